"""development: the pool test of tests/test_gpu_parity.py in a loop, with details of a failure"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle
import sz_amd as sz
from sz_amd.fields import m_field, s_field
cases = [(s_field(96, 128, 160), 1e-4), (m_field(96), 1e-4), (s_field(40, 70, 90, np.float64), 1e-6), (s_field(33, 65, 130), 1e-3), (m_field(64), 1e-4)]
refs, xs, metas = [], [], []
for d, eb in cases:
    ref, _ = oracle.compress(d, oracle.ABS, eb)
    refs.append(ref); xs.append(torch.from_numpy(d).cuda()); metas.append(ref[:4 + (28 if d.dtype == np.float32 else 36)])
print("stream sizes", [len(r) for r in refs])
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for lanes in (2, 3):
        pool = sz.HipPool(0, lanes)
        outs = [torch.empty(len(r) + (1 << 16), dtype=torch.uint8, device="cuda") for r in refs]
        for rounds in range(3):
            tks = [pool.submit(xs[i].data_ptr(), True, cases[i][0].shape, cases[i][0].dtype, cases[i][1], metas[i], None, outs[i].data_ptr(), outs[i].numel()) for i in range(len(cases))]
            for i, tk in enumerate(tks):
                try:
                    n, st = pool.wait(tk)
                    ok = n == len(refs[i]) and bytes(outs[i][:n].cpu().numpy()) == refs[i]
                    if not ok: print("MISMATCH rep", rep, "lanes", lanes, "round", rounds, "case", i, "n", n, "kernel", st.quant_kernel); bad += 1
                except Exception as e:
                    print("FAIL rep", rep, "lanes", lanes, "round", rounds, "case", i, "ticket", tk, str(e)[-60:]); bad += 1
        pool.close()
print("done, failures:", bad)
