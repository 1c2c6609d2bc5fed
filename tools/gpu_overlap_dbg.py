"""development: the context-reuse sequence of tests/test_gpu_parity.py with the library's timing points"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sz_amd
import oracle_lib as O
from sz_amd.fields import m_field, s_field
ctx = sz_amd.HipContext(0)
for d, eb in ((s_field(64, 64, 64), 1e-4), (m_field(40, np.float64), 1e-5), (s_field(24, 40, 56), 1e-3), (m_field(40, np.float64), 1e-5)):
    ref, st = O.compress(d, O.ABS, eb, want_stages=True)
    meta = ref[:4 + (28 if d.dtype == np.float32 else 36)]
    x = torch.from_numpy(d).cuda()
    print("case", d.shape, d.dtype, "reg blocks", st["reg_count"], "of", st["num_blocks"], flush=True)
    try:
        ptr, n, stats = ctx.compress(x.data_ptr(), True, d.shape, d.dtype, eb, meta, out_on_device=True)
        print("  ok", n == len(ref), flush=True)
        out = torch.empty_like(x)
        ctx.decompress(ptr, True, n, len(meta) + 8, d.shape, d.dtype, out.data_ptr(), True)
        ref_dec = torch.from_numpy(O.decompress(ref, d.shape, d.dtype)).cuda()
        print("  decoded equal", bool(torch.equal(out.view(torch.int32 if d.dtype == np.float32 else torch.int64), ref_dec.view(torch.int32 if d.dtype == np.float32 else torch.int64))), flush=True)
    except Exception as e:
        print("  FAILED", e, flush=True)
