"""development: time the 1-D path (one wavefront walks the chain).  usage: python tools/gpu_1d_time.py [n]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import sz_amd
assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
configs = ((np.float32, "0"), (np.float64, "0"), (np.float32, "1"), (np.float32, "one-thread"))
if os.environ.get("ONLY_SEG"): configs = configs[:2]          # under the profiler: the segmented walk only
for dt, serial in configs:
    os.environ["SZ_HIP_1D_SERIAL"] = "1" if serial == "1" else "0"
    if serial == "one-thread": os.environ["SZ_HIP_1D_REACH_PCT"] = "100000000"     # no cut ever: one thread walks the array
    rng = np.random.default_rng(1)
    d = np.ascontiguousarray((np.cumsum(rng.standard_normal(n)) * 0.01 + np.sin(np.arange(n) * 0.003)).astype(dt))
    for rep in range(2):
        t0 = time.time(); got = sz_amd.SZ_compress_args(d, 0, 1e-3, 0.0); t1 = time.time()
        sc = sz_amd.SZ_hip_last_stats(); qc = sc.ms_quant
        back = sz_amd.SZ_decompress(got, d.shape, d.dtype); t2 = time.time()
        sd = sz_amd.SZ_hip_last_stats(); qd = sd.ms_quant
    t3 = time.time(); ref, _ = O.compress(d, 0, 1e-3, 0.0); t4 = time.time()
    print(f"1-D {np.dtype(dt).name} n={n} serial={serial} launches={sc.quant_kernel_launches}: ratio {d.nbytes / len(got):.2f}  compress {t1 - t0:.3f} s (chain kernel {qc:.1f} ms = {n / qc / 1e3:.1f} Melem/s)  "
          f"decompress {t2 - t1:.3f} s (chain kernel {qd:.1f} ms = {n / qd / 1e3:.1f} Melem/s)  oracle compress {t4 - t3:.3f} s  same={got == ref}", flush=True)
