"""development: timings of the table-driven PW_REL form (MSST19) through the SZ_* API (host pointers, PCIe included) with the library's
own stage times.  usage: python tools/gpu_msst_time.py [edge]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sz_amd
from sz_amd.fields import s_field

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(1)
cases = [("3-D f32", (np.abs(s_field(edge, edge, edge, np.float64)) + 0.05).astype(np.float32)),
         ("3-D f64", (np.abs(s_field(edge // 2, edge, edge, np.float64)) + 0.05)),
         ("2-D f32", (np.abs(s_field(1, 4096, 4096, np.float64))[0] + 0.05).astype(np.float32)),
         ("1-D f32", np.exp(np.cumsum(rng.standard_normal(1 << 20)) * 1e-3).astype(np.float32))]
assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
for form in (1, 2, 0):                     # 1: MSST19 on the wavefront kernel, 2: MSST19 plane sweep, 0: log-domain form
    os.environ["SZ_HIP_MSST_SWEEP"] = "1" if form == 2 else "0"
    sz_amd.conf_params().accelerate_pw_rel_compression = 1 if form else 0
    for name, d in cases:
        d = np.ascontiguousarray(d)
        best = None
        for rep in range(3):
            t0 = time.perf_counter(); s = sz_amd.SZ_compress_args(d, sz_amd.PW_REL, 0.0, 0.0, 1e-3); t1 = time.perf_counter()
            st = sz_amd.SZ_hip_last_stats()
            cq, ce, ch = st.ms_quant, st.ms_entropy, st.ms_host
            t2 = time.perf_counter(); b = sz_amd.SZ_decompress(s, d.shape, d.dtype); t3 = time.perf_counter()
            sd = sz_amd.SZ_hip_last_stats()
            rec = (t1 - t0, t3 - t2, cq, ce, ch, sd.ms_quant)
            best = rec if best is None else tuple(min(a, b) for a, b in zip(best, rec))      # the least of three, metric by metric
        x = d.astype(np.float64); y = b.astype(np.float64)
        err = float((np.abs(y - x) / np.abs(x)).max())
        print(f"{('log   ', 'MSST19', 'MSST19-sweep')[form]} {name} {d.shape}: compress {best[0]*1e3:8.1f} ms ({d.nbytes/best[0]/1e9:6.2f} GB/s; quantise {best[2]:.1f}, entropy {best[3]:.1f}, host {best[4]:.1f}) "
              f"decompress {best[1]*1e3:8.1f} ms ({d.nbytes/best[1]/1e9:6.2f} GB/s; reconstruct {best[5]:.1f})  ratio {d.nbytes/len(s):.2f}  max rel err {err:.3e}", flush=True)
sz_amd.SZ_Finalize()
