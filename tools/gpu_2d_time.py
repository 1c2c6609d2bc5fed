"""development: timing of the 2-D path (device-resident input is not wired for the Python mirror here: host pointers, so the
phase times from szhip_stats are the numbers to read).  usage: python tools/gpu_2d_time.py [n] [dtype]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sz_amd
from sz_amd.fields import plane_field

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dt = np.float64 if len(sys.argv) > 2 and sys.argv[2] == "f64" else np.float32
assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
d = plane_field(n, n, dt)
for rep in range(3):
    t = time.time(); s = sz_amd.SZ_compress_args(d, sz_amd.ABS, 1e-4); tc = time.time() - t
    st = sz_amd.SZ_hip_last_stats()
    print(f"compress {n}x{n} {np.dtype(dt).name}: wall {tc*1e3:.1f} ms  prequant {st.ms_prequant:.2f} quant {st.ms_quant:.2f} entropy {st.ms_entropy:.2f} host {st.ms_host:.2f} "
          f"-> {d.nbytes/1e6/(st.ms_prequant+st.ms_quant+st.ms_entropy+st.ms_host):.2f} GB/s device phases; ratio {d.nbytes/len(s):.2f} reg {st.n_reg_blocks}/{st.n_blocks}")
    t = time.time(); o = sz_amd.SZ_decompress(s, d.shape, d.dtype); td = time.time() - t
    st = sz_amd.SZ_hip_last_stats()
    print(f"decompress: wall {td*1e3:.1f} ms  prequant {st.ms_prequant:.2f} quant {st.ms_quant:.2f} entropy {st.ms_entropy:.2f} host {st.ms_host:.2f}")
print("max err", float(np.abs(o.astype(np.float64) - d.astype(np.float64)).max()))
