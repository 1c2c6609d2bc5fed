"""development: where does an M-field compress call spend its wall time?  (wall per call vs the library's own ms_total and phases)"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sz_amd
from sz_amd.fields import m_field, s_field
n = 512
dev = torch.device("cuda:0")
ctx = sz_amd.HipContext(0)
out_cap = n ** 3 * 2 + (1 << 20)
ob = torch.empty(out_cap, dtype=torch.uint8, device=dev)
for name, host in (("S", s_field(n, n, n)), ("M", m_field(n))):
    x = torch.from_numpy(host).to(dev)
    for k in range(8):
        meta = sz_amd.make_meta(np.float32, err_mode=sz_amd.ABS, abs_bound=1e-4, vmin=0.0, vmax=0.0)
        out = ctypes.c_void_p(ob.data_ptr()); nn = ctypes.c_size_t(out_cap); st = sz_amd.szhip_stats()
        p = sz_amd.szhip_params(100, 0.99, 65536, 0, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = sz_amd.lib().szhip_compress(ctx._h, 0, x.data_ptr(), 1, n, n, n, 1e-4, ctypes.byref(p), meta, len(meta), 2, ctypes.byref(out), ctypes.byref(nn), ctypes.byref(st))
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{name} call {k}: wall {1e3*(t1-t0):7.2f} ms (+sync {1e3*(t2-t1):5.2f})  lib total {st.ms_total:6.2f}  prequant {st.ms_prequant:5.2f} quant {st.ms_quant:5.2f} entropy {st.ms_entropy:5.2f} host {st.ms_host:5.2f}  rc {rc}", flush=True)
print("nproc", os.cpu_count(), "loadavg", os.getloadavg())
