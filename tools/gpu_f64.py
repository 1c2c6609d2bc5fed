"""development: throughput of the per-GPU unit of BASELINE configs[3] (128x1024x1024 float64 slab, REL 1e-3)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sz_amd
from sz_amd.fields import s_field
shape = (128, 1024, 1024)
d = s_field(*shape, np.float64, z0=3 * 128)
eb = 1e-3 * (2 * 1.4834466)
x = torch.from_numpy(d).cuda()
ctx = sz_amd.HipContext(0)
meta = sz_amd.make_meta(np.float64, err_mode=sz_amd.REL, rel_ratio=1e-3, vmin=-1.4834466, vmax=1.4834466)
for it in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    ptr, n, st = ctx.compress(x.data_ptr(), True, shape, np.float64, eb, meta, out_on_device=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("compress %.2f ms  %.1f GB/s  bytes %d  phases pre %.2f quant %.2f entropy %.2f host %.2f  reg %d/%d" % (dt * 1e3, d.nbytes / dt / 1e9, n, st.ms_prequant, st.ms_quant, st.ms_entropy, st.ms_host, st.n_reg_blocks, st.n_blocks))
dec = torch.empty_like(x)
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    ds = ctx.decompress(ptr, True, n, 4 + 36 + 8, shape, np.float64, dec.data_ptr(), True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("decompress %.2f ms  %.1f GB/s  quant %.2f" % (dt * 1e3, d.nbytes / dt / 1e9, ds.ms_quant))
print("max err / eb", float((dec - x).abs().max().item()) / eb)
