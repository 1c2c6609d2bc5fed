"""round 6: per-call phase times of N compress calls of the 512^3 S-field (or m) under the environment given -- prints every call, then the medians."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sz_amd import api
from sz_amd.fields import m_field, s_field
edge = int(os.environ.get("EDGE", "512")); field = os.environ.get("FIELD", "s"); ncalls = int(os.environ.get("NCALLS", "24"))
dev = torch.device("cuda:0")
meta = api.make_meta(np.float32, api.ABS, 1e-4)
shape, dt, eb = (edge, edge, edge), np.float32, 1e-4
if field == "c4":      # BASELINE configs[3]: one rank's slab of the 1024^3 float64 S-field cut for 8 GPUs, REL 1e-3
    shape, dt = (132, 1024, 1024), np.float64
    host = s_field(132, 1024, 1024, np.float64, z0=0)
    lo, hi = float(host.min()), float(host.max())
    eb = 1e-3 * (hi - lo)
    meta = api.make_meta(np.float64, err_mode=api.REL, rel_ratio=1e-3, vmin=lo, vmax=hi)
    d = torch.from_numpy(host).to(dev)
else:
    d = torch.from_numpy(m_field(edge) if field == "m" else s_field(edge, edge, edge)).to(dev)
ctx = api.HipContext(0)
rows = []
for it in range(ncalls + 4):
    ptr, n, st = ctx.compress(d.data_ptr(), True, shape, dt, eb, meta, out_on_device=True)
    if it >= 4: rows.append((st.ms_total, st.ms_prequant, st.ms_quant, st.ms_entropy, st.ms_host))
a = np.array(rows)
print(os.environ.get("TAG", ""), "calls total/prequant/quant/entropy/host; per call total:", " ".join("%.2f" % x for x in a[:, 0]))
print(os.environ.get("TAG", ""), "median", " ".join("%.3f" % x for x in np.median(a, axis=0)), "max total %.3f" % a[:, 0].max(), "packing", st.packing, "size", n, "intervals", st.intervals, "reg", st.n_reg_blocks, "kernel", st.quant_kernel)
ctx.close()
