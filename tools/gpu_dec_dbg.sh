#!/bin/bash
# development: the inverse sweep's time (stats.ms_quant) under SZ_HIP_DBG bits (1: no result stores, 4: no loads -- wrong results on purpose)
cd $GRAFT_REPO_ROOT
for dbg in ${DBG_LIST:-0 1 4 5}; do
SZ_HIP_DBG=$dbg timeout 200 python - <<PY 2>&1 | grep -v "Warn\|amdgpu.ids"
import numpy as np, torch, sz_amd, ctypes
from sz_amd.fields import s_field
n = 512
x = torch.from_numpy(s_field(n, n, n)).cuda()
ctx = sz_amd.HipContext(0)
ob = torch.empty(x.numel() * 2 + (1 << 20), dtype=torch.uint8, device="cuda")
meta = sz_amd.make_meta(np.float32, err_mode=sz_amd.ABS, abs_bound=1e-4, vmin=0.0, vmax=0.0)
prm = sz_amd.szhip_params(100, 0.99, 65536, 0, 1)
import os
os.environ["SZ_HIP_DBG"] = "0"
out = ctypes.c_void_p(ob.data_ptr()); nn = ctypes.c_size_t(ob.numel()); st = sz_amd.szhip_stats()
rc = sz_amd.lib().szhip_compress(ctx._h, 0, x.data_ptr(), 1, n, n, n, 1e-4, ctypes.byref(prm), meta, len(meta), 2, ctypes.byref(out), ctypes.byref(nn), ctypes.byref(st))
os.environ["SZ_HIP_DBG"] = "$dbg"
dec = torch.empty_like(x)
q = []
for it in range(4):
    d = ctx.decompress(ob.data_ptr(), True, nn.value, 4 + 28 + 8, (n, n, n), np.float32, dec.data_ptr(), True)
    q.append(round(d.ms_quant, 3))
print("SZ_HIP_DBG=$dbg inverse sweep ms:", q, "kernel", d.quant_kernel)
PY
done
