#!/bin/bash
# development: in-kernel times of k_hdec_pass (library built with -DSZH_DBG_HDEC_TIME): prologue / decode cycles per wavefront
cd $GRAFT_REPO_ROOT
SZ_AMD_LIB=$PWD/sz_amd/csrc/variants/libszhip_htime.so timeout 200 python - <<PY 2>&1 | grep -v "Warn\|amdgpu.ids"
import numpy as np, torch, sz_amd
from sz_amd.fields import s_field
n = 512
x = torch.from_numpy(s_field(n, n, n)).cuda()
ctx = sz_amd.HipContext(0)
ob = torch.empty(x.numel() * 2 + (1 << 20), dtype=torch.uint8, device="cuda")
_, sz, st = ctx.compress_fast(x.data_ptr(), True, (n, n, n), np.float32, 1e-4, 0, ob.data_ptr(), ob.numel())
dec = torch.empty_like(x)
for it in range(3):
    ctx.decompress_fast(ob.data_ptr(), True, sz, (n, n, n), np.float32, dec.data_ptr(), True)
    sm = ctx.debug_fetch(12, 16, np.uint64)
    print("pass kernel(s): wavefronts %d, prologue cycles/wave %.0f, decode cycles/wave %.0f; look-ups total (cumulative) %d" % (sm[14], sm[12] / max(sm[14], 1), sm[13] / max(sm[14], 1), sm[15]))
PY
