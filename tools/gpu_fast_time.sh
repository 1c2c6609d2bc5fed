#!/bin/bash
# development: phase times inside k_fast_stat (library built with -DSZG_DBG_TIME): cycles >> 6 summed over workgroups, per wavefront
cd $GRAFT_REPO_ROOT
SZ_AMD_LIB=$PWD/sz_amd/csrc/variants/libszhip_time.so timeout 200 python - <<PY 2>&1 | grep -v "Warn\|amdgpu.ids"
import numpy as np, torch, sz_amd
from sz_amd.fields import s_field
n = 512
x = torch.from_numpy(s_field(n, n, n)).cuda()
ctx = sz_amd.HipContext(0)
ob = torch.empty(x.numel() * 2 + (1 << 20), dtype=torch.uint8, device="cuda")
for it in range(2):
    _, sz, st = ctx.compress_fast(x.data_ptr(), True, (n, n, n), np.float32, 1e-4, 0, ob.data_ptr(), ob.numel())
h0 = ctx.debug_fetch(4, 65536 + 2048, np.uint32)[65536 + 1024:65536 + 1040].astype(np.int64)
_, sz, st = ctx.compress_fast(x.data_ptr(), True, (n, n, n), np.float32, 1e-4, 0, ob.data_ptr(), ob.numel())
h1 = ctx.debug_fetch(4, 65536 + 2048, np.uint32)[65536 + 1024:65536 + 1040].astype(np.int64)
d = ((h1 - h0) % (1 << 32)).reshape(4, 4) * 64 / 768
print("per workgroup, cycles: rows = wavefront 0..3; columns = wait-at-top-barrier, commit, search+load-issue, walk")
print(d.round(0))
print("quant ms", st.ms_quant, "sum per wave (cycles)", d.sum(axis=1))
hh = ctx.debug_fetch(4, 65536 + 2048, np.uint32)
print("workgroup 0: wall_clock64 ticks", hh[65536 + 1100], "clock64 ticks", hh[65536 + 1101])
hh = ctx.debug_fetch(4, 65536 + 8192, np.uint32).astype(np.int64)
t0 = hh[65536 + 2048:65536 + 2048 + 768]; t1 = hh[65536 + 4096:65536 + 4096 + 768]
base = t0.min()
print("workgroup start times (us after the first): percentiles 0/25/50/75/100", np.percentile((t0 - base) / 100.0, [0, 25, 50, 75, 100]).round(1))
print("workgroup end times: percentiles", np.percentile((t1 - base) / 100.0, [0, 25, 50, 75, 100]).round(1), " lifetime mean us", ((t1 - t0) / 100.0).mean().round(1))
print("started within 20 us:", int(((t0 - base) < 2000).sum()), "of 768")
for it in range(3):
    _, sz, st = ctx.compress_fast(x.data_ptr(), True, (n, n, n), np.float32, 1e-4, 0, ob.data_ptr(), ob.numel())
hh = ctx.debug_fetch(4, 65536 + 2048, np.uint32)
print("after 300 calls: quant ms", st.ms_quant, "entropy", st.ms_entropy, "wall ticks", hh[65536 + 1100], "clock64 ticks", hh[65536 + 1101])
PY
