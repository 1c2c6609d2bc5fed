#!/bin/bash
# development: fast-mode phase times with every library variant under sz_amd/csrc/variants/ (results of the debug variants are wrong on purpose)
cd $GRAFT_REPO_ROOT
for so in sz_amd/csrc/libszhip.so $(ls sz_amd/csrc/variants/libszhip_*.so); do
  SZ_AMD_LIB=$PWD/$so timeout 200 python - <<PY
import numpy as np, torch, sz_amd
from sz_amd.fields import s_field
n = 512
x = torch.from_numpy(s_field(n, n, n)).cuda()
ctx = sz_amd.HipContext(0)
ob = torch.empty(x.numel() * 2 + (1 << 20), dtype=torch.uint8, device="cuda")
import time, os, sys
q = []
devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 2)
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    try:
        _, sz, st = ctx.compress_fast(x.data_ptr(), True, (n, n, n), np.float32, 1e-4, 0, ob.data_ptr(), ob.numel())
        r = (round(st.ms_quant, 3), round(st.ms_entropy, 3), sz)
    except Exception as e:
        r = ("fail", str(e)[-40:])
    torch.cuda.synchronize(); q.append((round((time.perf_counter() - t0) * 1e3, 3), r))
print("$so".split("/")[-1], q[-2:])
PY
done
