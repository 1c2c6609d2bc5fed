#!/bin/bash
# development: repeat the fast-mode decompression and report where the decoded array differs between calls
cd $GRAFT_REPO_ROOT
for so in sz_amd/csrc/libszhip.so $(ls sz_amd/csrc/variants/libszhip_*.so 2>/dev/null); do
echo "== $so"
SZ_AMD_LIB=$PWD/$so timeout 300 python - <<PY 2>&1 | grep -v "Warn\|amdgpu.ids" | cut -c1-250
import numpy as np, torch, sz_amd
from sz_amd.fields import s_field
n = 512
x = torch.from_numpy(s_field(n, n, n)).cuda()
ctx = sz_amd.HipContext(0)
ob = torch.empty(x.numel() * 2 + (1 << 20), dtype=torch.uint8, device="cuda")
_, sz, st = ctx.compress_fast(x.data_ptr(), True, (n, n, n), np.float32, 1e-4, 0, ob.data_ptr(), ob.numel())
dec = torch.empty_like(x)
ref = torch.from_numpy(ctx.debug_fetch(2, n * n * n, np.uint16).astype(np.int32))      # the encoder's code array
for it in range(1):
    try:
        ctx.decompress_fast(ob.data_ptr(), True, sz, (n, n, n), np.float32, dec.data_ptr(), True)
        err = float((dec - x).abs().max().item())
        msg = "ok max err %.3g" % err
    except Exception as e:
        msg = "FAIL " + str(e)[-60:]
    codes = torch.from_numpy(ctx.debug_fetch(2, n * n * n, np.uint16).astype(np.int32))
    if ref is None and msg.startswith("ok"): ref = codes.clone()
    if ref is not None:
        d = (codes != ref).nonzero().flatten()
        print(it, msg, "codes differing from the encoder's:", d.numel(), d[:8].tolist(), d[-3:].tolist() if d.numel() else "")
        if d.numel():
            i0 = int(d[0]); print("   at", i0, "decoded", codes[i0:i0 + 12].tolist(), "encoder", ref[i0:i0 + 12].tolist())
            for i in d[:12].tolist(): print("   ", i, i % 8, "decoded", codes[i - 2:i + 3].tolist(), "encoder", ref[i - 2:i + 3].tolist())
            print("   histogram of index % 8:", np.bincount(d.numpy() % 8, minlength=8).tolist())
            dd = d.numpy(); gaps = np.flatnonzero(np.diff(dd) > 1); print("   runs of differences:", len(gaps) + 1, "first run starts", dd[np.r_[0, gaps + 1]][:10].tolist())
    else: print(it, msg)
PY
done
