#!/bin/bash
# development: when do the workgroups of k_fast_stat start and end (library built with -DSZG_DBG_RES)
cd $GRAFT_REPO_ROOT
for w in ${WGS_LIST:-512 768 1024}; do
SZ_HIP_FAST_STAT_WGS=$w SZ_AMD_LIB=$PWD/sz_amd/csrc/variants/libszhip_res.so timeout 200 python - <<PY 2>&1 | grep -v "Warn\|amdgpu.ids"
import numpy as np, torch, sz_amd
from sz_amd.fields import s_field
n = 512; w = $w
x = torch.from_numpy(s_field(n, n, n)).cuda()
ctx = sz_amd.HipContext(0)
ob = torch.empty(x.numel() * 2 + (1 << 20), dtype=torch.uint8, device="cuda")
for it in range(4):
    _, sz, st = ctx.compress_fast(x.data_ptr(), True, (n, n, n), np.float32, 1e-4, 0, ob.data_ptr(), ob.numel())
hh = ctx.debug_fetch(4, 65536 + 8192, np.uint32).astype(np.int64)
t0 = hh[65536 + 2048:65536 + 2048 + w]; t1 = hh[65536 + 4096:65536 + 4096 + w]
base = t0.min()
print(w, "workgroups: quant ms %.3f; started within 20 us: %d; start pct 50/75/100 (us):" % (st.ms_quant, int(((t0 - base) < 2000).sum())), np.percentile((t0 - base) / 100.0, [50, 75, 100]).round(1),
      "\n  end pct 0/50/100:", np.percentile((t1 - base) / 100.0, [0, 50, 100]).round(1), "mean life %.1f us" % ((t1 - t0) / 100.0).mean())
nt = hh[65536 + 6144:65536 + 6144 + w] & 0xffff; sm = hh[65536 + 6144:65536 + 6144 + w] >> 16
life = (t1 - t0) / 100.0
print("  tiles per workgroup min/mean/max", nt.min(), nt.mean().round(1), nt.max(), " us per tile pct 0/50/100", np.percentile(life / np.maximum(nt, 1), [0, 50, 100]).round(2))
order = np.argsort(t1)[::-1][:8]
print("  latest finishers: (block, smid hex, tiles, start, end, us/tile)", [(int(b), hex(int(sm[b])), int(nt[b]), round((t0[b] - base) / 100.0, 1), round((t1[b] - base) / 100.0, 1), round(life[b] / max(nt[b], 1), 2)) for b in order])
import collections
per = collections.defaultdict(list)
for b in range(w): per[int(sm[b])].append(round(float(life[b] / max(nt[b], 1)), 1))
ks = sorted(per, key=lambda k: -np.mean(per[k]))
print("  distinct smid values", len(per), " slowest CUs:", [(hex(k), per[k]) for k in ks[:6]], " fastest:", [(hex(k), per[k]) for k in ks[-4:]])
cnt = collections.Counter(len(v) for v in per.values()); print("  workgroups per smid:", dict(cnt))
hi = collections.defaultdict(list)
for k in per: hi[k >> 4].append(np.mean(per[k]))
print("  mean us/tile by smid>>4:", {hex(k): round(float(np.mean(v)), 2) for k, v in sorted(hi.items())})
for xcc in range(0):
    m = (np.arange(w) % 8) == xcc
    print("  blockIdx%%8 == %d: mean us/tile %.2f, mean end %.1f" % (xcc, (life[m] / np.maximum(nt[m], 1)).mean(), ((t1[m] - base) / 100.0).mean()))
PY
done
