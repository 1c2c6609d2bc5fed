"""development (dev build): timeline of ONE cross-tile hand-off, column by column: when the producing pencil finished the step that made
column c of face row 0, when the STORE wavefront had forwarded it, when the FILL wavefront of the next tile had delivered it, and
when the consuming pencil finished the step that used it.  usage: SZ_AMD_LIB=.../libszhip_dev.so python tools/gpu_handoff.py [n] [TI] [TJ]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
TI = int(sys.argv[2]) if len(sys.argv) > 2 else 0
TJ = int(sys.argv[3]) if len(sys.argv) > 3 else 1
os.environ["SZ_HIP_TRACE"] = "1"
os.environ["SZ_HIP_TRACE_TILE"] = str((TI << 16) | TJ)
import sz_amd
from sz_amd.fields import s_field
d = s_field(n, n, n)
ctx = sz_amd.HipContext(0)
meta = sz_amd.make_meta(np.float32, abs_bound=1e-4, vmin=float(d.min()), vmax=float(d.max()))
for it in range(2):
    b, sz, st = ctx.compress(d.ctypes.data, False, d.shape, np.float32, 1e-4, meta)
nI = nJ = (n + 7) // 8
LOG = 2048
raw = ctx.debug_fetch(9, nI * nJ * 8 + 256 + 8 * LOG, np.uint64).astype(np.int64)
logs = raw[nI * nJ * 8 + 256:].reshape(4, LOG, 2)
P, S, F, C = logs
t0 = P[0, 0]
us = lambda x: (x - t0) / 100.0
print(f"ms_quant {st.ms_quant:.3f}; hand-off into tile ({TI},{TJ}) from its left neighbour; FILL mode {os.environ.get('SZ_HIP_FILL', 'default')}")
ns, nf = int((S[:, 0] > 0).sum()), int((F[:, 0] > 0).sum())
print(f"STORE rounds logged {ns}, FILL rounds logged {nf}")
if ns > 2:
    dS = np.diff(S[:ns, 0]) / 100.0
    print("STORE round period us: median %.2f p90 %.2f ; columns per round median %.1f" % (np.median(dS), np.percentile(dS, 90), np.median(np.diff(S[:ns, 1])[np.diff(S[:ns, 1]) > 0]) if (np.diff(S[:ns, 1]) > 0).any() else 0))
if nf > 2:
    act = F[:nf]
    dF = np.diff(act[:, 0]) / 100.0
    moved = np.diff(act[:, 1])
    print("FILL round period us: all median %.2f ; rounds that delivered: n %d median period %.2f, columns/round median %.1f" % (np.median(dF), (moved > 0).sum(), np.median(dF[moved > 0]) if (moved > 0).any() else 0, np.median(moved[moved > 0]) if (moved > 0).any() else 0))
print("col |  produced  stored(+)  filled(+)  consumed(+)   [us; + = after the previous stage]")
for c in list(range(0, 64, 4)) + list(range(64, n, 32)):
    tp = P[c + 7, 0] if c + 7 < LOG else 0
    si = np.argmax(S[:ns, 1] > c) if ns and (S[:ns, 1] > c).any() else -1
    fi = np.argmax(F[:nf, 1] > c) if nf and (F[:nf, 1] > c).any() else -1
    tc = C[c, 0] if c < LOG else 0
    if tp == 0 or si < 0 or fi < 0 or tc == 0: continue
    ts, tf = S[si, 0], F[fi, 0]
    print(f"{c:4d} | {us(tp):9.2f} {(ts - tp) / 100:9.2f} {(tf - ts) / 100:9.2f} {(tc - tf) / 100:9.2f}   total {(tc - tp) / 100:6.2f}")
pst = np.diff(P[: n + 14, 0]) / 100.0
cst = np.diff(C[: n + 14, 0]) / 100.0
print("producer step time us: median %.3f mean %.3f ; consumer step: median %.3f mean %.3f" % (np.median(pst), pst.mean(), np.median(cst), cst.mean()))
np.save(os.path.join(ROOT, "gpurun_out", f"handoff_{n}_{TI}_{TJ}.npy"), logs)
