"""development: per-pencil timeline of the wavefront kernel (SZ_HIP_TRACE=1)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SZ_HIP_TRACE"] = "1"
import sz_amd
from sz_amd.fields import s_field
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = s_field(n, n, n)
ctx = sz_amd.HipContext(0)
meta = sz_amd.make_meta(np.float32, abs_bound=1e-4, vmin=float(d.min()), vmax=float(d.max()))
for it in range(2):
    b, sz, st = ctx.compress(d.ctypes.data, False, d.shape, np.float32, 1e-4, meta)
nI = nJ = (n + 7) // 8
raw = ctx.debug_fetch(9, nI * nJ * 8 + 256, np.uint64).astype(np.int64)
tr = raw[:nI * nJ * 8].reshape(nI, nJ, 8)
det = raw[nI * nJ * 8:].reshape(64, 4)
t0 = tr[..., 0].min()
us = lambda x: (x - t0) / 100.0
print("ms_quant", st.ms_quant, "size", sz)
np.save(os.path.join(ROOT, "gpurun_out", f"trace_{n}.npy"), tr)
for (I, J) in [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 2), (5, 5), (nI // 2, nJ // 2), (nI - 1, nJ - 1)]:
    r = tr[I, J]
    print(f"pencil ({I},{J}) xcc={r[6]} start={us(r[0]):.1f} gate={us(r[1]):.1f} first_trip={us(r[2]):.1f} end={us(r[3]):.1f} spins={r[4]} naps={r[5]}")
dur = (tr[..., 3] - tr[..., 1]) / 100.0
print("pencil active duration us: min %.1f median %.1f max %.1f" % (dur.min(), np.median(dur), dur.max()))
print("starts: min %.1f max %.1f ; ends: min %.1f max %.1f" % (us(tr[..., 0].min()), us(tr[..., 0].max()), us(tr[..., 3].min()), us(tr[..., 3].max())))
# lag between a pencil's gate-pass and its J-predecessor's gate-pass
lagJ = (tr[:, 1:, 1] - tr[:, :-1, 1]) / 100.0
lagI = (tr[1:, :, 1] - tr[:-1, :, 1]) / 100.0
print("gate lag vs J-pred us: median %.1f ; vs I-pred: median %.1f" % (np.median(lagJ), np.median(lagI)))
endlagJ = (tr[:, 1:, 3] - tr[:, :-1, 3]) / 100.0
print("end lag vs J-pred us: median %.1f" % np.median(endlagJ))
print("spins total", tr[..., 4].sum(), "median per pencil", np.median(tr[..., 4]))

print("detail of pencil (%d,%d): per trip  wait-for-loads+step0 | remaining steps | flush  (us)" % (nI // 2, nJ // 2))
for i in range(min(40, (n + 14 + 15) // 16)):
    r = det[i]
    if r[0] == 0: break
    print("  trip %2d: top->step0 %.2f  steps %.2f  flush %.2f   (top at %.1f)" % (i, (r[1]-r[0])/100, (r[2]-r[1])/100, (r[3]-r[2])/100, us(r[0])))

# tile view (float tiles are 3 x 3 pencils): first-trip-end and end times of the pencils of the tile holding pencil (32,33)
if n == 512:
    I0, J0 = 33, 33
    print("tile at pencil (%d,%d): first_trip end (us) per pencil [pi][pj]  |  pencil end" % (I0, J0))
    for pi in range(3): print("   ", " ".join("%7.1f" % us(tr[I0 + pi, J0 + pj, 2]) for pj in range(3)), "  |  ", " ".join("%7.1f" % us(tr[I0 + pi, J0 + pj, 3]) for pj in range(3)))
    print("first_trip of the first pencils of the next tiles: right (%d,%d) %.1f ; below (%d,%d) %.1f" % (I0, J0 + 3, us(tr[I0, J0 + 3, 2]), I0 + 3, J0, us(tr[I0 + 3, J0, 2])))

if n == 512:
    print("waits (us) per pencil of that tile: J-producer | I-producer | ring space")
    for pi in range(3): print("   ", "  ".join("%6.0f %6.0f %5.0f" % (tr[I0 + pi, J0 + pj, 4] / 100, tr[I0 + pi, J0 + pj, 5] / 100, tr[I0 + pi, J0 + pj, 7] / 100) for pj in range(3)))
    w = tr[..., 4] + tr[..., 5]; 
    print("all pencils: median wait on producers %.0f us, on ring space %.0f us; pencil (0,0) %.0f" % (np.median(w) / 100, np.median(tr[..., 7]) / 100, w[0, 0] / 100))
