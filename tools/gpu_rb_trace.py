"""development: per-wavefront timeline of the ribbon kernel (dev build, SZ_HIP_TRACE=1)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SZ_HIP_TRACE"] = "1"
os.environ.setdefault("SZ_AMD_LIB", os.path.join(ROOT, "sz_amd", "csrc", "libszhip_dev.so"))
import sz_amd
from sz_amd.fields import s_field
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W, R = int(os.environ.get("RB_W", 8)), int(os.environ.get("RB_R", 2))
if len(sys.argv) > 3:      # explicit shape: one tile, long sweep -- the step time of free-running wavefronts
    shp = tuple(int(x) for x in sys.argv[1:4])
    d = s_field(*shp)
    ctx = sz_amd.HipContext(0)
    meta = sz_amd.make_meta(np.float32, abs_bound=1e-4, vmin=float(d.min()), vmax=float(d.max()))
    for it in range(3):
        try: b, sz, st = ctx.compress(d.ctypes.data, False, d.shape, np.float32, 1e-4, meta)
        except Exception as e: print("compress failed (expected with SZ_HIP_DBG):", str(e)[:80])
    nTI, nTJ = (shp[0] + W * R - 1) // (W * R), (shp[1] + 63) // 64
    tr = ctx.debug_fetch(9, nTI * nTJ * (W + 4) * 8, np.uint64).astype(np.int64).reshape(nTI, nTJ, W + 4, 8)
    for w in range(W):
        r = tr[0, 0, w]
        print("dbg=%s shape %s tile (0,0) wave %d: %.3f us/step (dur %.1f us) waits us up %.1f left %.1f room %.1f" % (os.environ.get("SZ_HIP_DBG", "0"), shp, w, (r[2] - r[0]) / 100.0 / (shp[2] + 80), (r[2] - r[0]) / 100.0, r[3] / 2400.0, r[4] / 2400.0, r[5] / 2400.0))
    sys.exit(0)
d = s_field(n, n, n)
ctx = sz_amd.HipContext(0)
meta = sz_amd.make_meta(np.float32, abs_bound=1e-4, vmin=float(d.min()), vmax=float(d.max()))
os.environ["SZ_HIP_TICKET_MODE"] = os.environ.get("SZ_HIP_TICKET_MODE", "2")
failed = False
for it in range(3):
    try:
        b, sz, st = ctx.compress(d.ctypes.data, False, d.shape, np.float32, 1e-4, meta)
    except Exception as e:
        print("compress failed:", e); failed = True; break
if os.environ.get("RB_DEC") and not failed:          # the inverse sweep instead: decompress the stream, the trace is of that launch
    out = np.empty_like(d)
    import ctypes
    buf = ctypes.create_string_buffer(b, len(b))
    for it in range(2):
        st = ctx.decompress(ctypes.addressof(buf), False, len(b), 4 + 28 + 8, d.shape, np.float32, out.ctypes.data, False)
    print("INVERSE sweep: max err", float(np.abs(out - d).max()))
nTI, nTJ = (n + W * R - 1) // (W * R), (n + 63) // 64
raw = ctx.debug_fetch(9, nTI * nTJ * (W + 4) * 8 + nTI * 4 * 48, np.uint64).astype(np.int64)
tl = raw[nTI * nTJ * (W + 4) * 8:].reshape(nTI, 4, 48)
raw = raw[:nTI * nTJ * (W + 4) * 8]
tr = raw.reshape(nTI, nTJ, W + 4, 8)
t0 = tr[:, :, :W, 0].min()
us = lambda x: (x - t0) / 100.0
if failed:
    to = tr[:, :, :W, 7]
    print("timeouts (first step) per tile, wave 0:"); print(np.where(to[:, :, 0] >= 1000000, to[:, :, 0] - 1000000, -1))
    print("timeouts (first step) per tile, wave W-1:"); print(np.where(to[:, :, W - 1] >= 1000000, to[:, :, W - 1] - 1000000, -1))
    for (TI, TJ) in [(0, 0), (0, 1), (1, 0), (1, 1)]:
        print("tile", TI, TJ, "steps/timeouts per wave", [int(x) for x in tr[TI, TJ, :W, 7]], "FILL_U n", int(tr[TI, TJ, W + 2, 7]), "rounds", int(tr[TI, TJ, W + 2, 3]))
    NT = (n + 62 + R + (W - 1) * (R - 1) + 15) // 16 * 16
    rf = ctx.debug_fetch(11, nTI * nTJ * NT * W * R, np.uint64).reshape(nTI, nTJ, NT, W * R)
    tags = (rf >> np.uint64(32)).astype(np.int64)
    vals, cnts = np.unique(tags[0, 0], return_counts=True)
    order = np.argsort(-cnts)[:8]
    print("most common tags in the right-face rows of tile (0,0):", [(int(vals[i]), int(cnts[i])) for i in order])
    ep = int(vals[order[0]]) if vals[order[0]] != 0 else int(vals[order[1]])
    t00 = tags[0, 0]
    print("rows x steps validity map of tile (0,0), steps 0..159 (1 = tag == epoch):")
    for q in range(W * R): print("  row %2d:" % q, "".join("1" if t00[t, q] == ep else "." for t in range(160)))
    print("epoch", ep, "right-face granules of tile (0,0): first step without the tag, per row:", [int(np.argmax(tags[0, 0, :, q] != ep)) if (tags[0, 0, :, q] != ep).any() else -1 for q in range(W * R)])
    print("  count valid per row:", [int((tags[0, 0, :, q] == ep).sum()) for q in range(W * R)])
    miss = np.where(tags[0, 0, :, W * R - 1] != ep)[0]; print("  row 15 missing steps:", miss[:40], len(miss))
    miss = np.where(tags[0, 0, :, 3] != ep)[0]; print("  row 3 missing steps:", miss[:40], len(miss))
    df = ctx.debug_fetch(10, nTI * nTJ * NT * 64, np.uint64).reshape(nTI, nTJ, NT, 64)
    tg = (df >> np.uint64(32)).astype(np.int64)
    print("down-face of tile (0,0): steps with all 64 tags:", int((tg[0, 0] == ep).all(axis=1).sum()), "of", NT, "; of tile (0,1):", int((tg[0, 1] == ep).all(axis=1).sum()))
    sys.exit(0)
print("ms_quant", st.ms_quant, "size", sz, "tiles", nTI, nTJ)
cyc = 2400.0
print("kernel span us: first start %.1f last end %.1f" % (us(tr[:, :, :W, 0].min()), us(tr[:, :, :W, 2].max())))
print("start spread of all compute wavefronts us: %.1f" % us(tr[:, :, :W, 0].max()))
for (TI, TJ) in [(0, 0), (0, 1), (1, 0), (1, 1), (0, nTJ - 1), (nTI // 2, nTJ // 2), (nTI - 1, 0), (nTI - 1, nTJ - 1)]:
    print("tile (%d,%d) xcc %d" % (TI, TJ, tr[TI, TJ, 0, 6]))
    for w in (0, 1, W - 1):
        r = tr[TI, TJ, w]
        print("   wave %d: start %.1f trip0 end %.1f end %.1f (dur %.1f us, %.3f us/step)  waits us: up %.1f left %.1f room %.1f store %.1f" %
              (w, us(r[0]), us(r[1]), us(r[2]), (r[2] - r[0]) / 100.0, (r[2] - r[0]) / 100.0 / max(1, r[7]), r[3] / cyc, r[4] / cyc, (r[5] & 0xffffffff) / cyc, (r[5] >> 32) / cyc))
    r = tr[TI, TJ, W]; print("   DRAIN  : end %.1f rounds %d empty %d steps %d" % (us(r[2]), r[3], r[4], r[7]))
    r = tr[TI, TJ, W + 3]; print("   STORE  : start %.1f end %.1f busy %.1f us, scans %d, blocks %d" % (us(r[0]), us(r[2]), r[3] / cyc, r[4], r[5]))
    r = tr[TI, TJ, W + 2]; print("   FILL_U : end %.1f rounds %d empty %d noroom %d steps %d" % (us(r[2]), r[3], r[4], r[5], r[7]))
e0 = tr[:, :, 0, 2]; eL = tr[:, :, W - 1, 2]
print("end of wave 0 per tile row (col 0):", " ".join("%.0f" % us(x) for x in e0[:, 0]))
print("end of wave 0 per tile col (row 0):", " ".join("%.0f" % us(x) for x in e0[0, :]))
print("end lag between tile rows (wave 0, col 0) us: median %.1f" % np.median(np.diff(e0[:, 0]) / 100.0))
print("end lag between tile cols (wave 0, row 0) us: median %.1f" % np.median(np.diff(e0[0, :]) / 100.0))
print("in-tile end lag wave w -> w+1 us: median %.2f" % np.median((tr[:, :, 1:W, 2] - tr[:, :, :W - 1, 2]) / 100.0))
dur = (tr[:, :, :W, 2] - tr[:, :, :W, 0]) / 100.0
print("compute wavefront duration us: min %.1f median %.1f max %.1f" % (dur.min(), np.median(dur), dur.max()))
wt = tr[:, :, :W, 3:6].copy(); wt[:, :, :, 2] &= 0xffffffff; wt = wt / cyc
print("median waits us: up %.1f left %.1f room %.1f" % tuple(np.median(wt.reshape(-1, 3), axis=0)))
np.save(os.path.join(ROOT, "gpurun_out", "rb_trace_%d.npy" % n), tr)

print("timeline, column 0 (us since kernel start): per 16 steps: wave0 | last wave | DRAIN | next tile: FILL_U, wave0")
for TI in (0, 1, 2, 10):
    print(" tile row", TI)
    for i in range(0, 37, 3):
        nxt = tl[TI + 1] if TI + 1 < nTI else None
        print("   steps<%4d: w0 %7.1f  w7 %7.1f  drain %7.1f | next fill_u %7.1f  next w0 %7.1f" % (16 * (i + 1), us(tl[TI, 0, i]), us(tl[TI, 1, i]), us(tl[TI, 2, i]), us(nxt[3, i]) if nxt is not None else 0, us(nxt[0, i]) if nxt is not None else 0))
