"""development: one case, GPU stream vs oracle stream, first difference and stage statistics."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import sz_amd
from sz_amd.fields import s_field
assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
d = s_field(128, 256, 256, np.float64)
ref, st = O.compress(d, O.REL, 0.0, 1e-3, want_stages=True)
got = sz_amd.SZ_compress_args(d, sz_amd.REL, 0.0, 1e-3)
gs = sz_amd.SZ_hip_last_stats()
print("oracle bytes", len(ref), "gpu bytes", len(got))
print("oracle: intervals", st["intervals"], "use_mean", st["use_mean"], "reg", st["reg_count"], "unpred", st["total_unpred"], "eb", st["eb"], "tree_bytes", st["tree_bytes"], "huff_bytes", st["huff_bytes"])
print("gpu   : intervals", gs.intervals, "use_mean", gs.use_mean, "reg", gs.n_reg_blocks, "unpred", gs.n_unpred)
n = min(len(ref), len(got))
a = np.frombuffer(ref[:n], np.uint8); b = np.frombuffer(got[:n], np.uint8)
diff = np.nonzero(a != b)[0]
print("first diffs at", diff[:10], "count", len(diff))
if len(diff):
    i = int(diff[0]); print("ref", ref[max(0, i - 8):i + 8].hex(), "\ngpu", got[max(0, i - 8):i + 8].hex())

# stage-level: codes in block order and natural order
ctx = sz_amd.HipContext(0)
meta = ref[:4 + 36]
out, nbytes, stats = ctx.compress(d.ctypes.data, False, d.shape, d.dtype, st["eb"], meta)
print("ctx stream identical:", out == ref, len(out))
blk = ctx.debug_fetch(3, d.size, np.uint16).astype(np.int32)
oc = st["codes"]
bad = np.nonzero(blk != oc)[0]
print("block-order code mismatches:", len(bad), "first", bad[:10])
if len(bad):
    i = int(bad[0]); print("gpu", blk[i - 4:i + 8], "oracle", oc[i - 4:i + 8])
nat = ctx.debug_fetch(2, d.size, np.uint16).astype(np.int32)
import sim_lib
nb = sim_lib.nat_to_blk(nat.astype(np.uint16), d.shape).astype(np.int32)
bad2 = np.nonzero(nb != oc)[0]
print("natural-order (re-permuted on the host) mismatches:", len(bad2), "first", bad2[:10])
