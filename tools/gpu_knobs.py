"""development: k_pencil time at 512^3 (compress and decompress) under combinations of the launch knobs (environment, read per call)."""
import itertools, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sz_amd
from sz_amd.fields import s_field
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
knobs = {}
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    knobs[k] = v.split(",")
d = s_field(n, n, n)
assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
ref = None
keys = list(knobs)
for combo in itertools.product(*[knobs[k] for k in keys]):
    for k, v in zip(keys, combo):
        os.environ[k] = v
    q, dq, tot = [], [], []
    for rep in range(6):
        s = sz_amd.SZ_compress_args(d, sz_amd.ABS, 1e-4); st = sz_amd.SZ_hip_last_stats(); q.append(st.ms_quant)
        if ref is None: ref = s
        assert s == ref, "stream changed"
        o = sz_amd.SZ_decompress(s, d.shape, d.dtype); st = sz_amd.SZ_hip_last_stats(); dq.append(st.ms_quant)
    print(" ".join(f"{k}={v}" for k, v in zip(keys, combo)), f": k_pencil compress min {min(q[1:]):.3f} med {np.median(q[1:]):.3f} ms | decompress min {min(dq[1:]):.3f} med {np.median(dq[1:]):.3f} ms", flush=True)
