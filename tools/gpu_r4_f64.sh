#!/bin/bash
# round 4: the OpenMP container on a float64 slab of BASELINE configs[3] (128 x 1024 x 1024, REL 1e-3 of the slab's own range; 4096 boxes of 32^3)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cat > /tmp/f64.py <<PY
import sys, time, json, numpy as np, torch
sys.path.insert(0, "$R")
import sz_amd
from sz_amd.fields import s_field
host = s_field(128, 1024, 1024, np.float64)
x = torch.from_numpy(host).to("cuda:0")
eb = 1e-3 * float(host.max() - host.min())
ctx = sz_amd.HipContext(0)
meta = bytes(32)
NBOX = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
y = torch.empty_like(x)
res = {}
for it in range(3):
    p, size, st = ctx.compress_omp(x.data_ptr(), True, host.shape, np.float64, eb, NBOX, meta, out_on_device=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(5):
    p, size, st = ctx.compress_omp(x.data_ptr(), True, host.shape, np.float64, eb, NBOX, meta, out_on_device=True)
torch.cuda.synchronize(); tc = (time.perf_counter() - t0) / 5
ctx.decompress_omp(p, True, size, len(meta), host.shape, np.float64, y.data_ptr(), True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(5):
    dst = ctx.decompress_omp(p, True, size, len(meta), host.shape, np.float64, y.data_ptr(), True)
torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 5
err = float((y - x).abs().max().item())
print(json.dumps({"what": "OpenMP container, float64 slab 128x1024x1024 (1 GiB), eb = 1e-3 x range, " + str(NBOX) + " boxes", "compress_GBps": round(host.nbytes / tc / 1e9, 2), "compress_ms": round(tc * 1e3, 3),
                  "decompress_GBps": round(host.nbytes / td / 1e9, 2), "out_bytes": int(size), "ratio": round(host.nbytes / size, 3), "max_abs_err": err, "eb": eb,
                  "phase_ms": {"prequant": round(st.ms_prequant, 3), "quant": round(st.ms_quant, 3), "entropy": round(st.ms_entropy, 3), "dec_entropy": round(dst.ms_entropy, 3), "dec_quant": round(dst.ms_quant, 3)},
                  "verbatim": int(st.n_unpred), "intervals": int(st.intervals)}))
assert err <= eb
PY
for NB in 4096 32768; do timeout 300 python /tmp/f64.py $NB 2>&1 | tail -1 | tee -a $O/r4_f64_omp.json; done
