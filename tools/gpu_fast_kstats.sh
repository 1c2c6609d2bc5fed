#!/bin/bash
# development: rocprofv3 per-kernel averages of the fast-mode compress call at 512^3 f32 -> gpurun_out/fast_kstats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof
cat > /tmp/fk.py <<PY
import numpy as np, torch, sz_amd
from sz_amd.fields import s_field
n = 512
x = torch.from_numpy(s_field(n, n, n)).cuda()
ctx = sz_amd.HipContext(0)
ob = torch.empty(x.numel() * 2 + (1 << 20), dtype=torch.uint8, device="cuda")
for it in range(8):
    _, sz, st = ctx.compress_fast(x.data_ptr(), True, (n, n, n), np.float32, 1e-4, 0, ob.data_ptr(), ob.numel())
dec = torch.empty_like(x)
for it in range(4):
    ctx.decompress_fast(ob.data_ptr(), True, sz, (n, n, n), np.float32, dec.data_ptr(), True)
print(sz, st.ms_quant, st.ms_entropy)
PY
PYTHONPATH=$R rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o fast --output-format csv -- python /tmp/fk.py > $R/gpurun_out/fast_kstats.log 2>&1
cp $(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/fast_kstats.csv
rm -rf $R/gpurun_out/prof
tail -1 $R/gpurun_out/fast_kstats.log
python3 - $R/gpurun_out/fast_kstats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-70s calls %4s avg %9.1f us  total %8.2f ms  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
