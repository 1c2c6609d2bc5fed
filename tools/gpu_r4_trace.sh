#!/bin/bash
# kernel timeline of ONE OpenMP-container compression + decompression (start/end timestamps of every kernel and copy): where the gaps are
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
T=${TAG:-r4tr}
cat > /tmp/one_omp.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
import sz_amd
from sz_amd.fields import s_field
n = 512
x = torch.from_numpy(s_field(n, n, n, np.float32)).to("cuda:0")
ctx = sz_amd.HipContext(0)
meta = bytes(32)
y = torch.empty_like(x)
for it in range(4):
    p, size, st = ctx.compress_omp(x.data_ptr(), True, (n, n, n), np.float32, 1e-4, 4096, meta, out_on_device=True)
    torch.cuda.synchronize()
    ctx.decompress_omp(p, True, size, len(meta), (n, n, n), np.float32, y.data_ptr(), True)
    torch.cuda.synchronize()
print("size", size, st.ms_total)
PY
rm -rf $O/tr
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/tr -o t --output-format csv -- python /tmp/one_omp.py > $O/${T}.log 2>&1
tail -2 $O/${T}.log
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:44]))
for f in glob.glob("$O/tr/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:30] + " " + r.get("Size", "")))
rows.sort()
# the last compress + decompress: from the last k_sample on
idx = [i for i, r in enumerate(rows) if "k_sample" in r[2]]
s0 = idx[-1]
t0 = rows[s0][0]; prev = t0
out = open("$O/${T}_timeline.txt", "w")
for a, b, n in rows[s0:]:
    line = "%9.1f us  +gap %7.1f  dur %8.1f  %s" % ((a - t0) / 1e3, (a - prev) / 1e3, (b - a) / 1e3, n)
    print(line); out.write(line + "\n"); prev = max(prev, b)
PY
rm -rf $O/tr
