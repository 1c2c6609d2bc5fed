#!/bin/bash
# kernel timeline of ONE SZ 2.1-path compression at 512^3 (start/end timestamps of every kernel and copy): what is on the critical path
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
T=${TAG:-r4tm}
cat > /tmp/one_main.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
import sz_amd
from sz_amd.fields import s_field
n = 512
x = torch.from_numpy(s_field(n, n, n, np.float32)).to("cuda:0")
ctx = sz_amd.HipContext(0)
meta = bytes(32)
for it in range(4):
    p, size, st = ctx.compress(x.data_ptr(), True, (n, n, n), np.float32, 1e-4, meta, out_on_device=True)
    torch.cuda.synchronize()
print("size", size, st.ms_total)
PY
rm -rf $O/tr
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/tr -o t --output-format csv -- python /tmp/one_main.py > $O/${T}.log 2>&1
tail -2 $O/${T}.log
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:44], r.get("Queue_Id", "")))
for f in glob.glob("$O/tr/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:30] + " " + r.get("Size", ""), ""))
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_fit_select" in r[2] or "k_sample" in r[2]]
# the last call: from the last k_sample / k_fit_select pair on
last = idx[-1]
while last - 1 in idx or (last > 0 and rows[last][0] - rows[last - 1][0] < 500000 and last - 1 >= idx[-2] - 3): last -= 1
t0 = rows[last][0]
out = open("$O/${T}_timeline.txt", "w")
for a, b, n, q in rows[last:]:
    line = "%9.1f .. %9.1f us  dur %8.1f  q%s  %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, q, n)
    print(line); out.write(line + "\n")
PY
rm -rf $O/tr
