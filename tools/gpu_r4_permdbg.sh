#!/bin/bash
# development: where k_permute<0>'s time goes -- an extra, early-stopping launch in front of the real one (SZ_HIP_PERM_DBG), durations from the kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
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
for D in ${DBGS:-8 4 5 6}; do
  rm -rf $O/tr
  env $EXTRA SZ_HIP_SLICES=1 SZ_HIP_PERM_DBG=$D timeout 60 rocprofv3 --kernel-trace -d $O/tr -o t --output-format csv -- python /tmp/one_main.py > /tmp/log.txt 2>&1
  tail -1 /tmp/log.txt
  python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_permute" in r["Kernel_Name"]: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
d = [(b - a) / 1e3 for a, b in rows]
print("DBG=$D  extra launch: %s   real launch: %s" % (["%.1f" % v for v in d[0::2]], ["%.1f" % v for v in d[1::2]]))
PY
done
rm -rf $O/tr
