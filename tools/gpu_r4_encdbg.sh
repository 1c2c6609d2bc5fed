#!/bin/bash
# development: which part of k_omp_encode_box3 costs what (SZ_HIP_OMP_DBG bits; the streams are wrong, only the kernel's duration counts)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cat > /tmp/one_cmp.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
import sz_amd
from sz_amd.fields import s_field
n = 512
x = torch.from_numpy(s_field(n, n, n, np.float32)).to("cuda:0")
ctx = sz_amd.HipContext(0)
for it in range(3):
    try:
        ctx.compress_omp(x.data_ptr(), True, (n, n, n), np.float32, 1e-4, 4096, bytes(32), out_on_device=True)
    except Exception as e:
        print("err", str(e)[:80])
    torch.cuda.synchronize()
PY
for D in 0 1 2 3 4 7; do
  rm -rf $O/tr
  SZ_HIP_OMP_DBG=$D timeout 200 rocprofv3 --kernel-trace --stats -d $O/tr -o t --output-format csv -- python /tmp/one_cmp.py > /tmp/log.txt 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob("$O/tr/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "encode_box3" in r["Name"]: print("DBG=$D encode_box3 avg %.1f us min %.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
rm -rf $O/tr
