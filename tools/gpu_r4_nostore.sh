#!/bin/bash
# development: is k_omp_col bound by HBM traffic?  The same sweep without its code stores (SZ_HIP_OMP_DBG_NOSTORE=1: streams wrong, timing only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
bash $R/tools/gpu_r4_encdbg.sh > /dev/null 2>&1   # writes /tmp/one_cmp.py
for D in 0 1; do
  rm -rf $O/tr
  SZ_HIP_OMP_DBG_NOSTORE=$D timeout 200 rocprofv3 --kernel-trace --stats -d $O/tr -o t --output-format csv -- python /tmp/one_cmp.py > /tmp/log.txt 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob("$O/tr/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_omp_col" in r["Name"]: print("NOSTORE=$D k_omp_col avg %.1f us min %.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
rm -rf $O/tr
