#!/bin/bash
# kernel stats of the OpenMP container on the float64 slab (32768 boxes of 4 x 32 x 32)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
bash $R/tools/gpu_r4_f64.sh > /dev/null 2>&1     # writes /tmp/f64.py
rm -rf $O/f64prof
timeout 300 rocprofv3 --kernel-trace --stats -d $O/f64prof -o p --output-format csv -- python /tmp/f64.py 32768 > $O/r4_f64_prof.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/f64prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["AverageNs"]) > 8000: print("%-56.56s calls %4s avg %9.1f us" % (r["Name"], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $O/f64prof
