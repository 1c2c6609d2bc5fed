#!/bin/bash
# kernel stats of a few plain SZ 2.1 compressions at 512^3 (ENV: extra environment, e.g. SZ_HIP_SLICES=1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
bash $R/tools/gpu_r4_trace_main.sh > /dev/null 2>&1    # writes /tmp/one_main.py
for V in ${VARIANTS:-"X=1"}; do
  rm -rf $O/tr
  env $(echo $V | tr ";" " ") timeout 200 rocprofv3 --kernel-trace --stats -d $O/tr -o t --output-format csv -- python /tmp/one_main.py > /tmp/log.txt 2>&1
  echo "== $V"
  python3 - <<PY
import csv, glob
for f in glob.glob("$O/tr/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["AverageNs"]) > 15000 and ("permute" in r["Name"] or not "$ONLY"): print("   %-50s avg %7.1f us min %7.1f  x%s" % (r["Name"][:50], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Calls"]))
PY
done
rm -rf $O/tr
