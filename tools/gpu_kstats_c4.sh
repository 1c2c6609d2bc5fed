#!/bin/bash
# development: rocprofv3 per-kernel averages of the configs[3] bench command -> gpurun_out/kstats_c4.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench --output-format csv -- python $R/bench.py --config c4 --steps 5 --warmup 2 > $R/gpurun_out/kstats_c4_bench.log 2>&1
cp $(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/kstats_c4.csv
rm -rf $R/gpurun_out/prof
python3 - $R/gpurun_out/kstats_c4.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-70s calls %4s avg %9.1f us  total %8.2f ms  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
