#!/bin/bash
# Collects the profile artefacts of a round (copy the outputs from gpurun_out/ into profiles/):
#   1. rocprofv3 --kernel-trace --stats of the default bench command  -> prof_kernel_stats.csv, prof_bench.log
#   2. PMC pass FETCH_SIZE, PMC pass WRITE_SIZE (separate passes, counters only) -> pmc_fetch.csv, pmc_write.csv (k_pencil rows)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench --output-format csv -- python $R/bench.py --timed-only > $R/gpurun_out/prof_bench.log 2>&1
grep '^{"metric"' $R/gpurun_out/prof_bench.log | tail -1 > $R/gpurun_out/prof_bench.json; cut -c1-700 $R/gpurun_out/prof_bench.json
cp $(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/prof_kernel_stats.csv
rm -rf $R/gpurun_out/prof
head -40 $R/gpurun_out/prof_kernel_stats.csv | cut -c1-130
for c in FETCH_SIZE WRITE_SIZE; do
  n=pmc_$(echo $c | tr A-Z a-z | cut -d_ -f1)
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/$n -o $n --output-format csv -- python $R/bench.py --timed-only > $R/gpurun_out/$n.log 2>&1
  f=$(find $R/gpurun_out/$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$R/gpurun_out/$n.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows: agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as o:
    o.write("kernel,counter,launches,mean_value,min,max\n")
    for (k, c), v in sorted(agg.items()):
        if "at::native" in k: continue
        o.write('"%s",%s,%d,%.1f,%.1f,%.1f\n' % (k, c, len(v), sum(v) / len(v), min(v), max(v)))
PY
  rm -rf $R/gpurun_out/$n
  grep -E "k_ribbon|k_pencil|k_fit|k_permute|k_encode|k_fast|k_hdec" $R/gpurun_out/$n.csv | cut -c1-150
done
