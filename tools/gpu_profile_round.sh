#!/bin/bash
# Collects the profile artefacts of a round (copy the outputs from gpurun_out/ into profiles/):
#   1. rocprofv3 --kernel-trace --stats of a command                                  -> <tag>_kernel_stats.csv, <tag>_bench.log / .json
#   2. PMC pass FETCH_SIZE, PMC pass WRITE_SIZE (separate passes, counters only)      -> <tag>_pmc_fetch.csv, <tag>_pmc_write.csv (per-launch means per kernel)
#      (counter collection runs one dispatch at a time: a sweep that waits for kernels of another stream -- the fed beam -- is taken in its unfed order there)
# usage: gpu_profile_round.sh [tag [command...]]      default: tag "prof", command `python bench.py --timed-only`
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-prof}; [ $# -gt 0 ] && shift
if [ $# -eq 0 ]; then set -- python $R/bench.py --timed-only; fi
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/${tag}_d
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_d -o bench --output-format csv -- "$@" > $R/gpurun_out/${tag}_bench.log 2>&1
grep '^{"metric"' $R/gpurun_out/${tag}_bench.log | tail -1 > $R/gpurun_out/${tag}_bench.json; cut -c1-400 $R/gpurun_out/${tag}_bench.json
cp $(find $R/gpurun_out/${tag}_d -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_kernel_stats.csv
rm -rf $R/gpurun_out/${tag}_d
head -24 $R/gpurun_out/${tag}_kernel_stats.csv | cut -c1-130
for c in FETCH_SIZE WRITE_SIZE; do
  n=${tag}_pmc_$(echo $c | tr A-Z a-z | cut -d_ -f1)
  SZ_HIP_BEAM_FEED=0 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/${n}_d -o $n --output-format csv -- "$@" > $R/gpurun_out/$n.log 2>&1
  f=$(find $R/gpurun_out/${n}_d -name "*counter_collection.csv" | head -1)
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
  rm -rf $R/gpurun_out/${n}_d
  grep -E "k_ribbon|k_beam|k_reg_points|k_pencil|k_fit|k_permute|k_encode|k_hdec" $R/gpurun_out/$n.csv | cut -c1-150
done
