#!/bin/bash
# round 4: the OpenMP container's GPU tests + bench object + kernel stats  (TAG=name for the output files)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${TAG:-r4b}
timeout 400 python -m pytest tests/test_zz_omp_hip.py -m gpu -x -q > $O/${T}_omp_tests.log 2>&1; echo "tests exit $?" >> $O/${T}_omp_tests.log; tail -4 $O/${T}_omp_tests.log
timeout 400 python bench.py --omp-boxes 4096 --no-cpu-baseline --no-m-field --no-fast > $O/${T}_omp_bench.log 2>&1
grep '^{"metric"' $O/${T}_omp_bench.log | tail -1 > $O/${T}_omp_bench.json
python3 - <<PY
import json
d = json.load(open("$O/${T}_omp_bench.json"))
o = d.get("omp_container")
print(json.dumps({k: o[k] for k in ("GB/s", "ms", "decompress_GBps", "out_bytes", "max_abs_err", "phase_ms")}, indent=1))
print(o["roofline"])
PY
cd /tmp
rm -rf $O/omp_prof
timeout 400 rocprofv3 --kernel-trace --stats -d $O/omp_prof -o omp --output-format csv -- python $R/bench.py --omp-boxes 4096 --no-cpu-baseline --no-m-field --no-fast > $O/${T}_omp_prof.log 2>&1
cp $(find $O/omp_prof -name "*kernel_stats.csv" | head -1) $O/${T}_omp_kernel_stats.csv 2>/dev/null
rm -rf $O/omp_prof
grep -E "k_omp|k_hist|k_sample|k_scan" $O/${T}_omp_kernel_stats.csv | awk -F'","' '{printf "%-60.60s calls %s avg_ns %s\n", $1, $2, $4}'
