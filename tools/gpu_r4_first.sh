#!/bin/bash
# round 4, first call: baseline timing of the OpenMP container as round 3 left it + kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_zz_omp_hip.py -m gpu -x -q > $O/r4a_omp_tests.log 2>&1; echo "tests exit $?" >> $O/r4a_omp_tests.log; tail -3 $O/r4a_omp_tests.log
timeout 400 python bench.py --omp-boxes 4096 --no-cpu-baseline --no-m-field --no-fast > $O/r4a_omp_bench.log 2>&1
grep '^{"metric"' $O/r4a_omp_bench.log | tail -1 > $O/r4a_omp_bench.json
python3 - <<PY
import json
d = json.load(open("$O/r4a_omp_bench.json"))
print(json.dumps(d.get("omp_container"), indent=1)[:2500])
print({k: d[k] for k in ("value", "ms_per_step")})
PY
cd /tmp
rm -rf $O/omp_prof
timeout 400 rocprofv3 --kernel-trace --stats -d $O/omp_prof -o omp --output-format csv -- python $R/bench.py --omp-boxes 4096 --no-cpu-baseline --no-m-field --no-fast > $O/r4a_omp_prof.log 2>&1
cp $(find $O/omp_prof -name "*kernel_stats.csv" | head -1) $O/r4a_omp_kernel_stats.csv 2>/dev/null
rm -rf $O/omp_prof
cut -c1-170 $O/r4a_omp_kernel_stats.csv | head -40
