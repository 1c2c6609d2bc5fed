#!/bin/bash
# First hardware run of the OpenMP container's HIP side (szh_omp.h) -- it was written after round 3's GPU minutes were spent.
#   gpurun --timeout 1500 -- 'bash tools/gpu_omp_first_run.sh'
# 1. its GPU tests (parity with the oracle, the recorded reference outputs, 256^3 / 512^3), each under its own timeout;
# 2. the differential runs of tools/omp_diff_fuzz.py through the built library;
# 3. bench.py --omp-boxes 4096 (the opt-in object of the bench line) and rocprofv3 --kernel-trace --stats of the same command.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_zz_omp_hip.py -m gpu -x -q > $O/omp_tests.log 2>&1; echo "tests exit $?" >> $O/omp_tests.log; tail -5 $O/omp_tests.log
SZ_FUZZ_GPU=1 timeout 600 python tools/omp_diff_fuzz.py 300 11 > $O/omp_fuzz.log 2>&1; tail -3 $O/omp_fuzz.log
timeout 600 python bench.py --omp-boxes 4096 --no-cpu-baseline --no-m-field --no-fast > $O/omp_bench.log 2>&1
grep '^{"metric"' $O/omp_bench.log | tail -1 > $O/omp_bench.json
python3 - <<PY
import json
d = json.load(open("$O/omp_bench.json"))
print(json.dumps(d.get("omp_container"), indent=1)[:1500])
PY
cd /tmp
rm -rf $O/omp_prof
timeout 600 rocprofv3 --kernel-trace --stats -d $O/omp_prof -o omp --output-format csv -- python $R/bench.py --omp-boxes 4096 --timed-only > $O/omp_prof.log 2>&1
cp $(find $O/omp_prof -name "*kernel_stats.csv" | head -1) $O/omp_kernel_stats.csv 2>/dev/null
rm -rf $O/omp_prof
grep -E "k_omp|k_hist|Name" $O/omp_kernel_stats.csv | cut -c1-160
