#!/bin/bash
# round 6: parity tests of the sweep and the packing passes, one call's device time line, then the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_beam_gpu.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|assert|rror" | grep -v szhip_decompress | tail -15 ) > gpurun_out/r6_check_tests.txt
cat gpurun_out/r6_check_tests.txt
bash tools/gpu_r6_trace.sh ${1:-s} chk > /dev/null 2>&1
grep field gpurun_out/r6_chk_log.txt | tail -8
head -75 gpurun_out/r6_chk_timeline.txt
( timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -3 ) > gpurun_out/r6_check_bench.txt
cut -c1-1500 gpurun_out/r6_check_bench.txt
