#!/bin/bash
# development: 1-D tests, the 1-D timing, the benchmark with the other paths, in one gpurun call
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "1d" > gpurun_out/oned_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/oned_tests.log
grep -n "passed\|failed\|FAILED\|pytest rc\|Error" gpurun_out/oned_tests.log | tail -8
timeout 120 python tools/gpu_1d_time.py 4000000 > gpurun_out/oned_time.log 2>&1; cat gpurun_out/oned_time.log
timeout 200 python bench.py --other-paths > gpurun_out/oned_bench.log 2>&1; tail -2 gpurun_out/oned_bench.log
timeout 60 python __graft_entry__.py smoke 2>&1 | tail -2
