#!/bin/bash
# development: 1-D tests, the 1-D timing and a fuzz run, in one gpurun call
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "1d or 1-D" > gpurun_out/oned_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/oned_tests.log
grep -n "passed\|failed\|FAILED\|pytest rc" gpurun_out/oned_tests.log | tail -8
timeout 120 python tools/gpu_1d_time.py 4000000 > gpurun_out/oned_time.log 2>&1; cat gpurun_out/oned_time.log
timeout 120 python tools/gpu_fuzz.py 2000 95 > gpurun_out/oned_fuzz.log 2>&1; tail -3 gpurun_out/oned_fuzz.log
