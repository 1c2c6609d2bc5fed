#!/bin/bash
# development: the GPU suite, the 1-D timing and fuzz runs with 1-D cases, in one gpurun call
mkdir -p gpurun_out
timeout 60 python tools/gpu_case.py > gpurun_out/oned_case.log 2>&1; cat gpurun_out/oned_case.log
timeout 420 python -m pytest tests -m gpu -x -q > gpurun_out/oned_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/oned_tests.log
grep -n "passed\|failed\|FAILED\|pytest rc" gpurun_out/oned_tests.log | tail -5
timeout 90 python tools/gpu_1d_time.py 4000000 > gpurun_out/oned_time.log 2>&1; cat gpurun_out/oned_time.log
timeout 120 python tools/gpu_fuzz.py 2000 81 > gpurun_out/oned_fuzz.log 2>&1; tail -4 gpurun_out/oned_fuzz.log
timeout 120 python tools/gpu_fuzz.py 2000 82 sz14 > gpurun_out/oned_fuzz14.log 2>&1; tail -4 gpurun_out/oned_fuzz14.log
