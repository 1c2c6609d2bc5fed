#!/bin/bash
# development: a recorded case, the 1-D tests, the 1-D timing and a fuzz run with 1-D cases, in one gpurun call
mkdir -p gpurun_out
timeout 60 python tools/gpu_case.py > gpurun_out/oned_case.log 2>&1; cat gpurun_out/oned_case.log
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "1d or unsupported or sz14" > gpurun_out/oned_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/oned_tests.log
tail -4 gpurun_out/oned_tests.log
timeout 90 python tools/gpu_1d_time.py 4000000 > gpurun_out/oned_time.log 2>&1; cat gpurun_out/oned_time.log
timeout 120 python tools/gpu_fuzz.py 1500 78 > gpurun_out/oned_fuzz.log 2>&1; tail -4 gpurun_out/oned_fuzz.log
timeout 120 python tools/gpu_fuzz.py 1500 79 sz14 > gpurun_out/oned_fuzz14.log 2>&1; tail -4 gpurun_out/oned_fuzz14.log
