#!/bin/bash
# development: GPU checks of the MSST19 path (tests, fuzz, timings); writes under gpurun_out/
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_msst19.py tests/test_ref_recorded.py -m gpu -q -x -k "msst or pw_rel or decodes_reference" > gpurun_out/msst_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/msst_tests.log
tail -5 gpurun_out/msst_tests.log
timeout 600 python tools/gpu_msst_time.py 256 > gpurun_out/msst_time.log 2>&1
cat gpurun_out/msst_time.log
