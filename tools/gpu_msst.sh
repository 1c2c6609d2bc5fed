#!/bin/bash
# development: GPU checks of the MSST19 path (tests, fuzz, timings); writes under gpurun_out/
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_msst19.py tests/test_ref_recorded.py -m gpu -q -x -k "msst or pw_rel or decodes_reference" > gpurun_out/msst_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/msst_tests.log
tail -5 gpurun_out/msst_tests.log
timeout 600 python tools/gpu_msst_time.py ${1:-256} > gpurun_out/msst_time.log 2>&1
cat gpurun_out/msst_time.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-m-field --no-fast > gpurun_out/msst_bench.log 2>&1
python - <<'P'
import json
for l in open('gpurun_out/msst_bench.log'):
    if l.startswith('{'):
        j = json.loads(l); print('bench', j['value'], j['ms_per_step'], j['roofline'])
P
