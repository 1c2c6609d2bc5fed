#!/bin/bash
# round 5: a quick look after a change: the beam / parity tests, one call's time line both ways (S and M fields), the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r5_check.log
timeout 900 python -m pytest tests/test_beam_gpu.py tests/test_gpu_parity.py tests/test_ref_recorded.py -m gpu -q -x 2>/dev/null | grep -E "passed in|failed in| passed,| failed,|^FAILED|^ERROR" | tail -4 >> gpurun_out/r5_check.log
python tools/gpu_r5_mtime.py 512 s,m 2>&1 | grep -E "field|header|sections|chain threads" >> gpurun_out/r5_check.log
python bench.py 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r5_check_bench.json
python3 -c "
import json; d=json.load(open('gpurun_out/r5_check_bench.json')); print(json.dumps({k:d.get(k) for k in ('value','ms_per_step','step_ms','decompress_GBps','m_field')}))" >> gpurun_out/r5_check.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> gpurun_out/r5_check.log
cat gpurun_out/r5_check.log
