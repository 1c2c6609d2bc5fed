#!/bin/bash
# round 5: the whole GPU suite, then one call's time line both ways for the S and the M field (512^3)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>/dev/null | grep -E "passed|failed|assert|Error:" | grep -v szhip_decompress | tail -8 > gpurun_out/r5_suite.log
python tools/gpu_r5_mtime.py 512 s,m 2>&1 | grep field >> gpurun_out/r5_suite.log
cat gpurun_out/r5_suite.log
