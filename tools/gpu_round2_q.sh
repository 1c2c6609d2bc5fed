#!/bin/bash
# round 2: point-wise relative bounds on the GPU (recorded reference outputs), then everything else of that file
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ref_recorded.py -m gpu -q > gpurun_out/q_tests.log 2>&1; grep -aE "^E  |passed|failed|FAILED" gpurun_out/q_tests.log | head -40 | cut -c1-300
