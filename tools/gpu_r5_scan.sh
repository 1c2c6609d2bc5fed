#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/gpu_r5_scan.py "$@" > gpurun_out/r5_scan.log 2>&1; echo "exit $?" >> gpurun_out/r5_scan.log; cat gpurun_out/r5_scan.log
