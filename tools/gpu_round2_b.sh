#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/gpu_knobs.py 512 SZ_HIP_FILL=0,1 SZ_HIP_HPRIO=3,0 SZ_HIP_BACKOFF=4 > gpurun_out/r2b_knobs.log 2>&1; cat gpurun_out/r2b_knobs.log
timeout 300 python tools/gpu_knobs.py 512 SZ_HIP_FILL=1 SZ_HIP_HPRIO=1,2 SZ_HIP_BACKOFF=1,16 >> gpurun_out/r2b_knobs2.log 2>&1; cat gpurun_out/r2b_knobs2.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2b_tests.log 2>&1; tail -3 gpurun_out/r2b_tests.log
SZ_AMD_LIB=$PWD/sz_amd/csrc/libszhip_dev.so timeout 300 python tools/gpu_trace.py 512 > gpurun_out/r2b_trace.log 2>&1; head -12 gpurun_out/r2b_trace.log
