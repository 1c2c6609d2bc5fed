#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/gpu_knobs.py 512 SZ_HIP_FILL=0,2 SZ_HIP_SPEC=0 > gpurun_out/r2d_knobs.log 2>&1; cat gpurun_out/r2d_knobs.log
timeout 600 python tools/gpu_knobs.py 512 SZ_HIP_FILL=2 SZ_HIP_SPEC=2,4,8 SZ_HIP_BACKOFF=1,4 > gpurun_out/r2d_knobs2.log 2>&1; cat gpurun_out/r2d_knobs2.log
SZ_AMD_LIB=$PWD/sz_amd/csrc/libszhip_dev.so timeout 300 python tools/gpu_trace.py 512 > gpurun_out/r2d_trace.log 2>&1; head -12 gpurun_out/r2d_trace.log
