#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SZ_AMD_LIB=$PWD/sz_amd/csrc/libszhip_dev.so
timeout 600 python tools/gpu_knobs_dev.py 512 SZ_HIP_FILL=0 SZ_HIP_DBG=0,1,2 > gpurun_out/r2c_dbg.log 2>&1; cat gpurun_out/r2c_dbg.log
timeout 600 python tools/gpu_knobs_dev.py 512 SZ_HIP_FILL=0 SZ_HIP_DBG=0 SZ_HIP_BACKOFF=1,4,32,128 > gpurun_out/r2c_backoff.log 2>&1; cat gpurun_out/r2c_backoff.log
timeout 600 python tools/gpu_knobs_dev.py 256 SZ_HIP_FILL=0 SZ_HIP_DBG=0,1,2 > gpurun_out/r2c_dbg256.log 2>&1; cat gpurun_out/r2c_dbg256.log
