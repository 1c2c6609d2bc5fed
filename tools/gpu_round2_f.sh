#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SZ_AMD_LIB=$PWD/sz_amd/csrc/variants/libszhip_g8.so timeout 600 python tools/gpu_knobs.py 512 SZ_HIP_FILL=0,2 SZ_HIP_SPEC=0,4 > gpurun_out/r2f_g8.log 2>&1; cat gpurun_out/r2f_g8.log
timeout 600 python tools/gpu_knobs.py 512 SZ_HIP_FILL=0 > gpurun_out/r2f_base.log 2>&1; cat gpurun_out/r2f_base.log
