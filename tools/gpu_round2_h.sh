#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/gpu_knobs.py 512 SZ_HIP_FILL=0,2 > gpurun_out/r2h_base.log 2>&1; cat gpurun_out/r2h_base.log
SZ_AMD_LIB=$PWD/sz_amd/csrc/variants/libszhip_g8.so timeout 600 python tools/gpu_knobs.py 512 SZ_HIP_FILL=0,2 > gpurun_out/r2h_g8.log 2>&1; cat gpurun_out/r2h_g8.log
SZ_AMD_LIB=$PWD/sz_amd/csrc/libszhip_dev.so SZ_HIP_FILL=0 timeout 300 python tools/gpu_handoff.py 512 0 1 2>&1 | head -8
SZ_AMD_LIB=$PWD/sz_amd/csrc/variants/libszhip_g8dev.so SZ_HIP_FILL=0 timeout 300 python tools/gpu_handoff.py 512 0 1 2>&1 | head -8
SZ_AMD_LIB=$PWD/sz_amd/csrc/variants/libszhip_g8dev.so SZ_HIP_FILL=0 timeout 300 python tools/gpu_handoff.py 512 10 11 2>&1 | head -8
SZ_AMD_LIB=$PWD/sz_amd/csrc/libszhip_dev.so timeout 600 python tools/gpu_knobs_dev.py 512 SZ_HIP_FILL=0 SZ_HIP_DBG=1,2
