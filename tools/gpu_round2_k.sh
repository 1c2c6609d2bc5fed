#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SZ_AMD_LIB=$PWD/sz_amd/csrc/libszhip_dev.so timeout 600 python tools/gpu_knobs_dev.py 512 SZ_HIP_FILL=0,2 SZ_HIP_DBG=0,3
SZ_AMD_LIB=$PWD/sz_amd/csrc/libszhip_dev.so SZ_HIP_DBG=3 SZ_HIP_FILL=0 timeout 300 python tools/gpu_handoff.py 512 10 11 2>&1 | head -12
