#!/bin/bash
cd $GRAFT_REPO_ROOT
SZ_AMD_LIB=$PWD/sz_amd/csrc/libszhip_dev.so timeout 600 python tools/gpu_knobs_dev.py 512 SZ_HIP_FILL=0 SZ_HIP_DBG=0,3,4,5
