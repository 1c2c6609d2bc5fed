#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SZ_AMD_LIB=$PWD/sz_amd/csrc/variants/libszhip_g8dev.so
for m in 0 2; do
SZ_HIP_FILL=$m timeout 300 python tools/gpu_handoff.py 512 0 1 2>&1 | head -24
SZ_HIP_FILL=$m timeout 300 python tools/gpu_handoff.py 512 10 11 2>&1 | head -24
done > gpurun_out/r2g_handoff_g8.log 2>&1
cat gpurun_out/r2g_handoff_g8.log
