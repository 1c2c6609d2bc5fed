#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SZ_AMD_LIB=$PWD/sz_amd/csrc/libszhip_dev.so
for m in 0 2; do
SZ_HIP_FILL=$m timeout 300 python tools/gpu_handoff.py 512 0 1 > gpurun_out/r2e_handoff_m${m}_0_1.log 2>&1; cat gpurun_out/r2e_handoff_m${m}_0_1.log
SZ_HIP_FILL=$m timeout 300 python tools/gpu_handoff.py 512 10 11 > gpurun_out/r2e_handoff_m${m}_10_11.log 2>&1; cat gpurun_out/r2e_handoff_m${m}_10_11.log
done
