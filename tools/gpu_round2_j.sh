#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/gpu_knobs.py 512 SZ_HIP_FILL=0,2 SZ_HIP_SPEC=0,4 > gpurun_out/r2j_base.log 2>&1; cat gpurun_out/r2j_base.log
timeout 600 python tools/gpu_knobs.py 512 SZ_HIP_FILL=0,2 SZ_HIP_WIDE=0 > gpurun_out/r2j_narrow.log 2>&1; cat gpurun_out/r2j_narrow.log
SZ_AMD_LIB=$PWD/sz_amd/csrc/libszhip_dev.so SZ_HIP_FILL=0 timeout 300 python tools/gpu_handoff.py 512 0 1 2>&1 | head -12
SZ_AMD_LIB=$PWD/sz_amd/csrc/libszhip_dev.so SZ_HIP_FILL=0 timeout 300 python tools/gpu_handoff.py 512 10 11 2>&1 | head -12
SZ_AMD_LIB=$PWD/sz_amd/csrc/libszhip_dev.so SZ_HIP_FILL=2 timeout 300 python tools/gpu_handoff.py 512 10 11 2>&1 | head -12
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
