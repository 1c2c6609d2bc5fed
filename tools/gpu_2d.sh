#!/bin/bash
# development: 2-D path on the GPU -- parity tests, fuzz, timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "2d or unsupported" 2>&1 | grep -vE "^\[SZ\]|^Error" | tail -15
timeout 600 python tools/gpu_fuzz.py 300 21 2>&1 | grep -vE "^\[SZ\]" | tail -8
timeout 120 python tools/gpu_2d_time.py 4096 2>&1 | grep -vE "^\[SZ\]" | tail -8
timeout 120 python tools/gpu_2d_time.py 2048 f64 2>&1 | grep -vE "^\[SZ\]" | tail -3
