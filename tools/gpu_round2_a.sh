#!/bin/bash
# round 2, first GPU trip: parity of the leaner wavefront kernel, bench line, per-pencil trace (dev build), shape scan
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "not pwr and not lossless and not zstd and not gzip" > gpurun_out/r2a_tests.log 2>&1; tail -5 gpurun_out/r2a_tests.log
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r2a_bench.log 2>&1; tail -1 gpurun_out/r2a_bench.log | cut -c1-1500
SZ_AMD_LIB=$PWD/sz_amd/csrc/libszhip_dev.so timeout 300 python tools/gpu_trace.py 512 > gpurun_out/r2a_trace.log 2>&1; head -40 gpurun_out/r2a_trace.log
timeout 300 python tools/gpu_shape_scan.py > gpurun_out/r2a_scan.log 2>&1; cat gpurun_out/r2a_scan.log
