#!/bin/bash
# development: what the driver runs at round end (GPU suite, smoke, default bench), in one gpurun call
mkdir -p gpurun_out
timeout 420 python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final_tests.log
grep -n "passed\|failed\|FAILED\|pytest rc" gpurun_out/final_tests.log | tail -5
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 200 python bench.py > gpurun_out/final_bench.log 2>&1; tail -1 gpurun_out/final_bench.log | cut -c1-400
