#!/bin/bash
# development: the whole GPU test-suite + the bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/check_tests.log 2>&1; tail -15 gpurun_out/check_tests.log | cut -c1-300
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/check_bench.log 2>&1; tail -1 gpurun_out/check_bench.log | cut -c1-1800
