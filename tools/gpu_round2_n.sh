#!/bin/bash
# round 2: fast mode + lossless stage on the GPU, then the bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fast_mode.py tests/test_ref_recorded.py -m gpu -q -k "not pwr" > gpurun_out/n_tests.log 2>&1; tail -25 gpurun_out/n_tests.log | cut -c1-400
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/n_bench.log 2>&1; tail -1 gpurun_out/n_bench.log | cut -c1-4000
