#!/bin/bash
# development (round 3): the 512^3 property test + the bench line with the ribbon kernel, then the same bench on k_pencil (SZ_HIP_RIBBON=0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "baseline_size or golden or f64_slab" > gpurun_out/rb_tests.log 2>&1; grep -aE "passed|failed|FAILED|^E  " gpurun_out/rb_tests.log | head -8 | cut -c1-300
timeout 300 python bench.py --steps 10 --warmup 3 ${BENCH_ARGS} > gpurun_out/rb_bench.log 2>&1; tail -1 gpurun_out/rb_bench.log | cut -c1-1500
if [ -n "$RB_AB" ]; then SZ_HIP_RIBBON=0 timeout 300 python bench.py --steps 10 --warmup 3 ${BENCH_ARGS} > gpurun_out/rb_bench_pencil.log 2>&1; tail -1 gpurun_out/rb_bench_pencil.log | cut -c1-700; fi
