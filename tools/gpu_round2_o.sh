#!/bin/bash
# round 2: fast mode after the k_fast_quant rewrite: tests, kernel stats (no M-field), bench object
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fast_mode.py -m gpu -q > gpurun_out/o_tests.log 2>&1; tail -5 gpurun_out/o_tests.log | cut -c1-400
BENCH_ARGS="--no-m-field" bash tools/gpu_kstats.sh 2>&1 | grep -E "k_fast|k_encode|k_hist|k_chunk|copyBuffer|fillBuffer|k_scan"
grep '^{"metric' gpurun_out/kstats_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value']); print(d['fast_mode'])"
