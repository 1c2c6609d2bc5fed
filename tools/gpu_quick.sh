#!/bin/bash
# development: recorded + parity GPU tests, then per-kernel averages of the bench command (no M-field, no fast mode)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ref_recorded.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/quick_tests.log 2>&1; grep -aE "^E  |[0-9]+ passed|failed|FAILED" gpurun_out/quick_tests.log | head -6 | cut -c1-300
BENCH_ARGS="--no-fast --no-m-field" bash tools/gpu_kstats.sh 2>&1 | grep -E "k_encode|k_hist|k_chunk|k_permute|k_pencil|k_fit|k_sample|k_unpred|k_hdec"
grep '^{"metric' gpurun_out/kstats_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phase_ms'], d['decompress_GBps'])"
