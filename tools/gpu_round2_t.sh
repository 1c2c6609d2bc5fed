#!/bin/bash
# round 2: parity (recorded + parity files) and kernel stats after the k_permute change
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ref_recorded.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/t_tests.log 2>&1; grep -aE "^E  |passed|failed|FAILED" gpurun_out/t_tests.log | head -8 | cut -c1-300
BENCH_ARGS="--no-fast" bash tools/gpu_kstats.sh 2>&1 | grep -E "k_encode|k_hist|k_chunk|k_permute|k_pencil|k_fit|k_sample|k_unpred|k_minmax"
grep '^{"metric' gpurun_out/kstats_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phase_ms']); print(d['m_field']['GB/s'], d['m_field']['ms_samples'], d['decompress_GBps'], d['e2e'])"
