#!/bin/bash
# round 2: entropy-stage kernels after the k_encode / k_hist_u16 changes: parity tests, kernel stats (no M-field), bench objects
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ref_recorded.py tests/test_fast_mode.py tests/test_gpu_parity.py -m gpu -q -x -k "not pwr" > gpurun_out/p_tests.log 2>&1; tail -5 gpurun_out/p_tests.log | cut -c1-400
BENCH_ARGS="--no-m-field" bash tools/gpu_kstats.sh 2>&1 | grep -E "k_fast|k_encode|k_hist|k_chunk|k_permute|k_pencil|k_minmax|k_fit|k_sample|k_unpred"
grep '^{"metric' gpurun_out/kstats_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['phase_ms']); print(d['fast_mode']['GB/s'], d['fast_mode']['phase_ms'])"
