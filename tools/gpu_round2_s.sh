#!/bin/bash
# round 2: range reduction fused into the fit pass + unpredictable gather on the second stream: whole GPU suite, then the bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/s_all.log 2>&1; grep -aE "^E  |passed|failed|FAILED" gpurun_out/s_all.log | head -20 | cut -c1-300
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/s_bench.log 2>&1; grep '^{"metric' gpurun_out/s_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phase_ms']); print(d['roofline']['frac'], d['m_field']['GB/s'], d['m_field']['ms_samples'], d['fast_mode']['GB/s'], d['e2e'])"
