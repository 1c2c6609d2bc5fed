#!/bin/bash
# round 2: staged host copies: host-pointer API tests + e2e numbers
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out

timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fast > gpurun_out/u_bench.log 2>&1; grep '^{"metric' gpurun_out/u_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['m_field']['GB/s'], d['e2e'])"
SZ_HIP_STAGED_COPY=0 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fast --no-m-field > gpurun_out/u_bench0.log 2>&1; grep '^{"metric' gpurun_out/u_bench0.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unstaged', d['e2e'])"
