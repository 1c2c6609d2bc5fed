#!/bin/bash
cd $GRAFT_REPO_ROOT
for g in 8 16 24; do
  echo "gate=$g"; SZ_HIP_GATE_STEPS=$g timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['out_bytes'], d['phase_ms'])"
done
SZ_HIP_GATE_STEPS=16 python tools/gpu_trace.py 512 2>&1 | tail -6
