#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_check.sh
for i in 1 2 3; do timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['phase_ms']['quant'], d['phase_ms']['decompress_quant'], d['decompress_GBps'])"; done
