#!/bin/bash
cd $GRAFT_REPO_ROOT
for sl in 0 4 8 16 32; do
echo "== slack=$sl"; SZ_HIP_SLACK=$sl timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['out_bytes'], d['phase_ms']['quant'], d['phase_ms']['decompress_quant'])"
done
