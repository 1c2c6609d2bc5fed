#!/bin/bash
cd $GRAFT_REPO_ROOT
for dbg in 0 5 0 5; do
echo "== dbg=$dbg"; SZ_HIP_DBG=$dbg timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['out_bytes'], d['phase_ms']['quant'], d['phase_ms']['decompress_quant'])"
done
