#!/bin/bash
# development: bench line under a few settings of an environment tunable
cd $GRAFT_REPO_ROOT
for b in 4 16 48 4 16 48; do
echo "SZ_HIP_BACKOFF=$b $(SZ_HIP_BACKOFF=$b timeout 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['out_bytes'], d['phase_ms']['quant'], d['phase_ms']['decompress_quant'])")"
done
