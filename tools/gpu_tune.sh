#!/bin/bash
cd $GRAFT_REPO_ROOT
for g in 4 8 12 16; do
  echo "gate=$g"; SZ_HIP_GATE_STEPS=$g timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['out_bytes'], d['phase_ms'], d['decompress_GBps'])"
done
python tools/gpu_trace.py 512 2>&1 | grep -E "ms_quant|active duration|gate lag|spins total|trip  [0-3]:|trip 1[0-2]:"
