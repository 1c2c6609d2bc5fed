#!/bin/bash
# development: the PCIe-inclusive numbers of bench.py (`e2e`) against the number of host threads of staged_copy
for t in 4 8 2; do
  SZ_HIP_STAGE_THREADS=$t timeout 300 python bench.py --steps 5 --warmup 2 --no-m-field --no-fast 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('threads $t', j['e2e'])
"
done
nproc
