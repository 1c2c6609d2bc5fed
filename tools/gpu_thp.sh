#!/bin/bash
# development: does the huge-page hint on the decompress output change the PCIe-inclusive decompress rate?
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>/dev/null
for t in 1 0 1 0; do
  SZ_HIP_THP=$t timeout 300 python bench.py --steps 5 --warmup 2 --no-m-field --no-fast 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('thp $t', j['e2e']['compress_GBps'], j['e2e']['decompress_GBps'])
"
done
