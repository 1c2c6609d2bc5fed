#!/bin/bash
# development: how a tile learns which tile it is (SZ_HIP_TICKET_MODE 0: atomic ticket + table, 1: blockIdx + table, 2: blockIdx, computed), same box, same binary
for rep in 1 2 3; do
  for m in ${MODES:-0 1 2}; do
    SZ_HIP_TICKET_MODE=$m timeout 300 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS:---no-m-field --no-fast} --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('ticket_mode $m:', j['value'], j['ms_per_step'], j['roofline']['avg_kernel_ms'], j['decompress_GBps'], j['phase_ms']['decompress_quant'], (j.get('m_field') or {}).get('ms_samples'))
"
  done
done
