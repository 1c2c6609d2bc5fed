#!/bin/bash
# development: A/B of one environment knob on one box, same binary.  usage: KNOB=NAME VALUES="0 1" [BENCH_ARGS=...] gpu_knob_ab.sh
for rep in 1 2 3; do
  for v in ${VALUES:-0 1}; do
    env $KNOB=$v timeout 300 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS:---no-m-field --no-fast} --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$KNOB=$v:', j['value'], j['ms_per_step'], j['roofline']['avg_kernel_ms'], j['decompress_GBps'], j['phase_ms']['decompress_quant'], (j.get('m_field') or {}).get('ms_samples'))
"
  done
done
