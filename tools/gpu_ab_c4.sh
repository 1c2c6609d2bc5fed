#!/bin/bash
# development: A/B of builds of the library on one box, configs[3] slab (132x1024x1024 float64).  usage: gpu_ab_c4.sh [reps] lib1.so ...
reps=${1:-2}; shift
for rep in $(seq $reps); do
  for lib in "$@"; do
    SZ_AMD_LIB=$PWD/$lib timeout 300 python bench.py --config c4 --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$lib', j['value'], j['ms_per_step'], j['phase_ms_rank0'])
"
  done
done
