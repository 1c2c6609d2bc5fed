#!/bin/bash
# development: A/B of builds of the library on one box (headline bench, S-field only).  usage: gpu_ab.sh [reps] lib1.so lib2.so ...
mkdir -p gpurun_out
reps=${1:-3}; shift
libs=${@:-sz_amd/csrc/libszhip_ab_old.so sz_amd/csrc/libszhip.so}
for rep in $(seq $reps); do
  for lib in $libs; do
    SZ_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-m-field --no-fast --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$lib', j['value'], j['ms_per_step'], j['roofline']['avg_kernel_ms'])
"
  done
done
