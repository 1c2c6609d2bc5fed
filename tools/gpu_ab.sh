#!/bin/bash
# development: A/B of two builds of the library on one box (headline bench, S-field only): sz_amd/csrc/libszhip_ab_old.so against libszhip.so
mkdir -p gpurun_out
for rep in 1 2 3; do
  for lib in sz_amd/csrc/libszhip_ab_old.so sz_amd/csrc/libszhip.so; do
    SZ_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-m-field --no-fast 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$lib', j['value'], j['ms_per_step'], j['roofline']['avg_kernel_ms'])
"
  done
done
