#!/bin/bash
# development: BASELINE configs[3] (one 132x1024x1024 float64 slab) with every library variant under sz_amd/csrc/variants/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for so in sz_amd/csrc/libszhip.so $(ls sz_amd/csrc/variants/libszhip_*.so); do
  SZ_AMD_LIB=$PWD/$so timeout 300 python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/var.log 2>&1
  grep '^{"metric' gpurun_out/var.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$so'.split('/')[-1], 'value', d['value'], 'ms', d['ms_per_step'], {k:v for k,v in d.items() if k in ('phase_ms','ratio')})"
done
