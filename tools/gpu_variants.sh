#!/bin/bash
# development: the bench line with every library variant under sz_amd/csrc/variants/ (SZ_AMD_LIB selects the shared object), twice
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for so in sz_amd/csrc/libszhip.so $(ls sz_amd/csrc/variants/libszhip_*.so); do
  SZ_AMD_LIB=$PWD/$so timeout 300 python bench.py --steps 10 --warmup 3 --inflight 1 --no-cpu-baseline --no-m-field --no-fast > gpurun_out/var.log 2>&1
  grep '^{"metric' gpurun_out/var.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$so'.split('/')[-1], 'value', d['value'], 'quant', d['phase_ms']['quant'], 'prequant', d['phase_ms']['prequant'], 'entropy', d['phase_ms']['entropy'], 'bytes', d['out_bytes'], 'err', d['max_abs_err'])"
done
done
