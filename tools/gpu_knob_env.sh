#!/bin/bash
# development: the bench line under different values of one environment knob:  gpu_knob_env.sh NAME v1 v2 ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
name=$1; shift
for v in "$@"; do
  env $name=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-m-field --no-fast > gpurun_out/knob.log 2>&1
  grep '^{"metric' gpurun_out/knob.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name=$v', 'value', d['value'], d['phase_ms'], 'bytes', d['out_bytes'], 'dec', d['decompress_GBps'])"
done
