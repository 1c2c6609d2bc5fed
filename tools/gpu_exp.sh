#!/bin/bash
cd $GRAFT_REPO_ROOT
for tg in 1 0 1 0; do
  echo "== tripgate=$tg"; SZ_HIP_TRIPGATE=$tg timeout 120 python tools/gpu_trace.py 512 2>&1 | grep -E "ms_quant|pencil \(32,32\)|pencil \(0,0\)|pencil \(63,63\)|active duration|gate of"
done
