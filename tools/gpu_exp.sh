#!/bin/bash
cd $GRAFT_REPO_ROOT
for g in 16; do
  echo "== gate=$g"; SZ_HIP_GATE_STEPS=$g timeout 120 python tools/gpu_trace.py 512 2>&1 | grep -E "ms_quant|pencil \(32,32\)|pencil \(63,63\)|active duration|end lag|tile|^    |gate of|first_trip of"
done
