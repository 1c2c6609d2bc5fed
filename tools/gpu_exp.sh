#!/bin/bash
cd $GRAFT_REPO_ROOT
for dbg in 0 1 2; do
  echo "== SZ_HIP_DBG=$dbg"; SZ_HIP_DBG=$dbg timeout 120 python tools/gpu_trace.py 256 2>&1 | grep -E "ms_quant|pencil \(0,0\)|active duration|gate lag"
done
