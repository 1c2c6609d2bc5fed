#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== free run (dbg=1)"; SZ_HIP_DBG=1 timeout 120 python tools/gpu_trace.py 512 2>&1 | grep -E "ms_quant|pencil \(32,32\)|pencil \(63,63\)|active duration"
echo "== normal"; timeout 120 python tools/gpu_trace.py 512 2>&1 | grep -E "ms_quant|pencil \(0,0\)|pencil \(32,32\)|pencil \(63,63\)|active duration|all pencils"
