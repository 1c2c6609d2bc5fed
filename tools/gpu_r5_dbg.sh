#!/bin/bash
# round 5: which part of a line of the beam sweep costs what (SZ_HIP_DBG: timing only, results wrong)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r5_dbg.log
for d in 0 1 2 4 8 14 15; do
  echo "== SZ_HIP_DBG=$d" >> gpurun_out/r5_dbg.log
  SZ_HIP_DBG=$d timeout 300 python tools/gpu_r5_scan.py ${@:-512x32x32 512x512x512} >> gpurun_out/r5_dbg.log 2>&1
done
cat gpurun_out/r5_dbg.log
