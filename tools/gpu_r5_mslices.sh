#!/bin/bash
# round 5: the fed M-field call with 4 (default) / 6 / 8 / 12 slices of the entropy stage
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r5_mslices.log
for ns in 4 6 8 12; do
  echo "== SZ_HIP_SLICES=$ns" >> gpurun_out/r5_mslices.log
  R5_DEC=0 SZ_HIP_SLICES=$ns python tools/gpu_r5_mtime.py 512 m 2>&1 | grep -E '"field"' | tail -3 >> gpurun_out/r5_mslices.log
done
cat gpurun_out/r5_mslices.log
