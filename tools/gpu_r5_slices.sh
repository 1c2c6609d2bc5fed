#!/bin/bash
# round 5: the entropy stage beside the beam sweep: what the progress words and the slices' kernels cost (S-field 512^3, beam forced)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r5_slices.log
export SZ_HIP_BEAM=2
for cfg in ${R5_CFGS:-"0 1 32" "1 4 32" "1 6 32" "1 8 32" "1 4 16" "1 8 16" "1 3 32"}; do
  set -- $cfg
  echo "== BEAM_SLICES=$1 SLICES=$2 PUB=$3" >> gpurun_out/r5_slices.log
  SZ_HIP_BEAM_SLICES=$1 SZ_HIP_SLICES=$2 SZ_HIP_BEAM_PUB=$3 python tools/gpu_r5_mtime.py 512 ${R5_FIELD:-s} 2>&1 | grep '"field"' | tail -2 >> gpurun_out/r5_slices.log
done
cat gpurun_out/r5_slices.log
