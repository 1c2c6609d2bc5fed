#!/bin/bash
# round 6: the same calls under several switch settings (one line of medians each)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_var.txt; : > $out
run() { tag="$1"; shift; env TAG="$tag" "$@" python tools/gpu_r6_calls.py 2>&1 | grep -E "median|per call" >> $out; }
run "default" 
run "slices=1" SZ_HIP_SLICES=1
run "slices=2" SZ_HIP_SLICES=2
run "slices=8" SZ_HIP_SLICES=8
run "segenc=0" SZ_HIP_SEGENC=0
run "segenc=0,slices=1" SZ_HIP_SEGENC=0 SZ_HIP_SLICES=1
run "tile12" SZ_HIP_SEG_TILE_KB=12
run "tile40" SZ_HIP_SEG_TILE_KB=40
cat $out
