#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_var.txt; : > $out
run() { tag="$1"; shift; env TAG="$tag" "$@" python tools/gpu_r6_calls.py 2>&1 | grep -E "median" >> $out; }
for rep in 1 2; do
run "default(tile,late)"
run "fit_tile=0" SZ_HIP_FIT_TILE=0
run "mean_first" SZ_HIP_MEAN_FIRST=1
run "mean_first,fit_tile=0" SZ_HIP_MEAN_FIRST=1 SZ_HIP_FIT_TILE=0
run "mean_first,early" SZ_HIP_MEAN_FIRST=1 SZ_HIP_SAMPLE_EARLY=1
done
cat $out
