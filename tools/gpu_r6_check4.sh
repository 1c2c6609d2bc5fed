#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_var2.txt; : > $out
run() { tag="$1"; shift; env TAG="$tag" "$@" python tools/gpu_r6_calls.py 2>&1 | grep -E "median" >> $out; }
run "c4 default" FIELD=c4 NCALLS=10
run "c4 segenc=0" FIELD=c4 NCALLS=10 SZ_HIP_SEGENC=0
run "c4 fit_tile=0" FIELD=c4 NCALLS=10 SZ_HIP_FIT_TILE=0
cat $out
FIELD=c4 NCALLS=3 TRACE_CMD="python tools/gpu_r6_calls.py" bash tools/gpu_r6_trace.sh s c4 > /dev/null 2>&1
grep -E "k_|fill" gpurun_out/r6_c4_timeline.txt | head -60
