#!/bin/bash
# round 6: quick look after a change of the packing passes: a few parity cases, one call's device time line, per-call medians under several settings
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_beam_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|assert|rror" | grep -v szhip_decompress | tail -5 ) > gpurun_out/r6_check_tests.txt
cat gpurun_out/r6_check_tests.txt
R5_DEC=0 bash tools/gpu_r6_trace.sh s chk > /dev/null 2>&1
grep -E "k_|fill|copy" gpurun_out/r6_chk_timeline.txt | awk '$1 > 3.0 && $1 < 4.8' | head -60
out=gpurun_out/r6_var.txt; : > $out
run() { tag="$1"; shift; env TAG="$tag" "$@" python tools/gpu_r6_calls.py 2>&1 | grep -E "median" >> $out; }
run "default"
run "seghist=0" SZ_HIP_SEGHIST=0
run "seghist=0,slices=1" SZ_HIP_SEGHIST=0 SZ_HIP_SLICES=1
run "segenc=0" SZ_HIP_SEGENC=0
run "rounds=0" SZ_HIP_SEG_ROUNDS=0
run "segb=14" SZ_HIP_SEG_SEGB=14
run "tile40" SZ_HIP_SEG_TILE_KB=40
run "m default" FIELD=m
run "m segenc=0" FIELD=m SZ_HIP_SEGENC=0
cat $out
