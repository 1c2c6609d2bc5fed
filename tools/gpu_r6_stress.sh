#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_multi_context_stress.txt
echo "# round 6: tools/gpu_r6_multictx.py -- T lone contexts on host threads, N calls each of the 512^3 M-field (303 450 regression blocks), every stream compared with a single call's" > $out
for cfg in "1 300" "2 5000" "4 2500"; do timeout 1500 python tools/gpu_r6_multictx.py $cfg 2>&1 | tail -1 >> $out; done
cat $out
python tools/gpu_r5_mtime.py 512 m 2>&1 | grep '"it"'
