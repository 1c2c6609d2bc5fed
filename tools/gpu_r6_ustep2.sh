#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r6_ustep2.log
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off"
i=0
for v in "1 0 1 1" "1 0 1 3" "1 0 1 9" "1 0 1 18" "0 0 1 9" "2 8 1 9" "4 0 1 9"; do
  set -- $v
  ( /opt/rocm/bin/hipcc $F -DVAR=$1 -DEXTRA=$2 -DFMA=$3 -DUNR=$4 -o /tmp/ub_step_$i tools/ubench/ub_step.hip 2>&1 | grep -E "error" -A3 ) &
  i=$((i+1))
done
wait
for j in $(seq 0 $((i-1))); do /tmp/ub_step_$j 3600 256 >> gpurun_out/r6_ustep2.log 2>&1; /tmp/ub_step_$j 3600 1 >> gpurun_out/r6_ustep2.log 2>&1; done
cat gpurun_out/r6_ustep2.log
