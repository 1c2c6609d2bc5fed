#!/bin/bash
# round 6: step-cost micro-benchmark (tools/ubench/ub_step.hip) in its variants
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r6_ustep.log
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off"
i=0
for v in "0 0 0" "0 0 1" "1 0 0" "1 0 1" "2 8 1" "2 16 1" "2 24 1" "2 32 1" "3 8 1" "3 16 1" "3 32 1" "4 0 1"; do
  set -- $v
  ( /opt/rocm/bin/hipcc $F -DVAR=$1 -DEXTRA=$2 -DFMA=$3 -o /tmp/ub_step_$i tools/ubench/ub_step.hip 2>&1 | grep -E "error" -A3 ) &
  i=$((i+1))
done
wait
for j in $(seq 0 $((i-1))); do /tmp/ub_step_$j 4000 256 >> gpurun_out/r6_ustep.log 2>&1; done
/tmp/ub_step_2 4000 1 >> gpurun_out/r6_ustep.log 2>&1
cat gpurun_out/r6_ustep.log
