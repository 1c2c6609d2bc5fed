#!/bin/bash
# round 5: what each part of a step of the beam sweep costs: tools/ubench/ub_beam.hip built with the step's development switches (SZH_BM_X)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r5_ubeam.log
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I sz_amd/csrc"
for x in ${@:-0 1 2 4 8 16 32 64 127}; do
  ( /opt/rocm/bin/hipcc $F -DSZH_BM_X=$x $UB_EXTRA -o /tmp/ub_beam_$x tools/ubench/ub_beam.hip 2>&1 | grep -E "error" -A3 ) &
done
wait
for x in ${@:-0 1 2 4 8 16 32 64 127}; do echo -n "X=$x: " >> gpurun_out/r5_ubeam.log; /tmp/ub_beam_$x ${UB_SHAPE:-512 32 32} >> gpurun_out/r5_ubeam.log 2>&1; done
cat gpurun_out/r5_ubeam.log
