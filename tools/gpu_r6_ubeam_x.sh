#!/bin/bash
# round 6: the beam sweep alone with parts switched off (SZH_BM_X, timing only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r6_ubeam_x.log
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -I sz_amd/csrc"
for x in ${@:-0 1 2 4 6 8 14}; do ( /opt/rocm/bin/hipcc $F -DSZH_BM_X=$x -o /tmp/ub_beam_$x tools/ubench/ub_beam.hip 2>&1 | grep -E "error" -A3 ) & done; wait
for x in ${@:-0 1 2 4 6 8 14}; do for sh in "512 32 32"; do echo -n "X=$x: " >> gpurun_out/r6_ubeam_x.log; timeout 60 /tmp/ub_beam_$x $sh >> gpurun_out/r6_ubeam_x.log 2>&1; done; done
cat gpurun_out/r6_ubeam_x.log
