#!/bin/bash
# round 6: the beam sweep alone (tools/ubench/ub_beam.hip) over a few shapes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r6_ubeam.log
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -I sz_amd/csrc"
/opt/rocm/bin/hipcc $F $UB_EXTRA -o /tmp/ub_beam tools/ubench/ub_beam.hip 2>&1 | grep -E "error" -A3
for sh in "512 32 32" "512 128 32" "512 32 128" "512 128 128" "512 512 512"; do timeout 60 /tmp/ub_beam $sh >> gpurun_out/r6_ubeam.log 2>&1; done
cat gpurun_out/r6_ubeam.log
