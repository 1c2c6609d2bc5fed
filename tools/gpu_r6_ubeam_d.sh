#!/bin/bash
# round 6: the beam sweep alone, compile-time variants given as "-D..." strings
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r6_ubeam_d.log
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -I sz_amd/csrc"
i=0; for d in "$@"; do ( /opt/rocm/bin/hipcc $F $d -o /tmp/ub_beam_d$i tools/ubench/ub_beam.hip 2>&1 | grep -E "error" -A3 ) & i=$((i+1)); done; wait
i=0; for d in "$@"; do for sh in "512 32 32" "512 128 128" "512 512 512"; do echo -n "[$d] " >> gpurun_out/r6_ubeam_d.log; timeout 60 /tmp/ub_beam_d$i $sh >> gpurun_out/r6_ubeam_d.log 2>&1; done; i=$((i+1)); done
cat gpurun_out/r6_ubeam_d.log
