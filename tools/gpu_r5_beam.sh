#!/bin/bash
# round 5: first hardware run of the beam sweep: parity (small arrays vs the oracle, 512^3 vs k_ribbon's stream), timings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/gpu_r5_beam.py ${1:-512} > gpurun_out/r5_beam.log 2>&1
echo "exit $?" >> gpurun_out/r5_beam.log
tail -40 gpurun_out/r5_beam.log
