#!/bin/bash
# round 2: HDF5 filter, entry points, CLI on the GPU, then the whole suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out

timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r_all.log 2>&1; grep -aE "^E  |passed|failed|FAILED" gpurun_out/r_all.log | head -20 | cut -c1-300
