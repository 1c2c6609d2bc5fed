#!/bin/bash
# round 5: the chains started on the first piece of the coefficients (SZ_HIP_CHAIN_EARLY): parity of the fed sweep, then the M-field call with it off / on
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r5_early.log
timeout 900 python -m pytest tests/test_beam_gpu.py -m gpu -q -x 2>/dev/null | grep -E "passed in|failed in| passed,| failed,|^FAILED|^ERROR" | tail -4 >> gpurun_out/r5_early.log
for early in 0 1 1; do
  echo "== CHAIN_EARLY=$early" >> gpurun_out/r5_early.log
  R5_DEC=0 SZ_HIP_CHAIN_EARLY=$early python tools/gpu_r5_mtime.py 512 m 2>&1 | grep -E '"field"|chain threads' | cut -c1-420 >> gpurun_out/r5_early.log
done
cat gpurun_out/r5_early.log
