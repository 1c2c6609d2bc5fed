#!/bin/bash
# round 5: the sliced entropy stage beside the beam sweep, parity first and then the time of a call (S and M fields, 512^3)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r5_slices2.log
SZ_HIP_BEAM_SLICES=1 timeout 900 python -m pytest tests/test_beam_gpu.py tests/test_gpu_parity.py -m gpu -q 2>/dev/null | grep -E "passed|failed|Error|assert" | head -20 >> gpurun_out/r5_slices2.log
for f in s m; do
for cfg in "0 1 32" "1 3 32" "1 4 32" "1 6 32"; do
  set -- $cfg
  echo "== field=$f BEAM_SLICES=$1 SLICES=$2 PUB=$3" >> gpurun_out/r5_slices2.log
  SZ_HIP_BEAM=2 SZ_HIP_BEAM_SLICES=$1 SZ_HIP_SLICES=$2 SZ_HIP_BEAM_PUB=$3 python tools/gpu_r5_mtime.py 512 $f 2>&1 | grep '"field"' | tail -2 >> gpurun_out/r5_slices2.log
done; done
echo "== field=s ribbon default" >> gpurun_out/r5_slices2.log
python tools/gpu_r5_mtime.py 512 s 2>&1 | grep '"field"' | tail -2 >> gpurun_out/r5_slices2.log
cat gpurun_out/r5_slices2.log
