#!/bin/bash
# round 5: the beam fed while it runs (M-field 512^3): parity, then a call's time line with the feed off / on and 6 - 16 slices
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r5_feed.log
timeout 900 python -m pytest tests/test_beam_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|assert|Error" | grep -v szhip_decompress | head -12 >> gpurun_out/r5_feed.log
for cfg in "1 12" "1 10" "0 12"; do
  set -- $cfg
  echo "== FEED=$1 FEED_SLICES=$2" >> gpurun_out/r5_feed.log
  R5_DEC=0 SZ_HIP_BEAM_FEED=$1 SZ_HIP_FEED_SLICES=$2 timeout 300 python tools/gpu_r5_mtime.py 512 m 2>&1 | grep -E "field|szhip|chain threads" | tail -4 >> gpurun_out/r5_feed.log
done
cat gpurun_out/r5_feed.log
