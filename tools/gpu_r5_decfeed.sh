#!/bin/bash
# round 5: the inverse sweep fed while it runs (S-field 512^3 decompression): parity, then the time of a call unfed / fed with 4 - 16 slices
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r5_decfeed.log
timeout 900 python -m pytest tests/test_beam_gpu.py tests/test_gpu_parity.py tests/test_ref_recorded.py -m gpu -q -x 2>/dev/null | grep -E "passed in|failed in| passed,| failed,|^FAILED|^ERROR" | tail -4 >> gpurun_out/r5_decfeed.log
for cfg in "0 8" "1 8" "1 4" "1 16"; do
  set -- $cfg
  echo "== DEC_FEED=$1 SLICES=$2" >> gpurun_out/r5_decfeed.log
  SZ_HIP_DEC_FEED=$1 SZ_HIP_DEC_FEED_SLICES=$2 python tools/gpu_r5_mtime.py 512 s 2>&1 | grep -E 'dec_it' >> gpurun_out/r5_decfeed.log
done
cat gpurun_out/r5_decfeed.log
