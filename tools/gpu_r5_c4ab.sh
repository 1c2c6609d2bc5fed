#!/bin/bash
# round 5: the C4 line (132 x 1024 x 1024 float64 slab) with the beam's sliced entropy stage / the feed on and off; one M-field call both ways; the GPU suite's summary
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r5_c4ab.log
for cfg in "1 1" "0 1" "1 0"; do
  set -- $cfg
  echo "== BEAM_SLICES=$1 BEAM_FEED=$2" >> gpurun_out/r5_c4ab.log
  SZ_HIP_BEAM_SLICES=$1 SZ_HIP_BEAM_FEED=$2 python bench.py --config c4 2>/dev/null | grep '^{"metric"' | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({k:d.get(k) for k in ('value','ms_per_step','phase_ms_rank0','decompress_GBps_per_gpu')}))" >> gpurun_out/r5_c4ab.log
done
python tools/gpu_r5_mtime.py 512 m 2>&1 | grep -E "field|header|sections" >> gpurun_out/r5_c4ab.log
timeout 1500 python -m pytest tests -m gpu -q 2>/dev/null | grep -E "passed in|failed in| passed,| failed,|^FAILED|^ERROR" | tail -5 >> gpurun_out/r5_c4ab.log
cat gpurun_out/r5_c4ab.log
