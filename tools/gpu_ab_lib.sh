#!/bin/bash
# development: A/B of two libraries on ONE box: bench --timed-only with K = 2 (and K = 1), alternating, ${REPS:-3} times each
cd $GRAFT_REPO_ROOT
A=${LIB_A:-sz_amd/csrc/libszhip.so}; B=${LIB_B}
for rep in $(seq 1 ${REPS:-3}); do for so in $A $B; do
  SZ_AMD_LIB=$PWD/$so timeout 200 python bench.py --timed-only --steps ${STEPS:-40} --warmup 8 --inflight ${K:-2} > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print("$so K=${K:-2}:", d["value"], d["ms_per_step"], "sweep", d["roofline"]["avg_kernel_ms"], "entropy", d.get("phases", d.get("phase_ms", {})).get("entropy"), "decompress", d["decompress_GBps"])
PY
done; done
