#!/bin/bash
# the entropy stage's passes on finished tile rows while the sweep runs (SZ_HIP_SLICES): timeline of one call and the bench line for 1 / 4 / 8 / 16 slices
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${TAG:-r4s}
if [ -n "$TESTS" ]; then timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_ref_recorded.py -x -q -m gpu > $O/${T}_tests.log 2>&1; grep -E 'passed|failed' $O/${T}_tests.log | tail -1; fi
if [ -n "$TL" ]; then TAG=${T}_tl bash tools/gpu_r4_trace_main.sh | tail -42; fi
for V in ${VARIANTS:-"SZ_HIP_SLICES=1" "SZ_HIP_SLICES=8"}; do
  S=$(echo $V | tr -c "A-Za-z0-9\n" "_")
  env $(echo $V | tr ";" " ") timeout 300 python bench.py --no-omp --no-other-paths --no-cpu-baseline --no-m-field --no-fast --steps 10 --warmup 3 ${BENCH_ARGS} > $O/${T}_b$S.log 2>&1
  grep '^{"metric"' $O/${T}_b$S.log | tail -1 > $O/${T}_b$S.json
  python3 - <<PY
import json
d = json.load(open("$O/${T}_b$S.json"))
print("$V value", d["value"], "single", d["single_call"], "phase", d["phase_ms"], "K", {k: (v["GB/s"], v["quant_ms"]) for k, v in d["concurrent"]["K"].items()})
PY
done
