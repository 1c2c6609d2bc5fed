#!/bin/bash
# the timed region (the driver's flags: 20 steps, 5 warm-up) with 2 / 3 / 4 arrays in flight, twice each
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for rep in 1 2; do
for K in ${LANES:-2 3 4}; do
  env $EXTRA timeout 300 python bench.py --no-omp --no-other-paths --no-cpu-baseline --no-m-field --no-fast $NOCONC --steps 20 --warmup 5 --inflight $K > $O/ln_$K.log 2>&1
  grep '^{"metric"' $O/ln_$K.log | tail -1 > $O/ln_$K.json
  python3 - <<PY
import json
d = json.load(open("$O/ln_$K.json"))
print("inflight $K value", d["value"], "ms/step", d["ms_per_step"], "single", d["single_call"]["GB/s"], "phase", d["phase_ms"])
PY
done
done
