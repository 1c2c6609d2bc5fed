#!/bin/bash
# the timed region's value is bimodal from run to run (341 / 247 GB/s on one box): per-call durations of the timed region of several runs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export SZ_BENCH_CALL_PHASES=1
for i in ${RUNS:-1 2 3 4 5 6 7 8}; do
  env $EXTRA timeout 300 python bench.py --no-omp --no-other-paths --no-cpu-baseline --no-m-field --no-fast --steps ${STEPS:-10} --warmup ${WARM:-3} --inflight ${INFLIGHT:-2} > $O/bi_$i.log 2>&1
  grep '^{"metric"' $O/bi_$i.log | tail -1 > $O/bi_$i.json
  python3 - <<PY
import json
d = json.load(open("$O/bi_$i.json"))
print("run $i value", d["value"], "ms/step", d["ms_per_step"], "calls", d["concurrent"].get("timed_region_call_ms"), "K2", d["concurrent"]["K"]["2"]["GB/s"], "K4", d["concurrent"]["K"]["4"]["GB/s"], "phases", [p for p, c in zip(d["concurrent"].get("timed_region_call_phases", []), d["concurrent"].get("timed_region_call_ms", [])) if c > 4.5])
PY
done
