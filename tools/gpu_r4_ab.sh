#!/bin/bash
# A/B of environment switches on the OpenMP container object of the bench: VARIANTS="NAME=VAL;NAME2=VAL2 ..." (space separated sets)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${TAG:-r4ab}
i=0
for V in ${VARIANTS:-"X=0"}; do
  i=$((i+1))
  env $(echo $V | tr ';' ' ') timeout 300 python bench.py --omp-boxes 4096 --no-cpu-baseline --no-m-field --no-fast --steps 4 --warmup 2 > $O/${T}_$i.log 2>&1
  grep '^{"metric"' $O/${T}_$i.log | tail -1 > $O/${T}_$i.json
  python3 - <<PY
import json
d = json.load(open("$O/${T}_$i.json")); o = d["omp_container"]
print("$V", o["GB/s"], o["ms"], "dec", o["decompress_GBps"], o["phase_ms"])
PY
done
