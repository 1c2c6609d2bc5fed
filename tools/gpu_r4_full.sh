#!/bin/bash
# round 4: the whole GPU suite, then the bench line (default flags + the OpenMP container) and its kernel stats  (TAG=name)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${TAG:-r4x}
timeout 900 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_tests.log 2>&1; echo "tests exit $?" >> $O/${T}_gpu_tests.log; tail -4 $O/${T}_gpu_tests.log
timeout 600 python bench.py --omp-boxes 4096 ${BENCH_FLAGS} > $O/${T}_bench.log 2>&1
grep '^{"metric"' $O/${T}_bench.log | tail -1 > $O/${T}_bench.json
python3 - <<PY
import json
d = json.load(open("$O/${T}_bench.json"))
o = d.get("omp_container")
print({k: d[k] for k in ("value", "ms_per_step")}, d.get("roofline"))
print("phase", d.get("phase_ms"))
print("m_field", {k: d["m_field"][k] for k in ("GB/s", "ms")} if d.get("m_field") else None)
print("decompress", d.get("decompress_GBps"), d.get("concurrent"))
if o: print(json.dumps({k: o[k] for k in ("GB/s", "ms", "decompress_GBps", "out_bytes", "max_abs_err", "phase_ms")}))
PY
cd /tmp
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python $R/bench.py --omp-boxes 4096 --no-cpu-baseline --no-fast > $O/${T}_prof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats.csv 2>/dev/null
rm -rf $O/prof
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/${T}_kernel_stats.csv")):
    n = r['Name']
    if float(r['AverageNs']) > 20000 or 'k_omp' in n:
        print("%-56.56s calls %4s avg %9.1f us  min %9.1f" % (n, r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
