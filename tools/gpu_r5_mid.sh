#!/bin/bash
# round 5: the default bench line, the C4 line, and per-kernel times of one S / M call both ways
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py 2> gpurun_out/r5_bench_default.err | grep '^{"metric"' | tail -1 > gpurun_out/r5_bench_default.json
python bench.py --config c4 2> gpurun_out/r5_bench_c4.err | grep '^{"metric"' | tail -1 > gpurun_out/r5_bench_c4.json
bash tools/gpu_r5_prof.sh mtime python $GRAFT_REPO_ROOT/tools/gpu_r5_mtime.py 512 s,m > /dev/null 2>&1
cut -c1-1500 gpurun_out/r5_bench_default.json; echo; cut -c1-600 gpurun_out/r5_bench_c4.json; echo; head -40 gpurun_out/r5_kstats_mtime.csv | cut -c1-150
