#!/bin/bash
# round 6: the C4 slab (132 x 1024 x 1024 float64, regression blocks) after the double sweep's prefetch distance went from 9 lines to 3: beam tests, per-call phases, per-kernel averages
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_c4_pd3.txt; : > $out
timeout 900 python -m pytest tests/test_beam_gpu.py tests/test_colenc.py -m gpu -x -q 2>&1 | tail -3 >> $out
FIELD=c4 NCALLS=12 TAG=c4 python tools/gpu_r6_calls.py 2>&1 | tail -1 >> $out
FIELD=c4 NCALLS=10 bash tools/gpu_r6_stats.sh c4pd3 12 >> $out 2>&1
python bench.py --config c4 --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r6_bench_c4_pd3.json
cat gpurun_out/r6_bench_c4_pd3.json | cut -c1-400 >> $out
cat $out
