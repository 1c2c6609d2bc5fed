#!/bin/bash
# round 6: per-kernel averages (rocprofv3 --kernel-trace --stats) of tools/gpu_r6_calls.py under the environment given; summary -> gpurun_out/r6_kstats_<tag>.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=${1:-x}
rm -rf $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o $tag --output-format csv -- python $R/tools/gpu_r6_calls.py > $R/gpurun_out/r6_prof_$tag.log 2>&1
cp $(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r6_kstats_$tag.csv
rm -rf $R/gpurun_out/prof_$tag
head -${2:-25} $R/gpurun_out/r6_kstats_$tag.csv | cut -c1-150
