#!/bin/bash
# round 5: per-kernel times (rocprofv3 --kernel-trace --stats) of a command; the summary goes to gpurun_out/r5_kstats_<tag>.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
rm -rf $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o $tag --output-format csv -- "$@" > $R/gpurun_out/r5_prof_$tag.log 2>&1
cp $(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r5_kstats_$tag.csv
rm -rf $R/gpurun_out/prof_$tag
head -30 $R/gpurun_out/r5_kstats_$tag.csv | cut -c1-160
