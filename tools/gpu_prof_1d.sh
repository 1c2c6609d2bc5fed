#!/bin/bash
# per-kernel time of the 1-D path (rocprofv3 --kernel-trace --stats): 4 M values, f32 and f64, segmented walk
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof1d
ONLY_SEG=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof1d -o oned --output-format csv -- python $R/tools/gpu_1d_time.py 4000000 > $R/gpurun_out/prof1d.log 2>&1
cat $R/gpurun_out/prof1d.log | tail -3
cp $(find $R/gpurun_out/prof1d -name "*kernel_stats.csv" | head -1) $R/gpurun_out/prof1d_kernel_stats.csv
rm -f $(find $R/gpurun_out/prof1d -name "*kernel_trace.csv")
cut -c1-160 $R/gpurun_out/prof1d_kernel_stats.csv | head -30
