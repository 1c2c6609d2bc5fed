#!/bin/bash
# per-kernel time of the bench command (rocprofv3 --kernel-trace --stats); summary copied to gpurun_out/prof_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
tail -1 $R/gpurun_out/prof_bench.log | cut -c1-400
find $R/gpurun_out/prof -name "*kernel_stats*" | head
cp $(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/prof_kernel_stats.csv
rm -f $(find $R/gpurun_out/prof -name "*kernel_trace.csv")
cat $R/gpurun_out/prof_kernel_stats.csv | cut -c1-200
