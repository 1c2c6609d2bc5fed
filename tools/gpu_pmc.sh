#!/bin/bash
# development: PMC passes over the bench (counters only; no sys-trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|Counter_Name)?\s*:?\s*(SQ_[A-Z_0-9]+|TCP_[A-Z_0-9a-z]+|TA_[A-Z_0-9a-z]+|TCC_[A-Z_0-9a-z]+)" | awk '{print $NF}' | sort -u > $R/gpurun_out/counters.txt
wc -l $R/gpurun_out/counters.txt
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/$name -o $name --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$name.log 2>&1; }
run pmc_sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
run pmc_tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
ls -R $R/gpurun_out/pmc_sq | head -20
