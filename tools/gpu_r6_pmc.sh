#!/bin/bash
# round 6: SQ counters of one kernel of the compress call (name pattern $1), a few calls of tools/gpu_r6_calls.py; counters only (no trace domains)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
PAT=${1:-k_col_encode}; OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_pmc_${2:-kernel}.txt; : > $OUT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_LDS_ATOMIC_RETURN SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc_out; NCALLS=2 rocprofv3 --pmc $set -d /tmp/pmc_out -o r --output-format csv -- python $GRAFT_REPO_ROOT/tools/gpu_r6_calls.py > /tmp/pmc_log.txt 2>&1
  echo "== $set" >> $OUT
  PAT=$PAT python3 - >> $OUT <<'PY'
import csv,glob,collections,os
f=glob.glob('/tmp/pmc_out/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(list)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if os.environ['PAT'] in r.get('Kernel_Name',''): acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k, 'per launch %.0f' % (sum(v)/len(v)), 'launches', len(v))
PY
done
cat $OUT
