#!/bin/bash
# round 6: SQ counters of the beam sweep alone (one tile / the whole array)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -I $GRAFT_REPO_ROOT/sz_amd/csrc"
/opt/rocm/bin/hipcc $F $UB_EXTRA -o /tmp/ub_beam $GRAFT_REPO_ROOT/tools/ubench/ub_beam.hip 2>&1 | grep -E "error" -A3
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_pmc_ubeam.txt; : > $OUT
for sh in "512 32 32" "512 512 512"; do
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_BUSY_CYCLES"; do
    rm -rf /tmp/pmc_out; rocprofv3 --pmc $set -d /tmp/pmc_out -o r --output-format csv -- /tmp/ub_beam $sh > /tmp/pmc_log.txt 2>&1
    echo "== $sh :: $set" >> $OUT; tail -1 /tmp/pmc_log.txt >> $OUT
    python3 - >> $OUT <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/pmc_out/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(list)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if 'k_beam' in r.get('Kernel_Name',''): acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, 'per launch', sum(v)/len(v), 'launches', len(v))
PY
  done
done
cat $OUT
