#!/bin/bash
# development: SQ counters of the fast-mode kernels (one pass per counter group; counters only) -> gpurun_out/fast_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/fk.py <<PY
import numpy as np, torch, sz_amd
from sz_amd.fields import s_field
n = 512
x = torch.from_numpy(s_field(n, n, n)).cuda()
ctx = sz_amd.HipContext(0)
ob = torch.empty(x.numel() * 2 + (1 << 20), dtype=torch.uint8, device="cuda")
for it in range(3):
    _, sz, st = ctx.compress_fast(x.data_ptr(), True, (n, n, n), np.float32, 1e-4, 0, ob.data_ptr(), ob.numel())
PY
: > $R/gpurun_out/fast_pmc.txt
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" ${PMC_EXTRA:+"$PMC_EXTRA"}; do
  rm -rf $R/gpurun_out/pmcx
  PYTHONPATH=$R rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/pmcx -o p --output-format csv -- python /tmp/fk.py > $R/gpurun_out/fast_pmc.log 2>&1
  f=$(find $R/gpurun_out/pmcx -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $R/gpurun_out/fast_pmc.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows: agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    if "k_fast_stat" in k or "k_fast_pack" in k or "k_fast_compact" in k:
        print("%-42s %-24s n=%d mean %.4g" % (k, c, len(v), sum(v) / len(v)))
PY
  rm -rf $R/gpurun_out/pmcx
done
cat $R/gpurun_out/fast_pmc.txt
