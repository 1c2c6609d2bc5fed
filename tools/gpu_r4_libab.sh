#!/bin/bash
# A/B of library builds (sz_amd/csrc/variants/*.so next to the product library) on one OpenMP-container compression + decompression at 512^3:
# per-kernel average durations of each.  LIBS="plain nt1" names variants; "." is the product library.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cat > /tmp/one_omp.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
import sz_amd
from sz_amd.fields import s_field
n = 512
x = torch.from_numpy(s_field(n, n, n, np.float32)).to("cuda:0")
ctx = sz_amd.HipContext(0)
meta = bytes(32)
y = torch.empty_like(x)
for it in range(5):
    p, size, st = ctx.compress_omp(x.data_ptr(), True, (n, n, n), np.float32, 1e-4, 4096, meta, out_on_device=True)
    torch.cuda.synchronize()
    ctx.decompress_omp(p, True, size, len(meta), (n, n, n), np.float32, y.data_ptr(), True)
    torch.cuda.synchronize()
print("size", size, "ms", st.ms_total, "equal", bool(((y - x).abs() <= 1e-4).all()))
PY
for L in ${LIBS:-.}; do
  rm -rf $O/tr
  if [ "$L" = "." ]; then unset SZ_AMD_LIB; else export SZ_AMD_LIB=$R/sz_amd/csrc/variants/libszhip_$L.so; fi
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/tr -o t --output-format csv -- python /tmp/one_omp.py > /tmp/log.txt 2>&1
  echo "== $L: $(tail -1 /tmp/log.txt)"
  python3 - <<PY
import csv, glob
for f in glob.glob("$O/tr/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "omp" in r["Name"] or "k_sample" in r["Name"]: print("   %-46s avg %7.1f us min %7.1f  x%s" % (r["Name"][:46], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Calls"]))
PY
done
rm -rf $O/tr
