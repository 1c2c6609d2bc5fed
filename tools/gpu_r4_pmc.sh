#!/bin/bash
# round 4: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes, counters + kernel trace only) and SQ counters of the OpenMP container's
# kernels and of the SZ 2.1 sweep, on one small driver script -> gpurun_out/${TAG}_pmc.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
T=${TAG:-r4pmc}
cat > /tmp/pmc_drv.py <<PY
import sys, ctypes, numpy as np, torch
sys.path.insert(0, "$R")
import sz_amd
from sz_amd.fields import s_field
n = 512
x = torch.from_numpy(s_field(n, n, n, np.float32)).to("cuda:0")
ctx = sz_amd.HipContext(0)
meta = bytes(32)
y = torch.empty_like(x)
for it in range(3):
    p, size, st = ctx.compress_omp(x.data_ptr(), True, (n, n, n), np.float32, 1e-4, 4096, meta, out_on_device=True)
    torch.cuda.synchronize()
    ctx.decompress_omp(p, True, size, len(meta), (n, n, n), np.float32, y.data_ptr(), True)
    torch.cuda.synchronize()
m2 = sz_amd.make_meta(np.float32, err_mode=sz_amd.ABS, abs_bound=1e-4, vmin=0.0, vmax=0.0)
ob = torch.empty(n * n * n * 2 + (1 << 20), dtype=torch.uint8, device="cuda:0")
for it in range(3):
    out = ctypes.c_void_p(ob.data_ptr()); nn = ctypes.c_size_t(ob.numel()); s2 = sz_amd.szhip_stats()
    prm = sz_amd.szhip_params(100, 0.99, 65536, 0, 1)
    rc = sz_amd.lib().szhip_compress(ctx._h, 0, x.data_ptr(), 1, n, n, n, 1e-4, ctypes.byref(prm), m2, len(m2), 2, ctypes.byref(out), ctypes.byref(nn), ctypes.byref(s2))
    assert rc == 0
    torch.cuda.synchronize()
PY
run() {  # name, counters...
  name=$1; shift
  rm -rf $O/pmc_$name
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $O/pmc_$name -o p --output-format csv -- python /tmp/pmc_drv.py > $O/${T}_$name.log 2>&1
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for d in ("fetch", "write", "sq1", "sq2"):
    for f in glob.glob("$O/pmc_" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if any(w in k for w in ("k_omp", "k_ribbon", "k_hist_u16", "k_encode", "k_permute", "k_sample", "k_fit_select")):
                agg[(k.split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("$O/${T}_pmc.csv", "w") as o:
    o.write("kernel,counter,launches,mean_value\n")
    for (k, c), v in sorted(agg.items()):
        o.write('"%s",%s,%d,%.1f\n' % (k, c, len(v), sum(v) / len(v)))
print(open("$O/${T}_pmc.csv").read())
PY
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2
