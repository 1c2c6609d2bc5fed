#!/bin/bash
# development: SQ counter pass over the bench (counters only; no sys-trace) -> gpurun_out/pmc_sq.csv (k_pencil rows)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_sq
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace -d $R/gpurun_out/pmc_sq -o pmc_sq --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-m-field --no-fast > $R/gpurun_out/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/gpurun_out/pmc_sq2 -o pmc_sq2 --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-m-field --no-fast > $R/gpurun_out/pmc_sq2.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(list)
for d in ("pmc_sq", "pmc_sq2"):
    for f in glob.glob(R + "/gpurun_out/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if os.environ.get("PMC_KERNEL", "k_pencil") in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:34], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(R + "/gpurun_out/pmc_sq.csv", "w") as o:
    o.write("kernel,counter,launches,mean_value\n")
    for (k, c), v in sorted(agg.items()):
        o.write('"%s",%s,%d,%.1f\n' % (k, c, len(v), sum(v) / len(v)))
print(open(R + "/gpurun_out/pmc_sq.csv").read())
PY
rm -rf $R/gpurun_out/pmc_sq $R/gpurun_out/pmc_sq2
