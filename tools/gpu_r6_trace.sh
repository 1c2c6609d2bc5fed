#!/bin/bash
# round 6: device time line of single calls (rocprofv3 --kernel-trace --memory-copy-trace is refused with PMC only; plain trace here):
# every kernel of the last compress / decompress call of tools/gpu_r5_mtime.py with its start and end relative to the call's first kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; field=${1:-s}; tag=${2:-trace}
rm -rf $R/gpurun_out/tr_$tag
rocprofv3 --kernel-trace -d $R/gpurun_out/tr_$tag -o t --output-format csv -- ${TRACE_CMD:-python $R/tools/gpu_r5_mtime.py 512 $field} > $R/gpurun_out/r6_${tag}_log.txt 2>&1
f=$(find $R/gpurun_out/tr_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $R/gpurun_out/r6_${tag}_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "at::native" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# calls: a gap of > 300 us between kernels separates them
calls, cur, last_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and s - last_end > 300000 and cur: calls.append(cur); cur = []
    cur.append(r); last_end = max(last_end or 0, e)
if cur: calls.append(cur)
def show(c, title):
    t0 = int(c[0]["Start_Timestamp"])
    print("==", title, "kernels", len(c), "span %.3f ms" % ((max(int(r["End_Timestamp"]) for r in c) - t0) / 1e6))
    for r in c:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print("%8.3f %8.3f %7.3f  q%-3s %s grid %s wg %s" % (s / 1e6, e / 1e6, (e - s) / 1e6, r.get("Queue_Id", "?"), r["Kernel_Name"][:60], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")))
big = [c for c in calls if len(c) >= 8]
names = lambda c: " ".join(r["Kernel_Name"] for r in c)
comp = [c for c in big if "k_fit" in names(c)]
dec = [c for c in big if "k_hdec_write" in names(c)]
if comp: show(comp[-1], "last compress call")
if dec: show(dec[-1], "last decompress call")
PY
rm -rf $R/gpurun_out/tr_$tag
grep field $R/gpurun_out/r6_${tag}_log.txt | tail -6
cat $R/gpurun_out/r6_${tag}_timeline.txt | head -120
