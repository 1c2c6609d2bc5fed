#!/bin/bash
# development: kernel + memcpy timeline of ONE compress and ONE decompress step (rocprofv3 traces), gaps included
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tl
rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/tl -o tl --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-m-field --no-fast > $R/gpurun_out/tl_bench.log 2>&1
tail -1 $R/gpurun_out/tl_bench.log | cut -c1-300
python3 - <<PY
import csv, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
ev=[]
for f in glob.glob(R+"/gpurun_out/tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for f in glob.glob(R+"/gpurun_out/tl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY "+r.get("Direction","")+" "+r.get("Bytes", r.get("Size",""))))
ev.sort()
# last compress step = the last k_pencil<float,false> ... find index of last gather_mean before it
idx=[i for i,e in enumerate(ev) if "k_gather_mean" in e[2]]
st=idx[3]                          # warmup 2 + steps 3: the fourth headline step
while st > 0 and "k_minmax" not in ev[st][2]: st -= 1
t0=ev[st][0]; prev=t0
out=open(R+"/gpurun_out/timeline.txt","w")
for s,e,n in ev[st:st+48]:
    line="%9.1f us  +gap %7.1f  dur %8.1f  %s"%((s-t0)/1e3,(s-prev)/1e3,(e-s)/1e3,n)
    print(line); out.write(line+"\n"); prev=e
# one decompress step: from the last k_hdec_init back to the copy before it, through the k_pencil<float, true> that follows
di=[i for i,e in enumerate(ev) if "k_hdec_init" in e[2]]
if di:
    st=di[-1]-3; t0=ev[st][0]; prev=t0
    out.write("---- decompress\n"); print("---- decompress")
    for s,e,n in ev[st:st+40]:
        line="%9.1f us  +gap %7.1f  dur %8.1f  %s"%((s-t0)/1e3,(s-prev)/1e3,(e-s)/1e3,n)
        print(line); out.write(line+"\n"); prev=e
PY
rm -rf $R/gpurun_out/tl
