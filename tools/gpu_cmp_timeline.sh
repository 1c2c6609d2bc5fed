#!/bin/bash
# development: kernel + copy timeline of ONE compression of the 512^3 headline array, one lane (rocprofv3 traces), gaps included
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tl
cat > /tmp/dt.py <<PY
import numpy as np, torch, sz_amd
from sz_amd.fields import s_field
n = 512
x = torch.from_numpy(s_field(n, n, n)).cuda()
ctx = sz_amd.HipContext(0)
ob = torch.empty(x.numel() * 2 + (1 << 20), dtype=torch.uint8, device="cuda")
meta = sz_amd.make_meta(np.float32, err_mode=sz_amd.ABS, abs_bound=1e-4, vmin=0.0, vmax=0.0)
prm = sz_amd.szhip_params(100, 0.99, 65536, 0, 1)
import ctypes
out = ctypes.c_void_p(ob.data_ptr()); nn = ctypes.c_size_t(ob.numel()); st = sz_amd.szhip_stats()
import time
for it in range(5):
    out = ctypes.c_void_p(ob.data_ptr()); nn = ctypes.c_size_t(ob.numel())
    torch.cuda.synchronize(); t = time.perf_counter()
    rc = sz_amd.lib().szhip_compress(ctx._h, 0, x.data_ptr(), 1, n, n, n, 1e-4, ctypes.byref(prm), meta, len(meta), 2, ctypes.byref(out), ctypes.byref(nn), ctypes.byref(st))
    torch.cuda.synchronize(); print("compress %.3f ms (prequant %.3f quant %.3f entropy %.3f)" % ((time.perf_counter() - t) * 1e3, st.ms_prequant, st.ms_quant, st.ms_entropy))
assert rc == 0
size = nn.value
dec = torch.empty_like(x)
import time
for it in range(0):
    torch.cuda.synchronize(); t = time.perf_counter()
    d = ctx.decompress(ob.data_ptr(), True, size, 4 + 28 + 8, (n, n, n), np.float32, dec.data_ptr(), True)
    torch.cuda.synchronize(); print("decompress %.3f ms (entropy %.3f quant %.3f)" % ((time.perf_counter() - t) * 1e3, d.ms_entropy, d.ms_quant))
PY
PYTHONPATH=$R rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/tl -o tl --output-format csv -- python /tmp/dt.py > $R/gpurun_out/tl_dec.log 2>&1
grep "^compress" $R/gpurun_out/tl_dec.log
python3 - <<PY
import csv, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
ev=[]
for f in glob.glob(R+"/gpurun_out/tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:56]))
for f in glob.glob(R+"/gpurun_out/tl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY "+r.get("Direction","")+" "+r.get("Bytes", r.get("Size",""))))
ev.sort()
di=[i for i,e in enumerate(ev) if "k_ribbon" in e[2]]
st=di[-1]
while st > 0 and "k_encode" not in ev[st-1][2]: st -= 1
t0=ev[st][0]; prev=t0
out=open(R+"/gpurun_out/timeline_cmp.txt","w")
for s,e,n in ev[st:]:
    line="%9.1f us  +gap %7.1f  dur %8.1f  %s"%((s-t0)/1e3,(s-prev)/1e3,(e-s)/1e3,n)
    print(line); out.write(line+"\n"); prev=e
PY
rm -rf $R/gpurun_out/tl
