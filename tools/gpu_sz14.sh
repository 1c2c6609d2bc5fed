#!/bin/bash
# development: SZ 1.4 path on the GPU -- parity tests, fuzz, timing at 512^3
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sz14" 2>&1 | grep -vE "^\[SZ\]|^Error" | tail -15
timeout 300 python tools/gpu_fuzz.py 600 31 sz14 2>&1 | grep -vE "^\[SZ\]" | tail -8
timeout 300 python - <<'PY' 2>&1 | grep -vE "^\[SZ\]" | tail -8
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import sz_amd
from sz_amd.fields import s_field
assert sz_amd.SZ_Init("tests/golden/sz_speed.config") == 0
sz_amd.conf_params().withRegression = 0
d = s_field(512, 512, 512)
for rep in range(3):
    s = sz_amd.SZ_compress_args(d, sz_amd.ABS, 1e-4); st = sz_amd.SZ_hip_last_stats()
    print(f"sz14 compress 512^3: size {len(s)} prequant {st.ms_prequant:.2f} quant {st.ms_quant:.2f} entropy {st.ms_entropy:.2f} host {st.ms_host:.2f} ms")
    o = sz_amd.SZ_decompress(s, d.shape, d.dtype); st = sz_amd.SZ_hip_last_stats()
    print(f"sz14 decompress: entropy {st.ms_entropy:.2f} quant {st.ms_quant:.2f} host {st.ms_host:.2f} ms  maxerr {float(np.abs(o-d).max()):.6g}")
PY
