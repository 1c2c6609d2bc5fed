"""development: k_pencil time against the length of the longest dependence path (r0 + r1 + r2 steps) for arrays of equal volume:
tells the step time of the wavefront with few / many tiles in flight.  usage: python tools/gpu_shape_scan.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sz_amd
from sz_amd.fields import s_field
assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
for shape in [(64, 64, 4096), (64, 64, 16384), (128, 128, 2048), (256, 256, 1024), (512, 512, 512), (1024, 1024, 128), (32, 32, 16384), (8, 8, 65536), (16, 24, 65536)]:
    d = s_field(*shape)
    ms = []
    for rep in range(4):
        s = sz_amd.SZ_compress_args(d, sz_amd.ABS, 1e-4); st = sz_amd.SZ_hip_last_stats(); ms.append(st.ms_quant)
    q = min(ms[1:]); steps = sum(shape) + 14
    tiles = -(-shape[0] // 32) * -(-shape[1] // 24)
    print(f"{shape}: k_pencil {q:.3f} ms, longest path {steps} steps -> {q * 1e3 / steps:.3f} us/step; {tiles} tiles, {d.nbytes / q / 1e6:.0f} GB/s")
