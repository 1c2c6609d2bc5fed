"""development: one recorded fuzz case, GPU stream against the oracle stream, field by field"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import sz_amd
assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
d = np.load(os.path.join(ROOT, "tools", "case_1d.npy"))
abs_b = 2.58783308225251e-05
for serial in ("0", "1"):
    os.environ["SZ_HIP_1D_SERIAL"] = serial
    for wr in (1, 0):
        sz_amd.conf_params().withRegression = wr
        ref, st = O.compress(d, 0, abs_b, 0.0, params=O.default_params(with_regression=wr), want_stages=True)
        got = sz_amd.SZ_compress_args(d, 0, abs_b, 0.0)
        diff = [i for i in range(min(len(ref), len(got))) if ref[i] != got[i]]
        print(f"serial={serial} with_regression={wr}: len {len(ref)}/{len(got)} differing bytes {len(diff)} first {diff[:12]}", flush=True)
        if diff:
            dec = O.decompress(got, d.shape, d.dtype)
            want = O.decompress(ref, d.shape, d.dtype)
            bad = np.nonzero(dec.view(np.uint32) != want.view(np.uint32))[0]
            print("   decoded values differing:", bad[:10], "ref bytes", ref[diff[0] - 2:diff[0] + 6].hex(), "gpu bytes", got[diff[0] - 2:diff[0] + 6].hex())
