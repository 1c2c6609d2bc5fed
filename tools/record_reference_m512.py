#!/usr/bin/env python3
"""Records what the UNMODIFIED reference (oracle/_ref/libSZ.so, built by oracle/build_ref.sh from /root/reference where it lies) gives for BASELINE
configs[2] at full size -- the 512^3 M-field, ABS 1e-4, SZ_BEST_SPEED -- into tests/golden/anchors.json: stream length and md5, regression blocks,
PSNR (formula of example/sz.c:598-607), max error.  Runs only in the build container (~15 s of CPU, 1 GiB of memory); the GPU test
tests/test_beam_gpu.py::test_m_field_512_full_size reads the JSON only.

    python tools/record_reference_m512.py [--check]
"""
import argparse, ctypes, hashlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sz_amd.fields import m_field  # noqa: E402

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--check", action="store_true"); ap.add_argument("--edge", type=int, default=512)
    args = ap.parse_args()
    lib = os.path.join(ROOT, "oracle", "_ref", "libSZ.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "build_ref.sh")])
    L = ctypes.CDLL(lib); sz = ctypes.c_size_t
    L.SZ_Init.argtypes = [ctypes.c_char_p]
    L.SZ_compress_args.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(sz), ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double] + [sz] * 5
    L.SZ_compress_args.restype = ctypes.c_void_p
    L.SZ_decompress.argtypes = [ctypes.c_int, ctypes.c_void_p, sz] + [sz] * 5
    L.SZ_decompress.restype = ctypes.c_void_p
    n = args.edge
    d = np.ascontiguousarray(m_field(n))
    assert L.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config").encode()) == 0
    out = sz(0)
    p = L.SZ_compress_args(0, d.ctypes.data, ctypes.byref(out), 0, 1e-4, 0.0, 0.0, 0, 0, n, n, n)
    stream = ctypes.string_at(p, out.value)
    q = L.SZ_decompress(0, p, out.value, 0, 0, n, n, n)
    dec = np.ctypeslib.as_array(ctypes.cast(q, ctypes.POINTER(ctypes.c_float)), shape=(n, n, n)).copy()
    ind_off = None
    # the indicator bits follow the code book: count regression blocks through the product's own parser instead (host C, no GPU needed)
    dd = d.astype(np.float64); ee = dec.astype(np.float64)
    rng = float(dd.max() - dd.min()); mse = float(((dd - ee) ** 2).mean())
    masked = bytearray(stream); masked[19] = 0        # parameter byte 15 is never written by convertSZParamsToBytes (ByteToolkit.c:874-972): heap garbage
    rec = {"stream_bytes": out.value, "md5_byte19_zeroed": hashlib.md5(bytes(masked)).hexdigest(), "decoded_md5": hashlib.md5(dec.tobytes()).hexdigest(),
           "psnr": round(float(20 * np.log10(rng) - 10 * np.log10(mse)), 6), "max_abs_err": float(np.abs(dd - ee).max()), "blocks": (n // 6) ** 3,
           "_source": "tools/record_reference_m512.py on oracle/_ref/libSZ.so (the unmodified reference built from /root/reference where it lies)"}
    key = "M%d_f32_abs1e-4_best_speed" % n
    path = os.path.join(ROOT, "tests", "golden", "anchors.json")
    A = json.load(open(path))
    if args.check:
        old = {k: A[key][k] for k in rec if k in A[key] and not k.startswith("_")}
        new = {k: rec[k] for k in old}
        print("identical" if old == new else "DIFFERENT: %r vs %r" % (old, new)); return 0 if old == new else 1
    if key in A: rec = {**A[key], **rec}
    A[key] = rec
    json.dump(A, open(path, "w"), indent=1); open(path, "a").write("\n")
    print(key, rec)

if __name__ == "__main__":
    sys.exit(main())
