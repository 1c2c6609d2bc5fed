#!/usr/bin/env python3
"""Records outputs of the UNMODIFIED reference library for every case of tests/ref_cases.py -> tests/golden/ref_recorded.json
(+ the small streams themselves under tests/golden/ref_streams/).

Runs only in the build container: it needs a libSZ.so of the unmodified reference (szcompressor/sz 2.1.12.4).  The one used
for the committed file is the survey's build of /root/reference (the reference's own CMake, Release, gcc 11.4, x86-64, no
FMA): /tmp/szbuild/sz/libSZ.so -- the same build whose outputs SURVEY.md section 6 quotes.  This script neither builds the
reference nor copies any of its source; it calls its public C API (SZ_Init / SZ_compress_args / SZ_decompress) through ctypes
and writes down what comes back: stream length, md5 of the stream, md5 and max error of the decoded array.  Nothing here runs
on the GPU box; the tests read the JSON only.

    python tools/record_reference_outputs.py [--lib /tmp/szbuild/sz/libSZ.so]
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_cases  # noqa: E402


def dims5(shape):
    d = list(shape)[::-1] + [0] * (5 - len(shape))
    return d[4], d[3], d[2], d[1], d[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "oracle", "_ref", "libSZ.so"), help="the reference library: oracle/build_ref.sh builds it")
    ap.add_argument("--only", default=None)
    ap.add_argument("--check", action="store_true", help="do not write: compare what the library gives with tests/golden/ref_recorded.json")
    args = ap.parse_args()
    if not os.path.exists(args.lib):
        import subprocess
        subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "build_ref.sh")])
    L = ctypes.CDLL(args.lib)
    sz = ctypes.c_size_t
    L.SZ_Init.argtypes = [ctypes.c_char_p]
    L.SZ_compress_args.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(sz), ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                   ctypes.c_double] + [sz] * 5
    L.SZ_compress_args.restype = ctypes.c_void_p
    L.SZ_decompress.argtypes = [ctypes.c_int, ctypes.c_void_p, sz] + [sz] * 5
    L.SZ_decompress.restype = ctypes.c_void_p
    L.SZ_Finalize.argtypes = []
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]

    out_path = os.path.join(ROOT, "tests", "golden", "ref_recorded.json")
    sdir = os.path.join(ROOT, "tests", "golden", "ref_streams")
    os.makedirs(sdir, exist_ok=True)
    rec = {}
    if (args.only or args.check) and os.path.exists(out_path):
        rec = json.load(open(out_path))["cases"]
    old = dict(rec)
    diffs = []
    with tempfile.TemporaryDirectory() as td:
        for c in ref_cases.CASES:
            if args.only and args.only not in c["name"]:
                continue
            data = np.ascontiguousarray(c["data"]())
            cfg = os.path.join(td, "sz.config")
            ref_cases.write_config(cfg, c["conf"])
            assert L.SZ_Init(cfg.encode()) == 0
            dt = 0 if data.dtype == np.float32 else 1
            n = sz(0)
            work = data.copy()       # the reference's MSST19 path overwrites the zeros of its input (sz_float_pwr.c:2053-2058): it gets a copy
            p = L.SZ_compress_args(dt, work.ctypes.data, ctypes.byref(n), c["mode"], c["abs"], c["rel"], c["pwr"], *dims5(data.shape))
            assert p, c["name"]
            stream = ctypes.string_at(p, n.value)
            libc.free(p)
            wrapped = c["conf"].get("szMode", "SZ_BEST_SPEED") != "SZ_BEST_SPEED"
            masked = []
            if not wrapped and data.size > 20 and not (stream[3] & 0x80):
                # params byte 15 (stream byte 19) is never written by convertSZParamsToBytes (ByteToolkit.c:874-972); the SZ 1.4
                # container, the constant and the raw-copy streams are malloc'd, so that byte is whatever the heap held.  It is
                # zeroed here AFTER the reference decoded its own stream; the tests zero it too before comparing.
                stream = stream[:19] + b"\0" + stream[20:]
                masked = [19]
            r = dict(shape=list(data.shape), dtype=str(data.dtype), input_md5=hashlib.md5(data.tobytes()).hexdigest(),
                     stream_bytes=len(stream), stream_md5=hashlib.md5(stream).hexdigest(), flags=stream[3] if len(stream) > 3 else None,
                     masked_bytes=masked)
            if data.size > 20:   # the reference's decompressor exits on its own <= 20-value raw copies (it looks for a version header)
                buf = ctypes.create_string_buffer(stream, len(stream))
                q = L.SZ_decompress(dt, buf, len(stream), *dims5(data.shape))
                assert q, c["name"]
                dec = np.ctypeslib.as_array(ctypes.cast(q, ctypes.POINTER(ctypes.c_float if dt == 0 else ctypes.c_double)),
                                            shape=(data.size,)).copy().reshape(data.shape)
                libc.free(q)
                d64, x64 = dec.astype(np.float64), data.astype(np.float64)
                err = np.abs(d64 - x64)[np.isfinite(x64)]                # (cases with NaN: over the numbers)
                x64 = x64[np.isfinite(x64)]
                nz = x64 != 0
                r.update(decoded_md5=hashlib.md5(dec.tobytes()).hexdigest(), max_abs_err=float(err.max()),
                         max_rel_err=float((err[nz] / np.abs(x64[nz])).max()) if nz.any() else 0.0)
            L.SZ_Finalize()
            # point-wise-relative streams are kept too: their sign bytes went through the reference's bundled zstd, which the tests cannot redo
            if len(stream) <= 16384 or (wrapped or (len(stream) > 3 and stream[3] & 0x20)) and len(stream) <= 65536:
                if not args.check:
                    with open(os.path.join(sdir, c["name"] + ".sz"), "wb") as f:
                        f.write(stream)
                r["stream_file"] = "ref_streams/" + c["name"] + ".sz"
            if args.check:
                o = old.get(c["name"])
                # a zstd / zlib frame around the stream carries the reference's uninitialised parameter byte 15 inside the frame: sizes and
                # what decodes from it are compared there, not the frame's bytes
                keys = ["shape", "dtype", "input_md5", "stream_bytes", "decoded_md5", "flags"] + ([] if wrapped else ["stream_md5"])
                bad = [k for k in keys if o is None or o.get(k) != r.get(k)]
                if bad:
                    diffs.append((c["name"], bad))
                continue
            rec[c["name"]] = r
            print(f"{c['name']:34s} {len(stream):9d} B  flags {r['flags']}  max err {r.get('max_abs_err', -1):.4g}  max rel {r.get('max_rel_err', -1):.4g}")
    if args.check:
        print("checked", len(ref_cases.CASES) if not args.only else "some", "cases against", out_path, ":", "identical" if not diffs else diffs)
        sys.exit(1 if diffs else 0)
    prov = ("Outputs of the UNMODIFIED reference (szcompressor/sz 2.1.12.4, /root/reference compiled where it lies by oracle/build_ref.sh: "
            "one gcc -O3 line, baseline x86-64, vendored zstd and zlib; oracle/_ref/libSZ.so -- bit-identical to the reference's own CMake "
            "Release build they were first recorded from, tools/record_reference_outputs.py --check), recorded through its "
            "public C API by tools/record_reference_outputs.py.  Inputs and sz.config keys of each case: tests/ref_cases.py.")
    json.dump({"_provenance": prov, "cases": rec}, open(out_path, "w"), indent=1, sort_keys=True)
    print("wrote", out_path, len(rec), "cases")


if __name__ == "__main__":
    main()
