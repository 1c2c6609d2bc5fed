#!/usr/bin/env python3
"""development, build container only: random differential run of the ORACLE against the unmodified reference library itself
(oracle/_ref/libSZ.so, built from /root/reference by oracle/build_ref.sh): random shapes (1-D .. 4-D), types, bound modes,
sz.config knobs; streams compared byte for byte (byte 19 masked where the reference leaves it undefined; sign bytes of PW_REL streams
compared decoded), decoded arrays bit for bit.  Nothing of this travels; what it finds becomes a recorded case in tests/ref_cases.py.

    python tools/ref_diff_fuzz.py [cases] [seed] [product]
product: the PRODUCT code (host C + HIP layer on the CPU shim of tests/sim, reading the same sz.config file) instead of the oracle"""
import ctypes
import os, hashlib, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import ref_cases
import test_ref_recorded as T
from sz_amd.fields import l_field, s_field

L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libSZ.so"))
sz = ctypes.c_size_t
L.SZ_Init.argtypes = [ctypes.c_char_p]
L.SZ_compress_args.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(sz), ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double] + [sz] * 5
L.SZ_compress_args.restype = ctypes.c_void_p
L.SZ_decompress.argtypes = [ctypes.c_int, ctypes.c_void_p, sz] + [sz] * 5
L.SZ_decompress.restype = ctypes.c_void_p
libc = ctypes.CDLL(None); libc.free.argtypes = [ctypes.c_void_p]

def dims5(shape):
    d = list(shape)[::-1] + [0] * (5 - len(shape))
    return d[4], d[3], d[2], d[1], d[0]

product = len(sys.argv) > 3 and sys.argv[3] == "product"
if product:
    import sim_lib, sz_amd
    from sz_amd import api
    api._lib = api._bind(ctypes.CDLL(sim_lib.shim_path()))
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0; done = 0
with tempfile.TemporaryDirectory() as td:
    for c in range(ncases):
        rng = np.random.default_rng(seed0 * 7919 + c)
        dt = np.float32 if rng.random() < 0.55 else np.float64
        nd = int(rng.choice([1, 2, 3, 3, 3, 4]))
        if nd == 1: shape = (int(rng.integers(21, 6000)),)
        elif nd == 2: shape = (int(rng.integers(2, 70)), int(rng.integers(2, 110)))
        elif nd == 3: shape = tuple(int(x) for x in rng.integers(2, 72 if os.environ.get("FUZZ_BIG") else 30, size=3))
        else: shape = (int(rng.integers(2, 4)), int(rng.integers(2, 5)), int(rng.integers(2, 16)), int(rng.integers(2, 24)))
        n = int(np.prod(shape))
        if n <= 20: continue
        skip4 = nd == 4
        kind = int(rng.integers(0, 8))
        sh3 = (1, 1, n) if nd == 1 else (1,) + shape if nd == 2 else shape if nd == 3 else (shape[0] * shape[1], shape[2], shape[3])
        if kind == 0: d = s_field(*sh3, dt)
        elif kind == 1: d = l_field(*sh3, dt, n_for_hash=max(sh3[2], 8))
        elif kind == 2: d = rng.random(sh3).astype(dt)
        elif kind == 3: d = (s_field(*sh3, dt) + (rng.random(sh3) - 0.5).astype(dt) * dt(10.0 ** rng.integers(-5, -1)))
        elif kind == 4: d = (np.cumsum(rng.standard_normal(n)) * 0.01).astype(dt).reshape(sh3)
        elif kind == 5: d = s_field(*sh3, dt); d[np.abs(d) < 0.6] = 0                     # mostly one value: the mean shortcut
        elif kind == 6: d = np.full(sh3, dt(rng.standard_normal()), dt)                   # constant
        else: d = (s_field(*sh3, dt) * dt(1e-3) + dt(7.5))                                # small range on a large offset
        mode = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 10, 10]))
        if mode == 10:
            mag = np.exp(2.0 * d.astype(np.float64) / max(float(np.abs(d).max()), 1e-30) + 0.05 * rng.standard_normal(sh3))
            s = rng.random()
            if s < 0.4: pass
            elif s < 0.8: mag = mag * np.sign(d.astype(np.float64) + 1e-300)
            else: mag = -mag
            if rng.random() < 0.5: mag[rng.random(sh3) < 0.04] = 0.0
            if rng.random() < 0.2: mag.reshape(-1)[0] = 0.0
            d = mag.astype(dt)
        d = np.ascontiguousarray(d.reshape(shape))
        if os.environ.get("FUZZ_FILL") and mode != 10:            # fill values (FUZZ_FILL=2: NaN / inf too, never in element 0) at a random density
            d = d.copy(); f = d.reshape(-1)
            vals = [1e30, -1e30, 9.96921e36] + ([np.nan, np.inf, -np.inf] if os.environ["FUZZ_FILL"] == "2" else [])
            k = max(1, int(f.size * 10.0 ** rng.uniform(-3, -1.2)))
            f[rng.integers(1, f.size, k)] = dt(rng.choice(vals)) if rng.random() < 0.7 else np.asarray(rng.choice(vals, k), dt)
        wide = bool(os.environ.get("FUZZ_WIDE"))          # the corners of the knobs
        conf = {"withLinearRegression": "YES" if rng.random() < 0.6 else "NO",
                "quantization_intervals": int(rng.choice([0, 0, 0, 32, 64, 1024, 65536] if wide else [0, 0, 0, 64, 1024])),
                "sampleDistance": int(rng.choice([1, 2, 3, 5, 10, 37, 100, 500] if wide else [100, 100, 10, 37])),
                "predThreshold": float(rng.choice([0.5, 0.9, 0.97, 0.99, 0.999, 1.0] if wide else [0.99, 0.9, 0.999])),
                "max_quant_intervals": int(rng.choice([32, 64, 256, 4096, 65536] if wide else [65536, 65536, 4096])), "accelerate_pw_rel_compression": int(rng.random() < 0.6),
                "protectValueRange": "YES" if rng.random() < 0.15 else "NO", "psnr": float(rng.choice([60, 80])), "normErr": 0.05}
        if product and rng.random() < 0.35:        # the lossless back ends: bytes depend on the zstd / zlib build, so only sizes and DECODED values are compared -- both ways
            conf["szMode"] = str(rng.choice(["SZ_BEST_COMPRESSION", "SZ_DEFAULT_COMPRESSION"]))
            conf["losslessCompressor"] = str(rng.choice(["ZSTD_COMPRESSOR", "GZIP_COMPRESSOR"]))
        fin = d[np.isfinite(d) & (np.abs(d) < 1e29)]                                   # (FUZZ_FILL: the bound comes from the numbers, not from the fill values)
        rngv = max(float(fin.max()) - float(fin.min()), 1e-6) if fin.size else 1.0
        case = dict(name=f"fuzz{c}", data=None, mode=mode, abs=float(10.0 ** rng.uniform(-5, -2)) * rngv, rel=float(10.0 ** rng.uniform(-5, -2)),
                    pwr=float(10.0 ** rng.uniform(-4, -1)), conf=conf)
        if skip4 and conf["withLinearRegression"] == "NO" and mode < 10: continue      # SZ 1.4 for 4-D arrays: not restated
        cfg = os.path.join(td, "sz.config")
        ref_cases.write_config(cfg, conf)
        assert L.SZ_Init(cfg.encode()) == 0
        work = d.copy(); nn = sz(0)
        p = L.SZ_compress_args(0 if dt == np.float32 else 1, work.ctypes.data, ctypes.byref(nn), mode, case["abs"], case["rel"], case["pwr"], *dims5(shape))
        ref = ctypes.string_at(p, nn.value); libc.free(p)
        buf = ctypes.create_string_buffer(ref, len(ref))
        q = L.SZ_decompress(0 if dt == np.float32 else 1, buf, len(ref), *dims5(shape))
        rdec = np.ctypeslib.as_array(ctypes.cast(q, ctypes.POINTER(ctypes.c_float if dt == np.float32 else ctypes.c_double)), shape=(n,)).copy().reshape(shape)
        libc.free(q); L.SZ_Finalize()
        po = T._pwr_oracle_params(O, dict(conf=dict(ref_cases.BASE_CONF, **conf), pwr=case["pwr"]))
        po.segment_size = 0
        try:
            if product:
                assert sz_amd.SZ_Init(cfg) == 0
                try:
                    got = sz_amd.SZ_compress_args(d.copy(), mode, case["abs"], case["rel"], case["pwr"])
                    odec = sz_amd.SZ_decompress(ref, shape, d.dtype)
                finally:
                    sz_amd.SZ_Finalize()
            else:
                got, _ = O.compress(d, mode, case["abs"], case["rel"], params=po)
                odec = O.decompress(ref, shape, d.dtype)
        except Exception as e:
            bad += 1; print("EXC", c, np.dtype(dt).name, shape, mode, conf, repr(e)); continue
        wrapped = conf.get("szMode", "SZ_BEST_SPEED") != "SZ_BEST_SPEED"
        if mode == 10 and conf["accelerate_pw_rel_compression"] and nd >= 2 and conf.get("losslessCompressor") == "GZIP_COMPRESSOR" and (d < 0).any():
            # the reference codes these sign bytes with zlib and reads them back with zstd (sz_float_pwr.c:2030 / szd_float_pwr.c:1438): it cannot decode its own
            # stream (signs from an uninitialised buffer).  The product reads it (zlib fallback) -- checked against the INPUT instead
            x64 = d.astype(np.float64); nzm = x64 != 0; nzm.reshape(-1)[0] = False
            if odec is None or not np.array_equal(np.sign(odec.astype(np.float64)[nzm]), np.sign(x64[nzm])): bad += 1; print("SIGNS LOST", c)
            done += 1; continue
        if product:                                  # the reference decodes the product's stream
            assert L.SZ_Init(cfg.encode()) == 0
            buf2 = ctypes.create_string_buffer(got, len(got))
            q2 = L.SZ_decompress(0 if dt == np.float32 else 1, buf2, len(got), *dims5(shape))
            xdec = np.ctypeslib.as_array(ctypes.cast(q2, ctypes.POINTER(ctypes.c_float if dt == np.float32 else ctypes.c_double)), shape=(n,)).copy().reshape(shape) if q2 else None
            if q2: libc.free(q2)
            L.SZ_Finalize()
            if xdec is None or not np.array_equal(xdec.view(np.uint8), rdec.view(np.uint8)):
                if not (mode == 10 and ((got[3] & 0x10) != (ref[3] & 0x10))):
                    bad += 1; print("CROSS-DECODE DIFF", c, np.dtype(dt).name, shape, "mode", mode, conf)
        a, b = bytearray(got), bytearray(ref)
        if wrapped:
            done += 1
            okd = odec is not None and np.array_equal(odec.view(np.uint8), rdec.view(np.uint8))
            if not okd or abs(len(a) - len(b)) > 0.03 * len(b) + 64:
                bad += 1; print("DIFF(wrapped)", c, np.dtype(dt).name, shape, "mode", mode, len(got), len(ref), "decode", okd, conf)
            continue
        if len(a) > 19 and len(b) > 19 and not (b[3] & 0x80): a[19] = 0; b[19] = 0
        same = bytes(a) == bytes(b)
        if not same and mode == 10:
            pa, sa = T._pwr_parts(bytes(a), d.dtype, n); pb, sb = T._pwr_parts(bytes(b), d.dtype, n); same = pa == pb and sa == sb
        okd = odec is not None and np.array_equal(odec.view(np.uint8), rdec.view(np.uint8))
        if not same and mode == 10 and ((a[3] & 0x10) != (b[3] & 0x10)) and abs(len(a) - len(b)) <= 64:
            # the zstd-coded sign bytes are a property of the zstd build (system library here, the bundled one there): within a few bytes of the
            # raw-copy threshold their size tips the decision.  Both streams are valid; counted apart
            soft = globals().get("soft", 0) + 1; globals()["soft"] = soft; same = True
        done += 1
        if not (same and okd):
            bad += 1
            if os.environ.get("FUZZ_DUMP"):
                idx = [i for i in range(min(len(a), len(b))) if a[i] != b[i]]
                print("  first differing bytes:", idx[:16], "of", len(idx), "| data[0..3]", d.reshape(-1)[:3], "min|x|", float(np.abs(d[d != 0]).min()) if (d != 0).any() else None, "zeros", int((d == 0).sum()), "neg", int((d < 0).sum()))
                print("  oracle:", bytes(a[40:80]).hex()); print("  ref   :", bytes(b[40:80]).hex())
            print("DIFF", c, np.dtype(dt).name, shape, "mode", mode, "stream", same, len(got), len(ref), "decode", okd, conf, f"abs={case['abs']:.3e} rel={case['rel']:.3e} pwr={case['pwr']:.3e}")
print(f"ref-vs-{'product(shim)' if product else 'oracle'}: {done} cases, {bad} differences" + (f" ({globals()['soft']} raw-copy decisions tipped by the size of the zstd-coded sign bytes)" if globals().get("soft") else ""))
