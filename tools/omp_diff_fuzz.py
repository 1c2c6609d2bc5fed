#!/usr/bin/env python3
"""Differential runs of the OpenMP container's HIP side against the oracle (oracle/szo_omp_impl.h): random box grids, shapes, types, bounds,
interval counts and data with zeros of both signs, denormals, NaN / inf, constant stretches, huge values.

    python tools/omp_diff_fuzz.py [cases] [seed]            # on the CPU shim (tests/sim; the product's .hip code, one workgroup at a time)
    SZ_FUZZ_GPU=1 python tools/omp_diff_fuzz.py ...         # through the built library on a GPU

Compares the stream byte for byte and the decoded array bit for bit."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def grid(threads):
    order = 0
    while (2 << order) <= threads:
        order += 1
    b = order // 3
    nx, ny = ((1 << b, 1 << b), (1 << (b + 1), 1 << b), (1 << (b + 1), 1 << (b + 1)))[order % 3]
    return nx, ny, threads // (nx * ny)


def make(rng, shape, dt):
    kind = rng.integers(0, 6)
    z, y, x = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in shape], indexing="ij")
    d = np.sin(0.11 * x + 0.07 * y) * np.cos(0.05 * z + 0.02 * x) + 0.3 * np.sin(0.013 * x * y / (1 + z))
    if kind == 1:
        d += rng.standard_normal(shape) * 10.0 ** rng.integers(-5, 1)
    elif kind == 2:
        d = rng.standard_normal(shape) * 10.0 ** rng.integers(-3, 6)
    elif kind == 3:
        d = np.round(d * 4) / 4                                   # plateaus: many exact zeros of the difference
    elif kind == 4:
        d *= 1e-40 if dt == np.float32 else 1e-310                # denormals
    d = d.astype(dt)
    f = d.ravel()
    for _ in range(int(rng.integers(0, 4))):
        a = int(rng.integers(0, f.size)); n = int(rng.integers(1, 40))
        f[a:a + n] = rng.choice([0.0, -0.0, 1.0, -1e30, 1e30, np.nan, np.inf, -np.inf, 1e-45])
    return d


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import oracle_lib as O
    import sz_amd
    from sz_amd import api
    if not os.environ.get("SZ_FUZZ_GPU"):
        import sim_lib
        api._lib = api._bind(ctypes.CDLL(sim_lib.shim_path()))
    ctx = sz_amd.HipContext(0)
    rng = np.random.default_rng(seed)
    meta = bytes(range(40, 72))
    bad = 0
    for c in range(cases):
        threads = int(rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64, 100, 128]))
        nx, ny, nz = grid(threads)
        c0, c1, c2 = int(rng.integers(1, 9)), int(rng.integers(1, 9)), int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 20, 33]))
        if rng.integers(0, 3) == 0:
            c0, c1 = int(rng.integers(8, 33)), int(rng.integers(8, 33))
            if c0 * c1 > 1024:
                c1 = 1024 // c0
        if os.environ.get("FUZZ_COL") or rng.integers(0, 4) == 0:
            # boxes with 32 x 32 faces, in pairs: the column-per-lane sweep, per-box entropy stage and look-up-table decoder of round 4 (szh_ompcol.h)
            threads = int(rng.choice([2, 4, 8, 16, 32]))
            nx, ny, nz = grid(threads)
            c0, c1, c2 = int(rng.choice([1, 2, 3, 4, 5, 8, 13])), 32, 32
        shape = (nx * c0, ny * c1, nz * c2)
        if shape[0] * shape[1] * shape[2] > (600000 if os.environ.get("SZ_FUZZ_GPU") else 300000):
            continue
        dt = np.float32 if rng.integers(0, 2) else np.float64
        d = make(rng, shape, dt)
        eb = float(10.0 ** rng.uniform(-6, 0)) * (1e-38 if (dt == np.float32 and rng.integers(0, 12) == 0) else 1.0)
        iv = int(rng.choice([0, 0, 4, 32, 256, 1024, 65536]))
        if os.environ.get("SZ_FUZZ_ONLY") and int(os.environ["SZ_FUZZ_ONLY"]) != c:
            continue
        if os.environ.get("SZ_FUZZ_DUMP"):
            np.save(os.environ["SZ_FUZZ_DUMP"], d)
        p = O.default_params(); p.quantization_intervals = iv
        ref = O.omp_compress(d, eb, threads, meta, p)
        try:
            got, n, st = ctx.compress_omp(d.ctypes.data, False, d.shape, d.dtype, eb, threads, meta, api.szhip_params(100, 0.99, 65536, iv))
        except Exception as e:                                   # noqa: BLE001
            print("case", c, shape, threads, dt.__name__, eb, iv, "FAILED", e); bad += 1; continue
        ok = got == ref
        out = np.empty_like(d)
        buf = ctypes.create_string_buffer(ref, len(ref))
        ctx.decompress_omp(ctypes.addressof(buf), False, len(ref), len(meta), d.shape, d.dtype, out.ctypes.data, False)
        want = O.omp_decompress(ref, len(meta), d.shape, d.dtype)
        iview = np.uint32 if dt == np.float32 else np.uint64
        okd = np.array_equal(out.view(iview), want.view(iview))
        if not (ok and okd):
            bad += 1
            first = next((i for i in range(min(len(got), len(ref))) if got[i] != ref[i]), -1)
            print("case", c, shape, threads, dt.__name__, eb, iv, "stream", ok, "decoded", okd, "| lengths", len(got), len(ref), "first difference at", first,
                  "| intervals", int.from_bytes(got[36 + d.itemsize:40 + d.itemsize], "big"), int.from_bytes(ref[36 + d.itemsize:40 + d.itemsize], "big"),
                  "| non-finite", int((~np.isfinite(d)).sum()))
    print(f"{cases} cases, seed {seed}: {bad} differing")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
