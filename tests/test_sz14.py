"""SZ 1.4 path (withLinearRegression = NO; sz_float.c:946, TightDataPointStorageF.c, szd_float.c:600): CPU-side checks.
The oracle's restatement is pinned to a recorded output of the reference in tests/test_oracle_pins.py (512^3 S-field: exact stream
size, exact-value count, PSNR).  Here: round trips and container structure of the oracle, and the product's kernels +
orchestration against the oracle on the HIP-on-CPU shim (the GPU runs are in test_gpu_parity.py)."""
import ctypes
import os
import struct

import numpy as np
import pytest

import sim_lib
from sz_amd.fields import plane_field, s_field


def _walk(n, dtype, seed=6):
    """a 1-D series: random walk with a few jumps (segments of the chain) and a run of zeros"""
    rng = np.random.default_rng(seed)
    x = np.cumsum(rng.standard_normal(n)) * 0.01
    x[rng.integers(0, n, 12)] += 20.0
    x[n // 2:n // 2 + 40] = 0.0
    return np.ascontiguousarray(x.astype(dtype))


def _noisy(shape, dtype, amp, seed=5):
    rng = np.random.default_rng(seed)
    return s_field(*shape, dtype) + (rng.random(shape).astype(dtype) - dtype(0.5)) * dtype(amp)


@pytest.mark.parametrize("shape,dtype,eb", [((20, 30, 40), np.float32, 1e-4), ((5, 7, 9), np.float32, 1e-3), ((2, 3, 50), np.float64, 1e-2),
                                            ((33, 20, 17), np.float64, 1e-5), ((24, 24, 24), np.float32, 1e-7)])
def test_oracle_sz14_round_trip_and_container(oracle, shape, dtype, eb):
    d = _noisy(shape, dtype, 3e-4)
    p = oracle.default_params(with_regression=0)
    stream, st = oracle.compress(d, oracle.ABS, eb, params=p, want_stages=True)
    dec = oracle.decompress(stream, shape, dtype)
    assert float(np.abs(dec.astype(np.float64) - d.astype(np.float64)).max()) <= eb
    if stream[3] & 0x10:   # raw fallback
        return
    esz = np.dtype(dtype).itemsize
    meta = 28 if dtype == np.float32 else 36
    q = 4 + meta
    assert not stream[3] & 0x80 and stream[3] & 0x40                       # SZ 1.4 container, 8-byte sizes
    n, = struct.unpack(">Q", stream[q:q + 8]); q += 8
    maxq, intervals = struct.unpack(">II", stream[q:q + 8]); q += 8
    q += esz                                                                # median
    req = stream[q]; q += 1
    ebs, = struct.unpack(">d", stream[q:q + 8]); q += 8
    type_size, exact_n, mid_n = struct.unpack(">QQQ", stream[q:q + 24]); q += 24
    assert n == d.size and maxq == 65536 and intervals == st["intervals"] and req == st["req_length"]
    assert ebs == float(dtype(eb)) and exact_n == st["exact_count"] == int((st["codes"] == 0).sum()) and mid_n == st["mid"].size
    resi = req % 8
    assert len(stream) == q + type_size + (exact_n * 2 + 7) // 8 + mid_n + ((exact_n * resi + 7) // 8 if resi else 0)
    assert st["codes"][0] == 0                                              # the first value is always stored exactly
    assert int(st["lead"].max()) <= 3 and mid_n == int(np.maximum(req // 8 - st["lead"].astype(np.int64), 0).sum())


@pytest.mark.parametrize("shape,dtype,eb", [((200, 300), np.float32, 1e-4), ((37, 45), np.float64, 1e-5), ((2, 500), np.float32, 1e-3)])
def test_oracle_sz14_2d_round_trip(oracle, shape, dtype, eb):
    """2-D arrays on the SZ 1.4 path (SZ_compress_float_2D_MDQ, sz_float.c:610): pinned by the recorded sz14-2D-* cases (tests/test_ref_recorded.py);
    the restatement shares everything but the optimiser's lattice with the 3-D one."""
    d = plane_field(*shape, dtype)
    stream, st = oracle.compress(d, oracle.ABS, eb, params=oracle.default_params(with_regression=0), want_stages=True)
    dec = oracle.decompress(stream, shape, dtype)
    assert float(np.abs(dec.astype(np.float64) - d.astype(np.float64)).max()) <= eb
    assert st is None or (st["codes"][0] == 0 and st["exact_count"] == int((st["codes"] == 0).sum()))


@pytest.mark.slow
def test_sz14_hip_layer_on_cpu_shim(oracle):
    """Kernels + orchestration of the product compiled against the HIP-on-CPU shim reproduce the oracle's SZ 1.4 streams byte
    for byte and decode them bit for bit (float and double; smooth, noisy, with a spike and zero planes)."""
    import sz_amd
    from sz_amd import api
    saved = api._lib
    try:
        api._lib = api._bind(ctypes.CDLL(sim_lib.shim_path()))
        assert sz_amd.SZ_Init(os.path.join(sim_lib.ROOT, "tests", "golden", "sz_speed.config")) == 0
        api.conf_params().withRegression = 0
        p = oracle.default_params(with_regression=0)
        spike = s_field(16, 16, 16); spike[3, 4, 5] = 1e4; spike[:, :2, :] = 0
        cases = (("smooth", s_field(10, 12, 40), 1e-4), ("noisy", _noisy((9, 17, 33), np.float32, 3e-4), 1e-5),
                 ("noisy-f64", _noisy((12, 10, 24), np.float64, 3e-4), 1e-6), ("spike", spike, 1e-3),
                 ("2d", plane_field(40, 70), 1e-4), ("2d-f64", plane_field(33, 40, np.float64), 1e-5),   # 2-D: sz_float.c:610
                 ("1d", _walk(3000, np.float32), 1e-3), ("1d-f64", _walk(2500, np.float64), 1e-4))        # 1-D: sz_float.c:353
        for name, d, eb in cases:
            ref, _ = oracle.compress(d, oracle.ABS, eb, params=p)
            got = sz_amd.SZ_compress_args(d, sz_amd.ABS, eb)
            assert got == ref, name
            dec = sz_amd.SZ_decompress(ref, d.shape, d.dtype)
            assert np.array_equal(dec.view(np.uint8), oracle.decompress(ref, d.shape, d.dtype).view(np.uint8)), name
        # the 1-D chain's other walks: the whole array by one wavefront, and the segmented walk cut too eagerly (flag -> fallback)
        d = _walk(1500, np.float32); ref, _ = oracle.compress(d, oracle.ABS, 1e-3, params=p)
        for knob, val in (("SZ_HIP_1D_SERIAL", "1"), ("SZ_HIP_1D_REACH_PCT", "3")):
            os.environ[knob] = val
            try:
                assert sz_amd.SZ_compress_args(d, sz_amd.ABS, 1e-3) == ref, knob
                assert sz_amd.SZ_hip_last_stats().quant_kernel_launches == 2, knob
                dec = sz_amd.SZ_decompress(ref, d.shape, d.dtype)
                assert np.array_equal(dec.view(np.uint8), oracle.decompress(ref, d.shape, d.dtype).view(np.uint8)), knob
            finally:
                del os.environ[knob]
        sz_amd.SZ_Finalize()
    finally:
        api._lib = saved
