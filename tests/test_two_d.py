"""2-D SZ 2.1 path (sz_float.c:5516, szd_float.c:3141): CPU-side checks.

The 2-D restatement is pinned by recorded outputs of the unmodified reference (tests/test_ref_recorded.py, round 2).  Kept from
round 1, when no such output existed: the oracle's 2-D restatement against itself
(round trips within the bound, stream structure), and the product's kernels + orchestration against the oracle (here on
the HIP-on-CPU shim; on the GPU in test_gpu_parity.py)."""
import ctypes
import os
import struct

import numpy as np
import pytest

import sim_lib
from sz_amd.fields import near_zero_planes, plane_field


@pytest.mark.parametrize("shape,dtype,eb", [((200, 300), np.float32, 1e-4), ((37, 45), np.float32, 1e-3), ((130, 257), np.float64, 1e-5),
                                            ((2, 500), np.float32, 1e-3), ((300, 3), np.float64, 1e-3), ((16, 16), np.float32, 1e-2)])
def test_oracle_2d_round_trip_and_stream_structure(oracle, shape, dtype, eb):
    d = plane_field(*shape, dtype)
    stream, st = oracle.compress(d, oracle.ABS, eb, want_stages=True)
    dec = oracle.decompress(stream, shape, dtype)
    assert float(np.abs(dec.astype(np.float64) - d.astype(np.float64)).max()) <= eb
    if st is None:        # raw fallback (the stream would have been larger than the data)
        return
    meta = 28 if dtype == np.float32 else 36
    body = 4 + meta + 8
    assert stream[3] & 0x80                                         # SZ 2.1 regression-type stream
    assert struct.unpack(">Q", stream[4 + meta:body])[0] == d.size
    assert struct.unpack(">I", stream[body:body + 4])[0] == 16      # block size of the 2-D path
    assert st["use_mean"] == 0                                      # forced off in 2-D (sz_float.c:5615)
    assert st["reg_params"].shape[0] == 3 and len(st["coeff_unpred"]) == 3
    nb = ((shape[0] // 16) or 1) * ((shape[1] // 16) or 1)
    assert st["num_blocks"] == nb and st["indicator"].size == nb
    # every code of a Lorenzo block is below the capacity the 2-D path leaves for it (intervals - 2)
    assert int(st["codes"].max()) < st["intervals"]


def test_oracle_2d_regression_plane_reconstruction(oracle):
    """All blocks regression: the decoded values are the decoded plane + code * 2eb, which the test recomputes from the stages."""
    d = near_zero_planes(1, 64, 96)[0]
    eb = 1e-4
    stream, st = oracle.compress(d, oracle.ABS, eb, want_stages=True)
    assert st["reg_count"] == st["num_blocks"] == 24
    dec = oracle.decompress(stream, d.shape, d.dtype)
    a, b, c = (st["coeff_dec"][e].astype(np.float32) for e in range(3))
    codes = st["codes"].reshape(4, 6, 16, 16)                       # block order: (block row, block column, row, column)
    radius = st["intervals"] // 2
    ii = np.arange(16, dtype=np.float32)[:, None]
    jj = np.arange(16, dtype=np.float32)[None, :]
    un = iter(st["unpred"])
    for bi in range(4):
        for bj in range(6):
            k = bi * 6 + bj
            pred = (a[k] * ii + b[k] * jj) + c[k]
            want = pred + (2 * (codes[bi, bj] - radius)).astype(np.float32) * np.float32(eb)
            got = dec[bi * 16:(bi + 1) * 16, bj * 16:(bj + 1) * 16]
            for (r, cc) in zip(*np.nonzero(codes[bi, bj] == 0)):
                want[r, cc] = next(un)
            assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), (bi, bj)


@pytest.mark.slow
def test_2d_hip_layer_on_cpu_shim(oracle):
    """Kernels + orchestration of the product, compiled against the HIP-on-CPU shim, reproduce the oracle's 2-D streams byte for byte
    (Lorenzo-only, regression-only and mixed blocks; float and double)."""
    import sz_amd
    from sz_amd import api
    saved = api._lib
    try:
        api._lib = api._bind(ctypes.CDLL(sim_lib.shim_path()))
        assert sz_amd.SZ_Init(os.path.join(sim_lib.ROOT, "tests", "golden", "sz_speed.config")) == 0
        cases = (("plane", plane_field(40, 70), 1e-4), ("planes-reg", near_zero_planes(1, 35, 50)[0], 1e-4),
                 ("plane-f64", plane_field(33, 40, np.float64), 1e-5))
        for name, d, eb in cases:
            ref, st = oracle.compress(d, oracle.ABS, eb, want_stages=True)
            got = sz_amd.SZ_compress_args(d, sz_amd.ABS, eb)
            assert got == ref, name
            dec = sz_amd.SZ_decompress(ref, d.shape, d.dtype)
            assert np.array_equal(dec.view(np.uint8), oracle.decompress(ref, d.shape, d.dtype).view(np.uint8)), name
        sz_amd.SZ_Finalize()
    finally:
        api._lib = saved
