"""The opt-in FAST mode (SZ_HIP_MODE=fast; sz_amd/csrc/szh_fast.h): a feedback-free quantiser with this library's own container.

Its parity gate is its own (there is no such mode in the reference):
  * the HIP path must reproduce the fast-mode oracle (oracle/szo_fast.c: the same rule as plain sequential loops) byte for byte and
    decode bit for bit -- everything after one multiply / rint / verification multiply per point is integer arithmetic;
  * the absolute bound always holds;
  * ratio no worse than the exact mode on Lorenzo-predicted fields (S), a stated loss where the exact mode uses regression (M, L);
    PSNR within 0.1 dB.
CPU (-m "not gpu"): the oracle against itself and against the exact oracle; the product code through the CPU shim on small arrays.
GPU (-m gpu): the HIP library against the oracle, round trips at the BASELINE size.
"""
import ctypes
import os

import numpy as np
import pytest

from sz_amd.fields import l_field, m_field, s_field

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cases():
    rng = np.random.default_rng(5)
    spike = s_field(20, 30, 70)
    spike[3, 4, 5] = 1e30; spike[10, 2, 66] = np.nan; spike[19, 29, 69] = -3e38        # raw points (q does not fit / NaN)
    return [("S-40x48x130", s_field(40, 48, 130), 1e-4, 0), ("M48", m_field(48), 1e-4, 0), ("L-14x19x33", l_field(14, 19, 33), 1e-4, 0),
            ("odd-17x25x38", s_field(17, 25, 38), 1e-3, 64), ("noise", rng.standard_normal((20, 24, 28)).astype(np.float32), 1e-4, 256),
            ("S-f64", s_field(24, 30, 70, np.float64), 1e-6, 0), ("raw-points", spike, 1e-4, 0), ("2D", s_field(1, 100, 120)[0], 1e-4, 0),
            ("1D", np.cumsum(rng.standard_normal(5000)).astype(np.float32) * np.float32(0.01), 1e-3, 0)]


def _small_cases():
    """for the CPU shim, which runs every lane of every workgroup as a fibre"""
    rng = np.random.default_rng(6)
    spike = s_field(8, 10, 70)
    spike[3, 4, 5] = 1e30; spike[7, 2, 66] = np.nan; spike[0, 0, 0] = -3e38
    return [("interior-tile-9x17x70", s_field(9, 17, 70), 1e-4, 0), ("odd-9x18x38", s_field(9, 18, 38), 1e-3, 64), ("S-f64", s_field(6, 17, 66, np.float64), 1e-6, 0), ("raw-points", spike, 1e-4, 0),
            ("2D", s_field(1, 30, 70)[0], 1e-4, 0), ("1D", np.cumsum(rng.standard_normal(2100)).astype(np.float32) * np.float32(0.01), 1e-3, 0)]


def _check_bound(d, dec, eb):
    fin = np.isfinite(d)
    assert float(np.abs(dec.astype(np.float64)[fin] - d.astype(np.float64)[fin]).max()) <= eb
    iv = np.uint32 if d.dtype == np.float32 else np.uint64
    assert np.array_equal(dec.view(iv)[~fin], d.view(iv)[~fin])          # what cannot be quantised is kept verbatim


def test_fast_oracle_round_trip_and_bound(oracle):
    for name, d, eb, iv in _cases():
        s = oracle.fast_compress(d, eb, iv)
        assert s[:4] == b"SZHF"
        dec = oracle.fast_decompress(s, d.shape, d.dtype)
        _check_bound(d, dec, np.dtype(d.dtype).type(eb))


def test_fast_mode_quality_next_to_exact_mode(oracle):
    """Where the exact mode predicts with Lorenzo (the S-field, the headline workload) the fast stream is no larger than the exact
    one (measured: -0.8 % at 64^3, -1.1 % at 128^3).  The fast mode has NO regression predictor (a regression block cannot be undone
    by prefix sums), so noise-dominated fields where the exact mode picks regression pay for it: +20 % on M, +42 % on L at 64^3.
    PSNR: the pre-quantisation error is uniform in [-eb, eb] like the exact mode's, within a few hundredths of a dB."""
    for name, d, tol in (("S", s_field(64, 64, 64), 0.0), ("M", m_field(64), 0.25), ("L", l_field(64, 64, 64), 0.45)):
        exact, _ = oracle.compress(d, oracle.ABS, 1e-4)
        fast = oracle.fast_compress(d, 1e-4)
        _, p_exact, _ = oracle.metrics(d, oracle.decompress(exact, d.shape, d.dtype))
        _, p_fast, _ = oracle.metrics(d, oracle.fast_decompress(fast, d.shape, d.dtype))
        print(f"{name}: exact {len(exact)} B PSNR {p_exact:.4f} | fast {len(fast)} B PSNR {p_fast:.4f}")
        assert len(fast) <= len(exact) * (1 + tol), name
        assert abs(p_fast - p_exact) < 0.1, name


@pytest.mark.slow
def test_product_fast_mode_on_cpu_shim_matches_oracle(oracle, built):
    import sim_lib
    import sz_amd
    from sz_amd import api
    saved = api._lib
    try:
        api._lib = api._bind(ctypes.CDLL(sim_lib.shim_path()))
        ctx = sz_amd.HipContext(0)
        for name, d, eb, iv in _small_cases():
            d3 = d.reshape((1,) * (3 - d.ndim) + d.shape)
            ref = oracle.fast_compress(d, eb, iv)
            for form in ("0", "1"):                      # the code-array form (default) and the two-pass form (szh_fast.h, round 3)
                os.environ["SZ_HIP_FAST2"] = form
                got, n, st = ctx.compress_fast(d3.ctypes.data, False, d3.shape, d.dtype, eb, iv)
                assert got == ref, (name, form)
                assert st.quant_kernel == (2 if form == "1" else 0), (name, form)
            out = np.empty_like(d3)
            buf = ctypes.create_string_buffer(ref, len(ref))
            ctx.decompress_fast(ctypes.addressof(buf), False, len(ref), d3.shape, d.dtype, out.ctypes.data, False)
            iview = np.uint32 if d.dtype == np.float32 else np.uint64
            assert np.array_equal(out.reshape(d.shape).view(iview), oracle.fast_decompress(ref, d.shape, d.dtype).view(iview)), name
        ctx.close()
        # through the reference API with SZ_HIP_MODE=fast
        os.environ["SZ_HIP_MODE"] = "fast"
        assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
        d = s_field(5, 17, 66)
        s = sz_amd.SZ_compress_args(d, sz_amd.ABS, 1e-4)
        assert s == oracle.fast_compress(d, 1e-4)
        back = sz_amd.SZ_decompress(s, d.shape, d.dtype)
        assert float(np.abs(back - d).max()) <= 1e-4
        sz_amd.SZ_Finalize()
    finally:
        os.environ.pop("SZ_HIP_MODE", None)
        os.environ.pop("SZ_HIP_FAST2", None)
        api._lib = saved


@pytest.mark.gpu
def test_hip_fast_mode_matches_oracle(oracle, built):
    import sz_amd
    ctx = sz_amd.HipContext(0)
    for name, d, eb, iv in _cases():
        d3 = np.ascontiguousarray(d.reshape((1,) * (3 - d.ndim) + d.shape))
        ref = oracle.fast_compress(d, eb, iv)
        try:
            for form in ("0", "1"):                          # the code-array form (default) and the two-pass form
                os.environ["SZ_HIP_FAST2"] = form
                got, n, st = ctx.compress_fast(d3.ctypes.data, False, d3.shape, d.dtype, eb, iv)
                assert got == ref, (name, form)
        finally:
            os.environ.pop("SZ_HIP_FAST2", None)
        out = np.empty_like(d3)
        buf = ctypes.create_string_buffer(ref, len(ref))
        ctx.decompress_fast(ctypes.addressof(buf), False, len(ref), d3.shape, d.dtype, out.ctypes.data, False)
        iview = np.uint32 if d.dtype == np.float32 else np.uint64
        assert np.array_equal(out.reshape(d.shape).view(iview), oracle.fast_decompress(ref, d.shape, d.dtype).view(iview)), name
    ctx.close()


@pytest.mark.gpu
def test_hip_fast_mode_512_round_trip_and_api_switch(built):
    import sz_amd
    d = s_field(256, 512, 512)
    os.environ["SZ_HIP_MODE"] = "fast"
    try:
        assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
        s = sz_amd.SZ_compress_args(d, sz_amd.ABS, 1e-4)
        assert s[:4] == b"SZHF" and len(s) < d.nbytes / 10
        back = sz_amd.SZ_decompress(s, d.shape, d.dtype)
        assert float(np.abs(back.astype(np.float64) - d).max()) <= 1e-4
        sz_amd.SZ_Finalize()
    finally:
        os.environ.pop("SZ_HIP_MODE", None)


def test_huffman_decoder_blocks_long_codes_and_repair_rounds_on_cpu_shim(oracle, built):
    """The round-3 Huffman decoder (look-up table over LDS-staged bits, szhip_kernels.h `k_hdec_*`) through the product code on the CPU
    shim, on payloads that span several workgroup blocks (256 sub-sequences = 32 KB each), with codes longer than the 10-bit window
    (wide symbol distributions) and with warm-up guesses that need the repair round: decoded values must be the oracle decoder's bit
    for bit, for the exact SZ 2.1 path and for the fast-mode container."""
    import sim_lib
    import sz_amd
    from sz_amd import api
    saved = api._lib
    rng = np.random.default_rng(11)
    cases = [("noise-wide", rng.standard_normal((40, 48, 64)).astype(np.float32), 2e-4),            # ~12 bits per symbol: long codes, 6 blocks
             ("smooth+noise", (s_field(30, 64, 96) + 0.003 * rng.standard_normal((30, 64, 96))).astype(np.float32), 1e-4),
             ("two-symbols", np.where(rng.random((24, 40, 70)) < 0.03, 1.0, 0.0).astype(np.float32), 1e-3)]   # 1-bit codes: 4 symbols per look-up
    try:
        api._lib = api._bind(ctypes.CDLL(sim_lib.shim_path()))
        assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
        ctx = sz_amd.HipContext(0)
        for name, d, eb in cases:
            ref_stream, _ = oracle.compress(d, oracle.ABS, eb)
            ref_dec = oracle.decompress(ref_stream, d.shape, d.dtype)
            got = sz_amd.SZ_decompress(ref_stream, d.shape, d.dtype)
            assert np.array_equal(got.view(np.uint32), ref_dec.view(np.uint32)), name
            fs = oracle.fast_compress(d, eb)
            out = np.empty_like(d)
            buf = ctypes.create_string_buffer(fs, len(fs))
            ctx.decompress_fast(ctypes.addressof(buf), False, len(fs), d.shape, d.dtype, out.ctypes.data, False)
            assert np.array_equal(out.view(np.uint32), oracle.fast_decompress(fs, d.shape, d.dtype).view(np.uint32)), name
        # the decoder runs two rounds without asking the device in between; a call whose starts were still moving is repeated with a
        # synchronisation per round (with_ticket_fallback): forced here
        os.environ["SZ_HIP_TEST_HDEC_FALLBACK"] = "1"
        name, d, eb = cases[1]
        ref_stream, _ = oracle.compress(d, oracle.ABS, eb)
        got = sz_amd.SZ_decompress(ref_stream, d.shape, d.dtype)
        assert np.array_equal(got.view(np.uint32), oracle.decompress(ref_stream, d.shape, d.dtype).view(np.uint32))
        ctx.close()
        sz_amd.SZ_Finalize()
    finally:
        os.environ.pop("SZ_HIP_TEST_HDEC_FALLBACK", None)
        api._lib = saved


def _ragged_inverse_cases():
    return [("S-33x70x50", s_field(33, 70, 50), 1e-4), ("S-f64-20x65x40", s_field(20, 65, 40, np.float64), 1e-6), ("S-17x130x38", s_field(17, 130, 38), 1e-3)]


def test_inverse_of_arrays_the_beam_does_not_take_on_cpu_shim(oracle, built):
    """rows that are no multiple of four values: the inverse sweep runs k_pencil (until round 5: the ribbon mapping); it must
    decode the oracle's streams bit for bit -- ragged tiles in every dimension, float and double."""
    import sim_lib
    import sz_amd
    from sz_amd import api
    saved = api._lib
    try:
        api._lib = api._bind(ctypes.CDLL(sim_lib.shim_path()))
        assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
        for mode in ("-",):
            for name, d, eb in _ragged_inverse_cases():
                ref_stream, _ = oracle.compress(d, oracle.ABS, eb)
                got = sz_amd.SZ_decompress(ref_stream, d.shape, d.dtype)
                iv = np.uint32 if d.dtype == np.float32 else np.uint64
                assert np.array_equal(got.view(iv), oracle.decompress(ref_stream, d.shape, d.dtype).view(iv)), (name, mode)
        sz_amd.SZ_Finalize()
    finally:
        pass
        api._lib = saved


@pytest.mark.gpu
def test_inverse_of_arrays_the_beam_does_not_take_on_gpu(oracle, built):
    import sz_amd
    try:
        assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
        for mode in ("-",):
            for name, d, eb in _ragged_inverse_cases() + [("S-100x200x300", s_field(100, 200, 300), 1e-4)]:
                ref_stream, _ = oracle.compress(d, oracle.ABS, eb)
                got = sz_amd.SZ_decompress(ref_stream, d.shape, d.dtype)
                iv = np.uint32 if d.dtype == np.float32 else np.uint64
                assert np.array_equal(got.view(iv), oracle.decompress(ref_stream, d.shape, d.dtype).view(iv)), (name, mode)
        sz_amd.SZ_Finalize()
    finally:
        pass
