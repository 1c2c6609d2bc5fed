"""Paths of the decompression that the main parity files do not single out: the Huffman decoder's fall-backs (code words longer than its look-up window, repair rounds with a
host round trip each) and the inverse sweep of arrays the beam does not take (rows that are no multiple of four values: k_pencil).  Until round 6 this file also held the tests
of the opt-in fast container, which was removed."""
import ctypes
import os

import numpy as np
import pytest

from sz_amd.fields import l_field, m_field, s_field

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_huffman_decoder_blocks_long_codes_and_repair_rounds_on_cpu_shim(oracle, built):
    """The round-3 Huffman decoder (look-up table over LDS-staged bits, szhip_kernels.h `k_hdec_*`) through the product code on the CPU
    shim, on payloads that span several workgroup blocks (256 sub-sequences = 32 KB each), with codes longer than the 10-bit window
    (wide symbol distributions) and with warm-up guesses that need the repair round: decoded values must be the oracle decoder's bit
    for bit."""
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
