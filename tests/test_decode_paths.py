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


@pytest.mark.gpu
def test_decompression_ignores_what_the_output_array_and_the_bytes_behind_the_stream_held(oracle, built):
    """The inverse sweep reads the output array (the unpredictable values are scattered into it first) and the decoder reads whole words up to the
    stream's end: neither what the array held before the call nor the bytes behind the stream may show in the result (bit for bit the oracle's values)."""
    import torch
    from sz_amd import api
    dev = torch.device("cuda:0")
    ctx = api.HipContext(0)
    try:
        for name, d, eb in (("S-96", s_field(96, 96, 96), 1e-4), ("M-72", m_field(72), 1e-4), ("S-f64-41x70x36", s_field(41, 70, 36, np.float64), 1e-3)):
            stream, _ = oracle.compress(d, oracle.ABS, eb)
            want = torch.from_numpy(oracle.decompress(stream, d.shape, d.dtype)).to(dev)
            body_off = 4 + (28 if d.dtype == np.float32 else 36) + 8
            other = torch.from_numpy(np.ascontiguousarray(d[::-1])).to(dev)
            it = torch.int32 if d.dtype == np.float32 else torch.int64
            fills = {"zeros": lambda o: o.zero_(), "0xff bytes": lambda o: o.view(torch.uint8).fill_(255), "the field upside down": lambda o: o.copy_(other),
                     "random bits": lambda o: o.view(torch.int32).random_(-2 ** 31, 2 ** 31 - 1)}
            for tail in ("zeros", "random"):
                s = torch.zeros(len(stream) + 4096, dtype=torch.uint8, device=dev)
                s[:len(stream)] = torch.frombuffer(bytearray(stream), dtype=torch.uint8).to(dev)
                if tail == "random": s[len(stream):] = torch.randint(0, 256, (4096,), dtype=torch.uint8, device=dev)
                for fname, f in fills.items():
                    out = torch.empty_like(want); f(out)
                    torch.cuda.synchronize()
                    ctx.decompress(s.data_ptr(), True, len(stream), body_off, d.shape, d.dtype, out.data_ptr(), True)
                    assert torch.equal(out.view(it), want.view(it)), (name, tail, fname)
    finally:
        ctx.close()
