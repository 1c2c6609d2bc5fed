"""Round 6: the packing passes that read the sweep's natural-order codes (sz_amd/csrc/szh_segenc.h: per-column histograms, k_col_bits / k_col_bits_h, k_col_scan,
k_col_encode; on the way back k_col_zeros, k_col_unpack), the fit pass from LDS tiles (szh_fittile.h) -- every form must give the oracle's stream,
byte for byte (the oracle: oracle/, pinned against the reference's recorded outputs; sz_float.c:7064-7359, Huffman.c:205-308, sz_float.c:6598-6633, :7083-7123)."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _cases():
    from sz_amd.fields import m_field, s_field
    rng = np.random.default_rng(5)
    return [("s40x44x48", s_field(40, 44, 48)),                                                  # blocks of 6 and 7, Lorenzo only
            ("m48", m_field(48)),                                                                # regression blocks: coefficient sections
            ("s13x17x100", s_field(13, 17, 100)),                                                # one block along dim 0 of 13: blocks of 6 / 7 and a long row
            ("s50x47x52", s_field(50, 47, 52)),                                                  # every dimension with early and late blocks
            ("noisy", (s_field(30, 36, 40) + 0.01 * rng.standard_normal((30, 36, 40))).astype(np.float32)),      # 1024 intervals: beyond the per-column histograms
            ("f64", s_field(24, 30, 36, np.float64)),
            ("s20x20x45", s_field(20, 20, 45)),                                                  # rows that are no multiple of four values: k_pencil
            ("s7x9x12", s_field(7, 9, 12)),                                                      # single blocks of 7 and 9: the general forms
            ("spiky", np.where(rng.random((24, 24, 64)) < 0.02, 100.0, s_field(24, 24, 64)).astype(np.float32)),  # 65536 / 16384 intervals: the block-ordered copy after all
            ("m30x41x56", np.ascontiguousarray(m_field(56)[:30, :41, :])),
            ("m-f64", m_field(36).astype(np.float64))]


SWITCHES = ["", "SZ_HIP_SEGENC=0", "SZ_HIP_SEGHIST=0", "SZ_HIP_SEG_SCAN1=0", "SZ_HIP_SEG_SEGB=3", "SZ_HIP_SEG_TILE_KB=4", "SZ_HIP_SEGENC=2", "SZ_HIP_FIT_TILE=1",
            "SZ_HIP_COL_UNPACK=0", "SZ_HIP_UNPACK_TILE_KB=4", "SZ_HIP_DEC_CHECKS_LAST=0"]


def _run(monkeypatch, switch, few=False):
    import oracle_lib as O
    import sz_amd
    for kv in filter(None, switch.split(";")):
        k, v = kv.split("=")
        monkeypatch.setenv(k, v)
    assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
    try:
        seen = set()
        for name, d in _cases():
            if few and name not in ("s40x44x48", "m48", "f64", "spiky", "s20x20x45"):      # (the CPU shim runs a lane at a time: the switches get five arrays, one bound)
                continue
            d = np.ascontiguousarray(d)
            for eb in ((1e-4,) if few or name not in ("s50x47x52", "m30x41x56") else (1e-4, 1e-2)):
                ref, _ = O.compress(d, O.ABS, eb)
                got = sz_amd.SZ_compress_args(d, sz_amd.ABS, eb)
                st = sz_amd.SZ_hip_last_stats()
                seen.add(int(st.packing))
                assert got == ref, (switch, name, eb, len(got), len(ref), int(st.packing), int(st.quant_kernel))
                dec = sz_amd.SZ_decompress(got, d.shape, d.dtype)
                want = O.decompress(ref, d.shape, d.dtype)
                assert np.array_equal(dec.view(np.uint8), want.view(np.uint8)), (switch, name, eb, "decoded values differ from the reference decoder's")
                assert float(np.abs(dec.astype(np.float64) - d).max()) <= eb
        return seen
    finally:
        sz_amd.SZ_Finalize()


@pytest.fixture
def shim():
    import sim_lib
    from sz_amd import api
    old = api._lib
    api._lib = api._bind(ctypes.CDLL(sim_lib.shim_path()))
    yield
    api._lib = old


@pytest.mark.parametrize("switch", SWITCHES)
def test_packing_from_natural_order_codes_on_the_cpu_shim(shim, monkeypatch, switch):
    seen = _run(monkeypatch, switch, few=switch != "")
    if switch == "SZ_HIP_SEGENC=0":
        assert seen == {0}
    else:
        assert seen == {0, 1}          # (the large alphabets still take the block-ordered copy)


@pytest.mark.gpu
@pytest.mark.parametrize("switch", SWITCHES)
def test_packing_from_natural_order_codes_on_the_gpu(monkeypatch, switch):
    _run(monkeypatch, switch)
