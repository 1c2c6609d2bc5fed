"""The HIP kernel BODIES, run by CPU simulators (tests/sim; test infrastructure only) against the oracle.
These check the logic the GPU executes -- lane/shuffle topology, halo rows, block order, scans, bit packing, stream
assembly -- bit for bit, without a GPU.  The real-hardware parity tests are in test_gpu_parity.py."""
import ctypes

import numpy as np
import pytest

import sim_lib
import sz_amd
from sz_amd import api
from sz_amd.fields import l_field, m_field, reg_beside_lorenzo, s_field


def _cases(c1):
    rng = np.random.default_rng(0)
    z = s_field(24, 24, 24)
    z[np.abs(z) < 0.7] = 0.0
    return [("S", s_field(24, 24, 40), 1e-4), ("M", m_field(32), 1e-4), ("L", l_field(14, 19, 33), 1e-4),
            ("odd", s_field(17, 25, 38), 1e-4), ("C1", c1, 1e-4), ("M-f64", m_field(24, np.float64), 1e-5),
            ("mean-rand", rng.random((13, 20, 17), dtype=np.float32), 1e-2), ("mean-zeros", z, 1e-3),
            # regression blocks whose plane is small at k < 0, next to Lorenzo blocks: nothing may leak into the zero halo
            ("reg-beside-lorenzo", reg_beside_lorenzo(24, 40, 32), 1e-4), ("reg-beside-lorenzo-f64", reg_beside_lorenzo(24, 40, 32, np.float64), 1e-4)]


def test_wavefront_kernel_body_codes_and_reconstruction(oracle, c1_data):
    for name, d, eb in _cases(c1_data):
        ref, st = oracle.compress(d, oracle.ABS, eb, want_stages=True)
        err, codes = sim_lib.quantize(d, st)
        assert err == 0
        assert np.array_equal(sim_lib.nat_to_blk(codes, d.shape), st["codes"]), name
        err, out = sim_lib.reconstruct(codes, d, st)
        dec = oracle.decompress(ref, d.shape, d.dtype)
        iv = np.uint32 if d.dtype == np.float32 else np.uint64
        assert err == 0 and np.array_equal(out.view(iv), dec.view(iv)), name


def test_fit_and_selection_bodies(oracle, c1_data):
    for name, d, eb in _cases(c1_data):
        _, st = oracle.compress(d, oracle.ABS, eb, want_stages=True)
        coef, bl = sim_lib.fit_select(d, st["eb"], st["use_mean"], st["mean"])
        assert np.array_equal(coef.view(np.uint8), st["reg_params"].view(np.uint8)), name
        assert np.array_equal(bl, st["indicator"]), name


def test_sampling_and_interval_decision(oracle, c1_data, built):
    L = sz_amd.lib()

    class Dec(ctypes.Structure):
        _fields_ = [("intervals", ctypes.c_uint), ("use_mean", ctypes.c_int), ("dense_pos", ctypes.c_double),
                    ("mean_freq", ctypes.c_double), ("sample_freq", ctypes.c_double)]
    L.szhost_decide.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                                ctypes.c_float, ctypes.c_double, ctypes.c_double, ctypes.POINTER(Dec)]
    for name, d, eb in _cases(c1_data):
        _, st = oracle.compress(d, oracle.ABS, eb, want_stages=True)
        mean, rh, fh, within, count = sim_lib.sample(d, st["eb"])
        assert int(rh.sum()) == count
        dec = Dec()
        L.szhost_decide(int(d.dtype == np.float64), rh.ctypes.data, 32768, fh.ctypes.data, count, within, np.float32(0.99), st["eb"], mean, ctypes.byref(dec))
        assert (dec.intervals, dec.use_mean) == (st["intervals"], st["use_mean"]), name
        if count:
            assert dec.dense_pos == st["dense_pos"] and dec.mean_freq == st["mean_freq"] and dec.sample_freq == st["sample_freq"], name


@pytest.mark.slow
def test_whole_hip_layer_on_cpu_shim(oracle, c1_data):
    """szhip.hip + host C compiled against the HIP-on-CPU shim: every kernel and the orchestration, end to end,
    must reproduce the oracle's stream byte for byte and decode it bit for bit."""
    saved = api._lib
    try:
        api._lib = api._bind(ctypes.CDLL(sim_lib.shim_path()))
        import os
        assert sz_amd.SZ_Init(os.path.join(sim_lib.ROOT, "tests", "golden", "sz_speed.config")) == 0
        for name, d, eb in (("C1", c1_data, 1e-4), ("M20", m_field(20), 1e-4), ("S-odd-rows", s_field(26, 30, 57), 1e-4)):   # odd r2: granule rows start at odd 8-byte slots
            ref, _ = oracle.compress(d, oracle.ABS, eb)
            got = sz_amd.SZ_compress_args(d, sz_amd.ABS, eb)
            assert got == ref, name
            dec = sz_amd.SZ_decompress(ref, d.shape, d.dtype)
            assert np.array_equal(dec.view(np.uint32), oracle.decompress(ref, d.shape, d.dtype).view(np.uint32)), name
        sz_amd.SZ_Finalize()
    finally:
        api._lib = saved


@pytest.mark.parametrize("switch", ["SZ_HIP_SLICES=1", "SZ_HIP_SLICES=4", "SZ_HIP_SLICES=16", "SZ_HIP_SLICES=3",
                                    "SZ_HIP_ENC32=0", "SZ_HIP_PERM_Y=2"])
def test_entropy_stage_on_finished_tile_rows_on_cpu_shim(oracle, monkeypatch, switch):
    """round 4: the histogram and block-ordering passes run slice by slice on finished tile rows (k_ribbon publishes every tile; on the shim the sweep
    is over when the host looks, but the slices, their bounds against the block rows and the third stream's order are the product's own).  An array
    of five tile rows (70 planes of 16) and ragged blocks: every slicing gives the oracle's bytes; so do the forms round 4 replaced."""
    for kv in switch.split(";"):
        k, v = kv.split("=")
        monkeypatch.setenv(k, v)
    saved = api._lib
    try:
        api._lib = api._bind(ctypes.CDLL(sim_lib.shim_path()))
        import os
        assert sz_amd.SZ_Init(os.path.join(sim_lib.ROOT, "tests", "golden", "sz_speed.config")) == 0
        for name, d, eb in (("S-70x20x45", s_field(70, 20, 45), 1e-4), ("S-f64-41x70x33", s_field(41, 70, 33, np.float64), 1e-3)):
            ref, _ = oracle.compress(d, oracle.ABS, eb)
            got = sz_amd.SZ_compress_args(d, sz_amd.ABS, eb)
            assert got == ref, (name, switch)
        sz_amd.SZ_Finalize()
    finally:
        api._lib = saved
