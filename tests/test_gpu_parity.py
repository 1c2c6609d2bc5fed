"""Parity tests proper: the HIP path on a real MI355X, called through the C ABI (SZ_* and szhip_*), against
(a) the oracle on the same seeded inputs, (b) the committed golden anchors of the reference, and (c) at the
BASELINE size, size-independent properties (bound, PSNR, stream size, round trips).  Bit-exact for streams and
decoded values: the path is integer/byte work wrapped around IEEE arithmetic evaluated in the reference's order."""
import ctypes
import hashlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sz(built):
    import sz_amd
    assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
    yield sz_amd
    sz_amd.SZ_Finalize()


def _fields():
    from sz_amd.fields import l_field, m_field, near_zero_planes, plane_field, reg_beside_lorenzo, s_field
    rng = np.random.default_rng(0)
    z = s_field(40, 40, 40)
    z[np.abs(z) < 0.7] = 0.0
    spike = s_field(30, 30, 30)
    spike[5, 7, 11] = 1e6  # forces wide intervals + unpredictable values
    return {
        "S40": (s_field(40, 40, 40), 0, 1e-4, 0.0),
        "M64": (m_field(64), 0, 1e-4, 0.0),
        "L": (l_field(30, 50, 70), 0, 1e-4, 0.0),
        "ragged": (s_field(37, 45, 70), 0, 1e-4, 0.0),
        "thin": (s_field(200, 9, 7), 0, 1e-4, 0.0),
        "min-dims": (s_field(2, 3, 50), 0, 1e-3, 0.0),
        "mean-rand": (rng.random((33, 20, 17), dtype=np.float32), 0, 1e-2, 0.0),
        "mean-zeros": (z, 0, 1e-3, 0.0),
        "spike": (spike, 0, 1e-2, 0.0),
        "M48-f64": (m_field(48, np.float64), 0, 1e-5, 0.0),
        "S-f64-rel": (s_field(32, 64, 64, np.float64), 1, 0.0, 1e-3),
        "abs-and-rel": (s_field(24, 32, 40), 2, 1e-3, 1e-4),
        "abs-or-rel": (s_field(24, 32, 40), 3, 1e-5, 1e-4),
        "4d": (s_field(12, 20, 24).reshape(3, 4, 20, 24), 0, 1e-4, 0.0),
        "reg-beside-lorenzo": (reg_beside_lorenzo(24, 40, 32), 0, 1e-4, 0.0),
        "reg-beside-lorenzo-f64": (reg_beside_lorenzo(48, 56, 64, np.float64), 0, 1e-4, 0.0),
        # 2-D (16-wide blocks, three-coefficient planes; the oracle's 2-D restatement is pinned by tests/test_ref_recorded.py)
        "2d-plane": (plane_field(200, 300), 0, 1e-4, 0.0),
        "2d-ragged": (plane_field(37, 45), 0, 1e-3, 0.0),
        "2d-wide": (plane_field(2, 500), 0, 1e-3, 0.0),
        "2d-tall": (plane_field(300, 3), 0, 1e-3, 0.0),
        "2d-all-regression": (near_zero_planes(1, 100, 150)[0], 0, 1e-4, 0.0),
        "2d-f64-rel": (plane_field(130, 257, np.float64), 1, 0.0, 1e-4),
        "2d-from-3d-shape": (plane_field(64, 80).reshape(64, 1, 80), 0, 1e-4, 0.0),
        "2d-1024": (plane_field(1024, 1024), 0, 1e-4, 0.0),
        "S128": (s_field(128, 128, 128), 0, 1e-4, 0.0),
        "M128": (m_field(128), 0, 1e-4, 0.0),
    }


@pytest.mark.parametrize("name", list(_fields().keys()))
def test_stream_and_decode_identical_to_oracle(sz, oracle, name):
    d, mode, ab, rel = _fields()[name]
    ref, st = oracle.compress(d, mode, ab, rel, want_stages=True)
    got = sz.SZ_compress_args(d, mode, ab, rel)
    assert len(got) == len(ref) and got == ref, f"{name}: stream differs (first diff at {next((i for i, (x, y) in enumerate(zip(got, ref)) if x != y), None)})"
    stats = sz.SZ_hip_last_stats()
    if st is not None:
        assert (stats.intervals, stats.use_mean, stats.n_reg_blocks, stats.n_unpred) == (st["intervals"], st["use_mean"], st["reg_count"], st["total_unpred"])
        assert stats.quant_kernel_launches == 1
    dec = sz.SZ_decompress(ref, d.shape, d.dtype)
    ref_dec = oracle.decompress(ref, d.shape, d.dtype)
    iv = np.uint32 if d.dtype == np.float32 else np.uint64
    assert np.array_equal(dec.view(iv), ref_dec.view(iv)), name
    if st is not None:
        assert float(np.abs(dec.astype(np.float64) - d.astype(np.float64)).max()) <= st["eb"]


def test_golden_c1_md5_without_oracle(sz, anchors, c1_data):
    a = anchors["C1_testfloat_8_8_128_abs1e-4_best_speed"]
    got = sz.SZ_compress_args(c1_data, sz.ABS, 1e-4)
    assert len(got) == a["stream_bytes"] and hashlib.md5(got).hexdigest() == a["md5"]
    dec = sz.SZ_decompress(got, c1_data.shape, np.float32)
    err = np.abs(dec - c1_data)
    assert abs(float(err.max()) - a["max_abs_err"]) < 1e-9


def test_edge_cases_constant_tiny_raw(sz, oracle):
    c = np.full((10, 12, 14), 3.25, dtype=np.float32)
    s = sz.SZ_compress_args(c, sz.ABS, 1e-3)
    assert s == oracle.compress(c, oracle.ABS, 1e-3)[0] and np.array_equal(sz.SZ_decompress(s, c.shape, np.float32), c)
    t = np.arange(18, dtype=np.float64).reshape(2, 3, 3)
    s = sz.SZ_compress_args(t, sz.ABS, 1e-3)
    assert len(s) == 18 * 8 and np.array_equal(sz.SZ_decompress(s, t.shape, np.float64), t)
    rng = np.random.default_rng(1)
    noise = rng.standard_normal((16, 16, 16)).astype(np.float32)
    s = sz.SZ_compress_args(noise, sz.ABS, 1e-7)  # expands -> stored raw, flag 0x10
    assert s == oracle.compress(noise, oracle.ABS, 1e-7)[0] and np.array_equal(sz.SZ_decompress(s, noise.shape, np.float32), noise)


def test_psnr_and_norm_modes_identical_to_oracle(sz, oracle):
    """errorBoundMode PSNR and NORM are turned into an absolute bound on the host (conf.c:54-65); the streams record the effective
    ABS mode.  Same bytes and same decoded values as the oracle."""
    from sz_amd.fields import m_field, s_field
    cp = sz.conf_params()
    saved = (cp.psnr, cp.normErr)
    try:
        for name, d, mode, psnr, norm in (("psnr-80", s_field(24, 32, 40), sz.PSNR, 80.0, 0.0), ("psnr-60-f64", m_field(32, np.float64), sz.PSNR, 60.0, 0.0),
                                          ("norm", s_field(30, 30, 30), sz.NORM, 0.0, 0.05)):
            cp.psnr, cp.normErr = psnr, norm
            ref, _ = oracle.compress(d, mode, 0.0, 0.0, params=oracle.default_params(psnr=psnr, norm_err=norm))
            got = sz.SZ_compress_args(d, mode, 0.0, 0.0)
            assert got == ref, name
            dec = sz.SZ_decompress(ref, d.shape, d.dtype)
            assert np.array_equal(dec.view(np.uint8), oracle.decompress(ref, d.shape, d.dtype).view(np.uint8)), name
    finally:
        cp.psnr, cp.normErr = saved


def test_unsupported_calls_fail_loudly(sz):
    import sz_amd
    with pytest.raises(sz_amd.SZError):
        sz.SZ_compress_args(np.random.default_rng(0).random((3, 4, 5, 6, 7), dtype=np.float32), sz.ABS, 1e-3)    # 5-D: the reference refuses too
    with pytest.raises(sz_amd.SZError):
        sz.SZ_compress_args(np.random.default_rng(0).random((8, 9, 10), dtype=np.float32), sz.PW_REL, 0, 0, 0.0)       # a point-wise ratio of 0
    with pytest.raises(sz_amd.SZError):
        sz.SZ_decompress(b"\x02\x01\x0c\xc0" + b"\x00" * 60, (8, 9, 10), np.float32)
    with pytest.raises(sz_amd.SZError):                                                                             # a point-wise-relative (MSST19-flagged) stream of zeros
        sz.SZ_decompress(b"\x02\x01\x0c\x68" + b"\x00" * 200, (8, 9, 10), np.float32)


def _series(n, dtype, seed):
    """1-D test series: a smooth walk, a noisy stretch, a jump and a run of exact zeros."""
    rng = np.random.default_rng(seed)
    x = np.cumsum(rng.standard_normal(n)) * 0.01 + np.sin(np.arange(n) * 0.003)
    a, b = n // 5, n // 5 + max(1, n // 50)
    x[a:b] += 30.0 * rng.standard_normal(b - a)
    x[n // 2:] += 4.0
    x[3 * n // 4:3 * n // 4 + n // 40] = 0.0
    return np.ascontiguousarray(x.astype(dtype))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,eb,mode", [(21, 1e-3, 0), (64, 1e-2, 0), (65, 1e-4, 0), (1000, 1e-3, 0), (4097, 1e-5, 0), (200000, 1e-3, 0),
                                        (200000, 1e-6, 0), (50000, 1e-4, 1)])
def test_1d_stream_and_decode_identical_to_oracle(sz, oracle, dtype, n, eb, mode):
    """1-D arrays (SZ_compress_float_1D_MDQ, sz_float.c:353; decompressDataSeries_float_1D, szd_float.c:185): the chain through
    the previous reconstructed value, walked by one wavefront.  The stream must match byte for byte, with and without the
    regression switch (a 1-D array ignores it), and decode bit for bit.  (The 1-D restatement itself is pinned by the recorded
    1D-* reference outputs, tests/test_ref_recorded.py.)"""
    d = _series(n, dtype, seed=n)
    ref, _ = oracle.compress(d, mode, eb, eb)
    got = sz.SZ_compress_args(d, mode, eb, eb)
    assert got == ref
    dec = sz.SZ_decompress(ref, d.shape, d.dtype)
    want = oracle.decompress(ref, d.shape, d.dtype)
    assert np.array_equal(dec.view(np.uint8), want.view(np.uint8))
    if dtype == np.float32 and mode == 0:   # the float chain re-checks the bound; the double chain of the reference does not
        assert float(np.abs(dec.astype(np.float64) - d).max()) <= eb * (1 + 1e-6)
    cp = sz.conf_params()
    saved = cp.withRegression
    try:
        cp.withRegression = 0
        assert sz.SZ_compress_args(d, mode, eb, eb) == ref
    finally:
        cp.withRegression = saved


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_1d_segment_check_and_one_wavefront_walk(sz, oracle, dtype, monkeypatch):
    """The 1-D chain is cut where it must restart and walked one thread per segment; each thread checks that its successor's
    first value really takes the exact branch.  Cutting too eagerly (SZ_HIP_1D_REACH_PCT shrinks the reach) must raise the flag
    and fall back to the one-wavefront walk of the whole array, with the same stream; SZ_HIP_1D_SERIAL=1 takes that walk directly."""
    d = _series(60000, dtype, seed=9)
    ref, _ = oracle.compress(d, 0, 1e-3, 0.0)
    want = oracle.decompress(ref, d.shape, d.dtype)
    assert sz.SZ_compress_args(d, 0, 1e-3, 0.0) == ref
    assert sz.SZ_hip_last_stats().quant_kernel_launches == 1
    monkeypatch.setenv("SZ_HIP_1D_REACH_PCT", "3")
    assert sz.SZ_compress_args(d, 0, 1e-3, 0.0) == ref
    assert sz.SZ_hip_last_stats().quant_kernel_launches == 2          # flag raised -> walked again by one wavefront
    monkeypatch.delenv("SZ_HIP_1D_REACH_PCT")
    monkeypatch.setenv("SZ_HIP_1D_SERIAL", "1")
    assert sz.SZ_compress_args(d, 0, 1e-3, 0.0) == ref
    dec = sz.SZ_decompress(ref, d.shape, d.dtype)
    assert np.array_equal(dec.view(np.uint8), want.view(np.uint8))


def test_1d_full_size_identical_to_oracle_and_idempotent(sz, oracle):
    """32 Mi values (the oracle still finishes in a second): stream and decoded values identical; compressing the decoded
    series again reproduces it exactly where the codes were non-zero (a reconstruction sits on its own quantisation lattice)."""
    n = 1 << 25
    rng = np.random.default_rng(11)
    d = np.cumsum(rng.standard_normal(n)) * 0.01 + np.sin(np.arange(n) * 0.003)
    d[rng.integers(0, n, 2000)] += 50.0
    d = np.ascontiguousarray(d.astype(np.float32))
    ref, _ = oracle.compress(d, 0, 1e-3, 0.0)
    got = sz.SZ_compress_args(d, 0, 1e-3, 0.0)
    assert got == ref
    assert sz.SZ_hip_last_stats().quant_kernel_launches == 1
    dec = sz.SZ_decompress(got, d.shape, d.dtype)
    assert np.array_equal(dec.view(np.uint32), oracle.decompress(ref, d.shape, d.dtype).view(np.uint32))
    assert float(np.abs(dec.astype(np.float64) - d).max()) <= 1e-3 * (1 + 1e-6)
    again = sz.SZ_decompress(sz.SZ_compress_args(dec, 0, 1e-3, 0.0), d.shape, d.dtype)
    assert float(np.abs(again.astype(np.float64) - dec).max()) <= 1e-3 * (1 + 1e-6)


def test_1d_constant_tiny_and_incompressible_arrays(sz, oracle):
    for d in (np.full(5000, 3.25, dtype=np.float32),                                  # constant: header + one value
              np.arange(20, dtype=np.float64),                                        # <= 20 values: stored as they are
              np.random.default_rng(5).standard_normal(30000).astype(np.float32)):    # white noise at a tiny bound: raw fallback
        eb = 1e-9 if d.size == 30000 else 1e-3
        ref, _ = oracle.compress(d, 0, eb, 0.0)
        assert sz.SZ_compress_args(d, 0, eb, 0.0) == ref
        dec = sz.SZ_decompress(ref, d.shape, d.dtype)
        assert np.array_equal(dec.view(np.uint8), oracle.decompress(ref, d.shape, d.dtype).view(np.uint8))


def test_context_reuse_and_device_resident_entry_points(sz, oracle):
    """Several shapes/dtypes through ONE context (buffers and epochs are reused), device-resident in and out."""
    import torch
    from sz_amd.fields import m_field, s_field
    ctx = sz.HipContext(0)
    for d, eb in ((s_field(64, 64, 64), 1e-4), (m_field(40, np.float64), 1e-5), (s_field(24, 40, 56), 1e-3), (s_field(64, 64, 64), 1e-4)):
        ref, st = oracle.compress(d, oracle.ABS, eb, want_stages=True)
        meta = ref[:4 + (28 if d.dtype == np.float32 else 36)]
        x = torch.from_numpy(d).cuda()
        ptr, n, stats = ctx.compress(x.data_ptr(), True, d.shape, d.dtype, eb, meta, out_on_device=True)
        assert n == len(ref)
        out = torch.empty_like(x)
        ctx.decompress(ptr, True, n, len(meta) + 8, d.shape, d.dtype, out.data_ptr(), True)
        iv = torch.int32 if d.dtype == np.float32 else torch.int64
        ref_dec = torch.from_numpy(oracle.decompress(ref, d.shape, d.dtype)).cuda()
        assert torch.equal(out.view(iv), ref_dec.view(iv))
        host, n2, _ = ctx.compress(x.data_ptr(), True, d.shape, d.dtype, eb, meta)
        assert host == ref
    ctx.close()


def test_pool_lanes_give_the_oracle_streams(sz, oracle):
    """szhip_pool (include/szhip.h): K arrays in flight on one GPU -- persistent sweep workgroups on atomic tickets, one context per lane.
    Every stream must be the oracle's, whatever shares the chip: S-fields (k_ribbon), M-fields (k_pencil next to the host coefficient
    chain), a double array and a ragged shape, interleaved over two and three lanes."""
    import torch
    from sz_amd.fields import m_field, s_field
    cases = [(s_field(96, 128, 160), 1e-4), (m_field(96), 1e-4), (s_field(40, 70, 90, np.float64), 1e-6), (s_field(33, 65, 130), 1e-3), (m_field(64), 1e-4)]
    refs, xs, metas = [], [], []
    for d, eb in cases:
        ref, _ = oracle.compress(d, oracle.ABS, eb)
        refs.append(ref); xs.append(torch.from_numpy(d).cuda()); metas.append(ref[:4 + (28 if d.dtype == np.float32 else 36)])
    for lanes in (2, 3):
        pool = sz.HipPool(0, lanes)
        outs = [torch.empty(len(r) + (1 << 16), dtype=torch.uint8, device="cuda") for r in refs]
        for rounds in range(3):
            tks = [pool.submit(xs[i].data_ptr(), True, cases[i][0].shape, cases[i][0].dtype, cases[i][1], metas[i], None, outs[i].data_ptr(), outs[i].numel())
                   for i in range(len(cases))]
            for i, tk in enumerate(tks):
                n, st = pool.wait(tk)
                assert n == len(refs[i]) and bytes(outs[i][:n].cpu().numpy()) == refs[i], (lanes, rounds, i)
        pool.close()


def test_baseline_size_properties(sz, anchors):
    """512^3 float32 S-field, ABS 1e-4 (BASELINE.json configs[1]/[4]): stream size and PSNR equal the reference's recorded
    numbers; every point within the bound; compress(decompress(x)) of the lossy output is a fixed point of the decoder."""
    import torch
    from sz_amd.fields import s_field
    a = anchors["S512_f32_abs1e-4_best_speed"]
    d = s_field(512, 512, 512)
    x = torch.from_numpy(d).cuda()
    ctx = sz.HipContext(0)
    meta = sz.make_meta(np.float32, abs_bound=1e-4, vmin=float(d.min()), vmax=float(d.max()))
    ptr, n, stats = ctx.compress(x.data_ptr(), True, d.shape, np.float32, 1e-4, meta, out_on_device=True)
    assert n == a["stream_bytes"] and stats.intervals == a["intervals"] and stats.n_reg_blocks == a["reg_blocks"] and stats.n_blocks == a["blocks"]
    dec = torch.empty_like(x)
    ctx.decompress(ptr, True, n, 4 + 28 + 8, d.shape, np.float32, dec.data_ptr(), True)
    err = (dec - x).abs()
    assert float(err.max().item()) <= 1e-4
    mse = float((err * err).double().sum().item()) / x.numel()
    psnr = 20 * np.log10(float((x.max() - x.min()).item())) - 10 * np.log10(mse)
    assert f"{psnr:.6f}" == f"{a['psnr']:.6f}"
    # the host-pointer API must give the same bytes as the device-resident one
    host, n2, _ = ctx.compress(x.data_ptr(), True, d.shape, np.float32, 1e-4, meta)
    dev_copy = torch.empty(n, dtype=torch.uint8, device="cuda")
    p2, n3, _ = ctx.compress(x.data_ptr(), True, d.shape, np.float32, 1e-4, meta, out_on_device=True)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy(C.c_void_p(dev_copy.data_ptr()), C.c_void_p(p2), C.c_size_t(n3), 3)
    assert n2 == n3 == n and bytes(dev_copy.cpu().numpy().tobytes()) == host
    ctx.close()


def _roundtrip_quality(sz, d, mode, abs_b, rel_b):
    """compress + decompress through the public API on host arrays; returns (stream, decoded, psnr with the reference's formula)"""
    stream = sz.SZ_compress_args(d, mode, abs_b, rel_b)
    dec = sz.SZ_decompress(stream, d.shape, d.dtype)
    err = dec.astype(np.float64) - d.astype(np.float64)
    rng = float(d.max()) - float(d.min())
    psnr = 20 * np.log10(rng) - 10 * np.log10(float(np.mean(err * err)))
    return stream, dec, psnr, float(np.abs(err).max())


def test_config3_adaptive_field_matches_recorded_reference(sz, anchors):
    """BASELINE configs[2] at the size the survey recorded (256^3 M-field, half Lorenzo / half regression blocks, ABS 1e-4):
    stream size, PSNR (6 decimals) and the error bound equal the unmodified reference's numbers -- no oracle involved."""
    from sz_amd.fields import m_field
    a = anchors["M256_f32_abs1e-4_best_speed"]
    d = m_field(256)
    stream, dec, psnr, maxerr = _roundtrip_quality(sz, d, sz.ABS, 1e-4, 0.0)
    assert len(stream) == a["stream_bytes"]
    assert f"{psnr:.6f}" == f"{a['psnr']:.6f}"
    assert maxerr <= 1e-4
    st = sz.SZ_hip_last_stats() if hasattr(sz, "SZ_hip_last_stats") else None
    if st is not None and st.n_blocks:
        assert abs(st.n_reg_blocks / st.n_blocks - a["reg_fraction"]) < 1e-3


def test_config4_recorded_f64_rel_slab(sz, anchors):
    """One slab of BASELINE configs[3] at the size the survey recorded (128x256x256 float64 S-field, REL 1e-3)."""
    from sz_amd.fields import s_field
    a = anchors["S_f64_slab_128x256x256_rel1e-3_best_speed"]
    d = s_field(128, 256, 256, np.float64)
    stream, dec, psnr, maxerr = _roundtrip_quality(sz, d, sz.REL, 0.0, 1e-3)
    assert len(stream) == a["stream_bytes"]
    assert f"{psnr:.6f}" == f"{a['psnr']:.6f}"
    assert maxerr <= a["eb"] * (1 + 1e-12)


@pytest.mark.slow
def test_config4_full_size_slab_properties(sz):
    """The per-GPU unit of BASELINE configs[3]: a 128x1024x1024 float64 slab (1 GiB) of the 1024^3 S-field, REL 1e-3 with the
    bound taken from the GLOBAL value range (what the all-reduce of sz_amd.slab.global_minmax delivers).  Too large for the
    CPU oracle in test time, so size-independent properties: every point within the bound, a second compress/decompress
    generation of the decoded array stays within the bound of the first, and the stream of this smooth field beats 20:1."""
    import torch
    from sz_amd.fields import s_field
    d = s_field(128, 1024, 1024, np.float64, z0=3 * 128)          # slab 3 of 8
    eb = 1e-3 * (2 * 1.4834466)                                      # range of the full field
    x = torch.from_numpy(d).cuda()
    ctx = sz.HipContext(0)
    meta = sz.make_meta(np.float64, err_mode=sz.REL, rel_ratio=1e-3, vmin=-1.4834466, vmax=1.4834466)
    ptr, n, stats = ctx.compress(x.data_ptr(), True, d.shape, np.float64, eb, meta, out_on_device=True)
    assert n * 20 < d.nbytes and stats.n_blocks == 21 * 170 * 170
    dec = torch.empty_like(x)
    ctx.decompress(ptr, True, n, 4 + 36 + 8, d.shape, np.float64, dec.data_ptr(), True)
    assert float((dec - x).abs().max().item()) <= eb
    ptr2, n2, _ = ctx.compress(dec.data_ptr(), True, d.shape, np.float64, eb, meta, out_on_device=True)
    dec2 = torch.empty_like(x)
    ctx.decompress(ptr2, True, n2, 4 + 36 + 8, d.shape, np.float64, dec2.data_ptr(), True)
    assert float((dec2 - dec).abs().max().item()) <= eb          # a second generation stays within the bound of the first
    ctx.close()


def _sz14_fields():
    from sz_amd.fields import l_field, m_field, near_zero_planes, plane_field, s_field
    rng = np.random.default_rng(7)
    spike = s_field(30, 30, 30); spike[5, 7, 11] = 1e6; spike[:, :3, :] = 0
    noisy = s_field(40, 48, 56) + (rng.random((40, 48, 56)).astype(np.float32) - np.float32(0.5)) * np.float32(3e-4)
    return {
        "S40": (s_field(40, 40, 40), 0, 1e-4, 0.0),
        "noisy": (noisy, 0, 1e-5, 0.0),
        "L": (l_field(30, 50, 70), 0, 1e-4, 0.0),
        "ragged": (s_field(37, 45, 70), 0, 1e-4, 0.0),
        "thin": (s_field(200, 9, 7), 0, 1e-4, 0.0),
        "min-dims": (s_field(2, 3, 50), 0, 1e-3, 0.0),
        "random": (rng.random((33, 20, 17), dtype=np.float32), 0, 1e-2, 0.0),
        "spike": (spike, 0, 1e-2, 0.0),
        "tight-bound-32-bit-exact": (s_field(24, 24, 24), 0, 1e-7, 0.0),
        "M48-f64": (m_field(48, np.float64), 0, 1e-5, 0.0),
        "planes-f64": (near_zero_planes(20, 30, 40, np.float64), 0, 1e-6, 0.0),
        "S-f64-rel": (s_field(32, 64, 64, np.float64), 1, 0.0, 1e-3),
        "abs-and-rel": (s_field(24, 32, 40), 2, 1e-3, 1e-4),
        "S128": (s_field(128, 128, 128), 0, 1e-4, 0.0),
        # 2-D arrays on the SZ 1.4 path (sz_float.c:610; pinned by the recorded sz14-2D cases)
        "2d-plane": (plane_field(200, 300), 0, 1e-4, 0.0),
        "2d-ragged-f64": (plane_field(37, 45, np.float64), 0, 1e-5, 0.0),
        "2d-wide": (plane_field(2, 500), 0, 1e-3, 0.0),
        "2d-tall": (plane_field(300, 3), 0, 1e-3, 0.0),
        "2d-1024": (plane_field(1024, 1024), 0, 1e-4, 0.0),
    }


@pytest.fixture()
def sz14(sz, oracle):
    """The same library with `withLinearRegression = NO`: the SZ 1.4 path (sz_float.c:2978)."""
    sz.conf_params().withRegression = 0
    yield sz, oracle.default_params(with_regression=0)
    sz.conf_params().withRegression = 1


@pytest.mark.parametrize("name", list(_sz14_fields().keys()))
def test_sz14_stream_and_decode_identical_to_oracle(sz14, oracle, name):
    sz, p = sz14
    d, mode, ab, rel = _sz14_fields()[name]
    ref, st = oracle.compress(d, mode, ab, rel, params=p, want_stages=True)
    got = sz.SZ_compress_args(d, mode, ab, rel)
    assert len(got) == len(ref) and got == ref, f"{name}: stream differs (first diff at {next((i for i, (x, y) in enumerate(zip(got, ref)) if x != y), None)})"
    if st is not None:
        stats = sz.SZ_hip_last_stats()
        assert (stats.intervals, stats.n_unpred) == (st["intervals"], st["exact_count"])
    dec = sz.SZ_decompress(ref, d.shape, d.dtype)
    ref_dec = oracle.decompress(ref, d.shape, d.dtype)
    iv = np.uint32 if d.dtype == np.float32 else np.uint64
    assert np.array_equal(dec.view(iv), ref_dec.view(iv)), name
    if st is not None:
        assert float(np.abs(dec.astype(np.float64) - d.astype(np.float64)).max()) <= st["eb"]


def test_sz14_recorded_reference_output_at_512(sz14, anchors):
    """The unmodified reference's recorded result for the 512^3 S-field with withLinearRegression = NO: exact stream size and
    exact-value count, compression ratio and PSNR to six decimals, maximum error to six significant digits -- no oracle involved."""
    sz, _ = sz14
    from sz_amd.fields import s_field
    a = anchors["S512_f32_abs1e-4_best_speed_no_regression_sz14"]
    d = s_field(512, 512, 512)
    stream = sz.SZ_compress_args(d, sz.ABS, 1e-4)
    st = sz.SZ_hip_last_stats()
    assert len(stream) == a["stream_bytes"] and st.n_unpred == a["exact_values"]
    assert f"{d.nbytes / len(stream):.6f}" == f"{a['ratio']:.6f}"
    dec = sz.SZ_decompress(stream, d.shape, d.dtype)
    err = float(np.abs(dec.astype(np.float64) - d.astype(np.float64)).max())
    mse = float(((dec.astype(np.float64) - d.astype(np.float64)) ** 2).mean())
    rng = float(d.max()) - float(d.min())
    psnr = 20 * np.log10(rng) - 10 * np.log10(mse)
    assert f"{err:.6g}" == f"{a['max_abs_err']:.6g}" and abs(psnr - a["psnr"]) < 2e-6


def test_2d_full_size_properties(sz):
    """4096 x 4096 float32 (64 MiB), ABS 1e-4: too large for the oracle in seconds, so size-independent properties: the decoded
    array is within the bound, decoding is deterministic, and re-compressing the decoded array decodes to within the bound of it."""
    from sz_amd.fields import plane_field
    d = plane_field(4096, 4096)
    eb = 1e-4
    stream = sz.SZ_compress_args(d, sz.ABS, eb)
    st = sz.SZ_hip_last_stats()
    assert st.n_blocks == 256 * 256 and st.use_mean == 0 and 0 < st.n_reg_blocks < st.n_blocks
    dec = sz.SZ_decompress(stream, d.shape, d.dtype)
    assert float(np.abs(dec.astype(np.float64) - d.astype(np.float64)).max()) <= eb
    assert np.array_equal(sz.SZ_decompress(stream, d.shape, d.dtype), dec)
    assert sz.SZ_compress_args(d, sz.ABS, eb) == stream
    stream2 = sz.SZ_compress_args(dec, sz.ABS, eb)
    dec2 = sz.SZ_decompress(stream2, d.shape, d.dtype)
    assert float(np.abs(dec2.astype(np.float64) - dec.astype(np.float64)).max()) <= eb
    assert len(stream) < d.nbytes / 3


def test_corrupt_streams_fail_cleanly(sz, oracle):
    """Malformed input must end in an error (or, for payload damage the format cannot detect, in finite garbage) -- never in
    a hang or a crash: truncated streams, damaged header fields, damaged Huffman tree, damaged payload."""
    from sz_amd.fields import m_field
    d = m_field(40)
    good = sz.SZ_compress_args(d, sz.ABS, 1e-4)
    assert np.array_equal(sz.SZ_decompress(good, d.shape, d.dtype).view(np.uint32), oracle.decompress(good, d.shape, d.dtype).view(np.uint32))
    rng = np.random.default_rng(3)
    outcomes = {"error": 0, "decoded": 0}

    def attempt(blob):
        try:
            out = sz.SZ_decompress(bytes(blob), d.shape, d.dtype)
            assert out.shape == d.shape
            outcomes["decoded"] += 1
        except sz.SZError:
            outcomes["error"] += 1

    for cut in (0, 3, 17, 40, 44, 60, 200, len(good) // 2, len(good) - 1):            # truncation
        attempt(good[:cut])
    body = 4 + 28 + 8
    for off in list(range(body, body + 24)) + [body + 30, body + 64, body + 200]:     # block size, bound, intervals, tree size, node count, tree
        for val in (0x00, 0xff, 0x7f):
            blob = bytearray(good); blob[off] = val
            attempt(blob)
    for _ in range(40):                                                                # random damage anywhere
        blob = bytearray(good)
        for _ in range(int(rng.integers(1, 6))):
            blob[int(rng.integers(4, len(blob)))] = int(rng.integers(0, 256))
        attempt(blob)
    assert outcomes["error"] > 20                       # the header checks fire ...
    assert sz.SZ_compress_args(d, sz.ABS, 1e-4) == good  # ... and the library is still healthy afterwards


@pytest.mark.parametrize("which", ["3-D", "1-D"])
def test_corrupt_sz14_streams_fail_cleanly(sz14, oracle, which):
    """The same for the SZ 1.4 container: truncation, damaged size fields (type array, exact-value count, mid-byte count), damaged
    tree, damaged lead / mid / residual sections -- an error or finite garbage, never a hang or a crash.  Also for a 1-D series
    (same container; the decoder's segments then start at whatever the damaged code array says)."""
    sz, p = sz14
    from sz_amd.fields import s_field
    rng = np.random.default_rng(4)
    d = s_field(24, 32, 40) + (rng.random((24, 32, 40)).astype(np.float32) - np.float32(0.5)) * np.float32(3e-4)
    if which == "1-D":
        d = _series(30720, np.float32, seed=4) * np.float32(1e-2)
    good = sz.SZ_compress_args(d, sz.ABS, 1e-5)
    assert good == oracle.compress(d, oracle.ABS, 1e-5, params=p)[0]
    outcomes = {"error": 0, "decoded": 0}

    def attempt(blob):
        try:
            out = sz.SZ_decompress(bytes(blob), d.shape, d.dtype)
            assert out.shape == d.shape
            outcomes["decoded"] += 1
        except sz.SZError:
            outcomes["error"] += 1

    for cut in (0, 3, 17, 40, 44, 60, 77, 200, len(good) // 2, len(good) - 1):
        attempt(good[:cut])
    body = 4 + 28 + 8
    for off in list(range(body, body + 4 + 4 + 4 + 1 + 8 + 24 + 8)) + [body + 60, body + 100, body + 300]:
        for val in (0x00, 0xff, 0x7f):
            blob = bytearray(good); blob[off] = val
            attempt(blob)
    for _ in range(60):
        blob = bytearray(good)
        for _ in range(int(rng.integers(1, 6))):
            blob[int(rng.integers(4, len(blob)))] = int(rng.integers(0, 256))
        attempt(blob)
    assert outcomes["error"] > 20
    assert sz.SZ_compress_args(d, sz.ABS, 1e-5) == good


def test_differential_fuzz_against_the_oracle(built):
    """400 random small cases (shape, dtype, field kind, bound mode and size all random): stream byte-identical and decode
    bit-identical to the oracle.  (more than 70 000 cases, 1-D and 2-D arrays and the SZ 1.4 path included, were run in development; tools/gpu_fuzz.py prints the failing seeds.)"""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz.py"), "400", "11"], capture_output=True, text=True, timeout=600)
    tail = [ln for ln in out.stdout.splitlines() if ln.startswith("fuzz:") or ln.startswith("FAIL")]
    assert tail and tail[-1].startswith("fuzz: 400 cases, 0 failures"), "\n".join(tail[-10:]) + out.stderr[-2000:]


def test_differential_fuzz_sz14_against_the_oracle(built):
    """The same with withLinearRegression = NO (SZ 1.4 path), 300 random 3-D cases."""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz.py"), "300", "13", "sz14"], capture_output=True, text=True, timeout=600)
    tail = [ln for ln in out.stdout.splitlines() if ln.startswith("fuzz:") or ln.startswith("FAIL")]
    assert tail and tail[-1].startswith("fuzz: 300 cases, 0 failures"), "\n".join(tail[-10:]) + out.stderr[-2000:]


def test_fuzz_pw_rel(built):
    """random differential cases for a path added in round 2: point-wise relative bounds (log-domain form; the MSST19 form has tests/test_msst19.py)"""
    import subprocess
    for args in (("300", "41", "pwr"),):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz.py"), *args], capture_output=True, text=True, timeout=600)
        tail = [ln for ln in out.stdout.splitlines() if ln.startswith("fuzz:") or ln.startswith("FAIL")]
        assert tail and tail[-1].startswith("fuzz: 300 cases, 0 failures"), "\n".join(tail[-10:]) + out.stderr[-2000:]


def test_config1_through_a_plain_c_caller(built, anchors, tmp_path):
    """BASELINE configs[0] plumbing: a C program that only knows include/sz.h + include/rw.h (examples/sz_cli.c, the option
    letters of the reference's `sz` tool) compresses example/testdata's 8x8x128 float file with ABS 1e-4 and reproduces the
    reference's stream (md5) and quality report (PSNR to 6 decimals)."""
    import shutil
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    a = anchors["C1_testfloat_8_8_128_abs1e-4_best_speed"]
    dat = tmp_path / "testfloat_8_8_128.dat"
    shutil.copy(os.path.join(ROOT, "tests", "golden", "testfloat_8_8_128.dat"), dat)
    cfg = os.path.join(ROOT, "tests", "golden", "sz_speed.config")
    cli = os.path.join(ROOT, "examples", "sz_cli")
    subprocess.check_call([cli, "-z", "-f", "-c", cfg, "-M", "ABS", "-A", "1e-4", "-i", str(dat), "-3", "8", "8", "128"])
    stream = open(str(dat) + ".sz", "rb").read()
    assert len(stream) == a["stream_bytes"] and hashlib.md5(stream).hexdigest() == a["md5"]
    out = subprocess.check_output([cli, "-x", "-f", "-s", str(dat) + ".sz", "-i", str(dat), "-3", "8", "8", "128", "-a"]).decode()
    assert f"PSNR = {a['psnr']:.6f}" in out and "Max absolute error = 0.0000999272" in out
    # the rest of the option surface (example/sz.c:30-88): -p, output names after -z / -x, -t text output, -M PW_REL -P, -v
    meta = subprocess.check_output([cli, "-p", "-s", str(dat) + ".sz"]).decode()
    assert "2.1.12" in meta and "Num of elements:" in meta and "8192" in meta and "ABS" in meta and "FLOAT" in meta
    named = tmp_path / "named.sz"
    subprocess.check_call([cli, "-z", str(named), "-f", "-c", cfg, "-M", "PW_REL", "-P", "1e-2", "-i", str(dat), "-3", "8", "8", "128"])
    assert named.exists() and open(named, "rb").read()[3] & 0x20
    txt = tmp_path / "dec.txt"
    subprocess.check_call([cli, "-x", str(txt), "-f", "-s", str(named), "-3", "8", "8", "128", "-t"])
    vals = np.array([float(v) for v in open(txt).read().split()], dtype=np.float64)
    ori = np.fromfile(dat, dtype=np.float32).astype(np.float64)
    nz = ori != 0
    assert vals.size == ori.size and float((np.abs(vals[nz] - ori[nz]) / np.abs(ori[nz])).max()) <= 1e-2
    assert "2.1.12" in subprocess.check_output([cli, "-v"]).decode()


def test_coefficient_hand_off_with_alternating_inputs(sz, oracle):
    """The coefficient chain runs next to the wavefront kernel and ships decoded coefficients into a buffer the kernel reads meanwhile
    (DESIGN section 8).  Two different arrays of one shape alternate through ONE context: a coefficient line left in an L2 by the previous
    launch, or a progress word of the previous launch, would give the other array's codes.  Small arrays on purpose: nothing evicts a
    stale line there."""
    import torch
    from sz_amd.fields import m_field
    ctx = sz.HipContext(0)
    for dtype, eb in ((np.float32, 1e-4), (np.float64, 1e-5)):
        d1 = m_field(48, dtype)
        d2 = np.ascontiguousarray((m_field(48, dtype)[::-1] * dtype(1.7) + dtype(0.3)).astype(dtype))
        refs = []
        for d in (d1, d2):
            ref, st = oracle.compress(d, oracle.ABS, eb, want_stages=True)
            assert st["reg_count"] > 100
            refs.append((d, ref, torch.from_numpy(d).cuda()))
        assert refs[0][1] != refs[1][1]
        for k in range(8):
            d, ref, x = refs[k & 1]
            meta = ref[:4 + (28 if dtype == np.float32 else 36)]
            got, n, stats = ctx.compress(x.data_ptr(), True, d.shape, d.dtype, eb, meta)
            assert got == ref, (str(dtype), k)
        # the same through the process's first context (the SZ_* API's): its two streams are the likeliest to sit on separate hardware
        # queues, which the overlapped form needs (a context whose streams share a queue falls back to the serial order by itself)
        overlapped = []
        for k in range(6):
            d, ref, x = refs[k & 1]
            assert sz.SZ_compress_args(d, sz.ABS, eb) == ref, (str(dtype), "api", k)
            overlapped.append(sz.SZ_hip_last_stats().chain_overlapped)
        print("chain overlapped:", overlapped)
    ctx.close()
