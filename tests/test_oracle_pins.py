"""The oracle (oracle/, a C restatement of the reference CPU path) against the reference's own recorded outputs
(tests/golden/anchors.json: stream md5 / byte sizes / PSNR of the unmodified reference)."""
import hashlib

import numpy as np
import pytest

from sz_amd.fields import m_field, s_field


def test_c1_stream_md5(oracle, anchors, c1_data):
    a = anchors["C1_testfloat_8_8_128_abs1e-4_best_speed"]
    stream, st = oracle.compress(c1_data, oracle.ABS, 1e-4, want_stages=True)
    assert len(stream) == a["stream_bytes"]
    assert hashlib.md5(stream).hexdigest() == a["md5"]
    assert st["intervals"] == a["intervals"] and st["node_count"] == a["node_count"]
    assert int(st["indicator"].sum()) == a["lorenzo_blocks"] and st["num_blocks"] == a["blocks"]
    assert len(np.unique(st["codes"])) == a["distinct_codes"]
    dec = oracle.decompress(stream, c1_data.shape, np.float32)
    mx, psnr, nrmse = oracle.metrics(c1_data, dec)
    assert f"{psnr:.6f}" == f"{a['psnr']:.6f}"
    assert abs(mx - a["max_abs_err"]) < 1e-9 and abs(nrmse - a["nrmse"]) < 1e-15


def test_m256_size_and_psnr(oracle, anchors):
    a = anchors["M256_f32_abs1e-4_best_speed"]
    d = m_field(256)
    stream, st = oracle.compress(d, oracle.ABS, 1e-4, want_stages=True)
    assert len(stream) == a["stream_bytes"]
    assert st["reg_count"] * 2 == st["num_blocks"] or abs(st["reg_count"] / st["num_blocks"] - a["reg_fraction"]) < 1e-3
    dec = oracle.decompress(stream, d.shape, np.float32)
    mx, psnr, _ = oracle.metrics(d, dec)
    assert f"{psnr:.6f}" == f"{a['psnr']:.6f}" and mx <= 1e-4


def test_f64_rel_slab(oracle, anchors):
    a = anchors["S_f64_slab_128x256x256_rel1e-3_best_speed"]
    d = s_field(128, 256, 256, np.float64)
    stream, st = oracle.compress(d, oracle.REL, 0.0, 1e-3, want_stages=True)
    assert len(stream) == a["stream_bytes"]
    assert abs(st["eb"] - a["eb"]) < 1e-9
    dec = oracle.decompress(stream, d.shape, np.float64)
    mx, psnr, _ = oracle.metrics(d, dec)
    assert f"{psnr:.6f}" == f"{a['psnr']:.6f}" and mx <= st["eb"]


@pytest.mark.slow
def test_s512_size_and_psnr(oracle, anchors):
    a = anchors["S512_f32_abs1e-4_best_speed"]
    d = s_field(512, 512, 512)
    stream, st = oracle.compress(d, oracle.ABS, 1e-4, want_stages=True)
    assert len(stream) == a["stream_bytes"]
    assert st["intervals"] == a["intervals"] and st["reg_count"] == a["reg_blocks"] and st["num_blocks"] == a["blocks"]
    dec = oracle.decompress(stream, d.shape, np.float32)
    mx, psnr, _ = oracle.metrics(d, dec)
    assert f"{psnr:.6f}" == f"{a['psnr']:.6f}" and mx <= 1e-4


@pytest.mark.slow
def test_s512_sz14_no_regression_size_exact_count_psnr(oracle, anchors):
    """SZ 1.4 path (withLinearRegression = NO): exact stream size, number of "exact" (unpredictable) values, PSNR and maximum error
    of the unmodified reference on the 512^3 S-field."""
    a = anchors["S512_f32_abs1e-4_best_speed_no_regression_sz14"]
    d = s_field(512, 512, 512)
    stream, st = oracle.compress(d, oracle.ABS, 1e-4, params=oracle.default_params(with_regression=0), want_stages=True)
    assert len(stream) == a["stream_bytes"] and st["exact_count"] == a["exact_values"]
    assert f"{d.nbytes / len(stream):.6f}" == f"{a['ratio']:.6f}"
    dec = oracle.decompress(stream, d.shape, np.float32)
    mx, psnr, _ = oracle.metrics(d, dec)
    assert f"{psnr:.6f}" == f"{a['psnr']:.6f}" and f"{mx:.6g}" == f"{a['max_abs_err']:.6g}"


def test_oracle_edge_cases(oracle):
    # constant array, tiny array (<= 20 values), 4-D folded to 3-D, expansion fallback on noise
    c = np.full((10, 12, 14), 3.25, dtype=np.float32)
    s, _ = oracle.compress(c, oracle.ABS, 1e-3)
    assert len(s) == 4 + 28 + 8 + 4 and np.array_equal(oracle.decompress(s, c.shape, np.float32), c)
    t = np.arange(18, dtype=np.float64).reshape(2, 3, 3)
    s, _ = oracle.compress(t, oracle.ABS, 1e-3)
    assert len(s) == 18 * 8 and np.array_equal(oracle.decompress(s, t.shape, np.float64), t)
    d4 = s_field(12, 20, 24).reshape(3, 4, 20, 24)
    s4, _ = oracle.compress(d4, oracle.ABS, 1e-4)
    s3, _ = oracle.compress(d4.reshape(12, 20, 24), oracle.ABS, 1e-4)
    assert s4 == s3
    rng = np.random.default_rng(1)
    noise = rng.standard_normal((16, 16, 16)).astype(np.float32)
    s, _ = oracle.compress(noise, oracle.ABS, 1e-7)
    assert s[3] & 0x10 and np.array_equal(oracle.decompress(s, noise.shape, np.float32), noise)  # stored raw
