"""Recorded outputs of the UNMODIFIED reference (tests/golden/ref_recorded.json, made by tools/record_reference_outputs.py from
the case list tests/ref_cases.py) replayed

  * on the CPU against the oracle (oracle/): stream length + md5 and the md5 of the decoded array must be the reference's --
    this is what pins the restatement for 2-D, 1-D, use_mean, f64 regression, PSNR/NORM, the sz.config knobs and PW_REL;
  * on the GPU (-m gpu) against the HIP library through the reference's C API (SZ_Init(config) / SZ_compress_args /
    SZ_decompress): same streams, same decoded bits, WITHOUT the oracle in between; and the reference-made streams stored under
    tests/golden/ref_streams/ are decoded by the HIP library.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import ref_cases
from ref_cases import PW_REL

HERE = os.path.dirname(os.path.abspath(__file__))
REC = json.load(open(os.path.join(HERE, "golden", "ref_recorded.json")))["cases"]

SZ_MODES = {"SZ_BEST_SPEED": 0, "SZ_BEST_COMPRESSION": 1, "SZ_DEFAULT_COMPRESSION": 2}


def _wrapped(c):
    return c["conf"].get("szMode", "SZ_BEST_SPEED") != "SZ_BEST_SPEED"


def _data(c):
    d = np.ascontiguousarray(c["data"]())
    r = REC[c["name"]]
    if hashlib.md5(d.tobytes()).hexdigest() != r["input_md5"]:
        pytest.skip(f"{c['name']}: this numpy generates a different input than the recorded one")
    return d, r


def _mask(stream, r):
    """bytes the reference itself leaves undefined (see tools/record_reference_outputs.py) are zeroed on both sides"""
    b = bytearray(stream)
    for i in r.get("masked_bytes", []):
        if i < len(b):
            b[i] = 0
    return bytes(b)


def _oracle_params(oracle, c):
    conf = dict(ref_cases.BASE_CONF)
    conf.update(c["conf"])
    return oracle.default_params(
        sample_distance=int(conf["sampleDistance"]), pred_threshold=float(conf["predThreshold"]),
        max_quant_intervals=int(conf["max_quant_intervals"]), quantization_intervals=int(conf["quantization_intervals"]),
        with_regression=1 if conf["withLinearRegression"] == "YES" else 0, sz_mode=0,
        protect_value_range=1 if conf["protectValueRange"] == "YES" else 0, psnr=float(conf["psnr"]), norm_err=float(conf["normErr"]),
        conf_rel_bound_ratio=float(conf["relBoundRatio"]))


PLAIN = [c for c in ref_cases.CASES if c["mode"] != PW_REL and not _wrapped(c)]
PWR = [c for c in ref_cases.CASES if c["mode"] == PW_REL]
WRAPPED = [c for c in ref_cases.CASES if _wrapped(c)]


@pytest.mark.parametrize("c", PLAIN, ids=[c["name"] for c in PLAIN])
def test_oracle_reproduces_recorded_reference_output(oracle, c):
    d, r = _data(c)
    stream, _ = oracle.compress(d, c["mode"], c["abs"], c["rel"], params=_oracle_params(oracle, c))
    assert len(stream) == r["stream_bytes"], c["name"]
    assert hashlib.md5(_mask(stream, r)).hexdigest() == r["stream_md5"], c["name"]
    if "decoded_md5" in r:
        dec = oracle.decompress(stream, d.shape, d.dtype)
        assert hashlib.md5(dec.tobytes()).hexdigest() == r["decoded_md5"], c["name"]


def _hip_roundtrip(c, tmp_path):
    import sz_amd
    d, r = _data(c)
    cfg = str(tmp_path / "sz.config")
    ref_cases.write_config(cfg, c["conf"])
    assert sz_amd.SZ_Init(cfg) == 0
    try:
        stream = sz_amd.SZ_compress_args(d, c["mode"], c["abs"], c["rel"], c["pwr"])
        dec = sz_amd.SZ_decompress(stream, d.shape, d.dtype) if d.size > 20 else None
    finally:
        sz_amd.SZ_Finalize()
    return d, r, stream, dec


@pytest.mark.gpu
@pytest.mark.parametrize("c", PLAIN + PWR, ids=[c["name"] for c in PLAIN + PWR])
def test_hip_reproduces_recorded_reference_output(built, c, tmp_path):
    d, r, stream, dec = _hip_roundtrip(c, tmp_path)
    assert len(stream) == r["stream_bytes"], c["name"]
    assert hashlib.md5(_mask(stream, r)).hexdigest() == r["stream_md5"], c["name"]
    if dec is not None:
        assert hashlib.md5(dec.tobytes()).hexdigest() == r["decoded_md5"], c["name"]


@pytest.mark.gpu
@pytest.mark.parametrize("c", WRAPPED, ids=[c["name"] for c in WRAPPED])
def test_hip_lossless_stage_round_trip(built, c, tmp_path):
    """szMode != SZ_BEST_SPEED: the wrapped bytes depend on the zstd / zlib build, the decoded values must be the reference's."""
    d, r, stream, dec = _hip_roundtrip(c, tmp_path)
    assert hashlib.md5(dec.tobytes()).hexdigest() == r["decoded_md5"], c["name"]
    assert abs(len(stream) - r["stream_bytes"]) <= 0.02 * r["stream_bytes"] + 64   # same content through a different zstd/zlib version


STORED = [c for c in ref_cases.CASES if "stream_file" in REC[c["name"]] and REC[c["name"]].get("decoded_md5")]


@pytest.mark.gpu
@pytest.mark.parametrize("c", STORED, ids=[c["name"] for c in STORED])
def test_hip_decodes_reference_made_stream(built, c):
    import sz_amd
    r = REC[c["name"]]
    stream = open(os.path.join(HERE, "golden", r["stream_file"]), "rb").read()
    assert hashlib.md5(stream).hexdigest() == r["stream_md5"]
    dec = sz_amd.SZ_decompress(stream, tuple(r["shape"]), np.dtype(r["dtype"]))
    assert hashlib.md5(dec.tobytes()).hexdigest() == r["decoded_md5"], c["name"]


# ---- the same replay WITHOUT a GPU: the product's HIP layer + host C compiled against the CPU shim (tests/sim), small cases only
# (one per path; the whole list is replayed on the GPU -- through the shim a case takes 5-30 s)
SMALL_NAMES = ("C1-f32", "mean-rand-f32", "2D-plane-70x90-f32", "sz14-S-20x24x40-f32", "1D-rand-5000-f32", "const-f64", "C1-zstd", "C1-gzip", "2D-plane-gzip-best")
SMALL = [c for c in ref_cases.CASES if c["name"] in SMALL_NAMES]


@pytest.mark.slow
@pytest.mark.parametrize("c", SMALL, ids=[c["name"] for c in SMALL])
def test_product_code_on_cpu_shim_reproduces_recorded_reference_output(built, c, tmp_path):
    import ctypes
    import sim_lib
    import sz_amd
    from sz_amd import api
    saved = api._lib
    try:
        api._lib = api._bind(ctypes.CDLL(sim_lib.shim_path()))
        d, r, stream, dec = _hip_roundtrip(c, tmp_path)
        if _wrapped(c):
            assert abs(len(stream) - r["stream_bytes"]) <= 0.02 * r["stream_bytes"] + 64
        else:
            assert len(stream) == r["stream_bytes"] and hashlib.md5(_mask(stream, r)).hexdigest() == r["stream_md5"], c["name"]
        if dec is not None:
            assert hashlib.md5(dec.tobytes()).hexdigest() == r["decoded_md5"], c["name"]
        if "stream_file" in r and r.get("decoded_md5"):           # and the reference-made stream itself
            assert sz_amd.SZ_Init(None) == 0
            ref_stream = open(os.path.join(HERE, "golden", r["stream_file"]), "rb").read()
            back = sz_amd.SZ_decompress(ref_stream, tuple(r["shape"]), np.dtype(r["dtype"]))
            sz_amd.SZ_Finalize()
            assert hashlib.md5(back.tobytes()).hexdigest() == r["decoded_md5"], c["name"]
    finally:
        api._lib = saved
