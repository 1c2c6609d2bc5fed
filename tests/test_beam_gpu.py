"""Round 5: the beam mapping of the SZ 2.1 sweep (sz_amd/csrc/szh_beam.h: k_beam, k_reg_points) on a real MI355X, through the C ABI.
(a) Shapes that exercise what is new against the oracle, byte for byte and bit for bit: several beams along k and j (hand-offs through LDS
and through HBM granules), ragged extents in every dimension, a last k-beam with four live lanes, float / double, the mean shortcut,
regression blocks (their points quantised by k_reg_points, passed through by the sweep), regression next to Lorenzo blocks at the array's
faces.  (b) BASELINE configs[2] at FULL size -- the 512^3 M-field -- against what the unmodified reference gave for it
(tests/golden/anchors.json, recorded by tools/record_reference_m512.py): stream length, md5, regression blocks, decoded md5, PSNR to 6 d.p.,
the bound.  (c) The same streams with the beam switched off (k_pencil): one switch, identical bytes."""
import hashlib
import json
import os

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


def _cases():
    from sz_amd.fields import l_field, m_field, near_zero_planes, reg_beside_lorenzo, s_field
    rng = np.random.default_rng(3)

    def noisy(shape, dt=np.float32, amp=3e-4):
        return (s_field(*shape, dt) + (rng.random(shape) - 0.5) * amp).astype(dt)
    mean = np.full((24, 40, 64), 1.25, dtype=np.float32); mean[::3, ::5, ::7] += 0.37
    mean += ((rng.random(mean.shape) - 0.5) * 1e-5).astype(np.float32)
    return {
        "one-workgroup": (s_field(12, 8, 32), 1e-4),
        "two-by-two-beams": (s_field(10, 40, 36), 1e-4),                 # the second k-beam has four live lanes
        "ragged-3x3": (s_field(9, 70, 68), 1e-4),
        "noisy-codes": (noisy((20, 33, 100)), 1e-4),
        "long-in-i": (noisy((150, 36, 40)), 1e-4),                       # blocks of lines without per-lane line checks
        "f64": (s_field(41, 70, 36, np.float64), 1e-3),
        "f64-noisy-long": (noisy((70, 36, 36), np.float64, 3e-4), 1e-3),
        "mean": (mean, 1e-4),
        "regression-half": (m_field(40), 1e-4),
        "regression-all": (l_field(30, 52, 72), 1e-4),
        "regression-f64": (m_field(36, np.float64), 1e-3),
        "regression-beside-lorenzo": (reg_beside_lorenzo(24, 40, 32), 1e-4),
        "regression-near-zero": (near_zero_planes(20, 36, 40), 1e-4),
        "S128": (s_field(128, 128, 128), 1e-4),
        "M128": (m_field(128), 1e-4),
    }


@pytest.mark.parametrize("name", list(_cases().keys()))
def test_beam_equals_the_oracle(sz, oracle, name, monkeypatch):
    d, eb = _cases()[name]
    ref, _ = oracle.compress(d, oracle.ABS, eb)
    want = oracle.decompress(ref, d.shape, d.dtype)
    for beam in ("2", "1", "0"):              # 2: the beam wherever it covers the array; 1: the library's own choice; 0: never
        monkeypatch.setenv("SZ_HIP_BEAM", beam)
        got = sz.SZ_compress_args(d, sz.ABS, eb)
        st = sz.SZ_hip_last_stats()
        if beam == "2":
            assert int(st.quant_kernel) == 2, "the beam sweep must be the one that ran"
        assert got == ref, (name, beam)
        dec = sz.SZ_decompress(ref, d.shape, d.dtype)
        assert np.array_equal(dec.view(np.uint8), want.view(np.uint8)), (name, beam)


@pytest.mark.parametrize("name", [k for k in _cases().keys() if k.startswith("regression") or k == "M128"])
def test_beam_fed_while_it_runs_equals_the_oracle(sz, oracle, name, monkeypatch):
    """round 5: the sweep of an array with regression blocks is launched while the host's coefficient chains still run; k_reg_points follows the
    chains in slices of block rows on the second stream and the sweep waits per plane (szh_beam.h wait_fed).  Same stream as with everything
    finished first, three times over (a hand-off that is late once in a while must not go unnoticed), and the statistics say which order ran."""
    d, eb = _cases()[name]
    ref, _ = oracle.compress(d, oracle.ABS, eb)
    monkeypatch.setenv("SZ_HIP_FEED_MIN_REG", "1")
    monkeypatch.setenv("SZ_HIP_CHAIN_EARLY_PIECE", "8")        # (the chains start on the first 8 coefficients of every array while the rest is on its way)
    for fed in ("1", "0", "1", "1"):
        monkeypatch.setenv("SZ_HIP_BEAM_FEED", fed)
        got = sz.SZ_compress_args(d, sz.ABS, eb)
        st = sz.SZ_hip_last_stats()
        assert got == ref, (name, fed)
        if fed == "0":
            assert int(st.chain_overlapped) != 2
        elif d.shape[1] * d.shape[2] % 128 == 0:          # (planes of whole cache lines: the library's condition for the feed)
            assert int(st.chain_overlapped) == 2, "the fed order must be the one that ran"


def test_m_field_512_full_size(sz):
    """BASELINE configs[2] at full size against the recorded output of the unmodified reference."""
    from sz_amd.fields import m_field
    A = json.load(open(os.path.join(ROOT, "tests", "golden", "anchors.json")))["M512_f32_abs1e-4_best_speed"]
    d = m_field(512)
    got = sz.SZ_compress_args(d, sz.ABS, 1e-4)
    st = sz.SZ_hip_last_stats()
    assert int(st.quant_kernel) == 2 and int(st.chain_overlapped) == 2      # the beam sweep, fed while the coefficient chains run
    assert len(got) == A["stream_bytes"] == 42782959
    assert int(st.n_reg_blocks) == 303450 and int(st.n_blocks) == A["blocks"] == 614125
    masked = bytearray(got); masked[19] = 0           # parameter byte 15: never written by the reference (heap garbage there, zero here)
    assert hashlib.md5(bytes(masked)).hexdigest() == A["md5_byte19_zeroed"]
    dec = sz.SZ_decompress(got, d.shape, d.dtype)
    assert hashlib.md5(dec.tobytes()).hexdigest() == A["decoded_md5"]
    dd, ee = d.astype(np.float64), dec.astype(np.float64)
    err = float(np.abs(dd - ee).max())
    assert err <= 1e-4 and abs(err - A["max_abs_err"]) < 1e-12
    psnr = 20 * np.log10(float(dd.max() - dd.min())) - 10 * np.log10(float(((dd - ee) ** 2).mean()))
    assert round(psnr, 6) == A["psnr"]
