"""Case list of tests/golden/ref_recorded.json: inputs, sz.config keys and call arguments of every recorded run of the
UNMODIFIED reference (tools/record_reference_outputs.py made the file; tests/test_ref_recorded.py replays the list against the
oracle on the CPU and against the HIP library on the GPU).  Inputs are pure functions (sz_amd/fields.py, seeded numpy
generators); each record also holds the md5 of its input so that a drifting generator shows up as a skip, not as a false alarm.

A case = dict(name, data=callable -> ndarray, mode, abs, rel, pwr, conf = {sz.config key: value}).
"""
import os

import numpy as np

from sz_amd.fields import l_field, m_field, plane_field, reg_beside_lorenzo, s_field

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABS, REL, ABS_AND_REL, ABS_OR_REL, PSNR, NORM, PW_REL = 0, 1, 2, 3, 4, 5, 10
ABS_AND_PW_REL, ABS_OR_PW_REL, REL_AND_PW_REL, REL_OR_PW_REL = 11, 12, 13, 14

BASE_CONF = {   # tests/golden/sz_speed.config
    "withLinearRegression": "YES", "protectValueRange": "NO", "sampleDistance": 100, "quantization_intervals": 0,
    "max_quant_intervals": 65536, "predThreshold": 0.99, "szMode": "SZ_BEST_SPEED", "losslessCompressor": "ZSTD_COMPRESSOR",
    "gzipMode": "Gzip_BEST_SPEED", "zstdMode": "Zstd_HIGH_SPEED", "randomAccess": 0, "errorBoundMode": "ABS", "absErrBound": 1e-4,
    "relBoundRatio": 1e-3, "psnr": 80, "normErr": 0.05, "pw_relBoundRatio": 1e-2, "accelerate_pw_rel_compression": 1,
    "snapshotCmprStep": 5,
}


def write_config(path, conf):
    c = dict(BASE_CONF)
    c.update(conf or {})
    with open(path, "w") as f:
        f.write("[ENV]\ndataEndianType = LITTLE_ENDIAN_DATA\nsol_name = SZ\n[PARAMETER]\n")
        for k, v in c.items():
            f.write(f"{k} = {v}\n")


def _c1(dtype=np.float32):
    n = "testfloat_8_8_128.dat" if dtype == np.float32 else "testdouble_8_8_128.dat"
    return np.fromfile(os.path.join(ROOT, "tests", "golden", n), dtype=dtype).reshape(128, 8, 8)


def _rand(shape, dtype, seed):
    return np.random.default_rng(seed).random(shape).astype(dtype)


def _mean_zeros(dtype=np.float32):
    z = s_field(24, 24, 24, dtype)
    z[np.abs(z) < 0.7] = 0
    return z


def _walk(n, dtype, seed=7, scale=0.01):
    return np.ascontiguousarray((np.cumsum(np.random.default_rng(seed).standard_normal(n)) * scale).astype(dtype))


def _walk_sine(n, dtype):
    t = np.arange(n, dtype=np.float64)
    return np.ascontiguousarray((np.cumsum(np.random.default_rng(3).standard_normal(n)) * 2e-3 + np.sin(t / 50.0)).astype(dtype))


def _pos(shape, dtype, seed=11):
    """strictly positive, several decades of dynamic range (point-wise relative bounds)"""
    r = np.random.default_rng(seed)
    base = s_field(*shape, np.float64) if len(shape) == 3 else s_field(1, *shape, np.float64)[0] if len(shape) == 2 else np.sin(np.arange(shape[0]) / 37.0)
    return np.ascontiguousarray(np.exp(3.0 * base + 0.05 * r.standard_normal(shape)).astype(dtype))


def _signed_with_zeros(shape, dtype, seed=13):
    r = np.random.default_rng(seed)
    d = (s_field(*shape, np.float64) if len(shape) == 3 else s_field(1, *shape, np.float64)[0]) * np.exp(r.standard_normal(shape))
    d[r.random(shape) < 0.05] = 0.0
    return np.ascontiguousarray(d.astype(dtype))


def _pwr_mix(shape, dtype, seed, zeros=0.05, signed=True, first=0.37, neg_near_zero=False):
    """point-wise-relative data whose FIRST element is not zero (computeRangeSize_float_MSST19 starts nearZero from it: with a zero there
    nothing replaces the zeros): magnitudes over a few decades, optional sign changes and zeros, optionally a negative value of least magnitude"""
    r = np.random.default_rng(seed)
    sh3 = shape if len(shape) == 3 else (1,) * (3 - len(shape)) + tuple(shape) if len(shape) < 3 else (shape[0] * shape[1],) + tuple(shape[2:])
    base = s_field(*sh3, np.float64).reshape(shape)
    d = np.exp(2.0 * base + 0.05 * r.standard_normal(shape))
    if signed: d = d * np.sign(base + 0.15)
    d[d == 0] = 1.0
    if zeros: d[r.random(shape) < zeros] = 0.0
    d.reshape(-1)[0] = first
    if neg_near_zero: d.reshape(-1)[d.size // 3] = -1e-3
    return np.ascontiguousarray(d.astype(dtype))


def _fill(shape, dtype=np.float32, every=7, value=1e30):
    """a smooth field with fill values (every 7th value 1e30): with an ordinary bound the interval optimisers' quotient
    (|prediction error| / eb + 1) / 2 leaves the range of `unsigned long` (sz_float.c:4664, :5092)"""
    sh3 = (1,) * (3 - len(shape)) + tuple(shape)
    d = s_field(*sh3, dtype).reshape(shape).copy()
    d.ravel()[::every] = value
    return d


def _nan(shape, dtype=np.float32):
    d = _fill(shape, dtype, every=197)
    d.ravel()[5::1001] = np.nan
    return d


def case(name, data, mode=ABS, abs=1e-4, rel=0.0, pwr=0.0, **conf):
    return dict(name=name, data=data, mode=mode, abs=abs, rel=rel, pwr=pwr, conf=conf)


f32, f64 = np.float32, np.float64
CASES = [
    # ---- SZ 2.1, 3-D (sz_float.c:6527 / sz_double.c:5904)
    case("C1-f32", lambda: _c1(f32)),
    case("C1-f64", lambda: _c1(f64)),
    case("S-40x48x56-f32", lambda: s_field(40, 48, 56)),
    case("S-odd-17x25x38-f32", lambda: s_field(17, 25, 38)),
    case("M48-f32", lambda: m_field(48)),
    case("M40-f64-1e-5", lambda: m_field(40, f64), abs=1e-5),
    case("L-14x19x33-f32", lambda: l_field(14, 19, 33)),
    case("L-20x26x31-f64", lambda: l_field(20, 26, 31, f64)),
    case("reg-beside-lorenzo-f32", lambda: reg_beside_lorenzo(24, 40, 32)),
    case("reg-beside-lorenzo-f64", lambda: reg_beside_lorenzo(24, 40, 32, f64)),
    case("mean-rand-f32", lambda: _rand((13, 20, 17), f32, 0), abs=1e-2),
    # degenerate 3-D extents (round 4): the interval optimiser walks the flat array, not rows and planes (k_sample_walk)
    case("degenerate-8x1x66-f32", lambda: s_field(8, 1, 66), abs=1e-3),
    case("degenerate-16x1x40-f64", lambda: s_field(16, 1, 40, f64), abs=1e-3),
    case("degenerate-9x7x1-f32", lambda: s_field(9, 7, 1), abs=1e-3),
    case("mean-rand-f64", lambda: _rand((16, 18, 21), f64, 1), abs=1e-2),
    case("mean-zeros-f32", lambda: _mean_zeros(f32), abs=1e-3),
    case("mean-zeros-f64", lambda: _mean_zeros(f64), abs=1e-3),
    case("S-rel1e-3-f32", lambda: s_field(32, 40, 48), mode=REL, rel=1e-3),
    case("S-rel1e-3-f64", lambda: s_field(32, 40, 48, f64), mode=REL, rel=1e-3),
    case("S-abs-and-rel-f32", lambda: s_field(32, 40, 48), mode=ABS_AND_REL, abs=1e-3, rel=1e-4),
    case("S-abs-or-rel-f32", lambda: s_field(32, 40, 48), mode=ABS_OR_REL, abs=1e-3, rel=1e-4),
    case("S-abs-or-rel-f64", lambda: s_field(32, 40, 48, f64), mode=ABS_OR_REL, abs=1e-5, rel=1e-3),
    case("M-psnr80-f32", lambda: m_field(40), mode=PSNR, psnr=80),
    case("M-psnr60-f64", lambda: m_field(36, f64), mode=PSNR, psnr=60),
    case("M-norm-f32", lambda: m_field(40), mode=NORM, normErr=0.05),
    case("4D-3x4x20x24-f32", lambda: s_field(12, 20, 24).reshape(3, 4, 20, 24)),
    # ---- sz.config knobs (conf.c:170-260)
    case("M48-intervals64", lambda: m_field(48), quantization_intervals=64),
    case("M48-intervals1024", lambda: m_field(48), quantization_intervals=1024),
    case("M48-intervals65536", lambda: m_field(48), quantization_intervals=65536),
    case("S-intervals256-f64", lambda: s_field(30, 36, 40, f64), quantization_intervals=256),
    case("M48-sampleDistance10", lambda: m_field(48), sampleDistance=10),
    case("M48-sampleDistance50", lambda: m_field(48), sampleDistance=50),
    case("M48-predThreshold0.9", lambda: m_field(48), predThreshold=0.9),
    case("M48-predThreshold0.999", lambda: m_field(48), predThreshold=0.999),
    case("rand-maxq256", lambda: _rand((20, 24, 28), f32, 2) * f32(0.01) + s_field(20, 24, 28), abs=1e-5, max_quant_intervals=256),
    case("rand-maxq4096", lambda: _rand((20, 24, 28), f32, 2) * f32(0.01) + s_field(20, 24, 28), abs=1e-6, max_quant_intervals=4096),
    case("M48-protectValueRange", lambda: m_field(48), abs=1e-3, protectValueRange="YES"),
    case("S-protectValueRange-f64", lambda: s_field(24, 30, 36, f64), abs=1e-2, protectValueRange="YES"),
    # ---- SZ 2.1, 2-D (sz_float.c:5516 / sz_double.c:4850)
    case("2D-plane-70x90-f32", lambda: plane_field(70, 90)),
    case("2D-plane-128x160-f64", lambda: plane_field(128, 160, f64), abs=1e-5),
    case("2D-S-100x120-f32", lambda: s_field(1, 100, 120)[0]),
    case("2D-L-65x77-f32", lambda: l_field(1, 65, 77)[0]),
    case("2D-rand-50x60-f32", lambda: _rand((50, 60), f32, 4), abs=1e-2),
    case("2D-plane-rel-f32", lambda: plane_field(96, 96), mode=REL, rel=1e-4),
    case("2D-plane-intervals128", lambda: plane_field(70, 90), quantization_intervals=128),
    # ---- SZ 1.4 (withLinearRegression = NO): 3-D sz_float.c:946, 2-D :610
    case("sz14-S-20x24x40-f32", lambda: s_field(20, 24, 40), withLinearRegression="NO"),
    case("sz14-S-20x24x40-f64", lambda: s_field(20, 24, 40, f64), withLinearRegression="NO"),
    case("sz14-M32-f32", lambda: m_field(32), withLinearRegression="NO"),
    case("sz14-rand-f32", lambda: _rand((12, 14, 16), f32, 5), abs=1e-3, withLinearRegression="NO"),
    case("sz14-2D-plane-70x90-f32", lambda: plane_field(70, 90), withLinearRegression="NO"),
    case("sz14-2D-S-64x80-f64", lambda: s_field(1, 64, 80, f64)[0], abs=1e-6, withLinearRegression="NO"),
    case("sz14-S-rel-f32", lambda: s_field(20, 24, 40), mode=REL, rel=1e-3, withLinearRegression="NO"),
    # protectValueRange in the TightDataPointStorage container: the float writer records it (TightDataPointStorageF.c:610), the double one does not
    case("sz14-S-protect-f32", lambda: s_field(20, 24, 40), withLinearRegression="NO", protectValueRange="YES"),
    case("sz14-S-protect-f64", lambda: s_field(20, 24, 40, f64), withLinearRegression="NO", protectValueRange="YES"),
    case("const-protect-f64", lambda: np.full((9, 11, 13), -2.5, f64), abs=1e-3, protectValueRange="YES"),
    case("const-protect-f32", lambda: np.full((9, 11, 13), 3.25, f32), abs=1e-3, protectValueRange="YES"),
    # a fixed interval count in the SZ 1.4 container: the header's max_quant_intervals field then holds that count (conf.c:193-197)
    case("sz14-S-intervals256-f32", lambda: s_field(20, 24, 40), withLinearRegression="NO", quantization_intervals=256),
    case("sz14-2D-plane-intervals64-f32", lambda: plane_field(70, 90), withLinearRegression="NO", quantization_intervals=64),
    case("1D-walk-intervals128-f32", lambda: _walk(30000, f32), quantization_intervals=128),
    # ---- 1-D (sz_float.c:353, sz_double.c:260)
    case("1D-walk-30000-f32", lambda: _walk(30000, f32)),
    case("1D-walk-30000-f64", lambda: _walk(30000, f64), abs=1e-5),
    case("1D-walksine-50000-f32", lambda: _walk_sine(50000, f32), abs=1e-3),
    case("1D-walksine-50000-f64", lambda: _walk_sine(50000, f64), abs=1e-3),
    case("1D-rand-5000-f32", lambda: _rand((5000,), f32, 6), abs=1e-3),
    case("1D-walk-rel-f32", lambda: _walk(20000, f32), mode=REL, rel=1e-4),
    # ---- edge cases (sz_float.c:37, :2728, :526)
    case("const-10x12x14-f32", lambda: np.full((10, 12, 14), 3.25, f32), abs=1e-3),
    case("const-f64", lambda: np.full((9, 11, 13), -2.5, f64), abs=1e-3),
    case("tiny-2x3x3-f32", lambda: np.arange(18, dtype=f32).reshape(2, 3, 3), abs=1e-3),
    case("noise-raw-f32", lambda: np.random.default_rng(1).standard_normal((16, 16, 16)).astype(f32), abs=1e-7),
    # dimensions of 1 are dropped before the dispatch (filterDimension, sz.c:128-200): 3-D shapes that are 2-D, 4-D shapes that are 2-D, 27 values
    case("dim1-1x40x50-f32", lambda: _walk(2000, f32, seed=41).reshape(1, 40, 50), abs=1e-3),
    case("dim1-40x1x50-f64", lambda: _walk(2000, f64, seed=42).reshape(40, 1, 50), abs=1e-3),
    case("dim1-7x1x1x40-f32", lambda: _walk(280, f32, seed=43).reshape(7, 1, 1, 40), abs=1e-3),
    case("tiny-3x3x3-f32", lambda: _walk(27, f32, seed=44).reshape(3, 3, 3), abs=1e-3),
    # more mode x shape x type combinations (recorded at the end of round 2)
    case("4D-3x4x20x24-rel-f64", lambda: s_field(12, 20, 24, f64).reshape(3, 4, 20, 24), mode=REL, rel=1e-3),
    case("1D-rel-f64", lambda: _walk_sine(20000, f64), mode=REL, rel=1e-4),
    case("1D-psnr70-f32", lambda: _walk_sine(20000, f32), mode=PSNR, psnr=70),
    case("sz14-norm-f32", lambda: s_field(20, 24, 40), mode=NORM, normErr=0.05, withLinearRegression="NO"),
    case("2D-abs-and-rel-f64", lambda: s_field(1, 64, 80, f64)[0], mode=ABS_AND_REL, abs=1e-4, rel=1e-3),
    case("S-gzip-default-f64", lambda: s_field(20, 24, 40, f64), szMode="SZ_DEFAULT_COMPRESSION", losslessCompressor="GZIP_COMPRESSOR"),
    # ---- point-wise relative bounds (sz_float_pwr.c / sz_double_pwr.c; dispatch sz_float.c:2888-2893)
    case("pwr-pos-3D-f32", lambda: _pos((20, 24, 28), f32), mode=PW_REL, pwr=1e-2),
    case("pwr-pos-3D-f64", lambda: _pos((20, 24, 28), f64), mode=PW_REL, pwr=1e-3),
    case("pwr-signed-zeros-3D-f32", lambda: _signed_with_zeros((20, 24, 28), f32), mode=PW_REL, pwr=1e-2),
    case("pwr-signed-zeros-3D-f64", lambda: _signed_with_zeros((18, 20, 22), f64), mode=PW_REL, pwr=1e-2),
    case("pwr-pos-2D-f32", lambda: _pos((60, 72), f32), mode=PW_REL, pwr=1e-2),
    case("pwr-signed-2D-f64", lambda: _signed_with_zeros((50, 64), f64), mode=PW_REL, pwr=1e-3),
    case("pwr-pos-1D-f32", lambda: _pos((20000,), f32), mode=PW_REL, pwr=1e-2),
    case("pwr-pos-1D-f64", lambda: _pos((20000,), f64), mode=PW_REL, pwr=1e-3),
    case("pwr-negative-3D-f32", lambda: -_pos((16, 20, 24), f32, 17), mode=PW_REL, pwr=1e-2),
    # ... the same form with zeros that ARE replaced (first element non-zero; the "signed-zeros" cases above start with a zero, which leaves
    # nearZero = 0 and the zeros in place), a negative value of least magnitude, a negative first element, a fixed interval count, 4-D, other knobs
    case("pwr-zeros-pos-3D-f32", lambda: _pwr_mix((18, 22, 26), f32, 21, signed=False), mode=PW_REL, pwr=1e-2),
    case("pwr-zeros-signed-3D-f64", lambda: _pwr_mix((16, 20, 24), f64, 22), mode=PW_REL, pwr=1e-3),
    case("pwr-zeros-signed-2D-f32", lambda: _pwr_mix((56, 70), f32, 23), mode=PW_REL, pwr=1e-2),
    case("pwr-zeros-signed-1D-f64", lambda: _pwr_mix((12000,), f64, 24), mode=PW_REL, pwr=1e-3),
    case("pwr-neg-nearzero-3D-f32", lambda: _pwr_mix((16, 20, 24), f32, 25, neg_near_zero=True), mode=PW_REL, pwr=1e-2),
    case("pwr-first-negative-3D-f32", lambda: _pwr_mix((16, 20, 24), f32, 26, first=-0.6), mode=PW_REL, pwr=1e-2),
    case("pwr-intervals256-3D-f32", lambda: _pwr_mix((18, 22, 26), f32, 27, zeros=0), mode=PW_REL, pwr=1e-3, quantization_intervals=256),
    case("pwr-4D-f32", lambda: _pwr_mix((3, 20, 24), f32, 28).reshape(3, 4, 5, 24), mode=PW_REL, pwr=1e-2),
    case("pwr-sd10-3D-f64", lambda: _pwr_mix((16, 20, 24), f64, 29, zeros=0.02), mode=PW_REL, pwr=1e-2, sampleDistance=10),
    case("pwr-ratio0.1-2D-f64", lambda: _pwr_mix((48, 64), f64, 30, zeros=0), mode=PW_REL, pwr=1e-1),
    case("pwr-wrapped-zstd-3D-f32", lambda: _pwr_mix((18, 22, 26), f32, 31), mode=PW_REL, pwr=1e-2, szMode="SZ_BEST_COMPRESSION"),
    case("pwr-protect-3D-f64", lambda: _pwr_mix((16, 20, 24), f64, 32), mode=PW_REL, pwr=1e-2, protectValueRange="YES"),
    # MSST19 falling back to the raw copy: the reference stores the array whose zeros it has already overwritten (sz_float_pwr.c:2053-2058, :2077)
    case("pwr-raw-zeros-2D-f32", lambda: _pwr_mix((14, 80), f32, 35, zeros=0.05, signed=False) * np.random.default_rng(36).random((14, 80)).astype(f32), mode=PW_REL, pwr=2e-5),
    case("pwr-noaccel-3D-f32", lambda: _pos((20, 24, 28), f32), mode=PW_REL, pwr=1e-2, accelerate_pw_rel_compression=0),
    # the log-domain form (accelerate_pw_rel_compression = 0, or a ratio below 1e-5: sz_float.c:2837-2838) -- the one the MI355X build writes
    case("pwrlog-pos-3D-f64", lambda: _pos((20, 24, 28), f64), mode=PW_REL, pwr=1e-3, accelerate_pw_rel_compression=0),
    case("pwrlog-signed-zeros-3D-f32", lambda: _signed_with_zeros((20, 24, 28), f32), mode=PW_REL, pwr=1e-2, accelerate_pw_rel_compression=0),
    case("pwrlog-signed-zeros-3D-f64", lambda: _signed_with_zeros((18, 20, 22), f64), mode=PW_REL, pwr=1e-2, accelerate_pw_rel_compression=0),
    case("pwrlog-pos-2D-f32", lambda: _pos((60, 72), f32), mode=PW_REL, pwr=1e-2, accelerate_pw_rel_compression=0),
    case("pwrlog-signed-2D-f64", lambda: _signed_with_zeros((50, 64), f64), mode=PW_REL, pwr=1e-3, accelerate_pw_rel_compression=0),
    case("pwrlog-pos-1D-f32", lambda: _pos((20000,), f32), mode=PW_REL, pwr=1e-2, accelerate_pw_rel_compression=0),
    case("pwrlog-pos-1D-f64", lambda: _pos((20000,), f64), mode=PW_REL, pwr=1e-3, accelerate_pw_rel_compression=0),
    case("pwrlog-negative-3D-f32", lambda: -_pos((16, 20, 24), f32, 17), mode=PW_REL, pwr=1e-2, accelerate_pw_rel_compression=0),
    case("pwrlog-tight-3D-f32", lambda: _pos((20, 24, 28), f32), mode=PW_REL, pwr=5e-6),
    case("pwrlog-abs-and-pwr-3D-f32", lambda: _pos((20, 24, 28), f32), mode=ABS_AND_PW_REL, abs=1e-3, pwr=1e-2, accelerate_pw_rel_compression=0,
         pw_relBoundRatio=1e-2),
    case("pwrlog-rel-or-pwr-2D-f64", lambda: _pwr_mix((48, 64), f64, 33), mode=REL_OR_PW_REL, rel=1e-3, pwr=1e-3, accelerate_pw_rel_compression=0, pw_relBoundRatio=1e-3),
    case("pwrlog-abs-or-pwr-1D-f32", lambda: _pwr_mix((9000,), f32, 34), mode=ABS_OR_PW_REL, abs=1e-3, pwr=1e-2, accelerate_pw_rel_compression=0, pw_relBoundRatio=1e-2),
    case("pwrlog-4D-f32", lambda: _pos((3, 20, 24), f32).reshape(3, 4, 5, 24), mode=PW_REL, pwr=1e-2, accelerate_pw_rel_compression=0),
    # ---- lossless back end (utility.c:156-214): the wrapped bytes depend on the zstd/zlib build, the decoded values do not
    case("C1-zstd", lambda: _c1(f32), szMode="SZ_BEST_COMPRESSION"),
    case("C1-gzip", lambda: _c1(f32), szMode="SZ_BEST_COMPRESSION", losslessCompressor="GZIP_COMPRESSOR"),
    case("M40-f64-zstd-default", lambda: m_field(40, f64), szMode="SZ_DEFAULT_COMPRESSION"),
    case("2D-plane-gzip-best", lambda: plane_field(70, 90), szMode="SZ_BEST_COMPRESSION", losslessCompressor="GZIP_COMPRESSOR",
         gzipMode="Gzip_BEST_COMPRESSION"),
    # ---- fill values (round 3): interval-optimiser quotients beyond the range of `unsigned long` -- what the reference's x86-64 build does with them
    # (every 7th value: the stream would be larger than the array, the reference stores the raw values; every 97th: about 8 % of the optimiser's
    #  samples have a fill value in their stencil)
    case("fill-1e30-28x30x36-f32", lambda: _fill((28, 30, 36))),
    case("fill-1e30-sparse-32x40x48-f32", lambda: _fill((32, 40, 48), every=97)),
    case("fill-1e30-sparse-32x40x48-f64", lambda: _fill((32, 40, 48), f64, every=97)),
    case("fill-1e30-sparse-sz14-32x40x48-f32", lambda: _fill((32, 40, 48), every=97), withLinearRegression="NO"),
    case("fill-1e30-sparse-2d-200x300-f32", lambda: _fill((200, 300), every=97)),
    case("fill-1e30-1d-5000-f32", lambda: _fill((5000,))),
    case("fill-1e30-sparse-1d-50000-f32", lambda: _fill((50000,), every=97)),
    # NaN (not in element 0): the range scan skips it (`if (min > data) .. else if (max < data)`, dataCompression.c:97-113), the optimisers
    # put its quotient into the last bin, the quantisers keep it verbatim
    case("nan-sparse-32x40x48-f32", lambda: _nan((32, 40, 48))),
    case("nan-sparse-sz14-32x40x48-f32", lambda: _nan((32, 40, 48)), withLinearRegression="NO"),
    case("nan-sparse-1d-20000-f32", lambda: _nan((20000,))),
]
BY_NAME = {c["name"]: c for c in CASES}
