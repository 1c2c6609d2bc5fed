"""PW_REL in the reference's default, table-driven form (accelerate_pw_rel_compression = 1, "MSST19"; szh_msst.h, oracle/szo_msst_impl.h).
The oracle is pinned by nine recorded reference outputs (test_ref_recorded.py); here the product code is held against the oracle on
random arrays -- on the CPU shim (every lane a fibre) without a GPU, and through tools/gpu_fuzz.py on the GPU."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _cases():
    out = []
    for c in range(14):
        rng = np.random.default_rng(9000 + c)
        dt = np.float32 if c % 2 == 0 else np.float64
        if c % 7 == 5: shape = (int(rng.integers(30, 3000)),)
        elif c % 7 in (2, 6): shape = (int(rng.integers(2, 40)), int(rng.integers(3, 60)))
        else: shape = tuple(int(x) for x in rng.integers(2, 14, size=3))
        if int(np.prod(shape)) <= 20: shape = (5, 6, 7)
        base = np.exp(rng.uniform(0.5, 3.0) * rng.standard_normal(shape).cumsum(axis=-1) * 0.05 + 0.02 * rng.standard_normal(shape))
        if c % 3 == 1: base = base * np.sign(rng.standard_normal(shape) + 0.3)
        if c % 3 == 2: base = -base
        if c % 2 == 1: base[rng.random(shape) < 0.05] = 0.0
        d = np.ascontiguousarray(base.astype(dt))
        if c in (3, 8): d.reshape(-1)[0] = 0          # nearZero = 0: zeros are not replaced, the optimiser's walk skips them
        out.append((c, d, float(10.0 ** rng.uniform(-3.5, -1.0)), int(rng.choice([0, 0, 64, 512]))))
    return out


@pytest.mark.slow
@pytest.mark.parametrize("sweep", [0, 1], ids=["wavefront-kernel", "plane-sweep"])
@pytest.mark.parametrize("c,d,ratio,iv", _cases(), ids=[f"{c}-{d.dtype.name}-{'x'.join(map(str, d.shape))}" for c, d, _, _ in _cases()])
def test_product_code_on_cpu_shim_matches_oracle(built, oracle, c, d, ratio, iv, sweep, tmp_path, monkeypatch):
    """both mappings of the quantiser: the wavefront kernel's third quantiser (szh_pencil.h, fmt 2; the default for 2-D/3-D arrays) and the
    plane-by-plane sweep (szh_msst.h; SZ_HIP_MSST_SWEEP=1), which share nothing but the host-built tables"""
    monkeypatch.setenv("SZ_HIP_MSST_SWEEP", str(sweep))
    import sim_lib
    import sz_amd
    from sz_amd import api
    saved = api._lib
    try:
        api._lib = api._bind(ctypes.CDLL(sim_lib.shim_path()))
        import ref_cases
        cfg = str(tmp_path / "sz.config")
        ref_cases.write_config(cfg, {"quantization_intervals": iv, "accelerate_pw_rel_compression": 1})
        assert sz_amd.SZ_Init(cfg) == 0
        try:
            po = oracle.default_params(quantization_intervals=iv); po.pw_rel_bound_ratio = ratio; po.segment_size = 0; po.accelerate_pw_rel = 1
            if iv: po.max_quant_intervals = iv            # conf.c:196: a fixed count is also the recorded maximum
            ref, _ = oracle.compress(d, oracle.PW_REL, 0.0, 0.0, params=po)
            before = d.copy()
            got = sz_amd.SZ_compress_args(d, sz_amd.PW_REL, 0.0, 0.0, ratio)
            assert np.array_equal(before.view(np.uint8), d.view(np.uint8))      # the caller's zeros stay zeros (the reference overwrites them)
            assert got == ref
            back = sz_amd.SZ_decompress(ref, d.shape, d.dtype)
            assert np.array_equal(back.view(np.uint8), oracle.decompress(ref, d.shape, d.dtype).view(np.uint8))
        finally:
            sz_amd.SZ_Finalize()
    finally:
        api._lib = saved


@pytest.mark.gpu
@pytest.mark.parametrize("sweep", [0, 1], ids=["wavefront-kernel", "plane-sweep"])
def test_fuzz_msst19_against_oracle(built, sweep):
    env = dict(os.environ, SZ_HIP_MSST_SWEEP=str(sweep))
    n = "300" if sweep == 0 else "150"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz.py"), n, "47", "msst"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert f"{n} cases, 0 failures" in out.stdout, out.stdout[-3000:]


@pytest.mark.gpu
def test_msst19_round_trip_at_size(built):
    """a size the oracle is not run at: the decoded values respect the reference's own tolerance (measured on the reference: the table's
    cells overshoot the ratio by a few per cent at most), zeros and signs are kept, and a second decode gives the same bits"""
    import sz_amd
    from sz_amd.fields import s_field
    rng = np.random.default_rng(5)
    d = (np.abs(s_field(64, 192, 256, np.float64)) + 0.05) * np.exp(0.3 * rng.standard_normal((64, 192, 256)))
    d = (d * np.sign(s_field(64, 192, 256, np.float64) + 0.2)).astype(np.float32)
    d[rng.random(d.shape) < 0.01] = 0.0
    assert sz_amd.SZ_Init(None) == 0
    try:
        stream = sz_amd.SZ_compress_args(d, sz_amd.PW_REL, 0.0, 0.0, 1e-2)
        assert stream[3] & 0x08
        a = sz_amd.SZ_decompress(stream, d.shape, d.dtype)
        b = sz_amd.SZ_decompress(stream, d.shape, d.dtype)
    finally:
        sz_amd.SZ_Finalize()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    x, y = d.astype(np.float64), a.astype(np.float64)
    nz = x != 0
    nz.reshape(-1)[0] = False                                  # the sign of element 0 is never recorded (dataCompression.c:129)
    assert np.all(y[x == 0] == 0)
    assert np.all(np.sign(y[nz]) == np.sign(x[nz]))
    assert float((np.abs(y[nz] - x[nz]) / np.abs(x[nz])).max()) <= 1.1e-2
    assert len(stream) < d.nbytes / 3
