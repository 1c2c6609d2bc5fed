"""1-D arrays (SZ_compress_float_1D_MDQ, sz/src/sz_float.c:353-540; SZ_compress_double_1D_MDQ, sz/src/sz_double.c:260-400;
decompressDataSeries_float_1D, sz/src/szd_float.c:185-282): CPU-side checks of the oracle's restatement.  Its pin is tests/test_ref_recorded.py (recorded
1-D outputs of the unmodified reference, round 2); these tests add a second, independent statement of the same chain written with
numpy scalars, and the restatement's own decoder.  The GPU runs are in test_gpu_parity.py."""
import numpy as np
import pytest


def _series(n, dtype, seed):
    rng = np.random.default_rng(seed)
    x = np.cumsum(rng.standard_normal(n)) * 0.01 + np.sin(np.arange(n) * 0.003)
    a, b = n // 5, n // 5 + max(1, n // 50)
    x[a:b] += 30.0 * rng.standard_normal(b - a)
    x[3 * n // 4:3 * n // 4 + n // 40] = 0.0
    return np.ascontiguousarray(x.astype(dtype))


def _keep_bits(x, median, req_len, T):
    """the leading req_len bits of (x - median), plus the median (dataCompression.c:454-477)"""
    U, nb = (np.uint32, 32) if T == np.float32 else (np.uint64, 64)
    norm = T(x - median)
    bits = np.array([norm], dtype=T).view(U)[0]
    ign = nb - req_len
    bits = U((int(bits) >> ign) << ign)
    return T(np.array([bits], dtype=U).view(T)[0] + median)


def _chain(data, eb, intervals, median, req_len):
    """codes and reconstructions of the 1-D chain, one numpy scalar operation per reference operation"""
    T = data.dtype.type
    eb = T(eb); recip = T(1) / eb; interval = T(2) * eb
    radius = intervals // 2
    check = T(intervals - 1) * eb
    codes = np.zeros(data.size, dtype=np.int32)
    rec = np.zeros(data.size, dtype=T)
    rec[0] = _keep_bits(data[0], median, req_len, T)
    pred = rec[1] = _keep_bits(data[1], median, req_len, T)
    for i in range(2, data.size):
        x = data[i]
        err = abs(T(x - pred))
        ok = err < check
        if ok:
            state = (int(T(T(err * recip) + T(1))) >> 1) if T == np.float32 else int(T(T(T(err * recip) + T(1)) * T(0.5)))
            step = T(T(state) * interval)
            p2 = T(pred + step) if x >= pred else T(pred - step)
            code = radius + state if x >= pred else radius - state
            if T == np.float32 and abs(T(x - p2)) > eb:
                ok = False
        if ok:
            codes[i], pred = code, p2
        else:
            pred = _keep_bits(x, median, req_len, T)
        rec[i] = pred
    return codes, rec


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,eb", [(21, 1e-3), (100, 1e-2), (3000, 1e-3), (3000, 1e-5), (20000, 1e-4)])
def test_oracle_1d_codes_and_values_against_a_second_statement(oracle, dtype, n, eb):
    d = _series(n, dtype, seed=n + 1)
    stream, st = oracle.compress(d, oracle.ABS, eb, want_stages=True)
    dec = oracle.decompress(stream, d.shape, dtype)
    if stream[3] & 0x10:      # stored raw: nothing more to look at
        assert np.array_equal(dec, d)
        return
    assert stream[3] & 0xC0 == 0x40                       # the SZ 1.4 container even with the regression switch on
    with np.errstate(over="ignore", invalid="ignore"):
        codes, rec = _chain(d, eb, st["intervals"], dtype(st["median"]), st["req_length"])
    assert np.array_equal(st["codes"], codes)
    assert st["exact_count"] == int((codes == 0).sum())
    assert codes[0] == 0 and codes[1] == 0
    assert np.array_equal(dec.view(np.uint8), rec.view(np.uint8))
    if dtype == np.float32:   # the double chain of the reference does not re-check the bound
        assert float(np.abs(dec.astype(np.float64) - d.astype(np.float64)).max()) <= eb


def test_oracle_1d_ignores_the_regression_switch_and_counts_samples(oracle):
    d = _series(5000, np.float32, seed=3)
    a, _ = oracle.compress(d, oracle.ABS, 1e-3)
    b, _ = oracle.compress(d, oracle.ABS, 1e-3, params=oracle.default_params(with_regression=0))
    assert a == b
    # interval optimiser (sz_float.c:5070): positions 2, 2+sd, ..., previous-value predictor
    p = oracle.default_params()
    sd, thr, maxr = 100, 0.99, 32768
    pos = np.arange(2, d.size, sd)
    err = np.abs((d[pos - 1] - d[pos]).astype(np.float32)).astype(np.float64)
    ri = np.minimum(((err / float(np.float32(1e-3)) + 1) / 2).astype(np.int64), maxr - 1)
    hist = np.bincount(ri, minlength=maxr)
    target = int(np.float32(len(pos)) * np.float32(thr))
    i = int(np.argmax(np.cumsum(hist) > target))
    want = max(32, 1 << int(np.ceil(np.log2(2 * (i + 1)))))
    _, st = oracle.compress(d, oracle.ABS, 1e-3, want_stages=True)
    assert st["intervals"] == want
    del p
