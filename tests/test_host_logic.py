"""CPU tests of the product's host C (sz_amd/csrc/szhost.c, sz_api.c, sz_conf.c) and of the C-ABI surface.
No compute call reaches a GPU here; the oracle is only the checker."""
import ctypes
import os
import re

import numpy as np
import pytest

import sz_amd
from sz_amd import api
from sz_amd.fields import m_field, s_field

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L(built):
    return sz_amd.lib()


def test_library_exports_every_declared_symbol(L):
    """Every function declared in include/szhip.h and include/sz.h must be exported by libszhip.so."""
    names = set()
    for hdr in ("szhip.h", "sz.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b((?:szhip_|SZ_)[A-Za-z0-9_]+)\s*\(", text))
        names |= set(re.findall(r"\b(computeDataLength|computeDimension|filterDimension|convertSZParamsToBytes|convertBytesToSZParams)\s*\(", text))
    assert len(names) > 25
    for n in sorted(names):
        assert hasattr(L, n), f"{n} is declared but not exported"
    for g in ("confparams_cpr", "confparams_dec", "exe_params", "dataEndianType", "sysEndianType", "versionNumber"):
        ctypes.c_int.in_dll(L, g)
    # include/rw.h: the file helpers of the reference's examples
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rw.h")).read(), flags=re.S)
    rw = set(re.findall(r"\b((?:read|write)[A-Za-z0-9_]+)\s*\(", text))
    assert len(rw) >= 4
    for n in sorted(rw):
        assert hasattr(L, n), f"{n} is declared in rw.h but not exported"
    # include/sz_slab.h: the slab container of the multi-GPU path for C callers
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "sz_slab.h")).read(), flags=re.S)
    sl = set(re.findall(r"\b(sz_slab_[A-Za-z0-9_]+)\s*\(", text))
    assert len(sl) == 7
    for n in sorted(sl):
        assert hasattr(L, n), f"{n} is declared in sz_slab.h but not exported"
    # include/sz_omp.h (round 5): the reference's OpenMP entry points (sz/include/sz_omp.h:24-47) and the thread helpers of an OpenMP build
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "sz_omp.h")).read(), flags=re.S)
    om = set(re.findall(r"\b(SZ_compress_[a-z]+_[123]D_MDQ_openmp|decompressDataSeries_[a-z]+_[123]D_openmp|sz_[a-z_]*thread[a-z_]*|SZ_hip_set_omp_threads)\s*\(", text))
    assert len(om) >= 14, om
    for n in sorted(om):
        assert hasattr(L, n), f"{n} is declared in sz_omp.h but not exported"


def test_hdf5_plugin_exports_every_declared_symbol(built):
    """include/H5Z_SZ.h against sz_amd/h5z/libhdf5sz.so (built when the HDF5 C library is present)"""
    so = os.path.join(ROOT, "sz_amd", "h5z", "libhdf5sz.so")
    if not os.path.exists(so):
        pytest.skip("no HDF5 C library in this environment: the plugin was not built")
    P = ctypes.CDLL(so)
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "H5Z_SZ.h")).read(), flags=re.S)
    names = set(re.findall(r"\b((?:H5Z_SZ_|SZ_)[A-Za-z0-9_]+|checkCDValuesWithErrors)\s*\(", text))
    assert len(names) >= 9
    for n in sorted(names | {"H5PLget_plugin_type", "H5PLget_plugin_info"}):
        assert hasattr(P, n), f"{n} is declared but not exported by the plugin"
    for g in ("load_conffile_flag", "init_sz_flag", "cfgFile"):
        ctypes.c_int.in_dll(P, g)


def test_no_cpu_fallback_without_gpu(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(sz_amd.SZError):
        sz_amd.HipContext(0)
    assert sz_amd.SZ_Init(None) == 0
    with pytest.raises(sz_amd.SZError):
        sz_amd.SZ_compress_args(s_field(24, 24, 24), sz_amd.ABS, 1e-4)
    sz_amd.SZ_Finalize()


def test_conf_loader_defaults_and_file(L, tmp_path):
    assert sz_amd.SZ_Init(None) == 0
    p = sz_amd.conf_params()  # sz/src/conf.c:99-141
    assert (p.max_quant_intervals, p.maxRangeRadius, p.quantization_intervals) == (65536, 32768, 0)
    assert p.sampleDistance == 100 and abs(p.predThreshold - 0.99) < 1e-7
    assert p.szMode == sz_amd.SZ_BEST_COMPRESSION and p.errorBoundMode == sz_amd.PSNR and p.psnr == 90
    assert p.withRegression == 1 and p.gzipMode == 3 and p.protectValueRange == 0
    assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
    p = sz_amd.conf_params()
    assert p.szMode == sz_amd.SZ_BEST_SPEED and p.errorBoundMode == sz_amd.ABS and p.absErrBound == 1e-4
    assert p.relBoundRatio == 1e-3 and p.gzipMode == 3 and p.withRegression == 1
    bad = tmp_path / "bad.config"
    bad.write_text("[ENV]\nsol_name = SZ\n[PARAMETER]\nszMode = SZ_FASTEST\nerrorBoundMode = ABS\n")
    assert sz_amd.SZ_Init(str(bad)) == sz_amd.SZ_NSCS
    odd = tmp_path / "odd.config"
    odd.write_text("[ENV]\nsol_name = SZ\n[PARAMETER]\nquantization_intervals = 33\nszMode = SZ_BEST_SPEED\nerrorBoundMode = ABS\n")
    assert sz_amd.SZ_Init(str(odd)) == sz_amd.SZ_NSCS
    assert sz_amd.SZ_Init("/nonexistent/sz.config") == sz_amd.SZ_NSCS
    sz_amd.SZ_Finalize()


def _filter_model(r5, r4, r3, r2, r1):
    """Straight restatement of the reference's filterDimension (sz/src/sz.c:162-282)."""
    dims = [r1, r2, r3, r4, r5]
    dim = 0
    while dim < 5 and dims[dim] != 0:
        dim += 1
    c = dims[:]
    if dim >= 2:
        for d in range(dim - 1, -1, -1):
            if dims[d] != 1:
                continue
            if d == dim - 1:
                c[d] = 0
            else:
                for k in range(d, 4):
                    c[k] = c[k + 1]
                if dim == 5:
                    c[4] = 0
    return c


def test_filter_dimension(L):
    L.filterDimension.argtypes = [ctypes.c_size_t] * 5 + [ctypes.POINTER(ctypes.c_size_t)]
    L.computeDataLength.restype = ctypes.c_size_t
    L.computeDataLength.argtypes = [ctypes.c_size_t] * 5
    import itertools
    for ndim in range(1, 6):
        for vals in itertools.product([1, 2, 7], repeat=ndim):
            dims = list(vals) + [0] * (5 - ndim)  # r1..r5
            out = (ctypes.c_size_t * 5)()
            L.filterDimension(dims[4], dims[3], dims[2], dims[1], dims[0], out)
            assert list(out) == _filter_model(dims[4], dims[3], dims[2], dims[1], dims[0]), dims
    assert L.computeDataLength(0, 0, 4, 5, 6) == 120 and L.computeDataLength(0, 0, 0, 0, 9) == 9


def test_meta_bytes_match_reference_header(oracle, c1_data):
    ref, _ = oracle.compress(c1_data, oracle.ABS, 1e-4)
    meta = sz_amd.make_meta(np.float32, abs_bound=1e-4, vmin=float(c1_data.min()), vmax=float(c1_data.max()))
    assert meta == ref[:32]
    d = s_field(10, 12, 14, np.float64)
    ref, st = oracle.compress(d, oracle.REL, 0.0, 1e-3, want_stages=True)
    meta = sz_amd.make_meta(np.float64, err_mode=sz_amd.REL, abs_bound=st["eb"], rel_ratio=1e-3, vmin=float(d.min()), vmax=float(d.max()))
    assert meta == ref[:40]


class _Huff(ctypes.Structure):
    _fields_ = [("state_num", ctypes.c_int), ("n_nodes", ctypes.c_int), ("code", ctypes.POINTER(ctypes.c_uint64)),
                ("len", ctypes.POINTER(ctypes.c_uint8)), ("L", ctypes.POINTER(ctypes.c_uint32)), ("R", ctypes.POINTER(ctypes.c_uint32)),
                ("C", ctypes.POINTER(ctypes.c_uint32)), ("t", ctypes.POINTER(ctypes.c_uint8)), ("total_bits", ctypes.c_uint64)]


@pytest.mark.parametrize("case", ["c1", "m32", "single"])
def test_host_huffman_matches_reference_stream(L, oracle, c1_data, case):
    """Tree bytes, code lengths and the packed payload produced by the product's host Huffman code must equal the bytes
    inside the oracle's (reference-identical) stream; the device decode table must walk back to the same symbols."""
    if case == "c1":
        data, eb = c1_data, 1e-4
    elif case == "m32":
        data, eb = m_field(32), 1e-4
    else:
        data, eb = np.linspace(0, 1e-9, 24 * 24 * 24, dtype=np.float32).reshape(24, 24, 24) + np.float32(1.0), 1e-3
        data[0, 0, 0] = 2.0  # not constant, but every point lands in one quantisation bin or is unpredictable
    ref, st = oracle.compress(data, oracle.ABS, eb, want_stages=True)
    codes = st["codes"]
    hist = np.bincount(codes, minlength=st["intervals"]).astype(np.uint32)
    L.szhost_huff_build.restype = ctypes.POINTER(_Huff)
    L.szhost_huff_build.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.szhost_huff_tree_size.restype = ctypes.c_size_t
    L.szhost_huff_tree_size.argtypes = [ctypes.c_void_p]
    L.szhost_huff_tree_write.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.szhost_huff_encode_i32.restype = ctypes.c_size_t
    L.szhost_huff_encode_i32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.szhost_huff_from_bytes.restype = ctypes.POINTER(_Huff)
    L.szhost_huff_from_bytes.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L.szhost_huff_decode_table.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.szhost_huff_free.argtypes = [ctypes.c_void_p]
    h = L.szhost_huff_build(2 * st["intervals"], hist.ctypes.data, None, hist.size)
    assert h
    assert h.contents.n_nodes == st["node_count"]
    tb = L.szhost_huff_tree_size(h)
    assert tb == st["tree_bytes"]
    tree = ctypes.create_string_buffer(tb)
    L.szhost_huff_tree_write(h, tree)
    esz = data.dtype.itemsize
    off = 4 + 28 + 8 + 4 + esz + 12  # meta | N | block size | eb | intervals, treeBytes, nodeCount
    assert tree.raw == ref[off:off + tb]
    lens = np.ctypeslib.as_array(h.contents.len, shape=(2 * st["intervals"],))
    assert np.array_equal(lens[:st["intervals"]], st["code_len"][:st["intervals"]])
    payload = np.zeros(st["huff_bytes"] + 16, dtype=np.uint8)
    c32 = np.ascontiguousarray(codes, dtype=np.int32)
    n = L.szhost_huff_encode_i32(h, c32.ctypes.data, c32.size, payload.ctypes.data)
    assert n == st["huff_bytes"] and bytes(payload[:n]) == ref[len(ref) - n:]
    assert h.contents.total_bits == int(lens[codes].astype(np.uint64).sum())
    # decode table from the serialised tree (what the GPU decoder walks)
    h2 = L.szhost_huff_from_bytes(2 * st["intervals"], tree, st["node_count"])
    assert h2
    table = np.zeros(2 * st["node_count"], dtype=np.uint32)
    L.szhost_huff_decode_table(h2, table.ctypes.data)
    bits = np.unpackbits(payload[:n])
    out, node, pos = [], 0, 0
    if st["node_count"] == 1:
        out = [int(table[0] & 0xffff)] * 50
    else:
        while len(out) < 50:
            nx = int(table[2 * node + int(bits[pos])]); pos += 1
            if nx & 0x80000000:
                out.append(nx & 0xffff); node = 0
            else:
                node = nx
    assert out == codes[:50].tolist()
    L.szhost_huff_free(h); L.szhost_huff_free(h2)


def test_slab_container_roundtrip():
    from sz_amd import slab
    b = slab.slab_bounds(1030, 8)
    # cuts on multiples of the block edge (6): every slab's block grid is the whole array's; the remainder goes to the last slab
    assert b[0] == (0, 132) and b[-1][1] == 1030 and all(x[1] == y[0] for x, y in zip(b, b[1:])) and all(x[0] % 6 == 0 for x in b)
    assert slab.slab_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]          # fewer blocks than ranks: plane-wise
    streams = [os.urandom(10 + 7 * r) for r in range(8)]
    blob = slab.pack_container(np.float64, (1030, 64, 32), b, streams)
    dt, dims, bounds, got = slab.unpack_container(blob)
    assert dt == np.float64 and dims == (1030, 64, 32) and bounds == b and [bytes(g) for g in got] == streams


class _Coeffs(ctypes.Structure):
    _fields_ = [("reg_count", ctypes.c_size_t), ("codes", ctypes.POINTER(ctypes.c_int) * 4), ("unpred", ctypes.c_void_p * 4),
                ("unpred_count", ctypes.c_size_t * 4), ("prec", ctypes.c_double * 4)]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("use_mean", [0, 1])
def test_fast_coefficient_chain_equals_the_reference_loop(L, dtype, use_mean, monkeypatch):
    """(SZ_HIP_CHAIN_TAB_MIN=0: the threshold-table form on these short chains too -- by default a chain shorter than 150 000 steps takes the
    reference's loop unless its table is cached already, round 6.)  round 5: szhost_coeff_chain_one_p keeps only subtract / compare / select / add on the loop-carried path (candidate interval numbers and their
    exact thresholds come from the original coefficients); it must give the codes, decoded coefficients and verbatim values of the reference's
    loop (szhost_coeff_chain_one_ref = sz_float.c:7126-7152 literally) bit for bit -- on random walks, values ON interval boundaries, jumps beyond
    the code range, sign changes around zero, NaN / infinity."""
    monkeypatch.setenv("SZ_HIP_CHAIN_TAB_MIN", "0")
    rng = np.random.default_rng(11)
    is_double = int(dtype == np.float64)
    eb = 1e-3 if is_double else 1e-4
    nb = 6000
    for kind in range(6):
        ind = (rng.random(nb) < 0.3).astype(np.uint8) if kind % 2 else np.zeros(nb, dtype=np.uint8)
        co = np.zeros((4, nb), dtype=dtype)
        for e in range(4):
            prec = 0.025 * eb / (6 if e < 3 else 1)
            if kind == 0:
                v = np.cumsum((rng.random(nb) - 0.5) * 40 * prec)
            elif kind == 1:
                v = (rng.integers(-10, 10, nb) * 2 + 1) * prec + (rng.random(nb) - 0.5) * 1e-9 * prec      # on the steps of the interval number
            elif kind == 2:
                v = (rng.random(nb) - 0.5) * 1e6 * prec                                                     # beyond the code range: verbatim
            elif kind == 3:
                v = (rng.random(nb) - 0.5) * 1.5 * prec                                                     # around zero: signs flip, codes 0 and +-1
            elif kind == 4:
                v = np.cumsum((rng.random(nb) - 0.5) * 6 * prec); v[::53] = np.nan; v[7::101] = np.inf
            else:
                v = np.cumsum(np.tile(np.arange(-3, 4), nb // 7 + 1)[:nb] * prec) + (rng.random(nb) - 0.5) * 2 * prec
            co[e] = v.astype(dtype)
        outs = []
        for fn in (L.szhost_coeff_chain_one_p, L.szhost_coeff_chain_one_ref):
            c = np.ascontiguousarray(co.copy())
            st = _Coeffs()
            L.szhost_coeff_chain_begin(is_double, ind.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nb), ctypes.c_double(eb), 6, 6, 6, 4, ctypes.byref(st))
            for e in range(4):
                fn(is_double, c.ctypes.data_as(ctypes.c_void_p), ind.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nb), use_mean, e, ctypes.byref(st), None)
            rc = st.reg_count
            codes = [np.ctypeslib.as_array(st.codes[e], shape=(rc,)).copy() for e in range(4)]
            un = [ctypes.string_at(st.unpred[e], st.unpred_count[e] * c.itemsize) for e in range(4)]
            outs.append((c.tobytes(), codes, un))
            L.szhost_coeffs_free(ctypes.byref(st))
        assert outs[0][0] == outs[1][0], (kind, "decoded coefficients")
        for e in range(4):
            assert np.array_equal(outs[0][1][e], outs[1][1][e]), (kind, e, "codes")
            assert outs[0][2][e] == outs[1][2][e], (kind, e, "verbatim coefficients")


def test_table_decode_of_coefficient_codes_equals_the_walk(L):
    """round 5: szhost_huff_decode_i32 looks the first 11 bits of a code up in a table and walks the rest; it must return what the bit-by-bit walk
    of the reference's decode() (Huffman.c:314-360) returns -- the symbols that were encoded --, for short and for very long codes, and report a
    payload that ends early instead of reading past it."""
    rng = np.random.default_rng(5)
    L.szhost_huff_build.restype = ctypes.c_void_p
    L.szhost_huff_from_bytes.restype = ctypes.c_void_p
    L.szhost_huff_tree_size.restype = ctypes.c_size_t
    L.szhost_huff_encode_i32.restype = ctypes.c_size_t
    for kind in range(3):
        states = 4096
        n = 60000
        if kind == 0:    # a narrow peak: codes of 1 - 12 bits
            sym = np.clip(np.rint(rng.normal(2048, 3, n)), 1, states - 1).astype(np.int32)
        elif kind == 1:  # geometric frequencies: code lengths up to ~25 bits, far beyond the table
            sym = np.minimum(rng.geometric(0.5, n), 30).astype(np.int32) + 100
            sym[::997] = rng.integers(1, states, len(sym[::997]))
        else:            # flat over 3000 symbols: every code longer than 11 bits
            sym = rng.integers(1, 3001, n).astype(np.int32)
        hist = np.bincount(sym, minlength=states).astype(np.uint32)
        h = L.szhost_huff_build(states, hist.ctypes.data_as(ctypes.c_void_p), None, ctypes.c_size_t(states))
        assert h
        ts = L.szhost_huff_tree_size(ctypes.c_void_p(h))
        tree = np.zeros(ts, dtype=np.uint8)
        L.szhost_huff_tree_write(ctypes.c_void_p(h), tree.ctypes.data_as(ctypes.c_void_p))
        pay = np.zeros(n * 8 + 64, dtype=np.uint8)
        nbytes = L.szhost_huff_encode_i32(ctypes.c_void_p(h), sym.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), pay.ctypes.data_as(ctypes.c_void_p))
        n_nodes = 2 * int((hist > 0).sum()) - 1
        h2 = L.szhost_huff_from_bytes(states, tree.ctypes.data_as(ctypes.c_void_p), n_nodes)
        assert h2
        for cut in (0, 1, 9, nbytes // 2):
            out = np.full(n, -1, dtype=np.int32)
            ok = L.szhost_huff_decode_i32(ctypes.c_void_p(h2), pay.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nbytes - cut), ctypes.c_size_t(n),
                                          out.ctypes.data_as(ctypes.c_void_p))
            if cut == 0:
                assert ok == 1 and np.array_equal(out, sym), kind
            elif cut > 1:
                assert ok == 0, (kind, cut)            # (one byte less may still hold every code: the last byte is padded)
        # a short array takes the walk alone: same symbols
        out = np.full(100, -1, dtype=np.int32)
        assert L.szhost_huff_decode_i32(ctypes.c_void_p(h2), pay.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nbytes), ctypes.c_size_t(100), out.ctypes.data_as(ctypes.c_void_p)) == 1
        assert np.array_equal(out, sym[:100])
        L.szhost_huff_free(ctypes.c_void_p(h)); L.szhost_huff_free(ctypes.c_void_p(h2))


def test_chain_started_before_all_coefficients_are_there(L, monkeypatch):
    """round 5: szhost_coeff_chain_one_pa starts a chain while the tail of its coefficient array is still on its way (the caller raises *avail from another
    thread); the chain waits where it runs into the mark and gives what szhost_coeff_chain_one_p gives on the complete array, bit for bit."""
    import threading, time
    monkeypatch.setenv("SZ_HIP_CHAIN_TAB_MIN", "0")
    rng = np.random.default_rng(3)
    for dtype in (np.float32, np.float64):
        is_double = int(dtype == np.float64)
        eb = 1e-3 if is_double else 1e-4
        nb = 50000
        ind = np.zeros(nb, dtype=np.uint8)
        co = np.zeros((4, nb), dtype=dtype)
        for e in range(4):
            prec = 0.025 * eb / (6 if e < 3 else 1)
            v = np.cumsum((rng.random(nb) - 0.5) * 30 * prec); v[::977] += 1e5 * prec
            co[e] = v.astype(dtype)
        outs = []
        for gated in (0, 1):
            c = np.ascontiguousarray(co.copy())
            if gated:
                c[:, 20000:] = np.nan                                   # (not there yet: whatever the chain read there too early would show)
            st = _Coeffs()
            L.szhost_coeff_chain_begin(is_double, ind.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nb), ctypes.c_double(eb), 6, 6, 6, 4, ctypes.byref(st))
            avail = ctypes.c_size_t(20000 if gated else nb)
            def late():
                time.sleep(0.05)
                c[:, 20000:] = co[:, 20000:]
                avail.value = nb
            th = threading.Thread(target=late) if gated else None
            if th: th.start()
            for e in range(4):
                L.szhost_coeff_chain_one_pa(is_double, c.ctypes.data_as(ctypes.c_void_p), ind.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nb), 0, e, ctypes.byref(st), None,
                                            ctypes.byref(avail) if gated else None)
            if th: th.join()
            rc = st.reg_count
            outs.append((c.tobytes(), [np.ctypeslib.as_array(st.codes[e], shape=(rc,)).copy() for e in range(4)]))
            L.szhost_coeffs_free(ctypes.byref(st))
        assert outs[0][0] == outs[1][0]
        for e in range(4):
            assert np.array_equal(outs[0][1][e], outs[1][1][e])


def test_chain_table_cache_full_and_short_chains(L, monkeypatch):
    """round 6 (ADVICE): (a) once the 16 slots of the threshold-table cache are taken, a chain with a new precision still runs in the fast form on a
    table of its own -- same results as the reference's loop -- instead of building a table, throwing it away and walking the slow loop; (b) a short
    chain does not build a table at all (it costs more than it saves) and gives the same results."""
    rng = np.random.default_rng(23)
    nb = 4000
    ind = np.zeros(nb, dtype=np.uint8)
    L.szhost_coeff_chain_begin.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_Coeffs)]
    for fn in (L.szhost_coeff_chain_one_p, L.szhost_coeff_chain_one_ref):
        fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_Coeffs), ctypes.c_void_p]
    L.szhost_coeffs_free.argtypes = [ctypes.POINTER(_Coeffs)]

    def run(fn, co, eb):
        c = _Coeffs()
        work = co.copy()
        L.szhost_coeff_chain_begin(0, ind.ctypes.data, nb, eb, 6, 6, 6, 4, ctypes.byref(c))
        for e in range(4):
            fn(0, work.ctypes.data, ind.ctypes.data, nb, 0, e, ctypes.byref(c), None)
        out = (work.copy(), [np.ctypeslib.as_array(c.codes[e], shape=(nb,)).copy() for e in range(4)], [int(c.unpred_count[e]) for e in range(4)])
        L.szhost_coeffs_free(ctypes.byref(c))
        return out

    for minsteps in ("0", None):                      # 0: every chain builds / finds a table (24 precisions x 4 > 16 slots); default: none of these short chains does
        if minsteps is None:
            monkeypatch.delenv("SZ_HIP_CHAIN_TAB_MIN", raising=False)
        else:
            monkeypatch.setenv("SZ_HIP_CHAIN_TAB_MIN", minsteps)
        for k in range(24):
            eb = 1e-4 * (1.0 + 0.37 * k)
            co = np.cumsum((rng.random((4, nb)) - 0.5) * 30 * 0.025 * eb, axis=1).astype(np.float32)
            a, b = run(L.szhost_coeff_chain_one_p, co, eb), run(L.szhost_coeff_chain_one_ref, co, eb)
            assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and a[2] == b[2], (minsteps, k)
            for e in range(4):
                assert np.array_equal(a[1][e], b[1][e]), (minsteps, k, e)
