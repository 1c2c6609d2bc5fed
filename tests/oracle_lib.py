"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY (see oracle/szo.h)."""
import ctypes
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "liboracle.so")

ABS, REL, ABS_AND_REL, ABS_OR_REL, PSNR, NORM = 0, 1, 2, 3, 4, 5
PW_REL, ABS_AND_PW_REL, ABS_OR_PW_REL, REL_AND_PW_REL, REL_OR_PW_REL = 10, 11, 12, 13, 14
SZ_FLOAT, SZ_DOUBLE = 0, 1


class Params(ctypes.Structure):
    _fields_ = [("sample_distance", ctypes.c_int), ("pred_threshold", ctypes.c_float),
                ("max_quant_intervals", ctypes.c_uint), ("quantization_intervals", ctypes.c_uint),
                ("with_regression", ctypes.c_int), ("sz_mode", ctypes.c_int), ("gzip_mode", ctypes.c_int),
                ("protect_value_range", ctypes.c_int), ("data_endian", ctypes.c_int), ("sol_id", ctypes.c_int),
                ("psnr", ctypes.c_double), ("norm_err", ctypes.c_double), ("conf_rel_bound_ratio", ctypes.c_double),
                ("pw_rel_bound_ratio", ctypes.c_double), ("segment_size", ctypes.c_int),
                ("accelerate_pw_rel", ctypes.c_int)]


class Stages(ctypes.Structure):
    _fields_ = [("num_elements", ctypes.c_size_t), ("num_blocks", ctypes.c_size_t), ("reg_count", ctypes.c_size_t),
                ("total_unpred", ctypes.c_size_t), ("intervals", ctypes.c_uint), ("use_mean", ctypes.c_int),
                ("mean", ctypes.c_double), ("eb", ctypes.c_double), ("dense_pos", ctypes.c_double),
                ("mean_freq", ctypes.c_double), ("sample_freq", ctypes.c_double),
                ("codes", ctypes.POINTER(ctypes.c_int)), ("indicator", ctypes.POINTER(ctypes.c_ubyte)),
                ("unpred", ctypes.c_void_p), ("reg_params", ctypes.c_void_p),
                ("coeff_codes", ctypes.POINTER(ctypes.c_int)), ("coeff_dec", ctypes.c_void_p),
                ("coeff_unpred_count", ctypes.c_size_t * 4), ("coeff_unpred", ctypes.c_void_p * 4),
                ("code_len", ctypes.POINTER(ctypes.c_ubyte)),
                ("tree_bytes", ctypes.c_size_t), ("node_count", ctypes.c_size_t), ("huff_bytes", ctypes.c_size_t)]


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.szo_compress_args.restype = ctypes.POINTER(ctypes.c_ubyte)
        L.szo_compress_args.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.POINTER(ctypes.c_size_t), ctypes.c_int, ctypes.c_double,
                                        ctypes.c_double] + [ctypes.c_size_t] * 5 + [ctypes.c_void_p]
        L.szo_decompress.restype = ctypes.c_void_p
        L.szo_decompress.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t] + [ctypes.c_size_t] * 5
        L.szo_omp_compress.restype = ctypes.POINTER(ctypes.c_ubyte)
        L.szo_omp_compress.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p] + [ctypes.c_size_t] * 3 + [ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.szo_omp_decompress.restype = ctypes.c_void_p
        L.szo_omp_decompress.argtypes = [ctypes.c_int, ctypes.c_void_p] + [ctypes.c_size_t] * 3
        L.szo_free_stages.argtypes = [ctypes.c_void_p]
        L.free.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def default_params(**kw):
    """Defaults of SZ_Init(NULL), except conf_rel_bound_ratio = 1E-3 to match tests/golden/sz_speed.config."""
    p = Params()
    lib().szo_default_params(ctypes.byref(p))
    p.conf_rel_bound_ratio = 1e-3
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _dims5(shape):
    """numpy shape (slowest..fastest) -> (r5, r4, r3, r2, r1) with r1 fastest."""
    d = list(shape)[::-1] + [0] * (5 - len(shape))
    return d[4], d[3], d[2], d[1], d[0]


def compress(data, mode=ABS, abs_err=1e-4, rel=0.0, params=None, want_stages=False):
    """Returns (stream bytes, stages dict or None)."""
    L = lib()
    p = params if params is not None else default_params()
    data = np.ascontiguousarray(data)
    dt = SZ_FLOAT if data.dtype == np.float32 else SZ_DOUBLE
    n = ctypes.c_size_t(0)
    st = Stages()
    out = L.szo_compress_args(ctypes.byref(p), dt, data.ctypes.data, ctypes.byref(n), mode, abs_err, rel,
                              *_dims5(data.shape), ctypes.byref(st) if want_stages else None)
    if not out:
        raise RuntimeError("oracle compress failed")
    b = bytes(out[:n.value])
    L.free(out)
    sd = None
    if want_stages and st.num_elements:
        T = data.dtype
        ne, nb, rc, tu = st.num_elements, st.num_blocks, st.reg_count, st.total_unpred
        ncoef = 3 if sum(1 for d in data.shape if d > 1) == 2 else 4   # 2-D streams carry a|b|c

        def arr(ptr, count, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            a = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(count,))
            return a.copy()
        if not p.with_regression or data.ndim == 1:   # SZ 1.4 path (1-D always takes it): the struct's fields carry other things (szo_sz14_impl.h)
            sd = dict(num_elements=ne, exact_count=tu, intervals=st.intervals, req_length=st.use_mean, median=st.mean, eb=st.eb,
                      codes=arr(st.codes, ne, np.int32), lead=arr(st.indicator, tu, np.uint8), mid=arr(st.unpred, nb, np.uint8),
                      code_len=arr(st.code_len, 2 * st.intervals, np.uint8),
                      tree_bytes=st.tree_bytes, node_count=st.node_count, huff_bytes=st.huff_bytes)
            L.szo_free_stages(ctypes.byref(st))
            return b, sd
        sd = dict(num_elements=ne, num_blocks=nb, reg_count=rc, total_unpred=tu, intervals=st.intervals,
                  use_mean=st.use_mean, mean=st.mean, eb=st.eb, dense_pos=st.dense_pos, mean_freq=st.mean_freq,
                  sample_freq=st.sample_freq,
                  codes=arr(st.codes, ne, np.int32), indicator=arr(st.indicator, nb, np.uint8),
                  unpred=arr(st.unpred, tu, T), reg_params=arr(st.reg_params, ncoef * nb, T).reshape(ncoef, nb),
                  coeff_codes=arr(st.coeff_codes, ncoef * rc, np.int32).reshape(ncoef, rc),
                  coeff_dec=arr(st.coeff_dec, ncoef * rc, T).reshape(ncoef, rc),
                  coeff_unpred=[arr(st.coeff_unpred[e], st.coeff_unpred_count[e], T) for e in range(ncoef)],
                  code_len=arr(st.code_len, 2 * st.intervals, np.uint8),
                  tree_bytes=st.tree_bytes, node_count=st.node_count, huff_bytes=st.huff_bytes)
        L.szo_free_stages(ctypes.byref(st))
    return b, sd


def decompress(stream, shape, dtype):
    L = lib()
    dt = SZ_FLOAT if np.dtype(dtype) == np.float32 else SZ_DOUBLE
    buf = ctypes.create_string_buffer(stream, len(stream))
    r = L.szo_decompress(dt, buf, len(stream), *_dims5(shape))
    if not r:
        raise RuntimeError("oracle decompress failed")
    n = int(np.prod(shape))
    a = np.ctypeslib.as_array(ctypes.cast(r, ctypes.POINTER(np.ctypeslib.as_ctypes_type(np.dtype(dtype)))), shape=(n,)).copy()
    L.free(r)
    return a.reshape(shape)


def metrics(ori, dec):
    L = lib()
    ori = np.ascontiguousarray(ori).ravel()
    dec = np.ascontiguousarray(dec).ravel()
    ma, ps, nr = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    f = L.szo_metrics_f32 if ori.dtype == np.float32 else L.szo_metrics_f64
    f(ori.ctypes.data_as(ctypes.c_void_p), dec.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(ori.size),
      ctypes.byref(ma), ctypes.byref(ps), ctypes.byref(nr))
    return ma.value, ps.value, nr.value


def _dims3(shape):
    d = [1] * (3 - len(shape)) + list(shape)
    return d[0], d[1], d[2]


def omp_compress(data, eb, threads, meta, params=None):
    """The reference's OpenMP container for a 3-D array (oracle/szo_omp_impl.h; sz/src/sz_omp.c:63-358).  meta: the stream's first
    4 + MetaDataByteLength bytes (configuration state of the writing library)."""
    L = lib()
    data = np.ascontiguousarray(data)
    assert data.ndim == 3
    p = params or default_params()
    n = ctypes.c_size_t(0)
    mb = ctypes.create_string_buffer(bytes(meta), len(meta))
    out = L.szo_omp_compress(ctypes.byref(p), SZ_FLOAT if data.dtype == np.float32 else SZ_DOUBLE, data.ctypes.data, *data.shape, eb, threads, mb, len(meta), ctypes.byref(n))
    b = bytes(out[:n.value])
    L.free(out)
    return b


def omp_decompress(stream, meta_len, shape, dtype):
    L = lib()
    buf = ctypes.create_string_buffer(stream[meta_len:], len(stream) - meta_len)
    r = L.szo_omp_decompress(SZ_FLOAT if np.dtype(dtype) == np.float32 else SZ_DOUBLE, buf, *shape)
    n = int(np.prod(shape))
    a = np.ctypeslib.as_array(ctypes.cast(r, ctypes.POINTER(np.ctypeslib.as_ctypes_type(np.dtype(dtype)))), shape=(n,)).copy()
    L.free(r)
    return a.reshape(shape)
