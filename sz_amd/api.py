"""ctypes mirror of include/sz.h and include/szhip.h (sz_amd/csrc/libszhip.so)."""
import ctypes
import os
import subprocess

import numpy as np

ABS, REL, VR_REL, ABS_AND_REL, ABS_OR_REL, PSNR, NORM, PW_REL = 0, 1, 1, 2, 3, 4, 5, 10
SZ_FLOAT, SZ_DOUBLE = 0, 1
SZ_BEST_SPEED, SZ_BEST_COMPRESSION, SZ_DEFAULT_COMPRESSION = 0, 1, 2
SZ_SCES, SZ_NSCS = 0, -1

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")


def lib_path():
    """The product library; SZ_AMD_LIB names another build of it (development: libszhip_dev.so, variants)."""
    return os.environ.get("SZ_AMD_LIB") or os.path.join(_CSRC, "libszhip.so")


class SZError(RuntimeError):
    pass


class sz_params(ctypes.Structure):  # include/sz.h (reference sz/include/sz.h:164-198)
    _fields_ = [("dataType", ctypes.c_int), ("max_quant_intervals", ctypes.c_uint),
                ("quantization_intervals", ctypes.c_uint), ("maxRangeRadius", ctypes.c_uint),
                ("sol_ID", ctypes.c_int), ("losslessCompressor", ctypes.c_int), ("sampleDistance", ctypes.c_int),
                ("predThreshold", ctypes.c_float), ("szMode", ctypes.c_int), ("gzipMode", ctypes.c_int),
                ("errorBoundMode", ctypes.c_int), ("absErrBound", ctypes.c_double), ("relBoundRatio", ctypes.c_double),
                ("psnr", ctypes.c_double), ("normErr", ctypes.c_double), ("pw_relBoundRatio", ctypes.c_double),
                ("segment_size", ctypes.c_int), ("pwr_type", ctypes.c_int), ("protectValueRange", ctypes.c_int),
                ("fmin", ctypes.c_float), ("fmax", ctypes.c_float), ("dmin", ctypes.c_double), ("dmax", ctypes.c_double),
                ("snapshotCmprStep", ctypes.c_int), ("predictionMode", ctypes.c_int),
                ("accelerate_pw_rel_compression", ctypes.c_int), ("plus_bits", ctypes.c_int),
                ("randomAccess", ctypes.c_int), ("withRegression", ctypes.c_int)]


class szhip_params(ctypes.Structure):  # include/szhip.h
    _fields_ = [("sample_distance", ctypes.c_int), ("pred_threshold", ctypes.c_float),
                ("max_quant_intervals", ctypes.c_uint), ("quantization_intervals", ctypes.c_uint), ("flags", ctypes.c_uint)]


class szhip_stats(ctypes.Structure):  # include/szhip.h
    _fields_ = [("ms_total", ctypes.c_double), ("ms_prequant", ctypes.c_double), ("ms_quant", ctypes.c_double),
                ("ms_entropy", ctypes.c_double), ("ms_host", ctypes.c_double),
                ("n_elements", ctypes.c_uint64), ("n_blocks", ctypes.c_uint64), ("n_reg_blocks", ctypes.c_uint64),
                ("n_unpred", ctypes.c_uint64), ("intervals", ctypes.c_uint), ("use_mean", ctypes.c_int),
                ("out_bytes", ctypes.c_uint64), ("quant_kernel_launches", ctypes.c_uint64), ("vmin", ctypes.c_double), ("vmax", ctypes.c_double), ("chain_overlapped", ctypes.c_int), ("quant_kernel", ctypes.c_int), ("packing", ctypes.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class szhost_meta(ctypes.Structure):  # sz_amd/csrc/szhost.h
    _fields_ = [("data_type", ctypes.c_int), ("err_mode", ctypes.c_int), ("abs_bound", ctypes.c_double),
                ("rel_ratio", ctypes.c_double), ("psnr", ctypes.c_double), ("pwr_ratio", ctypes.c_double), ("vmin", ctypes.c_double),
                ("vmax", ctypes.c_double), ("opt_quant_mode", ctypes.c_int), ("data_endian", ctypes.c_int),
                ("sz_mode", ctypes.c_int), ("gzip_mode", ctypes.c_int), ("sample_distance", ctypes.c_int),
                ("pred_threshold", ctypes.c_float), ("sol_id", ctypes.c_int), ("max_quant_intervals", ctypes.c_uint),
                ("quantization_intervals", ctypes.c_uint), ("protect_value_range", ctypes.c_int)]


def build_library(force=False):
    """Compile every HIP/C source for gfx950 into sz_amd/csrc/libszhip.so (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", _CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-C", _CSRC])
    return lib_path()


_lib = None


def lib():
    """Load the product library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise SZError(f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    _lib = _bind(ctypes.CDLL(p))
    return _lib


def _bind(L):
    """Declare the C prototypes on a loaded library object."""
    sz = ctypes.c_size_t
    L.SZ_Init.argtypes = [ctypes.c_char_p]; L.SZ_Init.restype = ctypes.c_int
    L.SZ_Init_Params.argtypes = [ctypes.POINTER(sz_params)]; L.SZ_Init_Params.restype = ctypes.c_int
    L.SZ_Finalize.argtypes = []; L.SZ_Finalize.restype = None
    L.SZ_compress.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(sz)] + [sz] * 5
    L.SZ_compress.restype = ctypes.c_void_p
    L.SZ_compress_args.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(sz), ctypes.c_int,
                                   ctypes.c_double, ctypes.c_double, ctypes.c_double] + [sz] * 5
    L.SZ_compress_args.restype = ctypes.c_void_p
    L.SZ_decompress.argtypes = [ctypes.c_int, ctypes.c_void_p, sz] + [sz] * 5
    L.SZ_decompress.restype = ctypes.c_void_p
    L.SZ_hip_last_stats.argtypes = [ctypes.POINTER(szhip_stats)]; L.SZ_hip_last_stats.restype = ctypes.c_int
    L.SZ_hip_set_device.argtypes = [ctypes.c_int]
    L.szhip_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]; L.szhip_create.restype = ctypes.c_int
    L.szhip_destroy.argtypes = [ctypes.c_void_p]; L.szhip_destroy.restype = None
    L.szhip_last_error.argtypes = [ctypes.c_void_p]; L.szhip_last_error.restype = ctypes.c_char_p
    L.szhip_minmax.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, sz,
                               ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.szhip_minmax.restype = ctypes.c_int
    L.szhip_compress.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, sz, sz, sz, ctypes.c_double,
                                 ctypes.POINTER(szhip_params), ctypes.c_char_p, sz, ctypes.c_int,
                                 ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz), ctypes.POINTER(szhip_stats)]
    L.szhip_compress.restype = ctypes.c_int
    if hasattr(L, "szhip_pool_create"):
        L.szhip_pool_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int]; L.szhip_pool_create.restype = ctypes.c_int
        L.szhip_pool_destroy.argtypes = [ctypes.c_void_p]; L.szhip_pool_destroy.restype = None
        L.szhip_pool_submit.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, sz, sz, sz, ctypes.c_double,
                                        ctypes.POINTER(szhip_params), ctypes.c_char_p, sz, ctypes.c_int, ctypes.c_void_p, sz, ctypes.POINTER(ctypes.c_int)]
        L.szhip_pool_submit.restype = ctypes.c_int
        L.szhip_pool_wait.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz), ctypes.POINTER(szhip_stats)]
        L.szhip_pool_wait.restype = ctypes.c_int
    L.szhip_decompress.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, sz, sz, sz, sz, sz,
                                   ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(szhip_stats)]
    L.szhip_decompress.restype = ctypes.c_int
    L.szhip_compress_sz14.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, sz, sz, sz, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_double, ctypes.POINTER(szhip_params), ctypes.c_char_p, sz, ctypes.c_int,
                                      ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz), ctypes.POINTER(szhip_stats)]
    L.szhip_compress_sz14.restype = ctypes.c_int
    L.szhip_decompress_sz14.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, sz, sz, sz, sz, sz,
                                        ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(szhip_stats)]
    L.szhip_decompress_sz14.restype = ctypes.c_int
    L.szhip_debug_fetch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, sz]; L.szhip_debug_fetch.restype = ctypes.c_int
    if hasattr(L, "szhip_compress_omp"):
        L.szhip_compress_omp.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, sz, sz, sz, ctypes.c_double, ctypes.c_int,
                                         ctypes.POINTER(szhip_params), ctypes.c_char_p, sz, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz), ctypes.POINTER(szhip_stats)]
        L.szhip_compress_omp.restype = ctypes.c_int
        L.szhip_decompress_omp.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, sz, sz, sz, sz, sz,
                                           ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(szhip_stats)]
        L.szhip_decompress_omp.restype = ctypes.c_int
    L.szhost_write_meta.argtypes = [ctypes.POINTER(szhost_meta), ctypes.c_ubyte, ctypes.c_char_p]
    L.szhost_write_meta.restype = sz
    L.free.argtypes = [ctypes.c_void_p]
    return L


def _dims5(shape):
    """numpy shape (slowest..fastest) -> (r5, r4, r3, r2, r1), r1 fastest, unused = 0 (reference convention)."""
    d = list(shape)[::-1] + [0] * (5 - len(shape))
    return d[4], d[3], d[2], d[1], d[0]


def _dtype_code(a):
    if a.dtype == np.float32:
        return SZ_FLOAT
    if a.dtype == np.float64:
        return SZ_DOUBLE
    raise SZError("only float32 / float64 arrays")


def SZ_Init(config_path=None):
    return lib().SZ_Init(config_path.encode() if config_path else None)


def SZ_Init_Params(params):
    return lib().SZ_Init_Params(ctypes.byref(params))


def SZ_Finalize():
    lib().SZ_Finalize()


def conf_params():
    """The live `confparams_cpr` struct (callers of the reference poke it directly)."""
    p = ctypes.POINTER(sz_params).in_dll(lib(), "confparams_cpr")
    if not p:
        raise SZError("SZ_Init has not been called")
    return p.contents


def SZ_compress_args(data, errBoundMode, absErrBound=0.0, relBoundRatio=0.0, pwrBoundRatio=0.0):
    """SZ_compress_args(dataType, data, &outSize, mode, abs, rel, pwr, r5..r1); returns the stream as bytes."""
    a = np.ascontiguousarray(data)
    n = ctypes.c_size_t(0)
    p = lib().SZ_compress_args(_dtype_code(a), a.ctypes.data, ctypes.byref(n), errBoundMode, absErrBound, relBoundRatio,
                               pwrBoundRatio, *_dims5(a.shape))
    if not p:
        raise SZError("SZ_compress_args returned NULL")
    out = ctypes.string_at(p, n.value)
    lib().free(p)
    return out


def SZ_compress(data):
    a = np.ascontiguousarray(data)
    n = ctypes.c_size_t(0)
    p = lib().SZ_compress(_dtype_code(a), a.ctypes.data, ctypes.byref(n), *_dims5(a.shape))
    if not p:
        raise SZError("SZ_compress returned NULL")
    out = ctypes.string_at(p, n.value)
    lib().free(p)
    return out


def SZ_decompress(stream, shape, dtype):
    dt = SZ_FLOAT if np.dtype(dtype) == np.float32 else SZ_DOUBLE
    buf = ctypes.create_string_buffer(bytes(stream), len(stream))
    p = lib().SZ_decompress(dt, buf, len(stream), *_dims5(shape))
    if not p:
        raise SZError("SZ_decompress returned NULL")
    n = int(np.prod(shape))
    ct = ctypes.c_float if dt == SZ_FLOAT else ctypes.c_double
    arr = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ct)), shape=(n,)).copy().reshape(shape)
    lib().free(p)
    return arr


def SZ_hip_last_stats():
    s = szhip_stats()
    lib().SZ_hip_last_stats(ctypes.byref(s))
    return s


def make_meta(dtype, err_mode=ABS, abs_bound=0.0, rel_ratio=0.0, vmin=0.0, vmax=0.0, sz_mode=SZ_BEST_SPEED, gzip_mode=1,
              sample_distance=100, pred_threshold=0.99, max_quant_intervals=65536, quantization_intervals=0,
              protect_value_range=0):
    """Version + flag + parameter bytes an SZ 2.1 regression-type stream starts with (szhost_write_meta)."""
    m = szhost_meta()
    m.data_type = SZ_FLOAT if np.dtype(dtype) == np.float32 else SZ_DOUBLE
    m.err_mode, m.abs_bound, m.rel_ratio, m.psnr = err_mode, abs_bound, rel_ratio, 0.0
    m.vmin, m.vmax = vmin, vmax
    m.opt_quant_mode = 1 if quantization_intervals == 0 else 0
    m.data_endian, m.sz_mode, m.gzip_mode = 0, sz_mode, gzip_mode
    m.sample_distance, m.pred_threshold, m.sol_id = sample_distance, pred_threshold, 101
    m.max_quant_intervals, m.quantization_intervals, m.protect_value_range = max_quant_intervals, quantization_intervals, protect_value_range
    buf = ctypes.create_string_buffer(64)
    flags = 0x80 | 0x40 | (0x04 if protect_value_range else 0)
    n = lib().szhost_write_meta(ctypes.byref(m), flags, buf)
    return buf.raw[:n]


class HipPool:
    """szhip_pool (include/szhip.h): `lanes` contexts and host threads on one GPU; submit() queues one szhip_compress call and returns
    a ticket, wait() blocks for it.  The arguments of a call (input, parameter bytes, output buffer) must stay alive until its wait()."""

    def __init__(self, device=0, lanes=2):
        self._h = ctypes.c_void_p()
        self.lanes = lanes
        rc = lib().szhip_pool_create(ctypes.byref(self._h), device, lanes)
        if rc != 0:
            raise SZError(f"szhip_pool_create failed ({rc})")
        self._keep = {}

    def close(self):
        if self._h:
            lib().szhip_pool_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def submit(self, ptr, on_device, shape3, dtype, eb, meta, params, out_ptr, out_cap):
        """Stream into the caller's device buffer (out_ptr, out_cap).  Returns the ticket."""
        t = ctypes.c_int(-1)
        p = params or szhip_params(100, 0.99, 65536, 0)
        rc = lib().szhip_pool_submit(self._h, 0 if np.dtype(dtype) == np.float32 else 1, ptr, int(on_device), shape3[0], shape3[1], shape3[2], eb,
                                     ctypes.byref(p), meta, len(meta), 2, out_ptr, out_cap, ctypes.byref(t))
        if rc != 0:
            raise SZError(f"szhip_pool_submit failed ({rc})")
        self._keep[t.value] = (p, meta)
        return t.value

    def wait(self, ticket):
        """Returns (size, stats); raises if the call failed.  SZHIP_CONSTANT (1: the array lies within the bound of one value -- the caller
        writes the reference's constant stream, as SZ_compress_args does) is a result, not a failure: size 0, stats with the range."""
        out = ctypes.c_void_p(); n = ctypes.c_size_t(0); st = szhip_stats()
        rc = lib().szhip_pool_wait(self._h, ticket, ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
        self._keep.pop(ticket, None)
        if rc == 1:
            return 0, st
        if rc != 0:
            raise SZError(f"pooled szhip_compress failed ({rc})")
        return n.value, st


class HipContext:
    """Thin owner of one szhip_ctx (one HIP stream + device workspaces).  Device-resident entry points for bench.py."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        rc = lib().szhip_create(ctypes.byref(self._h), device)
        if rc != 0:
            raise SZError(f"szhip_create failed ({rc}): no usable HIP device; there is no CPU fallback")

    def close(self):
        if self._h:
            lib().szhip_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self, rc, what):
        raise SZError(f"{what} failed ({rc}): {lib().szhip_last_error(self._h).decode(errors='replace')}")

    def minmax(self, ptr, on_device, n, dtype):
        lo, hi = ctypes.c_double(), ctypes.c_double()
        rc = lib().szhip_minmax(self._h, 0 if np.dtype(dtype) == np.float32 else 1, ptr, int(on_device), n,
                                ctypes.byref(lo), ctypes.byref(hi))
        if rc:
            self._err(rc, "szhip_minmax")
        return lo.value, hi.value

    def compress(self, ptr, on_device, shape3, dtype, eb, meta, params=None, out_on_device=False):
        """Returns (bytes | device pointer int, size, stats)."""
        p = params or szhip_params(100, 0.99, 65536, 0)
        out = ctypes.c_void_p()
        n = ctypes.c_size_t(0)
        st = szhip_stats()
        rc = lib().szhip_compress(self._h, 0 if np.dtype(dtype) == np.float32 else 1, ptr, int(on_device), shape3[0], shape3[1],
                                  shape3[2], eb, ctypes.byref(p), meta, len(meta), int(out_on_device), ctypes.byref(out),
                                  ctypes.byref(n), ctypes.byref(st))
        if rc == 1:                      # SZHIP_CONSTANT (range-from-data): nothing encoded, st.vmin / st.vmax say why
            return None, 0, st
        if rc:
            self._err(rc, "szhip_compress")
        if out_on_device:
            return out.value, n.value, st
        b = ctypes.string_at(out.value, n.value)
        lib().free(out)
        return b, n.value, st

    def compress_omp(self, ptr, on_device, shape3, dtype, eb, thread_num, meta, params=None, out_on_device=False):
        """The reference's OpenMP container (szhip_compress_omp; sz/src/sz_omp.c:63-358).  Returns (bytes | device pointer int, size, stats)."""
        p = params or szhip_params(100, 0.99, 65536, 0)
        out = ctypes.c_void_p()
        n = ctypes.c_size_t(0)
        st = szhip_stats()
        rc = lib().szhip_compress_omp(self._h, 0 if np.dtype(dtype) == np.float32 else 1, ptr, int(on_device), shape3[0], shape3[1], shape3[2],
                                      eb, thread_num, ctypes.byref(p), meta, len(meta), int(out_on_device), ctypes.byref(out), ctypes.byref(n),
                                      ctypes.byref(st))
        if rc:
            self._err(rc, "szhip_compress_omp")
        if out_on_device:
            return out.value, n.value, st
        b = ctypes.string_at(out.value, n.value)
        lib().free(out)
        return b, n.value, st

    def decompress_omp(self, stream_ptr, stream_on_device, stream_len, body_off, shape3, dtype, out_ptr, out_on_device):
        st = szhip_stats()
        rc = lib().szhip_decompress_omp(self._h, 0 if np.dtype(dtype) == np.float32 else 1, stream_ptr, int(stream_on_device), stream_len,
                                        body_off, shape3[0], shape3[1], shape3[2], out_ptr, int(out_on_device), ctypes.byref(st))
        if rc:
            self._err(rc, "szhip_decompress_omp")
        return st

    def debug_fetch(self, which, count, dtype):
        """Copy an internal workspace of the last call to host (tests only; see szhip_debug_fetch)."""
        a = np.zeros(count, dtype=dtype)
        rc = lib().szhip_debug_fetch(self._h, which, a.ctypes.data, a.nbytes)
        if rc:
            self._err(rc, "szhip_debug_fetch")
        return a

    def decompress(self, stream_ptr, stream_on_device, stream_len, body_off, shape3, dtype, out_ptr, out_on_device):
        st = szhip_stats()
        rc = lib().szhip_decompress(self._h, 0 if np.dtype(dtype) == np.float32 else 1, stream_ptr, int(stream_on_device), stream_len,
                                    body_off, shape3[0], shape3[1], shape3[2], out_ptr, int(out_on_device), ctypes.byref(st))
        if rc:
            self._err(rc, "szhip_decompress")
        return st
