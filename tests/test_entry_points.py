"""The other entry points of the reference's C API (sz/include/sz.h:276-334, sz/src/sz.c:399-481, :579-681, :1083-1201) on the product
library, called the way a C caller does -- raw ctypes on the shared object, no Python convenience layer in between:
SZ_compress (bounds from sz.config), SZ_compress_args2 and SZ_decompress_args (caller-owned buffers), SZ_getMetadata,
SZ_compress_customize / SZ_decompress_customize and their _threadsafe forms.  Expected streams come from the oracle.
GPU: libszhip.so; without a GPU the same calls run on the product code compiled against the CPU shim."""
import ctypes
import os

import numpy as np
import pytest

from sz_amd.fields import s_field

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SZ_FLOAT, SZ_DOUBLE, ABS, REL = 0, 1, 0, 1
SZ_SCES, SZ_NSCS = 0, -1


class sz_metadata(ctypes.Structure):
    _fields_ = [("versionNumber", ctypes.c_int * 3), ("isConstant", ctypes.c_int), ("isLossless", ctypes.c_int), ("sizeType", ctypes.c_int),
                ("dataSeriesLength", ctypes.c_size_t), ("defactoNBBins", ctypes.c_int), ("conf_params", ctypes.c_void_p)]


def _exercise(L, oracle):
    from sz_amd.api import sz_params
    sz = ctypes.c_size_t
    vp = ctypes.c_void_p
    L.SZ_Init.argtypes = [ctypes.c_char_p]
    L.SZ_compress.restype = vp
    L.SZ_compress.argtypes = [ctypes.c_int, vp, ctypes.POINTER(sz)] + [sz] * 5
    L.SZ_compress_args2.restype = ctypes.c_int
    L.SZ_compress_args2.argtypes = [ctypes.c_int, vp, vp, ctypes.POINTER(sz), ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double] + [sz] * 5
    L.SZ_decompress_args.restype = sz
    L.SZ_decompress_args.argtypes = [ctypes.c_int, vp, sz, vp] + [sz] * 5
    L.SZ_getMetadata.restype = ctypes.POINTER(sz_metadata)
    L.SZ_getMetadata.argtypes = [vp]
    for name in ("SZ_compress_customize", "SZ_compress_customize_threadsafe"):
        f = getattr(L, name); f.restype = vp
        f.argtypes = [ctypes.c_char_p, vp, ctypes.c_int, vp] + [sz] * 5 + [ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_int)]
    for name in ("SZ_decompress_customize", "SZ_decompress_customize_threadsafe"):
        f = getattr(L, name); f.restype = vp
        f.argtypes = [ctypes.c_char_p, vp, ctypes.c_int, vp, sz] + [sz] * 5 + [ctypes.POINTER(ctypes.c_int)]
    L.free.argtypes = [vp]

    def take(p, n):
        b = ctypes.string_at(p, n); L.free(p); return b

    assert L.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config").encode()) == 0
    d = s_field(12, 16, 20)                                       # r3 = 12, r2 = 16, r1 = 20
    dims = (0, 0, 12, 16, 20)
    ref_cfg, _ = oracle.compress(d, oracle.ABS, 1e-4)            # what sz_speed.config asks for
    ref_3, _ = oracle.compress(d, oracle.ABS, 1e-3)
    dec_3 = oracle.decompress(ref_3, d.shape, d.dtype)

    # SZ_compress: everything from the configuration (sz.c:399-405)
    n = sz(0)
    assert take(L.SZ_compress(SZ_FLOAT, d.ctypes.data, ctypes.byref(n), *dims), n.value) == ref_cfg
    # SZ_compress_args2 into the caller's buffer (sz.c:407-417)
    buf = ctypes.create_string_buffer(d.nbytes + 1024)
    n = sz(0)
    assert L.SZ_compress_args2(SZ_FLOAT, d.ctypes.data, buf, ctypes.byref(n), ABS, 1e-3, 0.0, 0.0, *dims) == SZ_SCES
    assert buf.raw[:n.value] == ref_3
    # SZ_decompress_args into the caller's array (sz.c:579-591)
    out = np.empty_like(d)
    sbuf = ctypes.create_string_buffer(ref_3, len(ref_3))
    assert L.SZ_decompress_args(SZ_FLOAT, sbuf, len(ref_3), out.ctypes.data, *dims) == d.size
    assert np.array_equal(out.view(np.uint32), dec_3.view(np.uint32))
    # SZ_getMetadata (sz.c:683-760)
    md = L.SZ_getMetadata(sbuf).contents
    assert list(md.versionNumber) == [2, 1, 12] and md.isConstant == 0 and md.isLossless == 0 and md.sizeType == 8
    assert md.dataSeriesLength == d.size and md.defactoNBBins > 0
    # SZ_compress_customize: "SZ2.1" with no parameters = SZ_compress; an unknown name is refused (sz.c:1083-1148).
    # The configuration is global state: the SZ_compress_args2 call above left its derived bound in confparams_cpr->absErrBound
    # (sz_float.c:2867), so SZ_compress now compresses with 1e-3 -- in the reference and here.
    n = sz(0); st = ctypes.c_int(7)
    p = L.SZ_compress_customize(b"SZ2.1", None, SZ_FLOAT, d.ctypes.data, *dims, ctypes.byref(n), ctypes.byref(st))
    assert st.value == SZ_SCES and take(p, n.value) == ref_3
    st = ctypes.c_int(7)
    assert not L.SZ_compress_customize(b"no-such-compressor", None, SZ_FLOAT, d.ctypes.data, *dims, ctypes.byref(n), ctypes.byref(st)) and st.value == SZ_NSCS
    # ... _threadsafe: bounds from the parameter block, the global configuration stays as it is (sz.c:1150-1178)
    par = sz_params()
    par.errorBoundMode = ABS; par.absErrBound = 1e-3; par.relBoundRatio = 0; par.pw_relBoundRatio = 0
    n = sz(0); st = ctypes.c_int(7)
    p = L.SZ_compress_customize_threadsafe(b"SZ2.1", ctypes.byref(par), SZ_FLOAT, d.ctypes.data, *dims, ctypes.byref(n), ctypes.byref(st))
    assert st.value == SZ_SCES and take(p, n.value) == ref_3
    # SZ_decompress_customize and _threadsafe (sz.c:1180-1201)
    for name in ("SZ_decompress_customize", "SZ_decompress_customize_threadsafe"):
        st = ctypes.c_int(7)
        q = getattr(L, name)(b"SZ2.1", None, SZ_FLOAT, sbuf, len(ref_3), *dims, ctypes.byref(st))
        assert st.value == SZ_SCES and q
        got = np.frombuffer(ctypes.string_at(q, d.nbytes), dtype=np.float32).reshape(d.shape); L.free(q)
        assert np.array_equal(got.view(np.uint32), dec_3.view(np.uint32))
    # a double array through the caller-buffer pair, REL bound
    d64 = s_field(10, 12, 14, np.float64)
    r64, _ = oracle.compress(d64, oracle.REL, 0.0, 1e-3)
    buf = ctypes.create_string_buffer(d64.nbytes + 1024); n = sz(0)
    assert L.SZ_compress_args2(SZ_DOUBLE, d64.ctypes.data, buf, ctypes.byref(n), REL, 0.0, 1e-3, 0.0, 0, 0, 10, 12, 14) == SZ_SCES
    assert buf.raw[:n.value] == r64
    out64 = np.empty_like(d64)
    assert L.SZ_decompress_args(SZ_DOUBLE, buf, n.value, out64.ctypes.data, 0, 0, 10, 12, 14) == d64.size
    assert np.array_equal(out64.view(np.uint64), oracle.decompress(r64, d64.shape, d64.dtype).view(np.uint64))
    L.SZ_Finalize()


@pytest.mark.gpu
def test_c_api_entry_points_on_the_gpu(built, oracle):
    import sz_amd
    _exercise(ctypes.CDLL(sz_amd.api.lib_path()), oracle)


@pytest.mark.slow
def test_c_api_entry_points_on_the_cpu_shim(built, oracle):
    import sim_lib
    _exercise(ctypes.CDLL(sim_lib.shim_path()), oracle)


@pytest.mark.slow
def test_staged_host_copies_on_the_cpu_shim(built, oracle, monkeypatch):
    """the bulk host <-> device copies of the host-pointer API (8 MiB chunks through pinned buffers, four host threads) with 1 KiB chunks,
    so that a small array takes that path: same stream, same decoded values"""
    import sim_lib
    monkeypatch.setenv("SZ_HIP_STAGE_CHUNK_KB", "1")
    _exercise(ctypes.CDLL(sim_lib.shim_path()), oracle)


@pytest.mark.slow
@pytest.mark.parametrize("mode", ["fallback", "atomic", "index+table"])
def test_tile_ticket_modes_on_the_cpu_shim(built, oracle, monkeypatch, mode):
    """how a tile of the wavefront kernel learns which tile it is: the workgroup index with the tile computed (default), the same with the order
    table, the atomic ticket -- and the repetition of a call with the atomic ticket after a (here: simulated) wait time-out under the default"""
    import sim_lib
    if mode == "fallback": monkeypatch.setenv("SZ_HIP_TEST_TICKET_FALLBACK", "1")
    else: monkeypatch.setenv("SZ_HIP_TICKET_MODE", "0" if mode == "atomic" else "1")
    _exercise(ctypes.CDLL(sim_lib.shim_path()), oracle)


@pytest.mark.parametrize("switch", ["SZ_HIP_ENC32=0", "SZ_HIP_SEGENC=0;SZ_HIP_ENC32=0", "SZ_HIP_OUT_IN_PLACE=0", "SZ_HIP_SLICES=1", "SZ_HIP_SLICES=8", "SZ_HIP_PERM_Y=3", "SZ_HIP_SLICES=3"])
def test_alternative_forms_of_the_entropy_stage_on_the_cpu_shim(built, oracle, monkeypatch, switch):
    """round 4's switches keep the forms they replaced alive (k_encode beside k_encode32, everything after the sweep beside the slices, a workgroup per
    segment in k_permute), round 5's too (the one-pass look-back packing, off by default, also with the repetition of a call whose look-back gave
    up -- here: simulated --; the stream copied into a caller's buffer instead of written there): every one of them gives the same streams and decoded values"""
    import sim_lib
    for kv in switch.split(";"):
        k, v = kv.split("=")
        monkeypatch.setenv(k, v)
    _exercise(ctypes.CDLL(sim_lib.shim_path()), oracle)


def test_chain_fallback_on_the_cpu_shim(built, oracle, monkeypatch):
    """a sweep that gave up waiting for the regression coefficients (here: simulated) is answered by one repetition of the call with the
    coefficient chain finished before the sweep starts: same stream, same decoded values"""
    import sim_lib
    monkeypatch.setenv("SZ_HIP_TEST_CHAIN_FALLBACK", "1")
    _exercise(ctypes.CDLL(sim_lib.shim_path()), oracle)


@pytest.mark.gpu
def test_chain_fallback_on_the_gpu(built, oracle, monkeypatch):
    import sz_amd
    monkeypatch.setenv("SZ_HIP_TEST_CHAIN_FALLBACK", "1")
    _exercise(ctypes.CDLL(sz_amd.api.lib_path()), oracle)


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["SZ_HIP_SEGENC=0", "SZ_HIP_SEGENC=0;SZ_HIP_ENC32=0", "SZ_HIP_OUT_IN_PLACE=0"])
def test_packing_forms_on_the_gpu(built, oracle, monkeypatch, switch):
    import sz_amd
    for kv in switch.split(";"):
        k, v = kv.split("=")
        monkeypatch.setenv(k, v)
    _exercise(ctypes.CDLL(sz_amd.api.lib_path()), oracle)


@pytest.mark.gpu
def test_ticket_fallback_on_the_gpu(built, oracle, monkeypatch):
    import sz_amd
    monkeypatch.setenv("SZ_HIP_TEST_TICKET_FALLBACK", "1")
    _exercise(ctypes.CDLL(sz_amd.api.lib_path()), oracle)


@pytest.mark.parametrize("backend", ["ZSTD_COMPRESSOR", "GZIP_COMPRESSOR"])
def test_first_64k_of_a_wrapped_stream_is_decoded_into_a_fixed_buffer(built, oracle, tmp_path, backend):
    """sz_lossless_decompress65536bytes (utility.c:216-234, used by `sz -p`): the front of a zstd / gzip wrapped stream, decoded INTO 64 KiB and no
    further (a crafted frame must not be able to ask for terabytes).  On the CPU shim: the bytes are the plain stream's front (the
    mode bits of the parameter byte aside); a frame that claims an absurd size costs nothing but the 64 KiB."""
    import sim_lib
    import ref_cases
    from sz_amd.fields import s_field
    L = ctypes.CDLL(sim_lib.shim_path())
    L.SZ_Init.argtypes = [ctypes.c_char_p]
    szt = ctypes.c_size_t
    L.SZ_compress_args.restype = ctypes.c_void_p
    L.SZ_compress_args.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(szt), ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double] + [szt] * 5
    L.sz_lossless_decompress65536bytes.restype = ctypes.c_uint64
    L.sz_lossless_decompress65536bytes.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]
    libc = ctypes.CDLL(None); libc.free.argtypes = [ctypes.c_void_p]
    d = (s_field(40, 48, 64) + (np.random.default_rng(1).random((40, 48, 64)) - 0.5) * 2e-3).astype(np.float32)      # a stream beyond 64 KiB

    def compress(conf):
        cfg = str(tmp_path / "c.config")
        ref_cases.write_config(cfg, conf)
        assert L.SZ_Init(cfg.encode()) == 0
        n = szt(0)
        p = L.SZ_compress_args(0, d.ctypes.data, ctypes.byref(n), 0, 1e-4, 0.0, 0.0, 0, 0, 40, 48, 64)
        assert p
        b = ctypes.string_at(p, n.value); libc.free(p); L.SZ_Finalize()
        return b
    plain = compress({"szMode": "SZ_BEST_SPEED"})
    wrapped = compress({"szMode": "SZ_BEST_COMPRESSION", "losslessCompressor": backend})
    assert wrapped != plain and len(plain) > 65536
    code = 1 if backend == "ZSTD_COMPRESSOR" else 0           # defines.h: GZIP_COMPRESSOR 0, ZSTD_COMPRESSOR 1
    out = ctypes.c_void_p()
    buf = ctypes.create_string_buffer(wrapped, len(wrapped))
    assert L.sz_lossless_decompress65536bytes(code, buf, len(wrapped), ctypes.byref(out)) == 65536 and out.value
    front = ctypes.string_at(out.value, 65536); libc.free(out)
    assert front[:4] == plain[:4] and front[5:] == plain[5:65536]
    if backend == "ZSTD_COMPRESSOR":
        # a frame header that claims 2^39 bytes of content: the call still returns its 64 KiB (zeros: the frame is empty behind the header)
        crafted = bytes([0x28, 0xB5, 0x2F, 0xFD, 0xE0]) + (1 << 39).to_bytes(8, "little") + b"\x01\x00\x00"
        out = ctypes.c_void_p()
        assert L.sz_lossless_decompress65536bytes(1, crafted, len(crafted), ctypes.byref(out)) == 65536 and out.value
        libc.free(out)


@pytest.mark.parametrize("defer_min", ["1", "1000000000"])
def test_coefficient_sections_decoded_beside_the_device_on_the_cpu_shim(built, oracle, monkeypatch, defer_min):
    """round 5: in a decompression the regression coefficients' sections are only LOCATED while the header is read; their Huffman decode and chains run on their
    own threads beside the device's decode of the type array and are joined where the coefficients are shipped (dec_header::finish).  Deferred (threshold 1) and
    not (threshold never reached): the reference decoder's bits; a damaged stream comes back as an error or as some array, never as a crash."""
    import sim_lib
    import sz_amd
    from sz_amd import api
    from sz_amd.fields import m_field
    monkeypatch.setenv("SZ_HIP_DEC_DEFER_MIN", defer_min)
    saved = api._lib
    api._lib = api._bind(ctypes.CDLL(sim_lib.shim_path()))
    try:
        assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
        for d, eb in ((m_field(36), 1e-4), (m_field(30, np.float64), 1e-3)):
            ref, _ = oracle.compress(d, oracle.ABS, eb)
            want = oracle.decompress(ref, d.shape, d.dtype)
            dec = sz_amd.SZ_decompress(ref, d.shape, d.dtype)
            assert np.array_equal(dec.view(np.uint8), want.view(np.uint8))
            bad = bytearray(ref); bad[len(bad) // 3] ^= 0x55; bad[len(bad) // 3 + 7] ^= 0xff
            try:
                sz_amd.SZ_decompress(bytes(bad), d.shape, d.dtype)
            except Exception:
                pass
        sz_amd.SZ_Finalize()
    finally:
        api._lib = saved
