"""sz_slab_compress_multi (include/sz_slab.h, sz_amd/csrc/sz_slab_multi.cpp) on the GPU box: one device is what the box has, so the driver runs
with ndev = 1 -- but through RCCL (communicator over one device: the all-reduce of the range and the all-gather of the sub-streams are
real RCCL calls, looked up in librccl.so at run time), with the gathered device copy compared against the host streams
(SZ_SLAB_VERIFY_GATHER).  The container is byte for byte sz_slab_compress's; the N > 1 exchange is covered on the CPU shim
(tests/test_distributed_cpu.py) with the host transport."""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Info(ctypes.Structure):
    _fields_ = [("devices", ctypes.c_int), ("used_rccl", ctypes.c_int), ("gathered_bytes", ctypes.c_size_t), ("seconds_total", ctypes.c_double),
                ("seconds_slowest_slab", ctypes.c_double)]


@pytest.mark.gpu
def test_multi_device_driver_with_rccl_on_one_device(built, oracle):
    import sz_amd
    from sz_amd.fields import s_field
    L = sz_amd.lib()
    szt = ctypes.c_size_t
    L.sz_slab_compress.restype = ctypes.c_void_p
    L.sz_slab_compress.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(szt), ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, szt, szt, szt, ctypes.c_int]
    L.sz_slab_compress_multi.restype = ctypes.c_void_p
    L.sz_slab_compress_multi.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(szt), ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, szt, szt, szt,
                                         ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(Info)]
    libc = ctypes.CDLL(None); libc.free.argtypes = [ctypes.c_void_p]
    assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
    os.environ["SZ_SLAB_VERIFY_GATHER"] = "1"
    try:
        for dt, mode, absb, rel in ((np.float32, sz_amd.ABS, 1e-4, 0.0), (np.float64, sz_amd.REL, 0.0, 1e-3), (np.float32, sz_amd.ABS_OR_REL, 1e-5, 1e-4)):
            d = s_field(48, 40, 64, dt)
            n1 = szt(0); p1 = L.sz_slab_compress(0 if dt == np.float32 else 1, d.ctypes.data, ctypes.byref(n1), mode, absb, rel, 0.0, *d.shape, 1)
            assert p1
            want = ctypes.string_at(p1, n1.value); libc.free(p1)
            info = Info(); n2 = szt(0)
            p2 = L.sz_slab_compress_multi(0 if dt == np.float32 else 1, d.ctypes.data, ctypes.byref(n2), mode, absb, rel, 0.0, *d.shape, 1, None, ctypes.byref(info))
            assert p2, "sz_slab_compress_multi failed"
            got = ctypes.string_at(p2, n2.value); libc.free(p2)
            assert got == want
            assert info.devices == 1
            assert info.used_rccl == 1, "librccl.so was not found or its communicator could not be created"
            assert info.gathered_bytes == len(want) - 40 - 24       # the one sub-stream, on the device, compared with the host copy inside the call
            # the sub-stream is the reference's stream of the slab
            eb = absb if mode == sz_amd.ABS else (rel * float(d.max() - d.min()) if mode == sz_amd.REL else max(np.float32(absb), np.float32(rel * float(np.float32(d.max()) - np.float32(d.min())))))
            assert got[64:] == oracle.compress(d, oracle.ABS, float(eb))[0]
    finally:
        os.environ.pop("SZ_SLAB_VERIFY_GATHER", None)
        sz_amd.SZ_Finalize()
