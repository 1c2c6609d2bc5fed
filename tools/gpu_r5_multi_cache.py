"""round 5: what sz_slab_compress_multi keeps from call to call (communicator, context, exchange buffers): seconds of five calls on one device, cache on / off"""
import ctypes, os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sz_amd
from sz_amd.fields import s_field
class Info(ctypes.Structure):
    _fields_ = [("devices", ctypes.c_int), ("used_rccl", ctypes.c_int), ("gathered_bytes", ctypes.c_size_t), ("seconds_total", ctypes.c_double), ("seconds_slowest_slab", ctypes.c_double)]
L = sz_amd.lib(); szt = ctypes.c_size_t
L.sz_slab_compress_multi.restype = ctypes.c_void_p
L.sz_slab_compress_multi.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(szt), ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, szt, szt, szt, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(Info)]
libc = ctypes.CDLL(None); libc.free.argtypes = [ctypes.c_void_p]
assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
d = s_field(256, 256, 256, np.float32)
import time
for cache in ("1", "0"):
    os.environ["SZ_SLAB_MULTI_CACHE"] = cache
    out = []
    for it in range(5):
        info = Info(); n = szt(0); t0 = time.perf_counter()
        p = L.sz_slab_compress_multi(0, d.ctypes.data, ctypes.byref(n), sz_amd.REL, 0.0, 1e-3, 0.0, *d.shape, 1, None, ctypes.byref(info))
        t1 = time.perf_counter()
        assert p; libc.free(p)
        out.append(round((t1 - t0) * 1e3, 1))
    print(json.dumps({"cache": cache, "used_rccl": info.used_rccl, "call_ms": out, "array": "256^3 float32 from host memory, REL 1e-3, one device"}), flush=True)
sz_amd.SZ_Finalize()
