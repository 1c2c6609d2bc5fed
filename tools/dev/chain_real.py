import ctypes, sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sz_amd, oracle_lib
from sz_amd.fields import m_field
class C(ctypes.Structure):
    _fields_ = [("reg_count", ctypes.c_size_t), ("codes", ctypes.POINTER(ctypes.c_int) * 4), ("unpred", ctypes.c_void_p * 4), ("unpred_count", ctypes.c_size_t * 4), ("prec", ctypes.c_double * 4)]
L = sz_amd.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
d = m_field(n)
ref, st = oracle_lib.compress(d, oracle_lib.ABS, 1e-4, want_stages=True)
nb = st["num_blocks"]; ind = np.ascontiguousarray(st["indicator"].astype(np.uint8))
co = np.ascontiguousarray(st["reg_params"].reshape(4, nb).astype(np.float32))
print("blocks", nb, "regression", int((ind == 0).sum()))
for name, fn in (("fast", L.szhost_coeff_chain_one_p), ("ref", L.szhost_coeff_chain_one_ref), ("fast", L.szhost_coeff_chain_one_p)):
    c = co.copy(); s = C()
    L.szhost_coeff_chain_begin(0, ind.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nb), ctypes.c_double(1e-4), 6, 6, 6, 4, ctypes.byref(s))
    t0 = time.perf_counter(); per = []
    for e in range(4):
        ta = time.perf_counter()
        fn(0, c.ctypes.data_as(ctypes.c_void_p), ind.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nb), 0, e, ctypes.byref(s), None)
        per.append(round((time.perf_counter() - ta) * 1e3, 3))
    t1 = time.perf_counter()
    codes3 = np.ctypeslib.as_array(s.codes[3], shape=(s.reg_count,)).astype(np.int64) - 32768
    print("   per coefficient ms", per, "| d codes: mean |q| %.0f, max %d" % (np.abs(codes3[codes3 > -32768]).mean(), np.abs(codes3[codes3 > -32768]).max()))
    print(name, "%.3f ms for 4 chains" % ((t1 - t0) * 1e3), "raw", list(s.unpred_count), "ns/step %.2f" % ((t1 - t0) * 1e9 / (4 * s.reg_count)))
    L.szhost_coeffs_free(ctypes.byref(s))
try:
    print("steps in the general form (all runs):", ctypes.c_ulong.in_dll(L, "g_chain_generic_steps").value)
except ValueError:
    pass
