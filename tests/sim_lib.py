"""ctypes binding of the TEST-ONLY simulators in tests/sim (never part of the product):
libszh_sim.so   -- 64-lane simulator instantiating the kernel bodies of sz_amd/csrc/szh_pencil.h / szh_core.h
libszhip_sim.so -- the product's szhip.hip + host C compiled against a HIP-on-CPU shim (one workgroup at a time)"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(ROOT, "tests", "sim")
_lanes = None


def lanes():
    global _lanes
    if _lanes is None:
        subprocess.check_call(["make", "-s", "-C", _DIR])
        _lanes = ctypes.CDLL(os.path.join(_DIR, "libszh_sim.so"))
    return _lanes


def shim_path():
    subprocess.check_call(["make", "-s", "-C", _DIR])
    return os.path.join(_DIR, "libszhip_sim.so")


def expand_coef(st):
    nb = st["num_blocks"]
    co = np.zeros((4, nb), dtype=st["coeff_dec"].dtype)
    co[:, np.where(st["indicator"] == 0)[0]] = st["coeff_dec"]
    return np.ascontiguousarray(co)


def quantize(d, st):
    S = lanes()
    suf = "f32" if d.dtype == np.float32 else "f64"
    f = getattr(S, "szh_sim_quantize_" + suf)
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                  ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    co, bl = expand_coef(st), np.ascontiguousarray(st["indicator"])
    codes = np.zeros(d.size, dtype=np.uint16)
    err = f(d.ctypes.data, *d.shape, st["eb"], st["intervals"], st["use_mean"], st["mean"], bl.ctypes.data, co.ctypes.data, codes.ctypes.data)
    return err, codes


def reconstruct(codes_nat, d, st):
    S = lanes()
    suf = "f32" if d.dtype == np.float32 else "f64"
    g = getattr(S, "szh_sim_reconstruct_" + suf)
    g.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                  ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    co, bl = expand_coef(st), np.ascontiguousarray(st["indicator"])
    out = np.zeros(d.shape, dtype=d.dtype)
    out.ravel()[codes_nat == 0] = d.ravel()[codes_nat == 0]  # the pre-scatter of unpredictable values
    err = g(out.ctypes.data, *d.shape, st["eb"], st["intervals"], st["use_mean"], st["mean"], bl.ctypes.data, co.ctypes.data, codes_nat.ctypes.data)
    return err, out


def nat_to_blk(codes_nat, shape):
    S = lanes()
    S.szh_sim_nat_to_blk_u16.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    blk = np.zeros(codes_nat.size, dtype=np.int32)
    S.szh_sim_nat_to_blk_u16(codes_nat.ctypes.data, *shape, blk.ctypes.data)
    return blk


def fit_select(d, eb, use_mean, mean):
    S = lanes()
    suf = "f32" if d.dtype == np.float32 else "f64"
    f = getattr(S, "szh_sim_fit_select_" + suf)
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
    nb = 1
    for n in d.shape:
        nb *= 1 if n <= 6 else n // 6
    coef = np.zeros((4, nb), dtype=d.dtype)
    bl = np.zeros(nb, dtype=np.uint8)
    f(d.ctypes.data, *d.shape, eb, use_mean, mean, coef.ctypes.data, bl.ctypes.data)
    return coef, bl


def sample(d, eb, sd=100, max_radius=32768):
    S = lanes()
    suf = "f32" if d.dtype == np.float32 else "f64"
    f = getattr(S, "szh_sim_sample_" + suf)
    f.restype = ctypes.c_double
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_uint,
                  ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    rh = np.zeros(max_radius, dtype=np.uint32)
    fh = np.zeros(8192, dtype=np.uint32)
    w, c = ctypes.c_uint64(0), ctypes.c_uint64(0)
    mean = f(d.ctypes.data, *d.shape, eb, sd, max_radius, rh.ctypes.data, fh.ctypes.data, ctypes.byref(w), ctypes.byref(c))
    return mean, rh, fh, w.value, c.value
