"""The HDF5 filter (id 32017, sz_amd/h5z/h5z_sz.c -- the reference's hdf5-filter/H5Z-SZ) driven through the HDF5 library itself by a small C
program (tests/h5/h5z_check.c; this image has libhdf5 1.10.6 under /opt/conda but no h5py): chunked float and double datasets written
and read back through the dynamically loaded plugin, and the raw chunks compared with the streams the oracle makes for the same chunks.
GPU: the plugin over libszhip.so.  Without a GPU: the same plugin source over the product code compiled against the CPU shim."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = os.path.join(ROOT, "tests", "sim")
HAVE_HDF5 = os.path.exists("/opt/conda/include/hdf5.h")
pytestmark = pytest.mark.skipif(not HAVE_HDF5, reason="no HDF5 C library in this environment")


def _unzstd(blob, cap):
    z = ctypes.CDLL("libzstd.so.1")
    z.ZSTD_decompress.restype = ctypes.c_size_t
    z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    out = ctypes.create_string_buffer(cap)
    n = z.ZSTD_decompress(out, cap, blob, len(blob))
    assert n <= cap
    return out.raw[:n]


def _run(plugin_dir, tmp_path, mode):
    d = str(tmp_path)
    if mode == "cfg":
        shutil.copy(os.path.join(ROOT, "tests", "golden", "sz_speed.config"), os.path.join(d, "sz.config"))
    env = dict(os.environ, HDF5_PLUGIN_PATH=plugin_dir)
    r = subprocess.run([os.path.join(SIM, "h5z_check"), d, mode], cwd=d, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    a = np.fromfile(os.path.join(d, "f32_in.bin"), np.float32).reshape(24, 40, 56)
    a2 = np.fromfile(os.path.join(d, "f32_out.bin"), np.float32).reshape(24, 40, 56)
    b = np.fromfile(os.path.join(d, "f64_in.bin"), np.float64).reshape(60, 72)
    b2 = np.fromfile(os.path.join(d, "f64_out.bin"), np.float64).reshape(60, 72)
    chunks = [open(os.path.join(d, n), "rb").read() for n in ("f32_chunk0.sz", "f32_chunk1.sz", "f64_chunk0.sz")]
    assert os.path.getsize(os.path.join(d, "t.h5")) < (a.nbytes + b.nbytes) / 2
    return a, a2, b, b2, chunks


def _check(plugin_dir, tmp_path, oracle):
    # bounds in the cd_values (SZ_errConfigToCdArray): everything else is SZ_Init(NULL)'s default, i.e. szMode = SZ_BEST_COMPRESSION with
    # zstd (conf.c:99-141) -- the chunk is the SZ stream wrapped by zstd
    a, a2, b, b2, chunks = _run(plugin_dir, tmp_path / "cd", "cd")
    assert float(np.abs(a2.astype(np.float64) - a).max()) <= 1e-3
    assert float(np.abs(b2 - b).max()) <= 1e-3 * float(b.max() - b.min())
    p = oracle.default_params(sz_mode=1, conf_rel_bound_ratio=1e-4)
    for k, raw in enumerate(chunks[:2]):
        want, _ = oracle.compress(a[12 * k:12 * k + 12], oracle.ABS, 1e-3, params=p)
        assert _unzstd(raw, a.nbytes) == want
        assert np.array_equal(a2[12 * k:12 * k + 12].view(np.uint32), oracle.decompress(want, (12, 40, 56), np.float32).view(np.uint32))
    want, _ = oracle.compress(b, oracle.REL, 0.0, 1e-3, params=p)
    assert _unzstd(chunks[2], b.nbytes) == want
    # no cd_values: the plugin reads ./sz.config (here tests/golden/sz_speed.config: ABS 1e-4, SZ_BEST_SPEED) -- plain SZ streams
    a, a2, b, b2, chunks = _run(plugin_dir, tmp_path / "cfg", "cfg")
    assert float(np.abs(a2.astype(np.float64) - a).max()) <= 1e-4 and float(np.abs(b2 - b).max()) <= 1e-4
    for k, raw in enumerate(chunks[:2]):
        want, _ = oracle.compress(a[12 * k:12 * k + 12], oracle.ABS, 1e-4)
        assert raw == want
    want, _ = oracle.compress(b, oracle.ABS, 1e-4)
    assert chunks[2] == want


@pytest.mark.slow
def test_hdf5_filter_over_the_cpu_shim(built, oracle, tmp_path):
    os.makedirs(tmp_path / "cd"); os.makedirs(tmp_path / "cfg")
    _check(os.path.join(SIM, "h5plugin"), tmp_path, oracle)


@pytest.mark.gpu
def test_hdf5_filter_on_the_gpu(built, oracle, tmp_path):
    os.makedirs(tmp_path / "cd"); os.makedirs(tmp_path / "cfg")
    _check(os.path.join(ROOT, "sz_amd", "h5z"), tmp_path, oracle)
