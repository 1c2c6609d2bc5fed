"""Synthetic test fields of SURVEY.md section 8(d) (no RNG; pure functions of the index).

S-field: smooth sinusoid (reference picks Lorenzo for every block)
L-field: slow ramp + hash noise (reference picks regression for every block)
M-field: S for z < N/2, L for z >= N/2 (exactly 50 % regression blocks at 256^3)
Arrays are indexed [z, y, x] with x fastest, i.e. SZ dims r1 = nx, r2 = ny, r3 = nz.
"""
import numpy as np


def _grid(nz, ny, nx, z0=0):
    z = np.arange(z0, z0 + nz, dtype=np.float64)[:, None, None]
    y = np.arange(ny, dtype=np.float64)[None, :, None]
    x = np.arange(nx, dtype=np.float64)[None, None, :]
    return z, y, x


def s_field(nz, ny, nx, dtype=np.float32, z0=0):
    z, y, x = _grid(nz, ny, nx, z0)
    tp = 2.0 * np.pi
    f = np.sin(tp * x / 64.0) * np.cos(tp * y / 96.0) * np.sin(tp * z / 128.0)
    f = f + 0.5 * np.sin(tp * (x + 2.0 * y + 3.0 * z) / 256.0)
    return np.ascontiguousarray(f.astype(dtype))


def l_field(nz, ny, nx, dtype=np.float32, z0=0, n_for_hash=None):
    """`idx = (z*N + y)*N + x` with N = n_for_hash (defaults to nx, the cube edge)."""
    N = np.uint64(n_for_hash if n_for_hash is not None else nx)
    z, y, x = _grid(nz, ny, nx, z0)
    tp = 2.0 * np.pi
    f = np.sin(tp * x / 2048.0) + np.cos(tp * y / 2048.0) + np.sin(tp * z / 2048.0)
    zi = np.arange(z0, z0 + nz, dtype=np.uint64)[:, None, None]
    yi = np.arange(ny, dtype=np.uint64)[None, :, None]
    xi = np.arange(nx, dtype=np.uint64)[None, None, :]
    idx = (zi * N + yi) * N + xi
    m32 = np.uint64(0xFFFFFFFF)
    h = (idx * np.uint64(2654435761)) & m32
    h = ((h ^ (h >> np.uint64(15))) * np.uint64(2246822519)) & m32
    h = h ^ (h >> np.uint64(13))
    u = (h & np.uint64(0xFFFFFF)).astype(np.float64) / float(1 << 24)
    f = f + (2.0 * u - 1.0) * 5e-4
    return np.ascontiguousarray(f.astype(dtype))


def m_field(n, dtype=np.float32):
    out = np.empty((n, n, n), dtype=dtype)
    h = n // 2
    out[:h] = s_field(h, n, n, dtype)
    out[h:] = l_field(n - h, n, n, dtype, z0=h, n_for_hash=n)
    return out


def near_zero_planes(nz, ny, nx, dtype=np.float32, seed=5):
    """Noisy planes close to zero: the regression predictor wins, and its plane evaluated OUTSIDE the array (k < 0) is a small,
    quantisable value.  Next to Lorenzo blocks this checks that such values never leak into the zero halo of the array faces."""
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    rng = np.random.default_rng(seed)
    return np.ascontiguousarray(((0.3 * z + 0.5 * y + 0.2 * x) * 2e-5 + (rng.random((nz, ny, nx)) - 0.5) * 1.2e-4).astype(dtype))


def reg_beside_lorenzo(nz, ny, nx, dtype=np.float32):
    """near_zero_planes for j < ny/2 - 1 (regression blocks), a scaled S-field beyond (Lorenzo blocks that read them as neighbours)."""
    d = s_field(nz, ny, nx, dtype) * np.dtype(dtype).type(0.02)
    h = ny // 2 - 1
    d[:, :h, :] = near_zero_planes(nz, h, nx, dtype)
    return np.ascontiguousarray(d)


def plane_field(ny, nx, dtype=np.float32):
    """2-D: slow ramp + hash noise for x < nx/2 (the 2-D selection picks regression there), smooth sinusoid beyond (Lorenzo)."""
    h = nx // 2
    f = np.empty((ny, nx), dtype=dtype)
    f[:, :h] = l_field(1, ny, h, dtype, n_for_hash=max(nx, 8))[0]
    f[:, h:] = s_field(1, ny, nx - h, dtype)[0] * dtype(0.5)
    return np.ascontiguousarray(f)
