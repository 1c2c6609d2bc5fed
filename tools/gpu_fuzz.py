"""development: differential fuzzing of the HIP path against the oracle (streams byte for byte, decode bit for bit).
usage: python tools/gpu_fuzz.py [cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import sz_amd
from sz_amd.fields import l_field, m_field, near_zero_planes, s_field

assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sz14 = len(sys.argv) > 3 and sys.argv[3] == "sz14"      # withLinearRegression = NO: the SZ 1.4 path
oparams = O.default_params(with_regression=0) if sz14 else None
if sz14: sz_amd.conf_params().withRegression = 0
fails = 0
t_start = time.time()
for c in range(ncases):
    rng = np.random.default_rng(seed0 * 100003 + c)
    dt = np.float32 if rng.random() < 0.6 else np.float64
    shape = tuple(int(x) for x in rng.integers(2, 72, size=3))
    if rng.random() < 0.2: shape = (shape[0], shape[1], int(rng.integers(60, 200)))
    two_d = rng.random() < 0.3     # a 2-D array: generated as one plane of the 3-D field
    if two_d: shape = (1, int(rng.integers(2, 150)), int(rng.integers(2, 300)))
    one_d = rng.random() < 0.15    # a 1-D array: one row of the field
    if one_d: two_d = False; shape = (1, 1, int(rng.integers(2, 20000)))
    if shape[0] * shape[1] * shape[2] <= 20: continue
    kind = int(rng.integers(0, 8))
    nz, ny, nx = shape
    if kind == 0: d = s_field(nz, ny, nx, dt)
    elif kind == 1: d = l_field(nz, ny, nx, dt, n_for_hash=max(nx, 8))
    elif kind == 2:
        d = s_field(nz, ny, nx, dt)
        if two_d: h = ny // 2; d[:, h:] = l_field(nz, ny - h, nx, dt, n_for_hash=max(nx, 8))
        else: h = nz // 2; d[h:] = l_field(nz - h, ny, nx, dt, n_for_hash=max(nx, 8))
    elif kind == 3: d = rng.random(shape).astype(dt)
    elif kind == 4:
        d = s_field(nz, ny, nx, dt) * dt(0.02); h = max(1, ny // 2 - 1); d[:, :h, :] = near_zero_planes(nz, h, nx, dt, seed=c)
    elif kind == 5:
        d = s_field(nz, ny, nx, dt); d[np.abs(d) < 0.6] = 0
    elif kind == 6:
        d = (s_field(nz, ny, nx, dt) + (rng.random(shape) - 0.5).astype(dt) * dt(10.0 ** rng.integers(-5, 0)))
    else:
        d = near_zero_planes(nz, ny, nx, dt, seed=c) * dt(10.0 ** rng.integers(-1, 3))
        if rng.random() < 0.5: d[rng.integers(0, nz), rng.integers(0, ny), rng.integers(0, nx)] = 1e5
    d = np.ascontiguousarray(d)
    if two_d: d = d.reshape(shape[1], shape[2])
    if one_d: d = d.reshape(shape[2])
    mode = int(rng.choice([0, 0, 0, 1, 2, 3]))
    rng_v = float(d.max()) - float(d.min())
    abs_b = float(10.0 ** rng.uniform(-5, -1)) * max(rng_v, 1e-6)
    rel_b = float(10.0 ** rng.uniform(-5, -2))
    if os.environ.get("FUZZ_VERBOSE"): print(f"case={c} dtype={np.dtype(dt).name} shape={d.shape} kind={kind} mode={mode} abs={abs_b:.3e} rel={rel_b:.3e}", flush=True)
    try:
        ref, _ = O.compress(d, mode, abs_b, rel_b, params=oparams)
        got = sz_amd.SZ_compress_args(d, mode, abs_b, rel_b)
        ok_stream = got == ref
        back = sz_amd.SZ_decompress(ref, d.shape, d.dtype)
        dec = O.decompress(ref, d.shape, d.dtype)
        iv = np.uint32 if dt == np.float32 else np.uint64
        ok_dec = np.array_equal(back.view(iv), dec.view(iv))
    except Exception as e:  # noqa
        ok_stream = ok_dec = False
        print("case", c, "EXCEPTION", repr(e))
    if not (ok_stream and ok_dec):
        fails += 1
        st = sz_amd.SZ_hip_last_stats()
        print(f"FAIL case={c} seed0={seed0} dtype={np.dtype(dt).name} shape={shape} kind={kind} mode={mode} abs={abs_b:.3e} rel={rel_b:.3e} "
              f"stream_ok={ok_stream} dec_ok={ok_dec} len ref/gpu {len(ref)}/{len(got) if 'got' in dir() else -1} intervals={st.intervals} reg={st.n_reg_blocks} unpred={st.n_unpred}")
print(f"fuzz: {ncases} cases, {fails} failures, {time.time() - t_start:.0f} s")
