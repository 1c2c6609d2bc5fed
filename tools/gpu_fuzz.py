"""development: differential fuzzing of the HIP path against the oracle (streams byte for byte, decode bit for bit).
usage: python tools/gpu_fuzz.py [cases] [seed] [sz14 | pwr | msst]
  msst: point-wise relative bounds in the reference's default table-driven form: streams byte for byte, decode bit for bit (both dtypes:
       no transcendental on the device); the bound itself is the reference's business there -- the largest excess is reported
  pwr: point-wise relative bounds (log-domain form): positive / sign-changing data with zeros, streams byte for byte (both sides code the
       sign bytes with the system zstd), float decode bit for bit, double decode to 1 ulp (exp2 of the libm vs the GPU's)"""
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
msst = len(sys.argv) > 3 and sys.argv[3] == "msst"
pwr = len(sys.argv) > 3 and sys.argv[3] in ("pwr", "msst")
worst_excess = 0.0
oparams = O.default_params(with_regression=0) if sz14 else None
if sz14: sz_amd.conf_params().withRegression = 0
if pwr: sz_amd.conf_params().accelerate_pw_rel_compression = 1 if msst else 0
fails = 0
t_start = time.time()
for c in range(ncases):
    rng = np.random.default_rng(seed0 * 100003 + c)
    dt = np.float32 if rng.random() < 0.6 else np.float64
    shape = tuple(int(x) for x in rng.integers(2, 72, size=3))
    if rng.random() < 0.2: shape = (shape[0], shape[1], int(rng.integers(60, 200)))
    big = int(os.environ.get("FUZZ_MAXDIM", "0"))      # larger 3-D arrays (many tiles of the sweep, many segments of the packing passes): every case 3-D, rows a multiple of 4 values two times in three
    if big:
        shape = tuple(int(x) for x in rng.integers(20, big, size=3))
        if rng.random() < 0.67: shape = (shape[0], shape[1], (shape[2] + 3) // 4 * 4)
    two_d = rng.random() < 0.3 and not big     # a 2-D array: generated as one plane of the 3-D field
    if two_d: shape = (1, int(rng.integers(2, 150)), int(rng.integers(2, 300)))
    one_d = rng.random() < 0.15 and not big    # a 1-D array: one row of the field
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
    if pwr:
        sgn = rng.random()
        mag = np.exp(rng.uniform(0.5, 4.0) * d.astype(np.float64) / max(float(np.abs(d).max()), 1e-30) + 0.05 * rng.standard_normal(d.shape))
        if sgn < 0.4: dd = mag
        elif sgn < 0.8: dd = mag * np.sign(d.astype(np.float64) + 1e-300)
        else: dd = -mag
        if rng.random() < 0.5: dd[rng.random(d.shape) < 0.03] = 0.0
        d = np.ascontiguousarray(dd.astype(dt))
        ratio = float(10.0 ** rng.uniform(-4, -1))
        if msst and rng.random() < 0.15: d.reshape(-1)[0] = 0          # nearZero = 0: the zeros stay (computeRangeSize_float_MSST19 starts from element 0)
        po = O.default_params(); po.pw_rel_bound_ratio = ratio; po.segment_size = 0; po.accelerate_pw_rel = 1 if msst else 0
        try:
            ref, _ = O.compress(d, O.PW_REL, 0.0, 0.0, params=po)
            got = sz_amd.SZ_compress_args(d, sz_amd.PW_REL, 0.0, 0.0, ratio)
            back = sz_amd.SZ_decompress(ref, d.shape, d.dtype)
            dec = O.decompress(ref, d.shape, d.dtype)
            if msst:
                assert ref[3] & 0x08 or ref[3] & 0x10
                okd = np.array_equal(back.view(np.uint8), dec.view(np.uint8))
            elif dt == np.float32: okd = np.array_equal(back.view(np.uint32), dec.view(np.uint32))
            else: okd = bool(np.all((back >= np.nextafter(dec, -np.inf)) & (back <= np.nextafter(dec, np.inf))) and np.array_equal(np.signbit(back), np.signbit(dec)))
            x = d.astype(np.float64); nzm = x != 0
            okb = (not nzm.any()) or float((np.abs(back.astype(np.float64)[nzm] - x[nzm]) / np.abs(x[nzm])).max()) <= ratio
            if not msst and not okb and got == ref and okd:
                # the reference's own stream, decoded to the reference's own values: what is left is the float rounding of ITS exp2 / narrowing
                # (seen: 1.00002 x the ratio) -- not this library's to fix; anything larger still fails
                okb = float((np.abs(back.astype(np.float64)[nzm] - x[nzm]) / np.abs(x[nzm])).max()) <= ratio * (1 + 1e-4)
                globals()["ref_round"] = globals().get("ref_round", 0) + 1
            if msst:
                if nzm.any() and np.isfinite(back).all(): worst_excess = max(worst_excess, float((np.abs(back.astype(np.float64)[nzm] - x[nzm]) / np.abs(x[nzm])).max()) / ratio)
                okb = True
            if dt == np.float64 and got != ref and len(got) == len(ref) and okd and okb:
                soft = globals().get("soft", 0) + 1; globals()["soft"] = soft      # log2 of a double: the GPU's and glibc's last bit differ on a few values
                continue
            if not (got == ref and okd and okb):
                fails += 1
                print(f"FAIL pwr case={c} seed0={seed0} dtype={np.dtype(dt).name} shape={d.shape} ratio={ratio:.3e} stream_ok={got == ref} dec_ok={okd} bound_ok={okb} len {len(ref)}/{len(got)}")
        except Exception as e:  # noqa
            fails += 1
            print("case", c, "EXCEPTION", repr(e))
        continue
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
print(f"fuzz: {ncases} cases, {fails} failures, {time.time() - t_start:.0f} s" + (f" (largest point-wise error / ratio, the reference's own: {worst_excess:.4f})" if msst else f" ({globals().get('soft', 0)} float64 streams of equal length differ in log2's last bit; {globals().get('ref_round', 0)} cases where the reference's own decoded values pass the ratio by < 1e-4 of it)" if pwr else ""))
