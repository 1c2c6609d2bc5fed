"""world_size-2 `gloo` test of the slab path's exchange steps (sz_amd/slab.py): the {min,max} all-reduce for the
range-based bound modes and the single all-gather of variable-length sub-streams.  No GPU; the per-slab streams are
produced by the oracle (the checker), which is exactly what each rank's GPU would produce."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    from sz_amd import slab
    from sz_amd.fields import s_field
    dims = (40, 24, 32)
    bounds = slab.slab_bounds(dims[0], world)
    z0, z1 = bounds[rank]
    mine = s_field(z1 - z0, dims[1], dims[2], np.float64, z0=z0)
    lo, hi = slab.global_minmax(float(mine.min()), float(mine.max()))
    eb = 1e-3 * (hi - lo)  # REL 1e-3 with the GLOBAL range, passed to every slab as an absolute bound
    stream, _ = O.compress(mine, O.ABS, eb)
    t = torch.frombuffer(bytearray(stream), dtype=torch.uint8)
    parts, sizes = slab.allgather_streams(t)
    # the overlapped form used by bench.py for N > 1: two gathers in flight, completed in order, same result
    sg = slab.StreamGather()
    t2 = torch.frombuffer(bytearray(stream[::-1]), dtype=torch.uint8)
    h1 = sg.begin(t, len(stream)); h2 = sg.begin(t2, len(stream))
    p1, s1 = sg.end(h1); p2, s2 = sg.end(h2)
    assert s1 == sizes and s2 == sizes and all(bytes(a.numpy().tobytes()) == bytes(b.numpy().tobytes()) for a, b in zip(p1, parts))
    assert bytes(p2[rank].numpy().tobytes()) == stream[::-1]
    # the benchmark's pipeline: one gather in flight while the next step runs, buffers reused every second step
    # (a result is a view of its buffers: it is checked when the gather is completed, before those buffers are used again)
    def payload_of(step, r):
        return bytes([(step * 7 + r) & 0xff]) * (1000 + 100 * step + 10 * r)

    def check(step, res):
        parts_k, sizes_k = res
        for r in range(world):
            assert sizes_k[r] == len(payload_of(step, r)) and bytes(parts_k[r].numpy().tobytes()) == payload_of(step, r)
    pending = []
    for step in range(5):
        mine_k = payload_of(step, rank)
        pending.append((step, sg.begin(torch.frombuffer(bytearray(mine_k), dtype=torch.uint8), len(mine_k))))
        if len(pending) > 1:
            k, h = pending.pop(0); check(k, sg.end(h))
    while pending:
        k, h = pending.pop(0); check(k, sg.end(h))
    blob = slab.pack_container(np.float64, dims, bounds, [bytes(p.numpy().tobytes()) for p in parts])
    if rank == 0:
        ret["blob"] = blob; ret["range"] = (lo, hi); ret["sizes"] = sizes
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
def test_slab_exchange_world2(oracle):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29517, ret), nprocs=world, join=True)
    from sz_amd import slab
    from sz_amd.fields import s_field
    whole = s_field(40, 24, 32, np.float64)
    lo, hi = ret["range"]
    assert lo == float(whole.min()) and hi == float(whole.max())
    dt, dims, bounds, streams = slab.unpack_container(ret["blob"])
    assert dims == (40, 24, 32) and [len(s) for s in streams] == list(ret["sizes"])
    eb = 1e-3 * (hi - lo)
    rec = np.concatenate([oracle.decompress(bytes(s), (z1 - z0, 24, 32), np.float64) for (z0, z1), s in zip(bounds, streams)])
    assert float(np.abs(rec - whole).max()) <= eb
    # each sub-stream is exactly what the reference produces for that slab on its own
    for (z0, z1), s in zip(bounds, streams):
        ref, _ = oracle.compress(whole[z0:z1], oracle.ABS, eb)
        assert bytes(s) == ref


def _worker_c4(rank, world, port, ret):
    """BASELINE configs[3] in miniature on the product's own code path (the HIP layer compiled against the CPU shim, tests/sim):
    global range over the process group -> eb -> this rank's slab through SZ_compress_args -> overlapped all-gather -> own
    sub-stream back through SZ_decompress."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import sim_lib
    os.environ["SZ_AMD_LIB"] = sim_lib.shim_path()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sz_amd
    from sz_amd import slab
    from sz_amd.fields import s_field
    dims = (36, 20, 40)
    bounds = slab.slab_bounds(dims[0], world)
    z0, z1 = bounds[rank]
    assert z0 % 6 == 0                                            # cuts on block multiples: per-slab block grids are the array's
    mine = s_field(z1 - z0, dims[1], dims[2], np.float64, z0=z0)
    lo, hi = slab.global_minmax(float(mine.min()), float(mine.max()))
    eb = 1e-3 * (hi - lo)
    assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
    sg = slab.StreamGather()
    got = None
    for step in range(3):                                         # first gather blocking (sizes unknown), the others overlapped
        stream = sz_amd.SZ_compress_args(mine, sz_amd.ABS, eb)
        h = sg.begin(torch.frombuffer(bytearray(stream), dtype=torch.uint8), len(stream))
        got = sg.end(h)
    parts, sizes = got
    back = sz_amd.SZ_decompress(bytes(parts[rank].numpy().tobytes()), mine.shape, mine.dtype)
    sz_amd.SZ_Finalize()
    assert float(np.abs(back - mine).max()) <= eb
    ret[rank] = (bytes(parts[rank].numpy().tobytes()), eb, (z0, z1))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
def test_config4_flow_world2_on_the_product_code_path(oracle):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_c4, args=(world, 29531, ret), nprocs=world, join=True)
    from sz_amd.fields import s_field
    whole = s_field(36, 20, 40, np.float64)
    for r in range(world):
        stream, eb, (z0, z1) = ret[r]
        assert abs(eb - 1e-3 * (float(whole.max()) - float(whole.min()))) < 1e-15
        ref, _ = oracle.compress(whole[z0:z1], oracle.ABS, eb)      # each sub-stream is the reference's stream for that slab
        assert stream == ref


def test_c_slab_container_equals_the_python_one_and_round_trips(oracle):
    """include/sz_slab.h on the product's code (HIP layer on the CPU shim): the container a C caller gets from sz_slab_compress is, byte
    for byte, what sz_amd/slab.py packs from the per-slab SZ_compress_args streams; sz_slab_decompress reads it back within the bound."""
    import ctypes
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sim_lib
    import sz_amd
    from sz_amd import api, slab
    from sz_amd.fields import s_field
    saved = api._lib
    try:
        L = ctypes.CDLL(sim_lib.shim_path())
        api._lib = api._bind(L)
        assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
        szt = ctypes.c_size_t
        L.sz_slab_compress.restype = ctypes.c_void_p
        L.sz_slab_compress.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(szt), ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, szt, szt, szt, ctypes.c_int]
        L.sz_slab_decompress.restype = ctypes.c_void_p
        L.sz_slab_decompress.argtypes = [ctypes.c_void_p, szt, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(szt * 3)]
        L.sz_slab_bounds.argtypes = [szt, ctypes.c_int, ctypes.c_int, ctypes.POINTER(szt)]
        libc = ctypes.CDLL(None); libc.free.argtypes = [ctypes.c_void_p]
        for (n0, world, dt, mode, absb, rel) in ((38, 3, np.float32, sz_amd.ABS, 1e-3, 0.0), (25, 2, np.float64, sz_amd.REL, 0.0, 1e-3), (7, 4, np.float32, sz_amd.ABS, 1e-2, 0.0),
                                                 (24, 2, np.float32, sz_amd.REL, 0.0, 1e-3)):
            whole = s_field(n0, 21, 40, dt)
            bb = (szt * (2 * world))()
            L.sz_slab_bounds(n0, world, 6, bb)
            bounds = [(int(bb[2 * r]), int(bb[2 * r + 1])) for r in range(world)]
            assert bounds == slab.slab_bounds(n0, world)
            eb = absb if mode == sz_amd.ABS else rel * float(whole.max() - whole.min())
            streams = [sz_amd.SZ_compress_args(np.ascontiguousarray(whole[z0:z1]), sz_amd.ABS, eb) if z1 > z0 else b"" for z0, z1 in bounds]
            for (z0, z1), st in zip(bounds, streams):        # every sub-stream is the reference's stream of that slab
                if z1 > z0:
                    assert st == oracle.compress(np.ascontiguousarray(whole[z0:z1]), oracle.ABS, eb)[0]
            want = slab.pack_container(dt, whole.shape, bounds, streams)
            n = szt(0)
            p = L.sz_slab_compress(0 if dt == np.float32 else 1, whole.ctypes.data, ctypes.byref(n), mode, absb, rel, 0.0, *whole.shape, world)
            assert p
            got = ctypes.string_at(p, n.value)
            libc.free(p)
            assert got == want
            dtc = ctypes.c_int(-1); dims = (szt * 3)()
            q = L.sz_slab_decompress(got, len(got), ctypes.byref(dtc), ctypes.byref(dims))
            assert q and tuple(dims) == whole.shape and dtc.value == (0 if dt == np.float32 else 1)
            back = np.ctypeslib.as_array(ctypes.cast(q, ctypes.POINTER(ctypes.c_float if dt == np.float32 else ctypes.c_double)), shape=(whole.size,)).copy().reshape(whole.shape)
            libc.free(q)
            assert float(np.abs(back.astype(np.float64) - whole).max()) <= eb
            assert L.sz_slab_decompress(got[:50], 50, ctypes.byref(dtc), ctypes.byref(dims)) is None     # a truncated container is refused
            # damaged tables (untrusted input): dimensions whose product wraps, a gap between two slabs, a slab past the end
            bad = bytearray(got); bad[16:24] = (1 << 60).to_bytes(8, "little")
            assert L.sz_slab_decompress(bytes(bad), len(bad), ctypes.byref(dtc), ctypes.byref(dims)) is None
            if world > 1 and bounds[1][1] > bounds[1][0]:
                bad = bytearray(got); bad[40 + 24:40 + 32] = (bounds[1][0] + 1).to_bytes(8, "little")      # slab 1 now begins a plane late
                assert L.sz_slab_decompress(bytes(bad), len(bad), ctypes.byref(dtc), ctypes.byref(dims)) is None
            bad = bytearray(got); bad[40 + 24 * (world - 1) + 8:40 + 24 * (world - 1) + 16] = (n0 + 5).to_bytes(8, "little")
            assert L.sz_slab_decompress(bytes(bad), len(bad), ctypes.byref(dtc), ctypes.byref(dims)) is None
            # the multi-device driver (sz_slab_compress_multi: a host thread + context per "device"; the shim has one device, named
            # `world` times, so the ranges and the sub-streams travel through host memory): the same bytes
            class Info(ctypes.Structure):
                _fields_ = [("devices", ctypes.c_int), ("used_rccl", ctypes.c_int), ("gathered_bytes", szt), ("seconds_total", ctypes.c_double), ("seconds_slowest_slab", ctypes.c_double)]
            L.sz_slab_compress_multi.restype = ctypes.c_void_p
            L.sz_slab_compress_multi.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(szt), ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, szt, szt, szt,
                                                 ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(Info)]
            info = Info(); n2 = szt(0); devs = (ctypes.c_int * world)(*([0] * world))
            p2 = L.sz_slab_compress_multi(0 if dt == np.float32 else 1, whole.ctypes.data, ctypes.byref(n2), mode, absb, rel, 0.0, *whole.shape, world, devs, ctypes.byref(info))
            assert p2 and info.devices == world and info.used_rccl == 0
            got2 = ctypes.string_at(p2, n2.value)
            libc.free(p2)
            assert got2 == want
            # round 5: the call keeps its per-device contexts (and, with RCCL, its communicator) for the next call with the same device list; the second
            # call runs on what the first one left, a call with the cache switched off builds everything anew: the same bytes all three times
            for cache in ("1", "0"):
                os.environ["SZ_SLAB_MULTI_CACHE"] = cache
                try:
                    n3 = szt(0)
                    p3 = L.sz_slab_compress_multi(0 if dt == np.float32 else 1, whole.ctypes.data, ctypes.byref(n3), mode, absb, rel, 0.0, *whole.shape, world, devs, ctypes.byref(info))
                    assert p3 and ctypes.string_at(p3, n3.value) == want, cache
                    libc.free(p3)
                finally:
                    os.environ.pop("SZ_SLAB_MULTI_CACHE", None)
        sz_amd.SZ_Finalize()
    finally:
        api._lib = saved


def test_c_slab_mode_comes_from_the_argument(oracle):
    """ADVICE (round 4, high): sz_slab_compress took the error-bound mode from the stored configuration as well as from its argument, so on a
    default-initialised library (SZ_Init(NULL): errorBoundMode = PSNR) an explicit ABS call was compressed with the PSNR-derived bound -- and a second,
    identical call (the first had rewritten the configuration) gave other bytes.  The mode is the argument's alone now, as in SZ_compress_args
    (sz.c:294-391).  Also the PSNR and NORM slab paths, which no test covered: bounds derived from the WHOLE array, honoured by every slab."""
    import ctypes
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sim_lib
    import sz_amd
    from sz_amd import api
    from sz_amd.fields import s_field
    saved = api._lib
    try:
        L = ctypes.CDLL(sim_lib.shim_path())
        api._lib = api._bind(L)
        szt = ctypes.c_size_t
        L.SZ_Init.argtypes = [ctypes.c_char_p]
        L.sz_slab_compress.restype = ctypes.c_void_p
        L.sz_slab_compress.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(szt), ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, szt, szt, szt, ctypes.c_int]
        L.sz_slab_decompress.restype = ctypes.c_void_p
        L.sz_slab_decompress.argtypes = [ctypes.c_void_p, szt, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(szt * 3)]
        libc = ctypes.CDLL(None); libc.free.argtypes = [ctypes.c_void_p]
        whole = s_field(26, 21, 40)

        def run(mode, absb, rel):
            n = szt(0)
            p = L.sz_slab_compress(0, whole.ctypes.data, ctypes.byref(n), mode, absb, rel, 0.0, *whole.shape, 2)
            assert p
            got = ctypes.string_at(p, n.value); libc.free(p)
            dtc = ctypes.c_int(-1); dims = (szt * 3)()
            q = L.sz_slab_decompress(got, len(got), ctypes.byref(dtc), ctypes.byref(dims))
            assert q
            back = np.ctypeslib.as_array(ctypes.cast(q, ctypes.POINTER(ctypes.c_float)), shape=(whole.size,)).copy().reshape(whole.shape)
            libc.free(q)
            return got, float(np.abs(back.astype(np.float64) - whole).max())

        assert L.SZ_Init(None) == 0                                  # the defaults of conf.c:99-141: errorBoundMode = PSNR, psnr = 90
        a, err_a = run(sz_amd.ABS, 1e-4, 0.0)
        b, err_b = run(sz_amd.ABS, 1e-4, 0.0)
        assert err_a <= 1e-4 and err_b <= 1e-4 and a == b           # the explicit bound holds on the first call, and both calls give the same bytes
        rng = float(whole.max() - whole.min())
        r1, err_r = run(sz_amd.REL, 0.0, 1e-3)
        assert err_r <= 1e-3 * rng * (1 + 1e-6) and r1 != a
        cp = api.conf_params()
        psnr_eb = rng * 10 ** ((float(cp.psnr) + 10 * np.log10(1 - 2.0 / 3.0 * float(cp.predThreshold))) / -20)      # conf.c:54-60 on the whole array's range
        p1, err_p = run(sz_amd.PSNR, 0.0, 0.0)
        assert err_p <= psnr_eb * (1 + 1e-6)
        cp.normErr = 0.05                                            # (callers of the reference poke the struct; 0 after SZ_Init(NULL))
        norm_eb = float(np.sqrt(3.0 / whole.size) * 0.05)                                                         # conf.c:62-65 on the whole array's element count
        n1, err_n = run(sz_amd.NORM, 0.0, 0.0)
        assert err_n <= norm_eb * (1 + 1e-6)
        a2, err_a2 = run(sz_amd.ABS, 1e-4, 0.0)                      # ... and an ABS call after them is the first call's stream again
        assert a2 == a
        sz_amd.SZ_Finalize()
    finally:
        api._lib = saved


@pytest.mark.slow
def test_bench_entry_starts_its_own_ranks_for_gpus_gt_1():
    """`python bench.py --gpus 2 ...` -- the shape of the driver's N = 1 command -- must start the two ranks itself and print ONE JSON
    line with n_gpus = 2.  Rehearsed on the CPU: gloo + the product's code on the HIP-on-CPU shim (--dry-run), tiny arrays."""
    import json
    import subprocess
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sim_lib
    env = dict(os.environ, SZ_AMD_LIB=sim_lib.shim_path())
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run", "--edge", "24"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["out_bytes"] > 0
