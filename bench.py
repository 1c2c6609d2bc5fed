#!/usr/bin/env python3
"""bench.py -- benchmarks of the MI355X SZ 2.1 hot path.

    python bench.py --gpus N --steps K --warmup W [--config headline|c4]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

--config headline (default; BASELINE.json configs[1]/[2], the configuration the metric is quoted on).  A "step" is one pass of
the hot path as SZ_compress_args runs it -- value-range scan, regression fit + predictor selection, interval optimiser,
predict/quantise, Huffman encode into the reference's SZ 2.1 stream -- over one 512x512x512 float32 array (smooth sinusoid
"S-field", ABS 1e-4) that is already resident in HBM; the stream is left in HBM.  With N ranks every rank owns one such slab of
an (N*512)x512x512 array (weak scaling, no data-path collective) and the step ends with one all-gather of the variable-length
sub-streams (RCCL), launched asynchronously so that it overlaps the next step's compression; every gather is completed inside
the timed region.  Since round 5 the timed region is ONE BLOCKING CALL AFTER THE OTHER (`--inflight 1`: what a caller of SZ_compress_args sees;
`value` = `value_single_call`; `step_ms` lists every step); `concurrent` gives 1 / 2 / 4 arrays in flight.  Rank 0 prints ONE JSON line.  Extra objects on that line:
  roofline     -- the predict+quantise wavefront kernel: algorithmic bytes (N*4, the array read once) / its average duration
                  measured with HIP events on the library's stream, against the 8 TB/s HBM3E peak.
  m_field      -- BASELINE configs[2]: the same step on the 512^3 "M-field" (half of the blocks choose the regression predictor: k_reg_points, the
                  beam sweep fed while the host's coefficient chains run), its decompression, and the sweep's own roofline from an unfed call.
  e2e          -- SZ_compress_args / SZ_decompress from and to HOST memory (pageable), PCIe included: never the `value`.
  e2e_default  -- the same two calls under the reference's SHIPPED mode (szMode = SZ_BEST_COMPRESSION: the zstd stage), beside the unmodified reference doing the same.
  cpu_baseline -- the unmodified reference (oracle/_ref/libSZ.so; kind "reference") pinned to one core, median of 3, with the oracle (a C restatement,
                  oracle/) beside it as `port`; cpu_baseline_mt: the reference's own OpenMP variant on 64 threads.

--config c4 (BASELINE.json configs[3]): 1024^3 float64 S-field, REL 1e-3, slab-sharded: rank r owns planes of the outer
dimension (at N = 8: 128x1024x1024 = 1 GiB; at N = 1 the bench runs ONE such slab, it does not pretend to hold the 8 GiB
array), the value range is all-reduced, eb = 1e-3 * range, every rank compresses its slab, the sub-streams are all-gathered,
every rank decompresses its own sub-stream and checks max|x - x'| <= eb.
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EDGE = 512
EB = 1e-4
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def cpu_info():
    model = platform.processor() or ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count()


def _mt_worker(args):
    idx, planes, n = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from sz_amd.fields import s_field
    d = s_field(planes, n, n, np.float32, z0=idx * planes)
    t = time.perf_counter()
    s, _ = O.compress(d, O.ABS, EB)
    return time.perf_counter() - t, len(s)


def _time_reference(sample, cfg_path):
    """oracle/_ref/libSZ.so (the unmodified reference, oracle/build_ref.sh) through its public C API: SZ_Init(config) once, then
    SZ_compress_args on the sample, szMode = SZ_BEST_SPEED (the protocol of example/sz.c:355-375 without the file I/O)."""
    import ctypes
    so = os.path.join(ROOT, "oracle", "_ref", "libSZ.so")
    if not os.path.exists(so):
        return None
    L = ctypes.CDLL(so)
    szt = ctypes.c_size_t
    L.SZ_Init.argtypes = [ctypes.c_char_p]
    L.SZ_compress_args.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(szt), ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double] + [szt] * 5
    L.SZ_compress_args.restype = ctypes.c_void_p
    libc = ctypes.CDLL(None); libc.free.argtypes = [ctypes.c_void_p]
    if L.SZ_Init(cfg_path.encode()) != 0:
        return None
    times, size = [], 0
    for _ in range(3):
        n = szt(0)
        t1 = time.perf_counter()
        p = L.SZ_compress_args(0, sample.ctypes.data, ctypes.byref(n), 0, EB, 0.0, 0.0, 0, 0, sample.shape[0], sample.shape[1], sample.shape[2])
        times.append(time.perf_counter() - t1)
        size = n.value
        libc.free(p)
    L.SZ_Finalize()
    return float(np.median(times)), size


def _omp_ref_child(n, threads):
    """child process (OMP_NUM_THREADS is set by the parent before the OpenMP runtime loads): the reference's OpenMP variant on the S-field"""
    import ctypes
    from sz_amd.fields import s_field
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libSZ_omp.so"))
    szt = ctypes.c_size_t
    L.SZ_Init.argtypes = [ctypes.c_char_p]
    assert L.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config").encode()) == 0
    d = s_field(n, n, n, np.float32)
    fn = L.SZ_compress_float_3D_MDQ_openmp
    fn.restype = ctypes.c_void_p
    fn.argtypes = [ctypes.c_void_p, szt, szt, szt, ctypes.c_float, ctypes.POINTER(szt)]
    libc = ctypes.CDLL(None); libc.free.argtypes = [ctypes.c_void_p]
    times, size = [], 0
    for _ in range(3):
        m = szt(0)
        t1 = time.perf_counter()
        p = fn(d.ctypes.data, n, n, n, EB, ctypes.byref(m))
        times.append(time.perf_counter() - t1)
        size = m.value
        libc.free(p)
    print("OMPREF " + json.dumps({"t": float(np.median(times)), "size": int(size), "times": times}))


def _time_reference_omp(n, ncpu):
    """the reference's OWN multi-thread variant (sz/src/sz_omp.c, oracle/_ref/libSZ_omp.so = the unmodified sources with
    -fopenmp) on the same array, thread count = the largest power of two <= the cores of this box (at most 64)."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libSZ_omp.so")) or n % 32:
        return None
    P = 1
    while P * 2 <= min(ncpu or 1, 64):
        P *= 2
    env = dict(os.environ, OMP_NUM_THREADS=str(P), OMP_DYNAMIC="false", OMP_PROC_BIND="false")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--omp-ref-child", str(n), str(P)], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("OMPREF ")]
    if not line:
        return None
    o = json.loads(line[0][7:])
    return {"value": round(n * n * n * 4 / o["t"] / 1e9, 4), "unit": "GB/s", "cores": P, "kind": "reference",
            "sample": f"median of 3 SZ_compress_float_3D_MDQ_openmp passes over the full {n}^3 float32 S-field by oracle/_ref/libSZ_omp.so (the unmodified "
                      f"reference built -fopenmp), OMP_NUM_THREADS = {P} = boxes ({o['t']:.2f} s per pass); stream {o['size']} B -- the OpenMP container, "
                      "the format `omp_container` above writes on the GPU",
            "stream_bytes": o["size"]}


def cpu_baselines(host, n, gpu_size):
    """The CPU beside the GPU number, on the host cores of this box, pinned to one core, median of 3 passes over the whole array:
    the unmodified reference library (oracle/_ref, kind "reference") when it is there, and the oracle (its restatement, kind "port");
    then P independent slabs in P processes (what the slab decomposition gives a multi-core CPU)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O  # the checker, timed here as a CPU baseline ("port" of the reference loops)
    model, ncpu = cpu_info()
    sample = np.ascontiguousarray(host if n <= 512 else host[:512])
    old = None
    try:
        old = os.sched_getaffinity(0)
        os.sched_setaffinity(0, {sorted(old)[len(old) // 2]})
    except (AttributeError, OSError):
        pass
    times, ref = [], None
    for _ in range(3):
        t1 = time.perf_counter()
        ref, _ = O.compress(sample, O.ABS, EB)
        times.append(time.perf_counter() - t1)
    refrun = None
    try:
        refrun = _time_reference(sample, os.path.join(ROOT, "tests", "golden", "sz_speed.config"))
    except Exception as e:  # noqa: BLE001 -- a baseline, not the product
        refrun = None
    if old is not None:
        os.sched_setaffinity(0, old)
    tc = float(np.median(times))
    port = {"value": round(sample.nbytes / tc / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"median of 3 compress passes over the full {sample.shape[0]}x{n}x{n} float32 S-field by oracle/liboracle.so, pinned to one "
                      f"core ({tc:.2f} s per pass)", "stream_bytes": len(ref), "gpu_stream_identical": bool(len(ref) == gpu_size)}
    if refrun is not None:
        tr, rsize = refrun
        one = {"value": round(sample.nbytes / tr / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "reference",
               "sample": f"median of 3 SZ_compress_args passes over the full {sample.shape[0]}x{n}x{n} float32 S-field by oracle/_ref/libSZ.so (the "
                         f"unmodified reference, gcc -O3, szMode = SZ_BEST_SPEED), pinned to one core ({tr:.2f} s per pass)",
               "cpu_model": model, "nproc": ncpu, "stream_bytes": int(rsize), "gpu_stream_identical": bool(rsize == gpu_size), "port": port}
    else:
        one = dict(port, cpu_model=model, nproc=ncpu,
                   note="oracle/_ref/libSZ.so is not in the tree (oracle/build_ref.sh builds it where /root/reference exists): the port is timed")
    mt = None
    try:
        mt = _time_reference_omp(n, ncpu)
    except Exception as e:  # noqa: BLE001 -- a baseline, not the product
        mt = None
    if mt is not None:
        return one, mt
    try:
        import multiprocessing as mp
        P = 1
        while P * 2 <= min(ncpu or 1, 8):
            P *= 2
        if P > 1:
            planes = sample.shape[0] // P
            with mp.get_context("fork").Pool(P) as pool:
                t1 = time.perf_counter()
                res = pool.map(_mt_worker, [(i, planes, n) for i in range(P)])
                wall = time.perf_counter() - t1
            mt = {"value": round(P * planes * n * n * 4 / wall / 1e9, 4), "unit": "GB/s", "cores": P, "kind": "port",
                  "sample": f"{P} slabs of {planes}x{n}x{n} compressed at the same time by {P} processes ({wall:.2f} s wall, data generation "
                            "included in no process's timed part but in the wall time)",
                  "slowest_slab_s": round(max(r[0] for r in res), 3)}
    except Exception as e:  # noqa: BLE001 -- a baseline, not the product
        mt = {"error": repr(e)}
    return one, mt


def _omp_traffic(n):
    """HBM bytes per launch of the OpenMP container's sweep from the PMC passes committed under profiles/ (only for the workload measured)"""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_traffic_omp_col.json")))
        return pm["traffic_bytes_per_launch"] if n == EDGE else None
    except (OSError, KeyError, ValueError):
        return None


def _sync(torch):
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def run_headline(args, torch, dist, world, rank, local_rank, dev):
    import ctypes
    import sz_amd
    from sz_amd import slab
    from sz_amd.fields import m_field, s_field

    n = args.edge
    # this rank's slab of the (world*n) x n x n S-field (z offset = rank*n); host generation is exact numpy float64 math
    host = s_field(n, n, n, np.float32, z0=rank * n)
    x = torch.from_numpy(host).to(dev)
    nbytes_in = host.nbytes
    ctx = sz_amd.HipContext(0 if getattr(args, "dry_run", False) else local_rank)     # (the CPU shim of --dry-run has one "device")
    out_cap = nbytes_in // 2 + (1 << 20)
    out_bufs = [torch.empty(out_cap, dtype=torch.uint8, device=dev) for _ in range(2)]   # alternate: a gather may still read the other
    gather = slab.StreamGather() if world > 1 else None
    pending = []
    step_no = [0]
    gather_host_s = [0.0]                          # host time inside gather.begin / gather.end (the exchange's visible cost on this rank)

    def one_step(src=x):
        # the whole hot path of SZ_compress_args for this call, into the stream.  The value range (computeRangeSize_float: what the
        # reference does first; the header records it) is reduced inside the library's fit pass (SZHIP_RANGE_FROM_DATA), exactly what
        # SZ_compress_args of this build does for an absolute bound -- one read of the input less than a separate range scan.
        meta = sz_amd.make_meta(np.float32, err_mode=sz_amd.ABS, abs_bound=EB, vmin=0.0, vmax=0.0)
        ob = out_bufs[step_no[0] & 1]
        step_no[0] += 1
        out = ctypes.c_void_p(ob.data_ptr())
        nn = ctypes.c_size_t(out_cap)
        st = sz_amd.szhip_stats()
        p = sz_amd.szhip_params(100, 0.99, 65536, 0, 1)
        rc = sz_amd.lib().szhip_compress(ctx._h, 0, src.data_ptr(), 1, n, n, n, EB, ctypes.byref(p), meta, len(meta), 2,
                                         ctypes.byref(out), ctypes.byref(nn), ctypes.byref(st))
        if rc:
            raise RuntimeError(f"szhip_compress failed: {sz_amd.lib().szhip_last_error(ctx._h)}")
        if world > 1:
            if os.environ.get("SZ_BENCH_SYNC_GATHER"):          # fallback: the plain, non-overlapped all-gather
                slab.allgather_streams(ob, nn.value)
            else:
                pending.append(gather.begin(ob, nn.value))
                if len(pending) > 1:
                    gather.end(pending.pop(0))
        return nn.value, st, ob

    def drain():
        while pending:
            gather.end(pending.pop(0))

    def sync_all():
        _sync(torch)
        if world > 1:
            dist.barrier()
        _sync(torch)

    # ---- the timed region: EXACTLY args.steps full compressions of the array, `inflight` of them in flight on this GPU at a time
    # (szhip_pool: one context + one host thread per lane; the passes around one array's sweep run beside the sweep of the next).
    # Every call is a complete szhip_compress of the same input into its own output buffer; the streams are byte-identical.
    inflight = max(1, args.inflight)
    pool_bufs = [torch.empty(out_cap, dtype=torch.uint8, device=dev) for _ in range(inflight + 1)]
    pools = {}
    # every lane compresses its OWN copy of the field (round 4): with all lanes reading one buffer a fraction of a millisecond apart, the
    # 256 MiB Infinity Cache could serve one lane the lines another had just fetched -- real callers compress different arrays
    lane_inputs = {id(x): [x]}

    def inputs_for(src, k):
        copies = lane_inputs.setdefault(id(src), [src])
        while len(copies) < k:
            copies.append(src.clone())
        return copies[:k]

    step_done_s = []          # when each call of the last run_steps came back (seconds after its start): a stall inside a timed region shows here

    def run_steps(k_lanes, nsteps, gather_too, src=None):
        """nsteps compressions with k_lanes in flight; returns (elapsed, per-call stats, (size, buffer) of the last call)."""
        meta = sz_amd.make_meta(np.float32, err_mode=sz_amd.ABS, abs_bound=EB, vmin=0.0, vmax=0.0)
        prm = sz_amd.szhip_params(100, 0.99, 65536, 0, 1)
        if k_lanes not in pools:
            # set-up, not a step: every lane's context allocates its workspaces (and loads its kernels) on its first calls, and a new
            # pool meets ONE stall of ~7 ms somewhere in its first ~10 calls (round 4, tools/gpu_r4_bimodal.sh: the lane's host thread
            # sits inside the HIP runtime between recording the call's first event and enqueuing its first kernel; once per pool and
            # process).  With two priming rounds that stall fell into the timed region in 5 of 10 runs on one box (247 instead of 341 GB/s
            # at 10 steps); with twelve, in 0 of 10.
            pools[k_lanes] = sz_amd.HipPool(0 if getattr(args, "dry_run", False) else local_rank, k_lanes)
            for _ in range(2 if getattr(args, "dry_run", False) else int(os.environ.get("SZ_BENCH_PRIME_ROUNDS", "12"))):
                tks = [pools[k_lanes].submit(x.data_ptr(), True, (n, n, n), np.float32, EB, meta, prm, pool_bufs[q % len(pool_bufs)].data_ptr(), out_cap) for q in range(k_lanes)]
                for tk in tks: pools[k_lanes].wait(tk)
        pool = pools[k_lanes]
        srcs = inputs_for(x if src is None else src, k_lanes)
        sync_all()
        t_begin = time.perf_counter()
        live, stats_all, last = [], [], None
        del step_done_s[:]
        for i in range(nsteps + k_lanes):
            if i >= k_lanes or i >= nsteps:                 # collect the oldest call (also drains the tail)
                if live:
                    tk, ob_ = live.pop(0)
                    sz_, st_ = pool.wait(tk)
                    stats_all.append(st_); last = (sz_, ob_)
                    step_done_s.append(time.perf_counter() - t_begin)
                    if gather_too and world > 1:
                        tg = time.perf_counter()
                        pending.append(gather.begin(ob_, sz_))
                        if len(pending) > 1:
                            gather.end(pending.pop(0))
                        gather_host_s[0] += time.perf_counter() - tg
            if i < nsteps:
                ob_ = pool_bufs[i % (k_lanes + 1)] if k_lanes + 1 <= len(pool_bufs) else pool_bufs[i % len(pool_bufs)]
                live.append((pool.submit(srcs[i % k_lanes].data_ptr(), True, (n, n, n), np.float32, EB, meta, prm, ob_.data_ptr(), out_cap), ob_))
        drain()
        sync_all()
        return time.perf_counter() - t_begin, stats_all, last

    # set-up, not steps: the HIP runtime stalls ONCE for ~7 ms somewhere in a process's first few hundred milliseconds of pooled submissions (all
    # lanes at once; profiles/r04_bench_priming_runs.txt) -- priming rounds alone left it inside the timed region in 2 of 14 four-lane runs.  The pool
    # is therefore exercised, untimed, for SZ_BENCH_REHEARSE_MS (600 ms) before the W warm-up steps; the `concurrent` object, measured seconds
    # later in the same process, never showed the stall.
    if not getattr(args, "dry_run", False):
        t_reh = time.perf_counter()
        while (time.perf_counter() - t_reh) * 1e3 < float(os.environ.get("SZ_BENCH_REHEARSE_MS", "600")):
            run_steps(inflight, 40, False)
    run_steps(inflight, max(args.warmup, inflight), True)
    gather_host_s[0] = 0.0
    elapsed, stats_all, (size, ob) = run_steps(inflight, args.steps, True)
    step_ms = [round((b - a) * 1e3, 3) for a, b in zip([0.0] + step_done_s[:-1], step_done_s)]
    gather_ms_per_step = gather_host_s[0] / args.steps * 1e3
    quant_ms = [st.ms_quant for st in stats_all]
    stats = stats_all[-1]
    per_rank = None
    if world > 1:
        # every rank's own clock and its host time in the gather, so that a flat or bent scaling curve can be read from the line
        mine = torch.tensor([elapsed / args.steps * 1e3, gather_ms_per_step, float(np.mean(quant_ms))], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"ms_per_step": [round(float(t[0]), 4) for t in allr], "gather_host_ms_per_step": [round(float(t[1]), 4) for t in allr],
                    "sweep_kernel_ms": [round(float(t[2]), 4) for t in allr]}
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * nbytes_in / (elapsed / args.steps) / 1e9
    # ---- the same steps as ONE blocking call after the other (what SZ_compress_args is): reported beside `value` in the same line
    if inflight == 1:
        single_el, single_stats = elapsed, stats_all
    else:
        run_steps(1, 2, True)
        single_el, single_stats, _ = run_steps(1, args.steps, True)
        if world > 1:
            t1 = torch.tensor([single_el], dtype=torch.float64, device=dev)
            dist.all_reduce(t1, op=dist.ReduceOp.MAX)
            single_el = float(t1.item())
    single_call = {"GB/s": round(world * nbytes_in / (single_el / args.steps) / 1e9, 3), "ms": round(single_el / args.steps * 1e3, 4),
                   "quant_ms": round(float(np.mean([t.ms_quant for t in single_stats])), 4)}

    # ---- quality of the result (outside the timed region): decompress on the GPU, compare with the input
    dec = torch.empty_like(x)
    dst = ctx.decompress(ob.data_ptr(), True, size, 4 + 28 + 8, (n, n, n), np.float32, dec.data_ptr(), True)
    err = (dec - x).abs()
    max_abs_err = float(err.max().item())
    mse = float((err * err).double().sum().item()) / x.numel()
    psnr = 20 * np.log10(float((x.max() - x.min()).item())) - 10 * np.log10(mse)
    del err
    _sync(torch); td = time.perf_counter()
    for _ in range(3):
        ctx.decompress(ob.data_ptr(), True, size, 4 + 28 + 8, (n, n, n), np.float32, dec.data_ptr(), True)
    _sync(torch); td = (time.perf_counter() - td) / 3

    if rank != 0:
        if world > 1:
            dist.barrier(); dist.destroy_process_group()
        return

    quant_avg_ms = float(np.mean(quant_ms))
    achieved = nbytes_in / (quant_avg_ms * 1e-3) / 1e9 if quant_avg_ms > 0 else 0.0     # (the CPU shim of --dry-run has no device events)
    # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside the timed process);
    # only valid for the workload it was measured on
    qk = getattr(stats, "quant_kernel", 0)
    kname = "k_beam<float,false,false,false>" if qk == 2 else "k_pencil<float,false>"
    traffic, traffic_src = None, None
    for name in (("r06_pmc_traffic_beam_sfield.json",) if qk == 2 else ("r02_pmc_traffic_pencil.json", "r01_pmc_traffic_pencil.json")):
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", name)))
            if n == EDGE:
                traffic = pm["traffic_bytes_per_launch"]
                traffic_src = f"profiles/{name} (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes)"
                break
        except (OSError, KeyError, ValueError):
            pass
    roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": nbytes_in, "avg_kernel_ms": round(quant_avg_ms, 4),
                # `frac` is per LAUNCH inside the timed region, where `launches_in_flight` sweeps share the GPU (each on a share of the CUs); the same
                # kernel with the GPU to itself, from the one-call-after-the-other run of this line:
                "launches_in_flight": inflight,
                "alone": {"avg_kernel_ms": single_call["quant_ms"], "frac": round(nbytes_in / (single_call["quant_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if single_call["quant_ms"] > 0 else None}}

    # ---- arrays in flight: the same args.steps compressions with 1 / 2 / 4 lanes (outside the timed region), with the sweep kernel's own
    # time per K so that interference is visible; and every stream of the last run against the one a single blocking call gives
    concurrent = None
    if world == 1 and not getattr(args, "dry_run", False) and not args.timed_only:
        ref_size, _, ref_ob = one_step()
        _sync(torch)
        ref_bytes = ref_ob[:ref_size].clone()
        concurrent = {"what": "szhip_pool: K contexts + host threads on this GPU, K full szhip_compress calls of the same array in flight; "
                              "GB/s over args.steps calls; quant_ms = average duration of the predict+quantise kernel inside those calls", "K": {}}
        for K in (1, 2, 4):
            if K + 1 > len(pool_bufs):
                pool_bufs.extend(torch.empty(out_cap, dtype=torch.uint8, device=dev) for _ in range(K + 1 - len(pool_bufs)))
            run_steps(K, max(2, K), False)
            el, sts, _ = run_steps(K, args.steps, False)
            same = all(bool(torch.equal(pool_bufs[i][:ref_size], ref_bytes)) for i in range(min(K + 1, args.steps)))
            concurrent["K"][str(K)] = {"GB/s": round(nbytes_in / (el / args.steps) / 1e9, 2), "ms_per_array": round(el / args.steps * 1e3, 4),
                                       "quant_ms": round(float(np.mean([t.ms_quant for t in sts])), 4),
                                       "prequant_ms": round(float(np.mean([t.ms_prequant for t in sts])), 4),
                                       "entropy_ms": round(float(np.mean([t.ms_entropy for t in sts])), 4), "streams_identical": same}
        concurrent["timed_region_lanes"] = inflight
        concurrent["timed_region_call_ms"] = [round(float(t.ms_total), 3) for t in stats_all]     # every call of the timed region, host clock inside the library
        if os.environ.get("SZ_BENCH_CALL_PHASES"):
            concurrent["timed_region_call_phases"] = [[round(float(v), 2) for v in (t.ms_prequant, t.ms_quant, t.ms_entropy, t.ms_host)] for t in stats_all]
        del ref_bytes

    # ---- BASELINE configs[2]: the adaptive case proper -- 512^3 M-field, half of the blocks regression (single GPU, outside the timed region)
    mfield = None
    if world == 1 and n == EDGE and not args.no_m_field:
        xm = torch.from_numpy(m_field(n)).to(dev)
        for _ in range(5):                                        # (the first calls create the chain threads and size the workspaces of the regression path)
            one_step(xm)
        msteps = max(3, min(args.steps, 5))
        tms = []
        for _ in range(msteps):                                   # each step timed on its own (7 ms: the sync is negligible); median reported
            _sync(torch); t1 = time.perf_counter()
            msize, mst, mob = one_step(xm)
            _sync(torch); tms.append(time.perf_counter() - t1)
        tm = float(np.median(tms))
        mdec = torch.empty_like(xm)
        ctx.decompress(mob.data_ptr(), True, msize, 4 + 28 + 8, (n, n, n), np.float32, mdec.data_ptr(), True)
        mob_keep = mob[:msize].clone()                            # (the context's output buffers alternate: keep the stream that is decoded)
        tdm = []
        for _ in range(3):
            _sync(torch); t1 = time.perf_counter()
            ctx.decompress(mob_keep.data_ptr(), True, msize, 4 + 28 + 8, (n, n, n), np.float32, mdec.data_ptr(), True)
            _sync(torch); tdm.append(time.perf_counter() - t1)
        mfield = {"GB/s": round(nbytes_in / tm / 1e9, 2), "ms": round(tm * 1e3, 3), "decompress_GBps": round(nbytes_in / float(np.median(tdm)) / 1e9, 2), "decompress_ms": round(float(np.median(tdm)) * 1e3, 3), "ms_samples": [round(t * 1e3, 3) for t in tms], "reg_blocks": int(mst.n_reg_blocks), "blocks": int(mst.n_blocks),
                  "out_bytes": int(msize), "ratio": round(nbytes_in / msize, 4), "max_abs_err": float((mdec - xm).abs().max().item()),
                  "sweep_fed_while_chains_run": bool(int(mst.chain_overlapped) == 2),
                  "phase_ms": {"prequant": round(mst.ms_prequant, 3), "quant_incl_waits_for_the_host_chains": round(mst.ms_quant, 3),
                               "entropy": round(mst.ms_entropy, 3), "host_glue": round(mst.ms_host, 3)}}
        if not getattr(args, "dry_run", False):
            # the sweep of this array on its own: the same call in the unfed order (chains first, then k_reg_points and the sweep back to back), whose
            # `quant` phase is the two kernels and nothing else; PMC traffic of that order from profiles/r06_pmc_traffic_beam_mfield.json
            os.environ["SZ_HIP_BEAM_FEED"] = "0"
            try:
                one_step(xm)
                _, ust, _ = one_step(xm)
            finally:
                os.environ.pop("SZ_HIP_BEAM_FEED", None)
            try:
                btr = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_traffic_beam_mfield.json")))["traffic_bytes_per_launch"]
            except (OSError, KeyError, ValueError):
                btr = None
            mfield["roofline"] = {"bound": "hbm", "kernel": "k_reg_points<float,0> + k_beam<float,false,false,true>", "unfed_call_quant_ms": round(ust.ms_quant, 4),
                                  "achieved": round(nbytes_in / (ust.ms_quant * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(nbytes_in / (ust.ms_quant * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": btr,
                                  "traffic_source": "profiles/r06_pmc_traffic_beam_mfield.json (k_beam alone: FETCH_SIZE x2 + WRITE_SIZE; the x2 of the guide overstates the 4- and 8-byte-per-lane reads)",
                                  "unfed_call_ms": round(ust.ms_total, 3)}
            # two M-field arrays in flight (szhip_pool): one array's host coefficient chain beside the other's kernels
            ref_m = mob[:msize].clone()
            for km in (2,):           # (three lanes: six streams share the process's hardware queues and the coefficient DMA of one lane can queue behind another lane's waiting kernel)
                if km + 1 > len(pool_bufs):
                    pool_bufs.extend(torch.empty(out_cap, dtype=torch.uint8, device=dev) for _ in range(km + 1 - len(pool_bufs)))
                run_steps(km, km, False, xm)
                elm, _, (lsz, lob) = run_steps(km, 3 * km, False, xm)
                mfield["two_in_flight" if km == 2 else "three_in_flight"] = {
                    "GB/s": round(nbytes_in / (elm / (3 * km)) / 1e9, 2), "ms_per_array": round(elm / (3 * km) * 1e3, 3),
                    "stream_identical_to_single_call": bool(lsz == msize and torch.equal(lob[:lsz], ref_m))}
        del xm, mdec

    # ---- the other paths of the same library, one line each (optional; outside the timed region; single GPU only): the SZ 1.4 container
    #      (withLinearRegression = NO) on the same array, a 2-D array through the SZ 2.1 path, a 1-D series
    other = None
    if not args.no_other_paths and not args.timed_only and world == 1 and n == EDGE:
        def timed(fn, reps=3):
            fn(); _sync(torch); t = time.perf_counter()
            for _ in range(reps):
                r = fn()
            _sync(torch)
            return (time.perf_counter() - t) / reps, r

        def run(fn_name, ptr, dims, extra, mbytes, eb=EB):
            out = ctypes.c_void_p(out_bufs[0].data_ptr()); nn = ctypes.c_size_t(out_cap); st = sz_amd.szhip_stats()
            p = sz_amd.szhip_params(100, 0.99, 65536, 0, 0)
            rc = getattr(sz_amd.lib(), fn_name)(ctx._h, 0, ptr, 1, *dims, eb, *extra, ctypes.byref(p), mbytes, len(mbytes), 2,
                                                ctypes.byref(out), ctypes.byref(nn), ctypes.byref(st))
            if rc:
                raise RuntimeError(f"{fn_name} failed: {sz_amd.lib().szhip_last_error(ctx._h)}")
            return nn.value
        vmin, vmax = ctx.minmax(x.data_ptr(), True, x.numel(), np.float32)
        rng = vmax - vmin
        med = float(np.float32(np.float32(vmin) + np.float32(rng) / np.float32(2)))
        meta0 = sz_amd.make_meta(np.float32, err_mode=sz_amd.ABS, abs_bound=EB, vmin=vmin, vmax=vmax)
        meta14 = bytes([meta0[0], meta0[1], meta0[2], 0x40]) + bytes(meta0[4:])
        t14, size14 = timed(lambda: run("szhip_compress_sz14", x.data_ptr(), (n, n, n), (float(np.float32(rng)), med), meta14))
        from sz_amd.fields import plane_field
        p2 = torch.from_numpy(plane_field(4096, 4096)).to(dev)
        meta2 = sz_amd.make_meta(np.float32, err_mode=sz_amd.ABS, abs_bound=EB, vmin=float(p2.min().item()), vmax=float(p2.max().item()))
        t2, size2 = timed(lambda: run("szhip_compress", p2.data_ptr(), (0, 4096, 4096), (), meta2))
        g = torch.Generator(device="cpu"); g.manual_seed(1)
        s1 = (torch.cumsum(torch.randn(1 << 24, generator=g, dtype=torch.float64), 0) * 0.01
              + torch.sin(torch.arange(1 << 24, dtype=torch.float64) * 0.003)).to(torch.float32).to(dev)
        lo1, hi1 = float(s1.min().item()), float(s1.max().item())
        rng1 = float(np.float32(np.float32(hi1) - np.float32(lo1)))
        med1 = float(np.float32(np.float32(lo1) + np.float32(rng1) / np.float32(2)))
        m1 = sz_amd.make_meta(np.float32, err_mode=sz_amd.ABS, abs_bound=1e-3, vmin=lo1, vmax=hi1)
        m1 = bytes([m1[0], m1[1], m1[2], 0x40]) + bytes(m1[4:])
        t1d, size1d = timed(lambda: run("szhip_compress_sz14", s1.data_ptr(), (0, 0, 1 << 24), (rng1, med1), m1, eb=1e-3))
        other = {"sz14_3d_512_f32": {"GB/s": round(nbytes_in / t14 / 1e9, 2), "ms": round(t14 * 1e3, 3), "out_bytes": size14},
                 "sz21_2d_4096x4096_f32": {"GB/s": round(p2.numel() * 4 / t2 / 1e9, 2), "ms": round(t2 * 1e3, 3), "out_bytes": size2},
                 "sz14_1d_16Mi_f32_abs1e-3": {"GB/s": round(s1.numel() * 4 / t1d / 1e9, 2), "ms": round(t1d * 1e3, 3), "out_bytes": size1d}}
        del p2, s1

    # ---- the reference's OpenMP container (szh_omp.h / szh_ompcol.h) on the same array, its own object in the default line since round 4
    #      (--omp-boxes N picks the box count, --no-omp leaves it out)
    omp = None
    omp_boxes = args.omp_boxes
    if omp_boxes < 0:                                              # default: boxes of 32^3 (4096 at 512^3) when the edge allows
        nb1 = n // 32
        omp_boxes = nb1 ** 3 if (n % 32 == 0 and nb1 >= 2 and nb1 & (nb1 - 1) == 0 and not args.no_omp and not getattr(args, "dry_run", False)) else 0
    if omp_boxes and world == 1:
        meta_o = sz_amd.make_meta(np.float32, err_mode=sz_amd.ABS, abs_bound=EB, vmin=0.0, vmax=0.0)
        meta_o = bytes([meta_o[0], meta_o[1], meta_o[2], 0xC0]) + bytes(meta_o[4:])
        y_o = torch.empty_like(x)
        for _ in range(2):
            optr, osize, ost = ctx.compress_omp(x.data_ptr(), True, (n, n, n), np.float32, EB, omp_boxes, meta_o, out_on_device=True)
        _sync(torch); t0 = time.perf_counter()
        oq, oe = [], []
        for _ in range(args.steps):
            optr, osize, ost = ctx.compress_omp(x.data_ptr(), True, (n, n, n), np.float32, EB, omp_boxes, meta_o, out_on_device=True)
            oq.append(ost.ms_quant); oe.append(ost.ms_entropy)
        _sync(torch); to = (time.perf_counter() - t0) / args.steps
        ctx.decompress_omp(optr, True, osize, len(meta_o), (n, n, n), np.float32, y_o.data_ptr(), True)
        _sync(torch); t0 = time.perf_counter()
        for _ in range(args.steps):
            odst = ctx.decompress_omp(optr, True, osize, len(meta_o), (n, n, n), np.float32, y_o.data_ptr(), True)
        _sync(torch); tod = (time.perf_counter() - t0) / args.steps
        oqm = max(float(np.mean(oq)), 1e-9)                        # (the CPU rehearsal has no event times)
        omp = {"container": "the reference's OpenMP container (SZ_compress_float_3D_MDQ_openmp, sz/src/sz_omp.c:63-358): independent boxes, one code book, "
                            "a payload per box; a stock OpenMP build of SZ reads it", "boxes": int(ost.n_blocks),
               "GB/s": round(nbytes_in / to / 1e9, 2), "ms": round(to * 1e3, 3), "decompress_GBps": round(nbytes_in / tod / 1e9, 2),
               "out_bytes": int(osize), "ratio": round(nbytes_in / osize, 4), "max_abs_err": float((y_o - x).abs().max().item()),
               "verbatim_values": int(ost.n_unpred), "intervals": int(ost.intervals),
               "phase_ms": {"prequant": round(ost.ms_prequant, 3), "quant": round(oqm, 3), "entropy": round(float(np.mean(oe)), 3), "host_glue": round(ost.ms_host, 3),
                            "compress_call_total": round(ost.ms_total, 3), "decompress_entropy": round(odst.ms_entropy, 3), "decompress_quant": round(odst.ms_quant, 3),
                            "decompress_total": round(odst.ms_total, 3)},
               "roofline": {"bound": "hbm", "kernel": "k_omp_col<float,32,32,false>" if n % 32 == 0 and omp_boxes == (n // 32) ** 3 else "k_omp_box<float,false,true>",
                            "achieved": round(nbytes_in / (oqm * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(nbytes_in / (oqm * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": _omp_traffic(n), "algorithmic_bytes_per_launch": nbytes_in,
                            "avg_kernel_ms": round(oqm, 4),
                            "achieved_incl_code_writes": round(6 * x.numel() / (oqm * 1e-3) / 1e9, 2),
                            "note": "algorithmic bytes = N * sizeof(T) READ (SURVEY 8d); the kernel also writes 2 N bytes of codes: `achieved_incl_code_writes`"}}
        del y_o

    # ---- host-pointer API, PCIe included (never the headline value)
    e2e = None
    if world == 1 and n == EDGE and not args.timed_only:
        assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
        L = sz_amd.lib()
        dims = (0, 0, n, n, n)
        tcs, tds, s2len, s2 = [], [], 0, None
        for _ in range(3):                                       # the C calls themselves: no Python-side copy inside the timed part
            nn2 = ctypes.c_size_t(0)
            t1 = time.perf_counter()
            p2 = L.SZ_compress_args(0, host.ctypes.data, ctypes.byref(nn2), sz_amd.ABS, EB, 0.0, 0.0, *dims)
            tcs.append(time.perf_counter() - t1)
            if not p2:
                raise RuntimeError("SZ_compress_args failed")
            s2len = nn2.value
            if s2 is None:
                s2 = ctypes.string_at(p2, s2len)
            L.free(p2)
        sbuf = ctypes.create_string_buffer(s2, len(s2))
        for _ in range(3):
            t1 = time.perf_counter()
            q2 = L.SZ_decompress(0, sbuf, len(s2), *dims)
            tds.append(time.perf_counter() - t1)
            if not q2:
                raise RuntimeError("SZ_decompress failed")
            L.free(q2)
        sz_amd.SZ_Finalize()
        e2e = {"compress_GBps": round(nbytes_in / float(np.median(tcs)) / 1e9, 2), "decompress_GBps": round(nbytes_in / float(np.median(tds)) / 1e9, 2),
               "what": "SZ_compress_args / SZ_decompress (the C entry points) on a pageable host array: 512 MiB staged to the device in 8 MiB chunks through "
                       "pinned buffers by four host threads + the stream back, and the reverse into a freshly malloc'd array (transparent huge pages requested for it); median of 3",
               "stream_identical_to_device_path": bool(s2len == size)}

    # ---- the same two calls under the mode the reference SHIPS (example/sz.config: szMode = SZ_BEST_COMPRESSION, the zstd stage behind the SZ stream, conf.c:114,
    #      utility.c:174-214), and the unmodified reference doing the same on one core: what a caller who changes nothing gets
    e2e_default = None
    if world == 1 and n == EDGE and not args.timed_only:
        import hashlib
        cfg_def = os.path.join(ROOT, "tests", "golden", "sz_default.config")
        assert sz_amd.SZ_Init(cfg_def) == 0
        L = sz_amd.lib()
        dims = (0, 0, n, n, n)
        tcs, tds, wrapped = [], [], None
        for _ in range(3):
            nn2 = ctypes.c_size_t(0)
            t1 = time.perf_counter()
            p2 = L.SZ_compress_args(0, host.ctypes.data, ctypes.byref(nn2), sz_amd.ABS, EB, 0.0, 0.0, *dims)
            tcs.append(time.perf_counter() - t1)
            if not p2:
                raise RuntimeError("SZ_compress_args (default mode) failed")
            if wrapped is None:
                wrapped = ctypes.string_at(p2, nn2.value)
            L.free(p2)
        sbuf = ctypes.create_string_buffer(wrapped, len(wrapped))
        dec_md5 = None
        L.SZ_decompress.restype = ctypes.c_void_p
        for _ in range(3):
            t1 = time.perf_counter()
            q2 = L.SZ_decompress(0, sbuf, len(wrapped), *dims)
            tds.append(time.perf_counter() - t1)
            if not q2:
                raise RuntimeError("SZ_decompress (default mode) failed")
            if dec_md5 is None:
                dec_md5 = hashlib.md5(ctypes.string_at(q2, host.nbytes)).hexdigest()
            L.free(q2)
        sz_amd.SZ_Finalize()
        e2e_default = {"compress_GBps": round(nbytes_in / float(np.median(tcs)) / 1e9, 2), "decompress_GBps": round(nbytes_in / float(np.median(tds)) / 1e9, 2),
                       "compress_ms": round(float(np.median(tcs)) * 1e3, 1), "decompress_ms": round(float(np.median(tds)) * 1e3, 1), "wrapped_bytes": len(wrapped),
                       "what": "SZ_compress_args / SZ_decompress on a pageable host array under tests/golden/sz_default.config (szMode = SZ_BEST_COMPRESSION, zstd level 3): the GPU "
                               "path + PCIe + the zstd stage on host threads (the stream in pieces, a frame each, the frames one behind the other: any ZSTD_decompress reads them as one "
                               "stream); median of 3"}
        # the unmodified reference: the same call (one pass: ~2.5 s), and ITS decoder on this library's wrapped stream
        so = os.path.join(ROOT, "oracle", "_ref", "libSZ.so")
        if os.path.exists(so) and not args.no_cpu_baseline:
            R = ctypes.CDLL(so)
            szt = ctypes.c_size_t
            R.SZ_Init.argtypes = [ctypes.c_char_p]
            R.SZ_compress_args.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(szt), ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double] + [szt] * 5
            R.SZ_compress_args.restype = ctypes.c_void_p
            R.SZ_decompress.argtypes = [ctypes.c_int, ctypes.c_void_p, szt] + [szt] * 5
            R.SZ_decompress.restype = ctypes.c_void_p
            libc = ctypes.CDLL(None); libc.free.argtypes = [ctypes.c_void_p]
            if R.SZ_Init(cfg_def.encode()) == 0:
                m = szt(0)
                t1 = time.perf_counter()
                rp = R.SZ_compress_args(0, host.ctypes.data, ctypes.byref(m), 0, EB, 0.0, 0.0, *dims)
                tr = time.perf_counter() - t1
                ref_wrapped = ctypes.string_at(rp, m.value)
                libc.free(rp)
                t1 = time.perf_counter()
                rq = R.SZ_decompress(0, sbuf, len(wrapped), *dims)          # the stock decoder reads the GPU library's wrapped stream
                trd = time.perf_counter() - t1
                ref_dec_md5 = hashlib.md5(ctypes.string_at(rq, host.nbytes)).hexdigest() if rq else None
                if rq:
                    libc.free(rq)
                rbuf = ctypes.create_string_buffer(ref_wrapped, len(ref_wrapped))
                q3 = L.SZ_decompress(0, rbuf, len(ref_wrapped), *dims)      # and this library the reference's
                back_md5 = hashlib.md5(ctypes.string_at(q3, host.nbytes)).hexdigest() if q3 else None
                if q3:
                    L.free(q3)
                R.SZ_Finalize()
                e2e_default["reference"] = {"compress_GBps": round(nbytes_in / tr / 1e9, 4), "decompress_GBps": round(nbytes_in / trd / 1e9, 4), "cores": 1, "wrapped_bytes": len(ref_wrapped),
                                            "what": "oracle/_ref/libSZ.so (the unmodified reference) on the same array and configuration, one pass each way; its decompression is of THIS library's wrapped stream"}
                e2e_default["decoded_md5_equal"] = bool(dec_md5 is not None and dec_md5 == ref_dec_md5 == back_md5)

    cpu, cpu_mt = (None, None)
    if not args.no_cpu_baseline and world == 1:
        cpu, cpu_mt = cpu_baselines(host, n, size)

    line = {"metric": f"compression GB/s (input), {n}^3 float32 ABS 1e-4", "value": round(value, 3), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n}x{n}x{n} float32 S-field (smooth sinusoid) per GPU, ABS 1e-4, SZ 2.1 path with adaptive Lorenzo+regression "
                                   "selection per block (on this field every block chooses Lorenzo; m_field is the mixed case), stream "
                                   "bit-identical to the reference; value-range reduction included in the step (fused into the fit pass); input and output resident in HBM; "
                                   + ("one blocking compression after the other (the shape of SZ_compress_args); `concurrent` gives 1 / 2 / 4 arrays in flight (szhip_pool)" if inflight == 1 else
                                      f"{inflight} compressions in flight per GPU, every lane on its own copy of the field (szhip_pool, one context per lane; "
                                      "`single_call_GBps` = one blocking call after the other, `concurrent` gives 1 / 2 / 4)"),
                       "error_bound_mode": "ABS", "abs_err_bound": EB, "slabs": world, "arrays_in_flight": inflight},
            "ratio": round(nbytes_in / size, 6), "out_bytes": size, "max_abs_err": max_abs_err, "psnr": round(psnr, 6),
            "intervals": stats.intervals, "reg_blocks": stats.n_reg_blocks, "unpredictable": stats.n_unpred,
            "decompress_GBps": round(nbytes_in / td / 1e9, 3),
            "value_single_call": single_call["GB/s"], "single_call_GBps": single_call["GB/s"], "single_call": single_call, "step_ms": step_ms,
            # (the HIP runtime stalls a call for 3 - 5 ms now and then -- one step in about a hundred, profiles/r05_timed_region_five_runs.txt and the step lists of
            #  the round's lines --; `value` is the contract's K steps over their total time and carries such a step when it meets one; the median step is beside it)
            "median_step_ms": round(float(np.median(step_ms)), 4) if step_ms else None,
            "value_from_median_step": round(world * nbytes_in / (float(np.median(step_ms)) * 1e-3) / 1e9, 3) if step_ms else None,
            "phase_ms": {"prequant": round(stats.ms_prequant, 3), "quant": round(stats.ms_quant, 3),
                         "entropy": round(stats.ms_entropy, 3), "host_glue": round(stats.ms_host, 3), "compress_call_total": round(stats.ms_total, 3),
                         "decompress_quant": round(dst.ms_quant, 3), "decompress_total": round(dst.ms_total, 3)},
            "per_rank": per_rank,
            "roofline": roofline, "concurrent": concurrent, "m_field": mfield, "other_paths": other, "omp_container": omp, "e2e": e2e, "e2e_default": e2e_default, "cpu_baseline": cpu, "cpu_baseline_mt": cpu_mt}
    print(json.dumps(line))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def run_c4(args, torch, dist, world, rank, local_rank, dev):
    """BASELINE configs[3]: 1024^3 float64, REL 1e-3, slabs of the outer dimension, RCCL all-reduce of the range + all-gather of the streams."""
    import ctypes
    import sz_amd
    from sz_amd import slab
    from sz_amd.fields import s_field
    N = args.c4_edge
    nslabs = max(world, 8) if world < 8 else world          # the array is cut for 8 GPUs; with fewer ranks each rank still runs ONE slab of that cut
    bounds = slab.slab_bounds(N, nslabs)
    z0, z1 = bounds[rank]
    planes = z1 - z0
    host = s_field(planes, N, N, np.float64, z0=z0)
    x = torch.from_numpy(host).to(dev)
    nbytes_in = host.nbytes
    ctx = sz_amd.HipContext(0 if getattr(args, "dry_run", False) else local_rank)     # (the CPU shim of --dry-run has one "device")
    out_cap = nbytes_in // 2 + (1 << 20)
    out_bufs = [torch.empty(out_cap, dtype=torch.uint8, device=dev) for _ in range(2)]
    gather = slab.StreamGather() if world > 1 else None
    pending, step_no = [], [0]
    eb_box = [0.0]

    def one_step():
        lo, hi = ctx.minmax(x.data_ptr(), True, x.numel(), np.float64)                 # local range scan on the GPU
        lo, hi = slab.global_minmax(lo, hi, device=dev)                                # 2 scalars over RCCL (sz_float.c:2845-2866 needs the GLOBAL range)
        eb = 1e-3 * (hi - lo)
        eb_box[0] = eb
        meta = sz_amd.make_meta(np.float64, err_mode=sz_amd.REL, rel_ratio=1e-3, vmin=lo, vmax=hi)
        ob = out_bufs[step_no[0] & 1]; step_no[0] += 1
        out = ctypes.c_void_p(ob.data_ptr()); nn = ctypes.c_size_t(out_cap); st = sz_amd.szhip_stats()
        p = sz_amd.szhip_params(100, 0.99, 65536, 0)
        rc = sz_amd.lib().szhip_compress(ctx._h, 1, x.data_ptr(), 1, planes, N, N, eb, ctypes.byref(p), meta, len(meta), 2,
                                         ctypes.byref(out), ctypes.byref(nn), ctypes.byref(st))
        if rc:
            raise RuntimeError(f"szhip_compress failed: {sz_amd.lib().szhip_last_error(ctx._h)}")
        if world > 1:
            pending.append(gather.begin(ob, nn.value))
            if len(pending) > 1:
                gather.end(pending.pop(0))
        return nn.value, st, ob

    def sync_all():
        _sync(torch)
        if world > 1:
            dist.barrier()
        _sync(torch)

    for _ in range(args.warmup):
        one_step()
    while pending:
        gather.end(pending.pop(0))
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        size, st, ob = one_step()
    parts = None
    while pending:
        parts = gather.end(pending.pop(0))
    sync_all()
    elapsed = time.perf_counter() - t0
    # every rank decompresses its OWN sub-stream (from the gathered set when there is one) and checks the bound
    eb = eb_box[0]
    dec = torch.empty_like(x)
    src = ob if parts is None else parts[0][rank].contiguous()
    _sync(torch); td = time.perf_counter()
    ctx.decompress(src.data_ptr(), True, size, 4 + 36 + 8, (planes, N, N), np.float64, dec.data_ptr(), True)
    _sync(torch); td = time.perf_counter() - td
    max_err = float((dec - x).abs().max().item())
    # ---- the same slab through the reference's OpenMP container (outside the timed region, rank 0, hardware only): boxes of
    #      4 x 32 x 32 (thread_num = the power of two whose grid gives 32 x 32 faces), the same absolute bound
    omp_c4 = None
    if rank == 0 and not getattr(args, "dry_run", False) and not args.no_omp and N % 32 == 0:
        try:
            # (the slabs of the SZ 2.1 path are cut on multiples of its block edge: 132 planes at N = 1024; the container's boxes need no such
            #  alignment but must divide the array: its slab is the plain N / 8 = 128 planes -- the first 128 of this rank's)
            nbx = N // 32
            po = (N // nslabs) // nbx * nbx if nbx & (nbx - 1) == 0 else 0
            tn = nbx ** 3 if po >= nbx and po <= planes else 0
            if tn:
                xo = x[:po]
                nbytes_o = xo.numel() * 8
                deco = torch.empty_like(xo)
                meta_o = bytes(32)
                for _ in range(2):
                    optr, osize, ost = ctx.compress_omp(xo.data_ptr(), True, (po, N, N), np.float64, eb, tn, meta_o, out_on_device=True)
                _sync(torch); t1 = time.perf_counter()
                for _ in range(max(3, min(args.steps, 5))):
                    optr, osize, ost = ctx.compress_omp(xo.data_ptr(), True, (po, N, N), np.float64, eb, tn, meta_o, out_on_device=True)
                _sync(torch); to = (time.perf_counter() - t1) / max(3, min(args.steps, 5))
                ctx.decompress_omp(optr, True, osize, len(meta_o), (po, N, N), np.float64, deco.data_ptr(), True)
                _sync(torch); t1 = time.perf_counter()
                for _ in range(3):
                    odst = ctx.decompress_omp(optr, True, osize, len(meta_o), (po, N, N), np.float64, deco.data_ptr(), True)
                _sync(torch); tod = (time.perf_counter() - t1) / 3
                oerr = float((deco - xo).abs().max().item())
                omp_c4 = {"container": f"the reference's OpenMP container on {po}x{N}x{N} of this rank's slab (not part of `value`)", "boxes": int(ost.n_blocks), "GB/s": round(nbytes_o / to / 1e9, 2),
                          "ms": round(to * 1e3, 3), "decompress_GBps": round(nbytes_o / tod / 1e9, 2), "out_bytes": int(osize), "ratio": round(nbytes_o / osize, 4),
                          "max_abs_err": oerr, "bound_held": bool(oerr <= eb),
                          "phase_ms": {"prequant": round(ost.ms_prequant, 3), "quant": round(ost.ms_quant, 3), "entropy": round(ost.ms_entropy, 3),
                                       "decompress_entropy": round(odst.ms_entropy, 3), "decompress_quant": round(odst.ms_quant, 3)},
                          "quant_kernel_frac_of_hbm_peak_on_N_sizeof_T": round(nbytes_o / (max(ost.ms_quant, 1e-9) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        except Exception as e:  # noqa: BLE001 -- an extra object, never the headline
            omp_c4 = {"error": repr(e)[:200]}
    stats_t = torch.tensor([elapsed, max_err, float(size), td], dtype=torch.float64, device=dev)
    if world > 1:
        allst = [torch.zeros_like(stats_t) for _ in range(world)]
        dist.all_gather(allst, stats_t)
    else:
        allst = [stats_t]
    if rank == 0:
        el = max(float(s[0]) for s in allst)
        worst = max(float(s[1]) for s in allst)
        total_out = sum(float(s[2]) for s in allst)
        line = {"metric": "compression GB/s (input), 1024^3 float64 REL 1e-3, slab-sharded", "value": round(world * nbytes_in / (el / args.steps) / 1e9, 3),
                "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"{N}^3 float64 S-field cut into {nslabs} slabs of the outer dimension (block-aligned cuts); {world} rank(s), each "
                                       f"running ONE slab ({planes}x{N}x{N}); REL 1e-3 on the all-reduced range; range scan, all-reduce and the "
                                       "all-gather of the sub-streams inside the step", "slabs_run": world, "slabs_of_array": nslabs},
                "eb": eb, "max_abs_err": worst, "bound_held": bool(worst <= eb), "out_bytes_all_ranks": int(total_out),
                "ratio": round(world * nbytes_in / total_out, 4), "decompress_GBps_per_gpu": round(nbytes_in / max(float(s[3]) for s in allst) / 1e9, 2),
                "phase_ms_rank0": {"prequant": round(st.ms_prequant, 3), "quant": round(st.ms_quant, 3), "entropy": round(st.ms_entropy, 3)},
                "omp_container": omp_c4}
        print(json.dumps(line))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="headline", choices=["headline", "c4"])
    ap.add_argument("--edge", type=int, default=EDGE, help="cube edge of the headline config (512 = the BASELINE config)")
    ap.add_argument("--c4-edge", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--other-paths", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.add_argument("--no-other-paths", action="store_true", help="skip the SZ 1.4 container, the 2-D array and the 1-D series (one line each)")
    ap.add_argument("--omp-boxes", type=int, default=-1, help="boxes (thread_num) of the OpenMP-container object; default: 32^3 boxes (4096 at 512^3); 0 = skip")
    ap.add_argument("--no-omp", action="store_true", help="skip the OpenMP-container object")
    ap.add_argument("--omp-ref-child", nargs=2, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-m-field", action="store_true")
    ap.add_argument("--timed-only", action="store_true", help="only the headline: priming, warm-up, the timed steps, one decompression (for "
                    "rocprofv3 --stats: every launch of the sweep kernel then runs as in the timed region)")
    ap.add_argument("--dry-run", action="store_true", help="CPU rehearsal of the entry on gloo + the HIP-on-CPU shim (tests only)")
    ap.add_argument("--inflight", type=int, default=1, help="arrays in flight per GPU in the timed region (szhip_pool lanes); 1 (the default since round 5: what a caller of "
                    "SZ_compress_args sees) = one blocking call after the other; the `concurrent` object of the line gives 1 / 2 / 4")
    args = ap.parse_args()
    if args.omp_ref_child:
        return _omp_ref_child(int(args.omp_ref_child[0]), int(args.omp_ref_child[1]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        # `python bench.py --gpus N` (the shape of the N = 1 command): start the N ranks here, one per GPU, exactly as
        # `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` would
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if args.timed_only:
        args.no_cpu_baseline = args.no_m_field = True
    if args.dry_run:
        # CPU rehearsal of the multi-rank entry (tests/test_distributed_cpu.py): gloo + the product's code on the HIP-on-CPU shim, tiny
        # arrays, no GPU.  Nothing it prints is a measurement.
        if not os.environ.get("SZ_AMD_LIB"):
            raise SystemExit("--dry-run needs SZ_AMD_LIB = tests/sim/libszhip_sim.so")
        dev = torch.device("cpu")
        args.no_cpu_baseline = args.no_m_field = True
        args.inflight = 1                      # (the shim executes one launch at a time)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        run_headline(args, torch, dist, world, rank, local_rank, dev)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    (run_c4 if args.config == "c4" else run_headline)(args, torch, dist, world, rank, local_rank, dev)


if __name__ == "__main__":
    main()
