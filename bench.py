#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X SZ 2.1 hot path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one pass of the hot path (fit + interval optimiser + predictor selection + predict/quantise + Huffman
encode, into the reference's SZ 2.1 stream) over one 512x512x512 float32 array (BASELINE.json configs[1]: smooth
sinusoid "S-field", ABS 1e-4) that is already resident in HBM; the stream is left in HBM.  With N ranks every rank
owns one such slab of an (N*512)x512x512 array (weak scaling, no data-path collective) and the step ends with one
all-gather of the variable-length sub-streams (RCCL), launched asynchronously so that it overlaps the next step's
compression; every gather is completed inside the timed region.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline     -- the predict+quantise wavefront kernel: algorithmic bytes (N*4, the array read once) / its average
                  duration measured with HIP events on the library's stream, against the 8 TB/s HBM3E peak.
  cpu_baseline -- the oracle (a C restatement of the reference CPU loops, oracle/), single core, on the same array.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EDGE = 512
EB = 1e-4
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--edge", type=int, default=EDGE, help="cube edge (512 = the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--other-paths", action="store_true",
                    help="also time the SZ 1.4 container, a 2-D array and a 1-D series (extra kernels: keep it off when profiling the headline kernel)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import sz_amd
    from sz_amd import slab
    from sz_amd.fields import s_field

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    n = args.edge
    # this rank's slab of the (world*n) x n x n S-field (z offset = rank*n); host generation is exact numpy float64 math
    host = s_field(n, n, n, np.float32, z0=rank * n)
    x = torch.from_numpy(host).to(dev)
    nbytes_in = host.nbytes
    vmin, vmax = float(host.min()), float(host.max())
    meta = sz_amd.make_meta(np.float32, err_mode=sz_amd.ABS, abs_bound=EB, vmin=vmin, vmax=vmax)
    ctx = sz_amd.HipContext(local_rank)
    out_cap = nbytes_in // 2 + (1 << 20)
    out_buf = torch.empty(out_cap, dtype=torch.uint8, device=dev)
    # N > 1: the all-gather of step k's sub-streams runs while step k+1 compresses (RCCL on its own stream; the payload travels
    # from a private copy).  All gathers are completed inside the timed region.
    gather = slab.StreamGather() if world > 1 else None
    pending = []

    import ctypes

    def one_step():
        out = ctypes.c_void_p(out_buf.data_ptr())
        nn = ctypes.c_size_t(out_cap)
        st = sz_amd.szhip_stats()
        p = sz_amd.szhip_params(100, 0.99, 65536, 0)
        rc = sz_amd.lib().szhip_compress(ctx._h, 0, x.data_ptr(), 1, n, n, n, EB, ctypes.byref(p), meta, len(meta), 2,
                                         ctypes.byref(out), ctypes.byref(nn), ctypes.byref(st))
        if rc:
            raise RuntimeError(f"szhip_compress failed: {sz_amd.lib().szhip_last_error(ctx._h)}")
        if world > 1:
            if os.environ.get("SZ_BENCH_SYNC_GATHER"):          # fallback: the plain, non-overlapped all-gather
                slab.allgather_streams(out_buf, nn.value)
            else:
                pending.append(gather.begin(out_buf, nn.value))
                if len(pending) > 1:
                    slab.StreamGather.end(pending.pop(0))
        return nn.value, st

    def drain():
        while pending:
            slab.StreamGather.end(pending.pop(0))

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    drain()
    sync_all()
    t0 = time.perf_counter()
    quant_ms, stats = [], None
    for _ in range(args.steps):
        size, stats = one_step()
        quant_ms.append(stats.ms_quant)
    drain()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * nbytes_in / (elapsed / args.steps) / 1e9

    # ---- quality of the result (outside the timed region): decompress on the GPU, compare with the input
    dec = torch.empty_like(x)
    dst = ctx.decompress(out_buf.data_ptr(), True, size, 4 + 28 + 8, (n, n, n), np.float32, dec.data_ptr(), True)
    err = (dec - x).abs()
    max_abs_err = float(err.max().item())
    mse = float((err * err).double().sum().item()) / x.numel()
    psnr = 20 * np.log10(float((x.max() - x.min()).item())) - 10 * np.log10(mse)
    # decompression throughput (same array, 3 passes)
    torch.cuda.synchronize(); td = time.perf_counter()
    for _ in range(3):
        ctx.decompress(out_buf.data_ptr(), True, size, 4 + 28 + 8, (n, n, n), np.float32, dec.data_ptr(), True)
    torch.cuda.synchronize(); td = (time.perf_counter() - td) / 3

    if rank != 0:
        if world > 1:
            dist.barrier(); dist.destroy_process_group()
        return

    quant_avg_ms = float(np.mean(quant_ms))
    achieved = nbytes_in / (quant_avg_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside the timed process);
    # only valid for the workload it was measured on
    traffic = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_pencil.json")))
        if n == EDGE:
            traffic = pm["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    roofline = {"bound": "hbm", "kernel": "k_pencil<float,false>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_source": "profiles/r01_pmc_traffic_pencil.json (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes)",
                "algorithmic_bytes_per_launch": nbytes_in, "avg_kernel_ms": round(quant_avg_ms, 4)}

    # ---- the other paths of the same library, one line each (outside the timed region; single GPU only): the SZ 1.4 container
    #      (withLinearRegression = NO) on the same array, and a 2-D array through the SZ 2.1 path
    other = None
    if args.other_paths and world == 1 and n == EDGE:
        def timed(fn, reps=3):
            fn(); torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(reps):
                r = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / reps, r

        def run(fn_name, ptr, dims, extra, mbytes, eb=EB):
            out = ctypes.c_void_p(out_buf.data_ptr()); nn = ctypes.c_size_t(out_cap); st = sz_amd.szhip_stats()
            p = sz_amd.szhip_params(100, 0.99, 65536, 0)
            rc = getattr(sz_amd.lib(), fn_name)(ctx._h, 0, ptr, 1, *dims, eb, *extra, ctypes.byref(p), mbytes, len(mbytes), 2,
                                                ctypes.byref(out), ctypes.byref(nn), ctypes.byref(st))
            if rc:
                raise RuntimeError(f"{fn_name} failed: {sz_amd.lib().szhip_last_error(ctx._h)}")
            return nn.value
        rng = vmax - vmin
        med = float(np.float32(np.float32(vmin) + np.float32(rng) / np.float32(2)))
        meta14 = bytes([meta[0], meta[1], meta[2], 0x40]) + bytes(meta[4:])
        t14, size14 = timed(lambda: run("szhip_compress_sz14", x.data_ptr(), (n, n, n), (float(np.float32(rng)), med), meta14))
        from sz_amd.fields import plane_field
        p2 = torch.from_numpy(plane_field(4096, 4096)).to(dev)
        meta2 = sz_amd.make_meta(np.float32, err_mode=sz_amd.ABS, abs_bound=EB, vmin=float(p2.min().item()), vmax=float(p2.max().item()))
        t2, size2 = timed(lambda: run("szhip_compress", p2.data_ptr(), (0, 4096, 4096), (), meta2))
        # a 1-D series (16 Mi values): a random walk with a sine, resident in HBM; the chain is cut at its certain restarts
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
                 "sz_1d_16Mi_f32_abs1e-3": {"GB/s": round(s1.numel() * 4 / t1d / 1e9, 2), "ms": round(t1d * 1e3, 3), "out_bytes": size1d}}
        del p2, s1

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O  # checker, timed here as the CPU baseline ("port" of the reference loops)
        sample = host if n <= 512 else host[:512]
        t1 = time.perf_counter()
        ref, _ = O.compress(sample, O.ABS, EB)
        tc = time.perf_counter() - t1
        cpu = {"value": round(sample.nbytes / tc / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
               "sample": f"one compress pass of the full {sample.shape[0]}x{n}x{n} float32 S-field by oracle/liboracle.so ({tc:.1f} s)",
               "stream_bytes": len(ref), "gpu_stream_identical": bool(len(ref) == size)}

    line = {"metric": "compression GB/s (input), 512^3 float32 ABS 1e-4", "value": round(value, 3), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n}x{n}x{n} float32 S-field (smooth sinusoid) per GPU, ABS 1e-4, adaptive Lorenzo+regression "
                                   "(SZ 2.1 stream, bit-identical to the reference); input and output resident in HBM",
                       "error_bound_mode": "ABS", "abs_err_bound": EB, "slabs": world},
            "ratio": round(nbytes_in / size, 6), "out_bytes": size, "max_abs_err": max_abs_err, "psnr": round(psnr, 6),
            "intervals": stats.intervals, "reg_blocks": stats.n_reg_blocks, "unpredictable": stats.n_unpred,
            "decompress_GBps": round(nbytes_in / td / 1e9, 3),
            "phase_ms": {"prequant": round(stats.ms_prequant, 3), "quant": round(stats.ms_quant, 3), "entropy": round(stats.ms_entropy, 3),
                         "host_glue": round(stats.ms_host, 3), "total": round(stats.ms_total, 3),
                         "decompress_quant": round(dst.ms_quant, 3), "decompress_total": round(dst.ms_total, 3)},
            "roofline": roofline, "cpu_baseline": cpu, "other_paths": other}
    print(json.dumps(line))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
