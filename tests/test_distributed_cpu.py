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
