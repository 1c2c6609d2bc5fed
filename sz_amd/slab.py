"""Outer-dimension slab sharding across the GPUs of one node.

Each rank compresses its own slab as an independent array (Lorenzo halo at the cut = zeros, exactly what the
reference does at array faces), so the parity oracle is the reference run on each slab.  The only exchanges are
(1) an all-reduce of {min,max} for the range-based bound modes and (2) one all-gather of the variable-length
sub-streams into the container below.  Plain torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in CPU tests).

Container layout (little endian):
    magic "SZSL" | u32 version=1 | u32 world | u32 dtype (0 f32, 1 f64) | u64 dims[3] (slowest..fastest, whole array)
    | world x { u64 z_begin, u64 z_end, u64 stream_bytes } | stream_0 | stream_1 | ...
"""
import struct

import numpy as np

MAGIC = b"SZSL"


def slab_bounds(n0, world, block=6):
    """Contiguous split of the slowest dimension into `world` slabs whose cuts fall on multiples of the block edge, so
    that every slab's block grid is the whole array's; the planes that do not fill a block go to the last slab.  Arrays with fewer
    blocks than ranks are cut plane-wise."""
    units = n0 // block
    if units < world:
        base, rem = divmod(n0, world)
        out, z = [], 0
        for r in range(world):
            h = base + (1 if r < rem else 0)
            out.append((z, z + h))
            z += h
        return out
    base, rem = divmod(units, world)
    out, z = [], 0
    for r in range(world):
        h = (base + (1 if r < rem else 0)) * block
        if r == world - 1:
            h = n0 - z
        out.append((z, z + h))
        z += h
    return out


def pack_container(dtype, dims, bounds, streams):
    world = len(streams)
    head = MAGIC + struct.pack("<III", 1, world, 0 if np.dtype(dtype) == np.float32 else 1) + struct.pack("<QQQ", *dims)
    for (z0, z1), s in zip(bounds, streams):
        head += struct.pack("<QQQ", z0, z1, len(s))
    return head + b"".join(bytes(s) for s in streams)


def unpack_container(blob):
    if blob[:4] != MAGIC:
        raise ValueError("not a slab container")
    ver, world, dt = struct.unpack_from("<III", blob, 4)
    if ver != 1:
        raise ValueError("unknown slab container version")
    dims = struct.unpack_from("<QQQ", blob, 16)
    off = 40
    table = [struct.unpack_from("<QQQ", blob, off + 24 * r) for r in range(world)]
    off += 24 * world
    streams = []
    for z0, z1, nbytes in table:
        streams.append(blob[off:off + nbytes])
        off += nbytes
    return (np.float32 if dt == 0 else np.float64), dims, [(t[0], t[1]) for t in table], streams


def global_minmax(local_min, local_max, device=None):
    """All-reduce of the value range (needed by REL / ABS_AND_REL / ABS_OR_REL / PSNR, sz_float.c:2845-2866)."""
    import torch
    import torch.distributed as dist
    lo = torch.tensor([local_min], dtype=torch.float64, device=device)
    hi = torch.tensor([local_max], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return float(lo.item()), float(hi.item())


def allgather_streams(local_stream, nbytes=None):
    """One all-gather of the variable-length sub-streams.  `local_stream`: 1-D uint8 torch tensor (any device) holding at
    least `nbytes` valid bytes.  Returns (list of per-rank uint8 tensors trimmed to their sizes, sizes)."""
    import torch
    import torch.distributed as dist
    n = int(local_stream.numel() if nbytes is None else nbytes)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local_stream[:n]], [n]
    world = dist.get_world_size()
    dev = local_stream.device
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, torch.tensor([n], dtype=torch.int64, device=dev))
    sizes = [int(x) for x in sizes.tolist()]
    cap = max(sizes)
    cap = (cap + 255) // 256 * 256
    send = torch.zeros(cap, dtype=torch.uint8, device=dev)
    send[:n] = local_stream[:n]
    recv = torch.empty(world * cap, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send)
    return [recv[r * cap:r * cap + sizes[r]] for r in range(world)], sizes


class StreamGather:
    """The same all-gather, overlapped with the next compression step: begin() launches the gather of this step's sub-stream and
    returns without touching the host; end() completes one.  No size exchange per step: every payload carries its length in its
    first 8 bytes and travels in a buffer sized from what the streams needed before (+25 %); the first gather (and one whose
    stream outgrew the buffer -- every rank sees that in the lengths) is done with the plain, blocking form.  The caller keeps
    `local_stream` untouched until end() (bench.py alternates two output buffers)."""

    def __init__(self, depth=2):
        self.depth = depth
        self.send = [None] * depth
        self.recv = [None] * depth
        self.slot = 0
        self.cap = 0

    def begin(self, local_stream, nbytes):
        import torch
        import torch.distributed as dist
        n = int(nbytes)
        world = dist.get_world_size()
        dev = local_stream.device
        if self.cap == 0:                                  # first use: learn the sizes the blocking way
            parts, sizes = allgather_streams(local_stream, n)
            self.cap = ((max(sizes) * 5 // 4 + 8) + 255) // 256 * 256
            return ("done", parts, sizes)
        cap = self.cap
        k = self.slot
        self.slot = (self.slot + 1) % self.depth
        if self.send[k] is None or self.send[k].numel() != cap:
            self.send[k] = torch.empty(cap, dtype=torch.uint8, device=dev)
            self.recv[k] = torch.empty(world * cap, dtype=torch.uint8, device=dev)
        send = self.send[k]
        send[:8] = torch.tensor([n], dtype=torch.int64).view(torch.uint8).to(dev, non_blocking=True)
        m = min(n, cap - 8)
        send[8:8 + m] = local_stream[:m]
        work = dist.all_gather_into_tensor(self.recv[k], send, async_op=True)
        return ("async", work, self.recv[k], cap, local_stream, n)

    def end(self, handle):
        if handle[0] == "done":
            return handle[1], handle[2]
        _, work, recv, cap, local_stream, n = handle
        work.wait()
        world = recv.numel() // cap
        sizes = [int(x) for x in recv.view(world, cap)[:, :8].contiguous().view(torch_int64()).flatten().tolist()]
        if max(sizes) > cap - 8:                           # a stream outgrew the buffers: every rank sees it, all redo this one
            parts, sizes = allgather_streams(local_stream, n)
            self.cap = ((max(sizes) * 5 // 4 + 8) + 255) // 256 * 256
            return parts, sizes
        return [recv[r * cap + 8:r * cap + 8 + sizes[r]] for r in range(world)], sizes


def torch_int64():
    import torch
    return torch.int64
