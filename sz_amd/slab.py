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


def slab_bounds(n0, world):
    """Contiguous split of the slowest dimension; the first n0 % world slabs get one extra plane."""
    base, rem = divmod(n0, world)
    out, z = [], 0
    for r in range(world):
        h = base + (1 if r < rem else 0)
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
    """The same all-gather, overlapped with the next compression step: begin() launches the gather of this step's sub-stream
    and returns; end() completes one.  The payload travels from a private copy, so the caller's buffer may be overwritten as
    soon as begin() returns; two gathers may be in flight (two sets of buffers).  What end() returns are views of those
    buffers: valid until the second begin() after the one that started this gather."""

    def __init__(self, depth=2):
        self.depth = depth
        self.send = [None] * depth
        self.recv = [None] * depth
        self.slot = 0

    def begin(self, local_stream, nbytes):
        import torch
        import torch.distributed as dist
        n = int(nbytes)
        world = dist.get_world_size()
        dev = local_stream.device
        sizes = torch.zeros(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(sizes, torch.tensor([n], dtype=torch.int64, device=dev))
        sizes = [int(x) for x in sizes.tolist()]
        cap = (max(sizes) + 255) // 256 * 256
        k = self.slot
        self.slot = (self.slot + 1) % self.depth
        if self.send[k] is None or self.send[k].numel() < cap:
            self.send[k] = torch.empty(cap + cap // 4, dtype=torch.uint8, device=dev)
            self.recv[k] = torch.empty(world * (cap + cap // 4), dtype=torch.uint8, device=dev)
        send = self.send[k][:cap]
        send[:n] = local_stream[:n]
        if send.is_cuda:
            torch.cuda.current_stream(dev).synchronize()   # the copy is done before the caller reuses local_stream (its producer
                                                           # runs on a stream of its own)
        recv = self.recv[k][:world * cap]
        work = dist.all_gather_into_tensor(recv, send, async_op=True)
        return (work, recv, sizes, cap)

    @staticmethod
    def end(handle):
        work, recv, sizes, cap = handle
        work.wait()
        return [recv[r * cap:r * cap + sizes[r]] for r in range(len(sizes))], sizes
