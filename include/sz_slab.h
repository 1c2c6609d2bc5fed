/*
 * sz_slab.h -- the slab container of the multi-GPU path for C callers.
 *
 * A large array is cut along its SLOWEST dimension into slabs whose cuts fall on multiples of the block edge; every slab is
 * compressed as an array of its own by SZ_compress_args (zero Lorenzo halo at the cut: what the reference does at array
 * faces, sz/src/sz_float.c:6685-6689), so every sub-stream is a plain SZ stream that SZ_decompress reads.  The container
 * only concatenates them (little endian):
 *     "SZSL" | u32 version = 1 | u32 slabs | u32 dtype (0 float, 1 double) | u64 dims[3] (slowest .. fastest, whole array)
 *     | slabs x { u64 z_begin, u64 z_end, u64 stream_bytes } | stream_0 | stream_1 | ...
 * sz_amd/slab.py writes the same bytes from N ranks (one slab per GPU, one all-gather); these entry points write and read
 * them without Python.  The reference has no counterpart (it has no multi-device path): additive, like szhip.h.
 */
#ifndef SZ_SLAB_H
#define SZ_SLAB_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct sz_slab_entry {
    size_t z_begin, z_end;      /* planes [z_begin, z_end) of the slowest dimension */
    size_t offset, bytes;       /* where the slab's SZ stream lies inside the container */
} sz_slab_entry;

/* cuts of a slowest dimension of n0 planes into `slabs` parts on multiples of `block` (6 for 3-D: SZ's block edge); bounds[2*s],
 * bounds[2*s+1] receive [begin, end) of slab s.  The rule of sz_amd/slab.py:slab_bounds. */
void sz_slab_bounds(size_t n0, int slabs, int block, size_t *bounds);

/* concatenate `slabs` finished sub-streams; returns a malloc'd container (caller frees) and its size, NULL on bad arguments */
unsigned char *sz_slab_pack(int dataType, const size_t dims[3], int slabs, const size_t *bounds, const unsigned char *const *streams,
                            const size_t *stream_bytes, size_t *outSize);

/* parse a container: fills dataType, dims, the number of slabs and up to max_entries entries; returns SZ_SCES or SZ_NSCS */
int sz_slab_unpack(const unsigned char *blob, size_t len, int *dataType, size_t dims[3], int *slabs, sz_slab_entry *entries, int max_entries);

/* compress a 3-D array (r3 slowest .. r1 fastest, the argument order of SZ_compress_args) as `slabs` slabs on the calling
 * process's GPU, one after the other, into one container.  Bounds that depend on the array -- REL, ABS_AND_REL, ABS_OR_REL (value range),
 * PSNR (value range), NORM (element count) -- are derived from the WHOLE array once, as the multi-GPU path does with its all-reduce, and
 * every slab is then compressed with that absolute bound.  Returns a malloc'd container or NULL. */
unsigned char *sz_slab_compress(int dataType, void *data, size_t *outSize, int errBoundMode, double absErrBound, double relBoundRatio,
                                double pwrBoundRatio, size_t r3, size_t r2, size_t r1, int slabs);

/* The same container from `ndev` GPUs of this node at once (round 4; sz_amd/csrc/sz_slab_multi.cpp): slab s on device devices[s] (NULL:
 * 0 .. ndev-1), one host thread and one HIP context per device, every slab through the ordinary SZ_compress_args.  Range-based bound modes
 * all-reduce the slabs' value ranges (ncclAllReduce, min / max); the sub-streams are all-gathered onto every device (ncclAllGather of the
 * sizes, the payloads by one grouped broadcast per slab) -- RCCL over xGMI, looked up at run time; without librccl.so, or when a device is
 * named twice, the exchange goes through host memory.  The bytes are those of sz_slab_compress(..., slabs = ndev).  ABS, REL, ABS_AND_REL,
 * ABS_OR_REL.  `info` (may be NULL) reports what was used. */
typedef struct sz_slab_multi_info {
    int devices, used_rccl;             /* used_rccl: 1 = ranges and sub-streams went over RCCL */
    size_t gathered_bytes;              /* bytes of sub-streams every device held after the all-gather (RCCL only) */
    double seconds_total, seconds_slowest_slab;
} sz_slab_multi_info;
unsigned char *sz_slab_compress_multi(int dataType, void *data, size_t *outSize, int errBoundMode, double absErrBound, double relBoundRatio,
                                      double pwrBoundRatio, size_t r3, size_t r2, size_t r1, int ndev, const int *devices, sz_slab_multi_info *info);

/* sz_slab_compress_multi keeps its communicator, its per-device contexts and exchange buffers for the next call with the same device list (a communicator
 * costs hundreds of milliseconds to make, a context its workspaces); this gives them back.  SZ_Finalize calls it.  SZ_SLAB_MULTI_CACHE=0: nothing is kept. */
void sz_slab_multi_release(void);

/* decompress every slab of a container into one malloc'd array of dims[0] x dims[1] x dims[2] values (caller frees); NULL on error */
void *sz_slab_decompress(const unsigned char *blob, size_t len, int *dataType, size_t dims[3]);

#ifdef __cplusplus
}
#endif
#endif
