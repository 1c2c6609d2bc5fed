/*
 * szhip.h -- C ABI of the MI355X (gfx950) HIP layer of the SZ 2.1 hot path.
 *
 * Everything below is `extern "C"`, plain pointers and sizes, no globals: this is what the
 * reference's own host code would bind instead of its CPU loops.  The reference has no FFI for
 * this path (it is one C library); the entry points therefore replace these internal call sites:
 *
 *   szhip_compress()      replaces  SZ_compress_float_3D_MDQ_nonblocked_with_blocked_regression
 *                                   (sz/src/sz_float.c:6527-7489) and the double twin
 *                                   (sz/src/sz_double.c:5904), called from SZ_compress_args_float
 *                                   (sz/src/sz_float.c:2974,3012) / _double
 *   szhip_decompress()    replaces  decompressDataSeries_float_3D_nonblocked_with_blocked_regression
 *                                   (sz/src/szd_float.c:3483-5866) / _double (szd_double.c:3316),
 *                                   called from SZ_decompress_args_float (sz/src/szd_float.c:133)
 *   szhip_minmax()        replaces  computeRangeSize_float/_double (sz/src/dataCompression.c:102,149)
 *
 * The streams produced/consumed are the reference's SZ 2.1 "raBytes" streams, byte for byte
 * (pre-lossless, i.e. what the reference emits with szMode = SZ_BEST_SPEED).
 */
#ifndef SZHIP_H
#define SZHIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SZHIP_OK            0
#define SZHIP_ERR_NODEVICE -1   /* no HIP device / runtime failure */
#define SZHIP_ERR_ARG      -2
#define SZHIP_ERR_UNSUP    -3   /* valid SZ input this layer does not cover yet */
#define SZHIP_ERR_STREAM   -4   /* malformed compressed stream */
#define SZHIP_ERR_INTERNAL -5   /* kernel-side timeout or inconsistency */
#define SZHIP_CONSTANT      1   /* szhip_compress with SZHIP_RANGE_FROM_DATA: max - min <= eb, nothing was encoded (stats.vmin / vmax are
                                   set, *out stays NULL): the caller writes the reference's constant-data stream (sz_float.c:2861) */

#define SZHIP_F32 0
#define SZHIP_F64 1

typedef struct szhip_ctx szhip_ctx; /* owns one HIP stream and grow-only device workspaces */

/* parameters the hot path reads from sz_params / sz_exedata (sz/include/sz.h:164-217), passed by value */
typedef struct szhip_params {
    int      sample_distance;        /* sampleDistance */
    float    pred_threshold;         /* predThreshold */
    unsigned max_quant_intervals;    /* max_quant_intervals (maxRangeRadius = half of it) */
    unsigned quantization_intervals; /* 0: optimise (optQuantMode 1); else fixed capacity */
    unsigned flags;                  /* SZHIP_RANGE_FROM_DATA: szhip_compress takes the array's value range from its own fit pass
                                        (one read of the input less than szhip_minmax + szhip_compress) and writes it into the
                                        range field of the parameter bytes in `meta` (min at +20, max = min + range, sz_float.c:2849);
                                        szhip_stats.vmin / vmax report it.  For callers whose bound does not depend on the range
                                        (ABS).  0: `meta` is used as given. */
} szhip_params;
#define SZHIP_RANGE_FROM_DATA 1u

/* per-call measurements (all times in milliseconds, device events on the ctx stream) */
typedef struct szhip_stats {
    double ms_total;        /* whole call, device side incl. host glue between kernels */
    double ms_prequant;     /* fit + sampling + selection kernels */
    double ms_quant;        /* the predict+quantise (or reconstruct) wavefront kernel alone */
    double ms_entropy;      /* histogram/permute/encode (or decode/permute) kernels */
    double ms_host;         /* host glue: tree build, coefficient chain, header */
    uint64_t n_elements, n_blocks, n_reg_blocks, n_unpred;
    unsigned intervals; int use_mean;
    uint64_t out_bytes;
    uint64_t quant_kernel_launches; /* launches of the wavefront kernel (1 per call) */
    double vmin, vmax;      /* the array's range when SZHIP_RANGE_FROM_DATA was set (else 0) */
    int chain_overlapped;   /* 1: the regression-coefficient chain ran next to the wavefront kernel */
    int quant_kernel;       /* which mapping of the wavefront kernel ran: 0 = k_pencil (8x8 pencils), 2 = k_beam (szh_beam.h) */
    int packing;            /* (round 6) 1: the Huffman packing read the sweep's natural-order codes segment by segment (szh_segenc.h); 0: block-ordered copy first */
} szhip_stats;

int  szhip_create(szhip_ctx **ctx, int device);
/* ---- several arrays in flight on one GPU (additive; the reference's API is one blocking call per array, sz/src/sz.c:294-391).
 * The predict+quantise sweep is bound by its dependency chain, not by the chip: the passes in front of it (fit, sampling) and
 * behind it (block ordering, histogram, Huffman packing) of one array fit beside the sweep of another.  A pool owns `lanes`
 * contexts and as many host threads; szhip_pool_submit queues one szhip_compress call (same arguments, which must stay valid
 * until the wait returns) and returns a ticket at once; szhip_pool_wait blocks until that call has finished and hands out what
 * szhip_compress would have returned.  Streams are byte for byte those of szhip_compress.  HDF5 chunks (one SZ call per chunk,
 * hdf5-filter/H5Z-SZ/src/H5Z_SZ.c:542-828) and time steps are the natural users. */
typedef struct szhip_pool szhip_pool;
int  szhip_pool_create(szhip_pool **pool, int device, int lanes);
void szhip_pool_destroy(szhip_pool *pool);
int  szhip_pool_submit(szhip_pool *pool, int dtype, const void *data, int data_on_device, size_t r0, size_t r1, size_t r2, double eb,
                       const szhip_params *params, const unsigned char *meta, size_t meta_len, int out_on_device,
                       unsigned char *out_buf, size_t out_cap, int *ticket);      /* out_on_device 2: into out_buf (capacity out_cap) */
int  szhip_pool_wait(szhip_pool *pool, int ticket, unsigned char **out, size_t *out_size, szhip_stats *stats);   /* the call's return code */

void szhip_destroy(szhip_ctx *ctx);
const char *szhip_last_error(szhip_ctx *ctx);

/* copy a host array into the context's device input buffer once, so that szhip_minmax and
 * szhip_compress can both run on it (data_on_device = 1) without a second PCIe transfer */
int szhip_stage_input(szhip_ctx *ctx, const void *host_data, size_t bytes, void **device_ptr);

/*
 * Device pointers and ordering (all calls below): a call runs on the context's own NON-BLOCKING streams and returns when its results are complete.
 * What it reads from device memory must be complete when the call is made: synchronise the stream that produces it first -- the null stream's
 * implicit ordering does not reach non-blocking streams, and a device-to-device hipMemcpy may return to the host before the copy has finished
 * (profiles/r06_device_input_ordering.txt: a tool that decoded a stream right behind such a copy read a stream whose end had not arrived).
 */
/* min / max of n values (device or host pointer) */
int szhip_minmax(szhip_ctx *ctx, int dtype, const void *data, int data_on_device, size_t n, double *vmin, double *vmax);

/*
 * Compress a 3-D array (r0 slowest ... r2 fastest, the callee convention of sz_float.c:6527) with
 * absolute bound `eb` (already derived from the user's mode by the caller, sz_float.c:2852-2868).
 * r0 == 0: a 2-D array r1 x r2 (SZ_compress_float_2D_MDQ_nonblocked_with_blocked_regression, sz_float.c:5516 /
 * sz_double.c:4900): 16-wide blocks, three plane coefficients per block.
 * `meta`/`meta_len`: the 3 version bytes + flag byte + parameter bytes the stream starts with.
 * Output: out_on_device = 0: *out is malloc'd host memory owned by the caller (free());
 *         out_on_device = 1: *out is a device pointer owned by ctx (valid until the next call);
 *         out_on_device = 2: *out is the CALLER's device buffer of capacity *out_size; the stream is written there -- in place when the buffer is
 *                            16-byte aligned and holds the stream plus 64 bytes of slack (bytes behind the stream, up to the capacity, may be
 *                            zeroed), else through the context's buffer and a copy.
 */
int szhip_compress(szhip_ctx *ctx, int dtype, const void *data, int data_on_device,
                   size_t r0, size_t r1, size_t r2, double eb, const szhip_params *params,
                   const unsigned char *meta, size_t meta_len,
                   int out_on_device, unsigned char **out, size_t *out_size, szhip_stats *stats);

/*
 * Decompress a SZ 2.1 regression-type stream.  `stream` points at the first byte of the whole stream
 * (version bytes); `body_off` is the offset of the block-size field (4 + 28|36 + 8).
 * `out`: device pointer (out_on_device) or host pointer to r0*r1*r2 values.  r0 == 0: a 2-D array r1 x r2
 * (decompressDataSeries_float_2D_nonblocked_with_blocked_regression, szd_float.c:3141 / szd_double.c:2974).
 */
int szhip_decompress(szhip_ctx *ctx, int dtype, const unsigned char *stream, int stream_on_device, size_t stream_len,
                     size_t body_off, size_t r0, size_t r1, size_t r2,
                     void *out, int out_on_device, szhip_stats *stats);

/*
 * SZ 1.4 ("withLinearRegression = NO") path for 3-D arrays: whole-array Lorenzo prediction with lossy "exact" values and the
 * TightDataPointStorage container.  Replaces SZ_compress_float_3D_MDQ + convertTDPStoFlatBytes_float (sz/src/sz_float.c:946-1415,
 * TightDataPointStorageF.c:379-479,590-663; doubles: sz_double.c:784, TightDataPointStorageD.c) as called from
 * SZ_compress_args_float_NoCkRngeNoGzip_3D (sz_float.c:1422).  `value_range` and `median` are the range scan's results
 * (max - min and min + range/2 in the data's type, dataCompression.c:102-119); `meta` as for szhip_compress, with the flag byte of
 * TightDataPointStorageF.c:600-611.  r0 == 0: a 2-D array r1 x r2 (SZ_compress_float_2D_MDQ, sz_float.c:610-894, called from :896).
 * r0 == 0 and r1 == 0: a 1-D array of r2 values (SZ_compress_float_1D_MDQ, sz_float.c:353-540, called from :561; doubles:
 * sz_double.c:260); the caller applies the call site's own raw-store rule (sz_float.c:2908).
 */
int szhip_compress_sz14(szhip_ctx *ctx, int dtype, const void *data, int data_on_device,
                        size_t r0, size_t r1, size_t r2, double eb, double value_range, double median,
                        const szhip_params *params, const unsigned char *meta, size_t meta_len,
                        int out_on_device, unsigned char **out, size_t *out_size, szhip_stats *stats);
/* Inverse: decompressDataSeries_float_3D (sz/src/szd_float.c:600-1138; doubles: szd_double.c) on a TightDataPointStorage stream
 * (parse: TightDataPointStorageF.c:54-265).  `body_off` is the offset of the max_quant_intervals field (4 + 28|36 + 8).
 * r0 == 0: a 2-D array (decompressDataSeries_float_2D, szd_float.c:284-598); r0 == 0 and r1 == 0: a 1-D array
 * (decompressDataSeries_float_1D, szd_float.c:185-282). */
int szhip_decompress_sz14(szhip_ctx *ctx, int dtype, const unsigned char *stream, int stream_on_device, size_t stream_len,
                          size_t body_off, size_t r0, size_t r1, size_t r2, void *out, int out_on_device, szhip_stats *stats);

/*
 * The reference's OpenMP container for 3-D arrays -- replaces SZ_compress_float_3D_MDQ_openmp (sz/src/sz_omp.c:63-358; double :578-863; box
 * quantiser SZ_compress_float_3D_MDQ_RA_block, sz/src/sz_float.c:4704-5012) and decompressDataSeries_float_3D_openmp (sz_omp.c:366-566):
 * the array is cut into the box grid of `thread_num` (sz_omp.c:88-117; what an OpenMP build takes from omp_get_max_threads()), every box is
 * quantised on its own from its own reconstructed neighbours, one Huffman code book covers all boxes, every box has a byte-aligned payload.
 * A stock OpenMP build of SZ reads the stream; the box count is in it.  `meta`: the 4 + MetaDataByteLength bytes the reference writes in
 * front (version, flag byte, parameter bytes); `eb`: the absolute bound (`realPrecision`).  params->quantization_intervals == 0: the
 * interval optimiser of the SZ 1.4 path over the whole array (optimize_intervals_float_3D_opt, sz_omp.c:73-82).
 * Restrictions (SZHIP_ERR_UNSUP otherwise): the box grid must divide the array (on an uneven grid the reference's code book depends on
 * uninitialised memory), a box face (dim 0 x dim 1 of a box) has at most 1024 rows -- thread_num 4096 cuts 512^3 into 32^3 boxes.
 * `body_off` of the inverse: offset of the thread_num field (4 + MetaDataByteLength).
 * Status (round 4): on MI355X byte-identical to the oracle (oracle/szo_omp_impl.h) and md5-identical to 12 recorded outputs of the reference built
 * with -fopenmp (8 float32; 4 float64 from the same sources at -O1).  Boxes with 32 x 32 faces run the column-per-lane sweep of szh_ompcol.h
 * (k_omp_col: 0.16 ms at 512^3 f32, the whole call 0.69 ms = 777 GB/s); other shapes the first form (k_omp_box).  SZ_HIP_OMP_COL=0 /
 * SZ_HIP_OMP_LEAN=0 select the round-3 kernels for comparison.
 */
int szhip_compress_omp(szhip_ctx *ctx, int dtype, const void *data, int data_on_device, size_t r0, size_t r1, size_t r2, double eb,
                       int thread_num, const szhip_params *params, const unsigned char *meta, size_t meta_len,
                       int out_on_device, unsigned char **out, size_t *out_size, szhip_stats *stats);
int szhip_decompress_omp(szhip_ctx *ctx, int dtype, const unsigned char *stream, int stream_on_device, size_t stream_len,
                         size_t body_off, size_t r0, size_t r1, size_t r2, void *out, int out_on_device, szhip_stats *stats);

/*
 * Point-wise relative bounds (PW_REL and its AND/OR combinations) in their log-domain form: the `_pwr_pre_log` functions of
 * sz/src/sz_float_pwr.c:1791-1975 / sz_double_pwr.c:1781-1965 (dispatch sz_float.c:2888-2996) and their inverses
 * szd_float_pwr.c:1353-1422.  Three steps, the sign bytes being compressed on the host (zstd) in between:
 *   szhip_pwr_prepare       log2|x| into a device array owned by the context (*d_log), sign bytes to `signs_host` (n bytes) when any
 *                           value is negative (*positive == 0), and what the quantiser needs: the absolute bound in the log domain
 *                           (*real_precision), the log array's range and median, and the header's minLogValue.
 *                           vmin / vmax: the array's range (szhip_minmax).
 *   szhip_compress_sz14_pwr szhip_compress_sz14 on that array, writing the extra PW_REL container fields (TightDataPointStorageF.c:
 *                           408-419, 454-467): radExpo, segment_size, the compressed sign bytes and minLogValue.
 *   szhip_sz14_pwr_locate   (host only) where a PW_REL stream keeps its sign bytes, and its minLogValue;
 *   szhip_decompress_sz14_pwr  the SZ 1.4 inverse on a stream with those fields, then x = exp2(l) (0 below minLogValue), signs applied
 *                           (`signs_host`: n bytes or NULL).
 * The reference's DEFAULT form of mode PW_REL (accelerate_pw_rel_compression = 1, ratio >= 1e-5: the table-driven "MSST19" quantiser,
 * sz_float.c:1824-2725, dispatch :2838, :2890; inverse szd_float.c:1702-2700, szd_float_pwr.c:1425-1528; stream flag 0x08):
 *   szhip_msst_prepare      the scan of computeRangeSize_float_MSST19 (dataCompression.c:121-166: sign bytes from element 1 on, nearZero) and
 *                           a device copy of the array with its zeros replaced by nearZero * (1+ratio)^-3.0001 (*d_prepared, owned by the
 *                           context); the header's median (sqrt|nearZero * vmax|) and minLogValue (nearZero / (1+ratio)^2).
 *   szhip_compress_sz14_pwr with pwr->msst19 = 1, `eb` = the ratio and `data` = *d_prepared: multiplicative Lorenzo predictor on the
 *                           reconstruction, codes from the look-up table of MultiLevelCacheTableWideInterval.c:53-107 (built on the host).
 *   szhip_decompress_sz14_pwr  recognises the form by the stream's flag byte.
 * 2-D and 3-D arrays run on the wavefront kernel (szh_pencil.h, fmt 2); SZ_HIP_MSST_SWEEP=1 selects the plane-by-plane sweep of szh_msst.h
 * instead (second mapping, same bits); 1-D arrays are a one-lane chain.
 */
typedef struct szhip_pwr {
    uint64_t segment_size;             /* confparams_cpr->segment_size, recorded in the header */
    const unsigned char *signs_blob;   /* compressed sign bytes (host memory) or NULL */
    uint32_t signs_blob_size;
    double min_log_value;
    unsigned char rad_expo;            /* 0 on this path */
    unsigned char msst19;              /* 1: the table-driven form (below); `eb` of szhip_compress_sz14_pwr is then the ratio itself */
    unsigned char plus_bits;           /* confparams_cpr->plus_bits (3, conf.c:97), recorded in an MSST19 header */
    double median_stored;              /* MSST19: the header's median field, sqrt|nearZero * max| (sz_float_pwr.c:2060) */
} szhip_pwr;
int szhip_pwr_prepare(szhip_ctx *ctx, int dtype, const void *data, int data_on_device, size_t n, double vmin, double vmax, double pwr_ratio,
                      void **d_log, unsigned char *signs_host, int *positive, double *real_precision, double *value_range, double *median,
                      double *min_log_value);
int szhip_msst_prepare(szhip_ctx *ctx, int dtype, const void *data, int data_on_device, size_t n, double vmax, double pwr_ratio,
                       void **d_prepared, unsigned char *signs_host, int *positive, double *near_zero, double *median_log, double *min_log_value);
int szhip_compress_sz14_pwr(szhip_ctx *ctx, int dtype, const void *data, int data_on_device,
                            size_t r0, size_t r1, size_t r2, double eb, double value_range, double median,
                            const szhip_params *params, const unsigned char *meta, size_t meta_len, const szhip_pwr *pwr,
                            int out_on_device, unsigned char **out, size_t *out_size, szhip_stats *stats);
int szhip_sz14_pwr_locate(int dtype, const unsigned char *stream, size_t stream_len, size_t body_off, size_t *blob_off, size_t *blob_size,
                          double *min_log_value);
int szhip_decompress_sz14_pwr(szhip_ctx *ctx, int dtype, const unsigned char *stream, int stream_on_device, size_t stream_len,
                              size_t body_off, size_t r0, size_t r1, size_t r2, const unsigned char *signs_host, void *out,
                              int out_on_device, szhip_stats *stats);

/* test/diagnostic hook: copy the first `bytes` bytes of an internal device workspace of the LAST call to host.
 * which: 0 coef (T SoA[4][nblocks]) 1 blk_lor (u8) 2 codes in natural order (u16) 3 codes in block order (u16)
 *        4 code histogram (u32) 5 per-column zero counts (u32) 6 per-column unpredictable offsets (u64)
 *        7 unpredictable values (T) 8 stream buffer */
int szhip_debug_fetch(szhip_ctx *ctx, int which, void *dst, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
