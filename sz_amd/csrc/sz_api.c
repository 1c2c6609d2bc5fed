/* sz_api.c -- the reference's public C API (include/sz.h) on top of the MI355X HIP layer.  Host C.
 *
 * Mirrors, function by function (paths relative to the reference tree):
 *   SZ_Init / SZ_Init_Params / SZ_Finalize          sz/src/sz.c:60-94, :1296-1320
 *   computeDimension / computeDataLength / filterDimension   sz/src/sz.c:96-282
 *   SZ_compress_args                                 sz/src/sz.c:294-391
 *   SZ_compress_args_float / _double (dispatch)      sz/src/sz_float.c:2811-3043, sz_double.c:2531...
 *   SZ_decompress / SZ_decompress_args_float         sz/src/sz.c:486-577, szd_float.c:50-183
 *   sz_lossless_compress / _decompress (zstd)        sz/src/utility.c:156-214
 * What the reference does on the CPU inside those (range scan, prediction, quantisation, Huffman)
 * is done by szhip_* (include/szhip.h).  There is no CPU fallback: without a GPU these return NULL.
 */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <pthread.h>
#include "sz.h"
#include "szhip.h"
#include "szhost.h"

int versionNumber[4] = {SZ_VER_MAJOR, SZ_VER_MINOR, SZ_VER_BUILD, SZ_VER_REVISION};
int dataEndianType = LITTLE_ENDIAN_DATA;
int sysEndianType = LITTLE_ENDIAN_SYSTEM;
sz_params *confparams_cpr = NULL;
sz_params *confparams_dec = NULL;
sz_exedata *exe_params = NULL;

int SZ_LoadConf(const char *sz_cfgFile); /* sz_conf.c */

/* ---- one HIP context per process (the library is single-threaded by contract) ---- */
static szhip_ctx *g_ctx = NULL;
static int g_device = -1;
static __thread szhip_stats g_last_stats;

/* ---- a per-THREAD view of that state, for the one caller that is not single-threaded: sz_slab_compress_multi (sz_slab_multi.cpp) runs a host
 * thread per GPU, and every thread calls the ordinary SZ_compress_args on its slab.  The library mutates its configuration on every call
 * (SURVEY 8b "Threading": the reference does too), so each of those threads binds its OWN copy of the two parameter structs and its own
 * HIP context; everything below this point reaches `confparams_cpr` / `exe_params` / the context through the bound view when there is one.
 * Threads that bind nothing see the process-wide state exactly as before. */
static __thread sz_params *t_cpr = NULL;
static __thread sz_exedata *t_exe = NULL;
static __thread szhip_ctx *t_ctx = NULL;
static inline sz_params **cpr_slot(void) { return t_cpr ? &t_cpr : &confparams_cpr; }
static inline sz_exedata **exe_slot(void) { return t_exe ? &t_exe : &exe_params; }
void sz_slab_thread_bind(struct szhip_ctx *ctx, sz_params *cpr_copy, sz_exedata *exe_copy) { t_ctx = ctx; t_cpr = cpr_copy; t_exe = exe_copy; }
#define confparams_cpr (*cpr_slot())
#define exe_params (*exe_slot())

static szhip_ctx *get_ctx(void)
{
    if (t_ctx) return t_ctx;
    if (g_ctx) return g_ctx;
    int dev = g_device;
    if (dev < 0) { const char *e = getenv("SZ_HIP_DEVICE"); dev = e ? atoi(e) : 0; }
    if (szhip_create(&g_ctx, dev) != SZHIP_OK) {
        printf("Error: the MI355X SZ build needs a HIP device (none usable); there is no CPU fallback.\n");
        g_ctx = NULL;
    }
    return g_ctx;
}

int SZ_hip_set_device(int device)
{
    if (g_ctx) { szhip_destroy(g_ctx); g_ctx = NULL; }
    g_device = device;
    return SZ_SCES;
}

int SZ_hip_last_stats(struct szhip_stats *out)
{
    if (!out) return SZ_NSCS;
    *out = g_last_stats;
    return SZ_SCES;
}

/* ---- zstd through the system library, resolved at run time (utility.c:174-214) ---- */
typedef size_t (*zstd_compress_fn)(void *, size_t, const void *, size_t, int);
typedef size_t (*zstd_decompress_fn)(void *, size_t, const void *, size_t);
typedef unsigned long long (*zstd_fcs_fn)(const void *, size_t);
typedef unsigned (*zstd_iserr_fn)(size_t);
typedef size_t (*zstd_framesize_fn)(const void *, size_t);
static struct { int tried; void *h; zstd_compress_fn compress; zstd_decompress_fn decompress; zstd_fcs_fn fcs; zstd_iserr_fn iserr; zstd_framesize_fn frame_size; } g_zstd;

static int zstd_load(void)
{
    if (g_zstd.tried) return g_zstd.h != NULL;
    g_zstd.tried = 1;
    const char *names[] = {"libzstd.so.1", "libzstd.so", NULL};
    for (int i = 0; names[i] && !g_zstd.h; i++) g_zstd.h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!g_zstd.h) return 0;
    g_zstd.compress = (zstd_compress_fn)dlsym(g_zstd.h, "ZSTD_compress");
    g_zstd.decompress = (zstd_decompress_fn)dlsym(g_zstd.h, "ZSTD_decompress");
    g_zstd.fcs = (zstd_fcs_fn)dlsym(g_zstd.h, "ZSTD_getFrameContentSize");
    g_zstd.iserr = (zstd_iserr_fn)dlsym(g_zstd.h, "ZSTD_isError");
    if (!g_zstd.compress || !g_zstd.decompress || !g_zstd.fcs || !g_zstd.iserr) { dlclose(g_zstd.h); g_zstd.h = NULL; return 0; }
    g_zstd.frame_size = (zstd_framesize_fn)dlsym(g_zstd.h, "ZSTD_findFrameCompressedSize");      /* (optional: frames decoded side by side) */
    return 1;
}
/* sz_lossless_compress with ZSTD_COMPRESSOR (utility.c:174-195: ONE ZSTD_compress call over the whole stream, on the calling thread -- at 512^3 that call is fifty
 * times the GPU's part of SZ_compress_args under the shipped szMode, SZ_BEST_COMPRESSION).  Streams of more than a few megabytes are cut into pieces, each piece
 * compressed by a host thread of its own into a frame of its own, the frames written one behind the other: ZSTD_decompress -- the stock sz_lossless_decompress's,
 * utility.c:207, zstd 1.3.5 -- reads "some number of frames" as one stream, ZSTD_getFrameContentSize answers for the first.  (The library's own worker threads,
 * ZSTD_c_nbWorkers, would give ONE frame, but the system's libzstd.so.1 of this image is built without them: the parameter is refused.)
 * SZ_HIP_ZSTD_WORKERS sets the number of pieces (0 / 1: the reference's single call). */
typedef struct { const unsigned char *src; size_t n; int level; unsigned char *out; size_t cap, got; int err; } zstd_piece;
static void *zstd_piece_run(void *arg)
{
    zstd_piece *p = (zstd_piece *)arg;
    p->out = (unsigned char *)malloc(p->cap);
    if (!p->out) { p->err = 1; return NULL; }
    p->got = g_zstd.compress(p->out, p->cap, p->src, p->n, p->level);
    p->err = g_zstd.iserr(p->got) ? 1 : 0;
    return NULL;
}
typedef struct { unsigned char *dst; const unsigned char *src; size_t n; } copy_piece;
static void *copy_piece_run(void *arg) { copy_piece *c = (copy_piece *)arg; memcpy(c->dst, c->src, c->n); return NULL; }
/* returns a malloc'd buffer holding the wrapped stream (*out_size bytes), NULL on failure */
static unsigned char *zstd_compress_stream(const unsigned char *src, size_t n, int level, size_t *out_size)
{
    int pieces = 1;
    const char *e = getenv("SZ_HIP_ZSTD_WORKERS");
    if (e) pieces = atoi(e);
    else if (n >= ((size_t)8 << 20)) {
        long nc = sysconf(_SC_NPROCESSORS_ONLN);
        pieces = nc > 32 ? 32 : (int)nc;
        if ((size_t)pieces > n / ((size_t)1 << 20)) pieces = (int)(n / ((size_t)1 << 20));
    }
    if (pieces > 64) pieces = 64;
    if (pieces < 2 || n < 2) {
        size_t est = n < 100 ? 200 : (size_t)(n * 1.2);
        unsigned char *z = (unsigned char *)malloc(est);
        if (!z) return NULL;
        size_t zs = g_zstd.compress(z, est, src, n, level);
        if (g_zstd.iserr(zs)) { free(z); return NULL; }
        *out_size = zs;
        return z;
    }
    zstd_piece P[64]; pthread_t th[64]; int started[64];
    const size_t per = (n + (size_t)pieces - 1) / (size_t)pieces;
    int np = 0;
    for (size_t off = 0; off < n; off += per, ++np) {
        P[np].src = src + off; P[np].n = n - off < per ? n - off : per; P[np].level = level;
        P[np].cap = P[np].n + P[np].n / 128 + 1024;            /* (>= ZSTD_compressBound: n + n / 256 + a small constant) */
        P[np].out = NULL; P[np].got = 0; P[np].err = 0;
    }
    for (int i = 0; i < np; i++) started[i] = pthread_create(&th[i], NULL, zstd_piece_run, &P[i]) == 0;
    for (int i = 0; i < np; i++) { if (started[i]) pthread_join(th[i], NULL); else zstd_piece_run(&P[i]); }
    size_t total = 0; int bad = 0;
    for (int i = 0; i < np; i++) { bad |= P[i].err; total += P[i].got; }
    unsigned char *z = bad ? NULL : (unsigned char *)malloc(total ? total : 1);
    if (z) {
        copy_piece C[64]; size_t at = 0;
        for (int i = 0; i < np; i++) { C[i].dst = z + at; C[i].src = P[i].out; C[i].n = P[i].got; at += P[i].got; }
        for (int i = 0; i < np; i++) started[i] = pthread_create(&th[i], NULL, copy_piece_run, &C[i]) == 0;
        for (int i = 0; i < np; i++) { if (started[i]) pthread_join(th[i], NULL); else copy_piece_run(&C[i]); }
        *out_size = total;
    }
    for (int i = 0; i < np; i++) free(P[i].out);
    return z;
}

/* the way back: a stream of several frames (zstd_compress_stream's, or anybody's) is decoded frame by frame on threads of their own when every frame says how long
 * its content is; anything else goes to ONE ZSTD_decompress call, as in the reference (utility.c:207).  Returns what ZSTD_decompress would (ZSTD_isError-testable). */
typedef struct { unsigned char *dst; size_t cap; const unsigned char *src; size_t n; size_t got; } unzstd_piece;
static void *unzstd_piece_run(void *arg) { unzstd_piece *p = (unzstd_piece *)arg; p->got = g_zstd.decompress(p->dst, p->cap, p->src, p->n); return NULL; }
static size_t zstd_decompress_stream(unsigned char *dst, size_t cap, const unsigned char *src, size_t n)
{
    unzstd_piece P[64]; int np = 0;
    const char *e = getenv("SZ_HIP_ZSTD_WORKERS");
    if (g_zstd.frame_size && !(e && atoi(e) <= 1)) {
        size_t at = 0, out = 0;
        while (at < n && np < 64) {
            const size_t fs = g_zstd.frame_size(src + at, n - at);
            if (g_zstd.iserr(fs) || fs == 0 || fs > n - at) { np = 0; break; }
            const unsigned long long cs = g_zstd.fcs(src + at, fs);
            if (cs == (unsigned long long)-1 || cs == (unsigned long long)-2 || cs > cap - out) { np = 0; break; }
            P[np].dst = dst + out; P[np].cap = (size_t)cs; P[np].src = src + at; P[np].n = fs; P[np].got = 0; ++np;
            at += fs; out += (size_t)cs;
        }
        if (at != n) np = 0;
    }
    if (np < 2) return g_zstd.decompress(dst, cap, src, n);
    pthread_t th[64]; int started[64];
    for (int i = 0; i < np; i++) started[i] = pthread_create(&th[i], NULL, unzstd_piece_run, &P[i]) == 0;
    for (int i = 0; i < np; i++) { if (started[i]) pthread_join(th[i], NULL); else unzstd_piece_run(&P[i]); }
    size_t total = 0;
    for (int i = 0; i < np; i++) { if (g_zstd.iserr(P[i].got) || P[i].got != P[i].cap) return g_zstd.iserr(P[i].got) ? P[i].got : (size_t)-1; total += P[i].got; }
    return total;
}

/* ---- zlib (GZIP_COMPRESSOR; callZlib.c:205-253 deflates with deflateInit(level), :496-527 inflates) through the system library ---- */
typedef int (*z_compress2_fn)(unsigned char *, unsigned long *, const unsigned char *, unsigned long, int);
typedef int (*z_uncompress_fn)(unsigned char *, unsigned long *, const unsigned char *, unsigned long);
typedef unsigned long (*z_bound_fn)(unsigned long);
static struct { int tried; void *h; z_compress2_fn compress2; z_uncompress_fn uncompress; z_bound_fn bound; } g_zlib;

static int zlib_load(void)
{
    if (g_zlib.tried) return g_zlib.h != NULL;
    g_zlib.tried = 1;
    const char *names[] = {"libz.so.1", "libz.so", NULL};
    for (int i = 0; names[i] && !g_zlib.h; i++) g_zlib.h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!g_zlib.h) return 0;
    g_zlib.compress2 = (z_compress2_fn)dlsym(g_zlib.h, "compress2");
    g_zlib.uncompress = (z_uncompress_fn)dlsym(g_zlib.h, "uncompress");
    g_zlib.bound = (z_bound_fn)dlsym(g_zlib.h, "compressBound");
    if (!g_zlib.compress2 || !g_zlib.uncompress || !g_zlib.bound) { dlclose(g_zlib.h); g_zlib.h = NULL; return 0; }
    return 1;
}
/* both back ends looked up once, before threads start (sz_slab_compress_multi) */
void sz_slab_preload_lossless(void) { (void)zstd_load(); (void)zlib_load(); }

/* ---- init / finalize ---- */
int SZ_Init(const char *configFilePath)
{
    if (confparams_cpr) { free(confparams_cpr); confparams_cpr = NULL; }
    if (exe_params) { free(exe_params); exe_params = NULL; }
    int r = SZ_LoadConf(configFilePath);
    if (r == SZ_NSCS) return SZ_NSCS;
    exe_params->SZ_SIZE_TYPE = sizeof(size_t);
    if (confparams_cpr->szMode == SZ_TEMPORAL_COMPRESSION) {
        printf("Error: time-step compression is outside the scope of the MI355X build.\n");
        return SZ_NSCS;
    }
    return SZ_SCES;
}

int SZ_Init_Params(sz_params *params)
{
    SZ_Init(NULL);
    if (params->losslessCompressor != GZIP_COMPRESSOR && params->losslessCompressor != ZSTD_COMPRESSOR)
        params->losslessCompressor = ZSTD_COMPRESSOR;
    if (params->max_quant_intervals > 0) params->maxRangeRadius = params->max_quant_intervals / 2;
    memcpy(confparams_cpr, params, sizeof(sz_params));
    if (params->quantization_intervals % 2 != 0) { printf("Error: quantization_intervals must be an even number!\n"); return SZ_NSCS; }
    return SZ_SCES;
}

void sz_slab_multi_release(void);
void SZ_Finalize(void)
{
    if (confparams_dec) { free(confparams_dec); confparams_dec = NULL; }
    if (confparams_cpr) { free(confparams_cpr); confparams_cpr = NULL; }
    if (exe_params) { free(exe_params); exe_params = NULL; }
    if (g_ctx) { szhip_destroy(g_ctx); g_ctx = NULL; }
    sz_slab_multi_release();               /* (what sz_slab_compress_multi kept per device: sz_slab_multi.cpp) */
}

/* ---- dimensions ---- */
int computeDimension(size_t r5, size_t r4, size_t r3, size_t r2, size_t r1)
{
    if (r1 == 0) return 0;
    if (r2 == 0) return 1;
    if (r3 == 0) return 2;
    if (r4 == 0) return 3;
    if (r5 == 0) return 4;
    return 5;
}

size_t computeDataLength(size_t r5, size_t r4, size_t r3, size_t r2, size_t r1)
{
    switch (computeDimension(r5, r4, r3, r2, r1)) {
    case 0: return 0;
    case 1: return r1;
    case 2: return r1 * r2;
    case 3: return r1 * r2 * r3;
    case 4: return r1 * r2 * r3 * r4;
    default: return r1 * r2 * r3 * r4 * r5;
    }
}

/* size-1 dimensions are dropped, from the slowest given dimension down (sz.c:162-282) */
int filterDimension(size_t r5, size_t r4, size_t r3, size_t r2, size_t r1, size_t *c)
{
    int changed = 0;
    int dim = computeDimension(r5, r4, r3, r2, r1);
    c[0] = r1; c[1] = r2; c[2] = r3; c[3] = r4; c[4] = r5;
    if (dim == 1) return r1 < 1 ? 2 : 0;
    if (dim < 2) return 0;
    size_t in[5] = {r1, r2, r3, r4, r5};
    /* the slowest dimension is zeroed when 1; every other size-1 dimension shifts the tail down */
    for (int d = dim - 1; d >= 0; d--) {
        if (in[d] != 1) continue;
        changed = 1;
        if (d == dim - 1) c[d] = 0;
        else { for (int k = d; k < 4; k++) c[k] = c[k + 1]; if (dim == 5) c[4] = 0; }
    }
    return changed;
}

/* ---- params bytes ---- */
static void fill_meta(szhost_meta *m, const sz_params *p, int data_type)
{
    memset(m, 0, sizeof(*m));
    m->data_type = data_type; m->err_mode = p->errorBoundMode;
    m->abs_bound = p->absErrBound; m->rel_ratio = p->relBoundRatio; m->psnr = p->psnr; m->pwr_ratio = p->pw_relBoundRatio;
    m->vmin = data_type == SZ_FLOAT ? p->fmin : p->dmin; m->vmax = data_type == SZ_FLOAT ? p->fmax : p->dmax;
    m->opt_quant_mode = exe_params->optQuantMode; m->data_endian = dataEndianType; m->sz_mode = p->szMode; m->gzip_mode = p->gzipMode;
    m->sample_distance = p->sampleDistance; m->pred_threshold = p->predThreshold; m->sol_id = p->sol_ID;
    m->max_quant_intervals = p->max_quant_intervals; m->quantization_intervals = p->quantization_intervals;
    m->protect_value_range = p->protectValueRange;
}

void convertSZParamsToBytes(sz_params *params, unsigned char *result)
{
    unsigned char tmp[4 + MetaDataByteLength_double];
    szhost_meta m; fill_meta(&m, params, params->dataType);
    size_t len = szhost_write_meta(&m, 0, tmp);
    memcpy(result, tmp + 4, len - 4);
}

void convertBytesToSZParams(unsigned char *bytes, sz_params *params)
{
    unsigned char flag1 = bytes[0];
    exe_params->optQuantMode = (char)((flag1 & 0x40) >> 6);
    dataEndianType = (flag1 & 0x20) >> 5;
    params->szMode = (flag1 & 0x0c) >> 2;
    switch (flag1 & 0x03) { case 0: params->gzipMode = 1; break; case 1: params->gzipMode = 0; break; case 2: params->gzipMode = 9; break; default: break; }
    params->sampleDistance = (short)((bytes[1] << 8) | bytes[2]);
    params->predThreshold = (float)(1.0 * (short)((bytes[3] << 8) | bytes[4]) / 10000.0);
    params->dataType = bytes[5] & 0x07;
    params->errorBoundMode = (bytes[5] & 0xf0) >> 4;
    switch (params->errorBoundMode) {
    case ABS: params->absErrBound = szhost_get_f32be(bytes + 6); break;
    case REL: params->relBoundRatio = szhost_get_f32be(bytes + 10); break;
    case ABS_AND_REL: case ABS_OR_REL: params->absErrBound = szhost_get_f32be(bytes + 6); params->relBoundRatio = szhost_get_f32be(bytes + 10); break;
    case PSNR: params->psnr = szhost_get_f32be(bytes + 6); break;
    case ABS_AND_PW_REL: case ABS_OR_PW_REL: params->absErrBound = szhost_get_f32be(bytes + 6); params->pw_relBoundRatio = szhost_get_f32be(bytes + 10); break;
    case REL_AND_PW_REL: case REL_OR_PW_REL: params->relBoundRatio = szhost_get_f32be(bytes + 6); params->pw_relBoundRatio = szhost_get_f32be(bytes + 10); break;
    case PW_REL: params->pw_relBoundRatio = szhost_get_f32be(bytes + 10); break;
    default: break;
    }
    params->sol_ID = (int)bytes[14];
    if (exe_params->optQuantMode == 1) { params->max_quant_intervals = szhost_get_u32be(bytes + 16); params->quantization_intervals = 0; }
    else { params->max_quant_intervals = 0; params->quantization_intervals = szhost_get_u32be(bytes + 16); }
    if (params->dataType == SZ_FLOAT) { params->fmin = szhost_get_f32be(bytes + 20); params->fmax = szhost_get_f32be(bytes + 24); }
    else if (params->dataType == SZ_DOUBLE) { params->dmin = szhost_get_f64be(bytes + 20); params->dmax = szhost_get_f64be(bytes + 28); }
}

/* valueRangeSize and confparams_cpr->{f,d}{min,max} from the scanned range: float arithmetic for float data, max = min + range (sz_float.c:2849) */
/* A freshly malloc'd output array is first touched by the copy-out threads: 131 072 page faults for 512 MiB.  Asking for transparent huge
 * pages on its 2 MiB-aligned interior cuts that to 256 (where the kernel allows madvise-d huge pages; elsewhere the call is a no-op).
 * SZ_HIP_THP=0 turns it off. */
static void hint_huge_pages(void *p, size_t bytes)
{
#ifdef MADV_HUGEPAGE
    const char *e = getenv("SZ_HIP_THP");
    if (!p || bytes < ((size_t)32 << 20) || (e && atoi(e) == 0)) return;
    const uintptr_t lo = ((uintptr_t)p + ((uintptr_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1), hi = ((uintptr_t)p + bytes) & ~(((uintptr_t)2 << 20) - 1);
    if (hi > lo) (void)madvise((void *)lo, (size_t)(hi - lo), MADV_HUGEPAGE);
#else
    (void)p; (void)bytes;
#endif
}

static void set_range(int dataType, double vmin, double vmax, double *valueRangeSize)
{
    if (dataType == SZ_FLOAT) {
        float fr = (float)vmax - (float)vmin; *valueRangeSize = fr;
        confparams_cpr->fmin = (float)vmin; confparams_cpr->fmax = (float)vmin + fr;
    } else { *valueRangeSize = vmax - vmin; confparams_cpr->dmin = vmin; confparams_cpr->dmax = vmin + *valueRangeSize; }
}

/* SZ_compress_args_float_withinRange, sz_float.c:2728: header + the first value */
static int constant_stream(int dataType, const szhost_meta *m, const void *oriData, size_t dataLength, unsigned char **newByteData, size_t *outSize)
{
    const size_t esz = dataType == SZ_FLOAT ? 4 : 8, meta_len = dataType == SZ_FLOAT ? MetaDataByteLength : MetaDataByteLength_double;
    unsigned char meta[4 + MetaDataByteLength_double];
    unsigned char same = 0x01 | 0x40;
    /* the float container records protectValueRange (TightDataPointStorageF.c:610); the double one does not (TightDataPointStorageD.c:596-608) */
    if (confparams_cpr->protectValueRange && dataType == SZ_FLOAT) same |= 0x04;
    szhost_write_meta(m, same, meta);
    size_t tot = 4 + meta_len + 8 + esz;
    unsigned char *o = (unsigned char *)malloc(tot);
    if (!o) return SZ_NSCS;
    memcpy(o, meta, 4 + meta_len);
    szhost_put_u64be(o + 4 + meta_len, dataLength);
    if (dataType == SZ_FLOAT) szhost_put_f32be(o + 4 + meta_len + 8, ((const float *)oriData)[0]);
    else szhost_put_f64be(o + 4 + meta_len + 8, ((const double *)oriData)[0]);
    *newByteData = o; *outSize = tot;
    return SZ_SCES;
}

/* the lossless stage after the SZ stream (sz_float.c:3027-3040); takes ownership of `tmp` */
static int finish_lossless(unsigned char *tmp, size_t tmpSize, unsigned char **newByteData, size_t *outSize, int status)
{
    if (confparams_cpr->szMode == SZ_BEST_SPEED) { *newByteData = tmp; *outSize = tmpSize; }
    else if (confparams_cpr->szMode == SZ_BEST_COMPRESSION || confparams_cpr->szMode == SZ_DEFAULT_COMPRESSION) {
        /* sz_lossless_compress, utility.c:174-195.  A missing back end is an error: returning the bare SZ stream would silently
         * change what szMode promises (and the reference's default is SZ_BEST_COMPRESSION, conf.c:114) */
        if (confparams_cpr->losslessCompressor == ZSTD_COMPRESSOR) {
            if (!zstd_load()) { printf("Error: szMode asks for the zstd back end but libzstd.so.1 cannot be loaded (use SZ_BEST_SPEED or GZIP_COMPRESSOR).\n"); free(tmp); return SZ_NSCS; }
            size_t zs = 0;
            unsigned char *z = zstd_compress_stream(tmp, tmpSize, confparams_cpr->gzipMode, &zs);
            if (!z) { printf("Error: ZSTD_compress failed.\n"); free(tmp); return SZ_NSCS; }
            free(tmp); *newByteData = z; *outSize = zs;
        } else if (confparams_cpr->losslessCompressor == GZIP_COMPRESSOR) {
            if (!zlib_load()) { printf("Error: szMode asks for the gzip back end but libz.so.1 cannot be loaded (use SZ_BEST_SPEED).\n"); free(tmp); return SZ_NSCS; }
            unsigned long zl = g_zlib.bound((unsigned long)tmpSize);
            unsigned char *z = (unsigned char *)malloc(zl);
            if (!z) { free(tmp); return SZ_NSCS; }
            int zr = g_zlib.compress2(z, &zl, tmp, (unsigned long)tmpSize, confparams_cpr->gzipMode);   /* zlib_compress5: deflateInit(level), one stream */
            if (zr != 0) { printf("Error: zlib compress2 failed (%d).\n", zr); free(z); free(tmp); return SZ_NSCS; }
            free(tmp); *newByteData = z; *outSize = (size_t)zl;
        } else { printf("Error: Unrecognized lossless compressor in sz_lossless_compress()\n"); free(tmp); return SZ_NSCS; }
    } else { printf("Error: Wrong setting of confparams_cpr->szMode in the compression.\n"); free(tmp); return SZ_MERR; }
    return status;
}

/* ---- compression ---- */
static unsigned char *omp_compress_at(int dataType, const void *data, int on_device, size_t r1, size_t r2, size_t r3, double realPrecision, size_t *comp_size);
static int omp_pick_threads(size_t r1, size_t r2, size_t r3);
#define SZ_HIP_OMP_MARK 0x4f      /* stream byte 19 (parameter byte 15, which convertSZParamsToBytes never writes: ByteToolkit.c:874-972) of an OpenMP container that
                                   * SZ_compress_args wrote under SZ_HIP_MODE=omp: how SZ_decompress of THIS library tells it from an SZ 2.1 stream (same flag byte) */
static int compress_fp(int dataType, int withRegression, unsigned char **newByteData, void *oriData,
                       size_t r5, size_t r4, size_t r3, size_t r2, size_t r1, size_t *outSize,
                       int errBoundMode, double absErr_Bound, double relBoundRatio, double pwRelBoundRatio)
{
    const size_t esz = dataType == SZ_FLOAT ? 4 : 8;
    const size_t meta_len = dataType == SZ_FLOAT ? MetaDataByteLength : MetaDataByteLength_double;
    confparams_cpr->dataType = dataType;
    confparams_cpr->errorBoundMode = errBoundMode;
    if (errBoundMode == PW_REL) confparams_cpr->pw_relBoundRatio = pwRelBoundRatio;
    *newByteData = NULL;
    size_t dataLength = computeDataLength(r5, r4, r3, r2, r1);
    if (dataLength <= MIN_NUM_OF_ELEMENTS) { /* SZ_skip_compress_float, sz_float.c:37 */
        *outSize = dataLength * esz;
        *newByteData = (unsigned char *)malloc(dataLength * esz != 0 ? dataLength * esz : 1);
        memcpy(*newByteData, oriData, dataLength * esz);
        return SZ_SCES;
    }
    szhip_ctx *ctx = get_ctx();
    if (!ctx) return SZ_NSCS;
    void *d_in = NULL;
    if (szhip_stage_input(ctx, oriData, dataLength * esz, &d_in) != SZHIP_OK) return SZ_NSCS;
    /* The value range (computeRangeSize_float, sz_float.c:2845).  With an absolute bound on the SZ 2.1 path nothing before the
     * quantiser depends on it -- it only goes into the header and decides the constant-data case -- so it is taken from the library's
     * own fit pass, which reads the array anyway (SZHIP_RANGE_FROM_DATA), instead of a separate pass over the input. */
    const int dim_in = computeDimension(r5, r4, r3, r2, r1);
    const char *hip_mode = getenv("SZ_HIP_MODE");
    const int fuse_range = errBoundMode == ABS && withRegression != SZ_NO_REGRESSION && dim_in >= 2 && dim_in <= 4 && !confparams_cpr->randomAccess && absErr_Bound > 0;
    double vmin = 0, vmax = 0;
    double valueRangeSize = 0;
    if (!fuse_range) {
        if (szhip_minmax(ctx, dataType == SZ_FLOAT ? SZHIP_F32 : SZHIP_F64, d_in, 1, dataLength, &vmin, &vmax) != SZHIP_OK) return SZ_NSCS;
        set_range(dataType, vmin, vmax, &valueRangeSize);
    }

    int status = SZ_SCES;
    double realPrecision = 0;
    if (confparams_cpr->errorBoundMode == PSNR) { /* conf.c:54-60 */
        confparams_cpr->errorBoundMode = ABS;
        double v1 = confparams_cpr->psnr + 10 * log10(1 - 2.0 / 3.0 * (double)confparams_cpr->predThreshold);
        realPrecision = confparams_cpr->absErrBound = valueRangeSize * pow(10, v1 / (-20));
    } else if (confparams_cpr->errorBoundMode == NORM) { /* conf.c:62-65 */
        confparams_cpr->errorBoundMode = ABS;
        realPrecision = confparams_cpr->absErrBound = sqrt(3.0 / dataLength) * confparams_cpr->normErr;
    } else { /* getRealPrecision_float/_double, dataCompression.c:288-332 */
        if (errBoundMode == ABS) realPrecision = absErr_Bound;
        else if (errBoundMode == REL) realPrecision = relBoundRatio * valueRangeSize;
        else if (errBoundMode == ABS_AND_REL || errBoundMode == ABS_OR_REL) {
            double b = relBoundRatio * valueRangeSize;
            if (dataType == SZ_FLOAT) { /* min_f / max_f narrow both operands to float */
                float fa = (float)absErr_Bound, fb = (float)b;
                realPrecision = errBoundMode == ABS_AND_REL ? (fa < fb ? fa : fb) : (fa > fb ? fa : fb);
            } else realPrecision = errBoundMode == ABS_AND_REL ? (absErr_Bound < b ? absErr_Bound : b) : (absErr_Bound > b ? absErr_Bound : b);
        } else if (errBoundMode == ABS_AND_PW_REL || errBoundMode == ABS_OR_PW_REL) realPrecision = absErr_Bound;
        else if (errBoundMode == REL_AND_PW_REL || errBoundMode == REL_OR_PW_REL) realPrecision = relBoundRatio * valueRangeSize;
        else if (errBoundMode == PW_REL) realPrecision = 0;
        else { printf("Error: error-bound-mode is incorrect!\n"); status = SZ_BERR; }
        confparams_cpr->absErrBound = realPrecision;
    }

    szhost_meta m; fill_meta(&m, confparams_cpr, dataType);
    unsigned char meta[4 + MetaDataByteLength_double];

    if (!fuse_range && valueRangeSize <= realPrecision) {
        int crc = constant_stream(dataType, &m, oriData, dataLength, newByteData, outSize);
        return crc == SZ_SCES ? status : crc;
    }

    int dim = computeDimension(r5, r4, r3, r2, r1);
    if (dim == 5) { printf("Error: doesn't support 5 dimensions for now.\n"); return SZ_DERR; }
    if (errBoundMode >= PW_REL) {
        /* Point-wise relative bounds: log2|x| through the SZ 1.4 quantiser with an absolute bound in the log domain -- the `_pwr_pre_log`
         * form (sz_float_pwr.c:1791-1975; dispatch sz_float.c:2888-2996: every mode >= PW_REL takes it, and only pwRelBoundRatio counts).
         * Mode PW_REL itself takes the table-driven MSST19 form instead when accelerate_pw_rel_compression is set (the default) and
         * the ratio is >= 1e-5 (sz_float.c:2838, :2890; szh_msst.h). */
        if (!(pwRelBoundRatio > 0)) { printf("Error: pw_relBoundRatio must be positive.\n"); return SZ_BERR; }
        const int dt = dataType == SZ_FLOAT ? SZHIP_F32 : SZHIP_F64;
        /* the table-driven form: mode PW_REL alone, the switch on (sticky off below a ratio of 1e-5, sz_float.c:2837-2838), at most 65536
         * intervals (:2890).  The other modes >= PW_REL reach the _MSST19 functions in the reference too, but without their sign scan
         * (:2838 tests `== PW_REL`), which loses the signs: they take the log-domain form here. */
        if (pwRelBoundRatio < 0.000009999) confparams_cpr->accelerate_pw_rel_compression = 0;
        const int msst19 = errBoundMode == PW_REL && confparams_cpr->accelerate_pw_rel_compression && confparams_cpr->maxRangeRadius <= 32768;
        unsigned char *signs = (unsigned char *)malloc(dataLength);
        if (!signs) return SZ_NSCS;
        void *d_log = NULL; int positive = 1; double rp = 0, lrange = 0, lmedian = 0, minlog = 0, msst_near_zero = 0;
        szhip_pwr pw; memset(&pw, 0, sizeof(pw));
        if (msst19) {
            double near_zero = 0, median_log = 0;
            const double ref_max = dataType == SZ_FLOAT ? (double)confparams_cpr->fmax : confparams_cpr->dmax;     /* `max = min + valueRangeSize` */
            int prc = szhip_msst_prepare(ctx, dt, d_in, 1, dataLength, ref_max, pwRelBoundRatio, &d_log, signs, &positive, &near_zero, &median_log, &minlog);
            if (prc != SZHIP_OK) { printf("Error: szhip_msst_prepare failed (%d): %s\n", prc, szhip_last_error(ctx)); free(signs); return SZ_NSCS; }
            pw.msst19 = 1; pw.plus_bits = (unsigned char)confparams_cpr->plus_bits; pw.median_stored = median_log;
            rp = pwRelBoundRatio; msst_near_zero = near_zero;
        } else {
            int prc = szhip_pwr_prepare(ctx, dt, d_in, 1, dataLength, vmin, vmax, pwRelBoundRatio, &d_log, signs, &positive, &rp, &lrange, &lmedian, &minlog);
            if (prc != SZHIP_OK) { printf("Error: szhip_pwr_prepare failed (%d): %s\n", prc, szhip_last_error(ctx)); free(signs); return SZ_NSCS; }
            if (!(rp > 0)) { printf("Error: pw_relBoundRatio %g leaves no room below the rounding margin of this data.\n", pwRelBoundRatio); free(signs); return SZ_BERR; }
        }
        pw.segment_size = (uint64_t)confparams_cpr->segment_size; pw.min_log_value = minlog;
        unsigned char *blob = NULL;
        if (!positive) {                                           /* sz_lossless_compress(ZSTD_COMPRESSOR, 3, signs, ...) (utility.c:174-195) */
            if (!zstd_load()) { printf("Error: PW_REL on data with negative values needs libzstd.so.1 for the sign bytes.\n"); free(signs); return SZ_NSCS; }
            size_t est = dataLength < 100 ? 200 : (size_t)(dataLength * 1.2);
            blob = (unsigned char *)malloc(est);
            if (!blob) { free(signs); return SZ_NSCS; }
            /* the 2-D/3-D MSST19 wrappers pass (losslessCompressor, gzipMode) (sz_float_pwr.c:2030, :2068) while every reader decodes the sign
             * bytes with zstd (szd_float_pwr.c:1438): zstd it is, at that level when zstd is the configured compressor */
            int level = 3;
            if (msst19 && dim >= 2 && confparams_cpr->losslessCompressor == ZSTD_COMPRESSOR) level = confparams_cpr->gzipMode;
            size_t zs = g_zstd.compress(blob, est, signs, dataLength, level);
            if (g_zstd.iserr(zs) || zs > 0xffffffffu) { printf("Error: ZSTD_compress failed on the sign bytes.\n"); free(blob); free(signs); return SZ_NSCS; }
            pw.signs_blob = blob; pw.signs_blob_size = (uint32_t)zs;
        }
        free(signs);
        unsigned char pflags = 0x40 | 0x20;                        /* TightDataPointStorageF.c:600-611: isPW_REL */
        if (msst19) pflags |= 0x08;                                /* :608-609 */
        if (confparams_cpr->protectValueRange && dataType == SZ_FLOAT) pflags |= 0x04;   /* not in the double container (TightDataPointStorageD.c:596-608) */
        szhost_write_meta(&m, pflags, meta);
        szhip_params php; memset(&php, 0, sizeof(php));
        php.sample_distance = confparams_cpr->sampleDistance; php.pred_threshold = confparams_cpr->predThreshold;
        /* the optimiser's range is maxRangeRadius; the header records confparams_cpr->max_quant_intervals (TightDataPointStorageF.c:399), which a
         * fixed quantization_intervals overwrites while leaving maxRangeRadius alone (conf.c:193-197) */
        php.max_quant_intervals = exe_params->optQuantMode == 1 ? confparams_cpr->maxRangeRadius * 2 : confparams_cpr->max_quant_intervals;
        php.quantization_intervals = exe_params->optQuantMode == 1 ? 0 : (unsigned)exe_params->intvCapacity;
        unsigned char *ptmp = NULL; size_t ptmpSize = 0;
        int crc = szhip_compress_sz14_pwr(ctx, dt, d_log, 1, dim == 4 ? r4 * r3 : r3, r2, r1, rp, lrange, lmedian, &php, meta, 4 + meta_len, &pw, 0,
                                          &ptmp, &ptmpSize, &g_last_stats);
        free(blob);
        if (crc != SZHIP_OK) { printf("Error: szhip_compress_sz14_pwr failed (%d): %s\n", crc, szhip_last_error(ctx)); return SZ_NSCS; }
        if (exe_params->optQuantMode == 1) { exe_params->intvCapacity = (int)g_last_stats.intervals; exe_params->intvRadius = exe_params->intvCapacity / 2; }
        if (ptmpSize > 3 + meta_len + exe_params->SZ_SIZE_TYPE + 1 + esz * dataLength) {     /* sz_float_pwr.c:1971 */
            size_t tot = 3 + meta_len + 8 + 1 + esz * dataLength;
            unsigned char *o = (unsigned char *)malloc(tot);
            if (!o) { printf("Error: out of memory (%zu bytes for the raw copy)\n", tot); free(ptmp); return SZ_NSCS; }
            memcpy(o, meta, 4 + meta_len);
            o[3] = 80;
            szhost_put_u64be(o + 4 + meta_len, dataLength);
            unsigned char *q = o + 4 + meta_len + 8;
            /* the reference's MSST19 wrappers store the array whose zeros they have overwritten (sz_float_pwr.c:2053-2058, :2077): the raw copy
             * holds nearZero * (1+ratio)^-3.0001 in their place */
            const float zf = msst19 ? (float)msst_near_zero * (float)pow(1 + pwRelBoundRatio, -3.0001) : 0.0f;
            const double zd = msst19 ? msst_near_zero * pow(1 + pwRelBoundRatio, -3.0001) : 0.0;
            for (size_t i = 0; i < dataLength; i++, q += esz) {
                if (dataType == SZ_FLOAT) { const float v = ((float *)oriData)[i]; szhost_put_f32be(q, v == 0 ? zf : v); }
                else { const double v = ((double *)oriData)[i]; szhost_put_f64be(q, v == 0 ? zd : v); }
            }
            free(ptmp); ptmp = o; ptmpSize = tot;
        }
        return finish_lossless(ptmp, ptmpSize, newByteData, outSize, status);
    }
    /* the SZ 1.4 path (sz_float.c:2938,2978): 2-D and 3-D in this build; a 1-D array takes its container whatever the switch says (:2885-2900) */
    const int sz14 = withRegression == SZ_NO_REGRESSION || dim == 1;
    if (!(dim >= 1 && dim <= 4) || (sz14 && dim == 4) || confparams_cpr->randomAccess) {
        printf("Error: the MI355X build covers 2-D/3-D/4-D float/double arrays with withLinearRegression=YES and 1-D/2-D/3-D arrays with "
               "withLinearRegression=NO; this call (dim=%d, withRegression=%d, randomAccess=%d) is not covered yet.\n", dim, withRegression, confparams_cpr->randomAccess);
        return SZ_NSCS;
    }
    /* opt-in (SZ_HIP_MODE=omp, round 5): 3-D arrays whose extents the box rule divides go out in the reference's OpenMP container (sz_omp.c:63-358: boxes
     * of the array quantised independently, one code book) instead of the SZ 2.1 stream -- no dependency front across the array, ~800 GB/s instead of
     * ~280 at 512^3, at a ratio of 10.8 instead of 14.4 on the S-field (every box face restarts the predictor).  The stream is what `sz_openmp -k` of an
     * OpenMP build of the reference reads and writes; the stock SZ_decompress does NOT read it (the reference has no flag for it); SZ_decompress of this
     * library does, by the mark below.  Arrays the rule does not fit, other dimensions and point-wise bounds take the ordinary path. */
    if (hip_mode && strcmp(hip_mode, "omp") == 0 && dim == 3 && errBoundMode < PW_REL && !confparams_cpr->randomAccess && realPrecision > 0 && omp_pick_threads(r3, r2, r1) > 0) {
        if (fuse_range) {            /* (the range was left to the fit pass, which this path does not run: the constant case is settled here) */
            if (szhip_minmax(ctx, dataType == SZ_FLOAT ? SZHIP_F32 : SZHIP_F64, d_in, 1, dataLength, &vmin, &vmax) != SZHIP_OK) return SZ_NSCS;
            set_range(dataType, vmin, vmax, &valueRangeSize);
            fill_meta(&m, confparams_cpr, dataType);
        }
        if (valueRangeSize > realPrecision) {
            size_t osz = 0;
            unsigned char *o = omp_compress_at(dataType, d_in, 1, r3, r2, r1, realPrecision, &osz);
            if (!o) return SZ_NSCS;
            o[19] = SZ_HIP_OMP_MARK;
            return finish_lossless(o, osz, newByteData, outSize, status);
        }
    }
    unsigned char flags = sz14 ? 0x40 : (0x80 | 0x40);        /* TightDataPointStorageF.c:600-611 / sz_float.c:7396 */
    if (confparams_cpr->protectValueRange && !(sz14 && dataType == SZ_DOUBLE)) flags |= 0x04;   /* the double TightDataPointStorage writer leaves it out (TightDataPointStorageD.c:596-608) */
    szhost_write_meta(&m, flags, meta);
    szhip_params hp; memset(&hp, 0, sizeof(hp));
    hp.flags = fuse_range ? SZHIP_RANGE_FROM_DATA : 0;
    hp.sample_distance = confparams_cpr->sampleDistance; hp.pred_threshold = confparams_cpr->predThreshold;
    hp.max_quant_intervals = exe_params->optQuantMode == 1 ? confparams_cpr->maxRangeRadius * 2 : confparams_cpr->max_quant_intervals;   /* as above */
    hp.quantization_intervals = exe_params->optQuantMode == 1 ? 0 : (unsigned)exe_params->intvCapacity;
    size_t s0 = dim == 4 ? r4 * r3 : r3;   /* 2-D: r3 == 0 tells the HIP layer so (sz_float.c:2942 passes (r2, r1)) */
    unsigned char *tmp = NULL; size_t tmpSize = 0;
    int rc;
    if (sz14) {   /* medianValue = min + valueRangeSize/2 in the data's type (dataCompression.c:117) */
        double median = dataType == SZ_FLOAT ? (double)(float)((float)vmin + (float)valueRangeSize / 2) : vmin + valueRangeSize / 2;
        rc = szhip_compress_sz14(ctx, dataType == SZ_FLOAT ? SZHIP_F32 : SZHIP_F64, d_in, 1, r3, r2, r1, realPrecision, valueRangeSize, median,   /* r3 == 0: 2-D; r3 == r2 == 0: 1-D */
                                 &hp, meta, 4 + meta_len, 0, &tmp, &tmpSize, &g_last_stats);
    } else
        rc = szhip_compress(ctx, dataType == SZ_FLOAT ? SZHIP_F32 : SZHIP_F64, d_in, 1, s0, r2, r1, realPrecision, &hp,
                            meta, 4 + meta_len, 0, &tmp, &tmpSize, &g_last_stats);
    if (rc != SZHIP_OK && rc != SZHIP_CONSTANT && fuse_range) {
        /* the fused pass failed: a constant array must still come out as the reference's constant stream -- settle that with the plain range scan */
        if (szhip_minmax(ctx, dataType == SZ_FLOAT ? SZHIP_F32 : SZHIP_F64, d_in, 1, dataLength, &vmin, &vmax) == SZHIP_OK) {
            set_range(dataType, vmin, vmax, &valueRangeSize);
            if (valueRangeSize <= realPrecision) {
                fill_meta(&m, confparams_cpr, dataType);
                int crc = constant_stream(dataType, &m, oriData, dataLength, newByteData, outSize);
                return crc == SZ_SCES ? status : crc;
            }
        }
    }
    if (rc != SZHIP_OK && rc != SZHIP_CONSTANT) { printf("Error: szhip_compress failed (%d): %s\n", rc, szhip_last_error(ctx)); return SZ_NSCS; }
    if (fuse_range) {   /* the range arrived with the stream: record it where the reference's callers look for it, and settle the constant case */
        set_range(dataType, g_last_stats.vmin, g_last_stats.vmax, &valueRangeSize);
        fill_meta(&m, confparams_cpr, dataType);
        szhost_write_meta(&m, flags, meta);
        if (rc == SZHIP_CONSTANT || valueRangeSize <= realPrecision) {   /* (the HIP layer stops after its fit pass when it sees this: SZHIP_CONSTANT) */
            free(tmp);
            int crc = constant_stream(dataType, &m, oriData, dataLength, newByteData, outSize);
            return crc == SZ_SCES ? status : crc;
        }
    }
    if (exe_params->optQuantMode == 1) { exe_params->intvCapacity = (int)g_last_stats.intervals; exe_params->intvRadius = exe_params->intvCapacity / 2; } /* updateQuantizationInfo */

    /* SZ_compress_args_float_StoreOriData, sz_float.c:526; '>=' on the SZ 2.1 path (:2975) and at the 1-D call site (:2908),
     * '>' on the 2-D/3-D SZ 1.4 path (:1469) */
    if (tmpSize + (sz14 && dim != 1 ? 0 : 1) > dataLength * esz + 3 + meta_len + exe_params->SZ_SIZE_TYPE + 1) {
        size_t tot = 3 + meta_len + 8 + 1 + esz * dataLength;
        unsigned char *o = (unsigned char *)malloc(tot);
        if (!o) { printf("Error: out of memory (%zu bytes for the raw copy)\n", tot); free(tmp); return SZ_NSCS; }
        memcpy(o, meta, 4 + meta_len);
        o[3] = 80;
        szhost_put_u64be(o + 4 + meta_len, dataLength);
        unsigned char *q = o + 4 + meta_len + 8;
        for (size_t i = 0; i < dataLength; i++, q += esz) {
            if (dataType == SZ_FLOAT) szhost_put_f32be(q, ((float *)oriData)[i]); else szhost_put_f64be(q, ((double *)oriData)[i]);
        }
        free(tmp); tmp = o; tmpSize = tot;
    }

    return finish_lossless(tmp, tmpSize, newByteData, outSize, status);
}

unsigned char *SZ_compress_args(int dataType, void *data, size_t *outSize, int errBoundMode, double absErrBound,
                                double relBoundRatio, double pwrBoundRatio, size_t r5, size_t r4, size_t r3, size_t r2, size_t r1)
{
    if (confparams_cpr == NULL) SZ_Init(NULL);
    else if (exe_params == NULL) exe_params = (sz_exedata *)calloc(1, sizeof(sz_exedata));
    if (exe_params->intvCapacity == 0) {
        exe_params->intvCapacity = confparams_cpr->maxRangeRadius * 2;
        exe_params->intvRadius = confparams_cpr->maxRangeRadius;
        exe_params->optQuantMode = 1;
    }
    if (exe_params->SZ_SIZE_TYPE == 0) exe_params->SZ_SIZE_TYPE = sizeof(size_t);
    size_t _r[5];
    filterDimension(r5, r4, r3, r2, r1, _r);
    confparams_cpr->dataType = dataType;
    if (dataType == SZ_FLOAT || dataType == SZ_DOUBLE) {
        unsigned char *newByteData = NULL;
        compress_fp(dataType, confparams_cpr->withRegression, &newByteData, data, _r[4], _r[3], _r[2], _r[1], _r[0], outSize,
                    errBoundMode, absErrBound, relBoundRatio, pwrBoundRatio);
        return newByteData;
    }
    printf("Error: the MI355X build handles SZ_FLOAT and SZ_DOUBLE; integer types are outside its scope.\n");
    return NULL;
}

int SZ_compress_args2(int dataType, void *data, unsigned char *compressed_bytes, size_t *outSize, int errBoundMode, double absErrBound,
                      double relBoundRatio, double pwrBoundRatio, size_t r5, size_t r4, size_t r3, size_t r2, size_t r1)
{
    unsigned char *bytes = SZ_compress_args(dataType, data, outSize, errBoundMode, absErrBound, relBoundRatio, pwrBoundRatio, r5, r4, r3, r2, r1);
    if (!bytes) return SZ_NSCS;
    memcpy(compressed_bytes, bytes, *outSize);
    free(bytes);
    return SZ_SCES;
}

unsigned char *SZ_compress(int dataType, void *data, size_t *outSize, size_t r5, size_t r4, size_t r3, size_t r2, size_t r1)
{
    if (confparams_cpr == NULL) SZ_Init(NULL);
    return SZ_compress_args(dataType, data, outSize, confparams_cpr->errorBoundMode, confparams_cpr->absErrBound,
                            confparams_cpr->relBoundRatio, confparams_cpr->pw_relBoundRatio, r5, r4, r3, r2, r1);
}

/* ---- decompression ---- */
static int is_zlib_format(unsigned char m1, unsigned char m2) /* callZlib.c:30-47 */
{
    if (m1 == 104) return m2 == 5 || m2 == 129 || m2 == 222;
    if (m1 == 120) return m2 == 1 || m2 == 94 || m2 == 156 || m2 == 218;
    return 0;
}

/* ---- the helpers the reference's command-line tool (example/sz.c) calls next to the API, so that it links against this library unchanged ---- */
/* utility.c:156-172: which lossless back end wrapped the stream (zstd frame -> ZSTD_COMPRESSOR, zlib header -> GZIP_COMPRESSOR), -1 for none */
int is_lossless_compressed_data(unsigned char *compressedBytes, size_t cmpSize)
{
    if (!compressedBytes || cmpSize < 2) return -1;
    if (cmpSize >= 4 && zstd_load() && g_zstd.fcs(compressedBytes, cmpSize) != (unsigned long long)-2) return ZSTD_COMPRESSOR;   /* != ZSTD_CONTENTSIZE_ERROR */
    if (is_zlib_format(compressedBytes[0], compressedBytes[1])) return GZIP_COMPRESSOR;
    return -1;
}
/* utility.c:216-234: the first 65536 bytes of the wrapped stream (its header is all `sz -p` wants), zero-filled behind what there is.  As the
 * reference does, the stream is inflated / decompressed INTO a 64 KiB buffer and no further (streaming entry points of the installed library,
 * found at run time like the rest): a crafted frame that claims terabytes costs 64 KiB and the work for 64 KiB (ADVICE, round 4). */
typedef struct { const void *src; size_t size, pos; } zstd_in_t;
typedef struct { void *dst; size_t size, pos; } zstd_out_t;
typedef struct { const unsigned char *next_in; unsigned avail_in; unsigned long total_in; unsigned char *next_out; unsigned avail_out; unsigned long total_out;
                 const char *msg; void *state; void *zalloc, *zfree, *opaque; int data_type; unsigned long adler, reserved; } zlib_stream_t;   /* z_stream of zlib 1.x, LP64 */
uint64_t sz_lossless_decompress65536bytes(int losslessCompressor, unsigned char *compressBytes, uint64_t cmpSize, unsigned char **oriData)
{
    if (!oriData) return 0;
    *oriData = (unsigned char *)calloc(65536, 1);
    if (!*oriData || !compressBytes) return 0;
    if (losslessCompressor == ZSTD_COMPRESSOR && zstd_load()) {
        void *(*create)(void) = (void *(*)(void))dlsym(g_zstd.h, "ZSTD_createDStream");
        size_t (*freeds)(void *) = (size_t (*)(void *))dlsym(g_zstd.h, "ZSTD_freeDStream");
        size_t (*step)(void *, zstd_out_t *, zstd_in_t *) = (size_t (*)(void *, zstd_out_t *, zstd_in_t *))dlsym(g_zstd.h, "ZSTD_decompressStream");
        void *ds = (create && freeds && step) ? create() : NULL;
        if (ds) {
            zstd_in_t in = {compressBytes, (size_t)cmpSize, 0};
            zstd_out_t out = {*oriData, 65536, 0};
            while (in.pos < in.size && out.pos < out.size) {
                const size_t r = step(ds, &out, &in);
                if (g_zstd.iserr(r) || r == 0) break;              /* an error, or the frame is complete */
            }
            freeds(ds);
        }
    } else if (losslessCompressor == GZIP_COMPRESSOR && zlib_load()) {
        int (*init)(zlib_stream_t *, const char *, int) = (int (*)(zlib_stream_t *, const char *, int))dlsym(g_zlib.h, "inflateInit_");
        int (*inflate_)(zlib_stream_t *, int) = (int (*)(zlib_stream_t *, int))dlsym(g_zlib.h, "inflate");
        int (*end)(zlib_stream_t *) = (int (*)(zlib_stream_t *))dlsym(g_zlib.h, "inflateEnd");
        zlib_stream_t zs; memset(&zs, 0, sizeof(zs));
        if (init && inflate_ && end && init(&zs, "1.2.11", (int)sizeof(zs)) == 0) {
            zs.next_in = compressBytes; zs.avail_in = cmpSize > 0xffffffffu ? 0xffffffffu : (unsigned)cmpSize;
            zs.next_out = *oriData; zs.avail_out = 65536;
            (void)inflate_(&zs, 0 /* Z_NO_FLUSH */);               /* stops when the 64 KiB are full or the stream ends */
            end(&zs);
        }
    } else printf("Error: Unrecognized lossless compressor\n");
    return 65536;
}
/* sz.c:768-...: print what SZ_getMetadata found */
void SZ_printMetadata(sz_metadata *metadata)
{
    if (!metadata || !metadata->conf_params) return;
    const sz_params *p = metadata->conf_params;
    printf("=================SZ Compression Meta Data=================\n");
    printf("Version:                        \t %d.%d.%d\n", metadata->versionNumber[0], metadata->versionNumber[1], metadata->versionNumber[2]);
    printf("Constant data?:                 \t %s\n", metadata->isConstant == 1 ? "YES" : "NO");
    printf("Lossless?:                      \t %s\n", metadata->isLossless == 1 ? "YES" : "NO");
    printf("Size type (size of # elements): \t %d bytes\n", metadata->sizeType);
    printf("Num of elements:                \t %zu\n", metadata->dataSeriesLength);
    printf("compressor Name: \t\t\t %s\n", p->sol_ID == SZ ? "SZ" : p->sol_ID == SZ_Transpose ? "SZ_Transpose" : "Other compressor");
    if (p->dataType == SZ_FLOAT) printf("Data type:                      \t FLOAT\nmin value of raw data:          \t %f\nmax value of raw data:          \t %f\n", p->fmin, p->fmax);
    else if (p->dataType == SZ_DOUBLE) printf("Data type:                      \t DOUBLE\nmin value of raw data:          \t %f\nmax value of raw data:          \t %f\n", p->dmin, p->dmax);
    else printf("Data type:                      \t %d (outside the MI355X build)\n", p->dataType);
    if (exe_params && exe_params->optQuantMode == 1) {
        printf("quantization_intervals:         \t 0\n");
        printf("max_quant_intervals:            \t %u\n", p->max_quant_intervals);
        printf("actual used # intervals:        \t %d\n", metadata->defactoNBBins);
    } else printf("quantization_intervals:         \t %u\n", p->quantization_intervals);
    printf("dataEndianType (prior raw data):\t %s\n", dataEndianType == BIG_ENDIAN_DATA ? "BIG_ENDIAN" : "LITTLE_ENDIAN");
    printf("sysEndianType (at compression): \t %s\n", sysEndianType == 1 ? "BIG_ENDIAN" : "LITTLE_ENDIAN");
    printf("sampleDistance:                 \t %d\n", p->sampleDistance);
    printf("predThreshold:                  \t %f\n", p->predThreshold);
    printf("szMode:                         \t %s\n", p->szMode == SZ_BEST_SPEED ? "SZ_BEST_SPEED (without Gzip)" : p->szMode == SZ_BEST_COMPRESSION ? "SZ_BEST_COMPRESSION (with Zstd or Gzip)" : "SZ_DEFAULT_COMPRESSION");
    static const char *modes[] = {"ABS", "REL", "ABS_AND_REL", "ABS_OR_REL", "PSNR", "NORM", "", "", "", "", "PW_REL", "ABS_AND_PW_REL", "ABS_OR_PW_REL", "REL_AND_PW_REL", "REL_OR_PW_REL"};
    printf("errBoundMode:                   \t %s\n", p->errorBoundMode >= 0 && p->errorBoundMode < 15 ? modes[p->errorBoundMode] : "?");
    printf("absErrBound:                    \t %g\nrelBoundRatio:                  \t %g\npw_relBoundRatio:               \t %g\n", p->absErrBound, p->relBoundRatio, p->pw_relBoundRatio);
}
/* utility.c:236-...: the inverse of the SZ_Transpose solution's axis swap (the reference's tool calls it only for sol_ID = SZ_Transpose, which this
 * build refuses at SZ_Init; kept for the link and for callers that use it on their own): element s of the input, read in the loop order below,
 * lands at the transposed index */
void *detransposeData(void *data, int dataType, size_t r5, size_t r4, size_t r3, size_t r2, size_t r1)
{
    const size_t len = computeDataLength(r5, r4, r3, r2, r1), esz = dataType == SZ_DOUBLE ? 8 : 4;
    const int dim = computeDimension(r5, r4, r3, r2, r1);
    if ((dataType != SZ_FLOAT && dataType != SZ_DOUBLE) || !data) { printf("Error: Unsupported Datatype by Transpose data\n"); return NULL; }
    unsigned char *o = (unsigned char *)malloc(len * esz != 0 ? len * esz : 1);
    const unsigned char *in = (const unsigned char *)data;
    if (!o) return NULL;
    size_t s = 0;
    if (dim <= 1 || dim > 4) memcpy(o, in, len * esz);
    else if (dim == 2) { for (size_t i = 0; i < r2; i++) for (size_t j = 0; j < r1; j++, s++) memcpy(o + (j * r2 + i) * esz, in + s * esz, esz); }
    else if (dim == 3) { const size_t B = r1 * r2; for (size_t i = 0; i < r2; i++) for (size_t j = 0; j < r1; j++) for (size_t k = 0; k < r3; k++, s++) memcpy(o + (k * B + i * r1 + j) * esz, in + s * esz, esz); }
    else { const size_t C = r2 * r1, B = r3 * C; for (size_t i = 0; i < r3; i++) for (size_t j = 0; j < r2; j++) for (size_t k = 0; k < r1; k++) for (size_t w = 0; w < r4; w++, s++) memcpy(o + (w * B + i * C + j * r1 + k) * esz, in + s * esz, esz); }
    return o;
}

static void omp_decompress(int dataType, void **data, size_t r1, size_t r2, size_t r3, unsigned char *comp_data, size_t avail);
static void *decompress_fp(int dataType, unsigned char *cmpBytes, size_t cmpSize, size_t r5, size_t r4, size_t r3, size_t r2, size_t r1)
{
    const size_t esz = dataType == SZ_FLOAT ? 4 : 8;
    const size_t meta_len = dataType == SZ_FLOAT ? MetaDataByteLength : MetaDataByteLength_double;
    size_t dataLength = computeDataLength(r5, r4, r3, r2, r1);
    unsigned char *sz = cmpBytes; size_t szlen = cmpSize; int owned = 0;

    /* the two sizes a constant-data stream can have (SZ_SIZE_TYPE 4 or 8) are never sniffed: szd_float.c:62, szd_double.c:32 */
    const size_t cbase = (dataType == SZ_FLOAT ? 8 : 12) + meta_len;
    if (cmpSize != cbase + 4 && cmpSize != cbase + 8) { /* szd_float.c:62-95 */
        int lossless = -1;
        if (cmpSize >= 4 && zstd_load() && g_zstd.fcs(cmpBytes, cmpSize) != (unsigned long long)-2) lossless = ZSTD_COMPRESSOR; /* != ZSTD_CONTENTSIZE_ERROR */
        else if (cmpSize >= 4 && cmpBytes[0] == 0x28 && cmpBytes[1] == 0xB5 && cmpBytes[2] == 0x2F && cmpBytes[3] == 0xFD && !zstd_load()) {
            printf("Error: zstd-wrapped stream but libzstd.so.1 is not available.\n"); return NULL;
        }
        if (lossless == -1 && cmpSize >= 2 && is_zlib_format(cmpBytes[0], cmpBytes[1])) lossless = GZIP_COMPRESSOR;
        confparams_dec->losslessCompressor = lossless;
        confparams_dec->szMode = lossless != -1 ? SZ_BEST_COMPRESSION : SZ_BEST_SPEED;
        if (lossless != -1) { /* sz_lossless_decompress, utility.c:197-214 */
            size_t target = dataLength * esz; if (target < 1000000) target = 1000000;   /* MIN_ZLIB_DEC_ALLOMEM_BYTES */
            target += 4 + meta_len + exe_params->SZ_SIZE_TYPE;
            if (lossless == ZSTD_COMPRESSOR) {
                unsigned long long fcs = g_zstd.fcs(cmpBytes, cmpSize);
                if (fcs != (unsigned long long)-1 && fcs > target) {
                    if (fcs > (unsigned long long)dataLength * esz * 2 + (64u << 20)) { printf("Error: implausible zstd frame size.\n"); return NULL; }
                    target = (size_t)fcs;
                }
                unsigned char *buf = (unsigned char *)malloc(target);
                if (!buf) { printf("Error: out of memory.\n"); return NULL; }
                size_t got = zstd_decompress_stream(buf, target, cmpBytes, cmpSize);
                if (g_zstd.iserr(got)) { printf("Error: ZSTD_decompress failed.\n"); free(buf); return NULL; }
                sz = buf; szlen = got; owned = 1;
            } else {
                if (!zlib_load()) { printf("Error: gzip-wrapped stream but libz.so.1 is not available.\n"); return NULL; }
                unsigned char *buf = (unsigned char *)malloc(target);
                if (!buf) { printf("Error: out of memory.\n"); return NULL; }
                unsigned long got = (unsigned long)target;
                int zr = g_zlib.uncompress(buf, &got, cmpBytes, (unsigned long)cmpSize);
                if (zr != 0) { printf("Error: zlib uncompress failed (%d).\n", zr); free(buf); return NULL; }
                sz = buf; szlen = (size_t)got; owned = 1;
            }
        }
    }
    if (dataLength <= MIN_NUM_OF_ELEMENTS) { /* raw copy written by SZ_skip_compress */
        void *o = malloc(dataLength * esz != 0 ? dataLength * esz : 1);
        if (!o) { printf("Error: out of memory\n"); if (owned) free(sz); return NULL; }
        memcpy(o, sz, dataLength * esz);
        if (owned) free(sz);
        return o;
    }
    if (szlen < 4 + meta_len + 8) { printf("Error: compressed stream too short.\n"); if (owned) free(sz); return NULL; }
    /* new_TightDataPointStorageF_fromFlatBytes, TightDataPointStorageF.c:54-126 */
    int version = sz[0] * 10000 + sz[1] * 100 + sz[2];
    if (version < 20108 && !(sz[0] == versionNumber[0] && sz[1] == versionNumber[1] && sz[2] == versionNumber[2])) {
        printf("Wrong version: \nCompressed-data version (%d.%d.%d)\n", sz[0], sz[1], sz[2]);
        printf("Current sz version: (%d.%d.%d)\n", versionNumber[0], versionNumber[1], versionNumber[2]);
        printf("Please double-check if the compressed data (or file) is correct.\n");
        exit(0);
    }
    unsigned char same = sz[3];
    exe_params->SZ_SIZE_TYPE = ((same & 0x40) >> 6) == 1 ? 8 : 4;
    confparams_dec->protectValueRange = (same & 0x04) >> 2;
    convertBytesToSZParams(sz + 4, confparams_dec);
    confparams_dec->sol_ID = sz[4 + 14];
    const size_t st = exe_params->SZ_SIZE_TYPE;
    const unsigned char *body = sz + 4 + meta_len + st;
    void *out = malloc(dataLength * esz);
    hint_huge_pages(out, dataLength * esz);
    if (!out) { printf("Error: out of memory.\n"); if (owned) free(sz); return NULL; }
    int ok = 1;
    if ((same & 0x80) && !(same & 0x31) && sz[19] == SZ_HIP_OMP_MARK && computeDimension(r5, r4, r3, r2, r1) == 3) {
        /* an OpenMP container written by SZ_compress_args of this library under SZ_HIP_MODE=omp (see compress_fp); its body starts behind
         * 4 + MetaDataByteLength bytes for both types (sz_omp.c:221, :733) */
        void *o2 = NULL;
        free(out);
        if (szlen >= 4 + (size_t)MetaDataByteLength) omp_decompress(dataType, &o2, r3, r2, r1, sz + 4 + MetaDataByteLength, szlen - 4 - (size_t)MetaDataByteLength);
        else printf("Error: truncated OpenMP-container stream.\n");
        if (owned) free(sz);
        return o2;
    }
    if (same & 0x10) { /* lossless raw copy, big-endian values (szd_float.c:106-118) */
        if (szlen < 4 + meta_len + st + dataLength * esz) ok = 0;
        else for (size_t i = 0; i < dataLength; i++) {
            if (dataType == SZ_FLOAT) ((float *)out)[i] = szhost_get_f32be(body + 4 * i); else ((double *)out)[i] = szhost_get_f64be(body + 8 * i);
        }
    } else if (same & 0x01) { /* constant */
        if (szlen < 4 + meta_len + st + esz) ok = 0;
        else for (size_t i = 0; i < dataLength; i++) {
            if (dataType == SZ_FLOAT) ((float *)out)[i] = szhost_get_f32be(body); else ((double *)out)[i] = szhost_get_f64be(body);
        }
    } else {
        int dim = computeDimension(r5, r4, r3, r2, r1);
        if ((same & 0x20) && !(same & (0x02 | 0x80)) && dim >= 1 && dim <= 4 && st == 8 && confparams_dec->sol_ID == SZ) {
            /* point-wise relative: flag 0x08 set = the table-driven form, decompressDataSeries_float_{1D,2D,3D}_pwr_pre_log_MSST19
             * (szd_float_pwr.c:1425-1528; the library tells them apart); else the
             * log-domain form: decompressDataSeries_float_{1D,2D,3D}_pwr_pre_log (szd_float_pwr.c:1353-1422; 4-D as
             * (r4*r3, r2, r1), szd_float.c:2838) */
            szhip_ctx *ctx = get_ctx();
            const int dt = dataType == SZ_FLOAT ? SZHIP_F32 : SZHIP_F64;
            size_t bo = 0, bs = 0; double thr = 0;
            unsigned char *signs = NULL;
            if (!ctx || szhip_sz14_pwr_locate(dt, sz, szlen, 4 + meta_len + st, &bo, &bs, &thr) != SZHIP_OK) { printf("Error: truncated PW_REL stream.\n"); ok = 0; }
            else if (bs > 0) {                                     /* sz_lossless_decompress(ZSTD_COMPRESSOR, ...) of the sign bytes */
                signs = (unsigned char *)malloc(dataLength ? dataLength : 1);
                if (!zstd_load()) { printf("Error: the sign bytes of this PW_REL stream need libzstd.so.1.\n"); ok = 0; }
                else if (!signs) ok = 0;
                else {
                    size_t got = g_zstd.decompress(signs, dataLength, sz + bo, bs);
                    if (g_zstd.iserr(got) || got != dataLength) {
                        /* the reference's 2-D / 3-D MSST19 wrappers code the sign bytes with the CONFIGURED back end (sz_float_pwr.c:2030, :2068) while its
                         * readers always ask zstd (szd_float_pwr.c:1438), i.e. it cannot read its own gzip-configured streams; zlib is tried here */
                        unsigned long zl = (unsigned long)dataLength;
                        if (!(zlib_load() && g_zlib.uncompress(signs, &zl, sz + bo, (unsigned long)bs) == 0 && zl == dataLength)) {
                            printf("Error: the sign bytes of this PW_REL stream do not decode to %zu bytes.\n", dataLength); ok = 0;
                        }
                    }
                }
            }
            if (ok) {
                int rc = szhip_decompress_sz14_pwr(ctx, dt, sz, 0, szlen, 4 + meta_len + st, dim == 4 ? r4 * r3 : r3, r2, r1, signs, out, 0, &g_last_stats);
                if (rc != SZHIP_OK) { printf("Error: szhip_decompress_sz14_pwr failed (%d): %s\n", rc, szhip_last_error(ctx)); ok = 0; }
            }
            free(signs);
        } else if ((same & (0x20 | 0x02)) || !(dim >= 1 && dim <= 4) || (!(same & 0x80) && dim == 4) || ((same & 0x80) && dim == 1) || st != 8 || confparams_dec->sol_ID != SZ) {
            printf("Error: the MI355X build decodes SZ 2.1 regression-type streams of 2-D/3-D/4-D arrays and SZ 1.4 streams of 1-D/2-D/3-D arrays "
                   "(float/double; point-wise-relative streams in both forms; no random-access form); this stream (flags 0x%02x, dim %d) is not covered yet.\n", same, dim);
            ok = 0;
        } else if (!(same & 0x80)) {   /* SZ 1.4 container: getSnapshotData_float_3D -> decompressDataSeries_float_3D (szd_float.c:146,600) */
            szhip_ctx *ctx = get_ctx();
            int rc = ctx ? szhip_decompress_sz14(ctx, dataType == SZ_FLOAT ? SZHIP_F32 : SZHIP_F64, sz, 0, szlen, 4 + meta_len + st,
                                                 r3, r2, r1, out, 0, &g_last_stats) : SZHIP_ERR_NODEVICE;
            if (rc != SZHIP_OK) { printf("Error: szhip_decompress_sz14 failed (%d): %s\n", rc, ctx ? szhip_last_error(ctx) : "no device"); ok = 0; }
        } else {
            szhip_ctx *ctx = get_ctx();
            size_t s0 = dim == 4 ? r4 * r3 : r3;
            int rc = ctx ? szhip_decompress(ctx, dataType == SZ_FLOAT ? SZHIP_F32 : SZHIP_F64, sz, 0, szlen, 4 + meta_len + st,
                                            s0, r2, r1, out, 0, &g_last_stats) : SZHIP_ERR_NODEVICE;
            if (rc != SZHIP_OK) { printf("Error: szhip_decompress failed (%d): %s\n", rc, ctx ? szhip_last_error(ctx) : "no device"); ok = 0; }
        }
    }
    if (ok && confparams_dec->protectValueRange) { /* szd_float.c:161-176 */
        if (dataType == SZ_FLOAT) {
            float *nd = (float *)out, mn = confparams_dec->fmin, mx = confparams_dec->fmax;
            for (size_t i = 0; i < dataLength; i++) { float v = nd[i]; if (v <= mx && v >= mn) continue; if (v < mn) nd[i] = mn; else if (v > mx) nd[i] = mx; }
        } else {
            double *nd = (double *)out, mn = confparams_dec->dmin, mx = confparams_dec->dmax;
            for (size_t i = 0; i < dataLength; i++) { double v = nd[i]; if (v <= mx && v >= mn) continue; if (v < mn) nd[i] = mn; else if (v > mx) nd[i] = mx; }
        }
    }
    if (owned) free(sz);
    if (!ok) { free(out); return NULL; }
    /* updateQuantizationInfo(tdps->intervals) (TightDataPointStorageF.c / szd_float.c): a later compression with fixed
     * quantization_intervals finds what this stream used, as after the reference's decompression */
    if (!(same & (0x10 | 0x01)) && g_last_stats.intervals) { exe_params->intvCapacity = (int)g_last_stats.intervals; exe_params->intvRadius = exe_params->intvCapacity / 2; }
    return out;
}

void *SZ_decompress(int dataType, unsigned char *bytes, size_t byteLength, size_t r5, size_t r4, size_t r3, size_t r2, size_t r1)
{
    size_t _r[5];
    filterDimension(r5, r4, r3, r2, r1, _r);
    if (confparams_dec == NULL) confparams_dec = (sz_params *)malloc(sizeof(sz_params));
    memset(confparams_dec, 0, sizeof(sz_params));
    if (exe_params == NULL) exe_params = (sz_exedata *)malloc(sizeof(sz_exedata));
    memset(exe_params, 0, sizeof(sz_exedata));
    exe_params->SZ_SIZE_TYPE = 8;
    sysEndianType = LITTLE_ENDIAN_SYSTEM;
    if (dataType == SZ_FLOAT || dataType == SZ_DOUBLE) return decompress_fp(dataType, bytes, byteLength, _r[4], _r[3], _r[2], _r[1], _r[0]);
    printf("Error: the MI355X build handles SZ_FLOAT and SZ_DOUBLE; integer types are outside its scope.\n");
    return NULL;
}

size_t SZ_decompress_args(int dataType, unsigned char *bytes, size_t byteLength, void *decompressed_array,
                          size_t r5, size_t r4, size_t r3, size_t r2, size_t r1)
{
    size_t _r[5];
    filterDimension(r5, r4, r3, r2, r1, _r);
    size_t nbEle = computeDataLength(_r[4], _r[3], _r[2], _r[1], _r[0]);
    void *data = SZ_decompress(dataType, bytes, byteLength, _r[4], _r[3], _r[2], _r[1], _r[0]);
    if (!data) return (size_t)SZ_NSCS;
    memcpy(decompressed_array, data, nbEle * (dataType == SZ_FLOAT ? 4 : 8));
    free(data);
    return nbEle;
}

sz_metadata *SZ_getMetadata(unsigned char *bytes) /* sz.c:683-760 */
{
    if (exe_params == NULL) exe_params = (sz_exedata *)calloc(1, sizeof(sz_exedata));
    sz_metadata *md = (sz_metadata *)calloc(1, sizeof(sz_metadata));
    for (int i = 0; i < 3; i++) md->versionNumber[i] = bytes[i];
    unsigned char same = bytes[3];
    md->isConstant = same & 0x01;
    md->isLossless = (same & 0x10) >> 4;
    md->sizeType = ((same & 0x40) >> 6) == 1 ? 8 : 4;
    exe_params->SZ_SIZE_TYPE = (unsigned)md->sizeType;
    md->conf_params = (sz_params *)calloc(1, sizeof(sz_params));
    convertBytesToSZParams(bytes + 4, md->conf_params);
    size_t meta_len = md->conf_params->dataType == SZ_DOUBLE ? MetaDataByteLength_double : MetaDataByteLength;
    md->dataSeriesLength = md->sizeType == 8 ? (size_t)szhost_get_u64be(bytes + 4 + meta_len) : (size_t)szhost_get_u32be(bytes + 4 + meta_len);
    if ((same & 0x80) && !md->isConstant && !md->isLossless) { /* SZ 2.1 stream: intervals follow block size and the bound */
        const unsigned char *q = bytes + 4 + meta_len + md->sizeType + 4 + (md->conf_params->dataType == SZ_DOUBLE ? 8 : 4);
        md->defactoNBBins = (int)szhost_get_u32be(q);
    }
    return md;
}

/* ---- customised entry points (sz.c:1362-1510) ---- */
static void maybe_init_with_user_params(sz_params *userPara, sz_params *current)
{
    if (userPara == NULL && current == NULL) SZ_Init(NULL);
    else if (userPara != NULL) SZ_Init_Params(userPara);
}

unsigned char *SZ_compress_customize(const char *cmprName, void *userPara, int dataType, void *data, size_t r5, size_t r4, size_t r3,
                                     size_t r2, size_t r1, size_t *outSize, int *status)
{
    unsigned char *result = NULL;
    if (strcmp(cmprName, "SZ2.0") == 0 || strcmp(cmprName, "SZ2.1") == 0 || strcmp(cmprName, "SZ") == 0) {
        maybe_init_with_user_params((sz_params *)userPara, confparams_cpr);
        result = SZ_compress(dataType, data, outSize, r5, r4, r3, r2, r1);
        *status = result ? SZ_SCES : SZ_NSCS;
    } else {
        printf("Error: compressor '%s' is outside the scope of the MI355X build (SZ / SZ2.0 / SZ2.1 only).\n", cmprName);
        *status = SZ_NSCS;
    }
    return result;
}

unsigned char *SZ_compress_customize_threadsafe(const char *cmprName, void *userPara, int dataType, void *data, size_t r5, size_t r4,
                                                size_t r3, size_t r2, size_t r1, size_t *outSize, int *status)
{
    /* the reference variant only skips SZ_Init; device state here is per process, so calls are serialised by the caller */
    unsigned char *result = NULL;
    if (strcmp(cmprName, "SZ2.0") == 0 || strcmp(cmprName, "SZ2.1") == 0 || strcmp(cmprName, "SZ") == 0) {
        sz_params *para = (sz_params *)userPara;
        if (confparams_cpr == NULL) SZ_Init(NULL);
        size_t _r[5];
        filterDimension(r5, r4, r3, r2, r1, _r);
        if (dataType == SZ_FLOAT || dataType == SZ_DOUBLE)
            compress_fp(dataType, SZ_WITH_LINEAR_REGRESSION, &result, data, _r[4], _r[3], _r[2], _r[1], _r[0], outSize,
                        para->errorBoundMode, para->absErrBound, para->relBoundRatio, para->pw_relBoundRatio);
        *status = result ? SZ_SCES : SZ_NSCS;
    } else *status = SZ_NSCS;
    return result;
}

void *SZ_decompress_customize(const char *cmprName, void *userPara, int dataType, unsigned char *bytes, size_t byteLength, size_t r5,
                              size_t r4, size_t r3, size_t r2, size_t r1, int *status)
{
    (void)userPara;
    void *result = NULL;
    if (strcmp(cmprName, "SZ2.0") == 0 || strcmp(cmprName, "SZ2.1") == 0 || strcmp(cmprName, "SZ") == 0 || strcmp(cmprName, "SZ1.4") == 0) {
        result = SZ_decompress(dataType, bytes, byteLength, r5, r4, r3, r2, r1);
        *status = result ? SZ_SCES : SZ_NSCS;
    } else *status = SZ_NSCS;
    return result;
}

void *SZ_decompress_customize_threadsafe(const char *cmprName, void *userPara, int dataType, unsigned char *bytes, size_t byteLength,
                                         size_t r5, size_t r4, size_t r3, size_t r2, size_t r1, int *status)
{
    return SZ_decompress_customize(cmprName, userPara, dataType, bytes, byteLength, r5, r4, r3, r2, r1, status);
}

/* ---- the reference's OpenMP container for 3-D arrays (sz/include/sz_omp.h:26, :29, :36, :40; sz/src/sz_omp.c:63-358, :366-566, :578-863, :871-).
 * Same names, arguments and stream as an OpenMP build of libSZ; the box count an OpenMP build takes from omp_get_max_threads() (it is
 * written into the stream) comes from SZ_hip_set_omp_threads / SZ_HIP_OMP_THREADS, by default the smallest power of two whose box grid
 * divides the array into boxes of at most 32768 points with a dim-0 x dim-1 face of at most 1024 rows (4096 boxes of 32^3 at 512^3). */
static int g_omp_threads = 0, g_omp_last_threads = 0;   /* set by the caller / picked for the last array */
void SZ_hip_set_omp_threads(int thread_num) { g_omp_threads = thread_num; }

static void omp_grid(int thread_num, size_t *nx, size_t *ny, size_t *nz)
{   /* sz_omp.c:88-117 */
    int order = 0; while ((2 << order) <= thread_num) ++order;
    const int b = order / 3;
    switch (order % 3) { case 0: *nx = (size_t)1 << b; *ny = (size_t)1 << b; break; case 1: *nx = (size_t)1 << (b + 1); *ny = (size_t)1 << b; break;
                         default: *nx = (size_t)1 << (b + 1); *ny = (size_t)1 << (b + 1); }
    *nz = (size_t)thread_num / (*nx * *ny);
}
static int omp_pick_threads(size_t r1, size_t r2, size_t r3)
{
    int t = g_omp_threads;
    if (t <= 0) { const char *e = getenv("SZ_HIP_OMP_THREADS"); if (e) t = atoi(e); }
    if (t > 0) {      /* a forced box count: only if its grid divides the array (else SZ_compress_args returned NULL instead of taking the ordinary path) */
        size_t nx, ny, nz; omp_grid(t, &nx, &ny, &nz);
        return (nx <= r1 && ny <= r2 && nz <= r3 && r1 % nx == 0 && r2 % ny == 0 && r3 % nz == 0) ? t : 0;
    }
    for (int o = 0; o <= 30; ++o) {
        size_t nx, ny, nz; omp_grid(1 << o, &nx, &ny, &nz);
        if (nx > r1 || ny > r2 || nz > r3) break;
        if (r1 % nx || r2 % ny || r3 % nz) continue;
        const size_t c0 = r1 / nx, c1 = r2 / ny, c2 = r3 / nz;
        if (c0 * c1 <= 1024 && c0 * c1 * c2 <= 32768) return 1 << o;
    }
    return 0;
}
static unsigned char *omp_compress(int dataType, const void *oriData, size_t r1, size_t r2, size_t r3, double realPrecision, size_t *comp_size)
{ return omp_compress_at(dataType, oriData, 0, r1, r2, r3, realPrecision, comp_size); }
static unsigned char *omp_compress_at(int dataType, const void *oriData, int on_device, size_t r1, size_t r2, size_t r3, double realPrecision, size_t *comp_size)
{
    if (comp_size) *comp_size = 0;
    if (!oriData || !comp_size) return NULL;
    if (confparams_cpr == NULL) SZ_Init(NULL);
    szhip_ctx *ctx = get_ctx();
    if (!ctx) return NULL;
    const int threads = omp_pick_threads(r1, r2, r3);
    g_omp_last_threads = threads;
    if (threads <= 0) { printf("Error: no power-of-two box grid divides %zu x %zu x %zu into boxes the MI355X build takes; set SZ_HIP_OMP_THREADS.\n", r1, r2, r3); return NULL; }
    /* initRandomAccessBytes (dataCompression.c:686-708): version, flag byte, parameter bytes of confparams_cpr->dataType */
    /* The body begins at byte 3 + 1 + MetaDataByteLength = 32 for BOTH types (sz_omp.c:221, :733): with confparams_cpr->dataType == SZ_DOUBLE
     * initRandomAccessBytes writes 36 parameter bytes and the body then overwrites the last eight of them; callers pass
     * comp + 4 + MetaDataByteLength to the decompressor either way (example/sz_openmp.c) */
    const int meta_type = confparams_cpr->dataType == SZ_DOUBLE ? SZ_DOUBLE : SZ_FLOAT;
    const size_t meta_len = MetaDataByteLength;
    szhost_meta m; fill_meta(&m, confparams_cpr, meta_type);
    unsigned char meta[4 + MetaDataByteLength_double];
    unsigned char flags = 0x80 | (exe_params->SZ_SIZE_TYPE == 8 ? 0x40 : 0);
    if (confparams_cpr->randomAccess) flags |= 0x02;
    if (confparams_cpr->protectValueRange) flags |= 0x04;
    szhost_write_meta(&m, flags, meta);
    szhip_params hp; memset(&hp, 0, sizeof(hp));
    hp.sample_distance = confparams_cpr->sampleDistance; hp.pred_threshold = confparams_cpr->predThreshold;
    hp.max_quant_intervals = exe_params->optQuantMode == 1 ? confparams_cpr->maxRangeRadius * 2 : confparams_cpr->max_quant_intervals;
    hp.quantization_intervals = exe_params->optQuantMode == 1 ? 0 : (unsigned)exe_params->intvCapacity;
    unsigned char *out = NULL; size_t n = 0;
    const int rc = szhip_compress_omp(ctx, dataType == SZ_FLOAT ? SZHIP_F32 : SZHIP_F64, oriData, on_device, r1, r2, r3, realPrecision, threads, &hp,
                                      meta, 4 + meta_len, 0, &out, &n, &g_last_stats);
    if (rc != SZHIP_OK) { printf("Error: szhip_compress_omp failed (%d): %s\n", rc, szhip_last_error(ctx)); return NULL; }
    *comp_size = n;
    return out;
}
/* `comp_data`: the stream behind its 4 + MetaDataByteLength leading bytes, as the reference's callers pass it (example/sz_openmp.c:580).
 * `avail`: the bytes there are behind that pointer -- SZ_decompress knows (a crafted frame must not send the table walk below out of the buffer:
 * every offset is checked against it before it is read); the reference-signature entry points have no length argument and pass SIZE_MAX:
 * there the extent is read off the stream's own tables, as the reference does */
static void omp_decompress(int dataType, void **data, size_t r1, size_t r2, size_t r3, unsigned char *comp_data, size_t avail)
{
    if (!data) return;
    *data = NULL;
    if (!comp_data) return;
    szhip_ctx *ctx = get_ctx();
    if (!ctx) return;
    const size_t esz = dataType == SZ_FLOAT ? 4 : 8;
    const unsigned char *q = comp_data;
    const size_t head = 4 + esz + 12;
    if (avail < head) { printf("Error: truncated OpenMP-container stream.\n"); return; }
    const unsigned thread_num = szhost_get_u32be(q);
    const size_t tree_bytes = szhost_get_u32be(q + 4 + esz + 4);
    if (thread_num == 0 || thread_num > (1u << 30)) { printf("Error: not an OpenMP-container stream.\n"); return; }
    size_t nx, ny, nz; omp_grid((int)thread_num, &nx, &ny, &nz);
    const size_t nb = nx * ny * nz;
    /* the box grid must fit the array (a box holds at least one value) and every table must lie within the stream */
    const size_t nvals = r1 * r2 * r3;
    if (nb == 0 || nb > nvals) { printf("Error: OpenMP-container stream: %zu boxes for a %zu x %zu x %zu array.\n", nb, r1, r2, r3); return; }
    size_t off = head;
    if (tree_bytes > avail - off) { printf("Error: truncated OpenMP-container stream.\n"); return; }
    off += tree_bytes;
    if (nb > (avail - off) / (4 + esz)) { printf("Error: truncated OpenMP-container stream.\n"); return; }
    q = comp_data + off;
    size_t total_un = 0;
    for (size_t b = 0; b < nb; ++b) { uint32_t c; memcpy(&c, q + b * 4, 4); total_un += c; }
    if (total_un > nvals) { printf("Error: OpenMP-container stream: more verbatim values than values.\n"); return; }
    off += nb * 4 + nb * esz;
    if (total_un > (avail - off) / esz) { printf("Error: truncated OpenMP-container stream.\n"); return; }
    off += total_un * esz;
    if (nb > (avail - off) / 8) { printf("Error: truncated OpenMP-container stream.\n"); return; }
    q = comp_data + off;
    size_t pay = 0;
    for (size_t b = 0; b < nb; ++b) { uint64_t s; memcpy(&s, q + b * 8, 8); if (s > avail || pay > avail) { pay = (size_t)-1; break; } pay += (size_t)s; }
    off += nb * 8;
    if (pay == (size_t)-1 || pay > avail - off) { printf("Error: truncated OpenMP-container stream.\n"); return; }
    const size_t total = off + pay;
    void *out = malloc(r1 * r2 * r3 * esz ? r1 * r2 * r3 * esz : 1);
    if (!out) return;
    const int rc = szhip_decompress_omp(ctx, dataType == SZ_FLOAT ? SZHIP_F32 : SZHIP_F64, comp_data, 0, total, 0, r1, r2, r3, out, 0, &g_last_stats);
    if (rc != SZHIP_OK) { printf("Error: szhip_decompress_omp failed (%d): %s\n", rc, szhip_last_error(ctx)); free(out); return; }
    *data = out;
}
unsigned char *SZ_compress_float_3D_MDQ_openmp(float *oriData, size_t r1, size_t r2, size_t r3, float realPrecision, size_t *comp_size)
{ return omp_compress(SZ_FLOAT, oriData, r1, r2, r3, (double)realPrecision, comp_size); }
unsigned char *SZ_compress_double_3D_MDQ_openmp(double *oriData, size_t r1, size_t r2, size_t r3, double realPrecision, size_t *comp_size)
{ return omp_compress(SZ_DOUBLE, oriData, r1, r2, r3, realPrecision, comp_size); }
void decompressDataSeries_float_3D_openmp(float **data, size_t r1, size_t r2, size_t r3, unsigned char *comp_data)
{ omp_decompress(SZ_FLOAT, (void **)data, r1, r2, r3, comp_data, (size_t)-1); }
void decompressDataSeries_double_3D_openmp(double **data, size_t r1, size_t r2, size_t r3, unsigned char *comp_data)
{ omp_decompress(SZ_DOUBLE, (void **)data, r1, r2, r3, comp_data, (size_t)-1); }

/* the rest of sz/include/sz_omp.h and sz.h's thread helpers, so that callers written for an OpenMP build link unchanged:
 * sz_set_num_threads (sz_omp.c:49-53) sets the box count as omp_set_num_threads does for an OpenMP build; the 1-D / 2-D entry points
 * are stubs in the reference itself (sz_omp.c:56-61, :360-364, :570-576, :866-870) and stay stubs here */
void sz_set_num_threads(int nthreads) { g_omp_threads = nthreads; }
/* the box count the next call uses when it is fixed (sz_set_num_threads / SZ_HIP_OMP_THREADS); otherwise the one picked for the last array */
int sz_get_max_threads(void)
{
    if (g_omp_threads > 0) return g_omp_threads;
    const char *e = getenv("SZ_HIP_OMP_THREADS");
    if (e && atoi(e) > 0) return atoi(e);
    return g_omp_last_threads > 0 ? g_omp_last_threads : 1;
}
int sz_get_thread_num(void) { return 0; }
double sz_wtime(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + (double)ts.tv_nsec / 1e9; }
unsigned char *SZ_compress_float_1D_MDQ_openmp(float *oriData, size_t r1, double realPrecision, size_t *comp_size)
{ (void)oriData; (void)r1; (void)realPrecision; (void)comp_size; return NULL; }
unsigned char *SZ_compress_float_2D_MDQ_openmp(float *oriData, size_t r1, size_t r2, double realPrecision, size_t *comp_size)
{ (void)oriData; (void)r1; (void)r2; (void)realPrecision; (void)comp_size; return NULL; }
unsigned char *SZ_compress_double_1D_MDQ_openmp(double *oriData, size_t r1, double realPrecision, size_t *comp_size)
{ (void)oriData; (void)r1; (void)realPrecision; (void)comp_size; return NULL; }
unsigned char *SZ_compress_double_2D_MDQ_openmp(double *oriData, size_t r1, size_t r2, double realPrecision, size_t *comp_size)
{ (void)oriData; (void)r1; (void)r2; (void)realPrecision; (void)comp_size; return NULL; }
void decompressDataSeries_float_1D_openmp(float **data, size_t r1, unsigned char *comp_data) { (void)data; (void)r1; (void)comp_data; }
void decompressDataSeries_float_2D_openmp(float **data, size_t r1, size_t r2, unsigned char *comp_data) { (void)data; (void)r1; (void)r2; (void)comp_data; }
void decompressDataSeries_double_1D_openmp(double **data, size_t r1, unsigned char *comp_data) { (void)data; (void)r1; (void)comp_data; }
void decompressDataSeries_double_2D_openmp(double **data, size_t r1, size_t r2, unsigned char *comp_data) { (void)data; (void)r1; (void)r2; (void)comp_data; }
