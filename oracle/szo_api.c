/*
 * szo_api.c -- ORACLE (test infrastructure only).
 * Restates the dispatch around the hot path:
 *   SZ_compress_args            sz/src/sz.c:294-391
 *   SZ_compress_args_float      sz/src/sz_float.c:2811-3043   (double: sz_double.c:2531...)
 *   computeRangeSize_*          sz/src/dataCompression.c:102-166
 *   getRealPrecision_*          sz/src/dataCompression.c:288-332
 *   convertSZParamsToBytes      sz/src/ByteToolkit.c:874-972
 *   initRandomAccessBytes       sz/src/dataCompression.c:686-709
 *   SZ_decompress / SZ_decompress_args_float   sz/src/sz.c:486-577, szd_float.c:50-183
 * The output is always the pre-lossless stream (what the reference returns with szMode=SZ_BEST_SPEED);
 * the zstd/zlib stage (utility.c:174) is a third-party post-pass outside the path.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "szo.h"

#define SZO_ABS 0
#define SZO_REL 1
#define SZO_ABS_AND_REL 2
#define SZO_ABS_OR_REL 3
#define SZO_PSNR 4
#define SZO_NORM 5
#define SZO_PW_REL 10
#define SZO_ABS_AND_PW_REL 11
#define SZO_ABS_OR_PW_REL 12
#define SZO_REL_AND_PW_REL 13
#define SZO_REL_OR_PW_REL 14
#define SZO_FLOAT 0
#define SZO_DOUBLE 1
#define SZO_META_F32 28
#define SZO_META_F64 36

void szo_default_params(szo_params *p)
{
    memset(p, 0, sizeof(*p));
    p->sample_distance = 100;
    p->pred_threshold = 0.99f;
    p->max_quant_intervals = 65536;
    p->quantization_intervals = 0;
    p->with_regression = 1;
    p->sz_mode = 0;   /* SZ_BEST_SPEED */
    p->gzip_mode = 1; /* Z_BEST_SPEED, as example/sz.config */
    p->sol_id = 101;
    p->psnr = 90;
    p->norm_err = 0.05;
    p->conf_rel_bound_ratio = 1E-4; /* conf.c:123 */
    p->pw_rel_bound_ratio = 1E-3;   /* conf.c:127 */
    p->segment_size = 36;           /* conf.c:128 */
    p->accelerate_pw_rel = 1;       /* conf.c:125 */
}

void szo_free_stages(szo_stages *s)
{
    if (!s) return;
    free(s->codes); free(s->indicator); free(s->unpred); free(s->reg_params);
    free(s->coeff_codes); free(s->coeff_dec); free(s->code_len);
    for (int e = 0; e < 4; e++) free(s->coeff_unpred[e]);
    memset(s, 0, sizeof(*s));
}

/* ---- small helpers shared by the type-generic bodies ---- */
typedef struct szo_grid { size_t num, early, late, split; } szo_grid;

/* SZ_COMPUTE_*_NUMBER_OF_BLOCKS + SZ_COMPUTE_BLOCKCOUNT (sz/include/sz.h:93-123) */
static szo_grid szo_make_grid(size_t count, size_t bs)
{
    szo_grid g;
    g.num = (count <= bs) ? 1 : count / bs;
    g.early = g.late = count / g.num;
    g.split = count % g.num;
    if (g.split) g.early++;
    return g;
}
static inline size_t szo_blk_start(const szo_grid *g, size_t b) { return b < g->split ? b * g->early : b * g->late + g->split; }
static inline size_t szo_blk_size(const szo_grid *g, size_t b) { return b < g->split ? g->early : g->late; }

static unsigned szo_round_up_pow2(unsigned v) /* conf.c:35 */
{
    v -= 1; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return v + 1;
}

static void szo_put_u16be(unsigned char *b, uint16_t v) { b[0] = (unsigned char)(v >> 8); b[1] = (unsigned char)v; }
static void szo_put_u32be(unsigned char *b, uint32_t v) { for (int i = 0; i < 4; i++) b[i] = (unsigned char)(v >> (24 - 8 * i)); }
static void szo_put_u64be(unsigned char *b, uint64_t v) { for (int i = 0; i < 8; i++) b[i] = (unsigned char)(v >> (56 - 8 * i)); }
static uint32_t szo_get_u32be(const unsigned char *b) { return ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3]; }
static uint64_t szo_get_u64be(const unsigned char *b) { uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | b[i]; return v; }
static void szo_put_be_f32(unsigned char *b, float v) { uint32_t u; memcpy(&u, &v, 4); szo_put_u32be(b, u); }
static void szo_put_be_f64(unsigned char *b, double v) { uint64_t u; memcpy(&u, &v, 8); szo_put_u64be(b, u); }
static float szo_get_be_f32(const unsigned char *b) { uint32_t u = szo_get_u32be(b); float v; memcpy(&v, &u, 4); return v; }
static double szo_get_be_f64(const unsigned char *b) { uint64_t u = szo_get_u64be(b); double v; memcpy(&v, &u, 8); return v; }

/* the extra container fields of a point-wise-relative stream */
typedef struct szo_pwr_extra { size_t segment_size; const unsigned char *blob; size_t blob_size; double min_log_value; int msst19, plus_bits; } szo_pwr_extra;

/* zstd for the sign bytes of the PW_REL path, loaded at run time (the image has libzstd.so.1 but no headers) */
#include <dlfcn.h>
static struct { int tried; void *h; size_t (*compress)(void *, size_t, const void *, size_t, int); size_t (*decompress)(void *, size_t, const void *, size_t);
                unsigned (*iserr)(size_t); } szo_zstd;
static int szo_zstd_load(void)
{
    if (!szo_zstd.tried) {
        szo_zstd.tried = 1;
        szo_zstd.h = dlopen("libzstd.so.1", RTLD_NOW);
        if (szo_zstd.h) {
            szo_zstd.compress = (size_t (*)(void *, size_t, const void *, size_t, int))dlsym(szo_zstd.h, "ZSTD_compress");
            szo_zstd.decompress = (size_t (*)(void *, size_t, const void *, size_t))dlsym(szo_zstd.h, "ZSTD_decompress");
            szo_zstd.iserr = (unsigned (*)(size_t))dlsym(szo_zstd.h, "ZSTD_isError");
        }
    }
    return szo_zstd.h && szo_zstd.compress && szo_zstd.decompress && szo_zstd.iserr;
}
static unsigned char *szo_zstd_compress(const unsigned char *src, size_t n, int level, size_t *out_size)
{
    if (!szo_zstd_load()) { fprintf(stderr, "szo: libzstd.so.1 not available\n"); return NULL; }
    size_t est = n < 100 ? 200 : (size_t)(n * 1.2);          /* utility.c:181-184 */
    unsigned char *o = (unsigned char *)malloc(est);
    size_t z = szo_zstd.compress(o, est, src, n, level);
    if (szo_zstd.iserr(z)) { free(o); return NULL; }
    *out_size = z;
    return o;
}
static unsigned char *szo_zstd_decompress(const unsigned char *src, size_t len, size_t n)
{
    if (!szo_zstd_load()) { fprintf(stderr, "szo: libzstd.so.1 not available\n"); return NULL; }
    unsigned char *o = (unsigned char *)malloc(n ? n : 1);
    size_t z = szo_zstd.decompress(o, n, src, len);
    if (szo_zstd.iserr(z) || z != n) { free(o); return NULL; }
    return o;
}

#define T float
#define SUF f32
#define FABS_T fabsf
#define IS_F64 0
#include "szo_sz21_impl.h"
#include "szo_sz14_impl.h"
#include "szo_omp_impl.h"
#include "szo_msst_impl.h"
#include "szo_pwr_impl.h"
#undef T
#undef SUF
#undef FABS_T
#undef IS_F64

#define T double
#define SUF f64
#define FABS_T fabs
#define IS_F64 1
#include "szo_sz21_impl.h"
#include "szo_sz14_impl.h"
#include "szo_omp_impl.h"
#include "szo_msst_impl.h"
#include "szo_pwr_impl.h"
#undef T
#undef SUF
#undef FABS_T
#undef IS_F64

/* convertSZParamsToBytes (ByteToolkit.c:874-972).  abs_bound is confparams_cpr->absErrBound at the time
 * of the call, i.e. the DERIVED bound (sz_float.c:2867). */
static void szo_params_to_bytes(const szo_params *p, int data_type, int err_mode, double abs_bound, double rel_ratio,
                                double fmin_, double fmax_, unsigned char *r, double pwr_ratio)
{
    unsigned char buf = (p->quantization_intervals == 0) ? 1 : 0; /* optQuantMode */
    buf = (unsigned char)((buf << 1) | (p->data_endian & 1));
    buf = (unsigned char)((buf << 1) | 0); /* sysEndianType little */
    buf = (unsigned char)((buf << 2) | (p->sz_mode & 3));
    int tmp = 0;
    switch (p->gzip_mode) { case 1: tmp = 0; break; case 0: tmp = 1; break; case 9: tmp = 2; break; default: tmp = 0; }
    buf = (unsigned char)((buf << 2) | tmp);
    r[0] = buf;
    szo_put_u16be(r + 1, (uint16_t)p->sample_distance);
    short t2 = (short)(p->pred_threshold * 10000);
    szo_put_u16be(r + 3, (uint16_t)t2);
    r[5] = (unsigned char)err_mode;
    r[5] = (unsigned char)((r[5] << 4) | (data_type & 0x17));
    switch (err_mode) {
    case SZO_ABS: szo_put_be_f32(r + 6, (float)abs_bound); memset(r + 10, 0, 4); break;
    case SZO_REL: memset(r + 6, 0, 4); szo_put_be_f32(r + 10, (float)rel_ratio); break;
    case SZO_ABS_AND_REL: case SZO_ABS_OR_REL:
        szo_put_be_f32(r + 6, (float)abs_bound); szo_put_be_f32(r + 10, (float)rel_ratio); break;
    case SZO_PSNR: szo_put_be_f32(r + 6, (float)p->psnr); memset(r + 9, 0, 4); break;
    case SZO_ABS_AND_PW_REL: case SZO_ABS_OR_PW_REL: szo_put_be_f32(r + 6, (float)abs_bound); szo_put_be_f32(r + 10, (float)pwr_ratio); break;
    case SZO_REL_AND_PW_REL: case SZO_REL_OR_PW_REL: szo_put_be_f32(r + 6, (float)rel_ratio); szo_put_be_f32(r + 10, (float)pwr_ratio); break;
    case SZO_PW_REL: memset(r + 6, 0, 4); szo_put_be_f32(r + 10, (float)pwr_ratio); break;
    default: break;
    }
    r[14] = (unsigned char)p->sol_id;
    if (p->quantization_intervals == 0) szo_put_u32be(r + 16, p->max_quant_intervals);
    else szo_put_u32be(r + 16, p->quantization_intervals);
    if (data_type == SZO_FLOAT) { szo_put_be_f32(r + 20, (float)fmin_); szo_put_be_f32(r + 24, (float)fmax_); }
    else { szo_put_be_f64(r + 20, fmin_); szo_put_be_f64(r + 28, fmax_); }
}

static size_t szo_data_length(size_t r5, size_t r4, size_t r3, size_t r2, size_t r1)
{
    if (r1 == 0) return 0;
    if (r2 == 0) return r1;
    if (r3 == 0) return r1 * r2;
    if (r4 == 0) return r1 * r2 * r3;
    if (r5 == 0) return r1 * r2 * r3 * r4;
    return r1 * r2 * r3 * r4 * r5;
}

/* filterDimension (sz.c:162-282): squeeze size-1 dimensions */
static void szo_filter_dims(size_t r5, size_t r4, size_t r3, size_t r2, size_t r1, size_t c[5])
{
    size_t in[5] = {r1, r2, r3, r4, r5};
    int dim = 0; while (dim < 5 && in[dim] != 0) dim++;
    size_t o[5] = {0, 0, 0, 0, 0}; int n = 0;
    if (dim <= 1) { for (int i = 0; i < 5; i++) c[i] = in[i]; return; }
    for (int i = 0; i < dim; i++) if (in[i] != 1) o[n++] = in[i];
    for (int i = 0; i < 5; i++) c[i] = o[i];
}

unsigned char *szo_compress_args(const szo_params *p, int data_type, const void *data, size_t *out_size,
                                 int err_mode, double abs_err, double rel_ratio,
                                 size_t r5, size_t r4, size_t r3, size_t r2, size_t r1, szo_stages *stages)
{
    size_t c[5];
    szo_filter_dims(r5, r4, r3, r2, r1, c);
    r1 = c[0]; r2 = c[1]; r3 = c[2]; r4 = c[3]; r5 = c[4];
    size_t n = szo_data_length(r5, r4, r3, r2, r1);
    size_t esz = data_type == SZO_FLOAT ? 4 : 8;
    size_t meta_len = data_type == SZO_FLOAT ? SZO_META_F32 : SZO_META_F64;
    if (stages) memset(stages, 0, sizeof(*stages));

    if (n <= 20) { /* SZ_skip_compress_float, sz_float.c:37 */
        unsigned char *o = (unsigned char *)malloc(n * esz);
        memcpy(o, data, n * esz); *out_size = n * esz; return o;
    }

    /* range scan: `else if`, so a new minimum is never tested against max (dataCompression.c:102-119) */
    double vmin, vmax, range, median;   /* median = min + range/2 in T (dataCompression.c:117) */
    if (data_type == SZO_FLOAT) {
        const float *d = (const float *)data; float mn = d[0], mx = d[0];
        for (size_t i = 1; i < n; i++) { float v = d[i]; if (mn > v) mn = v; else if (mx < v) mx = v; }
        float rg = mx - mn; range = rg; vmin = mn; vmax = (float)(mn + rg); /* max = min+valueRangeSize, sz_float.c:2849 */
        median = (float)(mn + rg / 2);
    } else {
        const double *d = (const double *)data; double mn = d[0], mx = d[0];
        for (size_t i = 1; i < n; i++) { double v = d[i]; if (mn > v) mn = v; else if (mx < v) mx = v; }
        range = mx - mn; vmin = mn; vmax = mn + range; median = mn + range / 2;
    }

    int eff_mode = err_mode;
    double eb;
    if (err_mode == SZO_PSNR) { /* conf.c:54-60 */
        eff_mode = SZO_ABS;
        double v1 = p->psnr + 10 * log10(1 - 2.0 / 3.0 * (double)p->pred_threshold);
        eb = range * pow(10, v1 / (-20));
    } else if (err_mode == SZO_NORM) {
        eff_mode = SZO_ABS;
        eb = sqrt(3.0 / n) * p->norm_err;
    } else if (err_mode == SZO_ABS) eb = abs_err;
    else if (err_mode == SZO_REL) eb = rel_ratio * range;
    else if (err_mode == SZO_ABS_AND_REL) {
        if (data_type == SZO_FLOAT) { float a = (float)abs_err, b = (float)(rel_ratio * range); eb = a < b ? a : b; } /* min_f */
        else { double b = rel_ratio * range; eb = abs_err < b ? abs_err : b; }
    } else if (err_mode == SZO_ABS_OR_REL) {
        if (data_type == SZO_FLOAT) { float a = (float)abs_err, b = (float)(rel_ratio * range); eb = a > b ? a : b; }
        else { double b = rel_ratio * range; eb = abs_err > b ? abs_err : b; }
    } else if (err_mode == SZO_ABS_AND_PW_REL || err_mode == SZO_ABS_OR_PW_REL) eb = abs_err;        /* getRealPrecision_float, dataCompression.c:314-325 */
    else if (err_mode == SZO_REL_AND_PW_REL || err_mode == SZO_REL_OR_PW_REL) eb = rel_ratio * range;
    else if (err_mode == SZO_PW_REL) eb = 0;
    else { fprintf(stderr, "szo: unsupported error bound mode %d\n", err_mode); return NULL; }

    unsigned char meta[4 + SZO_META_F64];
    memset(meta, 0, sizeof(meta));
    meta[0] = 2; meta[1] = 1; meta[2] = 12;
    /* confparams_cpr->pw_relBoundRatio takes the argument only in mode PW_REL (sz_float.c:2817-2820); the combinations record the configured one */
    szo_params_to_bytes(p, data_type, eff_mode, eb, p->conf_rel_bound_ratio, vmin, vmax, meta + 4, p->pw_rel_bound_ratio);

    if (range <= eb) {
        /* constant data: SZ_compress_args_float_withinRange (sz_float.c:2728) -> header + first value */
        unsigned char same = 0x01 | 0x40;
        if (p->protect_value_range && data_type == SZO_FLOAT) same |= 0x04;   /* TightDataPointStorageF.c:610; the double writer has no such line (TightDataPointStorageD.c:596-608) */
        meta[3] = same;
        unsigned char *o = (unsigned char *)malloc(4 + meta_len + 8 + esz);
        memcpy(o, meta, 4 + meta_len);
        szo_put_u64be(o + 4 + meta_len, n);
        if (data_type == SZO_FLOAT) szo_put_be_f32(o + 4 + meta_len + 8, ((const float *)data)[0]);
        else szo_put_be_f64(o + 4 + meta_len + 8, ((const double *)data)[0]);
        *out_size = 4 + meta_len + 8 + esz;
        return o;
    }

    int dim = (r2 == 0) ? 1 : (r3 == 0) ? 2 : (r4 == 0) ? 3 : (r5 == 0) ? 4 : 5;
    unsigned char *out = NULL; size_t osz = 0;
    int strict_raw_rule = 0;
    if (err_mode >= SZO_PW_REL && dim <= 4) {
        /* every mode >= PW_REL goes to the _pwr_pre_log functions with pwRelBoundRatio alone (sz_float.c:2888-2996); 4-D as (r4*r3, r2, r1) */
        size_t s0 = dim >= 3 ? (dim == 4 ? r4 * r3 : r3) : 1, s1 = dim >= 2 ? r2 : 1;
        meta[3] = 0x40 | 0x20 | (p->protect_value_range && data_type == SZO_FLOAT ? 0x04 : 0);   /* float container only, as above */
        /* the table-driven form: mode PW_REL alone, the switch on, a ratio of at least 1e-5 (sz_float.c:2837-2838, :2890) */
        const int msst19 = err_mode == SZO_PW_REL && p->accelerate_pw_rel && !(p->pw_rel_bound_ratio < 0.000009999) && p->max_quant_intervals <= 65536;
        if (msst19) {
            meta[3] |= 0x08;                                     /* TightDataPointStorageF.c:608-609 */
            if (data_type == SZO_FLOAT)
                out = szo_msst_compress_f32(p, meta, 4 + meta_len, (const float *)data, s0, s1, r1, p->pw_rel_bound_ratio, (float)vmax, (size_t)p->segment_size, &osz, stages);
            else
                out = szo_msst_compress_f64(p, meta, 4 + meta_len, (const double *)data, s0, s1, r1, p->pw_rel_bound_ratio, vmax, (size_t)p->segment_size, &osz, stages);
        } else if (data_type == SZO_FLOAT)
            out = szo_pwr_compress_f32(p, meta, 4 + meta_len, (const float *)data, s0, s1, r1, p->pw_rel_bound_ratio, (float)vmin, (float)vmax, (size_t)p->segment_size, &osz);
        else
            out = szo_pwr_compress_f64(p, meta, 4 + meta_len, (const double *)data, s0, s1, r1, p->pw_rel_bound_ratio, vmin, vmax, (size_t)p->segment_size, &osz);
        if (!out) return NULL;
        strict_raw_rule = 1;                                     /* '>' at sz_float_pwr.c:1971 */
    } else if ((dim == 3 || dim == 4) && p->with_regression) {
        size_t s = (dim == 4) ? r4 * r3 : r3; /* 4-D is treated as 3-D (r4*r3, r2, r1), sz_float.c:3010 */
        meta[3] = 0x80 | 0x40 | (p->protect_value_range ? 0x04 : 0);
        if (data_type == SZO_FLOAT)
            out = szo_sz21_compress_3d_f32(p, meta, 4 + meta_len, (const float *)data, s, r2, r1, (float)eb, &osz, stages);
        else
            out = szo_sz21_compress_3d_f64(p, meta, 4 + meta_len, (const double *)data, s, r2, r1, eb, &osz, stages);
    } else if (dim == 1 || ((dim == 3 || dim == 2) && !p->with_regression)) {
        /* 1-D, whatever the regression switch says: SZ_compress_args_float_NoCkRngeNoGzip_1D (sz_float.c:561, :2885-2900) */
        /* SZ 1.4 path: SZ_compress_args_float_NoCkRngeNoGzip_3D / _2D (sz_float.c:1422, :896), flag byte TightDataPointStorageF.c:600-611 */
        size_t s0 = dim == 3 ? r3 : 1;
        if (dim == 1) r2 = 1;
        meta[3] = 0x40 | (p->protect_value_range && data_type == SZO_FLOAT ? 0x04 : 0);          /* float container only, as above */
        if (data_type == SZO_FLOAT)
            out = szo_sz14_compress_3d_f32(p, meta, 4 + meta_len, (const float *)data, s0, r2, r1, (float)eb, (float)range, (float)median, &osz, stages, NULL);
        else
            out = szo_sz14_compress_3d_f64(p, meta, 4 + meta_len, (const double *)data, s0, r2, r1, eb, range, median, &osz, stages, NULL);
        /* strict '>' inside the 2-D/3-D callee (sz_float.c:1469, :940) and nothing after it; the 1-D call site adds a '>=' of
         * its own (sz_float.c:2908, sz_double.c:2624) */
        strict_raw_rule = dim != 1;
    } else if (dim == 2 && p->with_regression) {
        /* SZ 2.1 (2D), sz_float.c:2940-2944; pinned by the recorded 2D-* cases, see szo_sz21_impl.h */
        meta[3] = 0x80 | 0x40 | (p->protect_value_range ? 0x04 : 0);
        if (data_type == SZO_FLOAT)
            out = szo_sz21_compress_2d_f32(p, meta, 4 + meta_len, (const float *)data, r2, r1, (float)eb, &osz, stages);
        else
            out = szo_sz21_compress_2d_f64(p, meta, 4 + meta_len, (const double *)data, r2, r1, eb, &osz, stages);
    } else {
        fprintf(stderr, "szo: this dimensionality/regression setting is not restated yet (dim=%d)\n", dim);
        return NULL;
    }
    /* expansion fallback: SZ_compress_args_float_StoreOriData (sz_float.c:526) */
    if (osz + (strict_raw_rule ? 0 : 1) > n * esz + 3 + meta_len + 8 + 1) {
        size_t tot = 3 + meta_len + 8 + 1 + esz * n;
        unsigned char *o = (unsigned char *)malloc(tot);
        memcpy(o, meta, 4 + meta_len);
        o[3] = 80;
        szo_put_u64be(o + 4 + meta_len, n);
        unsigned char *q = o + 4 + meta_len + 8;
        /* the MSST19 wrappers have by now overwritten the zeros of the array they store (sz_float_pwr.c:2053-2058, :2077): the raw copy holds
         * nearZero * (1+ratio)^-3.0001 in their place (and decodes to that) */
        float zf = 0; double zd = 0;
        if (meta[3] & 0x08) {
            if (data_type == SZO_FLOAT) {
                const float *x = (const float *)data; float nz = x[0];
                for (size_t i = 1; i < n; i++) if (x[i] != 0 && fabsf(x[i]) < fabsf(nz)) nz = x[i];
                zf = nz * (float)pow(1 + p->pw_rel_bound_ratio, -3.0001);
            } else {
                const double *x = (const double *)data; double nz = x[0];
                for (size_t i = 1; i < n; i++) if (x[i] != 0 && fabs(x[i]) < fabs(nz)) nz = x[i];
                zd = nz * pow(1 + p->pw_rel_bound_ratio, -3.0001);
            }
        }
        for (size_t i = 0; i < n; i++, q += esz) {
            if (data_type == SZO_FLOAT) { float v = ((const float *)data)[i]; szo_put_be_f32(q, v == 0 ? zf : v); }
            else { double v = ((const double *)data)[i]; szo_put_be_f64(q, v == 0 ? zd : v); }
        }
        free(out); out = o; osz = tot;
    }
    *out_size = osz;
    return out;
}

void *szo_decompress(int data_type, const unsigned char *bytes, size_t byte_len,
                     size_t r5, size_t r4, size_t r3, size_t r2, size_t r1)
{
    size_t c[5];
    szo_filter_dims(r5, r4, r3, r2, r1, c);
    r1 = c[0]; r2 = c[1]; r3 = c[2]; r4 = c[3]; r5 = c[4];
    size_t n = szo_data_length(r5, r4, r3, r2, r1);
    size_t esz = data_type == SZO_FLOAT ? 4 : 8;
    size_t meta_len = data_type == SZO_FLOAT ? SZO_META_F32 : SZO_META_F64;
    void *out = malloc((n ? n : 1) * esz);
    if (n <= 20) { memcpy(out, bytes, n * esz); return out; } /* mirror of SZ_skip (callers size-sniff) */
    if (byte_len < 4 + meta_len + 8) { free(out); return NULL; }
    unsigned char same = bytes[3];
    const unsigned char *body = bytes + 4 + meta_len + 8;
    int dim = (r2 == 0) ? 1 : (r3 == 0) ? 2 : (r4 == 0) ? 3 : (r5 == 0) ? 4 : 5;
    if (same & 0x10) { /* lossless raw, big-endian values */
        for (size_t i = 0; i < n; i++) {
            if (data_type == SZO_FLOAT) ((float *)out)[i] = szo_get_be_f32(body + 4 * i);
            else ((double *)out)[i] = szo_get_be_f64(body + 8 * i);
        }
    } else if (same & 0x01) { /* constant */
        for (size_t i = 0; i < n; i++) {
            if (data_type == SZO_FLOAT) ((float *)out)[i] = szo_get_be_f32(body);
            else ((double *)out)[i] = szo_get_be_f64(body);
        }
    } else if ((same & 0x20) && (same & 0x08) && dim <= 4) {     /* point-wise relative, table-driven form (szd_float.c:2714, :2755, :2796, :2836) */
        size_t s0 = dim >= 3 ? (dim == 4 ? r4 * r3 : r3) : 1, s1 = dim >= 2 ? r2 : 1;
        int rc = data_type == SZO_FLOAT ? szo_msst_decompress_f32((float *)out, s0, s1, r1, body, byte_len - (size_t)(body - bytes))
                                        : szo_msst_decompress_f64((double *)out, s0, s1, r1, body, byte_len - (size_t)(body - bytes));
        if (rc) { free(out); return NULL; }
    } else if ((same & 0x20) && !(same & 0x08) && dim <= 4) {    /* point-wise relative, log-domain form (szd_float.c:2716, :2757, :2798, :2838) */
        size_t s0 = dim >= 3 ? (dim == 4 ? r4 * r3 : r3) : 1, s1 = dim >= 2 ? r2 : 1;
        int rc = data_type == SZO_FLOAT ? szo_pwr_decompress_f32((float *)out, s0, s1, r1, body, byte_len - (size_t)(body - bytes))
                                        : szo_pwr_decompress_f64((double *)out, s0, s1, r1, body, byte_len - (size_t)(body - bytes));
        if (rc) { free(out); return NULL; }
    } else if ((same & 0x80) && (dim == 3 || dim == 4)) {
        size_t s = (dim == 4) ? r4 * r3 : r3;
        if (data_type == SZO_FLOAT) szo_sz21_decompress_3d_f32((float *)out, s, r2, r1, body);
        else szo_sz21_decompress_3d_f64((double *)out, s, r2, r1, body);
    } else if (dim == 1 || (!(same & 0x80) && (dim == 3 || dim == 2))) {
        size_t s0 = dim == 3 ? r3 : 1;
        if (dim == 1) r2 = 1;
        int rc = data_type == SZO_FLOAT ? szo_sz14_decompress_3d_f32((float *)out, s0, r2, r1, body, byte_len - (size_t)(body - bytes), NULL)
                                        : szo_sz14_decompress_3d_f64((double *)out, s0, r2, r1, body, byte_len - (size_t)(body - bytes), NULL);
        if (rc) { free(out); return NULL; }
    } else if ((same & 0x80) && dim == 2) {
        if (data_type == SZO_FLOAT) szo_sz21_decompress_2d_f32((float *)out, r2, r1, body);
        else szo_sz21_decompress_2d_f64((double *)out, r2, r1, body);
    } else { free(out); return NULL; }
    /* protectValueRange clamp (szd_float.c:161-176) */
    if (same & 0x04) {
        if (data_type == SZO_FLOAT) {
            float mn = szo_get_be_f32(bytes + 4 + 20), mx = szo_get_be_f32(bytes + 4 + 24); float *d = (float *)out;
            for (size_t i = 0; i < n; i++) { if (d[i] < mn) d[i] = mn; else if (d[i] > mx) d[i] = mx; }
        } else {
            double mn = szo_get_be_f64(bytes + 4 + 20), mx = szo_get_be_f64(bytes + 4 + 28); double *d = (double *)out;
            for (size_t i = 0; i < n; i++) { if (d[i] < mn) d[i] = mn; else if (d[i] > mx) d[i] = mx; }
        }
    }
    return out;
}

/* metrics exactly as the CLI's -a report (example/sz.c:558-620) */
#define SZO_METRICS(NAME, T)                                                                              \
    void NAME(const T *ori, const T *dec, size_t n, double *max_abs_err, double *psnr, double *nrmse)     \
    {                                                                                                     \
        T Max = ori[0], Min = ori[0], diffMax = (T)fabs((double)(T)(dec[0] - ori[0]));                    \
        double sum = 0;                                                                                   \
        for (size_t i = 0; i < n; i++) {                                                                  \
            if (Max < ori[i]) Max = ori[i];                                                               \
            if (Min > ori[i]) Min = ori[i];                                                               \
            T err = (T)fabs((double)(T)(dec[i] - ori[i])); /* `float err`, example/sz.c:583 */            \
            if (diffMax < err) diffMax = err;                                                             \
            sum += err * err; /* product in T, accumulated in double (example/sz.c:599) */                \
        }                                                                                                 \
        double mse = sum / n, range = (T)(Max - Min);                                                     \
        *max_abs_err = diffMax;                                                                           \
        *psnr = 20 * log10(range) - 10 * log10(mse);                                                      \
        *nrmse = sqrt(mse) / range;                                                                       \
    }
SZO_METRICS(szo_metrics_f32, float)
SZO_METRICS(szo_metrics_f64, double)

/* ---- the reference's OpenMP container (szo_omp_impl.h): 3-D arrays cut into thread_num independent boxes.  meta: the stream's first
 * 4 + MetaDataByteLength bytes (from the configuration state of the writing library); decompress takes the stream without them. */
unsigned char *szo_omp_compress(const szo_params *p, int data_type, const void *data, size_t r1, size_t r2, size_t r3, double eb, int thread_num,
                                const unsigned char *meta, size_t meta_len, size_t *out_size)
{
    return data_type == 0 ? szo_omp_compress_f32(p, (const float *)data, r1, r2, r3, (float)eb, thread_num, meta, meta_len, out_size)
                          : szo_omp_compress_f64(p, (const double *)data, r1, r2, r3, eb, thread_num, meta, meta_len, out_size);
}
void *szo_omp_decompress(int data_type, const unsigned char *bytes, size_t r1, size_t r2, size_t r3)
{
    return data_type == 0 ? (void *)szo_omp_decompress_f32(bytes, r1, r2, r3) : (void *)szo_omp_decompress_f64(bytes, r1, r2, r3);
}
