/* szo_msst_impl.h -- ORACLE (test infrastructure only, see szo.h): point-wise relative bounds in the reference's DEFAULT form
 * (accelerate_pw_rel_compression = 1, "MSST19"): a multiplicative Lorenzo predictor on reconstructed values, the quotient value/prediction
 * looked up in a table that maps (exponent, leading mantissa bits) to the power of (1+ratio) that covers it.  Restated per type.
 *   range scan / signs      sz/src/dataCompression.c:121-166   computeRangeSize_float_MSST19           (double :168-198)
 *   caller                  sz/src/sz_float.c:2832-2850, :2888-2996 (dispatch)                          (sz_double.c:2555-2700)
 *   zeros, median, signs    sz/src/sz_float_pwr.c:1978-2084    SZ_compress_args_float_NoCkRngeNoGzip_{1,2,3}D_pwr_pre_log_MSST19
 *   quantisers              sz/src/sz_float.c:1824-1990 (1-D), :1992-2268 (2-D), :2270-2730 (3-D)       (sz_double.c:1552, :1721, :1996)
 *   interval optimisers     sz/src/sz_float.c:4468, :4518, :4578                                        (sz_double.c:4163, :4213, :4273)
 *   table                   sz/src/MultiLevelCacheTableWideInterval.c:53-107
 *   exact values            sz/src/dataCompression.c:479-501   compressSingleFloatValue_MSST19 (no median)
 *   container               the PW_REL one plus plus_bits / max_bits (TightDataPointStorageF.c:431-435), Huffman.c:818-863
 *   decompressors           sz/src/szd_float.c:1702, :1808, :2129; szd_float_pwr.c:1425-1528             (szd_double.c, szd_double_pwr.c:1427-)
 * The three quantisers spell their arithmetic differently (the float 3-D one multiplies in double, the 2-D one in float; one place of the
 * 3-D compressor has no fabs where its decompressor has one): each is followed as written.  Two things the reference does are NOT
 * copied: it overwrites the zeros of the CALLER'S array (sz_float_pwr.c:2053-2058) -- here a copy is changed -- and the sign of element
 * 0 is never recorded (the scan starts at 1), which IS followed. */
#define FN(name) SZO_CAT(name, SUF)
#if IS_F64
#define UT uint64_t
#define NBYTES 8
#else
#define UT uint32_t
#define NBYTES 4
#endif

#ifndef SZO_MSST_COMMON
#define SZO_MSST_COMMON
typedef struct szo_msst_tab {
    double *ptab;            /* precisionTable[intervals] */
    uint16_t *cells;         /* [(top - base + 1) << bits] */
    uint64_t base, range;    /* exponent field of the first sub-table, number of sub-tables - 1 */
    int bits;
} szo_msst_tab;

static inline uint64_t szo_bits_of(double v) { uint64_t u; memcpy(&u, &v, 8); return u; }
static inline double szo_msst_rebuild(uint16_t expo, uint64_t manti, int bits)
{
    uint64_t u = (uint64_t)expo << 52;
    u += manti << (52 - bits);
    double r; memcpy(&r, &u, 8); return r;
}
/* (size_t)/(uint64_t) of a double the way gcc's x86-64 SSE2 sequence does it (cvttsd2si below 2^63; above: subtract 2^63, convert, flip bit 63) */
static inline uint64_t szo_f64_to_u64(double v)
{
    if (v != v) return 0x8000000000000000ull;
    if (v >= 9223372036854775808.0) { double w = v - 9223372036854775808.0; return (w < 9223372036854775808.0 ? (uint64_t)(int64_t)w : 0x8000000000000000ull) ^ 0x8000000000000000ull; }
    if (v <= -9223372036854775808.0) return 0x8000000000000000ull;
    return (uint64_t)(int64_t)v;
}
static void szo_msst_tab_build(szo_msst_tab *t, double precision, unsigned count, int plus_bits)
{
    const int radius = (int)count / 2;
    t->ptab = (double *)malloc(sizeof(double) * count);
    const double inv = 2.0 - pow(2, -plus_bits);
    for (unsigned i = 0; i < count; i++) t->ptab[i] = pow(1 + precision, inv * ((int)i - radius));
    const uint16_t bits = (uint16_t)((uint16_t)(-((szo_bits_of(precision) >> 52) - 1023)) + plus_bits);
    t->bits = bits;
    const double bottom = t->ptab[1] / (1 + precision), top = t->ptab[count - 1] / (1 - precision);
    const uint16_t base = (uint16_t)(szo_bits_of(bottom) >> 52), topi = (uint16_t)(szo_bits_of(top) >> 52);
    t->base = base; t->range = (uint64_t)(topi - base);
    const uint32_t per = 1u << bits;
    t->cells = (uint16_t *)calloc((size_t)(t->range + 1) << bits, 2);
    uint32_t index = 0; int flag = 0;
    for (uint16_t i = 0; i <= topi - base; i++) {
        const uint16_t expo = (uint16_t)(i + base);
        for (uint32_t j = 0; j < per; j++) {
            const double sb = szo_msst_rebuild(expo, j, bits), stp = szo_msst_rebuild(expo, (uint64_t)j + 1, bits);
            const double bb = t->ptab[index] / (1 + precision), tb = t->ptab[index] / (1 - precision);
            uint16_t *cell = t->cells + ((size_t)i << bits) + j;
            if (stp < tb && sb > bb) { *cell = (uint16_t)index; flag = 1; }
            else if (flag && index < count - 1) { index++; *cell = (uint16_t)index; }
            else *cell = 0;
        }
    }
}
static void szo_msst_tab_free(szo_msst_tab *t) { free(t->ptab); free(t->cells); }
static inline int szo_msst_state(const szo_msst_tab *t, double quotient)
{
    const uint64_t u = szo_bits_of(quotient);
    const uint64_t e = ((u & 0x7fffffffffffffffull) >> 52) - t->base;
    if (e > t->range) return 0;
    return t->cells[(size_t)(e << t->bits) + (size_t)((u & 0x000fffffffffffffull) >> (52 - t->bits))];
}
#endif

/* ---- interval optimisers: the walks of the SZ 1.4 ones, zeros skipped, a log-ratio histogram ---- */
static unsigned FN(szo_msst_intervals)(const szo_params *p, const T *d, size_t r1, size_t r2, size_t r3, double precision)
{
    const unsigned maxr = p->max_quant_intervals / 2;
    size_t *iv = (size_t *)calloc(maxr, sizeof(size_t));
    const size_t sd = (size_t)p->sample_distance, len = r1 * r2 * r3, r23 = r2 * r3;
    const T divider = (T)(log2(1 + precision) * 2);
    size_t total = 0;
    if (r1 == 1 && r2 == 1) {
        for (size_t pos = 2; pos < len; pos += sd) {
            if (d[pos] == 0) continue;
            total++;
            const T pv = d[pos - 1];
            const double pe = fabs((double)d[pos] / pv);
            uint64_t ri = szo_f64_to_u64(fabs(log2(pe) / divider + 0.5));
            if (ri >= maxr) ri = maxr - 1;
            iv[ri]++;
        }
    } else if (r1 == 1) {
        size_t oc = sd - 1, n1 = 1, pos = r3 + oc;
        while (pos < len) {
            if (d[pos] == 0) { pos += sd; continue; }
            total++;
            const T pv = d[pos - 1] + d[pos - r3] - d[pos - r3 - 1];
            const T pe = (T)fabs((double)(T)(pv / d[pos]));
            uint64_t ri = szo_f64_to_u64(fabs(log2((double)pe) / divider + 0.5));
            if (ri >= maxr) ri = maxr - 1;
            iv[ri]++;
            oc += sd;
            if (oc >= r3) {
                n1++;
                const size_t oc2 = n1 % sd;
                pos += (r3 + sd - oc) + (sd - oc2);
                oc = sd - oc2;
                if (oc == 0) oc++;
            } else pos += sd;
        }
    } else {
        size_t oc = sd - 2, n1 = 1, n2 = 1, pos = r23 + r3 + oc;
        while (pos < len) {
            if (d[pos] == 0) { pos += sd; continue; }
            total++;
            const T pv = d[pos - 1] + d[pos - r3] + d[pos - r23] - d[pos - 1 - r23] - d[pos - r3 - 1] - d[pos - r3 - r23] + d[pos - r3 - r23 - 1];
            const T pe = (T)fabs((double)(T)(d[pos] / pv));
            uint64_t ri = szo_f64_to_u64(fabs(log2((double)pe) / divider + 0.5));
            if (ri >= maxr) ri = maxr - 1;
            iv[ri]++;
            oc += sd;
            if (oc >= r3) {
                n2++;
                if (n2 == r2) { n1++; n2 = 1; pos += r3; }
                const size_t oc2 = (n1 + n2) % sd;
                pos += (r3 + sd - oc) + (sd - oc2);
                oc = sd - oc2;
                if (oc == 0) oc++;
            } else pos += sd;
        }
    }
    const size_t target = (size_t)(total * p->pred_threshold);
    size_t sum = 0, i;
    for (i = 0; i < maxr; i++) { sum += iv[i]; if (sum > target) break; }
    if (i >= maxr) i = maxr - 1;
    unsigned pow2 = szo_round_up_pow2(2 * (unsigned)(i + 1));
    const unsigned floor_ = IS_F64 ? 64 : 32;                                   /* sz_double.c:4206, :4266, :4334 against sz_float.c */
    if (pow2 < floor_) pow2 = floor_;
    free(iv);
    return pow2;
}

/* one point.  `pred` has already been narrowed to T the way the calling function does it; hit: |pred| (or pred) times the table entry */
static inline int FN(szo_msst_point)(FN(szo_exact) *E, const szo_msst_tab *tb, T x, T pred, int use_fabs, T *rc)
{
    const double q = (double)(T)(x / pred);
    const int state = szo_msst_state(tb, q);
    if (state) { *rc = (T)((use_fabs ? fabs((double)pred) : (double)pred) * tb->ptab[state]); return state; }
    *rc = FN(szo_exact_add)(E, x);
    return 0;
}

/* `data`: zeros already replaced.  Returns the stream; pw carries the sign blob / minLogValue / segment size. */
static unsigned char *FN(szo_msst_quantise)(const szo_params *p, const unsigned char *meta, size_t meta_len, const T *data, size_t r1, size_t r2,
                                            size_t r3, double precision, T median_stored, size_t *out_size, szo_stages *st, const szo_pwr_extra *pw)
{
    const size_t n = r1 * r2 * r3, r23 = r2 * r3;
    unsigned intervals = p->quantization_intervals ? p->quantization_intervals : FN(szo_msst_intervals)(p, data, r1, r2, r3, precision);
    szo_msst_tab tb; szo_msst_tab_build(&tb, precision, intervals, pw->plus_bits);
    FN(szo_exact) E; memset(&E, 0, sizeof(E));
    E.median = 0;
    {
#if IS_F64
        const int reqExpo = (int)((szo_bits_of(precision) & 0x7FF0000000000000ull) >> 52) - 1023;
        E.req_len = 12 - (short)reqExpo;
#else
        const float pf = (float)precision; uint32_t u; memcpy(&u, &pf, 4);
        const int reqExpo = (int)((u & 0x7F800000u) >> 23) - 127;
        E.req_len = 9 - (short)reqExpo;
        if (r1 == 1 && r2 != 1)             /* the float 2-D quantiser asks the DOUBLE rule (sz_float.c:2041: computeReqLength_double_MSST19) */
            E.req_len = 12 - (short)((int)((szo_bits_of(precision) & 0x7FF0000000000000ull) >> 52) - 1023);
#endif
    }
    E.req_bytes = E.req_len / 8; E.resi_bits = E.req_len % 8;
    int *type = (int *)malloc(n * sizeof(int));
#define PT(IDX, X, PRED, FABS, RC) type[IDX] = FN(szo_msst_point)(&E, &tb, X, PRED, FABS, RC)
    if (r1 == 1 && r2 == 1) {                               /* 1-D: sz_float.c:1824-1990 */
        type[0] = 0; (void)FN(szo_exact_add)(&E, data[0]);
        type[1] = 0;
        T pred = FN(szo_exact_add)(&E, data[1]);
        for (size_t i = 2; i < n; i++) {
            const double q = (double)(T)(data[i] / pred);
            const int state = szo_msst_state(&tb, q);
            if (state) { type[i] = state; pred = (T)((double)pred * tb.ptab[state]); }      /* `pred *= precisionTable[state]` */
            else { type[i] = 0; pred = FN(szo_exact_add)(&E, data[i]); }
        }
    } else if (r1 == 1) {                                   /* 2-D: sz_float.c:1992-2268, products in T */
        T *P0 = (T *)malloc(r3 * sizeof(T)), *P1 = (T *)malloc(r3 * sizeof(T));
        type[0] = 0; P1[0] = FN(szo_exact_add)(&E, data[0]);
        PT(1, data[1], P1[0], 1, &P1[1]);
        for (size_t j = 2; j < r3; j++) { const T pr = (T)((T)(P1[j - 1] * P1[j - 1]) / P1[j - 2]); PT(j, data[j], pr, 1, &P1[j]); }
        for (size_t i = 1; i < r2; i++) {
            size_t idx = i * r3;
            PT(idx, data[idx], P1[0], 1, &P0[0]);
            for (size_t j = 1; j < r3; j++) {
                idx++;
                const T pr = (T)((T)(P0[j - 1] * P1[j]) / P1[j - 1]);
                PT(idx, data[idx], pr, 1, &P0[j]);
            }
            T *t_ = P1; P1 = P0; P0 = t_;
        }
        free(P0); free(P1);
    } else {                                                /* 3-D: sz_float.c:2270-2730, products in double */
        T *P0 = (T *)malloc(r23 * sizeof(T)), *P1 = (T *)malloc(r23 * sizeof(T));
        type[0] = 0; P1[0] = FN(szo_exact_add)(&E, data[0]);
        PT(1, data[1], P1[0], 1, &P1[1]);
        for (size_t j = 2; j < r3; j++) { const double t = P1[j - 1]; const T pr = (T)(t * t / P1[j - 2]); PT(j, data[j], pr, 1, &P1[j]); }
        for (size_t i = 1; i < r2; i++) {
            size_t idx = i * r3;
            PT(idx, data[idx], P1[idx - r3], 0, &P1[idx]);                                       /* no fabs here (:2459) */
            for (size_t j = 1; j < r3; j++) {
                idx = i * r3 + j;
                const double t = P1[idx - 1];
                const T pr = (T)(t * P1[idx - r3] / P1[idx - r3 - 1]);
                PT(idx, data[idx], pr, 1, &P1[idx]);
            }
        }
        for (size_t k = 1; k < r1; k++) {
            size_t idx = k * r23;
            PT(idx, data[idx], P1[0], 1, &P0[0]);
            for (size_t j = 1; j < r3; j++) {
                idx++;
                const double t = P0[j - 1];
                const T pr = (T)(t * P1[j] / P1[j - 1]);
                PT(idx, data[idx], pr, 1, &P0[j]);
            }
            for (size_t i = 1; i < r2; i++) {
                size_t q = i * r3;
                idx = k * r23 + q;
                { const double t = P0[q - r3]; const T pr = (T)(t * P1[q] / P1[q - r3]); PT(idx, data[idx], pr, 1, &P0[q]); }
                for (size_t j = 1; j < r3; j++) {
                    idx++; q = i * r3 + j;
                    const double t = P0[q - 1], t2 = P0[q - r3 - 1];
                    const T pr = (T)(t * P0[q - r3] * P1[q] * P1[q - r3 - 1] / (t2 * P1[q - r3] * P1[q - 1]));
                    PT(idx, data[idx], pr, 1, &P0[q]);
                }
            }
            T *t_ = P1; P1 = P0; P0 = t_;
        }
        free(P0); free(P1);
    }
#undef PT
    szo_msst_tab_free(&tb);
    return FN(szo_sz14_pack)(p, meta, meta_len, n, intervals, type, &E, precision, median_stored, out_size, st, pw);
}

static unsigned char *FN(szo_msst_compress)(const szo_params *p, const unsigned char *meta, size_t meta_len, const T *ori, size_t r1, size_t r2, size_t r3,
                                            double ratio, T vmax, size_t segment_size, size_t *out_size, szo_stages *st)
{
    const size_t n = r1 * r2 * r3;
    /* computeRangeSize_float_MSST19: signs from element 1 on, the non-zero value of least magnitude (starting from element 0, zero or not) */
    unsigned char *signs = (unsigned char *)calloc(n, 1);
    int positive = 1;
    T near_zero = ori[0];
    for (size_t i = 1; i < n; i++) {
        if (ori[i] < 0) { signs[i] = 1; positive = 0; }
        if (ori[i] != 0 && FABS_T(ori[i]) < FABS_T(near_zero)) near_zero = ori[i];
    }
    T *data = (T *)malloc(n * sizeof(T));
    const T multiplier = (T)pow(1 + ratio, -3.0001);
    for (size_t i = 0; i < n; i++) data[i] = ori[i] == 0 ? (T)(near_zero * multiplier) : ori[i];
    const T median_log = (T)sqrt(fabs((double)(T)(near_zero * vmax)));
    szo_pwr_extra pw; memset(&pw, 0, sizeof(pw));
    pw.segment_size = segment_size; pw.msst19 = 1; pw.plus_bits = 3;                 /* conf.c:97 */
    pw.min_log_value = (double)(T)((double)near_zero / ((1 + ratio) * (1 + ratio)));
    unsigned char *blob = NULL;
    if (!positive) {
        blob = szo_zstd_compress(signs, n, 3, &pw.blob_size);
        if (!blob) { free(data); free(signs); return NULL; }
        pw.blob = blob;
    }
    free(signs);
    unsigned char *out = FN(szo_msst_quantise)(p, meta, meta_len, data, r1, r2, r3, ratio, median_log, out_size, st, &pw);
    free(data); free(blob);
    return out;
}

/* ---- reconstruction (szd_float.c:1702-2700): `type` decoded, R at the first exact value ---- */
static void FN(szo_msst_reconstruct)(T *out, size_t r1, size_t r2, size_t r3, const int *type, FN(szo_exact_rd) *R, unsigned intervals,
                                     double ratio, int plus_bits)
{
    const size_t n = r1 * r2 * r3, r23 = r2 * r3;
    const int radius = (int)intervals / 2;
    double *ptab = (double *)malloc(sizeof(double) * intervals);
    const double inv = 2.0 - pow(2, -plus_bits);
    for (unsigned i = 0; i < intervals; i++) ptab[i] = pow(1 + ratio, inv * ((int)i - radius));
#define DEC(IDX, PRED) do { const int t_ = type[IDX]; out[IDX] = t_ ? (T)(fabs((double)(T)(PRED)) * ptab[t_]) : FN(szo_exact_next)(R); } while (0)
    if (r1 == 1 && r2 == 1) {
        T pv = 0;
        for (size_t i = 0; i < n; i++) { if (type[i]) pv = (T)(fabs((double)pv) * ptab[type[i]]); else pv = FN(szo_exact_next)(R); out[i] = pv; }
    } else if (r1 == 1) {                                   /* 2-D (szd_float.c:1808-2127): products in T */
        out[0] = FN(szo_exact_next)(R);
        DEC(1, out[0]);
        for (size_t j = 2; j < r3; j++) DEC(j, (T)(out[j - 1] * out[j - 1]) / out[j - 2]);
        for (size_t i = 1; i < r2; i++) {
            size_t idx = i * r3;
            DEC(idx, out[idx - r3]);
            for (size_t j = 1; j < r3; j++) { idx++; DEC(idx, (T)(out[idx - 1] * out[idx - r3]) / out[idx - r3 - 1]); }
        }
    } else {                                                /* 3-D (szd_float.c:2129-2700): products in double */
        out[0] = FN(szo_exact_next)(R);
        DEC(1, out[0]);
        for (size_t j = 2; j < r3; j++) DEC(j, (double)out[j - 1] * out[j - 1] / out[j - 2]);
        for (size_t i = 1; i < r2; i++) {
            size_t idx = i * r3;
            DEC(idx, out[idx - r3]);
            for (size_t j = 1; j < r3; j++) { idx++; DEC(idx, (double)out[idx - 1] * out[idx - r3] / out[idx - r3 - 1]); }
        }
        for (size_t k = 1; k < r1; k++) {
            size_t idx = k * r23;
            DEC(idx, out[idx - r23]);
            for (size_t j = 1; j < r3; j++) { idx++; DEC(idx, (double)out[idx - 1] * out[idx - r23] / out[idx - r23 - 1]); }
            for (size_t i = 1; i < r2; i++) {
                idx = k * r23 + i * r3;
                DEC(idx, (double)out[idx - r3] * out[idx - r23] / out[idx - r23 - r3]);
                for (size_t j = 1; j < r3; j++) {
                    idx++;
                    DEC(idx, (double)out[idx - 1] * out[idx - r3] * out[idx - r23] * out[idx - r23 - r3 - 1]
                             / ((double)out[idx - r3 - 1] * out[idx - r23 - r3] * out[idx - r23 - 1]));
                }
            }
        }
    }
#undef DEC
    free(ptab);
}

static int FN(szo_msst_decompress)(T *out, size_t r1, size_t r2, size_t r3, const unsigned char *b, size_t avail)
{
    const size_t n = r1 * r2 * r3;
    szo_pwr_extra pw; memset(&pw, 0, sizeof(pw));
    pw.msst19 = 1;
    if (FN(szo_sz14_decompress_3d)(out, r1, r2, r3, b, avail, &pw)) return -1;
    const T threshold = (T)pw.min_log_value;
    if (pw.blob_size > 0) {                                  /* szd_float_pwr.c:1430-1451 */
        unsigned char *signs = szo_zstd_decompress(pw.blob, pw.blob_size, n);
        if (!signs) return -1;
        for (size_t i = 0; i < n; i++) {
            if (out[i] < threshold && out[i] >= 0) { out[i] = 0; continue; }
            if (signs[i]) { UT u; memcpy(&u, &out[i], NBYTES); u |= (UT)1 << (8 * NBYTES - 1); memcpy(&out[i], &u, NBYTES); }
        }
        free(signs);
    } else
        for (size_t i = 0; i < n; i++) if (out[i] < threshold) out[i] = 0;
    return 0;
}
#undef UT
#undef NBYTES
#undef FN
