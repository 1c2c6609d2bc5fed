/*
 * szo_sz14_impl.h -- ORACLE (test infrastructure only), type-generic body of the SZ 1.4 ("no regression") 3-D path.
 * Included twice by szo_api.c (T = float / double).  Never part of the product.
 *
 * What the reference does (paths relative to the reference tree):
 *   interval optimiser      sz/src/sz_float.c:4644-4702  optimize_intervals_float_3D_opt          (double: sz_double.c:4337)
 *   compressor              sz/src/sz_float.c:946-1415   SZ_compress_float_3D_MDQ                 (double: sz_double.c:784)
 *   required bit length     sz/src/sz_float.c:45-56      computeReqLength_float                   (double: sz_double.c:44-55)
 *   lossy "exact" values    sz/src/dataCompression.c:454-477 compressSingleFloatValue, CompressElement.c:230-254
 *                           updateLossyCompElement_Float, dataCompression.c:562-592 leading bytes / addExactData
 *   container               sz/src/TightDataPointStorageF.c:275-327 (new_...), :379-479 (to bytes), :590-663 (flat bytes)
 *   type array blob         sz/src/Huffman.c:790-816 encode_withTree
 *   decompressor            sz/src/szd_float.c:600-1138  decompressDataSeries_float_3D            (double: szd_double.c)
 *
 * Every point is predicted from RECONSTRUCTED neighbours of the whole array (no blocks): the 7-point Lorenzo stencil in the
 * order  [-1] + [-r3] + [-r23] - [-r3-1] - [-r23-r3] - [-r23-1] + [-r23-r3-1]  (sz_float.c:1311), which on the array's faces
 * degenerates to the 3-point / 1-point forms the reference spells out (:1104, :1142, :1185, :1225, :1266 -- adding the missing
 * terms as zeros is exact) EXCEPT on the very first row, where it is P[0] for j = 1 and 2P[j-1] - P[j-2] beyond (:1034, :1072).
 * The first value is always stored "exactly".  "Exact" values are lossy: the value minus the median keeps its reqLength leading
 * bits, and THAT is what the neighbours see.
 *
 * Pinning: the 3-D path is pinned to a recorded output of the unmodified reference (tests/test_oracle_pins.py).  The 2-D path
 * (r1 == 1) and the 1-D path (r1 == r2 == 1: SZ_compress_float_1D_MDQ, sz_float.c:353-540, its optimiser :5070-5111, inverse
 * szd_float.c:185-282; doubles sz_double.c:260-400, :4747) share the pinned container and exact-value code; since round 2 they are pinned
 * by recorded outputs of the unmodified reference of their own (2-D and 1-D cases of tests/golden/ref_recorded.json).
 */

#ifndef SZO_CAT
#define SZO_CAT_(a, b) a##_##b
#define SZO_CAT(a, b) SZO_CAT_(a, b)
#endif
#define FN(name) SZO_CAT(name, SUF)

#if IS_F64
#define UT uint64_t
#define NBYTES 8
#else
#define UT uint32_t
#define NBYTES 4
#endif

/* ---- interval optimiser (sz_float.c:4644-4702): the sample walk of the SZ 2.1 optimiser, radius histogram only ---- */
static unsigned FN(szo_optimize_intervals_3d_opt)(const szo_params *p, const T *data, size_t r1, size_t r2, size_t r3, double ebD)
{
    const size_t r23 = r2 * r3, len = r1 * r2 * r3;
    unsigned maxRangeRadius = p->max_quant_intervals / 2;
    size_t *iv = (size_t *)calloc(maxRangeRadius, sizeof(size_t));
    const size_t sd = (size_t)p->sample_distance;
    size_t total = 0, oc = sd - 2, n1 = 1, n2 = 1;
    size_t pos = r23 + r3 + oc;
    while (pos < len) {
        const T *d = data + pos;
        total++;
        T pred = d[-1] + d[-(ptrdiff_t)r3] + d[-(ptrdiff_t)r23] - d[-1 - (ptrdiff_t)r23] - d[-(ptrdiff_t)r3 - 1]
                 - d[-(ptrdiff_t)r3 - (ptrdiff_t)r23] + d[-(ptrdiff_t)r3 - (ptrdiff_t)r23 - 1];
        T pred_err = (T)fabs((double)(T)(pred - *d));
        size_t ri = (size_t)(((double)pred_err / ebD + 1) / 2);
        if (ri >= maxRangeRadius) ri = maxRangeRadius - 1;
        iv[ri]++;
        oc += sd;
        if (oc >= r3) {
            n2++;
            if (n2 == r2) { n1++; n2 = 1; pos += r3; }
            size_t oc2 = (n1 + n2) % sd;
            pos += (r3 + sd - oc) + (sd - oc2);
            oc = sd - oc2;
            if (oc == 0) oc++;
        } else pos += sd;
    }
    size_t target = (size_t)(total * p->pred_threshold);   /* size_t * float */
    size_t sum = 0, i;
    for (i = 0; i < maxRangeRadius; i++) { sum += iv[i]; if (sum > target) break; }
    if (i >= maxRangeRadius) i = maxRangeRadius - 1;
    unsigned pow2 = szo_round_up_pow2(2 * (unsigned)(i + 1));
    if (pow2 < 32) pow2 = 32;
    free(iv);
    return pow2;
}

/* 2-D (sz_float.c:5015-5068): the lattice of the 2-D SZ 2.1 optimiser, radius histogram only */
static unsigned FN(szo_optimize_intervals_2d_opt)(const szo_params *p, const T *data, size_t r1, size_t r2, double ebD)
{
    const size_t len = r1 * r2;
    unsigned maxRangeRadius = p->max_quant_intervals / 2;
    size_t *iv = (size_t *)calloc(maxRangeRadius, sizeof(size_t));
    const size_t sd = (size_t)p->sample_distance;
    size_t total = 0, oc = sd - 1, n1 = 1;
    size_t pos = r2 + oc;
    while (pos < len) {
        const T *d = data + pos;
        total++;
        T pred = d[-1] + d[-(ptrdiff_t)r2] - d[-(ptrdiff_t)r2 - 1];
        T pred_err = (T)fabs((double)(T)(pred - *d));
        size_t ri = (size_t)(((double)pred_err / ebD + 1) / 2);
        if (ri >= maxRangeRadius) ri = maxRangeRadius - 1;
        iv[ri]++;
        oc += sd;
        if (oc >= r2) {
            n1++;
            size_t oc2 = n1 % sd;
            pos += (r2 + sd - oc) + (sd - oc2);
            oc = sd - oc2;
            if (oc == 0) oc++;
        } else pos += sd;
    }
    size_t target = (size_t)(total * p->pred_threshold);
    size_t sum = 0, i;
    for (i = 0; i < maxRangeRadius; i++) { sum += iv[i]; if (sum > target) break; }
    if (i >= maxRangeRadius) i = maxRangeRadius - 1;
    unsigned pow2 = szo_round_up_pow2(2 * (unsigned)(i + 1));
    if (pow2 < 32) pow2 = 32;
    free(iv);
    return pow2;
}

/* 1-D optimiser (optimize_intervals_float_1D_opt, sz_float.c:5070-5111 / sz_double.c:4747): previous-value predictor at
 * positions 2, 2+sampleDistance, ...  Pinned by the recorded 1-D cases of tests/golden/ref_recorded.json. */
static unsigned FN(szo_optimize_intervals_1d_opt)(const szo_params *p, const T *data, size_t len, double ebD)
{
    unsigned maxRangeRadius = p->max_quant_intervals / 2;
    size_t *iv = (size_t *)calloc(maxRangeRadius, sizeof(size_t));
    size_t total = 0;
    for (size_t pos = 2; pos < len; pos += (size_t)p->sample_distance) {
        total++;
        T pe = data[pos - 1] - data[pos];
        double pred_err = fabs((double)pe);
        size_t ri = (size_t)(uint64_t)((pred_err / ebD + 1) / 2);
        if (ri >= maxRangeRadius) ri = maxRangeRadius - 1;
        iv[ri]++;
    }
    size_t target = (size_t)(total * p->pred_threshold);
    size_t sum = 0, i;
    for (i = 0; i < maxRangeRadius; i++) { sum += iv[i]; if (sum > target) break; }
    if (i >= maxRangeRadius) i = maxRangeRadius - 1;
    unsigned pow2 = szo_round_up_pow2(2 * (unsigned)(i + 1));
    if (pow2 < 32) pow2 = 32;
    free(iv);
    return pow2;
}

/* required length of an "exact" value in bits, and the median it is taken against (sz_float.c:45-56 / sz_double.c:44-55) */
static int FN(szo_req_length)(double eb, T range, T *median)
{
    T half = range / 2;
    UT u; memcpy(&u, &half, NBYTES);
    uint64_t e; memcpy(&e, &eb, 8);
    int reqExpo = (int)((e & 0x7FF0000000000000ull) >> 52) - 1023;          /* getPrecisionReqLength_double */
#if IS_F64
    int radExpo = (int)((u & 0x7FF0000000000000ull) >> 52) - 1023;          /* getExponent_double */
    int req = 12 + (short)radExpo - (short)reqExpo;
    if (req < 12) req = 12;
    if (req > 64) { req = 64; *median = 0; }
#else
    int radExpo = (int)((u & 0x7F800000u) >> 23) - 127;                     /* getExponent_float */
    int req = 9 + (short)radExpo - (short)reqExpo + 1;
    if (req < 9) req = 9;
    if (req > 32) { req = 32; *median = 0; }
#endif
    return req;
}

typedef struct FN(szo_exact) {
    unsigned char *lead; size_t n, cap;          /* one entry per exact value: identical leading bytes with the previous one (0..3) */
    unsigned char *mid; size_t nmid, capmid;     /* the bytes [lead, reqBytes) of every exact value, concatenated */
    unsigned char *resi;                         /* the residual bits of every exact value (one byte each, right-aligned) */
    unsigned char pre[8];
    int req_len, req_bytes, resi_bits;
    T median;
} FN(szo_exact);

/* compressSingleFloatValue + updateLossyCompElement_Float + addExactData; returns what the neighbours will see */
static T FN(szo_exact_add)(FN(szo_exact) *E, T x)
{
    T norm = x - E->median;
    UT u; memcpy(&u, &norm, NBYTES);
    unsigned char cur[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < NBYTES; i++) cur[i] = (unsigned char)(u >> (8 * (NBYTES - 1 - i)));   /* big-endian bytes of the FULL value */
    int ign = 8 * NBYTES - E->req_len; if (ign < 0) ign = 0;
#if IS_F64
    int64_t s; memcpy(&s, &u, 8); s = (s >> ign) << ign; memcpy(&u, &s, 8);      /* arithmetic shift of the signed image */
#else
    int32_t s; memcpy(&s, &u, 4); s = (s >> ign) << ign; memcpy(&u, &s, 4);
#endif
    T kept; memcpy(&kept, &u, NBYTES);
    int lead = 0;
    for (int i = 0; i < NBYTES; i++) { if (E->pre[i] == cur[i]) lead++; else break; }
    if (lead > 3) lead = 3;
    if (E->n == E->cap) { E->cap = E->cap ? 2 * E->cap : 1024; E->lead = realloc(E->lead, E->cap); E->resi = realloc(E->resi, E->cap); }
    if (E->nmid + 8 > E->capmid) { E->capmid = E->capmid ? 2 * E->capmid : 4096; E->mid = realloc(E->mid, E->capmid); }
    E->lead[E->n] = (unsigned char)lead;
    for (int i = lead; i < E->req_bytes; i++) E->mid[E->nmid++] = cur[i];
    int rb = 0;
    if (E->resi_bits != 0 && E->req_bytes < 8) rb = (cur[E->req_bytes] & 0xFF) >> (8 - E->resi_bits);
    E->resi[E->n] = (unsigned char)rb;
    E->n++;
    memcpy(E->pre, cur, NBYTES);
    return kept + E->median;
}

/* one point: code (0 = exact) and reconstruction.  itv goes through DOUBLE here (`fabs`, sz_float.c:1040), unlike SZ 2.1 */
static inline int FN(szo_sz14_point)(FN(szo_exact) *E, T x, T pred, T eb, T recip, unsigned intervals, int radius, T *rc)
{
    T diff = x - pred;
    T itv = (T)(fabs((double)diff) * (double)recip + 1);
    if (itv < intervals) {
        if (diff < 0) itv = -itv;
        int code = (int)(itv / 2) + radius;
        T r = pred + 2 * (code - radius) * eb;
        if (fabs((double)(T)(x - r)) > (double)eb) { *rc = FN(szo_exact_add)(E, x); return 0; }
        *rc = r;
        return code;
    }
    *rc = FN(szo_exact_add)(E, x);
    return 0;
}

/* everything after the quantiser: type-array blob, exact-value arrays, container.  `median_stored` is the container's median field
 * (the MSST19 form stores another number there than the one its exact values are taken against, sz_float_pwr.c:2060-2062) */
static unsigned char *FN(szo_sz14_pack)(const szo_params *p, const unsigned char *meta, size_t meta_len, size_t n, unsigned intervals, int *type,
                                        FN(szo_exact) *Ep, double eb_field, T median_stored, size_t *out_size, szo_stages *st, const szo_pwr_extra *pw)
{
#define E (*Ep)

    /* type array blob (Huffman.c:790-816): nodeCount | stateNum/2 | tree | payload */
    szo_huff *h = szo_huff_from_symbols(2 * (int)intervals, type, n);
    size_t node_count = szo_huff_node_count(h);
    unsigned char *tree = NULL;
    size_t tree_bytes = szo_huff_tree_to_bytes(h, &tree);
    unsigned char *payload = (unsigned char *)calloc(n * sizeof(int) + 16, 1);
    size_t huff_bytes = szo_huff_encode(h, type, n, payload);
    const size_t type_size = 8 + tree_bytes + huff_bytes;

    /* 2-bit lead array (TypeManager.c:134-175) and residual bits, MSB first (:377-415) */
    const size_t lead_size = (E.n * 2 + 7) / 8;
    const size_t resi_size = E.resi_bits ? (E.n * (size_t)E.resi_bits + 7) / 8 : 0;

    /* container (TightDataPointStorageF.c:379-479): sizes are 8 bytes (SZ_SIZE_TYPE) */
    const size_t total = meta_len + 8 + 4 + 4 + NBYTES + 1 + 8 + 8 + 8 + 8 + type_size + lead_size + E.nmid + resi_size
                       + (pw ? 1 + 8 + 4 + NBYTES + pw->blob_size + (pw->msst19 ? 2 : 0) : 0);
    unsigned char *out = (unsigned char *)calloc(total + 8, 1);
    unsigned char *q = out;
    memcpy(q, meta, meta_len); q += meta_len;
    szo_put_u64be(q, n); q += 8;
    szo_put_u32be(q, p->max_quant_intervals); q += 4;
    if (pw) { *q++ = 0; /* radExpo */ szo_put_u64be(q, (uint64_t)pw->segment_size); q += 8; szo_put_u32be(q, (uint32_t)pw->blob_size); q += 4; }
    szo_put_u32be(q, intervals); q += 4;
    FN(szo_put_be)(q, median_stored); q += NBYTES;
    *q++ = (unsigned char)E.req_len;
    if (pw && pw->msst19) {                      /* plus_bits, max_bits (TightDataPointStorageF.c:431-435; max_bits: Huffman.c:828-833) */
        int max_bits = 0;
        for (unsigned i = 0; i < 2 * intervals; i++) if (h->len[i] > max_bits) max_bits = h->len[i];
        *q++ = (unsigned char)pw->plus_bits; *q++ = (unsigned char)max_bits;
    }
    szo_put_be_f64(q, eb_field); q += 8;
    szo_put_u64be(q, type_size); q += 8;
    szo_put_u64be(q, E.n); q += 8;
    szo_put_u64be(q, E.nmid); q += 8;
    if (pw) { FN(szo_put_be)(q, (T)pw->min_log_value); q += NBYTES; }
    szo_put_u32be(q, (uint32_t)node_count); szo_put_u32be(q + 4, intervals);
    memcpy(q + 8, tree, tree_bytes); memcpy(q + 8 + tree_bytes, payload, huff_bytes); q += type_size;
    if (pw && pw->blob_size) { memcpy(q, pw->blob, pw->blob_size); q += pw->blob_size; }
    for (size_t i = 0; i < E.n; i++) q[i >> 2] |= (unsigned char)(E.lead[i] << (6 - 2 * (i & 3)));
    q += lead_size;
    memcpy(q, E.mid, E.nmid); q += E.nmid;
    if (E.resi_bits) {
        size_t bit = 0;
        for (size_t i = 0; i < E.n; i++, bit += (size_t)E.resi_bits) {
            unsigned v = (unsigned)E.resi[i] << (16 - E.resi_bits - (bit & 7));      /* at most 7 bits, spans at most 2 bytes */
            q[bit >> 3] |= (unsigned char)(v >> 8);
            q[(bit >> 3) + 1] |= (unsigned char)v;                                   /* out has 8 spare bytes */
        }
        q += resi_size;
    }
    *out_size = (size_t)(q - out);

    if (st) {
        memset(st, 0, sizeof(*st));
        st->num_elements = n; st->total_unpred = E.n; st->intervals = intervals; st->eb = eb_field; st->mean = (double)median_stored;
        st->codes = type; type = NULL;
        st->num_blocks = E.nmid;                 /* SZ 1.4: number of mid bytes */
        st->use_mean = E.req_len;                /* SZ 1.4: reqLength */
        st->indicator = E.lead; E.lead = NULL;   /* SZ 1.4: lead numbers, one per exact value */
        st->unpred = E.mid; E.mid = NULL;        /* SZ 1.4: mid bytes */
        st->code_len = (unsigned char *)malloc(2 * (size_t)intervals);
        memcpy(st->code_len, h->len, 2 * (size_t)intervals);
        st->tree_bytes = tree_bytes; st->node_count = node_count; st->huff_bytes = huff_bytes;
    }
    free(type); free(tree); free(payload); szo_huff_free(h);
    free(E.lead); free(E.mid); free(E.resi);
    return out;
#undef E
}

/* r1 slowest ... r3 fastest (callee convention of sz_float.c:946).  `meta` = version bytes, flag byte and parameter bytes.
 * r1 == 1 is the 2-D compressor SZ_compress_float_2D_MDQ (sz_float.c:610-894; inverse szd_float.c:284-598): its predictors are
 * exactly those of layer 0 below (:686, :727, :772, :814); only its optimiser walks another lattice.  Pinned by the recorded
 * `sz14-2D-*` cases of tests/golden/ref_recorded.json. */
static unsigned char *FN(szo_sz14_compress_3d)(const szo_params *p, const unsigned char *meta, size_t meta_len,
                                               const T *data, size_t r1, size_t r2, size_t r3, T eb, T range, T median_in,
                                               size_t *out_size, szo_stages *st, const szo_pwr_extra *pw)
{   /* pw != NULL: the extra container fields of a point-wise-relative stream (TightDataPointStorageF.c:408-419, 454-467) */
    const size_t n = r1 * r2 * r3, r23 = r2 * r3;
    const T recip = 1 / eb;
    unsigned intervals = p->quantization_intervals ? p->quantization_intervals
                         : r1 == 1 && r2 == 1 ? FN(szo_optimize_intervals_1d_opt)(p, data, r3, (double)eb)
                         : r1 == 1 ? FN(szo_optimize_intervals_2d_opt)(p, data, r2, r3, (double)eb)
                                   : FN(szo_optimize_intervals_3d_opt)(p, data, r1, r2, r3, (double)eb);
    const int radius = (int)intervals / 2;
    FN(szo_exact) E; memset(&E, 0, sizeof(E));
    E.median = median_in;
    E.req_len = FN(szo_req_length)((double)eb, range, &E.median);
    E.req_bytes = E.req_len / 8; E.resi_bits = E.req_len % 8;

    int *type = (int *)malloc(n * sizeof(int));
    T *P0 = (T *)malloc(r23 * sizeof(T)), *P1 = (T *)malloc(r23 * sizeof(T));   /* P1: previous layer, P0: current (roles swap) */

    if (r1 == 1 && r2 == 1) {
        /* the 1-D compressor SZ_compress_float_1D_MDQ (sz_float.c:353-540) / SZ_compress_double_1D_MDQ (sz_double.c:260-400): the
         * first two values exact, then the previous reconstruction as predictor; its own quantiser (a radius test, the state
         * from a truncation and a shift); only the float version re-checks the bound.  Pinned by the recorded `1D-*` cases. */
        const T check_radius = (T)((intervals - 1) * eb), interval = 2 * eb;
        type[0] = 0; (void)FN(szo_exact_add)(&E, data[0]);
        type[1] = 0;
        T pred = FN(szo_exact_add)(&E, data[1]);
        for (size_t i = 2; i < n; i++) {
            const T x = data[i];
#if IS_F64
            const T err = fabs(x - pred);
#else
            const T err = fabsf(x - pred);
#endif
            if (err < check_radius) {
#if IS_F64
                const int state = (int)((err * recip + 1) * 0.5);
#else
                const int state = ((int)(T)(err * recip + 1)) >> 1;
#endif
                const T step = (T)(state * interval);
                if (x >= pred) { type[i] = radius + state; pred = pred + step; }
                else { type[i] = radius - state; pred = pred - step; }
#if !IS_F64
                if (fabs((double)(T)(x - pred)) > (double)eb) { type[i] = 0; pred = FN(szo_exact_add)(&E, x); }
#endif
                continue;
            }
            type[i] = 0;
            pred = FN(szo_exact_add)(&E, x);
        }
        goto pack;
    }

    /* layer 0 */
    type[0] = 0;
    P1[0] = FN(szo_exact_add)(&E, data[0]);
    if (r3 > 1) type[1] = FN(szo_sz14_point)(&E, data[1], P1[0], eb, recip, intervals, radius, &P1[1]);
    for (size_t j = 2; j < r3; j++) {
        T pred = 2 * P1[j - 1] - P1[j - 2];
        type[j] = FN(szo_sz14_point)(&E, data[j], pred, eb, recip, intervals, radius, &P1[j]);
    }
    for (size_t i = 1; i < r2; i++) {
        size_t idx = i * r3;
        type[idx] = FN(szo_sz14_point)(&E, data[idx], P1[idx - r3], eb, recip, intervals, radius, &P1[idx]);
        for (size_t j = 1; j < r3; j++) {
            idx = i * r3 + j;
            T pred = P1[idx - 1] + P1[idx - r3] - P1[idx - r3 - 1];
            type[idx] = FN(szo_sz14_point)(&E, data[idx], pred, eb, recip, intervals, radius, &P1[idx]);
        }
    }
    /* layers 1 .. r1-1 */
    for (size_t k = 1; k < r1; k++) {
        size_t idx = k * r23;
        type[idx] = FN(szo_sz14_point)(&E, data[idx], P1[0], eb, recip, intervals, radius, &P0[0]);
        for (size_t j = 1; j < r3; j++) {
            idx = k * r23 + j;
            T pred = P0[j - 1] + P1[j] - P1[j - 1];
            type[idx] = FN(szo_sz14_point)(&E, data[idx], pred, eb, recip, intervals, radius, &P0[j]);
        }
        for (size_t i = 1; i < r2; i++) {
            size_t q = i * r3;
            idx = k * r23 + q;
            T pred = P0[q - r3] + P1[q] - P1[q - r3];
            type[idx] = FN(szo_sz14_point)(&E, data[idx], pred, eb, recip, intervals, radius, &P0[q]);
            for (size_t j = 1; j < r3; j++) {
                q = i * r3 + j;
                idx = k * r23 + q;
                pred = P0[q - 1] + P0[q - r3] + P1[q] - P0[q - r3 - 1] - P1[q - r3] - P1[q - 1] + P1[q - r3 - 1];
                type[idx] = FN(szo_sz14_point)(&E, data[idx], pred, eb, recip, intervals, radius, &P0[q]);
            }
        }
        T *t = P1; P1 = P0; P0 = t;
    }
pack:
    free(P0); free(P1);
    return FN(szo_sz14_pack)(p, meta, meta_len, n, intervals, type, &E, (double)eb, E.median, out_size, st, pw);
}

/* ---- decompressor (szd_float.c:600-1138).  `b` points at the max_quant_intervals field (just after the element count) ---- */
typedef struct FN(szo_exact_rd) {
    const unsigned char *lead, *mid, *resi;
    size_t l, m, bit;
    unsigned char pre[8];
    int req_bytes, resi_bits;
    T median;
} FN(szo_exact_rd);

static T FN(szo_exact_next)(FN(szo_exact_rd) *R)
{
    unsigned rb = 0;
    if (R->resi_bits) {
        unsigned w = ((unsigned)R->resi[R->bit >> 3] << 8) | R->resi[(R->bit >> 3) + 1];
        rb = (w >> (16 - R->resi_bits - (R->bit & 7))) & ((1u << R->resi_bits) - 1);
        R->bit += (size_t)R->resi_bits;
    }
    unsigned char cur[8]; memset(cur, 0, 8);
    int lead = (R->lead[R->l >> 2] >> (6 - 2 * (R->l & 3))) & 3; R->l++;
    memcpy(cur, R->pre, (size_t)lead);
    for (int j = lead; j < R->req_bytes; j++) cur[j] = R->mid[R->m++];
    if (R->resi_bits) cur[R->req_bytes] = (unsigned char)(rb << (8 - R->resi_bits));
    UT u = 0;
    for (int i = 0; i < NBYTES; i++) u = (u << 8) | cur[i];
    T v; memcpy(&v, &u, NBYTES);
    memcpy(R->pre, cur, NBYTES);
    return v + R->median;
}

static void FN(szo_msst_reconstruct)(T *out, size_t r1, size_t r2, size_t r3, const int *type, FN(szo_exact_rd) *R, unsigned intervals,
                                     double ratio, int plus_bits);        /* szo_msst_impl.h */

static int FN(szo_sz14_decompress_3d)(T *out, size_t r1, size_t r2, size_t r3, const unsigned char *b, size_t avail, szo_pwr_extra *pw)
{   /* pw != NULL: a point-wise-relative stream; its extra fields are skipped here and handed back.  pw->msst19 (set by the caller from
     * flag 0x08): the table-driven form -- two more header bytes, another reconstruction (szo_msst_impl.h) */
    const size_t n = r1 * r2 * r3, r23 = r2 * r3;
    const unsigned char *q = b;
    q += 4;                                              /* max_quant_intervals */
    if (pw) { q += 1; pw->segment_size = (size_t)szo_get_u64be(q); q += 8; pw->blob_size = szo_get_u32be(q); q += 4; }
    unsigned intervals = szo_get_u32be(q); q += 4;
    FN(szo_exact_rd) R; memset(&R, 0, sizeof(R));
    R.median = FN(szo_get_be)(q); q += NBYTES;
    int req_len = *q++;
    const int msst19 = pw && pw->msst19;
    if (msst19) { pw->plus_bits = q[0]; q += 2; R.median = 0; }         /* plus_bits, max_bits (TightDataPointStorageF.c:164-168) */
    const double eb_field = szo_get_be_f64(q);
    T eb = (T)eb_field; q += 8;                          /* `float realPrecision = tdps->realPrecision`, szd_float.c:610 */
    size_t type_size = (size_t)szo_get_u64be(q); q += 8;
    size_t exact_n = (size_t)szo_get_u64be(q); q += 8;
    size_t mid_n = (size_t)szo_get_u64be(q); q += 8;
    if (pw) { pw->min_log_value = (double)FN(szo_get_be)(q); q += NBYTES; }
    if ((size_t)(q - b) + type_size + (pw ? pw->blob_size : 0) + (exact_n * 2 + 7) / 8 + mid_n > avail) return -1;
    int node_count = (int)szo_get_u32be(q);
    int *type = (int *)malloc(n * sizeof(int));
    {
        szo_huff *h = szo_huff_tree_from_bytes(2 * (int)intervals, q + 8, node_count);
        size_t tree_bytes = node_count <= 256 ? 1 + 7 * (size_t)node_count : node_count <= 65536 ? 1 + 9 * (size_t)node_count : 1 + 13 * (size_t)node_count;
        szo_huff_decode(h, q + 8 + tree_bytes, n, type);
        szo_huff_free(h);
    }
    q += type_size;
    if (pw) { pw->blob = q; q += pw->blob_size; }
    R.lead = q; q += (exact_n * 2 + 7) / 8;
    R.mid = q; q += mid_n;
    unsigned char *resi_pad = (unsigned char *)calloc((size_t)(b + avail - q) + 8, 1);   /* the bit reader looks one byte ahead */
    memcpy(resi_pad, q, (size_t)(b + avail - q));
    R.resi = resi_pad;
    R.req_bytes = req_len / 8; R.resi_bits = req_len % 8;
    const int radius = (int)intervals / 2;

    if (msst19) { FN(szo_msst_reconstruct)(out, r1, r2, r3, type, &R, intervals, eb_field, pw->plus_bits); free(type); free(resi_pad); return 0; }
#define SZO_DEC(IDX, PRED) do { int t_ = type[IDX]; out[IDX] = t_ ? (T)((PRED) + 2 * (t_ - radius) * eb) : FN(szo_exact_next)(&R); } while (0)
    if (r1 == 1 && r2 == 1) {   /* decompressDataSeries_float_1D (szd_float.c:185-282): codes 0 at positions 0 and 1 by construction */
        for (size_t i = 0; i < n; i++) { T pred = i ? out[i - 1] : 0; SZO_DEC(i, pred); }
        goto done;
    }
    out[0] = FN(szo_exact_next)(&R);
    if (r3 > 1) SZO_DEC(1, out[0]);
    for (size_t j = 2; j < r3; j++) { T pred = 2 * out[j - 1] - out[j - 2]; SZO_DEC(j, pred); }
    for (size_t i = 1; i < r2; i++) {
        size_t idx = i * r3;
        SZO_DEC(idx, out[idx - r3]);
        for (size_t j = 1; j < r3; j++) { idx = i * r3 + j; T pred = out[idx - 1] + out[idx - r3] - out[idx - r3 - 1]; SZO_DEC(idx, pred); }
    }
    for (size_t k = 1; k < r1; k++) {
        size_t idx = k * r23;
        SZO_DEC(idx, out[idx - r23]);
        for (size_t j = 1; j < r3; j++) { idx = k * r23 + j; T pred = out[idx - 1] + out[idx - r23] - out[idx - r23 - 1]; SZO_DEC(idx, pred); }
        for (size_t i = 1; i < r2; i++) {
            idx = k * r23 + i * r3;
            { T pred = out[idx - r3] + out[idx - r23] - out[idx - r23 - r3]; SZO_DEC(idx, pred); }
            for (size_t j = 1; j < r3; j++) {
                idx = k * r23 + i * r3 + j;
                T pred = out[idx - 1] + out[idx - r3] + out[idx - r23] - out[idx - r3 - 1] - out[idx - r23 - r3] - out[idx - r23 - 1]
                         + out[idx - r23 - r3 - 1];
                SZO_DEC(idx, pred);
            }
        }
    }
done:
#undef SZO_DEC
    free(type); free(resi_pad);
    return 0;
}

#undef UT
#undef NBYTES
#undef FN
