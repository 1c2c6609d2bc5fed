/*
 * szo_sz21_impl.h -- ORACLE (test infrastructure only), type-generic body.
 * Included twice by szo_api.c with T = float / double.
 *
 * Restates the SZ 2.1 "non-blocked Lorenzo with blocked regression" compressor and its inverse:
 *   3-D float  sz/src/sz_float.c:6396-6523 (interval optimiser), :6527-7489 (compressor)
 *   3-D double sz/src/sz_double.c:5773, :5904 (identical modulo the points marked IS_F64)
 *   inverse    sz/src/szd_float.c:3483-5866, szd_double.c:3316
 * The reference keeps two (bx+1)x(r2+1)x(r3+1) strip buffers; here one strip with a carried
 * bottom plane is used -- the values read by the 7-point stencil are the same.
 *
 * 2-D (block size 16, three coefficients): sz_float.c:5405-5515 (optimiser), :5516-6395 (compressor; the reference forces
 * use_mean = 0 at :5615, so only its "else" branch :5984-6282 is live), szd_float.c:3141-3482; sz_double.c:4790-4899, :4900-5757,
 * szd_double.c:2974.  Pinned since round 2 by the recorded `2D-*` outputs of the unmodified reference (tests/golden/ref_recorded.json,
 * tests/test_ref_recorded.py); see DESIGN.md section 2.
 */

#ifndef SZO_CAT
#define SZO_CAT_(a, b) a##_##b
#define SZO_CAT(a, b) SZO_CAT_(a, b)
#endif
#define FN(name) SZO_CAT(name, SUF)

/* ---- interval optimiser, 3-D (sz_float.c:6396-6523) ---- */
static unsigned FN(szo_optimize_intervals_3d)(const szo_params *p, const T *data, size_t r1, size_t r2, size_t r3,
                                              double ebD, T *dense_pos, T *max_freq, T *mean_freq)
{
    T mean = 0.0;
    size_t len = r1 * r2 * r3;
    size_t mean_distance = (size_t)(int)sqrt((double)len);
    size_t pos = 0, off1 = 0, off2 = 0, mean_count = 0;
    while (pos < len) {
        mean += data[pos];
        mean_count++;
        pos += mean_distance; off1 += mean_distance; off2 += mean_distance;
        if (off1 >= r3) { off1 = 0; pos -= 1; }
        if (off2 >= r2 * r3) { off2 = 0; pos -= 1; }
    }
    if (mean_count > 0) mean /= mean_count;

    const size_t range = 8192, radius = 4096;
    size_t *freq_iv = (size_t *)calloc(range, sizeof(size_t));
    unsigned maxRangeRadius = p->max_quant_intervals / 2;
    int sd = p->sample_distance;
    T predThreshold = p->pred_threshold;
    size_t *iv = (size_t *)calloc(maxRangeRadius, sizeof(size_t));
    size_t r23 = r2 * r3;
    size_t freq_count = 0, sample_count = 0;

    size_t oc = (size_t)(sd - 2);
    pos = r23 + r3 + oc;
    size_t n1 = 1, n2 = 1;
    while (pos < len) {
        const T *d = data + pos;
        T pred = d[-1] + d[-(ptrdiff_t)r3] + d[-(ptrdiff_t)r23] - d[-1 - (ptrdiff_t)r23] - d[-(ptrdiff_t)r3 - 1]
                 - d[-(ptrdiff_t)r3 - (ptrdiff_t)r23] + d[-(ptrdiff_t)r3 - (ptrdiff_t)r23 - 1];
        T pred_err = (T)fabs((double)(T)(pred - *d));
        if ((double)pred_err < ebD) freq_count++;
        size_t ri = (size_t)(((double)pred_err / ebD + 1) / 2);
        if (ri >= maxRangeRadius) ri = maxRangeRadius - 1;
        iv[ri]++;

        T mean_diff = *d - mean;
        ptrdiff_t fi;
        if (mean_diff > 0) fi = (ptrdiff_t)((double)mean_diff / ebD) + (ptrdiff_t)radius;
        else fi = (ptrdiff_t)((double)mean_diff / ebD) - 1 + (ptrdiff_t)radius;
        if (fi <= 0) freq_iv[0]++;
        else if ((size_t)fi >= range) freq_iv[range - 1]++;
        else freq_iv[fi]++;

        oc += (size_t)sd;
        if (oc >= r3) {
            n2++;
            if (n2 == r2) { n1++; n2 = 1; pos += r3; }
            size_t oc2 = (n1 + n2) % (size_t)sd;
            pos += (r3 + (size_t)sd - oc) + ((size_t)sd - oc2);
            oc = (size_t)sd - oc2;
            if (oc == 0) oc++;
        } else pos += (size_t)sd;
        sample_count++;
    }
    *max_freq = (T)(freq_count * 1.0 / sample_count);

    size_t target = (size_t)(sample_count * predThreshold);
    size_t sum = 0, i;
    for (i = 0; i < maxRangeRadius; i++) { sum += iv[i]; if (sum > target) break; }
    if (i >= maxRangeRadius) i = maxRangeRadius - 1;
    unsigned acc = 2 * (unsigned)(i + 1);
    unsigned pow2 = szo_round_up_pow2(acc);
    if (pow2 < 32) pow2 = 32;

    size_t max_sum = 0, max_index = 0;
    for (size_t q = 1; q < range - 2; q++) {
        size_t s2 = freq_iv[q] + freq_iv[q + 1];
        if (s2 > max_sum) { max_sum = s2; max_index = q; }
    }
    *dense_pos = (T)(mean + ebD * (double)(ptrdiff_t)(max_index + 1 - radius));
    *mean_freq = (T)(max_sum * 1.0 / sample_count);
    free(freq_iv); free(iv);
    return pow2;
}

/* one point of the quantiser: returns code (0 = unpredictable) and writes the reconstruction.
 * sz_float.c:7267-7287 (Lorenzo) / :7164-7183 (regression); capacity differs between the two. */
static inline int FN(szo_quant_point)(T x, T pred, T eb, T recip, int capacity, int radius, T *recon)
{
    T diff = x - pred;
    T itv = FABS_T(diff) * recip + 1;
    if (itv < capacity) {
        if (diff < 0) itv = -itv;
        int code = (int)(itv / 2) + radius;
        T rc = pred + 2 * (code - radius) * eb;
        if (FABS_T(x - rc) > eb) { *recon = x; return 0; }
        *recon = rc;
        return code;
    }
    *recon = x;
    return 0;
}

/* ---- SZ2.1 3-D compressor.  r1 slowest ... r3 fastest (callee convention of sz_float.c:6527) ---- */
static unsigned char *FN(szo_sz21_compress_3d)(const szo_params *p, const unsigned char *meta, size_t meta_len,
                                               const T *data, size_t r1, size_t r2, size_t r3, T eb,
                                               size_t *out_size, szo_stages *st)
{
    const T recip = 1 / eb;
    const size_t block_size = 6;
    szo_grid gx = szo_make_grid(r1, block_size), gy = szo_make_grid(r2, block_size), gz = szo_make_grid(r3, block_size);
    const size_t nb = gx.num * gy.num * gz.num, ne = r1 * r2 * r3;
    const size_t d0 = r2 * r3, d1 = r3;

    /* regression fit for every block (sz_float.c:6586-6637); SoA a|b|c|d */
    T *reg = (T *)malloc(nb * 4 * sizeof(T));
    {
        size_t b = 0;
        for (size_t bi = 0; bi < gx.num; bi++) for (size_t bj = 0; bj < gy.num; bj++) for (size_t bk = 0; bk < gz.num; bk++, b++) {
            size_t bx = szo_blk_size(&gx, bi), by = szo_blk_size(&gy, bj), bz = szo_blk_size(&gz, bk);
            const T *base = data + szo_blk_start(&gx, bi) * d0 + szo_blk_start(&gy, bj) * d1 + szo_blk_start(&gz, bk);
            T fx = 0.0, fy = 0.0, fz = 0.0, f = 0;
            for (size_t i = 0; i < bx; i++) {
                T sum_x = 0;
                for (size_t j = 0; j < by; j++) {
                    T sum_y = 0;
                    const T *row = base + i * d0 + j * d1;
                    for (size_t k = 0; k < bz; k++) { T c = row[k]; sum_y += c; fz += c * k; }
                    fy += sum_y * j;
                    sum_x += sum_y;
                }
                fx += sum_x * i;
                f += sum_x;
            }
            T coeff = (T)(1.0 / (double)(bx * by * bz));
            T a = (2 * fx / (bx - 1) - f) * 6 * coeff / (bx + 1);
            T bb = (2 * fy / (by - 1) - f) * 6 * coeff / (by + 1);
            T c = (2 * fz / (bz - 1) - f) * 6 * coeff / (bz + 1);
            T d = f * coeff - ((bx - 1) * a / 2 + (by - 1) * bb / 2 + (bz - 1) * c / 2);
            reg[b] = a; reg[nb + b] = bb; reg[2 * nb + b] = c; reg[3 * nb + b] = d;
        }
    }

    /* coefficient precisions (sz_float.c:6640-6645) */
    T rel_param_err = (T)0.025;
    T prec[4], rprec[4];
    prec[0] = rel_param_err * eb / gx.late; prec[1] = rel_param_err * eb / gy.late;
    prec[2] = rel_param_err * eb / gz.late; prec[3] = rel_param_err * eb;
    for (int e = 0; e < 4; e++) rprec[e] = 1 / prec[e];

    unsigned intervals; int use_mean = 0;
    T dense_pos = 0, sample_freq = -1, mean_freq = 0;
    if (p->quantization_intervals == 0) {
        intervals = FN(szo_optimize_intervals_3d)(p, data, r1, r2, r3, (double)eb, &dense_pos, &sample_freq, &mean_freq);
        if (mean_freq > 0.5 || mean_freq > sample_freq) use_mean = 1;
    } else intervals = p->quantization_intervals;

    T mean = 0;
    if (use_mean) { /* sz_float.c:6657-6669: sequential T-typed sum */
        T sum = 0.0; size_t cnt = 0;
        for (size_t i = 0; i < ne; i++) if (FABS_T(data[i] - dense_pos) < eb) { sum += data[i]; cnt++; }
        if (cnt > 0) mean = sum / cnt;
    }

    int *codes = (int *)calloc(ne, sizeof(int));
    T *unpred = (T *)malloc((ne ? ne : 1) * sizeof(T));
    size_t total_unpred = 0;
    unsigned char *indicator = (unsigned char *)calloc(nb, 1);
    int *ccodes = (int *)malloc(nb * 4 * sizeof(int));
    T *cunpred = (T *)malloc(nb * 4 * sizeof(T));
    T *cdec = (T *)malloc(nb * 4 * sizeof(T));
    size_t cunpred_n[4] = {0, 0, 0, 0};
    size_t reg_count = 0;
    T last[4] = {0, 0, 0, 0};

    const int cap = (int)intervals, radius = cap / 2, cap_sz = cap - 2;
    const int ccap = 65536, cradius = 32768;
    const T noise = (T)(eb * 1.22);

    /* reconstruction strip: plane 0 = last plane of previous strip, planes 1..bx = this strip;
     * row 0 / column 0 of every plane stay zero (array-face halo, sz_float.c:6685-6689) */
    const size_t s1 = r3 + 1, s0 = (r2 + 1) * (r3 + 1);
    T *strip = (T *)calloc((gx.early + 1) * s0, sizeof(T));

    size_t b = 0;
    for (size_t bi = 0; bi < gx.num; bi++) {
        const size_t bx = szo_blk_size(&gx, bi), ox = szo_blk_start(&gx, bi);
        for (size_t bj = 0; bj < gy.num; bj++) {
            const size_t by = szo_blk_size(&gy, bj), oy = szo_blk_start(&gy, bj);
            for (size_t bk = 0; bk < gz.num; bk++, b++) {
                const size_t bz = szo_blk_size(&gz, bk), oz = szo_blk_start(&gz, bk);
                const T *base = data + ox * d0 + oy * d1 + oz;
                int *type = codes + ox * d0 + oy * bx * d1 + bx * by * oz; /* sz_float.c:7064,7359 */
                T *pb = strip + s0 + (oy + 1) * s1 + (oz + 1);

                /* predictor selection on original data (sz_float.c:7083-7123 / :6747-6786) */
                int use_reg;
                {
                    T err_sz = 0.0, err_reg = 0.0;
                    int bs = (int)(bx < by ? (bx < bz ? bx : bz) : (by < bz ? by : bz));
                    T ra = reg[b], rb = reg[nb + b], rc = reg[2 * nb + b], rd = reg[3 * nb + b];
                    for (int i = 1; i < bs; i++) {
                        int bmi = bs - i;
                        int jj[4] = {i, i, bmi, bmi}, kk[4] = {i, bmi, i, bmi};
                        for (int q = 0; q < 4; q++) {
                            const T *c = base + (size_t)i * d0 + (size_t)jj[q] * d1 + (size_t)kk[q];
                            T x = *c;
                            T psz = c[-1] + c[-(ptrdiff_t)d1] + c[-(ptrdiff_t)d0] - c[-(ptrdiff_t)d1 - 1] - c[-(ptrdiff_t)d0 - 1]
                                    - c[-(ptrdiff_t)d0 - (ptrdiff_t)d1] + c[-(ptrdiff_t)d0 - (ptrdiff_t)d1 - 1];
                            T preg = ra * i + rb * jj[q] + rc * kk[q] + rd;
                            T e1 = FABS_T(psz - x) + noise;
                            if (use_mean) { T e2 = FABS_T(mean - x); err_sz += (e1 < e2 ? e1 : e2); }
                            else err_sz += e1;
                            err_reg += FABS_T(preg - x);
                        }
                    }
                    use_reg = (err_reg < err_sz);
                }

                if (use_reg) {
                    /* coefficient chain (sz_float.c:7126-7152 / :6790-6812) */
                    for (int e = 0; e < 4; e++) {
                        T cur = reg[(size_t)e * nb + b];
                        T diff = cur - last[e];
                        T itv;
#if IS_F64
                        itv = FABS_T(diff) * rprec[e] + 1;            /* sz_double.c: reciprocal in both branches */
#else
                        if (use_mean) itv = FABS_T(diff) * rprec[e] + 1; /* sz_float.c:6795 */
                        else itv = FABS_T(diff) / prec[e] + 1;           /* sz_float.c:7133 */
#endif
                        int cc = 0;
                        if (itv < ccap) {
                            if (diff < 0) itv = -itv;
                            cc = (int)(itv / 2) + cradius;
                            last[e] = last[e] + 2 * (cc - cradius) * prec[e];
                            if (FABS_T(cur - last[e]) > prec[e]) {
                                cc = 0; last[e] = cur; cunpred[(size_t)e * nb + cunpred_n[e]++] = cur;
                            }
                        } else { cc = 0; last[e] = cur; cunpred[(size_t)e * nb + cunpred_n[e]++] = cur; }
                        ccodes[(size_t)e * nb + reg_count] = cc;
                        cdec[(size_t)e * nb + reg_count] = last[e];
                    }
                    size_t idx = 0;
                    for (size_t ii = 0; ii < bx; ii++) for (size_t jj = 0; jj < by; jj++) for (size_t kk = 0; kk < bz; kk++, idx++) {
                        T x = base[ii * d0 + jj * d1 + kk];
                        T pred = last[0] * ii + last[1] * jj + last[2] * kk + last[3];
                        T rc;
                        int code = FN(szo_quant_point)(x, pred, eb, recip, cap, radius, &rc);
                        if (!code) unpred[total_unpred++] = x;
                        type[idx] = code;
                        pb[ii * s0 + jj * s1 + kk] = rc;
                    }
                    reg_count++;
                } else {
                    size_t idx = 0;
                    for (size_t ii = 0; ii < bx; ii++) for (size_t jj = 0; jj < by; jj++) for (size_t kk = 0; kk < bz; kk++, idx++) {
                        T x = base[ii * d0 + jj * d1 + kk];
                        T *c = pb + ii * s0 + jj * s1 + kk;
                        if (use_mean && FABS_T(x - mean) <= eb) { type[idx] = radius; *c = mean; continue; }
                        T pred = c[-1] + c[-(ptrdiff_t)s1] + c[-(ptrdiff_t)s0] - c[-(ptrdiff_t)s1 - 1] - c[-(ptrdiff_t)s0 - 1]
                                 - c[-(ptrdiff_t)s0 - (ptrdiff_t)s1] + c[-(ptrdiff_t)s0 - (ptrdiff_t)s1 - 1];
                        T rc;
                        int code = FN(szo_quant_point)(x, pred, eb, recip, cap_sz, radius, &rc);
                        if (!code) unpred[total_unpred++] = x;
                        else if (use_mean && code <= radius) code -= 1; /* sz_float.c:6944 */
                        type[idx] = code;
                        *c = rc;
                    }
                    indicator[b] = 1;
                }
            }
        }
        /* carry the strip's last plane down to plane 0 for the next strip */
        memcpy(strip, strip + bx * s0, s0 * sizeof(T));
    }
    free(strip);

    /* Huffman + stream (sz_float.c:7379-7473) */
    szo_huff *h = szo_huff_from_symbols(2 * (int)intervals, codes, ne);
    size_t node_count = szo_huff_node_count(h);
    unsigned char *tree = NULL;
    size_t tree_bytes = szo_huff_tree_to_bytes(h, &tree);

    size_t cap_bytes = meta_len + 8 + 4 + sizeof(T) + 4 + 4 + 4 + 5 * tree_bytes + 1 + sizeof(T) + nb / 8 + 1
                       + 4 * (sizeof(T) + 12 + 8 + 4) + nb * 4 * (sizeof(int) + sizeof(T)) + 4 * 13 * 131072
                       + 8 + total_unpred * sizeof(T) + ne * 8 + 64;
    unsigned char *out = (unsigned char *)calloc(cap_bytes, 1);
    unsigned char *q = out;
    memcpy(q, meta, meta_len); q += meta_len;
    szo_put_u64be(q, ne); q += 8;
    szo_put_u32be(q, (uint32_t)block_size); q += 4;
    FN(szo_put_be)(q, eb); q += sizeof(T);
    szo_put_u32be(q, intervals); q += 4;
    szo_put_u32be(q, (uint32_t)tree_bytes); q += 4;
    szo_put_u32be(q, (uint32_t)node_count); q += 4;
    memcpy(q, tree, tree_bytes); q += tree_bytes;
    *q++ = (unsigned char)use_mean;
    memcpy(q, &mean, sizeof(T)); q += sizeof(T);
    {
        size_t nbytes = (nb + 7) / 8;
        for (size_t i = 0; i < nb; i++) if (indicator[i] == 1) q[i >> 3] |= (unsigned char)(1u << (7 - (i & 7)));
        q += nbytes;
    }
    size_t coeff_tree_bytes = 0;
    if (reg_count > 0) {
        for (int e = 0; e < 4; e++) {
            szo_huff *ch = szo_huff_from_symbols(2 * ccap, ccodes + (size_t)e * nb, reg_count);
            size_t cnc = szo_huff_node_count(ch);
            unsigned char *ctree = NULL;
            size_t ctb = szo_huff_tree_to_bytes(ch, &ctree);
            coeff_tree_bytes += ctb;
            FN(szo_put_be)(q, prec[e]); q += sizeof(T);
            szo_put_u32be(q, (uint32_t)cradius); q += 4;
            szo_put_u32be(q, (uint32_t)ctb); q += 4;
            szo_put_u32be(q, (uint32_t)cnc); q += 4;
            memcpy(q, ctree, ctb); q += ctb;
            free(ctree);
            size_t enc = szo_huff_encode(ch, ccodes + (size_t)e * nb, reg_count, q + 8);
            szo_put_u64be(q, enc); q += 8 + enc;
            szo_put_u32be(q, (uint32_t)cunpred_n[e]); q += 4;
            memcpy(q, cunpred + (size_t)e * nb, cunpred_n[e] * sizeof(T)); q += cunpred_n[e] * sizeof(T);
            szo_huff_free(ch);
        }
    }
    memcpy(q, &total_unpred, 8); q += 8;
    memcpy(q, unpred, total_unpred * sizeof(T)); q += total_unpred * sizeof(T);
    size_t huff_bytes = szo_huff_encode(h, codes, ne, q);
    q += huff_bytes;
    *out_size = (size_t)(q - out);

    if (st) {
        memset(st, 0, sizeof(*st));
        st->num_elements = ne; st->num_blocks = nb; st->reg_count = reg_count; st->total_unpred = total_unpred;
        st->intervals = intervals; st->use_mean = use_mean; st->mean = (double)mean; st->eb = (double)eb;
        st->dense_pos = (double)dense_pos; st->mean_freq = (double)mean_freq; st->sample_freq = (double)sample_freq;
        st->codes = codes; codes = NULL;
        st->indicator = indicator; indicator = NULL;
        st->unpred = unpred; unpred = NULL;
        st->reg_params = reg; reg = NULL;
        st->coeff_codes = (int *)malloc((reg_count ? reg_count : 1) * 4 * sizeof(int));
        st->coeff_dec = malloc((reg_count ? reg_count : 1) * 4 * sizeof(T));
        for (int e = 0; e < 4; e++) {
            memcpy(st->coeff_codes + (size_t)e * reg_count, ccodes + (size_t)e * nb, reg_count * sizeof(int));
            memcpy((T *)st->coeff_dec + (size_t)e * reg_count, cdec + (size_t)e * nb, reg_count * sizeof(T));
            st->coeff_unpred_count[e] = cunpred_n[e];
            st->coeff_unpred[e] = malloc((cunpred_n[e] ? cunpred_n[e] : 1) * sizeof(T));
            memcpy(st->coeff_unpred[e], cunpred + (size_t)e * nb, cunpred_n[e] * sizeof(T));
        }
        st->code_len = (unsigned char *)malloc(2 * (size_t)intervals);
        memcpy(st->code_len, h->len, 2 * (size_t)intervals);
        st->tree_bytes = tree_bytes; st->node_count = node_count; st->huff_bytes = huff_bytes;
    }
    free(tree); szo_huff_free(h);
    free(codes); free(unpred); free(indicator); free(ccodes); free(cunpred); free(cdec); free(reg);
    (void)coeff_tree_bytes;
    return out;
}

/* ---- SZ2.1 3-D decompressor (szd_float.c:3483-5866); `ra` points just after the element count ---- */
static int FN(szo_sz21_decompress_3d)(T *out, size_t r1, size_t r2, size_t r3, const unsigned char *ra)
{
    const unsigned char *q = ra;
    size_t block_size = szo_get_u32be(q); q += 4;
    szo_grid gx = szo_make_grid(r1, block_size), gy = szo_make_grid(r2, block_size), gz = szo_make_grid(r3, block_size);
    const size_t nb = gx.num * gy.num * gz.num, ne = r1 * r2 * r3;
    const size_t d0 = r2 * r3, d1 = r3;
    T eb = FN(szo_get_be)(q); q += sizeof(T);
    unsigned intervals = szo_get_u32be(q); q += 4;
    unsigned tree_size = szo_get_u32be(q); q += 4;
    int node_count = (int)szo_get_u32be(q); q += 4;
    szo_huff *h = szo_huff_tree_from_bytes(2 * (int)intervals, q, node_count);
    q += tree_size;
    unsigned char use_mean = *q++;
    T mean; memcpy(&mean, q, sizeof(T)); q += sizeof(T);
    size_t ind_bytes = (nb - 1) / 8 + 1;
    unsigned char *indicator = (unsigned char *)malloc(nb);
    size_t reg_count = 0;
    for (size_t i = 0; i < nb; i++) { indicator[i] = (q[i >> 3] >> (7 - (i & 7))) & 1; if (!indicator[i]) reg_count++; }
    q += ind_bytes;

    int *ccodes[4] = {0, 0, 0, 0}; int cradius[4] = {0, 0, 0, 0}; T prec[4] = {0, 0, 0, 0};
    const unsigned char *cunpred[4] = {0, 0, 0, 0};
    if (reg_count > 0) {
        for (int e = 0; e < 4; e++) {
            prec[e] = FN(szo_get_be)(q); q += sizeof(T);
            cradius[e] = (int)szo_get_u32be(q); q += 4;
            unsigned ts = szo_get_u32be(q); q += 4;
            int cnc = (int)szo_get_u32be(q); q += 4;
            szo_huff *ch = szo_huff_tree_from_bytes(4 * cradius[e], q, cnc);
            q += ts;
            size_t enc = (size_t)szo_get_u64be(q); q += 8;
            ccodes[e] = (int *)malloc(reg_count * sizeof(int));
            szo_huff_decode(ch, q, reg_count, ccodes[e]);
            q += enc;
            unsigned cu = szo_get_u32be(q); q += 4;
            cunpred[e] = q; q += (size_t)cu * sizeof(T);
            szo_huff_free(ch);
        }
    }
    size_t total_unpred; memcpy(&total_unpred, q, 8); q += 8;
    const unsigned char *unpred = q; q += total_unpred * sizeof(T);
    int *codes = (int *)malloc(ne * sizeof(int));
    szo_huff_decode(h, q, ne, codes);
    szo_huff_free(h);

    const int radius = (int)intervals / 2;
    T last[4] = {0, 0, 0, 0}; size_t cu_n[4] = {0, 0, 0, 0}; size_t cidx = 0, un = 0;
    size_t b = 0;
    for (size_t bi = 0; bi < gx.num; bi++) {
        const size_t bx = szo_blk_size(&gx, bi), ox = szo_blk_start(&gx, bi);
        for (size_t bj = 0; bj < gy.num; bj++) {
            const size_t by = szo_blk_size(&gy, bj), oy = szo_blk_start(&gy, bj);
            for (size_t bk = 0; bk < gz.num; bk++, b++) {
                const size_t bz = szo_blk_size(&gz, bk), oz = szo_blk_start(&gz, bk);
                T *base = out + ox * d0 + oy * d1 + oz;
                const int *type = codes + ox * d0 + oy * bx * d1 + bx * by * oz;
                size_t idx = 0;
                if (indicator[b]) {
                    for (size_t ii = 0; ii < bx; ii++) for (size_t jj = 0; jj < by; jj++) for (size_t kk = 0; kk < bz; kk++, idx++) {
                        T *c = base + ii * d0 + jj * d1 + kk;
                        int t = type[idx];
                        if (use_mean && t == radius) { *c = mean; continue; }
                        if (t == 0) { memcpy(c, unpred + (un++) * sizeof(T), sizeof(T)); continue; }
                        if (use_mean && t < radius) t += 1;
                        /* neighbours outside the array are absent terms (szd_float.c:3751-5855);
                         * adding exact zeros instead gives the same value */
                        int hi = (ox + ii) > 0, hj = (oy + jj) > 0, hk = (oz + kk) > 0;
                        T n001 = hk ? c[-1] : 0, n010 = hj ? c[-(ptrdiff_t)d1] : 0, n100 = hi ? c[-(ptrdiff_t)d0] : 0;
                        T n011 = (hj && hk) ? c[-(ptrdiff_t)d1 - 1] : 0, n101 = (hi && hk) ? c[-(ptrdiff_t)d0 - 1] : 0;
                        T n110 = (hi && hj) ? c[-(ptrdiff_t)d0 - (ptrdiff_t)d1] : 0;
                        T n111 = (hi && hj && hk) ? c[-(ptrdiff_t)d0 - (ptrdiff_t)d1 - 1] : 0;
                        T pred = n001 + n010 + n100 - n011 - n101 - n110 + n111;
                        *c = pred + 2 * (t - radius) * eb;
                    }
                } else {
                    for (int e = 0; e < 4; e++) { /* szd_float.c:5809-5820 */
                        int t = ccodes[e][cidx];
                        if (t != 0) last[e] = last[e] + 2 * (t - cradius[e]) * prec[e];
                        else { memcpy(&last[e], cunpred[e] + (cu_n[e]++) * sizeof(T), sizeof(T)); }
                    }
                    cidx++;
                    for (size_t ii = 0; ii < bx; ii++) for (size_t jj = 0; jj < by; jj++) for (size_t kk = 0; kk < bz; kk++, idx++) {
                        T *c = base + ii * d0 + jj * d1 + kk;
                        int t = type[idx];
                        if (t != 0) {
                            T pred = last[0] * ii + last[1] * jj + last[2] * kk + last[3];
                            *c = pred + 2 * (t - radius) * eb;
                        } else memcpy(c, unpred + (un++) * sizeof(T), sizeof(T));
                    }
                }
            }
        }
    }
    for (int e = 0; e < 4; e++) free(ccodes[e]);
    free(codes); free(indicator);
    return 0;
}



/* ==================================================================================================================
 * 2-D
 * ================================================================================================================== */

/* ---- interval optimiser, 2-D (sz_float.c:5405-5515 / sz_double.c:4790-4899) ---- */
static unsigned FN(szo_optimize_intervals_2d)(const szo_params *p, const T *data, size_t r1, size_t r2,
                                              double ebD, T *dense_pos, T *max_freq, T *mean_freq)
{
    T mean = 0.0;
    size_t len = r1 * r2;
    size_t mean_distance = (size_t)(int)sqrt((double)len);
    size_t mean_count = 0;
    for (size_t pos = 0; pos < len; pos += mean_distance) { mean += data[pos]; mean_count++; }
    if (mean_count > 0) mean /= mean_count;

    const size_t range = 8192, radius = 4096;
    size_t *freq_iv = (size_t *)calloc(range, sizeof(size_t));
    unsigned maxRangeRadius = p->max_quant_intervals / 2;
    int sd = p->sample_distance;
    T predThreshold = p->pred_threshold;
    size_t *iv = (size_t *)calloc(maxRangeRadius, sizeof(size_t));
    size_t freq_count = 0, sample_count = 0;

    size_t n1 = 1, oc = (size_t)(sd - 1);
    size_t pos = r2 + oc;
    while (pos < len) {
        const T *d = data + pos;
        T pred = d[-1] + d[-(ptrdiff_t)r2] - d[-(ptrdiff_t)r2 - 1];
        T pred_err = (T)fabs((double)(T)(pred - *d));
        if ((double)pred_err < ebD) freq_count++;
        size_t ri = (size_t)(((double)pred_err / ebD + 1) / 2);
        if (ri >= maxRangeRadius) ri = maxRangeRadius - 1;
        iv[ri]++;

        T mean_diff = *d - mean;
        ptrdiff_t fi;
        if (mean_diff > 0) fi = (ptrdiff_t)((double)mean_diff / ebD) + (ptrdiff_t)radius;
        else fi = (ptrdiff_t)((double)mean_diff / ebD) - 1 + (ptrdiff_t)radius;
        if (fi <= 0) freq_iv[0]++;
        else if ((size_t)fi >= range) freq_iv[range - 1]++;
        else freq_iv[fi]++;

        oc += (size_t)sd;
        if (oc >= r2) {
            n1++;
            size_t oc2 = n1 % (size_t)sd;
            pos += (r2 + (size_t)sd - oc) + ((size_t)sd - oc2);
            oc = (size_t)sd - oc2;
            if (oc == 0) oc++;
        } else pos += (size_t)sd;
        sample_count++;
    }
    *max_freq = (T)(freq_count * 1.0 / sample_count);

    size_t target = (size_t)(sample_count * predThreshold);
    size_t sum = 0, i;
    for (i = 0; i < maxRangeRadius; i++) { sum += iv[i]; if (sum > target) break; }
    if (i >= maxRangeRadius) i = maxRangeRadius - 1;
    unsigned acc = 2 * (unsigned)(i + 1);
    unsigned pow2 = szo_round_up_pow2(acc);
    if (pow2 < 32) pow2 = 32;

    size_t max_sum = 0, max_index = 0;
    for (size_t q = 1; q < range - 2; q++) {
        size_t s2 = freq_iv[q] + freq_iv[q + 1];
        if (s2 > max_sum) { max_sum = s2; max_index = q; }
    }
    *dense_pos = (T)(mean + ebD * (double)(ptrdiff_t)(max_index + 1 - radius));
    *mean_freq = (T)(max_sum * 1.0 / sample_count);
    free(freq_iv); free(iv);
    return pow2;
}

/* ---- SZ2.1 2-D compressor.  r1 slow, r2 fast (callee convention of sz_float.c:5516) ---- */
static unsigned char *FN(szo_sz21_compress_2d)(const szo_params *p, const unsigned char *meta, size_t meta_len,
                                               const T *data, size_t r1, size_t r2, T eb, size_t *out_size, szo_stages *st)
{
    const T recip = 1 / eb;
    const size_t block_size = 16;
    szo_grid gx = szo_make_grid(r1, block_size), gy = szo_make_grid(r2, block_size);
    const size_t nb = gx.num * gy.num, ne = r1 * r2;
    const size_t d0 = r2;

    /* interval optimiser first (its result only sizes the code book; use_mean is then forced off, sz_float.c:5525-5533,5615) */
    unsigned intervals;
    T dense_pos = 0, sample_freq = -1, mean_freq = 0;
    if (p->quantization_intervals == 0)
        intervals = FN(szo_optimize_intervals_2d)(p, data, r1, r2, (double)eb, &dense_pos, &sample_freq, &mean_freq);
    else intervals = p->quantization_intervals;
    const int use_mean = 0;
    const T mean = 0;

    /* regression fit for every block (sz_float.c:5569-5605); SoA a|b|c */
    T *reg = (T *)malloc(nb * 3 * sizeof(T));
    {
        size_t b = 0;
        for (size_t bi = 0; bi < gx.num; bi++) for (size_t bj = 0; bj < gy.num; bj++, b++) {
            size_t bx = szo_blk_size(&gx, bi), by = szo_blk_size(&gy, bj);
            const T *base = data + szo_blk_start(&gx, bi) * d0 + szo_blk_start(&gy, bj);
            T fx = 0.0, fy = 0.0, f = 0;
            for (size_t i = 0; i < bx; i++) {
                T sum_x = 0;
                const T *row = base + i * d0;
                for (size_t j = 0; j < by; j++) { T c = row[j]; sum_x += c; fy += c * j; }
                fx += sum_x * i;
                f += sum_x;
            }
            T coeff = (T)(1.0 / (double)(bx * by));
            T a = (2 * fx / (bx - 1) - f) * 6 * coeff / (bx + 1);
            T bb = (2 * fy / (by - 1) - f) * 6 * coeff / (by + 1);
            T c = f * coeff - ((bx - 1) * a / 2 + (by - 1) * bb / 2);
            reg[b] = a; reg[nb + b] = bb; reg[2 * nb + b] = c;
        }
    }

    /* coefficient precisions (sz_float.c:5607-5612): rel_param_err = 0.15/3 */
    T rel_param_err = (T)(0.15 / 3);
    T prec[3], rprec[3];
    prec[0] = rel_param_err * eb / gx.late; prec[1] = rel_param_err * eb / gy.late; prec[2] = rel_param_err * eb;
    for (int e = 0; e < 3; e++) rprec[e] = 1 / prec[e];
    (void)rprec;

    int *codes = (int *)calloc(ne, sizeof(int));
    T *unpred = (T *)malloc((ne ? ne : 1) * sizeof(T));
    size_t total_unpred = 0;
    unsigned char *indicator = (unsigned char *)calloc(nb, 1);
    int *ccodes = (int *)malloc(nb * 3 * sizeof(int));
    T *cunpred = (T *)malloc(nb * 3 * sizeof(T));
    T *cdec = (T *)malloc(nb * 3 * sizeof(T));
    size_t cunpred_n[3] = {0, 0, 0};
    size_t reg_count = 0;
    T last[3] = {0, 0, 0};

    const int cap = (int)intervals, radius = cap / 2, cap_sz = cap - 2;
    const int ccap = 65536, cradius = 32768;
    const T noise = (T)(eb * 0.81);                        /* sz_float.c:5672 */

    /* reconstruction strip: row 0 = last row of the previous strip, rows 1..bx = this strip; column 0 stays zero */
    const size_t s0 = r2 + 1;
    T *strip = (T *)calloc((gx.early + 1) * s0, sizeof(T));

    size_t b = 0;
    for (size_t bi = 0; bi < gx.num; bi++) {
        const size_t bx = szo_blk_size(&gx, bi), ox = szo_blk_start(&gx, bi);
        for (size_t bj = 0; bj < gy.num; bj++, b++) {
            const size_t by = szo_blk_size(&gy, bj), oy = szo_blk_start(&gy, bj);
            const T *base = data + ox * d0 + oy;
            int *type = codes + ox * d0 + bx * oy;          /* block after block: `type += bx * by` (sz_float.c:6271) */
            T *pb = strip + s0 + (oy + 1);

            /* predictor selection on original data (sz_float.c:6003-6028).  NOTE the second sample of a pair evaluates the
             * plane at row (i - 1), not i -- kept as the reference has it. */
            int use_reg;
            {
                T err_sz = 0.0, err_reg = 0.0;
                int bs = (int)(bx < by ? bx : by);
                T ra = reg[b], rb = reg[nb + b], rc = reg[2 * nb + b];
                for (int i = 1; i < bs; i++) {
                    const T *c = base + (size_t)i * d0 + (size_t)i;
                    T x = *c;
                    T psz = c[-1] + c[-(ptrdiff_t)d0] - c[-(ptrdiff_t)d0 - 1];
                    T preg = ra * i + rb * i + rc;
                    err_sz += FABS_T(psz - x) + noise;
                    err_reg += FABS_T(preg - x);
                    int bmi = bs - i;
                    c = base + (size_t)i * d0 + (size_t)bmi;
                    x = *c;
                    psz = c[-1] + c[-(ptrdiff_t)d0] - c[-(ptrdiff_t)d0 - 1];
                    preg = ra * (i - 1) + rb * bmi + rc;
                    err_sz += FABS_T(psz - x) + noise;
                    err_reg += FABS_T(preg - x);
                }
                use_reg = (err_reg < err_sz);
            }

            if (use_reg) {
                /* coefficient chain (sz_float.c:6031-6055: division in the float file; sz_double.c:5417: reciprocal) */
                for (int e = 0; e < 3; e++) {
                    T cur = reg[(size_t)e * nb + b];
                    T diff = cur - last[e];
                    T itv;
#if IS_F64
                    itv = FABS_T(diff) * rprec[e] + 1;
#else
                    itv = FABS_T(diff) / prec[e] + 1;
#endif
                    int cc = 0;
                    if (itv < ccap) {
                        if (diff < 0) itv = -itv;
                        cc = (int)(itv / 2) + cradius;
                        last[e] = last[e] + 2 * (cc - cradius) * prec[e];
                        if (FABS_T(cur - last[e]) > prec[e]) {
                            cc = 0; last[e] = cur; cunpred[(size_t)e * nb + cunpred_n[e]++] = cur;
                        }
                    } else { cc = 0; last[e] = cur; cunpred[(size_t)e * nb + cunpred_n[e]++] = cur; }
                    ccodes[(size_t)e * nb + reg_count] = cc;
                    cdec[(size_t)e * nb + reg_count] = last[e];
                }
                size_t idx = 0;
                for (size_t ii = 0; ii < bx; ii++) for (size_t jj = 0; jj < by; jj++, idx++) {
                    T x = base[ii * d0 + jj];
                    T pred = last[0] * ii + last[1] * jj + last[2];
                    T rc;
                    int code = FN(szo_quant_point)(x, pred, eb, recip, cap, radius, &rc);
                    if (!code) unpred[total_unpred++] = x;
                    type[idx] = code;
                    pb[ii * s0 + jj] = rc;
                }
                reg_count++;
            } else {
                size_t idx = 0;
                for (size_t ii = 0; ii < bx; ii++) for (size_t jj = 0; jj < by; jj++, idx++) {
                    T x = base[ii * d0 + jj];
                    T *c = pb + ii * s0 + jj;
                    T pred = c[-1] + c[-(ptrdiff_t)s0] - c[-(ptrdiff_t)s0 - 1];
                    T rc;
                    int code = FN(szo_quant_point)(x, pred, eb, recip, cap_sz, radius, &rc);
                    if (!code) unpred[total_unpred++] = x;
                    type[idx] = code;
                    *c = rc;
                }
                indicator[b] = 1;
            }
        }
        /* carry the strip's last row up to row 0 for the next strip */
        memcpy(strip, strip + bx * s0, s0 * sizeof(T));
    }
    free(strip);

    /* Huffman + stream (sz_float.c:6283-6380) */
    szo_huff *h = szo_huff_from_symbols(2 * (int)intervals, codes, ne);
    size_t node_count = szo_huff_node_count(h);
    unsigned char *tree = NULL;
    size_t tree_bytes = szo_huff_tree_to_bytes(h, &tree);

    size_t cap_bytes = meta_len + 8 + 4 + sizeof(T) + 4 + 4 + 4 + 5 * tree_bytes + 1 + sizeof(T) + nb / 8 + 1
                       + 3 * (sizeof(T) + 12 + 8 + 4) + nb * 3 * (sizeof(int) + sizeof(T)) + 3 * 13 * 131072
                       + 8 + total_unpred * sizeof(T) + ne * 8 + 64;
    unsigned char *out = (unsigned char *)calloc(cap_bytes, 1);
    unsigned char *q = out;
    memcpy(q, meta, meta_len); q += meta_len;
    szo_put_u64be(q, ne); q += 8;
    szo_put_u32be(q, (uint32_t)block_size); q += 4;
    FN(szo_put_be)(q, eb); q += sizeof(T);
    szo_put_u32be(q, intervals); q += 4;
    szo_put_u32be(q, (uint32_t)tree_bytes); q += 4;
    szo_put_u32be(q, (uint32_t)node_count); q += 4;
    memcpy(q, tree, tree_bytes); q += tree_bytes;
    *q++ = (unsigned char)use_mean;
    memcpy(q, &mean, sizeof(T)); q += sizeof(T);
    {
        size_t nbytes = (nb + 7) / 8;
        for (size_t i = 0; i < nb; i++) if (indicator[i] == 1) q[i >> 3] |= (unsigned char)(1u << (7 - (i & 7)));
        q += nbytes;
    }
    if (reg_count > 0) {
        for (int e = 0; e < 3; e++) {
            szo_huff *ch = szo_huff_from_symbols(2 * ccap, ccodes + (size_t)e * nb, reg_count);
            size_t cnc = szo_huff_node_count(ch);
            unsigned char *ctree = NULL;
            size_t ctb = szo_huff_tree_to_bytes(ch, &ctree);
            FN(szo_put_be)(q, prec[e]); q += sizeof(T);
            szo_put_u32be(q, (uint32_t)cradius); q += 4;
            szo_put_u32be(q, (uint32_t)ctb); q += 4;
            szo_put_u32be(q, (uint32_t)cnc); q += 4;
            memcpy(q, ctree, ctb); q += ctb;
            free(ctree);
            size_t enc = szo_huff_encode(ch, ccodes + (size_t)e * nb, reg_count, q + 8);
            szo_put_u64be(q, enc); q += 8 + enc;
            szo_put_u32be(q, (uint32_t)cunpred_n[e]); q += 4;
            memcpy(q, cunpred + (size_t)e * nb, cunpred_n[e] * sizeof(T)); q += cunpred_n[e] * sizeof(T);
            szo_huff_free(ch);
        }
    }
    memcpy(q, &total_unpred, 8); q += 8;
    memcpy(q, unpred, total_unpred * sizeof(T)); q += total_unpred * sizeof(T);
    size_t huff_bytes = szo_huff_encode(h, codes, ne, q);
    q += huff_bytes;
    *out_size = (size_t)(q - out);

    if (st) {
        memset(st, 0, sizeof(*st));
        st->num_elements = ne; st->num_blocks = nb; st->reg_count = reg_count; st->total_unpred = total_unpred;
        st->intervals = intervals; st->use_mean = use_mean; st->mean = (double)mean; st->eb = (double)eb;
        st->dense_pos = (double)dense_pos; st->mean_freq = (double)mean_freq; st->sample_freq = (double)sample_freq;
        st->codes = codes; codes = NULL;
        st->indicator = indicator; indicator = NULL;
        st->unpred = unpred; unpred = NULL;
        st->reg_params = reg; reg = NULL;
        st->coeff_codes = (int *)malloc((reg_count ? reg_count : 1) * 3 * sizeof(int));
        st->coeff_dec = malloc((reg_count ? reg_count : 1) * 3 * sizeof(T));
        for (int e = 0; e < 3; e++) {
            memcpy(st->coeff_codes + (size_t)e * reg_count, ccodes + (size_t)e * nb, reg_count * sizeof(int));
            memcpy((T *)st->coeff_dec + (size_t)e * reg_count, cdec + (size_t)e * nb, reg_count * sizeof(T));
            st->coeff_unpred_count[e] = cunpred_n[e];
            st->coeff_unpred[e] = malloc((cunpred_n[e] ? cunpred_n[e] : 1) * sizeof(T));
            memcpy(st->coeff_unpred[e], cunpred + (size_t)e * nb, cunpred_n[e] * sizeof(T));
        }
        st->code_len = (unsigned char *)malloc(2 * (size_t)intervals);
        memcpy(st->code_len, h->len, 2 * (size_t)intervals);
        st->tree_bytes = tree_bytes; st->node_count = node_count; st->huff_bytes = huff_bytes;
    }
    free(tree); szo_huff_free(h);
    free(codes); free(unpred); free(indicator); free(ccodes); free(cunpred); free(cdec); free(reg);
    return out;
}

/* ---- SZ2.1 2-D decompressor (szd_float.c:3141-3482); `ra` points just after the element count ---- */
static int FN(szo_sz21_decompress_2d)(T *out, size_t r1, size_t r2, const unsigned char *ra)
{
    const unsigned char *q = ra;
    size_t block_size = szo_get_u32be(q); q += 4;
    szo_grid gx = szo_make_grid(r1, block_size), gy = szo_make_grid(r2, block_size);
    const size_t nb = gx.num * gy.num, ne = r1 * r2;
    const size_t d0 = r2;
    T eb = FN(szo_get_be)(q); q += sizeof(T);
    unsigned intervals = szo_get_u32be(q); q += 4;
    unsigned tree_size = szo_get_u32be(q); q += 4;
    int node_count = (int)szo_get_u32be(q); q += 4;
    szo_huff *h = szo_huff_tree_from_bytes(2 * (int)intervals, q, node_count);
    q += tree_size;
    unsigned char use_mean = *q++;
    T mean; memcpy(&mean, q, sizeof(T)); q += sizeof(T);
    size_t ind_bytes = (nb - 1) / 8 + 1;
    unsigned char *indicator = (unsigned char *)malloc(nb);
    size_t reg_count = 0;
    for (size_t i = 0; i < nb; i++) { indicator[i] = (q[i >> 3] >> (7 - (i & 7))) & 1; if (!indicator[i]) reg_count++; }
    q += ind_bytes;

    int *ccodes[3] = {0, 0, 0}; int cradius[3] = {0, 0, 0}; T prec[3] = {0, 0, 0};
    const unsigned char *cunpred[3] = {0, 0, 0};
    if (reg_count > 0) {
        for (int e = 0; e < 3; e++) {
            prec[e] = FN(szo_get_be)(q); q += sizeof(T);
            cradius[e] = (int)szo_get_u32be(q); q += 4;
            unsigned ts = szo_get_u32be(q); q += 4;
            int cnc = (int)szo_get_u32be(q); q += 4;
            szo_huff *ch = szo_huff_tree_from_bytes(4 * cradius[e], q, cnc);
            q += ts;
            size_t enc = (size_t)szo_get_u64be(q); q += 8;
            ccodes[e] = (int *)malloc(reg_count * sizeof(int));
            szo_huff_decode(ch, q, reg_count, ccodes[e]);
            q += enc;
            unsigned cu = szo_get_u32be(q); q += 4;
            cunpred[e] = q; q += (size_t)cu * sizeof(T);
            szo_huff_free(ch);
        }
    }
    size_t total_unpred; memcpy(&total_unpred, q, 8); q += 8;
    const unsigned char *unpred = q; q += total_unpred * sizeof(T);
    int *codes = (int *)malloc(ne * sizeof(int));
    szo_huff_decode(h, q, ne, codes);
    szo_huff_free(h);

    const int radius = (int)intervals / 2;
    T last[3] = {0, 0, 0}; size_t cu_n[3] = {0, 0, 0}; size_t cidx = 0, un = 0;
    size_t b = 0;
    for (size_t bi = 0; bi < gx.num; bi++) {
        const size_t bx = szo_blk_size(&gx, bi), ox = szo_blk_start(&gx, bi);
        for (size_t bj = 0; bj < gy.num; bj++, b++) {
            const size_t by = szo_blk_size(&gy, bj), oy = szo_blk_start(&gy, bj);
            T *base = out + ox * d0 + oy;
            const int *type = codes + ox * d0 + bx * oy;
            size_t idx = 0;
            if (indicator[b]) {
                for (size_t ii = 0; ii < bx; ii++) for (size_t jj = 0; jj < by; jj++, idx++) {
                    T *c = base + ii * d0 + jj;
                    int t = type[idx];
                    if (use_mean && t == radius) { *c = mean; continue; }
                    if (t == 0) { memcpy(c, unpred + (un++) * sizeof(T), sizeof(T)); continue; }
                    if (use_mean && t < radius) t += 1;
                    int hi = (ox + ii) > 0, hj = (oy + jj) > 0;
                    T n01 = hj ? c[-1] : 0, n10 = hi ? c[-(ptrdiff_t)d0] : 0, n11 = (hi && hj) ? c[-(ptrdiff_t)d0 - 1] : 0;
                    T pred = n01 + n10 - n11;
                    *c = pred + 2 * (t - radius) * eb;
                }
            } else {
                for (int e = 0; e < 3; e++) {
                    int t = ccodes[e][cidx];
                    if (t != 0) last[e] = last[e] + 2 * (t - cradius[e]) * prec[e];
                    else { memcpy(&last[e], cunpred[e] + (cu_n[e]++) * sizeof(T), sizeof(T)); }
                }
                cidx++;
                for (size_t ii = 0; ii < bx; ii++) for (size_t jj = 0; jj < by; jj++, idx++) {
                    T *c = base + ii * d0 + jj;
                    int t = type[idx];
                    if (t != 0) {
                        T pred = last[0] * ii + last[1] * jj + last[2];
                        *c = pred + 2 * (t - radius) * eb;
                    } else memcpy(c, unpred + (un++) * sizeof(T), sizeof(T));
                }
            }
        }
    }
    for (int e = 0; e < 3; e++) free(ccodes[e]);
    free(codes); free(indicator);
    return 0;
}

#undef FN
