/* szo_omp_impl.h -- CPU restatement (TEST INFRASTRUCTURE ONLY) of the reference's OpenMP container for 3-D arrays: the array is cut into
 * thread_num boxes, every box is quantised on its own (no value crosses a box face), ONE Huffman code book covers all boxes and every box
 * has its own byte-aligned payload.  Included twice by szo_api.c (T = float / double).  Follows
 *   container, box grid       sz/src/sz_omp.c:63-358   SZ_compress_float_3D_MDQ_openmp      (double: :578-863)
 *   box quantiser             sz/src/sz_float.c:4704-5012 SZ_compress_float_3D_MDQ_RA_block  (double: sz_double.c, same name)
 *   inverse                   sz/src/sz_omp.c:366-566  decompressDataSeries_float_3D_openmp, szd_float.c decompressDataSeries_float_3D_RA_block
 *   box grid macro            sz/include/sz.h:117 SZ_COMPUTE_BLOCKCOUNT
 * The stream's first 4 + MetaDataByteLength bytes (version, flag byte, parameter bytes: dataCompression.c:686) depend on the library's
 * configuration state, not on this function's arguments: the caller passes them in (`meta`), as szhip.h's entry points do.
 * Parity: pinned against oracle/_ref/libSZ_omp.so (tests/test_omp_container.py); "parity unpinned" where that library is absent.
 * Why it is here: this is the reference's own parallel mode -- thousands of independent boxes instead of one dependency front -- and the
 * next row of the hot path; the HIP side is sz_amd/csrc/szh_omp.h, checked against this file by tests/test_zz_omp_hip.py. */

#ifndef SZO_CAT
#define SZO_CAT_(a, b) a##_##b
#define SZO_CAT(a, b) SZO_CAT_(a, b)
#endif
#define FN(name) SZO_CAT(name, SUF)

/* thread_num -> box grid (sz_omp.c:88-117): the exponent of two is spread over the three dimensions, dim 0 first */
#ifndef SZO_OMP_GRID
#define SZO_OMP_GRID
static void szo_omp_grid(int thread_num, size_t *nx, size_t *ny, size_t *nz, int *threads_used)
{
    int order = 0; while ((2 << order) <= thread_num) ++order;          /* (int)log2(thread_num) */
    const int b = order / 3;
    size_t x, y;
    switch (order % 3) { case 0: x = (size_t)1 << b; y = (size_t)1 << b; break; case 1: x = (size_t)1 << (b + 1); y = (size_t)1 << b; break; default: x = (size_t)1 << (b + 1); y = (size_t)1 << (b + 1); }
    *nx = x; *ny = y; *nz = (size_t)thread_num / (x * y);
    *threads_used = (int)(*nx * *ny * *nz);
}
/* SZ_COMPUTE_BLOCKCOUNT: `count` cells into `num` boxes: the first `split` boxes have `early` cells, the others `late` */
static void szo_omp_blockcount(size_t count, size_t num, size_t *split, size_t *early, size_t *late)
{
    *split = count % num; *late = count / num; *early = *split ? *late + 1 : *late;
}
static size_t szo_omp_off(size_t i, size_t split, size_t early, size_t late) { return i < split ? i * early : i * late + split; }
static void szo_put_be32(unsigned char *p, unsigned v) { p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v; }
static unsigned szo_get_be32(const unsigned char *p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3]; }
#endif

/* one box (sz_float.c:4704-5012): `rec` receives the reconstruction (the reference keeps two planes of it), `type` the box's codes in its own
 * row-major order, `unpred` the values the quantiser could not take, verbatim.  Returns their number.  Predictors by position, all from
 * reconstructed values, float / double arithmetic left to right:
 *   plane 0: (0,0) the box's first value itself ("mean"), (0,1) its left neighbour, (0,j>=2) 2 left - left-left, (i>=1,0) the value above,
 *            elsewhere left + above - above-left;
 *   plane k>=1: (0,0) the value in the previous plane, first row / first column the 2-D form with the previous plane, elsewhere the 7-point form. */
static size_t FN(szo_omp_box)(const T *box, size_t d0, size_t d1, size_t b0, size_t b1, size_t b2, T eb, int intervals, int *type, T *unpred, T *rec)
{
    const T recip = 1 / eb;
    const int radius = intervals / 2;
    size_t nun = 0;
    const size_t p1 = b2, p0 = b1 * b2;             /* pitches of rec / type */
    for (size_t k = 0; k < b0; k++) for (size_t i = 0; i < b1; i++) for (size_t j = 0; j < b2; j++) {
        const size_t at = k * p0 + i * p1 + j;
        const T cur = box[k * d0 + i * d1 + j];
        T pred;
        if (k == 0) {
            if (i == 0) pred = j == 0 ? box[0] : j == 1 ? rec[at - 1] : 2 * rec[at - 1] - rec[at - 2];
            else pred = j == 0 ? rec[at - p1] : rec[at - 1] + rec[at - p1] - rec[at - p1 - 1];
        } else {
            if (i == 0) pred = j == 0 ? rec[at - p0] : rec[at - 1] + rec[at - p0] - rec[at - p0 - 1];
            else if (j == 0) pred = rec[at - p1] + rec[at - p0] - rec[at - p0 - p1];
            else pred = rec[at - 1] + rec[at - p1] + rec[at - p0] - rec[at - p1 - 1] - rec[at - p0 - p1] - rec[at - p0 - 1] + rec[at - p0 - p1 - 1];
        }
        const T diff = cur - pred;
        T itv = FABS_T(diff) * recip + 1;
        int t = 0;
        if (itv < intervals) {
            if (diff < 0) itv = -itv;
            t = (int)(itv / 2) + radius;
            rec[at] = pred + 2 * (t - radius) * eb;
            if (FABS_T(cur - rec[at]) > eb) t = 0;
        }
        if (t == 0) { rec[at] = cur; unpred[nun++] = cur; }
        type[at] = t;
    }
    return nun;
}
static void FN(szo_omp_unbox)(T *box, size_t d0, size_t d1, size_t b0, size_t b1, size_t b2, T first, T eb, int intervals, const int *type, const T *unpred, T *rec)
{
    const int radius = intervals / 2;
    size_t nun = 0;
    const size_t p1 = b2, p0 = b1 * b2;
    for (size_t k = 0; k < b0; k++) for (size_t i = 0; i < b1; i++) for (size_t j = 0; j < b2; j++) {
        const size_t at = k * p0 + i * p1 + j;
        T pred;
        if (k == 0) {
            if (i == 0) pred = j == 0 ? first : j == 1 ? rec[at - 1] : 2 * rec[at - 1] - rec[at - 2];
            else pred = j == 0 ? rec[at - p1] : rec[at - 1] + rec[at - p1] - rec[at - p1 - 1];
        } else {
            if (i == 0) pred = j == 0 ? rec[at - p0] : rec[at - 1] + rec[at - p0] - rec[at - p0 - 1];
            else if (j == 0) pred = rec[at - p1] + rec[at - p0] - rec[at - p0 - p1];
            else pred = rec[at - 1] + rec[at - p1] + rec[at - p0] - rec[at - p1 - 1] - rec[at - p0 - p1] - rec[at - p0 - 1] + rec[at - p0 - p1 - 1];
        }
        rec[at] = type[at] ? pred + 2 * (type[at] - radius) * eb : unpred[nun++];
        box[k * d0 + i * d1 + j] = rec[at];
    }
}

/* the container.  intervals: 0 = the interval optimiser of the SZ 1.4 path (optimize_intervals_*_3D_opt, szo_sz14_impl.h) */
static unsigned char *FN(szo_omp_compress)(const szo_params *p, const T *data, size_t r1, size_t r2, size_t r3, T eb, int thread_num,
                                           const unsigned char *meta, size_t meta_len, size_t *out_size)
{
    unsigned intervals = p->quantization_intervals ? p->quantization_intervals : FN(szo_optimize_intervals_3d_opt)(p, data, r1, r2, r3, (double)eb);
    size_t nx, ny, nz; int nt;
    szo_omp_grid(thread_num, &nx, &ny, &nz, &nt);
    size_t sx, ex, lx, sy, ey, ly, sz_, ez, lz;
    szo_omp_blockcount(r1, nx, &sx, &ex, &lx); szo_omp_blockcount(r2, ny, &sy, &ey, &ly); szo_omp_blockcount(r3, nz, &sz_, &ez, &lz);
    const size_t n = r1 * r2 * r3, nb = nx * ny * nz, d0 = r2 * r3, d1 = r3, maxbox = ex * ey * ez;
    int *type = (int *)calloc(n ? n : 1, sizeof(int));                  /* (the reference leaves gaps of an uneven grid uninitialised) */
    T *unpred = (T *)malloc((maxbox * nb ? maxbox * nb : 1) * sizeof(T)), *rec = (T *)malloc((maxbox ? maxbox : 1) * sizeof(T));
    unsigned *ucount = (unsigned *)calloc(nb, sizeof(unsigned));
    T *first = (T *)calloc(nb, sizeof(T));
    size_t *toff = (size_t *)calloc(nb, sizeof(size_t)), *bel = (size_t *)calloc(nb, sizeof(size_t));
    for (size_t id = 0; id < nb; id++) {
        const size_t i = id / (ny * nz), j = (id % (ny * nz)) / nz, k = id % nz;
        const size_t ox = szo_omp_off(i, sx, ex, lx), oy = szo_omp_off(j, sy, ey, ly), oz = szo_omp_off(k, sz_, ez, lz);
        const size_t cx = i < sx ? ex : lx, cy = j < sy ? ey : ly, cz = k < sz_ ? ez : lz;
        toff[id] = ox * d0 + oy * cx * d1 + oz * cx * cy; bel[id] = cx * cy * cz;             /* sz_omp.c:179 */
        const T *box = data + ox * d0 + oy * d1 + oz;
        first[id] = box[0];
        ucount[id] = (unsigned)FN(szo_omp_box)(box, d0, d1, cx, cy, cz, eb, (int)intervals, type + toff[id], unpred + id * maxbox, rec);
    }
    szo_huff *h = szo_huff_from_symbols(2 * (int)intervals, type, n);
    unsigned char *tree = NULL;
    const size_t tree_len = szo_huff_tree_to_bytes(h, &tree), nodes = szo_huff_node_count(h);
    size_t total_un = 0; for (size_t id = 0; id < nb; id++) total_un += ucount[id];
    unsigned char *out = (unsigned char *)malloc(meta_len + 16 + sizeof(T) + tree_len + nb * (4 + sizeof(T) + 8) + total_un * sizeof(T) + n * sizeof(int) + 64);
    unsigned char *q = out;
    memcpy(q, meta, meta_len); q += meta_len;
    szo_put_be32(q, (unsigned)nt); q += 4;
    { unsigned char e[sizeof(T)]; memcpy(e, &eb, sizeof(T)); for (size_t b = 0; b < sizeof(T); b++) q[b] = e[sizeof(T) - 1 - b]; q += sizeof(T); }     /* floatToBytes / doubleToBytes: big-endian */
    szo_put_be32(q, intervals); q += 4;
    szo_put_be32(q, (unsigned)tree_len); q += 4;
    szo_put_be32(q, (unsigned)nodes); q += 4;
    memcpy(q, tree, tree_len); q += tree_len;
    memcpy(q, ucount, nb * sizeof(unsigned)); q += nb * sizeof(unsigned);
    memcpy(q, first, nb * sizeof(T)); q += nb * sizeof(T);
    for (size_t id = 0; id < nb; id++) { memcpy(q, unpred + id * maxbox, ucount[id] * sizeof(T)); q += ucount[id] * sizeof(T); }
    unsigned char *sizes = q; q += nb * sizeof(size_t);
    for (size_t id = 0; id < nb; id++) {
        const size_t len = szo_huff_encode(h, type + toff[id], bel[id], q);
        memcpy(sizes + id * sizeof(size_t), &len, sizeof(size_t));
        q += len;
    }
    *out_size = (size_t)(q - out);
    szo_huff_free(h); free(tree); free(type); free(unpred); free(rec); free(ucount); free(first); free(toff); free(bel);
    return out;
}
/* `bytes`: the stream WITHOUT its first 4 + MetaDataByteLength bytes (what decompressDataSeries_*_3D_openmp is handed) */
static T *FN(szo_omp_decompress)(const unsigned char *bytes, size_t r1, size_t r2, size_t r3)
{
    const unsigned char *q = bytes;
    const int thread_num = (int)szo_get_be32(q); q += 4;
    T eb; { unsigned char e[sizeof(T)]; for (size_t b = 0; b < sizeof(T); b++) e[sizeof(T) - 1 - b] = q[b]; memcpy(&eb, e, sizeof(T)); q += sizeof(T); }
    const unsigned intervals = szo_get_be32(q); q += 4;
    const unsigned tree_len = szo_get_be32(q); q += 4;
    const unsigned nodes = szo_get_be32(q); q += 4;
    szo_huff *h = szo_huff_tree_from_bytes(2 * (int)intervals, q, (int)nodes); q += tree_len;
    size_t nx, ny, nz; int nt;
    szo_omp_grid(thread_num, &nx, &ny, &nz, &nt);
    size_t sx, ex, lx, sy, ey, ly, sz_, ez, lz;
    szo_omp_blockcount(r1, nx, &sx, &ex, &lx); szo_omp_blockcount(r2, ny, &sy, &ey, &ly); szo_omp_blockcount(r3, nz, &sz_, &ez, &lz);
    const size_t n = r1 * r2 * r3, nb = nx * ny * nz, d0 = r2 * r3, d1 = r3, maxbox = ex * ey * ez;
    const unsigned char *ucount = q; q += nb * sizeof(unsigned);
    const unsigned char *first = q; q += nb * sizeof(T);
    const unsigned char *un = q;
    size_t total_un = 0; for (size_t id = 0; id < nb; id++) { unsigned c; memcpy(&c, ucount + id * 4, 4); total_un += c; }
    q += total_un * sizeof(T);
    const unsigned char *sizes = q; q += nb * sizeof(size_t);
    T *out = (T *)malloc((n ? n : 1) * sizeof(T)), *rec = (T *)malloc((maxbox ? maxbox : 1) * sizeof(T)), *ub = (T *)malloc((maxbox ? maxbox : 1) * sizeof(T));
    int *type = (int *)malloc((maxbox ? maxbox : 1) * sizeof(int));
    size_t uoff = 0;
    for (size_t id = 0; id < nb; id++) {
        const size_t i = id / (ny * nz), j = (id % (ny * nz)) / nz, k = id % nz;
        const size_t ox = szo_omp_off(i, sx, ex, lx), oy = szo_omp_off(j, sy, ey, ly), oz = szo_omp_off(k, sz_, ez, lz);
        const size_t cx = i < sx ? ex : lx, cy = j < sy ? ey : ly, cz = k < sz_ ? ez : lz;
        size_t len; memcpy(&len, sizes + id * sizeof(size_t), sizeof(size_t));
        unsigned c; memcpy(&c, ucount + id * 4, 4);
        T f; memcpy(&f, first + id * sizeof(T), sizeof(T));
        szo_huff_decode(h, q, cx * cy * cz, type); q += len;
        memcpy(ub, un + uoff * sizeof(T), (size_t)c * sizeof(T)); uoff += c;
        FN(szo_omp_unbox)(out + ox * d0 + oy * d1 + oz, d0, d1, cx, cy, cz, f, eb, (int)intervals, type, ub, rec);
    }
    szo_huff_free(h); free(rec); free(ub); free(type);
    return out;
}
#undef FN
