/* szo_pwr_impl.h -- point-wise relative bounds in their log-domain form, restated per type.
 *   compress:   SZ_compress_args_float_NoCkRngeNoGzip_{1D,2D,3D}_pwr_pre_log, sz/src/sz_float_pwr.c:1791-1975 (doubles:
 *               sz_double_pwr.c:1781-1965); the three differ only in the SZ 1.4 quantiser they call
 *   decompress: decompressDataSeries_float_{1D,2D,3D}_pwr_pre_log, sz/src/szd_float_pwr.c:1353-1422
 * TEST INFRASTRUCTURE (see szo.h).  The sign bytes go through zstd level 3 (sz_lossless_compress, utility.c:174-195); the library
 * is loaded at run time, so its version -- not the reference's bundled one -- decides those bytes: tests compare them decoded. */
#define FN(name) SZO_CAT(name, SUF)
static unsigned char *FN(szo_pwr_compress)(const szo_params *p, const unsigned char *meta, size_t meta_len, const T *ori, size_t r1, size_t r2, size_t r3,
                                           double ratio, T vmin, T vmax, size_t segment_size, size_t *out_size)
{
    const size_t n = r1 * r2 * r3;
    T *log_data = (T *)malloc(n * sizeof(T));
    unsigned char *signs = (unsigned char *)calloc(n, 1);
    T max_abs_log_data;
    if (vmin == 0) max_abs_log_data = (T)fabs(log2(fabs(vmax)));
    else if (vmax == 0) max_abs_log_data = (T)fabs(log2(fabs(vmin)));
    else max_abs_log_data = (T)(fabs(log2(fabs(vmin))) > fabs(log2(fabs(vmax))) ? fabs(log2(fabs(vmin))) : fabs(log2(fabs(vmax))));
    T min_log_data = max_abs_log_data;
    int positive = 1;
    for (size_t i = 0; i < n; i++) {
        if (ori[i] < 0) { signs[i] = 1; log_data[i] = -ori[i]; positive = 0; }
        else log_data[i] = ori[i];
        if (log_data[i] > 0) {
            log_data[i] = (T)log2(log_data[i]);
            if (log_data[i] > max_abs_log_data) max_abs_log_data = log_data[i];
            if (log_data[i] < min_log_data) min_log_data = log_data[i];
        }
    }
    /* computeRangeSize_float on the log array (zeros are still 0): dataCompression.c:102-119 */
    T mn = log_data[0], mx = log_data[0];
    for (size_t i = 1; i < n; i++) { T v = log_data[i]; if (mn > v) mn = v; else if (mx < v) mx = v; }
    const T range = mx - mn, median = (T)(mn + range / 2);
    if (fabs(min_log_data) > max_abs_log_data) max_abs_log_data = (T)fabs(min_log_data);
    const double real_precision = log2(1.0 + ratio) - max_abs_log_data * (IS_F64 ? 2.23e-16 : 1.2e-7);
    for (size_t i = 0; i < n; i++) if (ori[i] == 0) log_data[i] = (T)(min_log_data - 2.0001 * real_precision);
    szo_pwr_extra pw; memset(&pw, 0, sizeof(pw));
    pw.segment_size = segment_size;
    pw.min_log_value = (double)(T)(min_log_data - 1.0001 * real_precision);
    unsigned char *blob = NULL;
    if (!positive) {
        blob = szo_zstd_compress(signs, n, 3, &pw.blob_size);
        if (!blob) { free(log_data); free(signs); return NULL; }
        pw.blob = blob;
    }
    free(signs);
    unsigned char *out = FN(szo_sz14_compress_3d)(p, meta, meta_len, log_data, r1, r2, r3, (T)real_precision, range, median, out_size, NULL, &pw);
    free(log_data); free(blob);
    return out;
}

static int FN(szo_pwr_decompress)(T *out, size_t r1, size_t r2, size_t r3, const unsigned char *b, size_t avail)
{
    const size_t n = r1 * r2 * r3;
    szo_pwr_extra pw; memset(&pw, 0, sizeof(pw));
    if (FN(szo_sz14_decompress_3d)(out, r1, r2, r3, b, avail, &pw)) return -1;
    const T threshold = (T)pw.min_log_value;
    unsigned char *signs = NULL;
    if (pw.blob_size > 0) { signs = szo_zstd_decompress(pw.blob, pw.blob_size, n); if (!signs) return -1; }
    for (size_t i = 0; i < n; i++) {
        if (out[i] < threshold) out[i] = 0; else out[i] = (T)exp2(out[i]);
        if (signs && signs[i]) out[i] = -out[i];
    }
    free(signs);
    return 0;
}
#undef FN
