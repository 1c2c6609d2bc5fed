/* sz_slab.c -- the slab container of the multi-GPU path for C callers (include/sz_slab.h; the byte layout and the cut rule are
 * those of sz_amd/slab.py).  Host C over the public SZ_* entry points: nothing here touches the device itself. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "sz.h"
#include "sz_slab.h"

static void put_u32(unsigned char *p, uint32_t v) { for (int i = 0; i < 4; i++) p[i] = (unsigned char)(v >> (8 * i)); }
static void put_u64(unsigned char *p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (unsigned char)(v >> (8 * i)); }
static uint32_t get_u32(const unsigned char *p) { uint32_t v = 0; for (int i = 0; i < 4; i++) v |= (uint32_t)p[i] << (8 * i); return v; }
static uint64_t get_u64(const unsigned char *p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i); return v; }

void sz_slab_bounds(size_t n0, int slabs, int block, size_t *bounds)
{
    if (slabs < 1 || !bounds) return;
    if (block < 1) block = 1;
    const size_t units = n0 / (size_t)block;
    size_t z = 0;
    if (units < (size_t)slabs) {                       /* fewer blocks than slabs: cut plane-wise */
        const size_t base = n0 / (size_t)slabs, rem = n0 % (size_t)slabs;
        for (int r = 0; r < slabs; r++) { const size_t h = base + ((size_t)r < rem ? 1 : 0); bounds[2 * r] = z; bounds[2 * r + 1] = z + h; z += h; }
        return;
    }
    const size_t base = units / (size_t)slabs, rem = units % (size_t)slabs;
    for (int r = 0; r < slabs; r++) {
        size_t h = (base + ((size_t)r < rem ? 1 : 0)) * (size_t)block;
        if (r == slabs - 1) h = n0 - z;                /* the planes that do not fill a block go to the last slab */
        bounds[2 * r] = z; bounds[2 * r + 1] = z + h; z += h;
    }
}

unsigned char *sz_slab_pack(int dataType, const size_t dims[3], int slabs, const size_t *bounds, const unsigned char *const *streams,
                            const size_t *stream_bytes, size_t *outSize)
{
    if (!dims || slabs < 1 || !bounds || !streams || !stream_bytes || !outSize || (dataType != SZ_FLOAT && dataType != SZ_DOUBLE)) return NULL;
    size_t total = 40 + 24 * (size_t)slabs;
    for (int r = 0; r < slabs; r++) { if (!streams[r]) return NULL; total += stream_bytes[r]; }
    unsigned char *o = (unsigned char *)malloc(total);
    if (!o) return NULL;
    memcpy(o, "SZSL", 4);
    put_u32(o + 4, 1); put_u32(o + 8, (uint32_t)slabs); put_u32(o + 12, dataType == SZ_FLOAT ? 0u : 1u);
    for (int i = 0; i < 3; i++) put_u64(o + 16 + 8 * i, dims[i]);
    unsigned char *q = o + 40;
    for (int r = 0; r < slabs; r++, q += 24) { put_u64(q, bounds[2 * r]); put_u64(q + 8, bounds[2 * r + 1]); put_u64(q + 16, stream_bytes[r]); }
    for (int r = 0; r < slabs; r++) { memcpy(q, streams[r], stream_bytes[r]); q += stream_bytes[r]; }
    *outSize = total;
    return o;
}

int sz_slab_unpack(const unsigned char *blob, size_t len, int *dataType, size_t dims[3], int *slabs, sz_slab_entry *entries, int max_entries)
{
    if (!blob || len < 40 || memcmp(blob, "SZSL", 4) != 0 || get_u32(blob + 4) != 1) return SZ_NSCS;
    const uint32_t n = get_u32(blob + 8), dt = get_u32(blob + 12);
    if (n == 0 || dt > 1 || (len - 40) / 24 < n) return SZ_NSCS;
    if (dataType) *dataType = dt == 0 ? SZ_FLOAT : SZ_DOUBLE;
    if (dims) for (int i = 0; i < 3; i++) dims[i] = (size_t)get_u64(blob + 16 + 8 * i);
    if (slabs) *slabs = (int)n;
    size_t off = 40 + 24 * (size_t)n;
    for (uint32_t r = 0; r < n; r++) {
        const unsigned char *q = blob + 40 + 24 * (size_t)r;
        const uint64_t z0 = get_u64(q), z1 = get_u64(q + 8), nb = get_u64(q + 16);
        if (z1 < z0 || nb > len - off) return SZ_NSCS;                    /* a table that points outside the container */
        if (entries && (int)r < max_entries) { entries[r].z_begin = (size_t)z0; entries[r].z_end = (size_t)z1; entries[r].offset = off; entries[r].bytes = (size_t)nb; }
        off += (size_t)nb;
    }
    return SZ_SCES;
}

unsigned char *sz_slab_compress(int dataType, void *data, size_t *outSize, int errBoundMode, double absErrBound, double relBoundRatio,
                                double pwrBoundRatio, size_t r3, size_t r2, size_t r1, int slabs)
{
    if (!data || !outSize || slabs < 1 || r3 < 1 || r2 < 1 || r1 < 1 || (dataType != SZ_FLOAT && dataType != SZ_DOUBLE)) return NULL;
    const size_t esz = dataType == SZ_FLOAT ? 4 : 8, plane = r2 * r1;
    size_t *bounds = (size_t *)malloc(2 * (size_t)slabs * sizeof(size_t));
    unsigned char **streams = (unsigned char **)calloc((size_t)slabs, sizeof(unsigned char *));
    size_t *bytes = (size_t *)calloc((size_t)slabs, sizeof(size_t));
    unsigned char *out = NULL;
    if (!bounds || !streams || !bytes) goto done;
    sz_slab_bounds(r3, slabs, 6, bounds);
    /* range-based bounds are derived from the range of the WHOLE array (the all-reduce of the multi-GPU path): turn them into the
     * absolute bound the reference would use (computeABSErrBoundFromABS_REL etc., dataCompression.c:288-332) once, here */
    int mode = errBoundMode; double abs_eb = absErrBound;
    if (confparams_cpr == NULL && SZ_Init(NULL) != SZ_SCES) goto done;
    /* the mode is the ARGUMENT's, as in SZ_compress_args (sz.c:294-391 assigns confparams_cpr->errorBoundMode = errBoundMode before it looks
     * at it): the configuration's own errorBoundMode -- PSNR after SZ_Init(NULL) -- must not override an explicit ABS / REL call */
    const int psnr_mode = errBoundMode == PSNR, norm_mode = errBoundMode == NORM;
    if (norm_mode) {                                     /* conf.c:62-65 on the element count of the whole array */
        abs_eb = sqrt(3.0 / (double)(r3 * plane)) * confparams_cpr->normErr;
        mode = ABS;
    }
    if (errBoundMode == REL || errBoundMode == ABS_AND_REL || errBoundMode == ABS_OR_REL || psnr_mode) {
        double lo, hi;
        const size_t n = r3 * plane;
        if (dataType == SZ_FLOAT) { const float *p = (const float *)data; float a = p[0], b = p[0]; for (size_t i = 1; i < n; i++) { if (p[i] < a) a = p[i]; else if (p[i] > b) b = p[i]; } lo = a; hi = b; }
        else { const double *p = (const double *)data; double a = p[0], b = p[0]; for (size_t i = 1; i < n; i++) { if (p[i] < a) a = p[i]; else if (p[i] > b) b = p[i]; } lo = a; hi = b; }
        const double range = dataType == SZ_FLOAT ? (double)((float)hi - (float)lo) : hi - lo, rel = relBoundRatio * range;
        if (psnr_mode) {                                 /* conf.c:54-60 on the range of the whole array */
            abs_eb = range * pow(10, (confparams_cpr->psnr + 10 * log10(1 - 2.0 / 3.0 * (double)confparams_cpr->predThreshold)) / (-20));
        } else if (errBoundMode == REL) abs_eb = rel;
        else if (dataType == SZ_FLOAT) {                 /* min_f / max_f narrow both operands to float (dataCompression.c:320-322) */
            const float fa = (float)absErrBound, fb = (float)rel;
            abs_eb = errBoundMode == ABS_AND_REL ? (fa < fb ? fa : fb) : (fa > fb ? fa : fb);
        } else abs_eb = errBoundMode == ABS_AND_REL ? (absErrBound < rel ? absErrBound : rel) : (absErrBound > rel ? absErrBound : rel);
        mode = ABS;
    }
    for (int s = 0; s < slabs; s++) {
        const size_t z0 = bounds[2 * s], h = bounds[2 * s + 1] - z0;
        if (h == 0) { streams[s] = (unsigned char *)malloc(1); bytes[s] = 0; if (!streams[s]) goto done; continue; }
        streams[s] = SZ_compress_args(dataType, (unsigned char *)data + z0 * plane * esz, &bytes[s], mode, abs_eb, relBoundRatio, pwrBoundRatio, 0, 0, h, r2, r1);
        if (!streams[s]) goto done;
    }
    { const size_t dims[3] = {r3, r2, r1}; out = sz_slab_pack(dataType, dims, slabs, bounds, (const unsigned char *const *)streams, bytes, outSize); }
done:
    if (streams) for (int s = 0; s < slabs; s++) free(streams[s]);
    free(streams); free(bytes); free(bounds);
    return out;
}

void *sz_slab_decompress(const unsigned char *blob, size_t len, int *dataType, size_t dims[3])
{
    int dt = 0, n = 0; size_t d[3];
    if (sz_slab_unpack(blob, len, &dt, d, &n, NULL, 0) != SZ_SCES) return NULL;
    sz_slab_entry *e = (sz_slab_entry *)malloc((size_t)n * sizeof(sz_slab_entry));
    if (!e || sz_slab_unpack(blob, len, &dt, d, &n, e, n) != SZ_SCES) { free(e); return NULL; }
    const size_t esz = dt == SZ_FLOAT ? 4 : 8;
    /* dimensions from an untrusted container: their product, times the element size, must not wrap (a wrapped `total` would give a
     * buffer far smaller than what the slabs below copy into it), and the slabs must tile [0, d0) without gaps or overlap (planes no
     * slab covers would come back as uninitialised memory) */
    if (d[0] == 0 || d[1] == 0 || d[2] == 0 || d[1] > SIZE_MAX / d[2] || d[0] > SIZE_MAX / (d[1] * d[2]) || d[0] * d[1] * d[2] > SIZE_MAX / esz ||
        d[0] * d[1] * d[2] >= ((size_t)1 << 46)) { free(e); return NULL; }
    const size_t plane = d[1] * d[2];
    const size_t total = d[0] * plane * esz;
    { size_t z = 0; for (int s = 0; s < n; s++) { if (e[s].z_begin != z || e[s].z_end < z || e[s].z_end > d[0]) { free(e); return NULL; } z = e[s].z_end; } if (z != d[0]) { free(e); return NULL; } }
    unsigned char *out = (unsigned char *)malloc(total);
    if (!out) { free(e); return NULL; }
    for (int s = 0; s < n; s++) {
        const size_t h = e[s].z_end - e[s].z_begin;
        if (h == 0) continue;
        void *part = SZ_decompress(dt, (unsigned char *)blob + e[s].offset, e[s].bytes, 0, 0, h, d[1], d[2]);
        if (!part) { free(out); free(e); return NULL; }
        memcpy(out + e[s].z_begin * plane * esz, part, h * plane * esz);
        free(part);
    }
    free(e);
    if (dataType) *dataType = dt;
    if (dims) { dims[0] = d[0]; dims[1] = d[1]; dims[2] = d[2]; }
    return out;
}
