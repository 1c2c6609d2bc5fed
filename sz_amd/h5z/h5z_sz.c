/*
 * h5z_sz.c -- HDF5 dynamically loaded filter 32017 over the MI355X SZ library (libszhip.so).
 * Interface and on-disk conventions of the reference's hdf5-filter/H5Z-SZ/src/H5Z_SZ.c:
 *   - plugin discovery: H5PLget_plugin_type / H5PLget_plugin_info (:36-37);
 *   - cd_values: [dim, dataType, extents (slowest first; a 1-D length as two 32-bit halves)] written by set_local (:383-508), optionally
 *     followed by the nine words of an error configuration {mode, abs, rel, pw_rel, psnr} the application passed to H5Pset_filter
 *     (SZ_errConfigToCdArray, :362-381);
 *   - a chunk is one SZ stream of SZ_compress_args / SZ_compress (:542-828); chunks of fewer than 20 values pass through.
 * Every chunk is one call into the GPU library: HDF5 hands over host buffers, the library stages them (H2D), compresses on the device
 * and returns the stream.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <H5PLextern.h>
#include "H5Z_SZ.h"

int load_conffile_flag = 0;
int init_sz_flag = 0;
char cfgFile[256] = "sz.config";

static herr_t h5z_sz_set_local(hid_t dcpl_id, hid_t type_id, hid_t chunk_space_id);
static size_t h5z_sz_filter(unsigned int flags, size_t cd_nelmts, const unsigned int cd_values[], size_t nbytes, size_t *buf_size, void **buf);

const H5Z_class2_t H5Z_SZ[1] = {{
    H5Z_CLASS_T_VERS, (H5Z_filter_t)H5Z_FILTER_SZ, 1, 1, "SZ compressor/decompressor for floating-point data.", NULL, h5z_sz_set_local, h5z_sz_filter,
}};

H5PL_type_t H5PLget_plugin_type(void) { return H5PL_TYPE_FILTER; }
const void *H5PLget_plugin_info(void) { return H5Z_SZ; }

int H5Z_SZ_Init(char *cfg)
{
    if (!init_sz_flag) { if (SZ_Init(cfg) != SZ_SCES) return -1; init_sz_flag = 1; load_conffile_flag = 1; }
    return H5Zregister(H5Z_SZ) < 0 ? -1 : 0;
}
int H5Z_SZ_Init_Params(sz_params *params)
{
    if (SZ_Init_Params(params) != SZ_SCES) return -1;
    init_sz_flag = 1;
    return H5Zregister(H5Z_SZ) < 0 ? -1 : 0;
}
sz_params *H5Z_SZ_Init_Default(void)
{
    if (SZ_Init(NULL) != SZ_SCES) return NULL;
    init_sz_flag = 1;
    if (H5Zregister(H5Z_SZ) < 0) return NULL;
    return confparams_cpr;
}
int H5Z_SZ_Finalize(void)
{
    SZ_Finalize();
    init_sz_flag = 0;
    return H5Zunregister(H5Z_FILTER_SZ) < 0 ? -1 : 0;
}

/* ---- cd_values ---- */
static unsigned hi32(double v) { unsigned long long u; memcpy(&u, &v, 8); return (unsigned)(u >> 32); }
static unsigned lo32(double v) { unsigned long long u; memcpy(&u, &v, 8); return (unsigned)u; }
static double from32(unsigned hi, unsigned lo) { unsigned long long u = ((unsigned long long)hi << 32) | lo; double v; memcpy(&v, &u, 8); return v; }
static int words_of_dims(int dim) { return dim == 1 ? 4 : dim + 2; }

void SZ_errConfigToCdArray(size_t *cd_nelmts, unsigned int **cd_values, int error_bound_mode, double abs_error, double rel_error, double pw_rel_error, double psnr)
{
    unsigned *v = (unsigned *)malloc(sizeof(unsigned) * 9);
    const double d[4] = {abs_error, rel_error, pw_rel_error, psnr};
    v[0] = (unsigned)error_bound_mode;
    for (int i = 0; i < 4; i++) { v[1 + 2 * i] = hi32(d[i]); v[2 + 2 * i] = lo32(d[i]); }   /* big-endian halves of the IEEE bits */
    *cd_values = v; *cd_nelmts = 9;
}

void SZ_cdArrayToMetaData(size_t cd_nelmts, const unsigned int cd_values[], int *dimSize, int *dataType, size_t *r5, size_t *r4, size_t *r3, size_t *r2, size_t *r1)
{
    *r1 = *r2 = *r3 = *r4 = *r5 = 0; *dimSize = 0; *dataType = 0;
    if (cd_nelmts < 4) return;
    *dimSize = (int)cd_values[0]; *dataType = (int)cd_values[1];
    switch (*dimSize) {
    case 1: *r1 = (size_t)(((unsigned long long)cd_values[2] << 32) | cd_values[3]); break;
    case 2: *r2 = cd_values[2]; *r1 = cd_values[3]; break;
    case 3: *r3 = cd_values[2]; *r2 = cd_values[3]; *r1 = cd_values[4]; break;
    case 4: *r4 = cd_values[2]; *r3 = cd_values[3]; *r2 = cd_values[4]; *r1 = cd_values[5]; break;
    default: *r5 = cd_values[2]; *r4 = cd_values[3]; *r3 = cd_values[4]; *r2 = cd_values[5]; *r1 = cd_values[6];
    }
}

void SZ_copymetaDataToCdArray(size_t *cd_nelmts, unsigned int *cd_values, int dataType, size_t r5, size_t r4, size_t r3, size_t r2, size_t r1)
{
    const int dim = computeDimension(r5, r4, r3, r2, r1);
    cd_values[0] = (unsigned)dim; cd_values[1] = (unsigned)dataType;
    switch (dim) {
    case 1: cd_values[2] = (unsigned)((unsigned long long)r1 >> 32); cd_values[3] = (unsigned)r1; break;
    case 2: cd_values[2] = (unsigned)r2; cd_values[3] = (unsigned)r1; break;
    case 3: cd_values[2] = (unsigned)r3; cd_values[3] = (unsigned)r2; cd_values[4] = (unsigned)r1; break;
    case 4: cd_values[2] = (unsigned)r4; cd_values[3] = (unsigned)r3; cd_values[4] = (unsigned)r2; cd_values[5] = (unsigned)r1; break;
    default: cd_values[2] = (unsigned)r5; cd_values[3] = (unsigned)r4; cd_values[4] = (unsigned)r3; cd_values[5] = (unsigned)r2; cd_values[6] = (unsigned)r1;
    }
    *cd_nelmts = (size_t)words_of_dims(dim);
}

void SZ_cdArrayToMetaDataErr(size_t cd_nelmts, const unsigned int cd_values[], int *dimSize, int *dataType, size_t *r5, size_t *r4, size_t *r3, size_t *r2,
                             size_t *r1, int *error_bound_mode, double *abs_error, double *rel_error, double *pw_rel_error, double *psnr)
{
    SZ_cdArrayToMetaData(cd_nelmts, cd_values, dimSize, dataType, r5, r4, r3, r2, r1);
    int k = words_of_dims(*dimSize);
    *error_bound_mode = (int)cd_values[k];
    *abs_error = from32(cd_values[k + 1], cd_values[k + 2]);
    *rel_error = from32(cd_values[k + 3], cd_values[k + 4]);
    *pw_rel_error = from32(cd_values[k + 5], cd_values[k + 6]);
    *psnr = from32(cd_values[k + 7], cd_values[k + 8]);
}

/* the chunk's extents with size-1 dimensions squeezed out, followed by the application's nine error words if it gave any */
void SZ_refreshDimForCdArray(int dataType, size_t old_cd_nelmts, unsigned int *old_cd_values, size_t *new_cd_nelmts, unsigned int **new_cd_values,
                             size_t r5, size_t r4, size_t r3, size_t r2, size_t r1)
{
    unsigned *v = (unsigned *)calloc(16, sizeof(unsigned));
    size_t c[5];
    filterDimension(r5, r4, r3, r2, r1, c);
    size_t n = 0;
    SZ_copymetaDataToCdArray(&n, v, dataType, c[4], c[3], c[2], c[1], c[0]);
    if (old_cd_nelmts != 0) { for (int i = 0; i < 9; i++) v[n + i] = old_cd_values[i]; n += 9; }
    *new_cd_values = v; *new_cd_nelmts = n;
}

int checkCDValuesWithErrors(size_t cd_nelmts, const unsigned int cd_values[])
{
    const int dim = (int)cd_values[0];
    if (dim < 1 || dim > 5) return -1;
    return cd_nelmts > (size_t)words_of_dims(dim) ? 1 : 0;
}

/* ---- callbacks ---- */
static herr_t h5z_sz_set_local(hid_t dcpl_id, hid_t type_id, hid_t chunk_space_id)
{
    unsigned int flags = 0;
    size_t mem_n = 9;
    unsigned int mem_values[16] = {0};
    if (H5Pget_filter_by_id2(dcpl_id, H5Z_FILTER_SZ, &flags, &mem_n, mem_values, 0, NULL, NULL) < 0) return -1;
    if (mem_n != 0 && mem_n != 9) {
        fprintf(stderr, "H5Z-SZ: cd_values must be empty (bounds from %s) or the nine words of SZ_errConfigToCdArray\n", cfgFile);
        return -1;
    }
    if (!init_sz_flag) {            /* bounds in the cd_values: defaults for everything else; otherwise the configuration file (H5Z_SZ.c:398-408) */
        if (SZ_Init(mem_n == 0 ? cfgFile : NULL) != SZ_SCES) return -1;
        init_sz_flag = 1;
    }
    H5T_class_t cls = H5Tget_class(type_id);
    size_t dsize = H5Tget_size(type_id);
    if (cls != H5T_FLOAT || (dsize != 4 && dsize != 8)) {
        fprintf(stderr, "H5Z-SZ (MI355X build): only 32- and 64-bit floating-point datasets are handled\n");
        return -1;
    }
    hsize_t dims[H5S_MAX_RANK];
    int ndims = H5Sget_simple_extent_dims(chunk_space_id, dims, NULL);
    if (ndims < 1 || ndims > 5) { fprintf(stderr, "H5Z-SZ: chunks of 1 to 5 dimensions\n"); return -1; }
    size_t r[5] = {0, 0, 0, 0, 0};   /* r[0] fastest */
    for (int i = 0; i < ndims; i++) { if (dims[i] > MAX_CHUNK_SIZE) return -1; r[ndims - 1 - i] = (size_t)dims[i]; }
    unsigned int *cd = NULL; size_t n = 0;
    SZ_refreshDimForCdArray(dsize == 4 ? SZ_FLOAT : SZ_DOUBLE, mem_n, mem_values, &n, &cd, r[4], r[3], r[2], r[1], r[0]);
    herr_t rc = H5Pmodify_filter(dcpl_id, H5Z_FILTER_SZ, flags, n, cd);
    free(cd);
    return rc < 0 ? -1 : 1;
}

static size_t h5z_sz_filter(unsigned int flags, size_t cd_nelmts, const unsigned int cd_values[], size_t nbytes, size_t *buf_size, void **buf)
{
    if (cd_nelmts == 0) return nbytes;
    const int with_err = checkCDValuesWithErrors(cd_nelmts, cd_values);
    if (with_err < 0) return 0;
    /* the form with error bounds carries exactly nine words behind the dimensions: anything else would be read out of bounds below */
    if (with_err && cd_nelmts != (size_t)words_of_dims((int)cd_values[0]) + 9) return 0;
    if (!with_err && cd_nelmts < (size_t)words_of_dims((int)cd_values[0])) return 0;
    size_t r1, r2, r3, r4, r5;
    int dim, dataType, mode = 0;
    double abs_e = 0, rel_e = 0, pwr_e = 0, psnr = 0;
    if (with_err) SZ_cdArrayToMetaDataErr(cd_nelmts, cd_values, &dim, &dataType, &r5, &r4, &r3, &r2, &r1, &mode, &abs_e, &rel_e, &pwr_e, &psnr);
    else SZ_cdArrayToMetaData(cd_nelmts, cd_values, &dim, &dataType, &r5, &r4, &r3, &r2, &r1);
    if (with_err && !(flags & H5Z_FLAG_REVERSE)) {                 /* bit patterns from a file: no NaN, infinite or negative bounds */
        const double b[4] = {abs_e, rel_e, pwr_e, psnr};
        for (int i = 0; i < 4; ++i) if (!(b[i] == b[i]) || b[i] < 0 || b[i] > 1.0e300) return 0;
    }
    const size_t n = computeDataLength(r5, r4, r3, r2, r1);
    if (n < 20) return nbytes;                                     /* H5Z_SZ.c:566 */
    if (dataType != SZ_FLOAT && dataType != SZ_DOUBLE) return 0;
    const size_t esz = dataType == SZ_FLOAT ? 4 : 8;
    if (flags & H5Z_FLAG_REVERSE) {
        void *data = SZ_decompress(dataType, (unsigned char *)*buf, nbytes, r5, r4, r3, r2, r1);
        if (!data) return 0;
        free(*buf); *buf = data; *buf_size = n * esz;
        return n * esz;
    }
    if (nbytes < n * esz) return 0;
    if (!init_sz_flag) { if (SZ_Init(with_err ? NULL : cfgFile) != SZ_SCES) return 0; init_sz_flag = 1; }
    size_t out = 0;
    unsigned char *bytes;
    if (with_err) {
        if (mode == PSNR) confparams_cpr->psnr = psnr;
        bytes = SZ_compress_args(dataType, *buf, &out, mode, abs_e, rel_e, pwr_e, r5, r4, r3, r2, r1);
    } else bytes = SZ_compress(dataType, *buf, &out, r5, r4, r3, r2, r1);
    if (!bytes) return 0;
    free(*buf); *buf = bytes; *buf_size = out;
    return out;
}
