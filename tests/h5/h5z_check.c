/* h5z_check.c -- TEST PROGRAM: writes and reads HDF5 datasets through filter 32017 the way an application does (the plugin is found by
 * HDF5 through HDF5_PLUGIN_PATH; this program links libhdf5 only), and leaves the input and the raw chunks behind for the Python side.
 *   h5z_check <dir> cd    bounds in the cd_values (nine words: mode, abs, rel, pw_rel, psnr as big-endian halves of doubles)
 *   h5z_check <dir> cfg   no cd_values: bounds from ./sz.config
 * Mirrors the reference's hdf5-filter/H5Z-SZ/test/szToHDF5.c and dszFromHDF5.c. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hdf5.h>

#define FILTER 32017
static void words(double v, unsigned *hi, unsigned *lo) { unsigned long long u; memcpy(&u, &v, 8); *hi = (unsigned)(u >> 32); *lo = (unsigned)u; }
static int dump(const char *dir, const char *name, const void *p, size_t n)
{
    char path[1024]; snprintf(path, sizeof(path), "%s/%s", dir, name);
    FILE *f = fopen(path, "wb"); if (!f) return -1;
    fwrite(p, 1, n, f); fclose(f); return 0;
}
#define CHECK(x) do { if ((x) < 0) { fprintf(stderr, "h5z_check: %s failed (line %d)\n", #x, __LINE__); return 2; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 3) return 1;
    const char *dir = argv[1];
    const int with_cd = strcmp(argv[2], "cd") == 0;
    char path[1024]; snprintf(path, sizeof(path), "%s/t.h5", dir);
    if (H5Zfilter_avail(FILTER) <= 0) { fprintf(stderr, "h5z_check: filter %d not available (HDF5_PLUGIN_PATH?)\n", FILTER); return 3; }

    /* ---- a 3-D float dataset in two chunks, ABS 1e-3 ---- */
    enum { N0 = 24, N1 = 40, N2 = 56 };
    float *a = (float *)malloc(sizeof(float) * N0 * N1 * N2);
    for (int i = 0; i < N0; i++) for (int j = 0; j < N1; j++) for (int k = 0; k < N2; k++)
        a[(i * N1 + j) * N2 + k] = (float)(sin(0.11 * i) * cos(0.07 * j) + 0.5 * sin(0.05 * k + 0.02 * i * j));
    /* ---- a 2-D double dataset in one chunk, REL 1e-3 ---- */
    enum { M0 = 60, M1 = 72 };
    double *b = (double *)malloc(sizeof(double) * M0 * M1);
    for (int i = 0; i < M0; i++) for (int j = 0; j < M1; j++) b[i * M1 + j] = 100.0 * sin(0.09 * i + 0.013 * j * j / 10.0) + 0.01 * j;

    hid_t file = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT); CHECK(file);
    unsigned cd[9]; memset(cd, 0, sizeof(cd));
    {
        hsize_t dims[3] = {N0, N1, N2}, chunk[3] = {N0 / 2, N1, N2};
        hid_t sp = H5Screate_simple(3, dims, NULL), pl = H5Pcreate(H5P_DATASET_CREATE);
        CHECK(H5Pset_chunk(pl, 3, chunk));
        cd[0] = 0 /* ABS */; words(1e-3, &cd[1], &cd[2]);
        CHECK(H5Pset_filter(pl, FILTER, H5Z_FLAG_MANDATORY, with_cd ? 9 : 0, cd));
        hid_t ds = H5Dcreate2(file, "f32", H5T_IEEE_F32LE, sp, H5P_DEFAULT, pl, H5P_DEFAULT); CHECK(ds);
        CHECK(H5Dwrite(ds, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, a));
        H5Dclose(ds); H5Pclose(pl); H5Sclose(sp);
    }
    {
        hsize_t dims[2] = {M0, M1};
        hid_t sp = H5Screate_simple(2, dims, NULL), pl = H5Pcreate(H5P_DATASET_CREATE);
        CHECK(H5Pset_chunk(pl, 2, dims));
        memset(cd, 0, sizeof(cd)); cd[0] = 1 /* REL */; words(1e-3, &cd[3], &cd[4]);
        CHECK(H5Pset_filter(pl, FILTER, H5Z_FLAG_MANDATORY, with_cd ? 9 : 0, cd));
        hid_t ds = H5Dcreate2(file, "f64", H5T_IEEE_F64LE, sp, H5P_DEFAULT, pl, H5P_DEFAULT); CHECK(ds);
        CHECK(H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, b));
        H5Dclose(ds); H5Pclose(pl); H5Sclose(sp);
    }
    CHECK(H5Fclose(file));

    /* ---- read back ---- */
    file = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT); CHECK(file);
    float *a2 = (float *)malloc(sizeof(float) * N0 * N1 * N2);
    double *b2 = (double *)malloc(sizeof(double) * M0 * M1);
    hid_t ds = H5Dopen2(file, "f32", H5P_DEFAULT); CHECK(ds);
    CHECK(H5Dread(ds, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, a2));
    hsize_t stored_a = H5Dget_storage_size(ds);
    for (int c = 0; c < 2; c++) {                                   /* the raw chunks */
        hsize_t off[3] = {(hsize_t)c * (N0 / 2), 0, 0}, sz = 0; uint32_t mask = 0;
        CHECK(H5Dget_chunk_storage_size(ds, off, &sz));
        unsigned char *raw = (unsigned char *)malloc(sz);
        CHECK(H5Dread_chunk(ds, H5P_DEFAULT, off, &mask, raw));
        char name[64]; snprintf(name, sizeof(name), "f32_chunk%d.sz", c);
        dump(dir, name, raw, sz); free(raw);
    }
    H5Dclose(ds);
    ds = H5Dopen2(file, "f64", H5P_DEFAULT); CHECK(ds);
    CHECK(H5Dread(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, b2));
    hsize_t stored_b = H5Dget_storage_size(ds);
    {
        hsize_t off[2] = {0, 0}, sz = 0; uint32_t mask = 0;
        CHECK(H5Dget_chunk_storage_size(ds, off, &sz));
        unsigned char *raw = (unsigned char *)malloc(sz);
        CHECK(H5Dread_chunk(ds, H5P_DEFAULT, off, &mask, raw));
        dump(dir, "f64_chunk0.sz", raw, sz); free(raw);
    }
    H5Dclose(ds); H5Fclose(file);
    dump(dir, "f32_in.bin", a, sizeof(float) * N0 * N1 * N2); dump(dir, "f32_out.bin", a2, sizeof(float) * N0 * N1 * N2);
    dump(dir, "f64_in.bin", b, sizeof(double) * M0 * M1); dump(dir, "f64_out.bin", b2, sizeof(double) * M0 * M1);
    double ea = 0, eb = 0;
    for (size_t i = 0; i < (size_t)N0 * N1 * N2; i++) { double e = fabs((double)a2[i] - a[i]); if (e > ea) ea = e; }
    for (size_t i = 0; i < (size_t)M0 * M1; i++) { double e = fabs(b2[i] - b[i]); if (e > eb) eb = e; }
    printf("f32: stored %llu of %zu bytes, max err %.6g\nf64: stored %llu of %zu bytes, max err %.6g\n", (unsigned long long)stored_a,
           sizeof(float) * N0 * N1 * N2, ea, (unsigned long long)stored_b, sizeof(double) * M0 * M1, eb);
    return 0;
}
