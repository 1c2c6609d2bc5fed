/* sz_cli.c -- a small command-line front end over the SZ API of this build, with the option letters of the
 * reference's `sz` tool (example/sz.c:30-88):
 *   sz_cli -z [out.sz] -f|-d -c sz.config [-M ABS|REL|ABS_AND_REL|ABS_OR_REL|PSNR|NORM|PW_REL] [-A abs] [-R rel] [-P pw_rel] [-S psnr] [-N norm]
 *          -i data.bin -1 nx | -2 nx ny | -3 nx ny nz | -4 nx ny nz nt     -> writes data.bin.sz (or out.sz)
 *   sz_cli -x [out.bin] -f|-d -s data.bin.sz -3 nx ny nz [-i data.bin -a] [-b|-t]
 *                                                   -> writes data.bin.sz.out (binary, or one value per line with -t); -a prints the quality report
 *   sz_cli -p -s data.bin.sz      prints what the stream's header records;   -v the version;   -h this text
 * nx is the fastest-varying dimension (r1), as in the reference.  -T (Tucker pre-processing) and -q (statistics build) are not
 * part of this build. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include "sz.h"
#include "rw.h"

static double now_s(void) { struct timeval t; gettimeofday(&t, NULL); return t.tv_sec + 1e-6 * t.tv_usec; }

static void report(int is_double, const void *ori, const void *dec, size_t n, size_t cmp_bytes)
{
    /* the formulas of the reference's `-a` report (example/sz.c:558-620): float differences for float data */
    double Max, Min, diffMax, sum = 0;
    if (is_double) {
        const double *o = ori, *d = dec; double mx = o[0], mn = o[0], dm = fabs(d[0] - o[0]);
        for (size_t i = 0; i < n; i++) { if (mx < o[i]) mx = o[i]; if (mn > o[i]) mn = o[i]; double e = fabs(d[i] - o[i]); if (dm < e) dm = e; sum += e * e; }
        Max = mx; Min = mn; diffMax = dm;
    } else {
        const float *o = ori, *d = dec; float mx = o[0], mn = o[0], dm = fabs(d[0] - o[0]);
        for (size_t i = 0; i < n; i++) { if (mx < o[i]) mx = o[i]; if (mn > o[i]) mn = o[i]; float e = fabs(d[i] - o[i]); if (dm < e) dm = e; sum += e * e; }
        Max = mx; Min = mn; diffMax = dm;
    }
    double mse = sum / n, range = is_double ? Max - Min : (double)(float)((float)Max - (float)Min);
    printf("Min=%.20G, Max=%.20G, range=%.20G\n", Min, Max, range);
    printf("Max absolute error = %.10f\n", diffMax);
    printf("Max relative error = %f\n", diffMax / (Max - Min));
    printf("PSNR = %f, NRMSE= %.20G\n", 20 * log10(range) - 10 * log10(mse), sqrt(mse) / range);
    printf("compressionRatio=%f\n", 1.0 * n * (is_double ? 8 : 4) / cmp_bytes);
}

static const char *mode_name(int m)
{
    static const char *n[] = {"ABS", "REL", "ABS_AND_REL", "ABS_OR_REL", "PSNR", "NORM", "?", "?", "?", "?", "PW_REL", "ABS_AND_PW_REL", "ABS_OR_PW_REL",
                              "REL_AND_PW_REL", "REL_OR_PW_REL"};
    return m >= 0 && m <= 14 ? n[m] : "?";
}

static int print_metadata(const char *cmp)   /* `sz -p -s file` (example/sz.c:330-346 -> SZ_printMetadata) */
{
    size_t len = 0; int st = 0;
    unsigned char *b = readByteData((char *)cmp, &len, &st);
    if (st != SZ_SCES || len < 4 + MetaDataByteLength + 4) { printf("Error: cannot read a stream header from %s\n", cmp); return 1; }
    sz_metadata *md = SZ_getMetadata(b);
    sz_params *p = md->conf_params;
    printf("=================SZ Compression Meta Data=================\n");
    printf("Version:                        \t %d.%d.%d\n", md->versionNumber[0], md->versionNumber[1], md->versionNumber[2]);
    printf("Constant data?:                 \t %s\n", md->isConstant ? "YES" : "NO");
    printf("Lossless?:                      \t %s\n", md->isLossless ? "YES" : "NO");
    printf("Size type (size of # elements): \t %d bytes\n", md->sizeType);
    printf("Num of elements:                \t %zu\n", md->dataSeriesLength);
    printf("Data type:                      \t %s\n", p->dataType == SZ_FLOAT ? "FLOAT" : p->dataType == SZ_DOUBLE ? "DOUBLE" : "other");
    if (md->defactoNBBins > 0) printf("quantization_intervals:         \t %d\n", md->defactoNBBins);
    printf("max_quant_intervals / fixed:    \t %u / %u\n", p->max_quant_intervals, p->quantization_intervals);
    printf("sampleDistance, predThreshold:  \t %d, %f\n", p->sampleDistance, p->predThreshold);
    printf("szMode:                         \t %s\n", p->szMode == SZ_BEST_SPEED ? "SZ_BEST_SPEED (without Gzip)" : p->szMode == SZ_BEST_COMPRESSION ? "SZ_BEST_COMPRESSION (with Zstd or Gzip)" : "SZ_DEFAULT_COMPRESSION (with Zstd or Gzip)");
    printf("errBoundMode:                   \t %s\n", mode_name(p->errorBoundMode));
    printf("absErrBound, relBoundRatio:     \t %g, %g\n", p->absErrBound, p->relBoundRatio);
    printf("pw_relBoundRatio, psnr:         \t %g, %g\n", p->pw_relBoundRatio, p->psnr);
    if (p->dataType == SZ_FLOAT) printf("value range:                    \t [%.9g, %.9g]\n", p->fmin, p->fmax); else printf("value range:                    \t [%.17g, %.17g]\n", p->dmin, p->dmax);
    free(p); free(md); free(b);
    return 0;
}

int main(int argc, char **argv)
{
    int compress = -1, is_double = -1, analyse = 0, mode = -1, text_out = 0, meta = 0;
    const char *cfg = NULL, *in = NULL, *cmp = NULL, *outname = NULL;
    double abs_b = -1, rel_b = -1, psnr = -1, norm = -1, pwr_b = -1;
    size_t r[5] = {0, 0, 0, 0, 0};
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (!strcmp(a, "-z") || !strcmp(a, "-x")) { compress = a[1] == 'z'; if (i + 1 < argc && argv[i + 1][0] != '-') outname = argv[++i]; }
        else if (!strcmp(a, "-p")) meta = 1;
        else if (!strcmp(a, "-h")) { printf("usage: see the head of examples/sz_cli.c (the option letters of the reference's `sz`)\n"); return 0; }
        else if (!strcmp(a, "-v")) { printf("version: %d.%d.%d (MI355X build)\n", SZ_VER_MAJOR, SZ_VER_MINOR, SZ_VER_BUILD); return 0; }
        else if (!strcmp(a, "-b")) text_out = 0; else if (!strcmp(a, "-t")) text_out = 1;
        else if (!strcmp(a, "-T") || !strcmp(a, "-q")) { printf("Error: option %s is not part of the MI355X build\n", a); return 1; }
        else if (!strcmp(a, "-P") && i + 1 < argc) pwr_b = atof(argv[++i]);
        else if (!strcmp(a, "-f")) is_double = 0; else if (!strcmp(a, "-d")) is_double = 1;
        else if (!strcmp(a, "-a")) analyse = 1;
        else if (!strcmp(a, "-c") && i + 1 < argc) cfg = argv[++i];
        else if (!strcmp(a, "-i") && i + 1 < argc) in = argv[++i];
        else if (!strcmp(a, "-s") && i + 1 < argc) cmp = argv[++i];
        else if (!strcmp(a, "-A") && i + 1 < argc) abs_b = atof(argv[++i]);
        else if (!strcmp(a, "-R") && i + 1 < argc) rel_b = atof(argv[++i]);
        else if (!strcmp(a, "-S") && i + 1 < argc) psnr = atof(argv[++i]);
        else if (!strcmp(a, "-N") && i + 1 < argc) norm = atof(argv[++i]);
        else if (!strcmp(a, "-M") && i + 1 < argc) {
            const char *m = argv[++i];
            mode = !strcmp(m, "ABS") ? ABS : !strcmp(m, "REL") ? REL : !strcmp(m, "ABS_AND_REL") ? ABS_AND_REL : !strcmp(m, "ABS_OR_REL") ? ABS_OR_REL
                 : !strcmp(m, "PSNR") ? PSNR : !strcmp(m, "NORM") ? NORM : !strcmp(m, "PW_REL") ? PW_REL : -2;
            if (mode == -2) { printf("Error: wrong error bound mode setting by using the option '-M'\n"); return 1; }
        }
        else if (a[0] == '-' && a[1] >= '1' && a[1] <= '5' && !a[2]) { int nd = a[1] - '0'; for (int k = 0; k < nd && i + 1 < argc; k++) r[k] = (size_t)atoll(argv[++i]); }
        else { printf("Error: unknown option %s\n", a); return 1; }
    }
    if (meta) { if (!cmp) { printf("Error: -p needs -s <compressed file>\n"); return 1; } return print_metadata(cmp); }
    if (compress < 0 || is_double < 0 || r[0] == 0) { printf("usage: see the head of examples/sz_cli.c\n"); return 1; }
    if (SZ_Init(cfg) == SZ_NSCS) return 1;
    if (mode >= 0) confparams_cpr->errorBoundMode = mode;          /* the reference's CLI pokes the globals the same way */
    if (abs_b >= 0) confparams_cpr->absErrBound = abs_b;
    if (rel_b >= 0) confparams_cpr->relBoundRatio = rel_b;
    if (psnr >= 0) confparams_cpr->psnr = psnr;
    if (norm >= 0) confparams_cpr->normErr = norm;
    if (pwr_b >= 0) confparams_cpr->pw_relBoundRatio = pwr_b;
    const int dt = is_double ? SZ_DOUBLE : SZ_FLOAT;
    size_t n = 0; int st = 0; void *ori = NULL; char path[4096];
    if (in) { ori = is_double ? (void *)readDoubleData((char *)in, &n, &st) : (void *)readFloatData((char *)in, &n, &st); if (st != SZ_SCES) return 1; }
    if (compress) {
        if (!ori) { printf("Error: -i is required with -z\n"); return 1; }
        size_t out = 0; double t0 = now_s();
        unsigned char *b = SZ_compress(dt, ori, &out, r[4], r[3], r[2], r[1], r[0]);
        if (!b) { printf("Error: compression failed\n"); return 1; }
        printf("compression time = %f\n", now_s() - t0);
        if (outname) snprintf(path, sizeof(path), "%s", outname); else snprintf(path, sizeof(path), "%s.sz", in);
        writeByteData(b, out, path, &st);
        printf("compressed data file: %s (%zu bytes)\n", path, out);
        free(b);
    } else {
        size_t len = 0;
        if (!cmp) { printf("Error: -s is required with -x\n"); return 1; }
        unsigned char *b = readByteData((char *)cmp, &len, &st);
        if (st != SZ_SCES) return 1;
        double t0 = now_s();
        void *dec = SZ_decompress(dt, b, len, r[4], r[3], r[2], r[1], r[0]);
        if (!dec) { printf("Error: decompression failed\n"); return 1; }
        printf("decompression time = %f seconds.\n", now_s() - t0);
        size_t ne = computeDataLength(r[4], r[3], r[2], r[1], r[0]);
        if (outname) snprintf(path, sizeof(path), "%s", outname); else snprintf(path, sizeof(path), "%s.out", cmp);
        if (text_out) {                                             /* writeFloatData / writeDoubleData (rw.c): one value per line */
            FILE *f = fopen(path, "w");
            if (!f) { printf("Error: cannot write %s\n", path); return 1; }
            for (size_t k = 0; k < ne; k++) { if (is_double) fprintf(f, "%.20G\n", ((double *)dec)[k]); else fprintf(f, "%.30G\n", (double)((float *)dec)[k]); }
            fclose(f);
        } else if (is_double) writeDoubleData_inBytes((double *)dec, ne, path, &st); else writeFloatData_inBytes((float *)dec, ne, path, &st);
        printf("decompressed data file: %s\n", path);
        if (analyse && ori) { if (n != ne) { printf("Error: size mismatch\n"); return 1; } report(is_double, ori, dec, ne, len); }
        free(dec); free(b);
    }
    free(ori);
    SZ_Finalize();
    return 0;
}
