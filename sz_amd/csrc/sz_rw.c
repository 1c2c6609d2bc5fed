/* sz_rw.c -- raw-array file helpers used by callers of the SZ API (include/rw.h; reference sz/src/rw.c). Host C. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "sz.h"
#include "rw.h"

size_t checkFileSize(char *srcFilePath, int *status)
{
    FILE *f = fopen(srcFilePath, "rb");
    if (!f) { printf("Failed to open input file. 1\n"); *status = SZ_FERR; return (size_t)-1; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fclose(f);
    *status = SZ_SCES;
    return (size_t)n;
}

unsigned char *readByteData(char *srcFilePath, size_t *byteLength, int *status)
{
    FILE *f = fopen(srcFilePath, "rb");
    if (!f) { printf("Failed to open input file. 1\n"); *status = SZ_FERR; return NULL; }
    fseek(f, 0, SEEK_END);
    *byteLength = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *buf = (unsigned char *)malloc(*byteLength ? *byteLength : 1);
    if (fread(buf, 1, *byteLength, f) != *byteLength) { fclose(f); free(buf); *status = SZ_FERR; return NULL; }
    fclose(f);
    *status = SZ_SCES;
    return buf;
}

static void swap_elems(unsigned char *b, size_t n, size_t w)
{
    for (size_t i = 0; i < n; i++) for (size_t k = 0; k < w / 2; k++) { unsigned char t = b[i * w + k]; b[i * w + k] = b[i * w + w - 1 - k]; b[i * w + w - 1 - k] = t; }
}

float *readFloatData(char *srcFilePath, size_t *nbEle, int *status)
{
    size_t bytes;
    unsigned char *b = readByteData(srcFilePath, &bytes, status);
    if (!b) return NULL;
    *nbEle = bytes / 4;
    if (dataEndianType != sysEndianType) swap_elems(b, *nbEle, 4);
    return (float *)b;
}

double *readDoubleData(char *srcFilePath, size_t *nbEle, int *status)
{
    size_t bytes;
    unsigned char *b = readByteData(srcFilePath, &bytes, status);
    if (!b) return NULL;
    *nbEle = bytes / 8;
    if (dataEndianType != sysEndianType) swap_elems(b, *nbEle, 8);
    return (double *)b;
}

void writeByteData(unsigned char *bytes, size_t byteLength, char *tgtFilePath, int *status)
{
    FILE *f = fopen(tgtFilePath, "wb");
    if (!f) { printf("Failed to open input file. 3\n"); *status = SZ_FERR; return; }
    fwrite(bytes, 1, byteLength, f);
    fclose(f);
    *status = SZ_SCES;
}

void writeFloatData_inBytes(float *data, size_t nbEle, char *tgtFilePath, int *status)
{
    writeByteData((unsigned char *)data, nbEle * sizeof(float), tgtFilePath, status);
}

void writeDoubleData_inBytes(double *data, size_t nbEle, char *tgtFilePath, int *status)
{
    writeByteData((unsigned char *)data, nbEle * sizeof(double), tgtFilePath, status);
}

/* the text-file helpers of the reference's tool (rw.c:22-31, :796-838, :989-1009): one value per line, "%.30G" / "%.20G" */
int checkFileExistance(char *filePath) { return filePath && access(filePath, F_OK) != -1 ? 1 : 0; }
void writeFloatData(float *data, size_t nbEle, char *tgtFilePath, int *status)
{
    FILE *f = fopen(tgtFilePath, "wb");
    if (!f) { printf("Failed to open input file. 3\n"); *status = SZ_FERR; return; }
    for (size_t i = 0; i < nbEle; i++) fprintf(f, "%.30G\n", data[i]);
    fclose(f); *status = SZ_SCES;
}
void writeDoubleData(double *data, size_t nbEle, char *tgtFilePath, int *status)
{
    FILE *f = fopen(tgtFilePath, "wb");
    if (!f) { printf("Failed to open input file. 3\n"); *status = SZ_FERR; return; }
    for (size_t i = 0; i < nbEle; i++) fprintf(f, "%.20G\n", data[i]);
    fclose(f); *status = SZ_SCES;
}
void writeStrings(int nbStr, char *str[], char *tgtFilePath, int *status)
{
    FILE *f = fopen(tgtFilePath, "wb");
    if (!f) { printf("Failed to open input file. 3\n"); *status = SZ_FERR; return; }
    for (int i = 0; i < nbStr; i++) fprintf(f, "%s\n", str[i]);
    fclose(f); *status = SZ_SCES;
}
