/* rw.h -- the handful of file helpers the reference's example programs use next to the SZ API
 * (reference: sz/include/rw.h:41-70).  Same names and signatures; data files are raw arrays, byte-swapped on read
 * when dataEndianType differs from the host's (sz/src/rw.c:425-455). */
#ifndef _SZ_RW_H
#define _SZ_RW_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
size_t checkFileSize(char *srcFilePath, int *status);
unsigned char *readByteData(char *srcFilePath, size_t *byteLength, int *status);
float *readFloatData(char *srcFilePath, size_t *nbEle, int *status);
double *readDoubleData(char *srcFilePath, size_t *nbEle, int *status);
void writeByteData(unsigned char *bytes, size_t byteLength, char *tgtFilePath, int *status);
void writeFloatData_inBytes(float *data, size_t nbEle, char *tgtFilePath, int *status);
void writeDoubleData_inBytes(double *data, size_t nbEle, char *tgtFilePath, int *status);
/* the text-file helpers of the reference's command-line tool (sz/include/rw.h; sz/src/rw.c:22, :796, :818, :989) */
int checkFileExistance(char *filePath);
void writeFloatData(float *data, size_t nbEle, char *tgtFilePath, int *status);
void writeDoubleData(double *data, size_t nbEle, char *tgtFilePath, int *status);
void writeStrings(int nbStr, char *str[], char *tgtFilePath, int *status);
#ifdef __cplusplus
}
#endif
#endif
