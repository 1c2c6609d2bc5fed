/*
 * sz_omp.h -- the header the reference's OpenMP callers include (example/sz_openmp.c:6): the entry points of the reference's OpenMP
 * container, served by the HIP layer of this library.
 *
 * Interface replaced: sz/include/sz_omp.h:24-47 of the reference (SZ_compress_{float,double}_{1D,2D,3D}_MDQ_openmp,
 * decompressDataSeries_{float,double}_{1D,2D,3D}_openmp) and the thread helpers of an OpenMP build (sz_omp.c:14-53).  Same names, argument
 * order and stream as an OpenMP build of libSZ: a stream written here is read by `sz_openmp -k` of the reference and the other way round
 * (tests/test_zz_omp_hip.py: 14 recorded outputs of the reference built -fopenmp).  r1 is the SLOWEST dimension in these calls, as in sz_omp.c.
 * `comp_data` of the inverse points behind the stream's first 4 + MetaDataByteLength bytes (example/sz_openmp.c:580).
 *
 * The 1-D and 2-D entry points are stubs in the reference itself (sz_omp.c:56-61, :360-364, :570-576, :866-870 return NULL / do nothing) and are
 * the same stubs here.  Huffman_init_openmp (sz_omp.h:44) is an internal of the reference's CPU coder and has no counterpart: the code book
 * is built by szhost_huff_build in the reference's heap order, the histogram on the GPU.
 *
 * The box count ("thread_num", written into the stream) is omp_get_max_threads() in the reference; here sz_set_num_threads / SZ_hip_set_omp_threads
 * set it (0 = pick one: boxes of at most 32768 points).  The HIP layer's restrictions (the box grid must divide the array) are in include/szhip.h.
 */
#ifndef _SZ_OMP_H
#define _SZ_OMP_H

#include <stdio.h>
#include <stdlib.h>
#ifdef _OPENMP
#include "omp.h"
#endif
#include "sz.h"

#ifdef __cplusplus
extern "C" {
#endif

/* sz_omp.h:26-29, :36-38 */
unsigned char *SZ_compress_float_1D_MDQ_openmp(float *oriData, size_t r1, double realPrecision, size_t *comp_size);
unsigned char *SZ_compress_float_2D_MDQ_openmp(float *oriData, size_t r1, size_t r2, double realPrecision, size_t *comp_size);
unsigned char *SZ_compress_float_3D_MDQ_openmp(float *oriData, size_t r1, size_t r2, size_t r3, float realPrecision, size_t *comp_size);
unsigned char *SZ_compress_double_1D_MDQ_openmp(double *oriData, size_t r1, double realPrecision, size_t *comp_size);
unsigned char *SZ_compress_double_2D_MDQ_openmp(double *oriData, size_t r1, size_t r2, double realPrecision, size_t *comp_size);
unsigned char *SZ_compress_double_3D_MDQ_openmp(double *oriData, size_t r1, size_t r2, size_t r3, double realPrecision, size_t *comp_size);

/* sz_omp.h:31-33, :40-42 */
void decompressDataSeries_float_1D_openmp(float **data, size_t r1, unsigned char *comp_data);
void decompressDataSeries_float_2D_openmp(float **data, size_t r1, size_t r2, unsigned char *comp_data);
void decompressDataSeries_float_3D_openmp(float **data, size_t r1, size_t r2, size_t r3, unsigned char *comp_data);
void decompressDataSeries_double_1D_openmp(double **data, size_t r1, unsigned char *comp_data);
void decompressDataSeries_double_2D_openmp(double **data, size_t r1, size_t r2, unsigned char *comp_data);
void decompressDataSeries_double_3D_openmp(double **data, size_t r1, size_t r2, size_t r3, unsigned char *comp_data);

/* sz_omp.c:14-53 */
void sz_set_num_threads(int nthreads);
int sz_get_max_threads(void);
int sz_get_thread_num(void);
double sz_wtime(void);

#ifdef __cplusplus
}
#endif

#endif /* _SZ_OMP_H */
