/*
 * sz.h -- public C API of the MI355X-native SZ 2.1 build.
 *
 * This header declares, with the same names, argument order and struct layouts, the part of the
 * reference's libSZ interface that fronts the GPU hot path, so that existing callers compile
 * against it unchanged (reference: sz/include/sz.h and sz/include/defines.h; each item cites the
 * line it mirrors).  The implementation behind it (sz_amd/csrc/sz_api.c, C) derives the absolute
 * bound, frames the stream and calls the HIP layer declared in szhip.h.
 *
 * Covered: SZ_FLOAT / SZ_DOUBLE; 1-D, 2-D, 3-D arrays and 4-D arrays on the SZ 2.1 path (folded to 3-D as
 * the reference does); withRegression = YES (SZ 2.1 stream) and NO (SZ 1.4 TightDataPointStorage
 * container, 1-D .. 3-D); error-bound modes ABS, REL/VR_REL, ABS_AND_REL, ABS_OR_REL, PSNR, NORM and
 * the point-wise relative family PW_REL, ABS_AND/OR_PW_REL, REL_AND/OR_PW_REL in both of the
 * reference's forms; the zstd / gzip lossless stage of szMode; protectValueRange.
 * Not covered (an explicit error, never a silent CPU fallback): integer types, the time-step and
 * random-access modes, withRegression = NO for 4-D arrays.
 */
#ifndef _SZ_H
#define _SZ_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* sz/include/defines.h:13-17 */
#define SZ_VERNUM 0x0200
#define SZ_VER_MAJOR 2
#define SZ_VER_MINOR 1
#define SZ_VER_BUILD 12
#define SZ_VER_REVISION 4

/* sz/include/defines.h:19-22 */
#define PASTRI 103
#define HZ 102
#define SZ 101
#define SZ_Transpose 104

#define MIN_NUM_OF_ELEMENTS 20 /* defines.h:27 */

/* error-bound modes, defines.h:29-41 */
#define ABS 0
#define REL 1
#define VR_REL 1
#define ABS_AND_REL 2
#define ABS_OR_REL 3
#define PSNR 4
#define NORM 5
#define PW_REL 10
#define ABS_AND_PW_REL 11
#define ABS_OR_PW_REL 12
#define REL_AND_PW_REL 13
#define REL_OR_PW_REL 14

/* data types, defines.h:43-52 */
#define SZ_FLOAT 0
#define SZ_DOUBLE 1
#define SZ_UINT8 2
#define SZ_INT8 3
#define SZ_UINT16 4
#define SZ_INT16 5
#define SZ_UINT32 6
#define SZ_INT32 7
#define SZ_UINT64 8
#define SZ_INT64 9

#define LITTLE_ENDIAN_DATA 0
#define BIG_ENDIAN_DATA 1
#define LITTLE_ENDIAN_SYSTEM 0
#define BIG_ENDIAN_SYSTEM 1

/* defines.h:67-73 */
#define SZ_BEST_SPEED 0
#define SZ_BEST_COMPRESSION 1
#define SZ_DEFAULT_COMPRESSION 2
#define SZ_TEMPORAL_COMPRESSION 3
#define SZ_NO_REGRESSION 0
#define SZ_WITH_LINEAR_REGRESSION 1

#define SZ_PWR_MIN_TYPE 0
#define SZ_PWR_AVG_TYPE 1
#define SZ_PWR_MAX_TYPE 2

/* status codes, defines.h:84-90 */
#define SZ_SCES 0
#define SZ_NSCS -1
#define SZ_FERR -2
#define SZ_TERR -3
#define SZ_DERR -4
#define SZ_MERR -5
#define SZ_BERR -6

#define MetaDataByteLength 28        /* defines.h:97 */
#define MetaDataByteLength_double 36 /* defines.h:98 */

#define GZIP_COMPRESSOR 0 /* defines.h:103 */
#define ZSTD_COMPRESSOR 1

/* sz/include/sz.h:164-198 -- field order and types are ABI */
typedef struct sz_params
{
	int dataType;
	unsigned int max_quant_intervals;
	unsigned int quantization_intervals;
	unsigned int maxRangeRadius;
	int sol_ID;
	int losslessCompressor;
	int sampleDistance;
	float predThreshold;
	int szMode;
	int gzipMode;
	int errorBoundMode;
	double absErrBound;
	double relBoundRatio;
	double psnr;
	double normErr;
	double pw_relBoundRatio;
	int segment_size;
	int pwr_type;

	int protectValueRange;
	float fmin, fmax;
	double dmin, dmax;

	int snapshotCmprStep;
	int predictionMode;

	int accelerate_pw_rel_compression;
	int plus_bits;

	int randomAccess;
	int withRegression;
} sz_params;

/* sz/include/sz.h:200-209 */
typedef struct sz_metadata
{
	int versionNumber[3];
	int isConstant;
	int isLossless;
	int sizeType;
	size_t dataSeriesLength;
	int defactoNBBins;
	struct sz_params* conf_params;
} sz_metadata;

/* sz/include/sz.h:211-217 */
typedef struct sz_exedata
{
	char optQuantMode;
	int intvCapacity;
	int intvRadius;
	unsigned int SZ_SIZE_TYPE;
} sz_exedata;

/* globals that callers poke directly (example/sz.c:317-338), sz/include/sz.h:232-240 */
extern int versionNumber[4];
extern int dataEndianType;
extern int sysEndianType;
extern sz_params *confparams_cpr;
extern sz_params *confparams_dec;
extern sz_exedata *exe_params;

/* sz/include/sz.h:255-334 */
int SZ_Init(const char *configFilePath);
int SZ_Init_Params(sz_params *params);
size_t computeDataLength(size_t r5, size_t r4, size_t r3, size_t r2, size_t r1);
int computeDimension(size_t r5, size_t r4, size_t r3, size_t r2, size_t r1);
int filterDimension(size_t r5, size_t r4, size_t r3, size_t r2, size_t r1, size_t* correctedDimension);

unsigned char *SZ_compress(int dataType, void *data, size_t *outSize, size_t r5, size_t r4, size_t r3, size_t r2, size_t r1);
unsigned char* SZ_compress_args(int dataType, void *data, size_t *outSize, int errBoundMode, double absErrBound,
double relBoundRatio, double pwrBoundRatio, size_t r5, size_t r4, size_t r3, size_t r2, size_t r1);
int SZ_compress_args2(int dataType, void *data, unsigned char* compressed_bytes, size_t *outSize,
int errBoundMode, double absErrBound, double relBoundRatio, double pwrBoundRatio,
size_t r5, size_t r4, size_t r3, size_t r2, size_t r1);

void *SZ_decompress(int dataType, unsigned char *bytes, size_t byteLength, size_t r5, size_t r4, size_t r3, size_t r2, size_t r1);
size_t SZ_decompress_args(int dataType, unsigned char *bytes, size_t byteLength, void* decompressed_array, size_t r5, size_t r4, size_t r3, size_t r2, size_t r1);

sz_metadata* SZ_getMetadata(unsigned char* bytes);
/* what the reference's command-line tool (example/sz.c:498-845) calls next to the API: sz.c:768, utility.c:156, :216, :236 */
void SZ_printMetadata(sz_metadata* metadata);
int is_lossless_compressed_data(unsigned char* compressedBytes, size_t cmpSize);
uint64_t sz_lossless_decompress65536bytes(int losslessCompressor, unsigned char* compressBytes, uint64_t cmpSize, unsigned char** oriData);
void* detransposeData(void* data, int dataType, size_t r5, size_t r4, size_t r3, size_t r2, size_t r1);
void SZ_Finalize(void);

void convertSZParamsToBytes(sz_params* params, unsigned char* result);
void convertBytesToSZParams(unsigned char* bytes, sz_params* params);

unsigned char* SZ_compress_customize(const char* appName, void* userPara, int dataType, void* data, size_t r5, size_t r4, size_t r3, size_t r2, size_t r1, size_t *outSize, int *status);
unsigned char* SZ_compress_customize_threadsafe(const char* cmprName, void* userPara, int dataType, void* data, size_t r5, size_t r4, size_t r3, size_t r2, size_t r1, size_t *outSize, int *status);
void* SZ_decompress_customize(const char* appName, void* userPara, int dataType, unsigned char* bytes, size_t byteLength, size_t r5, size_t r4, size_t r3, size_t r2, size_t r1, int* status);
void* SZ_decompress_customize_threadsafe(const char* cmprName, void* userPara, int dataType, unsigned char* bytes, size_t byteLength, size_t r5, size_t r4, size_t r3, size_t r2, size_t r1, int *status);

/* ---- additive entry points of this build (not in the reference) ---- */
/* device that SZ_* calls run on (default: env SZ_HIP_DEVICE or 0); call before the first compress */
int SZ_hip_set_device(int device);

/* The reference's OpenMP container for 3-D arrays (sz/include/sz_omp.h:26, :29, :36, :40): same names, arguments and stream as an OpenMP
 * build of libSZ.  r1 is the slowest dimension here, as in sz_omp.c.  `comp_data` of the inverse: the stream behind its first
 * 4 + MetaDataByteLength bytes (example/sz_openmp.c:580).  SZ_hip_set_omp_threads: the box count (omp_get_max_threads() of an OpenMP
 * build; 0 = pick one: boxes of at most 32768 points); the HIP layer's restrictions are in include/szhip.h (szhip_compress_omp). */
void SZ_hip_set_omp_threads(int thread_num);
/* sz_omp.h:44-47 of an OpenMP build (sz_omp.c:14-53): sz_set_num_threads sets the box count like SZ_hip_set_omp_threads; the 1-D / 2-D entry
 * points below are stubs in the reference itself (they return NULL / do nothing, sz_omp.c:56-61, :360-364, :570-576, :866-870) */
void sz_set_num_threads(int nthreads);
int sz_get_max_threads(void);
int sz_get_thread_num(void);
double sz_wtime(void);
unsigned char *SZ_compress_float_1D_MDQ_openmp(float *oriData, size_t r1, double realPrecision, size_t *comp_size);
unsigned char *SZ_compress_float_2D_MDQ_openmp(float *oriData, size_t r1, size_t r2, double realPrecision, size_t *comp_size);
unsigned char *SZ_compress_double_1D_MDQ_openmp(double *oriData, size_t r1, double realPrecision, size_t *comp_size);
unsigned char *SZ_compress_double_2D_MDQ_openmp(double *oriData, size_t r1, size_t r2, double realPrecision, size_t *comp_size);
void decompressDataSeries_float_1D_openmp(float **data, size_t r1, unsigned char *comp_data);
void decompressDataSeries_float_2D_openmp(float **data, size_t r1, size_t r2, unsigned char *comp_data);
void decompressDataSeries_double_1D_openmp(double **data, size_t r1, unsigned char *comp_data);
void decompressDataSeries_double_2D_openmp(double **data, size_t r1, size_t r2, unsigned char *comp_data);
unsigned char *SZ_compress_float_3D_MDQ_openmp(float *oriData, size_t r1, size_t r2, size_t r3, float realPrecision, size_t *comp_size);
unsigned char *SZ_compress_double_3D_MDQ_openmp(double *oriData, size_t r1, size_t r2, size_t r3, double realPrecision, size_t *comp_size);
void decompressDataSeries_float_3D_openmp(float **data, size_t r1, size_t r2, size_t r3, unsigned char *comp_data);
void decompressDataSeries_double_3D_openmp(double **data, size_t r1, size_t r2, size_t r3, unsigned char *comp_data);
/* per-call measurements of the last SZ_compress_args / SZ_decompress on this thread's context; see szhip.h */
struct szhip_stats;
int SZ_hip_last_stats(struct szhip_stats *out);

#ifdef __cplusplus
}
#endif

#endif /* _SZ_H */
