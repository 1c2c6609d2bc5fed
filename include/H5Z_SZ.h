/*
 * H5Z_SZ.h -- the HDF5 filter of the MI355X SZ build (filter id 32017, the one registered for SZ with The HDF Group).
 * Same interface as the reference's hdf5-filter/H5Z-SZ/include/H5Z_SZ.h:16-77: the plugin entry points HDF5 looks for, the
 * init/finalise calls of applications that register the filter themselves, and the helpers that pack dimensions and error
 * bounds into cd_values.  Datasets written through the reference's filter are read through this one and the other way round
 * (the chunk payload is an SZ stream, the cd_values layout is the reference's).  float and double datasets; integer types are
 * outside the scope of this build (the filter refuses them at H5Dcreate time).
 */
#ifndef H5Z_SZ_MI355X_H
#define H5Z_SZ_MI355X_H
#include <stddef.h>
#include <hdf5.h>
#include "sz.h"

#define H5Z_FILTER_SZ 32017
#define MAX_CHUNK_SIZE 4294967295u

#ifdef __cplusplus
extern "C" {
#endif
extern int load_conffile_flag;
extern int init_sz_flag;
extern char cfgFile[256];

int H5Z_SZ_Init(char *cfgFile);                       /* H5Z_SZ.c:39 */
int H5Z_SZ_Init_Params(sz_params *params);            /* :62 */
sz_params *H5Z_SZ_Init_Default(void);                 /* :72 */
int H5Z_SZ_Finalize(void);                            /* :101 */
void SZ_refreshDimForCdArray(int dataType, size_t old_cd_nelmts, unsigned int *old_cd_values, size_t *new_cd_nelmts, unsigned int **new_cd_values,
                             size_t r5, size_t r4, size_t r3, size_t r2, size_t r1);                                                       /* :235 */
void SZ_cdArrayToMetaData(size_t cd_nelmts, const unsigned int cd_values[], int *dimSize, int *dataType, size_t *r5, size_t *r4, size_t *r3,
                          size_t *r2, size_t *r1);                                                                                          /* :137 */
void SZ_copymetaDataToCdArray(size_t *cd_nelmts, unsigned int *cd_values, int dataType, size_t r5, size_t r4, size_t r3, size_t r2, size_t r1); /* :186 */
void SZ_cdArrayToMetaDataErr(size_t cd_nelmts, const unsigned int cd_values[], int *dimSize, int *dataType, size_t *r5, size_t *r4, size_t *r3,
                             size_t *r2, size_t *r1, int *error_bound_mode, double *abs_error, double *rel_error, double *pw_rel_error, double *psnr); /* :111 */
void SZ_errConfigToCdArray(size_t *cd_nelmts, unsigned int **cd_values, int error_bound_mode, double abs_error, double rel_error,
                           double pw_rel_error, double psnr);                                                                               /* :362 */
int checkCDValuesWithErrors(size_t cd_nelmts, const unsigned int cd_values[]);                                                             /* :510 */
#ifdef __cplusplus
}
#endif
#endif
