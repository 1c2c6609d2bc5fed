// szh_pwr.h -- the element-wise passes of the point-wise-relative modes (PW_REL and its AND/OR combinations) in their log-domain
// form: SZ_compress_args_float_NoCkRngeNoGzip_{1D,2D,3D}_pwr_pre_log (sz/src/sz_float_pwr.c:1791-1975; doubles sz_double_pwr.c:1781-1965)
// and decompressDataSeries_float_{1D,2D,3D}_pwr_pre_log (sz/src/szd_float_pwr.c:1353-1422; doubles szd_double_pwr.c).
//   compress:  l = log2|x| (0 stays 0 for now), sign bytes, reductions {min, max of l over x != 0; min, max of the array l; any x < 0};
//              then every x == 0 gets l = min_log - 2.0001 * realPrecision, and l goes through the SZ 1.4 quantiser with the
//              absolute bound realPrecision = log2(1 + ratio) - margin;
//   decompress: x' = l < threshold ? 0 : exp2(l), negated where the sign byte says so.
// log2 / exp2 are evaluated in double and narrowed (what the reference's C does for float data, where `log2` is the double function).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { PWR_MINLOG = 0, PWR_MAXLOG = 1, PWR_MINALL = 2, PWR_MAXALL = 3, PWR_NEG = 4, PWR_NONZERO = 5, PWR_RED = 8 };

template <class T>
__global__ __launch_bounds__(256) void k_pwr_log(const T *__restrict__ data, int64_t n, T *__restrict__ logd, unsigned char *__restrict__ signs, u64 *red)
{
    __shared__ u64 sh[4][6];
    u64 mnl = ~0ull, mxl = 0, mna = ~0ull, mxa = 0, neg = 0, nz = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const T x = data[i];
        const bool s = x < 0;
        T l = s ? -x : x;
        if (l > 0) {
            l = (T)log2((double)l);
            const u64 e = ord_enc(l);
            mnl = e < mnl ? e : mnl; mxl = e > mxl ? e : mxl; nz = 1;
        }
        logd[i] = l; signs[i] = (unsigned char)s; neg |= (u64)s;
        const u64 e = ord_enc(l);
        mna = e < mna ? e : mna; mxa = e > mxa ? e : mxa;
    }
    mnl = wave_min_u64(mnl); mxl = wave_max_u64(mxl); mna = wave_min_u64(mna); mxa = wave_max_u64(mxa); neg = wave_max_u64(neg); nz = wave_max_u64(nz);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[w][0] = mnl; sh[w][1] = mxl; sh[w][2] = mna; sh[w][3] = mxa; sh[w][4] = neg; sh[w][5] = nz; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 4; ++q) {
            sh[0][0] = sh[q][0] < sh[0][0] ? sh[q][0] : sh[0][0]; sh[0][1] = sh[q][1] > sh[0][1] ? sh[q][1] : sh[0][1];
            sh[0][2] = sh[q][2] < sh[0][2] ? sh[q][2] : sh[0][2]; sh[0][3] = sh[q][3] > sh[0][3] ? sh[q][3] : sh[0][3];
            sh[0][4] |= sh[q][4]; sh[0][5] |= sh[q][5];
        }
        atomicMin(&red[PWR_MINLOG], sh[0][0]); atomicMax(&red[PWR_MAXLOG], sh[0][1]);
        atomicMin(&red[PWR_MINALL], sh[0][2]); atomicMax(&red[PWR_MAXALL], sh[0][3]);
        if (sh[0][4]) atomicMax(&red[PWR_NEG], 1ull);
        if (sh[0][5]) atomicMax(&red[PWR_NONZERO], 1ull);
    }
}

template <class T>
__global__ __launch_bounds__(256) void k_pwr_zero(const T *__restrict__ data, int64_t n, T *__restrict__ logd, T zval)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        if (data[i] == 0) logd[i] = zval;
}

template <class T>
__global__ __launch_bounds__(256) void k_pwr_exp(const T *__restrict__ logd, int64_t n, T threshold, const unsigned char *__restrict__ signs, T *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const T l = logd[i];
        T r = l < threshold ? (T)0 : (T)exp2((double)l);
        if (signs && signs[i]) r = -r;
        out[i] = r;
    }
}
