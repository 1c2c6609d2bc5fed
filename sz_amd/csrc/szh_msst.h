// szh_msst.h -- point-wise relative bounds in the reference's DEFAULT ("MSST19", accelerate_pw_rel_compression = 1) form.
//   quantisers   sz/src/sz_float.c:1824-1990 (1-D), :1992-2268 (2-D), :2270-2730 (3-D); doubles sz_double.c:1552, :1721, :1996
//   inverses     sz/src/szd_float.c:1702, :1808, :2129 (szd_double.c likewise); wrappers szd_float_pwr.c:1425-1528
//   range scan   sz/src/dataCompression.c:121-166 (computeRangeSize_float_MSST19), zeros sz_float_pwr.c:2053-2058
//   optimisers   sz/src/sz_float.c:4468, :4518, :4578
// The predictor is a multiplicative Lorenzo stencil on RECONSTRUCTED values, the code the entry of a table indexed by the exponent and
// leading mantissa bits of value / prediction (MultiLevelCacheTableWideInterval.c:53-107, built on the host).
// Here: the element-wise passes around the quantiser, the 1-D chain, and the SECOND mapping of the 2-D / 3-D quantiser -- a sweep hyperplane
// by hyperplane (i + j + k = d), one launch per plane, the dependency being the launch order (SZ_HIP_MSST_SWEEP=1).  The first mapping is the
// wavefront kernel's third quantiser (szh_pencil.h, fmt 2); the two share only the host-built tables and the tests hold them against each
// other through the oracle.  The three reference quantisers spell their arithmetic differently -- the float 3-D one multiplies in double,
// the 2-D one in float, one boundary case of the 3-D compressor has no fabs where its inverse has one -- and each is followed as written.
#pragma once

enum { MS_MINMAG = 0, MS_MINIDX = 1, MS_NEG = 2, MS_RED = 4 };

template <class T> struct msst_bits;
template <> struct msst_bits<float> { using U = unsigned; static __device__ __forceinline__ U of(float v) { return __float_as_uint(v); } static __device__ __forceinline__ float to(U u) { return __uint_as_float(u); } };
template <> struct msst_bits<double> { using U = u64; static __device__ __forceinline__ U of(double v) { return (u64)__double_as_longlong(v); } static __device__ __forceinline__ double to(U u) { return __longlong_as_double((long long)u); } };

// signs from element 1 on (the reference's scan never looks at element 0), and the least magnitude among the non-zero values
template <class T>
__global__ __launch_bounds__(256) void k_msst_scan(const T *__restrict__ data, int64_t n, unsigned char *__restrict__ signs, u64 *red)
{
    using U = typename msst_bits<T>::U;
    u64 mn = ~0ull; unsigned neg = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const T x = data[i];
        const bool ng = i > 0 && x < 0;
        signs[i] = ng ? 1 : 0;
        neg |= ng ? 1u : 0u;
        if (x != 0) { const u64 m = (u64)(msst_bits<T>::of(x) & (U)~((U)1 << (sizeof(T) * 8 - 1))); if (m < mn) mn = m; }
    }
    if (mn != ~0ull) atomicMin((unsigned long long *)&red[MS_MINMAG], (unsigned long long)mn);
    if (neg) atomicOr((unsigned long long *)&red[MS_NEG], 1ull);
}
// ... and the first position that has it (the reference keeps the first one it meets: `fabsf(x) < fabsf(*nearZero)`)
template <class T>
__global__ __launch_bounds__(256) void k_msst_minidx(const T *__restrict__ data, int64_t n, u64 mag, u64 *red)
{
    using U = typename msst_bits<T>::U;
    u64 first = ~0ull;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const u64 m = (u64)(msst_bits<T>::of(data[i]) & (U)~((U)1 << (sizeof(T) * 8 - 1)));
        if (m == mag && (u64)i < first) first = (u64)i;
    }
    if (first != ~0ull) atomicMin((unsigned long long *)&red[MS_MINIDX], (unsigned long long)first);
}
template <class T>
__global__ __launch_bounds__(256) void k_msst_fill(const T *__restrict__ data, int64_t n, T *__restrict__ prep, T zval)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const T x = data[i]; prep[i] = x == 0 ? zval : x; }
}

// the quotient every sample of the interval optimiser takes the logarithm of (the logarithm itself is the host's: glibc's log2 decides
// the histogram bin in the reference).  Sample positions: the lattice of the SZ 1.4 optimisers (k_sample); one slot per possible
// sample, NaN = no sample there.  ndim 1: position 2 + slot * sd.
template <class T>
__global__ __launch_bounds__(256) void k_msst_sample(szh_geom3 G, int ndim, const T *__restrict__ data, int64_t nrows, int sd, int per_row, double *__restrict__ pe)
{
    const double none = __longlong_as_double(0x7ff8000000000001ll);
    if (ndim == 1) {
        const int64_t count = G.n > 2 ? (G.n - 2 + sd - 1) / sd : 0;
        for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < count; k += (int64_t)gridDim.x * 256) {
            const int64_t pos = 2 + k * sd;
            pe[k] = fabs((double)data[pos] / (double)data[pos - 1]);
        }
        return;
    }
    const int64_t rpp = G.g1.count - 1, r2 = G.g2.count;
    const bool two_d = ndim == 2;
    for (int64_t ridx = (int64_t)blockIdx.x * 256 + threadIdx.x; ridx < nrows; ridx += (int64_t)gridDim.x * 256) {
        const int64_t n1 = two_d ? 0 : ridx / rpp + 1, n2 = two_d ? ridx + 1 : ridx - (n1 - 1) * rpp + 1;
        const int64_t c0 = two_d ? szh_sample_col0_2d(n2, sd) : sd - ((n1 + n2) % sd);
        const int64_t origin = n1 * G.d0 + n2 * r2;
        bool live = true;
        for (int m = 0; m < per_row; ++m) {
            const int64_t col = c0 + (int64_t)m * sd, pos = origin + col;
            if ((m > 0 && col >= r2) || pos >= G.n) live = false;
            double v = none;
            if (live) {
                const T *d = data + pos;
                if (two_d) { const T pv = d[-1] + d[-r2] - d[-r2 - 1]; v = fabs((double)(T)(pv / d[0])); }
                else {
                    const int64_t s = G.d0;
                    const T pv = d[-1] + d[-r2] + d[-s] - d[-1 - s] - d[-r2 - 1] - d[-r2 - s] + d[-r2 - s - 1];
                    v = fabs((double)(T)(d[0] / pv));
                }
            }
            pe[ridx * per_row + m] = v;
        }
    }
}

struct msst_tab {
    const double *ptab;          // precisionTable[intervals]
    const uint16_t *cells;       // [(range + 1) << bits]   (compress only)
    u64 base, range;
    int bits, intervals;
};
__device__ __forceinline__ int msst_state(const msst_tab &t, double quotient)
{
    const u64 u = (u64)__double_as_longlong(quotient);
    const u64 e = ((u & 0x7fffffffffffffffull) >> 52) - t.base;
    if (e > t.range) return 0;
    return (int)t.cells[(size_t)(e << t.bits) + (size_t)((u & 0x000fffffffffffffull) >> (52 - t.bits))];
}
// an "exact" value of this path: the leading req_len bits of the value itself (compressSingleFloatValue_MSST19, dataCompression.c:479)
template <class T> __device__ __forceinline__ T msst_keep(T x, int ign_bits);
template <> __device__ __forceinline__ float msst_keep<float>(float x, int ign) { int s = (int)__float_as_uint(x); s = (s >> ign) << ign; return __uint_as_float((unsigned)s); }
template <> __device__ __forceinline__ double msst_keep<double>(double x, int ign) { long long s = __double_as_longlong(x); s = (s >> ign) << ign; return __longlong_as_double(s); }

// one hyperplane a + b + c = d of an r0 x r1 x r2 array (ndim 2: r0 = 1 and the 2-D functions' arithmetic).
//   compress: x (zeros replaced) -> codes, rec.     DEC: codes, rec (exact values already at the code-0 positions) -> rec
template <class T, bool DEC>
__global__ __launch_bounds__(256) void k_msst_plane(int r0, int r1, int r2, int ndim, int d, const T *__restrict__ x, T *rec, uint16_t *codes,
                                                    msst_tab tb, int ign_bits)
{
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int a_lo = d - (r1 - 1) - (r2 - 1) > 0 ? d - (r1 - 1) - (r2 - 1) : 0;
    const int a = a_lo + (int)(tid / r1), b = (int)(tid % r1), c = d - a - b;
    if (a >= r0 || a > d || c < 0 || c >= r2) return;
    const int64_t s1 = r2, s0 = (int64_t)r1 * r2, idx = a * s0 + b * s1 + c;
    if (idx == 0) { if (!DEC) { codes[0] = 0; rec[0] = msst_keep<T>(x[0], ign_bits); } return; }
    T pred; bool use_fabs = true;
    const T *R = rec + idx;
    if (ndim == 2) {                                             // products in T (sz_float.c:2091, :2152)
        if (b == 0) pred = c == 1 ? R[-1] : (T)((T)(R[-1] * R[-1]) / R[-2]);
        else if (c == 0) pred = R[-s1];
        else pred = (T)((T)(R[-1] * R[-s1]) / R[-s1 - 1]);
    } else if (a == 0) {                                         // products in double (sz_float.c:2403, :2488)
        if (b == 0) pred = c == 1 ? R[-1] : (T)((double)R[-1] * (double)R[-1] / (double)R[-2]);
        else if (c == 0) { pred = R[-s1]; use_fabs = DEC; }      // the compressor multiplies the signed prediction here (:2459), the inverse |.| (szd_float.c:2765)
        else pred = (T)((double)R[-1] * (double)R[-s1] / (double)R[-s1 - 1]);
    } else {
        if (b == 0) pred = c == 0 ? R[-s0] : (T)((double)R[-1] * (double)R[-s0] / (double)R[-s0 - 1]);
        else if (c == 0) pred = (T)((double)R[-s1] * (double)R[-s0] / (double)R[-s0 - s1]);
        else pred = (T)((double)R[-1] * (double)R[-s1] * (double)R[-s0] * (double)R[-s0 - s1 - 1]
                        / ((double)R[-s1 - 1] * (double)R[-s0 - s1] * (double)R[-s0 - 1]));
    }
    if (DEC) {
        const int t = codes[idx];
        if (t) rec[idx] = (T)(fabs((double)pred) * tb.ptab[t < tb.intervals ? t : 0]);          // a code beyond the table: a broken stream
    } else {
        const T v = x[idx];
        const int state = msst_state(tb, (double)(T)(v / pred));
        codes[idx] = (uint16_t)state;
        rec[idx] = state ? (T)((use_fabs ? fabs((double)pred) : (double)pred) * tb.ptab[state]) : msst_keep<T>(v, ign_bits);
    }
}

// 1-D: one chain through the previous reconstruction (sz_float.c:1824-1990; inverse szd_float.c:1702-1806): a single lane walks it.
// Values are fetched 16 at a time ahead of the chain; the tables sit in LDS when they fit.
template <class T, bool DEC>
__global__ __launch_bounds__(64) void k_msst_chain_1d(const T *__restrict__ x, T *out, uint16_t *codes, int64_t n, msst_tab tb, int intervals, int64_t cells_n,
                                                      int ign_bits, int tabs_in_lds)
{
    SZH_DYN_SMEM(msst_lds_raw);
    double *msst_lds = (double *)msst_lds_raw;
    msst_tab t = tb;
    if (tabs_in_lds) {
        double *lp = msst_lds; uint16_t *lc = (uint16_t *)(msst_lds + intervals);
        for (int i = threadIdx.x; i < intervals; i += 64) lp[i] = tb.ptab[i];
        if (!DEC) for (int64_t i = threadIdx.x; i < cells_n; i += 64) lc[i] = tb.cells[i];
        __syncthreads();
        t.ptab = lp; t.cells = lc;
    }
    if (threadIdx.x != 0) return;
    T pred = 0;
    for (int64_t base = 0; base < n; base += 16) {
        T xv[16]; int cv[16];
        const int m = (int)(n - base < 16 ? n - base : 16);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (DEC) { cv[u] = u < m ? (int)codes[base + u] : 1; xv[u] = (u < m && cv[u] == 0) ? out[base + u] : (T)0; }
            else xv[u] = u < m ? x[base + u] : (T)1;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (u >= m) break;
            const int64_t i = base + u;
            if (DEC) {
                if (cv[u]) { pred = (T)(fabs((double)pred) * t.ptab[cv[u] < intervals ? cv[u] : 0]); out[i] = pred; }
                else pred = xv[u];
            } else {
                int state = 0;
                if (i >= 2) state = msst_state(t, (double)(T)(xv[u] / pred));
                codes[i] = (uint16_t)state;
                pred = state ? (T)((double)pred * t.ptab[state]) : msst_keep<T>(xv[u], ign_bits);
            }
        }
    }
}

// after the inverse: values below the threshold are the zeros, then the signs (szd_float_pwr.c:1430-1455; both branches as written)
template <class T>
__global__ __launch_bounds__(256) void k_msst_post(const T *__restrict__ in, int64_t n, T threshold, const unsigned char *__restrict__ signs, T *__restrict__ out)
{
    using U = typename msst_bits<T>::U;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        T v = in[i];
        if (signs) {
            if (v < threshold && v >= 0) v = 0;
            else if (signs[i]) v = msst_bits<T>::to(msst_bits<T>::of(v) | ((U)1 << (sizeof(T) * 8 - 1)));
        } else if (v < threshold) v = 0;
        out[i] = v;
    }
}
