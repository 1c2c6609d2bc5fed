// szh_core.h -- per-block / per-sample arithmetic of the SZ 2.1 3-D path that has no loop-carried
// dependence across blocks (so it maps to one thread per block / per sample row on the GPU).
// Host+device inline functions; the HIP kernels and the test-only CPU simulator both call them.
//
// Reference (paths relative to the reference tree):
//   regression fit        sz/src/sz_float.c:6598-6633      (double: sz_double.c:5961-6012)
//   predictor selection   sz/src/sz_float.c:7083-7123, with mean :6747-6786
//   interval sampling     sz/src/sz_float.c:6396-6523      (double: sz_double.c:5773)
//   2-D: fit sz/src/sz_float.c:5569-5605, selection :6003-6028, sampling :5405-5515 (double: sz_double.c:4953-4989, :5384-5409, :4790)
#pragma once
#include "szh_geom.h"

// ---- regression fit of one block.  A(i,j,k) returns the original value at block-local (i,j,k).
// The accumulation order (sum_y over k, then j, then i; fz over the whole block) is part of the result.
// Rows are fetched into registers first (SZH_MAX_BLK independent reads in flight) and then accumulated in the reference's
// order; fetching inside the dependent chain would pay one LDS round trip per element.
#define SZH_MAX_BLK 12 /* widest block: a dimension of 7..11 forms a single block (sz.h:93-123) */
struct szh_no_visit { template <class T> SZH_HD void operator()(T) const {} };
// visit(c) sees every value of the block once (the kernel folds the array's min/max into this pass)
template <class T, class Acc, class Visit = szh_no_visit>
SZH_HD void szh_fit_block(const Acc &A, int s0, int s1, int s2, T *coef4, const Visit &visit = Visit())
{
    T fx = 0, fy = 0, fz = 0, f = 0;
    for (int i = 0; i < s0; ++i) {
        T sum_x = 0;
        for (int j = 0; j < s1; ++j) {
            T row[SZH_MAX_BLK];
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int k = 0; k < SZH_MAX_BLK; ++k) row[k] = k < s2 ? A(i, j, k) : (T)0;
            T sum_y = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int k = 0; k < SZH_MAX_BLK; ++k) {
                if (k < s2) {
                    const T c = row[k];
                    visit(c);
                    sum_y += c;
                    fz += c * (T)k;
                }
            }
            fy += sum_y * (T)j;
            sum_x += sum_y;
        }
        fx += sum_x * (T)i;
        f += sum_x;
    }
    const T coeff = (T)(1.0 / (double)((int64_t)s0 * s1 * s2));
    const T a = ((T)2 * fx / (T)(s0 - 1) - f) * (T)6 * coeff / (T)(s0 + 1);
    const T b = ((T)2 * fy / (T)(s1 - 1) - f) * (T)6 * coeff / (T)(s1 + 1);
    const T c = ((T)2 * fz / (T)(s2 - 1) - f) * (T)6 * coeff / (T)(s2 + 1);
    const T d = f * coeff - ((T)(s0 - 1) * a / (T)2 + (T)(s1 - 1) * b / (T)2 + (T)(s2 - 1) * c / (T)2);
    coef4[0] = a; coef4[1] = b; coef4[2] = c; coef4[3] = d;
}

// ---- predictor selection of one block: returns 1 if the regression plane wins.
// A(i,j,k) as above; all sampled stencil points have i,j,k >= 1 so they stay inside the block.
template <class T, class Acc>
SZH_HD int szh_select_block(const Acc &A, int s0, int s1, int s2, const T *coef4, T noise, int use_mean, T mean)
{
    T err_sz = 0, err_reg = 0;
    int bs = s0 < s1 ? s0 : s1; if (s2 < bs) bs = s2;
    for (int i = 1; i < bs; ++i) {
        const int bmi = bs - i;
        for (int q = 0; q < 4; ++q) {
            const int j = (q & 2) ? bmi : i;
            const int k = (q & 1) ? bmi : i;
            const T x = A(i, j, k);
            const T psz = A(i, j, k - 1) + A(i, j - 1, k) + A(i - 1, j, k) - A(i, j - 1, k - 1) - A(i - 1, j, k - 1)
                          - A(i - 1, j - 1, k) + A(i - 1, j - 1, k - 1);
            const T preg = coef4[0] * (T)i + coef4[1] * (T)j + coef4[2] * (T)k + coef4[3];
            const T e1 = szh_abs(psz - x) + noise;
            if (use_mean) { const T e2 = szh_abs(mean - x); err_sz += (e1 < e2 ? e1 : e2); }
            else err_sz += e1;
            err_reg += szh_abs(preg - x);
        }
    }
    return err_reg < err_sz ? 1 : 0;
}

// ---- 2-D: regression fit of one block.  A(i,j) returns the original value at block-local (i,j); coef3 = {a, b, c}.
template <class T, class Acc, class Visit = szh_no_visit>
SZH_HD void szh_fit_block_2d(const Acc &A, int s1, int s2, T *coef3, const Visit &visit = Visit())
{
    T fx = 0, fy = 0, f = 0;
    for (int i = 0; i < s1; ++i) {
        T sum_x = 0;
        for (int j0 = 0; j0 < s2; j0 += 8) {
            T row[8];
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int e = 0; e < 8; ++e) row[e] = j0 + e < s2 ? A(i, j0 + e) : (T)0;
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int e = 0; e < 8; ++e) {
                if (j0 + e < s2) {
                    const T c = row[e];
                    visit(c);
                    sum_x += c;
                    fy += c * (T)(j0 + e);
                }
            }
        }
        fx += sum_x * (T)i;
        f += sum_x;
    }
    const T coeff = (T)(1.0 / (double)((int64_t)s1 * s2));
    const T a = ((T)2 * fx / (T)(s1 - 1) - f) * (T)6 * coeff / (T)(s1 + 1);
    const T b = ((T)2 * fy / (T)(s2 - 1) - f) * (T)6 * coeff / (T)(s2 + 1);
    const T c = f * coeff - ((T)(s1 - 1) * a / (T)2 + (T)(s2 - 1) * b / (T)2);
    coef3[0] = a; coef3[1] = b; coef3[2] = c;
}

// ---- 2-D: predictor selection of one block (1 = the regression plane wins).  The second sample of each pair evaluates the
// plane at row i - 1 while the point itself lies in row i: that is what the reference computes (sz_float.c:6022).
template <class T, class Acc>
SZH_HD int szh_select_block_2d(const Acc &A, int s1, int s2, const T *coef3, T noise)
{
    T err_sz = 0, err_reg = 0;
    const int bs = s1 < s2 ? s1 : s2;
    for (int i = 1; i < bs; ++i) {
        {
            const T x = A(i, i);
            const T psz = A(i, i - 1) + A(i - 1, i) - A(i - 1, i - 1);
            const T preg = coef3[0] * (T)i + coef3[1] * (T)i + coef3[2];
            err_sz += szh_abs(psz - x) + noise;
            err_reg += szh_abs(preg - x);
        }
        {
            const int bmi = bs - i;
            const T x = A(i, bmi);
            const T psz = A(i, bmi - 1) + A(i - 1, bmi) - A(i - 1, bmi - 1);
            const T preg = coef3[0] * (T)(i - 1) + coef3[1] * (T)bmi + coef3[2];
            err_sz += szh_abs(psz - x) + noise;
            err_reg += szh_abs(preg - x);
        }
    }
    return err_reg < err_sz ? 1 : 0;
}

// ---- strided "mean" samples of the interval optimiser (sz_float.c:6405-6419) in closed form:
// the walk adds `md` per step and steps back by one whenever a running offset passes r2 / r1*r2.
struct szh_meanwalk { int64_t md, c1, c2, len; };
SZH_HD szh_meanwalk szh_make_meanwalk(int64_t len, int64_t r1r2, int64_t r2, int64_t md)
{
    szh_meanwalk w; w.md = md; w.len = len;
    w.c1 = (r2 + md - 1) / md;       // steps until the dim2 offset reaches r2
    w.c2 = (r1r2 + md - 1) / md;     // steps until the plane offset reaches r1*r2
    return w;
}
SZH_HD int64_t szh_meanwalk_pos(const szh_meanwalk &w, int64_t m) { return m * w.md - m / w.c1 - m / w.c2; }

// ---- lattice samples of the interval optimiser (sz_float.c:6442-6485).
// Logical row (n1,n2), n1>=1, 1<=n2<=r1-1, starts at origin n1*r1r2 + n2*r2 and is sampled at columns
// c0 + m*sd with c0 = sd - ((n1+n2) % sd); at least one sample per row (column c0 even if c0 >= r2),
// further ones while the column is < r2; the walk as a whole stops at the first position >= len.
// 2-D (sz_float.c:5441-5475): logical row n >= 1 is array row n, first column sd - 1 for n = 1 and sd - (n % sd) after it; the
// same "at least one sample per row" and "stop at the first position >= len" rules.  `r12` = 0 selects the 3-point stencil.
// (unsigned long)rq clamped to the table, as the reference's optimisers do it (sz_float.c:4664-4667, :5092-5095).  Outside that type's range the
// conversion is whatever the x86-64 build of the reference does -- and arrays with fill values (1e30, 9.97e36) get there with ordinary
// bounds: a NaN becomes 2^63 (clamped to the last bin), a quotient of 2^64 or more, infinity included, becomes 0 (cvttsd2si of
// x - 2^63 gives 2^63 again, and the xor with 2^63 that completes the unsigned conversion leaves 0): the FIRST bin
SZH_HD unsigned szh_radius_index(double rq, unsigned max_radius)
{
    if (!(rq < 18446744073709551616.0)) return rq != rq ? max_radius - 1 : 0u;
    return rq >= (double)max_radius ? max_radius - 1 : (unsigned)rq;
}

// the bin of a sampled value in the optimiser's histogram around the mean (sz_float.c:6466-6476)
template <class T>
SZH_HD int szh_freq_index(T x, T mean, double ebD)
{
    const T mean_diff = x - mean;
    const double fq = (double)mean_diff / ebD;
    // (ptrdiff_t)fq: outside the int64 range (or NaN) the x86 conversion of the reference yields
    // INT64_MIN, which lands in bin 0 after the clamp below
    int64_t fi;
    if (!(fq < 9.2233720368547758e18 && fq >= -9.2233720368547758e18)) fi = -(((int64_t)1) << 62);
    else fi = (int64_t)fq;
    if (!(mean_diff > 0)) fi -= 1;
    fi += 4096;
    return fi <= 0 ? 0 : (fi >= 8192 ? 8191 : (int)fi);
}

template <class T>
SZH_HD void szh_sample_point(const T *data, int64_t pos, int64_t r2, int64_t r12, double ebD, T mean,
                             unsigned max_radius, unsigned *radius_index, int *freq_index, int *within_eb)
{
    const T *d = data + pos;
    const T pred = r12 == 0 ? d[-1] + d[-r2] - d[-r2 - 1]
                            : d[-1] + d[-r2] + d[-r12] - d[-1 - r12] - d[-r2 - 1] - d[-r2 - r12] + d[-r2 - r12 - 1];
    const T pred_err = szh_abs((T)(pred - *d));
    *within_eb = ((double)pred_err < ebD) ? 1 : 0;
    double rq = ((double)pred_err / ebD + 1) / 2;
    const unsigned ri = szh_radius_index(rq, max_radius);
    *radius_index = ri;
    *freq_index = szh_freq_index<T>(*d, mean, ebD);
}

// number of logical sample rows the sequential walk of the reference visits before its first
// position >= len (rows are linearised as (n1-1)*(r1-1) + (n2-1)); host-side helper.
SZH_HD int64_t szh_sample_col0_2d(int64_t n, int sd) { return n == 1 ? sd - 1 : sd - (n % sd); }
inline int64_t szh_sample_row_limit(const szh_geom3 &G, int sd)
{
    const int64_t r1 = G.g1.count, r2 = G.g2.count, r0 = G.g0.count;
    if (G.ndim == 2) {   // logical rows 1 .. r1-1 are linearised as n - 1; positions n*r2 + col0(n) increase with n
        int64_t lim = r1 - 1;
        while (lim > 0 && lim * r2 + szh_sample_col0_2d(lim, sd) >= G.n) --lim;
        return lim;
    }
    const int64_t rows_per_plane = r1 - 1;
    const int64_t total = (r0 - 1) * rows_per_plane;
    if (total <= 0) return 0;
    // only rows whose single overflow sample (column c0 >= r2) can reach len matter: scan back from the end
    int64_t first_bad = total;
    int64_t scan = (int64_t)sd / (r2 > 0 ? r2 : 1) + 4 + rows_per_plane; // generous tail
    int64_t lo = total - scan; if (lo < 0) lo = 0;
    for (int64_t ridx = lo; ridx < total; ++ridx) {
        const int64_t n1 = ridx / rows_per_plane + 1, n2 = ridx % rows_per_plane + 1;
        const int64_t c0 = sd - ((n1 + n2) % sd);
        const int64_t pos = n1 * G.d0 + n2 * r2 + c0;
        if (pos >= G.n) { first_bad = ridx; break; }
    }
    return first_bad;
}
