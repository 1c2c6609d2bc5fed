// szh_fittile.h -- regression fit + predictor selection + value range of the SZ 2.1 path from LDS tiles (round 6).
//
// Reference: fit sz/src/sz_float.c:6598-6633 (double: sz_double.c:5961-6012), selection :7083-7123 with the mean shortcut :6747-6786, range
// computeRangeSize_float (dataCompression.c:97-113).  Same arithmetic, same order as szh_core.h's szh_fit_block / szh_select_block (which
// k_fit_select runs one thread per block straight from HBM: 216 four-byte loads a thread, a wavefront's load touching a dozen lines -- 0.24 ms
// at 512^3, bound by the address units, 2.3 TB/s).
//
// Here: a WAVEFRONT per (block column (b0, b1), segment of `segb` blocks along the contiguous dimension), a lane per block.  The column's planes
// come one after the other (a plane of a block column: s1 rows): whole 16-byte row pieces, every lane loading, into one of two LDS buffers; the
// next plane's pieces are on their way in registers while the lanes walk the current one.  A lane keeps the four moment sums of ITS block across
// the planes in the reference's order (sum_y over k, fz over every element, fy / sum_x over j, fx / f over i).  The selection's 4 (b - 1) sample
// stencils need planes i and i - 1: both buffers.  Its regression errors need the coefficients, which exist only after the last plane: the
// sampled values wait in registers; the Lorenzo errors are summed on the way, in the reference's order.  The range is reduced over the pieces
// as they arrive (every value once).
//
// Covers 3-D arrays whose blocks are at most FT_BMAX wide in every dimension and whose rows start on 16-byte boundaries; k_fit_select takes the rest.
#pragma once

namespace szh_ft {
constexpr int BMAX = 7;          // widest block (every extent >= 42 has blocks of 6 or 7; smaller extents may have wider ones)
constexpr int LPR = 1;           // 16-byte pieces per row and lane
template <class T> struct shape {
    static constexpr int VT = 16 / (int)sizeof(T);              // values per piece
    static constexpr int SEGB = sizeof(T) == 4 ? 32 : 16;       // blocks per segment: a row piece of at most 64 * LPR pieces (half of the lanes hold a block: the pass
                                                                // is bound by memory and by how many wavefronts a CU's LDS takes, not by the lanes' arithmetic)
    static constexpr int WIDTH = SEGB * BMAX + 2 * VT;          // values of a buffered row at most (vector boundaries on both sides)
    static constexpr int PITCH = WIDTH + VT;                    // (a multiple of VT: 16-byte LDS stores)
    static constexpr int BUF = BMAX * PITCH;                    // values per buffer
    static_assert((WIDTH + VT - 1) / VT <= 64 * LPR && PITCH % VT == 0, "row pieces per lane");
};
template <class T> inline bool applies(const szh_geom3 &G, const void *base)
{
    if (G.ndim != 3) return false;
    if (G.g0.early > BMAX || G.g1.early > BMAX || G.g2.early > BMAX) return false;
    if (G.g0.count < 2 || G.g1.count < 2 || G.g2.count < 2) return false;
    if ((G.g2.count % shape<T>::VT) != 0 || ((uintptr_t)base & 15) != 0) return false;
    return (double)G.g0.num * G.g1.num * ((G.g2.num + shape<T>::SEGB - 1) / shape<T>::SEGB) < 2.0e9;
}
}

template <class T>
__global__ __launch_bounds__(64) void k_fit_tile(szh_geom3 G, const T *__restrict__ data, T *coef, uint8_t *blk_lor, T noise, int use_mean, T mean, u64 *minmax, int nseg)
{
    using namespace szh_ft;
    typedef shape<T> S;
    constexpr int VT = S::VT;
    __shared__ __attribute__((aligned(16))) T buf[2][S::BUF];
    const int lane = (int)threadIdx.x;
    const int col = (int)(blockIdx.x / (unsigned)nseg), segi = (int)(blockIdx.x - (unsigned)col * (unsigned)nseg);
    const int b0 = col / G.g1.num, b1 = col - b0 * G.g1.num;
    const int s0 = szh_blk_size(G.g0, b0), s1 = szh_blk_size(G.g1, b1), o0 = szh_blk_start(G.g0, b0), o1 = szh_blk_start(G.g1, b1);
    const int bkbeg = segi * S::SEGB, bkend = bkbeg + S::SEGB < G.g2.num ? bkbeg + S::SEGB : G.g2.num;
    const int kbeg = szh_blk_start(G.g2, bkbeg), kend = bkend < G.g2.num ? szh_blk_start(G.g2, bkend) : G.g2.count;
    const int ka = kbeg / VT * VT, kb = (kend + VT - 1) / VT * VT, nvec = (kb - ka) / VT;
    // the lane's block
    const int b2 = bkbeg + lane;
    const bool has = b2 < bkend;
    const int s2 = has ? szh_blk_size(G.g2, b2) : 1, k0 = has ? szh_blk_start(G.g2, b2) - ka : 0;
    int bs = s0 < s1 ? s0 : s1; if (s2 < bs) bs = s2;

    // the range: minimum / maximum of the values that belong to this segment (a NaN takes no part: computeRangeSize_float's comparisons are false for it, and
    // fmin / fmax pass it over); ordered encodings only once per lane, at the end
    const T qnan = (T)__builtin_nanf("");
    T vmin = qnan, vmax = qnan;
    uint4 pf[BMAX][LPR];
    // plane i of the column's rows: the lane's pieces of every row (row j: pieces lane, lane + 64)
    auto fetch = [&](int i) {
#pragma unroll
        for (int j = 0; j < BMAX; ++j)
#pragma unroll
            for (int q = 0; q < LPR; ++q) {
                const int v = lane + 64 * q;
                pf[j][q] = uint4{0u, 0u, 0u, 0u};
                if (j < s1 && v < nvec) pf[j][q] = *reinterpret_cast<const uint4 *>(data + (int64_t)(o0 + i) * G.d0 + (int64_t)(o1 + j) * G.d1 + ka + v * VT);
            }
    };
    // ... into buffer `w`; the range over the values that belong to this segment (a NaN takes no part: computeRangeSize_float's comparisons are false for it)
    auto place = [&](int w) {
#pragma unroll
        for (int j = 0; j < BMAX; ++j)
#pragma unroll
            for (int q = 0; q < LPR; ++q) {
                const int v = lane + 64 * q;
                if (j < s1 && v < nvec) {
                    *reinterpret_cast<uint4 *>(&buf[w][j * S::PITCH + v * VT]) = pf[j][q];
                    T x[VT]; __builtin_memcpy(x, &pf[j][q], 16);
                    const int e_lo = kbeg - (ka + v * VT), e_hi = kend - (ka + v * VT);      // (only a row's first and last pieces reach beyond the segment)
#pragma unroll
                    for (int e = 0; e < VT; ++e) {
                        const T xv = (e >= e_lo && e < e_hi) ? x[e] : qnan;
                        vmin = sizeof(T) == 8 ? (T)__builtin_fmin((double)vmin, (double)xv) : (T)__builtin_fminf((float)vmin, (float)xv);
                        vmax = sizeof(T) == 8 ? (T)__builtin_fmax((double)vmax, (double)xv) : (T)__builtin_fmaxf((float)vmax, (float)xv);
                    }
                }
            }
    };
    fetch(0);
    T fx = 0, fy = 0, fz = 0, f = 0, err_sz = 0;
    T xs[4 * (BMAX - 1)];                                            // the selection's sampled values, in its order
#pragma unroll
    for (int e = 0; e < 4 * (BMAX - 1); ++e) xs[e] = 0;
#pragma unroll
    for (int i = 0; i < BMAX; ++i) {
        if (i >= s0) break;                                          // (uniform)
        const int w = i & 1;
        place(w);
        __syncthreads();
        if (i + 1 < s0) fetch(i + 1);
        if (has) {
            const T *P = &buf[w][k0], *Q = &buf[w ^ 1][k0];           // plane i, plane i - 1 of the lane's block
            // fit: the moment sums of this plane (sz_float.c:6604-6626)
            T sum_x = 0;
#pragma unroll
            for (int j = 0; j < BMAX; ++j) {
                if (j < s1) {
                    T row[BMAX];
#pragma unroll
                    for (int k = 0; k < BMAX; ++k) row[k] = k < s2 ? P[j * S::PITCH + k] : (T)0;
                    T sum_y = 0;
#pragma unroll
                    for (int k = 0; k < BMAX; ++k) if (k < s2) { const T c = row[k]; sum_y += c; fz += c * (T)k; }
                    fy += sum_y * (T)j;
                    sum_x += sum_y;
                }
            }
            fx += sum_x * (T)i;
            f += sum_x;
            // selection: the four sample stencils of this plane (sz_float.c:7088-7113)
            if (i >= 1 && i < bs) {
                const int bmi = bs - i;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = (q & 2) ? bmi : i, k = (q & 1) ? bmi : i;
                    const T x = P[j * S::PITCH + k];
                    const T psz = P[j * S::PITCH + k - 1] + P[(j - 1) * S::PITCH + k] + Q[j * S::PITCH + k] - P[(j - 1) * S::PITCH + k - 1] - Q[j * S::PITCH + k - 1]
                                  - Q[(j - 1) * S::PITCH + k] + Q[(j - 1) * S::PITCH + k - 1];
                    const T e1 = szh_abs(psz - x) + noise;
                    if (use_mean) { const T e2 = szh_abs(mean - x); err_sz += (e1 < e2 ? e1 : e2); }
                    else err_sz += e1;
                    xs[(i - 1) * 4 + q] = x;                               // (the planes' loop is unrolled: a compile-time place)
                }
            }
        }
        __syncthreads();                                             // (plane i - 1's buffer is free for plane i + 1)
    }
    if (has) {
        const T coeff = (T)(1.0 / (double)((int64_t)s0 * s1 * s2));
        const T a = ((T)2 * fx / (T)(s0 - 1) - f) * (T)6 * coeff / (T)(s0 + 1);
        const T b = ((T)2 * fy / (T)(s1 - 1) - f) * (T)6 * coeff / (T)(s1 + 1);
        const T c = ((T)2 * fz / (T)(s2 - 1) - f) * (T)6 * coeff / (T)(s2 + 1);
        const T d = f * coeff - ((T)(s0 - 1) * a / (T)2 + (T)(s1 - 1) * b / (T)2 + (T)(s2 - 1) * c / (T)2);
        T err_reg = 0;
#pragma unroll
        for (int i = 1; i < BMAX; ++i) {
            if (i < bs) {
                const int bmi = bs - i;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = (q & 2) ? bmi : i, k = (q & 1) ? bmi : i;
                    const T preg = a * (T)i + b * (T)j + c * (T)k + d;
                    err_reg += szh_abs(preg - xs[(i - 1) * 4 + q]);
                }
            }
        }
        const int64_t blk = ((int64_t)b0 * G.g1.num + b1) * G.g2.num + b2;
        coef[blk] = a; coef[G.nblocks + blk] = b; coef[2 * G.nblocks + blk] = c; coef[3 * G.nblocks + blk] = d;
        blk_lor[blk] = err_reg < err_sz ? 0 : 1;
    }
    u64 lmin = vmin == vmin ? ord_enc(vmin) : ~0ull, lmax = vmax == vmax ? ord_enc(vmax) : 0ull;
    lmin = wave_min_u64(lmin); lmax = wave_max_u64(lmax);
    // (fifteen thousand wavefronts and two words: only a wavefront that improves on what it sees sends its atomic -- what it sees may be old, i.e. too wide, never too narrow)
    if (lane == 0) {
        if (lmin < *(volatile u64 *)&minmax[0]) atomicMin(&minmax[0], lmin);
        if (lmax > *(volatile u64 *)&minmax[1]) atomicMax(&minmax[1], lmax);
    }
}
