// szh_fast.h -- the opt-in FAST mode (SZ_HIP_MODE=fast): a feedback-free predict + quantise for HBM speed.
//
// Why it exists: the reference's default quantiser predicts every point from RECONSTRUCTED neighbours, which forces the exact GPU
// path onto a 1 534-step wavefront (k_pencil, szh_pencil.h) and bounds it by latency, far from the HBM roofline.  The reference's
// own OpenMP variant already gives that coupling up between boxes (block-local Lorenzo: SZ_compress_float_3D_MDQ_RA_block,
// sz/src/sz_float.c:4704-5012, driven by sz/src/sz_omp.c:63-358; inverse szd_float.c:2848, sz_omp.c:366-566).  The fast mode gives
// it up between points as well, the way cuSZ's "dual-quantisation" does:
//   1. pre-quantise:  q = rint(x / 2eb) (int32); x' = q * 2eb is within eb of x -- verified per point, a point that fails (or
//      whose q does not fit) is kept verbatim ("raw");
//   2. integer Lorenzo on q over the whole array (zero outside it): delta = q - pred is exact integer arithmetic (mod 2^32), and q
//      depends on nothing but the point's own input, so prediction needs no feedback and every point is independent.  (A
//      tile-local zero halo, as in the OpenMP precedent, was tried first: its face points cost 27 % of the ratio on the S-field.)
//      The kernel works on tiles of 16 x 16 x 64 points and pre-quantises a one-point halo of its neighbours itself;
//   3. code = delta + radius (u16, 2 .. intervals-1); delta outside that range -> code 0, the delta goes to a side list;
//      raw point -> code 1 (its q counts as 0 for its neighbours; its delta and its value go to side lists);
//   4. the codes are Huffman-coded by the same kernels as the exact path (natural order = stream order).
// Inverse: codes -> deltas, then three inclusive scans (mod 2^32) along dim2, dim1, dim0 over the whole array -- the inverse of a
// 3-D Lorenzo difference -- and x' = q * 2eb.
// The container is this library's own ("SZHF" magic): a stock SZ reader rejects it at the version check.  The error bound always
// holds; codes, ratio and PSNR differ slightly from the exact mode (tests/test_fast_mode.py holds the numbers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SZF_TI 16
#define SZF_TJ 16
#define SZF_TK 64
#define SZF_KP (SZF_TK + 2)                 /* LDS row pitch in words: 1 halo column + 64 + 1 pad */
#define SZF_TILE_WORDS ((SZF_TI + 1) * (SZF_TJ + 1) * SZF_KP)
#define SZF_RAW INT32_MIN                   /* LDS marker of a raw point (a legal q is < 2^30 in magnitude) */

struct szf_geom { int r0, r1, r2; int n0, n1, n2; int64_t n; };   // extents, tiles per dimension
static inline szf_geom szf_make_geom(size_t r0, size_t r1, size_t r2)
{
    szf_geom g;
    g.r0 = (int)r0; g.r1 = (int)r1; g.r2 = (int)r2;
    g.n0 = (g.r0 + SZF_TI - 1) / SZF_TI; g.n1 = (g.r1 + SZF_TJ - 1) / SZF_TJ; g.n2 = (g.r2 + SZF_TK - 1) / SZF_TK;
    g.n = (int64_t)r0 * r1 * r2;
    return g;
}

// pre-quantisation of one value; returns false for a raw point
template <class T> __device__ __forceinline__ bool szf_prequant(T x, T recip, T twoeb, T eb, int32_t *q)
{
    const T s = x * recip;
    if (!(s > (T)-1073741824.0 && s < (T)1073741824.0)) { *q = 0; return false; }       // |q| < 2^30 (also rejects NaN)
    const T r = sizeof(T) == 4 ? (T)rintf((float)s) : (T)rint((double)s);
    const int32_t qi = (int32_t)r;
    const T back = (T)qi * twoeb;
    const T err = x - back;
    if (!((err < 0 ? -err : err) <= eb)) { *q = 0; return false; }
    *q = qi;
    return true;
}

// 7-point integer Lorenzo prediction from the LDS image of the tile + its one-point halo (tile point (i,j,k) sits at (i+1, j+1, k+1));
// raw neighbours and points outside the array count as 0
__device__ __forceinline__ uint32_t szf_nb(const int32_t *qs, int i, int j, int k)
{
    const int32_t v = qs[((i + 1) * (SZF_TJ + 1) + (j + 1)) * SZF_KP + (k + 1)];
    return v == SZF_RAW ? 0u : (uint32_t)v;
}
__device__ __forceinline__ uint32_t szf_pred(const int32_t *qs, int i, int j, int k)
{
    return szf_nb(qs, i, j, k - 1) + szf_nb(qs, i, j - 1, k) + szf_nb(qs, i - 1, j, k) - szf_nb(qs, i, j - 1, k - 1) - szf_nb(qs, i - 1, j, k - 1)
         - szf_nb(qs, i - 1, j - 1, k) + szf_nb(qs, i - 1, j - 1, k - 1);
}

// ---- predict + quantise: one workgroup per tile.  Reads the tile once (16-byte vectors along dim2), writes the u16 codes.
// Algorithmic bytes: N * sizeof(T) read + 2 N written.
template <class T>
__global__ __launch_bounds__(256) void k_fast_quant(szf_geom g, const T *__restrict__ data, uint16_t *__restrict__ codes, T eb, int radius)
{
    __shared__ int32_t qs[SZF_TILE_WORDS];
    const int tk = blockIdx.x % g.n2, tj = (blockIdx.x / g.n2) % g.n1, ti = blockIdx.x / (g.n2 * g.n1);
    const int i0 = ti * SZF_TI, j0 = tj * SZF_TJ, k0 = tk * SZF_TK;
    const T twoeb = eb + eb, recip = (T)1 / twoeb;
    constexpr int V = 16 / (int)sizeof(T);                          // values per 16-byte vector
    constexpr int VPR = SZF_TK / V;                                 // vectors per tile row
    const bool vec = (g.r2 % V) == 0 && k0 + SZF_TK <= g.r2;        // rows 16-byte aligned and whole
    // rows i0-1 .. i0+15, j0-1 .. j0+15 (the -1 rows are the halo), columns k0 .. k0+63 as vectors
    for (int v = threadIdx.x; v < (SZF_TI + 1) * (SZF_TJ + 1) * VPR; v += 256) {
        const int row = v / VPR, c = (v - row * VPR) * V;
        const int ih = row / (SZF_TJ + 1), jh = row - ih * (SZF_TJ + 1);        // halo-shifted row indices
        const int gi = i0 + ih - 1, gj = j0 + jh - 1;
        int32_t *dst = qs + row * SZF_KP + 1 + c;
        if (gi >= 0 && gj >= 0 && gi < g.r0 && gj < g.r1) {
            const T *src = data + ((int64_t)gi * g.r1 + gj) * g.r2 + k0 + c;
            T x[V];
            if (vec) { const uint4 w = *reinterpret_cast<const uint4 *>(src); __builtin_memcpy(x, &w, 16); }
            else { for (int e = 0; e < V; ++e) x[e] = k0 + c + e < g.r2 ? src[e] : (T)0; }
            for (int e = 0; e < V; ++e) { int32_t q; dst[e] = szf_prequant<T>(x[e], recip, twoeb, eb, &q) ? q : SZF_RAW; }
        } else { for (int e = 0; e < V; ++e) dst[e] = 0; }
    }
    // the halo column k0-1 of every row
    for (int row = threadIdx.x; row < (SZF_TI + 1) * (SZF_TJ + 1); row += 256) {
        const int ih = row / (SZF_TJ + 1), jh = row - ih * (SZF_TJ + 1);
        const int gi = i0 + ih - 1, gj = j0 + jh - 1;
        int32_t q = 0;
        if (k0 > 0 && gi >= 0 && gj >= 0 && gi < g.r0 && gj < g.r1) {
            int32_t t;
            q = szf_prequant<T>(data[((int64_t)gi * g.r1 + gj) * g.r2 + k0 - 1], recip, twoeb, eb, &t) ? t : SZF_RAW;
        }
        qs[row * SZF_KP] = q;
    }
    __syncthreads();
    for (int v = threadIdx.x; v < SZF_TI * SZF_TJ * (SZF_TK / 4); v += 256) {     // four codes (8 bytes) per store
        const int row = v / (SZF_TK / 4), c = (v - row * (SZF_TK / 4)) * 4;
        const int i = row / SZF_TJ, j = row - i * SZF_TJ;
        if (i0 + i >= g.r0 || j0 + j >= g.r1 || k0 + c >= g.r2) continue;
        uint16_t out[4];
        for (int e = 0; e < 4; ++e) {
            const int32_t q = qs[((i + 1) * (SZF_TJ + 1) + (j + 1)) * SZF_KP + 1 + c + e];
            unsigned code = 1;                                                      // raw
            if (q != SZF_RAW) {
                const int32_t delta = (int32_t)((uint32_t)q - szf_pred(qs, i, j, c + e));
                code = (delta >= 2 - radius && delta < radius) ? (unsigned)(delta + radius) : 0u;
            }
            out[e] = (uint16_t)code;
        }
        uint16_t *dst = codes + ((int64_t)(i0 + i) * g.r1 + (j0 + j)) * g.r2 + k0 + c;
        if ((g.r2 % 4) == 0 && k0 + c + 4 <= g.r2) { uint2 w; __builtin_memcpy(&w, out, 8); *reinterpret_cast<uint2 *>(dst) = w; }
        else { for (int e = 0; e < 4 && k0 + c + e < g.r2; ++e) dst[e] = out[e]; }
    }
}

// delta of ONE point, recomputed from the array (for the rare side-list entries): the same tile-local arithmetic
template <class T> __device__ uint32_t szf_q_at(const szf_geom &g, const T *data, int i, int j, int k, T recip, T twoeb, T eb)
{
    if (i < 0 || j < 0 || k < 0) return 0u;
    int32_t q;
    return szf_prequant<T>(data[((int64_t)i * g.r1 + j) * g.r2 + k], recip, twoeb, eb, &q) ? (uint32_t)q : 0u;
}
template <class T> __device__ int32_t szf_delta_at(const szf_geom &g, const T *data, int64_t p, T eb)
{
    const T twoeb = eb + eb, recip = (T)1 / twoeb;
    const int k = (int)(p % g.r2); const int64_t pj = p / g.r2; const int j = (int)(pj % g.r1), i = (int)(pj / g.r1);
    const uint32_t q = szf_q_at<T>(g, data, i, j, k, recip, twoeb, eb);
    const uint32_t pred = szf_q_at<T>(g, data, i, j, k - 1, recip, twoeb, eb) + szf_q_at<T>(g, data, i, j - 1, k, recip, twoeb, eb)
                        + szf_q_at<T>(g, data, i - 1, j, k, recip, twoeb, eb) - szf_q_at<T>(g, data, i, j - 1, k - 1, recip, twoeb, eb)
                        - szf_q_at<T>(g, data, i - 1, j, k - 1, recip, twoeb, eb) - szf_q_at<T>(g, data, i - 1, j - 1, k, recip, twoeb, eb)
                        + szf_q_at<T>(g, data, i - 1, j - 1, k - 1, recip, twoeb, eb);
    return (int32_t)(q - pred);
}

// ---- side lists, in stream (= natural) order.  counts: codes equal to 0 and to 1 per chunk of 2048 (one launch, both at once)
__global__ __launch_bounds__(256) void k_fast_count(const uint16_t *__restrict__ codes, int64_t n, u64 *cnt0, u64 *cnt1)
{
    __shared__ u64 sh[8];
    const int64_t e0 = (int64_t)blockIdx.x * 2048 + threadIdx.x * 8;
    u64 z = 0;
    for (int q = 0; q < 8; ++q) if (e0 + q < n) { const unsigned c = codes[e0 + q]; z += c == 0 ? 1ull : (c == 1 ? (1ull << 32) : 0ull); }
    u64 tot;
    block_excl_scan_256(z, sh, &tot);
    if (threadIdx.x == 0) { cnt0[blockIdx.x] = tot & 0xffffffffull; cnt1[blockIdx.x] = tot >> 32; }
}
// compress: listA[rank0] = delta of the rank0-th code-0 point; listBd[rank1] = delta and listB[rank1] = value of the rank1-th raw point
template <class T>
__global__ __launch_bounds__(256) void k_fast_lists(szf_geom g, const uint16_t *__restrict__ codes, const u64 *__restrict__ off0, const u64 *__restrict__ off1,
                                                    const T *__restrict__ data, T eb, int32_t *listA, int32_t *listBd, T *listB)
{
    __shared__ u64 sh[8];
    const int64_t e0 = (int64_t)blockIdx.x * 2048 + threadIdx.x * 8;
    unsigned m0 = 0, m1 = 0;
    for (int q = 0; q < 8; ++q) if (e0 + q < g.n) { const unsigned c = codes[e0 + q]; if (c == 0) m0 |= 1u << q; else if (c == 1) m1 |= 1u << q; }
    u64 tot;
    const u64 ex = block_excl_scan_256((u64)__builtin_popcount(m0) | ((u64)__builtin_popcount(m1) << 32), sh, &tot);
    u64 r0 = off0[blockIdx.x] + (ex & 0xffffffffull), r1 = off1[blockIdx.x] + (ex >> 32);
    for (int q = 0; q < 8; ++q) {
        if (m0 >> q & 1) listA[r0++] = szf_delta_at<T>(g, data, e0 + q, eb);
        if (m1 >> q & 1) { listBd[r1] = szf_delta_at<T>(g, data, e0 + q, eb); listB[r1] = data[e0 + q]; ++r1; }
    }
}
// decompress, pass 0 (only when there are side lists): the deltas of the side lists dropped into the workspace at their positions
__global__ __launch_bounds__(256) void k_fast_scatter(const uint16_t *__restrict__ codes, int64_t n, const u64 *__restrict__ off0, const u64 *__restrict__ off1,
                                                      const int32_t *__restrict__ listA, const int32_t *__restrict__ listBd, uint32_t *__restrict__ acc)
{
    __shared__ u64 sh[8];
    const int64_t e0 = (int64_t)blockIdx.x * 2048 + threadIdx.x * 8;
    unsigned m0 = 0, m1 = 0;
    for (int q = 0; q < 8; ++q) if (e0 + q < n) { const unsigned c = codes[e0 + q]; m0 |= (unsigned)(c == 0) << q; m1 |= (unsigned)(c == 1) << q; }
    u64 tot;
    u64 r0 = off0[blockIdx.x] + block_excl_scan_256((u64)__builtin_popcount(m0), sh, &tot);
    u64 r1 = off1[blockIdx.x] + block_excl_scan_256((u64)__builtin_popcount(m1), sh, &tot);
    for (int q = 0; q < 8; ++q) {
        if (m0 >> q & 1) acc[e0 + q] = (uint32_t)listA[r0++];
        if (m1 >> q & 1) acc[e0 + q] = (uint32_t)listBd[r1++];
    }
}
// pass 1: codes -> deltas and the inclusive scan along dim2 in the same pass.  One WAVEFRONT per row of the array, 8 consecutive
// codes per lane and round (one 16-byte load, two 16-byte stores), the running sum carried from round to round.
__global__ __launch_bounds__(256) void k_fast_expand_scan2(szf_geom g, const uint16_t *__restrict__ codes, int radius, uint32_t *acc)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)g.r0 * g.r1) return;
    const int64_t base = row * g.r2;
    const bool vec = (g.r2 & 7) == 0;
    uint32_t carry = 0;
    for (int k0 = 0; k0 < g.r2; k0 += 512) {
        const int k = k0 + lane * 8;
        uint32_t d[8];
        if (vec && k < g.r2) {
            uint16_t c[8];
            const uint4 w = *reinterpret_cast<const uint4 *>(codes + base + k); __builtin_memcpy(c, &w, 16);
            bool side = false;
            for (int e = 0; e < 8; ++e) { d[e] = (uint32_t)((int)c[e] - radius); side |= c[e] < 2; }
            if (side) for (int e = 0; e < 8; ++e) if (c[e] < 2) d[e] = acc[base + k + e];
        } else {
            for (int e = 0; e < 8; ++e) {
                d[e] = 0;
                if (k + e < g.r2) { const unsigned c = codes[base + k + e]; d[e] = c >= 2 ? (uint32_t)((int)c - radius) : acc[base + k + e]; }
            }
        }
        for (int e = 1; e < 8; ++e) d[e] += d[e - 1];
        uint32_t v = d[7];                                                    // inclusive wavefront scan of the lane totals
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
        const uint32_t before = v - d[7] + carry;
        for (int e = 0; e < 8; ++e) d[e] += before;
        if (vec && k < g.r2) {
            uint4 w0, w1; __builtin_memcpy(&w0, d, 16); __builtin_memcpy(&w1, d + 4, 16);
            *reinterpret_cast<uint4 *>(acc + base + k) = w0; *reinterpret_cast<uint4 *>(acc + base + k + 4) = w1;
        } else { for (int e = 0; e < 8; ++e) if (k + e < g.r2) acc[base + k + e] = d[e]; }
        carry += __shfl(v, 63, 64);
    }
}
// pass 2: inclusive scan along dim1, one thread per (i, k) column, coalesced along k, eight independent loads in flight per thread
__global__ __launch_bounds__(256) void k_fast_scan1(szf_geom g, uint32_t *acc)
{
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= (int64_t)g.r0 * g.r2) return;
    const int64_t i = x / g.r2, k = x - i * g.r2;
    uint32_t *c = acc + i * (int64_t)g.r1 * g.r2 + k;
    uint32_t s = 0;
    int j = 0;
    for (; j + 8 <= g.r1; j += 8) {
        uint32_t v[8];
        for (int e = 0; e < 8; ++e) v[e] = c[(int64_t)(j + e) * g.r2];
        for (int e = 0; e < 8; ++e) { s += v[e]; c[(int64_t)(j + e) * g.r2] = s; }
    }
    for (; j < g.r1; ++j) { s += c[(int64_t)j * g.r2]; c[(int64_t)j * g.r2] = s; }
}
// pass 3: inclusive scan along dim0 and x' = q * 2eb, one thread per (j, k) column; raw points receive their value afterwards
template <class T>
__global__ __launch_bounds__(256) void k_fast_scan0_out(szf_geom g, const uint32_t *__restrict__ acc, T *__restrict__ out, T eb)
{
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t plane = (int64_t)g.r1 * g.r2;
    if (x >= plane) return;
    const T twoeb = eb + eb;
    uint32_t s = 0;
    int i = 0;
    for (; i + 8 <= g.r0; i += 8) {
        uint32_t v[8];
        for (int e = 0; e < 8; ++e) v[e] = acc[(int64_t)(i + e) * plane + x];
        for (int e = 0; e < 8; ++e) { s += v[e]; out[(int64_t)(i + e) * plane + x] = (T)(int32_t)s * twoeb; }
    }
    for (; i < g.r0; ++i) { s += acc[(int64_t)i * plane + x]; out[(int64_t)i * plane + x] = (T)(int32_t)s * twoeb; }
}
// the raw values themselves, into their positions
template <class T>
__global__ __launch_bounds__(256) void k_fast_raw(const uint16_t *__restrict__ codes, int64_t n, const u64 *__restrict__ off1, const T *__restrict__ listB, T *out)
{
    __shared__ u64 sh[8];
    const int64_t e0 = (int64_t)blockIdx.x * 2048 + threadIdx.x * 8;
    unsigned m1 = 0;
    for (int q = 0; q < 8; ++q) if (e0 + q < n && codes[e0 + q] == 1) m1 |= 1u << q;
    u64 tot;
    u64 r1 = off1[blockIdx.x] + block_excl_scan_256((u64)__builtin_popcount(m1), sh, &tot);
    for (int q = 0; q < 8; ++q) if (m1 >> q & 1) out[e0 + q] = listB[r1++];
}
