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
#define SZF_KP (SZF_TK + 4)                 /* LDS row pitch in words: the halo column at word 3, the 64 tile columns 16-byte aligned from word 4 */
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

// ---- predict + quantise: one workgroup per tile.  Phase 1 pre-quantises the tile and its one-point halo (rows i0-1, j0-1, column
// k0-1) into LDS, 16-byte loads along dim2.  Phase 2: a thread owns 8 consecutive columns of one (j) row and walks 8 planes of dim0,
// keeping the previous plane's two rows in registers, so every LDS word is read about twice instead of eight times (the first
// version, one point at a time, was bound by LDS reads).  8 codes = one 16-byte store.  Raw neighbours and points outside the array
// count as 0.  Tiles are handed out so that consecutive tiles (neighbours along dim2 / dim1, who share halo rows) run on the same
// XCD and find each other's rows in its L2.  Algorithmic bytes: N * sizeof(T) read + 2 N written.
__device__ __forceinline__ void szf_row9(const int32_t *row, int kg, int32_t *v)
{
    v[0] = row[3 + kg * 8];
    const int4 a = *reinterpret_cast<const int4 *>(row + 4 + kg * 8), b = *reinterpret_cast<const int4 *>(row + 8 + kg * 8);
    v[1] = a.x; v[2] = a.y; v[3] = a.z; v[4] = a.w; v[5] = b.x; v[6] = b.y; v[7] = b.z; v[8] = b.w;
}
template <class T>
__global__ __launch_bounds__(256) void k_fast_quant(szf_geom g, const T *__restrict__ data, uint16_t *__restrict__ codes, T eb, int radius)
{
    __shared__ __attribute__((aligned(16))) int32_t qs[SZF_TILE_WORDS];
    const int64_t ntiles = (int64_t)g.n0 * g.n1 * g.n2, per = (ntiles + 7) / 8;
    const int64_t tile = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    const int tk = (int)(tile % g.n2), tj = (int)((tile / g.n2) % g.n1), ti = (int)(tile / ((int64_t)g.n2 * g.n1));
    const int i0 = ti * SZF_TI, j0 = tj * SZF_TJ, k0 = tk * SZF_TK;
    const T twoeb = eb + eb, recip = (T)1 / twoeb;
    constexpr int V = 16 / (int)sizeof(T);                          // values per 16-byte vector
    constexpr int VPR = SZF_TK / V;                                 // vectors per tile row
    const bool vec = (g.r2 % V) == 0 && k0 + SZF_TK <= g.r2;        // rows 16-byte aligned and whole
    constexpr int NVEC = (SZF_TI + 1) * (SZF_TJ + 1) * VPR;
    // the halo column k0-1 of every row: loads issued first, used last
    T hx[2] = {(T)0, (T)0}; bool hok[2];
    for (int b = 0; b < 2; ++b) {
        const int row = threadIdx.x + 256 * b;
        const int ih = row / (SZF_TJ + 1), jh = row - ih * (SZF_TJ + 1);
        const int gi = i0 + ih - 1, gj = j0 + jh - 1;
        hok[b] = row < (SZF_TI + 1) * (SZF_TJ + 1) && k0 > 0 && gi >= 0 && gj >= 0 && gi < g.r0 && gj < g.r1;
        if (hok[b]) hx[b] = data[((int64_t)gi * g.r1 + gj) * g.r2 + k0 - 1];
    }
    if (vec) {
        // whole, aligned rows: eight 16-byte loads in flight per thread before the first is used (one load per round trip made this
        // kernel latency-bound)
        constexpr int B = 8;
        for (int v0 = threadIdx.x; v0 < NVEC; v0 += 256 * B) {
            uint4 w[B]; bool ok[B];
            for (int b = 0; b < B; ++b) {
                const int v = v0 + 256 * b;
                const int row = v / VPR, c = (v - row * VPR) * V;
                const int ih = row / (SZF_TJ + 1), jh = row - ih * (SZF_TJ + 1);        // halo-shifted row indices
                const int gi = i0 + ih - 1, gj = j0 + jh - 1;
                ok[b] = v < NVEC && gi >= 0 && gj >= 0 && gi < g.r0 && gj < g.r1;
                w[b] = make_uint4(0, 0, 0, 0);
                if (ok[b]) w[b] = *reinterpret_cast<const uint4 *>(data + ((int64_t)gi * g.r1 + gj) * g.r2 + k0 + c);
            }
            for (int b = 0; b < B; ++b) {
                const int v = v0 + 256 * b;
                if (v >= NVEC) break;
                const int row = v / VPR, c = (v - row * VPR) * V;
                T x[V]; int32_t q[V];
                __builtin_memcpy(x, &w[b], 16);
                for (int e = 0; e < V; ++e) { int32_t t; q[e] = !ok[b] ? 0 : (szf_prequant<T>(x[e], recip, twoeb, eb, &t) ? t : SZF_RAW); }
                int32_t *dst = qs + row * SZF_KP + 4 + c;
                if (V == 4) { int4 o; __builtin_memcpy(&o, q, 16); *reinterpret_cast<int4 *>(dst) = o; }
                else { for (int e = 0; e < V; ++e) dst[e] = q[e]; }
            }
        }
    } else {
        for (int v = threadIdx.x; v < NVEC; v += 256) {
            const int row = v / VPR, c = (v - row * VPR) * V;
            const int ih = row / (SZF_TJ + 1), jh = row - ih * (SZF_TJ + 1);
            const int gi = i0 + ih - 1, gj = j0 + jh - 1;
            int32_t *dst = qs + row * SZF_KP + 4 + c;
            for (int e = 0; e < V; ++e) {
                int32_t q = 0, t;
                if (gi >= 0 && gj >= 0 && gi < g.r0 && gj < g.r1 && k0 + c + e < g.r2)
                    q = szf_prequant<T>(data[((int64_t)gi * g.r1 + gj) * g.r2 + k0 + c + e], recip, twoeb, eb, &t) ? t : SZF_RAW;
                dst[e] = q;
            }
        }
    }
    for (int b = 0; b < 2; ++b) {
        const int row = threadIdx.x + 256 * b;
        if (row < (SZF_TI + 1) * (SZF_TJ + 1)) {
            int32_t q = 0, t;
            if (hok[b]) q = szf_prequant<T>(hx[b], recip, twoeb, eb, &t) ? t : SZF_RAW;
            qs[row * SZF_KP + 3] = q;
        }
    }
    __syncthreads();
    const int kg = threadIdx.x & 7, j = (threadIdx.x >> 3) & 15, ih0 = (threadIdx.x >> 7) * 8;
    const int gj = j0 + j, gk = k0 + kg * 8;
    // rows of the LDS image: point (i, j) of the tile is row (i + 1) * 17 + (j + 1); `a` = row j-1, `b` = row j, masked for the neighbour role
    int32_t pa[9], pb[9], ca[9], cb[9];
    szf_row9(qs + ((ih0 + 0) * (SZF_TJ + 1) + j) * SZF_KP, kg, pa);
    szf_row9(qs + ((ih0 + 0) * (SZF_TJ + 1) + j + 1) * SZF_KP, kg, pb);
    for (int e = 0; e < 9; ++e) { pa[e] = pa[e] == SZF_RAW ? 0 : pa[e]; pb[e] = pb[e] == SZF_RAW ? 0 : pb[e]; }
    const bool inside_jk = gj < g.r1 && gk < g.r2;
    for (int s = 0; s < 8; ++s) {
        const int i = ih0 + s;
        szf_row9(qs + ((i + 1) * (SZF_TJ + 1) + j) * SZF_KP, kg, ca);
        szf_row9(qs + ((i + 1) * (SZF_TJ + 1) + j + 1) * SZF_KP, kg, cb);
        unsigned rawmask = 0;
        for (int e = 0; e < 9; ++e) { ca[e] = ca[e] == SZF_RAW ? 0 : ca[e]; if (cb[e] == SZF_RAW) { cb[e] = 0; rawmask |= 1u << e; } }
        uint16_t out[8];
        for (int e = 0; e < 8; ++e) {
            const uint32_t pred = (uint32_t)cb[e] + (uint32_t)ca[e + 1] + (uint32_t)pb[e + 1] - (uint32_t)ca[e] - (uint32_t)pb[e] - (uint32_t)pa[e + 1] + (uint32_t)pa[e];
            const int32_t delta = (int32_t)((uint32_t)cb[e + 1] - pred);
            unsigned code = (delta >= 2 - radius && delta < radius) ? (unsigned)(delta + radius) : 0u;
            if (rawmask >> (e + 1) & 1) code = 1;
            out[e] = (uint16_t)code;
        }
        if (inside_jk && i0 + i < g.r0) {
            uint16_t *dst = codes + ((int64_t)(i0 + i) * g.r1 + gj) * g.r2 + gk;
            if ((g.r2 & 7) == 0) { uint4 w; __builtin_memcpy(&w, out, 16); *reinterpret_cast<uint4 *>(dst) = w; }
            else { for (int e = 0; e < 8 && gk + e < g.r2; ++e) dst[e] = out[e]; }
        }
        for (int e = 0; e < 9; ++e) { pa[e] = ca[e]; pb[e] = cb[e]; }
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

// bit masks of the codes equal to 0 (m0) and to 1 (m1) among the eight codes at e0 (a multiple of 8): one 16-byte load
__device__ __forceinline__ void szf_masks(const uint16_t *__restrict__ codes, int64_t e0, int64_t n, unsigned *m0, unsigned *m1)
{
    uint16_t c[8];
    if (e0 + 8 <= n) { const uint4 w = *reinterpret_cast<const uint4 *>(codes + e0); __builtin_memcpy(c, &w, 16); }
    else { for (int q = 0; q < 8; ++q) c[q] = e0 + q < n ? codes[e0 + q] : (uint16_t)2; }
    unsigned a = 0, b = 0;
    for (int q = 0; q < 8; ++q) { a |= (unsigned)(c[q] == 0) << q; b |= (unsigned)(c[q] == 1) << q; }
    *m0 = a; *m1 = b;
}

// ---- side lists, in stream (= natural) order.  counts: codes equal to 0 and to 1 per chunk of 2048 (one launch, both at once)
__global__ __launch_bounds__(256) void k_fast_count(const uint16_t *__restrict__ codes, int64_t n, u64 *cnt0, u64 *cnt1)
{
    __shared__ u64 sh[8];
    const int64_t e0 = (int64_t)blockIdx.x * 2048 + threadIdx.x * 8;
    unsigned m0, m1;
    szf_masks(codes, e0, n, &m0, &m1);
    const u64 z = (u64)__builtin_popcount(m0) | ((u64)__builtin_popcount(m1) << 32);
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
    unsigned m0, m1;
    szf_masks(codes, e0, g.n, &m0, &m1);
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
    unsigned m0, m1;
    szf_masks(codes, e0, n, &m0, &m1);
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
    unsigned m0, m1;
    szf_masks(codes, e0, n, &m0, &m1);
    u64 tot;
    u64 r1 = off1[blockIdx.x] + block_excl_scan_256((u64)__builtin_popcount(m1), sh, &tot);
    for (int q = 0; q < 8; ++q) if (m1 >> q & 1) out[e0 + q] = listB[r1++];
}
