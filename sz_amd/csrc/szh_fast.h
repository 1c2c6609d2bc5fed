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

// =====================================================================================================================
// Round 3: the two-pass form of the compressor (same stream, byte for byte; the passes above remain for shapes it does not take).
// The quantiser is feedback-free, so recomputing a code is cheaper than storing it and reading it back four times:
//   pass A  k_fast_stat : pre-quantise + predict, code histogram (LDS), side-list counts per UNIT -- no code array is written.
//                         This is the predict+quantise kernel of the mode: N * sizeof(T) algorithmic bytes read, nothing but a few
//                         counters written.
//   (host: code book; device meanwhile: scan of the unit counts)
//   pass B  k_fast_pack : pre-quantise + predict again, every unit's codes Huffman-packed into a private slot + its bit length; the
//                         side lists written at their ranks
//   (scan of the unit bit lengths)
//   pass C  k_fast_compact: the slots shifted bit-exactly to their places in the one MSB-first stream of Huffman.c:205-308
// A UNIT is 64 consecutive symbols of one row (the k extent of a tile), so units in (i, j, k-tile) order are in stream order.
// Tiles are 8 x 16 x 64 here (41.6 KB of LDS: three workgroups per CU); a thread owns 8 consecutive columns of a row and walks 4 planes.
#define SZG_TI 8
#define SZG_ROWS ((SZG_TI + 1) * (SZF_TJ + 1))
#define SZG_TILE_WORDS (SZG_ROWS * SZF_KP)
#define SZG_HBINS 256                      /* LDS histogram window: codes radius - 128 .. radius + 127; the rest goes to global atomics */
#define SZG_SLOT_WORDS 64                  /* slot of a unit: 64 symbols x at most 32 bits */
#ifdef SZH_HIPSIM
#define SZG_SHR(v, o) __shfl_up((v), (o), 64)
#else
/* value of lane - o (o = 1, 2, 4) inside a row of 16 lanes: a DPP move instead of a trip through the LDS crossbar */
#define SZG_SHR(v, o) ((unsigned)__builtin_amdgcn_update_dpp(0, (int)(v), 0x110 + (o), 0xf, 0xf, true))
#endif
#ifdef SZH_HIPSIM
#define SZG_WAVE_SYNC() ((void)__all(1))   /* the shim's lanes are fibres: re-converge the wavefront where lock-step execution would */
#else
#define SZG_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

struct szg_geom { int r0, r1, r2, n0, n1, n2; int64_t ntiles; };
static inline szg_geom szg_make_geom(size_t r0, size_t r1, size_t r2)
{
    szg_geom g; g.r0 = (int)r0; g.r1 = (int)r1; g.r2 = (int)r2;
    g.n0 = (g.r0 + SZG_TI - 1) / SZG_TI; g.n1 = (g.r1 + SZF_TJ - 1) / SZF_TJ; g.n2 = (g.r2 + SZF_TK - 1) / SZF_TK;
    g.ntiles = (int64_t)g.n0 * g.n1 * g.n2;
    return g;
}
// tile of this workgroup (XCD-aware hand-out as in k_fast_quant) -> false if the index is past the end
__device__ __forceinline__ bool szg_tile_of_block(const szg_geom &g, int &i0, int &j0, int &k0, int64_t vb = -1)
{
    if (vb < 0) vb = blockIdx.x;
    const int64_t per = (g.ntiles + 7) / 8;
    const int64_t tile = (vb & 7) * per + (vb >> 3);
    if (tile >= g.ntiles) return false;
    if (g.ntiles < 0x7fffffff) {                                          // 32-bit divisions (scalar unit: ~35 instructions each, ~250 in 64 bits)
        const unsigned t = (unsigned)tile, tk = t % (unsigned)g.n2, tq = t / (unsigned)g.n2;
        k0 = (int)tk * SZF_TK; j0 = (int)(tq % (unsigned)g.n1) * SZF_TJ; i0 = (int)(tq / (unsigned)g.n1) * SZG_TI;
        return true;
    }
    k0 = (int)(tile % g.n2) * SZF_TK; j0 = (int)((tile / g.n2) % g.n1) * SZF_TJ; i0 = (int)(tile / ((int64_t)g.n2 * g.n1)) * SZG_TI;
    return true;
}
// phase 1: the tile and its one-point halo pre-quantised into LDS (row (ih, jh) = point (i0 + ih - 1, j0 + jh - 1); halo column at word 3).
// In two halves so that a persistent workgroup can have the NEXT tile's loads in flight while it works on the present one:
// szg_tile_load requests everything (SZG_NLD 16-byte vectors and one halo value per thread, registers), szg_tile_commit pre-quantises
// and stores the LDS image.
template <class T> __device__ __forceinline__ int32_t szg_prequant(T x, T recip, T twoeb, T eb, bool &ok)
{   // szf_prequant without branches: the same q when representable, SZF_RAW (and ok = false) otherwise
    const T s = x * recip;
    T r = sizeof(T) == 4 ? (T)rintf((float)s) : (T)rint((double)s);
    const T lim = (T)1073741824.0;
#ifndef SZH_HIPSIM
    if (sizeof(T) == 4) r = (T)__builtin_amdgcn_fmed3f((float)r, -1073741824.0f, 1073741824.0f);   // the cast below stays defined (NaN -> a limit)
    else
#endif
    r = (T)fmin(fmax((double)r, -1073741824.0), 1073741824.0);
    const int32_t qi = (int32_t)r;
    const T err = x - (T)qi * twoeb;
    ok = (s > -lim && s < lim) && ((err < 0 ? -err : err) <= eb);
    return ok ? qi : SZF_RAW;
}
// The loads are raw buffer loads from a resource that begins at the tile's first plane: a row outside the array gets an out-of-range
// offset and reads 0 (= the value whose pre-quantised form is 0, what the halo outside the array has to be), so there is no branch
// around a load and any row alignment goes as 16-byte vectors.  Planes too large for 32-bit offsets inside a tile (9 planes >= 2 GB)
// take plain loads at commit time instead (szg_big).
template <class T> struct szg_regs { szh_io::v4u w[(SZG_ROWS * (SZF_TK / (16 / (int)sizeof(T))) + 255) / 256]; T hx; };
__host__ __device__ __forceinline__ bool szg_big(const szg_geom &g, size_t esz) { return (uint64_t)(SZG_TI + 1) * (uint64_t)g.r1 * (uint64_t)g.r2 * esz >= 0x7ff00000ull; }
template <class T>
__device__ __forceinline__ void szg_tile_load(const szg_geom &g, const T *__restrict__ data, int i0, int j0, int k0, szg_regs<T> &R)
{
    constexpr int V = 16 / (int)sizeof(T), VPR = SZF_TK / V, NVEC = SZG_ROWS * VPR, NLD = (NVEC + 255) / 256;
    if (szg_big(g, sizeof(T))) return;
    const int ib = i0 > 0 ? i0 - 1 : 0;
    const int64_t base_el = (int64_t)ib * g.r1 * g.r2;
    const uint64_t span = (uint64_t)((int64_t)g.r0 * g.r1 * g.r2 - base_el) * sizeof(T);
    const szh_io::rsrc_t rs = szh_io::make_rsrc(data + base_el, span > 0x7ffffff0ull ? 0x7ffffff0u : (unsigned)span);
    R.hx = (T)0;
    {
        const int row = threadIdx.x;
        const int ih = row / (SZF_TJ + 1), jh = row - ih * (SZF_TJ + 1);
        const int gi = i0 + ih - 1, gj = j0 + jh - 1;
        if (row < SZG_ROWS && k0 > 0 && gi >= 0 && gj >= 0 && gi < g.r0 && gj < g.r1) R.hx = data[((int64_t)gi * g.r1 + gj) * g.r2 + k0 - 1];
    }
#pragma unroll
    for (int b = 0; b < NLD; ++b) {
        const int v = threadIdx.x + 256 * b;
        const int row = v / VPR, c = (v - row * VPR) * V;
        const int ih = row / (SZF_TJ + 1), jh = row - ih * (SZF_TJ + 1);
        const int gi = i0 + ih - 1, gj = j0 + jh - 1;
        const bool ok = v < NVEC && gi >= 0 && gj >= 0 && gi < g.r0 && gj < g.r1 && k0 + c < g.r2;
        const unsigned off = (unsigned)(((gi - ib) * g.r1 + gj) * g.r2 + k0 + c) * (unsigned)sizeof(T);
        R.w[b] = szh_io::bload16(rs, ok ? off : 0xffffffffu);
    }
}
template <class T>
__device__ __forceinline__ void szg_tile_commit(const szg_geom &g, const T *__restrict__ data, int i0, int j0, int k0, T eb, const szg_regs<T> &R, int32_t *qs,
                                                unsigned *rawflag = nullptr, bool sync = true)
{   // *rawflag (LDS, zero beforehand) is set when the image holds a raw point; sync = false: the caller has the closing barrier
    const T twoeb = eb + eb, recip = (T)1 / twoeb;
    constexpr int V = 16 / (int)sizeof(T), VPR = SZF_TK / V, NVEC = SZG_ROWS * VPR, NLD = (NVEC + 255) / 256;
    const bool big = szg_big(g, sizeof(T));
    const bool ktail = k0 + SZF_TK > g.r2;
    bool allok = true;
#pragma unroll
    for (int b = 0; b < NLD; ++b) {
        const int v = threadIdx.x + 256 * b;
        if (v >= NVEC) break;
        const int row = v / VPR, c = (v - row * VPR) * V;
        T x[V]; int32_t q[V];
        if (!big) {
            const unsigned w4[4] = {R.w[b].x, R.w[b].y, R.w[b].z, R.w[b].w};
            if (sizeof(T) == 4) { for (int e = 0; e < V; ++e) { float f; __builtin_memcpy(&f, &w4[e], 4); x[e] = (T)f; } }
            else { for (int e = 0; e < V; ++e) { const u64 u = (u64)w4[2 * e] | ((u64)w4[2 * e + 1] << 32); double d; __builtin_memcpy(&d, &u, 8); x[e] = (T)d; } }
        }
        else {
            const int ih = row / (SZF_TJ + 1), jh = row - ih * (SZF_TJ + 1);
            const int gi = i0 + ih - 1, gj = j0 + jh - 1;
            const bool ok = gi >= 0 && gj >= 0 && gi < g.r0 && gj < g.r1;
            for (int e = 0; e < V; ++e) x[e] = ok && k0 + c + e < g.r2 ? data[((int64_t)gi * g.r1 + gj) * g.r2 + k0 + c + e] : (T)0;
        }
        for (int e = 0; e < V; ++e) {
            bool ok;
            q[e] = szg_prequant<T>(x[e], recip, twoeb, eb, ok);
            if (ktail && k0 + c + e >= g.r2) { q[e] = 0; ok = true; }
            allok = allok && ok;
        }
        int32_t *dst = qs + row * SZF_KP + 4 + c;
        if (V == 4) { int4 o; __builtin_memcpy(&o, q, 16); *reinterpret_cast<int4 *>(dst) = o; }
        else { for (int e = 0; e < V; ++e) dst[e] = q[e]; }
    }
    if (threadIdx.x < SZG_ROWS) {
        T hx = R.hx;
        if (big) {
            const int ih = threadIdx.x / (SZF_TJ + 1), jh = threadIdx.x - ih * (SZF_TJ + 1);
            const int gi = i0 + ih - 1, gj = j0 + jh - 1;
            hx = (k0 > 0 && gi >= 0 && gj >= 0 && gi < g.r0 && gj < g.r1) ? data[((int64_t)gi * g.r1 + gj) * g.r2 + k0 - 1] : (T)0;
        }
        bool ok;
        qs[threadIdx.x * SZF_KP + 3] = szg_prequant<T>(hx, recip, twoeb, eb, ok);
        allok = allok && ok;
    }
    if (rawflag && !allok) *rawflag = 1u;
    if (sync) __syncthreads();
}
template <class T>
__device__ __forceinline__ void szg_stage_tile(const szg_geom &g, const T *__restrict__ data, int i0, int j0, int k0, T eb, int32_t *qs)
{
    szg_regs<T> R;
    szg_tile_load<T>(g, data, i0, j0, k0, R);
    szg_tile_commit<T>(g, data, i0, j0, k0, eb, R, qs);
}
// phase 2: the thread's 8 columns (kg) of row j in 4 consecutive planes from ih0; f(i, codes[8]) per plane.  Same arithmetic as k_fast_quant.
template <class F>
__device__ __forceinline__ void szg_walk(const int32_t *qs, int radius, int ih0, int j, int kg, F &&f)
{
    int32_t pa[9], pb[9], ca[9], cb[9];
    szf_row9(qs + ((ih0 + 0) * (SZF_TJ + 1) + j) * SZF_KP, kg, pa);
    szf_row9(qs + ((ih0 + 0) * (SZF_TJ + 1) + j + 1) * SZF_KP, kg, pb);
    for (int e = 0; e < 9; ++e) { pa[e] = pa[e] == SZF_RAW ? 0 : pa[e]; pb[e] = pb[e] == SZF_RAW ? 0 : pb[e]; }
    for (int s = 0; s < SZG_TI / 2; ++s) {
        const int i = ih0 + s;
        szf_row9(qs + ((i + 1) * (SZF_TJ + 1) + j) * SZF_KP, kg, ca);
        szf_row9(qs + ((i + 1) * (SZF_TJ + 1) + j + 1) * SZF_KP, kg, cb);
        unsigned rawmask = 0;
        for (int e = 0; e < 9; ++e) { ca[e] = ca[e] == SZF_RAW ? 0 : ca[e]; if (cb[e] == SZF_RAW) { cb[e] = 0; rawmask |= 1u << e; } }
        unsigned out[8]; int32_t dl[8];
        for (int e = 0; e < 8; ++e) {
            const uint32_t pred = (uint32_t)cb[e] + (uint32_t)ca[e + 1] + (uint32_t)pb[e + 1] - (uint32_t)ca[e] - (uint32_t)pb[e] - (uint32_t)pa[e + 1] + (uint32_t)pa[e];
            const int32_t delta = (int32_t)((uint32_t)cb[e + 1] - pred);
            unsigned code = (delta >= 2 - radius && delta < radius) ? (unsigned)(delta + radius) : 0u;
            if (rawmask >> (e + 1) & 1) code = 1;
            out[e] = code; dl[e] = delta;
        }
        f(i, out, dl);
        for (int e = 0; e < 9; ++e) { pa[e] = ca[e]; pb[e] = cb[e]; }
    }
}

// ---- pass A.  ucnt[unit] += (number of code-0 symbols) | (number of code-1 symbols) << 32, unit = ((i * r1 + j) * n2 + k-tile);
// ucnt is ZERO beforehand (a unit without side-list entries -- nearly all -- is not touched).
//
// The kernel is bound by instruction issue, not by HBM (measured: 110 instructions per point at 0.49 ms; 39 lane-operations per
// point is what 0.134 ms allows), so the common case has its own lean path:
//   * an INTERIOR tile without raw points (flag set by the commit) needs no bounds tests and no raw masks; its deltas come from
//     row differences, 3.25 subtractions per point: u = (row j) - (row j-1) per plane, t = u(plane i) - u(plane i-1),
//     delta[e] = t[e+1] - t[e] (the same integer sum as the 7-term predictor, modulo 2^32);
//   * the SZG_PW = 8 symbols around the radius are counted in REGISTERS, one byte each of a 64-bit word per tile (a thread sees 32
//     codes per tile), widened to 16-bit fields per tile and handed to the workgroup's LDS window every SZG_FLUSH tiles;
//   * everything else (a few per cent on smooth data) is collected as a bit mask per plane and handled in a short divergent loop:
//     LDS window of SZG_HBINS bins, the global histogram beyond it, side-list counts for the unpredictable.
// Edge tiles and tiles with raw points take the general walk (szg_walk).  Device-scope atomics on a handful of hot words were the
// earlier forms' undoing (per-tile flushes of ~10 hot bins: 1 ms; hist[0]/hist[1] from wherever code 0/1 occurred: 1 ms).
#define SZG_PW 8
#define SZG_CHUNK 2                        /* tiles per ticket: the kernel ends within one chunk time of the last ticket */
#define SZG_FLUSH 30                       /* tiles between two hand-overs of the register counters: 30 x 32 x 64 lanes < 2^16 */
// ---- pass A.  ucnt[unit] = (number of code-0 symbols) | (number of code-1 symbols) << 32, unit = ((i * r1 + j) * n2 + k-tile)
// Histogram: smooth data puts nearly every code on a handful of symbols around `radius`, and LDS atomics on a handful of words
// serialise (4 replicas per bin: 1.24 ms at 512^3; per-thread LDS words: 1.3 ms -- dependent read-modify-writes under divergent
// masks).  Every thread therefore counts the SZG_PW = 8 symbols around the radius in REGISTERS (16-bit fields of two 64-bit words:
// a thread sees 32 codes per tile, a wavefront 2048), the wavefronts add their fields up and one lane hands them to a
// workgroup-wide window of SZG_HBINS bins in LDS; codes outside the 8 go to that window by LDS atomics, beyond it to the global
// histogram directly (both rare).
#define SZG_PW 8
// the lean walk's row: u[e] = row_j[e] - row_{j-1}[e] for the thread's 9 columns
__device__ __forceinline__ void szg_rowdiff9(const int32_t *qs, int ih, int j, int kg, uint32_t *u)
{
    int32_t a[9], b[9];
    szf_row9(qs + (ih * (SZF_TJ + 1) + j) * SZF_KP, kg, a);
    szf_row9(qs + (ih * (SZF_TJ + 1) + j + 1) * SZF_KP, kg, b);
#pragma unroll
    for (int e = 0; e < 9; ++e) u[e] = (uint32_t)b[e] - (uint32_t)a[e];
}
// the eight deltas of a plane as 16-bit halves of four words (saturated: a delta outside 16 bits is unpredictable at any radius),
// so that a run-time element can be picked with three selects and a bit-field extract.  (A select tree over the elements of a
// register array is folded into a dynamically indexed private array, which hipcc then places in LDS: 8 KB per workgroup.)
__device__ __forceinline__ unsigned szg_pack2(int32_t a, int32_t b)
{
    const int32_t lo = a < -32768 ? -32768 : (a > 32767 ? 32767 : a), hi = b < -32768 ? -32768 : (b > 32767 ? 32767 : b);
    return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16);
}
__device__ __forceinline__ int32_t szg_pick8(unsigned p0, unsigned p1, unsigned p2, unsigned p3, unsigned sel)
{
    const unsigned a = sel & 2u ? p1 : p0, b = sel & 2u ? p3 : p2, w = sel & 4u ? b : a;
    return (int32_t)(w << (sel & 1u ? 0 : 16)) >> 16;
}
template <class T>
__global__ __launch_bounds__(256, 3) void k_fast_stat(szg_geom g, const T *__restrict__ data, T eb, int radius, unsigned nsym, unsigned *hist, u64 *ucnt, unsigned *ticket)
{
    // PERSISTENT: a few workgroups per CU walk the tiles and keep their histogram window in LDS until the end.
    __shared__ __attribute__((aligned(16))) int32_t qs[SZG_TILE_WORDS];
    __shared__ unsigned lh[SZG_HBINS];
    __shared__ unsigned rawflag[2];
    for (int b = threadIdx.x; b < SZG_HBINS; b += 256) lh[b] = 0;
    if (threadIdx.x < 2) rawflag[threadIdx.x] = 0;
    const int kg = threadIdx.x & 7, j = (threadIdx.x >> 3) & 15, ih0 = (threadIdx.x >> 7) * (SZG_TI / 2);
    const int lo = radius - SZG_HBINS / 2, plo = radius - SZG_PW / 2;
    u64 zall = 0;                      // this thread's codes 0 (low half) and 1 (high half): the side lists' sizes
    u64 acc_e = 0, acc_o = 0;          // register counters, 16-bit fields: symbols plo + 0, 2, 4, 6 / plo + 1, 3, 5, 7
    int since = 0, parity = 0;
    auto hand_over = [&]() {
        acc_e = wave_sum_u64(acc_e); acc_o = wave_sum_u64(acc_o);
        if ((threadIdx.x & 63) == 0) {
            for (int b = 0; b < SZG_PW; ++b) {
                const unsigned v = (unsigned)(((b & 1 ? acc_o : acc_e) >> ((b >> 1) * 16)) & 0xffffu);
                if (v && plo + b >= lo && plo + b < lo + SZG_HBINS) atomicAdd(&lh[plo + b - lo], v);
            }
        }
        acc_e = acc_o = 0; since = 0;
    };
    auto count_other = [&](unsigned code, u64 &z) {                     // a symbol outside the register window
        if (code < 2u) z += code == 0u ? 1ull : 1ull << 32;
        else { const unsigned w = code - (unsigned)lo; if (w < (unsigned)SZG_HBINS) atomicAdd(&lh[w], 1u); else atomicAdd(&hist[code], 1u); }
    };
    // Tiles are handed out in CHUNKS of SZG_CHUNK consecutive ones (natural order: neighbours along dim2) from one counter: a fixed
    // assignment left the slowest workgroup 60 % behind the fastest (lifetimes 161 .. 256 us at 512^3), chunks of 8 still 87 us.
    // A workgroup holds its chunk and the next one's first tile (loaded ahead), no more: what it holds when the counter runs dry is the tail.  The next tile's loads are in flight while this one is walked (registers; the LDS
    // image is written when the walk is over).
    szg_regs<T> R;
    __shared__ unsigned s_chunk;
    const unsigned nchunk = (unsigned)((g.ntiles + SZG_CHUNK - 1) / SZG_CHUNK);
    if (threadIdx.x == 0) s_chunk = atomicAdd(ticket, 1u);
    __syncthreads();
    unsigned chunk = s_chunk, r = 0, pend = 0;
    int ni0 = 0, nj0 = 0, nk0 = 0;
    auto tile_at = [&](unsigned c, unsigned rr) -> bool {
        const int64_t tile = (int64_t)c * SZG_CHUNK + rr;
        if (c >= nchunk || tile >= g.ntiles) return false;
        const unsigned t = (unsigned)tile, tk = t % (unsigned)g.n2, tq = t / (unsigned)g.n2;
        nk0 = (int)tk * SZF_TK; nj0 = (int)(tq % (unsigned)g.n1) * SZF_TJ; ni0 = (int)(tq / (unsigned)g.n1) * SZG_TI;
        return true;
    };
    bool have = tile_at(chunk, 0);
    if (have) szg_tile_load<T>(g, data, ni0, nj0, nk0, R);
    __syncthreads();
#ifdef SZG_DBG_RES
    const unsigned res_t0 = (unsigned)wall_clock64(); unsigned res_n = 0;
#endif
#ifdef SZG_DBG_TIME
    long long tm[5] = {0, 0, 0, 0, 0}, tq = clock64(); const long long tw0 = wall_clock64(), tc0 = tq;
#define SZG_TICK(k) { const long long t_ = clock64(); tm[k] += t_ - tq; tq = t_; }
#else
#define SZG_TICK(k)
#endif
    while (have) {
        const int i0 = ni0, j0 = nj0, k0 = nk0;
        __syncthreads();                                                 // the previous tile's image (and its raw flag) has been read
        SZG_TICK(0)
        if (threadIdx.x == 0) {
            rawflag[parity ^ 1] = 0;                                     // the next tile's flag
            if (r == SZG_CHUNK - 1) pend = atomicAdd(ticket, 1u);        // the next chunk: asked for now, published after this wavefront's
        }                                                                // share of the commit (a few microseconds: no stall on the reply)
        szg_tile_commit<T>(g, data, i0, j0, k0, eb, R, qs, &rawflag[parity], false);
        if (threadIdx.x == 0 && r == SZG_CHUNK - 1) s_chunk = pend;
        __syncthreads();
        SZG_TICK(1)
        if (++r == SZG_CHUNK) { r = 0; chunk = s_chunk; }
        have = tile_at(chunk, r);
        if (have) szg_tile_load<T>(g, data, ni0, nj0, nk0, R);
        SZG_TICK(2)
        const bool lean = i0 + SZG_TI <= g.r0 && j0 + SZF_TJ <= g.r1 && k0 + SZF_TK <= g.r2 && rawflag[parity] == 0u && radius > SZG_PW;
        parity ^= 1;
        u64 cnt = 0;                                                     // bytes: symbols plo .. plo + 7 of this tile
#ifdef SZG_DBG_NOWALK
        if (i0 < 0)
#else
        if (lean)
#endif
        {
            uint32_t up[9], uc[9];
            szg_rowdiff9(qs, ih0, j, kg, up);
#pragma unroll
            for (int s = 0; s < SZG_TI / 2; ++s) {
                szg_rowdiff9(qs, ih0 + s + 1, j, kg, uc);
                uint32_t t[9]; int32_t d[8];
#pragma unroll
                for (int e = 0; e < 9; ++e) t[e] = uc[e] - up[e];
                unsigned other = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    d[e] = (int32_t)(t[e + 1] - t[e]);
                    const unsigned pw = (unsigned)(d[e] + SZG_PW / 2);
                    const bool in = pw < (unsigned)SZG_PW;
#ifndef SZG_DBG_NOCNT
                    cnt += in ? 1ull << (pw * 8u) : 0ull;
#endif
                    other |= in ? 0u : 1u << e;
                }
#ifdef SZG_DBG_NORARE
                other = 0;
#endif
                if (other) {                                            // (divergent: a few lanes, one or two turns)
                    u64 z = 0;
                    const unsigned p0 = szg_pack2(d[0], d[1]), p1 = szg_pack2(d[2], d[3]), p2 = szg_pack2(d[4], d[5]), p3 = szg_pack2(d[6], d[7]);
                    do {
                        const unsigned e = (unsigned)__builtin_ctz(other);
                        other &= other - 1u;
                        const int32_t dd = szg_pick8(p0, p1, p2, p3, e);
                        count_other((dd >= 2 - radius && dd < radius) ? (unsigned)(dd + radius) : 0u, z);
                    } while (other);
                    if (z) { atomicAdd((unsigned long long *)&ucnt[((int64_t)(i0 + ih0 + s) * g.r1 + (j0 + j)) * g.n2 + k0 / SZF_TK], (unsigned long long)z); zall += z; }
                }
#pragma unroll
                for (int e = 0; e < 9; ++e) up[e] = uc[e];
            }
        }
#ifdef SZG_DBG_NOWALK
        else if (i0 < -1)
#else
        else
#endif
        {
            const int gj = j0 + j, gk = k0 + kg * 8;
            szg_walk(qs, radius, ih0, j, kg, [&](int i, const unsigned (&c)[8], const int32_t (&)[8]) {
                const bool row_in = i0 + i < g.r0 && gj < g.r1;
                u64 z = 0;
                for (int e = 0; e < 8; ++e) {
                    if (!(row_in && gk + e < g.r2)) continue;
                    const unsigned pw = c[e] - (unsigned)plo;
                    if (pw < (unsigned)SZG_PW && c[e] >= 2u) cnt += 1ull << (pw * 8u); else count_other(c[e], z);
                }
                if (z) { atomicAdd((unsigned long long *)&ucnt[((int64_t)(i0 + i) * g.r1 + gj) * g.n2 + k0 / SZF_TK], (unsigned long long)z); zall += z; }
            });
        }
        acc_e += cnt & 0x00ff00ff00ff00ffull; acc_o += (cnt >> 8) & 0x00ff00ff00ff00ffull;
        if (++since == SZG_FLUSH) hand_over();
        SZG_TICK(3)
#ifdef SZG_DBG_RES
        ++res_n;
#endif
    }
    hand_over();
#ifdef SZG_DBG_RES
    if (threadIdx.x == 0 && blockIdx.x < 2048) { hist[65536 + 2048 + blockIdx.x] = res_t0; hist[65536 + 4096 + blockIdx.x] = (unsigned)wall_clock64(); hist[65536 + 6144 + blockIdx.x] = res_n | (__smid() << 16); }
#endif
#ifdef SZG_DBG_TIME
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 4; ++k) atomicAdd(&hist[65536 + 1024 + 4 * (threadIdx.x >> 6) + k], (unsigned)(tm[k] >> 6));
    if (threadIdx.x == 0 && blockIdx.x == 0) { hist[65536 + 1100] = (unsigned)(wall_clock64() - tw0); hist[65536 + 1101] = (unsigned)(clock64() - tc0); }
    if (threadIdx.x == 0 && blockIdx.x < 2048) { hist[65536 + 2048 + blockIdx.x] = (unsigned)tw0; hist[65536 + 4096 + blockIdx.x] = (unsigned)wall_clock64(); }
#endif
    zall = wave_sum_u64(zall);
    if ((threadIdx.x & 63) == 0) {
        if (zall & 0xffffffffull) atomicAdd(&hist[0], (unsigned)(zall & 0xffffffffull));
        if (zall >> 32) atomicAdd(&hist[1], (unsigned)(zall >> 32));
    }
    __syncthreads();
    for (int b = threadIdx.x; b < SZG_HBINS; b += 256) {
        const unsigned c = lh[b];
        if (c && lo + b >= 2 && (unsigned)(lo + b) < nsym) atomicAdd(&hist[lo + b], c);
    }
}

// ---- pass B.  uoffc: exclusive scan of ucnt (low / high half: ranks in the two side lists); ubits[unit] = bits of the
// unit; slots[unit][SZG_SLOT_WORDS] = its bits, MSB first, in 32-bit words (not byte-swapped).
// Same skeleton as pass A (persistent, ticketed chunks, the next tile's loads in flight, lean walk for interior tiles without raw
// points).  The code table's window of SZG_PTAB symbols around the radius sits in LDS (codes are at most 32 bits here: the host takes
// the code-array form otherwise); symbols outside it -- and the table itself when `intervals` is large -- are read from global memory.
#define SZG_PTAB 512
#ifndef SZG_PACK_OCC
#define SZG_PACK_OCC 3
#endif
template <class T>
__global__ __launch_bounds__(256, SZG_PACK_OCC) void k_fast_pack(szg_geom g, const T *__restrict__ data, T eb, int radius, unsigned nsym,
                                                      const u64 *__restrict__ code, const uint8_t *__restrict__ len,
                                                      const u64 *__restrict__ uoffc,
                                                      int32_t *listA, int32_t *listBd, T *listB, u64 *ubits, unsigned *slots, unsigned *ticket)
{
    __shared__ __attribute__((aligned(16))) int32_t qs[SZG_TILE_WORDS];
    __shared__ unsigned lcode[SZG_PTAB];
    __shared__ uint8_t llen[SZG_PTAB];
    __shared__ unsigned stage[32 * SZG_SLOT_WORDS];      // 2 plane groups x 16 rows, one unit each per step
    __shared__ unsigned rawflag[2];
    __shared__ unsigned s_chunk;
    const int wlo = radius - SZG_PTAB / 2;
    for (int w = threadIdx.x; w < SZG_PTAB; w += 256) {
        const int c = wlo + w;
        const bool ok = c >= 0 && (unsigned)c < nsym;
        lcode[w] = ok ? (unsigned)code[c] : 0u; llen[w] = ok ? len[c] : (uint8_t)0;
    }
    if (threadIdx.x < 2) rawflag[threadIdx.x] = 0;
    const int kg = threadIdx.x & 7, j = (threadIdx.x >> 3) & 15, grp = threadIdx.x >> 7, ih0 = grp * (SZG_TI / 2);
    unsigned *const st = stage + (grp * 16 + j) * SZG_SLOT_WORDS;
    auto sym = [&](unsigned c, unsigned &l, unsigned &cw) {
        const unsigned w = c - (unsigned)wlo;
        if (w < (unsigned)SZG_PTAB) { l = llen[w]; cw = lcode[w]; } else { l = len[c]; cw = (unsigned)code[c]; }
    };
    szg_regs<T> R;
    const unsigned nchunk = (unsigned)((g.ntiles + SZG_CHUNK - 1) / SZG_CHUNK);
    if (threadIdx.x == 0) s_chunk = atomicAdd(ticket, 1u);
    __syncthreads();
    unsigned chunk = s_chunk, r = 0, pend = 0;
    int ni0 = 0, nj0 = 0, nk0 = 0, parity = 0;
    auto tile_at = [&](unsigned c, unsigned rr) -> bool {
        const int64_t tile = (int64_t)c * SZG_CHUNK + rr;
        if (c >= nchunk || tile >= g.ntiles) return false;
        const unsigned t = (unsigned)tile, tk = t % (unsigned)g.n2, tq = t / (unsigned)g.n2;
        nk0 = (int)tk * SZF_TK; nj0 = (int)(tq % (unsigned)g.n1) * SZF_TJ; ni0 = (int)(tq / (unsigned)g.n1) * SZG_TI;
        return true;
    };
    bool have = tile_at(chunk, 0);
    if (have) szg_tile_load<T>(g, data, ni0, nj0, nk0, R);
    __syncthreads();
    while (have) {
        const int i0 = ni0, j0 = nj0, k0 = nk0;
        __syncthreads();                                                 // the previous tile's image (and its raw flag) has been read
        if (threadIdx.x == 0) {
            rawflag[parity ^ 1] = 0;
            if (r == SZG_CHUNK - 1) pend = atomicAdd(ticket, 1u);
        }
        szg_tile_commit<T>(g, data, i0, j0, k0, eb, R, qs, &rawflag[parity], false);
        if (threadIdx.x == 0 && r == SZG_CHUNK - 1) s_chunk = pend;
        __syncthreads();
        if (++r == SZG_CHUNK) { r = 0; chunk = s_chunk; }
        have = tile_at(chunk, r);
        if (have) szg_tile_load<T>(g, data, ni0, nj0, nk0, R);
        const bool lean = i0 + SZG_TI <= g.r0 && j0 + SZF_TJ <= g.r1 && k0 + SZF_TK <= g.r2 && rawflag[parity] == 0u;
        parity ^= 1;
        const int gj = j0 + j, gk = k0 + kg * 8;
        // one plane of the thread's row segment: the 8 codes c (0 = unpredictable, 1 = raw), their deltas dl; `full`: every point inside
        auto emit = [&](int i, const unsigned (&c)[8], const int32_t (&dl)[8], bool full) {
            const bool row_in = full || (i0 + i < g.r0 && gj < g.r1);
            const int64_t unit = ((int64_t)(i0 + i) * g.r1 + gj) * g.n2 + k0 / SZF_TK;
            // bits of this thread's 8 codes (concatenated in a register), and where they start inside the unit
            unsigned s = 0, z0 = 0, z1 = 0, pos_in = 0;
            u64 acc = 0; int accn = 0;
            unsigned ovl = 0;                                            // codes that did not fit the 64-bit register (rare: > 8 bits on average)
            for (int e = 0; e < 8; ++e) {
                const bool in = full || (row_in && gk + e < g.r2);
                if (!in) continue;
                unsigned l, cw; sym(c[e], l, cw);
                z0 += c[e] == 0; z1 += c[e] == 1;
                if (accn + (int)l > 64) { ovl |= 1u << e; continue; }
                acc = (acc << l) | (u64)cw;                              // (l <= 32: the shift is defined; cw has no bits above l)
                accn += (int)l; s += l;
            }
            unsigned sfull = s;
            if (ovl) for (int e = 0; e < 8; ++e) if (ovl >> e & 1) { unsigned l, cw; sym(c[e], l, cw); sfull += l; }
            unsigned incl = sfull;
            { const unsigned t = SZG_SHR(incl, 1); if (kg >= 1) incl += t; }
            { const unsigned t = SZG_SHR(incl, 2); if (kg >= 2) incl += t; }
            { const unsigned t = SZG_SHR(incl, 4); if (kg >= 4) incl += t; }
            const unsigned total = __shfl(incl, (threadIdx.x & 63) | 7, 64);
            const unsigned pos = incl - sfull;
            (void)pos_in;
            // clear the words the unit fills, then OR the codes in
            for (unsigned w = kg; w < (total + 31) / 32; w += 8) st[w] = 0;
            SZG_WAVE_SYNC();
            if (!ovl) { if (accn) lds_put_bits(st, pos, acc, accn); }
            else {                                                       // the general form: code by code
                unsigned p = pos;
                for (int e = 0; e < 8; ++e) {
                    if (!(full || (row_in && gk + e < g.r2))) continue;
                    unsigned l, cw; sym(c[e], l, cw);
                    if (l) { lds_put_bits(st, p, (u64)cw, (int)l); p += l; }
                }
            }
            SZG_WAVE_SYNC();
            if (row_in) {
                for (unsigned w = kg; w < (total + 31) / 32; w += 8) slots[unit * SZG_SLOT_WORDS + w] = st[w];
                if (kg == 7) ubits[unit] = (u64)total;
            }
            // side lists, in stream order: rank = the unit's offset + the entries before this one inside the unit (rare: wavefronts without
            // an entry skip the prefix sums)
            unsigned zi = z0 | (z1 << 16);
            if (!__all(zi == 0u)) {
                const unsigned zown = zi;
                { const unsigned t = SZG_SHR(zi, 1); if (kg >= 1) zi += t; }
                { const unsigned t = SZG_SHR(zi, 2); if (kg >= 2) zi += t; }
                { const unsigned t = SZG_SHR(zi, 4); if (kg >= 4) zi += t; }
                if (row_in && zown) {
                    const u64 uo = uoffc[unit];
                    u64 ra = (uo & 0xffffffffull) + ((zi & 0xffffu) - z0), rb = (uo >> 32) + ((zi >> 16) - z1);
                    for (int e = 0; e < 8; ++e) {
                        if (gk + e >= g.r2) break;
                        if (c[e] == 0) listA[ra++] = dl[e];
                        else if (c[e] == 1) { listBd[rb] = dl[e]; listB[rb] = data[((int64_t)(i0 + i) * g.r1 + gj) * g.r2 + gk + e]; ++rb; }
                    }
                }
            }
            SZG_WAVE_SYNC();
        };
        if (lean) {
            uint32_t up[9], uc[9];
            szg_rowdiff9(qs, ih0, j, kg, up);
#pragma unroll
            for (int sI = 0; sI < SZG_TI / 2; ++sI) {
                szg_rowdiff9(qs, ih0 + sI + 1, j, kg, uc);
                uint32_t t[9]; int32_t d[8]; unsigned c[8];
#pragma unroll
                for (int e = 0; e < 9; ++e) t[e] = uc[e] - up[e];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    d[e] = (int32_t)(t[e + 1] - t[e]);
                    c[e] = (d[e] >= 2 - radius && d[e] < radius) ? (unsigned)(d[e] + radius) : 0u;
                }
                emit(ih0 + sI, c, d, true);
#pragma unroll
                for (int e = 0; e < 9; ++e) up[e] = uc[e];
            }
        } else {
            szg_walk(qs, radius, ih0, j, kg, [&](int i, const unsigned (&c)[8], const int32_t (&dl)[8]) { emit(i, c, dl, false); });
        }
    }
}

// ---- pass C: unit u's `ubits[u]` bits go to bit position bit0 + uoff[u] of the stream (MSB first within bytes, Huffman.c:205-308).
// A workgroup takes 256 consecutive units -- one contiguous stretch of the stream -- assembles it in LDS and writes whole words;
// only the two boundary words are shared with the neighbours (atomicOr; the stream is zero beforehand).  A stretch longer than the
// buffer (dense codes) goes word by word with atomicOr instead.
#define SZG_CBUF 4096                     /* words: 256 units x 16 words (512 bits per unit on average) */
__global__ __launch_bounds__(256) void k_fast_compact(int64_t nunits, const u64 *__restrict__ ubits, const u64 *__restrict__ uoff,
                                                      const unsigned *__restrict__ slots, u64 bit0, unsigned *out32)
{
    __shared__ unsigned buf[SZG_CBUF + 2];
    __shared__ u64 span[2];
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t ulast = min((int64_t)blockIdx.x * 256 + 255, nunits - 1);
    if (threadIdx.x == 0) { span[0] = bit0 + uoff[(int64_t)blockIdx.x * 256]; span[1] = bit0 + uoff[ulast] + ubits[ulast]; }
    __syncthreads();
    const u64 gbit = span[0], gend = span[1];
    const unsigned lead = (unsigned)(gbit & 31);
    const u64 w0 = gbit >> 5;
    const unsigned nwords = (unsigned)((lead + (gend - gbit) + 31) >> 5);
    const bool in_lds = nwords <= SZG_CBUF;
    if (in_lds) { for (unsigned w = threadIdx.x; w < nwords; w += 256) buf[w] = 0; }
    __syncthreads();
    if (u < nunits) {
        const unsigned nb = (unsigned)ubits[u];
        const unsigned *src = slots + u * SZG_SLOT_WORDS;
        const u64 d = bit0 + uoff[u];
        if (in_lds) {
            unsigned pos = (unsigned)(d - gbit) + lead;
            for (unsigned k = 0; k * 32 < nb; ++k) {
                const unsigned take = nb - k * 32 < 32 ? nb - k * 32 : 32;
                lds_put_bits(buf, pos, (u64)(src[k] >> (32 - take)), (int)take);
                pos += take;
            }
        } else if (nb) {
            const unsigned sh = (unsigned)(d & 31);
            const u64 w = d >> 5;
            const unsigned nw = (nb + 31) / 32;
            unsigned carry = 0;
            for (unsigned k = 0; k < nw; ++k) {
                const unsigned v = src[k];
                const unsigned word = carry | (sh ? (v >> sh) : v);
                if (word) atomicOr(&out32[w + k], __builtin_bswap32(word));
                carry = sh ? (v << (32 - sh)) : 0u;
            }
            if (carry) atomicOr(&out32[w + nw], __builtin_bswap32(carry));
        }
    }
    if (in_lds) {
        __syncthreads();
        for (unsigned w = threadIdx.x; w < nwords; w += 256) {
            const unsigned v = __builtin_bswap32(buf[w]);
            if (w == 0 || w == nwords - 1) { if (v) atomicOr(&out32[w0 + w], v); }
            else out32[w0 + w] = v;
        }
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
