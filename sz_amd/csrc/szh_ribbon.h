// szh_ribbon.h -- the predict+quantise (and inverse) kernel of the SZ 2.1 3-D path, second mapping ("ribbon", round 3).
//
// Same arithmetic as szh_pencil.h (7-point Lorenzo from RECONSTRUCTED neighbours, sz/src/sz_float.c:7253-7353, with mean :6914-7030;
// inverse sz/src/szd_float.c:3483-5866), same bit-exact results, different shape.  What round 2 measured about the pencil kernel
// (8x8 lanes per wavefront, one row per lane): a step costs ~130 instructions for 64 points, three of them trips through the LDS
// crossbar (ds_bpermute, 69 cycles each in a dependent chain against 9 for a VALU op and 18 for a DPP move: tools/ubench), and the
// sweep crosses 64 + 64 pencil boundaries.  Here:
//
//   * a wavefront owns a RIBBON: 64 lanes side by side along dim 1 (j = 64 J + lane), R consecutive rows of dim 0 per lane,
//     swept along dim 2 (the contiguous dimension).  At step t lane l works on k = t - l - r in its row r, so all R * 64 points of a
//     step lie on one hyperplane i + j + k and are independent (R-way instruction-level parallelism for the in-order wavefront);
//   * the (j - 1) neighbours come from lane l - 1 by ONE DPP `wave_shr:1` per row and step (the lane-0 value is the `old` operand:
//     the left halo); the (i - 1) neighbours are the lane's own registers (row r - 1), except for row 0;
//   * W such wavefronts stacked in dim 0 form a TILE (workgroup): wavefront w hands the reconstruction of its last row to
//     wavefront w + 1 through an LDS ring, one ds_write / ds_read per step, guarded by an LDS step counter;
//   * tiles hand their last row (downwards) and their last column (to the right) on through HBM-side granules {launch epoch, value bits}
//     -- naturally aligned 8-byte words written by one agent-scope store, the data being its own flag (MI355X_MICROARCH "handoff-1to1") --
//     laid out BY STEP: all 64 values of a step are one contiguous 512-byte row, so a hand-off is one coalesced store / load per step.
//     Three helper wavefronts per tile move them: DRAIN forwards the LDS ring of the tile's last wavefront and lane 63's values of every
//     wavefront to the granule rows (stores only, a nap between rounds), FILL_U / FILL_L poll the granules of the tiles above / on the
//     left and drop them into LDS rings (loads only).  There are no progress words: a granule's tag says whether it is there.
//     Why the compute wavefronts store no granules themselves (measured, tools/gpu_rb_trace.py with SZ_HIP_DBG): gfx9 counts loads and
//     stores in ONE counter and hipcc treats mixed traffic as out of order, so the one wait of a trip -- for input rows requested a
//     whole trip earlier -- is `s_waitcnt vmcnt(0)` and also waits for the acknowledgement of every store issued a moment ago (~3 us
//     per 16-step trip on an idle chip: 0.43 against 0.23 us per step).  For the same reason the codes of a trip are stored at the top
//     of the NEXT trip: a whole trip old when the next wait comes;
//   * per-wavefront step numbers are SHIFTED by w (R - 1) so that the ring position a wavefront reads (its producer's step) and the one
//     it writes are the same index: every ring access of an unrolled trip is `base + immediate`.
//
// Longest dependency path of a launch (512^3 float, R = 2, W = 8): r2 + 8 * 64 lane steps along dim 1 + 256 ring hand-offs + 32 + 8
// tile hand-offs, against 1536 steps + 128 pencil hops before; a step is ~60 instructions for 128 points.
//
// Codes: compression leaves them in RIBBON ORDER (below: one coalesced 1 KB store per instruction; k_permute<0> gathers the block
// order from it); decompression reads natural-order codes and writes natural-order values.
//
// Covers: 3-D arrays, float / double, compress / decompress, with / without the mean shortcut, Lorenzo-only block maps
// (`no_reg`; the host sends everything else to k_pencil).
#pragma once
#include "szh_pencil.h"

#ifndef SZH_RB_R_F32
#define SZH_RB_R_F32 2
#define SZH_RB_W_F32 8
#endif
#ifndef SZH_RB_R_F64
#define SZH_RB_R_F64 2
#define SZH_RB_W_F64 8
#endif
template <class T> struct szh_rb_shape;
// R rows per lane, W compute wavefronts per tile; ring lengths (steps): RL between wavefronts of a tile, RLU the up ring of wavefront 0
// (filled from granules: bursty), RLL the left rings (the left tile runs >= 64 steps ahead), RLR the right rings; U steps per trip (a
// trip's inputs sit in registers: 2 x R x U values).  RL >= 2 U: a wavefront checks the ring space of a whole trip at its top
#ifndef SZH_RB_RL_F32
#define SZH_RB_RL_F32 32
#endif
template <> struct szh_rb_shape<float> { static constexpr int R = SZH_RB_R_F32, W = SZH_RB_W_F32, U = 16, RL = SZH_RB_RL_F32, RLU = 64, RLL = 128, RLR = 32; };
template <> struct szh_rb_shape<double> { static constexpr int R = SZH_RB_R_F64, W = SZH_RB_W_F64, U = 8, RL = 16, RLU = 32, RLL = 128, RLR = 32; };
// inverse: rows of a trip per hand-over to the STORE wavefront (LDS: W x HB x 64 x U values; double has room for one row only)
template <class T> struct szh_rb_hb { static constexpr int HB = sizeof(T) == 4 ? szh_rb_shape<T>::R : 1; };
#define SZH_RB_INF (1 << 30)
#ifndef SZH_RB_KD
#define SZH_RB_KD 4            /* DRAIN: steps per face and round */
#endif
#ifndef SZH_RB_DRAIN_NAP
#define SZH_RB_DRAIN_NAP 2     /* DRAIN: s_sleep units between rounds */
#endif

// number of (shifted) steps every wavefront of a launch runs, a multiple of the trip length
SZH_HD int szh_rb_steps(int r2, int R, int W, int U) { return (r2 + 62 + R + (W - 1) * (R - 1) + U - 1) / U * U; }
template <class T> SZH_HD int szh_rb_steps_of(int r2) { return szh_rb_steps(r2, szh_rb_shape<T>::R, szh_rb_shape<T>::W, szh_rb_shape<T>::U); }

// RIBBON ORDER of the code array (compress): the codes leave the kernel in the order the wavefronts make them, so that every store
// instruction writes 64 x 16 contiguous, aligned bytes (in natural order a wavefront's 64 lanes write 16-byte pieces of 64 different
// rows, 2-byte aligned: measured at 512^3, those stores cost more than the whole sweep -- 2.16 ms against 0.95 ms without them).
//   [tile][trip][wavefront w][row r][group v of 8 steps][lane][8 codes]      (step s = 8 v + e of the trip, k = trip * U + s - w (R-1) - lane - r)
// Positions outside the array hold whatever was computed there; readers go by geometry.  k_permute<0> gathers from it (rb.on).
struct szh_rb_layout { int on, nTJ, NT, W, R, U; };
SZH_HD int64_t szh_rb_tile_elems(const szh_rb_layout &y) { return (int64_t)y.NT * y.W * y.R * 64; }
// index of the 8-code group that holds shifted step tt (a multiple of 8) of (tile, wavefront, row), for lane 0
SZH_HD int64_t szh_rb_group_index(const szh_rb_layout &y, int64_t tile, int w, int r, int tt)
{
    const int trip = tt / y.U, v = (tt % y.U) / 8;
    return tile * szh_rb_tile_elems(y) + ((((int64_t)trip * y.W + w) * y.R + r) * (y.U / 8) + v) * 512;
}

// RIBBON ORDER of a VALUE array (inverse, mode 2): as for the codes, but in groups of g = 16 / sizeof(T) values (one 16-byte vector per lane):
//   [tile][trip][wavefront w][row r][group v of g steps][lane][g values]
// index of point (i, j, k) of the array
SZH_HD int64_t szh_rb_value_index(const szh_rb_layout &y, int g, int i, int j, int k)
{
    const int WR = y.W * y.R, TI = i / WR, q = i - TI * WR, w = q / y.R, r = q - w * y.R, TJ = j >> 6, ln = j & 63;
    const int tt = k + w * (y.R - 1) + ln + r, trip = tt / y.U, s = tt - trip * y.U, v = s / g, e = s - v * g;
    return ((int64_t)TI * y.nTJ + TJ) * szh_rb_tile_elems(y) + (((((int64_t)trip * y.W + w) * y.R + r) * (y.U / g) + v) * 64 + ln) * g + e;
}

#if defined(__HIPCC__) || defined(SZH_HIPSIM)
namespace szh_rb {
#ifdef SZH_HIPSIM
struct v4u { unsigned x, y, z, w; };
#else
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#endif

#ifdef SZH_HIPSIM
#define SZH_LDS
struct rsrc_t { char *base; unsigned n; };
static inline rsrc_t make_rsrc(const void *p, unsigned bytes) { rsrc_t r = {(char *)const_cast<void *>(p), bytes}; return r; }
static inline v4u bload16(rsrc_t rs, unsigned off)
{
    unsigned w[4];
    for (int e = 0; e < 4; ++e) { w[e] = 0; if ((uint64_t)off + 4u * e + 4u <= rs.n) memcpy(&w[e], rs.base + off + 4u * e, 4); }
    v4u v = {w[0], w[1], w[2], w[3]}; return v;
}
static inline void bstore16(rsrc_t rs, unsigned off, v4u v)
{
    unsigned w[4] = {v.x, v.y, v.z, v.w};
    for (int e = 0; e < 4; ++e) if ((uint64_t)off + 4u * e + 4u <= rs.n) memcpy(rs.base + off + 4u * e, &w[e], 4);
}
static inline void bstore2(rsrc_t rs, unsigned off, unsigned short v) { if ((uint64_t)off + 2u <= rs.n) memcpy(rs.base + off, &v, 2); }
template <class T> static inline void bstoreT(rsrc_t rs, unsigned off, T v) { if ((uint64_t)off + sizeof(T) <= rs.n) memcpy(rs.base + off, &v, sizeof(T)); }
static inline void bstore8_wt(rsrc_t rs, unsigned off, unsigned soff, szh_u64 g) { if ((uint64_t)off + soff + 8u <= rs.n) __atomic_store_n((szh_u64 *)(rs.base + off + soff), g, __ATOMIC_RELAXED); }
static inline void bstore16_wt(rsrc_t rs, unsigned off, unsigned soff, szh_u64 g0, szh_u64 g1) { bstore8_wt(rs, off, soff, g0); bstore8_wt(rs, off + 8u, soff, g1); }
template <class E> static inline E lds_ld(const E *p)
{
    E v;
    if (sizeof(E) == 8) { const uint64_t u = __atomic_load_n((const uint64_t *)p, __ATOMIC_RELAXED); memcpy(&v, &u, sizeof(E)); }
    else { const uint32_t u = __atomic_load_n((const uint32_t *)p, __ATOMIC_RELAXED); memcpy(&v, &u, sizeof(E)); }
    return v;
}
template <class E> static inline void lds_st(E *p, E v)
{
    if (sizeof(E) == 8) { uint64_t u; memcpy(&u, &v, sizeof(E)); __atomic_store_n((uint64_t *)p, u, __ATOMIC_RELAXED); }
    else { uint32_t u; memcpy(&u, &v, sizeof(E)); __atomic_store_n((uint32_t *)p, u, __ATOMIC_RELAXED); }
}
static inline int uni(int v) { return __shfl(v, 0, 64); }
static inline void lds_fence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); (void)__all(1); __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void lds_order() { lds_fence(); }
template <class T> static inline T shr1(T old, T v) { const T s = __shfl_up(v, 1, 64); return (threadIdx.x & 63) == 0 ? old : s; }
static inline void prio(int) {}
static inline void keep(float &) {}
static inline void keep(double &) {}
static inline void keepu(unsigned &) {}
#else
#define SZH_LDS __attribute__((address_space(3)))   /* LDS pointers stay 32-bit offsets: no generic-pointer arithmetic in the sweep */
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000); }
// raw buffer accesses: out-of-range dwords read 0 / are dropped (tools/ubench/ub_mem.hip), any byte alignment
__device__ __forceinline__ v4u bload16(rsrc_t rs, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0); }
__device__ __forceinline__ void bstore16(rsrc_t rs, unsigned off, v4u v) { __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, 0); }
__device__ __forceinline__ void bstore2(rsrc_t rs, unsigned off, unsigned short v) { __builtin_amdgcn_raw_buffer_store_b16((short)v, rs, (int)off, 0, 0); }
__device__ __forceinline__ void bstoreT(rsrc_t rs, unsigned off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)off, 0, 0); }
__device__ __forceinline__ void bstoreT(rsrc_t rs, unsigned off, double v)
{
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    v2u w = {(unsigned)u, (unsigned)(u >> 32)};
    __builtin_amdgcn_raw_buffer_store_b64(w, rs, (int)off, 0, 0);
}
// granule stores: written through, not kept in this XCD's L2 (aux 17 = sc0 sc1, the agent-scope form); NOT volatile / atomic -- hipcc
// drains the memory queue behind those
__device__ __forceinline__ void bstore8_wt(rsrc_t rs, unsigned off, unsigned soff, szh_u64 g)
{
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    v2u w = {(unsigned)g, (unsigned)(g >> 32)};
    __builtin_amdgcn_raw_buffer_store_b64(w, rs, (int)off, (int)soff, 17);
}
__device__ __forceinline__ void bstore16_wt(rsrc_t rs, unsigned off, unsigned soff, szh_u64 g0, szh_u64 g1)
{
    v4u w = {(unsigned)g0, (unsigned)(g0 >> 32), (unsigned)g1, (unsigned)(g1 >> 32)};
    __builtin_amdgcn_raw_buffer_store_b128(w, rs, (int)off, (int)soff, 17);
    // Seen on gfx950 (round 3): with a REGISTER soffset hipcc assumes the ">64-bit store data" hazard does not exist and lets the very
    // next VALU instruction overwrite the data registers; now and then the store then wrote the NEW contents (an LDS address in the tag
    // word of a granule).  Keep the four registers alive across a few wait states.
    asm volatile("s_nop 3" :: "v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w));
}
template <class E> __device__ __forceinline__ E lds_ld(const SZH_LDS E *p) { return *(const volatile SZH_LDS E *)p; }
template <class E> __device__ __forceinline__ void lds_st(SZH_LDS E *p, E v) { *(volatile SZH_LDS E *)p = v; }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }     // one wavefront's LDS accesses execute in program order
// lane l receives v of lane l - 1; lane 0 keeps `old` (DPP wave_shr:1, bound_ctrl 0: tools/ubench/ub_valu.hip)
__device__ __forceinline__ float shr1(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ double shr1(double old, double v)
{
    const long long o = __double_as_longlong(old), s = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)o, (int)s, 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(s >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ void prio(int p) { if (p >= 3) __builtin_amdgcn_s_setprio(3); else if (p == 2) __builtin_amdgcn_s_setprio(2); else if (p == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
__device__ __forceinline__ void keep(float &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void keep(double &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void keepu(unsigned &v) { asm volatile("" : "+v"(v)); }
#endif
__device__ __forceinline__ szh_u64 ld_gran(const szh_u64 *p) { return __hip_atomic_load(const_cast<szh_u64 *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_gran(szh_u64 *p, szh_u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_flag(const unsigned *p) { return __hip_atomic_load(const_cast<unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_done(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void nap(int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1); }

// development build (SZH_DEV): per-wavefront stamps, [tile][W + 4 wavefronts][8] u64 in a.trace:
//   compute: {start, end of first trip, end, cycles waiting for the wavefront above / FILL_U, for FILL_L, for ring space, XCC, steps}
//   helpers: {start, -, end, rounds, rounds that moved nothing, -, XCC, steps}
#if SZH_DEV && !defined(SZH_HIPSIM)
__device__ __forceinline__ szh_u64 rb_wall() { return wall_clock64(); }
__device__ __forceinline__ szh_u64 rb_cyc() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ szh_u64 rb_xcc() { unsigned x = 0; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x; }
#else
__device__ __forceinline__ szh_u64 rb_wall() { return 0; }
__device__ __forceinline__ szh_u64 rb_cyc() { return 0; }
__device__ __forceinline__ szh_u64 rb_xcc() { return 0; }
#endif
template <class T> __device__ __forceinline__ szh_u64 *rb_trace_slot(const szh_qargs<T> &a, int TI, int TJ, int w)
{
    return (SZH_DEV && a.trace) ? a.trace + (((int64_t)TI * a.nJ + TJ) * (szh_rb_shape<T>::W + 4) + w) * 8 : nullptr;
}

// development: timeline of the tiles of column 0: [TI][4 agents: wavefront 0, last wavefront, DRAIN, FILL_U][48] wall clock when the agent
// passed step 16 i
template <class T> __device__ __forceinline__ szh_u64 *rb_timeline(const szh_qargs<T> &a, int TI, int TJ, int agent)
{
    return (SZH_DEV && a.trace && TJ == 0) ? a.trace + (int64_t)a.nI * a.nJ * (szh_rb_shape<T>::W + 4) * 8 + ((int64_t)TI * 4 + agent) * 48 : nullptr;
}

// LDS of one tile
template <class T> struct lds_t {
    SZH_LDS T *ring;    // ring 0: [RLU][64] (up ring of wavefront 0, filled by FILL_U); ring g = 1..W: [RL][64] written by wavefront g - 1
                        // (ring W: read by DRAIN)
    SZH_LDS T *lr;      // left rings [RLL][LS]: element 0 = the row above the tile (from the tile up-left), 1 + q = row q of the tile
    SZH_LDS T *rr;      // right rings [RLR][W * R]: lane 63's values, read by DRAIN
    SZH_LDS unsigned *P; // [W + 2][2]: {cnt[g], left[g]}: cnt[0] = indices of ring 0 filled, cnt[g] = steps completed by wavefront g - 1,
                        //              cnt[W + 1] = steps DRAIN has forwarded (of ring W and of every right ring); left[w] = indices of
                        //              the left rings complete for wavefront w
    SZH_LDS int *scratch; // [64]: FILL_L (per-row values for a per-wavefront minimum)
    SZH_LDS T *tbuf;    // inverse only: [W][HB][64][U]: a trip's results on their way to the STORE wavefront (store_trip, store_out)
    SZH_LDS unsigned *Q; // inverse only: [W][2]: {trips wavefront w has put into tbuf, trips STORE has taken out}
};

// bounded wait until an LDS word reaches `need` (a lost hand-off must end the launch, not hang the GPU)
__device__ __forceinline__ int wait_word(const SZH_LDS unsigned *p, int need, unsigned *err)
{
    int v = uni((int)lds_ld(p));
    unsigned spins = 0;
#pragma unroll 1
    while (v < need) {
        if (++spins > (1u << 22)) { st_flag(err, 1u); break; }
        if ((spins & 1023u) == 0 && uni((int)ld_flag(err)) != 0) break;
        nap(1);
        v = uni((int)lds_ld(p));
    }
    return v;
}

// The quantiser of szh_quant_sel (sz_float.c:7270-7287) with a shorter dependency chain, bit for bit the same results:
//   itv / 2 = (|diff| * recip + 1) / 2 = RN(|diff| * (recip / 2)) + 1/2        (scaling by 2 commutes with rounding)
//   (T)(2 * (int)(itv / 2)) * eb = trunc(itv / 2) * (2 eb)                      (2 t and 2 eb are exact)
//   the sign of diff goes onto the product; "+ 0" turns the -0 of a zero product back into the reference's +0
// rh = recip / 2, eb2 = 2 eb, caph = capacity / 2, radf = (T)radius.  Returns the code.
template <class T>
__device__ __forceinline__ int rb_quant(T x, T pred, T eb, T eb2, T rh, T caph, T radf, T *recon)
{
    const T diff = x - pred;
    const T h = szh_abs(diff) * rh + (T)0.5;
    const bool inr = h < caph;
    const T t = __builtin_trunc(h);
    const T ts = __builtin_copysign(t, diff);
    const T rc = pred + (ts * eb2 + (T)0);
    const bool ok = inr && !(szh_abs(x - rc) > eb);
    *recon = ok ? rc : x;
    // radius + q, through the float (exact: |q| < capacity <= 65536); the conversion of an out-of-range h is never used
    return ok ? (int)(radf + (inr ? ts : (T)0)) : 0;
}
template <> __device__ __forceinline__ int rb_quant<float>(float x, float pred, float eb, float eb2, float rh, float caph, float radf, float *recon)
{
    const float diff = x - pred;
    const float h = __builtin_fabsf(diff) * rh + 0.5f;
    const bool inr = h < caph;
    const float t = __builtin_truncf(h);
    const float ts = __builtin_copysignf(t, diff);
    const float rc = pred + (ts * eb2 + 0.0f);
    const bool ok = inr && !(__builtin_fabsf(x - rc) > eb);
    *recon = ok ? rc : x;
    return ok ? (int)(radf + (inr ? ts : 0.0f)) : 0;
}

// ------------------------------------------------------------------------------------------------------------------ compute
template <class T, bool DEC, bool USEMEAN>
__device__ __forceinline__ void ribbon_body(const szh_qargs<T> &a, const int TI, const int TJ, const int w, const lds_t<T> &L)
{
    using S = szh_rb_shape<T>;
    constexpr int R = S::R, W = S::W, U = S::U, LS = W * R + 2, WR = W * R;
    static_assert(S::RL >= 2 * U && U % 8 == 0, "ring space is checked per trip; codes travel in groups of 8");
    constexpr int VPT = 16 / (int)sizeof(T), NVEC = U / VPT, NW = szh_gran<T>::NW;
    const szh_geom3 &G = a.G;
    const int r0 = G.g0.count, r1 = G.g1.count, r2 = G.g2.count;
    const int lane = (int)(threadIdx.x & 63);
    const int sh = w * (R - 1);
    const int NT = szh_rb_steps_of<T>(r2);
    const T eb = a.eb, recip = a.recip, mean = a.mean;
    const int cap_lor = a.cap - 2, radius = a.radius;
    (void)mean;

    // rows of this lane; buffer offsets are relative to the first row of the tile
    const int64_t tile_base = (int64_t)TI * WR * G.d0;
    const int64_t rest = G.n - tile_base;
    // (the range check of a buffer access is per dword: the code array's last dword may hold one code -- the host allocates 64 bytes of
    //  slack behind it; offsets below the base wrap to >= 2^31: the record count stays below that)
    const uint64_t span_x = (uint64_t)rest * sizeof(T), span_c = (uint64_t)rest * 2 + 8;
    const bool xrb = DEC && a.codes_ribbon == 2;               // (uniform) inverse, mode 2: the value array is in ribbon order as well
    const rsrc_t rsx = xrb ? make_rsrc((T *)a.out + ((int64_t)TI * a.nJ + TJ) * ((int64_t)NT * WR * 64), (unsigned)((size_t)NT * WR * 64 * sizeof(T)))
                           : make_rsrc((DEC ? (const T *)a.out : a.data) + tile_base, span_x > 0x7ffffff0ull ? 0x7ffffff0u : (unsigned)span_x);
    // codes: natural order for the inverse; ribbon order (this tile's region) for compression
    const bool crb = !DEC || a.codes_ribbon;                   // (uniform)
    const rsrc_t rsc = !crb ? make_rsrc(a.codes + tile_base, span_c > 0x7ffffff0ull ? 0x7ffffff0u : (unsigned)span_c)
                           : make_rsrc(a.codes + ((int64_t)TI * a.nJ + TJ) * ((int64_t)NT * WR * 64), (unsigned)((size_t)NT * WR * 64 * 2));
    bool rowok[R], iok[R];
    int reli[R];             // the dim-0 part of rowrel (the same for every lane)
    int rowrel[R];           // element offset of (row, column 0 of dim 2) from the tile base
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = TI * WR + w * R + r, j = TJ * 64 + lane;
        rowok[r] = i < r0 && j < r1;
        const int ic = i < r0 ? i : r0 - 1, jc = j < r1 ? j : r1 - 1;
        rowrel[r] = (int)((int64_t)(ic - TI * WR) * G.d0 + (int64_t)jc * G.d1);
        reli[r] = (int)((int64_t)(ic - TI * WR) * G.d0); iok[r] = i < r0;
    }
    // LDS addresses
    const SZH_LDS T *const rin = L.ring + (w == 0 ? 0 : S::RLU * 64 + (w - 1) * S::RL * 64) + lane;     // ring w: read
    SZH_LDS T *const rout = L.ring + S::RLU * 64 + w * S::RL * 64 + lane;                                 // ring w + 1: write
    SZH_LDS T *const rwr = L.rr + w * R;
    const int rin_mask = w == 0 ? S::RLU - 1 : S::RL - 1;
    const SZH_LDS T *const lrd = L.lr + w * R;                                                            // {corner, b_0 .. b_{R-1}} of a step
    const SZH_LDS unsigned *const pin = L.P + 2 * w;                // {cnt[w], left[w]}
    SZH_LDS unsigned *const pout = L.P + 2 * (w + 1);               // cnt[w + 1] = my steps
    const SZH_LDS unsigned *const pnext = L.P + 2 * (w + 2);        // consumer of my ring (the last wavefront: DRAIN)
    const SZH_LDS unsigned *const pdrain = L.P + 2 * (W + 1);
    const bool has_right = TJ + 1 < a.nJ;
    // development (timing only, results become wrong): a.dbg & 1 no code / value stores, & 4 no input loads
    const bool dbg_nost = SZH_DEV && (a.dbg & 1), dbg_nold = SZH_DEV && (a.dbg & 4);

    // state
    T cur[R], prv[R], Bp[R], Bpp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { cur[r] = 0; prv[r] = 0; Bp[r] = 0; Bpp[r] = 0; }
    T Uprev = 0, Fprev = 0;
    int have_up = 0, have_left = 0;       // progress of the producers as last seen
    int room_dn = 0, room_rt = 0;         // the consumers' progress as last seen (ring space): the wavefront below / DRAIN
    int blocks_out = 0;                   // inverse: blocks STORE has taken out of this wavefront's tbuf, as last seen
    szh_u64 *const trc = rb_trace_slot(a, TI, TJ, w);
    szh_u64 *const tln = (w == 0 || w == W - 1) ? rb_timeline(a, TI, TJ, w == 0 ? 0 : 1) : nullptr;
    szh_u64 tw_up = 0, tw_left = 0, tw_room = 0, tw_store = 0;
    if (trc && lane == 0) { trc[0] = rb_wall(); trc[6] = rb_xcc(); }

    typedef T xbuf_t[R][U];
    typedef unsigned cbuf_t[R][U / 2];    // 16 codes per row, two per register
    xbuf_t xc, xn;                        // values of the current trip / of the next one (on their way)
    cbuf_t cc, cn;                        // codes: compress packs them into cc; decompress reads cc, cn is on its way

    auto load_x = [&](int tt0, xbuf_t &x) __attribute__((always_inline)) {           // values of the trip that starts at shifted step tt0
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int k0 = tt0 - sh - lane - r;
            const unsigned off = xrb ? (unsigned)(((((tt0 / U) * W + w) * R + r) * NVEC) * 1024 + lane * 16) : (unsigned)((rowrel[r] + k0) * (int)sizeof(T));
            const unsigned pitch = xrb ? 1024u : 16u;
#pragma unroll
            for (int v = 0; v < NVEC; ++v) {
                const v4u q = bload16(rsx, off + pitch * v);
                T tmp[VPT]; __builtin_memcpy(tmp, &q, 16);
#pragma unroll
                for (int e = 0; e < VPT; ++e) x[r][v * VPT + e] = tmp[e];
            }
        }
    };
    auto load_c = [&](int tt0, cbuf_t &c) __attribute__((always_inline)) {           // decompress: the codes of that trip
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int k0 = tt0 - sh - lane - r;
            // ribbon order (the layout compression stores, below): one coalesced 1 KB load per instruction; natural order: 16-byte pieces
            // of 64 different rows at 2-byte alignment (split by the memory pipeline: the 2.9 ms of the first inverse)
            const unsigned off = crb ? (unsigned)(((((tt0 / U) * W + w) * R + r) * (U / 8)) * 1024 + lane * 16) : (unsigned)((rowrel[r] + k0) * 2);
            const unsigned pitch = crb ? 1024u : 16u;
#pragma unroll
            for (int v = 0; v < U / 8; ++v) {
                const v4u q = bload16(rsc, off + pitch * v);
                c[r][4 * v] = q.x; c[r][4 * v + 1] = q.y; c[r][4 * v + 2] = q.z; c[r][4 * v + 3] = q.w;
            }
        }
    };
    // results of a finished trip: codes (compress) or values (decompress); `edge`: some position of the trip lies outside [0, r2)
    auto store_trip = [&](int tt0, bool edge, xbuf_t &x, cbuf_t &c) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int k0 = tt0 - sh - lane - r;
            if (!DEC) {          // ribbon order: 64 lanes x 16 contiguous bytes per store, whatever the position
                const unsigned off = (unsigned)(((((tt0 / U) * W + w) * R + r) * (U / 8)) * 1024 + lane * 16);
#pragma unroll
                for (int v = 0; v < U / 8; ++v) { v4u q = {c[r][4 * v], c[r][4 * v + 1], c[r][4 * v + 2], c[r][4 * v + 3]}; bstore16(rsc, off + 1024u * v, q); }
                continue;
            }
            if (DEC && xrb) {            // mode 2: back into the ribbon-order value array, 64 lanes x 16 contiguous bytes per store
                const unsigned off = (unsigned)(((((tt0 / U) * W + w) * R + r) * NVEC) * 1024 + lane * 16);
#pragma unroll
                for (int v = 0; v < NVEC; ++v) {
                    T tmp[VPT];
#pragma unroll
                    for (int e = 0; e < VPT; ++e) tmp[e] = x[r][v * VPT + e];
                    v4u q; __builtin_memcpy(&q, tmp, 16);
                    bstore16(rsx, off + 1024u * v, q);
                }
                continue;
            }
            if (DEC) {
                // The trip's U values of this row are U * sizeof(T) contiguous bytes per lane -- in 64 different rows of the array.  Stored
                // from here they cost 1.5 ms of the 2.5 ms the first inverse took (the sweep itself: 0.59 ms): this wavefront's one
                // memory counter makes every wait for a trip's inputs wait for the previous trip's scattered stores as well.  So the
                // trip's results (both rows: one hand-over per trip) go to LDS and the STORE wavefront (store_out) writes them,
                // transposed: NVEC neighbouring lanes write one row's bytes, an instruction covers 64 / NVEC rows with whole
                // 64-byte pieces.  This wavefront issues loads only.  tbuf: [w][r][lane][U], the 16-byte pieces of a lane XOR-swizzled
                // by the lane so that both sides are free of bank conflicts without padding (LDS is full: 159 KB with this buffer).
                constexpr int HB = szh_rb_hb<T>::HB;
                SZH_LDS T *const tb = L.tbuf + (w * HB + r % HB) * 64 * U;
                const int n = (tt0 / U) * (R / HB) + r / HB;                  // hand-over number
                if (r % HB == 0 && blocks_out < n) { const szh_u64 c0 = trc ? rb_cyc() : 0; blocks_out = wait_word(L.Q + 2 * w + 1, n, a.err); if (trc) tw_store += rb_cyc() - c0; }
#pragma unroll
                for (int v = 0; v < NVEC; ++v) {
                    T tmp[VPT];
#pragma unroll
                    for (int e = 0; e < VPT; ++e) tmp[e] = x[r][v * VPT + e];
                    v4u q; __builtin_memcpy(&q, tmp, 16);
                    *(SZH_LDS v4u *)(tb + lane * U + ((v ^ (lane % NVEC)) * VPT)) = q;
                }
                if (r % HB == HB - 1) { lds_fence(); if (lane == 0) lds_st(L.Q + 2 * w, (unsigned)(n + 1)); }
                continue;
            }
            if (!rowok[r]) continue;
            if (!edge) {
            } else {
#pragma unroll
                for (int s = 0; s < U; ++s) {
                    const int k = k0 + s;
                    if (DEC && (unsigned)k < (unsigned)r2) bstoreT(rsx, (unsigned)((rowrel[r] + k) * (int)sizeof(T)), x[r][s]);
                }
            }
        }
    };

    // Waits are bounded: a lost hand-off must end the launch, not hang the GPU.  After the first timeout every later wait of this
    // wavefront gives up at once (the results are garbage by then; the host sees the error flag).
    // (No memory access inside the sweep's wait loops: hipcc then counts the sweep's stores exactly and the wait at the top of a trip
    //  covers the loads it is for, not the granule stores issued a moment ago.  The flag is stored at the end.)
    unsigned spin_limit = 1u << 22;
    bool timed_out = false;
    int to_step = -1;                      // development: shifted step of the first timeout
    int cur_tt = 0;
    auto give_up = [&](unsigned &spins) __attribute__((always_inline)) -> bool {
        if (++spins <= spin_limit) return false;
        if (SZH_DEV && !timed_out) to_step = cur_tt;
        timed_out = true; spin_limit = 0;
        return true;
    };
    auto wait_counter = [&](const SZH_LDS unsigned *p, int need) __attribute__((always_inline)) -> int {
        int v = uni((int)lds_ld(p));
        unsigned spins = 0;
#pragma unroll 1
        while (v < need) { if (give_up(spins)) return SZH_RB_INF; nap(1); v = uni((int)lds_ld(p)); }
        return v;
    };
    const T eb2 = eb + eb, rh = recip * (T)0.5, caph = (T)cap_lor * (T)0.5, radf = (T)radius;

    // one trip of U steps.  EDGE: positions outside [0, r2) occur (their inputs read as zero; decompress / mean: they hand on zeros)
    auto trip = [&](const int tt0, auto edge_tag, xbuf_t &x, cbuf_t &c) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        // ring positions of this trip (a trip does not wrap a ring: base + immediate per step)
        const SZH_LDS T *const rin_t = rin + (tt0 & rin_mask) * 64;
        SZH_LDS T *const rout_t = rout + (tt0 & (S::RL - 1)) * 64;
        const SZH_LDS T *const lrd_t = lrd + (tt0 & (S::RLL - 1)) * LS;
        SZH_LDS T *const rwr_t = rwr + (tt0 & (S::RLR - 1)) * WR;
        // ring space for this trip's writes (the consumers lag by a few steps at most: checked per trip against the whole trip)
        if (room_dn < tt0 + U - S::RL) { const szh_u64 c0 = trc ? rb_cyc() : 0; room_dn = wait_counter(pnext, tt0 + U - S::RL); if (trc) tw_room += rb_cyc() - c0; }
        if (has_right && room_rt < tt0 + U - S::RLR) { const szh_u64 c0 = trc ? rb_cyc() : 0; room_rt = wait_counter(pdrain, tt0 + U - S::RLR); if (trc) tw_room += rb_cyc() - c0; }
        // the left rings are filled far ahead (the tile on the left runs >= 64 steps in front): one check per trip as a rule
        if (have_left < tt0 + U) { have_left = uni((int)lds_ld(pin + 1)); }
#pragma unroll
        for (int s = 0; s < U; ++s) {
            const int tt = tt0 + s;
            if (SZH_DEV) cur_tt = tt;
            if (have_left < tt + 1) { const szh_u64 c0 = trc ? rb_cyc() : 0; have_left = wait_counter(pin + 1, tt + 1); if (trc) tw_left += rb_cyc() - c0; }
            T hl[R + 1];                   // lane 0's left neighbours of this step: {corner, b_0 .. b_{R-1}} (every lane reads the same words)
#pragma unroll
            for (int e = 0; e <= R; ++e) hl[e] = lds_ld(lrd_t + s * LS + e);
            // the value of the row above, from ring w.  Counter and slot are read together: the LDS executes a wavefront's accesses in
            // order, so a counter that covers this step proves the slot read behind it (one round trip instead of two)
            T Unow = lds_ld(rin_t + s * 64);                 // (no tile above: FILL_U has filled ring 0 with the zero halo)
            if (have_up < tt + 1) {
                const szh_u64 c0 = trc ? rb_cyc() : 0;
                unsigned spins = 0;
#pragma unroll 1
                for (;;) {
                    have_up = uni((int)lds_ld(pin));
                    Unow = lds_ld(rin_t + s * 64);
                    if (have_up >= tt + 1 || give_up(spins)) break;
                    nap(1);
                }
                if (trc) tw_up += rb_cyc() - c0;
            }
            const T Fnow = shr1(hl[0], Uprev);
            T B[R], nv[R];
#pragma unroll
            for (int r = 0; r < R; ++r) B[r] = shr1(hl[1 + r], cur[r]);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const T c_ = r == 0 ? Unow : cur[r - 1], e_ = r == 0 ? Uprev : prv[r - 1];
                const T f_ = r == 0 ? Fnow : Bp[r - 1], g_ = r == 0 ? Fprev : Bpp[r - 1];
                // [-1] + [-s1] + [-s0] - [-s1-1] - [-s0-1] - [-s0-s1] + [-s0-s1-1], left to right (sz_float.c:7268)
                const T pred = cur[r] + B[r] + c_ - Bp[r] - e_ - f_ + g_;
                bool act = true;
                if (EDGE) act = (unsigned)(tt - sh - lane - r) < (unsigned)r2;
                T v;
                if (!DEC) {
                    const T xv = EDGE ? (act ? x[r][s] : (T)0) : x[r][s];
                    T rc;
                    int code = rb_quant<T>(xv, pred, eb, eb2, rh, caph, radf, &rc);
                    if (USEMEAN) {
                        if (code != 0 && code <= radius) code -= 1;                  // sz_float.c:6944
                        if (szh_abs(xv - mean) <= eb) { code = radius; rc = mean; }  // sz_float.c:6929
                        if (EDGE) rc = act ? rc : (T)0;
                    }
                    v = rc;
                    if (s & 1) c[r][s >> 1] |= (unsigned)code << 16; else c[r][s >> 1] = (unsigned)code;
                    keepu(c[r][s >> 1]);      // here, not where the trip's codes are stored (hipcc sinks the packing there and keeps 32 lane masks alive)
                } else {
                    const unsigned cw = c[r][s >> 1];
                    const int c0 = (int)((s & 1) ? (cw >> 16) : (cw & 0xffffu));
                    int cq = c0;
                    bool is_mean = false;
                    if (USEMEAN) { is_mean = cq == radius; if (cq != 0 && cq < radius) cq += 1; }    // szd_float.c:3784
                    v = pred + (T)(cq - radius) * eb2;          // = pred + (T)(2 (c - radius)) * eb (szd_float.c:5786): 2 (c - radius) and 2 eb are exact
                    if (USEMEAN && is_mean) v = mean;
                    if (c0 == 0) v = x[r][s];                                                      // pre-scattered unpredictable value
                    if (EDGE) v = act ? v : (T)0;
                    x[r][s] = v;
                }
                nv[r] = v;
            }
            // hand on: the last row to the wavefront below (the tile's last wavefront: to DRAIN), lane 63's rows to DRAIN
            lds_st(rout_t + s * 64, nv[R - 1]);
            if (has_right && lane == 63) {
#pragma unroll
                for (int r = 0; r < R; ++r) lds_st(rwr_t + s * WR + r, nv[r]);
            }
            lds_order();
            lds_st(pout, (unsigned)(tt + 1));
#pragma unroll
            for (int r = 0; r < R; ++r) { prv[r] = cur[r]; cur[r] = nv[r]; Bpp[r] = Bp[r]; Bp[r] = B[r]; }
            Uprev = Unow; Fprev = Fnow;
        }
    };
    auto is_edge = [&](int tt0) { return !(tt0 - sh - 63 - (R - 1) >= 0 && tt0 - sh + U - 1 < r2); };

    // Everything a trip needs from HBM was requested a whole trip earlier, into the `next` registers.  At the top of a trip: the wait
    // for it (`keep`; hipcc makes it a wait for the wavefront's whole memory queue), then the results of the PREVIOUS trip leave
    // (they sat in the `current` registers: by the next wait these stores are a whole trip old), then the copy next -> current and
    // the request for the trip after this one.
    if (!dbg_nold) { load_x(0, xn); if (DEC) load_c(0, cn); }
    for (int tt0 = 0; tt0 < NT; tt0 += U) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int s = 0; s < U; ++s) keep(xn[r][s]);
        }
        if (DEC) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int s = 0; s < U / 2; ++s) keepu(cn[r][s]);
            }
        }
        if (tt0 > 0 && !dbg_nost) store_trip(tt0 - U, is_edge(tt0 - U), xc, cc);
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int s = 0; s < U; ++s) xc[r][s] = xn[r][s];
        }
        if (DEC) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int s = 0; s < U / 2; ++s) cc[r][s] = cn[r][s];
            }
        }
        if (tt0 + U < NT && !dbg_nold) { load_x(tt0 + U, xn); if (DEC) load_c(tt0 + U, cn); }
        if (is_edge(tt0)) trip(tt0, std::true_type{}, xc, cc); else trip(tt0, std::false_type{}, xc, cc);
        if (trc && lane == 0 && tt0 == 0) trc[1] = rb_wall();
        if (tln && lane == 0 && tt0 / U < 48) tln[tt0 / U] = rb_wall();
    }
    if (!dbg_nost) store_trip(NT - U, is_edge(NT - U), xc, cc);
    if (trc && lane == 0) { trc[2] = rb_wall(); trc[3] = tw_up; trc[4] = tw_left; trc[5] = tw_room + (tw_store << 32); trc[7] = timed_out ? (szh_u64)(1000000 + to_step) : (szh_u64)NT; }
    if (timed_out) st_flag(a.err, 1u);
}

// ------------------------------------------------------------------------------------------------------------------ helpers
// granule rows: down faces  DF[tile][step][word][64 lanes], right faces RF[tile][step][word][W * R rows]; `step` is the producing
// wavefront's shifted step
template <class T> __device__ __forceinline__ int64_t tile_id(const szh_qargs<T> &a, int TI, int TJ) { return (int64_t)TI * a.nJ + TJ; }

// DRAIN (stores only): the ring of the tile's last wavefront -> down-face granule rows, lane 63's rows of every wavefront -> right-face
// granule rows.  One round handles up to KD steps of each; a nap between rounds (a helper that polls LDS without a pause takes the
// issue slots of the compute wavefronts on its SIMD).  P[W + 1].cnt = steps forwarded of EVERYTHING = ring space for the writers.
template <class T>
__device__ __forceinline__ void drain(const szh_qargs<T> &a, int TI, int TJ, const lds_t<T> &L)
{
    using S = szh_rb_shape<T>;
    constexpr int W = S::W, R = S::R, WR = W * R, NW = szh_gran<T>::NW, KD = SZH_RB_KD;
    const int lane = (int)(threadIdx.x & 63);
    const int NT = szh_rb_steps_of<T>(a.G.g2.count);
    SZH_LDS unsigned *const mine = L.P + 2 * (W + 1);
    const bool has_down = TI + 1 < a.nI, has_right = TJ + 1 < a.nJ;
    if (!has_down && !has_right) { lds_st(mine, (unsigned)SZH_RB_INF); return; }
    const int64_t tid = tile_id(a, TI, TJ);
    szh_u64 *const dstD = a.faceI + tid * NT * NW * 64 + lane;
    const bool enR = has_right && lane < WR;
    const int q = enR ? lane : 0, wq = q / R;
    szh_u64 *const dstR = a.faceJ + tid * NT * NW * WR + q;
    const SZH_LDS T *const ringD = L.ring + S::RLU * 64 + (W - 1) * S::RL * 64 + lane;
    const SZH_LDS T *const ringR = L.rr + q;
    const SZH_LDS unsigned *const srcD = L.P + 2 * W, *const srcR = L.P + 2 * (wq + 1);
    int nd = 0, nr = enR ? 0 : NT;            // steps forwarded: down face (uniform), right face (per row)
    unsigned idle = 0;
    szh_u64 *const trc = rb_trace_slot(a, TI, TJ, W);
    szh_u64 *const tlD = rb_timeline(a, TI, TJ, 2);
    szh_u64 rounds = 0, empty = 0;
    if (trc && lane == 0) { trc[0] = rb_wall(); trc[6] = rb_xcc(); }
    for (;;) {
        if (trc) ++rounds;
        const int cd = uni((int)lds_ld(srcD));
        int md = cd - nd; if (md > KD) md = KD;
#pragma unroll
        for (int e = 0; e < KD; ++e) {
            if (e < md) {                                  // (the ring is read either way: it also frees the slot)
                const T v = lds_ld(ringD + ((nd + e) & (S::RL - 1)) * 64);
                if (has_down) {
                    szh_u64 g[NW]; szh_gran<T>::pack(v, a.epoch, g);
#pragma unroll
                    for (int wd = 0; wd < NW; ++wd) st_gran(dstD + ((int64_t)(nd + e) * NW + wd) * 64, g[wd]);
                }
            }
        }
        if (md > 0) { if (tlD && lane == 0 && ((nd + md) >> 4) != (nd >> 4) && ((nd + md) >> 4) <= 48) tlD[((nd + md) >> 4) - 1] = rb_wall(); nd += md; }
        const int cr = (int)lds_ld(srcR);
        int mr = cr - nr; if (mr > KD) mr = KD; if (mr < 0) mr = 0;
#pragma unroll
        for (int e = 0; e < KD; ++e) {
            if (e < mr) {
                const T v = lds_ld(ringR + ((nr + e) & (S::RLR - 1)) * WR);
                szh_u64 g[NW]; szh_gran<T>::pack(v, a.epoch, g);
#pragma unroll
                for (int wd = 0; wd < NW; ++wd) st_gran(dstR + ((int64_t)(nr + e) * NW + wd) * WR, g[wd]);
            }
        }
        nr += mr;
        // what every ring has been read up to
        int mn = nr < nd ? nr : nd;
        for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(mn, o, 64); mn = t < mn ? t : mn; }
        lds_fence();                        // the ring slots have been read
        lds_st(mine, (unsigned)mn);
        if (mn >= NT) break;
        if (md <= 0 && __all(mr == 0 ? 1 : 0)) {
            if (trc) ++empty;
            if (++idle > (1u << 22)) { st_flag(a.err, 1u); break; }
            if ((idle & 1023u) == 0 && uni((int)ld_flag(a.err)) != 0) break;
        } else idle = 0;
        nap(SZH_RB_DRAIN_NAP);
    }
    if (trc && lane == 0) { trc[2] = rb_wall(); trc[3] = rounds; trc[4] = empty; trc[7] = (szh_u64)nd; }
}

// FILL_U: down-face granules of the tile above -> ring 0 (read by wavefront 0).  Consumer index u = the producer's step - W (R - 1).
template <class T>
__device__ __forceinline__ void fill_up(const szh_qargs<T> &a, int TI, int TJ, const lds_t<T> &L)
{
    using S = szh_rb_shape<T>;
    constexpr int W = S::W, R = S::R, NW = szh_gran<T>::NW, ADD = W * (R - 1);
    const int lane = (int)(threadIdx.x & 63);
    const int NT = szh_rb_steps_of<T>(a.G.g2.count);
    SZH_LDS T *const ring = L.ring + lane;
    if (TI == 0) {                             // no tile above: the zero halo
        for (int e = 0; e < S::RLU; ++e) lds_st(ring + e * 64, (T)0);
        lds_fence();
        lds_st(L.P, (unsigned)SZH_RB_INF);
        return;
    }
    const szh_u64 *const src = a.faceI + tile_id(a, TI - 1, TJ) * NT * NW * 64 + lane;
    const SZH_LDS unsigned *const cons = L.P + 2;      // steps completed by wavefront 0
    const int NU = NT - ADD;
    int n = 0, win = 2;
    unsigned idle = 0;
    szh_u64 *const trc = rb_trace_slot(a, TI, TJ, W + 2);
    constexpr int KF = 16;              // granule rows in flight per round
    szh_u64 *const tlU = rb_timeline(a, TI, TJ, 3);
    szh_u64 rounds = 0, empty = 0, noroom = 0;
    if (trc && lane == 0) { trc[0] = rb_wall(); trc[6] = rb_xcc(); }
    while (n < NU) {
        if (trc) ++rounds;
        const int room = uni((int)lds_ld(cons)) + S::RLU - n;
        if (trc && room <= 0) ++noroom;
        int m = win; if (m > room) m = room; if (m > NU - n) m = NU - n; if (m > KF) m = KF;
        szh_u64 g[KF][NW];
#pragma unroll
        for (int e = 0; e < KF; ++e) {
#pragma unroll
            for (int wd = 0; wd < NW; ++wd) g[e][wd] = e < m ? ld_gran(src + ((int64_t)(n + e + ADD) * NW + wd) * 64) : 0;
        }
        int lead = 0; bool run = true;
#pragma unroll
        for (int e = 0; e < KF; ++e) {
            bool ok = e < m;
#pragma unroll
            for (int wd = 0; wd < NW; ++wd) ok = ok && (unsigned)(g[e][wd] >> 32) == a.epoch;
            run = run && (__all(ok ? 1 : 0) != 0);
            if (run) { lds_st(ring + ((n + e) & (S::RLU - 1)) * 64, szh_gran<T>::unpack(g[e])); ++lead; }
        }
        if (lead > 0) {
            if (tlU && lane == 0 && ((n + lead) >> 4) != (n >> 4) && ((n + lead) >> 4) <= 48) tlU[((n + lead) >> 4) - 1] = rb_wall();
            n += lead;
            lds_order();
            lds_st(L.P, (unsigned)(n >= NU ? SZH_RB_INF : n));
            idle = 0;
        } else {
            if (trc) ++empty;
            if (++idle > (1u << 22)) { st_flag(a.err, 1u); break; }
            if ((idle & 1023u) == 0 && uni((int)ld_flag(a.err)) != 0) break;
            nap(a.backoff > 0 ? a.backoff : 1);
        }
        win = lead + 2;
    }
    if (trc && lane == 0) { trc[2] = rb_wall(); trc[3] = rounds; trc[4] = empty; trc[5] = noroom; trc[7] = (szh_u64)n; }
}

// FILL_L: right-face granules of the tile on the left (rows 1 ..) and of the tile up-left (row 0: the corner column of wavefront 0)
// -> left rings.  Consumer index = the producer's step - 63 (- W (R - 1) for row 0); one ring row per lane.
template <class T>
__device__ __forceinline__ void fill_left(const szh_qargs<T> &a, int TI, int TJ, const lds_t<T> &L)
{
    using S = szh_rb_shape<T>;
    constexpr int W = S::W, R = S::R, WR = W * R, LS = WR + 2, NW = szh_gran<T>::NW;
    const int lane = (int)(threadIdx.x & 63);
    const int NT = szh_rb_steps_of<T>(a.G.g2.count);
    for (int e = lane; e < S::RLL * LS; e += 64) lds_st(L.lr + e, (T)0);      // zero halo wherever nothing arrives
    lds_fence();
    if (TJ == 0) { if (lane < W) lds_st(L.P + 2 * lane + 1, (unsigned)SZH_RB_INF); return; }
    const bool en = lane <= WR && !(lane == 0 && TI == 0);
    const int qq = lane <= WR ? lane : 0;
    const int add = 63 + (qq == 0 ? W * (R - 1) : 0);
    const int64_t stile = qq == 0 ? tile_id(a, TI > 0 ? TI - 1 : 0, TJ - 1) : tile_id(a, TI, TJ - 1);
    const int srow = qq == 0 ? WR - 1 : qq - 1;
    const szh_u64 *const src = a.faceJ + stile * NT * NW * WR + srow;
    // who reads row qq: wavefront (qq - 1) / R as a neighbour row (qq >= 1), wavefront qq / R as its corner column (qq % R == 0, qq < WR)
    const int wa = qq >= 1 ? (qq - 1) / R : 0, wb = (qq % R == 0 && qq < WR) ? qq / R : wa;
    const SZH_LDS unsigned *const consa = L.P + 2 * (wa + 1), *const consb = L.P + 2 * (wb + 1);
    const int NLq = NT - add;
    int n = en ? 0 : NLq, win = 2;
    unsigned idle = 0;
    for (;;) {
        const int ca = (int)lds_ld(consa), cb = (int)lds_ld(consb);
        const int room = (ca < cb ? ca : cb) + S::RLL - n;
        int m = win; if (m > room) m = room; if (m > NLq - n) m = NLq - n; if (m > 8) m = 8; if (m < 0 || !en) m = 0;
        szh_u64 g[8][NW];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int wd = 0; wd < NW; ++wd) g[e][wd] = e < m ? ld_gran(src + ((int64_t)(n + e + add) * NW + wd) * WR) : 0;
        }
        int lead = 0; bool run = true;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bool ok = e < m;
#pragma unroll
            for (int wd = 0; wd < NW; ++wd) ok = ok && (unsigned)(g[e][wd] >> 32) == a.epoch;
            run = run && ok;
            if (run) { lds_st(L.lr + ((n + e) & (S::RLL - 1)) * LS + qq, szh_gran<T>::unpack(g[e])); ++lead; }
        }
        n += lead;
        win = lead + 2;
        // rows complete for wavefront w: the minimum over its R rows and its corner row
        lds_st(L.scratch + lane, (lane > WR || n >= NLq) ? SZH_RB_INF : n);
        lds_fence();
        if (lane < W) {
            int mn = SZH_RB_INF;
#pragma unroll
            for (int r = 0; r <= R; ++r) { const int x = lds_ld(L.scratch + lane * R + r); mn = x < mn ? x : mn; }
            lds_st(L.P + 2 * lane + 1, (unsigned)mn);
        }
        lds_fence();
        if (__all(n >= NLq ? 1 : 0)) break;
        if (__all(lead == 0 ? 1 : 0)) {
            if (++idle > (1u << 22)) { st_flag(a.err, 1u); break; }
            if ((idle & 1023u) == 0 && uni((int)ld_flag(a.err)) != 0) break;
            nap(a.backoff > 0 ? a.backoff : 1);
        } else idle = 0;
    }
}
// STORE (inverse only): writes the trips' results the compute wavefronts leave in tbuf to the output array.  Stores only.
// One wavefront serves the W compute wavefronts of the tile, so a hand-over must cost it little: the acknowledged counts live in
// registers (the loop over w is unrolled), all W request words are read with one LDS access, everything per lane is computed once,
// and a hand-over carries both rows of a trip.  (Row by row it took 0.5 us per row and the compute wavefronts spent a third of their
// time waiting for it: tools/gpu_rb_trace.py, `store` column and STORE line.)
template <class T> __device__ __forceinline__ void store_out(const szh_qargs<T> &a, const int TI, const int TJ, const lds_t<T> &L)
{
    if (a.codes_ribbon == 2) return;                 // (the compute wavefronts store ribbon-order results themselves: coalesced)
    using S = szh_rb_shape<T>;
    constexpr int R = S::R, W = S::W, U = S::U, WR = W * R, HB = szh_rb_hb<T>::HB;
    constexpr int VPT = 16 / (int)sizeof(T), NVEC = U / VPT;
    const szh_geom3 &G = a.G;
    const int r0 = G.g0.count, r1 = G.g1.count, r2 = G.g2.count;
    const int lane = (int)(threadIdx.x & 63);
    const int NT = szh_rb_steps_of<T>(r2), ntrips = NT / U * (R / HB);       // hand-overs per wavefront
    const int64_t tile_base = (int64_t)TI * WR * G.d0;
    const uint64_t span_x = (uint64_t)(G.n - tile_base) * sizeof(T);
    const rsrc_t rsx = make_rsrc((T *)a.out + tile_base, span_x > 0x7ffffff0ull ? 0x7ffffff0u : (unsigned)span_x);
    const int seg = lane % NVEC, jl = lane / NVEC;
    // per lane and piece q4: the column of dim 1 it belongs to, where that column's row starts, the position offset of the piece
    int ldsoff[NVEC], eloff[NVEC]; bool colok[NVEC];
#pragma unroll
    for (int q4 = 0; q4 < NVEC; ++q4) {
        const int jj = jl + (64 / NVEC) * q4, jg = TJ * 64 + jj;
        ldsoff[q4] = jj * U + ((seg ^ (jj % NVEC)) * VPT);
        colok[q4] = jg < r1;
        eloff[q4] = (jg < r1 ? jg : r1 - 1) * (int)G.d1 - jj + seg * VPT;      // + the row's start + (tt0 - sh - r) = the element offset of the piece
    }
    int done[W];
#pragma unroll
    for (int w = 0; w < W; ++w) done[w] = 0;
    unsigned idle = 0;
    szh_u64 *const trc = rb_trace_slot(a, TI, TJ, W + 3);
    szh_u64 t_busy = 0, n_scan = 0, n_blk = 0;
    if (trc && lane == 0) trc[0] = rb_wall();
    for (;;) {
        bool any = false, left = false;
        const szh_u64 c0 = trc ? rb_cyc() : 0;
        ++n_scan;
        const unsigned reqs = lds_ld(L.Q + 2 * (lane < W ? lane : 0));
#pragma unroll
        for (int w = 0; w < W; ++w) {
            if (done[w] >= ntrips) continue;
            left = true;
#ifdef SZH_HIPSIM
            const int req = (int)__shfl(reqs, w, 64);
#else
            const int req = __builtin_amdgcn_readlane((int)reqs, w);
#endif
            if (req <= done[w]) continue;
            any = true;
            const int n = done[w], tt0 = (n / (R / HB)) * U, rb0 = (n % (R / HB)) * HB, sh = w * (R - 1);
            v4u q[HB][NVEC];
#pragma unroll
            for (int r = 0; r < HB; ++r) {
                const SZH_LDS T *const tb = L.tbuf + (w * HB + r) * 64 * U;
#pragma unroll
                for (int q4 = 0; q4 < NVEC; ++q4) q[r][q4] = *(const SZH_LDS v4u *)(tb + ldsoff[q4]);
            }
            lds_fence();                                                  // the words are in registers: the buffer may be overwritten
            done[w] = n + 1; ++n_blk;
            if (lane == 0) lds_st(L.Q + 2 * w + 1, (unsigned)(n + 1));
#pragma unroll
            for (int rr = 0; rr < HB; ++rr) {
                const int r = rb0 + rr;
                const int i = TI * WR + w * R + r;
                if (i >= r0) continue;
                const int kb = tt0 - sh - r;                               // position of lane 0's first value of the trip
                const int base = (int)((int64_t)(i - TI * WR) * G.d0) + kb;
                if (kb - 63 >= 0 && kb + U <= r2) {                        // every position of the row's block lies inside the row
#pragma unroll
                    for (int q4 = 0; q4 < NVEC; ++q4) if (colok[q4]) bstore16(rsx, (unsigned)((base + eloff[q4]) * (int)sizeof(T)), q[rr][q4]);
                } else {
#pragma unroll
                    for (int q4 = 0; q4 < NVEC; ++q4) {
                        if (!colok[q4]) continue;
                        const int jj = jl + (64 / NVEC) * q4, k = kb - jj + seg * VPT;
                        const unsigned off = (unsigned)((base + eloff[q4]) * (int)sizeof(T));
                        if (k >= 0 && k + VPT <= r2) bstore16(rsx, off, q[rr][q4]);
                        else {
                            T tmp[VPT]; __builtin_memcpy(tmp, &q[rr][q4], 16);
#pragma unroll
                            for (int e = 0; e < VPT; ++e) if ((unsigned)(k + e) < (unsigned)r2) bstoreT(rsx, off + (unsigned)(e * (int)sizeof(T)), tmp[e]);
                        }
                    }
                }
            }
        }
        if (trc && any) t_busy += rb_cyc() - c0;
        if (!left) break;
        if (!any) {
            if ((++idle & 1023u) == 0 && uni((int)ld_flag(a.err)) != 0) break;      // a compute wavefront gave up: the launch is lost anyway
            nap(1);
        } else idle = 0;
    }
    if (trc && lane == 0) { trc[2] = rb_wall(); trc[3] = t_busy; trc[4] = n_scan; trc[5] = n_blk; }
}
} // namespace szh_rb

// a.nI x a.nJ = the TILE grid here; a.faceI / a.faceJ = the down- / right-face granule rows
template <class T, bool DEC, bool USEMEAN>
__global__ __launch_bounds__((szh_rb_shape<T>::W + 3 + (DEC ? 1 : 0)) * 64) void k_ribbon(szh_qargs<T> a)
{
    using S = szh_rb_shape<T>;
    using namespace szh_rb;
    constexpr int W = S::W, R = S::R, WR = W * R, LS = WR + 2;
    __shared__ T ring[(S::RLU + W * S::RL) * 64];
    __shared__ T lr[S::RLL * LS];
    __shared__ T rr[S::RLR * WR];
    __shared__ unsigned P[2 * (W + 2)];
    __shared__ int scratch[64];
    __shared__ __attribute__((aligned(16))) T tbuf[DEC ? W * szh_rb_hb<T>::HB * 64 * S::U : 4];
    __shared__ unsigned Q[2 * W];
    __shared__ unsigned tk_s;
    // PERSISTENT: gridDim.x workgroups walk the tiles in ticket order (the order of szh_pencil_order_at: a tile's predecessors have
    // smaller tickets, so the workgroup on the smallest unfinished ticket can always run -- no deadlock as long as the grid is
    // resident, one workgroup per CU).  With one workgroup per tile, 85 % of the resident workgroups were polling for predecessors that
    // had not started: ~40 of 256 tiles are active at a time (the dependency front), the others held their CU's LDS and registers
    // against the other arrays in flight and read granule rows over and over.
    const lds_t<T> L{(SZH_LDS T *)ring, (SZH_LDS T *)lr, (SZH_LDS T *)rr, (SZH_LDS unsigned *)P, (SZH_LDS int *)scratch, (SZH_LDS T *)tbuf, (SZH_LDS unsigned *)Q};
    const int w = uni((int)(threadIdx.x >> 6));
#ifndef SZH_RB_PRIO
#define SZH_RB_PRIO 1
#endif
#ifndef SZH_RB_PRIO_MIN
#define SZH_RB_PRIO_MIN 0            /* priority of the compute wavefronts 3 .. W-1 (other kernels' wavefronts run at 0) */
#endif
    if (SZH_RB_PRIO) prio(w >= W ? 3 : (w == 0 ? 2 : (w < 3 ? 1 : SZH_RB_PRIO_MIN)));
    const unsigned ntiles = (unsigned)(a.nI * a.nJ);
    int prev_tile = -1;
    for (unsigned it = 0;; ++it) {
        __syncthreads();                                             // the previous tile is finished by every wavefront
        // ... and its codes are published: a release at system scope writes this XCD's L2 back before the word goes to the host, so
        // a kernel the host launches on seeing it (another XCD, its L2 invalidated at launch) reads them from memory
        if (!DEC && a.tile_done && prev_tile >= 0 && threadIdx.x == 0) st_done(a.tile_done + prev_tile, a.epoch);
        if (threadIdx.x < 2 * (W + 2)) P[threadIdx.x] = 0;
        if (threadIdx.x < 2 * W) Q[threadIdx.x] = 0;
        if (threadIdx.x == 0) {
            const unsigned t = a.ticket_mode ? blockIdx.x + it * gridDim.x : atomicAdd(a.ticket, 1u);
            tk_s = t < ntiles ? szh_pencil_order_at(a.nI, a.nJ, t) : 0xffffffffu;
        }
        __syncthreads();
        const unsigned ij = (unsigned)uni((int)tk_s);
        if (ij == 0xffffffffu) break;
        const int TI = (int)(ij >> 16), TJ = (int)(ij & 0xffffu);
        prev_tile = TI * a.nJ + TJ;
        if (w < W) ribbon_body<T, DEC, USEMEAN>(a, TI, TJ, w, L);
        else if (w == W) drain<T>(a, TI, TJ, L);
        else if (w == W + 1) fill_up<T>(a, TI, TJ, L);
        else if (w == W + 2) fill_left<T>(a, TI, TJ, L);
        else store_out<T>(a, TI, TJ, L);
    }
}
#endif
