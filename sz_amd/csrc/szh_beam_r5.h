// szh_beam.h -- the predict+quantise (and inverse) sweep of the SZ 2.1 path, third mapping ("beam", round 5).
//
// Same arithmetic as szh_pencil.h / szh_ribbon.h (7-point Lorenzo from RECONSTRUCTED neighbours, sz/src/sz_float.c:7253-7353, with the
// mean shortcut :6914-7030; inverse sz/src/szd_float.c:3483-5866), bit for bit the same codes; the shape is k_omp_col's (szh_ompcol.h), which
// round 4 measured at 48 % of the HBM roofline on the reference's OpenMP container, carried over to ONE dependency front over the whole array:
//
//   * a lane owns a COLUMN: one position k of the contiguous dimension, C1 = 4 consecutive rows j, ALL planes i, and walks its cells
//     (i, j) in row-major order, one cell per step; lane m runs m steps behind lane 0.  32 such lanes are a HALF-BEAM (32 k x 4 j x r0 i),
//     a wavefront is two half-beams side by side in j (the upper one a whole line behind), a workgroup four wavefronts side by side in j:
//     32 k x 32 j.  No barrier inside the sweep.
//       - (i, j, k-1) is what lane m - 1 produced one step ago: ONE DPP `wave_shr:1` per step (lane 0 of a half: the k-face of the beam on
//         its left, read from a small LDS ring);
//       - (i, j-1, k) is the lane's own previous result; (i-1, j, k) what it produced LINE = C1 + 1 steps ago: a DELAY LINE of LINE registers
//         indexed by the step number modulo LINE, a compile-time index in the unrolled line; the same for the left lane's values;
//       - a line (i, :) takes LINE steps: its first is the VIRTUAL cell j = j0 - 1, whose "result" is the j-face of the half-beam below it:
//         the upper half of a wavefront takes it from the lower half's register (v_permlane32_swap), the lower half of wavefront w + 1 finds
//         it in its own LDS ring, where wavefront w's lanes put it the step they made it; across workgroups it travels through HBM granules.
//         With that, ONE predictor expression -- the reference's 7-point sum in its order -- serves every cell;
//   * memory: the rows of a line are read with 16 bytes per lane (8 row pieces of 128 bytes per instruction), eight such loads in flight per
//     wavefront, and dropped into an LDS RING of RL = 9 lines of row slots indexed by the cell's step number; lane m reads ITS value of the
//     row it is at (ds_read, conflict-free: the slot pitch is a multiple of 32 banks) one step ahead of its use and writes its code (and its
//     reconstruction: the faces are read from there) next to it; finished rows of codes leave NATURAL order, 8 bytes per lane and line;
//   * between workgroups: the k-face (lane 31's value of every cell) and the j-face (the last row of every line) go through HBM-side granules
//     {launch epoch, value bits} -- 8-byte words written write-through, the data being its own flag (MI355X_MICROARCH "handoff-1to1") -- stored
//     once per line by the wavefront itself; the consumer requests them DK lines ahead, validates the tag when it needs them and asks again
//     (bounded) if they have not arrived.  Granule buffers hold the whole launch: no ring space to manage, nothing to clear (the tag is the epoch).
//   * regression blocks (sz_float.c:7153-7252): their points do not depend on any neighbour -- prediction a i + b j + c k + d from the DECODED
//     coefficients -- so a separate, fully parallel pass (k_reg_points) quantises them; this sweep only needs their RECONSTRUCTIONS as
//     neighbours: it reads them where the other points' values are read and passes them through (HASREG: a flag byte per cell in the ring).
//
// Longest dependency path of a launch (512^3 float): LINE r0 = 2560 steps of one wavefront + the start-up lag of the last one: 32 + hop per
// beam along k (15), 2 LINE + hop per wavefront along j (63) -- against 1536 steps + ~300 hops of 6 - 10 us for k_ribbon.
//
// Covers: 3-D arrays with r2 a multiple of 4, float / double, compress / decompress, mean shortcut, regression blocks.
#pragma once
// (included by szhip_kernels.h after szh_ribbon.h and szh_ompcol.h, whose helpers it uses)

#ifndef SZH_DEV
#define SZH_DEV 0
#endif
#ifndef SZH_BM_X
#define SZH_BM_X 0     /* tools/ubench/ub_beam.hip only (timing, results wrong): 1 no face push, 2 no code write, 4 no value write, 8 no k-face read, 16 no half swap, 32 no quantiser, 64 no value read */
#endif
namespace szh_bm {
using szh_oc::mask_t;
using szh_oc::lane_mask;
using szh_oc::in_mask;
using szh_oc::for_n;
using szh_oc::lds_get;
using szh_oc::lds_put;
using szh_oc::lds_get16;
using szh_oc::lds_put16;
using szh_oc::order;
using szh_oc::wave_sync;
typedef szh_rb::v4u v4u;

constexpr int C1 = 4, LINE = C1 + 1, HL = 32, WPG = 4, RL = 9, RS = RL * LINE, KRL = 4, DK = 4;
constexpr int JW = 2 * C1, JG = JW * WPG;          // rows j per wavefront / per workgroup
constexpr int LAG = 7;                              // lines after which every lane has left a line (31 steps of skew)
#define SZH_BM_INF (1 << 30)

template <class T, bool HASREG> struct shape {
    static constexpr int SZ = (int)sizeof(T), VPL = 16 / SZ, LPR = HL / VPL, RPE = 64 / LPR, RH = RPE / 2, EV = C1 / RH, DV = 8, UL = DV / EV;
    static constexpr int HB = HL * SZ, VB = 2 * HB, CB = 128, PITCH = VB + CB + (HASREG ? 128 : 0), RINGB = RS * PITCH, NW = szh_gran<T>::NW;
    static constexpr int FOFF = VB + CB;            // flag bytes of a slot (HASREG), one per lane
    static constexpr int KRB = KRL * LINE * 2 * SZ + 16; // k-face ring of a wavefront (+ a write-only word)
    static_assert(PITCH % 128 == 0 && UL % DK == 0 && RL >= LAG + 2, "ring geometry");
};

// compile-time lane masks (both halves alike: lane = 32 h + m)
template <int U> constexpr mask_t virt_mask() { mask_t x = 0; for (int l = 0; l < 64; ++l) if ((((l & 31) - U) % LINE + LINE) % LINE == 0) x |= 1ull << l; return x; }
template <int U> constexpr mask_t push_mask() { mask_t x = 0; for (int l = 32; l < 64; ++l) if (((U - (l & 31)) % LINE + LINE) % LINE == C1) x |= 1ull << l; return x; }
constexpr mask_t FIRSTCOL = 1ull | (1ull << 32), UPPER = 0xffffffff00000000ull;

#ifdef SZH_HIPSIM
struct u2_t { unsigned x, y; };
#else
typedef unsigned int u2_t __attribute__((ext_vector_type(2)));
#endif
__device__ __forceinline__ u2_t lds_get8(OC_LDS unsigned char *base, unsigned off) { return *(OC_LDS u2_t *)(base + off); }
__device__ __forceinline__ void lds_put8(OC_LDS unsigned char *base, unsigned off, u2_t v) { *(OC_LDS u2_t *)(base + off) = v; }

#ifdef SZH_HIPSIM
template <class T> static inline T shr1(T v) { return __shfl_up(v, 1, 64); }
template <class T> static inline T low_to_high(T v) { const int l = (int)(threadIdx.x & 63); return __shfl(v, l >= 32 ? l - 32 : l, 64); }
static inline int uni(int v) { return __shfl(v, 0, 64); }
static inline void nap1() { __builtin_amdgcn_s_sleep(1); }
#else
__device__ __forceinline__ float shr1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ double shr1(double v)
{
    const long long s = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)s, 0x138, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp((int)(s >> 32), 0x138, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// lanes 32 + m receive the value of lane m (v_permlane32_swap: the upper 32 lanes of the first operand are swapped with the lower 32 of the second)
__device__ __forceinline__ unsigned low_to_high_u(unsigned v) { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return r[0]; }
__device__ __forceinline__ float low_to_high(float v) { return __uint_as_float(low_to_high_u(__float_as_uint(v))); }
__device__ __forceinline__ double low_to_high(double v)
{
    const unsigned long long s = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = low_to_high_u((unsigned)s), hi = low_to_high_u((unsigned)(s >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void nap1() { __builtin_amdgcn_s_sleep(1); }
#endif

__device__ __forceinline__ unsigned ld_err(const unsigned *p) { return __hip_atomic_load(const_cast<unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_err(unsigned *p, unsigned v) { atomicMax(p, v); }      // (2 = the regression points did not arrive, 1 = a face did not: the larger one stays)

// geometry of a launch
struct grid_t { int nKB, nJG; };
SZH_HD grid_t make_grid(const szh_geom3 &G) { grid_t g; g.nKB = (G.g2.count + HL - 1) / HL; g.nJG = (G.g1.count + JG - 1) / JG; return g; }
// granule words: k-face [workgroup][8 half-beams][LINE r0 cells][NW], j-face [workgroup][r0 lines][32 lanes][NW]
template <class T> SZH_HD size_t kface_words(const szh_geom3 &G) { const grid_t g = make_grid(G); return (size_t)g.nKB * g.nJG * 8 * LINE * (size_t)G.g0.count * szh_gran<T>::NW; }
template <class T> SZH_HD size_t jface_words(const szh_geom3 &G) { const grid_t g = make_grid(G); return (size_t)g.nKB * g.nJG * (size_t)G.g0.count * HL * szh_gran<T>::NW; }

#ifdef SZH_HIPSIM
#define SZH_SB
#else
#define SZH_SB __builtin_amdgcn_sched_barrier(0)
#endif
#ifdef SZH_HIPSIM
template <class V> static inline void hide(V &) {}
#else
__device__ __forceinline__ void hide(unsigned &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void hide(float &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void hide(double &v) { asm volatile("" : "+v"(v)); }
#endif
// select by a per-lane constant bit mask (all ones / all zeros): v_bfi_b32, no lane mask to set up in the scalar unit
#ifdef SZH_HIPSIM
static inline unsigned bsel(unsigned mask, unsigned a, unsigned b) { return (a & mask) | (b & ~mask); }
#else
__device__ __forceinline__ unsigned bsel(unsigned mask, unsigned a, unsigned b) { unsigned r; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b)); return r; }
#endif
__device__ __forceinline__ float bsel(unsigned mask, float a, float b) { return __uint_as_float(bsel(mask, __float_as_uint(a), __float_as_uint(b))); }
__device__ __forceinline__ double bsel(unsigned mask, double a, double b)
{
    const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
    const unsigned lo = bsel(mask, (unsigned)ua, (unsigned)ub), hi = bsel(mask, (unsigned)(ua >> 32), (unsigned)(ub >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// raw buffer accesses: a lane whose offset is out of range reads zeros / stores nothing, so every access is issued unconditionally (hipcc
// then counts the accesses in flight exactly: with accesses under a branch its `s_waitcnt vmcnt` fell to 0 once per line -- a memory round
// trip every five steps).  `soff` is wavefront-uniform; the range check is against offset + soff.  AUX: 0 plain, 2 non-temporal, 17 = sc0 sc1
// (granules: written through, read past this XCD's copies)
typedef szh_rb::rsrc_t rsrc_t;
using szh_rb::make_rsrc;
#define SZH_BM_OOB 0xffffffffu
#ifdef SZH_HIPSIM
static inline bool inr(rsrc_t rs, unsigned off, unsigned soff, unsigned bytes) { return (uint64_t)off + soff + bytes <= rs.n; }
template <int AUX> static inline v4u bld16(rsrc_t rs, unsigned off, unsigned soff) { v4u v = {0u, 0u, 0u, 0u}; if (inr(rs, off, soff, 16)) memcpy(&v, rs.base + off + soff, 16); return v; }
template <int AUX> static inline u2_t bld8(rsrc_t rs, unsigned off, unsigned soff) { u2_t v = {0u, 0u}; if (inr(rs, off, soff, 8)) memcpy(&v, rs.base + off + soff, 8); return v; }
template <int AUX> static inline void bst16(rsrc_t rs, unsigned off, unsigned soff, v4u v) { if (inr(rs, off, soff, 16)) memcpy(rs.base + off + soff, &v, 16); }
template <int AUX> static inline void bst8(rsrc_t rs, unsigned off, unsigned soff, u2_t v) { if (inr(rs, off, soff, 8)) memcpy(rs.base + off + soff, &v, 8); }
#else
template <int AUX> __device__ __forceinline__ v4u bld16(rsrc_t rs, unsigned off, unsigned soff) { return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, (int)soff, AUX); }
template <int AUX> __device__ __forceinline__ u2_t bld8(rsrc_t rs, unsigned off, unsigned soff) { return __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, (int)soff, AUX); }
template <int AUX> __device__ __forceinline__ void bst16(rsrc_t rs, unsigned off, unsigned soff, v4u v) { __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, (int)soff, AUX); }
template <int AUX> __device__ __forceinline__ void bst8(rsrc_t rs, unsigned off, unsigned soff, u2_t v) { __builtin_amdgcn_raw_buffer_store_b64(v, rs, (int)off, (int)soff, AUX); }
#endif

// granules: {32 value bits, epoch} per 8-byte word; a double is two of them in one 16-byte access
template <class T> struct gran_io;
template <> struct gran_io<float> {
    typedef u2_t reg_t;
    static constexpr int BYTES = 8;
    __device__ __forceinline__ static reg_t ld(rsrc_t rs, unsigned off, unsigned soff) { return bld8<17>(rs, off, soff); }
    __device__ __forceinline__ static void st(rsrc_t rs, unsigned off, unsigned soff, float v, unsigned ep) { u2_t g; g.x = __float_as_uint(v); g.y = ep; bst8<17>(rs, off, soff, g); }
    __device__ __forceinline__ static bool ok(reg_t g, unsigned ep) { return g.y == ep; }
    __device__ __forceinline__ static float val(reg_t g) { return __uint_as_float(g.x); }
};
template <> struct gran_io<double> {
    typedef v4u reg_t;
    static constexpr int BYTES = 16;
    __device__ __forceinline__ static reg_t ld(rsrc_t rs, unsigned off, unsigned soff) { return bld16<17>(rs, off, soff); }
    __device__ __forceinline__ static void st(rsrc_t rs, unsigned off, unsigned soff, double v, unsigned ep)
    {
        const unsigned long long u = (unsigned long long)__double_as_longlong(v);
        v4u g; g.x = (unsigned)u; g.y = ep; g.z = (unsigned)(u >> 32); g.w = ep;
        bst16<17>(rs, off, soff, g);
    }
    __device__ __forceinline__ static bool ok(reg_t g, unsigned ep) { return g.y == ep && g.w == ep; }
    __device__ __forceinline__ static double val(reg_t g) { return __longlong_as_double((long long)(((unsigned long long)g.z << 32) | g.x)); }
};

// an access of a wave line: `plain` = the lane's byte offset within a line of the array (SZH_BM_OOB: the lane takes no part), `half` = 1 if the
// lane's line is the one BEFORE the wave line (the upper half-beam runs a line behind), `stride` = bytes per line
struct role_t { unsigned plain, mid; int half; };
__device__ __forceinline__ role_t make_role(bool en, unsigned plain, int half, unsigned stride)
{
    role_t r; r.plain = en ? plain : SZH_BM_OOB; r.half = half; r.mid = en ? plain + (half ? 0u : stride) : SZH_BM_OOB;
    return r;
}

template <class T, bool DEC, bool USEMEAN, bool HASREG>
struct beam {
    typedef shape<T, HASREG> S;
    typedef gran_io<T> GIO;
    typedef typename GIO::reg_t greg_t;
    typedef u2_t cpiece_t;
    static constexpr int PITCH = S::PITCH, RINGB = S::RINGB, SZ = S::SZ, UL = S::UL, EV = S::EV, DV = S::DV, NW = S::NW, LP = LINE * S::PITCH;

    const szh_qargs<T> &a;
    OC_LDS unsigned char *ring, *nring, *kring;
    OC_LDS unsigned *prog;
    int lane, w, h, m, r0, dbg;                                    // dbg (development, timing only -- results become wrong): 1 no waits between the wavefronts, 2 / 4 / 8: no events at position 1 / 2 / 3
    bool has_prev, has_next, zero_face;                            // (uniform)
    unsigned spin_limit; bool timed_out;
    rsrc_t rs_v, rs_c, rs_k, rs_j;                                 // the array (values), the codes, the k- / j-face granules
    unsigned str_v, str_c, str_k, str_j;                           // bytes per line of each
    // per-lane state of the sweep
    T dl[LINE], lup[LINE], prev, Lprev, Bold, Bpold, cur_next, kf_next;
    unsigned tc_next, fl_next, vaddr, cdelta, fdelta, kaddr_h, trash, tstart;     // (vaddr, kaddr_h, trash: byte addresses in the workgroup's LDS window `lds0`)
    OC_LDS unsigned char *lds0;
    unsigned ring_end, nring_lo, pface;                           // first byte behind this ring; the next ring's first byte; where the lane's next face value goes (the virtual slot of its line in the next ring)
    T face_reg;                                                    // the lane's latest last-row result (upper half): handed on once per line
    unsigned m_first, m_vu[LINE], m_push[LINE];                   // per-lane select masks: first lane of a half; virtual cell of the upper half / last row of the upper half at position U
    T caphU[LINE];                                                 // the quantiser's range test per position: half the capacity, -1 where the lane's cell is virtual
    // events: roles and LDS places
    role_t rv[EV], rvs[EV], rc, rcs, rko, rki, rjo, rji;          // value rows in (out: inverse), code rows in (inverse) / out, granules
    unsigned vl[EV], cl, ko_lds, ki_lds, jo_lds, ji_lds;
    v4u gv[DV];                                                    // value rows on their way
    v4u gx[HASREG && !DEC ? DV : 1]; unsigned gf[HASREG ? DV : 1]; // HASREG: the regression points' reconstructions of the same rows (compress), their flag bytes
    rsrc_t rs_x, rs_f; role_t rf[EV]; unsigned fl_lds[EV];
    cpiece_t gc[UL], wqc; v4u wqv[EV]; unsigned wfl, cfl;                             // inverse: code rows on their way; rows on their way out
    greg_t gk[DK], gj[DK];                                         // k-face / j-face granules on their way
    T eb, eb2, rh, caph, radf, mean; int radius; unsigned epoch;
    unsigned lo_it;                                                // LDS offset of wave line `it` in the ring (uniform)
    bool wt_codes;                                                 // the codes are stored write-through (the host launches passes over finished lines while the sweep runs)
    bool pface_fixed;                                              // the lane hands nothing on (its `pface` is its write-only word)
    unsigned pv_prev, pv_next;                                     // the neighbouring wavefronts' progress words as read a step ago

    __device__ __forceinline__ beam(const szh_qargs<T> &args) : a(args) {}

    __device__ __forceinline__ bool give_up(unsigned &spins)
    {
        if (++spins <= spin_limit) return false;
        timed_out = true; spin_limit = 0;
        return true;
    }
    // The inputs of the first `planes` planes are in memory: compress, arrays with regression blocks -- the regression points (reconstructions, flags, codes:
    // k_reg_points); decompress (round 5) -- the codes in natural order and the pre-scattered unpredictable values (k_permute<1>, k_unpred).  They are made slice after slice
    // of block rows on another stream while this sweep runs, as the host's coefficient chains get there (szhip_sz21.inc, "feed"); after each slice a
    // one-thread kernel stores the number of finished planes into a.reg_ready.  Read past this XCD's L2 (system scope); the slices' own stores reached
    // memory when their kernel ended, and this launch has not touched a line of those planes before (planes are whole cache lines: checked on the host).
    // Bounded: a sweep that is not fed gives up (error 2) and the call is repeated with the chains finished first.
    int fed; bool feed_late;
    __device__ __forceinline__ void wait_fed(int planes)
    {
        if (planes > r0) planes = r0;
        if (fed >= planes) return;
        unsigned spins = 0;
#pragma unroll 1
        for (;;) {
            fed = uni((int)__hip_atomic_load(a.reg_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
            if (fed >= planes) break;
            if (++spins > (1u << 15) || spin_limit == 0) { timed_out = true; feed_late = true; spin_limit = 0; fed = r0; break; }     // (~70 ms)
#ifndef SZH_HIPSIM
            __builtin_amdgcn_s_sleep(64);          // (~2 us: a thousand waiting wavefronts ask one word)
#endif
        }
    }
    __device__ __forceinline__ void wait_prog(OC_LDS unsigned *p, int need)
    {
        unsigned spins = 0;
#pragma unroll 1
        for (;;) {
            const int v = uni((int)lds_get<unsigned>((OC_LDS unsigned char *)p, 0));
            if (v >= need || give_up(spins)) break;
            nap1();
        }
    }
    // LDS offset of the line n lines after wave line `it` (0 <= n < RL)
    __device__ __forceinline__ unsigned lo(int n) const { const unsigned x = lo_it + (unsigned)(n * LP); return x >= (unsigned)RINGB ? x - (unsigned)RINGB : x; }

    // offsets of a role's access of wave line X.  Inside the array (EDGE = false: every line any lane asks for exists) the line term is the
    // wavefront-uniform `soff` and the lane part a constant; at the array's first and last lines each lane checks its own line
    template <bool EDGE> __device__ __forceinline__ void place(const role_t &r, int X, unsigned stride, unsigned &off, unsigned &soff) const
    {
        if (!EDGE) { off = r.mid; soff = (unsigned)(X - 1) * stride; return; }
        const int L = X - r.half;
        off = (r.plain != SZH_BM_OOB && (unsigned)L < (unsigned)r0) ? r.plain + (unsigned)L * stride : SZH_BM_OOB;
        soff = 0u;
    }
    template <bool EDGE> __device__ __forceinline__ v4u load_v(int X, int ev) const { unsigned o, so; place<EDGE>(rv[ev], X, str_v, o, so); return bld16<DEC ? 0 : 2>(rs_v, o, so); }
    template <bool EDGE> __device__ __forceinline__ v4u load_x(int X, int ev) const { unsigned o, so; place<EDGE>(rv[ev], X, str_v, o, so); return bld16<0>(rs_x, o, so); }
    template <bool EDGE> __device__ __forceinline__ unsigned load_f(int X, int ev) const
    {
        unsigned o, so; place<EDGE>(rf[ev], X, str_v / (unsigned)SZ, o, so);
#ifdef SZH_HIPSIM
        unsigned v = 0; for (int e = 0; e < S::VPL; ++e) if (inr(rs_f, o + (unsigned)e, so, 1)) v |= (unsigned)(unsigned char)rs_f.base[o + so + (unsigned)e] << (8 * e);
        return v;
#else
        return S::VPL == 4 ? (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_f, (int)o, (int)so, 0) : (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs_f, (int)o, (int)so, 0);
#endif
    }
    // a row piece goes into the ring: HASREG: the regression points' values are their reconstructions (compress; the inverse finds them in the
    // array already), and their flag bytes go next to the values
    __device__ __forceinline__ void put_rows(unsigned at_line, int e, v4u xv, v4u xrv, unsigned fl)
    {
        if (HASREG) {
            if (!DEC) {
                T a_[S::VPL], b_[S::VPL];
                __builtin_memcpy(a_, &xv, 16); __builtin_memcpy(b_, &xrv, 16);
                for (int q = 0; q < S::VPL; ++q) a_[q] = ((fl >> (8 * q)) & 0xffu) ? b_[q] : a_[q];
                __builtin_memcpy(&xv, a_, 16);
            }
            if (S::VPL == 4) lds_put<unsigned>(ring, at_line + fl_lds[e], fl); else lds_put<uint16_t>(ring, at_line + fl_lds[e], (uint16_t)fl);
        }
        lds_put16(ring, at_line + vl[e], xv);
    }
    template <bool EDGE> __device__ __forceinline__ cpiece_t load_c(int X) const { unsigned o, so; place<EDGE>(rc, X, str_c, o, so); return bld8<0>(rs_c, o, so); }
    template <bool EDGE> __device__ __forceinline__ greg_t load_k(int X) const { unsigned o, so; place<EDGE>(rki, X, str_k, o, so); return GIO::ld(rs_k, o, so); }
    template <bool EDGE> __device__ __forceinline__ greg_t load_j(int X) const { unsigned o, so; place<EDGE>(rji, X, str_j, o, so); return GIO::ld(rs_j, o, so); }

    // the granules of wave line X have arrived -- or are asked for again (bounded); their values go into the k-face ring / the virtual slot
    template <bool EDGE, bool KFACE> __device__ __forceinline__ void take(int X, greg_t g, OC_LDS unsigned char *dst, unsigned at_line)
    {
        const role_t &r = KFACE ? rki : rji;
        // (lanes that take no part write to their write-only word: `ki_lds` / `ji_lds` is its whole offset for them)
        const unsigned at = (r.plain != SZH_BM_OOB ? at_line : 0u) + (KFACE ? ki_lds : ji_lds);
        const bool need = r.plain != SZH_BM_OOB && (!EDGE || (unsigned)(X - r.half) < (unsigned)r0);
        if (__all((!need || GIO::ok(g, epoch)) ? 1 : 0) == 0) {
            unsigned spins = 0;
#pragma unroll 1
            do {
                if (give_up(spins)) break;
                nap1();
                g = KFACE ? load_k<true>(X) : load_j<true>(X);
            } while (__all((!need || GIO::ok(g, epoch)) ? 1 : 0) == 0);
        }
        lds_put<T>(dst, at, need ? GIO::val(g) : (T)0);
    }

    // ---- the events of unrolled position U of wave line `it` (LL = it modulo UL, compile time)
    template <int U, int LL, bool EDGE> __device__ __forceinline__ void events(int it)
    {
        if constexpr (U == 1) {
            // finished rows (line it - LAG: every lane has left it) leave the ring: read here, stored one step later
            const unsigned loY = lo(RL - LAG);
            if (!DEC) wqc = lds_get8(ring, loY + cl);
            if (!DEC && HASREG) wfl = lds_get<unsigned>(ring, loY + cfl);
            else for_n<EV>([&](auto E) { constexpr int e = decltype(E)::value; wqv[e] = lds_get16(ring, loY + vl[e]); });
            unsigned o, so;
            place<EDGE>(rko, it - LAG, str_k, o, so);
            GIO::st(rs_k, o, so, lds_get<T>(ring, loY + ko_lds), epoch);
            place<EDGE>(rjo, it - LAG, str_j, o, so);
            GIO::st(rs_j, o, so, lds_get<T>(ring, loY + jo_lds), epoch);
        }
        if constexpr (U == 2) {
            unsigned o, so;
            if (!DEC) {
                if (HASREG) {
                    // the regression points of the row piece keep the codes k_reg_points gave them (the sweep passed their values through and has
                    // zeros there): per code a 16-bit mask from its flag byte
                    constexpr int set = LL % UL;
                    const cpiece_t old = gc[set];
                    const unsigned m0 = ((wfl & 0xffu) ? 0xffffu : 0u) | ((wfl & 0xff00u) ? 0xffff0000u : 0u), m1 = ((wfl & 0xff0000u) ? 0xffffu : 0u) | ((wfl & 0xff000000u) ? 0xffff0000u : 0u);
                    wqc.x = (old.x & m0) | (wqc.x & ~m0); wqc.y = (old.y & m1) | (wqc.y & ~m1);
                    gc[set] = load_c<true>(it - LAG + UL);
                }
                place<EDGE>(rcs, it - LAG, str_c, o, so);
                if (wt_codes) bst8<17>(rs_c, o, so, wqc); else bst8<0>(rs_c, o, so, wqc);      // (written through when the host follows this sweep's progress: see `pub` in run)
            }
            else for_n<EV>([&](auto E) { constexpr int e = decltype(E)::value; place<EDGE>(rvs[e], it - LAG, str_v, o, so); bst16<0>(rs_v, o, so, wqv[e]); });
            // rows that have arrived go in (line it + 1), and the register set that carried them is sent for the rows DV events further on
            const unsigned loX = lo(1);
            for_n<EV>([&](auto E) {
                constexpr int e = decltype(E)::value, set = ((LL + 1) * EV + e) % DV;
                put_rows(loX, e, gv[set], gx[HASREG && !DEC ? set : 0], gf[HASREG ? set : 0]);
                gv[set] = load_v<EDGE>(it + 1 + UL, e);
                if (HASREG && !DEC) gx[HASREG && !DEC ? set : 0] = load_x<EDGE>(it + 1 + UL, e);
                if (HASREG) gf[HASREG ? set : 0] = load_f<EDGE>(it + 1 + UL, e);
            });
            if (zero_face) { const v4u z = {0u, 0u, 0u, 0u}; if (lane < S::HB / 16) lds_put16(ring, loX + (unsigned)lane * 16u, z); }
            if (DEC) { constexpr int set = (LL + 1) % UL; lds_put8(ring, loX + cl, gc[set]); gc[set] = load_c<EDGE>(it + 1 + UL); }
        }
        if constexpr (U == 4) {
            pv_prev = lds_get<unsigned>((OC_LDS unsigned char *)(prog + (has_prev ? w - 1 : w)), 0);
            pv_next = lds_get<unsigned>((OC_LDS unsigned char *)(prog + (has_next ? w + 1 : w)), 0);
        }
        if constexpr (U == 3) {
            constexpr int set = (LL + 1) % DK, kl = (LL + 1) % KRL;
            take<EDGE, true>(it + 1, gk[set], kring, (unsigned)(kl * LINE * 2 * SZ));
            gk[set] = load_k<EDGE>(it + 1 + DK);
            take<EDGE, false>(it + 1, gj[set], ring, lo(1));
            gj[set] = load_j<EDGE>(it + 1 + DK);
        }
    }

    __device__ __forceinline__ static T tabs(T v) { return sizeof(T) == 8 ? (T)__builtin_fabs((double)v) : (T)__builtin_fabsf((float)v); }
    __device__ __forceinline__ static T ttrunc(T v) { return sizeof(T) == 8 ? (T)__builtin_trunc((double)v) : (T)__builtin_truncf((float)v); }
    __device__ __forceinline__ static T tsign(T mag, T from) { return sizeof(T) == 8 ? (T)__builtin_copysign((double)mag, (double)from) : (T)__builtin_copysignf((float)mag, (float)from); }

    // ---- one step: the cell the lane is at
    // One wavefront per SIMD: a dependent VALU instruction issues ~9 cycles after the one it waits for, an independent one after 4 (tools/ubench),
    // and hipcc's scheduler models neither: it emitted the 20 operations of the dependent chain (DPP, the 7-point sum left to right, the
    // quantiser, the bound check) back to back and everything else around them -- 480 cycles a step.  So the step is written in the order it
    // should issue, one chain operation and one or two independent ones per group, and the groups are pinned (SZH_SB: nothing crosses).
    template <int U, int LL, bool EDGE> __device__ __forceinline__ void step(int it)
    {
        const T cur_raw = cur_next, kf = kf_next;
        const unsigned tc_in = tc_next, fl_in = fl_next;
        const T Lraw = shr1(prev);                                       // (i, j, k-1): the left lane's previous result
        const unsigned y = vaddr + (unsigned)PITCH;
        SZH_SB;
        T L = bsel(m_first, kf, Lraw);                                   // (a half's first lane: the k-face of the beam on the left)
        const bool started = !EDGE || (unsigned)(it * LINE + U) >= tstart;      // (at the array's first lines: lanes that have not started hold zeros)
        if (EDGE) L = started ? L : (T)0;
        const T sw = (SZH_BM_X & 16) ? prev : low_to_high(prev);
        SZH_SB;
        const T B = dl[U], Bp = lup[U], C = Bold, Cp = Bpold;
        // [-1] + [-s1] + [-s0] - [-s1-1] - [-s0-1] - [-s0-s1] + [-s0-s1-1], left to right (sz_float.c:7268)
        const T s1 = L + prev;
        const unsigned vnext = y >= ring_end ? y - (unsigned)RINGB : y;
        SZH_SB;
        const T s2 = s1 + B;
        // a virtual cell: the j-face -- from the ring (lower half), from the lower half's previous result (upper half)
        const T cur = bsel(m_vu[U], sw, cur_raw);
        SZH_SB;
        const T s3 = s2 - Lprev;
        if (!(SZH_BM_X & 64)) cur_next = lds_get<T>(lds0, vnext);        // what the NEXT step needs from the rings is requested now
        if (DEC) tc_next = lds_get<uint16_t>(lds0, vnext + cdelta);
        if (HASREG) fl_next = lds_get<uint8_t>(lds0, vnext + fdelta);
        SZH_SB;
        const T s4 = s3 - Bp;
        {   // the k-face value of the next step's cell of lane 0 (every lane reads; only the halves' first lanes use it)
            constexpr int Un = (U + 1) % LINE, kl = (U + 1 == LINE ? LL + 1 : LL) % KRL;
            if (!(SZH_BM_X & 8)) kf_next = lds_get<T>(lds0, (unsigned)(kl * LINE * 2 * SZ + Un * 2 * SZ) + kaddr_h);
        }
        SZH_SB;
        const T s5 = s4 - C;
        SZH_SB;
        const T pred = s5 + Cp;
        SZH_SB;
        T rec;
        if (!DEC) {
            // the quantiser of szh_rb::rb_quant (sz_float.c:7270-7287 with a shorter dependency chain, bit for bit the same results); a virtual
            // cell fails the range test (its limit is -1) and hands its value on unchanged
            const T diff = cur - pred;
            SZH_SB;
            const T hq0 = tabs(diff) * rh;
            const unsigned caddr = vaddr + cdelta;
            SZH_SB;
            const T hq = hq0 + (T)0.5;
            SZH_SB;
            const T tq = ttrunc(hq);
            mask_t okm = lane_mask(hq < caphU[U]);
            if (HASREG) okm &= lane_mask(fl_in == 0u);
            SZH_SB;
            const T ts = tsign(tq, diff);
            SZH_SB;
            const T m1 = ts * eb2;
            const T cf = radf + ts;
            SZH_SB;
            const T m2 = m1 + (T)0;
            int code = (int)cf;
            SZH_SB;
            const T rcn = pred + m2;
            SZH_SB;
            const T err = cur - rcn;
            SZH_SB;
            okm &= lane_mask(!(tabs(err) > eb));
            const bool ok = in_mask(okm);
            code = ok ? code : 0;
            rec = ok ? rcn : cur;
            if (USEMEAN) {
                if (code != 0 && code <= radius) code -= 1;                                  // sz_float.c:6944
                mask_t nm = lane_mask(tabs(cur - mean) <= eb) & lane_mask(caphU[U] > (T)0);  // sz_float.c:6929
                if (HASREG) nm &= lane_mask(fl_in == 0u);
                if (in_mask(nm)) { code = radius; rec = mean; }
            }
            if (EDGE) rec = started ? rec : (T)0;
            if (SZH_BM_X & 32) rec = pred + cur;
            if (!(SZH_BM_X & 2)) lds_put<uint16_t>(lds0, caddr, (uint16_t)code);
            if (!(SZH_BM_X & 4)) lds_put<T>(lds0, vaddr, rec);
        } else {
            int cq = (int)tc_in;
            bool is_mean = false;
            if (USEMEAN) { is_mean = cq == radius; if (cq != 0 && cq < radius) cq += 1; }     // szd_float.c:3784
            const T mq = (T)(cq - radius) * eb2;
            mask_t um = lane_mask(tc_in != 0u) & lane_mask(caphU[U] > (T)0);
            if (HASREG) um &= lane_mask(fl_in == 0u);
            SZH_SB;
            T r = pred + mq;                                                                  // = pred + 2 (c - radius) eb (szd_float.c:5786)
            if (USEMEAN && is_mean) r = mean;
            rec = in_mask(um) ? r : cur;                                                       // zero code: the pre-scattered value
            if (EDGE) rec = started ? rec : (T)0;
            lds_put<T>(lds0, vaddr, rec);
        }
        // the last row of the upper half is the j-face of the wavefront above: kept where the lane made it, handed on at the end of the line
        if (!(SZH_BM_X & 1)) face_reg = bsel(m_push[U], rec, face_reg);
        dl[U] = rec; lup[U] = L;
        Bold = B; Bpold = Bp;
        Lprev = L; prev = rec;
        vaddr = vnext;
    }

    template <int LL, bool EDGE> __device__ __forceinline__ void line(int it)
    {
        // the wavefront below (in j) must be far enough ahead for the virtual cells read during this line, the one above not too far behind
        // (their progress words were read during the last step of the line before: no LDS round trip here unless one of them is late)
        if (!(SZH_DEV && (dbg & 1))) {
        if (has_prev && uni((int)pv_prev) < it + 3) wait_prog(prog + (w - 1), it + 3);
        if (has_next && uni((int)pv_next) < it - (RL - 2)) wait_prog(prog + (w + 1), it - (RL - 2));
        }
        for_n<LINE>([&](auto UU) {
            constexpr int U = decltype(UU)::value;
            wave_sync();
            if (!(SZH_DEV && ((dbg >> U) & 1)) || U == 0) events<U, LL, EDGE>(it);
            order();
            step<U, LL, EDGE>(it);
            order();
        });
        // the faces of this line go to the wavefront above: each lane's latest last-row result into the slot of that line's virtual cell in ITS
        // ring (lanes of the lower half, and the workgroup's last wavefront: into their write-only word).  One write per line, not per step:
        // the wavefront above waits for whole lines anyway (it + 3 below), so nothing arrives later than it is looked for
        if (!(SZH_BM_X & 1)) lds_put<T>(lds0, pface, face_reg);
        { const unsigned f = pface + (unsigned)LP; pface = pface_fixed ? pface : (f >= nring_lo + (unsigned)RINGB ? f - (unsigned)RINGB : f); }
        lds_put<unsigned>((OC_LDS unsigned char *)(prog + w), 0, (unsigned)(it + 1));
        lo_it = lo(1);
    }
    template <bool EDGE> __device__ __forceinline__ void block(int it0)
    {
        for_n<UL>([&](auto L_) { constexpr int LL = decltype(L_)::value; line<LL, EDGE>(it0 + LL); });
    }

    __device__ __forceinline__ void run(int kb, int jg, OC_LDS unsigned char *rings, OC_LDS unsigned *prog_)
    {
        const szh_geom3 &G = a.G;
        prog = prog_; dbg = a.dbg;
        r0 = G.g0.count;
        const int r1 = G.g1.count, r2 = G.g2.count;
        lane = (int)(threadIdx.x & 63u); w = uni((int)(threadIdx.x >> 6)); h = lane >> 5; m = lane & 31;
        ring = rings + w * RINGB; nring = rings + (w + 1 < WPG ? w + 1 : w) * RINGB; kring = rings + (WPG * RINGB + WPG * 64 * 8) + w * S::KRB;
        const grid_t gr = make_grid(G);
        const int k0 = kb * HL, jw0 = jg * JG + w * JW;
        has_prev = w > 0; has_next = w + 1 < WPG;
        const bool jf_in = w == 0 && jg > 0, jf_out = w == WPG - 1 && jg + 1 < gr.nJG, kf_in = kb > 0, kf_out = kb + 1 < gr.nKB;
        zero_face = w == 0 && jg == 0;
        spin_limit = 1u << 22; timed_out = false;
        eb = a.eb; eb2 = eb + eb; rh = a.recip * (T)0.5; caph = (T)(a.cap - 2) * (T)0.5; radf = (T)a.radius; mean = a.mean; radius = a.radius; epoch = a.epoch;
        const int64_t wg = (int64_t)kb * gr.nJG + jg;
        const uint64_t nbytes = (uint64_t)G.n * SZ;
        rs_v = make_rsrc(DEC ? (const void *)a.out : (const void *)a.data, (unsigned)nbytes);
        rs_c = make_rsrc(a.codes, (unsigned)((uint64_t)G.n * 2));
        rs_x = make_rsrc(HASREG && !DEC ? (const void *)a.xr : (const void *)a.codes, HASREG && !DEC ? (unsigned)nbytes : 0u);
        rs_f = make_rsrc(HASREG ? (const void *)a.ptflags : (const void *)a.codes, HASREG ? (unsigned)G.n : 0u);
        str_v = (unsigned)(G.d0 * SZ); str_c = (unsigned)(G.d0 * 2); str_k = (unsigned)(LINE * GIO::BYTES); str_j = (unsigned)(HL * GIO::BYTES);
        {   // value rows: 64 lanes x 16 B = RPE row pieces of HB bytes: RH rows of each half
            const int r = lane / S::LPR, p = lane - r * S::LPR, vh = r / S::RH;
            for (int e = 0; e < EV; ++e) {
                const int rr = r % S::RH + S::RH * e, j = jw0 + C1 * vh + rr, kk = k0 + p * S::VPL;
                const unsigned off = (unsigned)(((int64_t)j * G.d1 + kk) * SZ);
                const bool in = j < r1 && kk < r2;
                rv[e] = make_role(true, in ? off : 0u, vh, str_v);           // (rows outside the array: any readable place -- nobody looks at them)
                rvs[e] = make_role(in, off, vh, str_v);
                rf[e] = make_role(true, in ? off / (unsigned)SZ : 0u, vh, str_v / (unsigned)SZ);
                vl[e] = (unsigned)((1 + rr) * PITCH + vh * S::HB + p * 16);
                fl_lds[e] = (unsigned)((1 + rr) * PITCH + S::FOFF + vh * HL + p * S::VPL);
            }
        }
        {   // code rows: 64 lanes x 8 B = 8 row pieces of 64 bytes
            const int r = lane / 8, p = lane - r * 8, rr = r % C1, ch = r / C1, j = jw0 + C1 * ch + rr, kk = k0 + 4 * p;
            const unsigned off = (unsigned)(((int64_t)j * G.d1 + kk) * 2);
            const bool in = j < r1 && kk < r2;
            rc = make_role(true, in ? off : 0u, ch, str_c);
            rcs = make_role(in, off, ch, str_c);
            cl = (unsigned)((1 + rr) * PITCH + S::VB + ch * 64 + p * 8);
            cfl = (unsigned)((1 + rr) * PITCH + S::FOFF + ch * HL + p * 4);
        }
        {   // k-face granules: lane e < 10: cell e % 5 of half e / 5 of a wave line; [workgroup][8 half-beams][LINE r0 cells]
            const bool en = lane < 2 * LINE;
            const int khh = en ? lane / LINE : 0, kuu = en ? lane % LINE : 0;
            const uint64_t cells = (uint64_t)LINE * r0;
            const int64_t wgl = (int64_t)(kb > 0 ? kb - 1 : 0) * gr.nJG + jg;
            rs_k = make_rsrc(a.faceI, (unsigned)(kface_words<T>(G) * 8));
            rko = make_role(en && kf_out, (unsigned)((((uint64_t)(wg * 8 + 2 * w + khh)) * cells + kuu) * GIO::BYTES), khh, str_k);
            rki = make_role(en && kf_in, (unsigned)((((uint64_t)(wgl * 8 + 2 * w + khh)) * cells + kuu) * GIO::BYTES), khh, str_k);
            ko_lds = (unsigned)(kuu * PITCH + khh * S::HB + (HL - 1) * SZ);
            ki_lds = (en && kf_in) ? (unsigned)(kuu * 2 * SZ + khh * SZ) : (unsigned)(KRL * LINE * 2 * SZ);        // (other lanes: a write-only word behind the ring)
        }
        {   // j-face granules: lane e < 32: column e of the row; [workgroup][r0 lines][32]; out: the upper half's last row (a line behind), in: the lower half's virtual cell
            const bool en = lane < HL;
            const int64_t wgl = (int64_t)kb * gr.nJG + (jg > 0 ? jg - 1 : 0);
            rs_j = make_rsrc(a.faceJ, (unsigned)(jface_words<T>(G) * 8));
            rjo = make_role(en && jf_out, (unsigned)(((uint64_t)wg * r0 * HL + (unsigned)(lane & 31)) * GIO::BYTES), 1, str_j);
            rji = make_role(en && jf_in, (unsigned)(((uint64_t)wgl * r0 * HL + (unsigned)(lane & 31)) * GIO::BYTES), 0, str_j);
            jo_lds = (unsigned)(C1 * PITCH + S::HB + (lane & 31) * SZ);
            ji_lds = (en && jf_in) ? (unsigned)((lane & 31) * SZ) : (unsigned)(RINGB * (WPG - w) + (w * 64 + lane) * 8);     // (the others: their write-only slot behind the rings)
        }
        for (int u = 0; u < LINE; ++u) { dl[u] = 0; lup[u] = 0; }
        prev = 0; Lprev = 0; Bold = 0; Bpold = 0;
        lds0 = rings;                                   // (the addresses below are relative to the rings' first byte; the write-only words and the k-face rings follow the rings in one array)
        const unsigned ring_lo = (unsigned)(w * RINGB);
        ring_end = ring_lo + (unsigned)RINGB; nring_lo = ring_end;
        {   // after wave line `it` the lane's latest finished last-row cell belongs to beam line it - 1 - ceil(m / LINE): the slot of THAT line's
            // virtual cell in the next ring (lines before the array's first: slots nobody has looked at yet)
            const int L0 = -1 - (m + LINE - 1) / LINE;
            pface_fixed = !(h == 1 && has_next);
            pface = pface_fixed ? (unsigned)(WPG * RINGB + (w * 64 + lane) * 8)
                                : ring_end + (unsigned)((((L0 % RL) + RL) % RL) * LP + m * SZ);
            face_reg = 0;
        }
        vaddr = ring_lo + (unsigned)(((RS - m) % RS) * PITCH + h * S::HB + m * SZ);
        m_first = m == 0 ? 0xffffffffu : 0u;
        for (int u = 0; u < LINE; ++u) {
            const bool virt = ((u - m) % LINE + LINE) % LINE == 0, last = ((u - m) % LINE + LINE) % LINE == C1;
            m_vu[u] = (virt && h == 1) ? 0xffffffffu : 0u;
            m_push[u] = (last && h == 1) ? 0xffffffffu : 0u;
            caphU[u] = virt ? (T)-1 : caph;
            hide(m_vu[u]); hide(m_push[u]); hide(caphU[u]);        // (kept in registers: hipcc otherwise rebuilds them from the lane number as scalar lane masks, ~20 SGPRs and their spills)
        }
        hide(m_first);
        cdelta = (unsigned)(S::VB + h * 64 + m * 2) - (unsigned)(h * S::HB + m * SZ);
        fdelta = (unsigned)(S::FOFF + lane) - (unsigned)(h * S::HB + m * SZ);
        kaddr_h = (unsigned)(WPG * RINGB + WPG * 64 * 8 + w * S::KRB + h * SZ);
        trash = (unsigned)(WPG * RINGB + (w * 64 + lane) * 8);       // (the lane's write-only word behind the last ring)
        tstart = (unsigned)(m + LINE * h);
        lo_it = 0u;
        // the k-face ring reads zeros where nothing arrives (no beam on the left)
        for (int e = lane; e < S::KRB / 4; e += 64) lds_put<unsigned>(kring, (unsigned)e * 4u, 0u);
        // (arrays with regression blocks whose points arrive WHILE the sweep runs -- a.reg_ready, see wait_fed: nothing of a plane is asked for before it is there)
        fed = 0; feed_late = false;
        if (a.reg_ready) wait_fed(UL + 2);
        // ---- prologue: the first lines' rows are requested; line 0 goes into the ring
        {
            v4u first[EV], firstx[EV]; unsigned firstf[EV];
            for_n<EV>([&](auto E) {
                constexpr int e = decltype(E)::value;
                first[e] = load_v<true>(0, e);
                firstx[e] = (HASREG && !DEC) ? load_x<true>(0, e) : first[e];
                firstf[e] = HASREG ? load_f<true>(0, e) : 0u;
            });
            for_n<UL>([&](auto L_) { constexpr int LL = decltype(L_)::value; for_n<EV>([&](auto E) {
                constexpr int e = decltype(E)::value, set = ((LL + 1) * EV + e) % DV;
                gv[set] = load_v<true>(LL + 1, e);
                if (HASREG && !DEC) gx[HASREG && !DEC ? set : 0] = load_x<true>(LL + 1, e);
                if (HASREG) gf[HASREG ? set : 0] = load_f<true>(LL + 1, e);
            }); });
            for_n<EV>([&](auto E) { constexpr int e = decltype(E)::value; put_rows(0u, e, first[e], firstx[e], firstf[e]); });
            if (zero_face) { const v4u z = {0u, 0u, 0u, 0u}; if (lane < S::HB / 16) lds_put16(ring, (unsigned)lane * 16u, z); }
            if (!DEC && HASREG) for_n<UL>([&](auto L_) { constexpr int LL = decltype(L_)::value; gc[LL] = load_c<true>(LL - LAG); });
            if (DEC) {
                const cpiece_t c0 = load_c<true>(0);
                for_n<UL>([&](auto L_) { constexpr int LL = decltype(L_)::value; gc[(LL + 1) % UL] = load_c<true>(LL + 1); });
                lds_put8(ring, cl, c0);
            }
            const greg_t k0g = load_k<true>(0), j0g = load_j<true>(0);
            for_n<DK>([&](auto L_) { constexpr int LL = decltype(L_)::value; gk[(LL + 1) % DK] = load_k<true>(LL + 1); gj[(LL + 1) % DK] = load_j<true>(LL + 1); });
            take<true, true>(0, k0g, kring, 0u);
            take<true, false>(0, j0g, ring, 0u);
        }
        if (has_prev) wait_prog(prog + (w - 1), 3);
        pv_prev = 0u; pv_next = 0u;
        order();
        wave_sync();
        cur_next = lds_get<T>(lds0, vaddr);
        tc_next = DEC ? (unsigned)lds_get<uint16_t>(lds0, vaddr + cdelta) : 0u;
        fl_next = HASREG ? (unsigned)lds_get<uint8_t>(lds0, vaddr + fdelta) : 0u;
        kf_next = lds_get<T>(lds0, kaddr_h);
        { const v4u z = {0u, 0u, 0u, 0u}; for (int e = 0; e < EV; ++e) wqv[e] = z; wqc.x = 0u; wqc.y = 0u; wfl = 0u; }
        // ---- the lines: every lane has left line it - LAG when lane 0 enters line it.  Blocks of UL lines; the ones in which every line any
        // lane asks for or stores exists take the variant without per-lane line checks
        const int NWL = r0 + 1 + LAG, nblk = (NWL + UL - 1) / UL;
        wt_codes = !DEC && a.tile_done != nullptr;
        unsigned *const pub = (!DEC && a.tile_done) ? a.tile_done + (((int64_t)kb * gr.nJG + jg) * WPG + w) : nullptr;     // (uniform)
        const int PUBB = (a.pub_lines > 0 ? a.pub_lines : 32) / UL > 0 ? (a.pub_lines > 0 ? a.pub_lines : 32) / UL : 1;     // blocks between two words (a word every 32 lines by default)
#pragma unroll 1
        for (int b = 0; b < nblk; ++b) {
            const int it0 = b * UL;
            if (a.reg_ready) wait_fed(it0 + 2 * UL + 1);          // (the block's lines ask for rows up to wave line it0 + 2 UL)
#ifdef SZH_BM_ISA_MID_ONLY          /* (ISA inspection of the steady-state block only: wrong results) */
            block<false>(it0);
#else
            if (it0 >= LAG + 1 && it0 <= r0 - 1 - 2 * UL) block<false>(it0); else block<true>(it0);
#endif
            // the host starts the entropy stage's passes over lines every wavefront has passed (szhip.hip): how many of THIS wavefront's lines have
            // their codes in memory -- a release at system scope (this XCD's L2 is written back first), the launch's epoch beside the count
            if (pub && ((b + 1) % PUBB == 0 || b + 1 == nblk)) {
                int rows = b + 1 == nblk ? r0 : it0 + UL - 1 - LAG;
                rows = rows < 0 ? 0 : (rows > r0 ? r0 : rows);
                if (b + 1 == nblk || rows > 0) {
                    // the codes went out write-through (sc0 sc1): once this wavefront's memory counter is empty they are in memory, and the word may follow.
                    // (A release at system scope -- which writes the XCD's L2 back -- cost ~35 us a time here with every CU storing codes: 16 words a
                    // wavefront took the sweep from 1.05 to 1.6 ms at 512^3.)
#ifndef SZH_HIPSIM
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                    __hip_atomic_store(pub, ((epoch & 0xfffu) << 20) | (unsigned)rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        lds_put<unsigned>((OC_LDS unsigned char *)(prog + w), 0, (unsigned)SZH_BM_INF);
        if (timed_out) st_err(a.err, feed_late ? 2u : 1u);
    }
};
} // namespace szh_bm

// The points of the regression blocks (sz_float.c:7153-7252; inverse szd_float.c:5786-5838): prediction a ii + b jj + c kk + d from the DECODED
// coefficients, no neighbour involved -- all blocks at once.
//   MODE 0  compress, before the sweep: reconstructions -> vals (the sweep's neighbours), flags -> 1, codes -> codes (natural order; the sweep
//           keeps them where the flag is set when it stores a row of codes)
//   MODE 1  (the codes alone; not used any more)
//   MODE 2  decompress, before the sweep: values -> vals (= the output array) where the code is not zero (zero: the pre-scattered value stays), flags -> 1
template <class T, int MODE>
__global__ __launch_bounds__(256) void k_reg_points(szh_geom3 G, const uint8_t *__restrict__ blk_lor, const T *__restrict__ coef, int64_t cstride, const T *__restrict__ data,
                                                    T *__restrict__ vals, uint16_t *__restrict__ codes, uint8_t *__restrict__ flags, T eb, T recip, int cap, int radius, int b0_first)
{
    // a workgroup per block column (b0, b1): its threads stand side by side along the contiguous dimension, so that a row of the column's
    // blocks is read and written as whole lines (a wavefront per block touched 24-byte pieces of 2 KB-strided rows: 0.34 ms at 512^3 against
    // 0.1 here); a thread walks the s0 x s1 cross-section of ITS block at its position, if that block is a regression block
    const int b0r = (int)(blockIdx.x / (unsigned)G.g1.num), b1 = (int)(blockIdx.x - (unsigned)b0r * (unsigned)G.g1.num), b0 = b0r + b0_first;     // (a slice of block rows: b0_first)
    const int i0 = szh_blk_start(G.g0, b0), j0 = szh_blk_start(G.g1, b1), s0 = szh_blk_size(G.g0, b0), s1 = szh_blk_size(G.g1, b1);
    const int64_t bcol = ((int64_t)b0 * G.g1.num + b1) * G.g2.num;
    for (int k = (int)threadIdx.x; k < G.g2.count; k += (int)blockDim.x) {
        const int b2 = szh_blk_of(G.g2, k), kk = k - szh_blk_start(G.g2, b2);
        const int64_t b = bcol + b2;
        if (blk_lor[b] != 0) continue;
        const T ca = coef[b], cb = coef[cstride + b], cc = coef[2 * cstride + b], cd = coef[3 * cstride + b];
        for (int ii = 0; ii < s0; ++ii)
            for (int jj = 0; jj < s1; ++jj) {
                const int64_t idx = (int64_t)(i0 + ii) * G.d0 + (int64_t)(j0 + jj) * G.d1 + k;
                const T pred = ca * (T)ii + cb * (T)jj + cc * (T)kk + cd;     // sz_float.c:7165, left to right
                if (MODE == 2) {
                    const int c = (int)codes[idx];
                    if (c != 0) vals[idx] = pred + (T)(2 * (c - radius)) * eb;   // szd_float.c:5831
                    flags[idx] = 1;
                } else {
                    T rc;
                    const int c = szh_quant_sel<T>(data[idx], pred, eb, recip, cap, radius, &rc);   // capacity: the full interval count (sz_float.c:7170)
                    if (MODE == 0) { vals[idx] = rc; flags[idx] = 1; }
                    codes[idx] = (uint16_t)c;
                }
            }
    }
}

// (the feed's progress word: stream order puts it behind the slice's k_reg_points)
__global__ void k_store_u32(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// a.nI x a.nJ = the beam grid (k-beams x j-groups); a.faceI / a.faceJ = the k- / j-face granules
template <class T, bool DEC, bool USEMEAN, bool HASREG>
__global__ __launch_bounds__(szh_bm::WPG * 64, 1) void k_beam(szh_qargs<T> a)
{
    using namespace szh_bm;
    typedef shape<T, HASREG> S;
    __shared__ __attribute__((aligned(16))) unsigned char rings[WPG * S::RINGB + WPG * 64 * 8 + WPG * S::KRB];     // rings, write-only words of lanes without a face value, k-face rings
    __shared__ unsigned prog[WPG + 2];
    __shared__ unsigned tk_s;
    const unsigned ntiles = (unsigned)(a.nI * a.nJ);
    for (unsigned itile = 0;; ++itile) {
        __syncthreads();
        if (threadIdx.x < WPG + 2) prog[threadIdx.x] = 0;
        if (threadIdx.x == 0) {
            const unsigned t = a.ticket_mode ? blockIdx.x + itile * gridDim.x : atomicAdd(a.ticket, 1u);
            tk_s = t < ntiles ? szh_pencil_order_at(a.nI, a.nJ, t) : 0xffffffffu;
        }
        __syncthreads();
        const unsigned ij = (unsigned)szh_bm::uni((int)tk_s);
        if (ij == 0xffffffffu) break;
        beam<T, DEC, USEMEAN, HASREG> s(a);
        s.run((int)(ij >> 16), (int)(ij & 0xffffu), (OC_LDS unsigned char *)rings, (OC_LDS unsigned *)prog);
    }
}
