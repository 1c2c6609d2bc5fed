// szh_beam.h -- the predict+quantise (and inverse) sweep of the SZ 2.1 path: the "beam" mapping (round 5), rewritten lean in round 6.
//
// Arithmetic: 7-point Lorenzo from RECONSTRUCTED neighbours, sz/src/sz_float.c:7253-7353, with the mean shortcut :6914-7030; inverse
// sz/src/szd_float.c:3483-5866 -- bit for bit the reference's codes and values.
//
// Mapping (unchanged since round 5):
//   * a lane owns a COLUMN: one position k of the contiguous dimension, C1 = 4 consecutive rows j, ALL planes i, and walks its cells
//     (i, j) in row-major order, one cell per step; lane m runs m steps behind lane 0.  32 such lanes are a HALF-BEAM (32 k x 4 j x r0 i),
//     a wavefront is two half-beams side by side in j (the upper one a whole line behind), a workgroup four wavefronts side by side in j:
//     a tile of 32 k x 32 j x r0.  No barrier inside the sweep.
//       - (i, j, k-1) is what lane m - 1 produced one step ago: ONE DPP `wave_shr:1` per step (lane 0 of a half: the k-face of the tile on
//         its left, read from a small LDS ring);
//       - (i, j-1, k) is the lane's own previous result; (i-1, j, k) what it produced LINE = C1 + 1 steps ago: a DELAY LINE of LINE registers
//         indexed by the step number modulo LINE, a compile-time index in the unrolled line; the same for the left lane's values;
//       - a line (i, :) takes LINE steps: its first is the VIRTUAL cell j = j0 - 1, whose "result" is the j-face of the half-beam below it:
//         the upper half of a wavefront takes it from the lower half's register (v_permlane32_swap), the lower half of wavefront w + 1 finds
//         it in its own LDS ring, where wavefront w's lanes put it; across workgroups it travels through HBM granules.
//         With that, ONE predictor expression -- the reference's 7-point sum in its order -- serves every cell;
//   * regression blocks (sz_float.c:7153-7252): their points do not depend on any neighbour, so a separate, fully parallel pass
//     (k_reg_points) quantises them; this sweep only needs their RECONSTRUCTIONS as neighbours (HASREG: a flag per cell in the ring).
//
// What round 6 changed -- a wavefront alone on its SIMD issues ONE instruction every ~4 cycles, whatever the instruction is, and a step's
// dependent chain (DPP, 7-point sum, quantiser, bound check: 20 operations) takes ~150 cycles = 33 - 36 issue slots (tools/ubench/ub_step.hip:
// the chain alone 60 ns, a lean step of 33 instructions 76 ns).  Round 5's kernel issued 84 instructions per step (43 in the step, 41 in the
// events around it: scalar address arithmetic, ring wrap-arounds, progress words) and took 135 - 180 ns.  So:
//   * the ring holds RL = 9 lines and the loop is unrolled over exactly those 9 lines: every LDS address of a step or an event is
//     `lane constant + immediate`.  A lane's slot at unrolled step u is (u - m) modulo 45: two lane constants (before / after the lane's
//     wrap) selected by one compare per step (none for u >= 32); DS address arithmetic is modulo 2^32 (tools/ubench/ub_dswrap.hip), so the
//     "after" constant may lie below the ring;
//   * codes take 4 bytes per lane in the ring (float: the lane's code lies a CONSTANT away from its value), packed when a row leaves;
//   * per-stream scalar offsets advance by one add per line; the quantiser's last product is one exact v_fma (x * y + 0);
//   * the j-face between tiles travels SKEWED -- after each wave line every lane of the last row stores its LATEST finished value (the same
//     thing a wavefront hands to the one above it through LDS) instead of whole rows once all 32 lanes have passed: 3 lines + the memory
//     round trip per tile hop instead of LAG + 1 + the round trip;
//   * the lower face of the array is zeroed once (a virtual cell writes back what it read).
//
// Covers: 3-D arrays with r2 a multiple of 4, float / double, compress / decompress, mean shortcut, regression blocks.
#pragma once
// (included by szhip_kernels.h after szh_ribbon.h and szh_ompcol.h, whose helpers it uses)

#ifndef SZH_DEV
#define SZH_DEV 0
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
typedef szh_io::v4u v4u;

#ifndef SZH_BM_DK
#define SZH_BM_DK 1
#endif
#ifndef SZH_BM_PD
#define SZH_BM_PD 3        /* lines a wavefront asks for its rows ahead, see beam::PD (a divisor of UL) */
#endif
constexpr int C1 = 4, LINE = C1 + 1, HL = 32, WPG = 4, RL = 9, RS = RL * LINE, UL = RL, KRL = 3, DK = SZH_BM_DK;
constexpr int JW = 2 * C1, JG = JW * WPG;          // rows j per wavefront / per workgroup
constexpr int LAG = 7;                              // lines after which every lane has left a line (31 steps of skew + the upper half's line)
#define SZH_BM_INF (1 << 30)
#ifndef SZH_BM_X
#define SZH_BM_X 0     /* tools/ubench/ub_beam.hip only (timing, results wrong): 1 plain code stores, 2 no code stores, 4 no granule stores, 8 no progress checks, 16 no k-face read, 32 no value read, 64 no ring writes, 128 no events, 256 no swap, 512 no slot select, 1024 no face select, 2048 nothing at the end of a line, 4096 no progress reads */
#endif
static_assert(UL % KRL == 0 && UL % DK == 0 && RL >= LAG + 2, "ring geometry");

template <class T> struct shape {
    static constexpr int SZ = (int)sizeof(T), VPL = 16 / SZ, LPR = HL / VPL, RPE = 64 / LPR, RH = RPE / 2, EV = C1 / RH;
    static constexpr int HB = HL * SZ, VB = 2 * HB, CBW = 64 * 4, PITCH = VB + CBW, LP = LINE * PITCH, RINGB = RS * PITCH;
    static constexpr int KHS = 8 * SZ, KLS = 2 * KHS, KRB = KRL * KLS;   // k-face ring of a wavefront: per line and half 8 slots (LINE of them used: the line's cells), read 16 bytes at a time
    // LDS of a workgroup: the rings | a write-only word per lane (+ slack: a lane that takes no part in an event writes there, line term included) | the k-face rings
    // | the wavefronts' progress words |
    static constexpr int TRASH0 = WPG * RINGB, PROG0 = TRASH0 + WPG * 64 * 8 + 512, KR0 = PROG0 + 64, LDSB = KR0 + WPG * KRB;
    static constexpr bool CSEL = SZ != 4;           // the lane's code is NOT a constant away from its value: a second pair of lane constants
    static_assert(PITCH % 128 == 0 && EV >= 1, "ring geometry");
};

// compile-time lane masks (lane = 32 h + m)
constexpr mask_t FIRSTCOL = 1ull | (1ull << 32), UPPER = 0xffffffff00000000ull;
template <int U> constexpr mask_t virt_upper_mask() { mask_t x = 0; for (int l = 32; l < 64; ++l) if ((((l & 31) - U) % LINE + LINE) % LINE == 0) x |= 1ull << l; return x; }

#ifdef SZH_HIPSIM
struct u2_t { unsigned x, y; };
#else
typedef unsigned int u2_t __attribute__((ext_vector_type(2)));
#endif
__device__ __forceinline__ u2_t lds_get8(OC_LDS unsigned char *base, unsigned off) { return *(OC_LDS u2_t *)(base + off); }
__device__ __forceinline__ void lds_put8(OC_LDS unsigned char *base, unsigned off, u2_t v) { *(OC_LDS u2_t *)(base + off) = v; }

#ifdef SZH_HIPSIM
template <class T> static inline T shr1(T v) { return __shfl_up(v, 1, 64); }
template <class T> static inline T low_to_high(T v, T) { const int l = (int)(threadIdx.x & 63); return __shfl(v, l >= 32 ? l - 32 : l, 64); }
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
// lanes 32 + m receive the value of lane m (v_permlane32_swap: lanes 32..63 of the first operand are swapped with lanes 0..31 of the second;
// the first operand is a register whose contents nobody needs any more -- no copy of it, one of `v`)
__device__ __forceinline__ unsigned low_to_high_u(unsigned v, unsigned junk) { const auto r = __builtin_amdgcn_permlane32_swap(junk, v, false, false); return r[0]; }
__device__ __forceinline__ float low_to_high(float v, float junk) { return __uint_as_float(low_to_high_u(__float_as_uint(v), __float_as_uint(junk))); }
__device__ __forceinline__ double low_to_high(double v, double junk)
{
    const unsigned long long s = (unsigned long long)__double_as_longlong(v), j = (unsigned long long)__double_as_longlong(junk);
    const unsigned lo = low_to_high_u((unsigned)s, (unsigned)j), hi = low_to_high_u((unsigned)(s >> 32), (unsigned)(j >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void nap1() { __builtin_amdgcn_s_sleep(1); }
#endif
// the same in two parts, so that the copy the swap consumes is made early in a step and the swap itself sits among independent instructions (back
// to back they need wait states: two s_nop per step).  `copy_for_swap` must be issued at least two VALU instructions before `swap_into`.
#ifdef SZH_HIPSIM
template <class T> static inline T copy_for_swap(T v) { return v; }
template <class T> static inline T swap_into(T copy, T) { const int l = (int)(threadIdx.x & 63); return __shfl(copy, l >= 32 ? l - 32 : l, 64); }
static inline bool all_lanes(bool p) { return __all(p ? 1 : 0) != 0; }
#else
__device__ __forceinline__ float copy_for_swap(float v) { float t; asm("v_mov_b32 %0, %1" : "=v"(t) : "v"(v)); return t; }
__device__ __forceinline__ float swap_into(float copy, float junk) { asm("v_permlane32_swap_b32 %0, %1" : "+v"(junk), "+v"(copy)); return junk; }
__device__ __forceinline__ double copy_for_swap(double v) { double t; asm("v_mov_b32 %L0, %L1\n\tv_mov_b32 %H0, %H1" : "=&v"(t) : "v"(v)); return t; }
__device__ __forceinline__ double swap_into(double copy, double junk) { asm("v_permlane32_swap_b32 %L0, %L1\n\tv_permlane32_swap_b32 %H0, %H1" : "+v"(junk), "+v"(copy)); return junk; }
__device__ __forceinline__ bool all_lanes(bool p) { return __builtin_amdgcn_ballot_w64(p) == __builtin_amdgcn_ballot_w64(true); }
#endif

__device__ __forceinline__ void st_err(unsigned *p, unsigned v) { atomicMax(p, v); }      // (2 = the regression points did not arrive, 1 = a face did not: the larger one stays)

// geometry of a launch
struct grid_t { int nKB, nJG; };
SZH_HD grid_t make_grid(const szh_geom3 &G) { grid_t g; g.nKB = (G.g2.count + HL - 1) / HL; g.nJG = (G.g1.count + JG - 1) / JG; return g; }
SZH_HD int wave_lines(const szh_geom3 &G) { return ((G.g0.count + 1 + LAG + UL - 1) / UL) * UL; }       // wave lines a wavefront runs (whole unrolled blocks)
// granule words: k-face [workgroup][8 half-beams][LINE r0 cells][NW], j-face [workgroup][wave lines][32 lanes][NW] (row X: what each lane of the
// workgroup's last row had finished last when its wave line X ended)
template <class T> SZH_HD size_t kface_words(const szh_geom3 &G) { const grid_t g = make_grid(G); return (size_t)g.nKB * g.nJG * 8 * LINE * (size_t)G.g0.count * szh_gran<T>::NW; }
template <class T> SZH_HD size_t jface_words(const szh_geom3 &G) { const grid_t g = make_grid(G); return (size_t)g.nKB * g.nJG * (size_t)wave_lines(G) * HL * szh_gran<T>::NW; }

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
__device__ __forceinline__ void hide(mask_t &v) { asm volatile("" : "+s"(v)); }
#endif
// select by a per-lane constant bit mask (all ones / all zeros; hidden from hipcc where it is set up): v_bfi_b32, no lane mask to set up in the
// scalar unit.  (Plain C, which hipcc turns into that instruction: behind an inline `asm` it puts an s_nop in front of the next VALU instruction.)
__device__ __forceinline__ unsigned bsel(unsigned mask, unsigned a, unsigned b) { return (a & mask) | (b & ~mask); }
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
typedef szh_io::rsrc_t rsrc_t;
using szh_io::make_rsrc;
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
// two codes (the low halves of two ring words) in one word
#ifdef SZH_HIPSIM
static inline unsigned pack2(unsigned lo, unsigned hi) { return (lo & 0xffffu) | (hi << 16); }
#else
__device__ __forceinline__ unsigned pack2(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x05040100u); }
#endif

template <class T, bool DEC, bool USEMEAN, bool HASREG>
struct beam {
    typedef shape<T> S;
    typedef gran_io<T> GIO;
    typedef typename GIO::reg_t greg_t;
    typedef u2_t cpiece_t;
    static constexpr int PITCH = S::PITCH, RINGB = S::RINGB, SZ = S::SZ, EV = S::EV, LP = S::LP, VB = S::VB, VPL = S::VPL;

    // rows are asked for PD lines before they go into the ring (one register set per line in flight).  PD = UL = 9 lines (45 steps) until late in round 6: in
    // the double sweep over arrays with regression blocks -- three streams of 16-byte pieces, two pieces a line -- 180 registers, and that kernel spilled 174
    // (the C4 slab: 2.90 ms against round 5's 2.41).  3 lines (15 steps: 2 - 3 us) cover the memory's latency as well: that kernel 1.39 ms, and every other
    // form a little faster too, its lanes moving fewer values between the two register files (profiles/r06_sweep_prefetch_distance.txt)
    static constexpr int PD = SZH_BM_PD, NSET = PD * EV;
    static_assert(UL % PD == 0, "register sets are named by the unrolled line");

    const szh_qargs<T> &a;
    OC_LDS unsigned char *lds0;                                    // first byte of the workgroup's rings; every LDS address below is relative to it (modulo 2^32)
    OC_LDS unsigned *prog;
    int lane, w, h, m, r0, nwl;
    bool has_prev, has_next, kf_in, kf_out, jf_in, jf_out;         // (uniform)
    unsigned spin_limit; bool timed_out;
    rsrc_t rs_v, rs_c, rs_k, rs_j, rs_x, rs_f;                     // the array (values), the codes, the k- / j-face granules, regression points' values / flags
    unsigned str_v, str_c, str_k, str_j;                           // bytes per line of each
    unsigned so_vl, so_xl, so_fl, so_cl, so_vs, so_cs, so_ko, so_ki, so_jo, so_ji;   // (uniform) the streams' running scalar offsets, see place()
    // per-lane state of the sweep
    T dl[LINE], lup[LINE], prev, Lprev, Bold, Bpold, cur_next, kfv[LINE], kfn[LINE], face_reg, sw_junk, pcopy;
    mask_t wrapm_next;
    unsigned tc_next, fl_next;
    unsigned pA, pB, cA, cB, mreg, qreg;                           // the lane's slot: p? + u PITCH (A: before its wrap, u < m; B: after); its code (CSEL)
    unsigned pc, cc;                                               // the pointers of the CURRENT step (selected a step ago)
    unsigned pf_at[RL], ji_at[RL];                                 // face push into the next ring / j-face granules into this ring: the lane's virtual-cell slot of line-slot a - q (modulo RL), per a
    unsigned prog_prev_at, prog_next_at; mask_t idle_k, idle_j;
    T ko_val; unsigned prog_at, it_v;                              // the k-face value on its way out; this wavefront's progress word (LDS address), wave lines done
    unsigned kaddr_h, trash, tstart, ring_w;
    mask_t m_first, m_vu[LINE];                                    // lane masks (scalar registers, hidden from hipcc where they are set up: it would rebuild them at every use): first lane of a half; the upper half's lanes whose cell at position U is virtual
    T caphU[LINE];                                                 // the quantiser's range test per position: half the capacity, -1 where the lane's cell is virtual
    mask_t upper_m;
    // events: roles and LDS places
    role_t rv[EV], rvs[EV], rc, rcs, rko, rki, rjo, rji, rf[EV];
    unsigned vl[EV], cw[EV], cl, ko_lds, ki_lds;
    v4u gv[NSET];                                                  // value rows on their way
    v4u gx[HASREG && !DEC ? NSET : 1]; unsigned gf[HASREG ? NSET : 1]; // HASREG: the regression points' reconstructions of the same rows (compress), their flag bytes
    cpiece_t gc[PD]; v4u wqc4; v4u wqv[EV];                        // code rows on their way in (inverse; HASREG compress: k_reg_points' codes); rows on their way out
    greg_t gk[DK], gj[DK];                                         // k-face / j-face granules on their way
    T eb, eb2, rh, caph, radf, mean; int radius; unsigned epoch;
    unsigned pv_prev, pv_next;                                     // the neighbouring wavefronts' progress words as read a step ago

    __device__ __forceinline__ beam(const szh_qargs<T> &args) : a(args) {}

    __device__ __forceinline__ bool give_up(unsigned &spins)
    {
        if (++spins <= spin_limit) return false;
        timed_out = true; spin_limit = 0;
        return true;
    }
    // The inputs of the first `planes` planes are in memory: compress, arrays with regression blocks -- the regression points (reconstructions, flags, codes:
    // k_reg_points); decompress -- the codes in natural order and the pre-scattered unpredictable values (k_permute<1>, k_unpred).  They are made slice after slice
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

    // offsets of a role's access of wave line X.  Inside the array (EDGE = false: every line any lane asks for exists) the line term is the
    // stream's running scalar offset `so` (= (X - 1) stride, advanced by the caller) and the lane part a constant; at the array's first and last
    // lines each lane checks its own line
    template <bool EDGE> __device__ __forceinline__ void place(const role_t &r, int X, unsigned stride, unsigned so, unsigned &off, unsigned &soff) const
    {
        if (!EDGE) { off = r.mid; soff = so; return; }
        const int L = X - r.half;
        off = (r.plain != SZH_BM_OOB && (unsigned)L < (unsigned)r0) ? r.plain + (unsigned)L * stride : SZH_BM_OOB;
        soff = 0u;
    }
    // (j-face granule rows are wave lines, not lines of the array: no half, rows 0 .. nwl - 1)
    template <bool EDGE> __device__ __forceinline__ void place_row(const role_t &r, int X, unsigned stride, unsigned so, unsigned &off, unsigned &soff) const
    {
        if (!EDGE) { off = r.plain; soff = so; return; }
        off = (r.plain != SZH_BM_OOB && (unsigned)X < (unsigned)nwl) ? r.plain + (unsigned)X * stride : SZH_BM_OOB;
        soff = 0u;
    }
    template <bool EDGE> __device__ __forceinline__ v4u load_v(int X, int ev) const { unsigned o, so; place<EDGE>(rv[ev], X, str_v, so_vl, o, so); return bld16<DEC ? 0 : 2>(rs_v, o, so); }
    template <bool EDGE> __device__ __forceinline__ v4u load_x(int X, int ev) const { unsigned o, so; place<EDGE>(rv[ev], X, str_v, so_vl, o, so); return bld16<0>(rs_x, o, so); }
    template <bool EDGE> __device__ __forceinline__ unsigned load_f(int X, int ev) const
    {
        unsigned o, so; place<EDGE>(rf[ev], X, str_v / (unsigned)SZ, so_fl, o, so);
#ifdef SZH_HIPSIM
        unsigned v = 0; for (int e = 0; e < VPL; ++e) if (inr(rs_f, o + (unsigned)e, so, 1)) v |= (unsigned)(unsigned char)rs_f.base[o + so + (unsigned)e] << (8 * e);
        return v;
#else
        return VPL == 4 ? (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_f, (int)o, (int)so, 0) : (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs_f, (int)o, (int)so, 0);
#endif
    }
    template <bool EDGE> __device__ __forceinline__ cpiece_t load_c(int X) const { unsigned o, so; place<EDGE>(rc, X, str_c, so_cl, o, so); return bld8<0>(rs_c, o, so); }
    template <bool EDGE> __device__ __forceinline__ greg_t load_k(int X) const { unsigned o, so; place<EDGE>(rki, X, str_k, so_ki, o, so); return GIO::ld(rs_k, o, so); }
    template <bool EDGE> __device__ __forceinline__ greg_t load_j(int X) const { unsigned o, so; place_row<EDGE>(rji, X, str_j, so_ji, o, so); return GIO::ld(rs_j, o, so); }

    // a row piece goes into the ring (line-slot offset `at`): HASREG: the regression points' values are their reconstructions (compress; the
    // inverse finds them in the array already), and their flags go into the upper halves of the cells' code words
    __device__ __forceinline__ void put_rows(unsigned at, int e, v4u xv, v4u xrv, unsigned fl)
    {
        if (HASREG) {
            if (!DEC) {
                T a_[VPL], b_[VPL];
                __builtin_memcpy(a_, &xv, 16); __builtin_memcpy(b_, &xrv, 16);
                for (int q = 0; q < VPL; ++q) a_[q] = ((fl >> (8 * q)) & 0xffu) ? b_[q] : a_[q];
                __builtin_memcpy(&xv, a_, 16);
                // (compress: the lanes write their codes as 16-bit halves, so whole words may be written here)
                if (VPL == 4) { v4u f4 = {(fl & 0xffu) << 16, (fl & 0xff00u) << 8, fl & 0xff0000u, (fl >> 8) & 0xff0000u}; lds_put16(lds0, ring_w + at + cw[e], f4); }
                else { u2_t f2; f2.x = (fl & 0xffu) << 16; f2.y = (fl & 0xff00u) << 8; lds_put8(lds0, ring_w + at + cw[e], f2); }
            } else {
                // (inverse: the code row of the line has been written as whole words just before: the flags go into the upper halves)
                for (int q = 0; q < VPL; ++q) lds_put<uint16_t>(lds0, ring_w + at + cw[e] + (unsigned)(4 * q + 2), (uint16_t)((fl >> (8 * q)) & 0xffu));
            }
        }
        lds_put16(lds0, ring_w + at + vl[e], xv);
    }
    // (inverse) a row of codes goes into the ring: 4 codes of 16 bits -> 4 words
    __device__ __forceinline__ void put_codes(unsigned at, cpiece_t c)
    {
        v4u c4 = {c.x & 0xffffu, c.x >> 16, c.y & 0xffffu, c.y >> 16};
        lds_put16(lds0, ring_w + at + cl, c4);
    }

    // the granules of row X have arrived -- or are asked for again (bounded)
    // (`idle`: the lanes that need nothing of the row)
    template <bool KFACE> __device__ __forceinline__ greg_t arrived(int X, greg_t g, mask_t idle)
    {
        if (__builtin_expect((lane_mask(GIO::ok(g, epoch)) | idle) != ~0ull, 0)) {      // (unlikely: the fast path must fall through -- a taken branch costs an instruction fetch)
            unsigned spins = 0;
#pragma unroll 1
            do {
                if (give_up(spins)) break;
                nap1();
                g = KFACE ? load_k<true>(X) : load_j<true>(X);
            } while ((lane_mask(GIO::ok(g, epoch)) | idle) != ~0ull);
        }
        return g;
    }

    static constexpr unsigned lineoff(int n) { return (unsigned)((((n % RL) + RL) % RL) * LP); }      // LDS offset of the line-slot n lines after the block's first

    // ---- the events of unrolled position U of wave line `it` (LL = it modulo UL, compile time)
    template <int U, int LL, bool EDGE> __device__ __forceinline__ void events(int it)
    {
        if constexpr (U == 1) {
            // finished rows (line it - LAG: every lane has left it) leave the ring: read here, stored one step later
            constexpr unsigned loY = lineoff(LL - LAG);
            if (!DEC) wqc4 = lds_get16(lds0, ring_w + loY + cl);
            else for_n<EV>([&](auto E) { constexpr int e = decltype(E)::value; wqv[e] = lds_get16(lds0, ring_w + loY + vl[e]); });
            ko_val = lds_get<T>(lds0, ring_w + loY + ko_lds);
        }
        if constexpr (U == 2) {
            unsigned o, so;
            place<EDGE>(rko, it - LAG, str_k, so_ko, o, so);
            if (!(SZH_BM_X & 4)) GIO::st(rs_k, o, so, ko_val, epoch);
            so_ko += str_k;
            if (!DEC) {
                cpiece_t out; out.x = pack2(wqc4.x, wqc4.y); out.y = pack2(wqc4.z, wqc4.w);
                if (HASREG) {
                    // the regression points of the row piece keep the codes k_reg_points gave them (the sweep passed their values through and has
                    // zeros there): per code a 16-bit mask from the flag in the upper half of its ring word
                    const cpiece_t old = gc[LL % PD];
                    const unsigned m0 = (wqc4.x > 0xffffu ? 0xffffu : 0u) | (wqc4.y > 0xffffu ? 0xffff0000u : 0u), m1 = (wqc4.z > 0xffffu ? 0xffffu : 0u) | (wqc4.w > 0xffffu ? 0xffff0000u : 0u);
                    out.x = (old.x & m0) | (out.x & ~m0); out.y = (old.y & m1) | (out.y & ~m1);
                    gc[LL % PD] = load_c<true>(it - LAG + PD);
                }
                place<EDGE>(rcs, it - LAG, str_c, so_cs, o, so);
                if (SZH_BM_X & 1) bst8<0>(rs_c, o, so, out); else if (!(SZH_BM_X & 2)) bst8<17>(rs_c, o, so, out);      // (written through: the host may follow this sweep's progress, see `pub` in run; the codes are read next by other kernels, from memory either way)
                so_cs += str_c;
            } else {
                for_n<EV>([&](auto E) { constexpr int e = decltype(E)::value; place<EDGE>(rvs[e], it - LAG, str_v, so_vs, o, so); bst16<0>(rs_v, o, so, wqv[e]); });
                so_vs += str_v;
            }
            // rows that have arrived go in (line it + 1), and the register set that carried them is sent for the rows UL lines further on
            constexpr unsigned loX = lineoff(LL + 1);
            if (DEC) { put_codes(loX, gc[LL % PD]); gc[LL % PD] = load_c<EDGE>(it + 1 + PD); so_cl += str_c; wave_sync(); }      // (lock step: on the CPU shim the lanes are fibres, and the flags below go into words other lanes' code rows write whole)
            for_n<EV>([&](auto E) {
                constexpr int e = decltype(E)::value, set = (LL % PD) * EV + e;
                put_rows(loX, e, gv[set], gx[HASREG && !DEC ? set : 0], gf[HASREG ? set : 0]);
                gv[set] = load_v<EDGE>(it + 1 + PD, e);
                if (HASREG && !DEC) gx[HASREG && !DEC ? set : 0] = load_x<EDGE>(it + 1 + PD, e);
                if (HASREG) gf[HASREG ? set : 0] = load_f<EDGE>(it + 1 + PD, e);
            });
            so_vl += str_v; if (HASREG) so_fl += str_v / (unsigned)SZ;
        }
        if constexpr (U == 2) {
            // (a wavefront without a neighbour reads the word that always says "far ahead")
            pv_prev = lds_get<unsigned>(lds0, prog_prev_at);
            pv_next = lds_get<unsigned>(lds0, prog_next_at);
        }
        if constexpr (U == 3) {
            // Everything the NEXT wave line needs from others, tested in ONE branch (a conditional branch, taken or not, costs a wavefront alone on its
            // SIMD the better part of a step: it has to wait for the compare behind it): the k-face granules of wave line it + 1 and the j-face
            // granules of row it + 3 (lane m's value is the virtual cell of line it + 2 - q(m)) have arrived; the wavefront below (in j) is far enough
            // ahead for the virtual cells read from line it + 1 on (it has finished wave line it + 3), the one above not too far behind.  All of
            // that is the usual case; otherwise `late` sorts it out.
            constexpr int set = LL % DK, aj = (LL + 2) % RL;
            const bool need_k = rki.plain != SZH_BM_OOB && (!EDGE || (unsigned)(it + 1 - rki.half) < (unsigned)r0);
            const bool need_j = rji.plain != SZH_BM_OOB && (!EDGE || (unsigned)(it + 3) < (unsigned)nwl);
            const mask_t idk = EDGE ? ~lane_mask(need_k) : idle_k, idj = EDGE ? ~lane_mask(need_j) : idle_j;
            greg_t g_k = gk[set], g_j = gj[set];
            if (!(SZH_BM_X & 65536)) {
                const mask_t both = (lane_mask(GIO::ok(g_k, epoch)) | idk) & (lane_mask(GIO::ok(g_j, epoch)) | idj);
                const int pvp = uni((int)pv_prev), pvn = uni((int)pv_next);
                const unsigned behind = (unsigned)((pvp - (it + 4)) | (pvn - (it - (RL - 3)))) >> 31;      // (a sign bit: one of them is short)
                if (__builtin_expect((unsigned)(both != ~0ull) | ((SZH_BM_X & 8) ? 0u : behind), 0)) {
                    g_k = arrived<true>(it + 1, g_k, idk);
                    g_j = arrived<false>(it + 3, g_j, idj);
                    if (!(SZH_BM_X & 8)) {
                        if (pvp < it + 4) wait_prog(prog + (w - 1), it + 4);
                        if (pvn < it - (RL - 3)) wait_prog(prog + (w + 1), it - (RL - 3));
                    }
                }
                // (lanes and wavefronts that take no part write to their write-only words)
                lds_put<T>(lds0, ki_lds + (unsigned)(((LL + 1) % KRL) * S::KLS), (!EDGE || need_k) ? GIO::val(g_k) : (T)0);
                // (the upper half's lanes write into the upper half's part of the same slots: nobody reads a virtual cell of the upper half from the ring)
                lds_put<T>(lds0, ji_at[aj], (!EDGE || need_j) ? GIO::val(g_j) : (T)0);
            }
            if (!(SZH_BM_X & 131072)) { gk[set] = load_k<EDGE>(it + 1 + DK); gj[set] = load_j<EDGE>(it + 3 + DK); }
            so_ki += str_k; so_ji += str_j;
        }
        if constexpr (U == 4) read_kface<(LL + 1) % KRL>();      // (the k-face values of the next line's cells: one read per line, not one per step)
    }

    __device__ __forceinline__ static T tabs(T v) { return sizeof(T) == 8 ? (T)__builtin_fabs((double)v) : (T)__builtin_fabsf((float)v); }
    __device__ __forceinline__ static T ttrunc(T v) { return sizeof(T) == 8 ? (T)__builtin_trunc((double)v) : (T)__builtin_truncf((float)v); }
    __device__ __forceinline__ static T tsign(T mag, T from) { return sizeof(T) == 8 ? (T)__builtin_copysign((double)mag, (double)from) : (T)__builtin_copysignf((float)mag, (float)from); }
    // x y + 0 in ONE rounding: the product rounded, and a zero product +0 whatever its sign -- what `x * y` followed by `+ 0` gives (|x| >= 1 or x = 0 here: no underflow)
    __device__ __forceinline__ static T tmul0(T x, T y) { return sizeof(T) == 8 ? (T)__builtin_fma((double)x, (double)y, 0.0) : (T)__builtin_fmaf((float)x, (float)y, 0.0f); }

    // the k-face values of a whole line (cell U of the lane's half at kfv[U]): every lane reads, only the halves' first lanes use them
    template <int KL> __device__ __forceinline__ void read_kface()
    {
        if (SZH_BM_X & 16) return;
        const unsigned at = kaddr_h + (unsigned)(KL * S::KLS);
        if (SZ == 4) {
            const v4u q = lds_get16(lds0, at);
            kfn[0] = __uint_as_float(q.x); kfn[1] = __uint_as_float(q.y); kfn[2] = __uint_as_float(q.z); kfn[3] = __uint_as_float(q.w);
            kfn[4] = lds_get<T>(lds0, at + 16u);
        } else for_n<LINE>([&](auto UU) { constexpr int u_ = decltype(UU)::value; kfn[u_] = lds_get<T>(lds0, at + (unsigned)(u_ * SZ)); });
    }
    // the lanes that have not wrapped at unrolled step UN (m > UN), as a lane mask; `mreg` is hidden from hipcc block by block: it would hoist the
    // 31 compares out of the loop as 31 lane masks in scalar registers, and spill them
    template <int UN> __device__ __forceinline__ void next_wrap()
    {
        if (UN < HL - 1 && !(SZH_BM_X & 512)) wrapm_next = lane_mask(mreg > (unsigned)UN);
    }
    // ---- one step: the cell the lane is at
    // One wavefront per SIMD: a dependent VALU instruction issues ~9 cycles after the one it waits for, an independent one after 4 (tools/ubench),
    // and hipcc's scheduler models neither.  So the step is written in the order it should issue, one chain operation and one or two independent
    // ones per group, and the groups are pinned (SZH_SB: nothing crosses).
    template <int U, int LL, bool EDGE> __device__ __forceinline__ void step(int it)
    {
        constexpr int u = LL * LINE + U, un = (u + 1) % RS, un2 = (u + 2) % RS;
        const T cur_raw = cur_next, kf = kfv[U];
        const unsigned tc_in = tc_next, fl_in = fl_next;
        // (the copy the swap consumes was made at the end of the step before, and the lane mask of the slot select in the middle of it: back to
        // back with their consumers both need wait states -- three s_nop per step)
        const T Lraw = shr1(prev);                                       // (i, j, k-1): the left lane's previous result
        const mask_t wrapm = wrapm_next;
        const unsigned pn = (un < HL - 1 && !(SZH_BM_X & 512)) ? (in_mask(wrapm) ? pA : pB) : pB;
        const unsigned cn = !S::CSEL ? pn : (un < HL - 1 ? (in_mask(wrapm) ? cA : cB) : cB);
        SZH_SB;
        T L = in_mask(m_first) ? kf : Lraw;                              // (a half's first lane: the k-face of the tile on the left)
        const bool started = !EDGE || (unsigned)(it * LINE + U) >= tstart;      // (at the array's first lines: lanes that have not started hold zeros)
        if (EDGE) L = started ? L : (T)0;
        const T sw = (SZH_BM_X & 256) ? sw_junk : low_to_high(pcopy, sw_junk);
        SZH_SB;
        const T B = dl[U], Bp = lup[U], C = Bold, Cp = Bpold;
        // [-1] + [-s1] + [-s0] - [-s1-1] - [-s0-1] - [-s0-s1] + [-s0-s1-1], left to right (sz_float.c:7268)
        const T s1 = L + prev;
        if (!(SZH_BM_X & 32)) cur_next = lds_get<T>(lds0, pn + (unsigned)(un * PITCH));        // what the NEXT step needs from the rings is requested now
        if (DEC) tc_next = lds_get<uint16_t>(lds0, cn + (unsigned)(un * PITCH + (S::CSEL ? 0 : VB)));
        if (HASREG) fl_next = lds_get<uint8_t>(lds0, cn + (unsigned)(un * PITCH + (S::CSEL ? 0 : VB) + 2));
        SZH_SB;
        const T s2 = s1 + B;
        SZH_SB;
        const T s3 = s2 - Lprev;
        // a virtual cell: the j-face -- from the ring (lower half), from the lower half's previous result (upper half)
        const T cur = in_mask(m_vu[U]) ? sw : cur_raw;
        SZH_SB;
        const T s4 = s3 - Bp;
        SZH_SB;
        const T s5 = s4 - C;
        SZH_SB;
        const T pred = s5 + Cp;
        SZH_SB;
        T rec;
        if (!DEC) {
            // the quantiser (sz_float.c:7270-7287 with a shorter dependency chain, bit for bit the same results); a virtual
            // cell fails the range test (its limit is -1) and hands its value on unchanged
            const T diff = cur - pred;
            SZH_SB;
            const T hq0 = tabs(diff) * rh;
            SZH_SB;
            const T hq = hq0 + (T)0.5;
            next_wrap<un2>();
            SZH_SB;
            const T tq = ttrunc(hq);
            mask_t okm = lane_mask(hq < caphU[U]);
            if (HASREG) okm &= lane_mask(fl_in == 0u);
            SZH_SB;
            const T ts = tsign(tq, diff);
            SZH_SB;
            const T m2 = tmul0(ts, eb2);
            const T cf = radf + ts;
            SZH_SB;
            int code = (int)cf;
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
            if (!(SZH_BM_X & 64)) {
            lds_put<uint16_t>(lds0, cc + (unsigned)(u * PITCH + (S::CSEL ? 0 : VB)), (uint16_t)code);
            lds_put<T>(lds0, pc + (unsigned)(u * PITCH), rec);
            } else if (code == 12345) lds_put<T>(lds0, pc, rec);
        } else {
            int cq = (int)tc_in;
            bool is_mean = false;
            if (USEMEAN) { is_mean = cq == radius; if (cq != 0 && cq < radius) cq += 1; }     // szd_float.c:3784
            const T mq = (T)(cq - radius) * eb2;
            next_wrap<un2>();
            mask_t um = lane_mask(tc_in != 0u) & lane_mask(caphU[U] > (T)0);
            if (HASREG) um &= lane_mask(fl_in == 0u);
            SZH_SB;
            T r = pred + mq;                                                                  // = pred + 2 (c - radius) eb (szd_float.c:5786)
            if (USEMEAN && is_mean) r = mean;
            rec = in_mask(um) ? r : cur;                                                       // zero code: the pre-scattered value
            if (EDGE) rec = started ? rec : (T)0;
            lds_put<T>(lds0, pc + (unsigned)(u * PITCH), rec);
        }
        // the last row of the upper half is the j-face of the wavefront above: kept where the lane made it, handed on at the end of the line
        if (!(SZH_BM_X & 1024)) face_reg = in_mask(m_vu[(U + 1) % LINE]) ? rec : face_reg;      // (the upper half's lanes whose NEXT cell is virtual)
        dl[U] = rec; lup[U] = L;
        Bold = B; Bpold = Bp;
        Lprev = L; prev = rec; sw_junk = sw;
        pcopy = rec; hide(pcopy);                                        // (a register of its own: the swap overwrites half of it)
        pc = pn; cc = cn;
    }

    template <int LL, bool EDGE> __device__ __forceinline__ void line(int it)
    {
        // the wavefront below (in j) must be far enough ahead for the virtual cells read during this line, the one above not too far behind
        // (their progress words were read during the last step of the line before: no LDS round trip here unless one of them is late)
        for_n<LINE>([&](auto UU) { constexpr int u_ = decltype(UU)::value; kfv[u_] = kfn[u_]; });
        for_n<LINE>([&](auto UU) {
            constexpr int U = decltype(UU)::value;
            wave_sync();
            if (!(SZH_BM_X & 128) && !((SZH_BM_X >> (12 + U)) & 1)) events<U, LL, EDGE>(it);      /* (8192 / 16384 / 32768: no events at position 1 / 2 / 3) */
            order();
            step<U, LL, EDGE>(it);
            order();
        });
        // the faces of this line go up: each lane's latest last-row result -- the virtual cell of line it - 1 - q(m) -- into that slot of the ring
        // of the wavefront above (one write per line, not per step: the wavefront above waits for whole lines anyway), or, from the workgroup's
        // last wavefront, into row `it` of the j-face granules
        if (!(SZH_BM_X & 2048)) {
            constexpr int ap = (LL + RL - 1) % RL;
            // (the lower half's lanes write into the upper half's part of the same slots of the next ring: nobody reads those; the workgroup's
            // last wavefront: into the lanes' write-only words)
            lds_put<T>(lds0, pf_at[ap], face_reg);
            unsigned o, so;
            place_row<EDGE>(rjo, it, str_j, so_jo, o, so);
            if (!(SZH_BM_X & 4)) GIO::st(rs_j, o, so, face_reg, epoch);
            so_jo += str_j;
        }
        it_v += 1u;
        if (!(SZH_BM_X & 2048)) lds_put<unsigned>(lds0, prog_at, it_v);
    }
    template <bool EDGE> __device__ __forceinline__ void block(int it0)
    {
        for_n<UL>([&](auto L_) { constexpr int LL = decltype(L_)::value; line<LL, EDGE>(it0 + LL); });
    }

    __device__ __forceinline__ void run(int kb, int jg, OC_LDS unsigned char *rings, OC_LDS unsigned *prog_)
    {
        const szh_geom3 &G = a.G;
        prog = prog_;
        r0 = G.g0.count; nwl = wave_lines(G);
        const int r1 = G.g1.count, r2 = G.g2.count;
        lane = (int)(threadIdx.x & 63u); w = uni((int)(threadIdx.x >> 6)); h = lane >> 5; m = lane & 31;
        lds0 = rings;
        const grid_t gr = make_grid(G);
        const int k0 = kb * HL, jw0 = jg * JG + w * JW;
        has_prev = w > 0; has_next = w + 1 < WPG;
        jf_in = w == 0 && jg > 0; jf_out = w == WPG - 1 && jg + 1 < gr.nJG; kf_in = kb > 0; kf_out = kb + 1 < gr.nKB;
        spin_limit = 1u << 22; timed_out = false;
        eb = a.eb; eb2 = eb + eb; rh = a.recip * (T)0.5; caph = (T)(a.cap - 2) * (T)0.5; radf = (T)a.radius; mean = a.mean; radius = a.radius; epoch = a.epoch;
        const int64_t wg = (int64_t)kb * gr.nJG + jg;
        const uint64_t nbytes = (uint64_t)G.n * SZ;
        rs_v = make_rsrc(DEC ? (const void *)a.out : (const void *)a.data, (unsigned)nbytes);
        rs_c = make_rsrc(a.codes, (unsigned)((uint64_t)G.n * 2));
        rs_x = make_rsrc(HASREG && !DEC ? (const void *)a.xr : (const void *)a.codes, HASREG && !DEC ? (unsigned)nbytes : 0u);
        rs_f = make_rsrc(HASREG ? (const void *)a.ptflags : (const void *)a.codes, HASREG ? (unsigned)G.n : 0u);
        str_v = (unsigned)(G.d0 * SZ); str_c = (unsigned)(G.d0 * 2); str_k = (unsigned)(LINE * GIO::BYTES); str_j = (unsigned)(HL * GIO::BYTES);
        ring_w = (unsigned)(w * RINGB);
        {   // value rows: 64 lanes x 16 B = RPE row pieces of HB bytes: RH rows of each half
            const int r = lane / S::LPR, p = lane - r * S::LPR, vh = r / S::RH;
            for (int e = 0; e < EV; ++e) {
                const int rr = r % S::RH + S::RH * e, j = jw0 + C1 * vh + rr, kk = k0 + p * VPL;
                const unsigned off = (unsigned)(((int64_t)j * G.d1 + kk) * SZ);
                const bool in = j < r1 && kk < r2;
                rv[e] = make_role(true, in ? off : 0u, vh, str_v);           // (rows outside the array: any readable place -- nobody looks at them)
                rvs[e] = make_role(in, off, vh, str_v);
                rf[e] = make_role(true, in ? off / (unsigned)SZ : 0u, vh, str_v / (unsigned)SZ);
                vl[e] = (unsigned)((1 + rr) * PITCH + vh * S::HB + p * 16);
                cw[e] = (unsigned)((1 + rr) * PITCH + VB + (vh * HL + p * VPL) * 4);
            }
        }
        {   // code rows: 64 lanes x 8 B (4 codes; 16 B of the ring) = 8 row pieces of 64 bytes
            const int r = lane / 8, p = lane - r * 8, rr = r % C1, ch = r / C1, j = jw0 + C1 * ch + rr, kk = k0 + 4 * p;
            const unsigned off = (unsigned)(((int64_t)j * G.d1 + kk) * 2);
            const bool in = j < r1 && kk < r2;
            rc = make_role(true, in ? off : 0u, ch, str_c);
            rcs = make_role(in, off, ch, str_c);
            cl = (unsigned)((1 + rr) * PITCH + VB + (ch * HL + p * 4) * 4);
        }
        trash = (unsigned)(S::TRASH0 + (w * 64 + lane) * 8);       // (the lane's write-only word behind the last ring)
        {   // k-face granules: lane e < 10: cell e % 5 of half e / 5 of a wave line; [workgroup][8 half-beams][LINE r0 cells]
            const bool en = lane < 2 * LINE;
            const int khh = en ? lane / LINE : 0, kuu = en ? lane % LINE : 0;
            const uint64_t cells = (uint64_t)LINE * r0;
            const int64_t wgl = (int64_t)(kb > 0 ? kb - 1 : 0) * gr.nJG + jg;
            rs_k = make_rsrc(a.faceI, (unsigned)(kface_words<T>(G) * 8));
            rko = make_role(en && kf_out, (unsigned)((((uint64_t)(wg * 8 + 2 * w + khh)) * cells + kuu) * GIO::BYTES), khh, str_k);
            rki = make_role(en && kf_in, (unsigned)((((uint64_t)(wgl * 8 + 2 * w + khh)) * cells + kuu) * GIO::BYTES), khh, str_k);
            ko_lds = (unsigned)(kuu * PITCH + khh * S::HB + (HL - 1) * SZ);
            // (lanes that take no part write to their write-only word; the line term of the address stays within the slack behind those words)
            ki_lds = (en && kf_in) ? (unsigned)(S::KR0 + w * S::KRB + khh * S::KHS + kuu * SZ) : trash;
        }
        {   // j-face granules [workgroup][wave line][32]: out: the upper half's lanes, row `it`; in: the lower half's lanes, row it + 3
            const int64_t wgl = (int64_t)kb * gr.nJG + (jg > 0 ? jg - 1 : 0);
            rs_j = make_rsrc(a.faceJ, (unsigned)(jface_words<T>(G) * 8));
            rjo = make_role(h == 1 && jf_out, (unsigned)(((uint64_t)wg * (unsigned)nwl * HL + (unsigned)m) * GIO::BYTES), 0, str_j);
            rji = make_role(h == 0 && jf_in, (unsigned)(((uint64_t)wgl * (unsigned)nwl * HL + (unsigned)m) * GIO::BYTES), 0, str_j);
        }
        idle_k = ~lane_mask(rki.plain != SZH_BM_OOB); idle_j = ~lane_mask(rji.plain != SZH_BM_OOB);
        for (int u = 0; u < LINE; ++u) { dl[u] = 0; lup[u] = 0; }
        prev = 0; Lprev = 0; Bold = 0; Bpold = 0; sw_junk = 0; face_reg = 0; pcopy = 0;
        // the lane's slot at unrolled step u is (u - m) modulo RS: pB + u PITCH from u = m on, pA + u PITCH before
        pB = ring_w + (unsigned)(h * S::HB + m * SZ) - (unsigned)(m * PITCH); pA = pB + (unsigned)RINGB;
        cB = ring_w + (unsigned)(VB + lane * 4) - (unsigned)(m * PITCH); cA = cB + (unsigned)RINGB;
        mreg = (unsigned)m; qreg = (unsigned)((m + LINE - 1) / LINE);
        // faces: after wave line `it` the lane's latest finished last-row cell belongs to beam line it - 1 - q (q = ceil(m / LINE)): the slot of THAT
        // line's virtual cell -- in the next ring (push), or, for the granules of row X = it + 3, in this one (line it + 2 - q)
        // (the lanes of the half that takes no part aim at the other half's part of the same slots)
        for_n<RL>([&](auto A_) {
            constexpr int a_ = decltype(A_)::value;
            const unsigned ls = (unsigned)(((a_ - (int)qreg) % RL + RL) % RL) * (unsigned)LP;
            pf_at[a_] = has_next ? ring_w + (unsigned)RINGB + (unsigned)((1 - h) * S::HB + m * SZ) + ls : trash; hide(pf_at[a_]);
            ji_at[a_] = jf_in ? ring_w + (unsigned)(h * S::HB + m * SZ) + ls : trash; hide(ji_at[a_]);
        });
        prog_prev_at = (unsigned)(S::PROG0 + 4 * (has_prev ? w - 1 : WPG)); hide(prog_prev_at);
        prog_next_at = (unsigned)(S::PROG0 + 4 * (has_next ? w + 1 : WPG)); hide(prog_next_at);
        prog_at = (unsigned)(S::PROG0 + 4 * w); hide(prog_at); it_v = 0u;
        upper_m = UPPER;
        m_first = FIRSTCOL; hide(m_first);
        for_n<LINE>([&](auto UU) {
            constexpr int u = decltype(UU)::value;
            const bool virt = ((u - m) % LINE + LINE) % LINE == 0;
            m_vu[u] = virt_upper_mask<u>(); hide(m_vu[u]);
            caphU[u] = virt ? (T)-1 : caph; hide(caphU[u]);
        });
        kaddr_h = (unsigned)(S::KR0 + w * S::KRB + h * S::KHS); hide(kaddr_h);
        tstart = (unsigned)(m + LINE * h);
        {   // the k-face ring reads zeros where nothing arrives (no tile on the left); the virtual cells of the array's lower face read zeros (a
            // virtual cell writes back what it read, so they stay zeros), as do those of lines the wavefront below has not pushed yet
            OC_LDS unsigned char *kring = rings + S::KR0 + w * S::KRB;
            for (int e = lane; e < S::KRB / 4; e += 64) lds_put<unsigned>(kring, (unsigned)e * 4u, 0u);
            const v4u z = {0u, 0u, 0u, 0u};
            for (int e = lane; e < RL * (S::HB / 16); e += 64) lds_put16(lds0, ring_w + (unsigned)((e / (S::HB / 16)) * LP + (e % (S::HB / 16)) * 16), z);
        }
        // (arrays with regression blocks whose points arrive WHILE the sweep runs -- a.reg_ready, see wait_fed: nothing of a plane is asked for before it is there)
        fed = 0; feed_late = false;
        if (a.reg_ready) wait_fed(UL + 2);
        // ---- prologue: the first lines' rows are requested; line 0 goes into the ring
        so_vl = 0u; so_xl = 0u; so_fl = 0u; so_cl = 0u; so_vs = 0u; so_cs = 0u; so_ko = 0u; so_ki = 0u; so_jo = 0u; so_ji = 0u;
        {
            v4u first[EV], firstx[EV]; unsigned firstf[EV];
            for_n<EV>([&](auto E) {
                constexpr int e = decltype(E)::value;
                first[e] = load_v<true>(0, e);
                firstx[e] = (HASREG && !DEC) ? load_x<true>(0, e) : first[e];
                firstf[e] = HASREG ? load_f<true>(0, e) : 0u;
            });
            for_n<PD>([&](auto L_) { constexpr int LL = decltype(L_)::value; for_n<EV>([&](auto E) {
                constexpr int e = decltype(E)::value, set = LL * EV + e;
                gv[set] = load_v<true>(LL + 1, e);
                if (HASREG && !DEC) gx[HASREG && !DEC ? set : 0] = load_x<true>(LL + 1, e);
                if (HASREG) gf[HASREG ? set : 0] = load_f<true>(LL + 1, e);
            }); });
            if (DEC) {
                const cpiece_t c0 = load_c<true>(0);
                for_n<PD>([&](auto L_) { constexpr int LL = decltype(L_)::value; gc[LL] = load_c<true>(LL + 1); });
                put_codes(0u, c0);
                wave_sync();
            }
            for_n<EV>([&](auto E) { constexpr int e = decltype(E)::value; put_rows(0u, e, first[e], firstx[e], firstf[e]); });
            if (!DEC && HASREG) for_n<PD>([&](auto L_) { constexpr int LL = decltype(L_)::value; gc[LL] = load_c<true>(LL - LAG); });
            // granules: k-face row 0 (taken now), rows 1 .. DK on their way; j-face rows 0 .. 2 (lines -q .. 1 - q: taken now, the first virtual cells
            // are read from step 0 on), rows 3 .. 2 + DK on their way
            const greg_t k0g = load_k<true>(0);
            const greg_t j0g = load_j<true>(0), j1g = load_j<true>(1), j2g = load_j<true>(2);
            for_n<DK>([&](auto L_) { constexpr int LL = decltype(L_)::value; gk[LL] = load_k<true>(LL + 1); gj[LL] = load_j<true>(LL + 3); });
            if (kf_in) {
                const bool need = rki.plain != SZH_BM_OOB && (unsigned)(0 - rki.half) < (unsigned)r0;
                const greg_t g = arrived<true>(0, k0g, ~lane_mask(need));
                lds_put<T>(lds0, ki_lds, need ? GIO::val(g) : (T)0);
            }
            if (jf_in) {
                // row X carries the virtual cell of line X - 1 - q: rows 0 .. 2 hold lines that exist only for X - 1 - q >= 0
                const greg_t jg3[3] = {j0g, j1g, j2g};
                for (int X = 0; X < 3; ++X) {
                    const int ln = X - 1 - (int)qreg;
                    const bool need = rji.plain != SZH_BM_OOB && ln >= 0;
                    const greg_t g = arrived<false>(X, jg3[X], ~lane_mask(need));
                    if (need) lds_put<T>(lds0, ring_w + (unsigned)((ln % RL) * LP + m * SZ), GIO::val(g));
                }
            }
        }
        // (EDGE blocks place every access themselves, soff = 0; the first block inside the array starts its streams where they stand then)
        if (has_prev) wait_prog(prog + (w - 1), 3);
        pv_prev = (unsigned)SZH_BM_INF; pv_next = (unsigned)SZH_BM_INF;
        order();
        wave_sync();
        pc = mreg > 0u ? pA : pB; cc = S::CSEL ? (mreg > 0u ? cA : cB) : pc;
        wrapm_next = lane_mask(mreg > 1u);                                // (step 0 selects the slot of step 1)
        cur_next = lds_get<T>(lds0, pc);
        tc_next = DEC ? (unsigned)lds_get<uint16_t>(lds0, cc + (unsigned)(S::CSEL ? 0 : VB)) : 0u;
        fl_next = HASREG ? (unsigned)lds_get<uint8_t>(lds0, cc + (unsigned)((S::CSEL ? 0 : VB) + 2)) : 0u;
        for_n<LINE>([&](auto UU) { constexpr int u_ = decltype(UU)::value; kfn[u_] = 0; }); read_kface<0>();
        { const v4u z = {0u, 0u, 0u, 0u}; for (int e = 0; e < EV; ++e) wqv[e] = z; wqc4 = z; }
        // ---- the lines: every lane has left line it - LAG when lane 0 enters line it.  Blocks of UL lines; the ones in which every line any
        // lane asks for or stores exists take the variant without per-lane line checks
        const int nblk = nwl / UL;
        unsigned *const pub = (!DEC && a.tile_done) ? a.tile_done + (((int64_t)kb * gr.nJG + jg) * WPG + w) : nullptr;     // (uniform)
        const int PUBB = (a.pub_lines > 0 ? a.pub_lines : 32) / UL > 0 ? (a.pub_lines > 0 ? a.pub_lines : 32) / UL : 1;     // blocks between two words (a word every ~32 lines by default)
#pragma unroll 1
        for (int b = 0; b < nblk; ++b) {
            const int it0 = b * UL;
            hide(mreg);
            if (a.reg_ready) wait_fed(it0 + 2 * UL + 1);          // (the block's lines ask for rows up to wave line it0 + 2 UL)
            const bool mid = it0 >= LAG + 1 && it0 + 2 * UL <= r0 - 1;
            if (mid) {
                // the streams' scalar offsets for this block: wave line X of a stream is at (X - 1) stride (rows of the j-face: X stride)
                so_vl = (unsigned)(it0 + PD) * str_v; so_fl = (unsigned)(it0 + PD) * (str_v / (unsigned)SZ); so_cl = (unsigned)(it0 + PD) * str_c;
                so_vs = (unsigned)(it0 - LAG - 1) * str_v; so_cs = (unsigned)(it0 - LAG - 1) * str_c;
                so_ko = (unsigned)(it0 - LAG - 1) * str_k; so_ki = (unsigned)(it0 + DK) * str_k;
                so_jo = (unsigned)it0 * str_j; so_ji = (unsigned)(it0 + 3 + DK) * str_j;
                block<false>(it0);
            } else block<true>(it0);
            // the host starts the entropy stage's passes over lines every wavefront has passed (szhip.hip): how many of THIS wavefront's lines have
            // their codes in memory -- the launch's epoch beside the count
            if (pub && ((b + 1) % PUBB == 0 || b + 1 == nblk)) {
                int rows = b + 1 == nblk ? r0 : it0 + UL - 1 - LAG;
                rows = rows < 0 ? 0 : (rows > r0 ? r0 : rows);
                if (b + 1 == nblk || rows > 0) {
                    // the codes went out write-through (sc0 sc1): once this wavefront's memory counter is empty they are in memory, and the word may follow.
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
    typedef shape<T> S;
    __shared__ __attribute__((aligned(16))) unsigned char rings[S::LDSB];     // rings, write-only words of lanes without a face value, k-face rings
    __shared__ unsigned tk_s;
    OC_LDS unsigned *const prog = (OC_LDS unsigned *)((OC_LDS unsigned char *)rings + S::PROG0);
    const unsigned ntiles = (unsigned)(a.nI * a.nJ);
    for (unsigned itile = 0;; ++itile) {
        __syncthreads();
        if (threadIdx.x < WPG + 2) prog[threadIdx.x] = threadIdx.x == WPG ? (unsigned)SZH_BM_INF : 0u;
        if (threadIdx.x == 0) {
            const unsigned t = a.ticket_mode ? blockIdx.x + itile * gridDim.x : atomicAdd(a.ticket, 1u);
            tk_s = t < ntiles ? szh_pencil_order_at(a.nI, a.nJ, t) : 0xffffffffu;
        }
        __syncthreads();
        const unsigned ij = (unsigned)szh_bm::uni((int)tk_s);
        if (ij == 0xffffffffu) break;
        beam<T, DEC, USEMEAN, HASREG> s(a);
        s.run((int)(ij >> 16), (int)(ij & 0xffffu), (OC_LDS unsigned char *)rings, prog);
    }
}
