// szh_ompcol.h -- the box quantiser of the reference's OpenMP container (szh_omp.h) as a COLUMN-PER-LANE sweep (round 4).
//   box quantiser   sz/src/sz_float.c:4704-5012   SZ_compress_float_3D_MDQ_RA_block     (double: sz_double.c, same name)
//   inverse         sz/src/szd_float.c:2848       decompressDataSeries_float_3D_RA_block (called from sz_omp.c:366-566)
// Same arithmetic, same codes, same stream as k_omp_box (szh_omp.h); what changes is who does what:
//   * k_omp_box gave every ROW (k, i, :) of a box a lane and a 1024-lane workgroup a box: a barrier per step, a third of the lane-steps doing
//     work, every `global_load_dwordx4` touching 64 different 128-byte lines.  Measured on MI355X (round 4, first call): 2.88 ms at 512^3 f32.
//   * here a lane owns a COLUMN (:, :, j) of a box and walks its C0 x C1 cells in row-major order, one cell per step; lane j is one step
//     behind lane j - 1.  A wavefront is 64 / C2 boxes side by side (two 32^3 boxes), on its own: no barrier, no other wavefront to wait for.
//       - (k, i, j-1), the left neighbour, is what lane j - 1 produced one step ago: ONE DPP `wave_shr:1` per step;
//       - (k, i-1, j) is the lane's own previous result; (k-1, i, j) is what it produced C1 + 1 steps ago: a DELAY LINE of C1 + 1 registers,
//         indexed by the step number modulo C1 + 1 -- a compile-time index in a loop unrolled C1 + 1 times; the same for the left lane's
//         values ((k-1, i, j-1): what the DPP delivered C1 + 1 steps ago).  No LDS and no cross-lane traffic in the dependent chain but
//         that one DPP;
//       - a line (k, :) takes C1 + 1 steps: its first step is a VIRTUAL cell i = -1 whose result is forced to +0.  Every neighbour a cell
//         on a box face lacks (i - 1 for i = 0; k - 1 for k = 0: the delay lines start as zeros; j - 1 for j = 0: the DPP's `old`
//         operand) therefore arrives as +0 and ONE predictor expression -- the reference's 7-point sum, in its order -- serves every
//         cell but those of the row (0, 0, :) (see szh_omp.h for why adding +0 changes neither a code nor a reconstruction).
//   * memory: the rows of a box are read with 16 bytes per lane = 1 KB per instruction (whole 128-byte rows of 4 (float) / 2 (double)
//     consecutive rows of each box), eight loads in flight per wavefront, and dropped into an LDS RING of RS row slots indexed by the
//     cell's step number (a slot = the row of every box of the wavefront + its codes); lane j reads ITS value of the row it is at from the
//     ring (ds_read_b32, conflict-free: the slot pitch is a multiple of 32 banks) and writes its code next to it; finished rows of codes
//     leave as 16 bytes per lane = 8 rows per box and instruction.  The inverse is the mirror image (codes in, values out).
//   Steps of a 32^3 box: 32 * 33 + 31 = 1087, each one DPP + ~35 VALU instructions for 64 points; a CU holds 8 such wavefronts (LDS).
// Supported shapes: C1 (box extent along dim 1) = 32, C2 (along dim 2) = 32, any C0; everything else stays on k_omp_box.
#pragma once
#include <utility>

namespace szh_oc {

#ifdef SZH_HIPSIM
#define OC_LDS
static inline void wave_sync() { (void)__all(1); }        // (the lanes of the CPU shim are fibres: what lock-step execution gives for free)
#else
#define OC_LDS __attribute__((address_space(3)))
__device__ __forceinline__ void wave_sync() {}
#endif
typedef szh_io::v4u v4u;

template <class E> __device__ __forceinline__ E lds_get(OC_LDS unsigned char *base, unsigned off) { return *(volatile OC_LDS E *)(base + off); }
template <class E> __device__ __forceinline__ void lds_put(OC_LDS unsigned char *base, unsigned off, E v) { *(volatile OC_LDS E *)(base + off) = v; }
__device__ __forceinline__ v4u lds_get16(OC_LDS unsigned char *base, unsigned off) { return *(OC_LDS v4u *)(base + off); }
__device__ __forceinline__ void lds_put16(OC_LDS unsigned char *base, unsigned off, v4u v) { *(OC_LDS v4u *)(base + off) = v; }
__device__ __forceinline__ void order() { asm volatile("" ::: "memory"); }
// streaming accesses: every byte is read or written once, so the COMPRESSING sweep marks them non-temporal (measured, round 4, 512^3
// float: 0.153 -> 0.135 ms; the decompressing sweep reads codes and flags other kernels have just written and lost 0.025 ms with it:
// plain there).  -DSZH_OC_NT=0 gives the plain form everywhere.
#ifndef SZH_OC_NT
#define SZH_OC_NT 1
#endif
#ifndef SZH_OC_NT_DEC_LD
#define SZH_OC_NT_DEC_LD 0
#endif
#ifndef SZH_OC_NT_DEC_ST
#define SZH_OC_NT_DEC_ST 0
#endif
template <bool NT> __device__ __forceinline__ v4u ld_stream(const void *p)
{
#if !defined(SZH_HIPSIM)
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
#endif
    return *reinterpret_cast<const v4u *>(p);
}
template <bool NT> __device__ __forceinline__ void st_stream(void *p, v4u v)
{
#if !defined(SZH_HIPSIM)
    if (NT) { __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(p)); return; }
#endif
    *reinterpret_cast<v4u *>(p) = v;
}
#ifdef SZH_HIPSIM
static inline void pin(unsigned &) {}
#else
__device__ __forceinline__ void pin(unsigned &v) { asm volatile("" : "+v"(v)); }
#endif

// lane masks: which lanes of the wavefront a condition holds in.  On the GPU a compare already IS such a mask (an SGPR pair): AND-ing it
// with a compile-time mask is a scalar instruction, counting its bits too -- work the vector unit, which bounds this kernel, never sees
typedef unsigned long long mask_t;
#ifdef SZH_HIPSIM
static inline mask_t lane_mask(bool p) { return __ballot(p ? 1 : 0); }
static inline bool in_mask(mask_t m) { return (m >> (threadIdx.x & 63u)) & 1ull; }
#else
__device__ __forceinline__ mask_t lane_mask(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool in_mask(mask_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
#endif
// lane l receives v of lane l - 1, lane 0 receives +0 (DPP wave_shr:1 with bound_ctrl: no `old` operand to set up, unlike szh_io::shr1)
#ifdef SZH_HIPSIM
template <class T> static inline T shr1z(T v) { const T s = __shfl_up(v, 1, 64); return (threadIdx.x & 63) == 0 ? (T)0 : s; }
#else
__device__ __forceinline__ float shr1z(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ double shr1z(double v)
{
    const long long s = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)s, 0x138, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp((int)(s >> 32), 0x138, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
#endif
// the lanes whose cell at position U of the unrolled line is REAL: not the virtual one (lane j = U), in the first line not before the
// lane's first cell (j < U), in the last not after its last (j > U); the same for every box of the wavefront
template <int PHASE, int U, int C2> constexpr mask_t keep_mask()
{
    mask_t one = 0;
    for (int j = 0; j < C2; ++j) { const bool k = PHASE == 0 ? U > j : PHASE == 1 ? U != j : U < j; if (k) one |= 1ull << j; }
    mask_t m = 0;
    for (int b = 0; b < 64 / C2; ++b) m |= one << (b * C2);
    return m;
}
template <int C2> constexpr mask_t first_col_mask() { mask_t m = 0; for (int b = 0; b < 64 / C2; ++b) m |= 1ull << (b * C2); return m; }

constexpr int fdiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
constexpr int fmod(int a, int b) { return a - fdiv(a, b) * b; }
template <class F, int... I> __device__ __forceinline__ void for_seq(std::integer_sequence<int, I...>, F &&f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void for_n(F &&f) { for_seq(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f)); }

template <class T> struct ring_slots;
template <> struct ring_slots<float> { static constexpr int RS = 48; };      // 48 x 384 B = 18 KB per wavefront: 8 wavefronts in a CU's 160 KB
template <> struct ring_slots<double> { static constexpr int RS = 44; };     // 44 x 640 B = 27.5 KB: 5

enum { PH_FIRST = 0, PH_MID = 1, PH_LAST = 2 };
#define OC_FLAG_WORDS 64
enum { M_CMP = 0, M_DEC = 1, M_DECV = 2 };     // compress; inverse of boxes without verbatim values; inverse with them (pre-scattered into `out`)

template <class T, int C1, int C2>
struct shape {
    static constexpr int NB = 64 / C2, LINE = C1 + 1, RS = ring_slots<T>::RS;
    static constexpr int ROWB = C2 * (int)sizeof(T), VB = 64 * (int)sizeof(T), CROW = C2 * 2, CB = 128, PITCH = VB + CB;
    static constexpr int RPL = 1024 / VB, EVL = C1 / RPL;      // rows of a box per value event (16 B per lane), value events per line
    static constexpr int RPC = 1024 / CB, ECL = C1 / RPC;      // the same for code events (8 rows)
    static constexpr int DV = 8, DC = ECL;                     // loads in flight: value sets, code sets
    static_assert(64 % C2 == 0 && C1 % RPL == 0 && C1 % RPC == 0 && EVL % DV == 0, "unsupported box shape");
    // Event times relative to the step at which lane 0 enters line L (its virtual cell): row i of line L is cell number s = LINE L + i + 1, lane j is
    // at it at step s + j.  A ring slot holds cell s until cell s + RS takes it.
    //   value rows into the ring (K = 1: the old row was last READ by lane C2 - 1; K = RPL: it was last WRITTEN OUT, inverse with verbatim values)
    static constexpr int dvw(int e, int K) { return RPL * e + RPL + C2 + K - RS; }
    static constexpr int dco(int w) { return RPC * w + RPC + C2; }               // code rows out of the ring (lane C2 - 1 wrote the last one a step ago)
    static constexpr int dcw(int w) { return RPC * w + RPC + C2 + 1 - RS; }      // code rows into the ring (inverse)
    static constexpr int dvo(int e) { return RPL * e + RPL + C2; }               // value rows out of the ring (inverse)
    static_assert(dvw(0, RPL) <= -1 && dcw(0) <= -1, "ring too short: a row must be in the ring before lane 0 reaches it");
    static_assert(RPC + C2 <= RS && dco(ECL - 1) < 2 * LINE && dvo(EVL - 1) < 2 * LINE && dvw(0, 1) >= -LINE && dcw(0) >= -LINE, "event schedule");
};

template <class T> struct sweep_args {
    szh_omp_geom g;
    const T *data; T *out;             // M_CMP: data; inverse: out (M_DECV: holds the verbatim values at their places)
    T eb, recip; int intervals;
    uint16_t *codes;
    unsigned *ucount; szh_u64 *ucount64; T *first;        // M_CMP: written; inverse: `first` read, `ucount` = ONE error counter
    const szh_u64 *uoff;               // inverse: ranks of the boxes' verbatim values (which variant a wavefront runs)
    int dbg_no_code_stores;            // development (timing only, wrong streams): the sweep without its 2 N bytes of code stores
    const unsigned *vflags; int fw;    // inverse: per box `fw` words, bit (row / RPL) = that group of rows holds a verbatim value (k_omp_scatter); fw = 0: none
};

// COUNT (compress): count the verbatim values of every box here (false: the caller takes them from the boxes' histograms, k_omp_hist_box).
// (Round 4 also tried the histograms themselves in here -- the commonest code counted in a register per lane, the others by LDS atomics of
// the lanes that have them: 0.29 ms against 0.16 + 0.08 for the separate pass; the atomics and their exec masks sit in every step.  Removed.)
template <class T, int C1, int C2, int MODE, bool COUNT = true>
struct sweep {
    typedef shape<T, C1, C2> S;
    static constexpr int LINE = S::LINE, RS = S::RS, PITCH = S::PITCH;
    static constexpr bool CMP = MODE == M_CMP, VLOAD = MODE != M_DEC;
    static constexpr bool NT_LD = SZH_OC_NT && (CMP || SZH_OC_NT_DEC_LD), NT_ST = SZH_OC_NT && (CMP || SZH_OC_NT_DEC_ST);
    static constexpr int KV = MODE == M_DECV ? S::RPL : 1;

    OC_LDS unsigned char *ring;
    OC_LDS unsigned *fl;               // M_DECV: the flag words of the wavefront's boxes, [NB][OC_FLAG_WORDS]
    const sweep_args<T> &a;
    int lane, b, j;                    // box of the wavefront, column
    bool use_flags;
    // delay lines, by step number modulo LINE: own results / the left lane's.  TWO of each, read and written in turn line by line (PP): with one
    // array the value a step overwrites is still needed by the next step, and the compiler bridges that with two register copies per step
    // (only where the registers allow two wavefronts per SIMD with them: float compression; elsewhere ONE array and the copies)
    static constexpr bool PPONG = CMP && sizeof(T) == 4;
    T dl[PPONG ? 2 : 1][LINE], lup[PPONG ? 2 : 1][LINE];
    T prev, Lprev, Bold, Bpold;
    T first_v;
    unsigned nun;                      // M_CMP (COUNT): verbatim values of this column; inverse: zero codes met without a verbatim value
    unsigned vaddr, cdelta;            // LDS: the lane's value in the slot it is at; from there to its code
    v4u gv[S::DV], gc[S::DC], wq;      // loads in flight; rows on their way out
    T cur_next; unsigned tc_next;      // the next step's value / code, read from the ring a step ahead
    const unsigned char *vsrc;         // value events: the lane's piece of row 0 of line 0 of its box (global)
    unsigned char *vdst;
    unsigned char *cptr;               // code events: the lane's piece of row 0 of its box's codes
    unsigned vev_lds, cev_lds, ev_r, cev_r;   // lane parts of the LDS addresses of the events, the lane's row within an event
    int64_t line_bytes, row_bytes;     // one line (k) / one row (i) further in the array, in bytes

    __device__ __forceinline__ sweep(OC_LDS unsigned char *r, OC_LDS unsigned *f, const sweep_args<T> &args) : ring(r), fl(f), a(args) {}

    // the slot of cell s, s = LINE * L + i1 with i1 < LINE + 16 given as (uniform) `ls` = (LINE * L) mod RS plus a lane-dependent i1
    __device__ __forceinline__ unsigned slot_off(int ls, unsigned i1) const
    {
        unsigned x = (unsigned)ls + i1;                  // < 2 RS + 16 (RS > LINE)
        x -= x >= (unsigned)(2 * RS) ? (unsigned)(2 * RS) : x >= (unsigned)RS ? (unsigned)RS : 0u;
        return x * (unsigned)PITCH;
    }
    // the value part of the slot of a line's virtual cell (`ls` = (LINE * L) mod RS): zeros, 16 bytes per lane
    __device__ __forceinline__ void zero_virtual(int ls)
    {
        const v4u z = {0u, 0u, 0u, 0u};
        if (lane < S::VB / 16) lds_put16(ring, (unsigned)ls * (unsigned)PITCH + (unsigned)lane * 16u, z);
    }
    __device__ __forceinline__ int clampL(int L) const { return L < a.g.c0 ? L : a.g.c0 - 1; }
    __device__ __forceinline__ v4u load_vrows(int L, int e) const
    {
        const int Lc = clampL(L);
        const unsigned char *p = (CMP ? vsrc : (const unsigned char *)vdst) + (int64_t)Lc * line_bytes + (int64_t)(S::RPL * e) * row_bytes;
        if (MODE == M_DECV && use_flags) {
            // rows without a verbatim value are never looked at: their load goes to one line every lane shares (the load itself stays --
            // a load that may or may not happen would cost the compiler its count of the accesses in flight)
            const unsigned bit = (unsigned)(Lc * S::EVL + e);
            const unsigned wv = fl[b * OC_FLAG_WORDS + (bit >> 5)];
            p = (wv >> (bit & 31u) & 1u) ? p : (const unsigned char *)vdst;
        }
        return ld_stream<NT_LD>(p);
    }
    __device__ __forceinline__ v4u load_crows(int L, int w) const
    {
        return ld_stream<NT_LD>(cptr + ((int64_t)clampL(L) * C1 + S::RPC * w) * S::CROW);
    }
    // ---- the events of step (it, U): `ls_*`: (LINE * L) mod RS for L = it - 1, it, it + 1
    template <int U, int PHASE> __device__ __forceinline__ void events(int it, int ls_m1, int ls_0, int ls_p1)
    {
        auto ls_of = [&](int loff) { return loff < 0 ? ls_m1 : loff == 0 ? ls_0 : ls_p1; };
        // rows that are complete leave the ring first: read from the ring at their time (into `wq`), stored to memory one step later (an LDS
        // read and the store that needs it in the same step is a wait for the LDS in front of the store)
        if (CMP) for_n<S::ECL>([&](auto W) {
            constexpr int w = decltype(W)::value, d = S::dco(w);
            static_assert(fdiv(d, LINE) == 1 && fdiv(d + 1, LINE) == 1, "a finished row leaves during the next iteration");
            if constexpr (PHASE != PH_FIRST && fmod(d + 1, LINE) == U)
                if (!a.dbg_no_code_stores) st_stream<NT_ST>(cptr + ((int64_t)(it - 1) * C1 + S::RPC * w) * S::CROW, wq);
            if constexpr (PHASE != PH_FIRST && fmod(d, LINE) == U) {
                constexpr int loff = -fdiv(d, LINE);
                wq = lds_get16(ring, slot_off(ls_of(loff), (unsigned)(S::RPC * w + 1) + cev_r) + cev_lds);
            }
        });
        if (!CMP) for_n<S::EVL>([&](auto E) {
            constexpr int e = decltype(E)::value, d = S::dvo(e);
            static_assert(fdiv(d, LINE) == 1 && fdiv(d + 1, LINE) == 1, "a finished row leaves during the next iteration");
            if constexpr (PHASE != PH_FIRST && fmod(d + 1, LINE) == U)
                st_stream<NT_ST>(vdst + (int64_t)(it - 1) * line_bytes + (int64_t)(S::RPL * e) * row_bytes, wq);
            if constexpr (PHASE != PH_FIRST && fmod(d, LINE) == U) {
                constexpr int loff = -fdiv(d, LINE);
                wq = lds_get16(ring, slot_off(ls_of(loff), (unsigned)(S::RPL * e + 1) + ev_r) + vev_lds);
            }
        });
        // ... then rows that have arrived go in, and the register set that carried them is sent for the rows DV (DC) events further on
        if (VLOAD && PHASE != PH_LAST) for_n<S::EVL>([&](auto E) {
            constexpr int e = decltype(E)::value, d = S::dvw(e, KV);
            if constexpr (fmod(d, LINE) == U) {
                constexpr int loff = -fdiv(d, LINE);                 // 0 or +1
                lds_put16(ring, slot_off(ls_of(loff), (unsigned)(S::RPL * e + 1) + ev_r) + vev_lds, gv[e % S::DV]);
                if constexpr (e == 0) zero_virtual(ls_of(loff));     // (with the line's first rows: its virtual cell reads +0)
                constexpr int en = (e + S::DV) % S::EVL, ln = (e + S::DV) / S::EVL;
                gv[e % S::DV] = load_vrows(it + loff + ln, en);
            }
        });
        if (!CMP && PHASE != PH_LAST) for_n<S::ECL>([&](auto W) {
            constexpr int w = decltype(W)::value, d = S::dcw(w);
            if constexpr (fmod(d, LINE) == U) {
                constexpr int loff = -fdiv(d, LINE);
                lds_put16(ring, slot_off(ls_of(loff), (unsigned)(S::RPC * w + 1) + cev_r) + cev_lds, gc[w % S::DC]);
                gc[w % S::DC] = load_crows(it + loff + 1, w);
            }
        });
    }

    __device__ __forceinline__ static T tabs(T v) { return sizeof(T) == 8 ? (T)__builtin_fabs((double)v) : (T)__builtin_fabsf((float)v); }

    __device__ __forceinline__ static T tsign(T mag, T from) { return sizeof(T) == 8 ? (T)__builtin_copysign((double)mag, (double)from) : (T)__builtin_copysignf((float)mag, (float)from); }

    // ---- one step: the cell the lane is at (position U - j of the line, modulo LINE)
    template <int U, int PHASE, int PP> __device__ __forceinline__ void step(int radius, T fint)
    {
        constexpr mask_t KM = keep_mask<PHASE, U, C2>();            // the lanes whose cell is real
        // what the NEXT step needs from the ring is requested now (a read and its use in the same step: a wait for the LDS in every step)
        const unsigned y = vaddr + (unsigned)PITCH, y2 = y - (unsigned)(RS * PITCH);
        const unsigned vnext = y < y2 ? y : y2;
        const T cur_raw = cur_next;
        const unsigned tc_in = tc_next;
        if (CMP || MODE == M_DECV) cur_next = lds_get<T>(ring, vnext);
        if (!CMP) tc_next = lds_get<uint16_t>(ring, vnext + cdelta);
        // (a virtual cell reads +0: the loader zeroes its slot.  Only in the first line are there lanes that have not started: forced)
        const T cur = PHASE == PH_FIRST ? (in_mask(KM) ? cur_raw : (T)0) : cur_raw;
        const T Lraw = shr1z(prev);
        const T L = S::NB > 1 ? (in_mask(first_col_mask<C2>()) ? (T)0 : Lraw) : Lraw;
        // (k-1, i, *) and (k-1, i-1, *): what this lane / the left one had at this position and at the one before, a line ago (position -1: the
        // last of the line before that one, which is the OTHER array's last entry -- written one line earlier still, i.e. by this array's turn)
        constexpr int RD = PPONG ? PP : 0, WR = PPONG ? PP ^ 1 : 0;
        const T B = dl[RD][U], Bp = lup[RD][U];
        const T C = !PPONG ? Bold : U > 0 ? dl[RD][U > 0 ? U - 1 : 0] : dl[WR][LINE - 1], Cp = !PPONG ? Bpold : U > 0 ? lup[RD][U > 0 ? U - 1 : 0] : lup[WR][LINE - 1];
        T pred = L + prev + B - Lprev - C - Bp + Cp;                // sz_float.c:4939-4943: the 7-point sum, left to right
        if (PHASE == PH_FIRST) {
            // row (0, 0, :) (sz_float.c:4738-4799): the box's first value, then its left neighbour, then 2 left - left-left
            const T LL = shr1z(Lprev);
            if (U == j + 1) pred = j == 0 ? first_v : j == 1 ? L : 2 * L - LL;
        }
        T rec;
        if (CMP) {
            // sz_float.c:4762-4783: itv = |diff| / eb + 1 against the interval count, negated for a negative diff; code = (int)(itv / 2) + radius;
            // reconstruction pred + 2 (code - radius) eb, verified against the bound.  (The sign comes from diff's sign BIT: for diff = -0 --
            // where the reference's `diff < 0` is false -- itv is -1, (int)(-0.5) is 0 as for +1: the same code, the same +0 added.)
            const T diff = cur - pred;
            const T mag = tabs(diff) * a.recip + 1;
            const bool in_range = mag < fint;                         // (false for a NaN)
            const int q = (int)(tsign(mag, diff) / 2);                // (out of range: never used)
            const T r = pred + (T)(2 * q) * a.eb;
            const T err = cur - r;
            const mask_t okm = lane_mask(in_range) & lane_mask(!(tabs(err) > a.eb)) & KM;     // (two compares, two scalar ANDs: a `&&` of them goes through a vector register)
            const bool ok = in_mask(okm);
            const int tc = ok ? q + radius : 0;
            rec = ok ? r : cur;
            if (COUNT) { nun += in_mask(~okm & KM) ? 1u : 0u; pin(nun); }     // real cells kept verbatim (per lane: scalar pop-counts here cost 200 spilled SGPRs)
            lds_put<uint16_t>(ring, vaddr + cdelta, (uint16_t)tc);
        } else {
            const T m = (T)(2 * ((int)tc_in - radius)) * a.eb;        // szd_float.c: pred + 2 (type - radius) eb
            const T r = pred + m;
            const T verb = MODE == M_DECV ? cur : (T)0;
            const mask_t cm = lane_mask(tc_in != 0);
            rec = in_mask(cm & KM) ? r : verb;
            if (MODE == M_DEC) nun += in_mask(~cm & KM) ? 1u : 0u;    // a zero code in a box whose table entry says "no verbatim values"
            if (PHASE == PH_FIRST && MODE != M_DECV) rec = in_mask(KM) ? rec : (T)0;     // (M_DECV: `verb` is already +0 there)
            lds_put<T>(ring, vaddr, rec);
        }
        dl[WR][U] = rec; lup[WR][U] = L;
        if (!PPONG) { Bold = B; Bpold = Bp; }
        Lprev = L; prev = rec;
        vaddr = vnext;
    }

    template <int PHASE, int PP> __device__ __forceinline__ void line(int it, int ls_m1, int ls_0, int ls_p1, int radius, T fint)
    {
        for_n<LINE>([&](auto UU) {
            constexpr int U = decltype(UU)::value;
            wave_sync();
            events<U, PHASE>(it, ls_m1, ls_0, ls_p1);
            order();
            step<U, PHASE, PP>(radius, fint);
            order();
        });
    }

    __device__ __forceinline__ void run()
    {
        const szh_omp_geom &g = a.g;
        lane = (int)(threadIdx.x & 63u); b = lane / C2; j = lane - b * C2;
        const int box = (int)blockIdx.x * S::NB + b;
        line_bytes = g.d0 * (int64_t)sizeof(T); row_bytes = g.d1 * (int64_t)sizeof(T);
        const unsigned char *origin = (const unsigned char *)szh_omp_box_origin_bytes(g, box, sizeof(T), CMP ? (const void *)a.data : (const void *)a.out);
        {   // value events: 64 lanes x 16 B = NB boxes x RPL rows x ROWB bytes
            const int q = lane - b * (64 / S::NB), r = q / (S::ROWB / 16), c = q - r * (S::ROWB / 16);
            ev_r = (unsigned)r; vev_lds = (unsigned)(b * S::ROWB + c * 16);
            vsrc = origin + (int64_t)r * row_bytes + c * 16; vdst = const_cast<unsigned char *>(vsrc);
            const int rc = q / (S::CROW / 16), pc = q - rc * (S::CROW / 16);
            cev_r = (unsigned)rc; cev_lds = (unsigned)(S::VB + b * S::CROW + pc * 16);
            cptr = (unsigned char *)a.codes + ((int64_t)box * g.bel) * 2 + (int64_t)q * 16;
        }
        use_flags = false;
        if (MODE == M_DECV) {
            use_flags = a.vflags && a.fw > 0 && a.fw <= OC_FLAG_WORDS;          // (uniform)
            if (use_flags) {
                for (int i = lane; i < S::NB * a.fw; i += 64) { const int bb = i / a.fw, w = i - bb * a.fw; fl[bb * OC_FLAG_WORDS + w] = a.vflags[(int64_t)((int)blockIdx.x * S::NB + bb) * a.fw + w]; }
                wave_sync();
            }
        }
        for_n<LINE>([&](auto UU) { constexpr int U = decltype(UU)::value; dl[0][U] = 0; lup[0][U] = 0; if (PPONG) { dl[PPONG ? 1 : 0][U] = 0; lup[PPONG ? 1 : 0][U] = 0; } });
        prev = 0; Lprev = 0; Bold = 0; Bpold = 0; nun = 0;
        first_v = CMP ? *reinterpret_cast<const T *>(origin) : a.first[box];
        vaddr = (unsigned)(((RS - j) % RS) * PITCH + b * S::ROWB + j * (int)sizeof(T));
        cdelta = (unsigned)(S::VB + b * S::CROW + j * 2) - (unsigned)(b * S::ROWB + j * (int)sizeof(T));
        const int radius = a.intervals / 2;
        const T fint = (T)a.intervals;
        // ---- prologue: the first DV (DC) events' rows are requested; those whose time has come before step 0 go into the ring
        if (VLOAD) for_n<S::DV>([&](auto E) { constexpr int e = decltype(E)::value; gv[e] = load_vrows(e / S::EVL, e % S::EVL); });
        if (!CMP) for_n<S::DC>([&](auto W) { constexpr int w = decltype(W)::value; gc[w] = load_crows(0, w); });
        if (VLOAD) for_n<S::EVL>([&](auto E) {
            constexpr int e = decltype(E)::value;
            if constexpr (S::dvw(e, KV) < 0) {
                lds_put16(ring, slot_off(0, (unsigned)(S::RPL * e + 1) + ev_r) + vev_lds, gv[e % S::DV]);
                if constexpr (e == 0) zero_virtual(0);
                gv[e % S::DV] = load_vrows((e + S::DV) / S::EVL, (e + S::DV) % S::EVL);
            }
        });
        if (!CMP) for_n<S::ECL>([&](auto W) {
            constexpr int w = decltype(W)::value;
            if constexpr (S::dcw(w) < 0) { lds_put16(ring, slot_off(0, (unsigned)(S::RPC * w + 1) + cev_r) + cev_lds, gc[w % S::DC]); gc[w % S::DC] = load_crows(1, w); }
        });
        order();
        cur_next = (CMP || MODE == M_DECV) ? lds_get<T>(ring, vaddr) : (T)0;
        tc_next = CMP ? 0u : (unsigned)lds_get<uint16_t>(ring, vaddr + cdelta);
        { const v4u z = {0u, 0u, 0u, 0u}; wq = z; }
        // ---- the lines: lane 0 is on line `it`, lane j on it or on the one before
        int ls_m1 = (RS - LINE % RS) % RS, ls_0 = 0, ls_p1 = LINE % RS;
        auto next = [&]() { ls_m1 = ls_0; ls_0 = ls_p1; ls_p1 = (ls_p1 + LINE) % RS; };
        line<PH_FIRST, 0>(0, ls_m1, ls_0, ls_p1, radius, fint);
        int it = 1;
        for (; it + 1 < g.c0; it += 2) {                           // lines in pairs: the delay-line arrays change roles line by line
            next(); line<PH_MID, 1>(it, ls_m1, ls_0, ls_p1, radius, fint);
            next(); line<PH_MID, 0>(it + 1, ls_m1, ls_0, ls_p1, radius, fint);
        }
        if (it < g.c0) {
            next(); line<PH_MID, 1>(it, ls_m1, ls_0, ls_p1, radius, fint);
            next(); line<PH_LAST, 0>(g.c0, ls_m1, ls_0, ls_p1, radius, fint);
        } else { next(); line<PH_LAST, 1>(g.c0, ls_m1, ls_0, ls_p1, radius, fint); }
        // ---- per box: the count of verbatim values, the first value (sz_omp.c:246-262 writes both tables into the stream)
        unsigned tot = nun;
        for (int m = 1; m < C2; m <<= 1) tot += __shfl_xor(tot, m, 64);
        if (CMP) { if (j == 0) { if (COUNT) { a.ucount[box] = tot; a.ucount64[box] = tot; } a.first[box] = first_v; } }
    }
};

} // namespace szh_oc

// One wavefront per 64 / C2 boxes.  DEC: the wavefront looks its boxes up in `uoff` and runs the variant with or without verbatim values.
template <class T, int C1, int C2, bool DEC, bool COUNT = true>
__global__ __launch_bounds__(64, (sizeof(T) == 4 ? 2 : 1)) void k_omp_col(szh_oc::sweep_args<T> a)      // (float: two wavefronts per SIMD -- the register budget the compiler must keep)
{
    typedef szh_oc::shape<T, C1, C2> S;
    __shared__ __attribute__((aligned(16))) unsigned char ring_raw[S::RS * S::PITCH];
    __shared__ unsigned flags_raw[DEC ? S::NB * OC_FLAG_WORDS : 1];
    OC_LDS unsigned char *ring = (OC_LDS unsigned char *)ring_raw;
    OC_LDS unsigned *fl = (OC_LDS unsigned *)flags_raw;
    if (!DEC) { szh_oc::sweep<T, C1, C2, szh_oc::M_CMP, COUNT> s(ring, fl, a); s.run(); }
    else {
        const int b0 = (int)blockIdx.x * S::NB;
        const bool verb = a.uoff[b0 + S::NB] != a.uoff[b0];          // (uniform)
        if (verb) { szh_oc::sweep<T, C1, C2, szh_oc::M_DECV> s(ring, fl, a); s.run(); }
        else { szh_oc::sweep<T, C1, C2, szh_oc::M_DEC> s(ring, fl, a); s.run(); }
    }
}

// inverse, before the sweep: the verbatim values of a box go to their places in `out` (the sweep reads them there when it meets a zero
// code); boxes without any are skipped (the sweep counts zero codes in those).  `bad`: boxes whose zero codes and table entry disagree
// `vflags` (or null): per box `fw` words (zeroed by the caller), bit g = rows [g RPL, (g + 1) RPL) of the box, RPL = 16 / sizeof(T), hold a verbatim
// value -- the only rows of `out` the sweep has to read
template <class T>
__global__ __launch_bounds__(256) void k_omp_scatter(szh_omp_geom g, const uint16_t *__restrict__ codes, const u64 *__restrict__ uoff,
                                                     const T *__restrict__ unpred, T *__restrict__ out, unsigned *bad, unsigned *vflags, int fw)
{
    __shared__ u64 sh[8];
    __shared__ unsigned lflags[OC_FLAG_WORDS];
    const bool flags = vflags && fw > 0 && fw <= OC_FLAG_WORDS;
    if (flags) { for (int i = threadIdx.x; i < fw; i += 256) lflags[i] = 0; __syncthreads(); }
    const int b = blockIdx.x;
    const u64 cap = uoff[b + 1] - uoff[b];
    if (cap == 0) return;                                     // uniform
    T *box = reinterpret_cast<T *>(const_cast<void *>(szh_omp_box_origin_bytes(g, b, sizeof(T), out)));
    const uint16_t *cb = codes + (int64_t)b * g.bel;
    const T *src = unpred + uoff[b];
    u64 done = 0;
    const bool aligned = (g.bel & 7) == 0;
    uint16_t cn[8];
    omp_load8(cb, (int)threadIdx.x * 8, g.bel, aligned, cn);
    for (int base = 0; base < g.bel; base += 256 * 8) {
        const int p0 = base + (int)threadIdx.x * 8;
        unsigned mask = 0;
        for (int e = 0; e < 8; ++e) if (p0 + e < g.bel && cn[e] == 0) mask |= 1u << e;
        if (base + 256 * 8 < g.bel) omp_load8(cb, p0 + 256 * 8, g.bel, aligned, cn);      // (the next round's codes are on their way during the scan)
        u64 tot;
        u64 rank = done + block_excl_scan_256((u64)__builtin_popcount(mask), sh, &tot);
        for (int e = 0; e < 8; ++e) if (mask >> e & 1u) {
            const int p = p0 + e, k = p / (g.c1 * g.c2), r = p - k * (g.c1 * g.c2), i = r / g.c2, jj = r - i * g.c2;
            box[(int64_t)k * g.d0 + (int64_t)i * g.d1 + jj] = rank < cap ? src[rank] : (T)0;
            ++rank;
            if (flags) { const unsigned grp = (unsigned)(k * g.c1 + i) / (unsigned)(16 / sizeof(T)); if (grp < (unsigned)fw * 32u) atomicOr(&lflags[grp >> 5], 1u << (grp & 31u)); }
        }
        done += tot;
    }
    if (flags) { __syncthreads(); for (int i = threadIdx.x; i < fw; i += 256) vflags[(int64_t)b * fw + i] = lflags[i]; }
    if (threadIdx.x == 0 && done != cap) atomicAdd(bad, 1u);
}
