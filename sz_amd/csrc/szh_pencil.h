// szh_pencil.h -- the predict+quantise (and inverse) wavefront kernel body of the SZ 2.1 3-D path.
//
// What it computes: for every point, the 7-point Lorenzo prediction from RECONSTRUCTED neighbours
// (sz/src/sz_float.c:7253-7353; with mean: :6914-7030) or the block's regression plane
// (:7153-7252), the quantisation code and the reconstruction; and the inverse
// (sz/src/szd_float.c:3483-5866).  The reference does this strictly sequentially because every
// Lorenzo point depends on its reconstructed -1 neighbours in all three dimensions.
//
// How it is parallelised (MI355X-native, bit-identical results): the only exact parallel schedule of
// that recurrence is the i+j+k hyperplane wavefront.  One 64-lane wavefront owns a "pencil": an
// 8x8 cross-section in (dim0,dim1), swept along dim2 (the fastest, contiguous dimension).  Lane
// (il,jl) handles k = t - il - jl at step t, so its three face neighbours were produced by lanes
// l-1 / l-8 / l-9 one or two steps earlier and travel by cross-lane moves.  Pencil (I,J) needs the dim1-face of (I,J-1)
// and the dim0-face of (I-1,J), delayed by 7 steps: its step t reads what its producers finished in their step t+7.
//
// Tiles.  A workgroup owns a TILE of TPI x TPJ neighbouring pencils, one COMPUTE wavefront per pencil, plus two helper
// wavefronts.  Every hand-off a compute wavefront takes part in is LDS only: it reads its producers' face rows from LDS
// rings (waiting on an LDS step counter) and writes its own faces into its ring.  Faces between pencils of the same tile
// never leave the CU.  Faces that cross a tile boundary travel as granules, moved by the helpers:
//   STORE  forwards the rings of the tile's last row / last column of pencils to the granule buffers (stores only);
//   FILL   follows the progress words of the producers in the neighbouring tiles, fetches their granules and drops them
//          into "virtual producer" rings of this tile (loads only).
// On gfx950 loads and stores of a wavefront share one in-order memory queue and agent-scope stores are acknowledged
// slowly, so a wavefront that both computes and talks to the fabric pays a fabric round trip per step; the split keeps
// the sweep at LDS latency.  The third producer of a pencil, (I-1,J-1), is not needed: its corner column is the row-7
// halo of (I-1,J), which forwards it as a ninth row of its I-face.
//
// The body is written once and instantiated by two back ends:
//   * the HIP kernel (szhip_kernels.h): NL = 1 value per thread, collectives = bpermute / readlane;
//   * a CPU lane simulator used ONLY by tests/ (tests/sim): NL = 64, collectives = array moves.
#pragma once
#include "szh_geom.h"

typedef unsigned long long szh_u64;

template <class T> struct szh_qargs {
    szh_geom3 G;
    const T *data;            // compress: original values
    T *out;                   // decompress: output (unpredictable values pre-scattered)
    uint16_t *codes;          // NATURAL-order codes: written by compress, read by decompress
    const uint8_t *blk_lor;   // per block: 1 = Lorenzo, 0 = regression (the stream's indicator)
    const T *coef;            // decoded regression coefficients, SoA [4][coef_stride]
    int64_t coef_stride;      // nblocks, or nblocks rounded up to whole 128-byte lines when the coefficients arrive during the launch
    T eb, recip, mean;
    int cap, radius, use_mean;
    szh_u64 *faceI, *faceJ;   // granule buffers: faceJ [pencil][8 rows][r2][NW], faceI [pencil][9 rows][r2][NW] (row 8 = forwarded corner column)
    unsigned epoch;
    unsigned *tile_done;      // k_ribbon, compress, optional (host-coherent memory): [tile] <- epoch once every code of the tile is in device memory
                              // (the host starts the entropy stage's passes over finished tile rows while the sweep is still running)
    int nI, nJ;
    const unsigned *order;    // ticket -> (tile row << 16) | tile column, anti-diagonal order over the tiles
    unsigned *ticket;
    int no_reg;               // 1: the caller knows that no block is a regression block (the per-pencil scan of blk_lor in front of the sweep is skipped)
    int ticket_mode;          // 0: atomic ticket + order table; 1: blockIdx.x as the ticket (+ table); 2: blockIdx.x and the tile computed (szh_pencil_order_at)
    int persist;              // k_pencil: persistent workgroups draw tile after tile from the ticket counter (gridDim.x < number of tiles)
    int codes_ribbon;         // (unused since round 6: the ribbon mapping is gone)
    unsigned *err;            // set to 1 if a halo wait timed out
    const szh_u64 *coef_progress; // compress, optional: number of blocks (scan order) whose decoded coefficients have arrived in `coef`;
                              // the host's coefficient chain runs NEXT TO this kernel and ships them as it goes (nullptr: all there)
    szh_u64 *progress;        // [pencil][2] (J-face, I-face): {epoch, steps whose face values have been published}: the consumers' FILL
                              // wavefront polls these words and then fetches only granules that exist
    int fmt;                  // 0: SZ 2.1 block path; 1: SZ 1.4 whole-array Lorenzo (sz_float.c:946): no blocks, capacity = intervals,
                              //    interval number through double, lossy "exact" values, second-order predictor on the first row
                              // 2: the table-driven point-wise-relative form ("MSST19", sz_float.c:2270-2730): multiplicative Lorenzo predictor,
                              //    code = table[exponent, leading mantissa bits of x / prediction], reconstruction |prediction| * ptab[code]
    T median;                 // fmt 1: exact values are kept as reqLength leading bits of (x - median)
    int ign_bits;             // fmt 1, 2: 8*sizeof(T) - reqLength, >= 0
    const double *ptab;       // fmt 2: precisionTable[cap]
    const uint16_t *cells;    // fmt 2, compress: [(trange + 1) << tbits] (MultiLevelCacheTableWideInterval.c:53-107, built on the host)
    szh_u64 tbase, trange;    // fmt 2: exponent field of the first sub-table, number of sub-tables - 1
    int tbits;                // fmt 2: mantissa bits per sub-table
    int f32arith;             // fmt 2: 1 = the float 2-D quantiser, whose products are float (sz_float.c:2091, :2152); else products in double
    int ndim3;                // fmt 2: 1 = a 3-D array (one boundary case of ITS compressor has no fabs, sz_float.c:2459)
    int backoff;              // FILL wavefront: sleep units between two rounds that delivered nothing
    int wide;                 // 1: the granule rows of a tile lie within 4 GB of the tile's first row: 16-byte buffer accesses with 32-bit offsets
    int trace_tile;           // development: (TI << 16) | TJ of the tile whose left-hand hand-off is logged round by round (SZH_TRACE_LOG)
    const T *xr;              // k_beam, compress, arrays with regression blocks: the RECONSTRUCTIONS of their points (k_reg_points), natural layout
    int pub_lines;            // k_beam with tile_done: lines between two progress words of a wavefront (a multiple of 8; 0: 32)
    const uint8_t *ptflags;   // k_beam, arrays with regression blocks: per point, 1 = the point lies in a regression block (k_reg_points; zeros elsewhere)
    const unsigned *reg_ready; // k_beam, when its inputs are made in slices beside the sweep (compress: k_reg_points; decompress: k_permute<1> + k_unpred): planes whose inputs are in memory (nullptr: all, before the launch)
    int dbg;                  // development: 1 = no hand-off at all (timing only, results become WRONG), 4 = no issue priorities
    szh_u64 *trace;           // optional (development): per pencil {t_start, t_start, t_first_trip, t_end, wait spins, -, xcc, 0} + per-trip detail
};

template <class T> struct szh_gran;
template <> struct szh_gran<float> {
    static constexpr int NW = 1;
    SZH_HD static void pack(float v, unsigned epoch, szh_u64 *w)
    {
        unsigned u; __builtin_memcpy(&u, &v, 4);
        w[0] = ((szh_u64)epoch << 32) | u;
    }
    SZH_HD static float unpack(const szh_u64 *w)
    {
        unsigned u = (unsigned)w[0]; float v; __builtin_memcpy(&v, &u, 4); return v;
    }
};
template <> struct szh_gran<double> {
    static constexpr int NW = 2;
    SZH_HD static void pack(double v, unsigned epoch, szh_u64 *w)
    {
        szh_u64 u; __builtin_memcpy(&u, &v, 8);
        w[0] = ((szh_u64)epoch << 32) | (u & 0xffffffffull);
        w[1] = ((szh_u64)epoch << 32) | (u >> 32);
    }
    SZH_HD static double unpack(const szh_u64 *w)
    {
        szh_u64 u = (w[0] & 0xffffffffull) | (w[1] << 32); double v; __builtin_memcpy(&v, &u, 8); return v;
    }
};

// LDS of one tile (the workgroup).  Slots 0 .. NP-1 are the tile's pencils (pi * TPJ + pj); slots NP .. NP+TPI-1 are the
// "virtual producers" left of the tile (J-faces of pencils (I, J0-1)), slots NP+TPI .. NP+TPI+TPJ-1 those above it.
// B::ring(k) maps a column to its ring slot.
template <class T> struct szh_tile_lds {
    uint16_t *cring;          // [NP][(SZH_XC + 1)][64] quantisation codes in flight per pencil (+ trash column)
    T *faces;                 // [NP + NV][SZH_FROWS][row stride] face rings: rows 0-7 = J-face (il), 8-15 = I-face (jl), 16 = corner column;
                              // the row stride (RL + 2) spreads the skewed columns of neighbouring rows over different LDS banks
                              // forwarded to the pencil below
    int ftrash;               // element offset in faces[] of 64 x (SZH_U + 1) write-only slots (SZH_U + 1 per lane): lanes without a face row
                              // store there, so that the ring accesses need no exec-mask juggling (SZH_FTRASH elements)
    unsigned *cstep;          // [NP + NV] steps completed (all face values of those steps are in the ring); virtual: same scale, by FILL
    unsigned *spubJ, *spubI;  // [NP] columns of the J- / I-face already forwarded by STORE (ring space for the producer)
    int *scratch;             // [2][64] helper wavefronts (STORE, FILL): per-row values for a per-slot minimum
};
#define SZH_FROWS 17

// development plumbing (timing-only "free run", per-pencil trace stamps) is compiled out of the production library; `make dev`
// builds libszhip_dev.so with it (SZ_HIP_DBG / SZ_HIP_TRACE, tools/gpu_trace.py)
#ifndef SZH_DEV
#define SZH_DEV 0
#endif
#define SZH_TRACE_LOG 2048   /* entries {clock, value} per hand-off log (development): producer steps, STORE rounds, FILL rounds, consumer steps */

#define SZH_U 16 /* steps per loop trip: a trip first requests all of its inputs (values and halo granules), then steps */
#if defined(__HIPCC__)
#define SZH_UNROLL _Pragma("unroll")
#else
#define SZH_UNROLL
#endif
#define SZH_FORL for (int l = 0; l < NL; ++l)
#define SZH_FTRASH (64 * (SZH_U + 1))
#define SZH_XC 32 /* columns of the per-pencil LDS code ring ([column % SZH_XC][lane]); column SZH_XC is a write-only trash column */

// branch-free form of szh_quant_point (same arithmetic, same results)
template <class T>
SZH_HD int szh_quant_sel(T x, T pred, T eb, T recip, int capacity, int radius, T *recon)
{
    const T diff = x - pred;
    T itv = szh_abs(diff) * recip + 1;
    const bool inr = itv < (T)capacity;
    itv = inr ? itv : (T)0;                 // keeps the int conversion in range on every back end
    const T sitv = diff < 0 ? -itv : itv;
    const int q = (int)(sitv / 2);
    const T rc = pred + (T)(2 * q) * eb;
    const bool ok = inr && !(szh_abs(x - rc) > eb);
    *recon = ok ? rc : x;
    return ok ? q + radius : 0;
}

// SZ 1.4 flavour (sz_float.c:1040-1069): the interval number goes through double (`fabs`), the capacity is the full interval
// count, and an unpredictable value is reconstructed as its reqLength leading bits against the median (dataCompression.c:454-477)
SZH_HD float szh_keep_bits(float x, float median, int ign)
{
    const float norm = x - median;
    int32_t s; __builtin_memcpy(&s, &norm, 4);
    s = (int32_t)((uint32_t)(s >> ign) << ign);
    float kept; __builtin_memcpy(&kept, &s, 4);
    return kept + median;
}
SZH_HD double szh_keep_bits(double x, double median, int ign)
{
    const double norm = x - median;
    int64_t s; __builtin_memcpy(&s, &norm, 8);
    s = (int64_t)((uint64_t)(s >> ign) << ign);
    double kept; __builtin_memcpy(&kept, &s, 8);
    return kept + median;
}
template <class T>
SZH_HD int szh_quant_sel14(T x, T pred, T eb, T recip, int capacity, int radius, T median, int ign, bool force_exact, T *recon)
{
    const T diff = x - pred;
    T itv = (T)((double)szh_abs(diff) * (double)recip + 1.0);
    const bool inr = itv < (T)capacity;
    itv = inr ? itv : (T)0;
    const T sitv = diff < 0 ? -itv : itv;
    const int q = (int)(sitv / 2);
    const T rc = pred + (T)(2 * q) * eb;
    const bool ok = inr && !(szh_abs(x - rc) > eb) && !force_exact;
    *recon = ok ? rc : szh_keep_bits(x, median, ign);
    return ok ? q + radius : 0;
}

// the MSST19 flavour: an exact value is the leading bits of the value itself (compressSingleFloatValue_MSST19, dataCompression.c:479-501)
SZH_HD float szh_keep_bits_msst(float x, int ign)
{
    int32_t s; __builtin_memcpy(&s, &x, 4);
    s = (int32_t)((uint32_t)(s >> ign) << ign);
    float kept; __builtin_memcpy(&kept, &s, 4);
    return kept;
}
SZH_HD double szh_keep_bits_msst(double x, int ign)
{
    int64_t s; __builtin_memcpy(&s, &x, 8);
    s = (int64_t)((uint64_t)(s >> ign) << ign);
    double kept; __builtin_memcpy(&kept, &s, 8);
    return kept;
}
// the table look-up of the MSST19 quantisers (sz_float.c:2409-2415): 0 = no entry (the value is stored "exactly")
template <class T>
SZH_HD int szh_msst_state(const szh_qargs<T> &a, double quotient)
{
    szh_u64 u; __builtin_memcpy(&u, &quotient, 8);
    const szh_u64 e = ((u & 0x7fffffffffffffffull) >> 52) - a.tbase;
    if (e > a.trange) return 0;
    return (int)a.cells[(size_t)(e << a.tbits) + (size_t)((u & 0x000fffffffffffffull) >> (52 - a.tbits))];
}

// B: back end. Requires: NL, lane(l), shfl_up(dst,src,d), shfl_up1(dst,src) (d = 1; lanes with lane % 8 == 0 may receive anything),
//    readlane(src,lane), all(pred), lds_order(), touch(v) (the value must be in its register here),
//    ld_gran(p), st_gran(p,v), ld_flag(p), st_flag(p,v), backoff(n), nap(), clock(), where(),
//    gbuf_t, make_gbuf(base), ld_gran2_b(buf, byte offset, a, b), st_gran2_b(...): two granules in one 16-byte access relative to a wavefront-uniform base,
//    ld16(p, T(&)[16/sizeof T]), st16(p, const T(&)[...]) -- one 16-byte vector access (4-byte aligned for 4/8-byte T),
//    lds_ld(p), lds_st(p,v), lds_ld_u(p) (wavefront-uniform address), lds_fence() (orders this wavefront's LDS accesses),
//    RL / ring(k) (face-ring length), face_stride(r2) (elements per ring), TPI / TPJ (tile shape).

// tile-local slot numbers
template <class B> SZH_HD int szh_slot(int I, int J) { return (I % B::TPI) * B::TPJ + (J % B::TPJ); }
template <class B> SZH_HD int szh_vslot_left(int I) { return B::TPI * B::TPJ + (I % B::TPI); }
template <class B> SZH_HD int szh_vslot_top(int J) { return B::TPI * B::TPJ + B::TPI + (J % B::TPJ); }

// COMPUTE wavefront: the sweep of one pencil.  Memory traffic: its own rows of values and codes (plain 16-byte accesses);
// every hand-off is LDS.
// HASREG: the pencil touches at least one regression block (otherwise the block bookkeeping and the
//         regression quantiser are compiled out); USEMEAN: the stream's use_mean flag.
template <class T, bool DEC, bool HASREG, bool USEMEAN, class B, int FMT = 0>
SZH_HD void szh_pencil_body(const szh_qargs<T> &a, int I, int J, const szh_tile_lds<T> &L)
{
    static_assert(FMT == 0 || (!HASREG && !USEMEAN), "the SZ 1.4 path has neither blocks nor the mean shortcut");
    constexpr int NL = B::NL;
    static_assert(B::RL % SZH_U == 0, "a trip must not wrap the face rings: ring(t0 + s) = ring(t0) + s");
    const szh_geom3 &G = a.G;
    const int r0 = G.g0.count, r1 = G.g1.count, r2 = G.g2.count;
    const int nbz = G.g2.num;
    const int cap_lor = a.cap - 2, cap_reg = a.cap, radius = a.radius;
    const T eb = a.eb, recip = a.recip, mean = a.mean;
    const bool free_run = SZH_DEV && a.dbg == 1;   // development: no hand-off at all (timing only, results are wrong)
    szh_u64 *const trace = SZH_DEV ? a.trace : nullptr;
    const int pi = I % B::TPI, pj = J % B::TPJ;
    const bool cut = SZH_DEV && a.dbg == 2;     // development: tiles cut apart (no hand-off across tile boundaries; timing only, results are wrong)
    const bool pubJ = (J + 1 < a.nJ) && !free_run && !(cut && pj == B::TPJ - 1), pubI = (I + 1 < a.nI) && !free_run && !(cut && pi == B::TPI - 1);
    const int myslot = szh_slot<B>(I, J);
    uint16_t *const cring = L.cring + (size_t)myslot * (SZH_XC + 1) * 64;
    // producers: a pencil of this tile, or the virtual producer the FILL wavefront keeps filled
    const bool hasPJ = J > 0 && !free_run && !(cut && pj == 0), hasPI = I > 0 && !free_run && !(cut && pi == 0);
    const int slotPJ = pj > 0 ? myslot - 1 : szh_vslot_left<B>(I), slotPI = pi > 0 ? myslot - B::TPJ : szh_vslot_top<B>(J);
    // consumers: a pencil of this tile (its step counter tells which ring slots it has read), or the STORE wavefront
    const bool consJ_in = pubJ && pj + 1 < B::TPJ, consI_in = pubI && pi + 1 < B::TPI;
    const int stride = B::face_stride(r2), RS = B::face_rowstride(r2);
    const int mybase = myslot * stride;

    // ---- per-lane constants ----
    int il[NL], jl[NL], skew[NL], askew[NL], hsk[NL], hrd[NL], st1[NL], st2[NL], cidx[NL], ctrash[NL];
    bool inb[NL], corner[NL], top[NL], left[NL];
    int64_t rowoff[NL], blkrow[NL];
    T fii[NL], fjj[NL];
    constexpr int NEVER = -(1 << 30);          // as a skew: (unsigned)(t - NEVER) < r2 never holds
    SZH_FORL {
        const int lane = B::lane(l);
        il[l] = lane >> 3; jl[l] = lane & 7;
        const int i = 8 * I + il[l], j = 8 * J + jl[l];
        inb[l] = (i < r0) && (j < r1);
        top[l] = i == 0; left[l] = j == 0;     // fmt 2: no neighbour in dim 0 / dim 1 (the multiplicative stencil takes 1 there, below)
        const int ic = i < r0 ? i : r0 - 1, jc = j < r1 ? j : r1 - 1;
        const int b0 = szh_blk_of(G.g0, ic), b1 = szh_blk_of(G.g1, jc);
        fii[l] = (T)(ic - szh_blk_start(G.g0, b0));
        fjj[l] = (T)(jc - szh_blk_start(G.g1, b1));
        rowoff[l] = (int64_t)ic * G.d0 + (int64_t)jc * G.d1;
        blkrow[l] = ((int64_t)b0 * G.g1.num + b1) * nbz;
        skew[l] = il[l] + jl[l];
        askew[l] = inb[l] ? skew[l] : NEVER;                  // "this lane has a point at step t": (unsigned)(t - askew) < r2
        const int trash = L.ftrash + lane * (SZH_U + 1);      // SZH_U + 1 write-only face slots per lane (odd stride: no bank conflicts)
        ctrash[l] = SZH_XC * 64 + lane;
        cidx[l] = lane - (skew[l] << 6);                      // code-ring slot of column k = t - skew: ((t << 6) + cidx) mod (SZH_XC * 64)
        // halo role: which face row this lane reads (for k = t - hsk); hlds = element offset of that row in faces[], -1: none
        int hskew = 0, hlds = -1;
        if (jl[l] == 0 && hasPJ && i < r0) { hskew = il[l]; hlds = slotPJ * stride + il[l] * RS; }                   // (i, 8J-1, k): J-face of (I,J-1), row il
        else if (il[l] == 0 && jl[l] > 0 && hasPI && j < r1) { hskew = jl[l]; hlds = slotPI * stride + (8 + jl[l]) * RS; } // (8I-1, j, k): I-face of (I-1,J), row jl
        else if (lane == 63 && hasPI) hlds = slotPI * stride + 8 * RS;                                              // for lane 0: (8I-1, 8J, k)
        else if (lane == 62 && hasPI && hasPJ) hlds = slotPI * stride + 16 * RS;                                    // for lane 0: (8I-1, 8J-1, k), forwarded by (I-1,J)
        hsk[l] = hlds >= 0 ? hskew : NEVER;
        hrd[l] = hlds >= 0 ? hlds : 0;         // lanes without a halo row read (and discard) some valid slot
        // face stores.  A lane's value goes into the ring row of its role at the position of the step (whether or not the step
        // is inside the lane's k range: positions outside it are never read -- consumers and the STORE wavefront check the range
        // themselves), so the address is `row base + ring(t)`: one add per TRIP and an immediate offset per step.
        //   first store : J-face row il (lanes jl = 7), else I-face row jl (lanes il = 7)
        //   second store: I-face row 7 for lane (7,7) when it also owns a J-face row; the corner column (the lane's HALO value) for lane (7,0)
        const int pjb = (pubJ && jl[l] == 7 && inb[l]) ? mybase + il[l] * RS : -1;
        const int pib = (pubI && il[l] == 7 && inb[l]) ? mybase + (8 + jl[l]) * RS : -1;
        const int pcb = (pubI && hasPJ && il[l] == 7 && jl[l] == 0 && inb[l]) ? mybase + 16 * RS : -1;
        corner[l] = pcb >= 0;
        st1[l] = pjb >= 0 ? pjb : (pib >= 0 ? pib : trash);
        st2[l] = (pjb >= 0 && pib >= 0) ? pib : (pcb >= 0 ? pcb : trash);
        (void)hskew;
    }
    const int ftrash0 = L.ftrash;

    // ---- per-lane block tracking along dim2 (only when the pencil touches regression blocks) ----
    int kk[NL], bz[NL], bk[NL];
    bool lor[NL], nlor[NL], nnlor[NL];
    T ca[NL], cb[NL], cc[NL], cd[NL], na[NL], nb_[NL], nc[NL], nd[NL], pbase[NL];
    SZH_FORL {
        bk[l] = 0; kk[l] = 0; bz[l] = szh_blk_size(G.g2, 0);
        ca[l] = cb[l] = cc[l] = cd[l] = 0; na[l] = nb_[l] = nc[l] = nd[l] = 0; pbase[l] = 0;
        lor[l] = nlor[l] = nnlor[l] = true;
        if (HASREG && inb[l]) {
            const int64_t b = blkrow[l];
            lor[l] = a.blk_lor[b] != 0;
            if (nbz > 1) nlor[l] = a.blk_lor[b + 1] != 0;
            if (nbz > 2) nnlor[l] = a.blk_lor[b + 2] != 0;
            if (!lor[l]) {
                ca[l] = B::ld_coef(a.coef + b); cb[l] = B::ld_coef(a.coef + a.coef_stride + b); cc[l] = B::ld_coef(a.coef + 2 * a.coef_stride + b); cd[l] = B::ld_coef(a.coef + 3 * a.coef_stride + b);
                pbase[l] = ca[l] * fii[l] + cb[l] * fjj[l];
            }
            if (!nlor[l]) {
                na[l] = B::ld_coef(a.coef + b + 1); nb_[l] = B::ld_coef(a.coef + a.coef_stride + b + 1); nc[l] = B::ld_coef(a.coef + 2 * a.coef_stride + b + 1); nd[l] = B::ld_coef(a.coef + 3 * a.coef_stride + b + 1);
            }
        }
    }

    // ---- per-trip inputs.  Nothing stays in flight across the loop back-edge (hipcc guards rotated in-flight registers with
    //      s_waitcnt vmcnt(0) = one memory round trip per STEP); a trip requests everything it needs at its top.
    //      Every lane moves its OWN row's next SZH_U values as 16-byte vectors; the 2-byte codes go through a small LDS ring
    //      and are moved between ring and HBM as aligned 16-byte row segments.
    constexpr int VPT = 16 / (int)sizeof(T);      // values per 16-byte vector
    constexpr int NVEC = SZH_U / VPT;             // vectors per lane per trip
    // compress: originals; decompress: pre-scattered values in, reconstruction out.  Two buffers: while a trip steps through one, the
    // NEXT trip's values are on their way into the other (compress), so the sweep never waits for HBM at the top of a trip -- and,
    // since a wavefront's loads and stores return in issue order, never for the acknowledgement of the code stores issued before them
    typedef T xbuf_t[SZH_U][NL];
    xbuf_t xr0, xr1;
    const bool vec_codes = (r2 % 8) == 0;         // rows of the u16 code array are 16-byte aligned

    // decompress: which columns of this lane's OWN row (row = lane, as in move_codes) hold a zero code, one bit per ring column.
    // Only those positions need the pre-scattered value, so a lane fetches its 16 values of a trip only when one of them does.
    unsigned zcols[NL];
    SZH_FORL zcols[l] = 0;
    const bool no_ld = SZH_DEV && (a.dbg == 3 || a.dbg == 5), no_st = SZH_DEV && (a.dbg == 3 || a.dbg == 4);   // development: the sweep loads / stores nothing (timing only, results are wrong)
    auto load_x = [&](int t0, xbuf_t &xr) {
        const T *src = DEC ? a.out : a.data;
        if (no_ld) { SZH_FORL { SZH_UNROLL for (int s = 0; s < SZH_U; ++s) xr[s][l] = (T)0; } return; }
        // interior trips: the SZH_U columns of every lane (skews 0..14) lie inside its row, so the whole wavefront takes the vector
        // path -- a wavefront-uniform branch (a per-lane choice between the two paths makes hipcc wait for the vector loads before
        // it issues the other path's loads into the same registers, which would undo the prefetch).  Lanes outside the array load
        // their clamped row: their values reach nothing (no code, no face, no valid neighbour).
        const bool interior = t0 >= SZH_U && t0 + SZH_U <= r2;
        if (interior) {
            SZH_FORL {
                const int k0 = t0 - skew[l];
                if (DEC && SZH_XC == 32) {
                    const unsigned sh = (unsigned)k0 & 31u;
                    const unsigned win = (zcols[l] >> sh) | (sh ? zcols[l] << (32u - sh) : 0u);      // bit s = column k0 + s
                    if ((win & 0xffffu) == 0) continue;                                              // no unpredictable value in this lane's trip
                }
                SZH_UNROLL
                for (int v = 0; v < NVEC; ++v) {
                    T tmp[VPT];
                    B::ld16(src + rowoff[l] + k0 + v * VPT, tmp);
                    SZH_UNROLL
                    for (int e = 0; e < VPT; ++e) xr[v * VPT + e][l] = tmp[e];
                }
            }
        } else {
            SZH_FORL {
                const int k0 = t0 - skew[l];
                SZH_UNROLL
                for (int s = 0; s < SZH_U; ++s) {
                    const int k = k0 + s;
                    xr[s][l] = (inb[l] && (unsigned)k < (unsigned)r2) ? src[rowoff[l] + k] : (T)0;
                }
            }
        }
    };
    auto store_out = [&](int t0, xbuf_t &xr) {
        if (no_st) return;
        const bool interior = t0 >= SZH_U && t0 + SZH_U <= r2;   // wavefront-uniform, as in load_x
        if (interior) {
            SZH_FORL {
                const int k0 = t0 - skew[l];
                if (inb[l]) {
                    SZH_UNROLL
                    for (int v = 0; v < NVEC; ++v) {
                        T tmp[VPT];
                        SZH_UNROLL
                        for (int e = 0; e < VPT; ++e) tmp[e] = xr[v * VPT + e][l];
                        B::st16(a.out + rowoff[l] + k0 + v * VPT, tmp);
                    }
                }
            }
        } else {
            SZH_FORL {
                const int k0 = t0 - skew[l];
                SZH_UNROLL
                for (int s = 0; s < SZH_U; ++s) {
                    const int k = k0 + s;
                    if (inb[l] && (unsigned)k < (unsigned)r2) a.out[rowoff[l] + k] = xr[s][l];
                }
            }
        }
    };
    // code ring <-> code array, columns [c0, c0+8) of all 64 rows of the pencil (c0 multiple of 8): one 16-byte row segment per lane
    auto row_off = [&](int row, int64_t &off) -> bool {
        const int i = 8 * I + (row >> 3), j = 8 * J + (row & 7);
        off = (int64_t)i * G.d0 + (int64_t)j * G.d1;
        return i < r0 && j < r1;
    };
    auto move_codes = [&](int c0) {
        if (no_st) return;
        if (vec_codes && c0 + 8 <= r2) {
            SZH_FORL {
                const int row = B::lane(l);
                int64_t off;
                if (row_off(row, off)) {
                    uint16_t tmp[8];
                    if (DEC) {
                        B::ld16(a.codes + off + c0, tmp);
                        unsigned z = 0;
                        SZH_UNROLL
                        for (int e = 0; e < 8; ++e) { cring[((c0 + e) % SZH_XC) * 64 + row] = tmp[e]; z |= (tmp[e] == 0 ? 1u : 0u) << e; }
                        const unsigned sh = (unsigned)c0 & 31u;                                   // c0 is a multiple of 8: the byte of the mask
                        zcols[l] = (zcols[l] & ~(0xffu << sh)) | (z << sh);
                    } else {
                        SZH_UNROLL
                        for (int e = 0; e < 8; ++e) tmp[e] = cring[((c0 + e) % SZH_XC) * 64 + row];
                        B::st16(a.codes + off + c0, tmp);
                    }
                }
            }
        } else {
            for (int q = 0; q < 8; ++q) {
                SZH_FORL {
                    const int lane = B::lane(l);
                    const int row = q * 8 + (lane >> 3), col = c0 + (lane & 7);
                    int64_t off;
                    if (row_off(row, off) && col < r2) {
                        if (DEC) cring[(col % SZH_XC) * 64 + row] = a.codes[off + col];
                        else a.codes[off + col] = cring[(col % SZH_XC) * 64 + row];
                    }
                }
            }
            if (DEC) { SZH_FORL zcols[l] |= 0xffu << ((unsigned)c0 & 31u); }   // ragged rows: treat the group as "may hold zeros"
        }
    };
    // bounded wait until an LDS counter reaches `need`; returns the value seen.  A lost hand-off ends the launch, it must not hang the GPU
    szh_u64 tr_spins = 0;
    auto wait_ctr = [&](const unsigned *ctr, int need) -> int {
        int v = (int)B::lds_ld_u(ctr);
        unsigned spins = 0;
        while (v < need && !free_run) {
            if (++spins > (1u << 24)) { SZH_FORL { if (B::lane(l) == 0) B::st_flag(a.err, 1u); } break; }
            if ((spins & 4095u) == 0 && B::ld_flag(a.err) != 0) break;
            B::backoff(1);
            v = (int)B::lds_ld_u(ctr);
        }
        if (SZH_DEV) tr_spins += spins;
        return v;
    };

    // rolling neighbour state (values at the previous step)
    T cur[NL], A1[NL], B1[NL], C1[NL], cur2[NL];
    SZH_FORL { cur[l] = 0; A1[l] = 0; B1[l] = 0; C1[l] = 0; cur2[l] = 0; }
    const bool first_pencil = FMT >= 1 && I == 0 && J == 0;   // holds the array's first row (its lane 0)

    const int tsteps = r2 + 14;
    int flushed = 0, filled = 0;   // code-ring columns already written back (compress) / already brought in (decompress)
    szh_u64 tr_start = 0, tr_first = 0;
    if (trace) tr_start = B::clock();
    const bool detail = trace && I == a.nI / 2 && J == a.nJ / 2;
    szh_u64 *steplog = nullptr;   // development: clock at the end of every step of the observed producer / consumer pencil
    if (trace) {
        const int oTI = a.trace_tile >> 16, oTJ = a.trace_tile & 0xffff;
        szh_u64 *logs = trace + (int64_t)a.nI * a.nJ * 8 + 256;
        if (I == oTI * B::TPI && J == oTJ * B::TPJ - 1) steplog = logs;                              // producer: last pencil column of the tile to the left
        if (I == oTI * B::TPI && J == oTJ * B::TPJ) steplog = logs + 3 * 2 * SZH_TRACE_LOG;         // consumer: first pencil of the observed tile
    }
    szh_u64 *dt = trace ? trace + (int64_t)a.nI * a.nJ * 8 : nullptr;
    // (decompress learns which values it needs from the trip's own codes: it asks at the top of the trip; double has no registers to spare for
    //  a second buffer next to 2 x 4 pencils per workgroup)
    constexpr bool PREFETCH = !DEC && sizeof(T) == 4;
    auto trip = [&](const int t0, xbuf_t &xr, xbuf_t &xnext) {
        if (detail && t0 / SZH_U < 64) { SZH_FORL { if (B::lane(l) == 0) dt[(t0 / SZH_U) * 4 + 0] = B::clock(); } }
        // ring space: the consumers must have read the slots this trip overwrites.  A pencil of the tile reads column k no later
        // than its step k+7; the STORE wavefront counts the columns it has forwarded.  (This trip writes columns <= t0 + SZH_U - 8.)
        if (pubJ) { if (consJ_in) wait_ctr(L.cstep + myslot + 1, t0 + SZH_U - B::RL); else wait_ctr(L.spubJ + myslot, t0 + SZH_U - 7 - B::RL); }
        if (pubI) { if (consI_in) wait_ctr(L.cstep + myslot + B::TPJ, t0 + SZH_U - B::RL); else wait_ctr(L.spubI + myslot, t0 + SZH_U - 7 - B::RL); }
        // how far are the producers?  (step t needs their step t+7 finished)
        int pstepJ = hasPJ ? (int)B::lds_ld_u(L.cstep + slotPJ) : (1 << 30);
        int pstepI = hasPI ? (int)B::lds_ld_u(L.cstep + slotPI) : (1 << 30);
        if (DEC) { while (filled < t0 + SZH_U && filled < r2) { move_codes(filled); filled += 8; } } // ring holds columns < filled
        if (PREFETCH) {
            // ALL of this trip's global memory traffic is issued here, in one batch: first the wait for this trip's own values (asked for
            // a whole trip ago -- hipcc waits for the wavefront's entire memory queue, so the wait must come BEFORE anything new is
            // issued: B::touch makes it), then the next trip's values, then the code columns that every lane has passed.  Whatever
            // the next trip's wait finds in the queue is then a whole trip (SZH_U steps) old.  The code ring just holds it: columns
            // [t0 - SZH_U, t0 + SZH_U) are in flight during the trip (SZH_XC = 2 * SZH_U).
            static_assert(DEC || SZH_XC >= 2 * SZH_U, "code ring too small for one flush per trip");
            SZH_FORL { SZH_UNROLL for (int s = 0; s < SZH_U; ++s) B::touch(xr[s][l]); }
            if (t0 + SZH_U < tsteps) load_x(t0 + SZH_U, xnext);
        } else load_x(t0, xr);
        if (!DEC) { while (flushed + 8 <= t0 - 14 && flushed < r2) { move_codes(flushed); flushed += 8; } }
        // face-store addresses of this trip: row base + ring(t0) (write-only slots keep their place); ring(t0 + s) = ring(t0) + s
        int stb1[NL], stb2[NL];
        SZH_FORL {
            const int w0 = B::ring(t0);
            stb1[l] = st1[l] >= ftrash0 ? st1[l] : st1[l] + w0;
            stb2[l] = st2[l] >= ftrash0 ? st2[l] : st2[l] + w0;
        }
        if (detail && t0 / SZH_U < 64) { SZH_FORL { if (B::lane(l) == 0) dt[(t0 / SZH_U) * 4 + 1] = B::clock(); } }
        SZH_UNROLL
        for (int s = 0; s < SZH_U; ++s) {
            const int t = t0 + s;
            // -- halo for this step: the producers' step t+7 must be complete (no halo is needed once t - hskew >= r2)
            T hval[NL];
            {
                const int need = t + 8 < tsteps ? t + 8 : tsteps;
                if (pstepJ < need) pstepJ = wait_ctr(L.cstep + slotPJ, need);
                if (pstepI < need) pstepI = wait_ctr(L.cstep + slotPI, need);
                // A face value sits at the ring position of the PRODUCER'S STEP that made it (column + the producing lane's skew), so
                // every lane of this wavefront reads position t + 7 and every lane of the producer wrote position t: one
                // wavefront-uniform offset per step instead of per-lane address arithmetic.
                const int rpos = B::ring(t + 7);
                SZH_FORL {      // every lane reads (lanes without a halo row read a don't-care slot)
                    const bool lact = (unsigned)(t - hsk[l]) < (unsigned)r2;
                    const T v = B::lds_ld(L.faces + hrd[l] + rpos);
                    hval[l] = lact ? v : (T)0;
                }
            }
            // -- neighbours through cross-lane moves (values of the previous step) --
            T shA[NL], shB[NL], shCi[NL], shCj[NL];
            B::shfl_up1(shA, cur);
            B::shfl_up(shB, cur, 8);
            B::shfl_up(shCi, A1, 8);
            B::shfl_up1(shCj, B1);
            const T h63 = B::readlane(hval, 63), h62 = B::readlane(hval, 62);

            SZH_FORL {
                const int k = t - skew[l];
                const bool act = (unsigned)(t - askew[l]) < (unsigned)r2;
                const int lane = B::lane(l);
                const int cslot = act ? (((t << 6) + cidx[l]) & (SZH_XC * 64 - 1)) : ctrash[l];
                const T nA = jl[l] > 0 ? shA[l] : hval[l];
                const T nB = il[l] > 0 ? shB[l] : (lane == 0 ? h63 : hval[l]);
                const T nC = il[l] > 0 ? shCi[l] : (jl[l] > 0 ? shCj[l] : h62);
                // [-1] + [-s1] + [-s0] - [-s1-1] - [-s0-1] - [-s0-s1] + [-s0-s1-1], left to right
                // (SZ 1.4 subtracts [-s0-s1] before [-s0-1], sz_float.c:1311; its first row predicts 2P[k-1] - P[k-2] from k = 2 on)
                T pred = FMT == 1 ? cur[l] + nA + nB - A1[l] - nC - B1[l] + C1[l] : cur[l] + nA + nB - A1[l] - B1[l] - nC + C1[l];
                if (FMT == 1 && first_pencil && lane == 0 && k >= 2) pred = (T)2 * cur[l] - cur2[l];
                bool nofabs = false; (void)nofabs;
                if (FMT == 2) {
                    // [-1] * [-s1] * [-s0] * [-s0-s1-1] / ([-s1-1] * [-s0-s1] * [-s0-1]) in double, left to right (sz_float.c:2654-2656); on the
                    // array's faces and edges the reference spells the 2-D / 1-D forms out (:2403-2620) -- they ARE this form with 1 in place
                    // of every neighbour that does not exist (a factor of exactly 1 changes no bit) -- except on the very first row, where it is
                    // P[k-1]^2 / P[k-2] from k = 2 on (:2403).  The float 2-D quantiser multiplies in float ((a * b) / c, :2091, :2152).
                    const bool k0 = k == 0;
                    const T one = (T)1;
                    const T mc = k0 ? one : cur[l], mA = left[l] ? one : nA, mB = top[l] ? one : nB;
                    const T mA1 = (left[l] || k0) ? one : A1[l], mB1 = (top[l] || k0) ? one : B1[l];
                    const T mC = (top[l] || left[l]) ? one : nC, mC1 = (top[l] || left[l] || k0) ? one : C1[l];
                    if (top[l] && left[l] && k >= 2) pred = a.f32arith ? (T)((T)(cur[l] * cur[l]) / cur2[l]) : (T)((double)cur[l] * (double)cur[l] / (double)cur2[l]);
                    else if (a.f32arith) pred = (T)((T)(mc * mA) / mA1);
                    else pred = (T)((double)mc * (double)mA * (double)mB * (double)mC1 / ((double)mA1 * (double)mC * (double)mB1));
                    nofabs = !DEC && a.ndim3 && top[l] && !left[l] && k0;          // the compressor multiplies the signed prediction here (:2459)
                }
                T predr = 0;
                if (HASREG) predr = pbase[l] + cc[l] * (T)kk[l] + cd[l];
                const bool is_lor = HASREG ? lor[l] : true;
                T nv;
                if (!DEC) {
                    const T x = xr[s][l];
                    T rcl;
                    int code;
                    if (FMT == 2) {
                        code = szh_msst_state<T>(a, (double)(T)(x / pred));
                        if (first_pencil && lane == 0 && k == 0) code = 0;          // the array's first value is always exact
                        rcl = code ? (T)((nofabs ? (double)pred : (double)szh_abs(pred)) * a.ptab[code]) : szh_keep_bits_msst(x, a.ign_bits);
                    } else
                    code = FMT == 1 ? szh_quant_sel14<T>(x, pred, eb, recip, cap_reg, radius, a.median, a.ign_bits,
                                                             first_pencil && lane == 0 && k == 0, &rcl)
                                        : szh_quant_sel<T>(x, pred, eb, recip, cap_lor, radius, &rcl);
                    if (USEMEAN) {
                        if (code != 0 && code <= radius) code -= 1;                 // sz_float.c:6944
                        if (szh_abs(x - mean) <= eb) { code = radius; rcl = mean; } // sz_float.c:6929
                    }
                    nv = rcl;
                    if (HASREG) {
                        T rcr;
                        const int cr = szh_quant_sel<T>(x, predr, eb, recip, cap_reg, radius, &rcr);
                        code = is_lor ? code : cr;
                        nv = is_lor ? rcl : rcr;
                    }
                    B::lds_st(cring + cslot, (uint16_t)code);
                } else {
                    const int cread = (int)B::lds_ld(cring + cslot);
                    const int c0 = act ? cread : radius;
                    int c = c0;
                    const T p = is_lor ? pred : predr;
                    bool is_mean = false;
                    if (USEMEAN) {
                        is_mean = is_lor && (c == radius);
                        if (is_lor && c != 0 && c < radius) c += 1;                 // szd_float.c:3784
                    }
                    nv = p + (T)(2 * (c - radius)) * eb;
                    if (FMT == 2) nv = (T)((double)szh_abs(pred) * a.ptab[c < cap_reg ? c : 0]);   // szd_float.c:2927 (a code beyond the table: a broken stream)
                    if (USEMEAN && is_mean) nv = mean;
                    if (act && c0 == 0) nv = xr[s][l];                              // pre-scattered unpredictable value (read at the top of the trip)
                    xr[s][l] = nv;
                }
                // faces for the pencils to the right / below (in this tile or, through the STORE wavefront, in the next one)
                if (pubJ || pubI) B::lds_st(L.faces + stb1[l] + s, nv);
                if (pubI) B::lds_st(L.faces + stb2[l] + s, corner[l] ? hval[l] : nv);   // lane (7,0): its halo value IS the corner column of the pencil below, same k
                // roll the neighbour state.  Lanes outside the k range must hand on ZEROS (the reference's zero halo).  With
                // Lorenzo-only data they produce zeros by themselves (zero input, zero neighbours); the mean shortcut, stale
                // pre-scattered values and a regression plane (non-zero prediction at k < 0) need the mask.
                if (FMT >= 1) cur2[l] = cur[l];
                cur[l] = (USEMEAN || DEC || HASREG || FMT >= 1) ? (act ? nv : (T)0) : nv;
                A1[l] = nA; B1[l] = nB; C1[l] = nC;
                // advance along dim2
                if (HASREG && act) {
                    kk[l] += 1;
                    if (kk[l] == bz[l]) {
                        bk[l] += 1; kk[l] = 0;
                        if (bk[l] < nbz) {
                            bz[l] = szh_blk_size(G.g2, bk[l]);
                            lor[l] = nlor[l]; ca[l] = na[l]; cb[l] = nb_[l]; cc[l] = nc[l]; cd[l] = nd[l];
                            pbase[l] = ca[l] * fii[l] + cb[l] * fjj[l];
                            nlor[l] = nnlor[l];
                            const int64_t b = blkrow[l] + bk[l];
                            if (bk[l] + 1 < nbz && !nlor[l]) {
                                na[l] = B::ld_coef(a.coef + b + 1); nb_[l] = B::ld_coef(a.coef + a.coef_stride + b + 1);
                                nc[l] = B::ld_coef(a.coef + 2 * a.coef_stride + b + 1); nd[l] = B::ld_coef(a.coef + 3 * a.coef_stride + b + 1);
                            }
                            nnlor[l] = (bk[l] + 2 < nbz) ? (a.blk_lor[b + 2] != 0) : true;
                        }
                    }
                }
            }
            // step t is complete: its face values are in the ring.  The LDS executes one wavefront's accesses in order, so the counter
            // store needs no wait behind the face stores -- only the compiler must keep the order (lds_order)
            B::lds_order();
            SZH_FORL { B::lds_st(L.cstep + myslot, (unsigned)(t + 1)); }
            if (SZH_DEV && steplog && t < SZH_TRACE_LOG) { SZH_FORL { if (B::lane(l) == 0) { steplog[2 * t] = B::clock(); steplog[2 * t + 1] = (szh_u64)(pstepJ < pstepI ? pstepJ : pstepI); } } }
            // half-way through the trip: write back the code columns that are complete (keeps the 32-column ring from wrapping)
        }
        if (detail && t0 / SZH_U < 64) { SZH_FORL { if (B::lane(l) == 0) dt[(t0 / SZH_U) * 4 + 2] = B::clock(); } }
        if (DEC) store_out(t0, xr);
        if (detail && t0 / SZH_U < 64) { SZH_FORL { if (B::lane(l) == 0) dt[(t0 / SZH_U) * 4 + 3] = B::clock(); } }
        if (trace && t0 == 0) tr_first = B::clock();
    };
    if (PREFETCH) {
        load_x(0, xr0);
        for (int t0 = 0; t0 < tsteps; t0 += 2 * SZH_U) {
            trip(t0, xr0, xr1);
            if (t0 + SZH_U < tsteps) trip(t0 + SZH_U, xr1, xr0);
        }
    } else {
        for (int t0 = 0; t0 < tsteps; t0 += SZH_U) trip(t0, xr0, xr0);
    }
    if (!DEC) { for (; flushed < r2; flushed += 8) move_codes(flushed); }
    if (trace) {
        const szh_u64 tr_end = B::clock();
        SZH_FORL {
            if (B::lane(l) == 0) {
                szh_u64 *tp = trace + ((int64_t)I * a.nJ + J) * 8;
                tp[0] = tr_start; tp[1] = tr_start; tp[2] = tr_first; tp[3] = tr_end; tp[4] = tr_spins; tp[6] = B::where(); tp[7] = 0;
            }
        }
    }
}

// does pencil (I,J) touch a regression block?  (collective over the wavefront's lanes)
template <class T, class B>
SZH_HD bool szh_pencil_has_reg(const szh_qargs<T> &a, int I, int J)
{
    constexpr int NL = B::NL;
    const szh_geom3 &G = a.G;
    const int i0 = 8 * I, i1 = (8 * I + 7 < G.g0.count ? 8 * I + 7 : G.g0.count - 1);
    const int j0 = 8 * J, j1 = (8 * J + 7 < G.g1.count ? 8 * J + 7 : G.g1.count - 1);
    const int b0lo = szh_blk_of(G.g0, i0), b0hi = szh_blk_of(G.g0, i1);
    const int b1lo = szh_blk_of(G.g1, j0), b1hi = szh_blk_of(G.g1, j1);
    const int nbz = G.g2.num;
    const int ncol = (b0hi - b0lo + 1) * (b1hi - b1lo + 1);
    bool none[NL];
    SZH_FORL {
        none[l] = true;
        for (int e = B::lane(l); e < ncol * nbz; e += 64) {
            const int c = e / nbz, z = e - c * nbz;
            const int b0 = b0lo + c / (b1hi - b1lo + 1), b1 = b1lo + c % (b1hi - b1lo + 1);
            if (a.blk_lor[((int64_t)b0 * G.g1.num + b1) * nbz + z] == 0) none[l] = false;
        }
    }
    return !B::all(none);
}

// L: this pencil's view of its tile's LDS (HIP) / plain memory (simulator); face rings and step counters must be zero at launch
// MS: the kernel instance of the table-driven point-wise-relative quantiser (fmt 2) -- a kernel of its own, so that the code of the other
// formats stays what it is without it (measured: with fmt 2 inside the same kernel the 512^3 ABS headline ran 1.5-5 % slower)
template <class T, bool DEC, class B, bool MS = false>
SZH_HD void szh_pencil_run(const szh_qargs<T> &a, int I, int J, const szh_tile_lds<T> &L)
{
    if (MS) { szh_pencil_body<T, DEC, false, false, B, 2>(a, I, J, L); return; }
    if (a.fmt == 1) { szh_pencil_body<T, DEC, false, false, B, 1>(a, I, J, L); return; }
#ifdef SZH_EXP_NOREG
    const bool hasreg = false;
#else
    const bool hasreg = a.no_reg ? false : szh_pencil_has_reg<T, B>(a, I, J);
#endif
    if (!DEC && hasreg && a.coef_progress) {
        // the decoded coefficients are still arriving (the host's chain runs next to this launch): wait until the last block this pencil
        // reads -- scan order: its largest (b0, b1), whole row of dim2 -- is final.  Bounded like every other wait of the kernel.
        const szh_geom3 &G = a.G;
        const int i1 = 8 * I + 7 < G.g0.count ? 8 * I + 7 : G.g0.count - 1, j1 = 8 * J + 7 < G.g1.count ? 8 * J + 7 : G.g1.count - 1;
        const szh_u64 need = ((szh_u64)szh_blk_of(G.g0, i1) * G.g1.num + szh_blk_of(G.g1, j1) + 1) * (szh_u64)G.g2.num;
        // the word is {launch epoch << 40 | blocks}: it is written by the host's DMA only and never cleared -- a word that a kernel (a
        // memset) has stored stays in that XCD's L2, where this wavefront's loads would keep finding the old value
        const szh_u64 tag = (szh_u64)(a.epoch & 0xffffffu);
        auto arrived = [&]() { const szh_u64 v = B::ld_sys_u64(a.coef_progress); return (v >> 40) == tag && (v & ((1ull << 40) - 1)) >= need; };
        unsigned spins = 0;
        while (!arrived()) {
            if (++spins > (1u << 22)) { B::st_flag(a.err, 2u); break; }        // 2: the coefficients did not arrive
            if ((spins & 1023u) == 0 && B::ld_flag(a.err) != 0) break;
            B::nap();
        }
    }
    if (a.use_mean) {
        if (hasreg) szh_pencil_body<T, DEC, true, true, B>(a, I, J, L);
        else szh_pencil_body<T, DEC, false, true, B>(a, I, J, L);
    } else {
        if (hasreg) szh_pencil_body<T, DEC, true, false, B>(a, I, J, L);
        else szh_pencil_body<T, DEC, false, false, B>(a, I, J, L);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Helper wavefronts of a tile (TI, TJ) = pencils [TI*TPI, ...) x [TJ*TPJ, ...).  One ROW of a face per lane:
//   lanes 0 .. 8*TPI-1          J-face rows: pencil row pi = lane / 8, face row il = lane % 8
//   lanes 8*TPI .. 8*TPI+9*TPJ-1  I-face rows: pencil column pj = (lane - 8*TPI) / 9, face row r = (lane - 8*TPI) % 9 (8 = corner column)
// sr = the step offset at which the PRODUCING pencil finishes column k of that row (k + sr); hs = the step offset at which
// the CONSUMING pencil reads it (k + hs).
template <class B> struct szh_rowmap {
    bool isJ, valid; int pp, r, ringrow, sr, hs;
    SZH_HD explicit szh_rowmap(int lane)
    {
        constexpr int NJ = 8 * B::TPI, NI = 9 * B::TPJ;
        valid = lane < NJ + NI; isJ = lane < NJ;
        if (isJ) { pp = lane >> 3; r = lane & 7; ringrow = r; sr = r + 7; hs = r; }
        else { const int q = valid ? lane - NJ : 0; pp = q / 9; r = q - pp * 9; ringrow = 8 + r; sr = r < 8 ? 7 + r : 7; hs = r < 8 ? r : 0; }
    }
};
// minimum of v over the lanes of each pencil group; result returned to every lane of the group (through the LDS scratch row)
template <class B> SZH_HD void szh_group_min(int *scratch, const int (&v)[B::NL], int (&out)[B::NL])
{
    constexpr int NL = B::NL;
    SZH_FORL { B::lds_st(scratch + B::lane(l), v[l]); }
    B::lds_fence();
    SZH_FORL {
        const szh_rowmap<B> m(B::lane(l));
        const int base = m.isJ ? m.pp * 8 : 8 * B::TPI + m.pp * 9, cnt = m.isJ ? 8 : 9;
        int mn = 1 << 30;
        if (m.valid) { for (int e = 0; e < cnt; ++e) { const int x = B::lds_ld(scratch + base + e); mn = x < mn ? x : mn; } }
        out[l] = mn;
    }
    B::lds_fence();
}

// STORE wavefront (stores only): forwards the face rings of the tile's last column / last row of pencils to the granule buffers.
template <class T, class B>
SZH_HD void szh_tile_store(const szh_qargs<T> &a, int TI, int TJ, const szh_tile_lds<T> &L)
{
    constexpr int NL = B::NL;
    constexpr int NW = szh_gran<T>::NW;
    constexpr int KP = 8;                       // columns per row per round
    const int r0 = a.G.g0.count, r1 = a.G.g1.count, r2 = a.G.g2.count;
    const int stride = B::face_stride(r2);
    int pk[NL], slot[NL], rbase[NL];
    bool en[NL];
    szh_u64 *dst[NL];
    int64_t penc[NL];
    // granules travel as 16-byte `sc0 sc1` buffer accesses (two neighbouring granules), addressed by a 32-bit offset from the first
    // row of the tile's J- / I-face group (a.wide; otherwise as single 8-byte agent-scope atomics).  hipcc waits for the completion of
    // a VOLATILE 16-byte access before it issues the next one -- one fabric round trip per access; these are ordinary accesses with
    // the cache policy in the instruction, so a round's stores (and the FILL wavefront's loads) are all in flight together.
    szh_u64 *const gbaseJ = a.faceJ + ((int64_t)(TI * B::TPI) * a.nJ + (TJ * B::TPJ + B::TPJ - 1)) * 8 * (int64_t)r2 * NW;
    szh_u64 *const gbaseI = a.faceI + ((int64_t)(TI * B::TPI + B::TPI - 1) * a.nJ + TJ * B::TPJ) * 9 * (int64_t)r2 * NW;
    const typename B::gbuf_t bufJ = B::make_gbuf(gbaseJ), bufI = B::make_gbuf(gbaseI);
    const bool wide = a.wide != 0;
    SZH_FORL {
        const szh_rowmap<B> m(B::lane(l));
        // J-face rows come from the pencils of the tile's LAST column, I-face rows from its LAST row
        const int I = m.isJ ? TI * B::TPI + m.pp : TI * B::TPI + B::TPI - 1, J = m.isJ ? TJ * B::TPJ + B::TPJ - 1 : TJ * B::TPJ + m.pp;
        en[l] = m.valid && I < a.nI && J < a.nJ && !(SZH_DEV && (a.dbg == 1 || a.dbg == 2));
        if (m.isJ) en[l] = en[l] && J + 1 < a.nJ && 8 * I + m.r < r0;
        else en[l] = en[l] && I + 1 < a.nI && (m.r < 8 ? 8 * J + m.r < r1 : J > 0);
        penc[l] = (int64_t)I * a.nJ + J;
        slot[l] = en[l] ? szh_slot<B>(I, J) : 0;
        rbase[l] = slot[l] * stride + m.ringrow * B::face_rowstride(r2);
        dst[l] = m.isJ ? a.faceJ + (penc[l] * 8 + m.r) * (int64_t)r2 * NW : a.faceI + (penc[l] * 9 + m.r) * (int64_t)r2 * NW;
        pk[l] = en[l] ? 0 : r2;
    }
    int reported[NL];
    SZH_FORL reported[l] = -1;
    unsigned idle = 0;
    int slog = 0;
    for (;;) {
        bool none[NL], fin[NL];
        int steps[NL];
        SZH_FORL {
            const szh_rowmap<B> m(B::lane(l));
            const int cs = (int)B::lds_ld(L.cstep + slot[l]);
            int avail = cs - m.sr; if (avail > r2) avail = r2;
            int n = avail - pk[l]; if (n > KP) n = KP; if (n < 0 || !en[l]) n = 0;
            // granules leave as 16-byte stores wherever two of them are neighbours at a 16-byte boundary (half the fabric writes):
            // double: the two words of a column; float: the columns k, k+1 with an even granule index
            szh_u64 gw[KP + 1][NW];
            SZH_UNROLL
            for (int e = 0; e < KP; ++e) {
                const int k = pk[l] + e;
                const T v = e < n ? B::lds_ld(L.faces + rbase[l] + B::ring(k + m.sr)) : (T)0;   // ring position = the producer's step
                szh_gran<T>::pack(v, a.epoch, gw[e]);
            }
            SZH_UNROLL
            for (int w = 0; w < NW; ++w) gw[KP][w] = 0;
            auto put2 = [&](szh_u64 *p, szh_u64 wa, szh_u64 wb) {   // two granules at a 16-byte boundary
                if (wide) {
                    if (m.isJ) B::st_gran2_b(bufJ, (unsigned)((p - gbaseJ) * 8), wa, wb); else B::st_gran2_b(bufI, (unsigned)((p - gbaseI) * 8), wa, wb);
                } else { B::st_gran(p, wa); B::st_gran(p + 1, wb); }
            };
            if (NW == 2) {
                SZH_UNROLL
                for (int e = 0; e < KP; ++e) { if (e < n) put2(dst[l] + (int64_t)(pk[l] + e) * 2, gw[e][0], gw[e][NW - 1]); }
            } else {
                szh_u64 *const p0 = dst[l] + pk[l];
                const int odd = (int)(((uintptr_t)p0 >> 3) & 1);          // the first granule sits in the upper half of a 16-byte slot
                if (odd && n > 0) B::st_gran(p0, gw[0][0]);
                SZH_UNROLL
                for (int j = 0; j < KP / 2; ++j) {
                    const int e = 2 * j + odd;                             // pair (e, e+1), 16-byte aligned
                    const szh_u64 wa = odd ? gw[2 * j + 1][0] : gw[2 * j][0], wb = odd ? gw[2 * j + 2][0] : gw[2 * j + 1][0];
                    if (e + 1 < n) put2(p0 + e, wa, wb);
                    else if (e < n) B::st_gran(p0 + e, wa);
                }
            }
            pk[l] += n;
            none[l] = n == 0; fin[l] = pk[l] >= r2;
            steps[l] = en[l] ? (pk[l] >= r2 ? (1 << 29) : pk[l] + m.sr) : (1 << 30);
        }
        if (SZH_DEV && a.trace && TI == (a.trace_tile >> 16) && TJ + 1 == (a.trace_tile & 0xffff) && slog < SZH_TRACE_LOG) {
            SZH_FORL { if (B::lane(l) == 0) { szh_u64 *lg = a.trace + (int64_t)a.nI * a.nJ * 8 + 256 + 1 * 2 * SZH_TRACE_LOG; lg[2 * slog] = B::clock(); lg[2 * slog + 1] = (szh_u64)pk[l]; } }
            ++slog;
        }
        // per pencil: ring space for the producer (columns forwarded on every row) and the progress word for the consumers' FILL
        int cols[NL], gcols[NL], gsteps[NL];
        SZH_FORL cols[l] = en[l] ? pk[l] : (1 << 30);
        szh_group_min<B>(L.scratch, cols, gcols);
        szh_group_min<B>(L.scratch, steps, gsteps);
        SZH_FORL {
            const szh_rowmap<B> m(B::lane(l));
            const bool leader = m.valid && m.r == 0 && gcols[l] < (1 << 30);
            if (leader) {
                const int I = m.isJ ? TI * B::TPI + m.pp : TI * B::TPI + B::TPI - 1, J = m.isJ ? TJ * B::TPJ + B::TPJ - 1 : TJ * B::TPJ + m.pp;
                if (I < a.nI && J < a.nJ) B::lds_st((m.isJ ? L.spubJ : L.spubI) + szh_slot<B>(I, J), (unsigned)gcols[l]);
            }
        }
        // progress words (one per pencil and face: [pencil][0] J-face, [pencil][1] I-face)
        SZH_FORL {
            const szh_rowmap<B> m(B::lane(l));
            if (m.valid && m.r == 0 && gsteps[l] < (1 << 30)) {
                int v = gsteps[l]; if (v > r2 + 14) v = r2 + 14;
                if (v != reported[l]) { reported[l] = v; B::st_gran(a.progress + penc[l] * 2 + (m.isJ ? 0 : 1), ((szh_u64)a.epoch << 32) | (unsigned)v); }
            }
        }
        if (B::all(fin)) break;
        if (B::all(none)) {
            if (++idle > (1u << 24)) { SZH_FORL { if (B::lane(l) == 0) B::st_flag(a.err, 1u); } break; }
            if ((idle & 4095u) == 0 && B::ld_flag(a.err) != 0) break;
            B::backoff(1);
        } else idle = 0;
    }
}

// FILL wavefront (loads only): follows the progress words of the producing pencils in the neighbouring tiles, fetches their
// granules and delivers them to this tile's virtual-producer rings.  Only granules that the progress word covers are
// requested (polling the granules themselves from every waiting tile floods the fabric); every granule is validated by its tag.
template <class T, class B>
SZH_HD void szh_tile_fill(const szh_qargs<T> &a, int TI, int TJ, const szh_tile_lds<T> &L)
{
    constexpr int NL = B::NL;
    constexpr int NW = szh_gran<T>::NW;
    constexpr int KF = sizeof(T) == 8 ? 16 : 8;   // (measured: 8 for float, 16 for double)                     // columns per row per round
    const int r0 = a.G.g0.count, r1 = a.G.g1.count, r2 = a.G.g2.count;
    const int stride = B::face_stride(r2);
    int fk[NL], cslot[NL], wbase[NL];
    bool en[NL];
    const szh_u64 *src[NL];
    // (see the STORE wavefront: 16-byte buffer loads with 32-bit offsets from the first producing row of each face group)
    // (two granules below the group's first row: a 16-byte pair may start one granule before a row)
    const szh_u64 *const gbaseJ = a.faceJ + ((int64_t)(TI * B::TPI) * a.nJ + (TJ > 0 ? TJ * B::TPJ - 1 : 0)) * 8 * (int64_t)r2 * NW - 2;
    const szh_u64 *const gbaseI = a.faceI + ((int64_t)(TI > 0 ? TI * B::TPI - 1 : 0) * a.nJ + TJ * B::TPJ) * 9 * (int64_t)r2 * NW - 2;
    const typename B::gbuf_t bufJ = B::make_gbuf(gbaseJ), bufI = B::make_gbuf(gbaseI);
    const bool wide = a.wide != 0;
    const szh_u64 *prog[NL];
    SZH_FORL {
        const szh_rowmap<B> m(B::lane(l));
        // J-face rows feed the pencils of the tile's FIRST column (from the tile to the left), I-face rows its FIRST row (from above)
        const int I = m.isJ ? TI * B::TPI + m.pp : TI * B::TPI, J = m.isJ ? TJ * B::TPJ : TJ * B::TPJ + m.pp;
        en[l] = m.valid && I < a.nI && J < a.nJ && !(SZH_DEV && (a.dbg == 1 || a.dbg == 2));
        if (m.isJ) en[l] = en[l] && J > 0 && 8 * I + m.r < r0;
        else en[l] = en[l] && I > 0 && (m.r < 8 ? 8 * J + m.r < r1 : J > 0);
        const int64_t pp = m.isJ ? (int64_t)I * a.nJ + (J - 1) : (int64_t)(I - 1) * a.nJ + J;   // the producing pencil
        src[l] = en[l] ? (m.isJ ? a.faceJ + (pp * 8 + m.r) * (int64_t)r2 * NW : a.faceI + (pp * 9 + m.r) * (int64_t)r2 * NW) : a.faceJ;
        prog[l] = a.progress + (en[l] ? pp * 2 + (m.isJ ? 0 : 1) : 0);
        cslot[l] = en[l] ? szh_slot<B>(I, J) : 0;
        const int vslot = m.isJ ? szh_vslot_left<B>(I) : szh_vslot_top<B>(J);
        wbase[l] = vslot * stride + m.ringrow * B::face_rowstride(r2);
        fk[l] = en[l] ? 0 : r2;
    }
    auto get2 = [&](bool isJ, const szh_u64 *p, szh_u64 &wa, szh_u64 &wb) {   // two granules at a 16-byte boundary
        if (wide) {
            if (isJ) B::ld_gran2_b(bufJ, (unsigned)((p - gbaseJ) * 8), wa, wb); else B::ld_gran2_b(bufI, (unsigned)((p - gbaseI) * 8), wa, wb);
        } else { wa = B::ld_gran(p); wb = B::ld_gran(p + 1); }
    };
    unsigned idle = 0;
    int flog = 0;
    for (;;) {
        // how far have the producers got?  (every lane asks for its own row's producer: a handful of distinct words per round)
        szh_u64 g[KF][NW][NL];
        bool none[NL], fin[NL];
        int vsteps[NL];
        SZH_FORL {
            const szh_rowmap<B> m(B::lane(l));
            int avail = 0;
            if (en[l] && fk[l] < r2) {
                const szh_u64 p = B::ld_gran(prog[l]);
                const int ps = (unsigned)(p >> 32) == a.epoch ? (int)(unsigned)p : 0;
                avail = ps - m.sr; if (avail > r2) avail = r2;
                const int room = (int)B::lds_ld(L.cstep + cslot[l]) - 7 + B::RL;      // slots of columns < consumer step - 7 are free again
                if (avail > room) avail = room;
            }
            int n = avail - fk[l]; if (n > KF) n = KF; if (n < 0) n = 0;
            // 16-byte loads (two granules) wherever possible, see STORE
            if (NW == 2) {
                SZH_UNROLL
                for (int e = 0; e < KF; ++e) {
                    szh_u64 wa = 0, wb = 0;
                    if (e < n) get2(m.isJ, src[l] + (int64_t)(fk[l] + e) * 2, wa, wb);
                    g[e][0][l] = wa; g[e][NW - 1][l] = wb;
                }
            } else {
                const szh_u64 *const p0 = src[l] + fk[l];
                const int odd = (int)(((uintptr_t)p0 >> 3) & 1);
                szh_u64 q[KF + 2];                                           // granules p0 - odd ... (pairs at 16-byte boundaries)
                SZH_UNROLL
                for (int j = 0; j <= KF / 2; ++j) {
                    szh_u64 wa = 0, wb = 0;
                    if (n > 0 && 2 * j - odd < n && (j < KF / 2 || odd)) get2(m.isJ, p0 - odd + 2 * j, wa, wb);
                    q[2 * j] = wa; q[2 * j + 1] = wb;
                }
                SZH_UNROLL
                for (int e = 0; e < KF; ++e) g[e][0][l] = e < n ? (odd ? q[e + 1] : q[e]) : 0;
            }
            int lead = 0; bool run = true;
            SZH_UNROLL
            for (int e = 0; e < KF; ++e) {
                bool v = e < n;
                szh_u64 w2[NW];
                SZH_UNROLL
                for (int w = 0; w < NW; ++w) { w2[w] = g[e][w][l]; v = v && ((unsigned)(w2[w] >> 32) == a.epoch); }
                run = run && v;                    // a row advances by its run of leading delivered columns
                if (run) { B::lds_st(L.faces + wbase[l] + B::ring(fk[l] + e + m.sr), szh_gran<T>::unpack(w2)); ++lead; }
            }
            fk[l] += lead;
            none[l] = lead == 0; fin[l] = fk[l] >= r2;
            // the consumer reads column k of this row at its step k + hs: "steps covered" on the scale of the step counters (+7)
            vsteps[l] = en[l] ? (fk[l] >= r2 ? (1 << 29) : fk[l] + m.hs + 7) : (1 << 30);
        }
        if (SZH_DEV && a.trace && TI == (a.trace_tile >> 16) && TJ == (a.trace_tile & 0xffff) && flog < SZH_TRACE_LOG) {
            SZH_FORL { if (B::lane(l) == 0) { szh_u64 *lg = a.trace + (int64_t)a.nI * a.nJ * 8 + 256 + 2 * 2 * SZH_TRACE_LOG; lg[2 * flog] = B::clock(); lg[2 * flog + 1] = (szh_u64)fk[l]; } }
            ++flog;
        }
        B::lds_fence();          // the values are in the ring before the counter says so
        int gv[NL];
        szh_group_min<B>(L.scratch + 64, vsteps, gv);
        SZH_FORL {
            const szh_rowmap<B> m(B::lane(l));
            if (m.valid && m.r == 0) {
                const int I = m.isJ ? TI * B::TPI + m.pp : TI * B::TPI, J = m.isJ ? TJ * B::TPJ : TJ * B::TPJ + m.pp;
                const int vslot = m.isJ ? szh_vslot_left<B>(I) : szh_vslot_top<B>(J);
                B::lds_st(L.cstep + vslot, (unsigned)gv[l]);
            }
        }
        if (B::all(fin)) break;
        if (B::all(none)) {
            if (++idle > (1u << 22)) { SZH_FORL { if (B::lane(l) == 0) B::st_flag(a.err, 1u); } break; }
            if ((idle & 1023u) == 0 && B::ld_flag(a.err) != 0) break;
            B::backoff(a.backoff);
        } else idle = 0;
    }
}

// the t-th entry of szh_fill_pencil_order(nI, nJ), computed: no table, no memory round trip in front of a tile's start
SZH_HD unsigned szh_pencil_order_at(int nI, int nJ, unsigned t)
{
    const int m = nI < nJ ? nI : nJ, M = nI < nJ ? nJ : nI;
    const unsigned tri = (unsigned)m * (unsigned)(m - 1) / 2;         // diagonals 0 .. m-2 (lengths 1 .. m-1)
    const unsigned mid = (unsigned)(M - m + 1) * (unsigned)m;         // diagonals m-1 .. M-1, m tiles each
    int d; unsigned off;
    if (t < tri) {
        d = (int)((__builtin_sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((unsigned)(d + 1) * (unsigned)(d + 2) / 2 <= t) ++d;
        while ((unsigned)d * (unsigned)(d + 1) / 2 > t) --d;
        off = t - (unsigned)d * (unsigned)(d + 1) / 2;
    } else if (t < tri + mid) { d = m - 1 + (int)((t - tri) / (unsigned)m); off = (t - tri) % (unsigned)m; }
    else {                                                            // the shrinking part, counted from the end
        const unsigned total = (unsigned)nI * (unsigned)nJ, r = total - 1 - t;
        int e = (int)((__builtin_sqrtf(8.0f * (float)r + 1.0f) - 1.0f) * 0.5f);
        while ((unsigned)(e + 1) * (unsigned)(e + 2) / 2 <= r) ++e;
        while ((unsigned)e * (unsigned)(e + 1) / 2 > r) --e;
        d = nI + nJ - 2 - e;
        off = (unsigned)e - (r - (unsigned)e * (unsigned)(e + 1) / 2);
    }
    int lo = d - (nJ - 1); if (lo < 0) lo = 0;
    const int I = lo + (int)off;
    return ((unsigned)I << 16) | (unsigned)(d - I);
}

// anti-diagonal start order (of the tiles): every dependency has a smaller ticket
inline void szh_fill_pencil_order(int nI, int nJ, unsigned *order)
{
    int n = 0;
    for (int d = 0; d <= nI + nJ - 2; ++d) {
        int lo = d - (nJ - 1); if (lo < 0) lo = 0;
        int hi = d < nI - 1 ? d : nI - 1;
        for (int I = lo; I <= hi; ++I) order[n++] = ((unsigned)I << 16) | (unsigned)(d - I);
    }
}
