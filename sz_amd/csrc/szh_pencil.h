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
// l-1 / l-8 / l-9 one or two steps earlier and travel by cross-lane shuffles -- no LDS, no barrier.
// Pencil (I,J) needs the faces of pencils (I-1,J), (I,J-1) and one corner column of (I-1,J-1);
// these travel through HBM-side "granules": 8-byte {epoch tag, value bits} words written with one
// agent-scope store and polled with agent-scope loads (data is its own flag; MI355X_MICROARCH
// "handoff-1to1"/R2).  Pencils are started in anti-diagonal order through an atomic ticket so
// every dependency is already resident: no deadlock, no grid barrier.
//
// Tiles.  A workgroup owns a TILE of TPI x TPJ neighbouring pencils (float: 4 x 4 = 16 wavefronts = one whole CU;
// double: 4 x 2), one wavefront per pencil.  Faces between pencils of the same tile never leave the CU: the producer
// writes them into an LDS ring and bumps an LDS step counter, the consumer reads them a few hundred cycles later.
// Only faces that cross a tile boundary travel as granules through the fabric, so the chain of slow hops across a
// 512 x 512 cross-section is 30 long instead of 126.  The third producer of a pencil, (I-1,J-1), is not needed: its
// corner column is the row-7 halo of (I-1,J), which forwards it as a ninth row of its I-face.
//
// The body is written once and instantiated by two back ends:
//   * the HIP kernel (szh_kernels.hip): NL = 1 value per thread, collectives = DPP/bpermute;
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
    const T *coef;            // decoded regression coefficients, SoA [4][nblocks]
    T eb, recip, mean;
    int cap, radius, use_mean;
    szh_u64 *faceI, *faceJ;   // granule buffers: faceJ [pencil][8 rows][r2][NW], faceI [pencil][9 rows][r2][NW] (row 8 = forwarded corner column)
    unsigned epoch;
    int nI, nJ;
    const unsigned *order;    // ticket -> (tile row << 16) | tile column, anti-diagonal order over the tiles
    unsigned *ticket;
    unsigned *err;            // set to 1 if a halo wait timed out
    szh_u64 *progress;        // per pencil: {epoch, steps completed}; a cheap "has my neighbour got going" word
    int gate_steps;           // a pencil starts once both producers have completed this many steps
    int backoff;              // sleep units between two polls of a missing granule
    int tripgate;             // 1: a tile-edge pencil sleeps at the top of a trip until its producers' progress covers the trip
    int dbg;                  // development timing experiments (results become WRONG): 1 = no hand-off at all, 2 = no publishing stores
    szh_u64 *trace;           // optional (development): per pencil {t_start, t_gate, t_first, t_end, spins, naps, cu, 0}
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

// LDS of one pencil inside its tile's workgroup.  B::ring(k) maps a column to its ring slot; B::face_slot(I,J,nJ) indexes the
// per-pencil arrays (tile-local on the GPU, global in the CPU simulator).
template <class T> struct szh_tile_lds {
    uint16_t *cring;          // [SZH_XC + 1][64] this pencil's quantisation codes in flight (+ trash column)
    T *faces;                 // [slots][RL][SZH_FROWS] face rings of every pencil of the tile: rows 0-7 = J-face (il), 8-15 = I-face (jl),
                              // 16 = corner column forwarded to the pencil below
    int ftrash;               // element offset in faces[] of 64 write-only / don't-care slots (one per lane): masked-off lanes go there,
                              // so that the ring accesses need no exec-mask juggling
    unsigned *cstep;          // [slots] steps completed (all face values of those steps are in the ring)
};
#define SZH_FROWS 17

#define SZH_U 16 /* steps per loop trip: a trip first requests all of its inputs (values and halo granules), then steps */
#if defined(__HIPCC__)
#define SZH_UNROLL _Pragma("unroll")
#else
#define SZH_UNROLL
#endif
#define SZH_FORL for (int l = 0; l < NL; ++l)
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

// B: back end. Requires: NL, lane(l), shfl_up(dst,src,d), readlane(src,lane), all(pred),
//    ld_gran(p), st_gran(p,v), ld_flag(p), st_flag(p,v), backoff(n), nap(), clock(), where(),
//    ld16(p, T(&)[16/sizeof T]), st16(p, const T(&)[...]) -- one 16-byte vector access (4-byte aligned for 4/8-byte T),
//    lds_ld(p), lds_st(p,v), lds_ld_u(p) (wavefront-uniform address), lds_fence() (orders this wavefront's LDS accesses),
//    RL / ring(k) (face-ring length), face_stride(r2) (elements per pencil slot), TPI / TPJ (tile shape), face_slot(I,J,nJ), same_tile(I,J,I2,J2).
// HASREG: the pencil touches at least one regression block (otherwise the block bookkeeping and the
//         regression quantiser are compiled out); USEMEAN: the stream's use_mean flag.
template <class T, bool DEC, bool HASREG, bool USEMEAN, class B>
SZH_HD void szh_pencil_body(const szh_qargs<T> &a, int I, int J, const szh_tile_lds<T> &L)
{
    uint16_t *const cring = L.cring;
    constexpr int NL = B::NL;
    constexpr int NW = szh_gran<T>::NW;
    const szh_geom3 &G = a.G;
    const int r0 = G.g0.count, r1 = G.g1.count, r2 = G.g2.count;
    const int nbz = G.g2.num;
    const int cap_lor = a.cap - 2, cap_reg = a.cap, radius = a.radius;
    const T eb = a.eb, recip = a.recip, mean = a.mean;
    const bool pubJ = (J + 1 < a.nJ), pubI = (I + 1 < a.nI);

    // ---- this pencil's neighbours: inside the tile (LDS) or across a tile boundary (granules) ----
    const bool predJ_lds = J > 0 && B::same_tile(I, J, I, J - 1), predI_lds = I > 0 && B::same_tile(I, J, I - 1, J);
    const bool consJ_lds = pubJ && B::same_tile(I, J, I, J + 1), consI_lds = pubI && B::same_tile(I, J, I + 1, J);
    const int myslot = B::face_slot(I, J, a.nJ);
    const int slotPJ = predJ_lds ? B::face_slot(I, J - 1, a.nJ) : 0, slotPI = predI_lds ? B::face_slot(I - 1, J, a.nJ) : 0;
    const int slotCJ = consJ_lds ? B::face_slot(I, J + 1, a.nJ) : 0, slotCI = consI_lds ? B::face_slot(I + 1, J, a.nJ) : 0;
    const int mybase = myslot * B::face_stride(r2);
    const bool has_gran = (J > 0 && !predJ_lds) || (I > 0 && !predI_lds);    // this pencil reads granules (it sits on a tile edge)

    // ---- per-lane constants ----
    int il[NL], jl[NL], skew[NL], hskew[NL], hkind[NL], hlds[NL], pjb[NL], pib[NL], pcb[NL], trash[NL], ctrash[NL];
    bool inb[NL], pJ[NL], pI[NL], pC[NL];
    int64_t rowoff[NL], blkrow[NL], hoff[NL], pubJoff[NL], pubIoff[NL];
    const szh_u64 *hbuf[NL];
    T fii[NL], fjj[NL];
    SZH_FORL {
        const int lane = B::lane(l);
        il[l] = lane >> 3; jl[l] = lane & 7;
        const int i = 8 * I + il[l], j = 8 * J + jl[l];
        inb[l] = (i < r0) && (j < r1);
        const int ic = i < r0 ? i : r0 - 1, jc = j < r1 ? j : r1 - 1;
        const int b0 = szh_blk_of(G.g0, ic), b1 = szh_blk_of(G.g1, jc);
        fii[l] = (T)(ic - szh_blk_start(G.g0, b0));
        fjj[l] = (T)(jc - szh_blk_start(G.g1, b1));
        rowoff[l] = (int64_t)ic * G.d0 + (int64_t)jc * G.d1;
        blkrow[l] = ((int64_t)b0 * G.g1.num + b1) * nbz;
        skew[l] = il[l] + jl[l];
        // halo role: which face row this lane fetches (for k = t - hskew), and from where: hkind 0 none, 1 granules, 2 LDS ring
        hkind[l] = 0; hoff[l] = 0; hskew[l] = 0; hbuf[l] = a.faceJ; hlds[l] = -1;
        trash[l] = L.ftrash + lane; ctrash[l] = SZH_XC * 64 + lane;
        const int64_t pencJ = (int64_t)I * a.nJ + (J - 1), pencI = (int64_t)(I - 1) * a.nJ + J;
        if (jl[l] == 0 && J > 0 && i < r0) {           // (i, 8J-1, k): J-face of pencil (I,J-1), row il
            hskew[l] = il[l];
            if (predJ_lds) { hkind[l] = 2; hlds[l] = slotPJ * B::face_stride(r2) + il[l]; }
            else { hkind[l] = 1; hbuf[l] = a.faceJ; hoff[l] = (pencJ * 8 + il[l]) * r2; }
        } else if (il[l] == 0 && jl[l] > 0 && I > 0 && j < r1) { // (8I-1, j, k): I-face of (I-1,J), row jl
            hskew[l] = jl[l];
            if (predI_lds) { hkind[l] = 2; hlds[l] = slotPI * B::face_stride(r2) + 8 + jl[l]; }
            else { hkind[l] = 1; hbuf[l] = a.faceI; hoff[l] = (pencI * 9 + jl[l]) * r2; }
        } else if (lane == 63 && I > 0) {               // for lane 0: (8I-1, 8J, k) = I-face of (I-1,J), row 0
            if (predI_lds) { hkind[l] = 2; hlds[l] = slotPI * B::face_stride(r2) + 8; }
            else { hkind[l] = 1; hbuf[l] = a.faceI; hoff[l] = (pencI * 9 + 0) * r2; }
        } else if (lane == 62 && I > 0 && J > 0) {      // for lane 0: (8I-1, 8J-1, k), forwarded by (I-1,J) as its ninth row
            if (predI_lds) { hkind[l] = 2; hlds[l] = slotPI * B::face_stride(r2) + 16; }
            else { hkind[l] = 1; hbuf[l] = a.faceI; hoff[l] = (pencI * 9 + 8) * r2; }
        }
        pubJoff[l] = (((int64_t)I * a.nJ + J) * 8 + il[l]) * r2;
        pubIoff[l] = (((int64_t)I * a.nJ + J) * 9 + jl[l]) * r2;
        pJ[l] = pubJ && jl[l] == 7 && inb[l];
        pI[l] = pubI && il[l] == 7 && inb[l];
        pC[l] = pubI && J > 0 && il[l] == 7 && jl[l] == 0 && inb[l];   // forwards its own halo value: the corner column of (I+1,J)
        if (a.dbg == 1) { hkind[l] = 0; hlds[l] = -1; pJ[l] = false; pI[l] = false; pC[l] = false; }
        if (a.dbg == 2) { pJ[l] = false; pI[l] = false; pC[l] = false; }
        // LDS ring offsets of the rows this lane publishes (-1: none)
        pjb[l] = (pJ[l] && consJ_lds) ? mybase + il[l] : -1;
        pib[l] = (pI[l] && consI_lds) ? mybase + 8 + jl[l] : -1;
        pcb[l] = (pC[l] && consI_lds) ? mybase + 16 : -1;
    }
    const bool gran_pubJ = pubJ && !consJ_lds && a.dbg == 0, gran_pubI = pubI && !consI_lds && a.dbg == 0;
    const int64_t pubCoff = (((int64_t)I * a.nJ + J) * 9 + 8) * r2;

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
                ca[l] = a.coef[b]; cb[l] = a.coef[G.nblocks + b]; cc[l] = a.coef[2 * G.nblocks + b]; cd[l] = a.coef[3 * G.nblocks + b];
                pbase[l] = ca[l] * fii[l] + cb[l] * fjj[l];
            }
            if (!nlor[l]) {
                na[l] = a.coef[b + 1]; nb_[l] = a.coef[G.nblocks + b + 1]; nc[l] = a.coef[2 * G.nblocks + b + 1]; nd[l] = a.coef[3 * G.nblocks + b + 1];
            }
        }
    }

    // ---- per-trip inputs.  Nothing stays in flight across the loop back-edge (hipcc guards rotated in-flight registers
    //      with s_waitcnt vmcnt(0) = one memory round trip per STEP); a trip requests everything it needs at its top.
    //      Memory shape: every lane moves its OWN row's next SZH_U values as 16-byte vectors (64 B contiguous per lane
    //      instead of sixteen 4-byte requests); the 2-byte codes go through a small LDS ring and are moved between ring
    //      and HBM as aligned 16-byte row segments (a per-step 2-byte store per lane was 134 M separate L2 write requests).
    constexpr int VPT = 16 / (int)sizeof(T);      // values per 16-byte vector
    constexpr int NVEC = SZH_U / VPT;             // vectors per lane per trip
    T xr[SZH_U][NL];
    T ov[SZH_U][NL];
    szh_u64 hr[SZH_U][NW][NL];
    const bool vec_codes = (r2 % 8) == 0;         // rows of the u16 code array are 16-byte aligned

    // this lane's row, steps t0..t0+SZH_U-1  <->  k = t0 - skew .. t0 - skew + SZH_U - 1
    // compress: the original values -> xr; decompress: the pre-scattered unpredictable values -> ov (overwritten step by step)
    auto load_x = [&](int t0) {
        const T *src = DEC ? a.out : a.data;
        T (&dst)[SZH_U][NL] = DEC ? ov : xr;
        SZH_FORL {
            const int k0 = t0 - skew[l];
            if (inb[l] && k0 >= 0 && k0 + SZH_U <= r2) {
                SZH_UNROLL
                for (int v = 0; v < NVEC; ++v) {
                    T tmp[VPT];
                    B::ld16(src + rowoff[l] + k0 + v * VPT, tmp);
                    SZH_UNROLL
                    for (int e = 0; e < VPT; ++e) dst[v * VPT + e][l] = tmp[e];
                }
            } else {
                SZH_UNROLL
                for (int s = 0; s < SZH_U; ++s) {
                    const int k = k0 + s;
                    dst[s][l] = (inb[l] && (unsigned)k < (unsigned)r2) ? src[rowoff[l] + k] : (T)0;
                }
            }
        }
    };
    auto store_out = [&](int t0) {
        SZH_FORL {
            const int k0 = t0 - skew[l];
            if (inb[l] && k0 >= 0 && k0 + SZH_U <= r2) {
                SZH_UNROLL
                for (int v = 0; v < NVEC; ++v) {
                    T tmp[VPT];
                    SZH_UNROLL
                    for (int e = 0; e < VPT; ++e) tmp[e] = ov[v * VPT + e][l];
                    B::st16(a.out + rowoff[l] + k0 + v * VPT, tmp);
                }
            } else {
                SZH_UNROLL
                for (int s = 0; s < SZH_U; ++s) {
                    const int k = k0 + s;
                    if (inb[l] && (unsigned)k < (unsigned)r2) a.out[rowoff[l] + k] = ov[s][l];
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
        if (vec_codes && c0 + 8 <= r2) {
            SZH_FORL {
                const int row = B::lane(l);
                int64_t off;
                if (row_off(row, off)) {
                    uint16_t tmp[8];
                    if (DEC) {
                        B::ld16(a.codes + off + c0, tmp);
                        SZH_UNROLL
                        for (int e = 0; e < 8; ++e) cring[((c0 + e) % SZH_XC) * 64 + row] = tmp[e];
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
        }
    };

    auto load_halo = [&](int t, szh_u64 (&hv)[NW][NL]) {    // granule-sourced rows only
        SZH_FORL {
            const int kh = t - hskew[l];
            const bool hact = hkind[l] == 1 && (unsigned)kh < (unsigned)r2;
            SZH_UNROLL
            for (int w = 0; w < NW; ++w) hv[w][l] = hact ? B::ld_gran(hbuf[l] + (hoff[l] + kh) * NW + w) : 0;
        }
    };
    // bounded wait until an LDS step counter reaches `need`; returns the value seen
    auto wait_cstep = [&](int slot, int need) -> int {
        int v = (int)B::lds_ld_u(L.cstep + slot);
        unsigned spins = 0;
        while (v < need && !a.dbg) {
            if (++spins > (1u << 22)) { SZH_FORL { if (B::lane(l) == 0) B::st_flag(a.err, 1u); } break; }
            if ((spins & 1023u) == 0 && B::ld_flag(a.err) != 0) break;
            B::backoff(1);
            v = (int)B::lds_ld_u(L.cstep + slot);
        }
        return v;
    };

    // rolling neighbour state (values at the previous step)
    T cur[NL], A1[NL], B1[NL], C1[NL];
    SZH_FORL { cur[l] = 0; A1[l] = 0; B1[l] = 0; C1[l] = 0; }

    const int tsteps = r2 + 14;
    int flushed = 0, filled = 0;   // code-ring columns already written back (compress) / already brought in (decompress)
    szh_u64 tr_start = 0, tr_gate = 0, tr_first = 0, tr_spins = 0, tr_naps = 0;
    if (a.trace) tr_start = B::clock();
    // ---- start gate: sleep (one lane polls ONE word per producer, long naps) until the producers are under way.
    // Thousands of queued wavefronts polling granules instead would flood the fabric and slow every hand-off.
    {
        const int need = a.dbg ? 0 : (a.gate_steps < tsteps ? a.gate_steps : tsteps);
        const szh_u64 want = ((szh_u64)a.epoch << 32) | (unsigned)need;
        unsigned naps = 0;
        for (;;) {
            bool ok[NL];
            if (a.dbg) break;
            SZH_FORL {
                ok[l] = true;
                const int lane = B::lane(l);
                if (lane == 0 && J > 0 && !predJ_lds) { const szh_u64 p = B::ld_gran(a.progress + ((int64_t)I * a.nJ + (J - 1))); ok[l] = (p >> 32) == a.epoch && p >= want; }
                if (lane == 1 && I > 0 && !predI_lds) { const szh_u64 p = B::ld_gran(a.progress + ((int64_t)(I - 1) * a.nJ + J)); ok[l] = (p >> 32) == a.epoch && p >= want; }
            }
            if (B::all(ok)) break;
            if (++naps > (1u << 18) || B::ld_flag(a.err) != 0) break; // bounded; the granule waits below still guard correctness
            B::nap();
        }
        tr_naps = naps;
    }
    if (a.trace) tr_gate = B::clock();
    const bool detail = a.trace && I == a.nI / 2 && J == a.nJ / 2;
    szh_u64 *dt = a.trace ? a.trace + (int64_t)a.nI * a.nJ * 8 : nullptr;
    for (int t0 = 0; t0 < tsteps; t0 += SZH_U) {
        if (detail && t0 / SZH_U < 64) { SZH_FORL { if (B::lane(l) == 0) dt[(t0 / SZH_U) * 4 + 0] = B::clock(); } }
        // in-tile consumers must have read the ring slots this trip overwrites (they read column k no later than their step k+7)
        if (consJ_lds) wait_cstep(slotCJ, t0 + SZH_U - B::RL);
        if (consI_lds) wait_cstep(slotCI, t0 + SZH_U - B::RL);
        // how far are the in-tile producers?  (step t needs their step t+7 finished)
        int pstepJ = predJ_lds ? (int)B::lds_ld_u(L.cstep + slotPJ) : (1 << 30);
        int pstepI = predI_lds ? (int)B::lds_ld_u(L.cstep + slotPI) : (1 << 30);
        if (DEC) { while (filled < t0 + SZH_U && filled < r2) { move_codes(filled); filled += 8; } } // ring holds columns < filled
        load_x(t0);
        // granule-sourced rows (tile edge): request this trip's 16 granules per row.  If some are not there yet, this pencil
        // has caught up with a producer in another tile: SLEEP until that producer's progress word says the whole trip is
        // covered, and ask again.  (Polling granule by granule inside the steps costs a fabric round trip per step and
        // throttles the whole tile behind this wavefront.)
        if (has_gran) {
            const int needp = t0 + SZH_U + 7 < tsteps ? t0 + SZH_U + 7 : tsteps;
            unsigned tries = 0;
            for (;;) {
                szh_u64 pw[NL];
                SZH_FORL {
                    const int lane = B::lane(l);
                    pw[l] = ~0ull;
                    if (lane == 0 && J > 0 && !predJ_lds) pw[l] = B::ld_gran(a.progress + ((int64_t)I * a.nJ + (J - 1)));
                    if (lane == 1 && I > 0 && !predI_lds) pw[l] = B::ld_gran(a.progress + ((int64_t)(I - 1) * a.nJ + J));
                }
                SZH_UNROLL
                for (int s = 0; s < SZH_U; ++s) load_halo(t0 + s, hr[s]);
                bool ok[NL];
                SZH_FORL {
                    ok[l] = true;
                    if (hkind[l] == 1) {
                        SZH_UNROLL
                        for (int s = 0; s < SZH_U; ++s) {
                            const int kh = t0 + s - hskew[l];
                            if ((unsigned)kh < (unsigned)r2) {
                                SZH_UNROLL
                                for (int w = 0; w < NW; ++w) ok[l] = ok[l] && ((unsigned)(hr[s][w][l] >> 32) == a.epoch);
                            }
                        }
                    }
                }
                if (B::all(ok) || a.dbg || !a.tripgate) break;
                const szh_u64 p0 = B::readlane(pw, 0), p1 = B::readlane(pw, 1);
                const szh_u64 want = ((szh_u64)a.epoch << 32) | (unsigned)needp;
                const bool covered = (p0 == ~0ull || ((p0 >> 32) == a.epoch && p0 >= want)) && (p1 == ~0ull || ((p1 >> 32) == a.epoch && p1 >= want));
                if (covered) break;          // the stragglers (stores still in flight) are picked up by the per-step wait below
                if (++tries > (1u << 20)) { SZH_FORL { if (B::lane(l) == 0) B::st_flag(a.err, 1u); } break; }
                if ((tries & 255u) == 0 && B::ld_flag(a.err) != 0) break;
                B::nap();
                ++tr_naps;
            }
        }
        SZH_UNROLL
        for (int s = 0; s < SZH_U; ++s) {
            const int t = t0 + s;
            // -- halo for this step.  Fast path: the granule requested at the top of the trip already carries this launch's epoch.
            //    Slow path (cold): poll with fresh loads into a private copy; it never touches the ring registers.
            T hval[NL];
            {
                // in-tile producers: their step t+7 must be complete (no halo is needed once t - hskew >= r2)
                const int need = t + 8 < tsteps ? t + 8 : tsteps;
                if (pstepJ < need) pstepJ = wait_cstep(slotPJ, need);
                if (pstepI < need) pstepI = wait_cstep(slotPI, need);
                // LDS-sourced rows: every lane reads (lanes without such a row read a don't-care slot)
                SZH_FORL {
                    const int kh = t - hskew[l];
                    const bool lact = hlds[l] >= 0 && (unsigned)kh < (unsigned)r2;
                    const T v = B::lds_ld(L.faces + (lact ? hlds[l] + B::ring(kh) * SZH_FROWS : trash[l]));
                    hval[l] = lact ? v : (T)0;
                }
                if (has_gran) {       // granule-sourced rows (tile edge).  Fast path: the granule requested at the top of the trip is valid
                    szh_u64 g[NW][NL];
                    bool ok[NL], gact[NL];
                    SZH_FORL {
                        const int kh = t - hskew[l];
                        gact[l] = hkind[l] == 1 && (unsigned)kh < (unsigned)r2;
                        bool v = true;
                        SZH_UNROLL
                        for (int w = 0; w < NW; ++w) { g[w][l] = hr[s][w][l]; v = v && ((unsigned)(g[w][l] >> 32) == a.epoch); }
                        ok[l] = !gact[l] || v;
                    }
                    if (!B::all(ok) && !a.dbg) {
                        unsigned spins = 0;
                        for (;;) {
                            // bounded wait: a lost hand-off must end the launch, not hang the GPU
                            if (++spins > (1u << 20)) { SZH_FORL { if (!ok[l]) B::st_flag(a.err, 1u); } break; }
                            if ((spins & 255u) == 0 && B::ld_flag(a.err) != 0) break; // another wavefront already gave up
                            B::backoff(a.backoff);
                            SZH_FORL {
                                if (!ok[l]) {
                                    const int kh = t - hskew[l];
                                    bool v = true;
                                    SZH_UNROLL
                                    for (int w = 0; w < NW; ++w) {
                                        g[w][l] = B::ld_gran(hbuf[l] + (hoff[l] + kh) * NW + w);
                                        v = v && ((unsigned)(g[w][l] >> 32) == a.epoch);
                                    }
                                    ok[l] = v;
                                }
                            }
                            if (B::all(ok)) break;
                        }
                        tr_spins += spins;
                    }
                    SZH_FORL {
                        szh_u64 w2[NW];
                        SZH_UNROLL
                        for (int w = 0; w < NW; ++w) w2[w] = g[w][l];
                        if (gact[l]) hval[l] = szh_gran<T>::unpack(w2);
                    }
                }
            }
            // -- neighbours through cross-lane moves (values of the previous step) --
            T shA[NL], shB[NL], shCi[NL], shCj[NL];
            B::shfl_up(shA, cur, 1);
            B::shfl_up(shB, cur, 8);
            B::shfl_up(shCi, A1, 8);
            B::shfl_up(shCj, B1, 1);
            const T h63 = B::readlane(hval, 63), h62 = B::readlane(hval, 62);

            SZH_FORL {
                const int k = t - skew[l];
                const bool act = inb[l] && (unsigned)k < (unsigned)r2;
                const int lane = B::lane(l);
                const T nA = jl[l] > 0 ? shA[l] : hval[l];
                const T nB = il[l] > 0 ? shB[l] : (lane == 0 ? h63 : hval[l]);
                const T nC = il[l] > 0 ? shCi[l] : (jl[l] > 0 ? shCj[l] : h62);
                // [-1] + [-s1] + [-s0] - [-s1-1] - [-s0-1] - [-s0-s1] + [-s0-s1-1], left to right
                const T pred = cur[l] + nA + nB - A1[l] - B1[l] - nC + C1[l];
                T predr = 0;
                if (HASREG) predr = pbase[l] + cc[l] * (T)kk[l] + cd[l];
                const bool is_lor = HASREG ? lor[l] : true;
                T nv;
                if (!DEC) {
                    const T x = xr[s][l];
                    T rcl;
                    int code = szh_quant_sel<T>(x, pred, eb, recip, cap_lor, radius, &rcl);
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
                    B::lds_st(cring + (act ? ((unsigned)k % SZH_XC) * 64 + B::lane(l) : ctrash[l]), (uint16_t)code);
                } else {
                    const int cread = (int)B::lds_ld(cring + (act ? ((unsigned)k % SZH_XC) * 64 + B::lane(l) : ctrash[l]));
                    const int c0 = act ? cread : radius;
                    int c = c0;
                    const T p = is_lor ? pred : predr;
                    bool is_mean = false;
                    if (USEMEAN) {
                        is_mean = is_lor && (c == radius);
                        if (is_lor && c != 0 && c < radius) c += 1;                 // szd_float.c:3784
                    }
                    nv = p + (T)(2 * (c - radius)) * eb;
                    if (USEMEAN && is_mean) nv = mean;
                    if (act && c0 == 0) nv = ov[s][l];                              // pre-scattered unpredictable value (read at the top of the trip)
                    ov[s][l] = nv;
                }
                // publish faces for the pencils to the right / below.  Inside the tile: LDS ring, every lane stores (lanes without a
                // face row, or outside the k range, store to their trash slot); across a tile boundary: granules
                if (consJ_lds) B::lds_st(L.faces + ((act && pjb[l] >= 0) ? pjb[l] + B::ring(k) * SZH_FROWS : trash[l]), nv);
                if (consI_lds) {
                    B::lds_st(L.faces + ((act && pib[l] >= 0) ? pib[l] + B::ring(k) * SZH_FROWS : trash[l]), nv);
                    // lane (7,0): its halo value IS the corner column of the pencil below, same k
                    B::lds_st(L.faces + ((act && pcb[l] >= 0) ? pcb[l] + B::ring(k) * SZH_FROWS : trash[l]), hval[l]);
                }
                if (gran_pubJ || gran_pubI) {
                    szh_u64 w2[NW];
                    szh_gran<T>::pack(nv, a.epoch, w2);
                    if (gran_pubJ && act && pJ[l]) { SZH_UNROLL for (int w = 0; w < NW; ++w) B::st_gran(a.faceJ + (pubJoff[l] + k) * NW + w, w2[w]); }
                    if (gran_pubI && act && pI[l]) { SZH_UNROLL for (int w = 0; w < NW; ++w) B::st_gran(a.faceI + (pubIoff[l] + k) * NW + w, w2[w]); }
                    if (gran_pubI && act && pC[l]) {
                        szh_u64 w3[NW];
                        szh_gran<T>::pack(hval[l], a.epoch, w3);
                        SZH_UNROLL for (int w = 0; w < NW; ++w) B::st_gran(a.faceI + (pubCoff + k) * NW + w, w3[w]);
                    }
                }
                // roll the neighbour state.  Lanes outside the k range see zero inputs and zero neighbours, so
                // they produce zeros by themselves; only the mean shortcut and stale pre-scattered values need masking.
                if (detail && s == 0 && t0 / SZH_U < 64 && B::lane(l) == 0) dt[(t0 / SZH_U) * 4 + 1] = B::clock();
                cur[l] = (USEMEAN || DEC) ? (act ? nv : (T)0) : nv;
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
                                na[l] = a.coef[b + 1]; nb_[l] = a.coef[G.nblocks + b + 1];
                                nc[l] = a.coef[2 * G.nblocks + b + 1]; nd[l] = a.coef[3 * G.nblocks + b + 1];
                            }
                            nnlor[l] = (bk[l] + 2 < nbz) ? (a.blk_lor[b + 2] != 0) : true;
                        }
                    }
                }
            }
            // step t is complete: its face values are in the ring (LDS executes a wavefront's accesses in order)
            if (consJ_lds || consI_lds || predJ_lds || predI_lds) {   // consumers wait for it; producers watch it for ring space
                B::lds_fence();
                SZH_FORL { B::lds_st(L.cstep + myslot, (unsigned)(t + 1)); }
            }
            // half-way through the trip: write back the code columns that are complete (keeps the 32-column ring from wrapping)
            if (!DEC && s == SZH_U / 2 - 1) { while (flushed + 8 <= t + 1 - 14 && flushed < r2) { move_codes(flushed); flushed += 8; } }
            if (s == SZH_U / 2 - 1 && (gran_pubJ || gran_pubI)) {
                SZH_FORL { if (B::lane(l) == 0) B::st_gran(a.progress + ((int64_t)I * a.nJ + J), ((szh_u64)a.epoch << 32) | (unsigned)(t + 1)); }
            }
        }
        if (detail && t0 / SZH_U < 64) { SZH_FORL { if (B::lane(l) == 0) dt[(t0 / SZH_U) * 4 + 2] = B::clock(); } }
        if (DEC) store_out(t0);
        else {
            // after this trip every lane is past column t0 + SZH_U - 15: flush the 8-column groups that are complete
            while (flushed + 8 <= t0 + SZH_U - 14 && flushed < r2) { move_codes(flushed); flushed += 8; }
        }
        if (detail && t0 / SZH_U < 64) { SZH_FORL { if (B::lane(l) == 0) dt[(t0 / SZH_U) * 4 + 3] = B::clock(); } }
        if (a.trace && t0 == 0) tr_first = B::clock();
        // tell the consumers how far this pencil has got (one word, one lane)
        if ((pubJ && !consJ_lds) || (pubI && !consI_lds)) {
            SZH_FORL { if (B::lane(l) == 0) B::st_gran(a.progress + ((int64_t)I * a.nJ + J), ((szh_u64)a.epoch << 32) | (unsigned)(t0 + SZH_U)); }
        }
    }
    if (!DEC) { for (; flushed < r2; flushed += 8) move_codes(flushed); }
    if (a.trace) {
        const szh_u64 tr_end = B::clock();
        SZH_FORL {
            if (B::lane(l) == 0) {
                szh_u64 *tp = a.trace + ((int64_t)I * a.nJ + J) * 8;
                tp[0] = tr_start; tp[1] = tr_gate; tp[2] = tr_first; tp[3] = tr_end; tp[4] = tr_spins; tp[5] = tr_naps; tp[6] = B::where(); tp[7] = 0;
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
template <class T, bool DEC, class B>
SZH_HD void szh_pencil_run(const szh_qargs<T> &a, int I, int J, const szh_tile_lds<T> &L)
{
#ifdef SZH_EXP_NOREG
    const bool hasreg = false;
#else
    const bool hasreg = szh_pencil_has_reg<T, B>(a, I, J);
#endif
    if (a.use_mean) {
        if (hasreg) szh_pencil_body<T, DEC, true, true, B>(a, I, J, L);
        else szh_pencil_body<T, DEC, false, true, B>(a, I, J, L);
    } else {
        if (hasreg) szh_pencil_body<T, DEC, true, false, B>(a, I, J, L);
        else szh_pencil_body<T, DEC, false, false, B>(a, I, J, L);
    }
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
