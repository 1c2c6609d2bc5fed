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
    szh_u64 *faceI, *faceJ;   // granule buffers: [pencil][8 rows][r2][NW]
    unsigned epoch;
    int nI, nJ;
    const unsigned *order;    // ticket -> (I<<16)|J, anti-diagonal order
    unsigned *ticket;
    unsigned *err;            // set to 1 if a halo wait timed out
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

#define SZH_U 8 /* steps per software-pipelined chunk */
#if defined(__HIPCC__)
#define SZH_UNROLL _Pragma("unroll")
#else
#define SZH_UNROLL
#endif
#define SZH_FORL for (int l = 0; l < NL; ++l)

// B: back end. Requires: NL, lane(l), shfl_up(dst,src,d), readlane(src,lane), all(pred),
//    ld_gran(p), st_gran(p,v), ld_flag(p), st_flag(p,v), backoff().
template <class T, bool DEC, class B>
SZH_HD void szh_pencil_run(const szh_qargs<T> &a, int I, int J)
{
    constexpr int NL = B::NL;
    constexpr int NW = szh_gran<T>::NW;
    const szh_geom3 &G = a.G;
    const int r0 = G.g0.count, r1 = G.g1.count, r2 = G.g2.count;
    const int nbz = G.g2.num;
    const int cap_lor = a.cap - 2, cap_reg = a.cap, radius = a.radius;
    const T eb = a.eb, recip = a.recip, mean = a.mean;
    const bool use_mean = a.use_mean != 0;
    const bool pubJ = (J + 1 < a.nJ), pubI = (I + 1 < a.nI);

    // ---- per-lane constants ----
    int il[NL], jl[NL], skew[NL], hskew[NL];
    bool inb[NL], hrole[NL];
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
        // halo role: which granule row this lane fetches, and for which k = t - hskew
        hrole[l] = false; hoff[l] = 0; hskew[l] = 0; hbuf[l] = a.faceJ;
        if (jl[l] == 0 && J > 0 && i < r0) {           // (i, 8J-1, k): J-face of pencil (I,J-1), row il
            hrole[l] = true; hbuf[l] = a.faceJ; hskew[l] = il[l];
            hoff[l] = (((int64_t)I * a.nJ + (J - 1)) * 8 + il[l]) * r2;
        } else if (il[l] == 0 && jl[l] > 0 && I > 0 && j < r1) { // (8I-1, j, k): I-face of (I-1,J), row jl
            hrole[l] = true; hbuf[l] = a.faceI; hskew[l] = jl[l];
            hoff[l] = (((int64_t)(I - 1) * a.nJ + J) * 8 + jl[l]) * r2;
        } else if (lane == 63 && I > 0) {               // for lane 0: (8I-1, 8J, k)
            hrole[l] = true; hbuf[l] = a.faceI; hskew[l] = 0;
            hoff[l] = (((int64_t)(I - 1) * a.nJ + J) * 8 + 0) * r2;
        } else if (lane == 62 && I > 0 && J > 0) {      // for lane 0: (8I-1, 8J-1, k)
            hrole[l] = true; hbuf[l] = a.faceI; hskew[l] = 0;
            hoff[l] = (((int64_t)(I - 1) * a.nJ + (J - 1)) * 8 + 7) * r2;
        }
        pubJoff[l] = (((int64_t)I * a.nJ + J) * 8 + il[l]) * r2;
        pubIoff[l] = (((int64_t)I * a.nJ + J) * 8 + jl[l]) * r2;
    }

    // ---- per-lane block tracking along dim2 ----
    int kk[NL], bz[NL], bk[NL];
    bool lor[NL], nlor[NL], nnlor[NL];
    T ca[NL], cb[NL], cc[NL], cd[NL], na[NL], nb_[NL], nc[NL], nd[NL], pbase[NL];
    SZH_FORL {
        bk[l] = 0; kk[l] = 0; bz[l] = szh_blk_size(G.g2, 0);
        ca[l] = cb[l] = cc[l] = cd[l] = 0; na[l] = nb_[l] = nc[l] = nd[l] = 0; pbase[l] = 0;
        lor[l] = nlor[l] = nnlor[l] = true;
        if (inb[l]) {
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

    // ---- software-pipelined inputs: values / codes and halo granules for SZH_U steps ----
    T xcur[SZH_U][NL], xnext[SZH_U][NL];
    uint16_t qcur[SZH_U][NL], qnext[SZH_U][NL];
    szh_u64 hcur[SZH_U][NW][NL], hnext[SZH_U][NW][NL];

    auto fetch = [&](int tbase, T (&xv)[SZH_U][NL], uint16_t (&qv)[SZH_U][NL], szh_u64 (&hv)[SZH_U][NW][NL]) {
SZH_UNROLL
        for (int s = 0; s < SZH_U; ++s) {
            SZH_FORL {
                const int k = tbase + s - skew[l];
                const bool act = inb[l] && k >= 0 && k < r2;
                if (DEC) { qv[s][l] = act ? a.codes[rowoff[l] + k] : (uint16_t)1; xv[s][l] = 0; }
                else { xv[s][l] = act ? a.data[rowoff[l] + k] : (T)0; qv[s][l] = 0; }
                const int kh = tbase + s - hskew[l];
                const bool hact = hrole[l] && kh >= 0 && kh < r2;
SZH_UNROLL
                for (int w = 0; w < NW; ++w)
                    hv[s][w][l] = hact ? B::ld_gran(hbuf[l] + (hoff[l] + kh) * NW + w) : 0;
            }
        }
    };

    // rolling neighbour state (values at the previous step)
    T cur[NL], A1[NL], B1[NL], C1[NL];
    SZH_FORL { cur[l] = 0; A1[l] = 0; B1[l] = 0; C1[l] = 0; }

    const int tsteps = r2 + 14;
    fetch(0, xcur, qcur, hcur);
    for (int t0 = 0; t0 < tsteps; t0 += SZH_U) {
        if (t0 + SZH_U < tsteps) fetch(t0 + SZH_U, xnext, qnext, hnext);
SZH_UNROLL
        for (int s = 0; s < SZH_U; ++s) {
            const int t = t0 + s;
            // -- halo for this step: wait until the producer's granules carry this launch's epoch --
            T hval[NL];
            {
                bool ok[NL];
                unsigned spins = 0;
                for (;;) {
                    SZH_FORL {
                        const int kh = t - hskew[l];
                        const bool hact = hrole[l] && kh >= 0 && kh < r2;
                        bool v = true;
SZH_UNROLL
                        for (int w = 0; w < NW; ++w) v = v && ((unsigned)(hcur[s][w][l] >> 32) == a.epoch);
                        ok[l] = !hact || v;
                    }
                    if (B::all(ok)) break;
                    // bounded wait: a lost hand-off must end the launch, not hang the GPU
                    if (++spins > (1u << 20)) { SZH_FORL { if (!ok[l]) B::st_flag(a.err, 1u); } break; }
                    if ((spins & 255u) == 0 && B::ld_flag(a.err) != 0) break; // another wavefront already gave up
                    B::backoff();
                    SZH_FORL {
                        if (!ok[l]) {
                            const int kh = t - hskew[l];
SZH_UNROLL
                            for (int w = 0; w < NW; ++w) hcur[s][w][l] = B::ld_gran(hbuf[l] + (hoff[l] + kh) * NW + w);
                        }
                    }
                }
                SZH_FORL {
                    const int kh = t - hskew[l];
                    const bool hact = hrole[l] && kh >= 0 && kh < r2;
                    szh_u64 w2[NW];
SZH_UNROLL
                    for (int w = 0; w < NW; ++w) w2[w] = hcur[s][w][l];
                    hval[l] = hact ? szh_gran<T>::unpack(w2) : (T)0;
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
                const bool act = inb[l] && k >= 0 && k < r2;
                const int lane = B::lane(l);
                const T nA = jl[l] > 0 ? shA[l] : hval[l];
                const T nB = il[l] > 0 ? shB[l] : (lane == 0 ? h63 : hval[l]);
                const T nC = il[l] > 0 ? shCi[l] : (jl[l] > 0 ? shCj[l] : h62);
                // [-1] + [-s1] + [-s0] - [-s1-1] - [-s0-1] - [-s0-s1] + [-s0-s1-1], left to right
                const T pred = cur[l] + nA + nB - A1[l] - B1[l] - nC + C1[l];
                const T fkk = (T)kk[l];
                const T predr = pbase[l] + cc[l] * fkk + cd[l];
                T nv;
                if (!DEC) {
                    const T x = xcur[s][l];
                    T rcl, rcr;
                    int cl = szh_quant_point<T>(x, pred, eb, recip, cap_lor, radius, &rcl);
                    const int cr = szh_quant_point<T>(x, predr, eb, recip, cap_reg, radius, &rcr);
                    if (use_mean) {
                        if (cl != 0 && cl <= radius) cl -= 1;               // sz_float.c:6944
                        if (szh_abs(x - mean) <= eb) { cl = radius; rcl = mean; } // sz_float.c:6929
                    }
                    const int code = lor[l] ? cl : cr;
                    nv = lor[l] ? rcl : rcr;
                    if (act) a.codes[rowoff[l] + k] = (uint16_t)code;
                } else {
                    int c = qcur[s][l];
                    const T p = lor[l] ? pred : predr;
                    bool is_mean = false;
                    if (lor[l] && use_mean) {
                        is_mean = (c == radius);
                        if (c != 0 && c < radius) c += 1;                   // szd_float.c:3784
                    }
                    nv = p + (T)(2 * (c - radius)) * eb;
                    if (is_mean) nv = mean;
                    if (act) {
                        if (qcur[s][l] == 0) nv = a.out[rowoff[l] + k];      // pre-scattered unpredictable value
                        else a.out[rowoff[l] + k] = nv;
                    }
                }
                // publish faces for the pencils to the right / below
                if (act) {
                    szh_u64 w2[NW];
                    szh_gran<T>::pack(nv, a.epoch, w2);
                    if (pubJ && jl[l] == 7) {
SZH_UNROLL
                        for (int w = 0; w < NW; ++w) B::st_gran(a.faceJ + (pubJoff[l] + k) * NW + w, w2[w]);
                    }
                    if (pubI && il[l] == 7) {
SZH_UNROLL
                        for (int w = 0; w < NW; ++w) B::st_gran(a.faceI + (pubIoff[l] + k) * NW + w, w2[w]);
                    }
                }
                // roll the neighbour state; lanes outside the array or the k range carry zeros
                cur[l] = act ? nv : (T)0;
                A1[l] = act ? nA : (T)0;
                B1[l] = act ? nB : (T)0;
                C1[l] = act ? nC : (T)0;
                // advance along dim2
                if (act) {
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
        }
        // rotate the software pipeline
SZH_UNROLL
        for (int s = 0; s < SZH_U; ++s) {
            SZH_FORL {
                xcur[s][l] = xnext[s][l]; qcur[s][l] = qnext[s][l];
SZH_UNROLL
                for (int w = 0; w < NW; ++w) hcur[s][w][l] = hnext[s][w][l];
            }
        }
    }
}

// anti-diagonal start order of the pencils: every dependency of a pencil has a smaller ticket
inline void szh_fill_pencil_order(int nI, int nJ, unsigned *order)
{
    int n = 0;
    for (int d = 0; d <= nI + nJ - 2; ++d) {
        int lo = d - (nJ - 1); if (lo < 0) lo = 0;
        int hi = d < nI - 1 ? d : nI - 1;
        for (int I = lo; I <= hi; ++I) order[n++] = ((unsigned)I << 16) | (unsigned)(d - I);
    }
}
