// szh_sim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
// A CPU lane simulator that instantiates the exact kernel bodies of sz_amd/csrc/szh_pencil.h with
// NL = 64 lanes per "wavefront" and runs pencils sequentially in ticket order, so that the
// index/shuffle/halo logic of the HIP kernel can be checked against the oracle without a GPU.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../sz_amd/csrc/szh_pencil.h"
#include "../../sz_amd/csrc/szh_core.h"
#include <cmath>

struct SimBackend {
    static constexpr int NL = 64;
    static int lane(int l) { return l; }
    template <class T> static void shfl_up(T (&dst)[64], const T (&src)[64], int d)
    {
        T tmp[64];
        for (int l = 0; l < 64; ++l) tmp[l] = (l >= d) ? src[l - d] : src[l];
        for (int l = 0; l < 64; ++l) dst[l] = tmp[l];
    }
    template <class T> static void shfl_up1(T (&dst)[64], const T (&src)[64]) { shfl_up(dst, src, 1); }
    template <class T> static T readlane(const T (&src)[64], int lane) { return src[lane]; }
    // tiles of 2 x 3 pencils.  The simulator runs a tile's FILL, its pencils and its STORE one after the other, so a ring must
    // keep every column: RL covers all of dim2.
    static constexpr int TPI = 2, TPJ = 3, RL = 1 << 20;
    static int ring(int k) { return k; }
    static int face_rowstride(int r2) { return r2 + 32; }   // ring position = the producing step: < r2 + 14 rounded up to whole trips
    static int face_stride(int r2) { return (r2 + 32) * SZH_FROWS; }
    template <class E> static E lds_ld(const E *p) { return *p; }
    template <class E> static E lds_ld_u(const E *p) { return *p; }
    template <class E> static void lds_st(E *p, E v) { *p = v; }
    static void lds_fence() {}
    static void lds_order() {}
    template <class E> static void touch(E &) {}
    static bool all(const bool (&p)[64]) { for (int l = 0; l < 64; ++l) if (!p[l]) return false; return true; }
    static szh_u64 ld_gran(const szh_u64 *p) { return *p; }
    static void st_gran(szh_u64 *p, szh_u64 v) { *p = v; }
    typedef const szh_u64 *gbuf_t;
    static gbuf_t make_gbuf(const szh_u64 *base) { return base; }
    static void st_gran2_b(gbuf_t b, unsigned off, szh_u64 x, szh_u64 y) { szh_u64 *p = const_cast<szh_u64 *>(b) + off / 8; p[0] = x; p[1] = y; }
    static void ld_gran2_b(gbuf_t b, unsigned off, szh_u64 &x, szh_u64 &y) { x = b[off / 8]; y = b[off / 8 + 1]; }
    static unsigned ld_flag(const unsigned *p) { return *p; }
    static void st_flag(unsigned *p, unsigned v) { *p = v; }
    static szh_u64 ld_sys_u64(const szh_u64 *p) { return *p; }
    template <class U> static U ld_coef(const U *p) { return *p; }
    static void backoff(int) {}
    static void nap() {}
    template <class E, int N> static void ld16(const E *p, E (&v)[N]) { memcpy(v, p, 16); }
    template <class E, int N> static void st16(E *p, const E (&v)[N]) { memcpy(p, v, 16); }
    static szh_u64 clock() { return 0; }
    static szh_u64 where() { return 0; }
};

template <class T, bool DEC>
static int run_all(szh_qargs<T> a)
{
    const int r0 = a.G.g0.count, r1 = a.G.g1.count, r2 = a.G.g2.count;
    a.nI = (r0 + 7) / 8; a.nJ = (r1 + 7) / 8;
    const size_t ng = (size_t)a.nI * a.nJ * 9 * r2 * szh_gran<T>::NW;
    std::vector<szh_u64> fI(ng, 0), fJ(ng, 0);
    a.faceI = fI.data(); a.faceJ = fJ.data(); a.epoch = 7;
    using B = SimBackend;
    const int nTI = (a.nI + B::TPI - 1) / B::TPI, nTJ = (a.nJ + B::TPJ - 1) / B::TPJ;
    std::vector<unsigned> order((size_t)nTI * nTJ);
    szh_fill_pencil_order(nTI, nTJ, order.data());
    unsigned err = 0; a.err = &err;
    std::vector<szh_u64> prog((size_t)a.nI * a.nJ * 2, 0);
    a.progress = prog.data(); a.backoff = 1; a.wide = getenv("SZH_SIM_NARROW") ? 0 : 1;
    constexpr int NP = B::TPI * B::TPJ, NV = B::TPI + B::TPJ;
    std::vector<uint16_t> ring((size_t)NP * (SZH_XC + 1) * 64);
    // "LDS" of one tile: face arrays [r2][SZH_FROWS] per slot, poisoned so that a value used before it is written shows
    const size_t fsz = (size_t)(NP + NV) * B::face_stride(r2);
    std::vector<T> faces(fsz + SZH_FTRASH);
    std::vector<unsigned> cstep(NP + NV), spubJ(NP), spubI(NP);
    std::vector<int> scratch(128);
    for (size_t tk = 0; tk < order.size(); ++tk) {
        const int TI = (int)(order[tk] >> 16), TJ = (int)(order[tk] & 0xffff);
        std::fill(ring.begin(), ring.end(), (uint16_t)0xDEAD);
        std::fill(faces.begin(), faces.end(), (T)-777);
        std::fill(cstep.begin(), cstep.end(), 0u); std::fill(spubJ.begin(), spubJ.end(), 0u); std::fill(spubI.begin(), spubI.end(), 0u);
        szh_tile_lds<T> L{ring.data(), faces.data(), (int)fsz, cstep.data(), spubJ.data(), spubI.data(), scratch.data()};
        szh_tile_fill<T, B>(a, TI, TJ, L);
        for (int pi = 0; pi < B::TPI; ++pi)
            for (int pj = 0; pj < B::TPJ; ++pj) {
                const int I = TI * B::TPI + pi, J = TJ * B::TPJ + pj;
                if (I < a.nI && J < a.nJ) szh_pencil_run<T, DEC, B>(a, I, J, L);
            }
        szh_tile_store<T, B>(a, TI, TJ, L);
    }
    return (int)err;
}

template <class T>
static int quantize(const T *data, int r0, int r1, int r2, double eb, int cap, int use_mean, double mean,
                    const uint8_t *blk_lor, const T *coef, uint16_t *codes_nat)
{
    szh_qargs<T> a; memset(&a, 0, sizeof(a));
    a.G = szh_make_geom3(r0, r1, r2);
    a.data = data; a.codes = codes_nat; a.blk_lor = blk_lor; a.coef = coef; a.coef_stride = a.G.nblocks;
    a.eb = (T)eb; a.recip = 1 / a.eb; a.mean = (T)mean; a.cap = cap; a.radius = cap / 2; a.use_mean = use_mean;
    return run_all<T, false>(a);
}
template <class T>
static int reconstruct(T *out, int r0, int r1, int r2, double eb, int cap, int use_mean, double mean,
                       const uint8_t *blk_lor, const T *coef, const uint16_t *codes_nat)
{
    szh_qargs<T> a; memset(&a, 0, sizeof(a));
    a.G = szh_make_geom3(r0, r1, r2);
    a.out = out; a.codes = const_cast<uint16_t *>(codes_nat); a.blk_lor = blk_lor; a.coef = coef; a.coef_stride = a.G.nblocks;
    a.eb = (T)eb; a.recip = 1 / a.eb; a.mean = (T)mean; a.cap = cap; a.radius = cap / 2; a.use_mean = use_mean;
    return run_all<T, true>(a);
}

extern "C" {
int szh_sim_quantize_f32(const float *d, int r0, int r1, int r2, double eb, int cap, int um, double mean,
                         const uint8_t *bl, const float *coef, uint16_t *codes)
{ return quantize<float>(d, r0, r1, r2, eb, cap, um, mean, bl, coef, codes); }
int szh_sim_quantize_f64(const double *d, int r0, int r1, int r2, double eb, int cap, int um, double mean,
                         const uint8_t *bl, const double *coef, uint16_t *codes)
{ return quantize<double>(d, r0, r1, r2, eb, cap, um, mean, bl, coef, codes); }
int szh_sim_reconstruct_f32(float *o, int r0, int r1, int r2, double eb, int cap, int um, double mean,
                            const uint8_t *bl, const float *coef, const uint16_t *codes)
{ return reconstruct<float>(o, r0, r1, r2, eb, cap, um, mean, bl, coef, codes); }
int szh_sim_reconstruct_f64(double *o, int r0, int r1, int r2, double eb, int cap, int um, double mean,
                            const uint8_t *bl, const double *coef, const uint16_t *codes)
{ return reconstruct<double>(o, r0, r1, r2, eb, cap, um, mean, bl, coef, codes); }

// natural <-> block order maps (plain loops over the geometry; reference for the GPU permute kernels)
void szh_sim_nat_to_blk_u16(const uint16_t *nat, int r0, int r1, int r2, int32_t *blk)
{
    szh_geom3 G = szh_make_geom3(r0, r1, r2);
    for (int b0 = 0; b0 < G.g0.num; ++b0) for (int b1 = 0; b1 < G.g1.num; ++b1) for (int b2 = 0; b2 < G.g2.num; ++b2) {
        int64_t p = szh_code_base(G, b0, b1, b2);
        int o0 = szh_blk_start(G.g0, b0), o1 = szh_blk_start(G.g1, b1), o2 = szh_blk_start(G.g2, b2);
        int s0 = szh_blk_size(G.g0, b0), s1 = szh_blk_size(G.g1, b1), s2 = szh_blk_size(G.g2, b2);
        for (int i = 0; i < s0; ++i) for (int j = 0; j < s1; ++j) for (int k = 0; k < s2; ++k)
            blk[p++] = nat[(int64_t)(o0 + i) * G.d0 + (int64_t)(o1 + j) * G.d1 + (o2 + k)];
    }
}
void szh_sim_blk_to_nat_u16(const int32_t *blk, int r0, int r1, int r2, uint16_t *nat)
{
    szh_geom3 G = szh_make_geom3(r0, r1, r2);
    for (int b0 = 0; b0 < G.g0.num; ++b0) for (int b1 = 0; b1 < G.g1.num; ++b1) for (int b2 = 0; b2 < G.g2.num; ++b2) {
        int64_t p = szh_code_base(G, b0, b1, b2);
        int o0 = szh_blk_start(G.g0, b0), o1 = szh_blk_start(G.g1, b1), o2 = szh_blk_start(G.g2, b2);
        int s0 = szh_blk_size(G.g0, b0), s1 = szh_blk_size(G.g1, b1), s2 = szh_blk_size(G.g2, b2);
        for (int i = 0; i < s0; ++i) for (int j = 0; j < s1; ++j) for (int k = 0; k < s2; ++k)
            nat[(int64_t)(o0 + i) * G.d0 + (int64_t)(o1 + j) * G.d1 + (o2 + k)] = (uint16_t)blk[p++];
    }
}
}

// ---- fit / select / sampling through the shared per-block functions (global-memory accessor) ----
template <class T> struct GAcc {
    const T *base; int64_t d0, d1;
    T operator()(int i, int j, int k) const { return base[(int64_t)i * d0 + (int64_t)j * d1 + k]; }
};
template <class T>
static void fit_select(const T *data, int r0, int r1, int r2, double eb, int use_mean, double mean, T *coef, uint8_t *blk_lor)
{
    szh_geom3 G = szh_make_geom3(r0, r1, r2);
    const T noise = (T)((T)eb * 1.22);
    int64_t b = 0;
    for (int b0 = 0; b0 < G.g0.num; ++b0) for (int b1 = 0; b1 < G.g1.num; ++b1) for (int b2 = 0; b2 < G.g2.num; ++b2, ++b) {
        GAcc<T> A{data + (int64_t)szh_blk_start(G.g0, b0) * G.d0 + (int64_t)szh_blk_start(G.g1, b1) * G.d1 + szh_blk_start(G.g2, b2), G.d0, G.d1};
        T c4[4];
        szh_fit_block<T>(A, szh_blk_size(G.g0, b0), szh_blk_size(G.g1, b1), szh_blk_size(G.g2, b2), c4);
        for (int e = 0; e < 4; ++e) coef[e * G.nblocks + b] = c4[e];
        blk_lor[b] = szh_select_block<T>(A, szh_blk_size(G.g0, b0), szh_blk_size(G.g1, b1), szh_blk_size(G.g2, b2), c4, noise, use_mean, (T)mean) ? 0 : 1;
    }
}
template <class T>
static double sample(const T *data, int r0, int r1, int r2, double ebD, int sd, unsigned max_radius,
                     uint32_t *radius_hist, uint32_t *freq_hist, uint64_t *within, uint64_t *count)
{
    szh_geom3 G = szh_make_geom3(r0, r1, r2);
    // strided mean, closed-form positions, sequential sum
    szh_meanwalk w = szh_make_meanwalk(G.n, G.d0, r2, (int64_t)(int)std::sqrt((double)G.n));
    T mean = 0; int64_t mc = 0;
    for (int64_t m = 0;; ++m) { int64_t p = szh_meanwalk_pos(w, m); if (p >= G.n) break; mean += data[p]; mc++; }
    if (mc > 0) mean /= (T)mc;
    const int64_t nrows = szh_sample_row_limit(G, sd);
    *within = 0; *count = 0;
    for (int64_t ridx = 0; ridx < nrows; ++ridx) {
        const int64_t n1 = ridx / (r1 - 1) + 1, n2 = ridx % (r1 - 1) + 1;
        const int64_t c0 = sd - ((n1 + n2) % sd);
        for (int64_t m = 0;; ++m) {
            const int64_t col = c0 + m * sd;
            if (m > 0 && col >= r2) break;
            const int64_t pos = n1 * G.d0 + n2 * r2 + col;
            if (pos >= G.n) break;
            unsigned ri; int fi, we;
            szh_sample_point<T>(data, pos, r2, G.d0, ebD, mean, max_radius, &ri, &fi, &we);
            radius_hist[ri]++; freq_hist[fi]++; *within += we; (*count)++;
        }
    }
    return (double)mean;
}
extern "C" {
void szh_sim_fit_select_f32(const float *d, int r0, int r1, int r2, double eb, int um, double mean, float *coef, uint8_t *bl)
{ fit_select<float>(d, r0, r1, r2, eb, um, mean, coef, bl); }
void szh_sim_fit_select_f64(const double *d, int r0, int r1, int r2, double eb, int um, double mean, double *coef, uint8_t *bl)
{ fit_select<double>(d, r0, r1, r2, eb, um, mean, coef, bl); }
double szh_sim_sample_f32(const float *d, int r0, int r1, int r2, double eb, int sd, unsigned mr, uint32_t *rh, uint32_t *fh, uint64_t *w, uint64_t *c)
{ return sample<float>(d, r0, r1, r2, eb, sd, mr, rh, fh, w, c); }
double szh_sim_sample_f64(const double *d, int r0, int r1, int r2, double eb, int sd, unsigned mr, uint32_t *rh, uint32_t *fh, uint64_t *w, uint64_t *c)
{ return sample<double>(d, r0, r1, r2, eb, sd, mr, rh, fh, w, c); }
}
