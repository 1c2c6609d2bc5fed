// szh_segenc.h -- the Huffman packing of the SZ 2.1 type array STRAIGHT FROM NATURAL ORDER (round 6).
//
// Reference: the quantisation codes are stored block by block (sz_float.c:7064, :7359), coded as one bit string (Huffman.c:205-308,
// encode_withTree), and the unpredictable values follow in the same order (sz_float.c:7288).
//
// Until round 5 a pass of its own (k_permute<0>) turned the sweep's natural-order code array into a block-ordered copy, which the packing
// passes (k_chunk_bits, k_encode32) and k_unpred then read: 2 N bytes written and read again, 0.33 ms of kernel time at 512^3 (0.5 beside
// the sweep), most of it left over behind the sweep's end.  But the block order only permutes codes WITHIN a block column (b0, b1): a
// column is s0 * s1 whole rows of the natural array and ONE contiguous range of the bit string.  So, with a workgroup per block column:
//   k_col_hist    (alphabets of up to SZH_COL_HBINS symbols) the code histogram every code book needs (Huffman.c:165-174), kept PER COLUMN;
//   k_hist_reduce their sum = the array's histogram -> host -> tree;
//   k_col_bits_h  a column's bits = sum(count * code length), its zero codes = count[0]                           -> col_bits, col_zeros
//   k_col_bits    (larger alphabets, after k_hist_u16) the same two numbers by a pass over the column's rows
//   k_col_scan    both exclusive scans, and the word of the stream at every column boundary cleared: the two workgroups that share it OR
//                 their bits in; every other word is written whole by exactly one workgroup -- the stream buffer needs no memset
//   k_col_encode  the column in SEGMENTS of `segb` blocks: a segment's rows into LDS (natural order, coalesced; the next segment's rows are
//                 on their way meanwhile), then read in block order -- a thread takes NR consecutive runs (a run = one row of one block,
//                 s2 codes side by side in LDS) --, packed as k_encode32 packs (lengths summed, one scan per round, 64-bit accumulator,
//                 words ORed into an LDS window, the window out in whole 4-byte words; a partial last word stays for the next round);
//                 the unpredictable values of its zero codes go to their places in the list on the way.
// No block-ordered copy, no k_permute, no k_unpred, no memset of the stream.
#pragma once

namespace szh_se {
constexpr int NR = 4;            // runs per thread and round
constexpr int INNER = 7;         // codes of a run in the fast form (block widths: 6 or 7 for every extent >= 42, and most below)
constexpr int PF = 8;            // 16-byte pieces of the next segment's rows a thread keeps on their way

struct col_t { int b0, b1, s0, s1, o0, o1, rows; };
struct seg_t {
    int bkbeg, bkend, kbeg, kend;
    int nE, E, L;                // the segment's first nE blocks are E = g2.early wide, the others L = g2.late
    int ka, nvec, pitch, kshift; // natural side: rows cover [ka, ka + nvec * VW) (vector boundaries); LDS row pitch in codes (even, pitch / 2 odd: rows
                                 // NR apart start in different banks)
    int nblk, nruns;
    int lg_nvec;                 // smallest power of two >= nvec, as its exponent
};
SZH_HD int seg_pitch(int width) { int p = (width + 1) & ~1; if (((p >> 1) & 1) == 0) p += 2; return p; }
SZH_HD col_t make_col(const szh_geom3 &G, int col)
{
    col_t c;
    c.b0 = col / G.g1.num; c.b1 = col - c.b0 * G.g1.num;
    c.s0 = szh_blk_size(G.g0, c.b0); c.s1 = szh_blk_size(G.g1, c.b1); c.o0 = szh_blk_start(G.g0, c.b0); c.o1 = szh_blk_start(G.g1, c.b1);
    c.rows = c.s0 * c.s1;
    return c;
}
// VW: codes per load (8 / 4 / 1: the contiguous extent a multiple of 8 / of 4 / anything)
SZH_HD seg_t make_seg(const szh_geom3 &G, int rows, int segi, int segb, int VW)
{
    seg_t s;
    s.bkbeg = segi * segb; s.bkend = s.bkbeg + segb < G.g2.num ? s.bkbeg + segb : G.g2.num;
    s.kbeg = szh_blk_start(G.g2, s.bkbeg); s.kend = s.bkend < G.g2.num ? szh_blk_start(G.g2, s.bkend) : G.g2.count;
    s.E = G.g2.early; s.L = G.g2.late;
    int nE = G.g2.split - s.bkbeg; if (nE < 0) nE = 0; if (nE > s.bkend - s.bkbeg) nE = s.bkend - s.bkbeg;
    s.nE = nE;
    s.ka = s.kbeg / VW * VW;
    const int kb = (s.kend + VW - 1) / VW * VW;
    s.nvec = (kb - s.ka) / VW;
    s.pitch = seg_pitch(kb - s.ka); s.kshift = s.kbeg - s.ka;
    s.nblk = s.bkend - s.bkbeg; s.nruns = s.nblk * rows;
    s.lg_nvec = 0; while ((1 << s.lg_nvec) < s.nvec) ++s.lg_nvec;
    return s;
}
// LDS bytes of the widest segment's rows (host: sizes the launch)
inline size_t seg_tile_bytes(const szh_geom3 &G, int segb, int VW)
{
    const size_t rows = (size_t)G.g0.early * G.g1.early;
    const int width = segb * G.g2.early + 2 * VW;
    return (rows * (size_t)seg_pitch(width) * 2 + 15) & ~(size_t)15;
}
// VW-code pieces of the widest segment's rows per thread (host: the prefetch keeps PF of them in registers)
inline size_t seg_pieces_per_thread(const szh_geom3 &G, int segb, int VW)
{
    const size_t rows = (size_t)G.g0.early * G.g1.early;
    const size_t nvec = ((size_t)segb * G.g2.early + 2 * VW) / VW + 1;
    size_t p2 = 1; while (p2 < nvec) p2 <<= 1;
    return (rows * p2 + 255) / 256;
}
inline size_t seg_window_words(const szh_geom3 &G, unsigned maxlen) { return (size_t)256 * NR * (size_t)G.g2.early * maxlen / 32 + 8; }      // (a round: 256 NR runs of at most g2.early codes)
}

#ifdef SZH_HIPSIM
static inline bool any_lane(bool p) { return __ballot(p ? 1 : 0) != 0ull; }
#else
__device__ __forceinline__ bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
#endif
// x / d by one multiplication: m = magic_of(d); exact while x * d < 2^32 (floor(x / d) = x * ceil(2^32 / d) >> 32)
__device__ __forceinline__ unsigned magic_of(unsigned d) { return d > 1u ? 0xffffffffu / d + 1u : 0u; }
__device__ __forceinline__ unsigned div_by(unsigned x, unsigned d, unsigned m) { return d > 1u ? (unsigned)(((u64)x * m) >> 32) : x; }

// ---- small alphabets: the histogram of every block column (one workgroup each, whole rows: every lane loads)
// LDS: [nbins][R = 1 << rshift replicas]
#define SZH_COL_HBINS 256
__global__ __launch_bounds__(256) void k_col_hist(szh_geom3 G, const uint16_t *__restrict__ codes, unsigned nbins, int rshift, int vw, unsigned *__restrict__ col_hist)
{
    SZH_DYN_SMEM(smem);
    unsigned *sh = reinterpret_cast<unsigned *>(smem);
    const unsigned R = 1u << rshift;
    for (unsigned i = threadIdx.x; i < (nbins << rshift); i += 256) sh[i] = 0;
    __syncthreads();
    const szh_se::col_t c = szh_se::make_col(G, (int)blockIdx.x);
    const int nvec = G.g2.count / vw;
    const int lane = (int)threadIdx.x & 63, wid = (int)threadIdx.x >> 6;
    const unsigned rep = threadIdx.x & (R - 1);
    const unsigned m_s1 = magic_of((unsigned)c.s1);
    for (int x = lane; x < nvec; x += 64) {
        for (int r0 = wid; r0 < c.rows; r0 += 16) {                 // four rows of this wavefront at a time: their loads are in flight together
            uint4 w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + 4 * q;
                w[q] = uint4{0u, 0u, 0u, 0u};
                if (r < c.rows) {
                    const int i = (int)div_by((unsigned)r, (unsigned)c.s1, m_s1), j = r - i * c.s1;
                    const uint16_t *src = codes + (int64_t)(c.o0 + i) * G.d0 + (int64_t)(c.o1 + j) * G.d1 + x * vw;
                    if (vw == 8) w[q] = *reinterpret_cast<const uint4 *>(src);
                    else if (vw == 4) { const uint2 t = *reinterpret_cast<const uint2 *>(src); w[q].x = t.x; w[q].y = t.y; }
                    else w[q].x = src[0];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (r0 + 4 * q < c.rows) {
                    const unsigned wv[4] = {w[q].x, w[q].y, w[q].z, w[q].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (e < vw) { const unsigned cd = (e & 1) ? wv[e >> 1] >> 16 : wv[e >> 1] & 0xffffu; atomicAdd(&sh[(cd << rshift) + rep], 1u); }
                }
            }
        }
    }
    __syncthreads();
    for (unsigned b = threadIdx.x; b < nbins; b += 256) {
        unsigned t = 0;
        for (unsigned q = 0; q < R; ++q) t += sh[(b << rshift) + q];
        col_hist[(size_t)blockIdx.x * nbins + b] = t;
    }
}
// the array's histogram = the sum of the columns' (a few workgroups, one atomic per bin each)
__global__ __launch_bounds__(256) void k_hist_reduce(const unsigned *__restrict__ col_hist, unsigned nbins, int64_t ncols, unsigned *__restrict__ hist)
{
    __shared__ unsigned acc[SZH_COL_HBINS];
    for (unsigned b = threadIdx.x; b < nbins; b += 256) acc[b] = 0;
    __syncthreads();
    // a thread walks columns of ONE bin (thread t: bin t % nbins, columns from t / nbins on in steps of the threads per bin): coalesced, no division in the loop
    const unsigned tpb = 256u / nbins > 0u ? 256u / nbins : 1u;            // threads per bin (nbins <= 256)
    if (threadIdx.x < tpb * nbins) {
        const unsigned bin = threadIdx.x % nbins, lane_col = threadIdx.x / nbins;
        unsigned t = 0;
        for (int64_t col = (int64_t)blockIdx.x * tpb + lane_col; col < ncols; col += (int64_t)gridDim.x * tpb) t += col_hist[(size_t)col * nbins + bin];
        if (t) atomicAdd(&acc[bin], t);
    }
    __syncthreads();
    for (unsigned b = threadIdx.x; b < nbins; b += 256) if (acc[b]) atomicAdd(&hist[b], acc[b]);
}
// a column's bits and zero codes from its histogram
__global__ __launch_bounds__(256) void k_col_bits_h(const unsigned *__restrict__ col_hist, const uint8_t *__restrict__ len, unsigned nbins, int64_t ncols, u64 *__restrict__ col_bits, u64 *__restrict__ col_zeros)
{
    __shared__ uint8_t llen[SZH_COL_HBINS];
    for (unsigned i = threadIdx.x; i < nbins; i += 256) llen[i] = len[i];
    __syncthreads();
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= ncols) return;
    const unsigned *h = col_hist + (size_t)e * nbins;
    u64 bits = 0;
    for (unsigned b = 0; b < nbins; ++b) bits += (u64)h[b] * llen[b];
    col_bits[e] = bits; col_zeros[e] = h[0];
}

// ---- any alphabet: sum of the code lengths / number of zero codes of every block column, by a pass over its rows.  len: code length per symbol
__global__ __launch_bounds__(256) void k_col_bits(szh_geom3 G, const uint16_t *__restrict__ codes, const uint8_t *__restrict__ len, unsigned nsym, int vw,
                                                  u64 *__restrict__ col_bits, u64 *__restrict__ col_zeros)
{
    SZH_DYN_SMEM(smem);
    __shared__ u64 red[8];
    uint8_t *llen = reinterpret_cast<uint8_t *>(smem);
    const bool tab_lds = nsym <= 16384;
    if (tab_lds) { for (unsigned i = threadIdx.x; i < nsym; i += 256) llen[i] = len[i]; __syncthreads(); }
    const szh_se::col_t c = szh_se::make_col(G, (int)blockIdx.x);
    const int nvec = G.g2.count / vw;
    const int lane = (int)threadIdx.x & 63, wid = (int)threadIdx.x >> 6;
    const unsigned m_s1 = magic_of((unsigned)c.s1);
    u64 bits = 0; unsigned zeros = 0;
    for (int x = lane; x < nvec; x += 64) {
        for (int r0 = wid; r0 < c.rows; r0 += 16) {
            uint4 w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + 4 * q;
                w[q] = uint4{0u, 0u, 0u, 0u};
                if (r < c.rows) {
                    const int i = (int)div_by((unsigned)r, (unsigned)c.s1, m_s1), j = r - i * c.s1;
                    const uint16_t *src = codes + (int64_t)(c.o0 + i) * G.d0 + (int64_t)(c.o1 + j) * G.d1 + x * vw;
                    if (vw == 8) w[q] = *reinterpret_cast<const uint4 *>(src);
                    else if (vw == 4) { const uint2 t = *reinterpret_cast<const uint2 *>(src); w[q].x = t.x; w[q].y = t.y; }
                    else w[q].x = src[0];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (r0 + 4 * q < c.rows) {
                    const unsigned wv[4] = {w[q].x, w[q].y, w[q].z, w[q].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (e < vw) {
                        const unsigned cd = (e & 1) ? wv[e >> 1] >> 16 : wv[e >> 1] & 0xffffu;
                        bits += tab_lds ? llen[cd] : len[cd]; zeros += cd == 0u ? 1u : 0u;
                    }
                }
            }
        }
    }
    bits = wave_sum_u64(bits); const u64 z = wave_sum_u64((u64)zeros);
    if (lane == 0) { red[wid] = bits; red[4 + wid] = z; }
    __syncthreads();
    if (threadIdx.x == 0) { col_bits[blockIdx.x] = red[0] + red[1] + red[2] + red[3]; col_zeros[blockIdx.x] = red[4] + red[5] + red[6] + red[7]; }
}

// the stream's word at every column boundary (and at the payload's end) starts as zero (the form for more columns than k_col_scan takes: after scan_u64)
__global__ __launch_bounds__(256) void k_col_bounds(const u64 *__restrict__ col_bitoff, int64_t nent, const u64 *__restrict__ total_bits, u64 bit0, unsigned *out32)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e > nent) return;
    const u64 at = bit0 + (e < nent ? col_bitoff[e] : *total_bits);
    out32[at >> 5] = 0u;
    if (e == nent) out32[(at >> 5) + 1] = 0u;
}
// both scans and the boundary words in ONE launch of one workgroup (up to 2^13 columns, eight per thread in registers -- sixteen spilled under the 1024-thread register bound --: six launches of the general scan less)
#define SZH_COL_SCAN_PER 8
__global__ __launch_bounds__(1024) void k_col_scan(const u64 *__restrict__ col_bits, const u64 *__restrict__ col_zeros, int nent, u64 *__restrict__ col_bitoff, u64 *__restrict__ col_zoff,
                                                   u64 *total_bits, u64 *total_zeros, u64 bit0, unsigned *out32)
{
    __shared__ u64 shb[16], shz[16];
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lo = tid * SZH_COL_SCAN_PER;
    u64 vb[SZH_COL_SCAN_PER], vz[SZH_COL_SCAN_PER];
#pragma unroll
    for (int q = 0; q < SZH_COL_SCAN_PER; ++q) { const bool in = lo + q < nent; vb[q] = in ? col_bits[lo + q] : 0ull; vz[q] = in ? col_zeros[lo + q] : 0ull; }
    u64 b = 0, z = 0;
#pragma unroll
    for (int q = 0; q < SZH_COL_SCAN_PER; ++q) { b += vb[q]; z += vz[q]; }
    u64 ib = b, iz = z;
    for (int o = 1; o < 64; o <<= 1) { const u64 tb = __shfl_up(ib, o, 64), tz = __shfl_up(iz, o, 64); if (lane >= o) { ib += tb; iz += tz; } }
    if (lane == 63) { shb[wid] = ib; shz[wid] = iz; }
    __syncthreads();
    u64 baseb = 0, basez = 0, totb = 0, totz = 0;
    for (int w = 0; w < 16; ++w) { if (w < wid) { baseb += shb[w]; basez += shz[w]; } totb += shb[w]; totz += shz[w]; }
    u64 rb = baseb + ib - b, rz = basez + iz - z;
#pragma unroll
    for (int q = 0; q < SZH_COL_SCAN_PER; ++q) {
        if (lo + q < nent) { col_bitoff[lo + q] = rb; col_zoff[lo + q] = rz; out32[(bit0 + rb) >> 5] = 0u; }
        rb += vb[q]; rz += vz[q];
    }
    if (tid == 0) { *total_bits = totb; *total_zeros = totz; out32[(bit0 + totb) >> 5] = 0u; out32[((bit0 + totb) >> 5) + 1] = 0u; }
}

// (the way back) the columns' zero counts (k_col_zeros) -> where each column's unpredictable values begin in the stream's list, and their number: one launch of one workgroup
// (up to 2^13 columns) in place of a widening pass and the general scan's three
__global__ __launch_bounds__(1024) void k_col_zscan(const unsigned *__restrict__ col_zeros, int nent, u64 *__restrict__ col_zoff, u64 *total_zeros)
{
    __shared__ u64 shz[16];
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lo = tid * SZH_COL_SCAN_PER;
    u64 vz[SZH_COL_SCAN_PER], z = 0;
#pragma unroll
    for (int q = 0; q < SZH_COL_SCAN_PER; ++q) { vz[q] = lo + q < nent ? (u64)col_zeros[lo + q] : 0ull; z += vz[q]; }
    u64 iz = z;
    for (int o = 1; o < 64; o <<= 1) { const u64 tz = __shfl_up(iz, o, 64); if (lane >= o) iz += tz; }
    if (lane == 63) shz[wid] = iz;
    __syncthreads();
    u64 basez = 0, totz = 0;
    for (int w = 0; w < 16; ++w) { if (w < wid) basez += shz[w]; totz += shz[w]; }
    u64 rz = basez + iz - z;
#pragma unroll
    for (int q = 0; q < SZH_COL_SCAN_PER; ++q) { if (lo + q < nent) col_zoff[lo + q] = rz; rz += vz[q]; }
    if (tid == 0) *total_zeros = totz;
}

// dynamic LDS: [(nsym + 1) x u64 table][a segment's rows][16 bytes: a run of the null symbol][window]
// table entry of symbol s: low word = code length (bit 16 set for symbol 0: the zero codes are counted in the same sum), high word = the code; entry nsym (the null
// symbol: places of a thread's share that hold no code) = 0
// WIDE: every block is 6 or 7 codes wide along the contiguous dimension (at least two blocks along it, none wider than 7: every extent >= 42, and most below): a run
// is read as four 4-byte words from the word boundary at or below it, the thread keeps its codes' table entries in registers between the two passes
// segs: the nseg segments' geometry (the same for every column: worked out once, on the host -- make_seg's divisions were a third of this kernel's scalar work)
template <class T, bool WIDE>
__global__ __launch_bounds__(256) void k_col_encode(szh_geom3 G, const uint16_t *__restrict__ codes, const u64 *__restrict__ table, unsigned nsym, const szh_se::seg_t *__restrict__ segs, int nseg, int vw,
                                                    size_t tile_bytes, unsigned win_words, const u64 *__restrict__ col_bitoff, const u64 *__restrict__ col_zoff, u64 bit0,
                                                    unsigned *out32, const T *__restrict__ data, T *__restrict__ unpred)
{
    using namespace szh_se;
    SZH_DYN_SMEM(smem);
    __shared__ u64 sh[8];
    __shared__ unsigned rowbase[128];                                // element offset of every row of the column (rows <= 121; the array: < 2^32 elements)
    u64 *ltab = reinterpret_cast<u64 *>(smem);
    uint16_t *tile = reinterpret_cast<uint16_t *>(smem + ((size_t)nsym + 1) * 8);
    uint16_t *nullrun = reinterpret_cast<uint16_t *>(smem + ((size_t)nsym + 1) * 8 + tile_bytes);
    unsigned *win = reinterpret_cast<unsigned *>(smem + ((size_t)nsym + 1) * 8 + tile_bytes + 16);
    const int tid = (int)threadIdx.x;
    const col_t c = make_col(G, (int)blockIdx.x);
    for (unsigned i = tid; i < nsym; i += 256) ltab[i] = table[i];
    if (tid == 0) ltab[nsym] = 0;
    if (tid < 8) nullrun[tid] = (uint16_t)nsym;
    for (unsigned w = tid; w < win_words; w += 256) win[w] = 0u;     // (from here on every word that is written out is cleared where it is read)
    const unsigned m_rows = magic_of((unsigned)c.rows);
    if (tid < c.rows) { const int i = tid / c.s1, j = tid - i * c.s1; rowbase[tid] = (unsigned)((int64_t)(c.o0 + i) * G.d0 + (int64_t)(c.o1 + j) * G.d1); }
    __syncthreads();
    // a segment's rows: piece p of the thread = vector (p * 256 + tid) of the rows x nvec of them; PF pieces travel in registers while the segment before is packed
    uint4 pf[PF];
    // (piece p of the thread = vector cv = x & mask of row r = x >> lg, x = p * 256 + tid: a power-of-two pitch for the vector index -- no division; up to a
    // third of the places hold no vector)
    auto fetch = [&](const seg_t &s) {
        const int lg = s.lg_nvec, mask = (1 << lg) - 1;
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int x = p * 256 + tid, r = x >> lg, cv = x & mask;
            pf[p] = uint4{0u, 0u, 0u, 0u};
            if (r < c.rows && cv < s.nvec) {
                const uint16_t *src = codes + (size_t)rowbase[r] + (unsigned)(s.ka + cv * vw);
                if (vw == 8) pf[p] = *reinterpret_cast<const uint4 *>(src);
                else if (vw == 4) { const uint2 t = *reinterpret_cast<const uint2 *>(src); pf[p].x = t.x; pf[p].y = t.y; }
                else pf[p].x = src[0];
            }
        }
    };
    auto place = [&](const seg_t &s) {
        const int lg = s.lg_nvec, mask = (1 << lg) - 1;
        auto put_piece = [&](int r, int cv, const uint4 &w) {
            uint16_t *dst = tile + r * s.pitch + cv * vw;
            if (vw == 8) { unsigned *d = reinterpret_cast<unsigned *>(dst); d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w; }
            else if (vw == 4) { unsigned *d = reinterpret_cast<unsigned *>(dst); d[0] = w.x; d[1] = w.y; }
            else dst[0] = (uint16_t)w.x;
        };
#pragma unroll
        for (int p = 0; p < PF; ++p) { const int x = p * 256 + tid, r = x >> lg, cv = x & mask; if (r < c.rows && cv < s.nvec) put_piece(r, cv, pf[p]); }
        // (a segment of more pieces than the registers take -- geometries outside the host's choice of segb: the rest straight from memory)
        for (int x = PF * 256 + tid; (x >> lg) < c.rows; x += 256) {
            const int r = x >> lg, cv = x & mask;
            if (cv >= s.nvec) continue;
            const uint16_t *src = codes + (size_t)rowbase[r] + (unsigned)(s.ka + cv * vw);
            uint4 w = {0u, 0u, 0u, 0u};
            if (vw == 8) w = *reinterpret_cast<const uint4 *>(src);
            else if (vw == 4) { const uint2 t = *reinterpret_cast<const uint2 *>(src); w.x = t.x; w.y = t.y; }
            else w.x = src[0];
            put_piece(r, cv, w);
        }
    };
    seg_t s = segs[0]; s.nruns = s.nblk * c.rows;
    fetch(s);
    const u64 col_bit = bit0 + col_bitoff[blockIdx.x];
    u64 bits_done = 0, zeros_done = col_zoff[blockIdx.x];
    const int null_at = (int)(nullrun - tile);
    for (int segi = 0; segi < nseg; ++segi) {
        // (the last segment's readers of the tile are behind a barrier of its last round: the one in front of the window's way out, or the one that stands in for it)
        place(s);
        __syncthreads();
        seg_t sn = s;
        if (segi + 1 < nseg) { sn = segs[segi + 1]; sn.nruns = sn.nblk * c.rows; fetch(sn); }
        const int nrounds = (s.nruns + 256 * NR - 1) / (256 * NR);
        const int edge_k = s.nE * s.E;
        for (int rd = 0; rd < nrounds; ++rd) {
            const int q0 = (rd * 256 + tid) * NR;
            // where the thread's first run lies: block `bl` of the segment, row `row` of it
            const int bl0 = (int)div_by((unsigned)q0, (unsigned)c.rows, m_rows), row0 = q0 - bl0 * c.rows;
            // pass 1: the lengths of the thread's codes (low half of `sum1`), its zero codes (high half)
            unsigned sum1 = 0;
            unsigned elen[WIDE ? NR * INNER : 1], ecode[WIDE ? NR * INNER : 1];
            bool need7[NR];                                             // (wavefront-uniform) some lane's j-th run is INNER codes wide
            if (WIDE) {
                int bl = bl0, row = row0;
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const bool on = q0 + j < s.nruns;
                    const int s2 = bl < s.nE ? s.E : s.L, koff = bl < s.nE ? bl * s.E : edge_k + (bl - s.nE) * s.L;
                    const int at = on ? row * s.pitch + s.kshift + koff : null_at;
                    const unsigned *w = reinterpret_cast<const unsigned *>(tile + (at & ~1));
                    unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];          // (a run at the end of the last row reads into the null run and the window's first words)
                    if (at & 1) { w0 = (w0 >> 16) | (w1 << 16); w1 = (w1 >> 16) | (w2 << 16); w2 = (w2 >> 16) | (w3 << 16); w3 >>= 16; }
                    const unsigned wv[4] = {w0, w1, w2, w3};
#pragma unroll
                    for (int kk = 0; kk < INNER; ++kk) {
                        unsigned cd = (kk & 1) ? wv[kk >> 1] >> 16 : wv[kk >> 1] & 0xffffu;
                        if (kk == INNER - 1) {                                         // (blocks are INNER - 1 or INNER wide here: most runs have no such code)
                            cd = (on && s2 == INNER) ? cd : nsym;
                            need7[j] = any_lane(on && s2 == INNER);                  // (every lane comes by here)
                            if (!need7[j]) { elen[j * INNER + kk] = 0u; ecode[j * INNER + kk] = 0u; continue; }
                        }
                        const u64 en = ltab[cd];
                        elen[j * INNER + kk] = (unsigned)en; ecode[j * INNER + kk] = (unsigned)(en >> 32);
                        sum1 += (unsigned)en;
                    }
                    if (++row == c.rows) { row = 0; ++bl; }
                }
            } else {
                int bl = bl0, row = row0;
                for (int j = 0; j < NR; ++j) {
                    if (q0 + j >= s.nruns) break;
                    const int s2 = bl < s.nE ? s.E : s.L, koff = bl < s.nE ? bl * s.E : edge_k + (bl - s.nE) * s.L;
                    const uint16_t *p = tile + row * s.pitch + s.kshift + koff;
                    for (int kk = 0; kk < s2; ++kk) sum1 += (unsigned)ltab[p[kk]];
                    if (++row == c.rows) { row = 0; ++bl; }
                }
            }
            const unsigned bsum = sum1 & 0xffffu, zsum = sum1 >> 16;
            u64 tot2;
            const u64 ex2 = block_excl_scan_256((u64)bsum | ((u64)zsum << 32), sh, &tot2);     // (its barriers also separate the last round's reads of the window from this round's writes)
            const unsigned tot = (unsigned)tot2, ex = (unsigned)ex2;
            const u64 gbit = col_bit + bits_done;
            const unsigned lead = (unsigned)(gbit & 31u);
            const unsigned nwords = (lead + tot + 31u) >> 5;           // words the round touches (word 0 holds `lead` bits of what came before: the last round's partial word stayed there)
            if (bsum) {
                const unsigned bitpos = lead + ex;
                unsigned wpos = bitpos >> 5, nb = bitpos & 31u;
                u64 acc = 0;
                // (every word by an atomic OR: a thread's first and last words are shared with its neighbours, and telling them from the others costs more than the OR)
                auto put = [&](unsigned le, unsigned cw) {
                    acc = (acc << (le & 63u)) | cw;
                    nb += le & 0xffffu;
                    if (nb >= 32u) { atomicOr(&win[wpos], (unsigned)(acc >> (nb - 32u))); ++wpos; nb -= 32u; }
                };
                if (WIDE) {
#pragma unroll
                    for (int e = 0; e < NR * INNER; ++e) {
                        if (e % INNER == INNER - 1 && !need7[e / INNER]) continue;      // (nobody's run is that wide)
                        put(elen[e], ecode[e]);
                    }
                } else {
                    int bl = bl0, row = row0;
                    for (int j = 0; j < NR; ++j) {
                        if (q0 + j >= s.nruns) break;
                        const int s2 = bl < s.nE ? s.E : s.L, koff = bl < s.nE ? bl * s.E : edge_k + (bl - s.nE) * s.L;
                        const uint16_t *p = tile + row * s.pitch + s.kshift + koff;
                        for (int kk = 0; kk < s2; ++kk) { const u64 en = ltab[p[kk]]; put((unsigned)en, (unsigned)(en >> 32)); }
                        if (++row == c.rows) { row = 0; ++bl; }
                    }
                }
                if (nb) atomicOr(&win[wpos], (unsigned)(acc << (32u - nb)));   // the last, partial word
            }
            if (zsum) {
                // the unpredictable values of the thread's zero codes: the originals, at their places in the list (sz_float.c:7288) -- few threads get here
                u64 zat = zeros_done + (ex2 >> 32);
                int bl = bl0, row = row0;
                for (int j = 0; j < NR; ++j) {
                    if (q0 + j >= s.nruns) break;
                    const int s2 = bl < s.nE ? s.E : s.L, koff = bl < s.nE ? bl * s.E : edge_k + (bl - s.nE) * s.L;
                    const uint16_t *p = tile + row * s.pitch + s.kshift + koff;
                    const T *drow = data + (size_t)rowbase[row] + (unsigned)(s.kbeg + koff);
                    for (int kk = 0; kk < s2; ++kk) if (p[kk] == 0) unpred[zat++] = drow[kk];
                    if (++row == c.rows) { row = 0; ++bl; }
                }
            }
            const bool last_round = rd + 1 == nrounds && segi + 1 == nseg;
            if (tot) {                                                   // (uniform)
                __syncthreads();
                // the window goes out: whole words by plain stores; the column's first word (if bits of the column before lie in front of it)
                // and its last one (if it ends inside it) are shared with the neighbouring workgroups -- cleared by k_col_scan / k_col_bounds, ORed in; a
                // round's last, partial word moves to the window's first place, where the next round goes on with it
                const bool partial = ((lead + tot) & 31u) != 0u;
                const unsigned nout = (partial && !last_round) ? nwords - 1 : nwords;
                const u64 w0 = gbit >> 5;
                unsigned keep = 0u;
                if (tid == 0 && partial && !last_round) { keep = win[nwords - 1]; if (nwords > 1) win[nwords - 1] = 0u; }      // (nobody else looks at that word: it is not written out)
                for (unsigned w = tid; w < nout; w += 256) {
                    const unsigned x = __builtin_bswap32(win[w]);
                    win[w] = w == 0 ? keep : 0u;                           // (the window is clear again behind the words that left; the kept word is the next round's first)
                    const bool shared = (w == 0 && bits_done == 0 && lead != 0u) || (w == nwords - 1 && last_round && partial);
                    if (shared) { if (x) atomicOr(&out32[w0 + w], x); }
                    else out32[w0 + w] = x;
                }
                bits_done += tot;
            } else if (rd + 1 == nrounds && segi + 1 < nseg) __syncthreads();      // (nothing to write out: the barrier that lets the next segment's rows into the tile)
            zeros_done += tot2 >> 32;
        }
        s = sn;
    }
}

// ------------------------------------------------------------------ the way back (round 6): block-ordered codes -> natural order, unpredictable values into the array
// Until round 6: k_permute<1> (0.35 ms at 512^3), three scan launches and k_unpred<1> (0.08 ms) between the Huffman decode and the inverse sweep.
// k_col_zeros: the zero codes of every block column (its codes are one contiguous range of the block-ordered array).
__global__ __launch_bounds__(256) void k_col_zeros(szh_geom3 G, const uint16_t *__restrict__ blk, unsigned *__restrict__ col_zeros)
{
    __shared__ unsigned red[4];
    const szh_se::col_t c = szh_se::make_col(G, (int)blockIdx.x);
    const int64_t base = szh_code_base01(G, c.b0, c.b1), len = (int64_t)c.rows * G.g2.count;
    const int head = (int)(base & 7);
    const int64_t ngroups = (head + len + 7) / 8;
    unsigned z = 0;
    for (int64_t g = threadIdx.x; g < ngroups; g += 256) {
        const uint4 w = *reinterpret_cast<const uint4 *>(blk + (base - head) + g * 8);       // (the array has slack behind its last code)
        const unsigned wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t at = g * 8 + e - head;
            const unsigned cd = (e & 1) ? wv[e >> 1] >> 16 : wv[e >> 1] & 0xffffu;
            z += (at >= 0 && at < len && cd == 0u) ? 1u : 0u;
        }
    }
    z = wave_sum_u32(z);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = z;
    __syncthreads();
    if (threadIdx.x == 0) col_zeros[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// k_col_unpack: a workgroup per block column, the column in the segments of `segs`: a segment's block-ordered codes into LDS as they lie (one contiguous range, 16-byte
// loads); every thread walks its share of them for zero codes, whose values -- the next ones of the stream's list, col_zoff says where the column's begin; never beyond
// `ucap` entries -- go to their places in `out` (szd_float.c:3784: the pre-scattered values the inverse sweep finds); then the segment's rows leave in natural order,
// 16 bytes at a time, gathered from LDS.
template <class T>
__global__ __launch_bounds__(256) void k_col_unpack(szh_geom3 G, const uint16_t *__restrict__ blk, uint16_t *__restrict__ nat, const szh_se::seg_t *__restrict__ segs, int nseg, int vw,
                                                    const u64 *__restrict__ col_zoff, const T *__restrict__ unpred, u64 ucap, T *__restrict__ out)
{
    using namespace szh_se;
    SZH_DYN_SMEM(smem);
    __shared__ u64 sh[8];
    __shared__ unsigned rowbase[128];
    uint16_t *L = reinterpret_cast<uint16_t *>(smem);
    const int tid = (int)threadIdx.x;
    const col_t c = make_col(G, (int)blockIdx.x);
    if (tid < c.rows) { const int i = tid / c.s1, j = tid - i * c.s1; rowbase[tid] = (unsigned)((int64_t)(c.o0 + i) * G.d0 + (int64_t)(c.o1 + j) * G.d1); }
    const int64_t col_base = szh_code_base01(G, c.b0, c.b1);
    u64 zeros_done = col_zoff[blockIdx.x];
    for (int segi = 0; segi < nseg; ++segi) {
        const seg_t s = segs[segi];
        const int klen = s.kend - s.kbeg, total = c.rows * klen;
        const int64_t start = col_base + (int64_t)c.rows * s.kbeg;
        const int head = (int)(start & 7), ngroups = (head + total + 7) / 8;
        __syncthreads();                                             // (the segment before has left the buffer; the row offsets are there)
        for (int g = tid; g < ngroups; g += 256) *reinterpret_cast<uint4 *>(L + g * 8) = *reinterpret_cast<const uint4 *>(blk + (start - head) + (int64_t)g * 8);
        __syncthreads();
        const uint16_t *B = L + head;                                // the segment's codes in block order: [block][row][kk]
        const int esz = c.rows * s.E, lsz = c.rows * s.L, eregion = s.nE * esz;
        // the unpredictable values: a thread's share = `per` consecutive codes
        {
            const int per = (total + 255) / 256, e0 = tid * per, e1 = e0 + per < total ? e0 + per : total;
            unsigned z = 0;
            for (int e = e0; e < e1; ++e) z += B[e] == 0 ? 1u : 0u;
            u64 tot;
            u64 rank = zeros_done + block_excl_scan_256((u64)z, sh, &tot);
            if (z) {
                for (int e = e0; e < e1; ++e) {
                    if (B[e] != 0) continue;
                    int bl, rem, s2, koff;
                    if (e < eregion) { bl = e / esz; rem = e - bl * esz; s2 = s.E; koff = bl * s.E; }
                    else { const int e2 = e - eregion; bl = e2 / lsz; rem = e2 - bl * lsz; s2 = s.L; koff = s.nE * s.E + bl * s.L; }
                    const int row = rem / s2, kk = rem - row * s2;
                    if (rank < ucap) out[(size_t)rowbase[row] + (unsigned)(s.kbeg + koff + kk)] = unpred[rank];
                    ++rank;
                }
            }
            zeros_done += tot;
        }
        // natural order out: vector cv of row r (a power-of-two pitch for the vector index); the vector's codes are read run after run: where the first one
        // lies is worked out once, the step from a run's end to the same row of the next block is (rows - 1) s2 (+ r (s2' - s2) where the block width changes)
        const int lg = s.lg_nvec, mask = (1 << lg) - 1;
        const unsigned m_E = magic_of((unsigned)s.E), m_L = magic_of((unsigned)s.L);
        const int edge_k = s.nE * s.E;
        for (int x = tid; (x >> lg) < c.rows; x += 256) {
            const int r = x >> lg, cv = x & mask;
            if (cv >= s.nvec) continue;
            const int k0 = s.ka + cv * vw;                             // the vector's first k; its codes that lie in [kbeg, kend) are this segment's
            const int kl = k0 - s.kbeg;                                // (may be negative in a row's first vector)
            const int e_lo = kl < 0 ? -kl : 0, e_hi = klen - kl < vw ? klen - kl : vw;
            uint16_t v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0;
            if (e_lo < e_hi) {
                const int q = kl + e_lo;
                int bl, kk, s2;
                if (q < edge_k) { bl = (int)div_by((unsigned)q, (unsigned)s.E, m_E); kk = q - bl * s.E; s2 = s.E; }
                else { const int q2 = q - edge_k; const int b2 = (int)div_by((unsigned)q2, (unsigned)s.L, m_L); kk = q2 - b2 * s.L; bl = s.nE + b2; s2 = s.L; }
                int idx = (bl < s.nE ? bl * esz : eregion + (bl - s.nE) * lsz) + r * s2 + kk, left = s2 - kk;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (e >= e_lo && e < e_hi) {
                        v[e] = B[idx];
                        ++idx;
                        if (--left == 0) {
                            ++bl;
                            const int s2n = bl < s.nE ? s.E : s.L;
                            idx += (c.rows - 1) * s2 + r * (s2n - s2);
                            s2 = s2n; left = s2;
                        }
                    }
                }
            }
            uint16_t *dst = nat + (size_t)rowbase[r] + (unsigned)k0;
            if (kl >= 0 && kl + vw <= klen) {
                if (vw == 8) { uint4 w; __builtin_memcpy(&w, v, 16); *reinterpret_cast<uint4 *>(dst) = w; }
                else if (vw == 4) { uint2 w; __builtin_memcpy(&w, v, 8); *reinterpret_cast<uint2 *>(dst) = w; }
                else dst[0] = v[0];
            } else {
                for (int e = 0; e < vw; ++e) if (kl + e >= 0 && kl + e < klen) dst[e] = v[e];      // (a vector that straddles two segments: each writes its own codes)
            }
        }
    }
}
