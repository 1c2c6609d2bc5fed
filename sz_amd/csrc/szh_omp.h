// szh_omp.h -- the reference's OpenMP container for 3-D arrays on the GPU: the array is cut into thread_num boxes, every box is
// quantised on its own (no value crosses a box face), ONE Huffman code book covers all boxes, every box has its own byte-aligned payload.
//   container, box grid   sz/src/sz_omp.c:63-358    SZ_compress_float_3D_MDQ_openmp          (double: :578-863)
//   box quantiser         sz/src/sz_float.c:4704-5012  SZ_compress_float_3D_MDQ_RA_block      (double: sz_double.c, same name)
//   inverse               sz/src/sz_omp.c:366-566   decompressDataSeries_float_3D_openmp,  szd_float.c decompressDataSeries_float_3D_RA_block
// Why it is on the hot path: thousands of independent boxes instead of ONE dependency front (k_ribbon / k_pencil: ~40 of 256 tiles are
// on the front at a time), and a stock OpenMP build of SZ reads the stream.
//
// The box quantiser predicts from RECONSTRUCTED neighbours (left, above, previous plane), so inside a box the points of one hyperplane
// k + i + j = t are independent and the hyperplanes follow each other.  A workgroup takes a box; thread (k, i) owns the ROW (k, i, :)
// and walks it along j, one point per step, at step t = k + i + j:
//   * its own two previous reconstructions stay in registers (the left neighbours);
//   * the three neighbours of other rows -- (k, i-1, j), (k-1, i, j), (k-1, i-1, j) -- were written at steps t-1, t-1, t-2 into a ring of
//     four values per row in LDS; their predecessors along j are what the thread read one step earlier (registers again).  A value j is
//     read at steps t+1 and t+2 and its ring place is overwritten at t+4: ONE barrier per step is enough;
//   * the row is read in chunks of four values (one 16-byte load for float, two for double) and its codes leave as 8-byte stores (four
//     codes): every lane of a wavefront works on another row, so narrower accesses would spend a full line transfer per 4 (2) bytes.
//     All lanes load and store at the same steps, two groups of four steps ahead of the use (see the kernel).
// Box faces of up to 1024 rows (c0 * c1 <= 1024: a 32^3 box is 1024 rows of 32), the box grid must divide the array (on an uneven grid
// the reference counts Huffman frequencies over uninitialised gaps of its code array: nothing to be identical to).
// Round 4: run on hardware (tests/test_zz_omp_hip.py -m gpu, tools/omp_diff_fuzz.py on the GPU); boxes of 32 x 32 columns go through the
// column sweep of szh_ompcol.h instead of k_omp_box, and the entropy stage through the per-box kernels at the end of this file.
#pragma once

struct szh_omp_geom {
    int nx, ny, nz;            // boxes along dim 0 / 1 / 2
    int c0, c1, c2;            // points of a box along dim 0 / 1 / 2
    int64_t d0, d1;            // pitches of the array
    int nb, bel;               // boxes, points per box
    int cpb;                   // chunks of SZH_ENC_CHUNK codes per box (the last one may be short)
    int vec;                   // rows may be read / written 16 bytes at a time, codes 8 bytes at a time (host: picks k_omp_box<.., VEC>)
    int tile8, pitch;          // lanes of a wavefront = an 8 x 8 tile of rows; line pitch of the LDS ring (k_omp_box)
};
#define SZH_OMP_MAX_ROWS 1024

template <class T> struct szh_omp_chunk { struct alignas(16) type { T x, y, z, w; }; };     // four values of a row: one 16-byte load (float), two (double)
// (named members, not an array: picking v[j & 3] out of an array member sent the chunks to scratch memory)

__device__ __forceinline__ const void *szh_omp_box_origin_bytes(const szh_omp_geom &g, int b, size_t elem, const void *data)
{
    const int bi = b / (g.ny * g.nz), bj = (b / g.nz) % g.ny, bk = b % g.nz;
    return (const char *)data + ((int64_t)bi * g.c0 * g.d0 + (int64_t)bj * g.c1 * g.d1 + (int64_t)bk * g.c2) * (int64_t)elem;
}

// DEC = false: data -> codes (0 = the value is kept verbatim), the count of such values per box, the box's first value
// DEC = true : codes + the box's verbatim values (in the box's row-major order, uoff[b] .. uoff[b + 1]) + its first value -> data;
//              `ucount` is then ONE counter of boxes whose codes and table entry disagree
template <class T, bool DEC, bool VEC>
__global__ __launch_bounds__(1024) void k_omp_box(szh_omp_geom g, const T *__restrict__ data, T *__restrict__ out, T eb, T recip, int intervals,
                                                  uint16_t *__restrict__ codes, unsigned *__restrict__ ucount, u64 *__restrict__ ucount64, T *__restrict__ first,
                                                  const T *__restrict__ unpred, const u64 *__restrict__ uoff)
{
    SZH_DYN_SMEM(smem);
    T *ring = reinterpret_cast<T *>(smem);                  // [4][c0 * pitch]
    __shared__ unsigned s_un;
    __shared__ unsigned s_scan[SZH_OMP_MAX_ROWS];
    typedef typename szh_omp_chunk<T>::type chunkT;
    const int b = blockIdx.x, tid = threadIdx.x, rows = g.c0 * g.c1;
    if ((int)blockDim.x != rows) return;                    // exactly one lane per row (the unconditional stores below have nowhere else to go)
    // lane -> row.  Rows (k, i) start at step k + i, so a wavefront works from the first start of its rows to the last end: with 64
    // consecutive rows (two lines of 32) that is 32 + 33 of the 94 steps of a 32^3 box, with an 8 x 8 tile of rows 32 + 15 (g.tile8: both
    // face dimensions multiples of 8).  r: the row's number in the box's row-major order (codes, ranks); rp: its place in the ring, whose
    // lines are g.pitch apart (a pitch of 8 or 24 modulo 32 spreads the 8 x 8 tile over the LDS banks two lanes deep, the minimum)
    int k, i;
    if (g.tile8) { const int w = tid >> 6, l = tid & 63, tpr = g.c1 >> 3; k = (w / tpr) * 8 + (l >> 3); i = (w % tpr) * 8 + (l & 7); }
    else { k = tid / g.c1; i = tid - k * g.c1; }
    const int r = k * g.c1 + i, rp = k * g.pitch + i, rstride = g.c0 * g.pitch;
    const int64_t row_off = (int64_t)k * g.d0 + (int64_t)i * g.d1;
    const T *row_in = DEC ? nullptr : reinterpret_cast<const T *>(szh_omp_box_origin_bytes(g, b, sizeof(T), data)) + row_off;
    T *row_out = DEC ? reinterpret_cast<T *>(const_cast<void *>(szh_omp_box_origin_bytes(g, b, sizeof(T), out))) + row_off : nullptr;
    uint16_t *crow = codes + (int64_t)b * g.bel + (int64_t)r * g.c2;
    const int radius = intervals / 2;
    if (tid == 0) s_un = 0u;
    T first_v;
    unsigned urank = 0, ucap = 0;
    if (DEC) {
        // rank of the row's first verbatim value among the box's: zeros of the rows before it (row-major order = row order)
        unsigned z = 0;
        if (VEC) {                                              // (four codes per load, eight loads in flight: one at a time is a memory round trip per code)
#pragma unroll 8
            for (int m = 0; m < (g.c2 >> 2); ++m) {
                const u64 w = *reinterpret_cast<const u64 *>(crow + 4 * m);
                z += ((w & 0xffffull) == 0) + ((w & 0xffff0000ull) == 0) + ((w & 0xffff00000000ull) == 0) + ((w >> 48) == 0);
            }
        } else for (int j = 0; j < g.c2; ++j) z += crow[j] == 0;
        s_scan[r] = z;
        __syncthreads();
        for (int o = 1; o < (int)blockDim.x; o <<= 1) {
            const unsigned add = r >= o ? s_scan[r - o] : 0u;
            __syncthreads();
            s_scan[r] += add;
            __syncthreads();
        }
        urank = s_scan[r] - z;
        first_v = first[b];
        // a damaged stream: the box's codes call for another number of verbatim values than its table entry says (`ucount`: the error
        // counter here); the reads below stay inside the box's values either way
        ucap = (unsigned)(uoff[b + 1] - uoff[b]);
        if (r == rows - 1 && s_scan[r] != ucap) atomicAdd(ucount, 1u);
    } else {
        first_v = *reinterpret_cast<const T *>(szh_omp_box_origin_bytes(g, b, sizeof(T), data));
        __syncthreads();
    }
    const T *ub = DEC ? unpred + uoff[b] : nullptr;
    // The step.  The reference's predictors by position (sz_float.c:4749-4990), all from reconstructed values: plane 0 -- (0,0,0) the box's first
    // value, (0,0,1) the left neighbour, (0,0,j>=2) 2 left - left-left, (0,i>=1,0) the value above, elsewhere left + above - above-left; plane
    // k>=1 -- (k,0,0) the previous plane's value, first row / first column the 2-D form with the previous plane, elsewhere the 7-point form.
    // ONE predictor expression serves every row but (0, 0): the 7-point form with the absent neighbours as +0 and the
    // registers of the j - 1 neighbours starting at +0 --
    //   row (0, i>=1):  l1 + A + 0 - Ap - 0 - 0 + 0   = left + above - above-left        (j = 0: A)
    //   row (k>=1, 0):  l1 + 0 + B - 0 - 0 - Bp + 0   = left + back - back-left          (j = 0: B)
    //   elsewhere:      the reference's own sum, left to right                            (j = 0: 0 + A + B - 0 - C - 0 + 0)
    // Adding a +0 is exact; it can only turn a -0 SUM into +0, and the sign of a zero prediction reaches neither the code (|cur - pred|,
    // `diff < 0` is false for both zeros) nor the reconstruction (pred + 2 q eb: q = 0 adds +0 and gives +0 either way).  So the lanes
    // of a wavefront (rows of every kind) run the same instructions; the lone (0, 0) row of a box keeps its own branch.
    //
    // Memory (VEC): a lane's row is four-value chunks.  The lanes of a wavefront stand at different j, so "load when j % 4 == 0" would
    // put a partly masked load -- and the wait for it -- into EVERY step.  Instead all lanes load at the same steps (t % 4 == 0): during
    // the four steps that follow a lane needs the chunks m and m + 1 of its j (v0, v1); chunk m + 2 is requested now (vl) and taken
    // over four steps later.  A finished chunk of results (four codes / four values) waits for the same steps to be stored.
    const bool i0 = i == 0, k0 = k == 0, row00 = i0 && k0;
    const int ia = i0 ? rp : rp - 1, ib = k0 ? rp : rp - g.pitch, ic = (i0 || k0) ? rp : rp - g.pitch - 1;     // (own place when absent: read, not used)
    T l1 = 0, l2 = 0, Ap = 0, Bp = 0, Cp = 0;
    unsigned nun = 0;
    const int steps = g.c0 + g.c1 + g.c2 - 2, j_first = -(k + i), nch = g.c2 >> 2;
    const T fint = (T)intervals;
    chunkT va, vb, vc, vd, vacc, vdone;
    { const chunkT z = {0, 0, 0, 0}; va = z; vb = z; vc = z; vd = z; vacc = z; vdone = z; }
    u64 ca = 0, cb = 0, cc = 0, cd = 0, cacc = 0, cdone = 0;
    int done_j = 0;
    // (every lane loads, from a chunk index clamped into its row -- rows that do not exist stand on row 0: a load that may or may not
    //  happen would make the compiler wait for it right where it is issued)
    auto clampc = [&](int m) { return m < 0 ? 0 : m >= nch ? nch - 1 : m; };
    const uint16_t *cr = crow;
    if (VEC) {                                              // the first four steps' chunks, and the one that is in flight during them
        const int mb = j_first >> 2;
        if (!DEC) {
            va = *reinterpret_cast<const chunkT *>(row_in + 4 * clampc(mb));
            vb = *reinterpret_cast<const chunkT *>(row_in + 4 * clampc(mb + 1));
            vc = *reinterpret_cast<const chunkT *>(row_in + 4 * clampc(mb + 2));
        } else {
            ca = *reinterpret_cast<const u64 *>(cr + 4 * clampc(mb));
            cb = *reinterpret_cast<const u64 *>(cr + 4 * clampc(mb + 1));
            cc = *reinterpret_cast<const u64 *>(cr + 4 * clampc(mb + 2));
        }
    }
    // A GROUP of four steps (unrolled) with the chunks m (x0) and m + 1 (x1) of the lane's j; chunk m + 2 was requested at the end of
    // the group before and is in flight, chunk m + 3 is requested at the end of this one into `xn`.  The four chunk variables take
    // these roles in turn (the loop below is unrolled by four groups), so no chunk is ever copied: with "v0 = v1; v1 = vl" at a
    // group's end the compiler placed those copies behind the new load and waited for it on the spot -- a memory round trip in every
    // group, which is what this arrangement exists to avoid.  Steps past the last one find no lane inside its row.
    auto group = [&](const int tg, const chunkT &x0, const chunkT &x1, chunkT &xn, const u64 y0, const u64 y1, u64 &yn) __attribute__((always_inline)) {
        const int mb = (tg + j_first) >> 2;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = tg + u + j_first;
            if ((unsigned)j < (unsigned)g.c2) {
                const int slot = (j & 3) * rstride;
                T A = ring[slot + ia], B = ring[slot + ib], C = ring[slot + ic];
                A = i0 ? (T)0 : A; B = k0 ? (T)0 : B; C = (i0 || k0) ? (T)0 : C;
                T pred = l1 + A + B - Ap - C - Bp + Cp;
                if (row00) pred = j == 0 ? first_v : j == 1 ? l1 : 2 * l1 - l2;
                const bool hi = (j >> 2) != mb;
                const int e4 = j & 3;
                T rec;
                if (!DEC) {
                    T cur;
                    if (VEC) {
                        const T q0 = hi ? x1.x : x0.x, q1 = hi ? x1.y : x0.y, q2v = hi ? x1.z : x0.z, q3 = hi ? x1.w : x0.w;
                        cur = e4 == 0 ? q0 : e4 == 1 ? q1 : e4 == 2 ? q2v : q3;
                    } else cur = row_in[j];
                    // sz_float.c:4762-4783: |diff| / eb + 1 against the interval count, truncation, the bound verified on the result
                    const T diff = cur - pred;
                    const T mag = (diff < 0 ? -diff : diff) * recip + 1;
                    const T itv = diff < 0 ? -mag : mag;
                    const bool in_range = mag < fint;                         // (false for a NaN)
                    const int q2 = 2 * (int)((in_range ? itv : (T)0) / 2);    // 2 (code - radius)
                    const T r = pred + (T)q2 * eb;
                    const T err = cur - r;
                    const bool ok = in_range && !((err < 0 ? -err : err) > eb);
                    const int tc = ok ? (q2 >> 1) + radius : 0;
                    rec = ok ? r : cur;
                    nun += ok ? 0u : 1u;
                    if (VEC) {
                        cacc |= (u64)(unsigned)tc << (16 * e4);
                        if (e4 == 3) { cdone = cacc; done_j = j - 3; cacc = 0; }
                    } else crow[j] = (uint16_t)tc;
                } else {
                    unsigned tc;
                    if (VEC) tc = (unsigned)((hi ? y1 : y0) >> (16 * e4)) & 0xffffu;
                    else tc = crow[j];
                    if (tc) rec = pred + (T)(2 * ((int)tc - radius)) * eb;
                    else { const unsigned ui = urank + nun++; rec = ui < ucap ? ub[ui] : (T)0; }
                    if (VEC) {
                        vacc.x = e4 == 0 ? rec : vacc.x; vacc.y = e4 == 1 ? rec : vacc.y; vacc.z = e4 == 2 ? rec : vacc.z; vacc.w = e4 == 3 ? rec : vacc.w;
                        if (e4 == 3) { vdone = vacc; done_j = j - 3; }
                    } else row_out[j] = rec;
                }
                ring[slot + rp] = rec;
                l2 = l1; l1 = rec; Ap = A; Bp = B; Cp = C;
            }
            __syncthreads();
        }
        if (VEC) {
            if (!DEC) xn = *reinterpret_cast<const chunkT *>(row_in + 4 * clampc(mb + 3));
            else yn = *reinterpret_cast<const u64 *>(cr + 4 * clampc(mb + 3));
            // the lane's latest finished chunk of results (one per four steps; a lane that has finished none since the last time writes
            // the same chunk again, one that has not started yet writes zeros where its first chunk will go: a store that may or may
            // not happen would leave the compiler unable to count the accesses in flight, and it would wait for all of them)
            if (!DEC) *reinterpret_cast<u64 *>(crow + done_j) = cdone; else *reinterpret_cast<chunkT *>(row_out + done_j) = vdone;
        }
    };
    // (the first group stands in front of the loop: entered from the prologue's three loads the loop's first group would have to
    //  assume the fewest accesses in flight, and so wait for the newest one on every later pass too)
    group(0, va, vb, vd, ca, cb, cd);
    for (int tg = 4; tg < steps; tg += 16) {
        group(tg, vb, vc, va, cb, cc, ca);
        group(tg + 4, vc, vd, vb, cc, cd, cb);
        group(tg + 8, vd, va, vc, cd, ca, cc);
        group(tg + 12, va, vb, vd, ca, cb, cd);
    }
    if (!DEC) {
        if (nun) atomicAdd(&s_un, nun);
        __syncthreads();
        if (tid == 0) { ucount[b] = s_un; ucount64[b] = s_un; first[b] = first_v; }
    }
}

// the values the quantiser kept verbatim, in the box's row-major order, behind those of the boxes before it (sz_omp.c:246-262)
template <class T>
__global__ __launch_bounds__(256) void k_omp_gather(szh_omp_geom g, const T *__restrict__ data, const uint16_t *__restrict__ codes,
                                                    const unsigned *__restrict__ ucount, const u64 *__restrict__ uoff, T *__restrict__ unpred)
{
    __shared__ u64 sh[8];
    const int b = blockIdx.x;
    if (ucount[b] == 0) return;                               // uniform
    const T *box = reinterpret_cast<const T *>(szh_omp_box_origin_bytes(g, b, sizeof(T), data));
    const uint16_t *cb = codes + (int64_t)b * g.bel;
    T *dst = unpred + uoff[b];
    u64 done = 0;
    for (int base = 0; base < g.bel; base += 256 * 8) {
        const int p0 = base + (int)threadIdx.x * 8;
        unsigned mask = 0;
        for (int e = 0; e < 8; ++e) if (p0 + e < g.bel && cb[p0 + e] == 0) mask |= 1u << e;
        u64 tot;
        u64 rank = done + block_excl_scan_256((u64)__builtin_popcount(mask), sh, &tot);
        for (int e = 0; e < 8; ++e) if (mask >> e & 1u) {
            const int p = p0 + e, k = p / (g.c1 * g.c2), r = p - k * (g.c1 * g.c2), i = r / g.c2, j = r - i * g.c2;
            dst[rank++] = box[(int64_t)k * g.d0 + (int64_t)i * g.d1 + j];
        }
        done += tot;
    }
}

// ---- Huffman packing with one byte-aligned payload per box (sz_omp.c:300-330: `encode` per box into its own buffer, then memcpy)
__device__ __forceinline__ void omp_load8(const uint16_t *__restrict__ cb, int p0, int bel, bool aligned, uint16_t (&c)[8])
{
    if (aligned && p0 + 8 <= bel) { const uint4 w = *reinterpret_cast<const uint4 *>(cb + p0); __builtin_memcpy(c, &w, 16); }
    else { for (int q = 0; q < 8; ++q) c[q] = p0 + q < bel ? cb[p0 + q] : (uint16_t)0; }
}
// bits of chunk q of box b -> chunk_bits[b * cpb + q]
__global__ __launch_bounds__(256) void k_omp_chunk_bits(szh_omp_geom g, const uint16_t *__restrict__ codes, const uint8_t *__restrict__ len, u64 *chunk_bits)
{
    __shared__ u64 sh[4];
    const int c = blockIdx.x, b = c / g.cpb, q = c - b * g.cpb;
    const uint16_t *cb = codes + (int64_t)b * g.bel;
    const int p0 = q * SZH_ENC_CHUNK + (int)threadIdx.x * 8;
    uint16_t cc[8];
    omp_load8(cb, p0, g.bel, (g.bel & 7) == 0, cc);
    unsigned s = 0;
    for (int e = 0; e < 8; ++e) if (p0 + e < g.bel) s += len[cc[e]];
    const u64 ws = wave_sum_u64((u64)s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ws;
    __syncthreads();
    if (threadIdx.x == 0) chunk_bits[c] = sh[0] + sh[1] + sh[2] + sh[3];
}
// bytes of box b = its bits rounded up (Huffman.c encode: the last byte is padded with zero bits)
__global__ __launch_bounds__(256) void k_omp_box_bytes(int nb, int cpb, const u64 *__restrict__ chunk_off, const u64 *__restrict__ total_bits, u64 *box_bytes)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nb) return;
    const u64 lo = chunk_off[(int64_t)b * cpb], hi = b + 1 < nb ? chunk_off[(int64_t)(b + 1) * cpb] : *total_bits;
    box_bytes[b] = (hi - lo + 7) >> 3;
}
// out32: 4-byte aligned base of the stream buffer (zeroed); bit0: bit position of the first box's payload in it
__global__ __launch_bounds__(256) void k_omp_encode(szh_omp_geom g, const uint16_t *__restrict__ codes, const u64 *__restrict__ code,
                                                    const uint8_t *__restrict__ len, const u64 *__restrict__ chunk_off,
                                                    const u64 *__restrict__ box_off, u64 bit0, unsigned *out32)
{
    __shared__ unsigned buf[SZH_ENC_CHUNK * 2 + 2];
    __shared__ u64 sh[8];
    const int c = blockIdx.x, b = c / g.cpb, q = c - b * g.cpb;
    const uint16_t *cb = codes + (int64_t)b * g.bel;
    const int p0 = q * SZH_ENC_CHUNK + (int)threadIdx.x * 8;
    uint16_t cc[8];
    omp_load8(cb, p0, g.bel, (g.bel & 7) == 0, cc);
    const u64 gbit = bit0 + box_off[b] * 8 + (chunk_off[c] - chunk_off[(int64_t)b * g.cpb]);
    const unsigned lead = (unsigned)(gbit & 31);
    unsigned l[8]; unsigned s = 0;
    for (int e = 0; e < 8; ++e) { l[e] = p0 + e < g.bel ? (unsigned)len[cc[e]] : 0u; s += l[e]; }
    u64 tot;
    const u64 ex = block_excl_scan_256((u64)s, sh, &tot);
    for (unsigned w = threadIdx.x; w < (unsigned)((lead + tot + 31) >> 5) + 1; w += 256) buf[w] = 0;
    __syncthreads();
    unsigned pos = lead + (unsigned)ex;
    u64 acc = 0; int accn = 0;
    for (int e = 0; e < 8; ++e) {
        if (!l[e]) continue;
        const u64 cw = code[cc[e]];
        if (accn + (int)l[e] > 64) { lds_put_bits(buf, pos, acc, accn); pos += accn; acc = 0; accn = 0; }
        acc = l[e] == 64 ? cw : ((acc << l[e]) | (cw & ((1ull << l[e]) - 1)));
        accn += (int)l[e];
    }
    if (accn) lds_put_bits(buf, pos, acc, accn);
    __syncthreads();
    const unsigned nwords = (unsigned)((lead + tot + 31) >> 5);
    const u64 w0 = gbit >> 5;
    for (unsigned w = threadIdx.x; w < nwords; w += 256) {
        const unsigned v = __builtin_bswap32(buf[w]);
        if (w == 0 || w == nwords - 1) { if (v) atomicOr(&out32[w0 + w], v); }
        else out32[w0 + w] = v;
    }
}

// ---- Huffman decoding, a workgroup per box.  A box's payload starts at a byte boundary and holds `bel` symbols, so the boxes are
// independent; inside a box the 256 lanes take equal stretches of its bits and find their first codeword boundary by the
// self-synchronising rule of k_hdec_pass (a stretch starts where the one before it ends; repeated until no start moves -- the fixed
// point is the sequential decode, reached after a couple of rounds because Huffman codes fall into step within a few symbols).
// Bit by bit through the node table (`table[2 node + bit]`: next node, or 0x80000000 | symbol): the simple form first.
__device__ __forceinline__ unsigned omp_hdec_run(const unsigned char *__restrict__ bits, unsigned total, const unsigned *__restrict__ table,
                                                 unsigned pos, unsigned limit, unsigned *endpos, uint16_t *out, unsigned o, unsigned oend)
{
    unsigned cnt = 0, p = pos, last = pos, node = 0, cur = 0;
    if (p < total && (p & 7u)) cur = bits[p >> 3];
    while (p < total) {
        if ((p & 7u) == 0) cur = bits[p >> 3];
        const unsigned bit = (cur >> (7u - (p & 7u))) & 1u;
        ++p;
        const unsigned nx = table[2 * node + bit];
        if (nx & 0x80000000u) {
            if (out && o + cnt < oend) out[o + cnt] = (uint16_t)(nx & 0xffffu);
            ++cnt; node = 0; last = p;
            if (p >= limit) break;
        } else node = nx;
    }
    *endpos = last;
    return cnt;
}
// dynamic LDS: [the node table when tab_lds (2 n_nodes words)] [the box's payload when it has at most pay_cap bytes, + 8] -- the walk is one
// dependent table read and (every eighth step) one payload read per bit; both come from LDS then
__global__ __launch_bounds__(256) void k_omp_hdec(int bel, const unsigned char *__restrict__ payload, const u64 *__restrict__ box_off,
                                                  const u64 *__restrict__ box_bytes, const unsigned *__restrict__ table, int n_nodes, int tab_lds,
                                                  unsigned pay_cap, int single_symbol, uint16_t *__restrict__ codes, unsigned *__restrict__ bad)
{
    SZH_DYN_SMEM(smem);
    __shared__ unsigned s_start[257], s_flag[2];
    __shared__ u64 sh[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    uint16_t *out = codes + (int64_t)b * bel;
    if (single_symbol >= 0) { for (int p = tid; p < bel; p += 256) out[p] = (uint16_t)single_symbol; return; }
    const unsigned char *bits = payload + box_off[b];
    const unsigned nbytes = (unsigned)box_bytes[b];
    const unsigned total = nbytes * 8;
    {
        unsigned *ltab = reinterpret_cast<unsigned *>(smem);
        if (tab_lds) { for (int i = tid; i < 2 * n_nodes; i += 256) ltab[i] = table[i]; table = ltab; }
        if (nbytes <= pay_cap && nbytes > 0) {                    // whole words from the word the payload starts in (the stream lies in a padded buffer)
            unsigned *lpay = ltab + (tab_lds ? 2 * n_nodes : 0);
            const unsigned lead = (unsigned)((uintptr_t)bits & 3u);
            const unsigned *src = reinterpret_cast<const unsigned *>(bits - lead);
            for (unsigned i = tid; i < (lead + nbytes + 3) / 4; i += 256) lpay[i] = src[i];
            bits = reinterpret_cast<const unsigned char *>(lpay) + lead;
        }
        __syncthreads();
    }
    unsigned sb = (total + 255u) / 256u; if (sb < 64u) sb = 64u;
    const unsigned first = (unsigned)tid * sb, limit = first + sb;
    unsigned start = first, endp = first, cnt = 0;
    bool redo = true;
    if (tid == 0) { s_flag[0] = 0u; s_flag[1] = 0u; }
    __syncthreads();
    for (int round = 0; round < 257; ++round) {
        if (redo) {
            if (start < limit && start < total) cnt = omp_hdec_run(bits, total, table, start, limit, &endp, nullptr, 0, 0);
            else { cnt = 0; endp = start; }
            s_start[tid + 1] = endp;
        }
        __syncthreads();
        if (tid == 0) s_flag[(round + 1) & 1] = 0u;            // (the flag of the NEXT round; this round's is read below, after the barrier)
        redo = false;
        if (tid > 0) { const unsigned ns = s_start[tid]; if (ns != start) { start = ns; redo = true; } }
        if (redo) s_flag[round & 1] = 1u;
        __syncthreads();
        if (!s_flag[round & 1]) break;                         // uniform
    }
    u64 tot;
    const unsigned o = (unsigned)block_excl_scan_256((u64)cnt, sh, &tot);
    if (tot < (u64)bel) { if (tid == 0) atomicAdd(bad, 1u); }   // a payload that holds fewer symbols than the box has points
    if (cnt && o < (unsigned)bel) { unsigned e; omp_hdec_run(bits, total, table, start, limit, &e, out, o, (unsigned)bel); }
}

// =====================================================================================================================
// Round 4: the entropy stage with ONE pass over the codes on either side of the code book (measured before, 512^3 f32, 4096 boxes:
// k_hist_u16 0.07-0.2 + k_omp_gather 0.22 + k_omp_chunk_bits 0.10 + two scans + k_omp_encode 0.22 ms; k_omp_hdec 1.96 ms).
//   k_omp_hist_box   a workgroup per box: the box's histogram (LDS, lane-private copies) -> hist_box[b][*] and the global histogram
//   k_omp_box_bits   bytes of a box's payload = sum of hist_box[b][s] * len[s], rounded up      (or, large alphabets: from its codes)
//   k_omp_encode_box a workgroup per box walks the box's chunks with a running bit position (no chunk table, no chunk scan) and
//                    drops the verbatim values (code 0) into the stream's table on the way (the box's rank table: uoff)
//   k_omp_hdec_lut   a workgroup per box, the payload staged in LDS, decoded with the multi-symbol look-up table of k_hdec_* (hdec_run_lut)
// =====================================================================================================================
// ucount / ucount64 (or null): the box's count of verbatim values = its bin 0 (when the sweep did not count them, k_omp_col<.., COUNT = false>)
__global__ __launch_bounds__(256) void k_omp_hist_box(int bel, const uint16_t *__restrict__ codes, unsigned nbins, int rshift, unsigned *__restrict__ hist_box, unsigned *hist,
                                                      unsigned *ucount, u64 *ucount64, int nb, int per_wg)
{
    SZH_DYN_SMEM(smem);
    unsigned *sh = reinterpret_cast<unsigned *>(smem);
    const unsigned R = 1u << rshift;
    const unsigned rep = threadIdx.x & (R - 1);
    // `per_wg` boxes one after the other (small boxes: 32 768 workgroups of two loads per thread were bound by their own dispatch, 0.38 ms)
    for (int box = (int)blockIdx.x * per_wg; box < nb && box < ((int)blockIdx.x + 1) * per_wg; ++box) {
        for (unsigned i = threadIdx.x; i < nbins * R; i += 256) sh[i] = 0;
        __syncthreads();
        const uint16_t *cb = codes + (int64_t)box * bel;
        const int nvec = bel / 8;
        const uint4 *v4 = reinterpret_cast<const uint4 *>(cb);
        for (int i = (int)threadIdx.x; i < nvec; i += 4 * 256) {         // four loads in flight per thread
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i + u * 256 < nvec) v[u] = v4[i + u * 256];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i + u * 256 >= nvec) break;
                const unsigned wv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned c0 = wv[q] & 0xffffu, c1 = wv[q] >> 16;
                    if (c0 < nbins) atomicAdd(&sh[(c0 << rshift) + rep], 1u);
                    if (c1 < nbins) atomicAdd(&sh[(c1 << rshift) + rep], 1u);
                }
            }
        }
        for (int i = nvec * 8 + (int)threadIdx.x; i < bel; i += 256) { const unsigned c = cb[i]; if (c < nbins) atomicAdd(&sh[(c << rshift) + rep], 1u); }
        __syncthreads();
        for (unsigned b = threadIdx.x; b < nbins; b += 256) {
            unsigned s = 0;
            for (unsigned r = 0; r < R; ++r) s += sh[(b << rshift) + r];
            hist_box[(int64_t)box * nbins + b] = s;
            if (s) atomicAdd(&hist[b], s);
            if (b == 0 && ucount) { ucount[box] = s; ucount64[box] = s; }
        }
        __syncthreads();
    }
}
// bytes of box b = its bits rounded up (Huffman.c encode: the last byte is padded with zero bits); from the box's histogram ...
__global__ __launch_bounds__(256) void k_omp_box_bits_h(int nb, unsigned nbins, const unsigned *__restrict__ hist_box, const uint8_t *__restrict__ len, u64 *box_bytes)
{
    const int b = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;        // a wavefront per box
    if (b >= nb) return;
    u64 s = 0;
    for (unsigned i = lane; i < nbins; i += 64) s += (u64)hist_box[(int64_t)b * nbins + i] * len[i];
    s = wave_sum_u64(s);
    if (lane == 0) box_bytes[b] = (s + 7) >> 3;
}
// the same from the packed table `code << 8 | len` of the fast path
__global__ __launch_bounds__(256) void k_omp_box_bits_p(int nb, unsigned nbins, const unsigned *__restrict__ hist_box, const u64 *__restrict__ packed, u64 *box_bytes)
{
    const int b = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;        // a wavefront per box
    if (b >= nb) return;
    u64 s = 0;
    for (unsigned i = lane; i < nbins; i += 64) s += (u64)hist_box[(int64_t)b * nbins + i] * (unsigned)(packed[i] & 0xffu);
    s = wave_sum_u64(s);
    if (lane == 0) box_bytes[b] = (s + 7) >> 3;
}
// ... or from its codes (alphabets too large for per-box histograms)
__global__ __launch_bounds__(256) void k_omp_box_bits_c(int bel, const uint16_t *__restrict__ codes, const uint8_t *__restrict__ len, u64 *box_bytes)
{
    __shared__ u64 sh[4];
    const uint16_t *cb = codes + (int64_t)blockIdx.x * bel;
    u64 s = 0;
    for (int p0 = (int)threadIdx.x * 8; p0 < bel; p0 += 256 * 8) {
        uint16_t cc[8];
        omp_load8(cb, p0, bel, (bel & 7) == 0, cc);
        for (int e = 0; e < 8; ++e) if (p0 + e < bel) s += len[cc[e]];
    }
    const u64 ws = wave_sum_u64(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ws;
    __syncthreads();
    if (threadIdx.x == 0) box_bytes[blockIdx.x] = (sh[0] + sh[1] + sh[2] + sh[3] + 7) >> 3;
}
// out32: 4-byte aligned base of the stream buffer (zeroed); bit0: bit position of the first box's payload in it.  unpred: the verbatim values
// of all boxes, box after box (uoff), in the box's row-major order (sz_omp.c:246-262).  TAB_LDS: code words and lengths in (dynamic) LDS --
// [u64 code[nbins]][u8 len[nbins]] -- instead of two dependent global look-ups per code.  The next chunk's codes are requested before the
// present one is packed (a workgroup walks its box chunk by chunk: every load it waits for is a memory round trip on its critical path).
template <class T, bool TAB_LDS>
__global__ __launch_bounds__(256) void k_omp_encode_box(szh_omp_geom g, const T *__restrict__ data, const uint16_t *__restrict__ codes, const u64 *__restrict__ code,
                                                        const uint8_t *__restrict__ len, unsigned nbins, const u64 *__restrict__ box_off, const u64 *__restrict__ box_bytes,
                                                        const u64 *__restrict__ uoff, const unsigned *__restrict__ ucount, u64 bit0, unsigned *out32,
                                                        T *__restrict__ unpred, unsigned *bad)
{
    SZH_DYN_SMEM(smem);
    __shared__ unsigned buf[SZH_ENC_CHUNK * 2 + 2];
    __shared__ u64 sh[8];
    const int b = blockIdx.x;
    const u64 *lcode = code; const uint8_t *llen = len;
    if (TAB_LDS) {
        u64 *lc = reinterpret_cast<u64 *>(smem); uint8_t *ll = reinterpret_cast<uint8_t *>(smem) + (size_t)nbins * 8;
        for (unsigned i = threadIdx.x; i < nbins; i += 256) { lc[i] = code[i]; ll[i] = len[i]; }
        lcode = lc; llen = ll;
        __syncthreads();
    }
    const uint16_t *cb = codes + (int64_t)b * g.bel;
    const T *box = reinterpret_cast<const T *>(szh_omp_box_origin_bytes(g, b, sizeof(T), data));
    const bool aligned = (g.bel & 7) == 0;
    const u64 gbit0 = bit0 + box_off[b] * 8;
    u64 done_bits = 0, done_zero = 0;
    const u64 ubase = uoff[b];
    uint16_t cn[8];
    omp_load8(cb, (int)threadIdx.x * 8, g.bel, aligned, cn);
    for (int q = 0; q < g.cpb; ++q) {
        const int p0 = q * SZH_ENC_CHUNK + (int)threadIdx.x * 8;
        uint16_t cc[8];
        for (int e = 0; e < 8; ++e) cc[e] = cn[e];
        if (q + 1 < g.cpb) omp_load8(cb, p0 + SZH_ENC_CHUNK, g.bel, aligned, cn);
        unsigned l[8]; unsigned s = 0, zmask = 0;
        for (int e = 0; e < 8; ++e) { const bool in = p0 + e < g.bel; l[e] = in ? (unsigned)llen[cc[e]] : 0u; s += l[e]; if (in && cc[e] == 0) zmask |= 1u << e; }
        u64 tot;
        const u64 ex2 = block_excl_scan_256(((u64)__builtin_popcount(zmask) << 32) | s, sh, &tot);     // (a chunk: < 2^32 bits)
        const unsigned ex = (unsigned)ex2, tot_bits = (unsigned)tot;
        const u64 gbit = gbit0 + done_bits;
        const unsigned lead = (unsigned)(gbit & 31);
        for (unsigned w = threadIdx.x; w < ((lead + tot_bits + 31) >> 5) + 1; w += 256) buf[w] = 0;
        __syncthreads();
        // verbatim values: requested here, stored at the end of the chunk (they come from HBM: waited for on the spot, every chunk of the
        // box's walk would carry a memory round trip -- measured, 0.26 ms of this kernel at 512^3)
        T vals[8];
        if (zmask) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int p = p0 + e, k = p / (g.c1 * g.c2), r = p - k * (g.c1 * g.c2), i = r / g.c2, j = r - i * g.c2;
                vals[e] = (zmask >> e & 1u) ? box[(int64_t)k * g.d0 + (int64_t)i * g.d1 + j] : (T)0;
            }
        }
        unsigned pos = lead + ex;
        u64 acc = 0; int accn = 0;
        for (int e = 0; e < 8; ++e) {
            if (!l[e]) continue;
            const u64 cw = lcode[cc[e]];
            if (accn + (int)l[e] > 64) { lds_put_bits(buf, pos, acc, accn); pos += accn; acc = 0; accn = 0; }
            acc = l[e] == 64 ? cw : ((acc << l[e]) | (cw & ((1ull << l[e]) - 1)));
            accn += (int)l[e];
        }
        if (accn) lds_put_bits(buf, pos, acc, accn);
        __syncthreads();
        const unsigned nwords = (lead + tot_bits + 31) >> 5;
        const u64 w0 = gbit >> 5;
        for (unsigned w = threadIdx.x; w < nwords; w += 256) {
            const unsigned v = __builtin_bswap32(buf[w]);
            if (w == 0 || w == nwords - 1) { if (v) atomicOr(&out32[w0 + w], v); }
            else out32[w0 + w] = v;
        }
        if (zmask) {
            u64 rank = ubase + done_zero + (ex2 >> 32);
#pragma unroll
            for (int e = 0; e < 8; ++e) if (zmask >> e & 1u) unpred[rank++] = vals[e];
        }
        done_bits += tot_bits; done_zero += tot >> 32;
        __syncthreads();                                          // the buffer is read out before the next chunk clears it
    }
    if (threadIdx.x == 0 && (((done_bits + 7) >> 3) != box_bytes[b] || done_zero != (u64)ucount[b])) atomicAdd(bad, 1u);
}

// Everything between the code book and the packing in ONE launch of one workgroup: the boxes' payload sizes (from their histograms), where
// the payloads and the verbatim values of every box begin (two scans), the totals.  Before: k_omp_box_bits_h + two three-launch scans +
// their gaps, ~50 us for 4096 boxes.
// have_bytes: box_bytes is already there (k_omp_box_bits_p, many boxes: one workgroup reading every box's histogram took 0.22 ms for 32 768 boxes)
__global__ __launch_bounds__(1024) void k_omp_layout(int nb, unsigned nbins, const unsigned *__restrict__ hist_box, const u64 *__restrict__ packed, const u64 *__restrict__ ucount64,
                                                     u64 *box_bytes, u64 *box_off, u64 *uoff, u64 *total_bytes, u64 *total_unpred, int have_bytes)
{
    __shared__ u64 sa[16], sb[16];
    __shared__ unsigned char llen[2048];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (unsigned i = tid; i < nbins; i += 1024) llen[i] = (unsigned char)(packed[i] & 0xffu);
    __syncthreads();
    u64 carry_b = 0, carry_u = 0;
    for (int base = 0; base < nb; base += 1024) {
        const int b = base + tid;
        u64 bytes = 0, un = 0;
        if (b < nb) {
            if (have_bytes) bytes = box_bytes[b];
            else {
                u64 bits = 0;
                const unsigned *h = hist_box + (int64_t)b * nbins;
                for (unsigned i = 0; i < nbins; ++i) bits += (u64)h[i] * llen[i];
                bytes = (bits + 7) >> 3;
                box_bytes[b] = bytes;
            }
            un = ucount64[b];
        }
        u64 ib = bytes, iu = un;
        for (int o = 1; o < 64; o <<= 1) { const u64 tb = __shfl_up(ib, o, 64), tu = __shfl_up(iu, o, 64); if (lane >= o) { ib += tb; iu += tu; } }
        if (lane == 63) { sa[wid] = ib; sb[wid] = iu; }
        __syncthreads();
        u64 pb = carry_b, pu = carry_u, tb = 0, tu = 0;
        for (int w = 0; w < 16; ++w) { if (w < wid) { pb += sa[w]; pu += sb[w]; } tb += sa[w]; tu += sb[w]; }
        if (b < nb) { box_off[b] = pb + ib - bytes; uoff[b] = pu + iu - un; }
        carry_b += tb; carry_u += tu;
        __syncthreads();
    }
    if (tid == 0) { *total_bytes = carry_b; *total_unpred = carry_u; uoff[nb] = carry_u; }
}

// Third form (round 4): the two above spend ~50 instructions per code -- eight codes a thread, every one of them an LDS atomic, a block scan
// and three barriers per 2048 codes.  Here a thread takes 32 CONSECUTIVE codes of a round of 8192 and packs them one after the other
// through a 64-bit accumulator into the workgroup's LDS window: whole words are plain stores (nobody else owns them), only its first and
// last word are shared with its neighbours (atomic OR).  One scan and two barriers per 8192 codes.  Needs every code word <= 32 bits
// (host: else k_omp_encode_box) and packed table entries `code << 8 | len`.  Dynamic LDS: [u64 entry[nbins]][window: 8192 maxlen / 32 + 4 words].
#define SZH_OMP_R3 8192
// what the kernel also writes into the stream (every box its own entries; byte stores: the tables lie wherever the tree's size puts them):
// the header bytes (box 0), ucount[b], first[b], the payload size, the verbatim values -- sz_omp.c:233-262,279-280
struct szh_omp_tables {
    unsigned char *stream;             // null: nothing of this (the caller copies the tables)
    const unsigned char *hdr; unsigned hdr_len;
    u64 off_ucount, off_first, off_unpred, off_sizes;
    const void *first;
    int dbg;                           // development: 1 = no verbatim values, 2 = no packing, 4 = no read-out, 8 = no counting pass (wrong streams: timing only)
};
// the e-th point after one at (row offset off0, column j0) of a box, e < 64: over the row's end into the next rows / the next plane
__device__ __forceinline__ int64_t omp_point_off(const szh_omp_geom &g, int64_t off0, int j0, int e)
{
    int j = j0 + e; int64_t off = off0;
    if (j >= g.c2) {                                                // (rare for boxes 32 wide: a thread's 32 codes are one row)
        // rows are g.d1 apart; after the last row of a plane comes the first of the next: step row by row
        int i = (int)((off0 % g.d0) / g.d1);
        while (j >= g.c2) { j -= g.c2; ++i; off += g.d1; if (i == g.c1) { i = 0; off += g.d0 - (int64_t)g.c1 * g.d1; } }
    }
    return off + j;
}
__device__ __forceinline__ void omp_put_bytes(unsigned char *dst, const void *src, int n) { const unsigned char *q = (const unsigned char *)src; for (int i = 0; i < n; ++i) dst[i] = q[i]; }
template <class T>
__global__ __launch_bounds__(256) void k_omp_encode_box3(szh_omp_geom g, const T *__restrict__ data, const uint16_t *__restrict__ codes, const u64 *__restrict__ packed,
                                                         unsigned nbins, unsigned maxlen, const u64 *__restrict__ box_off, const u64 *__restrict__ box_bytes,
                                                         const u64 *__restrict__ uoff, const unsigned *__restrict__ ucount, u64 bit0, unsigned *out32,
                                                         T *__restrict__ unpred, unsigned *bad, szh_omp_tables tb)
{
    SZH_DYN_SMEM(smem);
    __shared__ u64 sh[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    u64 *ltab = reinterpret_cast<u64 *>(smem);
    unsigned *win = reinterpret_cast<unsigned *>(smem + (size_t)nbins * 8);
    for (unsigned i = tid; i < nbins; i += 256) ltab[i] = packed[i];
    const uint16_t *cb = codes + (int64_t)b * g.bel;
    const T *box = reinterpret_cast<const T *>(szh_omp_box_origin_bytes(g, b, sizeof(T), data));
    const u64 gbit0 = bit0 + box_off[b] * 8, ubase = uoff[b];
    u64 done_bits = 0, done_zero = 0;
    const int rounds = (g.bel + SZH_OMP_R3 - 1) / SZH_OMP_R3;
    auto load32 = [&](int r, uint4 (&v)[4]) {                      // the thread's 32 codes of round r (bel: a multiple of 8; pieces past the end: 0xffff)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = r * SZH_OMP_R3 + tid * 32 + k * 8;
            v[k] = (r < rounds && p + 8 <= g.bel) ? *reinterpret_cast<const uint4 *>(cb + p) : make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
        }
    };
    uint4 vn[4];
    load32(0, vn);
    if (tb.stream) {
        if (tid == 0) {
            const unsigned uc = ucount[b]; const T fv = reinterpret_cast<const T *>(tb.first)[b]; const u64 bb = box_bytes[b];
            omp_put_bytes(tb.stream + tb.off_ucount + (u64)b * 4, &uc, 4);
            omp_put_bytes(tb.stream + tb.off_first + (u64)b * sizeof(T), &fv, (int)sizeof(T));
            omp_put_bytes(tb.stream + tb.off_sizes + (u64)b * 8, &bb, 8);
        }
        if (b == 0) for (unsigned i = tid; i < tb.hdr_len; i += 256) tb.stream[i] = tb.hdr[i];
    }
    auto put_val = [&](u64 rank, T val) {
        if (unpred) unpred[rank] = val; else omp_put_bytes(tb.stream + tb.off_unpred + rank * sizeof(T), &val, (int)sizeof(T));
    };
    __syncthreads();                                               // (the table is in LDS)
    for (int r = 0; r < rounds; ++r) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = vn[k];
        load32(r + 1, vn);
        const int p0 = r * SZH_OMP_R3 + tid * 32;
        // ---- bits and zero codes of the thread's 32 codes
        unsigned s = 0, z = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned wv[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
            if (p0 + k * 8 + 8 <= g.bel)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned c0 = wv[q] & 0xffffu, c1 = wv[q] >> 16;
                    s += (unsigned)(ltab[c0] & 0xffu) + (unsigned)(ltab[c1] & 0xffu); z += (c0 == 0) + (c1 == 0);
                }
        }
        u64 tot;
        const u64 ex2 = block_excl_scan_256(((u64)z << 32) | s, sh, &tot);
        const unsigned tot_bits = (unsigned)tot;
        const u64 gbit = gbit0 + done_bits;
        const unsigned lead = (unsigned)(gbit & 31);
        const unsigned nwords = (lead + tot_bits + 31) >> 5;
        for (unsigned w = tid; w < nwords + 1; w += 256) win[w] = 0;
        // ---- verbatim values: requested now, stored after the round (they come from HBM: waited for on the spot, every round of the
        // box's walk would carry a memory round trip).  The thread's 32 codes are consecutive points of the box: (k, i, j) of the first
        // one by division, the others by stepping; the zero codes as a bit mask
        unsigned zm = 0;
        T vals[4]; int nv = 0;                                      // (a thread with more than four of them fetches the rest at the end)
        int64_t zoff0 = 0; int zj0 = 0;
        if (z && !(tb.dbg & 1)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned wv[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                if (p0 + k * 8 + 8 <= g.bel)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const unsigned c = (e & 1) ? wv[e >> 1] >> 16 : wv[e >> 1] & 0xffffu; zm |= (c == 0 ? 1u : 0u) << (k * 8 + e); }
            }
            const int kk = p0 / (g.c1 * g.c2), rr = p0 - kk * (g.c1 * g.c2), i = rr / g.c2;
            zj0 = rr - i * g.c2; zoff0 = (int64_t)kk * g.d0 + (int64_t)i * g.d1;
            unsigned m = zm;
            for (int q = 0; q < 4 && m; ++q) {
                const int e = __builtin_ctz(m); m &= m - 1;
                const T val = box[omp_point_off(g, zoff0, zj0, e)];
                if (q == 0) vals[0] = val; else if (q == 1) vals[1] = val; else if (q == 2) vals[2] = val; else vals[3] = val;
                ++nv;
            }
        }
        __syncthreads();
        // ---- pack: acc holds `nb` pending bits (top-aligned at bit nb - 1); a full word leaves as soon as there are 32
        if (s && !(tb.dbg & 2)) {
            const unsigned bitpos = lead + (unsigned)ex2;
            unsigned wpos = bitpos >> 5, nb = bitpos & 31u;
            u64 acc = 0;
            bool first = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned wv[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                if (p0 + k * 8 + 8 <= g.bel)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned c = (e & 1) ? wv[e >> 1] >> 16 : wv[e >> 1] & 0xffffu;
                        const u64 en = ltab[c];
                        const unsigned le = (unsigned)(en & 0xffu);
                        acc = (acc << le) | (en >> 8);
                        nb += le;
                        if (nb >= 32) {
                            const unsigned word = (unsigned)(acc >> (nb - 32));
                            if (first) { atomicOr(&win[wpos], word); first = false; } else win[wpos] = word;
                            ++wpos; nb -= 32;
                        }
                    }
            }
            if (nb) {                                              // the last, partial word (shared with the next thread); a first word that never filled up too
                const unsigned word = (unsigned)(acc << (32 - nb));
                // (if `first` is still set the pending bits include the lead-in positions, which are zero in acc: the OR leaves them alone)
                atomicOr(&win[wpos], word);
            }
        }
        __syncthreads();
        const u64 w0 = gbit >> 5;
        if (!(tb.dbg & 4)) for (unsigned w = tid; w < nwords; w += 256) {
            const unsigned x = __builtin_bswap32(win[w]);
            if (w == 0 || w == nwords - 1) { if (x) atomicOr(&out32[w0 + w], x); }
            else out32[w0 + w] = x;
        }
        if (zm) {
            u64 rank = ubase + done_zero + (ex2 >> 32);
            if (nv > 0) put_val(rank, vals[0]);
            if (nv > 1) put_val(rank + 1, vals[1]);
            if (nv > 2) put_val(rank + 2, vals[2]);
            if (nv > 3) put_val(rank + 3, vals[3]);
            unsigned m = zm;
            for (int q = 0; q < 4 && m; ++q) m &= m - 1;
            rank += 4;
            while (m) { const int e = __builtin_ctz(m); m &= m - 1; put_val(rank++, box[omp_point_off(g, zoff0, zj0, e)]); }
        }
        done_bits += tot_bits; done_zero += tot >> 32;
        __syncthreads();                                          // the window is read out before the next round clears it
    }
    if (tid == 0 && (((done_bits + 7) >> 3) != box_bytes[b] || done_zero != (u64)ucount[b])) atomicAdd(bad, 1u);
}

// ---- decoding with the look-up table of k_hdec_* (szhip_kernels.h: hdec_run_lut, SZH_LUT_BITS bits and up to four symbols a look-up).
// A workgroup per box; dynamic LDS: [the payload: byte-swapped words, swizzled (SZH_HDEC_SWZ), `stage_bytes` of them + slack]
// [the look-up table SZH_LUT_BYTES][the node table when tab_lds].  The box's 256 stretches find their starts as in k_omp_hdec.
__global__ __launch_bounds__(256) void k_omp_hdec_lut(int bel, const unsigned char *__restrict__ payload, unsigned bytes_before, const u64 *__restrict__ box_off,
                                                      const u64 *__restrict__ box_bytes, const unsigned *__restrict__ table, int n_nodes, int tab_lds,
                                                      const uint4 *__restrict__ lut, unsigned stage_bytes, uint16_t *__restrict__ codes, unsigned *__restrict__ bad, int nb, int per_wg)
{
    SZH_DYN_SMEM(smem);
    __shared__ unsigned s_start[257], s_flag[2];
    __shared__ u64 sh[8];
    const int tid = threadIdx.x;
    // the tables once per workgroup; then `per_wg` boxes one after the other (small boxes: 20 KB of tables per 1 KB of payload otherwise)
    const unsigned lds_words = (stage_bytes + 16) / 4;
    char *q = smem + ((SZH_HDEC_SWZ(lds_words) * 4 + 15) / 16 * 16);
    const SZH_LDS void *lutw = (const SZH_LDS void *)(q + SZH_LUT_SIZE * 16);
    const SZH_LDS void *lut4 = (const SZH_LDS void *)q;
    hdec_copy16<false>(reinterpret_cast<uint4 *>(q), lut, SZH_LUT_BYTES / 16);
    q += SZH_LUT_BYTES;
    const SZH_LDS unsigned *ltab = nullptr;
    if (tab_lds) { hdec_copy16<false>(reinterpret_cast<uint4 *>(q), reinterpret_cast<const uint4 *>(table), (2 * n_nodes + 3) / 4); ltab = (const SZH_LDS unsigned *)q; }
    for (int b = (int)blockIdx.x * per_wg; b < nb && b < ((int)blockIdx.x + 1) * per_wg; ++b) {
    __syncthreads();                                               // (the previous box's payload has been read)
    uint16_t *out = codes + (int64_t)b * bel;
    const unsigned nbytes = (unsigned)box_bytes[b];
    // ---- staging: from a 16-byte aligned address at or below the payload's first byte
    const unsigned char *bits = payload + box_off[b];
    unsigned lead = (unsigned)((uintptr_t)bits & 15u);
    if ((u64)lead > (u64)bytes_before + box_off[b]) lead = 0;       // (never in front of the buffer; then the loads below are unaligned but valid)
    const unsigned n16 = (lead + nbytes + 15) / 16;
    hdec_stage16(reinterpret_cast<unsigned *>(smem), reinterpret_cast<const uint4 *>(bits - lead), (int)n16);
    if (tid == 0) { reinterpret_cast<unsigned *>(smem)[SZH_HDEC_SWZ(4 * n16)] = 0u; s_flag[0] = 0u; s_flag[1] = 0u; }     // the slack word hdec_w32 may touch
    __syncthreads();
    const SZH_LDS unsigned *l = (const SZH_LDS unsigned *)smem;
    const unsigned base = lead * 8, total = base + nbytes * 8;           // local bit positions: the payload is [base, total)
    unsigned sb = (nbytes * 8 + 255u) / 256u; if (sb < 64u) sb = 64u;
    const unsigned first = base + (unsigned)tid * sb, limit = first + sb;
    unsigned start = first, endp = first, cnt = 0;
    bool redo = true;
    // (the warm-up start of k_hdec_pass -- decoding from 128 bits in front of the stretch -- was tried here in round 4: no gain, 0.53 against 0.51 ms)
    for (int round = 0; round < 257; ++round) {
        const bool run = redo && start < limit && start < total;
        if (redo) {
            // (all lanes of a wavefront enter: hdec_run_lut has no wavefront-wide operations, `run` only spares the work)
            cnt = hdec_run_lut<false>(l, total, ltab, table, lutw, run ? start : 0u, run ? limit : 0u, &endp, nullptr, 0, 0, run);
            if (!run) { cnt = 0; endp = start; }
            s_start[tid + 1] = endp;
        }
        __syncthreads();
        if (tid == 0) s_flag[(round + 1) & 1] = 0u;
        redo = false;
        if (tid > 0) { const unsigned ns = s_start[tid]; if (ns != start) { start = ns; redo = true; } }
        if (redo) s_flag[round & 1] = 1u;
        __syncthreads();
        if (!s_flag[round & 1]) break;                         // uniform
    }
    u64 tot;
    const unsigned o = (unsigned)block_excl_scan_256((u64)cnt, sh, &tot);
    if (tot < (u64)bel) { if (tid == 0) atomicAdd(bad, 1u); }   // a payload that holds fewer symbols than the box has points
    const bool wr = cnt && o < (unsigned)bel;
    unsigned e;
    int64_t oend = wr ? (int64_t)o + cnt : (int64_t)o;
    if (oend > bel) oend = bel;
    hdec_run_lut<true>(l, total, ltab, table, lut4, wr ? start : 0u, wr ? limit : 0u, &e, out, (int64_t)o, oend, wr);
    }
}
