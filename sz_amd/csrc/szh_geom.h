// szh_geom.h -- block-grid geometry and the quantisation rule of the SZ 2.1 hot path,
// shared by host C++ and device code.
//
// Reference semantics (paths relative to the reference tree):
//   block grid        sz/include/sz.h:93-123 (SZ_COMPUTE_3D_NUMBER_OF_BLOCKS, SZ_COMPUTE_BLOCKCOUNT)
//   type-array order  sz/src/sz_float.c:7064,7359  (codes are stored block by block)
//   quantiser rule    sz/src/sz_float.c:7267-7287 (Lorenzo), :7164-7183 (regression)
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#define SZH_HD __host__ __device__ __forceinline__
#else
#define SZH_HD inline
#endif

#define SZH_BLOCK_SIZE 6      /* 3-D: sz/src/sz_float.c:6531 */
#define SZH_BLOCK_SIZE_2D 16  /* 2-D: sz/src/sz_float.c:5518 */

// one dimension of the block grid
struct szh_grid1 {
    int count;  // extent of this dimension
    int num;    // number of blocks
    int early;  // width of the first `split` blocks
    int late;   // width of the others
    int split;
};

SZH_HD szh_grid1 szh_make_grid1(int count, int block_size = SZH_BLOCK_SIZE)
{
    szh_grid1 g;
    g.count = count;
    g.num = (count <= block_size) ? 1 : count / block_size;
    g.early = g.late = count / g.num;
    g.split = count % g.num;
    if (g.split) g.early += 1;
    return g;
}
SZH_HD int szh_blk_start(const szh_grid1 &g, int b) { return b < g.split ? b * g.early : b * g.late + g.split; }
SZH_HD int szh_blk_size(const szh_grid1 &g, int b) { return b < g.split ? g.early : g.late; }
// block containing coordinate x
SZH_HD int szh_blk_of(const szh_grid1 &g, int x)
{
    int edge = g.split * g.early;
    return x < edge ? x / g.early : g.split + (x - edge) / g.late;
}

// 3-D geometry; dim 0 is the slowest, dim 2 the fastest (the callee convention of sz_float.c:6527).
// A 2-D array (r1 x r2, sz_float.c:5516) is carried as 1 x r1 x r2 with `ndim` = 2: one block layer in dim 0, 16-wide blocks in
// the other two.  The 7-point Lorenzo stencil then degenerates to the reference's 3-point one exactly (the dim-0 neighbours are
// the zero halo, and x + 0 is exact); the regression plane keeps a zero dim-0 coefficient.
struct szh_geom3 {
    szh_grid1 g0, g1, g2;
    int64_t d0, d1;      // element strides of dim 0 and dim 1
    int64_t n;           // number of elements
    int64_t nblocks;
    int ndim;            // 3, or 2 for the carried 2-D case
    int block_size;      // what the stream header records
};

SZH_HD szh_geom3 szh_make_geom3(int r0, int r1, int r2)
{
    szh_geom3 G;
    G.g0 = szh_make_grid1(r0); G.g1 = szh_make_grid1(r1); G.g2 = szh_make_grid1(r2);
    G.d1 = r2; G.d0 = (int64_t)r1 * r2; G.n = G.d0 * r0;
    G.nblocks = (int64_t)G.g0.num * G.g1.num * G.g2.num;
    G.ndim = 3; G.block_size = SZH_BLOCK_SIZE;
    return G;
}
SZH_HD szh_geom3 szh_make_geom2(int r1, int r2)
{
    szh_geom3 G;
    G.g0 = szh_make_grid1(1, SZH_BLOCK_SIZE_2D); G.g1 = szh_make_grid1(r1, SZH_BLOCK_SIZE_2D); G.g2 = szh_make_grid1(r2, SZH_BLOCK_SIZE_2D);
    G.d1 = r2; G.d0 = (int64_t)r1 * r2; G.n = G.d0;
    G.nblocks = (int64_t)G.g1.num * G.g2.num;
    G.ndim = 2; G.block_size = SZH_BLOCK_SIZE_2D;
    return G;
}

// position in the block-ordered type array of the first element of block column (b0,b1) ...
SZH_HD int64_t szh_code_base01(const szh_geom3 &G, int b0, int b1)
{
    int64_t o0 = szh_blk_start(G.g0, b0), o1 = szh_blk_start(G.g1, b1), s0 = szh_blk_size(G.g0, b0);
    return o0 * G.d0 + o1 * s0 * G.d1;
}
// ... and of block (b0,b1,b2)
SZH_HD int64_t szh_code_base(const szh_geom3 &G, int b0, int b1, int b2)
{
    int64_t s0 = szh_blk_size(G.g0, b0), s1 = szh_blk_size(G.g1, b1), o2 = szh_blk_start(G.g2, b2);
    return szh_code_base01(G, b0, b1) + s0 * s1 * o2;
}

// --- the quantisation rule, identical for predictor kinds apart from `capacity` ---
// returns the code (0 = unpredictable); *recon receives the value the decompressor will see.
SZH_HD float szh_abs(float x) { return __builtin_fabsf(x); }
SZH_HD double szh_abs(double x) { return __builtin_fabs(x); }

template <class T>
SZH_HD int szh_quant_point(T x, T pred, T eb, T recip, int capacity, int radius, T *recon)
{
    T diff = x - pred;
    T itv = szh_abs(diff) * recip + 1;
    if (itv < (T)capacity) {
        if (diff < 0) itv = -itv;
        int code = (int)(itv / 2) + radius;
        T rc = pred + (T)(2 * (code - radius)) * eb;
        if (szh_abs(x - rc) > eb) { *recon = x; return 0; }
        *recon = rc;
        return code;
    }
    *recon = x;
    return 0;
}
