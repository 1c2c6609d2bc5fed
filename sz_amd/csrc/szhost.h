/* szhost.h -- host-side C helpers of the MI355X SZ2 path (product code, plain C).
 * These are the short, inherently serial pieces that sit between the HIP kernels:
 * the Huffman tree (heap order decides the code book), the interval decision from
 * the sampled histograms, the regression-coefficient chain, and the stream framing.
 * Reference citations (relative to the reference tree) are given per function in szhost.c. */
#ifndef SZHOST_H
#define SZHOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- Huffman code book ---- */
typedef struct szhost_huff {
    int       state_num;   /* alphabet size the reference allocates: 2*intervals */
    int       n_nodes;     /* nodes of the tree = 2*distinct-1 */
    /* per symbol */
    uint64_t *code;        /* right-aligned code bits */
    uint8_t  *len;         /* code length, 0 if unused */
    /* pre-order serialisation arrays (index-based tree), n_nodes entries each */
    uint32_t *L, *R, *C;
    uint8_t  *t;
    uint64_t  total_bits;  /* sum freq*len */
} szhost_huff;

/* build from a histogram of `nbins` bins (bins >= nbins are zero); returns NULL if every bin is zero */
szhost_huff *szhost_huff_build(int state_num, const uint32_t *hist32, const uint64_t *hist64, size_t nbins);
/* rebuild tables from the serialised tree of a stream */
szhost_huff *szhost_huff_from_bytes(int state_num, const unsigned char *bytes, int node_count);
size_t szhost_huff_tree_size(const szhost_huff *h);                 /* serialised size */
size_t szhost_huff_serial_size(int node_count);                     /* serialised size of a tree with that many nodes */
void   szhost_huff_tree_write(const szhost_huff *h, unsigned char *out);
size_t szhost_huff_encode_i32(const szhost_huff *h, const int *s, size_t n, unsigned char *out);
/* decodes n symbols from at most in_bytes bytes; returns 0 if the payload runs out first (corrupt stream) */
int    szhost_huff_decode_i32(const szhost_huff *h, const unsigned char *in, size_t in_bytes, size_t n, int *out);
/* device decode table: entry[2*node+bit] = child index, or 0x80000000|symbol when the child is a leaf */
void   szhost_huff_decode_table(const szhost_huff *h, uint32_t *table);
void   szhost_huff_free(szhost_huff *h);

/* ---- interval optimiser decision from the sampled histograms ---- */
typedef struct szhost_decision {
    unsigned intervals;
    int      use_mean;
    double   dense_pos;    /* T-rounded */
    double   mean_freq, sample_freq;
} szhost_decision;
void szhost_decide(int is_double, const uint32_t *radius_hist, unsigned max_radius, const uint32_t *freq_hist,
                   uint64_t sample_count, uint64_t within_eb, float pred_threshold, double ebD, double mean,
                   szhost_decision *out);
/* sequential T-typed sum of the strided samples (the `mean` of the optimiser) */
double szhost_seq_mean(int is_double, const void *samples, size_t count);

/* ---- regression coefficient chain ---- */
typedef struct szhost_coeffs {
    size_t   reg_count;
    int     *codes[4];       /* [reg_count] each */
    void    *unpred[4];      /* T each */
    size_t   unpred_count[4];
    double   prec[4];        /* T-rounded precisions */
} szhost_coeffs;
/* coef: SoA [ncoef][nblocks] fitted coefficients (T); indicator: 1 = Lorenzo.  Overwrites coef for regression blocks
 * with the DECODED coefficients (what the decompressor will use).  ncoef = 4: 3-D; ncoef = 3: 2-D (late0 unused). */
void szhost_coeff_chain(int is_double, void *coef, const unsigned char *indicator, size_t nblocks, double eb,
                        int late0, int late1, int late2, int use_mean, int ncoef, szhost_coeffs *out);
/* the same in pieces: the chains of the ncoef coefficients are independent (one thread each) */
void szhost_coeff_chain_begin(int is_double, const unsigned char *indicator, size_t nblocks, double eb,
                              int late0, int late1, int late2, int ncoef, szhost_coeffs *out);
void szhost_coeff_chain_one(int is_double, void *coef, const unsigned char *indicator, size_t nblocks, int use_mean, int e, szhost_coeffs *out);
/* the same, publishing the number of regression blocks finished so far in *progress (release stores, every 1024 blocks and at the end) */
void szhost_coeff_chain_one_p(int is_double, void *coef, const unsigned char *indicator, size_t nblocks, int use_mean, int e, szhost_coeffs *out,
                              size_t *progress);
/* the same, started before all coefficients are in memory: *avail = blocks whose coefficients (all of the chain's array) have arrived, raised by the caller
 * from another thread up to nblocks; the chain waits where it runs into the mark */
void szhost_coeff_chain_one_pa(int is_double, void *coef, const unsigned char *indicator, size_t nblocks, int use_mean, int e, szhost_coeffs *out,
                               size_t *progress, const size_t *avail);
/* the reference's loop, literally (the fall-back and the tests' yardstick of szhost_coeff_chain_one_p, which takes the arithmetic off the chain) */
void szhost_coeff_chain_one_ref(int is_double, void *coef, const unsigned char *indicator, size_t nblocks, int use_mean, int e, szhost_coeffs *out,
                                size_t *progress);
void szhost_coeffs_free(szhost_coeffs *c);
/* inverse: codes+unpred -> decoded coefficients written into coef SoA [ncoef][nblocks] for regression blocks */
void szhost_coeff_unchain(int is_double, void *coef, const unsigned char *indicator, size_t nblocks,
                          int *const codes[4], const int radius[4], const double prec[4],
                          const unsigned char *const unpred[4], int ncoef);

/* ---- stream framing ---- */
typedef struct szhost_meta {
    int data_type;          /* 0 float, 1 double */
    int err_mode;           /* effective mode written in the params bytes */
    double abs_bound, rel_ratio, psnr, pwr_ratio;
    double vmin, vmax;
    int opt_quant_mode;     /* exe_params->optQuantMode */
    int data_endian, sz_mode, gzip_mode;
    int sample_distance; float pred_threshold;
    int sol_id;
    unsigned max_quant_intervals, quantization_intervals;
    int protect_value_range;
} szhost_meta;
/* writes 3 version bytes + flag byte + 28/36 parameter bytes; returns the length */
size_t szhost_write_meta(const szhost_meta *m, unsigned char flags, unsigned char *out);

void     szhost_put_u32be(unsigned char *b, uint32_t v);
void     szhost_put_u64be(unsigned char *b, uint64_t v);
uint32_t szhost_get_u32be(const unsigned char *b);
uint64_t szhost_get_u64be(const unsigned char *b);
void     szhost_put_f32be(unsigned char *b, float v);
void     szhost_put_f64be(unsigned char *b, double v);
float    szhost_get_f32be(const unsigned char *b);
double   szhost_get_f64be(const unsigned char *b);

#ifdef __cplusplus
}
#endif
#endif
