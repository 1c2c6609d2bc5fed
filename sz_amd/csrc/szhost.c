/* szhost.c -- host-side C of the MI355X SZ2 path (see szhost.h).  Plain C99, no GPU calls.
 *
 * Reference behaviour reproduced here (paths relative to the reference tree):
 *   Huffman tree: leaves inserted in ascending symbol order into a 1-based binary min-heap whose
 *     sift-up stops on parent<=child and whose sift-down takes the right child only on strict <
 *     (sz/src/Huffman.c:76-114,165-185); of the two nodes removed per merge the first becomes the
 *     RIGHT child (bit 1), the second the LEFT child (bit 0) -- gcc evaluates the arguments of
 *     new_node(...,qremove(),qremove()) right to left (Huffman.c:181);
 *   tree bytes: pre-order arrays L,R,C,t behind one endian byte (Huffman.c:443-585);
 *   bit packing: MSB first, zero padded (Huffman.c:205-308);
 *   interval decision: sz/src/sz_float.c:6486-6522,6650;
 *   coefficient chain: sz/src/sz_float.c:7126-7152 (with mean :6790-6812), inverse szd_float.c:5809-5820;
 *   params bytes: sz/src/ByteToolkit.c:874-972; flag byte: sz/src/dataCompression.c:686-709.
 */
#include "szhost.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

void szhost_put_u32be(unsigned char *b, uint32_t v) { b[0] = (unsigned char)(v >> 24); b[1] = (unsigned char)(v >> 16); b[2] = (unsigned char)(v >> 8); b[3] = (unsigned char)v; }
void szhost_put_u64be(unsigned char *b, uint64_t v) { szhost_put_u32be(b, (uint32_t)(v >> 32)); szhost_put_u32be(b + 4, (uint32_t)v); }
uint32_t szhost_get_u32be(const unsigned char *b) { return ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | (uint32_t)b[3]; }
uint64_t szhost_get_u64be(const unsigned char *b) { return ((uint64_t)szhost_get_u32be(b) << 32) | szhost_get_u32be(b + 4); }
void szhost_put_f32be(unsigned char *b, float v) { uint32_t u; memcpy(&u, &v, 4); szhost_put_u32be(b, u); }
void szhost_put_f64be(unsigned char *b, double v) { uint64_t u; memcpy(&u, &v, 8); szhost_put_u64be(b, u); }
float szhost_get_f32be(const unsigned char *b) { uint32_t u = szhost_get_u32be(b); float v; memcpy(&v, &u, 4); return v; }
double szhost_get_f64be(const unsigned char *b) { uint64_t u = szhost_get_u64be(b); double v; memcpy(&v, &u, 8); return v; }

/* ------------------------------------------------------------------ Huffman */

typedef struct { uint64_t w; int lch, rch; uint32_t sym; } hnode; /* lch < 0: leaf */

static void heap_push(int *hp, int *hn, const hnode *nd, int id)
{
    int i = ++(*hn);
    while (i > 1) {
        int par = i >> 1;
        if (nd[hp[par]].w <= nd[id].w) break;
        hp[i] = hp[par];
        i = par;
    }
    hp[i] = id;
}

static int heap_pop(int *hp, int *hn, const hnode *nd)
{
    int top = hp[1];
    int last = hp[*hn];
    (*hn)--;
    int n = *hn; /* valid entries 1..n after moving `last` to the root */
    if (n >= 1) {
        hp[1] = last;
        int i = 1;
        for (;;) {
            int c = i << 1;
            if (c > n) break;
            if (c + 1 <= n && nd[hp[c + 1]].w < nd[hp[c]].w) c++;
            if (nd[hp[i]].w > nd[hp[c]].w) { int t = hp[i]; hp[i] = hp[c]; hp[c] = t; i = c; }
            else break;
        }
    }
    return top;
}

static szhost_huff *huff_alloc(int state_num, int n_nodes)
{
    szhost_huff *h = (szhost_huff *)calloc(1, sizeof(*h));
    h->state_num = state_num;
    h->n_nodes = n_nodes;
    h->code = (uint64_t *)calloc((size_t)state_num, sizeof(uint64_t));
    h->len = (uint8_t *)calloc((size_t)state_num, 1);
    h->L = (uint32_t *)calloc((size_t)n_nodes, 4);
    h->R = (uint32_t *)calloc((size_t)n_nodes, 4);
    h->C = (uint32_t *)calloc((size_t)n_nodes, 4);
    h->t = (uint8_t *)calloc((size_t)n_nodes, 1);
    return h;
}

void szhost_huff_free(szhost_huff *h)
{
    if (!h) return;
    free(h->code); free(h->len); free(h->L); free(h->R); free(h->C); free(h->t); free(h);
}

/* assign codes by walking the serialised (pre-order) tree: left edge 0, right edge 1.
 * The arrays may come from an untrusted stream: every node must be reached exactly once, an internal node has two children with
 * larger (pre-order) indices, so the walk can neither cycle nor outgrow its stacks. */
static int huff_codes_from_arrays(szhost_huff *h)
{
    int n = h->n_nodes;
    uint32_t *stk = (uint32_t *)malloc((size_t)(n + 2) * sizeof(uint32_t));
    uint64_t *sbits = (uint64_t *)malloc((size_t)(n + 2) * sizeof(uint64_t));
    uint8_t *slen = (uint8_t *)malloc((size_t)(n + 2));
    uint8_t *seen = (uint8_t *)calloc((size_t)n + 1, 1);
    int sp = 0, ok = stk && sbits && slen && seen;
    int visited = 0;
    if (ok) { stk[0] = 0; sbits[0] = 0; slen[0] = 0; sp = 1; seen[0] = 1; }
    while (ok && sp) {
        sp--;
        uint32_t nd = stk[sp]; uint64_t bits = sbits[sp]; int len = slen[sp];
        visited++;
        if (h->t[nd]) {
            if (len > 64 || h->C[nd] >= (uint32_t)h->state_num) { ok = 0; break; }
            h->code[h->C[nd]] = bits;
            h->len[h->C[nd]] = (uint8_t)len;
            continue;
        }
        const uint32_t l = h->L[nd], r = h->R[nd];
        if (len >= 64 || l <= nd || r <= nd || l >= (uint32_t)n || r >= (uint32_t)n || l == r || seen[l] || seen[r]) { ok = 0; break; }
        seen[l] = seen[r] = 1;
        stk[sp] = r; sbits[sp] = (bits << 1) | 1; slen[sp] = (uint8_t)(len + 1); sp++;
        stk[sp] = l; sbits[sp] = bits << 1; slen[sp] = (uint8_t)(len + 1); sp++;
    }
    if (ok && visited != n) ok = 0;   /* unreachable nodes: not a tree of n nodes */
    free(stk); free(sbits); free(slen); free(seen);
    return ok;
}

szhost_huff *szhost_huff_build(int state_num, const uint32_t *hist32, const uint64_t *hist64, size_t nbins)
{
    size_t lim = (size_t)state_num * 2; /* the reference scans allNodes = 2*stateNum bins */
    if (nbins < lim) lim = nbins;
    size_t distinct = 0;
    for (size_t s = 0; s < lim; s++) if (hist32 ? hist32[s] : hist64[s]) distinct++;
    if (!distinct) return NULL;
    int total = (int)(2 * distinct - 1);
    hnode *nd = (hnode *)malloc((size_t)total * sizeof(hnode));
    int *hp = (int *)malloc((distinct + 2) * sizeof(int));
    int hn = 0, nn = 0;
    uint64_t total_bits_weight = 0;
    for (size_t s = 0; s < lim; s++) {
        uint64_t f = hist32 ? hist32[s] : hist64[s];
        if (!f) continue;
        nd[nn].w = f; nd[nn].lch = -1; nd[nn].rch = -1; nd[nn].sym = (uint32_t)s;
        heap_push(hp, &hn, nd, nn);
        nn++;
    }
    while (hn > 1) {
        int first = heap_pop(hp, &hn, nd);
        int second = heap_pop(hp, &hn, nd);
        nd[nn].w = nd[first].w + nd[second].w;
        nd[nn].rch = first;   /* smallest -> right, bit 1 */
        nd[nn].lch = second;  /* next     -> left,  bit 0 */
        nd[nn].sym = 0;
        heap_push(hp, &hn, nd, nn);
        nn++;
    }
    int root = hp[1];
    szhost_huff *h = huff_alloc(state_num, total);
    /* pre-order numbering, left subtree first (pad_tree_*, Huffman.c:443-501) */
    {
        int *stk = (int *)malloc((size_t)(total + 1) * sizeof(int));
        int *spar = (int *)malloc((size_t)(total + 1) * sizeof(int));
        unsigned char *sright = (unsigned char *)malloc((size_t)(total + 1));
        int sp = 0; uint32_t next = 0;
        stk[0] = root; spar[0] = -1; sright[0] = 0; sp = 1;
        while (sp) {
            sp--;
            int n = stk[sp], par = spar[sp]; unsigned char isr = sright[sp];
            uint32_t idx = next++;
            if (par >= 0) { if (isr) h->R[par] = idx; else h->L[par] = idx; }
            h->C[idx] = nd[n].sym;
            h->t[idx] = nd[n].lch < 0 ? 1 : 0;
            if (nd[n].lch >= 0) {
                stk[sp] = nd[n].rch; spar[sp] = (int)idx; sright[sp] = 1; sp++;
                stk[sp] = nd[n].lch; spar[sp] = (int)idx; sright[sp] = 0; sp++;
            }
        }
        free(stk); free(spar); free(sright);
    }
    free(nd); free(hp);
    if (!huff_codes_from_arrays(h)) { fprintf(stderr, "szhost: Huffman code longer than 64 bits is not supported\n"); szhost_huff_free(h); return NULL; }
    for (size_t s = 0; s < lim; s++) {
        uint64_t f = hist32 ? hist32[s] : hist64[s];
        if (f) total_bits_weight += f * h->len[s];
    }
    h->total_bits = total_bits_weight;
    return h;
}

static int tree_width(int n_nodes) { return n_nodes <= 256 ? 1 : (n_nodes <= 65536 ? 2 : 4); }

size_t szhost_huff_serial_size(int node_count)
{
    size_t n = (size_t)node_count;
    return 1 + 2 * (size_t)tree_width(node_count) * n + 4 * n + n;
}

size_t szhost_huff_tree_size(const szhost_huff *h)
{
    size_t n = (size_t)h->n_nodes;
    return 1 + 2 * (size_t)tree_width(h->n_nodes) * n + 4 * n + n;
}

void szhost_huff_tree_write(const szhost_huff *h, unsigned char *out)
{
    int w = tree_width(h->n_nodes);
    size_t n = (size_t)h->n_nodes;
    unsigned char *p = out;
    *p++ = 0; /* little-endian system */
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t *src = pass ? h->R : h->L;
        for (size_t i = 0; i < n; i++) {
            if (w == 1) p[0] = (unsigned char)src[i];
            else if (w == 2) { uint16_t v = (uint16_t)src[i]; memcpy(p, &v, 2); }
            else memcpy(p, &src[i], 4);
            p += w;
        }
    }
    memcpy(p, h->C, 4 * n); p += 4 * n;
    memcpy(p, h->t, n);
}

szhost_huff *szhost_huff_from_bytes(int state_num, const unsigned char *bytes, int node_count)
{
    if (node_count <= 0) return NULL;
    szhost_huff *h = huff_alloc(state_num, node_count);
    int w = tree_width(node_count);
    size_t n = (size_t)node_count;
    const unsigned char *pL = bytes + 1, *pR = pL + (size_t)w * n, *pC = pR + (size_t)w * n, *pt = pC + 4 * n;
    for (size_t i = 0; i < n; i++) {
        uint32_t l = 0, r = 0;
        if (w == 1) { l = pL[i]; r = pR[i]; }
        else if (w == 2) { uint16_t a, b; memcpy(&a, pL + 2 * i, 2); memcpy(&b, pR + 2 * i, 2); l = a; r = b; }
        else { memcpy(&l, pL + 4 * i, 4); memcpy(&r, pR + 4 * i, 4); }
        if (l >= n || r >= n) { szhost_huff_free(h); return NULL; }
        h->L[i] = l; h->R[i] = r;
        memcpy(&h->C[i], pC + 4 * i, 4);
        h->t[i] = pt[i];
    }
    if (node_count > 256) h->t[0] = 0; /* the reference builds the root with t=0 in the wide forms (Huffman.c:740,780) */
    if (!huff_codes_from_arrays(h)) { szhost_huff_free(h); return NULL; }
    return h;
}

void szhost_huff_decode_table(const szhost_huff *h, uint32_t *table)
{
    for (int i = 0; i < h->n_nodes; i++) {
        uint32_t kids[2] = { h->L[i], h->R[i] };
        for (int b = 0; b < 2; b++) {
            uint32_t c = kids[b];
            if (h->t[i] || c == 0) table[2 * i + b] = 0x80000000u | h->C[i]; /* leaf (only reachable for a 1-node tree) */
            else table[2 * i + b] = h->t[c] ? (0x80000000u | h->C[c]) : c;
        }
    }
}

/* MSB-first packing of the code words (encode(), Huffman.c:205-308).  A 64-bit accumulator that is emptied 32 bits at a time: with fewer than 32 bits
 * pending, a piece of up to 32 bits always fits, so a code is one shift-or and at most one store (round 5: 1.2 -> 0.4 ms for the 303 450 codes of
 * one coefficient section of the 512^3 M-field, the part of a section that the call waits for); codes of 33 - 64 bits go in as two pieces. */
size_t szhost_huff_encode_i32(const szhost_huff *h, const int *s, size_t n, unsigned char *out)
{
    uint64_t acc = 0; int fill = 0; size_t o = 0;
#define PUT_PIECE(bits_, len_) do {                                                                  \
        acc = (acc << (len_)) | (uint64_t)(bits_); fill += (len_);                                   \
        if (fill >= 32) { fill -= 32; szhost_put_u32be(out + o, (uint32_t)(acc >> fill)); o += 4; }  \
    } while (0)
    for (size_t i = 0; i < n; i++) {
        const int len = h->len[s[i]];
        const uint64_t bits = h->code[s[i]];
        if (len <= 32) { if (len > 0) PUT_PIECE(bits, len); }
        else { PUT_PIECE(bits >> 32, len - 32); PUT_PIECE(bits & 0xffffffffu, 32); }
    }
#undef PUT_PIECE
    if (fill) {
        const uint32_t w = (uint32_t)(acc << (32 - fill));
        const int nbytes = (fill + 7) / 8;
        for (int b = 0; b < nbytes; b++) out[o++] = (unsigned char)(w >> (24 - 8 * b));
    }
    return o;
}

/* The walk down the tree, bit by bit, of the reference's decode (Huffman decode of the coefficient codes, szd_float.c:5790-5800 ->
 * decode(), Huffman.c:314-360).  Round 5: the first 12 bits of a code are looked up in a table built by that same walk -- for a code of up to 12 bits
 * the entry holds the SYMBOL and the length, so a common code costs one load off a 64-bit window that is refilled every few symbols; for a longer code it
 * holds the node reached, and the walk goes on bit by bit from there.  The last eight bytes of the payload are decoded bit by bit throughout (the
 * window load reads eight bytes).  Same symbols, same failure as the walk. */
int szhost_huff_decode_i32(const szhost_huff *h, const unsigned char *in, size_t in_bytes, size_t n, int *out)
{
    if (h->t[0]) { for (size_t i = 0; i < n; i++) out[i] = (int)h->C[0]; return 1; }
    enum { K = 12 };
    const size_t max_bits = in_bytes * 8;
    size_t bit = 0, cnt = 0; uint32_t nd = 0;
    if (n >= 4096) {
        /* entry: leaf within K bits: 0x80000000 | symbol << 4 | length (symbols are below 2^27: state_num is at most 131072 here and 65536 * 2 elsewhere);
         * otherwise node << 4 | K */
        uint32_t *tab = (uint32_t *)malloc(sizeof(uint32_t) << K);
        if (tab && h->state_num <= (1 << 27)) {
            for (uint32_t p = 0; p < (1u << K); p++) {
                uint32_t x = 0; int len = 0;
                while (len < K) { x = ((p >> (K - 1 - len)) & 1) ? h->R[x] : h->L[x]; len++; if (h->t[x]) break; }
                tab[p] = h->t[x] ? (0x80000000u | (h->C[x] << 4) | (uint32_t)len) : ((x << 4) | (uint32_t)len);
            }
            while (cnt < n && (bit >> 3) + 8 <= in_bytes) {
                uint64_t w; memcpy(&w, in + (bit >> 3), 8);
                w = __builtin_bswap64(w) << (bit & 7);
                int avail = 64 - (int)(bit & 7);                 /* valid bits at the top of w */
                while (avail >= K && cnt < n) {
                    const uint32_t e = tab[w >> (64 - K)];
                    const int len = (int)(e & 15u);
                    if (e & 0x80000000u) { out[cnt++] = (int)((e >> 4) & 0x7ffffffu); w <<= len; bit += (size_t)len; avail -= len; continue; }
                    /* a code longer than K bits: the rest of the walk, from memory */
                    uint32_t x = e >> 4;
                    bit += (size_t)len;
                    for (;;) {
                        if (bit >= max_bits) { free(tab); return 0; }
                        const int b = (in[bit >> 3] >> (7 - (bit & 7))) & 1; bit++;
                        x = b ? h->R[x] : h->L[x];
                        if (h->t[x]) break;
                    }
                    out[cnt++] = (int)h->C[x];
                    avail = 0;                                   /* refill the window at the new position */
                }
            }
        }
        free(tab);
    }
    while (cnt < n) {
        if (bit >= max_bits) return 0;               /* the payload ends before n symbols: corrupt stream */
        int b = (in[bit >> 3] >> (7 - (bit & 7))) & 1; bit++;
        nd = b ? h->R[nd] : h->L[nd];
        if (h->t[nd]) { out[cnt++] = (int)h->C[nd]; nd = 0; }
    }
    return 1;
}

/* ------------------------------------------------------------------ interval decision */

static unsigned round_up_pow2(unsigned v) { v -= 1; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }

double szhost_seq_mean(int is_double, const void *samples, size_t count)
{
    if (is_double) {
        const double *s = (const double *)samples; double m = 0.0;
        for (size_t i = 0; i < count; i++) m += s[i];
        if (count > 0) m /= (double)count;
        return m;
    } else {
        const float *s = (const float *)samples; float m = 0.0f;
        for (size_t i = 0; i < count; i++) m += s[i];
        if (count > 0) m /= (float)count;
        return (double)m;
    }
}

void szhost_decide(int is_double, const uint32_t *radius_hist, unsigned max_radius, const uint32_t *freq_hist,
                   uint64_t sample_count, uint64_t within_eb, float pred_threshold, double ebD, double mean,
                   szhost_decision *out)
{
    double sample_freq_d = (double)within_eb * 1.0 / (double)sample_count; /* 0/0 -> NaN as in the reference */
    uint64_t target;
    if (is_double) target = (uint64_t)((double)sample_count * (double)pred_threshold);
    else target = (uint64_t)((float)sample_count * pred_threshold);
    uint64_t sum = 0; unsigned i;
    for (i = 0; i < max_radius; i++) { sum += radius_hist[i]; if (sum > target) break; }
    if (i >= max_radius) i = max_radius - 1;
    unsigned p2 = round_up_pow2(2 * (i + 1));
    if (p2 < 32) p2 = 32;
    uint64_t max_sum = 0; size_t max_index = 0;
    for (size_t q = 1; q < 8192 - 2; q++) {
        uint64_t s2 = (uint64_t)freq_hist[q] + freq_hist[q + 1];
        if (s2 > max_sum) { max_sum = s2; max_index = q; }
    }
    double dense = mean + ebD * (double)((int64_t)max_index + 1 - 4096);
    double mean_freq_d = (double)max_sum * 1.0 / (double)sample_count;
    if (is_double) { out->dense_pos = dense; out->mean_freq = mean_freq_d; out->sample_freq = sample_freq_d; }
    else { out->dense_pos = (double)(float)dense; out->mean_freq = (double)(float)mean_freq_d; out->sample_freq = (double)(float)sample_freq_d; }
    out->intervals = p2;
    out->use_mean = (out->mean_freq > 0.5 || out->mean_freq > out->sample_freq) ? 1 : 0;
}

/* ------------------------------------------------------------------ coefficient chain */

void szhost_coeffs_free(szhost_coeffs *c)
{
    for (int e = 0; e < 4; e++) { free(c->codes[e]); free(c->unpred[e]); c->codes[e] = NULL; c->unpred[e] = NULL; }
}

/* A chain may start before all of its coefficients are in memory (round 5: the arrays arrive from the device in two pieces, szhost_coeff_chain_one_pa): the
 * calling thread's g_chain_avail points at the number of blocks whose coefficients have arrived; a loop that comes to a block behind that mark waits.  One
 * compare per step against a local copy, off the chain.  NULL: everything is there. */
static __thread const size_t *g_chain_avail = NULL;
#if defined(__SSE2__)
#define CHAIN_RELAX() __builtin_ia32_pause()
#else
#define CHAIN_RELAX() do { } while (0)
#endif
#define CHAIN_AVAIL_LOCALS const size_t *const avail_p = g_chain_avail; size_t avail_c = avail_p ? 0 : (size_t)-1;
#define CHAIN_WAIT_AVAIL(b_) do { if (__builtin_expect((b_) >= avail_c, 0)) { while ((avail_c = __atomic_load_n(avail_p, __ATOMIC_ACQUIRE)) <= (b_)) CHAIN_RELAX(); } } while (0)

/* one coefficient's chain: the four (three) chains of a block sequence are independent of each other, so callers may run them on
 * different threads (szhost_coeff_chain_begin once, then szhost_coeff_chain_one per e) */
#define CHAIN_ONE(T, FABS, DIVIDE_IN_NOMEAN)                                                              \
    T *cf = (T *)coef + (size_t)e * nblocks;                                                               \
    const T prec = (T)out->prec[e], rprec = 1 / prec;                                                      \
    T last = 0;                                                                                            \
    T *un = (T *)out->unpred[e];                                                                           \
    int *codes = out->codes[e];                                                                            \
    size_t ci = 0, nun = 0;                                                                                \
    CHAIN_AVAIL_LOCALS                                                                                     \
    for (size_t b = 0; b < nblocks; b++) {                                                                 \
        if (indicator[b]) continue;                                                                        \
        CHAIN_WAIT_AVAIL(b);                                                                               \
        T cur = cf[b];                                                                                     \
        T diff = cur - last;                                                                               \
        T itv;                                                                                             \
        if (DIVIDE_IN_NOMEAN && !use_mean) itv = FABS(diff) / prec + 1;                                    \
        else itv = FABS(diff) * rprec + 1;                                                                 \
        int cc = 0;                                                                                        \
        if (itv < 65536) {                                                                                 \
            if (diff < 0) itv = -itv;                                                                      \
            cc = (int)(itv / 2) + 32768;                                                                   \
            last = last + 2 * (cc - 32768) * prec;                                                         \
            if (FABS(cur - last) > prec) { cc = 0; last = cur; un[nun++] = cur; }                          \
        } else { cc = 0; last = cur; un[nun++] = cur; }                                                    \
        codes[ci++] = cc;                                                                                  \
        cf[b] = last;                                                                                      \
        if (progress && (ci & 1023) == 0) __atomic_store_n(progress, ci, __ATOMIC_RELEASE);               \
    }                                                                                                      \
    out->unpred_count[e] = nun;                                                                            \
    if (progress) __atomic_store_n(progress, ci, __ATOMIC_RELEASE);

/* ncoef = 4: 3-D planes {a,b,c,d}, precisions from late0..late2; ncoef = 3: 2-D planes {a,b,c} (sz_float.c:6031-6055), precisions
 * from late1, late2 (late0 unused).  `coef` is SoA [ncoef][nblocks]. */
void szhost_coeff_chain_begin(int is_double, const unsigned char *indicator, size_t nblocks, double eb,
                              int late0, int late1, int late2, int ncoef, szhost_coeffs *out)
{
    memset(out, 0, sizeof(*out));
    size_t reg_count = 0;
    for (size_t b = 0; b < nblocks; b++) if (!indicator[b]) reg_count++;
    out->reg_count = reg_count;
    const size_t esz = is_double ? 8 : 4;
    for (int e = 0; e < ncoef; e++) {
        out->codes[e] = (int *)malloc((reg_count ? reg_count : 1) * sizeof(int));
        out->unpred[e] = malloc((reg_count ? reg_count : 1) * esz);
    }
    /* the precisions, in the data's type (sz_float.c:5608 (2-D), :6640 (3-D)) */
    if (is_double) {
        double ebT = eb, rel = ncoef == 3 ? (double)(0.15 / 3) : (double)0.025, p[4] = {0, 0, 0, 0};
        if (ncoef == 3) { p[0] = rel * ebT / late1; p[1] = rel * ebT / late2; p[2] = rel * ebT; }
        else { p[0] = rel * ebT / late0; p[1] = rel * ebT / late1; p[2] = rel * ebT / late2; p[3] = rel * ebT; }
        for (int e = 0; e < ncoef; e++) out->prec[e] = p[e];
    } else {
        float ebT = (float)eb, rel = ncoef == 3 ? (float)(0.15 / 3) : (float)0.025, p[4] = {0, 0, 0, 0};
        if (ncoef == 3) { p[0] = rel * ebT / late1; p[1] = rel * ebT / late2; p[2] = rel * ebT; }
        else { p[0] = rel * ebT / late0; p[1] = rel * ebT / late1; p[2] = rel * ebT / late2; p[3] = rel * ebT; }
        for (int e = 0; e < ncoef; e++) out->prec[e] = (double)p[e];
    }
}

/* The reference's loop, literally (sz_float.c:7126-7152 / :6790-6812): ~45 cycles of dependent latency per block and coefficient (subtract,
 * divide, add, halve, truncate, convert back, multiply, add), 3.9 ms for the 3 x 10^5 regression blocks of the 512^3 M-field.  Kept as the
 * fall-back of single steps and as what the tests compare the fast form against. */
void szhost_coeff_chain_one_ref(int is_double, void *coef, const unsigned char *indicator, size_t nblocks, int use_mean, int e, szhost_coeffs *out,
                                size_t *progress)
{
    if (is_double) { CHAIN_ONE(double, fabs, 0) }
    else { CHAIN_ONE(float, fabsf, 1) }
}

/* The same chain with most of the arithmetic taken OFF the loop-carried path (round 5).  The interval number of a step,
 * q = (int)((|d| / prec + 1) / 2), is a monotone step function of |d| = |c_k - last_{k-1}|.  Its steps lie at T[n] = the smallest magnitude
 * with q >= n, found once per precision with the reference's own expression (a few evaluations around (2 n - 1) prec each: tables of 32 769
 * entries, cached); and since |c_{k-1} - last_{k-1}| <= prec after every step, |d| lies within prec of |c_k - c_{k-1}|, which is known
 * without the chain: three candidates n0, n0 + 1, n0 + 2 chosen from the ORIGINAL coefficients cover it with a margin.  What remains on the
 * chain is subtract, compares, a select among the candidates' pre-multiplied addends, and the add; the table look-ups of later steps run
 * ahead of it.  A magnitude outside the candidates' range (or anything not finite) takes the reference's expression for that step.  Bit for
 * bit the reference's codes, decoded values and verbatim coefficients (tests/test_host_logic.py compares the two forms on adversarial
 * sequences). */
__thread unsigned long g_chain_generic_steps = 0;   /* (development: steps of THIS thread's chains that took the general form) */
#define COMMA ,
#define CHAIN_TAB_N 32768
typedef struct chain_tab { int is_double, variant; uint64_t prec_bits; void *thr, *pos, *neg; } chain_tab;
#define CHAIN_TAB_SLOTS 16
static chain_tab g_chain_tabs[CHAIN_TAB_SLOTS];
static int g_chain_tab_count = 0;
static pthread_mutex_t g_chain_tab_mu = PTHREAD_MUTEX_INITIALIZER;

#define CHAIN_TAB_BUILD(T, NEXTAFTER, HUGE)                                                                          \
    T *tt = (T *)malloc((CHAIN_TAB_N + 4) * sizeof(T)), *ta = (T *)malloc((CHAIN_TAB_N + 4) * sizeof(T)),            \
      *tn = (T *)malloc((CHAIN_TAB_N + 4) * sizeof(T));                                                               \
    if (!tt || !ta || !tn) { free(tt); free(ta); free(tn); return NULL; }                                             \
    const T prec = (T)precd, rprec = 1 / prec;                                                                        \
    tt[0] = 0;                                                                                                        \
    for (int n = 0; n <= CHAIN_TAB_N + 3; n++) {                                                                      \
        ta[n] = (T)(2 * n) * prec; tn[n] = (T)0 - ta[n];                                                              \
        if (n == 0) continue;                                                                                         \
        if (n > CHAIN_TAB_N) { tt[n] = HUGE; continue; }                                                              \
        const T want = (T)(2 * n);                                                                                    \
        T x = (T)(2 * n - 1) * prec;                                                                                  \
        int guard = 0;                                                                                                \
        while (x > 0 && (variant ? x * rprec + 1 : x / prec + 1) >= want && guard++ < 4096) x = NEXTAFTER(x, (T)0);   \
        while ((variant ? x * rprec + 1 : x / prec + 1) < want && guard++ < 8192) x = NEXTAFTER(x, HUGE);             \
        if (guard >= 4096) { free(tt); free(ta); free(tn); return NULL; }                                             \
        tt[n] = x;                                                                                                    \
    }                                                                                                                 \
    t.thr = tt; t.pos = ta; t.neg = tn;

/* `*owned` = 1: the table is the caller's (the cache is full: entries are never evicted, a chain may still be reading them), to be released with
 * chain_tab_release after the chain.  `steps`: the length of the chain the table is wanted for -- building one costs ~1.7 ms, which a short chain
 * never earns back (3.6 against 11.7 ns a step): below CHAIN_TAB_MIN_STEPS a table is used only if it is there already */
#define CHAIN_TAB_MIN_STEPS 150000
static void chain_tab_release(const chain_tab *t) { if (t) { free(t->thr); free(t->pos); free(t->neg); free((void *)t); } }
static const chain_tab *chain_tab_get(int is_double, int variant, double precd, size_t steps, int *owned)
{
    *owned = 0;
    uint64_t bits; memcpy(&bits, &precd, 8);
    if (!(precd > 0) || !isfinite(precd) || !isfinite(precd * 131080.0) || (is_double ? precd < 1e-290 : precd < 1e-30)) return NULL;
    pthread_mutex_lock(&g_chain_tab_mu);
    for (int i = 0; i < g_chain_tab_count; i++)
        if (g_chain_tabs[i].is_double == is_double && g_chain_tabs[i].variant == variant && g_chain_tabs[i].prec_bits == bits) {
            pthread_mutex_unlock(&g_chain_tab_mu);
            return &g_chain_tabs[i];
        }
    pthread_mutex_unlock(&g_chain_tab_mu);
    { const char *mn = getenv("SZ_HIP_CHAIN_TAB_MIN"); if (steps < (mn ? (size_t)atol(mn) : (size_t)CHAIN_TAB_MIN_STEPS)) return NULL; }      /* (tests set 0: the fast form on short chains) */
    chain_tab t; t.is_double = is_double; t.variant = variant; t.prec_bits = bits; t.thr = t.pos = t.neg = NULL;
    if (is_double) { CHAIN_TAB_BUILD(double, nextafter, HUGE_VAL) }
    else { CHAIN_TAB_BUILD(float, nextafterf, HUGE_VALF) }
    pthread_mutex_lock(&g_chain_tab_mu);
    for (int i = 0; i < g_chain_tab_count; i++)                            /* another thread built the same table meanwhile */
        if (g_chain_tabs[i].is_double == is_double && g_chain_tabs[i].variant == variant && g_chain_tabs[i].prec_bits == bits) {
            pthread_mutex_unlock(&g_chain_tab_mu);
            free(t.thr); free(t.pos); free(t.neg);
            return &g_chain_tabs[i];
        }
    if (g_chain_tab_count == CHAIN_TAB_SLOTS) {                            /* (a full cache stops caching: the table just built serves this chain and goes) */
        pthread_mutex_unlock(&g_chain_tab_mu);
        chain_tab *mine = (chain_tab *)malloc(sizeof(chain_tab));
        if (!mine) { free(t.thr); free(t.pos); free(t.neg); return NULL; }
        *mine = t; *owned = 1;
        return mine;
    }
    g_chain_tabs[g_chain_tab_count] = t;
    const chain_tab *r = &g_chain_tabs[g_chain_tab_count++];
    pthread_mutex_unlock(&g_chain_tab_mu);
    return r;
}


/* The lean form of a step (most steps), in loops of their own (their register allocation is not the general loop's): when |c_k - c_{k-1}|
 * exceeds the precision by a margin, the SIGN of c_k - last is the sign of c_k - c_{k-1}, and two candidates n, n + 1 with ONE threshold between
 * them cover |c_k - last| (its range is one step of the interval number wide): what is left on the chain is subtract, abs, ONE compare whose
 * result is the select mask, and the add -- all in the vector registers (SSE2 scalar forms).  n is the interval number of
 * |c_k - c_{k-1}| - prec, by the reference's own expression (a divide, but off the chain).  Everything the selection assumes is checked by
 * predictable branches; a step that fails a check ends the lean loop, which returns how far it came: the general form takes that step. */
#if defined(__SSE2__)
#include <emmintrin.h>
#define CHAIN_LEAN_FN(NAME, T, V, LOADS, GETS, SUBS, ADDS, ANDV, ANDNV, ORV, CMPLES, COMILT, COMIGE, COMIGT, MOVMSK, ABSMASK, FABS)     \
static size_t NAME(const T *cf, size_t b, size_t nblocks, const unsigned char *indicator, T *plast, T *pprevc, T *cfo, int *codes,     \
                   size_t *pci, const T *tt, const T *ta, const T *tn, T prec, int variant, size_t *progress, T *un, size_t *pnun)     \
{                                                                                                                                  \
    const T rprec = 1 / prec, lim = prec * (T)1.5, pm = prec * (T)1.0001, nlimit = (T)(2 * (CHAIN_TAB_N - 4)) * prec;                   \
    const T nbeyond = (T)(2 * (CHAIN_TAB_N + 4)) * prec;                                                                           \
    const V absmask = ABSMASK, vprec = LOADS(&prec);                                                                               \
    V vlast = LOADS(plast);                                                                                                        \
    T prevc = *pprevc;                                                                                                             \
    size_t ci = *pci;                                                                                                              \
    CHAIN_AVAIL_LOCALS                                                                                                             \
    for (; b < nblocks; b++) {                                                                                                     \
        if (indicator[b]) continue;                                                                                                \
        CHAIN_WAIT_AVAIL(b);                                                                                                       \
        const T cur = cf[b], Dk = cur - prevc, aD = FABS(Dk);                                                                      \
        if (__builtin_expect(!(aD > lim && aD < nlimit), 0)) {                                                                     \
            const V vcur0 = LOADS(&cf[b]);                                                                                         \
            if (aD >= nbeyond && aD <= (T)3.0e38) {                                                                                \
                /* certainly beyond the code range whatever `last` is (|c_k - last| >= |c_k - c_{k-1}| - prec): the coefficient     \
                 * verbatim (sz_float.c:7150) -- and the chain starts afresh from it */                                             \
                const V vd0 = SUBS(vcur0, vlast);                                                                                  \
                if (!COMIGE(ANDV(vd0, absmask), LOADS(&tt[CHAIN_TAB_N]))) break;          /* (checked all the same) */              \
                vlast = vcur0; prevc = cur; codes[ci++] = 0; cfo[b] = cur; un[(*pnun)++] = cur;                                    \
                if (progress && (ci & 1023) == 0) __atomic_store_n(progress, ci, __ATOMIC_RELEASE);                               \
                continue;                                                                                                          \
            }                                                                                                                      \
            if (aD <= lim) {                                                                                                       \
                /* the coefficient barely moved: |c_k - last| < 2.5 prec, the interval number is 0 or 1 -- one threshold, the       \
                 * addend's sign from the difference itself (still no divide on the chain) */                                       \
                const V vd0 = SUBS(vcur0, vlast), vad0 = ANDV(vd0, absmask);                                                       \
                const V m0 = CMPLES(LOADS(&tt[1]), vad0), mn0 = CMPLES(vad0, vad0);       /* (mn0: all ones unless NaN) */          \
                if (!(MOVMSK(mn0) & 1) || !COMILT(vad0, LOADS(&tt[2]))) break;                                                     \
                const V sgn0 = ANDNV(absmask, vd0);                                        /* the sign bit of the difference */      \
                const V add0 = ANDV(m0, ORV(LOADS(&ta[1]), sgn0));                         /* +-A[1] or +0 (q = 0 adds (T)0, sz_float.c:7140) */ \
                const V vnew0 = ADDS(vlast, add0);                                                                                 \
                prevc = cur;                                                                                                       \
                if (__builtin_expect(COMIGT(ANDV(SUBS(vcur0, vnew0), absmask), vprec), 0)) {                                       \
                    vlast = vcur0; codes[ci++] = 0; cfo[b] = cur; un[(*pnun)++] = cur;                                             \
                } else {                                                                                                           \
                    const int q0 = MOVMSK(m0) & 1, ng0 = MOVMSK(vd0) & 1;                                                          \
                    vlast = vnew0; codes[ci++] = (ng0 ? -q0 : q0) + 32768; cfo[b] = GETS(vnew0);                                   \
                }                                                                                                                  \
                if (progress && (ci & 1023) == 0) __atomic_store_n(progress, ci, __ATOMIC_RELEASE);                               \
                continue;                                                                                                          \
            }                                                                                                                      \
            break;                                                     /* (NaN, infinity, the last intervals of the range) */      \
        }                                                                                                                          \
        const T lo1 = aD - pm;                                                                                                     \
        const int n1 = (int)(((variant ? lo1 * rprec : lo1 / prec) + 1) * (T)0.5);                                                 \
        const int neg = Dk < 0;                                                                                                    \
        const T *sg = neg ? tn : ta;                                                                                               \
        const V vt = LOADS(&tt[n1 + 1]), vcur = LOADS(&cf[b]);                                                                     \
        const V vd = SUBS(vcur, vlast), vad = ANDV(vd, absmask);                                                                   \
        const V m = CMPLES(vt, vad);                                                                                               \
        const V vnew = ADDS(vlast, ORV(ANDV(m, LOADS(&sg[n1 + 1])), ANDNV(m, LOADS(&sg[n1]))));                                    \
        /* what the selection assumed: the magnitude within the two candidates' range, the sign as predicted, the bound kept */     \
        if (!(COMIGE(vad, LOADS(&tt[n1])) && COMILT(vad, LOADS(&tt[n1 + 2])))) break;                                              \
        if (neg != (MOVMSK(vd) & 1)) break;                                                                                        \
        prevc = cur;                                                                                                               \
        if (__builtin_expect(COMIGT(ANDV(SUBS(vcur, vnew), absmask), vprec), 0)) {   /* the bound does not hold: the coefficient verbatim (sz_float.c:7143) */ \
            vlast = vcur; codes[ci++] = 0; cfo[b] = cur; un[(*pnun)++] = cur;                                                      \
        } else {                                                                                                                   \
            const int q = n1 + (MOVMSK(m) & 1);                                                                                    \
            vlast = vnew;                                                                                                          \
            codes[ci++] = (neg ? -q : q) + 32768;                                                                                  \
            cfo[b] = GETS(vnew);                                                                                                   \
        }                                                                                                                          \
        if (progress && (ci & 1023) == 0) __atomic_store_n(progress, ci, __ATOMIC_RELEASE);                                       \
    }                                                                                                                              \
    *plast = GETS(vlast); *pprevc = prevc; *pci = ci;                                                                              \
    return b;                                                                                                                      \
}
CHAIN_LEAN_FN(chain_lean_f32, float, __m128, _mm_load_ss, _mm_cvtss_f32, _mm_sub_ss, _mm_add_ss, _mm_and_ps, _mm_andnot_ps, _mm_or_ps, _mm_cmple_ss,
              _mm_comilt_ss, _mm_comige_ss, _mm_comigt_ss, _mm_movemask_ps, _mm_castsi128_ps(_mm_set_epi32(0, 0, 0, 0x7fffffff)), fabsf)
CHAIN_LEAN_FN(chain_lean_f64, double, __m128d, _mm_load_sd, _mm_cvtsd_f64, _mm_sub_sd, _mm_add_sd, _mm_and_pd, _mm_andnot_pd, _mm_or_pd, _mm_cmple_sd,
              _mm_comilt_sd, _mm_comige_sd, _mm_comigt_sd, _mm_movemask_pd, _mm_castsi128_pd(_mm_set_epi64x(0, 0x7fffffffffffffffll)), fabs)
/* (data whose steps are mostly NOT lean -- coefficients that differ by about a precision -- would pay for a failed attempt at every step: after 8 in
 *  a row that got nowhere the next 64 steps go straight to the general form) */
#define CHAIN_LEAN_CALL(FN) { if (lean_skip > 0 || ref_steps > 0) { if (lean_skip > 0) --lean_skip; } else { const size_t b_in = b;                                      \
        b = FN(cf, b, nblocks, indicator, &last, &prevc, cf, codes, &ci, tt, ta, tn, prec, variant, progress, un, &nun);                    \
        if (b == b_in) { if (++lean_miss >= 8) { lean_miss = 0; lean_skip = 64; } } else lean_miss = 0;                           \
        if (b >= nblocks) break; if (indicator[b]) continue; } }
#define CHAIN_LEAN_F32 CHAIN_LEAN_CALL(chain_lean_f32)
#define CHAIN_LEAN_F64 CHAIN_LEAN_CALL(chain_lean_f64)
#else
#define CHAIN_LEAN_F32
#define CHAIN_LEAN_F64
#endif

#define CHAIN_FAST(T, U, FABS, CHAIN_LEAN_TRY, VDECL)                                                                                       \
    T *cf = (T *)coef + (size_t)e * nblocks;                                                                           \
    const T prec = (T)out->prec[e], rprec = 1 / prec, prec2 = prec + prec, nlimit = (T)(2 * (CHAIN_TAB_N - 4));        \
    const T *tt = (const T *)tab->thr, *ta = (const T *)tab->pos, *tn = (const T *)tab->neg;                           \
    T last = 0, prevc = 0;                                                                                             \
    T *un = (T *)out->unpred[e];                                                                                       \
    int *codes = out->codes[e];                                                                                        \
    size_t ci = 0, nun = 0;                                                                                            \
    int lean_miss = 0, lean_skip = 0; (void)lean_miss; (void)lean_skip;                                               \
    int ref_steps = 0, odd = 0; size_t next_eval = 1024;   /* data on which the candidates rarely hold (verbatim coefficients all over): the plain loop for a while */ \
    CHAIN_AVAIL_LOCALS                                                                                                 \
    for (size_t b = 0; b < nblocks; b++) {                                                                             \
        if (indicator[b]) continue;                                                                                    \
        CHAIN_WAIT_AVAIL(b);                                                                                           \
        CHAIN_LEAN_TRY                                             /* as many lean steps as come in a row; then this one in the general form */ \
        const T cur = cf[b];                                                                                           \
        /* off the chain: the candidates from the original coefficients */                                             \
        const T Dk = cur - prevc, aD = FABS(Dk);                                                                       \
        const T lo = aD - prec2, ql = lo * rprec;                                                                      \
        prevc = cur;                                                                                                   \
        int cc = 0;                                                                                                    \
        if (ref_steps > 0) --ref_steps;                                                                                \
        if (ci >= next_eval) { if (odd > 128) ref_steps = 4096; odd = 0; next_eval = ci + 1024; }   /* (of ALL steps, the lean ones included) */ \
        ++g_chain_generic_steps;                                                                                       \
        const int fast = ql < nlimit && ref_steps == 0;          /* (false for NaN) */                                 \
        const T qc = fast ? (ql > 0 ? ql : (T)0) : (T)0;         /* (selects, not branches: the data decide them) */   \
        int n0 = (int)((qc + 1) * (T)0.5) - 1;                                                                         \
        n0 = n0 < 0 ? 0 : n0;                                                                                          \
        const T t0 = tt[n0], t1 = tt[n0 + 1], t2 = tt[n0 + 2], t3 = tt[n0 + 3];                                        \
        U p0, p1, p2, g0, g1, g2;                                                                                      \
        memcpy(&p0, &ta[n0], sizeof(T)); memcpy(&p1, &ta[n0 + 1], sizeof(T)); memcpy(&p2, &ta[n0 + 2], sizeof(T));     \
        memcpy(&g0, &tn[n0], sizeof(T)); memcpy(&g1, &tn[n0 + 1], sizeof(T)); memcpy(&g2, &tn[n0 + 2], sizeof(T));     \
        /* on the chain */                                                                                             \
        const T diff = cur - last, ad = FABS(diff);                                                                    \
        if (__builtin_expect(fast && ad >= t0 && ad < t3, 1)) {                                                        \
            const int s1 = ad >= t1, s2 = ad >= t2, neg = diff < 0;                                                    \
            const U m1 = (U)0 - (U)s1, m2 = (U)0 - (U)s2, mn = (U)0 - (U)neg;                                          \
            const U pos = (p2 & m2) | (((p1 & m1) | (p0 & ~m1)) & ~m2), ngv = (g2 & m2) | (((g1 & m1) | (g0 & ~m1)) & ~m2); \
            const U sel = (ngv & mn) | (pos & ~mn);                                                                    \
            T add; memcpy(&add, &sel, sizeof(T));                                                                      \
            const int q = n0 + s1 + s2;                                                                                \
            cc = (neg ? -q : q) + 32768;                                                                               \
            last = last + add;                                                                                         \
            if (__builtin_expect(FABS(cur - last) > prec, 0)) { cc = 0; last = cur; un[nun++] = cur; ++odd; }          \
        } else {                                                   /* the reference's expression for this step */      \
            ++odd;                                                                                                     \
            T itv;                                                                                                     \
            if (!variant) itv = FABS(diff) / prec + 1; else itv = FABS(diff) * rprec + 1;                              \
            if (itv < 65536) {                                                                                         \
                if (diff < 0) itv = -itv;                                                                              \
                cc = (int)(itv / 2) + 32768;                                                                           \
                last = last + 2 * (cc - 32768) * prec;                                                                 \
                if (FABS(cur - last) > prec) { cc = 0; last = cur; un[nun++] = cur; }                                  \
            } else { cc = 0; last = cur; un[nun++] = cur; }                                                            \
        }                                                                                                              \
        codes[ci++] = cc;                                                                                              \
        cf[b] = last;                                                                                                  \
        if (progress && (ci & 1023) == 0) __atomic_store_n(progress, ci, __ATOMIC_RELEASE);                           \
    }                                                                                                                  \
    out->unpred_count[e] = nun;                                                                                        \
    if (progress) __atomic_store_n(progress, ci, __ATOMIC_RELEASE);

void szhost_coeff_chain_one_p(int is_double, void *coef, const unsigned char *indicator, size_t nblocks, int use_mean, int e, szhost_coeffs *out,
                              size_t *progress)
{
    const int variant = (is_double || use_mean) ? 1 : 0;          /* 1: |diff| * (1 / prec), 0: |diff| / prec (sz_float.c:7133 against :6795, sz_double.c) */
    const char *sw = getenv("SZ_HIP_CHAIN_FAST");
    int owned = 0;
    size_t steps = 0;
    for (size_t b = 0; b < nblocks; b++) steps += indicator[b] == 0;      /* the chain's length: the regression blocks */
    const chain_tab *tab = (sw && sw[0] == '0') ? NULL : chain_tab_get(is_double, variant, out->prec[e], steps, &owned);
    if (!tab) { szhost_coeff_chain_one_ref(is_double, coef, indicator, nblocks, use_mean, e, out, progress); return; }
    if (is_double) { CHAIN_FAST(double, uint64_t, fabs, CHAIN_LEAN_F64, ) }
    else { CHAIN_FAST(float, uint32_t, fabsf, CHAIN_LEAN_F32, ) }
    if (owned) chain_tab_release(tab);
}
void szhost_coeff_chain_one_pa(int is_double, void *coef, const unsigned char *indicator, size_t nblocks, int use_mean, int e, szhost_coeffs *out,
                               size_t *progress, const size_t *avail)
{
    g_chain_avail = avail;
    szhost_coeff_chain_one_p(is_double, coef, indicator, nblocks, use_mean, e, out, progress);
    g_chain_avail = NULL;
}
void szhost_coeff_chain_one(int is_double, void *coef, const unsigned char *indicator, size_t nblocks, int use_mean, int e, szhost_coeffs *out)
{
    szhost_coeff_chain_one_p(is_double, coef, indicator, nblocks, use_mean, e, out, NULL);
}

void szhost_coeff_chain(int is_double, void *coef, const unsigned char *indicator, size_t nblocks, double eb,
                        int late0, int late1, int late2, int use_mean, int ncoef, szhost_coeffs *out)
{
    szhost_coeff_chain_begin(is_double, indicator, nblocks, eb, late0, late1, late2, ncoef, out);
    for (int e = 0; e < ncoef; e++) szhost_coeff_chain_one(is_double, coef, indicator, nblocks, use_mean, e, out);
}

void szhost_coeff_unchain(int is_double, void *coef, const unsigned char *indicator, size_t nblocks,
                          int *const codes[4], const int radius[4], const double prec[4],
                          const unsigned char *const unpred[4], int ncoef)
{
    size_t ci = 0, un[4] = {0, 0, 0, 0};
    if (is_double) {
        double *cf = (double *)coef, last[4] = {0, 0, 0, 0};
        for (size_t b = 0; b < nblocks; b++) {
            if (indicator[b]) continue;
            for (int e = 0; e < ncoef; e++) {
                int t = codes[e][ci];
                if (t != 0) last[e] = last[e] + 2 * (t - radius[e]) * prec[e];
                else { memcpy(&last[e], unpred[e] + 8 * (un[e]++), 8); }
                cf[(size_t)e * nblocks + b] = last[e];
            }
            ci++;
        }
    } else {
        float *cf = (float *)coef, last[4] = {0, 0, 0, 0};
        for (size_t b = 0; b < nblocks; b++) {
            if (indicator[b]) continue;
            for (int e = 0; e < ncoef; e++) {
                int t = codes[e][ci];
                if (t != 0) last[e] = last[e] + 2 * (t - radius[e]) * (float)prec[e];
                else { memcpy(&last[e], unpred[e] + 4 * (un[e]++), 4); }
                cf[(size_t)e * nblocks + b] = last[e];
            }
            ci++;
        }
    }
}

/* ------------------------------------------------------------------ stream framing */

size_t szhost_write_meta(const szhost_meta *m, unsigned char flags, unsigned char *out)
{
    size_t plen = m->data_type == 0 ? 28 : 36;
    memset(out, 0, 4 + plen);
    out[0] = 2; out[1] = 1; out[2] = 12; /* versionNumber, sz/include/defines.h:13-17 */
    out[3] = flags;
    unsigned char *r = out + 4;
    unsigned char buf = (unsigned char)(m->opt_quant_mode & 1);
    buf = (unsigned char)((buf << 1) | (m->data_endian & 1));
    buf = (unsigned char)((buf << 1) | 0); /* little-endian system */
    buf = (unsigned char)((buf << 2) | (m->sz_mode & 3));
    int tmp = 0;
    if (m->gzip_mode == 1) tmp = 0; else if (m->gzip_mode == 0) tmp = 1; else if (m->gzip_mode == 9) tmp = 2;
    buf = (unsigned char)((buf << 2) | tmp);
    r[0] = buf;
    r[1] = (unsigned char)((unsigned)m->sample_distance >> 8); r[2] = (unsigned char)m->sample_distance;
    short t2 = (short)(m->pred_threshold * 10000);
    r[3] = (unsigned char)((unsigned short)t2 >> 8); r[4] = (unsigned char)t2;
    r[5] = (unsigned char)m->err_mode;
    r[5] = (unsigned char)((r[5] << 4) | (m->data_type & 0x17));
    switch (m->err_mode) {
    case 0: szhost_put_f32be(r + 6, (float)m->abs_bound); break;
    case 1: szhost_put_f32be(r + 10, (float)m->rel_ratio); break;
    case 2: case 3: szhost_put_f32be(r + 6, (float)m->abs_bound); szhost_put_f32be(r + 10, (float)m->rel_ratio); break;
    case 4: szhost_put_f32be(r + 6, (float)m->psnr); memset(r + 9, 0, 4); break;
    case 11: case 12: szhost_put_f32be(r + 6, (float)m->abs_bound); szhost_put_f32be(r + 10, (float)m->pwr_ratio); break;   /* ABS_AND/OR_PW_REL (ByteToolkit.c:935-939) */
    case 13: case 14: szhost_put_f32be(r + 6, (float)m->rel_ratio); szhost_put_f32be(r + 10, (float)m->pwr_ratio); break;   /* REL_AND/OR_PW_REL */
    case 10: szhost_put_f32be(r + 10, (float)m->pwr_ratio); break;                                                          /* PW_REL */
    default: break;
    }
    r[14] = (unsigned char)m->sol_id;
    szhost_put_u32be(r + 16, m->opt_quant_mode == 1 ? m->max_quant_intervals : m->quantization_intervals);
    if (m->data_type == 0) { szhost_put_f32be(r + 20, (float)m->vmin); szhost_put_f32be(r + 24, (float)m->vmax); }
    else { szhost_put_f64be(r + 20, m->vmin); szhost_put_f64be(r + 28, m->vmax); }
    return 4 + plen;
}
