/*
 * szo_huffman.c -- ORACLE (test infrastructure only).
 * CPU restatement of the reference Huffman coder, sz/src/Huffman.c:
 *   histogram + heap tree        init()            Huffman.c:165-185
 *   heap tie-breaking            qinsert/qremove   Huffman.c:76-114
 *   code assignment              build_code()      Huffman.c:122-157
 *   tree (de)serialisation       convert_HuffTree_to_bytes_anyStates / reconstruct_... Huffman.c:443-788
 *   MSB-first bit packing        encode()          Huffman.c:205-308
 *   bit-serial decode            decode()          Huffman.c:310-343
 * Written index-based (no node pointers); codes are limited to 64 bits (the reference
 * keeps 128; >64 needs Fibonacci-like frequencies over more than 1e13 symbols).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "szo.h"

static szo_huff *huff_alloc(int state_num)
{
    szo_huff *h = (szo_huff *)calloc(1, sizeof(*h));
    size_t cap = (size_t)state_num * 4 + 4; /* reference pool: allNodes*2 = 4*stateNum, Huffman.c:22-24 */
    h->state_num = state_num;
    h->root = -1;
    h->freq = (uint64_t *)calloc(cap, sizeof(uint64_t));
    h->left = (int *)malloc(cap * sizeof(int));
    h->right = (int *)malloc(cap * sizeof(int));
    h->sym = (unsigned *)calloc(cap, sizeof(unsigned));
    h->leaf = (unsigned char *)calloc(cap, 1);
    h->code = (uint64_t *)calloc((size_t)state_num * 2 + 2, sizeof(uint64_t));
    h->len = (unsigned char *)calloc((size_t)state_num * 2 + 2, 1);
    h->used = (unsigned char *)calloc((size_t)state_num * 2 + 2, 1);
    for (size_t i = 0; i < cap; i++) h->left[i] = h->right[i] = -1;
    return h;
}

void szo_huff_free(szo_huff *h)
{
    if (!h) return;
    free(h->freq); free(h->left); free(h->right); free(h->sym); free(h->leaf);
    free(h->code); free(h->len); free(h->used); free(h);
}

/* build_code(): left edge = 0, right edge = 1, pre-order; iterative with an explicit stack */
static void assign_codes(szo_huff *h)
{
    if (h->root < 0) return;
    int *stk_node = (int *)malloc((size_t)(h->n_nodes + 1) * sizeof(int));
    uint64_t *stk_bits = (uint64_t *)malloc((size_t)(h->n_nodes + 1) * sizeof(uint64_t));
    int *stk_len = (int *)malloc((size_t)(h->n_nodes + 1) * sizeof(int));
    int sp = 0;
    stk_node[0] = h->root; stk_bits[0] = 0; stk_len[0] = 0; sp = 1;
    while (sp) {
        sp--;
        int n = stk_node[sp]; uint64_t bits = stk_bits[sp]; int len = stk_len[sp];
        if (h->leaf[n]) {
            if (len > 64) { fprintf(stderr, "szo_huffman: code longer than 64 bits\n"); abort(); }
            unsigned c = h->sym[n];
            h->code[c] = len ? (bits << (64 - len)) : 0; /* Huffman.c:128: out1 << (64-len) */
            h->len[c] = (unsigned char)len;
            h->used[c] = 1;
            continue;
        }
        /* push right first so that left is processed first (order is irrelevant for the result) */
        stk_node[sp] = h->right[n]; stk_bits[sp] = (bits << 1) | 1; stk_len[sp] = len + 1; sp++;
        stk_node[sp] = h->left[n];  stk_bits[sp] = (bits << 1);     stk_len[sp] = len + 1; sp++;
    }
    free(stk_node); free(stk_bits); free(stk_len);
}

/* init() from a ready histogram.  nfreq = number of histogram bins that may be non-zero
 * (the reference scans allNodes = 2*stateNum bins, Huffman.c:176). */
szo_huff *szo_huff_from_freq(int state_num, const uint64_t *freq, size_t nfreq)
{
    szo_huff *h = huff_alloc(state_num);
    size_t limit = (size_t)state_num * 2;
    if (nfreq < limit) limit = nfreq;
    /* 1-based binary min-heap of node indices (qq[], Huffman.c:33) */
    int *heap = (int *)malloc(((size_t)state_num * 4 + 4) * sizeof(int));
    int qend = 1;
    for (size_t s = 0; s < limit; s++) {
        if (!freq[s]) continue;
        int n = h->n_nodes++;
        h->freq[n] = freq[s]; h->sym[n] = (unsigned)s; h->leaf[n] = 1;
        /* qinsert: sift up while parent.freq > n.freq (stop on <=, Huffman.c:81) */
        int i = qend++, j;
        while ((j = i >> 1)) {
            if (h->freq[heap[j]] <= h->freq[n]) break;
            heap[i] = heap[j]; i = j;
        }
        heap[i] = n;
    }
    while (qend > 2) {
        int removed[2];
        for (int r = 0; r < 2; r++) {
            /* qremove, Huffman.c:87-114 */
            int top = heap[1];
            qend--;
            heap[1] = heap[qend];
            int i = 1, l;
            while ((l = i << 1) < qend) {
                if (l + 1 < qend && h->freq[heap[l + 1]] < h->freq[heap[l]]) l++;
                if (h->freq[heap[i]] > h->freq[heap[l]]) {
                    int t = heap[i]; heap[i] = heap[l]; heap[l] = t; i = l;
                } else break;
            }
            removed[r] = top;
        }
        /* new_node(0,0, qremove(), qremove()) with gcc's right-to-left argument evaluation:
         * the FIRST removed node is `b` (right, bit 1), the second is `a` (left, bit 0).
         * Huffman.c:181; SURVEY Appendix B probe. */
        int n = h->n_nodes++;
        h->right[n] = removed[0];
        h->left[n] = removed[1];
        h->freq[n] = h->freq[removed[0]] + h->freq[removed[1]];
        h->leaf[n] = 0;
        int i = qend++, j;
        while ((j = i >> 1)) {
            if (h->freq[heap[j]] <= h->freq[n]) break;
            heap[i] = heap[j]; i = j;
        }
        heap[i] = n;
    }
    h->root = (qend > 1) ? heap[1] : -1;
    free(heap);
    assign_codes(h);
    return h;
}

szo_huff *szo_huff_from_symbols(int state_num, const int *s, size_t n)
{
    size_t bins = (size_t)state_num * 2;
    uint64_t *freq = (uint64_t *)calloc(bins, sizeof(uint64_t));
    for (size_t i = 0; i < n; i++) freq[s[i]]++;
    szo_huff *h = szo_huff_from_freq(state_num, freq, bins);
    free(freq);
    return h;
}

size_t szo_huff_node_count(const szo_huff *h)
{
    /* sz_float.c:7385-7387: count symbols with a code, then 2*count-1 */
    size_t cnt = 0;
    for (int i = 0; i < h->state_num; i++) cnt += h->used[i];
    return cnt * 2 - 1;
}

/* pad_tree_*(): pre-order numbering, left subtree first (Huffman.c:443-501) */
size_t szo_huff_tree_to_bytes(const szo_huff *h, unsigned char **out)
{
    size_t nc = szo_huff_node_count(h);
    uint32_t *L = (uint32_t *)calloc(nc, 4), *R = (uint32_t *)calloc(nc, 4), *C = (uint32_t *)calloc(nc, 4);
    unsigned char *t = (unsigned char *)calloc(nc, 1);
    /* iterative pre-order: stack of (node, slot-to-fill) */
    int *stk = (int *)malloc((nc + 1) * sizeof(int));
    int *stk_parent = (int *)malloc((nc + 1) * sizeof(int));
    int *stk_isright = (int *)malloc((nc + 1) * sizeof(int));
    int sp = 0; uint32_t next = 0;
    stk[sp] = h->root; stk_parent[sp] = -1; stk_isright[sp] = 0; sp++;
    while (sp) {
        sp--;
        int n = stk[sp]; int par = stk_parent[sp]; int isr = stk_isright[sp];
        uint32_t idx = next++;
        if (par >= 0) { if (isr) R[par] = idx; else L[par] = idx; }
        C[idx] = h->sym[n]; t[idx] = h->leaf[n];
        if (!h->leaf[n]) {
            stk[sp] = h->right[n]; stk_parent[sp] = (int)idx; stk_isright[sp] = 1; sp++;
            stk[sp] = h->left[n];  stk_parent[sp] = (int)idx; stk_isright[sp] = 0; sp++;
        }
    }
    free(stk); free(stk_parent); free(stk_isright);
    size_t w = nc <= 256 ? 1 : (nc <= 65536 ? 2 : 4);
    size_t total = 1 + 2 * w * nc + 4 * nc + nc;
    unsigned char *b = (unsigned char *)malloc(total);
    b[0] = 0; /* sysEndianType: LITTLE_ENDIAN_SYSTEM (Huffman.c:520) */
    unsigned char *p = b + 1;
    for (int pass = 0; pass < 2; pass++) {
        uint32_t *src = pass ? R : L;
        for (size_t i = 0; i < nc; i++) {
            if (w == 1) { *p = (unsigned char)src[i]; }
            else if (w == 2) { uint16_t v = (uint16_t)src[i]; memcpy(p, &v, 2); }
            else { memcpy(p, &src[i], 4); }
            p += w;
        }
    }
    memcpy(p, C, 4 * nc); p += 4 * nc;
    memcpy(p, t, nc);
    free(L); free(R); free(C); free(t);
    *out = b;
    return total;
}

/* reconstruct_HuffTree_from_bytes_anyStates (little-endian host only) */
szo_huff *szo_huff_tree_from_bytes(int state_num, const unsigned char *bytes, int node_count)
{
    szo_huff *h = huff_alloc(state_num > node_count ? state_num : node_count);
    h->state_num = state_num;
    size_t nc = (size_t)node_count;
    size_t w = nc <= 256 ? 1 : (nc <= 65536 ? 2 : 4);
    const unsigned char *pL = bytes + 1, *pR = pL + w * nc, *pC = pR + w * nc, *pt = pC + 4 * nc;
    /* the serialised arrays are already an index-based tree: node i has children L[i], R[i] (0 = none) */
    for (size_t i = 0; i < nc; i++) {
        uint32_t l = 0, r = 0, c;
        if (w == 1) { l = pL[i]; r = pR[i]; }
        else if (w == 2) { uint16_t a, b2; memcpy(&a, pL + 2 * i, 2); memcpy(&b2, pR + 2 * i, 2); l = a; r = b2; }
        else { memcpy(&l, pL + 4 * i, 4); memcpy(&r, pR + 4 * i, 4); }
        memcpy(&c, pC + 4 * i, 4);
        h->sym[i] = c; h->leaf[i] = pt[i];
        h->left[i] = l ? (int)l : -1;
        h->right[i] = r ? (int)r : -1;
    }
    h->n_nodes = node_count;
    h->root = 0;
    /* quirk kept: for nodeCount>256 the reference creates the root with t=0 regardless (Huffman.c:740,780) */
    if (nc > 256) h->leaf[0] = 0;
    return h;
}

/* encode(): MSB-first concatenation; returns ceil(total_bits/8) (Huffman.c:205-308) */
size_t szo_huff_encode(const szo_huff *h, const int *s, size_t n, unsigned char *out)
{
    uint64_t acc = 0; int nacc = 0; size_t o = 0;
    for (size_t i = 0; i < n; i++) {
        int st = s[i];
        int len = h->len[st];
        uint64_t code = len ? (h->code[st] >> (64 - len)) : 0;
        while (len > 0) {
            int take = 64 - nacc; if (take > len) take = len;
            uint64_t part = (take == 64) ? code : ((code >> (len - take)) & ((1ULL << take) - 1));
            acc = (take == 64) ? part : ((acc << take) | part);
            nacc += take; len -= take;
            if (nacc == 64) {
                for (int b = 7; b >= 0; b--) out[o++] = (unsigned char)(acc >> (8 * b));
                acc = 0; nacc = 0;
            }
        }
    }
    int rem = nacc;
    while (rem > 0) {
        int take = rem >= 8 ? 8 : rem;
        unsigned char byte = (unsigned char)((acc >> (rem - take)) & ((1u << take) - 1));
        if (take < 8) byte = (unsigned char)(byte << (8 - take));
        out[o++] = byte; rem -= take;
    }
    return o;
}

void szo_huff_decode(const szo_huff *h, const unsigned char *in, size_t n, int *out)
{
    int root = h->root;
    if (h->leaf[root]) { for (size_t c = 0; c < n; c++) out[c] = (int)h->sym[root]; return; }
    size_t bit = 0, count = 0; int node = root;
    while (count < n) {
        int b = (in[bit >> 3] >> (7 - (bit & 7))) & 1; bit++;
        node = b ? h->right[node] : h->left[node];
        if (h->leaf[node]) { out[count++] = (int)h->sym[node]; node = root; }
    }
}
