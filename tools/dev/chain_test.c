#include "szhost.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
static unsigned long long rs = 88172645463325252ull;
static double rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (rs >> 11) * (1.0 / 9007199254740992.0); }
int run(int is_double, size_t nb, int kind, double eb, int use_mean)
{
    const size_t esz = is_double ? 8 : 4;
    unsigned char *ind = calloc(nb, 1);
    void *c0 = malloc(nb * 4 * esz), *c1 = malloc(nb * 4 * esz);
    for (size_t b = 0; b < nb; b++) if (kind & 1) ind[b] = rnd() < 0.3;
    double walk[4] = {0, 0, 0, 0};
    for (int e = 0; e < 4; e++) for (size_t b = 0; b < nb; b++) {
        double prec = 0.025 * eb / (e < 3 ? 6 : 1), v;
        switch (kind >> 1) {
        case 0: walk[e] += (rnd() - 0.5) * 40 * prec; v = walk[e]; break;                    /* random walk, steps of some intervals */
        case 1: v = (double)((long)(rnd() * 20) - 10) * prec + (rnd() < 0.5 ? 1e-9 * prec : -1e-9 * prec) * (b & 1); walk[e] = v; break;  /* on interval boundaries */
        case 2: v = (rnd() - 0.5) * 1e6 * prec; break;                                      /* huge jumps: raw values */
        case 3: walk[e] += prec * ((b % 7) - 3) * 1.0; v = walk[e] + (rnd() - 0.5) * 2.0 * prec; break;
        case 4: v = (b % 50 == 0) ? (rnd() < 0.5 ? NAN : INFINITY) : (rnd() - 0.5) * 10 * prec; break;
        default: v = 0; break;
        }
        if (is_double) ((double *)c0)[e * nb + b] = v; else ((float *)c0)[e * nb + b] = (float)v;
    }
    memcpy(c1, c0, nb * 4 * esz);
    szhost_coeffs a, r; memset(&a, 0, sizeof a); memset(&r, 0, sizeof r);
    szhost_coeff_chain_begin(is_double, ind, nb, eb, 6, 6, 6, 4, &a);
    szhost_coeff_chain_begin(is_double, ind, nb, eb, 6, 6, 6, 4, &r);
    double t0 = now();
    for (int e = 0; e < 4; e++) szhost_coeff_chain_one_p(is_double, c0, ind, nb, use_mean, e, &a, NULL);
    double t1 = now();
    for (int e = 0; e < 4; e++) szhost_coeff_chain_one_ref(is_double, c1, ind, nb, use_mean, e, &r, NULL);
    double t2 = now();
    int bad = memcmp(c0, c1, nb * 4 * esz) != 0;
    for (int e = 0; e < 4; e++) {
        if (a.unpred_count[e] != r.unpred_count[e]) bad |= 2;
        else if (memcmp(a.unpred[e], r.unpred[e], a.unpred_count[e] * esz)) bad |= 4;
        if (memcmp(a.codes[e], r.codes[e], a.reg_count * sizeof(int))) bad |= 8;
    }
    printf("f%d nb %zu kind %d mean %d: fast %.3f ms ref %.3f ms (4 chains in a row) raw %zu/%zu %s\n", is_double ? 64 : 32, nb, kind, use_mean, t1 - t0, t2 - t1, a.unpred_count[0], a.reg_count, bad ? "DIFFERENT" : "identical");
    szhost_coeffs_free(&a); szhost_coeffs_free(&r); free(ind); free(c0); free(c1);
    return bad;
}
int main2(void);
int main()
{
#ifdef THREADS_MAIN
    return main2();
#endif
    int bad = 0;
    for (int d = 0; d < 2; d++) for (int kind = 0; kind < 10; kind++) for (int um = 0; um < 2; um++) bad |= run(d, 20000 + kind * 77, kind, d ? 1e-3 : 1e-4, um);
    bad |= run(0, 303450, 0, 1e-4, 0); bad |= run(0, 303450, 0, 1e-4, 0); bad |= run(1, 303450, 6, 2.9668932e-3, 0);
    printf(bad ? "FAILED\n" : "all identical\n");
    return bad;
}

/* ---- the four chains on four freshly created threads, as the product runs them (compile with -DTHREADS_MAIN) */
#ifdef THREADS_MAIN
#include <pthread.h>
#include <unistd.h>
typedef struct { int e; void *c; unsigned char *ind; size_t nb; szhost_coeffs *st; int fast; } job_t;
static void *worker(void *p) { job_t *j = (job_t *)p; if (j->fast) szhost_coeff_chain_one_p(0, j->c, j->ind, j->nb, 0, j->e, j->st, NULL); else szhost_coeff_chain_one_ref(0, j->c, j->ind, j->nb, 0, j->e, j->st, NULL); return NULL; }
int main2(void)
{
    const size_t nb = 303450; const double eb = 1e-4;
    unsigned char *ind = calloc(nb, 1);
    float *c0 = malloc(nb * 16);
    double walk[4] = {0, 0, 0, 0};
    for (int e = 0; e < 4; e++) for (size_t b = 0; b < nb; b++) { double prec = 0.025 * eb / (e < 3 ? 6 : 1); walk[e] += (rnd() - 0.5) * 40 * prec; c0[e * nb + b] = (float)walk[e]; }
    printf("online cpus %ld\n", sysconf(_SC_NPROCESSORS_ONLN));
    for (int rep = 0; rep < 8; rep++) {
        const int fast = rep != 3;
        float *c = malloc(nb * 16); memcpy(c, c0, nb * 16);
        szhost_coeffs st; memset(&st, 0, sizeof st);
        szhost_coeff_chain_begin(0, ind, nb, eb, 6, 6, 6, 4, &st);
        usleep(3000);                                    /* the caller was busy elsewhere: cores asleep */
        pthread_t th[4]; job_t j[4];
        double t0 = now();
        for (int e = 0; e < 4; e++) { j[e] = (job_t){e, c, ind, nb, &st, fast}; pthread_create(&th[e], NULL, worker, &j[e]); }
        for (int e = 0; e < 4; e++) pthread_join(th[e], NULL);
        printf("rep %d %s: 4 chains on 4 new threads %.3f ms\n", rep, fast ? "fast" : "ref ", now() - t0);
        szhost_coeffs_free(&st); free(c);
    }
    return 0;
}
#endif
