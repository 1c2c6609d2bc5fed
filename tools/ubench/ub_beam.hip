// micro-benchmark of the beam sweep (szh_beam.h) on its own: one launch over an r0 x r1 x r2 float array, timed with events, no host library around it.
// Variants are compiled in with -D switches of szh_beam.h's step (development only: results wrong): what each part of a step costs.
// build (on the GPU box): hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I sz_amd/csrc [-D...] -o ub_beam tools/ubench/ub_beam.hip
#include "szhip_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main(int argc, char **argv)
{
    const int r0 = argc > 1 ? atoi(argv[1]) : 512, r1 = argc > 2 ? atoi(argv[2]) : 32, r2 = argc > 3 ? atoi(argv[3]) : 32;
    const szh_geom3 G = szh_make_geom3(r0, r1, r2);
    std::vector<float> h((size_t)G.n);
    for (int i = 0; i < r0; ++i) for (int j = 0; j < r1; ++j) for (int k = 0; k < r2; ++k)
        h[((size_t)i * r1 + j) * r2 + k] = sinf(k * 0.098f) * cosf(j * 0.065f) * sinf(i * 0.049f) + 0.5f * sinf((k + 2 * j + 3 * i) * 0.0245f);
    float *d; uint16_t *codes; szh_u64 *fk, *fj; unsigned *small;
    CK(hipMalloc(&d, G.n * 4)); CK(hipMalloc(&codes, G.n * 2 + 64)); CK(hipMalloc(&small, 256));
    CK(hipMalloc(&fk, szh_bm::kface_words<float>(G) * 8 + 64)); CK(hipMalloc(&fj, szh_bm::jface_words<float>(G) * 8 + 64));
    CK(hipMemset(fk, 0, szh_bm::kface_words<float>(G) * 8)); CK(hipMemset(fj, 0, szh_bm::jface_words<float>(G) * 8)); CK(hipMemset(small, 0, 256));
    CK(hipMemcpy(d, h.data(), G.n * 4, hipMemcpyHostToDevice));
    szh_qargs<float> a; memset(&a, 0, sizeof(a));
    const szh_bm::grid_t g = szh_bm::make_grid(G);
    a.G = G; a.data = d; a.codes = codes; a.eb = 1e-4f; a.recip = 1 / a.eb; a.cap = 32; a.radius = 16; a.faceI = fk; a.faceJ = fj; a.nI = g.nKB; a.nJ = g.nJG;
    a.ticket = small; a.err = small + 8; a.ticket_mode = 1;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        a.epoch = (unsigned)(it + 1);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_beam<float, false, false, false>), dim3(g.nKB * g.nJG), dim3(szh_bm::WPG * 64), 0, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    unsigned err; CK(hipMemcpy(&err, small + 8, 4, hipMemcpyDeviceToHost));
    std::vector<uint16_t> hc((size_t)G.n); CK(hipMemcpy(hc.data(), codes, G.n * 2, hipMemcpyDeviceToHost));
    unsigned long long sum = 0; for (auto c : hc) sum += c;
    printf("%dx%dx%d: %.4f ms, %.1f ns per wave step (%d steps), err flag %u, code sum %llu\n", r0, r1, r2, best, best * 1e6 / (5.0 * (r0 + 8)), 5 * (r0 + 8), err, sum);
    return 0;
}
