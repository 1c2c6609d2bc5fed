// micro-benchmark (round 6): what one step of the lane-per-column sweep costs on one wavefront per SIMD, by what is in it.
// A step = DPP from the left lane, the 7-point sum left to right, the quantiser (sz_float.c:7268-7287 in szh_beam.h's form), selects.
//   VAR 0  the dependent chain alone (value from a register)
//   VAR 1  + the value read from an LDS ring a step ahead, code and reconstruction written to it, running address (add, sub, min)
//   VAR 2  VAR 1 with EXTRA independent VALU instructions per step (-DEXTRA=n)
//   VAR 3  VAR 1 with EXTRA independent SALU instructions per step
//   VAR 4  VAR 1 with two wavefronts per SIMD (512 threads)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DVAR=1 -DEXTRA=0 -o ub_step tools/ubench/ub_step.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef VAR
#define VAR 0
#endif
#ifndef EXTRA
#define EXTRA 0
#endif
#ifndef FMA
#define FMA 0
#endif
#ifndef UNR
#define UNR 1        /* lines per loop trip (straight-line code of UNR x 5 steps) */
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define SB __builtin_amdgcn_sched_barrier(0)
#define LDS __attribute__((address_space(3)))
__device__ __forceinline__ float shr1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ unsigned bsel(unsigned mask, unsigned a, unsigned b) { unsigned r; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float bself(unsigned mask, float a, float b) { return __uint_as_float(bsel(mask, __float_as_uint(a), __float_as_uint(b))); }

constexpr int LINE = 5, RS = 45, PITCH = 512;
constexpr int NTHR = VAR == 4 ? 512 : 256;

__global__ __launch_bounds__(NTHR, VAR == 4 ? 2 : 1) void k_step(float *io, int nlines, float eb, float kfv)
{
    __shared__ __attribute__((aligned(16))) unsigned char ring[NTHR / 64][RS * PITCH / (VAR == 4 ? 2 : 1)];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    LDS unsigned char *rg = (LDS unsigned char *)ring[w];
    for (int s = 0; s < RS / (VAR == 4 ? 2 : 1); ++s) *(LDS float *)(rg + s * PITCH + lane * 4) = io[(s * 64 + lane) & 1023];
    float dl[LINE], lup[LINE], prev = 0, Lprev = 0, Bold = 0, Bpold = 0;
    for (int u = 0; u < LINE; ++u) { dl[u] = 0; lup[u] = 0; }
    unsigned mfirst = (lane & 31) == 0 ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(mfirst));
    const float eb2 = eb + eb, rh = 0.5f / eb, caph = 32767.0f, radf = 32768.0f;
    constexpr unsigned RB = (unsigned)(RS / (VAR == 4 ? 2 : 1)) * PITCH;
    unsigned vaddr = (unsigned)(lane * 4);
    float cur_next = VAR >= 1 ? *(volatile LDS float *)(rg + vaddr) : io[lane];
    unsigned dummy[16]; for (int i = 0; i < 16; ++i) dummy[i] = lane + i;
    unsigned sd = (unsigned)nlines;
#pragma unroll 1
    for (int it = 0; it < nlines; it += UNR) {
#pragma unroll
        for (int UU = 0; UU < LINE * UNR; ++UU) {
            const int U = UU % LINE;
            const float cur = cur_next;
            const float Lraw = shr1(prev);
            SB;
            const float L = bself(mfirst, kfv, Lraw);
            unsigned vnext = vaddr;
            if (VAR >= 1) { const unsigned y = vaddr + PITCH, y2 = y - RB; vnext = y < y2 ? y : y2; }
            SB;
            const float B = dl[U], Bp = lup[U], C = Bold, Cp = Bpold;
            const float s1 = L + prev;
            SB;
            const float s2 = s1 + B;
            if (VAR >= 1) cur_next = *(volatile LDS float *)(rg + vnext);
            SB;
            const float s3 = s2 - Lprev;
            SB;
            const float s4 = s3 - Bp;
            SB;
            const float s5 = s4 - C;
            SB;
            const float pred = s5 + Cp;
            SB;
            const float diff = cur - pred;
            SB;
            const float hq0 = __builtin_fabsf(diff) * rh;
            SB;
            const float hq = hq0 + 0.5f;
            SB;
            const float tq = __builtin_truncf(hq);
            const bool inr = hq < caph;
            SB;
            const float ts = __builtin_copysignf(tq, diff);
            SB;
#if FMA
            const float m2 = __builtin_fmaf(ts, eb2, 0.0f);
#else
            const float m1 = ts * eb2;
            SB;
            const float m2 = m1 + 0.0f;
#endif
            const float cf = radf + ts;
            SB;
            int code = (int)cf;
            const float rcn = pred + m2;
            SB;
            const float err = cur - rcn;
            SB;
            const bool ok = inr && !(__builtin_fabsf(err) > eb);
            code = ok ? code : 0;
            const float rec = ok ? rcn : cur;
            if (VAR >= 1) {
                *(volatile LDS unsigned short *)(rg + vaddr + 256) = (unsigned short)code;
                *(volatile LDS float *)(rg + vaddr) = rec;
            } else dummy[0] += (unsigned)code;
            if (VAR == 2) {
#pragma unroll
                for (int e = 0; e < EXTRA; ++e) asm volatile("v_add_u32 %0, %0, %1" : "+v"(dummy[e & 15]) : "v"(dummy[(e + 1) & 15]));
            }
            if (VAR == 3) {
#pragma unroll
                for (int e = 0; e < EXTRA; ++e) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sd));
            }
            dl[U] = rec; lup[U] = L; Bold = B; Bpold = Bp; Lprev = L; prev = rec; vaddr = vnext;
            SB;
        }
    }
    unsigned s = sd; for (int i = 0; i < 16; ++i) s += dummy[i];
    io[threadIdx.x + blockIdx.x * NTHR] = prev + (float)s;
}

int main(int argc, char **argv)
{
    const int nlines = argc > 1 ? atoi(argv[1]) : 4000, nwg = argc > 2 ? atoi(argv[2]) : 256;
    float *d; CK(hipMalloc(&d, 4 << 20));
    std::vector<float> h(1 << 20); for (size_t i = 0; i < h.size(); ++i) h[i] = 0.001f * (float)(i % 977);
    CK(hipMemcpy(d, h.data(), 4 << 20, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_step, dim3(nwg), dim3(NTHR), 0, 0, d, nlines, 1e-4f, 0.25f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("VAR=%d EXTRA=%d FMA=%d UNR=%d: %d lines x 5 steps, %d WGs of %d: %.4f ms = %.1f ns per step\n", VAR, EXTRA, FMA, UNR, nlines, nwg, NTHR, best, best * 1e6 / (nlines * 5.0));
    return 0;
}
