// micro-benchmark (round 3): cost of one workgroup barrier per step with 8..12 wavefronts, next to a dependent VALU chain;
// and of an LDS write by 64 lanes to ONE address (the step counter) against a write by lane 0 only
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ inline unsigned long long clk() { return __builtin_readcyclecounter(); }

template <int WORK, int BAR>
__global__ void k_bar(float *out, unsigned long long *cyc, int iters)
{
    __shared__ float buf[1024];
    float a = out[threadIdx.x];
    const float c = out[1000];
    const int w = threadIdx.x >> 6;
    unsigned long long t0 = clk();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < WORK; ++r) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
        if (BAR == 1) { buf[threadIdx.x] = a; __syncthreads(); a += buf[(threadIdx.x + 64) & 1023]; }
        if (BAR == 2) { buf[threadIdx.x] = a; __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); a += buf[(threadIdx.x + 64) & 1023]; }
    }
    unsigned long long t1 = clk();
    out[threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0) cyc[w] = t1 - t0;
}
template <int MODE>
__global__ void k_ctr(float *out, unsigned long long *cyc, int iters)
{
    __shared__ unsigned ctr[256];
    const int lane = threadIdx.x & 63;
    unsigned long long t0 = clk();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == 0) ((volatile unsigned *)ctr)[0] = it;                       // all lanes, one address
            if (MODE == 1) { if (lane == 0) ((volatile unsigned *)ctr)[0] = it; }    // lane 0 only
            if (MODE == 2) ((volatile unsigned *)ctr)[lane == 0 ? 0 : 64 + lane] = it; // lane 0 the word, the others their own trash word
        }
    }
    unsigned long long t1 = clk();
    out[threadIdx.x] = (float)ctr[lane];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    float *d; unsigned long long *c;
    CK(hipMalloc(&d, 4096 * 4)); CK(hipMalloc(&c, 64 * 8));
    CK(hipMemset(d, 0, 4096 * 4));
    const int iters = 4000;
    unsigned long long h[16];
    auto rep = [&](const char *name) {
        CK(hipDeviceSynchronize()); CK(hipMemcpy(h, c, 16 * 8, hipMemcpyDeviceToHost));
        printf("%-60s wave0 %7.1f  wave7 %7.1f cycles per iteration\n", name, (double)h[0] / iters, (double)h[7] / iters);
    };
    for (int pass = 0; pass < 2; ++pass) {
        k_bar<0, 1><<<1, 512>>>(d, c, iters); rep("8 waves: LDS write + __syncthreads + LDS read, no work");
        k_bar<0, 1><<<1, 768>>>(d, c, iters); rep("12 waves: same");
        k_bar<24, 0><<<1, 768>>>(d, c, iters); rep("12 waves: 24 dependent adds, no barrier");
        k_bar<24, 1><<<1, 768>>>(d, c, iters); rep("12 waves: 24 dependent adds + write/__syncthreads/read");
        k_bar<24, 2><<<1, 768>>>(d, c, iters); rep("12 waves: 24 dependent adds + write/lgkmcnt/s_barrier/read");
        k_bar<24, 2><<<1, 512>>>(d, c, iters); rep("8 waves: same");
        k_bar<24, 2><<<256, 768>>>(d, c, iters); rep("12 waves x 256 blocks: same");
    }
    const char *nm[3] = {"ds_write by 64 lanes to one address", "ds_write by lane 0 (exec mask)", "ds_write: lane 0 the word, other lanes own trash words"};
    k_ctr<0><<<1, 64>>>(d, c, iters); CK(hipDeviceSynchronize()); CK(hipMemcpy(h, c, 8, hipMemcpyDeviceToHost)); printf("%-60s %6.1f cycles per write\n", nm[0], (double)h[0] / iters / 16);
    k_ctr<1><<<1, 64>>>(d, c, iters); CK(hipDeviceSynchronize()); CK(hipMemcpy(h, c, 8, hipMemcpyDeviceToHost)); printf("%-60s %6.1f cycles per write\n", nm[1], (double)h[0] / iters / 16);
    k_ctr<2><<<1, 64>>>(d, c, iters); CK(hipDeviceSynchronize()); CK(hipMemcpy(h, c, 8, hipMemcpyDeviceToHost)); printf("%-60s %6.1f cycles per write\n", nm[2], (double)h[0] / iters / 16);
    return 0;
}
