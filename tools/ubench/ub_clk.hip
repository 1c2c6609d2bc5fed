// calibration: what do __builtin_readcyclecounter() (s_memtime) and wall_clock64() (s_memrealtime) tick at, and how long does a
// dependent v_add_f32 take in nanoseconds -- cold (first launch after idle) and after one second of continuous load.
// build: hipcc --offload-arch=gfx950 -O3 -o ub_clk ub_clk.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void k_chain(float *out, unsigned long long *t, int iters)
{
    float a = out[threadIdx.x];
    const float c = out[100];
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
template <int BYTES> __global__ __launch_bounds__(256) void k_lds(float *out)
{
    __shared__ float buf[BYTES / 4];
    buf[threadIdx.x] = out[threadIdx.x]; __syncthreads(); out[threadIdx.x] = buf[255 - threadIdx.x];
}
template <int BYTES> void occ()
{
    int nb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_lds<BYTES>, 256, 0));
    printf("LDS %6d B per workgroup of 256: occupancy API says %d workgroups per CU\n", BYTES, nb);
}
int main()
{
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, CUs %d, regsPerBlock %d\n", pr.sharedMemPerBlock, pr.maxSharedMemoryPerMultiProcessor, pr.multiProcessorCount, pr.regsPerBlock);
    occ<32768>(); occ<40960>(); occ<42640>(); occ<50848>(); occ<53248>(); occ<54528>(); occ<65536>();
    float *d; unsigned long long *t, h[2];
    CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096)); CK(hipMalloc(&t, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;                       // 1.6 M dependent adds
    for (int phase = 0; phase < 3; ++phase) {
        const int grid = phase == 2 ? 2048 : 1;     // phase 2: the whole chip busy
        if (phase >= 1) { for (int k = 0; k < 300; ++k) hipLaunchKernelGGL(k_chain, dim3(2048), dim3(256), 0, 0, d, t, 20000); CK(hipDeviceSynchronize()); }
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_chain, dim3(grid), dim3(64), 0, 0, d, t, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));
            printf("phase %d (%s) rep %d: %.3f ms for %d dependent adds = %.2f ns each; s_memtime %llu ticks (%.1f MHz), wall_clock64 %llu ticks (%.1f MHz)\n", phase,
                   phase == 0 ? "cold" : phase == 1 ? "after load, 1 wave" : "after load, 2048 x 64", rep, ms, iters * 16, ms * 1e6 / (iters * 16.0),
                   h[0], h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3));
        }
    }
    return 0;
}
