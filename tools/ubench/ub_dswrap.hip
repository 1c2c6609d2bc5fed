// does a DS access compute (VGPR address + 16-bit offset) modulo 2^32?  (round 6: lane base pointers below the ring's first byte)
#include <hip/hip_runtime.h>
#include <cstdio>
#define LDS __attribute__((address_space(3)))
__global__ void k(unsigned *out)
{
    __shared__ unsigned buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) buf[i] = 0xA0000000u + i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(LDS unsigned *)buf;
    unsigned neg = base + threadIdx.x * 4u - 0x3000u;          // below the buffer for small lanes (wraps below zero when base < 0x3000)
    unsigned v, w;
    asm volatile("ds_read_b32 %0, %1 offset:0x3000\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(neg));
    asm volatile("ds_write_b32 %0, %1 offset:0x3400\n\ts_waitcnt lgkmcnt(0)" :: "v"(neg), "v"(0xB0000000u + threadIdx.x) : "memory");
    __syncthreads();
    w = buf[256 + threadIdx.x];
    out[threadIdx.x] = v; out[64 + threadIdx.x] = w; out[128] = base;
}
int main()
{
    unsigned *d, h[129]; hipMalloc(&d, 129 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 129 * 4, hipMemcpyDeviceToHost);
    int ok = 1; for (int i = 0; i < 64; ++i) if (h[i] != 0xA0000000u + i || h[64 + i] != 0xB0000000u + i) ok = 0;
    printf("lds base %u: read %08x %08x ... write-back %08x %08x: negative base + offset %s\n", h[128], h[0], h[63], h[64], h[127], ok ? "WRAPS (usable)" : "does NOT wrap");
    return 0;
}
