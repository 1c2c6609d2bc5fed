// micro-benchmarks, memory side (round 3): 2-byte-aligned 16-byte global accesses, raw buffer loads out of range,
// granule ping-pong between two workgroups (same / other XCD, store flavours).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

// every lane writes 16 bytes at byte offset base + lane*stride + mis
__global__ void k_unal_store(unsigned char *p, int mis, size_t stride_b, int reps, size_t span)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < reps; ++r) {
        unsigned char *q = p + ((tid * stride_b + (size_t)r * 16) % span) + mis;
        v4u v = {(unsigned)tid, (unsigned)r, 0x12345678u, 0x9abcdef0u};
        __builtin_nontemporal_store(v, (v4u *)q);
    }
}
__global__ void k_unal_check_w(unsigned short *p, int mis2)   // one wave: lane l writes 8 u16 = (l<<8 | e) at element offset l*24 + mis2
{
    const int l = threadIdx.x;
    unsigned short tmp[8];
    for (int e = 0; e < 8; ++e) tmp[e] = (unsigned short)((l << 8) | e);
    v4u v; __builtin_memcpy(&v, tmp, 16);
    *(v4u *)(p + l * 24 + mis2) = v;
}
__global__ void k_unal_check_r(const unsigned short *p, int mis2, unsigned short *out)
{
    const int l = threadIdx.x;
    v4u v = *(const v4u *)(p + l * 24 + mis2);
    unsigned short tmp[8]; __builtin_memcpy(tmp, &v, 16);
    for (int e = 0; e < 8; ++e) out[l * 8 + e] = tmp[e];
}
__global__ void k_rawbuf(const float *p, int n, float *out)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, n * 4, 0x00020000);
    const int l = threadIdx.x;
    // offsets: lane 0: -16 bytes; lane 1: 0; lane 2: n*4-16; lane 3: n*4-8 (straddles the end); lane 4: n*4; lane 5: 2 (misaligned)
    int off = 0;
    if (l == 0) off = -16; else if (l == 1) off = 0; else if (l == 2) off = n * 4 - 16; else if (l == 3) off = n * 4 - 8; else if (l == 4) off = n * 4; else if (l == 5) off = 2; else off = l * 16;
    v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    float f[4]; __builtin_memcpy(f, &v, 16);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = f[e];
}
// ping-pong between block 0 and block `other`: granule = {seq<<32 | payload}
template <int MODE>   // 0: sc1 atomic store + sc1 load; 1: plain store + sc1 load (same XCD only)
__global__ void k_pp(unsigned long long *g, int other, int iters, unsigned long long *res, unsigned *xcc)
{
    if (blockIdx.x != 0 && (int)blockIdx.x != other) return;
    const bool first = blockIdx.x == 0;
    unsigned x = 0; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) xcc[first ? 0 : 1] = x;
    if (threadIdx.x != 0) return;
    unsigned long long *mine = g + (first ? 0 : 64), *theirs = g + (first ? 64 : 0);
    unsigned long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        if (first) {
            if (MODE == 0) __hip_atomic_store(mine, (unsigned long long)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *(volatile unsigned long long *)mine = it;
            while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)it) { }
        } else {
            while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)it) { }
            if (MODE == 0) __hip_atomic_store(mine, (unsigned long long)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *(volatile unsigned long long *)mine = it;
        }
    }
    unsigned long long t1 = wall_clock64();
    if (first) res[0] = t1 - t0;
}

int main()
{
    // 1. correctness of 2-byte-aligned 16-byte accesses
    unsigned short *d16, *dout; CK(hipMalloc(&d16, 64 * 24 * 2 + 64)); CK(hipMalloc(&dout, 64 * 8 * 2));
    for (int mis = 0; mis < 8; ++mis) {
        CK(hipMemset(d16, 0, 64 * 24 * 2 + 64));
        k_unal_check_w<<<1, 64>>>(d16, mis);
        k_unal_check_r<<<1, 64>>>(d16, mis, dout);
        std::vector<unsigned short> h(64 * 8), raw(64 * 24 + 32);
        CK(hipMemcpy(h.data(), dout, 64 * 8 * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(raw.data(), d16, raw.size() * 2, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) { if (h[l * 8 + e] != ((l << 8) | e)) ++bad; if (raw[l * 24 + mis + e] != ((l << 8) | e)) ++bad; }
        printf("16-byte access at u16 offset %d (byte %d mod 16): %s\n", mis, (mis * 2) % 16, bad ? "WRONG" : "ok");
    }
    // 2. bandwidth of misaligned 16-byte stores, each lane its own 64-byte line region (stride 2048 B) vs contiguous
    {
        const size_t span = 512u << 20;
        unsigned char *p; CK(hipMalloc(&p, span + 4096));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int stride : {16, 2048}) for (int mis : {0, 4, 2}) {
            const int blocks = 4096, reps = 64;
            k_unal_store<<<blocks, 256>>>(p, mis, stride, reps, span);
            CK(hipEventRecord(e0));
            k_unal_store<<<blocks, 256>>>(p, mis, stride, reps, span);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("stores: lane stride %5d B, misalignment %d B: %.1f GB/s\n", stride, mis, (double)blocks * 256 * reps * 16 / ms / 1e6);
        }
    }
    // 3. raw buffer load out of range
    {
        const int n = 1024; float *p, *o; CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&o, 64 * 16));
        std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = (float)(i + 1);
        CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
        k_rawbuf<<<1, 64>>>(p, n, o);
        float r[64 * 4]; CK(hipMemcpy(r, o, 64 * 16, hipMemcpyDeviceToHost));
        const char *nm[6] = {"offset -16", "offset 0", "last 16 B", "straddles end", "at end", "offset 2 (misaligned)"};
        for (int l = 0; l < 6; ++l) printf("raw buffer load, %-22s: %g %g %g %g\n", nm[l], r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3]);
    }
    // 4. granule ping-pong
    {
        unsigned long long *g, *res; unsigned *xcc; CK(hipMalloc(&g, 4096)); CK(hipMalloc(&res, 64)); CK(hipMalloc(&xcc, 64));
        const int iters = 2000;
        for (int other : {8, 1, 16, 3}) {
            for (int mode = 0; mode < 2; ++mode) {
                CK(hipMemset(g, 0, 4096));
                if (mode == 0) k_pp<0><<<256, 64>>>(g, other, iters, res, xcc); else k_pp<1><<<256, 64>>>(g, other, iters, res, xcc);
                hipError_t e = hipDeviceSynchronize();
                if (e != hipSuccess) { printf("ping-pong mode %d other %d: %s\n", mode, other, hipGetErrorString(e)); return 1; }
                unsigned long long r; unsigned x[2]; CK(hipMemcpy(&r, res, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost));
                printf("ping-pong block 0 (xcc %u) <-> block %d (xcc %u), %s: %.0f ns one way\n", x[0], other, x[1], mode == 0 ? "sc1 store + sc1 load " : "plain store + sc1 load", (double)r * 10.0 / (2.0 * iters));
                if (mode == 1 && x[0] != x[1]) { }
            }
        }
    }
    return 0;
}
