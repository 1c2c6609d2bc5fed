// micro-benchmarks that settle design questions of the predict+quantise kernel (round 3):
//   dependent / independent VALU issue, DPP wave_shr:1 (semantics + latency), ds_bpermute latency,
//   LDS hand-off latency between two wavefronts of one workgroup, packed f32 adds.
// build: hipcc --offload-arch=gfx950 -O3 -o ub_valu ub_valu.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ inline unsigned long long clk() { return __builtin_readcyclecounter(); }
__device__ inline unsigned long long rt() { return wall_clock64(); }

template <int ILP>
__global__ void k_chain(float *out, unsigned long long *cyc, int iters)
{
    float a[ILP];
    for (int i = 0; i < ILP; ++i) a[i] = out[threadIdx.x + i];
    const float c = out[100];
    unsigned long long t0 = clk();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        }
    }
    unsigned long long t1 = clk();
    float s = 0; for (int i = 0; i < ILP; ++i) s += a[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// chain add -> dpp wave_shr:1 -> add ...
__global__ void k_dpp(float *out, unsigned long long *cyc, int iters)
{
    float a = out[threadIdx.x];
    const float c = out[100];
    unsigned long long t0 = clk();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int v = __builtin_amdgcn_update_dpp(__float_as_int(a), __float_as_int(a), 0x138, 0xf, 0xf, false);
            a = __int_as_float(v) + c;
        }
    }
    unsigned long long t1 = clk();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// dpp fused into the add as src0 modifier
__global__ void k_dpp_fused(float *out, unsigned long long *cyc, int iters)
{
    float a = out[threadIdx.x];
    unsigned long long t0 = clk();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("v_add_f32_dpp %0, %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(a));
    }
    unsigned long long t1 = clk();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_dpp_sem(int *out)
{
    int lane = threadIdx.x;
    int old = 1000 + lane;
    out[lane] = __builtin_amdgcn_update_dpp(old, lane, 0x138, 0xf, 0xf, false);            // wave_shr:1, bound_ctrl 0
    out[64 + lane] = __builtin_amdgcn_update_dpp(old, lane, 0x138, 0xf, 0xf, true);       // bound_ctrl 1
    out[128 + lane] = __builtin_amdgcn_update_dpp(old, lane, 0x111, 0xf, 0xf, false);     // row_shr:1
    out[192 + lane] = __builtin_amdgcn_update_dpp(old, lane, 0x130, 0xf, 0xf, false);     // wave_shl:1
}
__global__ void k_bperm(float *out, unsigned long long *cyc, int iters)
{
    float a = out[threadIdx.x];
    const float c = out[100];
    const int idx = ((threadIdx.x + 56) & 63) * 4;
    unsigned long long t0 = clk();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int v = __builtin_amdgcn_ds_bpermute(idx, __float_as_int(a));
            a = __int_as_float(v) + c;
        }
    }
    unsigned long long t1 = clk();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// LDS read-after-write by the same wave (latency of ds_write -> ds_read round trip)
__global__ void k_lds_rt(float *out, unsigned long long *cyc, int iters)
{
    __shared__ float buf[256];
    float a = out[threadIdx.x];
    const float c = out[100];
    unsigned long long t0 = clk();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            buf[threadIdx.x] = a;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            a = ((volatile float *)buf)[threadIdx.x ^ 1] + c;
        }
    }
    unsigned long long t1 = clk();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// two wavefronts of one workgroup ping-pong through LDS words: one-way latency = cycles / (2 * iters)
__global__ void k_pingpong(unsigned long long *cyc, int iters)
{
    __shared__ volatile unsigned flag[2];
    const int w = threadIdx.x >> 6;
    if (threadIdx.x < 2) flag[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long t0 = clk();
    for (int it = 1; it <= iters; ++it) {
        if (w == 0) {
            flag[0] = it;
            while (flag[1] < (unsigned)it) { }
        } else {
            while (flag[0] < (unsigned)it) { }
            flag[1] = it;
        }
    }
    unsigned long long t1 = clk();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k_pk(float *out, unsigned long long *cyc, int iters)
{
    f2 a = {out[threadIdx.x], out[threadIdx.x + 1]}, b = {out[threadIdx.x + 2], out[threadIdx.x + 3]};
    const f2 c = {out[100], out[101]};
    unsigned long long t0 = clk();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(b) : "v"(c));
        }
    }
    unsigned long long t1 = clk();
    out[threadIdx.x] = a.x + a.y + b.x + b.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// the quantiser chain as the compiler sees it: cycles per dependent "step" of R interleaved rows
template <int R>
__global__ void k_quant(float *out, unsigned long long *cyc, int iters, float eb, float recip)
{
    float cur[R], x[R];
    for (int r = 0; r < R; ++r) { cur[r] = out[threadIdx.x + r]; x[r] = out[threadIdx.x + 8 + r]; }
    int acc = 0;
    unsigned long long t0 = clk();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float pred = cur[r];
                const float diff = x[r] - pred;
                float itv = __builtin_fabsf(diff) * recip + 1.0f;
                const bool inr = itv < 30.0f;
                itv = inr ? itv : 0.0f;
                const float sitv = diff < 0 ? -itv : itv;
                const int q = (int)(sitv / 2);
                const float rc = pred + (float)(2 * q) * eb;
                const bool ok = inr && !(__builtin_fabsf(x[r] - rc) > eb);
                cur[r] = ok ? rc : x[r];
                acc += ok ? q + 16 : 0;
                x[r] += 1e-3f;
            }
        }
    }
    unsigned long long t1 = clk();
    float sum = 0; for (int r = 0; r < R; ++r) sum += cur[r];
    out[threadIdx.x] = sum + acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    float *d; unsigned long long *c;
    CK(hipMalloc(&d, 4096 * 4)); CK(hipMalloc(&c, 64 * 8));
    std::vector<float> h(4096, 1.0f);
    CK(hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    unsigned long long hc[4];
    const int iters = 2000;
    auto rep = [&](const char *name, double per) {
        CK(hipDeviceSynchronize()); CK(hipMemcpy(hc, c, 8, hipMemcpyDeviceToHost));
        printf("%-40s %8.2f cycles per op (total %llu)\n", name, (double)hc[0] / per, hc[0]);
    };
    for (int pass = 0; pass < 2; ++pass) {
        k_chain<1><<<1, 64>>>(d, c, iters); rep("dependent v_add chain (1 wave)", iters * 16.0);
        k_chain<2><<<1, 64>>>(d, c, iters); rep("2 interleaved chains, per instr", iters * 32.0);
        k_chain<4><<<1, 64>>>(d, c, iters); rep("4 interleaved chains, per instr", iters * 64.0);
        k_chain<8><<<1, 64>>>(d, c, iters); rep("8 interleaved chains, per instr", iters * 128.0);
        k_chain<1><<<1, 128>>>(d, c, iters); rep("dependent chain, 2 waves/CU (diff SIMD?)", iters * 16.0);
        k_chain<1><<<1, 512>>>(d, c, iters); rep("dependent chain, 8 waves/CU (2/SIMD)", iters * 16.0);
        k_chain<4><<<1, 512>>>(d, c, iters); rep("4 chains, 8 waves/CU (2/SIMD), per instr", iters * 64.0);
        k_chain<1><<<1, 1024>>>(d, c, iters); rep("dependent chain, 16 waves/CU (4/SIMD)", iters * 16.0);
        k_dpp<<<1, 64>>>(d, c, iters); rep("add + mov_dpp wave_shr chain, per pair", iters * 16.0);
        k_dpp_fused<<<1, 64>>>(d, c, iters); rep("v_add_dpp wave_shr fused (+s_nop 1)", iters * 16.0);
        k_bperm<<<1, 64>>>(d, c, iters); rep("add + ds_bpermute chain, per pair", iters * 16.0);
        k_lds_rt<<<1, 64>>>(d, c, iters); rep("LDS write -> read -> add, per round", iters * 16.0);
        k_pingpong<<<1, 128>>>(c, iters); rep("LDS flag ping-pong, per one-way hop", iters * 2.0);
        k_pk<<<1, 64>>>(d, c, iters); rep("v_pk_add_f32 2 chains, per instr", iters * 32.0);
        k_quant<1><<<1, 64>>>(d, c, iters, 1e-4f, 1e4f); rep("quantiser chain R=1, per point", iters * 8.0);
        k_quant<2><<<1, 64>>>(d, c, iters, 1e-4f, 1e4f); rep("quantiser R=2, per point", iters * 16.0);
        k_quant<4><<<1, 64>>>(d, c, iters, 1e-4f, 1e4f); rep("quantiser R=4, per point", iters * 32.0);
        k_quant<8><<<1, 64>>>(d, c, iters, 1e-4f, 1e4f); rep("quantiser R=8, per point", iters * 64.0);
        k_quant<4><<<1, 512>>>(d, c, iters, 1e-4f, 1e4f); rep("quantiser R=4, 2 waves/SIMD, per point", iters * 32.0);
    }
    int *di; CK(hipMalloc(&di, 256 * 4));
    k_dpp_sem<<<1, 64>>>(di);
    int hi[256]; CK(hipMemcpy(hi, di, 256 * 4, hipMemcpyDeviceToHost));
    const char *nm[4] = {"wave_shr:1 bc0", "wave_shr:1 bc1", "row_shr:1 bc0", "wave_shl:1 bc0"};
    for (int v = 0; v < 4; ++v) {
        printf("%s:", nm[v]);
        for (int l = 0; l < 64; ++l) if (l < 3 || (l >= 14 && l <= 18) || (l >= 30 && l <= 34) || l >= 61) printf(" [%d]=%d", l, hi[v * 64 + l]);
        printf("\n");
    }
    // clock rate: cycles vs wall clock
    return 0;
}
