// hip_runtime.h -- TEST INFRASTRUCTURE ONLY: a minimal "HIP on the CPU" shim so that the product's
// sz_amd/csrc/szhip.hip (kernels + host orchestration) can be compiled by g++ and run, one workgroup at
// a time, in the GPU-less test container.  Each GPU thread is an OS thread from a pool; __syncthreads and
// the wave-64 collectives are barriers + exchange buffers.  It checks LOGIC (indexing, scans, bit packing,
// stream assembly); it says nothing about the GPU memory model or performance.
#pragma once
#include <pthread.h>
#include <sched.h>
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define SZH_HIPSIM 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define SZH_SYNC_LAUNCH 1   /* a launch returns when the kernel has finished: nothing on the host can overlap with it */
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct uint4 { unsigned x, y, z, w; };
struct int4 { int x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r = {x, y, z, w}; return r; }
struct uint2 { unsigned x, y; };

namespace hipsim {
struct TIdx { unsigned x, y, z; };
extern TIdx t_threadIdx;
extern dim3 t_blockIdx;
void fiber_yield();
extern dim3 g_blockDim, g_gridDim;
extern char *g_dyn_smem;
void sync_block();
void wave_exchange(const void *src, void *dst_all, size_t elem); // every lane of the calling thread's wave contributes one element
int lanes_in_wave();
void launch(const std::function<void()> &body, dim3 grid, dim3 block, size_t shmem);
}
#define threadIdx hipsim::t_threadIdx
#define blockIdx hipsim::t_blockIdx
#define blockDim hipsim::g_blockDim
#define gridDim hipsim::g_gridDim
inline char *hipsim_dyn_smem() { return hipsim::g_dyn_smem; }

inline void __syncthreads() { hipsim::sync_block(); }

template <class T> inline T __shfl(T v, int lane, int = 64)
{
    T all[64]; hipsim::wave_exchange(&v, all, sizeof(T));
    return all[lane & 63];
}
template <class T> inline T __shfl_up(T v, int d, int = 64)
{
    T all[64]; hipsim::wave_exchange(&v, all, sizeof(T));
    const int me = threadIdx.x & 63;
    return me >= d ? all[me - d] : v;
}
template <class T> inline T __shfl_xor(T v, int m, int = 64)
{
    T all[64]; hipsim::wave_exchange(&v, all, sizeof(T));
    const int me = threadIdx.x & 63, src = me ^ m;
    return src < hipsim::lanes_in_wave() ? all[src] : v;
}
inline unsigned long long __ballot(int p)
{
    int all[64]; hipsim::wave_exchange(&p, all, sizeof(int));
    unsigned long long m = 0;
    for (int l = 0; l < hipsim::lanes_in_wave(); ++l) if (all[l]) m |= 1ull << l;
    return m;
}
inline int __all(int p)
{
    int all[64]; hipsim::wave_exchange(&p, all, sizeof(int));
    for (int l = 0; l < hipsim::lanes_in_wave(); ++l) if (!all[l]) return 0;
    return 1;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline long long __double_as_longlong(double d) { long long u; memcpy(&u, &d, 8); return u; }
inline void __builtin_amdgcn_s_sleep(int) { hipsim::fiber_yield(); } // a waiting lane hands the processor to the next fibre
inline unsigned long long wall_clock64() { return 0; }
inline int min(int a, int b) { return a < b ? a : b; }

template <class T> inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicMin(T *p, T v)
{
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
template <class T> inline T atomicMax(T *p, T v)
{
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
template <class T> inline T __hip_atomic_load(T *p, int, int) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
template <class T> inline void __hip_atomic_store(T *p, T v, int, int) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }

static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline double __longlong_as_double(long long u) { double f; memcpy(&f, &u, 8); return f; }

// ---- host runtime API subset ----
typedef int hipError_t;
#define hipSuccess 0
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
#define hipStreamNonBlocking 1
#define hipHostMallocDefault 0
#define hipHostMallocCoherent 0x40000000
#define hipHostMallocMapped 0x2
inline const char *hipGetErrorString(hipError_t) { return "hipsim"; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; }
template <class K> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *nb, K, int, size_t) { *nb = 2; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, int) { *s = (void *)1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, int, int) { *s = (void *)1; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (void *)1; return hipSuccess; }
#define hipEventDisableTiming 2
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, int) { *e = (void *)1; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, int) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
#define hipErrorNotReady 600
// (never "ready": a caller that polls an event next to device-written progress words must get there by the words -- a missing word hangs the test)
inline hipError_t hipEventQuery(hipEvent_t) { return hipErrorNotReady; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); memset(*p, 0xA5, n); return *p ? hipSuccess : 1; } // poison: catch reads of uninitialised workspaces
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, int) { *p = malloc(n ? n : 1); return hipSuccess; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipsim::launch([&]() { kernel(__VA_ARGS__); }, (grid), (block), (shmem))
