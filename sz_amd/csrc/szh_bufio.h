// szh_bufio.h -- raw-buffer, LDS and cross-lane helpers shared by the sweep kernels (szh_beam.h, szh_ompcol.h): 16-byte buffer loads / stores with
// out-of-range offsets that read zeros / store nothing, write-through granule stores, LDS accesses that stay 32-bit offsets, the CPU shim's stand-ins.
// (Until round 6 the head of szh_ribbon.h, the hyperplane "ribbon" mapping of the sweep -- rounds 3 - 5, 1.13 ms at 512^3 float --, which the lean beam of
// round 6 replaced; arrays the beam does not take run k_pencil.)
#pragma once
#include "szh_pencil.h"

namespace szh_io {
#ifdef SZH_HIPSIM
struct v4u { unsigned x, y, z, w; };
#else
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#endif

#ifdef SZH_HIPSIM
#define SZH_LDS
struct rsrc_t { char *base; unsigned n; };
static inline rsrc_t make_rsrc(const void *p, unsigned bytes) { rsrc_t r = {(char *)const_cast<void *>(p), bytes}; return r; }
static inline v4u bload16(rsrc_t rs, unsigned off)
{
    unsigned w[4];
    for (int e = 0; e < 4; ++e) { w[e] = 0; if ((uint64_t)off + 4u * e + 4u <= rs.n) memcpy(&w[e], rs.base + off + 4u * e, 4); }
    v4u v = {w[0], w[1], w[2], w[3]}; return v;
}
static inline void bstore16(rsrc_t rs, unsigned off, v4u v)
{
    unsigned w[4] = {v.x, v.y, v.z, v.w};
    for (int e = 0; e < 4; ++e) if ((uint64_t)off + 4u * e + 4u <= rs.n) memcpy(rs.base + off + 4u * e, &w[e], 4);
}
static inline void bstore2(rsrc_t rs, unsigned off, unsigned short v) { if ((uint64_t)off + 2u <= rs.n) memcpy(rs.base + off, &v, 2); }
template <class T> static inline void bstoreT(rsrc_t rs, unsigned off, T v) { if ((uint64_t)off + sizeof(T) <= rs.n) memcpy(rs.base + off, &v, sizeof(T)); }
static inline void bstore8_wt(rsrc_t rs, unsigned off, unsigned soff, szh_u64 g) { if ((uint64_t)off + soff + 8u <= rs.n) __atomic_store_n((szh_u64 *)(rs.base + off + soff), g, __ATOMIC_RELAXED); }
static inline void bstore16_wt(rsrc_t rs, unsigned off, unsigned soff, szh_u64 g0, szh_u64 g1) { bstore8_wt(rs, off, soff, g0); bstore8_wt(rs, off + 8u, soff, g1); }
template <class E> static inline E lds_ld(const E *p)
{
    E v;
    if (sizeof(E) == 8) { const uint64_t u = __atomic_load_n((const uint64_t *)p, __ATOMIC_RELAXED); memcpy(&v, &u, sizeof(E)); }
    else { const uint32_t u = __atomic_load_n((const uint32_t *)p, __ATOMIC_RELAXED); memcpy(&v, &u, sizeof(E)); }
    return v;
}
template <class E> static inline void lds_st(E *p, E v)
{
    if (sizeof(E) == 8) { uint64_t u; memcpy(&u, &v, sizeof(E)); __atomic_store_n((uint64_t *)p, u, __ATOMIC_RELAXED); }
    else { uint32_t u; memcpy(&u, &v, sizeof(E)); __atomic_store_n((uint32_t *)p, u, __ATOMIC_RELAXED); }
}
static inline int uni(int v) { return __shfl(v, 0, 64); }
static inline void lds_fence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); (void)__all(1); __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void lds_order() { lds_fence(); }
template <class T> static inline T shr1(T old, T v) { const T s = __shfl_up(v, 1, 64); return (threadIdx.x & 63) == 0 ? old : s; }
static inline void prio(int) {}
static inline void keep(float &) {}
static inline void keep(double &) {}
static inline void keepu(unsigned &) {}
#else
#define SZH_LDS __attribute__((address_space(3)))   /* LDS pointers stay 32-bit offsets: no generic-pointer arithmetic in the sweep */
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000); }
// raw buffer accesses: out-of-range dwords read 0 / are dropped (tools/ubench/ub_mem.hip), any byte alignment
__device__ __forceinline__ v4u bload16(rsrc_t rs, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0); }
__device__ __forceinline__ void bstore16(rsrc_t rs, unsigned off, v4u v) { __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, 0); }
__device__ __forceinline__ void bstore2(rsrc_t rs, unsigned off, unsigned short v) { __builtin_amdgcn_raw_buffer_store_b16((short)v, rs, (int)off, 0, 0); }
__device__ __forceinline__ void bstoreT(rsrc_t rs, unsigned off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)off, 0, 0); }
__device__ __forceinline__ void bstoreT(rsrc_t rs, unsigned off, double v)
{
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    v2u w = {(unsigned)u, (unsigned)(u >> 32)};
    __builtin_amdgcn_raw_buffer_store_b64(w, rs, (int)off, 0, 0);
}
// granule stores: written through, not kept in this XCD's L2 (aux 17 = sc0 sc1, the agent-scope form); NOT volatile / atomic -- hipcc
// drains the memory queue behind those
__device__ __forceinline__ void bstore8_wt(rsrc_t rs, unsigned off, unsigned soff, szh_u64 g)
{
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    v2u w = {(unsigned)g, (unsigned)(g >> 32)};
    __builtin_amdgcn_raw_buffer_store_b64(w, rs, (int)off, (int)soff, 17);
}
__device__ __forceinline__ void bstore16_wt(rsrc_t rs, unsigned off, unsigned soff, szh_u64 g0, szh_u64 g1)
{
    v4u w = {(unsigned)g0, (unsigned)(g0 >> 32), (unsigned)g1, (unsigned)(g1 >> 32)};
    __builtin_amdgcn_raw_buffer_store_b128(w, rs, (int)off, (int)soff, 17);
    // Seen on gfx950 (round 3): with a REGISTER soffset hipcc assumes the ">64-bit store data" hazard does not exist and lets the very
    // next VALU instruction overwrite the data registers; now and then the store then wrote the NEW contents (an LDS address in the tag
    // word of a granule).  Keep the four registers alive across a few wait states.
    asm volatile("s_nop 3" :: "v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w));
}
template <class E> __device__ __forceinline__ E lds_ld(const SZH_LDS E *p) { return *(const volatile SZH_LDS E *)p; }
template <class E> __device__ __forceinline__ void lds_st(SZH_LDS E *p, E v) { *(volatile SZH_LDS E *)p = v; }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }     // one wavefront's LDS accesses execute in program order
// lane l receives v of lane l - 1; lane 0 keeps `old` (DPP wave_shr:1, bound_ctrl 0: tools/ubench/ub_valu.hip)
__device__ __forceinline__ float shr1(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ double shr1(double old, double v)
{
    const long long o = __double_as_longlong(old), s = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)o, (int)s, 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(s >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ void prio(int p) { if (p >= 3) __builtin_amdgcn_s_setprio(3); else if (p == 2) __builtin_amdgcn_s_setprio(2); else if (p == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
__device__ __forceinline__ void keep(float &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void keep(double &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void keepu(unsigned &v) { asm volatile("" : "+v"(v)); }
#endif
__device__ __forceinline__ szh_u64 ld_gran(const szh_u64 *p) { return __hip_atomic_load(const_cast<szh_u64 *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_gran(szh_u64 *p, szh_u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_flag(const unsigned *p) { return __hip_atomic_load(const_cast<unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_done(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void nap(int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1); }

} // namespace szh_io
