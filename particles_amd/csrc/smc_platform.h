// smc_platform.h -- the one place that names the HIP runtime.
//
// Product builds (hipcc --offload-arch=gfx950) include <hip/hip_runtime.h>.
// The CPU test-suite additionally compiles the same sources with g++
// -DSMC_EMULATE against tests/emu/hip_emu.h (a fiber emulator of the handful
// of HIP constructs used here) so kernel logic can be checked without a GPU;
// that build is test infrastructure and is never loaded by the product.
#pragma once

#ifdef SMC_EMULATE
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define SMC_LAUNCH(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), 0, (stream), __VA_ARGS__)
#endif

#include <cstdint>

typedef unsigned long long u64;
typedef long long i64;
typedef unsigned int u32;

// 4 doubles per lane: the C/D operand of v_mfma_f64_16x16x4_f64
#ifdef SMC_EMULATE
struct smc_v4d {
    double v[4];
    double& operator[](int i) { return v[i]; }
    const double& operator[](int i) const { return v[i]; }
};
#define smc_mfma_f64_16x16x4(a, b, c) hipemu_mfma_f64_16x16x4((a), (b), (c))
#else
typedef double smc_v4d __attribute__((ext_vector_type(4)));
#define smc_mfma_f64_16x16x4(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#endif

// Pointers read out of the argument block are generic to the compiler and turn
// into flat_* accesses, which count on lgkmcnt as well: every wait for an LDS
// read then also waits for the global loads in flight.  SMC_GLOBAL marks them
// as global memory (address space 1) so they become global_* (vmcnt only).
#ifdef SMC_EMULATE
#define SMC_GLOBAL(T) T*
#define SMC_AS_GLOBAL(T, p) (p)
#else
#define SMC_GLOBAL(T) T __attribute__((address_space(1)))*
#define SMC_AS_GLOBAL(T, p) ((T __attribute__((address_space(1)))*)(p))
#endif

// Loads / stores through a generic pointer known to address global memory
// (see SMC_GLOBAL): one or two consecutive elements (8 / 16 bytes).
#ifdef SMC_EMULATE
template <class T> inline T smc_ldg(const T* p) { return *p; }
template <class T> inline void smc_stg(T* p, T v) { *p = v; }
template <class T> inline void smc_ld2g(const T* p, T& a, T& b) { a = p[0]; b = p[1]; }
template <class T> inline void smc_st2g(T* p, T a, T b) { p[0] = a; p[1] = b; }
template <class T> inline void smc_st2g_nt(T* p, T a, T b) { p[0] = a; p[1] = b; }
template <class T> inline void smc_st4g_nt(T* p, const T (&v)[4]) { for (int i = 0; i < 4; ++i) p[i] = v[i]; }
template <class T> inline void smc_ld4g(const T* p, T (&o)[4]) { for (int i = 0; i < 4; ++i) o[i] = p[i]; }
template <class T> inline void smc_st4g(T* p, const T (&v)[4]) { for (int i = 0; i < 4; ++i) p[i] = v[i]; }
#else
template <class T> __device__ __forceinline__ T smc_ldg(const T* p) { return *SMC_AS_GLOBAL(const T, p); }
template <class T> __device__ __forceinline__ void smc_stg(T* p, T v) { *SMC_AS_GLOBAL(T, p) = v; }
template <class T> __device__ __forceinline__ void smc_ld2g(const T* p, T& a, T& b)
{
    typedef T v2 __attribute__((ext_vector_type(2)));
    const v2 v = *SMC_AS_GLOBAL(const v2, p);
    a = v.x; b = v.y;
}
template <class T> __device__ __forceinline__ void smc_st2g(T* p, T a, T b)
{
    typedef T v2 __attribute__((ext_vector_type(2)));
    v2 v; v.x = a; v.y = b;
    *SMC_AS_GLOBAL(v2, p) = v;
}
// streaming variants (`nt`): the lines leave the L2 as they are written instead of waiting for
// the write-back at the end of the kernel -- 1.5 us less per launch when the launch is short
// (C2: 16 MB of new particles per 10 us kernel), no gain when it is long
template <class T> __device__ __forceinline__ void smc_st2g_nt(T* p, T a, T b)
{
    typedef T v2 __attribute__((ext_vector_type(2)));
    v2 v; v.x = a; v.y = b;
    // (inline asm: behind a run-time flag the optimiser merges a __builtin_nontemporal_store with
    //  the plain store of the other branch and drops the hint)
    static_assert(sizeof(v2) == 16, "16-byte store");
    asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
}
template <class T> __device__ __forceinline__ void smc_st4g_nt(T* p, const T (&v)[4])
{
    typedef T v4 __attribute__((ext_vector_type(4)));
    v4 w; w.x = v[0]; w.y = v[1]; w.z = v[2]; w.w = v[3];
    static_assert(sizeof(v4) == 16, "16-byte store");
    asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(w) : "memory");
}
template <class T> __device__ __forceinline__ void smc_ld4g(const T* p, T (&o)[4])
{
    typedef T v4 __attribute__((ext_vector_type(4)));
    const v4 v = *SMC_AS_GLOBAL(const v4, p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <class T> __device__ __forceinline__ void smc_st4g(T* p, const T (&v)[4])
{
    typedef T v4 __attribute__((ext_vector_type(4)));
    v4 w; w.x = v[0]; w.y = v[1]; w.z = v[2]; w.w = v[3];
    *SMC_AS_GLOBAL(v4, p) = w;
}
#endif
