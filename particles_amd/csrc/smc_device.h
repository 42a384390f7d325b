// smc_device.h -- device-side building blocks shared by the smc kernels.
//
// Workgroups are 256 threads = 4 wave64; all cross-lane code assumes 64 lanes
// and runs on the DPP path (smc_dpp.h).  Compiled with -ffp-contract=off: every
// expression that mirrors a line of the reference is evaluated with the
// reference's roundings (no silent FMA).
#pragma once
#include "smc_platform.h"

#include <cmath>

#include "smc_dpp.h"
#include "smc_math.h"

#define SMC_BLOCK 256
#define SMC_NWAVE (SMC_BLOCK / 64)
#define SMC_SM (2 * SMC_NWAVE)          /* LDS scratch slots the collectives may use */
#define SMC_C_NORM 0.9189385332046727   /* scipy _norm_pdf_logC (distributions.py:273) */
#define SMC_HALFLOG2PI 0.91893853320467267 /* 0.5*log(2*pi) (distributions.py:212) */
#define SMC_Q62 4611686018427387904.0   /* 2^62 */

#define SMC_STREAM_NORMAL 0u
#define SMC_STREAM_RESAMPLE 1u
#define SMC_STREAM_SPACINGS 2u

__device__ __forceinline__ int smc_lane() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int smc_wave() { return (int)(threadIdx.x >> 6); }

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Counter layout used everywhere:
//   ctr = (index, t, island, stream), key = (seed lo32, seed hi32)
// One call yields two 64-bit words.  Restated in oracle/smc_oracle.py and
// oracle/oracle.c; tests check the integer stream bit-for-bit.
// ---------------------------------------------------------------------------
#ifndef SMC_PHILOX_ROUNDS
#define SMC_PHILOX_ROUNDS 10
#endif
// a ^ b ^ c in one instruction (gfx950's v_bitop3_b32, truth table 0x96); left to itself the compiler keeps the
// two v_xor_b32 of each of a round's two words -- 40 of a pair of calls' 116 vector instructions
__host__ __device__ __forceinline__ u32 smc_xor3(u32 a, u32 b, u32 c)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SMC_EMULATE) && !defined(SMC_NO_BITOP3)
    return (u32)__builtin_amdgcn_bitop3_b32((int)a, (int)b, (int)c, 0x96);
#else
    return a ^ b ^ c;
#endif
}
template <bool XOR3>
__host__ __device__ __forceinline__ void smc_philox_t(u32 c0, u32 c1, u32 c2, u32 c3, u64 seed,
                                             u64& x01, u64& x23)
{
    u32 k0 = (u32)seed, k1 = (u32)(seed >> 32);
#pragma unroll
    for (int r = 0; r < SMC_PHILOX_ROUNDS; ++r) {
        if (r > 0) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        const u64 p0 = (u64)0xD2511F53u * c0, p1 = (u64)0xCD9E8D57u * c2;   // v_mad_u64_u32
        const u32 h1 = (u32)(p1 >> 32), h0 = (u32)(p0 >> 32);
        const u32 n0 = XOR3 ? smc_xor3(h1, c1, k0) : (h1 ^ c1 ^ k0), n2 = XOR3 ? smc_xor3(h0, c3, k1) : (h0 ^ c3 ^ k1);
        c0 = n0; c1 = (u32)p1; c2 = n2; c3 = (u32)p0;
    }
    x01 = ((u64)c1 << 32) | c0;
    x23 = ((u64)c3 << 32) | c2;
}
__host__ __device__ __forceinline__ void smc_philox(u32 c0, u32 c1, u32 c2, u32 c3, u64 seed, u64& x01, u64& x23)
{
    smc_philox_t<true>(c0, c1, c2, c3, seed, x01, x23);
}
// the same function for a call whose inputs are all wave-uniform (the systematic scheme's one draw): plain xors, which
// the compiler keeps on the scalar unit together with the multiplies
__host__ __device__ __forceinline__ void smc_philox_uniform(u32 c0, u32 c1, u32 c2, u32 c3, u64 seed, u64& x01, u64& x23)
{
    smc_philox_t<false>(c0, c1, c2, c3, seed, x01, x23);
}

// (0,1): 52 random bits + 1/2 ulp, exact in fp64, never 0 or 1
__device__ __forceinline__ double smc_u01_open(u64 x)
{
    return ((double)(x >> 12) + 0.5) * 0x1.0p-52;
}
// [0,1): numpy's rand() convention, 53 random bits
__device__ __forceinline__ double smc_u01_halfopen(u64 x)
{
    return (double)(x >> 11) * 0x1.0p-53;
}

// Two standard normals from one Philox call (Box-Muller):
//   r = sqrt(-2 log u1);  (z_even, z_odd) = r * (cos, sin)(2 pi u2),  u = ((x >> 12) + 1/2) 2^-52
// ntab: the tables of smc_math.h's smc_bm_pair staged in LDS (SMC_NTAB_LDS + smc_ntab_stage + a
// barrier at the top of the kernel).  -DSMC_BM_LEGACY (A/B builds only): the table-free
// evaluation of rounds 1-2 (same uniforms, results equal to a few ulp).
#define SMC_NTAB_LDS(name) __shared__ SmcD2 name[SMC_NTAB_LDS_N]
__device__ __forceinline__ void smc_normal_pair(const SmcD2* ntab, u64 seed, u32 pair, u32 t, u32 island,
                                                u32 stream, double& z0, double& z1)
{
    u64 a, b;
    smc_philox(pair, t, island, stream, seed, a, b);
#ifdef SMC_BM_LEGACY
    (void)ntab;
    const double r = sqrt(-2.0 * smc_log_pos(smc_u01_open(a)));
    double sn, cs;
    smc_sincospi_02(2.0 * smc_u01_open(b), &sn, &cs);
    z0 = r * cs;
    z1 = r * sn;
#else
    smc_bm_pair(ntab, a, b, z0, z1);
#endif
}

// the two halves of smc_normal_pair: the counter-based bits need no table -- a kernel whose tables are still
// on their way to LDS draws them first (k_propagate's prologue) -- the transform does
__device__ __forceinline__ void smc_normal_bits(u64 seed, u32 pair, u32 t, u32 island, u32 stream, u64& a, u64& b)
{
    smc_philox(pair, t, island, stream, seed, a, b);
}
__device__ __forceinline__ void smc_normal_from_bits(const SmcD2* ntab, const u64 a, const u64 b, double& z0, double& z1)
{
#ifdef SMC_BM_LEGACY
    (void)ntab;
    const double r = sqrt(-2.0 * smc_log_pos(smc_u01_open(a)));
    double sn, cs;
    smc_sincospi_02(2.0 * smc_u01_open(b), &sn, &cs);
    z0 = r * cs;
    z1 = r * sn;
#else
    smc_bm_pair(ntab, a, b, z0, z1);
#endif
}

// ---------------------------------------------------------------------------
// Publishing a few 8-byte values to another workgroup of the same launch
// without fences (MI355X: per-XCD L2s, per-CU L1s): the producer stores them
// with relaxed AGENT-scope atomics (write-through, `sc1`), drains its stores
// (s_waitcnt vmcnt(0)) and then takes a ticket with an agent-scope atomicAdd;
// the consumer that draws the last ticket reads them with relaxed agent-scope
// atomic loads (served from L2, never from a stale L1).  A __threadfence()
// here would write back the whole dirty L2 of the XCD per workgroup.
// ---------------------------------------------------------------------------
#ifdef SMC_EMULATE
__device__ __forceinline__ void smc_st_agent(u64* p, u64 v) { *p = v; }
__device__ __forceinline__ u64 smc_ld_agent(const u64* p) { return *p; }
__device__ __forceinline__ void smc_drain_stores() {}
__device__ __forceinline__ void smc_spin_pause() {}
#else
__device__ __forceinline__ void smc_st_agent(u64* p, u64 v)
{
    __hip_atomic_store(SMC_AS_GLOBAL(u64, p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 smc_ld_agent(const u64* p)
{
    return __hip_atomic_load(SMC_AS_GLOBAL(const u64, p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void smc_drain_stores()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void smc_spin_pause() { __builtin_amdgcn_s_sleep(2); }
#endif
// A value every lane of the wavefront holds identically (loaded through a per-lane
// address the compiler cannot prove uniform) -> an SGPR copy: conditions on it become
// scalar branches instead of exec-mask regions.
#ifdef SMC_EMULATE
__device__ __forceinline__ double smc_uniform(double v) { return v; }
#else
__device__ __forceinline__ double smc_uniform(double v)
{
    const u64 x = (u64)__double_as_longlong(v);
    const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)x);
    const u32 hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(x >> 32));
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}
#endif
__device__ __forceinline__ u64 smc_uniform_u64(u64 v)
{
#ifdef SMC_EMULATE
    return v;
#else
    const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
    const u32 hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32));
    return ((u64)hi << 32) | lo;
#endif
}
__device__ __forceinline__ void smc_st_agent_f64(double* p, double v)
{
    smc_st_agent(reinterpret_cast<u64*>(p), (u64)__double_as_longlong(v));
}
__device__ __forceinline__ double smc_ld_agent_f64(const double* p)
{
    return __longlong_as_double((long long)smc_ld_agent(reinterpret_cast<const u64*>(p)));
}

// ---------------------------------------------------------------------------
// Q62 fixed-point CDF contract (oracle/smc_oracle.py "Q62"):
//   q = rint(W 2^62), T = ceil(su 2^62); both exact (power-of-two scaling).
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ u64 smc_q62_w(double w)
{
    return (w > 0.0) ? (u64)rint(fmin(w, 2.0) * SMC_Q62) : 0ull;
}
__host__ __device__ __forceinline__ u64 smc_q62_t(double su)
{
    return (su > 0.0) ? (u64)ceil(fmin(su, 2.0) * SMC_Q62) : 0ull;
}

// ---------------------------------------------------------------------------
// workgroup collectives (256 threads).  Every thread of the workgroup must
// call; `sm` is LDS scratch (SMC_NWAVE slots per value) that must not be in use
// when the call starts (each routine ends with all threads past its reads only
// after the next barrier, so callers alternate scratch areas or pass `fence`).
// All results have a fixed association order.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double smc_block_max(double v, double* sm)
{
    v = smc_wave_max(v);
    __syncthreads();
    if (smc_lane() == 0) sm[smc_wave()] = v;
    __syncthreads();
    double r = sm[0];
#pragma unroll
    for (int w = 1; w < SMC_NWAVE; ++w) r = smc_max2(r, sm[w]);
    return r;
}
__device__ __forceinline__ double smc_block_sum(double v, double* sm)
{
    v = smc_wave_sum(v);
    __syncthreads();
    if (smc_lane() == 0) sm[smc_wave()] = v;
    __syncthreads();
    double r = sm[0];
#pragma unroll
    for (int w = 1; w < SMC_NWAVE; ++w) r = r + sm[w];
    return r;
}
// two sums with one LDS exchange; sm needs 2*SMC_NWAVE slots
__device__ __forceinline__ void smc_block_sum2(double& a, double& b, double* sm)
{
    a = smc_wave_sum(a);
    b = smc_wave_sum(b);
    __syncthreads();
    if (smc_lane() == 0) { sm[smc_wave()] = a; sm[SMC_NWAVE + smc_wave()] = b; }
    __syncthreads();
    double ra = sm[0], rb = sm[SMC_NWAVE];
#pragma unroll
    for (int w = 1; w < SMC_NWAVE; ++w) { ra = ra + sm[w]; rb = rb + sm[SMC_NWAVE + w]; }
    a = ra;
    b = rb;
}
__device__ __forceinline__ u64 smc_block_sum_u64(u64 v, u64* sm)
{
    v = smc_wave_sum_u64(v);
    __syncthreads();
    if (smc_lane() == 0) sm[smc_wave()] = v;
    __syncthreads();
    u64 r = sm[0];
#pragma unroll
    for (int w = 1; w < SMC_NWAVE; ++w) r = r + sm[w];
    return r;
}
// exclusive prefix over the workgroup's threads (thread order) + total
__device__ __forceinline__ u64 smc_block_exscan_u64(u64 v, u64* sm, u64& total)
{
    const u64 inc = smc_wave_scan_add_u64(v);
    __syncthreads();
    if (smc_lane() == 63) sm[smc_wave()] = inc;
    __syncthreads();
    u64 base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SMC_NWAVE; ++w) {
        if (w < smc_wave()) base += sm[w];
        tot += sm[w];
    }
    total = tot;
    return base + inc - v;
}
// the same for integer-valued doubles whose sums stay below 2^53 (exact, order-free)
__device__ __forceinline__ double smc_block_exscan_f64(double v, double* sm, double& total)
{
    const double inc = smc_wave_scan_add_f64(v);
    __syncthreads();
    if (smc_lane() == 63) sm[smc_wave()] = inc;
    __syncthreads();
    double base = 0.0, tot = 0.0;
#pragma unroll
    for (int w = 0; w < SMC_NWAVE; ++w) {
        if (w < smc_wave()) base += sm[w];
        tot += sm[w];
    }
    total = tot;
    return base + inc - v;
}
// the same for positive terms whose exclusive prefix must keep its RELATIVE accuracy (inc - v cancels where a thread's
// own term dominates everything before it): the left neighbour's inclusive sum, nothing subtracted
__device__ __forceinline__ double smc_block_exscan_pos_f64(double v, double* sm, double& total)
{
    const double inc = smc_wave_scan_add_f64(v);
    const double exc = smc_dpp_f64<SMC_DPP_WAVE_SHR1, 0xf, false>(inc);   // (lane 0: 0.0)
    __syncthreads();
    if (smc_lane() == 63) sm[smc_wave()] = inc;
    __syncthreads();
    double base = 0.0, tot = 0.0;
#pragma unroll
    for (int w = 0; w < SMC_NWAVE; ++w) {
        if (w < smc_wave()) base += sm[w];
        tot += sm[w];
    }
    total = tot;
    return base + exc;
}
// exclusive prefix of `v` over threads AND the workgroup sum of `extra`, with a
// single LDS exchange; sm needs 2*SMC_NWAVE slots
__device__ __forceinline__ u64 smc_block_exscan_plus_sum_u64(u64 v, u64 extra, u64* sm,
                                                             u64& total, u64& extra_sum)
{
    const u64 inc = smc_wave_scan_add_u64(v);
    const u64 es = smc_wave_sum_u64(extra);
    __syncthreads();
    if (smc_lane() == 63) { sm[smc_wave()] = inc; sm[SMC_NWAVE + smc_wave()] = es; }
    __syncthreads();
    u64 base = 0, tot = 0, et = 0;
#pragma unroll
    for (int w = 0; w < SMC_NWAVE; ++w) {
        if (w < smc_wave()) base += sm[w];
        tot += sm[w];
        et += sm[SMC_NWAVE + w];
    }
    total = tot;
    extra_sum = et;
    return base + inc - v;
}

// ---------------------------------------------------------------------------
// log-sum-exp accumulators: (m, s, ss) = (max, sum exp(lw-m), sum exp(2(lw-m)))
// ---------------------------------------------------------------------------
struct SmcLse {
    double m, s, ss;
};
__device__ __forceinline__ SmcLse smc_lse_empty()
{
    SmcLse a;
    a.m = -INFINITY; a.s = 0.0; a.ss = 0.0;
    return a;
}
// one exp per element (online form)
__device__ __forceinline__ void smc_lse_push(SmcLse& a, double lw)
{
    if (!(lw > -INFINITY)) return;             // -inf (and NaN, sanitised earlier) weigh 0
    const double d = lw - a.m;                 // +inf on the first element
    const double e = smc_exp_nonpos(-fabs(d));
    if (d > 0.0) {
        a.s = a.s * e + 1.0;
        a.ss = a.ss * (e * e) + 1.0;
        a.m = lw;
    } else {
        a.s += e;
        a.ss += e * e;
    }
}
// Combine the per-thread accumulators of a workgroup: every thread returns the
// workgroup's (m, s, ss).  sm: 2*SMC_NWAVE doubles.
__device__ __forceinline__ SmcLse smc_lse_block(SmcLse a, double* sm)
{
    const double m = smc_block_max(a.m, sm);
    double sc = 0.0;
    if (a.m > -INFINITY) sc = smc_exp_nonpos(a.m - m);
    SmcLse r;
    r.m = m;
    r.s = a.s * sc;
    r.ss = a.ss * (sc * sc);
    smc_block_sum2(r.s, r.ss, sm);
    return r;
}
// Reduce `n` per-workgroup partials (SoA: pm, ps, pss) to the global (m,s,ss).
// Called by every thread of a workgroup; all workgroups obtain identical bits.
// merge another accumulator (m2, s2, ss2) into a
__device__ __forceinline__ void smc_lse_merge(SmcLse& a, double m2, double s2, double ss2)
{
    if (!(m2 > -INFINITY)) return;
    const double m = smc_max2(a.m, m2);
    const double e1 = (a.m > -INFINITY) ? smc_exp_nonpos(a.m - m) : 0.0;
    const double e2 = smc_exp_nonpos(m2 - m);
    a.s = a.s * e1 + s2 * e2;
    a.ss = a.ss * (e1 * e1) + ss2 * (e2 * e2);
    a.m = m;
}
// Reduce `count` per-workgroup partials (SoA: pm, ps, pss; entries first,
// first+stride, ...) to one (m,s,ss).  Called by every thread of ONE workgroup.
// AGENT: the partials were published by other workgroups of the same launch
// (smc_st_agent) and are read with agent-scope loads; four entries of each
// array are requested back to back so a batch costs one memory round trip.
template <bool AGENT = false>
__device__ __forceinline__ SmcLse smc_lse_reduce_partials(const double* pm, const double* ps,
                                                          const double* pss, int count,
                                                          double* sm, int first = 0,
                                                          int stride = 1)
{
    SmcLse acc = smc_lse_empty();
    for (int base = 0; base < count; base += 4 * SMC_BLOCK) {
        double vm[4], vs[4], vq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = base + k * SMC_BLOCK + (int)threadIdx.x;
            const bool in = i < count;
            const int ii = first + (in ? i : 0) * stride;
            vm[k] = AGENT ? smc_ld_agent_f64(pm + ii) : pm[ii];
            vs[k] = AGENT ? smc_ld_agent_f64(ps + ii) : ps[ii];
            vq[k] = AGENT ? smc_ld_agent_f64(pss + ii) : pss[ii];
            if (!in) vm[k] = -INFINITY;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) smc_lse_merge(acc, vm[k], vs[k], vq[k]);
    }
    return smc_lse_block(acc, sm);
}

// smallest index i in [0, n) with T <= c[i]; n if none.  `c` non-decreasing.
__device__ __forceinline__ int smc_lower_bound_u64(const u64* c, int n, u64 T)
{
    int lo = 0, len = n;
    while (len > 0) {
        const int half = len >> 1;
        const bool less = c[lo + half] < T;
        lo = less ? lo + half + 1 : lo;
        len = less ? len - half - 1 : half;
    }
    return lo;
}
