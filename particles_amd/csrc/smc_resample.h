// smc_resample.h -- device machinery for inverse-CDF resampling
// (particles/resampling.py:484-509 inverse_cdf, :599-610 stratified/systematic,
// :540-558 multinomial), shared by the stand-alone kernels and the fused step.
//
// Formulation ("scatter by tile"): the N weights are cut into tiles of
// TILE = 256*IPT consecutive particles, one workgroup per tile.  With the
// exact Q62 CDF C_j (smc_device.h), tile b owns precisely the outputs
//     n in [ count(C_{j0-1}), count(C_{j1-1}) ),  count(C) = #{n : T_n <= C},
// a contiguous range because the sorted uniforms are monotone.  For
// systematic / stratified draws count() has a closed form, so a workgroup
// finds its outputs without any global search; the parent of each output is
// then a binary search in the tile's CDF held in LDS.  Per-tile totals Q_b
// come from a preceding pass; a workgroup's exclusive prefix is the (exact,
// order-free) integer sum of its predecessors' totals -- no spinning, no
// inter-workgroup hand-off inside a launch.
#pragma once
#include "smc_device.h"

enum { SMC_MULTINOMIAL_ = 0, SMC_STRATIFIED_ = 1, SMC_SYSTEMATIC_ = 2 };

// Where the sorted uniforms su_n of one resampling come from.
struct SmcSu {
    int scheme;        // SMC_*_
    i64 M;             // number of outputs
    double dM;         // (double)M
    const double* u;   // replay: systematic u[0]; stratified u[n]; multinomial su[n] (sorted)
                       // (multinomial always reads su from memory)
    double u_sys;      // systematic: the single uniform
    u64 seed;          // Philox (u == nullptr, stratified)
    u32 t, island;
    // stratified, Philox: the last pair of uniforms drawn (consecutive offspring share a call)
    mutable u64 c_pair = ~0ull, c_a = 0ull, c_b = 0ull;
    // multinomial, one-pass uniform_spacings (k_f_spacing_onepass): the integer prefix sums of the draws as 32-bit
    // prefixes inside their tile of 1024, zo, and the tiles' own prefixes, zE -- Z_n = zE[n >> 10] + zo[n] -- and
    // (double)Z_N: su_n = Z_n / Z_N, the quotient resampling.py:537 forms (null: `u` holds the quotients)
    const u32* zo = nullptr;
    const u64* zE = nullptr;
    double dall = 1.0;
    // M a power of two: 1 / M -- the division by M is then an exact scaling and x * rM the same double as x / M
    // (0: divide)
    double rM = 0.0;
};
__device__ __forceinline__ u64 smc_su_z(const SmcSu& s, const i64 n) { return s.zE[n >> 10] + (u64)s.zo[n]; }
__device__ __forceinline__ double smc_su_div(const SmcSu& s, const double y) { return s.rM != 0.0 ? y * s.rM : y / s.dM; }

// the nc-th uniform of the stratified draw
__device__ __forceinline__ double smc_strat_u(const SmcSu& s, u64 nc)
{
    if (s.u) return s.u[nc];
    const u64 pr = nc >> 1;
    if (pr != s.c_pair) {
        smc_philox((u32)pr, s.t, s.island, SMC_STREAM_RESAMPLE, s.seed, s.c_a, s.c_b);
        s.c_pair = pr;
    }
    return smc_u01_halfopen((nc & 1) ? s.c_b : s.c_a);
}

// su_n in fp64 exactly as the reference forms it: (u + n) / M with a true
// division (resampling.py:602, :609), or the n-th sorted uniform (:536-537).
__device__ __forceinline__ double smc_su_at(const SmcSu& s, i64 n)
{
    if (s.scheme == SMC_SYSTEMATIC_) return smc_su_div(s, s.u_sys + (double)n);
    if (s.scheme == SMC_STRATIFIED_) {
        double un;
        if (s.u) {
            un = s.u[n];
        } else {
            u64 a, b;
            smc_philox((u32)(n >> 1), s.t, s.island, SMC_STREAM_RESAMPLE, s.seed, a, b);
            un = smc_u01_halfopen((n & 1) ? b : a);
        }
        return smc_su_div(s, un + (double)n);
    }
    return s.zo ? (double)smc_su_z(s, n) / s.dall : s.u[n];
}

// Both members of the pair (2p, 2p+1) with one Philox call.
__device__ __forceinline__ void smc_su_pair(const SmcSu& s, i64 p, double& su0, double& su1)
{
    const i64 n0 = 2 * p, n1 = 2 * p + 1;
    if (s.scheme == SMC_STRATIFIED_ && !s.u) {
        u64 a, b;
        smc_philox((u32)p, s.t, s.island, SMC_STREAM_RESAMPLE, s.seed, a, b);
        su0 = smc_su_div(s, smc_u01_halfopen(a) + (double)n0);
        su1 = smc_su_div(s, smc_u01_halfopen(b) + (double)n1);
        return;
    }
    su0 = (n0 < s.M) ? smc_su_at(s, n0) : 2.0;
    su1 = (n1 < s.M) ? smc_su_at(s, n1) : 2.0;
}

// count(C) = #{ n in [0,M) : T(su_n) <= C } = first n whose threshold exceeds C.
// T(su_n) is non-decreasing in n, so this is a partition point; a closed-form
// guess is corrected with the exact predicate (and a bisection as the
// always-correct fallback).
__device__ __forceinline__ bool smc_su_le(const SmcSu& s, i64 n, u64 C)
{
    return smc_q62_t(smc_su_at(s, n)) <= C;
}
__device__ inline i64 smc_su_count_le(const SmcSu& s, u64 C)
{
    i64 lo = 0, hi = s.M;       // invariant: pred true on [0,lo), false on [hi,M)
    if (s.scheme != SMC_MULTINOMIAL_) {
        // su_n ~ (n + u)/M  =>  su_n <= x  <=>  n <= x*M - u
        const double x = (double)C * 0x1.0p-62;
        double r = x * s.dM - (s.scheme == SMC_SYSTEMATIC_ ? s.u_sys : 0.0);
        i64 g = (r < 0.0) ? 0 : (r >= s.dM ? s.M : (i64)r);
        // bounded local correction
        int it = 0;
        while (g < s.M && it < 4 && smc_su_le(s, g, C)) { ++g; ++it; }
        if (it < 4) {
            int jt = 0;
            while (g > 0 && jt < 4 && !smc_su_le(s, g - 1, C)) { --g; ++jt; }
            if (jt < 4) return g;
            hi = g;
        } else {
            lo = g;
        }
    }
    while (lo < hi) {
        const i64 mid = lo + ((hi - lo) >> 1);
        if (smc_su_le(s, mid, C)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Systematic draws with M = 2^k outputs: su_n = fl(u+n) / 2^k is an exact
// scaling, so T_n = ceil(fl(u+n) * 2^(62-k)) and, because n <= fl(u+n) <= n+1,
//     count(C) = nc + [T_nc <= C],   nc = floor(C / 2^(62-k))
// exactly: one shift, one add, one compare -- no division, no search.
__device__ __forceinline__ i64 smc_sys_count_pow2(u64 C, double u, int k, i64 M)
{
    const int sh = 62 - k;
    const u64 nc = C >> sh;
    if (nc >= (u64)M) return M;
    const double scale = __longlong_as_double((long long)(1023 + sh) << 52);   // 2^sh
    const u64 T = (u64)ceil((u + (double)(i64)nc) * scale);
    return (i64)nc + (T <= C ? 1 : 0);
}
// Integer-only shortcut of the same function.  fl(u+n) differs from u+n by at
// most half an ulp of a number below 2^k, i.e. T_n = n 2^sh + u 2^sh +- 2^(sh+k-53)
// = n 2^sh + Us +- 512 with Us = floor(u 2^sh).  So unless the fractional part
// of C (below 2^sh) is within 1024 of Us, the comparison T_nc <= C is decided by
// integers alone; the rare near-boundary case takes the exact path above.
__device__ __forceinline__ i64 smc_sys_count_pow2_fast(u64 C, double u, u64 Us, int k, i64 M)
{
    const int sh = 62 - k;
    const u64 nc = C >> sh;
    if (nc >= (u64)M) return M;
    const u64 frac = C & ((1ull << sh) - 1ull);
    if (frac >= Us + 1024ull) return (i64)nc + 1;
    if (frac + 1024ull <= Us) return (i64)nc;
    return smc_sys_count_pow2(C, u, k, M);
}

// floor(c * Q / t) for 0 <= c <= t <= 2^60, Q < 2^63 (the two-level CDF of the step loop: a
// position c of a tile's local integer CDF, total t, mapped onto the tile's share Q of the
// global 2^62 scale).  fp64 estimate, exact 128-bit remainder, one fp64 correction, +-1 fix-up.
__device__ __forceinline__ u64 smc_muldiv_floor(u64 c, u64 Q, u64 t)
{
    if (c == 0ull) return 0ull;
    if (c >= t) return Q;
    const double td = (double)t;
    u64 q = (u64)(((double)c / td) * (double)Q);               // within ~2^11 of the quotient
    const u64 lo1 = c * Q, hi1 = __umul64hi(c, Q);
    const u64 lo2 = q * t, hi2 = __umul64hi(q, t);
    const u64 rlo = lo1 - lo2;
    const i64 rhi = (i64)(hi1 - hi2 - (lo1 < lo2 ? 1ull : 0ull));    // R = c Q - q t, |R| < 2^72
    const double Rd = (double)rhi * 18446744073709551616.0 + (double)rlo;
    q += (u64)(i64)floor(Rd / td);
    i64 r = (i64)(c * Q - q * t);                              // now |r| < 2 t <= 2^61: 64 bits do
    if (r < 0) { q -= 1ull; r += (i64)t; }
    if (r < 0) { q -= 1ull; r += (i64)t; }
    if (r >= (i64)t) { q += 1ull; r -= (i64)t; }
    if (r >= (i64)t) { q += 1ull; r -= (i64)t; }
    return q;
}

// Stratified draws with M = 2^k outputs: su_n = fl(u_n + n) / 2^k lies in [n, n+1] / 2^k just
// like the systematic ones, so the same closed form holds with the n-th uniform:
//     count(C) = nc + [T_nc <= C],   nc = floor(C / 2^(62-k)),  T_nc = ceil(fl(u_nc + nc) 2^(62-k))
// (one Philox call, or one tape read, per evaluation).
__device__ __forceinline__ i64 smc_strat_count_pow2(u64 C, const SmcSu& s, int k, i64 M)
{
    const int sh = 62 - k;
    const u64 nc = C >> sh;
    if (nc >= (u64)M) return M;
    const double un = smc_strat_u(s, nc);
    const double scale = __longlong_as_double((long long)(1023 + sh) << 52);   // 2^sh
    const u64 T = (u64)ceil((un + (double)(i64)nc) * scale);
    return (i64)nc + (T <= C ? 1 : 0);
}

// ---------------------------------------------------------------------------
// count(C) for a C known only to within +-E (E a power of two, far below 2^(62-k)): decided
// whenever neither nc = floor(C / 2^(62-k)) nor the comparison T_nc <= C can change inside the
// error band; returns -1 otherwise (the caller then forms C exactly).
__device__ __forceinline__ i64 smc_count_pow2_band(u64 Ch, u64 E, const SmcSu& s, double u_sys,
                                                   u64 Us, int k, i64 M)
{
    const int sh = 62 - k;
    const u64 nc = Ch >> sh;
    const u64 frac = Ch & ((1ull << sh) - 1ull);
    if (frac < E || frac + E >= (1ull << sh)) return -1;        // nc itself could move
    if (nc >= (u64)M) return M;
    if (s.scheme == SMC_SYSTEMATIC_) {
        if (frac >= Us + 1024ull + E) return (i64)nc + 1;
        if (frac + 1024ull + E <= Us) return (i64)nc;
        return -1;
    }
    const double un = smc_strat_u(s, nc);                       // stratified: the nc-th uniform
    const double scale = __longlong_as_double((long long)(1023 + sh) << 52);   // 2^sh
    const u64 T = (u64)ceil((un + (double)(i64)nc) * scale);
    if (T + E <= Ch) return (i64)nc + 1;
    if (T > Ch + E) return (i64)nc;
    return -1;
}

// ---------------------------------------------------------------------------
// Per-tile CDF in LDS.
//   wq[i] (i < IPT): this thread's quantised weights, particles j0+tid*IPT+i
//   Qtiles[0..b):    totals of the preceding tiles (global memory)
// Writes sC[tid*IPT+i] = inclusive CDF of particle j0+tid*IPT+i and returns the
// tile's exclusive prefix; tile_total receives its total.  All threads call.
// ---------------------------------------------------------------------------
template <int IPT>
__device__ __forceinline__ u64 smc_tile_cdf(const u64 (&wq)[IPT], const u64* Qtiles, int b,
                                            u64* sC, u64* sm, u64& tile_total)
{
    u64 tsum = 0;
#pragma unroll
    for (int i = 0; i < IPT; ++i) tsum += wq[i];
    u64 pre = 0;
    for (int i = (int)threadIdx.x; i < b; i += SMC_BLOCK) pre += Qtiles[i];
    pre = smc_block_sum_u64(pre, sm);
    u64 run = pre + smc_block_exscan_u64(tsum, sm, tile_total);
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        run += wq[i];
        sC[threadIdx.x * IPT + i] = run;
    }
    __syncthreads();
    return pre;
}

// The same count for sorted uniforms that sit in memory (multinomial), by a whole wavefront:
// a 64-ary search -- 64 probes per round, log64(M) dependent loads instead of log2(M) (a
// bisection by one lane costs 22 round trips to memory at M = 2^22, per tile).
// Every lane of the calling wave must take part; all of them get the result.
__device__ __forceinline__ i64 smc_su_count_le_wave(const SmcSu& s, u64 C)
{
    i64 lo = 0, hi = s.M;                 // pred true on [0, lo), false on [hi, M)
    const int lane = smc_lane();
    while (lo < hi) {
        const i64 width = hi - lo;
        const i64 stride = (width + 63) >> 6;
        const i64 p = lo + (i64)lane * stride;
        const bool in = p < hi;
        const bool le = in && (smc_q62_t(s.u[p]) <= C);
        const int cnt = (int)smc_wave_sum_u64(le ? 1ull : 0ull);      // monotone: the first cnt probes
        const int nprobe = (int)((width + stride - 1) / stride);
        const i64 nlo = cnt > 0 ? lo + (i64)(cnt - 1) * stride + 1 : lo;
        const i64 nhi = cnt < nprobe ? lo + (i64)cnt * stride : hi;
        lo = nlo;
        hi = nhi;
    }
    return lo;
}

// Output range [n_lo, n_hi) of tile b (every thread gets the same values).  BS = threads of
// the workgroup (a one-wave workgroup does both ends with its only wave).
template <int BS = SMC_BLOCK>
__device__ __forceinline__ void smc_tile_outputs(const SmcSu& su, int b, int ntiles, u64 pre,
                                                 u64 tile_total, i64* sn, i64& n_lo, i64& n_hi)
{
    constexpr int W1 = BS > 64 ? 1 : 0;           // the wave / thread that takes the upper end
    if (su.scheme == SMC_MULTINOMIAL_) {          // one wave per end, cooperatively
        if (smc_wave() == 0) {
            const i64 v = (b == 0) ? 0 : smc_su_count_le_wave(su, pre);
            if (smc_lane() == 0) sn[0] = v;
        }
        if (smc_wave() == W1) {
            const i64 v = (b == ntiles - 1) ? su.M : smc_su_count_le_wave(su, pre + tile_total);
            if (smc_lane() == 0) sn[1] = v;
        }
    } else {
        if (threadIdx.x == 0) sn[0] = (b == 0) ? 0 : smc_su_count_le(su, pre);
        if (threadIdx.x == 64 * W1) sn[1] = (b == ntiles - 1) ? su.M : smc_su_count_le(su, pre + tile_total);
    }
    __syncthreads();
    n_lo = sn[0];
    n_hi = sn[1];
}


// ---------------------------------------------------------------------------
// STRICT mode: the reference's own inverse CDF, literally (resampling.py:500-509):
//     s = W[0]; j = 0;  for n: while su[n] > s: j += 1; s += W[j];  A[n] = j
// i.e. A[n] = the first j with su[n] <= S_j, S_j = ((W_0 + W_1) + W_2) + ... accumulated LEFT TO
// RIGHT in fp64.  The order of the additions is the result, so the prefix cannot be formed in
// parallel: one wavefront walks the weights -- its lanes fetch 64 consecutive weights at a time
// (coalesced), the 64 additions of a chunk are a dependent chain on the scalar value of the running
// sum (v_readlane feeds it) -- about 10 cycles per weight, 4-5 ms at N = 2^20.  The searches
// against S are then independent.  Opt-in (SMC_FLAG_STRICT_ANCESTORS / smc_inverse_cdf_strict): the
// default exact integer CDFs give the same ancestors except where su[n] lies within rounding
// distance of a step of S, at 1/500 of the cost.
// ---------------------------------------------------------------------------
// first j in [0, n) with x <= S[j]; n - 1 if none (the reference would run off the end of W there:
// IndexError in Python, an unchecked read under numba)
__device__ __forceinline__ i64 smc_first_ge(const double* S, const i64 n, const double x)
{
    i64 lo = 0, hi = n;
    while (lo < hi) {
        const i64 mid = lo + ((hi - lo) >> 1);
        if (S[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo < n ? lo : n - 1;
}
