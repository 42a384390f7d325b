// smc_filter_kernels.h -- device side of the fused SMC step loop.
//
// Replaces, for the closed model family of the hot-path configs, the body of
// particles.SMC.__next__ (particles/core.py:369-383):
//     setup_auxiliary_weights / resample_move  (core.py:307-337)
//     reweight_particles                        (core.py:323-324)
//     compute_summaries                         (core.py:351-359)
// for `n_islands` independent filters advancing in lock step.
//
// Two kernels per time step, no host round trip, no in-kernel spinning:
//
//   k_prepare(t): every workgroup reduces the per-workgroup log-sum-exp
//       partials of step t-1 to (max, sum, sum of squares) -> ESS, log-mean
//       weight and the resample decision of step t (core.py:181-183); if
//       resampling, it converts its tile of log-weights to Q62 fixed point,
//       q_i = rint(exp(lw_i-m)/s * 2^62), stores them and the tile total.
//   k_move(t): one workgroup per tile of 1024 consecutive parents.  From the q
//       of its tile and the totals of the preceding tiles it knows the exact CDF
//       of its parents, hence the contiguous range of offspring it owns
//       (smc_resample.h).  Offspring are produced 4 per thread per pass:
//         systematic, N a power of two: closed-form first-offspring index per
//           parent, scattered into LDS and expanded by a max-scan (no search);
//         otherwise: per-offspring binary search in the tile's CDF in LDS;
//       then gather of the parent state from LDS, propagation
//       x = loc(xp)+scale*z with a counted Philox normal (or a replayed draw),
//       the weight increment log G, coalesced 32-byte stores of (A, X, lw) and
//       the online log-sum-exp partial of the new weights.
//
// The time index lives in device memory (ctl[0]/ctl[1], ping-ponged between the
// two kernels) so the same launches -- or one hipGraph holding many of them --
// serve every step.
//
// HBM traffic per particle-step on a resampling step (d = 1):
//   k_prepare: read lw (8), write q (8);  k_move: read q, X (16), write A, X, lw (24)
// = 56 B, the algorithmic figure of SURVEY 8d (W itself is never materialised).
#pragma once
#include "smc_internal.h"
#include "smc_resample.h"

#define F_IPT 4
#define F_TILE (SMC_BLOCK * F_IPT)
#define F_PASS (SMC_BLOCK * 4)      /* offspring per pass: 4 per thread */
#define SUMM_STRIDE 8   /* ESS, log_mean, loglt, logLt, rs_flag, m, s, - */
#define PARAM_STRIDE 16
#define INFO_STRIDE 8     /* per-island step record written by k_prepare: t, flag, y_t, m, 1/s */

struct FArgs {
    i64 N, T;
    int ntiles, n_islands, scheme, rng_mode, island_offset;
    int log2N;             // k if N == 2^k, else -1
    double ess_thresh;
    u64 seed;
    double *X0, *X1, *lw0, *lw1;
    i64* A;
    u64* q;                // (n_islands, N) Q62 weights of the parents
    u64* Q;                // (n_islands, ntiles) tile totals of q
    double *pm, *ps, *pss;
    double* summ;          // (n_islands, T+1, SUMM_STRIDE)
    const double* params;  // (n_islands, PARAM_STRIDE)
    const double* y;       // (T,)
    i64* ctl;              // [0] = time index of the next step (advanced by k_move)
    double* info;          // (n_islands, INFO_STRIDE) record of the step being run
    const double* zt;      // replay normals (T, n_islands, N) or null
    const double* ut;      // replay uniforms (T, n_islands, K) or null
    i64 ut_stride;         // K
    double* su;            // multinomial, Philox mode: (n_islands, N) sorted uniforms
    u64* E;                // multinomial, Philox mode: spacing tile sums (n_islands, ntiles1)
    int ntiles1;
    double spacing_scale;
};

// ---------------------------------------------------------------------------
// model family
// ---------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ double m_trans_loc(const double* p, double xp)
{
    if (KIND == SMC_MODEL_LINGAUSS) return p[0] * xp;           // kalman.py:430-431
    return p[4] + p[1] * xp;                                    // state_space_models.py:465-470
}
template <int KIND>
__device__ __forceinline__ double m_trans_scale(const double* p)
{
    return (KIND == SMC_MODEL_LINGAUSS) ? p[1] : p[2];
}
template <int KIND>
__device__ __forceinline__ double m_init_loc(const double* p)
{
    return (KIND == SMC_MODEL_LINGAUSS) ? 0.0 : p[0];           // kalman.py:427 ; ssm.py:462
}
// log p(y_t | x_t) as scipy.stats.norm.logpdf evaluates it
template <int KIND>
__device__ __forceinline__ double m_obs_logpdf(const double* p, double y, double x)
{
    if (KIND == SMC_MODEL_LINGAUSS) {                           // kalman.py:433-434
        const double v = (y - x) / p[2];
        return -(v * v) / 2.0 - SMC_C_NORM - p[4];
    }
    const double sc = exp(0.5 * x);                             // ssm.py:472-473
    const double v = (y - 0.0) / sc;
    return -(v * v) / 2.0 - SMC_C_NORM - log(sc);
}
__device__ __forceinline__ double m_norm_logpdf(double x, double loc, double scale, double lscale)
{
    const double v = (x - loc) / scale;
    return -(v * v) / 2.0 - SMC_C_NORM - lscale;
}

// one particle of one step: returns the new state, writes the weight increment
template <int KIND, int FK>
__device__ __forceinline__ double m_step(const double* p, bool first, double y, double xp,
                                         double z, double& inc)
{
    if (FK == SMC_FK_BOOTSTRAP) {
        const double x = first ? m_init_loc<KIND>(p) + p[3] * z
                               : m_trans_loc<KIND>(p, xp) + m_trans_scale<KIND>(p) * z;
        inc = m_obs_logpdf<KIND>(p, y, x);
        return x;
    }
    // guided filter with LinearGauss' optimal proposal (kalman.py:436-446,
    // state_space_models.py:374-392)
    if (first) {
        const double mu = p[12] * (y / p[8]);
        const double x = mu + p[13] * z;
        inc = (m_norm_logpdf(x, 0.0, p[3], p[6]) + m_obs_logpdf<KIND>(p, y, x))
              - m_norm_logpdf(x, mu, p[13], p[14]);
        return x;
    }
    const double mu = p[9] * (p[0] * xp / p[7] + y / p[8]);
    const double x = mu + p[10] * z;
    inc = (m_norm_logpdf(x, p[0] * xp, p[1], p[5]) + m_obs_logpdf<KIND>(p, y, x))
          - m_norm_logpdf(x, mu, p[10], p[11]);
    return x;
}

// W_i as the device defines it: exp(lw_i - m) * (1/s)   (resampling.py:222,225)
__device__ __forceinline__ double f_weight(double lw, double m, double rs)
{
    return smc_exp_nonpos(lw - m) * rs;
}

// ---------------------------------------------------------------------------
// 4 consecutive elements per thread, as two 16-byte accesses when possible
// ---------------------------------------------------------------------------
struct alignas(16) F2u { u64 a, b; };
struct alignas(16) F2d { double a, b; };

template <class T, class T2>
__device__ __forceinline__ void f_load4(const T* p, i64 j, i64 N, bool vec, T fill, T (&o)[4])
{
    if (vec && j + 3 < N) {
        const T2 v0 = *reinterpret_cast<const T2*>(p + j);
        const T2 v1 = *reinterpret_cast<const T2*>(p + j + 2);
        o[0] = v0.a; o[1] = v0.b; o[2] = v1.a; o[3] = v1.b;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (j + i < N) ? p[j + i] : fill;
    }
}
template <class T, class T2>
__device__ __forceinline__ void f_store4(T* p, i64 n, bool full_vec, const bool (&ok)[4],
                                         const T (&v)[4])
{
    if (full_vec) {
        T2 v0, v1;
        v0.a = v[0]; v0.b = v[1]; v1.a = v[2]; v1.b = v[3];
        *reinterpret_cast<T2*>(p + n) = v0;
        *reinterpret_cast<T2*>(p + n + 2) = v1;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ok[i]) p[n + i] = v[i];
    }
}

// ---------------------------------------------------------------------------
// k_prepare
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(SMC_BLOCK)
k_prepare(FArgs a, int finalize_only)
{
    __shared__ double smd[SMC_SM];
    __shared__ u64 smu[SMC_SM];
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y;
    const i64 t = a.ctl[0];
    double* info = a.info + (i64)isl * INFO_STRIDE;
    if (t == 0) {                     // nothing to finalise; step 0 never resamples
        if (b == 0 && threadIdx.x == 0) {
            info[0] = 0.0; info[1] = 0.0; info[2] = a.y[0]; info[3] = 0.0; info[4] = 0.0;
        }
        return;
    }
    const i64 tp = t - 1;
    // issue this tile's log-weight loads first: their latency hides behind the
    // reduction of the partials below (they are only used when resampling)
    const double* lw = ((tp & 1) ? a.lw1 : a.lw0) + (i64)isl * a.N;
    const bool vec = (a.N & 3) == 0;
    const i64 j0 = (i64)b * F_TILE + (i64)threadIdx.x * F_IPT;
    double l4[4];
    if (t < a.T && !finalize_only) f_load4<double, F2d>(lw, j0, a.N, vec, -INFINITY, l4);
    const double* pm = a.pm + (i64)isl * a.ntiles;
    const double* ps = a.ps + (i64)isl * a.ntiles;
    const double* pss = a.pss + (i64)isl * a.ntiles;
    const SmcLse r = smc_lse_reduce_partials(pm, ps, pss, a.ntiles, smd);
    const bool bad = !(r.m > -INFINITY) || !(r.m < INFINITY);
    const double ess = bad ? NAN : (r.s * r.s) / r.ss;                  // resampling.py:226
    const double log_mean = bad ? NAN : r.m + log(r.s / (double)a.N);   // resampling.py:224
    const bool flag = (t < a.T) && (ess < a.ess_thresh);                // core.py:181-183
    const double rs = 1.0 / r.s;
    if (b == 0 && threadIdx.x == 0) {
        double* row = a.summ + ((i64)isl * (a.T + 1) + tp) * SUMM_STRIDE;
        double loglt, logLt;                                            // core.py:355-359
        if (tp == 0 || row[4] != 0.0) loglt = log_mean;
        else loglt = log_mean - row[1 - SUMM_STRIDE];
        logLt = (tp == 0 ? 0.0 : row[3 - SUMM_STRIDE]) + loglt;
        row[0] = ess;
        row[1] = log_mean;
        row[2] = loglt;
        row[3] = logLt;
        row[5] = r.m;
        row[6] = bad ? NAN : rs;
        if (t < a.T) {
            row[SUMM_STRIDE + 4] = flag ? 1.0 : 0.0;
            // everything k_move(t) needs, in one 64-byte record
            info[0] = (double)t; info[1] = flag ? 1.0 : 0.0; info[2] = a.y[t];
            info[3] = r.m; info[4] = rs;
        } else {
            info[0] = (double)t;
        }
    }
    if (t >= a.T || finalize_only || !flag) return;
    // Q62 weights of step t-1's particles (the parents of step t) + tile total
    u64* q = a.q + (i64)isl * a.N;
    u64 q4[4], s = 0;
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ok[i] = j0 + i < a.N;
        q4[i] = ok[i] ? smc_q62_w(f_weight(l4[i], r.m, rs)) : 0ull;
        s += q4[i];
    }
    f_store4<u64, F2u>(q, j0, vec && ok[3], ok, q4);
    s = smc_block_sum_u64(s, smu);
    if (threadIdx.x == 0) a.Q[(i64)isl * a.ntiles + b] = s;
}

// ---------------------------------------------------------------------------
// multinomial, Philox mode: sorted uniforms by exponential spacings
// (resampling.py:512-537), batched over islands, skipped when not resampling
// ---------------------------------------------------------------------------
__device__ __forceinline__ u64 f_spacing_q(const FArgs& a, u32 t, u32 gisl, i64 n)
{
    u64 x, y;
    smc_philox((u32)(n >> 1), t, gisl, SMC_STREAM_SPACINGS, a.seed, x, y);
    return (u64)rint(-log(smc_u01_open((n & 1) ? y : x)) * a.spacing_scale);
}

__global__ void __launch_bounds__(SMC_BLOCK)
k_f_spacing_sums(FArgs a)
{
    __shared__ u64 smu[SMC_SM];
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y;
    const double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)info[0];
    if (t >= a.T || t == 0 || info[1] == 0.0) return;
    const i64 n0 = (i64)b * F_TILE + (i64)threadIdx.x * F_IPT;
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < F_IPT; ++i)
        if (n0 + i <= a.N) s += f_spacing_q(a, (u32)t, (u32)(a.island_offset + isl), n0 + i);
    s = smc_block_sum_u64(s, smu);
    if (threadIdx.x == 0) a.E[(i64)isl * a.ntiles1 + b] = s;
}

__global__ void __launch_bounds__(SMC_BLOCK)
k_f_spacing_write(FArgs a)
{
    __shared__ u64 smu[SMC_SM];
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y;
    const double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)info[0];
    if (t >= a.T || t == 0 || info[1] == 0.0) return;
    const u64* E = a.E + (i64)isl * a.ntiles1;
    const i64 n0 = (i64)b * F_TILE + (i64)threadIdx.x * F_IPT;
    u64 q[F_IPT], tsum = 0;
#pragma unroll
    for (int i = 0; i < F_IPT; ++i) {
        q[i] = (n0 + i <= a.N) ? f_spacing_q(a, (u32)t, (u32)(a.island_offset + isl), n0 + i) : 0ull;
        tsum += q[i];
    }
    u64 pre = 0, all = 0;
    for (int i = (int)threadIdx.x; i < a.ntiles1; i += SMC_BLOCK) {
        const u64 e = E[i];
        all += e;
        if (i < b) pre += e;
    }
    all = smc_block_sum_u64(all, smu);
    u64 tot, pre_sum;
    u64 run = smc_block_exscan_plus_sum_u64(tsum, pre, smu, tot, pre_sum);
    run += pre_sum;
    const double dall = (double)all;
    double* su = a.su + (i64)isl * a.N;
#pragma unroll
    for (int i = 0; i < F_IPT; ++i) {
        run += q[i];
        if (n0 + i < a.N) su[n0 + i] = (double)run / dall;
    }
}

// ---------------------------------------------------------------------------
// k_move
// ---------------------------------------------------------------------------
template <int KIND, int FK>
__global__ void __launch_bounds__(SMC_BLOCK)
k_move(FArgs a)
{
    __shared__ double sX[F_TILE];      // states of the tile's parents
    __shared__ u64 sC[F_TILE];         // inclusive CDF of the tile (search path)
    __shared__ __attribute__((aligned(16))) u32 sP[F_PASS];   // parent of each offspring of a
                                                               // pass (scatter path)
    __shared__ u64 smu[SMC_SM];
    __shared__ double smd[SMC_SM];
    __shared__ i64 sn[2];
    __shared__ u32 smx[SMC_NWAVE];
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y;
    const int tid = (int)threadIdx.x;
    const double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)info[0];
    if (t >= a.T) return;
    if (b == 0 && isl == 0 && tid == 0) a.ctl[0] = t + 1;

    const i64 N = a.N;
    const double* p = a.params + (i64)isl * PARAM_STRIDE;
    const double yt = info[2];
    const u32 gisl = (u32)(a.island_offset + isl);
    const int cur = (int)(t & 1);
    double* Xn = (cur ? a.X1 : a.X0) + (i64)isl * N;
    const double* Xo = (cur ? a.X0 : a.X1) + (i64)isl * N;
    double* lwn = (cur ? a.lw1 : a.lw0) + (i64)isl * N;
    const double* lwo = (cur ? a.lw0 : a.lw1) + (i64)isl * N;
    i64* A = a.A + (i64)isl * N;
    const double* zt = a.zt ? a.zt + ((i64)t * a.n_islands + isl) * N : nullptr;
    const bool first = (t == 0);
    const bool resample = !first && info[1] != 0.0;
    const bool vec = (N & 3) == 0;
    const i64 j0 = (i64)b * F_TILE;

    SmcLse acc = smc_lse_empty();
    i64 n_lo = j0, n_hi = (j0 + F_TILE < N) ? j0 + F_TILE : N;   // element-wise: own tile
    i64 ns[F_IPT + 1];                                           // scatter path
    bool scatter = false;
    SmcSu su;

    if (resample) {
        // ---- the tile's parents: q, states, exact CDF
        u64 q4[4];
        double x4[4];
        const i64 jt = j0 + (i64)tid * F_IPT;
        f_load4<u64, F2u>(a.q + (i64)isl * N, jt, N, vec, 0ull, q4);
        f_load4<double, F2d>(Xo, jt, N, vec, 0.0, x4);
#pragma unroll
        for (int i = 0; i < 4; ++i) sX[tid * F_IPT + i] = x4[i];
        const u64* Qt = a.Q + (i64)isl * a.ntiles;
        u64 pre_part = 0;
        for (int i = tid; i < b; i += SMC_BLOCK) pre_part += Qt[i];
        const u64 tsum = q4[0] + q4[1] + q4[2] + q4[3];
        u64 total, pre;
        u64 cex = smc_block_exscan_plus_sum_u64(tsum, pre_part, smu, total, pre);
        cex += pre;                                   // exclusive CDF of this thread's 1st parent
        su.scheme = a.scheme;
        su.M = N;
        su.dM = (double)N;
        su.u = a.ut ? a.ut + ((i64)t * a.n_islands + isl) * a.ut_stride
                    : (a.scheme == SMC_MULTINOMIAL_ ? a.su + (i64)isl * N : nullptr);
        su.u_sys = 0.0;
        su.seed = a.seed;
        su.t = (u32)t;
        su.island = gisl;
        if (a.scheme == SMC_SYSTEMATIC_) {
            if (su.u) {
                su.u_sys = su.u[0];
            } else {
                u64 x0, x1;
                smc_philox(0u, su.t, su.island, SMC_STREAM_RESAMPLE, su.seed, x0, x1);
                su.u_sys = smc_u01_halfopen(x0);
            }
        }
        scatter = (a.scheme == SMC_SYSTEMATIC_) && a.log2N >= 0;
        if (scatter) {
            // first offspring of each parent, closed form (smc_resample.h)
            u64 c = cex;
            const u64 Us = (u64)(su.u_sys *
                                 __longlong_as_double((long long)(1023 + 62 - a.log2N) << 52));
#pragma unroll
            for (int i = 0; i <= F_IPT; ++i) {
                const i64 j = jt + i;
                ns[i] = (j == 0) ? 0
                                 : (j >= N ? N
                                           : smc_sys_count_pow2_fast(c, su.u_sys, Us, a.log2N, N));
                if (i < F_IPT) c += q4[i];
            }
            if (tid == 0) sn[0] = ns[0];
            if (tid == SMC_BLOCK - 1) sn[1] = ns[F_IPT];
            __syncthreads();
            n_lo = sn[0];
            n_hi = sn[1];
        } else {
            u64 c = cex;
#pragma unroll
            for (int i = 0; i < F_IPT; ++i) {
                c += q4[i];
                sC[tid * F_IPT + i] = c;
            }
            __syncthreads();
            smc_tile_outputs(su, b, a.ntiles, pre, total, sn, n_lo, n_hi);
        }
    }
    const int nvalid = (int)((N - j0 < F_TILE) ? (N - j0) : F_TILE);

    // ---- offspring, 4 consecutive ones per thread per pass
    for (i64 pb = n_lo & ~(i64)3; pb < n_hi; pb += F_PASS) {
        const i64 n0 = pb + (i64)tid * 4;
        bool ok[4];
        int par[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ok[i] = (n0 + i >= n_lo) && (n0 + i < n_hi);
            par[i] = tid * 4 + i;                      // element-wise: the particle itself
        }
        if (resample && scatter) {
            __syncthreads();                           // previous pass has read sP
            *reinterpret_cast<uint4*>(&sP[tid * 4]) = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < F_IPT; ++i) {
                const i64 lo = ns[i] > pb ? ns[i] : pb;
                const i64 hi = ns[i + 1] < pb + F_PASS ? ns[i + 1] : pb + F_PASS;
                if (lo < hi) sP[lo - pb] = (u32)(tid * F_IPT + i);
            }
            __syncthreads();
            const uint4 v = *reinterpret_cast<const uint4*>(&sP[tid * 4]);
            const u32 m0 = v.x, m1 = m0 > v.y ? m0 : v.y, m2 = m1 > v.z ? m1 : v.z,
                      m3 = m2 > v.w ? m2 : v.w;
            const u32 inc = smc_wave_scan_u32(m3, SmcOpMaxU32());
            u32 ex = smc_mov_dpp<SMC_DPP_WAVE_SHR1>(inc);
            if (smc_lane() == 0) ex = 0u;
            if (smc_lane() == 63) smx[smc_wave()] = inc;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < SMC_NWAVE - 1; ++w)
                if (w < smc_wave()) ex = ex > smx[w] ? ex : smx[w];
            par[0] = (int)(m0 > ex ? m0 : ex);
            par[1] = (int)(m1 > ex ? m1 : ex);
            par[2] = (int)(m2 > ex ? m2 : ex);
            par[3] = (int)(m3 > ex ? m3 : ex);
        } else if (resample) {
            double s4[4];
            smc_su_pair(su, n0 >> 1, s4[0], s4[1]);
            smc_su_pair(su, (n0 >> 1) + 1, s4[2], s4[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int jl = ok[i] ? smc_lower_bound_u64(sC, F_TILE, smc_q62_t(s4[i])) : 0;
                par[i] = jl < nvalid ? jl : nvalid - 1;
            }
        }
        if (!(ok[0] || ok[1] || ok[2] || ok[3])) continue;

        double z4[4];
        if (zt) {
            f_load4<double, F2d>(zt, n0, N, vec, 0.0, z4);
        } else {
#if defined(ABL_NO_RNG)
            z4[0] = 0.1 * (double)(n0 & 15); z4[1] = -z4[0]; z4[2] = 0.3; z4[3] = -0.7;
#elif defined(ABL_NO_BM)
            { u64 x0, x1, x2, x3;
              smc_philox((u32)(n0 >> 1), (u32)t, gisl, 0u, a.seed, x0, x1);
              smc_philox((u32)(n0 >> 1) + 1u, (u32)t, gisl, 0u, a.seed, x2, x3);
              z4[0] = smc_u01_open(x0) - 0.5; z4[1] = smc_u01_open(x1) - 0.5;
              z4[2] = smc_u01_open(x2) - 0.5; z4[3] = smc_u01_open(x3) - 0.5; }
#else
            smc_normal_pair(a.seed, (u32)(n0 >> 1), (u32)t, gisl, SMC_STREAM_NORMAL, z4[0], z4[1]);
            smc_normal_pair(a.seed, (u32)(n0 >> 1) + 1u, (u32)t, gisl, SMC_STREAM_NORMAL, z4[2],
                            z4[3]);
#endif
        }
        double xo4[4] = {0.0, 0.0, 0.0, 0.0}, lo4[4] = {0.0, 0.0, 0.0, 0.0};
        if (!resample && !first) {
            f_load4<double, F2d>(Xo, n0, N, vec, 0.0, xo4);
            f_load4<double, F2d>(lwo, n0, N, vec, 0.0, lo4);
        }
        double xn4[4], lw4[4];
        i64 a4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double xp = resample ? sX[par[i]] : (first ? 0.0 : xo4[i]);
            double inc;
            xn4[i] = m_step<KIND, FK>(p, first, yt, xp, z4[i], inc);
            double lw = (resample || first) ? inc : lo4[i] + inc;   // resampling.py:241-244
            if (lw != lw) lw = -INFINITY;                            // resampling.py:220
            lw4[i] = lw;
            a4[i] = resample ? j0 + par[i] : n0 + i;                 // core.py:329 / :335
#if defined(ABL_NO_LSE)
            if (ok[i]) { acc.m = fmax(acc.m, lw); acc.s += 1.0; acc.ss += 1.0; }
#else
            if (ok[i]) smc_lse_push(acc, lw);
#endif
        }
        const bool full = vec && ok[0] && ok[3];
#if defined(ABL_NO_STORE)
        if (lw4[0] == 1.2345) f_store4<double, F2d>(Xn, n0, full, ok, xn4);
        continue;
#endif
        f_store4<double, F2d>(Xn, n0, full, ok, xn4);
        f_store4<double, F2d>(lwn, n0, full, ok, lw4);
        if (!first) f_store4<i64, F2u>(A, n0, full, ok, a4);
    }
    const SmcLse r = smc_lse_block(acc, smd);
    if (tid == 0) {
        const i64 o = (i64)isl * a.ntiles + b;
        a.pm[o] = r.m;
        a.ps[o] = r.s;
        a.pss[o] = r.ss;
    }
}

// W = exp(lw - m)/s for one island (SMC.W)
__global__ void __launch_bounds__(SMC_BLOCK)
k_f_write_W(const double* lw, i64 N, const double* row, double* W)
{
    const double m = row[5], rs = row[6];
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < N) W[i] = f_weight(lw[i], m, rs);
}

__global__ void __launch_bounds__(SMC_BLOCK)
k_f_gather1(const double* X, const i64* A, i64 N, double* Xp)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < N) Xp[i] = X[A[i]];
}
