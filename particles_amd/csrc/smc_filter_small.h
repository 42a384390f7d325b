// smc_filter_small.h -- the whole T-loop of a small particle filter in ONE launch.
//
// With N <= 1024 particles a filter is a single tile: one workgroup holds its particles in
// registers (4 per thread), resamples through LDS and needs no other workgroup -- so there
// is nothing to wait for between steps and the loop over time stays inside the kernel.
// grid = (1, n_islands): every island is one persistent workgroup (the regime of SMC^2 and
// PMMH, smc_samplers.py:1038-1167: many filters of 10^2..10^3 particles), and a step costs a
// few microseconds instead of two kernel launches.
//
// Same arithmetic, in the same order, as k_ancestors<true> + k_propagate (f_tile_offspring,
// m_step, the workgroup log-sum-exp): results are bit-identical to the multi-kernel path
// (tests: check_small_filter_equals_general).  State is written back every step (X, lw, A,
// summary row) -- a few KB -- so every API call behaves as after the general path.
#pragma once
#include "smc_filter_kernels.h"

// BS = 256 threads for N <= 1024, one wavefront (BS = 64) for N <= 256: then the workgroup
// barriers are free and 4x as many filters are resident.  The workgroup collectives of
// smc_device.h combine SMC_NWAVE per-wave slots; with fewer waves the unused slots hold the
// neutral element (set once here, never written again), so the same routines -- and the same
// association order, hence the same bits -- serve both sizes.
template <int KIND, int FK, int BS>
__global__ void __launch_bounds__(BS)
k_filter_small(const FArgs av, const int nsteps)
{
    const FArgs& a = av;
    __shared__ u64 sC[BS * F_IPT];
    __shared__ __attribute__((aligned(16))) u32 sP[BS * 4];
    __shared__ double sX[BS * F_IPT];
    __shared__ u64 smu[SMC_SM];
    __shared__ double smd[SMC_SM];          // sums
    __shared__ double smm[SMC_SM];          // maxima
    __shared__ i64 sn[2];
    __shared__ u32 smx[SMC_NWAVE];
    SMC_NTAB_LDS(s_ntab);
    const int isl = (int)blockIdx.y;
    const int tid = (int)threadIdx.x;
    if (tid < SMC_SM) { smu[tid] = 0ull; smd[tid] = 0.0; smm[tid] = -INFINITY; }
    smc_ntab_stage<BS>(s_ntab, tid);
    // the island's model constants in LDS (as in k_propagate: read through the argument block's pointer they are global
    // loads repeated behind every step's stores, each with a wait that also drains those stores)
    __shared__ double s_par[PARAM_STRIDE];
    static_assert(BS >= PARAM_STRIDE, "one constant per thread");
    if (tid < PARAM_STRIDE) s_par[tid] = smc_ldg(a.params + (i64)isl * PARAM_STRIDE + tid);
    __syncthreads();
    const i64 N = a.N;
    double* info = a.info + (i64)isl * INFO_STRIDE;
    const u32 gisl = (u32)(a.island_offset + isl);
    const double* p = s_par;
    const i64 jt = (i64)tid * F_IPT;                       // this thread's particles jt..jt+3
    const bool vec = (N & 3) == 0;

    i64 t = (i64)smc_uniform(smc_ldg(info));
    if (t >= a.T) return;                                  // done, or frozen by k_theta_update
    bool resample = smc_uniform(smc_ldg(info + 1)) != 0.0;
    double m = smc_uniform(smc_ldg(info + 3)), rs = smc_uniform(smc_ldg(info + 4));
    // APF (core.py:299-313): resampling runs on the AUXILIARY weights lw + logeta; (m, rs) then
    // normalise those, and cconst = log_mean_exp(logeta, W) is what the weights are reset to
    constexpr bool APF = f_is_apf(FK);
    double cconst = APF ? smc_uniform(smc_ldg(info + 6)) : 0.0;
    double prev_log_mean = 0.0, prev_logLt = 0.0;          // of step t-1 (core.py:355-359)
    if (t > 0) {
        const double* prow = a.summ + ((i64)isl * (a.T + 1) + (t - 1)) * SUMM_STRIDE;
        prev_log_mean = smc_uniform(smc_ldg(prow + 1));
        prev_logLt = smc_uniform(smc_ldg(prow + 3));
    }
    double y_next = (t < a.T) ? a.y[t * a.dy] : 0.0;
    double aux_next = (m_has_aux<KIND>() && a.aux && t < a.T) ? a.aux[t] : 0.0;
    double x[4], lw[4];
    if (t > 0) {                                           // continue a run
        f_load4<double>(f_X(a, t - 1) + (i64)isl * N, jt, N, vec, 0.0, x);
        f_load4<double>(f_lw(a, t - 1) + (i64)isl * N, jt, N, vec, -INFINITY, lw);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { x[k] = 0.0; lw[k] = -INFINITY; }
    }

    for (int step = 0; step < nsteps && t < a.T; ++step, ++t) {
        const bool first = (t == 0);
        const bool rsp = !first && resample;
        const double yt = y_next, aux = aux_next;
        if (t + 1 < a.T) {                                 // requested a whole step ahead
            y_next = a.y[(t + 1) * a.dy];
            if (m_has_aux<KIND>() && a.aux) aux_next = a.aux[t + 1];
        }
        double xp[4], lwp[4];
        u32* A = f_A(a, t) + (i64)isl * N;
        if (rsp) {
            // ---- ancestors: exactly k_ancestors<true> for the single tile b = 0
            u64 q4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // (APF: the auxiliary weight of the parent, logeta(t - 1, x) with data[t] -- core.py:307-313)
                const double la = APF ? lw[i] + m_logeta<KIND>(p, x[i], yt) : lw[i];
                q4[i] = (jt + i < N) ? smc_q62_w(f_weight(la, m, rs)) : 0ull;
            }
            const u64 tsum = q4[0] + q4[1] + q4[2] + q4[3];
            u64 total, pre;
            const u64 cex = smc_block_exscan_plus_sum_u64(tsum, 0ull, smu, total, pre);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) sX[jt + k] = x[k];                 // parents' states for the gather
            i64 an[4] = {0, 0, 0, 0};
            f_tile_offspring<BS>(a, isl, t, 0, jt, 0, q4, cex, 0ull, total, sC, sP, sn, smx,
                             [&](i64 n0, const bool (&ok)[4], const i64 (&a4)[4]) {
                                 // N <= 1024: one pass, n0 == jt
#pragma unroll
                                 for (int i = 0; i < 4; ++i) an[i] = ok[i] ? a4[i] : 0;
                                 const u32 a32[4] = {(u32)an[0], (u32)an[1], (u32)an[2], (u32)an[3]};
                                 if (vec && ok[0] && ok[3]) {
                                     smc_st4g(A + n0, a32);
                                 } else {
#pragma unroll
                                     for (int i = 0; i < 4; ++i)
                                         if (ok[i]) smc_stg(A + n0 + i, a32[i]);
                                 }
                             });
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {                                           // core.py:332
                xp[k] = sX[an[k]];
                // core.py:299-305 reset_weights: log_mean_exp(logeta, W) - logeta[A] for the APF
                lwp[k] = APF ? cconst - m_logeta<KIND>(p, xp[k], yt) : 0.0;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) { xp[k] = first ? 0.0 : x[k]; lwp[k] = first ? 0.0 : lw[k]; }
        }
        // ---- standard normals (same counters as k_propagate), propagate, weigh
        double z[4];
        if (a.zt) {
            const double* zt = a.zt + ((i64)t * a.zt_ts + (i64)isl * N);
#pragma unroll
            for (int k = 0; k < 4; ++k) z[k] = (jt + k < N) ? smc_ldg(zt + jt + k) : 0.0;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k += 2)
                smc_normal_pair(s_ntab, a.seed, (u32)((jt + k) >> 1), (u32)t, gisl, SMC_STREAM_NORMAL, z[k], z[k + 1]);
        }
        bool okp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            okp[k] = jt + k < N;
            double inc;
            x[k] = m_step<KIND, FK>(p, first, yt, aux, xp[k], z[k], inc);
            double l = first ? inc : lwp[k] + inc;                                  // resampling.py:241-244
                                                                                    // (lwp = 0 after a reset)
            if (l != l) l = -INFINITY;                                              // resampling.py:220
            lw[k] = okp[k] ? l : -INFINITY;
        }
        f_store4<double>(f_X(a, t) + (i64)isl * N, jt, vec && okp[3], okp, x);
        f_store4<double>(f_lw(a, t) + (i64)isl * N, jt, vec && okp[3], okp, lw);
        // ---- log-sum-exp of the workgroup = of the filter (as in k_propagate)
        double tm = lw[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) tm = smc_max2(tm, lw[k]);
        const double gm = smc_block_max(tm, smm);
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double e = (lw[k] > -INFINITY) ? smc_exp_nonpos(lw[k] - gm) : 0.0;
            s1 += e;
            s2 = fma(e, e, s2);
        }
        smc_block_sum2(s1, s2, smd);
        // ---- finalise step t (every thread computes the same scalars), decide step t+1
        const bool bad = !(gm > -INFINITY) || !(gm < INFINITY);
        const double ess = bad ? NAN : (s1 * s1) / s2;                              // resampling.py:226
        const double log_mean = bad ? NAN : gm + log(s1 / (double)N);               // resampling.py:224
        double* row = a.summ + ((i64)isl * (a.T + 1) + t) * SUMM_STRIDE;
        double loglt;                                                               // core.py:355-359
        if (first || rsp) loglt = log_mean;
        else loglt = log_mean - prev_log_mean;
        const double logLt = (first ? 0.0 : prev_logLt) + loglt;
        prev_log_mean = log_mean;
        prev_logLt = logLt;
        if (tid == 0) {
            row[0] = ess;
            row[1] = log_mean;
            row[2] = loglt;
            row[3] = logLt;
            row[4] = rsp ? 1.0 : 0.0;
            row[5] = gm;
            row[6] = bad ? NAN : 1.0 / s1;
        }
        m = gm;
        rs = bad ? NAN : 1.0 / s1;
        resample = (t + 1 < a.T) && (ess < a.ess_thresh);                           // core.py:181-183
        if (APF && t + 1 < a.T) {
            // auxiliary weights of the NEXT step: aux = wgts.add(logeta(t, X)) (core.py:307-313);
            // the decision, the CDF's normalisation and the reset constant come from them
            double la[4], tma = -INFINITY;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                la[k] = okp[k] ? lw[k] + m_logeta<KIND>(p, x[k], y_next) : -INFINITY;
                if (la[k] != la[k]) la[k] = -INFINITY;
                tma = smc_max2(tma, la[k]);
            }
            __syncthreads();
            const double gma = smc_block_max(tma, smm);
            double a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double e = (la[k] > -INFINITY) ? smc_exp_nonpos(la[k] - gma) : 0.0;
                a1 += e;
                a2 = fma(e, e, a2);
            }
            __syncthreads();
            smc_block_sum2(a1, a2, smd);
            const bool bada = !(gma > -INFINITY) || !(gma < INFINITY);
            const double essa = bada ? NAN : (a1 * a1) / a2;
            resample = essa < a.ess_thresh;
            m = gma;
            rs = bada ? NAN : 1.0 / a1;
            cconst = (gma - gm) + log(a1 / s1);           // log sum_i W_i exp(logeta_i)
        }
    }
    if (tid == 0) {                 // the step record the other kernels / the next launch read
        info[0] = (double)t;
        info[1] = resample ? 1.0 : 0.0;
        info[2] = (t < a.T) ? a.y[t * a.dy] : 0.0;
        info[3] = m;
        info[4] = rs;
        info[5] = (a.aux && t < a.T) ? a.aux[t] : 0.0;
        if (APF) info[6] = cconst;
    }
}
