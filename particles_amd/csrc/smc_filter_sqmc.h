// smc_filter_sqmc.h -- the SQMC step (particles/core.py:339-349, Gerber & Chopin's sequential quasi-Monte
// Carlo) of a univariate fused filter, on the kernels of the two-level step:
//
//     u = sobol(N, 2);  tau = argsort(u[:, 0])                    -> closed form for N = 2^k (smc_qmc.h)
//     h_order = hilbert_sort(X) = argsort(X)            (d = 1)   -> the radix sort (smc_sort.hip)
//     A = h_order[inverse_cdf(u[tau, 0], W[h_order])]             -> k_sq_permute: the tile partials and integer
//                                                                    CDFs of the weights IN SORTED ORDER; k_reduce2;
//                                                                    k_ancestors2<MID, MULTI, .., SQ>: the multinomial
//                                                                    counts with the thresholds a FUNCTION of n (the
//                                                                    sorted first coordinates are a regular grid:
//                                                                    f2_sq_T), storing h_order[A] (FArgs::sq_perm)
//     X = Gamma(t, X[A], u[tau, 1])  (ppf of the Normal kernel)   -> k_propagate unchanged, its standard normals
//                                                                    z_n = ndtri(u[tau_n, 1]) read from a tape
//                                                                    k_sq_permute wrote
// SQMC always resamples (core.py:340): the filter's ESS threshold is +inf.  The points are the stand-alone
// operator's (smc_sobol / smc_sobol_sorted with scramble = safe = 1): point set `ctr0 + t` of the stream keyed
// by `pseed`, so that a run of SMC(qmc=True) on the operators and the fused run see the same points.
#pragma once
#include "smc_filter_kernels.h"
#include "smc_qmc.h"

// t = 0 (core.py:315-321 generate_particles: X_0 = Gamma0(sobol(N, 1))): the tape of step 0
__global__ void __launch_bounds__(SMC_BLOCK)
k_sq_init(const FArgs av, double* zbuf, const u64 pseed, const u64 ctr0)
{
    const FArgs& a = av;
    const int isl = (int)blockIdx.y;
    const i64 n = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (n >= a.N) return;
    const u64 ctr = ctr0 + ((u64)(u32)(a.island_offset + isl) << 32);
    const u32 sh0 = smc_sobol_shift(pseed, ctr, 0u);
    u32 x0, x1;
    smc_sobol2((u32)n ^ ((u32)n >> 1), sh0, 0u, x0, x1);
    zbuf[(i64)isl * a.N + n] = smc_ndtri(smc_sobol_safe(x0));
}

// step t >= 1, one workgroup per tile of 1024 SORTED positions (ownership as in the tail-free k_propagate):
// leaves what k_propagate(t-1) left for the unsorted order -- (K_b, S_b, SS_b), the tile's integer CDF, t_b --
// for the sorted one; writes the tape of step t's moves.  The log-weights of step t - 1 in sorted order:
//   RECOMPUTE (bootstrap filters whose log G depends on the new particle only): SQMC always resamples, so
//     lw_{t-1} = log G_{t-1}(x) = m_obs_logpdf(y_{t-1}, x) -- evaluated again from the SORTED KEYS (the sort's
//     key images decode to the particles: a sequential read) by the very function k_propagate stored it with,
//     hence the same bits (check_sqmc_fused runs both forms and compares the ESS rows with ==);
//   else gathered through the permutation (a random 8-byte gather of 8 MB at N = 2^20: 20 of the kernel's 30 us).
template <int KIND, bool RECOMPUTE>
__global__ void __launch_bounds__(SMC_BLOCK)
k_sq_permute(const FArgs av, const u64* perm, const u64* skeys, double* zbuf, const u64 pseed, const u64 ctr0)
{
    const FArgs& a = av;
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y, tid = (int)threadIdx.x;
    const i64 N = a.N;
    const i64 t = (i64)smc_uniform(smc_ldg(a.info2 + (i64)isl * INFO_STRIDE));
    if (t >= a.T || t == 0) return;
    const FOwn own = f_own<true>(b, tid, N);
    double lw[4];
    if (RECOMPUTE) {
        const u64* sk = skeys + (i64)isl * N;
        u64 k4[4];
        smc_ld2g(sk + own.na, k4[0], k4[1]);
        smc_ld2g(sk + own.nb, k4[2], k4[3]);
        const double* p = a.params + (i64)isl * PARAM_STRIDE;
        const double y = a.y[(t - 1) * a.dy];
        const double aux = (m_has_aux<KIND>() && a.aux) ? a.aux[t - 1] : 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u64 e = k4[k];                                   // rs_decode (fp64 keys)
            const double x = __longlong_as_double((long long)((e >> 63) ? (e & 0x7fffffffffffffffull) : ~e));
            double l = m_obs_logpdf<KIND>(p, y, x, 0.0, t - 1 == 0, aux);
            if (l != l) l = -INFINITY;                             // resampling.py:220
            lw[k] = l;
        }
    } else {
        const u64* pi = perm + (i64)isl * N;
        const double* lwo = f_lw(a, t - 1) + (i64)isl * N;
        u64 p4[4];
        smc_ld2g(pi + own.na, p4[0], p4[1]);
        smc_ld2g(pi + own.nb, p4[2], p4[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) lw[k] = smc_ldg(lwo + p4[k]);
    }
    // the step's points, while the loads are on their way.  Second coordinate of the point with Gray code g:
    // the XOR of the direction numbers v_k over the set bits of g -- looked up 7 bits at a time in tables the
    // workgroup builds once (5 x 128 words of LDS) instead of walking the 20 bits of every g
    __shared__ u32 s_v2[5 * 128];
    // the tail arguments of ndtri (27 % of the tile's 1024: smc_qmc.h), packed: slot k * 256 + tid of s_y holds the
    // argument, then the result; every wave that ran the tail branch for one lane paid for all 64 (25 of this
    // kernel's 25 us at N = 2^20 were VALU), a packed queue is 1.1 passes per wave instead of 4
    __shared__ double s_y[4 * SMC_BLOCK];
    __shared__ unsigned short s_q[4 * SMC_BLOCK];
    __shared__ unsigned s_nq;
    if (tid == 0) s_nq = 0u;
    for (int e = tid; e < 5 * 128; e += SMC_BLOCK) {
        const int c = e >> 7, bits = e & 127;
        u32 m = 1u, acc = 0u;
        for (int k = 0; k < 7 * c + 7 && k < SOBOL_BITS; ++k) {
            if (k >= 7 * c && ((bits >> (k - 7 * c)) & 1)) acc ^= m << (SOBOL_BITS - 1 - k);
            m ^= m << 1;
        }
        s_v2[e] = acc;
    }
    __syncthreads();
    const u64 ctr = ctr0 + (u64)t + ((u64)(u32)(a.island_offset + isl) << 32);
    const u32 sh0 = smc_sobol_shift(pseed, ctr, 0u), sh1 = smc_sobol_shift(pseed, ctr, 1u);
    double z[4];
    bool tl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32 g = smc_sobol_sorted_gray((u32)f_own_idx(own, k), sh0, a.log2N);
        const u32 x1 = sh1 ^ s_v2[g & 127u] ^ s_v2[128 + ((g >> 7) & 127u)] ^ s_v2[256 + ((g >> 14) & 127u)] ^
                       s_v2[384 + ((g >> 21) & 127u)] ^ s_v2[512 + ((g >> 28) & 127u)];
        const double y0 = smc_sobol_safe(x1);
        z[k] = smc_ndtri_centre(y0, tl[k]);
        if (tl[k]) {
            const unsigned q = atomicAdd(&s_nq, 1u);
            s_q[q] = (unsigned short)(k * SMC_BLOCK + tid);
            s_y[k * SMC_BLOCK + tid] = y0;
        }
    }
    __syncthreads();
    {
        const int nq = (int)s_nq;
        for (int i = tid; i < nq; i += SMC_BLOCK) {
            const int slot = (int)s_q[i];
            s_y[slot] = smc_ndtri_tail(s_y[slot]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (tl[k]) z[k] = s_y[k * SMC_BLOCK + tid];
    // (the sorted first coordinates are not written: k_ancestors2<SQ> forms the thresholds from n -- f2_sq_T)
    double* zb = zbuf + (i64)isl * N;
    smc_st2g(zb + own.na, z[0], z[1]);
    smc_st2g(zb + own.nb, z[2], z[3]);
    u64 cx[4];
    const F2Tile r = f2_tile_weights(lw, cx);
    u64* cq = a.cq + (i64)isl * a.ncq;
    smc_st2g(cq + own.na, cx[0], cx[1]);
    smc_st2g(cq + own.nb, cx[2], cx[3]);
    if (tid == 0) {
        const i64 o = (i64)isl * a.nparts;
        a.pm[o + b] = r.K;
        a.ps[o + b] = r.S;
        a.pss[o + b] = r.SS;
        a.tq[o + b] = r.tb;
    }
}

// ---- multivariate filters (MVLinearGauss, 2 <= d <= 9; the flat step): h_order = hilbert_sort(X) (hilbert.py:33-58,
// the stand-alone operator's kernels), the step's d + 1 Sobol' coordinates sorted by the first (smc_sobol_points).
// k_sqmv_tapes: the tape of the moves z = ndtri(u[:, 1:]) and -- t >= 1 -- the log-weights of step t - 1 in
// Hilbert order (to a scratch row, copied over the slot afterwards: the flat resampling kernels form their exact
// Q62 CDF straight from that slot) and the sorted first coordinates in the buffer the flat multinomial search
// reads its sorted uniforms from (a.su).
__global__ void __launch_bounds__(SMC_BLOCK)
k_sqmv_tapes(const FArgs av, const int isl, const i64* perm, const double* U, const int du, double* lw_sorted, double* zbuf)
{
    const FArgs& a = av;
    const i64 n = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (n >= a.N) return;
    const int d = a.dx;
    const double* row = U + n * du;
    double* z = zbuf + ((i64)isl * a.N + n) * d;
    for (int j = 0; j < d; ++j) z[j] = smc_ndtri(row[du - d + j]);
    if (perm) {
        const i64 t = (i64)smc_uniform(smc_ldg(a.info + (i64)isl * INFO_STRIDE));
        if (t >= a.T || t == 0) return;
        lw_sorted[n] = smc_ldg(f_lw(a, t - 1) + (i64)isl * a.N + perm[n]);
        a.su[(i64)isl * a.N + n] = row[0];
    }
}
// A_t <- h_order[A_t] (int64 permutation of the Hilbert sort)
__global__ void __launch_bounds__(SMC_BLOCK)
k_sqmv_compose(const FArgs av, const int isl, const i64* perm)
{
    const FArgs& a = av;
    const double* info = a.info + (i64)isl * INFO_STRIDE;
    // (the flat step's record: info[0] = the step being run, info[1] = its decision)
    const i64 t = (i64)smc_uniform(smc_ldg(info));
    if (t >= a.T || t == 0 || smc_uniform(smc_ldg(info + 1)) == 0.0) return;
    const i64 n = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (n >= a.N) return;
    u32* A = f_A(a, t) + (i64)isl * a.N;
    A[n] = (u32)perm[A[n]];
}
