// smc_filter_mv.h -- propagate kernel of the fused step loop for the
// multivariate linear Gaussian model (particles/kalman.py:296-361
// MVLinearGauss) under the bootstrap (state_space_models.py:299-349) and the
// guided filter with the model's optimal proposal (state_space_models.py:
// 352-398, kalman.py:348-356, filter_step kalman.py:196-229).
//
// Per particle, with the (d,d) matrices shared by all particles:
//   bootstrap  x = F xp + L_X z                     log G = log N(y; G x, covY)
//   guided     m = F xp ; mu = m + K (y - G m) = B xp + K y ;  x = mu + L_P z
//              log G = log N(x; m, covX) + log N(y; G x, covY) - log N(x; mu, P)
// (distributions.py:946-959: rvs = loc + Z L^T, logpdf via L^-1 (x-loc)).
// The last term needs no solve: x - mu = L_P z, so it is -|z|^2/2 - c_P.
// Triangular solves become products with the precomputed inverse factors.
// COLL (opts.flags & SMC_FLAG_COLLAPSED_PROPOSAL, guided only): with the model's OPTIMAL
// proposal the three terms of log G collapse analytically,
//      p(x_t | x_{t-1}) p(y_t | x_t) / q(x_t | x_{t-1}, y_t) = p(y_t | x_{t-1})
//      log G = log N(y_t; G F xp, S),  S = G covX G' + covY          (t = 0: the constant log p(y_0))
// (kalman.py:215-229: S is filter_step's innovation covariance): one product with -(L_S^-1 G F)
// instead of three -- 44 instead of 72 MFMAs per 16 particles, and no |z|^2, |u|^2.  Same
// particles bit for bit, log-weights equal up to rounding (the three-term form cancels two
// O(d) numbers): hence a flag, the default keeps the reference's expression.
//
// Mapping (MI355X).  This is the one GEMM-shaped piece of the path -- (N, d)
// particle rows times (d, d) matrices, fp64 -- and it runs on the matrix cores:
// v_mfma_f64_16x16x4_f64 computes D(16x16) += A(16x4) B(4x16) per wavefront.
// A wave works on 16 particles at a time in the TRANSPOSED product
//      out^T (d x 16 particles) = M (d x d) . v^T (d x 16 particles)
// with A = a 16x4 block of the constant matrix (one double per lane, kept in
// LDS in exactly the lane order the instruction wants) and B = 4 consecutive
// dimensions of the 16 particles.  Lane l = 16 g + n holds, of particle n,
// the dimensions congruent to g modulo 4 -- and that is also where the
// instruction leaves its results (row = (lane>>4) + 4 reg, col = lane&15), so
// the output of one product is the B operand of the next without moving a
// single register: five chained products per particle (guided), each 16 (12
// for the triangular factors, whose upper blocks are skipped) MFMAs per 16
// particles, nothing staged through LDS but the matrices themselves.
// A wave keeps two 16-particle groups in flight so that every matrix fragment
// read feeds two independent accumulator chains.
// FK = SMC_FK_APF (AuxiliaryPF of the model, kalman.py:358-361): the guided filter above plus the auxiliary
// weights of core.py:299-313.  logeta(t-1, x) = log p(y_t | x_{t-1} = x) = log N(y_t; G F x, S) is the collapsed
// form's one product; k_mv_aux(t) evaluates it for the particles of step t-1 at the start of step t, leaves
// eta and the plain log-weights aside and puts lw + eta in their place, k_mv_aux_restate redoes the island's
// reduction, decision and normalisation on them (the resampling kernels then run unchanged), and the
// propagate kernel resets a resampled particle's weight to log_mean_exp(eta, W) - eta[A] (the constant
// travels in the step record) or restores the plain weight.
// DG (FArgs::mv_diag: G, covX, covY and cov0 are all DIAGONAL -- independent noises, e.g. kalman.py:364-394
// MVLinearGauss_Guarniero_etal, BASELINE config C4): the Cholesky factors L_X, L_Y, L_0, the proposal's L_P = chol(covX
// - K G covX) and their inverses are diagonal too, so "x = mu + L z", "u = L_X^-1 (x - m)" and "w = L_Y^-1 (y - G x)" are
// one fma per element instead of 12 + 12 + 16 MFMAs per 16 particles: 32 of the guided step's 72 remain (F xp, B xp).
// Same bits: in the dense product every other term of an element's sum is (+-0) . v_j, which changes no finite
// accumulator; `check_mv_diag_equals_dense` compares the two forms particle for particle (SMC_PATH_MV_DENSE keeps the
// dense kernel selectable: the verification twin, and bench.py's `c4_dense` leg).
// Philox normals follow the usual contract (pair kp of particle n -> dimensions
// 2kp, 2kp+1; counter n*ceil(d/2)+kp): the two lanes that own the halves of a
// pair each generate half of the pairs and swap the other element.
#pragma once
#include "smc_filter_kernels.h"

// ---- layout of the constants block `mvc` (doubles).  Every matrix is padded
// with zeros to DP x DP (DP = 16 or 32) and stored as MFMA A-fragments:
//   frag[(jb * DP/4 + kb) * 64 + lane] = M[16 jb + (lane & 15)][4 kb + (lane >> 4)]
#define MV_F 0        /* F                                   */
#define MV_B 1        /* (I - K G) F            (guided)     */
#define MV_LZ 2       /* factor applied to z, t >= 1: L_X (bootstrap) / L_P (guided) */
#define MV_LZ0 3      /* same at t = 0: L_0 / L_P0            */
#define MV_XINV 4     /* L_X^-1                  (guided)     */
#define MV_X0INV 5    /* L_0^-1                  (guided)     */
#define MV_NGY 6      /* -(L_Y^-1 G)                          */
#define MV_NGF 7      /* -(L_S^-1 G F), S = G covX G' + covY   (guided, collapsed form) */
#define MV_NMAT 8
#define MV_VEC(dp) (MV_NMAT * (dp) * (dp))         /* mu0[dp], mup0[dp] */
#define MV_SCAL(dp) (MV_VEC(dp) + 2 * (dp))        /* cX, cY, cP, c0, cP0, cS, log p(y_0), - */
#define MV_DIAGV(dp) (MV_SCAL(dp) + 8)             /* diagonals of LZ, LZ0, XINV, X0INV, NGY (what the DG kernels apply) */
#define MV_NDIAGV 5
#define MV_STEP(dp) (MV_DIAGV(dp) + MV_NDIAGV * (dp)) /* per t: yw_t[dp] = L_Y^-1 y_t, ky_t[dp] = K y_t, ys_t[dp] = L_S^-1 y_t */
#define MV_NSTEPV 3
#define MV_SIZE(dp, T) (MV_STEP(dp) + MV_NSTEPV * (size_t)(dp) * (T))

#define MV_G 1        /* 16-particle groups a wave keeps in flight */

// acc[gi][jb] += M . v[gi]   for the MV_G groups; `frag` = the matrix's fragments in LDS.
// v[gi][kb] = dimension 4 kb + g of the lane's particle; acc[gi][jb][r] = dimension
// 16 jb + 4 r + g.  LOWER: M is lower triangular, blocks above the diagonal are skipped.
template <int DP, bool LOWER>
__device__ __forceinline__ void mv_product(const double* frag, const double (&v)[MV_G][DP / 4],
                                           smc_v4d (&acc)[MV_G][DP / 16], const int lane)
{
    // kb outer: consecutive MFMAs go to different accumulators (no dependent stalls)
#pragma unroll
    for (int kb = 0; kb < DP / 4; ++kb) {
#pragma unroll
        for (int jb = 0; jb < DP / 16; ++jb) {
            if (LOWER && 4 * kb >= 16 * (jb + 1)) continue;
            const double m = frag[(jb * (DP / 4) + kb) * 64 + lane];
#pragma unroll
            for (int gi = 0; gi < MV_G; ++gi)
                acc[gi][jb] = smc_mfma_f64_16x16x4(m, v[gi][kb], acc[gi][jb]);
        }
    }
}

// sum over the 4 lanes (g = 0..3) that share a particle; every one of them gets it
__device__ __forceinline__ double mv_sum_g(double v)
{
    return smc_sum_rows(v);       // two lane swaps on the VALU (smc_dpp.h), no LDS round trips
}

// minimum waves per SIMD the register allocation must leave room for (A/B builds: -DSMC_MV_MINW=.. / -DSMC_MV_MINW_DG=..).
// Dense form: 3 (168 registers, no spill; with 50 KB of LDS three workgroups share a CU, and the host picks 4 chunks
// per workgroup at N = 2^20: 222 -> 217 us, profiles/r15_c4_ab.txt).  Element-wise form: unconstrained (187 registers,
// 2 waves per SIMD; 3 or 4 cost spills and 10 - 80 %, same file).
#ifndef SMC_MV_MINW
#define SMC_MV_MINW 3
#endif
#ifndef SMC_MV_MINW_DG
#define SMC_MV_MINW_DG 1
#endif
template <int FK, int DP, bool DFULL, bool COLL = false, bool DG = false>
__global__ void __launch_bounds__(SMC_BLOCK, DG ? SMC_MV_MINW_DG : SMC_MV_MINW)
k_propagate_mv(const FArgs av, const double* __restrict__ C)
{
    static_assert(!COLL || FK == SMC_FK_GUIDED, "the collapsed form is the guided filter's");
    constexpr bool APF = FK == SMC_FK_APF;
    constexpr bool GUIDED = FK == SMC_FK_GUIDED || APF;           // the optimal proposal
    constexpr int NV = DP / 4;                    // dimensions per lane
    constexpr int NJ = DP / 16;                   // 16-row blocks of a product
    constexpr int MM = DP * DP;
    constexpr bool GUIDED3 = GUIDED && !COLL;                     // the reference's three-term weight
    // DG: the factors are diagonal and live in sDg; only F, B and (COLL) -(L_S^-1 G F) are staged as fragments
    constexpr int NSLOT = DG ? (GUIDED ? 2 : 1) : (GUIDED3 ? 5 : 3);
    constexpr int S_F = 0, S_B = COLL ? 0 : 1;
    constexpr int S_LZ = GUIDED3 ? 2 : 1;
    constexpr int S_XINV = 3;
    constexpr int S_NGY = DG ? 1 : (GUIDED3 ? 4 : 2);        // COLL: holds -(L_S^-1 G F)
    const FArgs& a = av;
    __shared__ double sM[NSLOT * MM];
    __shared__ double sVec[4 * DP];               // mu (t = 0) or K y_t | mu0 (t = 0) or 0 | L_Y^-1 y_t | the scalars
    __shared__ double sDg[DG ? 3 * DP : 1];       // DG: the diagonals of LZ | XINV | NGY of this step
    __shared__ double smd[SMC_SM];
    __shared__ int s_last;
    SMC_NTAB_LDS(s_ntab);
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y;
    const int tid = (int)threadIdx.x;
    smc_ntab_stage<SMC_BLOCK>(s_ntab, tid);       // (the barrier behind the matrices covers it)
    const int lane = tid & 63, wv = tid >> 6;
    const int g = lane >> 4, pn = lane & 15;
    double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)smc_uniform(info[0]);
    if (t >= a.T) return;
    const i64 N = a.N;
    const int d = a.dx;
    const u32 gisl = (u32)(a.island_offset + isl);
    SMC_GLOBAL(double) Xn = SMC_AS_GLOBAL(double, f_X(a, t) + (i64)isl * N * d);
    SMC_GLOBAL(const double) Xo = SMC_AS_GLOBAL(const double, f_X(a, t - 1) + (i64)isl * N * d);
    SMC_GLOBAL(double) lwn = SMC_AS_GLOBAL(double, f_lw(a, t) + (i64)isl * N);
    SMC_GLOBAL(const double) lwo = SMC_AS_GLOBAL(const double, f_lw(a, t - 1) + (i64)isl * N);
    SMC_GLOBAL(const u32) A = SMC_AS_GLOBAL(const u32, f_A(a, t) + (i64)isl * N);
    SMC_GLOBAL(const double) zt =
        SMC_AS_GLOBAL(const double, a.zt ? a.zt + ((i64)t * a.zt_ts + (i64)isl * N * d) : nullptr);
    const bool first = (t == 0);
    const bool resample = !first && smc_uniform(info[1]) != 0.0;
    const double* scal = C + MV_SCAL(DP);
    const double* yw = C + MV_STEP(DP) + (size_t)t * MV_NSTEPV * DP;
    const double* ky = yw + DP;
    const double* ys = yw + 2 * DP;

    // ---- the matrices of this step -> LDS (fragment order, as stored)
    {
        const int m_lz = first ? MV_LZ0 : MV_LZ, m_xinv = first ? MV_X0INV : MV_XINV;
        for (int i = tid; i < MM; i += SMC_BLOCK) {
            if (!first) {
                if (!COLL) sM[S_F * MM + i] = C[MV_F * MM + i];
                if (GUIDED) sM[S_B * MM + i] = C[MV_B * MM + i];
            }
            if (!DG) sM[S_LZ * MM + i] = C[m_lz * MM + i];
            if (GUIDED3 && !DG) sM[S_XINV * MM + i] = C[m_xinv * MM + i];
            if (!DG || COLL) sM[S_NGY * MM + i] = C[(COLL ? MV_NGF : MV_NGY) * MM + i];
        }
        if (DG && tid < DP) {
            const double* dv = C + MV_DIAGV(DP);
            sDg[tid] = dv[(first ? 1 : 0) * DP + tid];
            sDg[DP + tid] = dv[(first ? 3 : 2) * DP + tid];
            sDg[2 * DP + tid] = dv[4 * DP + tid];
        }
        // per-lane reads of these vectors come from LDS: indexed by g from the constant
        // block they turn into scalar loads plus a select chain per element
        if (tid < DP) {
            const double* mu = C + MV_VEC(DP) + (GUIDED ? DP : 0);
            sVec[tid] = first ? mu[tid] : (GUIDED ? ky[tid] : 0.0);
            sVec[DP + tid] = first ? C[MV_VEC(DP) + tid] : 0.0;
            sVec[2 * DP + tid] = COLL ? ys[tid] : yw[tid];
            // (the per-particle tail reads its constants from LDS: as scalar loads inside the loop they cost a
            //  round trip to the scalar cache every fourth iteration, and the SGPR file is full)
            if (tid < 8) sVec[3 * DP + tid] = scal[tid];
        }
    }
    __syncthreads();
    const double* vMu = sVec + g;                 // element 16 jb + 4 r of the lane's view
    const double* vMu0 = sVec + DP + g;
    const double* vYw = sVec + 2 * DP + g;
    const double* vDg = sDg + g;                  // (DG) element 16 jb + 4 r of the lane's view of a diagonal

    SmcLse lacc = smc_lse_empty();
    const int hp = (d + 1) / 2;
    // rows are requested one iteration ahead (and their ancestor index two ahead):
    // at 2 waves per SIMD nothing else hides the two dependent round trips
    const int nit = a.mv_chunks * (4 / MV_G);
    auto particle = [&](int it, int gi) -> i64 {
        return ((i64)b * a.mv_chunks + it / (4 / MV_G)) * SMC_BLOCK + (i64)wv * 64 +
               (i64)((it % (4 / MV_G)) * MV_G + gi) * 16 + pn;
    };
    // loads are unconditional on clamped addresses and masked afterwards (selects, not
    // exec-mask regions: those cost an SGPR pair each and spill)
    constexpr bool dfull = DFULL;               // d == DP: no padded dimensions to mask
    // The ancestor word of iteration it is REQUESTED two iterations ahead and only looked at
    // when its row is requested (one iteration ahead): touching it earlier -- even to widen it
    // to 64 bits -- makes the compiler wait for it, and with it for the row loads in flight.
    auto parent_raw = [&](int it, int gi) -> u32 {
        const i64 nn = particle(it, gi);
        return A[nn < N ? nn : N - 1];
    };
    auto parent_row = [&](int it, int gi, u32 raw) -> i64 {               // core.py:332 / :336
        const i64 nn = particle(it, gi);
        const bool ok = !first && it < nit && nn < N;
        return ok ? (resample ? (i64)raw : nn) : -1;
    };
    auto load_row = [&](SMC_GLOBAL(const double) base, i64 row, double (&dst)[NV]) {
        const bool rv = row >= 0;
        SMC_GLOBAL(const double) pr = base + (rv ? row : 0) * d;
#pragma unroll
        for (int kb = 0; kb < NV; ++kb) {
            const int k = 4 * kb + g;
            const bool kin = dfull || k < d;
            const double x = pr[kin ? k : 0];
            dst[kb] = (rv && kin) ? x : 0.0;
        }
    };
    static_assert(MV_G == 1, "the per-particle tail below assumes one group per iteration");
    double kw = 0.0, kz = 0.0, ku = 0.0, lwprev = 0.0;
    u32 anext[MV_G];
    double nx[MV_G][NV];
#pragma unroll
    for (int gi = 0; gi < MV_G; ++gi) {
        load_row(Xo, parent_row(0, gi, parent_raw(0, gi)), nx[gi]);
        anext[gi] = parent_raw(1, gi);
    }
#pragma unroll 1
    for (int it = 0; it < nit; ++it) {
        i64 n[MV_G];
        bool valid[MV_G];
        double v[MV_G][NV];                 // current B operand: xp, then z, x, x - m
        smc_v4d am[MV_G][NJ], ax[MV_G][NJ];
        // the previous log-weight of the particle this lane will own at the end of the quad (not resampled:
        // resampling.py:241-244), requested three iterations before it is added
        if ((it & 3) == 0 && !APF && !first && !resample) {
            const i64 np = particle(it, 0) - pn + lane;
            lwprev = lwo[np < N ? np : N - 1];
        }
        // ---- the parents' rows (requested during the previous iteration)
#pragma unroll
        for (int gi = 0; gi < MV_G; ++gi) {
            n[gi] = particle(it, gi);
            valid[gi] = n[gi] < N;
#pragma unroll
            for (int kb = 0; kb < NV; ++kb) v[gi][kb] = nx[gi][kb];
            load_row(Xo, parent_row(it + 1, gi, anext[gi]), nx[gi]);
            anext[gi] = parent_raw(it + 2, gi);
        }
        // ---- mean of the proposal -> ax ; guided keeps m = F xp in am
#pragma unroll
        for (int gi = 0; gi < MV_G; ++gi)
#pragma unroll
            for (int jb = 0; jb < NJ; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ax[gi][jb][r] = vMu[16 * jb + 4 * r];                  // mu (t = 0) / K y_t / 0
                    am[gi][jb][r] = vMu0[16 * jb + 4 * r];                 // mu0 (t = 0) / 0
                }
        if (COLL) {
            // w = L_S^-1 (y - G F xp) accumulates in am (started from L_S^-1 y_t), mu in ax
#pragma unroll
            for (int gi = 0; gi < MV_G; ++gi)
#pragma unroll
                for (int jb = 0; jb < NJ; ++jb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) am[gi][jb][r] = vYw[16 * jb + 4 * r];
            if (!first) {
                mv_product<DP, false>(sM + S_B * MM, v, ax, lane);          // mu = B xp + K y
                mv_product<DP, false>(sM + S_NGY * MM, v, am, lane);        // w
            }
        } else if (!first) {
            if (GUIDED) {
                mv_product<DP, false>(sM + S_F * MM, v, am, lane);          // m = F xp
                mv_product<DP, false>(sM + S_B * MM, v, ax, lane);          // mu = B xp + K y
            } else {
                mv_product<DP, false>(sM + S_F * MM, v, ax, lane);          // mu = F xp
            }
        }
        // ---- z -> v, zz = |z|^2
        double zz[MV_G];
#pragma unroll
        for (int gi = 0; gi < MV_G; ++gi) {
            if (zt) {
                load_row(zt, valid[gi] ? n[gi] : -1, v[gi]);
            } else {
                // lane (g = 2h + e) needs element e of the pairs kp = 2 j + h, j < NV; the
                // e = 0 lane generates j < NV/2, its e = 1 neighbour (lane + 16) the rest
                const int h = g >> 1, e = g & 1;
#pragma unroll
                for (int jj = 0; jj < NV / 2; ++jj) {
                    const int j = jj + e * (NV / 2);
                    const int kp = 2 * j + h;
                    double z0, z1;
                    smc_normal_pair(s_ntab, a.seed, (u32)(n[gi] * hp + kp), (u32)t, gisl,
                                    SMC_STREAM_NORMAL, z0, z1);
                    if (!dfull) {
                        if (2 * kp >= d) z0 = 0.0;
                        if (2 * kp + 1 >= d) z1 = 0.0;
                    }
                    // the e = 0 lane keeps its z0 and takes its neighbour's (pair jj + NV/2); the e = 1 lane keeps
                    // its z1 and takes its neighbour's (pair jj): rows 1, 3 of z0 trade places with rows 0, 2 of z1
                    // -- one lane swap per word puts both elements where they belong, no select
                    smc_swap16_f64(z0, z1);
                    v[gi][jj] = z0;
                    v[gi][jj + NV / 2] = z1;
                }
            }
            double q = 0.0;
            if (GUIDED3) {
#pragma unroll
                for (int kb = 0; kb < NV; ++kb) q = fma(v[gi][kb], v[gi][kb], q);
            }
            zz[gi] = q;
        }
        // ---- x = mu + L z
        if (DG) {
#pragma unroll
            for (int gi = 0; gi < MV_G; ++gi)
#pragma unroll
                for (int jb = 0; jb < NJ; ++jb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ax[gi][jb][r] = fma(vDg[16 * jb + 4 * r], v[gi][4 * jb + r], ax[gi][jb][r]);
        } else {
            mv_product<DP, true>(sM + S_LZ * MM, v, ax, lane);
        }
        double uu[MV_G];
        if (GUIDED3) {
            // u = L_X^-1 (x - m)
            smc_v4d au[MV_G][NJ];
#pragma unroll
            for (int gi = 0; gi < MV_G; ++gi)
#pragma unroll
                for (int jb = 0; jb < NJ; ++jb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[gi][4 * jb + r] = ax[gi][jb][r] - am[gi][jb][r];
                        au[gi][jb][r] = DG ? fma(vDg[DP + 16 * jb + 4 * r], v[gi][4 * jb + r], 0.0) : 0.0;
                    }
            if (!DG) mv_product<DP, true>(sM + S_XINV * MM, v, au, lane);
#pragma unroll
            for (int gi = 0; gi < MV_G; ++gi) {
                double q = 0.0;
#pragma unroll
                for (int jb = 0; jb < NJ; ++jb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) q = fma(au[gi][jb][r], au[gi][jb][r], q);
                uu[gi] = q;
            }
        }
        // ---- w = L_Y^-1 (y - G x) ; rows of x out
#pragma unroll
        for (int gi = 0; gi < MV_G; ++gi)
#pragma unroll
            for (int jb = 0; jb < NJ; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[gi][4 * jb + r] = ax[gi][jb][r];
                    if (!COLL) am[gi][jb][r] = vYw[16 * jb + 4 * r];
                }
#pragma unroll
        for (int gi = 0; gi < MV_G; ++gi)
            if (valid[gi]) {
                SMC_GLOBAL(double) px = Xn + n[gi] * d + g;
                if (dfull) {
#pragma unroll
                    for (int kb = 0; kb < NV; ++kb) px[4 * kb] = v[gi][kb];
                } else {
#pragma unroll
                    for (int kb = 0; kb < NV; ++kb)
                        if (4 * kb + g < d) px[4 * kb] = v[gi][kb];
                }
            }
        if (!COLL && !DG) mv_product<DP, false>(sM + S_NGY * MM, v, am, lane);
        if (!COLL && DG) {
#pragma unroll
            for (int gi = 0; gi < MV_G; ++gi)
#pragma unroll
                for (int jb = 0; jb < NJ; ++jb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        am[gi][jb][r] = fma(vDg[2 * DP + 16 * jb + 4 * r], v[gi][4 * jb + r], am[gi][jb][r]);
        }
#pragma unroll
        for (int gi = 0; gi < MV_G; ++gi) {
            double ww = 0.0;
#pragma unroll
            for (int jb = 0; jb < NJ; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) ww = fma(am[gi][jb][r], am[gi][jb][r], ww);
            // the sums over the 4 lanes of a particle reach all of them; lane (n, g) keeps
            // those of the chunk's group q == g, so that after 4 groups lane l owns particle
            // l of the wave's 64 and the per-particle tail runs once, on full wavefronts
            const int q = it & 3;
            ww = mv_sum_g(ww);
            const double zs = mv_sum_g(zz[gi]);
            const double us = GUIDED3 ? mv_sum_g(uu[gi]) : 0.0;
            if (q == g) { kw = ww; kz = zs; ku = us; }
            if (q == 3) {
                const i64 np = particle(it - 3, 0) - pn + lane;
                const double* sc = sVec + 3 * DP;
                double inc = -0.5 * kw - sc[1];                                // kalman.py:345-346
                if (GUIDED3)                                                   // ssm.py:380-392
                    inc = ((-0.5 * ku - sc[first ? 3 : 0]) + inc) - (-0.5 * kz - sc[first ? 4 : 2]);
                if (COLL) inc = first ? sc[6] : -0.5 * kw - sc[5];             // log p(y_t | x_{t-1})
                if (np < N) {
                    double lw;
                    if (APF && !first) {
                        // core.py:299-305 reset_weights: log_mean_exp(logeta, W) - logeta[A] (the constant comes
                        // with the record, k_mv_aux_restate); not resampled: the plain weight k_mv_aux set aside
                        const double plain = a.lwsv[(i64)isl * N + np];
                        const double prev = resample ? smc_uniform(info[6]) - a.eta[(i64)isl * N + A[np]] : plain;
                        lw = prev + inc;
                        // history slots: step t-1's slot gets its PLAIN weights back (k_mv_aux put lw + eta there for the
                        // resampling kernels, which are done: a launch boundary lies in between)
                        if (a.hist) (f_lw(a, t - 1) + (i64)isl * N)[np] = plain;
                    } else
                    lw = (resample || first) ? inc : lwprev + inc;             // resampling.py:241-244
                    if (lw != lw) lw = -INFINITY;                              // resampling.py:220
                    lwn[np] = lw;
                    smc_lse_push(lacc, lw);
                }
            }
        }
    }
    f_step_tail(a, isl, b, t, first, resample, smc_lse_block(lacc, smd), smd, s_last, info);
}

// ---- AuxiliaryPF (see the top): eta_i = logeta(t-1, X_{t-1,i}) with y_t, the auxiliary log-weights in place
// of the plain ones (which wait in a.lwsv), the workgroup's log-sum-exp partial of them.  Same blocking as
// the propagate kernel (nparts partials per island).
template <int DP, bool DFULL>
__global__ void __launch_bounds__(SMC_BLOCK)
k_mv_aux(const FArgs av, const double* __restrict__ C)
{
    constexpr int NV = DP / 4, NJ = DP / 16, MM = DP * DP;
    const FArgs& a = av;
    __shared__ double sM[MM];
    __shared__ double sY[DP];
    __shared__ double smd[SMC_SM];
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y, tid = (int)threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, g = lane >> 4, pn = lane & 15;
    const double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)smc_uniform(info[0]);                  // the step about to run
    if (t >= a.T || t == 0) return;
    const i64 N = a.N;
    const int d = a.dx;
    const double* Xo = f_X(a, t - 1) + (i64)isl * N * d;
    double* lw = f_lw(a, t - 1) + (i64)isl * N;
    const double* ys = C + MV_STEP(DP) + (size_t)t * MV_NSTEPV * DP + 2 * DP;     // L_S^-1 y_t
    const double cS = C[MV_SCAL(DP) + 5];
    for (int i = tid; i < MM; i += SMC_BLOCK) sM[i] = C[MV_NGF * MM + i];
    if (tid < DP) sY[tid] = ys[tid];
    __syncthreads();
    SmcLse lacc = smc_lse_empty();
    double kw = 0.0;
    const int nit = a.mv_chunks * 4;
#pragma unroll 1
    for (int it = 0; it < nit; ++it) {
        const i64 n = ((i64)b * a.mv_chunks + it / 4) * SMC_BLOCK + (i64)wv * 64 + (i64)(it & 3) * 16 + pn;
        const bool valid = n < N;
        double v[1][NV];
        smc_v4d acc[1][NJ];
        const double* pr = Xo + (valid ? n : 0) * d;
#pragma unroll
        for (int kb = 0; kb < NV; ++kb) {
            const int k = 4 * kb + g;
            const bool kin = DFULL || k < d;
            const double x = pr[kin ? k : 0];
            v[0][kb] = (valid && kin) ? x : 0.0;
        }
#pragma unroll
        for (int jb = 0; jb < NJ; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[0][jb][r] = sY[16 * jb + 4 * r + g];
        mv_product<DP, false>(sM, v, acc, lane);                        // w = L_S^-1 (y_t - G F x)
        double ww = 0.0;
#pragma unroll
        for (int jb = 0; jb < NJ; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) ww = fma(acc[0][jb][r], acc[0][jb][r], ww);
        ww = mv_sum_g(ww);
        const int q = it & 3;
        if (q == g) kw = ww;
        if (q == 3) {                     // lane l now owns particle l of the wave's 64
            const i64 np = ((i64)b * a.mv_chunks + it / 4) * SMC_BLOCK + (i64)wv * 64 + lane;
            if (np < N) {
                const double eta = -0.5 * kw - cS;                      // kalman.py:358-361
                const double l = lw[np];
                double la = l + eta;                                    // core.py:307-313 (Weights.add)
                if (la != la) la = -INFINITY;
                a.eta[(i64)isl * N + np] = eta;
                a.lwsv[(i64)isl * N + np] = l;
                lw[np] = la;
                smc_lse_push(lacc, la);
            }
        }
    }
    const SmcLse r = smc_lse_block(lacc, smd);
    if (tid == 0) {
        const i64 o = (i64)isl * a.nparts;
        a.pm[o + b] = r.m;
        a.ps[o + b] = r.s;
        a.pss[o + b] = r.ss;
    }
}
// one workgroup per island: the auxiliary weights' (max, sum, sum of squares) -> ESS, the decision of step t
// (fk.time_to_resample on aux.ESS, core.py:181-183), the normalisation the resampling kernels use and the
// reset constant log_mean_exp(eta, W) = log-mean of the auxiliary weights - log-mean of the plain ones
__global__ void __launch_bounds__(SMC_BLOCK)
k_mv_aux_restate(const FArgs av)
{
    const FArgs& a = av;
    __shared__ double smd[SMC_SM];
    const int isl = (int)blockIdx.x;
    double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)smc_uniform(info[0]);
    if (t >= a.T || t == 0) return;
    const i64 o = (i64)isl * a.nparts;
    const SmcLse g = smc_lse_reduce_partials<true>(a.pm + o, a.ps + o, a.pss + o, a.nparts, smd);
    if (threadIdx.x == 0) {
        const bool bad = !(g.m > -INFINITY) || !(g.m < INFINITY);
        const double ess = bad ? NAN : (g.s * g.s) / g.ss;
        const double log_mean = bad ? NAN : g.m + log(g.s / (double)a.N);
        const double* row = a.summ + ((i64)isl * (a.T + 1) + (t - 1)) * SUMM_STRIDE;
        info[1] = (ess < a.ess_thresh) ? 1.0 : 0.0;
        info[3] = g.m;
        info[4] = bad ? NAN : 1.0 / g.s;
        info[6] = log_mean - row[1];
    }
}

// X_{t-1}[A] for SMC.Xp, (N,d)
__global__ void __launch_bounds__(SMC_BLOCK)
k_f_gather_rows(const double* X, const u32* A, i64 N, int d, double* Xp)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < N * d) {
        const i64 nn = i / d;
        Xp[i] = X[(i64)A[nn] * d + (i - nn * d)];
    }
}

// ---------------------------------------------------------------------------
// host: derived constants (dense d x d algebra, d <= 32)
// ---------------------------------------------------------------------------
#include <vector>
namespace mvh {
typedef std::vector<double> Mat;     // row-major

inline bool chol(const Mat& A, int n, Mat& L)
{
    L.assign((size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
        if (!(s > 0.0)) return false;
        L[j * n + j] = sqrt(s);
        for (int i = j + 1; i < n; ++i) {
            double v = A[i * n + j];
            for (int k = 0; k < j; ++k) v -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = v / L[j * n + j];
        }
    }
    return true;
}
inline Mat tri_inv(const Mat& L, int n)
{
    Mat R((size_t)n * n, 0.0);
    for (int c = 0; c < n; ++c)
        for (int i = c; i < n; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) s -= L[i * n + k] * R[k * n + c];
            R[i * n + c] = s / L[i * n + i];
        }
    return R;
}
inline Mat mul(const Mat& A, int ra, int ca, const Mat& B, int cb)   // (ra,ca)(ca,cb)
{
    Mat C((size_t)ra * cb, 0.0);
    for (int i = 0; i < ra; ++i)
        for (int k = 0; k < ca; ++k) {
            const double v = A[i * ca + k];
            for (int j = 0; j < cb; ++j) C[i * cb + j] += v * B[k * cb + j];
        }
    return C;
}
inline Mat tr(const Mat& A, int r, int c)
{
    Mat T((size_t)r * c);
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) T[j * r + i] = A[i * c + j];
    return T;
}
inline double logdiag(const Mat& L, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += log(L[i * n + i]);
    return s;
}
// Kalman gain and filtered covariance for predictive covariance P
// (kalman.py:215-229): S = G P G' + R ; K = P G' S^-1 ; Pf = P - K G P
inline bool gain(const Mat& P, const Mat& G, const Mat& R, int dx, int dy, Mat& K, Mat& Pf)
{
    Mat GP = mul(G, dy, dx, P, dx);                       // (dy,dx)
    Mat S = mul(GP, dy, dx, tr(G, dy, dx), dy);           // (dy,dy)
    for (int i = 0; i < dy * dy; ++i) S[i] += R[i];
    Mat Ls;
    if (!chol(S, dy, Ls)) return false;
    Mat Li = tri_inv(Ls, dy);
    Mat Sinv = mul(tr(Li, dy, dy), dy, dy, Li, dy);       // S^-1 = L^-T L^-1
    K = mul(tr(GP, dy, dx), dx, dy, Sinv, dy);            // (dx,dy)
    Mat KGP = mul(K, dx, dy, GP, dx);
    Pf = P;
    for (int i = 0; i < dx * dx; ++i) Pf[i] -= KGP[i];
    // symmetrise (round-off)
    for (int i = 0; i < dx; ++i)
        for (int j = 0; j < i; ++j) Pf[i * dx + j] = Pf[j * dx + i] = 0.5 * (Pf[i * dx + j] + Pf[j * dx + i]);
    return true;
}
// store M (r x c), zero-padded to dp x dp, as MFMA A-fragments (layout at the top)
inline void put_frag(double* dst, int dp, const Mat& M, int r, int c, double sign = 1.0)
{
    for (int jb = 0; jb < dp / 16; ++jb)
        for (int kb = 0; kb < dp / 4; ++kb)
            for (int lane = 0; lane < 64; ++lane) {
                const int row = 16 * jb + (lane & 15), col = 4 * kb + (lane >> 4);
                dst[(jb * (dp / 4) + kb) * 64 + lane] =
                    (row < r && col < c) ? sign * M[row * c + col] : 0.0;
            }
}
}  // namespace mvh

// Builds the constants block; returns false if a covariance is not positive definite
// (the reference raises ValueError in MvNormal.__init__, distributions.py:935-940).
inline bool mv_is_diag(const mvh::Mat& M, int r, int c)
{
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j)
            if (i != j && M[i * c + j] != 0.0) return false;
    return true;
}
inline void mv_put_diag(double* dst, const mvh::Mat& M, int r, int c, double sign = 1.0)
{
    for (int i = 0; i < r && i < c; ++i) dst[i] = sign * M[i * c + i];
}
// *diag (optional): every factor the propagate kernel applies after F / B is a diagonal matrix
inline bool mv_build_constants(const smc_model* m, int fk, int dp, i64 T, const double* y,
                               std::vector<double>& out, bool* diag = nullptr)
{
    using namespace mvh;
    const int dx = m->dx, dy = m->dy;
    Mat F(m->F_host, m->F_host + dx * dx), G(m->G_host, m->G_host + dy * dx);
    Mat QX(m->covX_host, m->covX_host + dx * dx), R(m->covY_host, m->covY_host + dy * dy);
    Mat Q0(m->cov0_host, m->cov0_host + dx * dx), mu0(m->mu0_host, m->mu0_host + dx);
    out.assign(MV_SIZE(dp, T), 0.0);
    double* C = out.data();
    Mat LX, LY, L0;
    if (!chol(QX, dx, LX) || !chol(R, dy, LY) || !chol(Q0, dx, L0)) return false;
    Mat LYi = tri_inv(LY, dy);
    Mat GY = mul(LYi, dy, dy, G, dx);                                     // (dy,dx)
    put_frag(C + MV_F * dp * dp, dp, F, dx, dx);
    put_frag(C + MV_NGY * dp * dp, dp, GY, dy, dx, -1.0);
    double* dgv = C + MV_DIAGV(dp);
    mv_put_diag(dgv + 4 * dp, GY, dy, dx, -1.0);
    bool dg = mv_is_diag(GY, dy, dx);
    double* scal = C + MV_SCAL(dp);
    scal[0] = logdiag(LX, dx) + dx * SMC_HALFLOG2PI;
    scal[1] = logdiag(LY, dy) + dy * SMC_HALFLOG2PI;
    scal[3] = logdiag(L0, dx) + dx * SMC_HALFLOG2PI;
    for (int i = 0; i < dx; ++i) C[MV_VEC(dp) + i] = mu0[i];
    Mat K, K0, LSi;
    if (fk == SMC_FK_GUIDED || fk == SMC_FK_APF) {
        Mat P, P0, LP, LP0;
        if (!gain(QX, G, R, dx, dy, K, P) || !gain(Q0, G, R, dx, dy, K0, P0)) return false;
        if (!chol(P, dx, LP) || !chol(P0, dx, LP0)) return false;
        // B = (I - K G) F
        Mat KG = mul(K, dx, dy, G, dx), IKG((size_t)dx * dx, 0.0);
        for (int i = 0; i < dx; ++i)
            for (int j = 0; j < dx; ++j) IKG[i * dx + j] = (i == j ? 1.0 : 0.0) - KG[i * dx + j];
        put_frag(C + MV_B * dp * dp, dp, mul(IKG, dx, dx, F, dx), dx, dx);
        put_frag(C + MV_LZ * dp * dp, dp, LP, dx, dx);
        put_frag(C + MV_LZ0 * dp * dp, dp, LP0, dx, dx);
        const Mat LXi = tri_inv(LX, dx), L0i = tri_inv(L0, dx);
        put_frag(C + MV_XINV * dp * dp, dp, LXi, dx, dx);
        put_frag(C + MV_X0INV * dp * dp, dp, L0i, dx, dx);
        mv_put_diag(dgv, LP, dx, dx);
        mv_put_diag(dgv + dp, LP0, dx, dx);
        mv_put_diag(dgv + 2 * dp, LXi, dx, dx);
        mv_put_diag(dgv + 3 * dp, L0i, dx, dx);
        dg = dg && mv_is_diag(LP, dx, dx) && mv_is_diag(LP0, dx, dx) && mv_is_diag(LXi, dx, dx) && mv_is_diag(L0i, dx, dx);
        scal[2] = logdiag(LP, dx) + dx * SMC_HALFLOG2PI;
        scal[4] = logdiag(LP0, dx) + dx * SMC_HALFLOG2PI;
        // collapsed form: S = G covX G' + covY, -(L_S^-1 G F), log p(y_0) = log N(y_0; G mu0, G cov0 G' + covY)
        {
            Mat GQ = mul(G, dy, dx, QX, dx), S = mul(GQ, dy, dx, tr(G, dy, dx), dy);
            Mat GQ0 = mul(G, dy, dx, Q0, dx), S0 = mul(GQ0, dy, dx, tr(G, dy, dx), dy);
            for (int i = 0; i < dy * dy; ++i) { S[i] += R[i]; S0[i] += R[i]; }
            Mat LS, LS0;
            if (!chol(S, dy, LS) || !chol(S0, dy, LS0)) return false;
            LSi = tri_inv(LS, dy);
            put_frag(C + MV_NGF * dp * dp, dp, mul(LSi, dy, dy, mul(G, dy, dx, F, dx), dx), dy, dx, -1.0);
            scal[5] = logdiag(LS, dy) + dy * SMC_HALFLOG2PI;
            Mat LS0i = tri_inv(LS0, dy);
            double q = 0.0;
            for (int i = 0; i < dy; ++i) {
                double v = 0.0;
                for (int j = 0; j <= i; ++j) {
                    double r = y[j];
                    for (int k = 0; k < dx; ++k) r -= G[j * dx + k] * mu0[k];
                    v += LS0i[i * dy + j] * r;
                }
                q += v * v;
            }
            scal[6] = -0.5 * q - (logdiag(LS0, dy) + dy * SMC_HALFLOG2PI);
        }
        // proposal0 mean: mu0 + K0 (y0 - G mu0)           (kalman.py:353-356)
        for (int i = 0; i < dx; ++i) {
            double v = mu0[i];
            for (int j = 0; j < dy; ++j) {
                double r = y[j];
                for (int k = 0; k < dx; ++k) r -= G[j * dx + k] * mu0[k];
                v += K0[i * dy + j] * r;
            }
            C[MV_VEC(dp) + dp + i] = v;
        }
    } else {
        put_frag(C + MV_LZ * dp * dp, dp, LX, dx, dx);
        put_frag(C + MV_LZ0 * dp * dp, dp, L0, dx, dx);
        mv_put_diag(dgv, LX, dx, dx);
        mv_put_diag(dgv + dp, L0, dx, dx);
        dg = dg && mv_is_diag(LX, dx, dx) && mv_is_diag(L0, dx, dx);
    }
    if (diag) *diag = dg;
    for (i64 t = 0; t < T; ++t) {
        double* yw = C + MV_STEP(dp) + (size_t)t * MV_NSTEPV * dp;
        const double* yt = y + t * dy;
        for (int i = 0; i < dy; ++i) {
            double v = 0.0;
            for (int j = 0; j <= i; ++j) v += LYi[i * dy + j] * yt[j];
            yw[i] = v;
        }
        if (fk == SMC_FK_GUIDED || fk == SMC_FK_APF) {
            for (int i = 0; i < dx; ++i) {
                double v = 0.0;
                for (int j = 0; j < dy; ++j) v += K[i * dy + j] * yt[j];
                yw[dp + i] = v;
            }
            for (int i = 0; i < dy; ++i) {
                double v = 0.0;
                for (int j = 0; j <= i; ++j) v += LSi[i * dy + j] * yt[j];
                yw[2 * dp + i] = v;
            }
        }
    }
    return true;
}
