// smc_filter_kernels.h -- device side of the fused SMC step loop.
//
// Replaces, for the closed model family of the hot-path configs, the body of
// particles.SMC.__next__ (particles/core.py:369-383):
//     setup_auxiliary_weights / resample_move  (core.py:307-337)
//     reweight_particles                        (core.py:323-324)
//     compute_summaries                         (core.py:351-359)
// for `n_islands` independent filters advancing in lock step.
//
// Kernels of a time step (no host round trip; the argument block FArgs travels
// by value in the kernarg segment):
//
//   k_ancestors<FUSED>(t): integer-only resampling, only when step t resamples.
//       One workgroup per tile of 1024 consecutive parents: Q62 weights
//       q_i = rint(exp(lw_i-m)/s * 2^62), the tile's exact CDF (tile totals are
//       exchanged between workgroups, see the kernel), the contiguous range of
//       offspring the tile owns (smc_resample.h) and the parent of each, 4 per
//       thread per pass:
//         systematic, N a power of two: closed-form first-offspring index per
//           parent, scattered into LDS and expanded by a max-scan (no search);
//         otherwise: per-offspring binary search in the tile's CDF in LDS.
//   [k_prepare(t): beyond 2048 workgroups per launch the tile totals are computed and
//       scanned by a launch of their own; k_ancestors<false> reads the scanned prefixes]
//   k_propagate<KIND,FK,OPT>(t): element-wise over the new particles: gather of
//       the parent state X_{t-1}[A], x = loc(xp)+scale*z with a counted Philox
//       normal (or a replayed draw), the weight increment log G, 32-byte stores
//       of (X, lw) and the online log-sum-exp partial.  The LAST workgroup of
//       each island to finish (two-level tickets) reduces the partials to
//       (max, sum, sum of squares) -> ESS, log-mean weight, loglt/logLt of step
//       t (core.py:355-359), the resample decision of step t+1 (core.py:181-183)
//       and writes the 64-byte step record the next launches read.
//   (k_propagate_mv in smc_filter_mv.h is the multivariate counterpart.)
//   [k_reduce2] -> k_ancestors2 -> k_propagate<.., TAIL = false>: the same step on the
//       two-level exact CDF (systematic / stratified, N = 2^k, 2..1024 tiles per island):
//       no exchange inside any launch -- see "Two-level CDF" below.
//
// The time index lives in that device-resident record, so the same launches
// -- or one hipGraph holding many of them -- serve every step.
//
// HBM traffic per particle-step on a resampling step (d = 1):
//   k_ancestors: read lw (8), write A (8);  k_propagate: read A, X (16), write X, lw (16)
// = 48 B (+8 B when k_prepare reads lw for the totals) against SURVEY 8d's 56 B: neither
// W nor its fixed-point image is ever materialised.
#pragma once
#include "smc_internal.h"
#include "smc_resample.h"

#define F_HMAX 64      /* heavy parents registered per island and step */
#define F_HLOC 8       /* ... per tile */
#define F_IPT 4
#define F_TILE (SMC_BLOCK * F_IPT)
#define F_PASS (SMC_BLOCK * 4)      /* offspring per pass: 4 per thread */
#define SUMM_STRIDE 8   /* ESS, log_mean, loglt, logLt, rs_flag, m, 1/s, - */
#define PARAM_STRIDE 32  /* 16 model constants (the host's params row), then RN(1 / constant) of each */
#ifdef SMC_PARAMS_IN_GLOBAL              /* (A/B builds: k_propagate reading the constants through the argument's pointer) */
#define SMC_PARAMS_GLOBAL 1
#else
#define SMC_PARAMS_GLOBAL 0
#endif
#define PARAM_HOST 16
#define INFO_STRIDE 8   /* per-island step record: t, rs_flag, y_t, m, 1/s of step t-1, aux_t */

struct FArgs {
    i64 N, T;
    int ntiles, n_islands, scheme, rng_mode, island_offset;
    int nparts;            // workgroups of k_propagate per island (= LSE partials)
    int log2N;             // k if N == 2^k, else -1
    double ess_thresh;
    u64 seed;
    // states / log-weights / ancestors of step t live in slot f_slot(t): two alternating
    // slots, or -- keep_history -- one slot per time step (the history IS the buffer)
    double *X, *lw;        // (nslots, n_islands, N[, dx])
    u32* A;                // (1 or T, n_islands, N) ancestors, 32-bit in HBM (N < 2^32; the
                           // ABI hands out int64 like the reference, resampling.py:503)
    i64 xslot, lslot;      // elements per slot: n_islands*N*dx, n_islands*N
    int hist;              // 0: two alternating slots; 1: one slot per time step (the whole history
                           // stays resident); k >= 2: a ring of k slots (the k most recent steps)
    int par;               // t & 1 of the step this launch runs, or -1 (history slots): lets the
                           // kernels form their addresses before the step record has arrived
    i64 tk;                // the time index the HOST expects this launch to run (eager launches), or -1
                           // (inside a captured graph): work that depends on t only -- the step's
                           // normals -- starts on it while the step record is still on its way, and is
                           // redone with the record's t if the two differ (same bits either way)
    u64* Q;                // (n_islands, ntiles) tile totals of q
    u64* Qpre;             // (n_islands, ntiles) exclusive prefixes of Q
    double *pm, *ps, *pss; // (n_islands, ntiles) log-sum-exp partials (two-level path: K_b, S_b, SS_b)
    u64* cq;               // two-level path: (n_islands, N) exclusive integer CDF of every particle in its tile
    u64* tq;               // two-level path: (n_islands, ntiles) the tiles' integer totals t_b
    unsigned* cnt;         // (n_islands, 2, F_CNT_WORDS) completion tickets: k_propagate, k_prepare
    double* spart;         // (n_islands, 3, 32) shard-level log-sum-exp partials
    double* summ;          // (n_islands, T+1, SUMM_STRIDE)
    const double* params;  // (n_islands, PARAM_STRIDE)
    const double* y;       // (T,)
    double* mom;           // (n_islands, T, 2*dx) weighted mean | variance per step, or null
    double* mpart;         // (n_islands, nmb, dx, 3) partial sums of k_f_moments_partials
    int nmb;
    const double* aux;     // (T,) per-step scalar of the transition (GORDON: d cos(e (t-1))) or null
    double* info;          // (n_islands, INFO_STRIDE) record of the step being run
    // heavy parents (>= 2048 offspring: at least one whole 1024-block of offspring is theirs):
    // registered by the ancestors kernel, which then skips those blocks; k_propagate fills them
    unsigned* hcnt;        // (n_islands, 2) entries per parity of t, or null (feature off)
    i64* hlist;            // (n_islands, 2, F_HMAX, 3): first offspring, one past the last (both
                           // multiples of 1024), parent
    int nt;                // bulk stores of X, lw, A as streaming (`nt`) stores: resident launches
    int exact_counts;      // two-level path, tests: always form c Q_b / t_b exactly (no fp64 band shortcut)
    double* info2;         // two-level path: (n_islands, INFO_STRIDE) [0] = t for k_ancestors2 (written by
                           // k_propagate's workgroup 0; `info` then is written by k_ancestors2's)
    const double* zt;      // replay normals (T, n_islands, N) or null
    const double* ut;      // replay uniforms (T, n_islands, K) or null
    i64 ut_stride;         // K
    double* su;            // multinomial, Philox mode: (n_islands, N) sorted uniforms
    u64* E;                // multinomial, Philox mode: spacing tile sums (n_islands, ntiles1)
    int ntiles1;
    double spacing_scale;
    int dx, dy, dp;        // state / observation dimension, dx padded to 4, 8, 16 or 32
    int mv_chunks;         // 256-particle chunks per workgroup of k_propagate_mv
    const double* mvc;     // MVLINGAUSS: derived constants (see smc_filter_mv.h)
    u64* trace;            // SMC_TRACE builds: shader-clock stamps of the last step's workgroups
    // (late additions live at the END of the block: k_propagate sits at the scalar-register limit, and
    //  a field inserted among the ones it loads made the compiler spill SGPRs in its prologue -- an
    //  early s_waitcnt on the kernarg loads, 0.3 us per launch; profiles/r03n A/B)
    int kform;             // two-level step: the row's (K, 1/s) normalise weights kept as (p, k) pairs, and
                           // the count of steps done lives in info2 (the side kernels: moments)
    i64 ncq;               // per-island stride of cq: ntiles x 1024 (the last tile may be ragged)
    u64* sst;              // one-pass uniform_spacings: (n_islands, sp_nwg) look-back status words, 0 between steps
    int sp_tpw, sp_nwg;    // ... tiles of draws per workgroup (0: the three-pass form) and workgroups per island;
                           // `su` then holds the integer prefix sums Z_n (u64), not the quotients
    double *pm2, *ps2, *pss2;   // APF on the two-level path: the tile partials of the PLAIN weights
                           // (pm/ps/pss, cq, tq then describe the AUXILIARY weights lw + logeta,
                           // which decide and drive the resampling -- core.py:307-313); else null
    i64 zt_ts;             // elements per time step of the tape zt: n_islands x N x dx (replay), 0: ONE step's buffer,
                           // rewritten before every step (SMC_FLAG_SQMC: ndtri of the step's second Sobol' coordinates)
    u64 sq_seed, sq_ctr;   // SMC_FLAG_SQMC (smc_filter_sqmc.h): the point stream's key and the point set of step t = sq_ctr + t
    u64* sdec;             // one-pass uniform_spacings with the island's reduction as its workgroup 0 (sp_epoch != 0):
    u64 sp_epoch;          // (n_islands) decision words, (epoch << 2) | 2 resample, | 1 not; epoch: unique per launch
    const u64* sq_perm;    // SMC_FLAG_SQMC: (n_islands, N) h_order of the step being run -- k_ancestors2<SQ> counts in sorted
                           // positions and stores h_order[position] (core.py:344: A = h_order[inverse_cdf(...)])
    double *eta, *lwsv;    // APF of MVLINGAUSS (smc_filter_mv.h k_mv_aux): (n_islands, N) logeta of the step's
                           // parents and their plain log-weights, set aside while lw + eta drives the resampling
    int xcd_chunks;        // two-level step: workgroup -> tile map that keeps CONSECUTIVE tiles on one XCD (f_tile_xcd)
    int mv_diag;           // MVLINGAUSS: G, covX, covY, cov0 are all diagonal -- the three triangular factors of the step are
                           // diagonal too and k_propagate_mv<..., DG> applies them element by element (smc_filter_mv.h)
    int strict_e;          // SMC_FLAG_STRICT_ANCESTORS on the two-level step: nobody reads the tile's integer CDF, so `cq` holds
                           // every particle's weight on its TILE's scale instead, e_j = p_j 2^(k_j - K_b) (a double): what
                           // k_strict_classify forms W_j from without evaluating an exponential again
};
// Workgroups go round the 8 XCDs (each with its own L2): with tile = workgroup index, neighbouring tiles sit on
// different XCDs -- but a tile's offspring start in the tile next door as soon as the weights drift, and their parents'
// states were written there: k_propagate then reads A and gathers X through another XCD's L2.  With the map below
// XCD x owns a contiguous run of tiles: neighbours share an L2 (C2: r12v A/B).  Any bijection is correct -- a workgroup
// just processes the tile the map gives it.  Workgroup bx sits on XCD c = bx mod 8 and is that XCD's q-th (q = bx / 8);
// XCD c receives ceil((ntiles - c) / 8) workgroups, so its run starts at c floor(ntiles / 8) + min(c, ntiles mod 8):
// a bijection for ANY number of tiles (round sizes like N = 10^6 included).
__device__ __forceinline__ int f_tile_xcd_geom(const int geom, const int bx)      // geom: ntiles | xcd_chunks << 30
{
    if (!(geom >> 30)) return bx;
    const int ntiles = geom & 0x3FFFFFFF;
    const int c = bx & 7, base = ntiles >> 3, rem = ntiles & 7;
    return c * base + (c < rem ? c : rem) + (bx >> 3);
}
__device__ __forceinline__ int f_tile_xcd(const FArgs& a, const int bx)
{
    if (!a.xcd_chunks) return bx;
    const int c = bx & 7, base = a.ntiles >> 3, rem = a.ntiles & 7;
    return c * base + (c < rem ? c : rem) + (bx >> 3);
}

__host__ __device__ __forceinline__ i64 f_slot(const FArgs& a, i64 t)
{
    if (a.hist == 0) return t & 1;
    if (t < 0) t = 0;
    return a.hist == 1 ? t : (i64)((u32)t % (u32)a.hist);
}
__host__ __device__ __forceinline__ double* f_X(const FArgs& a, i64 t) { return a.X + f_slot(a, t) * a.xslot; }
__host__ __device__ __forceinline__ double* f_lw(const FArgs& a, i64 t) { return a.lw + f_slot(a, t) * a.lslot; }
__host__ __device__ __forceinline__ u32* f_A(const FArgs& a, i64 t) { return a.A + (a.hist ? f_slot(a, t) : 0) * a.lslot; }

#ifdef SMC_TRACE
#define F_STAMP(k) do { if (threadIdx.x == 0) a.trace[((i64)blockIdx.y * a.nparts + blockIdx.x) * 8 + (k)] = (u64)wall_clock64(); } while (0)
#define F_STAMP_A(k) do { if (threadIdx.x == 0) a.trace[((i64)a.n_islands * a.nparts + (i64)blockIdx.y * a.ntiles + blockIdx.x) * 8 + (k)] = (u64)wall_clock64(); } while (0)
// (the one-pass spacings kernel, island 0: the rows the strict step's stamps use in strict filters -- smc_debug_trace_strict)
#define F_STAMP_S(k) do { if (threadIdx.x == 0 && blockIdx.y == 0) a.trace[((i64)a.n_islands * (a.nparts + a.ntiles) + blockIdx.x) * 8 + (k)] = (u64)wall_clock64(); } while (0)
#else
#define F_STAMP(k) do { } while (0)
#define F_STAMP_A(k) do { } while (0)
#define F_STAMP_S(k) do { } while (0)
#endif

// a / b for a model constant b whose correctly rounded reciprocal rb = RN(1/b) sits next to it in
// the params row (computed on the host): q0 = RN(a rb) is within 1 ulp of a/b, r = a - b q0 is
// exact in the fma, and RN(q0 + r rb) = RN(a / b) (Markstein's theorem; 4e8 random and
// adversarial pairs checked against the division in tools/micro/markstein.c) -- the IEEE
// quotient, i.e. the reference's bits, in 3 instructions instead of 11.  Outside the range
// where nothing can overflow or go subnormal on the way (a = 0, inf, NaN included; rb = 0
// marks a divisor the host would not vouch for) the division itself is executed.
__device__ __forceinline__ double smc_div_c(const double a, const double b, const double rb)
{
    const double q0 = a * rb;
    const double r = fma(-q0, b, a);
    double q = fma(r, rb, q0);
    const double m = fabs(q0);
    if (__builtin_expect(!(m > 1e-280 && m < 1e280), 0)) {
        // (a real branch: left as a plain `if` the compiler executes the division for everybody and selects --
        //  13 instructions per quotient on the common path, v_rcp_f64 among them; the operands pass through an
        //  empty asm so that nothing of it can be hoisted)
        double aa = a, bb = b;
#ifndef SMC_EMULATE
        asm volatile("" : "+v"(aa), "+v"(bb));
#endif
        q = aa / bb;
    }
    return q;
}

// ---------------------------------------------------------------------------
// model family
// ---------------------------------------------------------------------------
// params row p[16] (host: particles_amd/*._device_params):
//   LINGAUSS  rho, sigmaX, sigmaY, sigma0, log sigmaY, ... guided constants
//   STOCHVOL  mu, rho, sigma, sig0, (1-rho) mu
//   GORDON    b, sigmaX, c, sigma0 (= 2), -, a          aux_t = d cos(e (t-1))
//   THETALOG  tau0, sigmaX, sigmaY, sigma0 (= 1), log sigmaY, tau1, tau2
//   SVLEVERAGE as STOCHVOL + 5 phi, 6 sqrt(1 - phi^2)
//   DISCRETECOX mu, phi, sigma, sig0                   aux_t = gammaln(y_t + 1)
// models with a per-step scalar aux_t (FArgs::aux, slot 5 of the step record)
template <int KIND>
__host__ __device__ constexpr bool m_has_aux()
{
    return KIND == SMC_MODEL_GORDON || KIND == SMC_MODEL_DISCRETECOX;
}
template <int KIND>
__device__ __forceinline__ double m_trans_loc(const double* p, double xp, double aux)
{
    if (KIND == SMC_MODEL_LINGAUSS) return p[0] * xp;           // kalman.py:430-431
    if (KIND == SMC_MODEL_GORDON)                               // state_space_models.py:568-574
        return (p[0] * xp + (p[2] * xp) / (1.0 + xp * xp)) + aux;
    if (KIND == SMC_MODEL_THETALOGISTIC)                        // state_space_models.py:677-680
        return (xp + p[0]) - p[5] * exp(p[6] * xp);
    if (KIND == SMC_MODEL_DISCRETECOX)                          // state_space_models.py:626-627
        return p[0] + p[1] * (xp - p[0]);
    return p[4] + p[1] * xp;                                    // state_space_models.py:465-470
}
template <int KIND>
__device__ __forceinline__ double m_trans_scale(const double* p)
{
    return (KIND == SMC_MODEL_STOCHVOL || KIND == SMC_MODEL_SVLEVERAGE ||
            KIND == SMC_MODEL_DISCRETECOX) ? p[2] : p[1];
}
template <int KIND>
__device__ __forceinline__ double m_init_loc(const double* p)
{
    return (KIND == SMC_MODEL_STOCHVOL || KIND == SMC_MODEL_SVLEVERAGE ||
            KIND == SMC_MODEL_DISCRETECOX) ? p[0] : 0.0;       // ssm.py:462, :621 ; kalman.py:427, ssm.py:565, :675
}
// log p(y_t | x_t) as scipy.stats.norm.logpdf evaluates it
template <int KIND>
__device__ __forceinline__ double m_obs_logpdf(const double* p, double y, double x, double xp,
                                               bool first, double aux)
{
    if (KIND == SMC_MODEL_DISCRETECOX) {                        // ssm.py:629-630, distributions.py:528-529
        // scipy.stats.poisson._logpmf: xlogy(k, mu) - gammaln(k + 1) - mu, mu = exp(x); log(mu) is x itself
        // (as for StochVol below: one exp, no log; equal to the 1-2 ulp of either route)
        const double mu = exp(x);
        const double xl = (y == 0.0 && mu == mu) ? 0.0 : y * x;
        return (xl - aux) - mu;
    }
    if (KIND == SMC_MODEL_SVLEVERAGE) {                         // ssm.py:531-541
        const double u = first ? smc_div_c(x - p[0], p[3], p[16 + 3])
                               : smc_div_c(x - (p[4] + p[1] * xp), p[2], p[16 + 2]);
        const double sx = exp(0.5 * x);
        const double loc = sx * p[5] * u, sc = sx * p[6];
        const double v = (y - loc) / sc;
        return -(v * v) / 2.0 - SMC_C_NORM - log(sc);
    }
    if (KIND == SMC_MODEL_LINGAUSS || KIND == SMC_MODEL_THETALOGISTIC) {   // kalman.py:433-434, ssm.py:682-683
        const double v = smc_div_c(y - x, p[2], p[16 + 2]);
        return -(v * v) / 2.0 - SMC_C_NORM - p[4];
    }
    if (KIND == SMC_MODEL_GORDON) {                             // ssm.py:576-577: Normal(loc=a x^2), scale 1
        const double v = (y - p[5] * (x * x)) / 1.0;
        return -(v * v) / 2.0 - SMC_C_NORM - 0.0;
    }
    // StochVol, ssm.py:472-473: Normal(loc=0, scale=exp(x / 2)).logpdf(y) = -v^2 / 2 - log sqrt(2 pi) - log(scale),
    // v = y / scale.  scipy forms scale, divides by it and takes its log; here log(exp(h)) is h itself and the
    // quotient is y exp(-h): one exp instead of exp + division + log (a quarter of k_propagate's arithmetic for
    // this model), the same value to within the 1-2 ulp of either route (tests: 1e-12 on the log-weights).
    const double h = 0.5 * x;
    const double v = y * exp(-h);
    return -(v * v) / 2.0 - SMC_C_NORM - h;
}
__device__ __forceinline__ double m_norm_logpdf(double x, double loc, double scale, double rscale,
                                                double lscale)
{
    const double v = smc_div_c(x - loc, scale, rscale);
    return -(v * v) / 2.0 - SMC_C_NORM - lscale;
}

// StochVol, Pitt & Shephard (state_space_models.py:475-498): the proposal's mean
// _xhat(xst, sig, y) = xst + 0.5 sig^2 (y^2 exp(-xst) - 1), half_s2 = 0.5 * sig ** 2 from the host
__device__ __forceinline__ double m_sv_xhat(double xst, double half_s2, double y)
{
    return xst + half_s2 * ((y * y) * exp(-xst) - 1.0);
}
// ... and the auxiliary function logeta(t, x) with y = data[t + 1] (:491-498)
__device__ __forceinline__ double m_sv_logeta(const double* p, double x, double y)
{
    const double xst = p[4] + p[1] * x;
    const double xstmmu = xst - p[0];
    const double xhatmmu = m_sv_xhat(xst, p[7], y) - p[0];
    return p[9] * (xhatmmu * xhatmmu - xstmmu * xstmmu) - ((0.5 * (y * y)) * exp(-xst)) * (1.0 + xstmmu);
}

// the auxiliary function logeta(t, x) of the models whose APF is fused, y = data[t + 1]:
//   STOCHVOL  Pitt & Shephard's (state_space_models.py:491-498)
//   LINGAUSS  log N(y; rho x, sqrt(sigmaX^2 + sigmaY^2)) (kalman.py:448-452) as scipy evaluates it;
//             p[15] = that scale (from the host, with its rounding), its log formed here
template <int KIND>
__device__ __forceinline__ double m_logeta(const double* p, double x, double y)
{
    if (KIND == SMC_MODEL_LINGAUSS) {
        const double v = smc_div_c(y - p[0] * x, p[15], p[16 + 15]);
        return -(v * v) / 2.0 - SMC_C_NORM - log(p[15]);
    }
    return m_sv_logeta(p, x, y);
}

// auxiliary filters: AuxiliaryPF (guided move) and AuxiliaryBootstrap (bootstrap move) share the auxiliary weights
__host__ __device__ constexpr bool f_is_apf(const int fk) { return fk == SMC_FK_APF || fk == SMC_FK_APF_BOOT; }
__host__ __device__ constexpr bool f_boot_move(const int fk) { return fk == SMC_FK_BOOTSTRAP || fk == SMC_FK_APF_BOOT; }
// one particle of one step: returns the new state, writes the weight increment
template <int KIND, int FK>
__device__ __forceinline__ double m_step(const double* p, bool first, double y, double aux,
                                         double xp, double z, double& inc)
{
    if (KIND == SMC_MODEL_STOCHVOL && !f_boot_move(FK)) {
        // GuidedPF / AuxiliaryPF of StochVol (state_space_models.py:374-392 with :481-489)
        const double xst = first ? 0.0 : p[4] + p[1] * xp;             // proposal0 centres _xhat at 0 (:483)
        const double sc = first ? p[3] : p[2], rsc = first ? p[16 + 3] : p[16 + 2];
        const double lsc = first ? p[6] : p[5];
        const double xhat = m_sv_xhat(xst, first ? p[8] : p[7], y);
        const double x = xhat + sc * z;
        const double prior_loc = first ? p[0] : xst;                   // PX0 = N(mu, sig0), PX = N(EXt(xp), sigma)
        inc = (m_norm_logpdf(x, prior_loc, sc, rsc, lsc) + m_obs_logpdf<KIND>(p, y, x, xp, first, aux))
              - m_norm_logpdf(x, xhat, sc, rsc, lsc);
        return x;
    }
    if (f_boot_move(FK)) {
        const double x = first ? m_init_loc<KIND>(p) + p[3] * z
                               : m_trans_loc<KIND>(p, xp, aux) + m_trans_scale<KIND>(p) * z;
        inc = m_obs_logpdf<KIND>(p, y, x, xp, first, aux);
        return x;
    }
    // guided filter with LinearGauss' optimal proposal (kalman.py:436-446,
    // state_space_models.py:374-392)
    if (first) {
        const double mu = p[12] * smc_div_c(y, p[8], p[16 + 8]);
        const double x = mu + p[13] * z;
        inc = (m_norm_logpdf(x, 0.0, p[3], p[16 + 3], p[6]) + m_obs_logpdf<KIND>(p, y, x, xp, first, aux))
              - m_norm_logpdf(x, mu, p[13], p[16 + 13], p[14]);
        return x;
    }
    const double mu = p[9] * (smc_div_c(p[0] * xp, p[7], p[16 + 7]) + smc_div_c(y, p[8], p[16 + 8]));
    const double x = mu + p[10] * z;
    inc = (m_norm_logpdf(x, p[0] * xp, p[1], p[16 + 1], p[5]) + m_obs_logpdf<KIND>(p, y, x, xp, first, aux))
          - m_norm_logpdf(x, mu, p[10], p[16 + 10], p[11]);
    return x;
}

// W_i as the device defines it: exp(lw_i - m) * (1/s)   (resampling.py:222,225)
__device__ __forceinline__ double f_weight(double lw, double m, double rs)
{
    return smc_exp_nonpos(lw - m) * rs;
}

// ---------------------------------------------------------------------------
// 4 consecutive elements per thread as 16-byte accesses when possible
// ---------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ void f_load4(const T* p, i64 j, i64 N, bool vec, T fill, T (&o)[4])
{
    if (vec && j + 3 < N) {
        smc_ld2g(p + j, o[0], o[1]);
        smc_ld2g(p + j + 2, o[2], o[3]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (j + i < N) ? smc_ldg(p + j + i) : fill;
    }
}
template <class T>
__device__ __forceinline__ void f_store4(T* p, i64 n, bool full_vec, const bool (&ok)[4],
                                         const T (&v)[4])
{
    if (full_vec) {
        smc_st2g(p + n, v[0], v[1]);
        smc_st2g(p + n + 2, v[2], v[3]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ok[i]) smc_stg(p + n + i, v[i]);
    }
}
// Ticket of the "last workgroup done" pattern (smc_device.h, "Publishing ..."):
// thread 0 has published this workgroup's values with smc_st_agent*, drains its
// stores and takes a ticket; the workgroup that draws the last ticket reads
// everybody's values with smc_ld_agent*.  Nobody spins, nobody fences.
// One atomic word sustains only ~90 returning atomics per microsecond on
// MI355X, so the ticket is two-level: 32 shard counters (64 B apart), whose
// last arrivers take a ticket on the top counter.
// Returns true in every thread of the last workgroup; it also re-arms the
// counters for the next launch.
// Up to this many tiles every k_ancestors workgroup adds up the totals of the
// tiles before it by itself (<= 8 loads per thread, in parallel with its other
// loads); beyond, the O(tiles^2) traffic would show and the last workgroup of
// k_prepare scans the totals once instead (a serial tail of a few microseconds).
#define F_DIRECT_PREFIX_MAX 2048
#define F_CNT_STRIDE 16                       /* unsigned per counter: one 64-byte line */
#define F_CNT_WORDS (34 * F_CNT_STRIDE)       /* top + 32 shards (+ pad) */
__device__ __forceinline__ bool f_last_block(unsigned* cnt, int b, int nblocks, int* s_flag)
{
    if (threadIdx.x == 0) {
        smc_drain_stores();
        const int shards = nblocks >= 64 ? 32 : 1;
        const int s = b & (shards - 1);
        const int size_s = nblocks / shards + (s < nblocks % shards ? 1 : 0);
        bool last = atomicAdd(cnt + (1 + s) * F_CNT_STRIDE, 1u) == (unsigned)(size_s - 1);
        if (last) last = atomicAdd(cnt, 1u) == (unsigned)(shards - 1);
        if (last)
            for (int i = 0; i <= shards; ++i) cnt[i * F_CNT_STRIDE] = 0u;
        *s_flag = last;
    }
    __syncthreads();
    return *s_flag != 0;
}

// ---------------------------------------------------------------------------
// k_prepare
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(SMC_BLOCK)
k_prepare(const FArgs av)
{
    const FArgs& a = av;
    __shared__ u64 smu[SMC_SM];
    __shared__ int s_last;
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y;
    const double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)smc_uniform(info[0]);
    if (t >= a.T || smc_uniform(info[1]) == 0.0) return;          // step t does not resample
    const double m = smc_uniform(info[3]), rs = smc_uniform(info[4]);
    // Q62 weights of step t-1's particles (the parents of step t) + tile total
    const double* lw = f_lw(a, t - 1) + (i64)isl * a.N;
    const bool vec = (a.N & 3) == 0;
    const i64 j0 = (i64)b * F_TILE + (i64)threadIdx.x * F_IPT;
    double l4[4];
    f_load4<double>(lw, j0, a.N, vec, -INFINITY, l4);
    u64 q4[4], s = 0;
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ok[i] = j0 + i < a.N;
        q4[i] = ok[i] ? smc_q62_w(f_weight(l4[i], m, rs)) : 0ull;
        s += q4[i];
    }
    s = smc_block_sum_u64(s, smu);
    u64* Q = a.Q + (i64)isl * a.ntiles;
    if (threadIdx.x == 0) smc_st_agent(Q + b, s);
    if (!f_last_block(a.cnt + (isl * 2 + 1) * F_CNT_WORDS, b, a.ntiles, &s_last)) return;
    // last workgroup: exclusive prefixes of the tile totals (exact integers)
    u64* Qpre = a.Qpre + (i64)isl * a.ntiles;
    const int per = (a.ntiles + SMC_BLOCK - 1) / SMC_BLOCK;
    const int i0 = (int)threadIdx.x * per;
    u64 loc = 0;
    if (per <= 4) {               // up to 1024 tiles: all loads of a thread in flight at once
        u64 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (k < per && i0 + k < a.ntiles) ? smc_ld_agent(Q + i0 + k) : 0ull;
        loc = v[0] + v[1] + v[2] + v[3];
        u64 tot;
        u64 run = smc_block_exscan_u64(loc, smu, tot);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < per && i0 + k < a.ntiles) { Qpre[i0 + k] = run; run += v[k]; }
        return;
    }
    for (int i = i0; i < i0 + per && i < a.ntiles; ++i) loc += smc_ld_agent(Q + i);
    u64 tot;
    u64 run = smc_block_exscan_u64(loc, smu, tot);
    for (int i = i0; i < i0 + per && i < a.ntiles; ++i) {
        Qpre[i] = run;
        run += smc_ld_agent(Q + i);
    }
}

// ---------------------------------------------------------------------------
// multinomial, Philox mode: sorted uniforms by exponential spacings
// (resampling.py:512-537), batched over islands, skipped when not resampling
// ---------------------------------------------------------------------------
// The integer spacing of draw n in [0, N] of step t: rint(-log(u_n) 2^s), u_n the open-interval
// uniform of Philox word n & 1 of call n >> 1 (stream SPACINGS); log by the table-driven routine
// of the normals (smc_log_u52: <= 3 ulp), the rounding read off the mantissa (q < 2^51).
// The prefix sum of the draws INSIDE a tile of 1024 saturates at 2^32 - 1 (part of the contract -- oracle
// philox_spacings -- so that the one-pass kernel can store it as a 32-bit word; with the scale of at most 2^21 a tile's
// sum is 2^31 +- 2^26: the bound is 32 standard deviations away): Z_n = E[tile] + min(prefix in tile, 2^32 - 1).
#define SP_OFF_MAX 0xFFFFFFFFull
__device__ __forceinline__ u64 f_spacing_int(const SmcD2* ntab, const u64 x, const double scale)
{
    const double m = fma(-smc_log_u52(ntab, x), scale, 4503599627370496.0);      // 2^52 + rint(-log(u) 2^s)
    return (u64)__double_as_longlong(m) & 0x000FFFFFFFFFFFFFull;
}
// the 4 draws n0 .. n0 + 3 of a thread (n0 a multiple of 4: two Philox calls); 0 beyond draw N
__device__ __forceinline__ void f_spacing_q4(const FArgs& a, const SmcD2* ntab, const u32 t, const u32 gisl,
                                             const i64 n0, u64 (&q)[4])
{
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
        u64 x, y;
        smc_philox((u32)((n0 + i) >> 1), t, gisl, SMC_STREAM_SPACINGS, a.seed, x, y);
        q[i] = (n0 + i <= a.N) ? f_spacing_int(ntab, x, a.spacing_scale) : 0ull;
        q[i + 1] = (n0 + i + 1 <= a.N) ? f_spacing_int(ntab, y, a.spacing_scale) : 0ull;
    }
}

// Pass 1 of uniform_spacings: the tile sums of the integer spacings.  Nothing else is written: pass 2
// (k_f_spacing_write) makes the same draws again -- 54 instructions per draw with the table-driven log,
// against 16 bytes of traffic per draw for parking them (round 2 parked: its log was libm's).
__global__ void __launch_bounds__(SMC_BLOCK)
k_f_spacing_sums(const FArgs av)
{
    const FArgs& a = av;
    __shared__ u64 smu[SMC_SM];
    SMC_NTAB_LDS(s_ntab);
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y;
    smc_ntab_stage<SMC_BLOCK>(s_ntab, (int)threadIdx.x);
    __syncthreads();
    const double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)smc_uniform(info[0]);
    if (t >= a.T || t == 0 || smc_uniform(info[1]) == 0.0) return;
    const i64 n0 = (i64)b * F_TILE + (i64)threadIdx.x * F_IPT;
    u64 q[4];
    f_spacing_q4(a, s_ntab, (u32)t, (u32)(a.island_offset + isl), n0, q);
    const u64 s = smc_block_sum_u64(q[0] + q[1] + q[2] + q[3], smu);
    if (threadIdx.x == 0) a.E[(i64)isl * (a.ntiles1 + 1) + b] = s;
}

// E (n_islands, ntiles1 + 1): the tile totals -> their exclusive prefixes, the grand total in the
// extra slot; one workgroup per island between the two passes (a ticket in k_f_spacing_sums -- 4097
// atomics on 33 addresses at N = 2^22 -- cost 8 us, this launch costs 4)
__global__ void __launch_bounds__(SMC_BLOCK)
k_f_spacing_scan(const FArgs av)
{
    const FArgs& a = av;
    __shared__ u64 smu[SMC_SM];
    const int isl = (int)blockIdx.x;
    const double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)smc_uniform(info[0]);
    if (t >= a.T || t == 0 || smc_uniform(info[1]) == 0.0) return;
    u64* E = a.E + (i64)isl * (a.ntiles1 + 1);
    const int per = (a.ntiles1 + SMC_BLOCK - 1) / SMC_BLOCK;
    const int i0 = (int)threadIdx.x * per;
    const int i1 = (i0 + per < a.ntiles1) ? i0 + per : a.ntiles1;
    if (per <= 24) {                            // up to 6144 tiles: every load in flight at once
        u64 v[24], loc = 0;
#pragma unroll
        for (int k = 0; k < 24; ++k) { v[k] = (i0 + k < i1) ? smc_ldg(E + i0 + k) : 0ull; }
#pragma unroll
        for (int k = 0; k < 24; ++k) loc += v[k];
        u64 tot;
        u64 run = smc_block_exscan_u64(loc, smu, tot);
#pragma unroll
        for (int k = 0; k < 24; ++k)
            if (i0 + k < i1) { E[i0 + k] = run; run += v[k]; }
        if (threadIdx.x == 0) E[a.ntiles1] = tot;
        return;
    }
    u64 loc = 0;
    for (int i = i0; i < i1; i += 8) {          // 8 loads in flight at a time
        u64 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (i + k < i1) ? smc_ldg(E + i + k) : 0ull;
#pragma unroll
        for (int k = 0; k < 8; ++k) loc += v[k];
    }
    u64 tot;
    u64 run = smc_block_exscan_u64(loc, smu, tot);
    for (int i = i0; i < i1; i += 8) {
        u64 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (i + k < i1) ? smc_ldg(E + i + k) : 0ull;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i + k < i1) { E[i + k] = run; run += v[k]; }
    }
    if (threadIdx.x == 0) E[a.ntiles1] = tot;
}

__global__ void __launch_bounds__(SMC_BLOCK)
k_f_spacing_write(const FArgs av)
{
    const FArgs& a = av;
    __shared__ u64 smu[SMC_SM];
    SMC_NTAB_LDS(s_ntab);
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y;
    smc_ntab_stage<SMC_BLOCK>(s_ntab, (int)threadIdx.x);
    __syncthreads();
    const double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)smc_uniform(info[0]);
    if (t >= a.T || t == 0 || smc_uniform(info[1]) == 0.0) return;
    const u64* E = a.E + (i64)isl * (a.ntiles1 + 1);
    const u64 pre = smc_uniform_u64(smc_ldg(E + b)), all = smc_uniform_u64(smc_ldg(E + a.ntiles1));
    const i64 n0 = (i64)b * F_TILE + (i64)threadIdx.x * F_IPT;
    u64 q[F_IPT];
    f_spacing_q4(a, s_ntab, (u32)t, (u32)(a.island_offset + isl), n0, q);      // the same draws as pass 1
    const u64 tsum = q[0] + q[1] + q[2] + q[3];
    u64 tot;
    u64 run = smc_block_exscan_u64(tsum, smu, tot);
    const double dall = (double)all;
    double* su = a.su + (i64)isl * a.N;
#pragma unroll
    for (int i = 0; i < F_IPT; ++i) {
        run += q[i];
        if (n0 + i < a.N) su[n0 + i] = (double)(pre + (run < SP_OFF_MAX ? run : SP_OFF_MAX)) / dall;
    }
}

// uniform_spacings in ONE pass (two-level step, production mode): a workgroup makes the draws of TPW consecutive tiles
// of 1024, four tiles at a time (16 draws per thread in registers), and writes every draw's prefix sum INSIDE ITS TILE
// as a 32-bit word right away -- o_n = q_{1024 k} + .. + q_n (saturating at 2^32 - 1: the scale of the draws,
// spacing_scale = 2^21 at most, puts a tile's sum 32 standard deviations below that) -- so that
//      Z_n = E[n >> 10] + o_n ,   su_n = fl(Z_n) / fl(Z_N)        (resampling.py:536-537)
// with E[k] the sum of the tiles in front of tile k.  Only the E[k] need the workgroups in front: the workgroup
// publishes its total, obtains the total of everything before it by a decoupled look-back over the workgroups' status
// words and writes its TPW tile prefixes -- 4 bytes per draw go to memory, once, and they leave while the next tiles'
// logarithms are being computed (round 5 stored the 8-byte Z_n themselves, all of them behind the look-back:
// generate 9.8 us, wait 3, write 5.3-7.5 at N = 2^22, profiles/r15_trace_spacing_before.txt).  Status word of
// workgroup w: (its total << 2) | 1, 0 = not there yet (k_ancestors2 zeroes the words again).  The grid is sized by
// the host so that EVERY workgroup is resident at once (smc_filter_create asks the runtime how many fit): a spinning
// workgroup can then never keep the one it waits for off the chip, whatever the dispatch order; the spin is bounded
// all the same.
#define SP_FLAG_AGG 1ull
// a.sp_epoch != 0: the launch carries one workgroup more per island -- workgroup 0 is k_reduce2 (the island's
// reduction, the decision of step t, the tiles' shares), the others draw while it works and look at its
// decision before they publish anything (their offsets are in memory by then: on a step that does not resample
// nobody reads them): the 9 us of a one-workgroup launch on the critical path of every multinomial step
// (4096 tiles) disappear behind the 4 M logarithms.
__device__ __forceinline__ int f2_reduce2_island(const FArgs& a, const int isl, double* smd, double* sme,
                                                 u64* dec_word = nullptr, const u64 dec_epoch = 0ull);
#ifndef SMC_SP_MINW
#define SMC_SP_MINW 5        /* waves per SIMD the register allocation leaves room for (A/B builds) */
#endif
template <int TPW>
__global__ void __launch_bounds__(SMC_BLOCK, SMC_SP_MINW)
k_f_spacing_onepass(const FArgs av)
{
    const FArgs& a = av;
    constexpr int TG = TPW > 4 ? 4 : TPW;          // tiles in flight (registers): 16 draws per thread
    constexpr int NG = TPW / TG;
    __shared__ u64 s_w[2][TG][SMC_NWAVE];          // (two areas: ONE barrier per group of tiles)
    __shared__ u64 s_pre;
    SMC_NTAB_LDS(s_ntab);
    const bool merged = a.sp_epoch != 0ull;
    const int isl = (int)blockIdx.y;
    const int tid = (int)threadIdx.x, lane = smc_lane(), wave = smc_wave();
    F_STAMP_S(0);
    if (merged && blockIdx.x == 0) {
        __shared__ double smd[SMC_SM];
        __shared__ double sme[SMC_SM];
#ifndef SMC_EMULATE
        // the launch ends with this workgroup (15 us beside 1024 workgroups drawing, 8.5 alone): its waves go first
        // wherever they share a SIMD
        __builtin_amdgcn_s_setprio(3);
#endif
        const int dec = f2_reduce2_island(a, isl, smd, sme, a.sdec + isl, a.sp_epoch);
        if (dec < 0 && tid == 0) smc_st_agent(a.sdec + isl, (a.sp_epoch << 2) | 1ull);     // (no step to run)
        F_STAMP_S(5);
        return;
    }
    const int w = (int)blockIdx.x - (merged ? 1 : 0);
    smc_ntab_stage<SMC_BLOCK>(s_ntab, tid);
    __syncthreads();
    // (merged: the record is being written by workgroup 0 -- the count of steps done is k_propagate's)
    const double* info = merged ? a.info2 + (i64)isl * INFO_STRIDE : a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)smc_uniform(info[0]);
    if (t >= a.T || t == 0 || (!merged && smc_uniform(info[1]) == 0.0)) return;
    F_STAMP_S(1);
    const u32 gisl = (u32)(a.island_offset + isl);
    u64* st = a.sst + (i64)isl * a.sp_nwg;
    u64* E = a.E + (i64)isl * (a.ntiles1 + 1);
    u32* O = reinterpret_cast<u32*>(a.su) + (i64)isl * a.N * 2;         // (the island's slot of N doubles: N words used)
    const bool vec = (a.N & 3) == 0;
    // ---- the draws of my tiles (tile r: draws 1024 (w TPW + r) + 4 tid ..), tile by tile: wave scans, the
    // tile's total, every draw's prefix inside its tile -> memory
    // (tiles of the N uniforms: k < ntiles.  The (N+1)-th draw, which only the total needs, sits in the last
    //  of them unless N is a multiple of 1024 -- then the last workgroup makes it by itself, below: the
    //  grid stays ceil(ntiles / TPW) workgroups)
    u64 tile_tot[TPW];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        u64 q[TG][4], inc[TG];
#pragma unroll
        for (int r = 0; r < TG; ++r) {
            const i64 k = (i64)w * TPW + g * TG + r;
            if (k < a.ntiles) f_spacing_q4(a, s_ntab, (u32)t, gisl, k * F_TILE + (i64)tid * F_IPT, q[r]);
            else { q[r][0] = q[r][1] = q[r][2] = q[r][3] = 0ull; }
            inc[r] = smc_wave_scan_add_u64(q[r][0] + q[r][1] + q[r][2] + q[r][3]);
        }
        if (lane == 63) {
#pragma unroll
            for (int r = 0; r < TG; ++r) s_w[g & 1][r][wave] = inc[r];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < TG; ++r) {
            const i64 k = (i64)w * TPW + g * TG + r;
            u64 run = 0ull, tot = 0ull;
#pragma unroll
            for (int v = 0; v < SMC_NWAVE; ++v) {
                if (v < wave) run += s_w[g & 1][r][v];
                tot += s_w[g & 1][r][v];
            }
            tile_tot[g * TG + r] = tot;
            if (k >= a.ntiles) continue;
            run += inc[r] - (q[r][0] + q[r][1] + q[r][2] + q[r][3]);
            const i64 n0 = k * F_TILE + (i64)tid * F_IPT;
            u32 o4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                run += q[r][i];
                o4[i] = (u32)(run < SP_OFF_MAX ? run : SP_OFF_MAX);
            }
            if (n0 + 3 < a.N && vec) {
                smc_st4g(O + n0, o4);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (n0 + i < a.N) smc_stg(O + n0 + i, o4[i]);
            }
        }
    }
    F_STAMP_S(2);
    u64 total = 0ull;
#pragma unroll
    for (int r = 0; r < TPW; ++r) total += tile_tot[r];
    u64 q_last = 0ull;                             // draw N when it has a tile of its own
    if (w == a.sp_nwg - 1 && a.ntiles1 > a.ntiles && tid == 0) {
        u64 ql[4];
        f_spacing_q4(a, s_ntab, (u32)t, gisl, a.N, ql);       // (N a multiple of 1024 here: draw N is slot 0)
        q_last = ql[0];
    }
    // ---- publish my total; the prefix = the totals of ALL workgroups before me (<= 1024 of them: up to 4
    // per thread, every load in flight at once -- the workgroups generate side by side and publish within
    // a microsecond of each other, so a chained look-back would only add its w / 64 dependent rounds).
    // Status word: (total << 22) | tag.  merged: the tag is the launch's own (from its epoch), so the words are
    // published -- and the look-back done -- BEFORE the decision is known (a launch that turns out not to resample
    // leaves words no later launch can mistake for its own); otherwise the decision came with the record, only
    // resampling launches get here, tag 1, and k_ancestors2 zeroes the words again.
    const u64 tag = merged ? (a.sp_epoch % 0x3FFFFEull) + 2ull : 1ull;
    if (tid == 0) smc_st_agent(st + w, (total << 22) | tag);
    u64 part = 0ull;
    for (int idx = tid; idx < w; idx += SMC_BLOCK) {
        u64 word = smc_ld_agent(st + idx);
        for (int spin = 0; (word & 0x3FFFFFull) != tag && spin < (1 << 22); ++spin) {
            smc_spin_pause();
            word = smc_ld_agent(st + idx);
        }
        part += word >> 22;
    }
    __syncthreads();                                                       // (s_w has been read)
    part = smc_wave_sum_u64(part);
    if (lane == 0) s_w[0][0][wave] = part;
    __syncthreads();
    u64 excl = 0ull;
#pragma unroll
    for (int v = 0; v < SMC_NWAVE; ++v) excl += s_w[0][0][v];
    F_STAMP_S(3);
    if (merged) {                                  // the decision of step t (workgroup 0's: published as soon as it has it)
        if (tid == 0) {
            u64 word = smc_ld_agent(a.sdec + isl);
            for (int spin = 0; (word >> 2) != a.sp_epoch && spin < (1 << 24); ++spin) {
                smc_spin_pause();
                word = smc_ld_agent(a.sdec + isl);
            }
            s_pre = word;
        }
        __syncthreads();
        const u64 word = s_pre;
        if ((word >> 2) != a.sp_epoch || (word & 3ull) != 2ull) return;
    }
    F_STAMP_S(4);
    // ---- the tiles' exclusive prefixes (the window search of k_ancestors2 probes them; Z_n = E[n >> 10] + o_n)
    if (tid == 0) {
        u64 run = excl;
#pragma unroll
        for (int r = 0; r < TPW; ++r) {
            const i64 k = (i64)w * TPW + r;
            if (k < a.ntiles) E[k] = run;
            run += tile_tot[r];
        }
        if (w == a.sp_nwg - 1) {
            if (a.ntiles1 > a.ntiles) E[a.ntiles] = excl + total;         // the prefix of draw N's own tile
            E[a.ntiles1] = excl + total + q_last;                         // Z_N
        }
    }
    F_STAMP_S(5);
}

// ---------------------------------------------------------------------------
// The offspring of one tile of parents (shared by k_ancestors and the single-workgroup
// filter of smc_filter_small.h).  q4: the Q62 weights of this thread's 4 parents jt..jt+3,
// cex: their exclusive CDF (tile prefix included), pre/total: the tile's prefix and total.
// Calls sink(n0, ok[4], a4[4]) once per pass of 1024 offspring with the parents a4 of the
// offspring n0..n0+3 this thread owns in the pass (ok: inside the tile's range).
// ---------------------------------------------------------------------------
// The scatter passes: given the first offspring ns[i] of this thread's 4 parents (ns[4]: of the
// next thread's first), every parent writes its index at its first offspring's slot of the
// pass (LDS) and a running maximum over the slots gives each offspring its parent.
// (out of line: the tiles that call it are rare, the ones that do not keep their registers)
__device__ __attribute__((noinline)) int f_register_heavy(unsigned* hcnt, i64* hlist, const i64 jt,
                                                           const i64 n0, const i64 n1, const i64 n2,
                                                           const i64 n3, const i64 n4, i64* sH, unsigned* sHn)
{
    const i64 ns[F_IPT + 1] = {n0, n1, n2, n3, n4};
    if (threadIdx.x == 0) *sHn = 0u;
    __syncthreads();
    for (int i = 0; i < F_IPT; ++i) {
        const i64 bs = ((ns[i] + F_TILE - 1) / F_TILE) * F_TILE, be = (ns[i + 1] / F_TILE) * F_TILE;
        if (ns[i + 1] - ns[i] >= 2 * (i64)F_TILE && be > bs) {
            const unsigned g = atomicAdd(hcnt, 1u);
            if (g < F_HMAX) {
                const unsigned k = atomicAdd(sHn, 1u);
                i64* e = hlist + (i64)g * 3;
                if (k < F_HLOC) {
                    e[0] = bs; e[1] = be; e[2] = jt + i;
                    sH[2 * k] = bs; sH[2 * k + 1] = be;
                } else {                           // no room here: the entry stays harmless (empty)
                    e[0] = 0; e[1] = 0; e[2] = 0;
                }
            }
        }
    }
    __syncthreads();
    return (int)(*sHn < F_HLOC ? *sHn : F_HLOC);
}

// A parent with >= 2048 offspring owns whole 1024-blocks of them: it is registered (hlist) and
// the passes that lie inside those blocks are skipped -- k_propagate's workgroup of such a block
// takes the parent from the list and writes the block's ancestors itself, so a collapsed weight
// vector costs the tile's workgroup a few passes instead of N / 1024.
template <int BS = SMC_BLOCK, class Sink>
__device__ __forceinline__ void f_scatter_passes(const FArgs& a, const int isl, const i64 t, const i64 jt,
                                                 const i64 j0, const i64 n_lo, const i64 n_hi,
                                                 const i64 (&ns)[F_IPT + 1], u32* sP, u32* smx,
                                                 Sink&& sink)
{
    constexpr int PASS = BS * 4, NW = BS / 64;
    const int tid = (int)threadIdx.x;
    __shared__ i64 sH[2 * F_HLOC];
    __shared__ unsigned sHn;
    const bool heavy = a.hcnt && (n_hi - n_lo >= 2 * (i64)F_TILE);        // (same in every thread)
    int nH = 0;
    if (heavy) nH = f_register_heavy(a.hcnt + (i64)isl * 2 + (t & 1),
                                     a.hlist + ((i64)isl * 2 + (t & 1)) * F_HMAX * 3, jt,
                                     ns[0], ns[1], ns[2], ns[3], ns[4], sH, &sHn);
    for (i64 pb = n_lo & ~(i64)3; pb < n_hi; pb += PASS) {
        if (nH) {                                  // the whole pass inside a registered parent's blocks?
            const i64 w_lo = pb > n_lo ? pb : n_lo, w_hi = pb + PASS < n_hi ? pb + PASS : n_hi;
            i64 jump = 0;                          // passes to leave out, this one included
            for (int k = 0; k < nH; ++k)
                if (sH[2 * k] <= w_lo && w_hi <= sH[2 * k + 1]) {
                    const i64 whole = (sH[2 * k + 1] - pb) / PASS;      // passes that end inside the blocks
                    jump = whole > 1 ? whole : 1;
                }
            if (jump) { pb += (jump - 1) * PASS; continue; }
        }
        const i64 n0 = pb + (i64)tid * 4;
        bool ok[4];
        i64 a4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ok[i] = (n0 + i >= n_lo) && (n0 + i < n_hi);
        __syncthreads();                           // previous pass has read sP
        *reinterpret_cast<uint4*>(&sP[tid * 4]) = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < F_IPT; ++i) {
            const i64 lo = ns[i] > pb ? ns[i] : pb;
            const i64 hi = ns[i + 1] < pb + PASS ? ns[i + 1] : pb + PASS;
            if (lo < hi) sP[lo - pb] = (u32)(tid * F_IPT + i);
        }
        __syncthreads();
        const uint4 v = *reinterpret_cast<const uint4*>(&sP[tid * 4]);
        const u32 m0 = v.x, m1 = m0 > v.y ? m0 : v.y, m2 = m1 > v.z ? m1 : v.z,
                  m3 = m2 > v.w ? m2 : v.w;
        const u32 inc = smc_wave_scan_max_u32(m3);
        u32 ex = smc_mov_dpp<SMC_DPP_WAVE_SHR1>(inc);
        if (smc_lane() == 0) ex = 0u;
        if (smc_lane() == 63) smx[smc_wave()] = inc;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NW - 1; ++w)
            if (w < smc_wave()) ex = ex > smx[w] ? ex : smx[w];
        a4[0] = j0 + (i64)(m0 > ex ? m0 : ex);
        a4[1] = j0 + (i64)(m1 > ex ? m1 : ex);
        a4[2] = j0 + (i64)(m2 > ex ? m2 : ex);
        a4[3] = j0 + (i64)(m3 > ex ? m3 : ex);
        sink(n0, ok, a4);
    }
}

template <int BS = SMC_BLOCK, class Sink>
__device__ __forceinline__ void f_tile_offspring(const FArgs& a, const int isl, const i64 t,
                                                 const int b, const i64 jt, const i64 j0,
                                                 const u64 (&q4)[4], const u64 cex, const u64 pre,
                                                 const u64 total, u64* sC, u32* sP, i64* sn, u32* smx,
                                                 Sink&& sink)
{
    constexpr int TILE = BS * F_IPT, PASS = BS * 4;                   // of THIS workgroup size
    const int tid = (int)threadIdx.x;
    const i64 N = a.N;
    const u32 gisl = (u32)(a.island_offset + isl);
    SmcSu su;
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
            smc_philox_uniform(0u, su.t, su.island, SMC_STREAM_RESAMPLE, su.seed, x0, x1);
            su.u_sys = smc_u01_halfopen(x0);
        }
    }
    const bool scatter = (a.scheme == SMC_SYSTEMATIC_ || a.scheme == SMC_STRATIFIED_) && a.log2N >= 0;
    i64 n_lo, n_hi;
    i64 ns[F_IPT + 1];
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
                     : (a.scheme == SMC_SYSTEMATIC_ ? smc_sys_count_pow2_fast(c, su.u_sys, Us, a.log2N, N)
                                                    : smc_strat_count_pow2(c, su, a.log2N, N)));
            if (i < F_IPT) c += q4[i];
        }
        if (tid == 0) sn[0] = ns[0];
        if (tid == BS - 1) sn[1] = ns[F_IPT];
        __syncthreads();
        n_lo = sn[0];
        n_hi = sn[1];
        f_scatter_passes<BS>(a, isl, t, jt, j0, n_lo, n_hi, ns, sP, smx, sink);
        return;
    } else {
        u64 c = cex;
#pragma unroll
        for (int i = 0; i < F_IPT; ++i) {
            c += q4[i];
            sC[tid * F_IPT + i] = c;
        }
        __syncthreads();
        smc_tile_outputs<BS>(su, b, a.ntiles, pre, total, sn, n_lo, n_hi);
    }
    const int nvalid = (int)((N - j0 < TILE) ? (N - j0) : TILE);

    // ---- offspring, 4 consecutive ones per thread per pass
    for (i64 pb = n_lo & ~(i64)3; pb < n_hi; pb += PASS) {
        const i64 n0 = pb + (i64)tid * 4;
        bool ok[4];
        i64 a4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ok[i] = (n0 + i >= n_lo) && (n0 + i < n_hi);
        {
            double s4[4];
            smc_su_pair(su, n0 >> 1, s4[0], s4[1]);
            smc_su_pair(su, (n0 >> 1) + 1, s4[2], s4[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int jl = ok[i] ? smc_lower_bound_u64(sC, TILE, smc_q62_t(s4[i])) : 0;
                a4[i] = j0 + (jl < nvalid ? jl : nvalid - 1);
            }
        }
        sink(n0, ok, a4);
    }
}

// ---------------------------------------------------------------------------
// k_ancestors(t): integer-only.  One workgroup per tile of 1024 parents: exact
// CDF of the tile from q and its exclusive prefix, the contiguous range of
// offspring it owns, and the parent index of each of them -> A.
// ---------------------------------------------------------------------------
// SPEC: the buffer slots follow from a.par (kernarg), so the tile's log-weights are
// requested right behind the step record instead of one memory round trip after it.
template <bool FUSED, bool SPEC>
__global__ void __launch_bounds__(SMC_BLOCK)
k_ancestors(const FArgs av)
{
    const FArgs& a = av;
    __shared__ u64 sC[F_TILE];         // inclusive CDF of the tile (search path)
    __shared__ __attribute__((aligned(16))) u32 sP[F_PASS];   // parent of each offspring of a
                                                               // pass (scatter path)
    __shared__ u64 smu[SMC_SM];
    __shared__ i64 sn[2];
    __shared__ u32 smx[SMC_NWAVE];
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y;
    const int tid = (int)threadIdx.x;
    const double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 N = a.N;
    const bool vec = (N & 3) == 0;
    const i64 j0 = (i64)b * F_TILE;
    const i64 jt = j0 + (i64)tid * F_IPT;
    F_STAMP_A(0);
    const double r0 = smc_ldg(info), r1 = smc_ldg(info + 1), r3 = smc_ldg(info + 3),
                 r4 = smc_ldg(info + 4);                      // requested first ...
    double l4[4];
    if (FUSED && SPEC)                                         // ... the data right behind
        f_load4<double>(a.lw + (i64)(a.par ^ 1) * a.lslot + (i64)isl * N, jt, N, vec, -INFINITY, l4);
    const i64 t = (i64)smc_uniform(r0);
    if (t >= a.T || t == 0 || smc_uniform(r1) == 0.0) return;          // step t does not resample
    F_STAMP_A(1);
    u32* A = f_A(a, t) + (i64)isl * N;

    // ---- the tile's parents: q and their exact CDF
    u64 q4[4];
    u64 total, pre;
    u64 cex;
    if (FUSED) {
        // Q62 weights straight from the log-weights of step t-1 (f_weight); the
        // tile total is published as total+1 (0 = "not there yet") and the totals
        // of the tiles before this one are picked up the same way: all tiles of
        // an island publish a few microseconds into the launch, lower-numbered
        // workgroups are dispatched first, so the wait is short and cannot cycle.
        // k_propagate(t) zeroes Q again.
        const double m = smc_uniform(r3), rs = smc_uniform(r4);
        if (!SPEC) f_load4<double>(f_lw(a, t - 1) + (i64)isl * N, jt, N, vec, -INFINITY, l4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            q4[i] = (jt + i < N) ? smc_q62_w(f_weight(l4[i], m, rs)) : 0ull;
        const u64 tsum = q4[0] + q4[1] + q4[2] + q4[3];
        u64* Qt = a.Q + (i64)isl * a.ntiles;
        F_STAMP_A(2);
        const u64 mine = smc_block_sum_u64(tsum, smu);
        if (tid == 0) smc_st_agent(Qt + b, mine + 1ull);
        F_STAMP_A(3);
        constexpr int NPRE = F_DIRECT_PREFIX_MAX / SMC_BLOCK;
        u64 v[NPRE];
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int i = tid + k * SMC_BLOCK;
            v[k] = (i < b) ? smc_ld_agent(Qt + i) : 1ull;
        }
        u64 part = 0;
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            while (v[k] == 0ull) {
                smc_spin_pause();
                v[k] = smc_ld_agent(Qt + tid + k * SMC_BLOCK);
            }
            part += v[k] - 1ull;
        }
        F_STAMP_A(4);
        cex = smc_block_exscan_plus_sum_u64(tsum, part, smu, total, pre);
    } else {
        // k_prepare kept only the tile totals: the Q62 weights are formed again from the
        // log-weights (same expression, same bits) -- 8 B/particle of q traffic less each way
        const double m = smc_uniform(r3), rs = smc_uniform(r4);
        f_load4<double>(f_lw(a, t - 1) + (i64)isl * N, jt, N, vec, -INFINITY, l4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            q4[i] = (jt + i < N) ? smc_q62_w(f_weight(l4[i], m, rs)) : 0ull;
        const u64 tsum = q4[0] + q4[1] + q4[2] + q4[3];
        pre = a.Qpre[(i64)isl * a.ntiles + b];
        cex = smc_block_exscan_u64(tsum, smu, total);
    }
    cex += pre;                                                      // exclusive CDF, 1st parent
    F_STAMP_A(5);
    f_tile_offspring(a, isl, t, b, jt, j0, q4, cex, pre, total, sC, sP, sn, smx,
                     [&](i64 n0, const bool (&ok)[4], const i64 (&a4)[4]) {                // core.py:329
                         const u32 a32[4] = {(u32)a4[0], (u32)a4[1], (u32)a4[2], (u32)a4[3]};
                         if (vec && ok[0] && ok[3]) {
                             if (a.nt & 8) smc_st4g_nt(A + n0, a32);
                             else smc_st4g(A + n0, a32);
                         } else {
#pragma unroll
                             for (int i = 0; i < 4; ++i)
                                 if (ok[i]) smc_stg(A + n0 + i, a32[i]);
                         }
                     });
    F_STAMP_A(6);
}

// Step t is complete and its log-weights reduce to g = (max, sum e, sum e^2): summary row of t
// (resampling.py:224-226, core.py:355-359), the resample decision of t+1 (core.py:181-183) and
// the step record the launches of t+1 read.  One thread.
__device__ __forceinline__ void f_finalise_step(const FArgs& a, const int isl, const i64 t,
                                                const bool first, const bool resample,
                                                const SmcLse g, double* info)
{
    const bool bad = !(g.m > -INFINITY) || !(g.m < INFINITY);
    const double ess = bad ? NAN : (g.s * g.s) / g.ss;                  // resampling.py:226
    const double log_mean = bad ? NAN : g.m + log(g.s / (double)a.N);   // resampling.py:224
    const double rs = bad ? NAN : 1.0 / g.s;
    double* row = a.summ + ((i64)isl * (a.T + 1) + t) * SUMM_STRIDE;
    double loglt, logLt;                                                // core.py:355-359
    if (first || resample) loglt = log_mean;
    else loglt = log_mean - row[1 - SUMM_STRIDE];
    logLt = (first ? 0.0 : row[3 - SUMM_STRIDE]) + loglt;
    row[0] = ess;
    row[1] = log_mean;
    row[2] = loglt;
    row[3] = logLt;
    row[4] = resample ? 1.0 : 0.0;
    row[5] = g.m;
    row[6] = rs;
    const bool flag = (t + 1 < a.T) && (ess < a.ess_thresh);            // core.py:181-183
    info[1] = flag ? 1.0 : 0.0;
    info[2] = (t + 1 < a.T) ? a.y[(t + 1) * a.dy] : 0.0;
    info[5] = (a.aux && t + 1 < a.T) ? a.aux[t + 1] : 0.0;
    info[3] = g.m;
    info[4] = rs;
    info[0] = (double)(t + 1);
}

// ---------------------------------------------------------------------------
// End of a propagate kernel: publish the workgroup's log-sum-exp partial `r` and
// let the last workgroup of the island finalise step t and decide step t+1.
// Called by every thread.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void f_step_tail(const FArgs& a, const int isl, const int b,
                                            const i64 t, const bool first, const bool resample,
                                            const SmcLse r, double* smd, int& s_last,
                                            double* info)
{
    const int tid = (int)threadIdx.x;
    {   // re-arm the tile totals k_ancestors<true> publishes (0 = "not there yet")
        const i64 g = (i64)b * SMC_BLOCK + tid;
        if (g < a.ntiles) a.Q[(i64)isl * a.ntiles + g] = 0ull;
    }
    F_STAMP(4);
    const i64 o = (i64)isl * a.nparts;
    // ---- publish the partial; two-level "last one reduces" (no spinning):
    // the last workgroup of each of the 32 shards reduces its shard's partials,
    // the last of those reduces the 32 shard results and finalises the step
    const int shards = a.nparts >= 64 ? 32 : 1;
    const int sh = b & (shards - 1);
    const int size_s = a.nparts / shards + (sh < a.nparts % shards ? 1 : 0);
    unsigned* cnt = a.cnt + (isl * 2) * F_CNT_WORDS;
    double* spart = a.spart + (i64)isl * 96;
    if (tid == 0) {
        smc_st_agent_f64(a.pm + o + b, r.m);
        smc_st_agent_f64(a.ps + o + b, r.s);
        smc_st_agent_f64(a.pss + o + b, r.ss);
        smc_drain_stores();
        s_last = atomicAdd(cnt + (1 + sh) * F_CNT_STRIDE, 1u) == (unsigned)(size_s - 1);
    }
    __syncthreads();
    F_STAMP(5);
    if (!s_last) return;
    // Up to 2048 partials the last workgroup of the island reduces them all itself (<= 8
    // loads per thread and array, one round trip): the shard level then only counts.
    const bool direct = a.nparts <= 2048;
    SmcLse gs = smc_lse_empty();
    if (!direct) {
        gs = smc_lse_reduce_partials<true>(a.pm + o, a.ps + o, a.pss + o, size_s, smd, sh, shards);
        __syncthreads();
    }
    if (tid == 0) {
        cnt[(1 + sh) * F_CNT_STRIDE] = 0u;                 // re-arm for the next launch
        if (!direct) {
            smc_st_agent_f64(spart + sh, gs.m);
            smc_st_agent_f64(spart + 32 + sh, gs.s);
            smc_st_agent_f64(spart + 64 + sh, gs.ss);
            smc_drain_stores();
        }
        s_last = atomicAdd(cnt, 1u) == (unsigned)(shards - 1);
        if (s_last) cnt[0] = 0u;
    }
    __syncthreads();
    F_STAMP(6);
    if (!s_last) return;

    // ---- last workgroup of this island: finalise step t, decide step t+1
    const SmcLse g = direct
        ? smc_lse_reduce_partials<true>(a.pm + o, a.ps + o, a.pss + o, a.nparts, smd)
        : smc_lse_reduce_partials<true>(spart, spart + 32, spart + 64, shards, smd);
    if (tid == 0) f_finalise_step(a, isl, t, first, resample, g, info);
    F_STAMP(7);
}

// ---------------------------------------------------------------------------
// k_propagate(t): element-wise over the new particles, OPT consecutive ones per
// thread.  x = loc(X_{t-1}[A]) + scale z, weight increment, log-weights, online
// log-sum-exp partial; the last workgroup of the island finalises the step.
// ---------------------------------------------------------------------------
// ---- two-level CDF path (contract: see "Two-level CDF" below): what k_propagate leaves per tile
#define F2_QBITS 49                     /* local CDF: q_i = rint(e_i 2^49), t_b < 2^60 */
#define F2_SBITS 52                     /* shares: Q_b = rint(W_b 2^52) */
struct F2Red {
    double K, s, ss, ess, log_mean, rs;
    bool bad;
};
struct F2Tile {
    double K, S, SS;
    u64 tb;
};
// The tile's partial and integer CDF from the 4 log-weights each thread holds (ownership: f_own);
// cx: the exclusive CDF positions of this thread's particles.  Two barriers; all threads get the
// result.  Association order of S, SS: the thread's 4 values in slot order, a balanced tree over
// the 64 lanes, the 4 waves left to right (oracle.c orc_tile_partials).
__device__ __forceinline__ F2Tile f2_tile_weights(const double (&lw)[4], u64 (&cx)[4])
{
    __shared__ double s_k[SMC_NWAVE];
    __shared__ double s_s[2 * SMC_NWAVE];
    __shared__ u64 s_c[SMC_NWAVE];
    const int lane = smc_lane(), wave = smc_wave();
    double p[4], k[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p[i] = smc_expk(lw[i], k[i]);
        p[i] = (lw[i] > -INFINITY) ? p[i] : 0.0;   // (k is -inf by itself there: rint(-inf log2 e); NaNs never arrive,
                                                   //  the callers turn them into -inf)
    }
    double km = smc_max2(smc_max2(k[0], k[1]), smc_max2(k[2], k[3]));
    km = smc_wave_max(km);
    if (lane == 0) s_k[wave] = km;
    __syncthreads();
    F2Tile r;
    r.K = s_k[0];
#pragma unroll
    for (int w = 1; w < SMC_NWAVE; ++w) r.K = smc_max2(r.K, s_k[w]);
    double s1 = 0.0, s2 = 0.0;
    u64 q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double e = smc_scale_pk(p[i], k[i], r.K);
        s1 += e;
        s2 = fma(e, e, s2);
        // q = rint(e 2^49) < 2^50 read off the mantissa of e 2^49 + 2^52 (one rounding, to the nearest
        // even integer: the same value as (u64)rint(e * 2^49), in 2 instructions instead of 10)
        q[i] = (u64)__double_as_longlong(fma(e, 562949953421312.0, 4503599627370496.0)) & 0x000FFFFFFFFFFFFFull;
    }
    smc_wave_sum2(s1, s2);
    // the thread's pairs are 128 particles apart (f_own): the wave's first pairs come first
    const u64 sa = q[0] + q[1], sb = q[2] + q[3];
    u64 incA = sa, incB = sb;                      // (q < 2^50: the pairs' sums fit 51 bits)
    smc_wave_scan_add_u51x2(incA, incB);
    const u64 totA = smc_readlane64(incA, 63), totB = smc_readlane64(incB, 63);
    if (lane == 0) { s_s[wave] = s1; s_s[SMC_NWAVE + wave] = s2; s_c[wave] = totA + totB; }
    __syncthreads();
    r.S = s_s[0];
    r.SS = s_s[SMC_NWAVE];
    u64 base = 0ull;
    r.tb = 0ull;
#pragma unroll
    for (int w = 0; w < SMC_NWAVE; ++w) {
        if (w > 0) { r.S = r.S + s_s[w]; r.SS = r.SS + s_s[SMC_NWAVE + w]; }
        if (w < wave) base += s_c[w];
        r.tb += s_c[w];
    }
    cx[0] = base + incA - sa;
    cx[1] = cx[0] + q[0];
    cx[2] = base + totA + incB - sb;
    cx[3] = cx[2] + q[2];
    return r;
}
// The same partial WITHOUT the integer CDF (FArgs::strict_e): the tile-scale weights e_i themselves come back (their bit
// patterns: they take the integer CDF's place in `cq`).  S, SS: the operations of f2_tile_weights in its order, hence its bits.
__device__ __forceinline__ F2Tile f2_tile_weights_e(const double (&lw)[4], u64 (&cx)[4])
{
    __shared__ double s_k[SMC_NWAVE];
    __shared__ double s_s[2 * SMC_NWAVE];
    const int lane = smc_lane(), wave = smc_wave();
    double p[4], k[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p[i] = smc_expk(lw[i], k[i]);
        p[i] = (lw[i] > -INFINITY) ? p[i] : 0.0;
    }
    double km = smc_max2(smc_max2(k[0], k[1]), smc_max2(k[2], k[3]));
    km = smc_wave_max(km);
    if (lane == 0) s_k[wave] = km;
    __syncthreads();
    F2Tile r;
    r.K = s_k[0];
#pragma unroll
    for (int w = 1; w < SMC_NWAVE; ++w) r.K = smc_max2(r.K, s_k[w]);
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double e = smc_scale_pk(p[i], k[i], r.K);
        s1 += e;
        s2 = fma(e, e, s2);
        cx[i] = (u64)__double_as_longlong(e);
    }
    smc_wave_sum2(s1, s2);
    if (lane == 0) { s_s[wave] = s1; s_s[SMC_NWAVE + wave] = s2; }
    __syncthreads();
    r.S = s_s[0];
    r.SS = s_s[SMC_NWAVE];
#pragma unroll
    for (int w = 1; w < SMC_NWAVE; ++w) { r.S = r.S + s_s[w]; r.SS = r.SS + s_s[SMC_NWAVE + w]; }
    r.tb = 0ull;
    return r;
}
// The workgroup's log-sum-exp partial (max, sum e, sum e^2) of the OPT log-weights each thread
// holds (-inf beyond N): max first, then ONE exp per particle against the workgroup's max (no
// per-thread rescaling, no branches).  Fixed association order: the thread's OPT values left to
// right, a balanced tree over the 64 lanes, the 4 waves left to right (oracle.c orc_tile_partials).
template <int OPT>
__device__ __forceinline__ SmcLse f_tile_lse(const double (&lw)[OPT], double* smd)
{
    SmcLse r;
    double tm = lw[0];
#pragma unroll
    for (int k = 1; k < OPT; ++k) tm = smc_max2(tm, lw[k]);
    r.m = smc_block_max(tm, smd);
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < OPT; ++k) {
        const double e = (lw[k] > -INFINITY) ? smc_exp_nonpos(lw[k] - r.m) : 0.0;
        s1 += e;
        s2 = fma(e, e, s2);
    }
    smc_block_sum2(s1, s2, smd);
    r.s = s1;
    r.ss = s2;
    return r;
}

// Which 4 new particles a thread of k_propagate owns: the pairs (na, na+1) and (nb, nb+1) with
// na = first of the wave's span of 256 + 2 lane, nb = na + 128 -- so that each 16-byte access of
// the wave (64 lanes x 16 B) covers 1 KB of CONTIGUOUS memory.  With 4 consecutive particles per
// thread a store instruction wrote every other 16 bytes: harmless for plain stores (the halves
// merge in the L2), but streaming (`nt`) stores left the L2 as half lines and WRITE_SIZE was
// 1.7x the bytes stored (profiles/r01n, r02b).  Slot k of a thread: k = 0, 1 the first pair,
// 2, 3 the second.
struct FOwn {
    i64 na, nb;
    bool full;             // the wave's whole span lies inside N (wave-uniform)
};
// PAIRS = false: 4 consecutive particles per thread (the flat-CDF path keeps it: its one-launch
// twin k_filter_small holds 4 consecutive particles per thread and promises the same bits)
template <bool PAIRS>
__device__ __forceinline__ FOwn f_own(const int b, const int tid, const i64 N)
{
    FOwn o;
    if (PAIRS) {
        const i64 wb = ((i64)b * SMC_BLOCK + (tid & ~63)) * 4;
        o.na = wb + 2 * (tid & 63);
        o.nb = o.na + 128;
        o.full = (N & 1) == 0 && wb + 256 <= N;
    } else {
        o.na = ((i64)b * SMC_BLOCK + tid) * 4;
        o.nb = o.na + 2;
        o.full = (N & 3) == 0 && o.na + 4 <= N;
    }
    return o;
}
__device__ __forceinline__ i64 f_own_idx(const FOwn& o, const int k) { return (k < 2 ? o.na : o.nb) + (k & 1); }

// RAGGED (tail-free launches only): N is not a multiple of the tile.  1: N even -- the LOADS stay
// unconditional 16-byte accesses (X, lw and A are allocated with a tile of padding, the lanes beyond N
// read a neighbour's valid entries and are masked afterwards), only the stores test their indices;
// `full` remains a compile-time constant where it matters (see the loads below).  2: N odd -- pairs
// would straddle the islands' boundaries: every access tests its index.
template <int KIND, int FK, int OPT, bool SPEC, bool TAIL = true, int RAGGED = 0>
__global__ void __launch_bounds__(SMC_BLOCK)
k_propagate(const u32* __restrict__ pre_A, double* __restrict__ pre_info, const double* __restrict__ pre_params,
            const unsigned* __restrict__ pre_hcnt, const i64 pre_N, const int pre_geom, const FArgs av)
{
    // (leading scalar arguments = fields of the block that the launch's first loads are addressed with: the command
    //  processor preloads them into SGPRs -- -amdgpu-kernarg-preload-count -- and those loads leave without waiting for
    //  a scalar load of the kernarg segment; pre_geom: ntiles | xcd_chunks << 30)
    static_assert(OPT == 4, "two pairs per thread");
    const FArgs& a = av;
    __shared__ double smd[SMC_SM];
    __shared__ int s_last;
    SMC_NTAB_LDS(s_ntab);
    const int b = f_tile_xcd_geom(pre_geom, (int)blockIdx.x), isl = (int)blockIdx.y;
    const int tid = (int)threadIdx.x;
    F_STAMP(0);
    double* info = pre_info + (i64)isl * INFO_STRIDE;
    const i64 N = pre_N;
    const FOwn own = f_own<!TAIL>(b, tid, N);
    const bool full = (TAIL || RAGGED == 2) ? own.full : true;   // loads (tail-free: every thread reads 4)
    const bool full_st = (TAIL || RAGGED) ? own.full : true;     // stores
    const double r0 = smc_ldg(info), r1 = smc_ldg(info + 1), r2 = smc_ldg(info + 2),
                 r5 = smc_ldg(info + 5);
    constexpr bool APF = f_is_apf(FK);             // (tail-free two-level path only: see the tail)
    const double r6 = APF ? smc_ldg(info + 6) : 0.0;
    unsigned nh0 = 0u, nh1 = 0u;                   // registered heavy parents, either parity of t
    if (pre_hcnt) { nh0 = smc_ldg(pre_hcnt + (i64)isl * 2); nh1 = smc_ldg(pre_hcnt + (i64)isl * 2 + 1); }
    // SPEC: slots from a.par (kernarg): the ancestor indices are requested right behind the
    // step record, and the gather X_{t-1}[A] can leave as soon as they are back -- without
    // waiting for the record (read in vain on the steps that do not resample)
    u32 an[OPT] = {0u, 0u, 0u, 0u};
    auto load_anc = [&](const u32* Ap) {
        if (full) {
            smc_ld2g(Ap + own.na, an[0], an[1]);
            smc_ld2g(Ap + own.nb, an[2], an[3]);
        } else {
#pragma unroll
            for (int k = 0; k < OPT; ++k) {
                const i64 n = f_own_idx(own, k);
                an[k] = (n < N) ? smc_ldg(Ap + n) : 0u;
            }
        }
    };
    // ---- the step's standard normals depend on (seed, island, particle, t) only: with the host's
    // t (a.tk) one pair is generated while the ancestor indices are on their way and the other
    // behind the gather they trigger -- the first bytes of a dependent launch take 1.2-2 us to
    // arrive (profiles/r02i_trace_step.txt: "record"), the gather another ~0.8 us.  (Loads, first
    // pair and gather sit in ONE conditional block: a load left pending across a branch makes the
    // compiler wait for everything at the next write of its destination registers.)
    const u32 gisl = (u32)(a.island_offset + isl);
    double zs[OPT] = {0.0, 0.0, 0.0, 0.0};
    // tail-free launches over whole tiles: every thread owns 4 particles (no exec-mask region around
    // the speculative block: the loads below stay in flight across the staging barrier).  RAGGED == 1 (N even,
    // a ragged last tile) too: its loads are unconditional on padded arrays -- the threads beyond N read the
    // padding's valid entries (A is zero there) and never store
    constexpr bool ALL_IN = !TAIL && RAGGED != 2;
    const bool mine = ALL_IN || own.na < N;
    const bool spec_z = a.tk >= 0 && !a.zt && mine;
    double xg[OPT];
    // ---- request order: step record (above), Box-Muller tables (6 KB, L2-resident), ancestor
    // indices; the tables go to LDS as soon as they are back -- the indices are still on their way
    SmcNtabRegs<SMC_BLOCK> ntr;
    smc_ntab_fetch<SMC_BLOCK>(ntr, tid);
    // the island's model constants: staged in LDS with the tables (same barrier).  Read through the kernel
    // argument's pointer they are global loads the compiler has to repeat behind every store it cannot see through
    // (the streaming stores are inline asm with a memory clobber, the plain ones may alias) -- and a repeated global
    // load in the middle of the move / weigh phase waits, in order, for the stores just issued in front of it
    // (s_waitcnt vmcnt(0): a round trip to HBM).  From LDS they are ds_reads that wait for nothing of the kind.
    __shared__ double s_par[PARAM_STRIDE];
    double par_reg = 0.0;
    if (!SMC_PARAMS_GLOBAL && tid < PARAM_STRIDE) par_reg = smc_ldg(pre_params + (i64)isl * PARAM_STRIDE + tid);
#ifndef SMC_EMULATE
    asm volatile("" ::: "memory");                    // (the table loads are ISSUED first: vmcnt retires in order)
#endif
    if (SPEC && mine) load_anc(pre_A + (i64)isl * N);   // A always holds valid indices (zeros before the first resampling)
    // the tables go to LDS here, behind the requests for the indices: the wait in front of the ds_writes is for the
    // tables only (vmcnt retires in order), the indices stay in flight across the Philox calls below.  (Measured
    // against: the tables staged before the indices are requested, through global_load_lds_dwordx4, and the
    // indices requested first of all -- within 0.15 us of each other, this order ahead: profiles/r12_same_box_ab.txt r12ay)
    smc_ntab_store<SMC_BLOCK>(ntr, s_ntab, tid);
    // the Philox calls of the step's normals need no table: they run while it is on its way
    u64 pa0 = 0ull, pa1 = 0ull, pb0 = 0ull, pb1 = 0ull;
#ifndef SMC_PHILOX_LATE
    if (spec_z) {
        smc_normal_bits(a.seed, (u32)(own.na >> 1), (u32)a.tk, gisl, SMC_STREAM_NORMAL, pa0, pa1);
        smc_normal_bits(a.seed, (u32)(own.nb >> 1), (u32)a.tk, gisl, SMC_STREAM_NORMAL, pb0, pb1);
    }
#endif
    if (!SMC_PARAMS_GLOBAL && tid < PARAM_STRIDE) s_par[tid] = par_reg;
    __syncthreads();
#ifdef SMC_PHILOX_LATE                             // (A/B builds: the calls behind the table barrier, as in r04)
    if (spec_z) {
        smc_normal_bits(a.seed, (u32)(own.na >> 1), (u32)a.tk, gisl, SMC_STREAM_NORMAL, pa0, pa1);
        smc_normal_bits(a.seed, (u32)(own.nb >> 1), (u32)a.tk, gisl, SMC_STREAM_NORMAL, pb0, pb1);
    }
#endif
    if (SPEC && mine) {
        if (spec_z) smc_normal_from_bits(s_ntab, pa0, pa1, zs[0], zs[1]);
        const double* Xs = a.X + (i64)(a.par ^ 1) * a.xslot + (i64)isl * N;
#pragma unroll
        for (int k = 0; k < OPT; ++k) xg[k] = smc_ldg(Xs + an[k]);
    } else if (spec_z) {
        smc_normal_from_bits(s_ntab, pa0, pa1, zs[0], zs[1]);
    }
    if (spec_z) smc_normal_from_bits(s_ntab, pb0, pb1, zs[2], zs[3]);
    const i64 t = (i64)smc_uniform(r0);
    if (t >= a.T) return;
    F_STAMP(1);
    const double* p = SMC_PARAMS_GLOBAL ? a.params + (i64)isl * PARAM_STRIDE : s_par;
    const double yt = smc_uniform(r2);
    const double aux = m_has_aux<KIND>() ? smc_uniform(r5) : 0.0;
    double* Xn = (SPEC ? a.X + (i64)a.par * a.xslot : f_X(a, t)) + (i64)isl * N;
    const double* Xo = (SPEC ? a.X + (i64)(a.par ^ 1) * a.xslot : f_X(a, t - 1)) + (i64)isl * N;
    double* lwn = (SPEC ? a.lw + (i64)a.par * a.lslot : f_lw(a, t)) + (i64)isl * N;
    const double* lwo = (SPEC ? a.lw + (i64)(a.par ^ 1) * a.lslot : f_lw(a, t - 1)) + (i64)isl * N;
    const u32* A = f_A(a, t) + (i64)isl * N;
    const double* zt = a.zt ? a.zt + ((i64)t * a.zt_ts + (i64)isl * N) : nullptr;
    const bool first = (t == 0);
    const bool resample = !first && smc_uniform(r1) != 0.0;
    // ---- is this block of offspring wholly a registered heavy parent's?  Then that parent is
    // everybody's ancestor here and the block's A entries are ours to write
    i64 heavy_parent = -1;
    {
        const unsigned nh = smc_uniform((t & 1) ? nh1 : nh0);
        if (resample && nh) {
            const i64 w0 = (i64)b * SMC_BLOCK * OPT;
            const i64* hl = a.hlist + ((i64)isl * 2 + (t & 1)) * F_HMAX * 3;
            for (unsigned e = 0; e < nh && e < F_HMAX; ++e)
                if (hl[3 * e] <= w0 && w0 + (i64)SMC_BLOCK * OPT <= hl[3 * e + 1]) heavy_parent = hl[3 * e + 2];
        }
        if (a.hcnt && b == 0 && tid == 0) a.hcnt[(i64)isl * 2 + ((t & 1) ^ 1)] = 0u;    // next step's list
    }

    double lw[OPT];
    double xkeep[OPT] = {0.0, 0.0, 0.0, 0.0};      // APF: the new particles, for the auxiliary weights
#pragma unroll
    for (int k = 0; k < OPT; ++k) lw[k] = -INFINITY;
    if (mine) {                                     // (whole tiles: every thread -- no path around the block on which
        // the speculative loads would still be pending, which cost a vmcnt(0) -- the X / lw stores' acknowledgement
        // -- at the join, in front of the tile's weights)
        double xp[OPT], lwp[OPT], z[OPT];
        // ---- ancestor indices (when resampled) or the particle's own state and
        // log-weight: requested first, consumed after the normals are generated
        if (resample && !SPEC) {
            load_anc(A);
        } else if (!first && !resample) {
            if (full) {
                smc_ld2g(Xo + own.na, xp[0], xp[1]);
                smc_ld2g(lwo + own.na, lwp[0], lwp[1]);
                smc_ld2g(Xo + own.nb, xp[2], xp[3]);
                smc_ld2g(lwo + own.nb, lwp[2], lwp[3]);
            } else {
#pragma unroll
                for (int k = 0; k < OPT; ++k) {
                    const i64 n = f_own_idx(own, k);
                    xp[k] = (n < N) ? smc_ldg(Xo + n) : 0.0;
                    lwp[k] = (n < N) ? smc_ldg(lwo + n) : 0.0;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < OPT; ++k) { xp[k] = 0.0; lwp[k] = 0.0; }
        }
        // ---- standard normals: one Philox call per (even, odd) pair, or the tape
        if (zt) {
#pragma unroll
            for (int k = 0; k < OPT; ++k) {
                const i64 n = f_own_idx(own, k);
                z[k] = (n < N) ? smc_ldg(zt + n) : 0.0;
            }
        } else if (spec_z && t == a.tk) {
#pragma unroll
            for (int k = 0; k < OPT; ++k) z[k] = zs[k];
        } else {
            smc_normal_pair(s_ntab, a.seed, (u32)(own.na >> 1), (u32)t, gisl, SMC_STREAM_NORMAL, z[0], z[1]);
            smc_normal_pair(s_ntab, a.seed, (u32)(own.nb >> 1), (u32)t, gisl, SMC_STREAM_NORMAL, z[2], z[3]);
        }
        if (resample && heavy_parent >= 0) {
            const double xh = smc_ldg(Xo + heavy_parent);
            u32* Aw = f_A(a, t) + (i64)isl * N;
#pragma unroll
            for (int k = 0; k < OPT; ++k) {
                const i64 n = f_own_idx(own, k);
                xp[k] = xh;
                lwp[k] = 0.0;
                if (n < N) Aw[n] = (u32)heavy_parent;
            }
        } else if (resample) {
#pragma unroll
            for (int k = 0; k < OPT; ++k) { xp[k] = SPEC ? xg[k] : smc_ldg(Xo + an[k]); lwp[k] = 0.0; }    // core.py:332
        }
        if (APF && resample) {
            // core.py:299-305 reset_weights: log_mean_exp(logeta, W) - logeta[A]; the constant comes
            // with the record (k_reduce2), logeta of the gathered parent is formed again here
            const double cconst = smc_uniform(r6);
#pragma unroll
            for (int k = 0; k < OPT; ++k) lwp[k] = cconst - m_logeta<KIND>(p, xp[k], yt);
        }
        F_STAMP(2);
        double xn[OPT];
#pragma unroll
        for (int k = 0; k < OPT; ++k) {
            double inc;
            xn[k] = m_step<KIND, FK>(p, first, yt, aux, xp[k], z[k], inc);
            double l = (first || (resample && !APF)) ? inc : lwp[k] + inc;    // resampling.py:241-244
            if (l != l) l = -INFINITY;                                     // resampling.py:220
            lw[k] = (ALL_IN && RAGGED == 0) ? l : ((f_own_idx(own, k) < N) ? l : -INFINITY);   // (whole tiles: all in)
            if (APF) xkeep[k] = xn[k];
        }
        if (full_st) {
            if (a.nt & 1) {
                smc_st2g_nt(Xn + own.na, xn[0], xn[1]);
                smc_st2g_nt(Xn + own.nb, xn[2], xn[3]);
            } else {
                smc_st2g(Xn + own.na, xn[0], xn[1]);
                smc_st2g(Xn + own.nb, xn[2], xn[3]);
            }
            if (a.nt & 2) {
                smc_st2g_nt(lwn + own.na, lw[0], lw[1]);
                smc_st2g_nt(lwn + own.nb, lw[2], lw[3]);
            } else {
                smc_st2g(lwn + own.na, lw[0], lw[1]);
                smc_st2g(lwn + own.nb, lw[2], lw[3]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < OPT; ++k) {
                const i64 n = f_own_idx(own, k);
                if (n < N) { smc_stg(Xn + n, xn[k]); smc_stg(lwn + n, lw[k]); }
            }
        }
    }
    F_STAMP(3);
    if (TAIL) {
        // ---- the workgroup's (max, sum e, sum e^2); its last arriver finalises the step
        const SmcLse r = f_tile_lse<OPT>(lw, smd);
        f_step_tail(a, isl, b, t, first, resample, r, smd, s_last, info);
    } else {
        // two-level path: the tile's partial and its integer CDF are all this launch owes;
        // k_ancestors2(t+1) -- every workgroup of it -- reduces the partials, so nobody waits for
        // a last workgroup here (tiles are full, or RAGGED: -inf weights beyond N)
        u64 cx[4];
        if (APF) {
            // the partial of the PLAIN weights first (evidence, the logged ESS, W): pm2 / ps2 / pss2;
            // then everything the resampling of step t+1 reads -- partial, integer CDF, tile total --
            // from the auxiliary weights lw + logeta(t, X) with data[t+1] (core.py:307-313)
            const F2Tile r2 = f2_tile_weights(lw, cx);
            if (tid == 0) {
                const i64 o2 = (i64)isl * a.nparts;
                a.pm2[o2 + b] = r2.K;
                a.ps2[o2 + b] = r2.S;
                a.pss2[o2 + b] = r2.SS;
            }
            if (t + 1 < a.T) {
                const double y_next = a.y[(t + 1) * a.dy];
#pragma unroll
                for (int k = 0; k < OPT; ++k) {
                    double la = lw[k] + m_logeta<KIND>(p, xkeep[k], y_next);
                    if (la != la) la = -INFINITY;
                    lw[k] = (lw[k] > -INFINITY) ? la : -INFINITY;
                }
            }
        }
        const F2Tile r = (!APF && a.strict_e) ? f2_tile_weights_e(lw, cx) : f2_tile_weights(lw, cx);
        u64* cq = a.cq + (i64)isl * (RAGGED ? a.ncq : N);          // (whole tiles: ncq == N, already in registers)
        if (a.nt & 4) { smc_st2g_nt(cq + own.na, cx[0], cx[1]); smc_st2g_nt(cq + own.nb, cx[2], cx[3]); }
        else { smc_st2g(cq + own.na, cx[0], cx[1]); smc_st2g(cq + own.nb, cx[2], cx[3]); }
        if (tid == 0) {
            const i64 o = (i64)isl * a.nparts;
            a.pm[o + b] = r.K;
            a.ps[o + b] = r.S;
            a.pss[o + b] = r.SS;
            a.tq[o + b] = r.tb;
            if (b == 0) a.info2[(i64)isl * INFO_STRIDE] = (double)(t + 1);
        }
        F_STAMP(4);
    }
}

// ---------------------------------------------------------------------------
// Two-level CDF (N = 2^k, at least 2 tiles per island; systematic / stratified with closed-form
// counts, multinomial with counts by search over the sorted uniforms): the step loop without
// any intra-launch exchange.  Contract (restated in oracle/oracle.c, checked bit for bit):
//
// Weights travel as pairs (p, k), exp(lw) = p 2^k (smc_expk): no maximum is needed to form them
// and every change of reference is an exact power-of-two scaling.
// k_propagate<.., TAIL = false>(t-1) ends, per aligned tile b of 1024 new particles, with
//   K_b = max k_i,  e_i = p_i 2^(k_i - K_b),  the partial (K_b, S_b = sum e, SS_b = sum e^2),
//   the tile's integer CDF  q_i = rint(e_i 2^49), c_j = sum_{i<j} q_i (stored: 8 B per particle),
//   t_b = sum q_i  -- no tickets, no last workgroup.
// EVERY workgroup of k_ancestors2(t) reduces the <= 1024 partials itself (same loads, same
// order, same bits everywhere; k_reduce2 does it once per island for larger grids):
//   K = max K_b, s = sum S_b 2^(K_b-K), ss likewise -> ESS, the resample decision, and each tile's
//   share of the 2^52 scale  Q_b = rint(S_b 2^(K_b-K) / s 2^52),  G_b = sum_{b' < b} Q_b'
//   (integers below 2^53, summed exactly in fp64).
// Parent j of tile b then owns the offspring n with
//     count(G_b + floor(c_j Q_b / t_b)) <= n < count(G_b + floor(c_{j+1} Q_b / t_b)),
//     count(C) = #{ n : T_n <= C },  T_n = ceil(su_n 2^52)
// -- i.e. offspring n with G_b < T_n <= G_b + Q_b belongs to the parent j of tile b with
// c_j Q_b < (T_n - G_b) t_b <= c_{j+1} Q_b: an exact rational comparison, deterministic for any
// schedule.  It agrees with the sequential fp64 CDF of the reference (resampling.py:500-509)
// except within rounding distance of a CDF step.  Neither kernel of the step evaluates an exp
// more than once per particle, and k_ancestors2 none at all.
// Workgroup 0 writes the summary row of step t-1 and the record k_propagate(t) reads.
// ---------------------------------------------------------------------------
// S_b 2^(K_b - K) and SS_b 2^(2 (K_b - K)): a tile's sums on the island's reference (exact)
__device__ __forceinline__ void f2_rescale(const double Kb, const double K, const double S, const double SS,
                                           double& v, double& w)
{
    double d = Kb - K;
    d = (d > -2000.0) ? d : -2000.0;
    const int di = (int)d;
    v = ldexp(S, di);
    w = ldexp(SS, 2 * di);
}
__device__ __forceinline__ double f2_share(const double v, const double rs)
{
    const double w = v * rs;
    return (w > 0.0) ? rint(fmin(w, 2.0) * 4503599627370496.0) : 0.0;      // 2^52
}
__device__ __forceinline__ void f2_finish(const FArgs& a, F2Red& r)
{
    r.bad = !(r.K > -INFINITY) || !(r.K < INFINITY);
    r.ess = r.bad ? NAN : (r.s * r.s) / r.ss;                              // resampling.py:226
    r.rs = r.bad ? NAN : 1.0 / r.s;
}
// log of the mean weight, m + log(s / N) with m = K ln 2 (resampling.py:224); one thread
__device__ __forceinline__ double f2_log_mean(const FArgs& a, const F2Red& r)
{
    if (r.bad) return NAN;
    return r.K * 6.93147180369123816490e-01 + (r.K * 1.90821492927058770002e-10 + log(r.s / (double)a.N));
}
// What the record of step t takes from memory (earlier launches wrote all of it): requested in ONE go, and early -- the
// writer is one thread of one workgroup, and written as "load, use, store, load ..." its five loads were five round trips
// in a row (a store may alias the next load, so the compiler keeps the order; vmcnt counts stores, so every wait for a load
// also waited for the stores in front of it): 2 us during which that workgroup's other threads stood at the next barrier,
// in a kernel that takes 6.5 us because its slowest workgroup does.
struct F2RecIn {
    double row4, lm_prev, cum_prev, y, aux;
};
__device__ __forceinline__ F2RecIn f2_record_loads(const FArgs& a, const int isl, const i64 t, const bool row_only = false)
{
    F2RecIn in;
    const double* row = a.summ + ((i64)isl * (a.T + 1) + (t - 1)) * SUMM_STRIDE;
    const bool first = t == 1;
    in.row4 = smc_ldg(row + 4);
    in.lm_prev = first ? 0.0 : smc_ldg(row + 1 - SUMM_STRIDE);
    in.cum_prev = first ? 0.0 : smc_ldg(row + 3 - SUMM_STRIDE);
    in.y = row_only ? 0.0 : smc_ldg(a.y + t * a.dy);          // (row_only: the last row, t = T -- there is no y_T)
    in.aux = (a.aux && !row_only) ? smc_ldg(a.aux + t) : 0.0;
    return in;
}
// summary row of step ts (the step the partials belong to) -- core.py:355-359
__device__ __forceinline__ void f2_write_row(const FArgs& a, const int isl, const i64 ts, const F2Red& r, const F2RecIn& in)
{
    double* row = a.summ + ((i64)isl * (a.T + 1) + ts) * SUMM_STRIDE;
    const bool first = (ts == 0);
    const bool resampled = in.row4 != 0.0;             // written when step ts was decided
    const double log_mean = f2_log_mean(a, r);
    const double loglt = (first || resampled) ? log_mean : log_mean - in.lm_prev;
    row[0] = r.ess;
    row[1] = log_mean;
    row[2] = loglt;
    row[3] = (first ? 0.0 : in.cum_prev) + loglt;
    row[5] = r.K;                                      // W = p 2^(k - K) / s  (k_f_write_W)
    row[6] = r.rs;
}
__device__ __forceinline__ void f2_write_row(const FArgs& a, const int isl, const i64 ts, const F2Red& r)
{
    f2_write_row(a, isl, ts, r, f2_record_loads(a, isl, ts + 1, true));
}

// the sorted uniforms of step t (systematic: the one draw; stratified: read per offspring);
// kq: the "k" the count functions of smc_resample.h are called with -- they use 2^(62-k) per
// offspring, the shares here live on the 2^52 scale, hence kq = log2 N + 10
__device__ __forceinline__ void f2_su(const FArgs& a, const int isl, const i64 t, SmcSu& su, u64& Us, const int scheme_k = 0)
{
    const int scheme = scheme_k ? scheme_k : a.scheme;       // (scheme_k: the kernel's compile-time copy, see k_ancestors2)
    su.scheme = scheme;
    su.M = a.N;
    su.dM = (double)a.N;
    su.u = a.ut ? a.ut + ((i64)t * a.n_islands + isl) * a.ut_stride
                : (scheme == SMC_MULTINOMIAL_ ? a.su + (i64)isl * a.N : nullptr);
    su.u_sys = 0.0;
    su.seed = a.seed;
    su.t = (u32)t;
    su.island = (u32)(a.island_offset + isl);
    if (scheme == SMC_SYSTEMATIC_) {
        if (su.u) {
            su.u_sys = su.u[0];
        } else {
            u64 x0, x1;
            smc_philox_uniform(0u, su.t, su.island, SMC_STREAM_RESAMPLE, su.seed, x0, x1);
            su.u_sys = smc_u01_halfopen(x0);
        }
    }
    Us = a.log2N >= 0 ? (u64)(su.u_sys * __longlong_as_double((long long)(1023 + F2_SBITS - a.log2N) << 52))
                      : 0ull;                      // (the integer shortcut of N = 2^k only)
}
// ---- any N (not a power of two).  su_n = fl(fl(u_n + n) / N) lies in [fl(n / N), fl((n + 1) / N)]
// (fl(u_n + n) in [n, n + 1]; rounding is monotone), so with B_n = ceil(fl(n / N) 2^52) the thresholds
// T_n = ceil(su_n 2^52) satisfy B_n <= T_n <= B_{n+1} and, exactly as for N = 2^k,
//     count(C) = #{ n : T_n <= C } = nc + [T_nc <= C],   nc = max{ n <= N : B_n <= C }.
// nc is floor(C N 2^-52) up to the roundings of the definition: an fp64 guess, fixed with the
// definition itself (a step or two); the oracle counts by bisection on the same definition.
__device__ __forceinline__ u64 f2_t52_div(const double x, const double dN)
{
    const double v = x / dN;
    return (v > 0.0) ? (u64)ceil(fmin(v, 2.0) * 4503599627370496.0) : 0ull;
}
__device__ __forceinline__ i64 f2_nc_general(const u64 C, const double dN, const i64 N)
{
    const double z = (double)C * (dN * 0x1.0p-52);
    i64 g = (z >= dN) ? N : (i64)z;
    while (g < N && f2_t52_div((double)(g + 1), dN) <= C) ++g;
    while (g > 0 && f2_t52_div((double)g, dN) > C) --g;
    return g;
}
__device__ __forceinline__ i64 f2_count_general(const u64 C, const SmcSu& su)
{
    const i64 nc = f2_nc_general(C, su.dM, su.M);
    if (nc >= su.M) return su.M;
    const double un = (su.scheme == SMC_SYSTEMATIC_) ? su.u_sys : smc_strat_u(su, (u64)nc);
    return nc + (f2_t52_div(un + (double)nc, su.dM) <= C ? 1 : 0);
}
// the same for a C known only to within +-E (E far below 2^52 / N): decided whenever neither nc nor
// the comparison can change inside the band; -1 otherwise (cf. smc_count_pow2_band).
// z = Ch N 2^-52 in fp64; B_n differs from n 2^52 / N by < 2 units of the
// scale, z from its real value by < N 2^-52 -- so unless z lies within (E + 4) N 2^-52 of an integer,
// nc = floor(z) for every C of the band and only T_nc (one division) is left to compare
__device__ __forceinline__ i64 f2_count_band_general_fast(const u64 Ch, const u64 E, const SmcSu& su)
{
    const double sN = su.dM * 0x1.0p-52;
    const double z = (double)Ch * sN;
    const double fz = floor(z);
    const double d = z - fz, delta = (double)(E + 4ull) * sN;
    if (!(d > delta && d < 1.0 - delta) || !(fz < su.dM)) return -1;
    const i64 nc = (i64)fz;
    const double un = (su.scheme == SMC_SYSTEMATIC_) ? su.u_sys : smc_strat_u(su, (u64)nc);
    const u64 T = f2_t52_div(un + fz, su.dM);
    if (T + E <= Ch) return nc + 1;
    if (T > Ch + E) return nc;
    return -1;
}
// POW2: a compile-time copy of a.log2N >= 0 -- the kernels of N = 2^k carry none of the general code
template <bool POW2>
__device__ __forceinline__ i64 f2_count(const FArgs& a, const SmcSu& su, const u64 Us, const u64 C)
{
    if (!POW2) return f2_count_general(C, su);
    const int kq = a.log2N + (62 - F2_SBITS);
    return su.scheme == SMC_SYSTEMATIC_ ? smc_sys_count_pow2_fast(C, su.u_sys, Us, kq, a.N)
                                       : smc_strat_count_pow2(C, su, kq, a.N);
}
struct F2Fast {
    double Gd, r, u, eps, one_m_eps, dN;
    u64 Gb, Qb, tb;
};
// First offspring of the parent at position c (0 <= c <= t_b) of tile b's local CDF, systematic,
// N = 2^k: ns = count(C), C = G_b + floor(c Q_b / t_b), count(C) = #{n : fl(u + n) 2^sh <= C}
// (sh = 52 - k).  With Y = C 2^-sh: ns = floor(Y - u) + 1 whenever Y - u is not within the
// rounding errors of an integer.  fp64 evaluation z = fma((double)c, r, Gd) - u with
// r = fl(Q_b / fl(t_b)) 2^-sh, Gd = G_b 2^-sh (exact): |z - (Y - u)| <= (3 ulp on the product,
// 1 on the fma, 1 on the subtraction, 1 for fl(u + n), 2^-sh for the floor in C) x N 2^-53
// < 8 N 2^-53; the band is 16 N 2^-53.  Outside it the floor and the comparison are decided;
// inside (probability 2^-28 per parent at N = 2^20) the exact integer route is taken.
template <bool POW2>
__device__ __forceinline__ i64 f2_ns_sys(const FArgs& a, const SmcSu& su, const u64 Us, const F2Fast& f,
                                         const u64 c)
{
    const double z = fma((double)c, f.r, f.Gd) - f.u;
    const double fl = floor(z);
    const double d = z - fl;
    if (d > f.eps && d < f.one_m_eps) {
        double v = fl + 1.0;
        v = v < 0.0 ? 0.0 : v;
        v = v > f.dN ? f.dN : v;
        return (i64)(u32)v;
    }
    if (!POW2) return -1;                 // general N: resolved by f2_resolve_general (one copy of the code)
    return f2_count<POW2>(a, su, Us, f.Gb + smc_muldiv_floor(c, f.Qb, f.tb));
}
// The same for the stratified draw, N = 2^k: nc = floor(Y) (Y not within the band of an integer), then
// count = nc + [T_nc <= C] with T_nc <= C <=> fl(u_nc + nc) <= Y <=> u_nc <= Y - nc up to the same roundings:
// decided unless u_nc lies within the band of Y - nc.  (Round 3: this replaces the integer band test of
// smc_count_pow2_band on this path -- the same uniforms, the same counts, ~30 instructions less per parent.)
__device__ __forceinline__ i64 f2_ns_strat(const FArgs& a, const SmcSu& su, const u64 Us, const F2Fast& f, const u64 c)
{
    const double Y = fma((double)c, f.r, f.Gd);
    const double fl = floor(Y);
    const double d = Y - fl;
    if (d > f.eps && d < f.one_m_eps) {
        if (!(fl < f.dN)) return (i64)f.dN;
        const double e = d - smc_strat_u(su, (u64)fl);
        if (e > f.eps) return (i64)(u32)fl + 1;
        if (e < -f.eps) return (i64)(u32)fl;
    }
    return f2_count<true>(a, su, Us, f.Gb + smc_muldiv_floor(c, f.Qb, f.tb));
}
// ... with the uniforms of the offspring nfirst, nfirst + 1, ... staged in LDS by the workgroup (k_ancestors2): every
// floor(Y) of the tile lies inside the staged window (the caller checks the tile's share against its size)
#define F2_SU_PAIRS 640                 /* staged pairs of stratified uniforms per tile: 1280 offspring, 1.25 x the average */
__device__ __forceinline__ i64 f2_ns_strat_lds(const FArgs& a, const SmcSu& su, const u64 Us, const F2Fast& f, const u64 c,
                                               const double* sU, const double nfirst)
{
    const double Y = fma((double)c, f.r, f.Gd);
    const double fl = floor(Y);
    const double d = Y - fl;
    if (d > f.eps && d < f.one_m_eps) {
        if (!(fl < f.dN)) return (i64)f.dN;
        const int k = (int)(fl - nfirst);
        const double e = d - sU[k];
        if (e > f.eps) return (i64)(u32)fl + 1;
        if (e < -f.eps) return (i64)(u32)fl;
    }
    return f2_count<true>(a, su, Us, f.Gb + smc_muldiv_floor(c, f.Qb, f.tb));
}
// general N: the counts the fast tests left open (bit i of `need`: position c_i), by the definition.
// A loop that is NOT unrolled around the one inlined copy of the exact route: it runs for one parent
// in 2^28 (systematic) and must cost the common path neither registers nor a call.
__device__ __forceinline__ void f2_resolve_general(const SmcSu& su, unsigned need, const u64 Gb, const u64 Qb,
                                                   const u64 tb, const u64 (&c)[7], i64 (&ns)[7])
{
#pragma unroll 1
    for (int i = 0; i < 7; ++i) {
        if (!((need >> i) & 1u)) continue;
        u64 ci = c[0];
#pragma unroll
        for (int k = 1; k < 7; ++k) ci = (i == k) ? c[k] : ci;
        const u64 pos = (ci == 0ull) ? 0ull : (ci >= tb ? Qb : smc_muldiv_floor(ci, Qb, tb));
        const i64 v = f2_count_general(Gb + pos, su);
#pragma unroll
        for (int k = 0; k < 7; ++k) ns[k] = (i == k) ? v : ns[k];
    }
}
// ---- multinomial on the two-level path: the sorted uniforms sit in memory (the tape, or the
// exponential spacings k_f_spacing_* left in a.su), thresholds T_n = ceil(su_n 2^52) on the
// scale of the shares; count(C) = #{ n : T_n <= C } is a search
#define F2_MW 2048                      /* thresholds staged in LDS per tile (its offspring: 1024 +- a few dozen) */
__device__ __forceinline__ u64 f2_t52(const double su)
{
    return (su > 0.0) ? (u64)ceil(fmin(su, 2.0) * 4503599627370496.0) : 0ull;
}
// by a whole wavefront: a 64-ary search (log64 M dependent loads); every lane takes part and
// gets the result (cf. smc_su_count_le_wave, the flat path's twin on the 2^62 scale)
__device__ __forceinline__ i64 f2_count_sorted_wave(const double* u, const i64 M, const u64 C)
{
    i64 lo = 0, hi = M;                   // T_n <= C on [0, lo), > C on [hi, M)
    const int lane = smc_lane();
    while (lo < hi) {
        const i64 width = hi - lo;
        const i64 stride = (width + 63) >> 6;
        const i64 p = lo + (i64)lane * stride;
        const bool in = p < hi;
        const bool le = in && (f2_t52(smc_ldg(u + (in ? p : lo))) <= C);
        const int cnt = (int)smc_wave_sum_u64(le ? 1ull : 0ull);      // monotone: the first cnt probes
        const int nprobe = (int)((width + stride - 1) / stride);
        const i64 nlo = cnt > 0 ? lo + (i64)(cnt - 1) * stride + 1 : lo;
        const i64 nhi = cnt < nprobe ? lo + (i64)cnt * stride : hi;
        lo = nlo;
        hi = nhi;
    }
    return lo;
}
// #{ i < n : T[i] <= C } in a sorted LDS window
__device__ __forceinline__ int f2_count_lds(const u64* T, const int n, const u64 C)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (T[mid] <= C) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// the same over memory, one lane: first index in [lo, hi) whose threshold exceeds C (tiles with
// more than F2_MW offspring: collapsed weights, where speed is not the concern)
__device__ inline i64 f2_count_sorted_range(const double* u, i64 lo, i64 hi, const u64 C)
{
    while (lo < hi) {
        const i64 mid = lo + ((hi - lo) >> 1);
        if (f2_t52(smc_ldg(u + mid)) <= C) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---- multinomial, production (Philox) mode.  Draw n has the integer spacing q_n (f_spacing_q4),
// Z_n = q_0 + .. + q_n, su_n = fl(Z_n) / fl(Z_N) (resampling.py:536-537: z[:-1] / z[-1]), written ONCE by
// k_f_spacing_write; T_n = ceil(su_n 2^52).  k_f_spacing_sums / _scan leave the exclusive prefix of every
// tile of 1024 draws (E[k] = Z_{1024 k - 1}) and the total (E[ntiles1] = Z_N), so the LAST threshold of
// every tile is known without touching the uniforms: B_k = T_{1024 k - 1} = t52(E[k] / all).  Hence
//     count(C) = #{n : T_n <= C} = 1024 k* + #{n in tile k* : T_n <= C},  k* = max{k : B_k <= C}  (B_0 = 0)
// -- one probe of the 32 KB of prefixes (a window of 64 tiles centred on the expected one) instead of
// four dependent rounds of a 64-ary search over the N uniforms; the tiles k*(G_b) .. k*(G_b + Q_b) (2,
// sometimes 1 or 3: a tile of 1024 parents owns 1024 +- a few dozen offspring) are then staged in LDS
// whole and every boundary of the tile is a bisection there.
// (Measured alternative, r04d: never writing the uniforms and REGENERATING those tiles inside
//  k_ancestors2 brings its traffic down to 1.05x the bytes it must move but costs 2 700 VALU
//  instructions per wave -- 94 us against 42 at N = 2^22: on this part 8 bytes from HBM are four times
//  cheaper than 54 instructions.)
struct F2Regen {
    const u64* E;          // (ntiles1 + 1) tile prefixes, the total last
    int ntiles1;
    double dall, rdall;    // fl(Z_N) and its correctly rounded reciprocal (smc_div_c: the IEEE quotient)
};
__device__ __forceinline__ u64 f2_regen_B(const F2Regen& g, const i64 k)
{
    return f2_t52(smc_div_c((double)smc_ldg(g.E + k), g.dall, g.rdall));
}
// #{k in [0, ntiles1] : B_k <= C} >= 1 by a whole wavefront: first a window of 64 tiles centred on the
// expected one (the uniforms deviate from n / N by O(1 / sqrt N): +- a tile or two), then 64-ary.
// Bl, Bh: the bracketing values B_{k*} <= C < B_{k* + 1} (the last threshold of tile k* - 1 and of
// tile k*): they come out of the probe's own lanes when the window brackets the answer (one round
// trip in all), else they are loaded.
__device__ __forceinline__ i64 f2_regen_tiles_le_wave(const F2Regen& g, const u64 C, const i64 guess, u64& Bl, u64& Bh)
{
    const i64 M = (i64)g.ntiles1 + 1;
    const int lane = smc_lane();
    i64 lo = 0, hi = M;                                        // B_k <= C on [0, lo), > C on [hi, M)
    bool have = false;
    {
        i64 w0 = guess - 32;
        w0 = w0 > M - 64 ? M - 64 : w0;
        w0 = w0 < 0 ? 0 : w0;
        const i64 k = w0 + lane;
        const bool in = k < M;
        const u64 Bk = in ? f2_regen_B(g, k) : ~0ull;
        const bool le = in && Bk <= C;
        const int cnt = (int)smc_wave_sum_u64(le ? 1ull : 0ull);
        const int nprobe = (int)(M - w0 < 64 ? M - w0 : 64);
        if (cnt > 0) lo = w0 + cnt;
        if (cnt < nprobe) hi = w0 + cnt;
        if (cnt == 0) hi = w0;
        if (hi < lo) hi = lo;
        if (cnt > 0 && cnt < nprobe) {                         // bracketed inside the window
            Bl = smc_readlane64(Bk, cnt - 1);
            Bh = smc_readlane64(Bk, cnt);
            have = true;
        }
    }
    while (lo < hi) {
        const i64 width = hi - lo;
        const i64 stride = (width + 63) >> 6;
        const i64 p = lo + (i64)lane * stride;
        const bool in = p < hi;
        const bool le = in && f2_regen_B(g, in ? p : lo) <= C;
        const int cnt = (int)smc_wave_sum_u64(le ? 1ull : 0ull);
        const int nprobe = (int)((width + stride - 1) / stride);
        const i64 nlo = cnt > 0 ? lo + (i64)(cnt - 1) * stride + 1 : lo;
        const i64 nhi = cnt < nprobe ? lo + (i64)cnt * stride : hi;
        lo = nlo;
        hi = nhi;
    }
    if (!have) {
        Bl = f2_regen_B(g, lo - 1);
        Bh = lo < M ? f2_regen_B(g, lo) : ~0ull;
    }
    return lo;
}
// The same answer from a window of 64 tile prefixes that was requested BEFORE C was known (k_ancestors2 asks for
// the prefixes around its own tile index with its first loads: the tile shares deviate from 1 / ntiles by a few
// per cent, the window covers +- 32 tiles) -- when it brackets C the probe costs no round trip of its own.
// Ek: prefix w0 + lane (any value beyond the table).  Returns the count, or -1: not bracketed, probe again.
__device__ __forceinline__ i64 f2_regen_tiles_le_spec(const F2Regen& g, const u64 C, const i64 w0, const u64 Ek,
                                                      u64& Bl, u64& Bh)
{
    const i64 M = (i64)g.ntiles1 + 1;
    const int lane = smc_lane();
    const i64 k = w0 + lane;
    const bool in = k < M;
    const u64 Bk = in ? f2_t52(smc_div_c((double)Ek, g.dall, g.rdall)) : ~0ull;
    const bool le = in && Bk <= C;
    const int cnt = (int)smc_wave_sum_u64(le ? 1ull : 0ull);
    const int nprobe = (int)(M - w0 < 64 ? M - w0 : 64);
    if (cnt > 0 && cnt < nprobe) {
        Bl = smc_readlane64(Bk, cnt - 1);
        Bh = smc_readlane64(Bk, cnt);
        return w0 + cnt;
    }
    return -1;
}
// #{ i < n : T[i] <= C } in a sorted LDS window of integer-valued doubles
__device__ __forceinline__ int f2_count_lds_f64(const double* T, const int n, const double C)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (T[mid] <= C) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---- SQMC (smc_filter_sqmc.h): the n-th smallest first coordinate of the N = 2^k points of step t is
// x_0 = (n << (30 - k)) | (low bits of the digital shift), su_n = safe_generate(x_0 2^-30) (rqmc.py:9-13): the
// thresholds T_n = ceil(su_n 2^52) are a function of n, and count(C) = #{n : T_n <= C} is a guess fixed with
// that definition (T is monotone in n)
struct F2Sq {
    u32 low;               // the shift's low 30 - k bits
    int sh;                // 30 - k
    i64 N;
};
__device__ __forceinline__ void f2_sq_init(const FArgs& a, const int isl, const i64 t, F2Sq& q)
{
    const u64 ctr = a.sq_ctr + (u64)t + ((u64)(u32)(a.island_offset + isl) << 32);
    u64 x0, x1;
    smc_philox_uniform(0u, (u32)ctr, (u32)(ctr >> 32), SMC_STREAM_RESAMPLE, a.sq_seed, x0, x1);
    const u32 sh0 = (u32)(x0 >> 34);                                      // (smc_sobol_shift: 30 bits)
    q.sh = 30 - a.log2N;
    q.low = sh0 & ((1u << q.sh) - 1u);
    q.N = a.N;
}
__device__ __forceinline__ u64 f2_sq_T(const F2Sq& q, const i64 n)
{
    const u32 x = ((u32)n << q.sh) | q.low;
    const double u = (double)x * (1.0 / 1073741824.0);
    return f2_t52(0.5 + (1.0 - 1e-10) * (u - 0.5));                       // smc_sobol_safe
}
__device__ __forceinline__ i64 f2_sq_count(const F2Sq& q, const u64 C)
{
    const double y = ((double)C * 0x1.0p-52 - 0.5) / (1.0 - 1e-10) + 0.5;
    double g = floor((y * 1073741824.0 - (double)q.low) * __longlong_as_double((long long)(1023 - q.sh) << 52));
    g = g < -1.0 ? -1.0 : g;
    g = g > (double)(q.N - 1) ? (double)(q.N - 1) : g;
    i64 n = (i64)g;                                                        // last threshold at or below C (guess)
    while (n + 1 < q.N && f2_sq_T(q, n + 1) <= C) ++n;
    while (n >= 0 && f2_sq_T(q, n) > C) --n;
    return n + 1;
}

// first offspring ns[i] of the parents jt+i at positions cx[i] (cx[4]: the next thread's first
// parent, t_b for the last thread), any scheme: fp64 quotient within 2^12 of the truth, count
// decided unless the position lies within that band of a threshold, else formed exactly
template <bool POW2>
__device__ __forceinline__ void f2_first_offspring(const FArgs& a, const SmcSu& su, const u64 Us,
                                                   const u64 (&cx)[F_IPT + 1], const u64 tb,
                                                   const u64 Gb, const u64 Qb, const i64 jt,
                                                   i64 (&ns)[F_IPT + 1])
{
    const i64 N = a.N;
    const int kq = a.log2N + (62 - F2_SBITS);
    const double qscale = (double)Qb / (double)(tb ? tb : 1ull);
#pragma unroll
    for (int i = 0; i <= F_IPT; ++i) {
        const i64 j = jt + i;
        const u64 c = cx[i];
        if (j == 0) ns[i] = 0;
        else if (j >= N) ns[i] = N;
        else {
            u64 qh = (u64)((double)c * qscale);
            qh = qh > Qb ? Qb : qh;
            i64 cnt = a.exact_counts ? -1
                    : (POW2 ? smc_count_pow2_band(Gb + qh, 1ull << 13, su, su.u_sys, Us, kq, N)
                            : f2_count_band_general_fast(Gb + qh, 1ull << 13, su));
            if (cnt < 0 && POW2) cnt = f2_count<POW2>(a, su, Us, Gb + smc_muldiv_floor(c, Qb, tb));
            ns[i] = cnt;                          // (general N: -1 = open, see f2_resolve_general)
        }
    }
    if (!POW2) {
        unsigned need = 0u;
        u64 c7[7] = {cx[0], cx[1], cx[2], cx[3], cx[4], 0ull, 0ull};
        i64 n7[7] = {ns[0], ns[1], ns[2], ns[3], ns[4], 0, 0};
#pragma unroll
        for (int i = 0; i <= F_IPT; ++i) need |= (ns[i] < 0) ? (1u << i) : 0u;
        if (need) {
            f2_resolve_general(su, need, Gb, Qb, tb, c7, n7);
#pragma unroll
            for (int i = 0; i <= F_IPT; ++i) ns[i] = n7[i];
        }
    }
}

// the decision of step t and what k_propagate(t) reads (one thread)
__device__ __forceinline__ void f2_write_record(const FArgs& a, const int isl, const i64 t, const F2Red& r,
                                                const bool resample, const F2RecIn& in)
{
    double* info = a.info + (i64)isl * INFO_STRIDE;
    f2_write_row(a, isl, t - 1, r, in);
    a.summ[((i64)isl * (a.T + 1) + t) * SUMM_STRIDE + 4] = resample ? 1.0 : 0.0;
    info[0] = (double)t;
    info[1] = resample ? 1.0 : 0.0;
    info[2] = in.y;
    info[3] = r.K;
    info[4] = r.rs;
    info[5] = in.aux;
}
__device__ __forceinline__ void f2_write_record(const FArgs& a, const int isl, const i64 t, const F2Red& r,
                                                const bool resample)
{
    f2_write_record(a, isl, t, r, resample, f2_record_loads(a, isl, t));
}
// APF: the row of step t-1 comes from the PLAIN weights (evidence, logged ESS, W), the decision
// and the shares from the auxiliary ones; the constant the weights are reset to,
// log_mean_exp(logeta, W) = log(sum exp(lw + logeta) / sum exp(lw)), travels in slot 6
__device__ __forceinline__ void f2_write_record_apf(const FArgs& a, const int isl, const i64 t,
                                                    const F2Red& r_aux, const F2Red& r_plain,
                                                    const bool resample)
{
    double* info = a.info + (i64)isl * INFO_STRIDE;
    const F2RecIn in = f2_record_loads(a, isl, t);
    f2_write_row(a, isl, t - 1, r_plain, in);
    a.summ[((i64)isl * (a.T + 1) + t) * SUMM_STRIDE + 4] = resample ? 1.0 : 0.0;
    info[0] = (double)t;
    info[1] = resample ? 1.0 : 0.0;
    info[2] = in.y;
    info[3] = r_aux.K;
    info[4] = r_aux.rs;
    info[5] = in.aux;
    info[6] = (r_aux.K - r_plain.K) * 6.93147180559945286227e-01 + log(r_aux.s / r_plain.s);
}
// The island's reduction by ONE workgroup for any number of tiles, in chunks of 1024 partials
// (thread tid: the four from 4*tid on of every chunk).  Up to 1024 tiles that is operation for
// operation what every workgroup of k_ancestors2 does, hence the same bits.
__device__ __forceinline__ F2Red f2_reduce_island(const FArgs& a, const int isl, double* smd,
                                                  const bool plain = false)
{
    // plain: the APF's partials of the plain weights (pm2 / ps2 / pss2) instead of the auxiliary ones
    const double* Pm = plain ? a.pm2 : a.pm;
    const double* Ps = plain ? a.ps2 : a.ps;
    const double* Pss = plain ? a.pss2 : a.pss;
    const i64 o = (i64)isl * a.nparts;
    const bool pvec = (a.nparts & 3) == 0;
    const int nchunks = (a.nparts + 4 * SMC_BLOCK - 1) / (4 * SMC_BLOCK);
    F2Red r;
    double tm = -INFINITY;
    for (int c = 0; c < nchunks; ++c) {
        double pm4[4];
        f_load4<double>(Pm + o, (i64)c * 4 * SMC_BLOCK + (i64)threadIdx.x * 4, a.nparts, pvec, -INFINITY, pm4);
        tm = smc_max2(tm, smc_max2(smc_max2(pm4[0], pm4[1]), smc_max2(pm4[2], pm4[3])));
    }
    r.K = smc_block_max(tm, smd);
    double s1 = 0.0, s2 = 0.0;
    for (int c = 0; c < nchunks; ++c) {
        const i64 i0 = (i64)c * 4 * SMC_BLOCK + (i64)threadIdx.x * 4;
        double pm4[4], ps4[4], pss4[4];
        f_load4<double>(Pm + o, i0, a.nparts, pvec, -INFINITY, pm4);
        f_load4<double>(Ps + o, i0, a.nparts, pvec, 0.0, ps4);
        f_load4<double>(Pss + o, i0, a.nparts, pvec, 0.0, pss4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double v, w;
            f2_rescale(pm4[k], r.K, ps4[k], pss4[k], v, w);
            s1 = s1 + v;
            s2 = s2 + w;
        }
    }
    smc_block_sum2(s1, s2, smd);
    r.s = s1;
    r.ss = s2;
    f2_finish(a, r);
    return r;
}

// The same reduction with the partials of up to NC chunks (4096 tiles at NC = 4) held in registers:
// every load of the island is in flight at once and the three passes (max, sums, shares) read
// registers -- the same operations in the same order, hence the same bits, at one memory latency
// instead of one per chunk and pass (k_reduce2 is ONE workgroup on the critical path of the step).
template <int NC>
__device__ __forceinline__ F2Red f2_reduce_island_cached(const FArgs& a, const int isl, double* smd,
                                                         double (&pm)[NC][4], double (&ps)[NC][4],
                                                         const bool plain = false)
{
    const double* Pm = plain ? a.pm2 : a.pm;
    const double* Ps = plain ? a.ps2 : a.ps;
    const double* Pss = plain ? a.pss2 : a.pss;
    const i64 o = (i64)isl * a.nparts;
    const bool pvec = (a.nparts & 3) == 0;
    const int nchunks = (a.nparts + 4 * SMC_BLOCK - 1) / (4 * SMC_BLOCK);
    double pss[NC][4];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const i64 i0 = (i64)c * 4 * SMC_BLOCK + (i64)threadIdx.x * 4;
        if (c < nchunks) {
            f_load4<double>(Pm + o, i0, a.nparts, pvec, -INFINITY, pm[c]);
            f_load4<double>(Ps + o, i0, a.nparts, pvec, 0.0, ps[c]);
            f_load4<double>(Pss + o, i0, a.nparts, pvec, 0.0, pss[c]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) { pm[c][k] = -INFINITY; ps[c][k] = 0.0; pss[c][k] = 0.0; }
        }
    }
    F2Red r;
    double tm = -INFINITY;
#pragma unroll
    for (int c = 0; c < NC; ++c)
        if (c < nchunks)
            tm = smc_max2(tm, smc_max2(smc_max2(pm[c][0], pm[c][1]), smc_max2(pm[c][2], pm[c][3])));
    r.K = smc_block_max(tm, smd);
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c)
        if (c < nchunks) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double v, w;
                f2_rescale(pm[c][k], r.K, ps[c][k], pss[c][k], v, w);
                s1 = s1 + v;
                s2 = s2 + w;
            }
        }
    smc_block_sum2(s1, s2, smd);
    r.s = s1;
    r.ss = s2;
    f2_finish(a, r);
    return r;
}

// k_reduce2(t): one workgroup per island reduces the partials of step t-1, decides step t,
// writes the record, the summary row and every tile's (G_b, Q_b) (integer-valued doubles: shares of 2^52; a strict
// filter reads G_b only as its estimate of the sum in front of tile b and gets fractions of 1, relatively accurate)
// (a function of the island: k_reduce2 is one launch of it per step; the one-pass spacings kernel of the
//  multinomial scheme runs it as ITS workgroup 0, side by side with the workgroups that draw.)
// Returns -1: no step to run (t = 0, or the filter is past T / frozen), 0: step t does not resample, 1: it does.
// dec_word (the one-pass spacings kernel's workgroup 0): the decision is PUBLISHED -- (dec_epoch << 2) | (2 resample, 1 not) -- the
// moment it is known, before the tiles' shares are formed and written: the launch's other workgroups wait for it
// (nothing else of this function's results is theirs to read), the shares are the next launch's.
__device__ __forceinline__ int f2_reduce2_island(const FArgs& a, const int isl, double* smd, double* sme,
                                                 u64* dec_word, const u64 dec_epoch)
{
    const int tid = (int)threadIdx.x;
    double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)smc_uniform(smc_ldg(a.info2 + (i64)isl * INFO_STRIDE));
    if (t >= a.T) {
        if (tid == 0) info[0] = (double)t;
        return -1;
    }
    if (t == 0) return -1;
    F2RecIn rin = {0.0, 0.0, 0.0, 0.0, 0.0};                   // (what the record's writer reads: on its way during the reduction)
    if (tid == 0 && !a.pm2) rin = f2_record_loads(a, isl, t);
    const i64 o = (i64)isl * a.nparts;
    const bool pvec = (a.nparts & 3) == 0;
    const int nchunks = (a.nparts + 4 * SMC_BLOCK - 1) / (4 * SMC_BLOCK);
    double* G = reinterpret_cast<double*>(a.Qpre) + (i64)isl * a.ntiles;
    double* Q = reinterpret_cast<double*>(a.Q) + (i64)isl * a.ntiles;
    constexpr int NC = 4;
    if (nchunks <= NC) {
        double pmc[NC][4], psc[NC][4];
        const F2Red r = f2_reduce_island_cached<NC>(a, isl, smd, pmc, psc);
        const bool resample = r.ess < a.ess_thresh;
        if (dec_word && tid == 0) smc_st_agent(dec_word, (dec_epoch << 2) | (resample ? 2ull : 1ull));
        if (a.pm2) {
            const F2Red r2 = f2_reduce_island(a, isl, sme, true);
            if (tid == 0) f2_write_record_apf(a, isl, t, r, r2, resample);
        } else if (tid == 0) {
            f2_write_record(a, isl, t, r, resample, rin);
        }
        if (!resample) return 0;
        // every tile's share and the shares before it.  The chunks' four scans share ONE exchange (wave scans of
        // the four per-thread sums side by side, 16 wave totals through LDS, two barriers in all instead of eight):
        // the shares are integers below 2^53, their sums exact in any order
        __shared__ double s_x4[NC * SMC_NWAVE];
        double Q4[NC][4], run[NC], inc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const i64 i0 = (i64)c * 4 * SMC_BLOCK + (i64)tid * 4;
            run[c] = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double v, w;
                f2_rescale(pmc[c][k], r.K, psc[c][k], 0.0, v, w);
                Q4[c][k] = (c < nchunks && i0 + k < a.nparts) ? (a.strict_e ? v * r.rs : f2_share(v, r.rs)) : 0.0;
                run[c] += Q4[c][k];
            }
            inc[c] = smc_wave_scan_add_f64(run[c]);
        }
        // (the sums in front of a thread: its left neighbour's inclusive ones, nothing subtracted -- the same exact
        //  integers for the shares; for a strict filter, whose G_b is the estimate of the normalised sum in front of
        //  tile b as a fraction, inc - run would cancel in front of a heavy tile and lose the RELATIVE accuracy the
        //  classification's margin assumes)
        double exl[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) exl[c] = smc_dpp_f64<SMC_DPP_WAVE_SHR1, 0xf, false>(inc[c]);
        __syncthreads();
        if (smc_lane() == 63) {
#pragma unroll
            for (int c = 0; c < NC; ++c) s_x4[c * SMC_NWAVE + smc_wave()] = inc[c];
        }
        __syncthreads();
        double carry = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c >= nchunks) break;
            const i64 i0 = (i64)c * 4 * SMC_BLOCK + (i64)tid * 4;
            double base = 0.0, tot = 0.0;
#pragma unroll
            for (int w = 0; w < SMC_NWAVE; ++w) {
                if (w < smc_wave()) base += s_x4[c * SMC_NWAVE + w];
                tot += s_x4[c * SMC_NWAVE + w];
            }
            double g = carry + (base + exl[c]);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + k < a.nparts) { G[i0 + k] = g; Q[i0 + k] = Q4[c][k]; g += Q4[c][k]; }
            carry += tot;
        }
        return 1;
    }
    const F2Red r = f2_reduce_island(a, isl, smd);
    const bool resample = r.ess < a.ess_thresh;
    if (dec_word && tid == 0) smc_st_agent(dec_word, (dec_epoch << 2) | (resample ? 2ull : 1ull));
    if (a.pm2) {
        const F2Red r2 = f2_reduce_island(a, isl, sme, true);
        if (tid == 0) f2_write_record_apf(a, isl, t, r, r2, resample);
    } else if (tid == 0) {
        f2_write_record(a, isl, t, r, resample, rin);
    }
    if (!resample) return 0;
    // every tile's share and the shares before it, chunk by chunk (a running carry across chunks)
    double carry = 0.0;
    for (int c = 0; c < nchunks; ++c) {
        const i64 i0 = (i64)c * 4 * SMC_BLOCK + (i64)tid * 4;
        double pm4[4], ps4[4];
        f_load4<double>(a.pm + o, i0, a.nparts, pvec, -INFINITY, pm4);
        f_load4<double>(a.ps + o, i0, a.nparts, pvec, 0.0, ps4);
        double Q4[4], run = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double v, w;
            f2_rescale(pm4[k], r.K, ps4[k], 0.0, v, w);
            Q4[k] = (i0 + k < a.nparts) ? (a.strict_e ? v * r.rs : f2_share(v, r.rs)) : 0.0;
            run += Q4[k];
        }
        double tot;
        double g = carry + smc_block_exscan_pos_f64(run, sme, tot);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i0 + k < a.nparts) { G[i0 + k] = g; Q[i0 + k] = Q4[k]; g += Q4[k]; }
        carry += tot;
    }
    return 1;
}
// k_reduce2 for islands of 1025 .. 4096 tiles (C3's N = 2^22) as ONE workgroup of 1024 threads: thread (c, i) = chunk c
// of 1024 tiles, slot i of 256 -- what thread i of k_reduce2 does for its chunk c.  The loads, the power-of-two
// rescalings and the shares of the 16 tiles a slot owns are done by four threads side by side (sixteen waves hide
// each other's latencies where four could not: the kernel is one workgroup on the critical path of every step); the
// slot's two sums are then formed by thread (0, i) from LDS in k_reduce2's order -- chunk by chunk, tile by tile --
// and reduced over the 256 slots by waves 0 .. 3 exactly as smc_block_sum2 does: the same operations in the same
// order, hence the same bits (tests: SMC_TWO_LEVEL_MID against the resident step, C3's runs against the oracle).
__global__ void __launch_bounds__(4 * SMC_BLOCK)
k_reduce2w(const FArgs av)
{
    const FArgs& a = av;
    constexpr int NC = 4;
    __shared__ double s_v[NC][4][SMC_BLOCK];                 // [chunk][k][slot]: conflict-free for the slot's owner
    __shared__ double s_w[NC][4][SMC_BLOCK];
    __shared__ double s_m[NC * SMC_NWAVE];
    __shared__ double s_s[2 * SMC_NWAVE];
    __shared__ double s_x4[NC * SMC_NWAVE];
    const int isl = (int)blockIdx.x, tid = (int)threadIdx.x;
    const int c = tid >> 8, i = tid & (SMC_BLOCK - 1), wv = i >> 6, lane = tid & 63;
    double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)smc_uniform(smc_ldg(a.info2 + (i64)isl * INFO_STRIDE));
    if (t >= a.T) {
        if (tid == 0) info[0] = (double)t;
        return;
    }
    if (t == 0) return;
    F2RecIn rin = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (tid == 0) rin = f2_record_loads(a, isl, t);
    const i64 o = (i64)isl * a.nparts;
    const bool pvec = (a.nparts & 3) == 0;
    const int nchunks = (a.nparts + 4 * SMC_BLOCK - 1) / (4 * SMC_BLOCK);
    const i64 i0 = (i64)c * 4 * SMC_BLOCK + (i64)i * 4;
    double pm[4], ps[4], pss[4];
    if (c < nchunks) {
        f_load4<double>(a.pm + o, i0, a.nparts, pvec, -INFINITY, pm);
        f_load4<double>(a.ps + o, i0, a.nparts, pvec, 0.0, ps);
        f_load4<double>(a.pss + o, i0, a.nparts, pvec, 0.0, pss);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { pm[k] = -INFINITY; ps[k] = 0.0; pss[k] = 0.0; }
    }
    F2Red r;
    {
        double tm = smc_max2(smc_max2(pm[0], pm[1]), smc_max2(pm[2], pm[3]));
        tm = smc_wave_max(tm);
        if (lane == 0) s_m[tid >> 6] = tm;
        __syncthreads();
        r.K = s_m[0];
#pragma unroll
        for (int w = 1; w < NC * SMC_NWAVE; ++w) r.K = smc_max2(r.K, s_m[w]);
    }
    double v4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double w;
        f2_rescale(pm[k], r.K, ps[k], pss[k], v4[k], w);
        s_v[c][k][i] = v4[k];
        s_w[c][k][i] = w;
    }
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    if (c == 0) {
        for (int cc = 0; cc < nchunks; ++cc)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s1 = s1 + s_v[cc][k][i];
                s2 = s2 + s_w[cc][k][i];
            }
    }
    s1 = smc_wave_sum(s1);
    s2 = smc_wave_sum(s2);
    if (c == 0 && lane == 0) { s_s[wv] = s1; s_s[SMC_NWAVE + wv] = s2; }
    __syncthreads();
    {
        double ra = s_s[0], rb = s_s[SMC_NWAVE];
#pragma unroll
        for (int w = 1; w < SMC_NWAVE; ++w) { ra = ra + s_s[w]; rb = rb + s_s[SMC_NWAVE + w]; }
        r.s = ra;
        r.ss = rb;
    }
    f2_finish(a, r);
    const bool resample = r.ess < a.ess_thresh;
    if (tid == 0) f2_write_record(a, isl, t, r, resample, rin);
    if (!resample) return;
    // every tile's share and the shares before it (f2_reduce2_island's expressions, one chunk per group of four waves)
    double* G = reinterpret_cast<double*>(a.Qpre) + (i64)isl * a.ntiles;
    double* Q = reinterpret_cast<double*>(a.Q) + (i64)isl * a.ntiles;
    double Q4[4], run = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        Q4[k] = (c < nchunks && i0 + k < a.nparts) ? (a.strict_e ? v4[k] * r.rs : f2_share(v4[k], r.rs)) : 0.0;
        run += Q4[k];
    }
    const double inc = smc_wave_scan_add_f64(run);
    const double exl = smc_dpp_f64<SMC_DPP_WAVE_SHR1, 0xf, false>(inc);
    if (lane == 63) s_x4[c * SMC_NWAVE + wv] = inc;
    __syncthreads();
    if (c >= nchunks) return;
    double carry = 0.0;
    for (int cc = 0; cc < c; ++cc) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < SMC_NWAVE; ++w) tot += s_x4[cc * SMC_NWAVE + w];
        carry += tot;
    }
    double base = 0.0;
#pragma unroll
    for (int w = 0; w < SMC_NWAVE; ++w)
        if (w < wv) base += s_x4[c * SMC_NWAVE + w];
    double g = carry + (base + exl);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < a.nparts) { G[i0 + k] = g; Q[i0 + k] = Q4[k]; g += Q4[k]; }
}
__global__ void __launch_bounds__(SMC_BLOCK)
k_reduce2(const FArgs av)
{
    __shared__ double smd[SMC_SM];
    __shared__ double sme[SMC_SM];
    (void)f2_reduce2_island(av, (int)blockIdx.x, smd, sme);
}


// ---------------------------------------------------------------------------
// k_ancestors2(t): one workgroup per tile of 1024 parents.  No exp, no division per particle:
//   * loads the tile's stored integer CDF c_j (k_propagate wrote it) and, unless k_reduce2 ran
//     (MID), the island's partials: K, s, ss, ESS, decision, (G_b, Q_b) -- every exchange has
//     its own LDS slots and costs ONE barrier;
//   * systematic: the first offspring of a parent is floor(Y - u) + 1 with the parent's position
//     Y = (G_b + c Q_b / t_b) 2^-(52-k) in offspring units evaluated in fp64, the exact integer
//     route only inside the error band (f2_ns_sys); the tile's range [n_lo, n_hi) from the same
//     function evaluated by every thread (uniform values) instead of an LDS exchange;
//   * ONE scatter pass over a window of 2048 offspring (8 KB of LDS, zeroed while the loads are
//     in flight): a tile owns 1024 +- a few dozen offspring.
// MID: k_reduce2 ran first: grids too large for every workgroup to repeat the reduction.
// ---------------------------------------------------------------------------
// REGEN (MID, MULTI, Philox mode): the tile's window of the sorted uniforms is found through the tile
// prefixes of the spacings (one probe, see F2Regen above) and staged whole; tapes keep the search.
// SQ (SMC_FLAG_SQMC; MID, MULTI, N = 2^k): the sorted uniforms are the sorted first coordinates of the step's
// Sobol' points -- a regular grid (the digital shift's low bits, rqmc's safe_generate map): threshold n is a
// function of n, counts are a guess fixed with the definition, nothing is read.
// SCH (closed-form counts): SMC_STRATIFIED_ / SMC_SYSTEMATIC_ as a compile-time constant -- the kernel of one scheme
// carries none of the other's code (the stratified draw inlines a dozen Philox calls); 0: a.scheme at run time.
#ifndef SMC_A2M_MINW
#define SMC_A2M_MINW 8      /* REGEN form: waves per SIMD the register allocation leaves room for (64 registers, 64 B of
                               scratch; with 14.6 KB of LDS eight workgroups share a CU: 74.0 -> 72.0 us per step at C3,
                               profiles/r15_c3_multinomial_ab.txt; A/B builds: -DSMC_A2M_MINW=1) */
#endif
template <bool MID, bool MULTI = false, bool POW2 = true, bool REGEN = false, bool SQ = false, int SCH = 0>
__global__ void __launch_bounds__(SMC_BLOCK, REGEN ? SMC_A2M_MINW : 1)
k_ancestors2(const FArgs av)
{
    static_assert(!REGEN || (MID && MULTI), "regenerated thresholds: multinomial behind k_reduce2");
    static_assert(!SQ || (MID && MULTI && POW2 && !REGEN), "SQMC: the tape form of the multinomial search");
    const FArgs& a = av;
    constexpr int WIN = 2 * F_PASS;                                        // offspring per pass
    // the scatter window sP (8 KB), and -- REGEN -- the staged thresholds sT (12 KB) IN THE SAME MEMORY: sT is dead
    // when the counts are done, sP is zeroed then (behind the barrier the counts end with) -- 14.6 instead of 22.8 KB of
    // LDS per workgroup, 8 workgroups per CU instead of 7: N = 2^22's 4096 tiles are two full rounds of the chip
    constexpr int WMAX = 1536;                                             // staged thresholds
    __shared__ __attribute__((aligned(16))) double s_raw[REGEN ? WMAX + 4 : WIN / 2];
    u32* const sP = reinterpret_cast<u32*>(s_raw);
    __shared__ double s_max[SMC_NWAVE];                                    // one area per exchange
    __shared__ double s_sum[2 * SMC_NWAVE];
    __shared__ double s_g[SMC_NWAVE + 1];
    __shared__ u32 s_mx[2 * SMC_NWAVE];
    // SQ: the tile's parents are SORTED positions; what is stored is the particle at that position, h_order[j0 + j] --
    // one sequential read of the tile's 1024 entries, an LDS look-up per offspring (round 5: a launch of its own,
    // k_sq_compose, 5 us + a seam per step: a random 8-byte gather per offspring)
    __shared__ u32 s_pm[SQ ? F_TILE : 4];
    const int b = f_tile_xcd(av, (int)blockIdx.x), isl = (int)blockIdx.y;
    const int tid = (int)threadIdx.x;
    const int lane = smc_lane(), wave = smc_wave();
    const i64 N = a.N;
    const i64 j0 = (i64)b * F_TILE;
    const i64 jt = j0 + (i64)tid * F_IPT;
    F_STAMP_A(0);
    u64 pq[4] = {0ull, 0ull, 0ull, 0ull};
    if (SQ) {
        const u64* pp = a.sq_perm + (i64)isl * N + jt;
        smc_ld2g(pp, pq[0], pq[1]);
        smc_ld2g(pp + 2, pq[2], pq[3]);
    }
    double* info = a.info + (i64)isl * INFO_STRIDE;
    const double r0 = smc_ldg(MID ? info : a.info2 + (i64)isl * INFO_STRIDE);
    const double r1 = MID ? smc_ldg(info + 1) : 0.0;
    const i64 o = (i64)isl * a.nparts;
    double pm4[4], ps4[4], pss4[4];
    double Gmid = 0.0, Qmid = 0.0;
    if (MID) {
        Gmid = smc_ldg(reinterpret_cast<const double*>(a.Qpre) + (i64)isl * a.ntiles + b);
        Qmid = smc_ldg(reinterpret_cast<const double*>(a.Q) + (i64)isl * a.ntiles + b);
    }
    // the tile's integer CDF: this thread's 4 positions and the next thread's first
    const u64* cq = a.cq + (i64)isl * ((POW2 && !MULTI) ? N : a.ncq);     // (N = 2^k: whole tiles, ncq == N)
    u64 cx[F_IPT + 1];
    smc_ld2g(cq + jt, cx[0], cx[1]);
    smc_ld2g(cq + jt + 2, cx[2], cx[3]);
    cx[4] = (tid < SMC_BLOCK - 1) ? smc_ldg(cq + jt + 4) : 0ull;
    const u64 tb_raw = smc_ldg(a.tq + o + b);
    // (REGEN) the spacings' total and the tile prefixes around this tile, requested with the first loads
    u64 spec_all = 0ull, spec_E = 0ull;
    i64 spec_w0 = 0;
    if (REGEN) {
        const u64* Eb = a.E + (i64)isl * (a.ntiles1 + 1);
        const i64 M1 = (i64)a.ntiles1 + 1;
        spec_all = smc_ldg(Eb + a.ntiles1);
        spec_w0 = (i64)b - 32;
        spec_w0 = spec_w0 > M1 - 64 ? M1 - 64 : spec_w0;
        spec_w0 = spec_w0 < 0 ? 0 : spec_w0;
        const i64 ks = spec_w0 + lane;
        spec_E = smc_ldg(Eb + (ks < M1 ? ks : M1 - 1));
    }
    if (!MID) {
        const bool pvec = (a.nparts & 3) == 0;
        f_load4<double>(a.pm + o, (i64)tid * 4, a.nparts, pvec, -INFINITY, pm4);
        f_load4<double>(a.ps + o, (i64)tid * 4, a.nparts, pvec, 0.0, ps4);
        f_load4<double>(a.pss + o, (i64)tid * 4, a.nparts, pvec, 0.0, pss4);
    }
    // the scatter window of the first pass, while the loads are on their way (REGEN: see s_raw)
    if (!REGEN) {
        *reinterpret_cast<uint4*>(&sP[tid * 4]) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(&sP[F_PASS + tid * 4]) = make_uint4(0u, 0u, 0u, 0u);
    }
    const i64 t = (i64)smc_uniform(r0);
    if (t >= a.T) {
        if (!MID && b == 0 && tid == 0) info[0] = (double)t;   // k_propagate returns on it
        return;
    }
    if (t == 0) return;                                        // the host wrote the record of step 0
    F2RecIn rin = {0.0, 0.0, 0.0, 0.0, 0.0};                   // (what the record's writer reads: on its way during the reduction)
    if (!MID && b == 0 && tid == 0) rin = f2_record_loads(a, isl, t);
    if (MID && smc_uniform(r1) == 0.0) return;                 // k_reduce2: step t does not resample
    F_STAMP_A(1);
    SmcSu su;                                                  // (the step's uniform: one Philox call,
    u64 Us;                                                    //  all inputs uniform: scalar unit)
    f2_su(a, isl, t, su, Us, SCH);
    double Gd, Qd;
    if (MID) {
        Gd = smc_uniform(Gmid);
        Qd = smc_uniform(Qmid);
        if (SQ) *reinterpret_cast<uint4*>(&s_pm[tid * 4]) = make_uint4((u32)pq[0], (u32)pq[1], (u32)pq[2], (u32)pq[3]);
        __syncthreads();                                       // sP zeroed (SQ: s_pm staged)
    } else {
        // ---- all partials -> K, (s, ss), ESS, the decision (f2_reduce_island's operations)
        // (measured, r05p: ONE exchange -- every wave summing on its own maximum, the four wave results put on K
        //  afterwards, the same bits because power-of-two scalings commute with the roundings -- is 0.12 us
        //  SLOWER per step: the extra ldexp pairs cost more than the barrier they save)
        double tm = smc_max2(smc_max2(pm4[0], pm4[1]), smc_max2(pm4[2], pm4[3]));
        tm = smc_wave_max(tm);
        if (lane == 0) s_max[wave] = tm;
        __syncthreads();                                       // (1) also: sP zeroed
        F2Red r;
        r.K = s_max[0];
#pragma unroll
        for (int w = 1; w < SMC_NWAVE; ++w) r.K = smc_max2(r.K, s_max[w]);
        double v4[4], s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double w;
            f2_rescale(pm4[k], r.K, ps4[k], pss4[k], v4[k], w);
            s1 = s1 + v4[k];
            s2 = s2 + w;
        }
        s1 = smc_wave_sum(s1);
        s2 = smc_wave_sum(s2);
        if (lane == 0) { s_sum[wave] = s1; s_sum[SMC_NWAVE + wave] = s2; }
        __syncthreads();                                       // (2)
        s1 = s_sum[0];
        s2 = s_sum[SMC_NWAVE];
#pragma unroll
        for (int w = 1; w < SMC_NWAVE; ++w) { s1 = s1 + s_sum[w]; s2 = s2 + s_sum[SMC_NWAVE + w]; }
        r.s = s1;
        r.ss = s2;
        f2_finish(a, r);
        const bool resample = r.ess < a.ess_thresh;            // core.py:181-183 (t < T here)
        if (b == 0 && tid == 0) f2_write_record(a, isl, t, r, resample, rin);
        F_STAMP_A(3);
        if (!resample) return;
        // ---- this tile's share Q_b of the 2^52 scale and the shares before it, G_b
        double qbefore = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid * 4 + k;
            const double Qk = (i < a.nparts) ? f2_share(v4[k], r.rs) : 0.0;
            qbefore += (i < b) ? Qk : 0.0;
            if (i == b) s_g[SMC_NWAVE] = Qk;
        }
        qbefore = smc_wave_sum(qbefore);                       // (integers below 2^53: exact)
        if (lane == 0) s_g[wave] = qbefore;
        __syncthreads();                                       // (3)
        Gd = s_g[0];
#pragma unroll
        for (int w = 1; w < SMC_NWAVE; ++w) Gd = Gd + s_g[w];
        Qd = s_g[SMC_NWAVE];
    }
    const u64 tb = smc_uniform_u64(tb_raw);
    if (tid == SMC_BLOCK - 1) cx[4] = tb;
    F_STAMP_A(4);
    // ---- first offspring of each parent; the tile's range [n_lo, n_hi)
    const u64 Gb = (u64)Gd, Qb = (u64)Qd;
    i64 ns[F_IPT + 1], n_lo, n_hi;
    if (REGEN) {
        // thresholds as integer-valued doubles (<= 2^52: exact, compared in fp64; no 64-bit conversions)
        constexpr int MARGIN = 128;                                        // > 8 sigma of a position inside a tile
        double* const sT = s_raw;                                          // (WMAX + 4 guard slots)
        __shared__ i64 s_k[2];
        __shared__ double s_b[4];
        __shared__ i64 s_edge[SMC_BLOCK + 1];
        F2Regen g;
        g.E = a.E + (i64)isl * (a.ntiles1 + 1);
        g.ntiles1 = a.ntiles1;
        g.dall = (double)smc_uniform_u64(spec_all);
        g.rdall = 1.0 / g.dall;
        // ---- the tiles of draws the tile's two ends fall into (one wave per end, one probe each) and,
        // by interpolation between the tile's first and last threshold, where in them
        const double per52 = (double)(N + 1) * 0x1.0p-62;                  // tiles of draws per unit of the scale
        if (wave < 2) {
            const u64 C = wave == 0 ? Gb : Gb + Qb;
            u64 Bl, Bh;
            i64 v = f2_regen_tiles_le_spec(g, C, spec_w0, spec_E, Bl, Bh);
            if (v < 0) v = f2_regen_tiles_le_wave(g, C, (i64)((double)C * per52), Bl, Bh);
            v -= 1;
            if (lane == 0) { s_k[wave] = v; s_b[2 * wave] = (double)Bl; s_b[2 * wave + 1] = (double)(Bh == ~0ull ? Bl + 1ull : Bh); }
        }
        __syncthreads();
        F_STAMP_A(2);                                                      // (REGEN: tile prefixes probed)
        const i64 k_lo = s_k[0];
        i64 k_hi = s_k[1];
        k_hi = k_hi > (i64)a.ntiles1 - 1 ? (i64)a.ntiles1 - 1 : k_hi;       // (C >= 2^52: every draw counted)
        const i64 w_lo = k_lo * F_TILE;                                    // the tiles' draws: [w_lo, w_hi)
        i64 w_hi = (k_hi + 1) * F_TILE;
        w_hi = w_hi > N ? N : w_hi;
        const double Cl = (double)Gb, Ch_ = (double)(Gb + Qb);
        const double f_lo = (Cl - s_b[0]) / (s_b[1] - s_b[0]), f_hi = (Ch_ - s_b[2]) / (s_b[3] - s_b[2]);
        i64 s0 = w_lo + (i64)(f_lo * (double)F_TILE) - MARGIN;
        i64 s1 = k_hi * F_TILE + (i64)(f_hi * (double)F_TILE) + MARGIN;
        s0 = (s0 < w_lo ? w_lo : s0) & ~(i64)1;
        s1 = s1 > w_hi ? w_hi : s1;
        s1 = s1 < s0 ? s0 : s1;
        const int nw = (int)(s1 - s0);
        bool staged = nw <= WMAX;
        const bool zform = a.sp_tpw > 0;           // one-pass spacings: Z_n = E[n >> 10] + o_n (k_f_spacing_onepass), su_n = fl(Z_n) / fl(Z_N)
        const u32* zo = reinterpret_cast<const u32*>(a.su) + (i64)isl * N * 2;
        if (zform && b < a.sp_nwg && tid == 0) a.sst[(i64)isl * a.sp_nwg + b] = 0ull;     // look-back words: re-armed
        const i64 kw0 = smc_uniform_u64((u64)(s0 >> 10));
        u64 Ew[4] = {0ull, 0ull, 0ull, 0ull};
        if (zform && staged) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const i64 kk = kw0 + j < (i64)a.ntiles1 ? kw0 + j : (i64)a.ntiles1;
                Ew[j] = smc_ldg(g.E + kk);          // (requested with the window's words: nothing waits for them here)
            }
        }
        if (staged && zform) {
            // (WMAX = 3 x 512: at most three pairs per thread -- all of them, and the prefixes of the at most four
            //  tiles of draws the window touches, requested before anything is used.  Loads on clamped addresses,
            //  masked when they are written to LDS: no exec-mask region, hence no wait, between two requests.
            //  s0 is even and a tile of draws starts at a multiple of 1024: a pair lies inside ONE tile)
            constexpr int NPAIR = WMAX / (2 * SMC_BLOCK);
            u32 o0[NPAIR], o1[NPAIR];
            int kt[NPAIR];
            const i64 last_pair = (N - 1) & ~(i64)1;       // (N odd: the pair (N - 1, N) -- the island's slot has the room)
#pragma unroll
            for (int r = 0; r < NPAIR; ++r) {
                i64 idx = s0 + tid * 2 + r * 2 * SMC_BLOCK;
                idx = idx < last_pair ? idx : last_pair;
                smc_ld2g(zo + idx, o0[r], o1[r]);
                kt[r] = (int)((idx >> 10) - kw0);
            }
#pragma unroll
            for (int r = 0; r < NPAIR; ++r) {
                const int i = tid * 2 + r * 2 * SMC_BLOCK;
                const u64 eb = kt[r] <= 0 ? Ew[0] : (kt[r] == 1 ? Ew[1] : (kt[r] == 2 ? Ew[2] : Ew[3]));
                const double u0 = smc_div_c((double)(eb + (u64)o0[r]), g.dall, g.rdall);
                const double u1 = smc_div_c((double)(eb + (u64)o1[r]), g.dall, g.rdall);
                if (i < nw) sT[i] = ceil(u0 * 4503599627370496.0);
                if (i + 1 < nw) sT[i + 1] = ceil(u1 * 4503599627370496.0);
            }
        } else if (staged) {
            constexpr int NPAIR = WMAX / (2 * SMC_BLOCK);
            double w0[NPAIR], w1[NPAIR];
            const bool even = (N & 1) == 0;
#pragma unroll
            for (int r = 0; r < NPAIR; ++r) {
                const int i = tid * 2 + r * 2 * SMC_BLOCK;
                w0[r] = 2.0;
                w1[r] = 2.0;
                if (i + 1 < nw && even) smc_ld2g(su.u + s0 + i, w0[r], w1[r]);
                else if (i < nw) {
                    w0[r] = smc_ldg(su.u + s0 + i);
                    if (i + 1 < nw) w1[r] = smc_ldg(su.u + s0 + i + 1);
                }
            }
#pragma unroll
            for (int r = 0; r < NPAIR; ++r) {
                const int i = tid * 2 + r * 2 * SMC_BLOCK;
                if (i < nw) sT[i] = ceil(w0[r] * 4503599627370496.0);
                if (i + 1 < nw) sT[i + 1] = ceil(w1[r] * 4503599627370496.0);
            }
        }
        if (staged) {
            __syncthreads();
            // the window must hold every threshold in (G_b, G_b + Q_b]: the one before it at or below
            // G_b (or the window starts with its tile), the one after it above G_b + Q_b (or it ends
            // with its tile) -- else (skewed spacings: never seen) the slow path
            const bool ok_lo = s0 == w_lo || (nw > 0 && sT[0] <= Cl);
            const bool ok_hi = s1 == w_hi || (nw > 0 && sT[nw - 1] > Ch_);
            staged = ok_lo && ok_hi;
        }
        // ---- every boundary: position on the scale as G_b + floor(c Q_b / t_b) with the quotient in fp64
        // (within 2^12 of the exact one; all values integers below 2^53: exact doubles), the count decided
        // unless a threshold lies within 2^13 of it -- then the exact 128-bit quotient.  A thread's boundaries
        // are consecutive and a parent owns about one offspring: ONE bisection for the first, then each
        // count is the previous one plus the number of the next 4 thresholds at or below the new position
        // (all lanes do the same 4 comparisons; a parent with more than 3 offspring: 2 % of them: bisects).
        const double qscale = (double)Qb / (double)(tb ? tb : 1ull);
        const double BAND = 8192.0, Gbd = (double)Gb, Qbd = (double)Qb;
        if (staged) {
            // guards: sT[-1] below everything this tile compares with (see ok_lo), 4 slots of +inf at the end
            for (int i = tid; i < 4; i += SMC_BLOCK) sT[nw + i] = INFINITY;
            __syncthreads();
            F_STAMP_A(3);                                                  // (REGEN: window staged)
            int kprev = 0;
            bool have_prev = false;
#pragma unroll
            for (int i = 0; i <= F_IPT; ++i) {
                const i64 j = jt + i;
                const u64 c = cx[i];
                // (the 5th boundary is the next thread's first -- taken from it below -- except the last
                //  thread's: the tile's upper end, c = t_b)
                const bool open = j > 0 && j < N && (i < F_IPT || tid == SMC_BLOCK - 1);
                ns[i] = (j == 0) ? 0 : N;
                if (!open) continue;
                const bool end = c == 0ull || c >= tb;                    // the tile's own ends: exact integers
                bool exact = end || a.exact_counts != 0;
                double Cd;
                if (end) Cd = c == 0ull ? Gbd : Gbd + Qbd;
                else if (exact) Cd = (double)(Gb + smc_muldiv_floor(c, Qb, tb));
                else Cd = Gbd + fmin(floor((double)c * qscale), Qbd);
                int k;
                if (have_prev) {
                    // (a parent with more than 3 offspring -- 2 % of them, but 3 waves in 4 hold one -- takes a
                    //  second and a third step of 4 probes before it bisects: a step is ONE LDS latency, the
                    //  bisection eleven; r05t: the counts went from 4.9 to ... us per workgroup)
                    k = kprev;
                    int adv = 4;
#pragma unroll
                    for (int s_ = 0; s_ < 3; ++s_) {
                        if (adv == 4) {
                            const double t0 = sT[k], t1 = sT[k + 1], t2 = sT[k + 2], t3 = sT[k + 3];
                            adv = (t0 <= Cd ? 1 : 0) + (t1 <= Cd ? 1 : 0) + (t2 <= Cd ? 1 : 0) + (t3 <= Cd ? 1 : 0);
                            k += adv;
                        }
                    }
                    if (adv == 4) k += f2_count_lds_f64(sT + k, nw - k, Cd);
                } else {
                    k = f2_count_lds_f64(sT, nw, Cd);
                }
                if (!exact) {
                    const bool below = k == 0 || sT[k - 1] + BAND <= Cd;   // (k == 0: see ok_lo)
                    const bool above = sT[k] > Cd + BAND;                  // (k == nw: the +inf guard)
                    if (!(below && above)) {
                        Cd = (double)(Gb + smc_muldiv_floor(c, Qb, tb));
                        k = f2_count_lds_f64(sT, nw, Cd);
                    }
                }
                kprev = k;
                have_prev = true;
                const i64 cnt = s0 + k;
                ns[i] = cnt < N ? cnt : N;
            }
        } else {
#pragma unroll 1
            for (int i = 0; i <= F_IPT; ++i) {
                const i64 j = jt + i;
                u64 c = cx[0];
#pragma unroll
                for (int q = 1; q <= F_IPT; ++q) c = (i == q) ? cx[q] : c;
                const bool open = j > 0 && j < N && (i < F_IPT || tid == SMC_BLOCK - 1);
                i64 v = (j == 0) ? 0 : N;
                if (open) {
                    const u64 pos = (c == 0ull) ? 0ull : (c >= tb ? Qb : smc_muldiv_floor(c, Qb, tb));
                    i64 lo_ = w_lo, hi_ = w_hi;                            // first threshold in the tiles above C
                    while (lo_ < hi_) {
                        const i64 mid = lo_ + ((hi_ - lo_) >> 1);
                        double um = zform ? smc_div_c((double)(smc_ldg(g.E + (mid >> 10)) + (u64)smc_ldg(zo + mid)), g.dall, g.rdall)
                                          : smc_ldg(su.u + mid);
                        if (f2_t52(um) <= Gb + pos) lo_ = mid + 1; else hi_ = mid;
                    }
                    v = lo_ < N ? lo_ : N;
                }
#pragma unroll
                for (int q = 0; q <= F_IPT; ++q) ns[q] = (i == q) ? v : ns[q];
            }
        }
        // ---- the next thread's first boundary; the tile's range
        __syncthreads();
        // (every count is made: the thresholds' memory becomes the scatter window of the first pass)
        *reinterpret_cast<uint4*>(&sP[tid * 4]) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(&sP[F_PASS + tid * 4]) = make_uint4(0u, 0u, 0u, 0u);
        s_edge[tid] = ns[0];
        if (tid == SMC_BLOCK - 1) s_edge[SMC_BLOCK] = ns[F_IPT];
        __syncthreads();
        if (tid < SMC_BLOCK - 1) ns[F_IPT] = s_edge[tid + 1];
        n_lo = s_edge[0];
        n_hi = s_edge[SMC_BLOCK];
    } else if (MULTI) {
        // ---- multinomial: the tile's range by two cooperative searches (one wave per end), its
        // thresholds staged in LDS, every parent's count a search there; the position of a parent
        // on the tile's share is formed exactly (128-bit product)
        __shared__ i64 s_nm[2];
        __shared__ u64 sT[F2_MW];
        F2Sq sq;
        if (SQ) f2_sq_init(a, isl, t, sq);
        if (wave == 0) {
            const i64 v = (b == 0) ? 0 : (SQ ? f2_sq_count(sq, Gb) : f2_count_sorted_wave(su.u, N, Gb));
            if (lane == 0) s_nm[0] = v;
        }
        if (wave == 1) {
            const i64 v = (b == a.ntiles - 1) ? N : (SQ ? f2_sq_count(sq, Gb + Qb) : f2_count_sorted_wave(su.u, N, Gb + Qb));
            if (lane == 0) s_nm[1] = v;
        }
        __syncthreads();
        n_lo = s_nm[0];
        n_hi = s_nm[1];
        const i64 width = n_hi - n_lo;
        const bool staged = width <= F2_MW;
        if (staged) {
            for (int i = tid; i < (int)width; i += SMC_BLOCK)
                sT[i] = SQ ? f2_sq_T(sq, n_lo + i) : f2_t52(smc_ldg(su.u + n_lo + i));
            __syncthreads();
        }
        // position of a parent on the tile's share: fp64 quotient within 2^12 of floor(c Q_b / t_b)
        // (f2_first_offspring); the count is decided unless a threshold lies within 2^13 of it --
        // then, and in tiles too wide to stage, the exact 128-bit quotient is formed
        const double qscale = (double)Qb / (double)(tb ? tb : 1ull);
        const u64 E = 1ull << 13;
#pragma unroll
        for (int i = 0; i <= F_IPT; ++i) {
            const i64 j = jt + i;
            const u64 c = cx[i];
            if (j == 0) ns[i] = 0;
            else if (j >= N) ns[i] = N;
            else {
                i64 cnt = -1;
                if (staged && !a.exact_counts) {
                    u64 qh = (u64)((double)c * qscale);
                    qh = qh > Qb ? Qb : qh;
                    const u64 Ch = Gb + qh;
                    const int k = f2_count_lds(sT, (int)width, Ch);
                    const bool below = k == 0 || sT[k - 1] + E <= Ch;
                    const bool above = k == (int)width || sT[k] > Ch + E;
                    if (below && above) cnt = n_lo + k;
                }
                if (cnt < 0) {
                    const u64 pos = (c == 0ull) ? 0ull : (c >= tb ? Qb : smc_muldiv_floor(c, Qb, tb));
                    cnt = staged ? n_lo + f2_count_lds(sT, (int)width, Gb + pos)
                                 : (SQ ? f2_sq_count(sq, Gb + pos) : f2_count_sorted_range(su.u, n_lo, n_hi, Gb + pos));
                }
                ns[i] = cnt;
            }
        }
    } else if ((SCH ? SCH : a.scheme) == SMC_SYSTEMATIC_) {
        F2Fast f;
        const double down = (double)N * 0x1.0p-52;        // offspring per unit of the 2^52 scale (2^-sh for N = 2^k)
        f.Gb = Gb; f.Qb = Qb; f.tb = tb;
        f.Gd = Gd * down;
        f.r = tb ? (Qd / (double)tb) * down : 0.0;
        f.u = su.u_sys;
        f.dN = (double)N;
        f.eps = a.exact_counts ? 2.0 : f.dN * 0x1.0p-49;                 // (2.0: always the exact route)
        f.one_m_eps = 1.0 - f.eps;
#pragma unroll
        for (int i = 0; i <= F_IPT; ++i) {
            const i64 j = jt + i;
            ns[i] = (j == 0) ? 0 : (j >= N ? N : f2_ns_sys<POW2>(a, su, Us, f, cx[i]));
        }
        if (POW2) {
            // the tile's range: its first thread's first count and its last thread's last (positions 0 and t_b),
            // handed round through LDS instead of evaluated by every thread (2 of its 7 counts)
            __shared__ i64 s_nsys[2];
            if (tid == 0) s_nsys[0] = ns[0];
            if (tid == SMC_BLOCK - 1) s_nsys[1] = ns[F_IPT];
            __syncthreads();
            n_lo = s_nsys[0];
            n_hi = s_nsys[1];
        } else {
            n_lo = (b == 0) ? 0 : f2_ns_sys<POW2>(a, su, Us, f, 0ull);
            n_hi = (b == a.ntiles - 1) ? N : f2_ns_sys<POW2>(a, su, Us, f, tb);
        }
        if (!POW2) {
            unsigned need = (n_lo < 0 ? 32u : 0u) | (n_hi < 0 ? 64u : 0u);
#pragma unroll
            for (int i = 0; i <= F_IPT; ++i) need |= (ns[i] < 0) ? (1u << i) : 0u;
            if (need) {
                const u64 c7[7] = {cx[0], cx[1], cx[2], cx[3], cx[4], 0ull, tb};
                i64 n7[7] = {ns[0], ns[1], ns[2], ns[3], ns[4], n_lo, n_hi};
                f2_resolve_general(su, need, Gb, Qb, tb, c7, n7);
#pragma unroll
                for (int i = 0; i <= F_IPT; ++i) ns[i] = n7[i];
                n_lo = n7[5];
                n_hi = n7[6];
            }
        }
    } else {
        if (POW2) {
            F2Fast f;
            const double down = (double)N * 0x1.0p-52;
            f.Gb = Gb; f.Qb = Qb; f.tb = tb;
            f.Gd = Gd * down;
            f.r = tb ? (Qd / (double)tb) * down : 0.0;
            f.u = 0.0;
            f.dN = (double)N;
            f.eps = a.exact_counts ? 2.0 : f.dN * 0x1.0p-49;
            f.one_m_eps = 1.0 - f.eps;
            // ---- the stratified uniforms of the tile's offspring, staged in LDS.  A parent at position Y of the
            // offspring scale needs u_n for n = floor(Y): generated per parent that is one Philox call per boundary
            // (5 per thread, each behind a divergent "same pair as before?" test -- 9.4 us of C3's 25.8 us launch
            // against the systematic scheme's 16.4).  The tile's boundaries all lie in [floor(G_b down), that +
            // Q_b down + 2]: the workgroup generates that window once -- F2_SU_PAIRS Philox calls for the tile,
            // 2 or 3 per thread, each giving the pair (u_2p, u_2p+1) -- and a boundary READS its uniform.  The same
            // counters, the same uniforms, the same counts.  Tiles whose window does not fit (a share beyond 1.24 x
            // the average) and the tape mode keep the per-boundary calls.
            __shared__ __attribute__((aligned(16))) double sU[2 * F2_SU_PAIRS];
            const double nfirst = floor(f.Gd * 0.5) * 2.0;                 // first staged offspring (even)
#ifdef SMC_NO_SU_STAGE                     /* (A/B builds: tools/build_ablations.sh) */
            const bool stage = false;
#else
            const bool stage = !su.u && (Qd * down + 4.0 <= (double)(2 * F2_SU_PAIRS - 2)) && nfirst < f.dN;
#endif
            if (stage) {
                const u32 p0 = (u32)(nfirst * 0.5);
#pragma unroll
                for (int r = 0; r < (F2_SU_PAIRS + SMC_BLOCK - 1) / SMC_BLOCK; ++r) {
                    const int q = tid + r * SMC_BLOCK;
                    if (q < F2_SU_PAIRS) {                                 // (whole waves: F2_SU_PAIRS % 64 == 0)
                        u64 xa, xb;
                        smc_philox(p0 + (u32)q, su.t, su.island, SMC_STREAM_RESAMPLE, su.seed, xa, xb);
                        double2 uu;
                        uu.x = smc_u01_halfopen(xa);
                        uu.y = smc_u01_halfopen(xb);
                        *reinterpret_cast<double2*>(&sU[2 * q]) = uu;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int i = 0; i <= F_IPT; ++i) {
                    const i64 j = jt + i;
                    ns[i] = (j == 0) ? 0 : (j >= N ? N : f2_ns_strat_lds(a, su, Us, f, cx[i], sU, nfirst));
                }
            } else {
#pragma unroll
                for (int i = 0; i <= F_IPT; ++i) {
                    const i64 j = jt + i;
                    ns[i] = (j == 0) ? 0 : (j >= N ? N : f2_ns_strat(a, su, Us, f, cx[i]));
                }
            }
        } else {
            f2_first_offspring<POW2>(a, su, Us, cx, tb, Gb, Qb, jt, ns);
        }
        __shared__ i64 s_n[2];
        if (tid == 0) s_n[0] = ns[0];
        if (tid == SMC_BLOCK - 1) s_n[1] = ns[F_IPT];
        __syncthreads();
        n_lo = s_n[0];
        n_hi = s_n[1];
    }
    F_STAMP_A(5);
    u32* A = f_A(a, t) + (i64)isl * N;
    // ---- heavy parents (>= 2048 offspring): registered, their whole blocks left to k_propagate
    __shared__ i64 sH[2 * F_HLOC];
    __shared__ unsigned sHn;
    int nH = 0;
    if (a.hcnt && (n_hi - n_lo >= 2 * (i64)F_TILE))               // (same in every thread)
        nH = f_register_heavy(a.hcnt + (i64)isl * 2 + (t & 1),
                              a.hlist + ((i64)isl * 2 + (t & 1)) * F_HMAX * 3, jt,
                              ns[0], ns[1], ns[2], ns[3], ns[4], sH, &sHn);
    // (32-bit arithmetic from here on: N <= 2^30 on this path, offspring indices fit)
    u32 nsu[F_IPT + 1];
#pragma unroll
    for (int i = 0; i <= F_IPT; ++i) nsu[i] = (u32)ns[i];
    const u32 lo = (u32)n_lo, hi = (u32)n_hi;
    bool first_pass = true;
    for (u32 pb = lo & ~3u; pb < hi; pb += WIN) {
        if (nH) {                                  // the whole pass inside a registered parent's blocks?
            const i64 w_lo = pb > lo ? pb : lo, w_hi = pb + WIN < hi ? pb + WIN : hi;
            i64 jump = 0;                          // passes to leave out, this one included
            for (int k = 0; k < nH; ++k)
                if (sH[2 * k] <= w_lo && w_hi <= sH[2 * k + 1]) {
                    const i64 whole = (sH[2 * k + 1] - (i64)pb) / WIN;      // passes that end inside the blocks
                    jump = whole > 1 ? whole : 1;
                }
            if (jump) { pb += (u32)(jump - 1) * WIN; continue; }
        }
        if (!first_pass) {
            __syncthreads();                       // previous pass has read sP
            *reinterpret_cast<uint4*>(&sP[tid * 4]) = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(&sP[F_PASS + tid * 4]) = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
        }
        first_pass = false;
        // every parent writes its index at its first offspring's slot of the window
        int rel[F_IPT + 1];
#pragma unroll
        for (int i = 0; i <= F_IPT; ++i) {
            const int d = (int)(nsu[i] - pb);
            rel[i] = d < 0 ? 0 : (d > WIN ? WIN : d);
        }
#pragma unroll
        for (int i = 0; i < F_IPT; ++i)
            if (rel[i] < rel[i + 1]) sP[rel[i]] = (u32)(tid * F_IPT + i);
        const bool two = hi > pb + F_PASS;         // does the second half of the window hold offspring?
        __syncthreads();
        // a running maximum over the slots gives each offspring its parent: both halves at once
        const uint4 v = *reinterpret_cast<const uint4*>(&sP[tid * 4]);
        uint4 w = make_uint4(0u, 0u, 0u, 0u);
        if (two) w = *reinterpret_cast<const uint4*>(&sP[F_PASS + tid * 4]);
        const u32 m0 = v.x, m1 = m0 > v.y ? m0 : v.y, m2 = m1 > v.z ? m1 : v.z, m3 = m2 > v.w ? m2 : v.w;
        const u32 k0 = w.x, k1 = k0 > w.y ? k0 : w.y, k2 = k1 > w.z ? k1 : w.z, k3 = k2 > w.w ? k2 : w.w;
        const u32 inc1 = smc_wave_scan_max_u32(m3);
        u32 ex1 = smc_mov_dpp<SMC_DPP_WAVE_SHR1>(inc1);
        if (lane == 0) ex1 = 0u;
        u32 inc2 = 0u, ex2 = 0u;
        if (two) {
            inc2 = smc_wave_scan_max_u32(k3);
            ex2 = smc_mov_dpp<SMC_DPP_WAVE_SHR1>(inc2);
            if (lane == 0) ex2 = 0u;
        }
        if (lane == 63) { s_mx[wave] = inc1; s_mx[SMC_NWAVE + wave] = inc2; }
        __syncthreads();
        u32 all1 = 0u;
#pragma unroll
        for (int ww = 0; ww < SMC_NWAVE; ++ww) {
            const u32 x1 = s_mx[ww], x2 = s_mx[SMC_NWAVE + ww];
            all1 = all1 > x1 ? all1 : x1;
            if (ww < wave) {
                ex1 = ex1 > x1 ? ex1 : x1;
                ex2 = ex2 > x2 ? ex2 : x2;
            }
        }
        ex2 = ex2 > all1 ? ex2 : all1;             // the second half continues the first
        const u32 jb = (u32)j0;
        {
            const u32 n0 = pb + (u32)tid * 4u;
            u32 a32[4] = {jb + (m0 > ex1 ? m0 : ex1), jb + (m1 > ex1 ? m1 : ex1),
                          jb + (m2 > ex1 ? m2 : ex1), jb + (m3 > ex1 ? m3 : ex1)};
            if (SQ) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a32[i] = s_pm[a32[i] - jb];
            }
            if (n0 >= lo && n0 + 3u < hi) {                                             // core.py:329
                if (a.nt & 8) smc_st4g_nt(A + n0, a32);
                else smc_st4g(A + n0, a32);
            } else {
#pragma unroll
                for (u32 i = 0; i < 4u; ++i)
                    if (n0 + i >= lo && n0 + i < hi) smc_stg(A + n0 + i, a32[i]);
            }
        }
        if (two) {
            const u32 n0 = pb + F_PASS + (u32)tid * 4u;
            u32 a32[4] = {jb + (k0 > ex2 ? k0 : ex2), jb + (k1 > ex2 ? k1 : ex2),
                          jb + (k2 > ex2 ? k2 : ex2), jb + (k3 > ex2 ? k3 : ex2)};
            if (SQ) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a32[i] = s_pm[a32[i] - jb];
            }
            if (n0 + 3u < hi) {                    // (n0 >= lo: the second half starts 1024 past it)
                if (a.nt & 8) smc_st4g_nt(A + n0, a32);
                else smc_st4g(A + n0, a32);
            } else {
#pragma unroll
                for (u32 i = 0; i < 4u; ++i)
                    if (n0 + i < hi) smc_stg(A + n0 + i, a32[i]);
            }
        }
    }
    F_STAMP_A(6);
}

// ---- SMC_FLAG_STRICT_ANCESTORS: the reference's sequential fp64 CDF (smc_resample.h "STRICT") on the
// filter's own normalised weights, in place of the exact integer CDFs.  W_{t-1} is written out (the
// same values smc_filter_get(SMC_FIELD_W) returns), k_seq_cdf turns it into S in place, every
// offspring searches S with its own sorted uniform.
__global__ void __launch_bounds__(SMC_BLOCK)
k_strict_W(const FArgs av, double* Wout, double* tsum)
{
    // one tile of 1024 weights per workgroup, 4 per thread; tsum (may be null): the tile's fp64 sum -- the estimate
    // smc_seqx.h's classification starts from (k_seq_tile_sums, saved a launch)
    const FArgs& a = av;
    __shared__ double smd[SMC_SM];
    const int isl = (int)blockIdx.y, b = (int)blockIdx.x;
    const double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)smc_uniform(info[0]);
    if (t >= a.T || t == 0 || smc_uniform(info[1]) == 0.0) return;
    const double* row = a.summ + ((i64)isl * (a.T + 1) + (t - 1)) * SUMM_STRIDE;
    const double m = row[5], rs = row[6];
    const double* lwp = f_lw(a, t - 1) + (i64)isl * a.N;
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const i64 i = (i64)b * 1024 + (i64)threadIdx.x * 4 + k;
        if (i >= a.N) continue;
        const double lw = lwp[i];
        double W;
        if (a.kform) {
            double kk;
            double p = smc_expk(lw, kk);
            const bool ok = lw > -INFINITY;
            p = ok ? p : 0.0;
            kk = ok ? kk : -INFINITY;
            W = smc_scale_pk(p, kk, m) * rs;
        } else {
            W = f_weight(lw, m, rs);
        }
        Wout[(i64)isl * a.N + i] = W;
        v += W;
    }
    if (tsum) {
        v = smc_block_sum(v, smd);
        if (threadIdx.x == 0) tsum[(i64)isl * gridDim.x + b] = v;
    }
}
// (k_seq_cdf needs the decision too: a wrapper that returns early when the step does not resample)
__global__ void __launch_bounds__(64)
k_strict_cdf(const FArgs av, double* WS)
{
    const FArgs& a = av;
    const int isl = (int)blockIdx.y, lane = (int)threadIdx.x;
    const double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)smc_uniform(info[0]);
    if (t >= a.T || t == 0 || smc_uniform(info[1]) == 0.0) return;
    double* o = WS + (i64)isl * a.N;
    const i64 n = a.N;
    double s = 0.0;
    bool first = true;
    for (i64 c = 0; c < n; c += 64) {
        const i64 i = c + lane;
        const double wi = i < n ? o[i] : 0.0;
        double mine = 0.0;
        const int m = (int)(n - c < 64 ? n - c : 64);
        for (int k = 0; k < m; ++k) {
            const double wk = smc_readlane_f64(wi, k);
            s = first ? wk : s + wk;
            first = false;
            if (lane == k) mine = s;
        }
        if (i < n) o[i] = mine;
    }
}
// (the searches against a MATERIALISED S: filters of one tile, the flat test paths, the literal walk; the two-level step
//  never writes S -- smc_filter_strict.h)
__global__ void __launch_bounds__(SMC_BLOCK)
k_strict_search_S(const FArgs av, const double* S, const double* su_mem)
{
    const FArgs& a = av;
    const int isl = (int)blockIdx.y;
    const double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)smc_uniform(info[0]);
    if (t >= a.T || t == 0 || smc_uniform(info[1]) == 0.0) return;
    const i64 N = a.N;
    SmcSu su;
    su.scheme = a.scheme;
    su.M = N;
    su.dM = (double)N;
    su.u = a.ut ? a.ut + ((i64)t * a.n_islands + isl) * a.ut_stride
                : (a.scheme == SMC_MULTINOMIAL_ ? su_mem + (i64)isl * N : nullptr);
    su.u_sys = 0.0;
    su.seed = a.seed;
    su.t = (u32)t;
    su.island = (u32)(a.island_offset + isl);
    if (a.scheme == SMC_SYSTEMATIC_) {
        if (su.u) su.u_sys = su.u[0];
        else {
            u64 x0, x1;
            smc_philox_uniform(0u, su.t, su.island, SMC_STREAM_RESAMPLE, su.seed, x0, x1);
            su.u_sys = smc_u01_halfopen(x0);
        }
    }
    const i64 p = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;       // offspring 2p, 2p + 1
    if (2 * p >= N) return;
    double s0, s1;
    smc_su_pair(su, p, s0, s1);
    u32* A = f_A(a, t) + (i64)isl * N;
    const double* Si = S + (i64)isl * N;
    A[2 * p] = (u32)smc_first_ge(Si, N, s0);
    if (2 * p + 1 < N) A[2 * p + 1] = (u32)smc_first_ge(Si, N, s1);
}

// smc_filter_spacings: the sorted uniforms the multinomial resampling of step t draws in production
// mode, written out (inspection / tests: the step loop itself never materialises them).  ONE
// workgroup walks the tiles of draws in order: same integers, same prefix sums, same quotients as
// k_f_spacing_sums / _scan / k_ancestors2<REGEN> (and k_f_spacing_write on the flat step).
__global__ void __launch_bounds__(SMC_BLOCK)
k_f_spacings_out(const FArgs av, const i64 t, const int isl, double* out)
{
    const FArgs& a = av;
    __shared__ u64 smu[SMC_SM];
    SMC_NTAB_LDS(s_ntab);
    const int tid = (int)threadIdx.x;
    smc_ntab_stage<SMC_BLOCK>(s_ntab, tid);
    __syncthreads();
    const u32 gisl = (u32)(a.island_offset + isl);
    u64* Z = reinterpret_cast<u64*>(out);
    u64 carry = 0ull;
    for (int k = 0; k < a.ntiles1; ++k) {
        const i64 n0 = (i64)k * F_TILE + (i64)tid * F_IPT;
        u64 q[4];
        f_spacing_q4(a, s_ntab, (u32)t, gisl, n0, q);
        u64 tot;
        u64 run = smc_block_exscan_u64(q[0] + q[1] + q[2] + q[3], smu, tot);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            run += q[i];
            if (n0 + i < a.N) Z[n0 + i] = carry + (run < SP_OFF_MAX ? run : SP_OFF_MAX);
        }
        carry += tot;
        __syncthreads();
    }
    const double dall = (double)carry;
    for (i64 n = tid; n < a.N; n += SMC_BLOCK) out[n] = (double)Z[n] / dall;
}

// the same inside the step loop (strict mode, Philox multinomial): step and decision from the record
__global__ void __launch_bounds__(SMC_BLOCK)
k_f_spacings_step(const FArgs av, const int isl, double* out)
{
    const FArgs& a = av;
    __shared__ u64 smu[SMC_SM];
    SMC_NTAB_LDS(s_ntab);
    const int tid = (int)threadIdx.x;
    smc_ntab_stage<SMC_BLOCK>(s_ntab, tid);
    __syncthreads();
    const double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)smc_uniform(info[0]);
    if (t >= a.T || t == 0 || smc_uniform(info[1]) == 0.0) return;
    const u32 gisl = (u32)(a.island_offset + isl);
    u64* Z = reinterpret_cast<u64*>(out);
    u64 carry = 0ull;
    for (int k = 0; k < a.ntiles1; ++k) {
        const i64 n0 = (i64)k * F_TILE + (i64)tid * F_IPT;
        u64 q[4];
        f_spacing_q4(a, s_ntab, (u32)t, gisl, n0, q);
        u64 tot;
        u64 run = smc_block_exscan_u64(q[0] + q[1] + q[2] + q[3], smu, tot);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            run += q[i];
            if (n0 + i < a.N) Z[n0 + i] = carry + (run < SP_OFF_MAX ? run : SP_OFF_MAX);
        }
        carry += tot;
        __syncthreads();
    }
    const double dall = (double)carry;
    for (i64 n = tid; n < a.N; n += SMC_BLOCK) out[n] = (double)Z[n] / dall;
}

// the summary row of the last step done and the (K, 1/s) W is formed with: enqueued at the end
// of every smc_filter_step call of the two-level path (one workgroup per island; idempotent --
// k_ancestors2 of the next step writes the same row again)
__global__ void __launch_bounds__(SMC_BLOCK)
k_flush2(const FArgs av)
{
    const FArgs& a = av;
    __shared__ double smd[SMC_SM];
    const int isl = (int)blockIdx.x;
    const i64 t = (i64)smc_uniform(smc_ldg(a.info2 + (i64)isl * INFO_STRIDE));       // steps done
    if (t <= 0 || t > a.T) return;                      // (t > T: records frozen by k_theta_update)
    F2Red r;
    const bool plain = a.pm2 != nullptr;                // APF: the row describes the plain weights
    if (a.nparts <= 4 * 4 * SMC_BLOCK) {                // all partials in registers: one memory latency
        double pmc[4][4], psc[4][4];
        r = f2_reduce_island_cached<4>(a, isl, smd, pmc, psc, plain);
    } else {
        r = f2_reduce_island(a, isl, smd, plain);
    }
    if (threadIdx.x == 0) f2_write_row(a, isl, t - 1, r);
}

// smc_filter_set_state: the log-weights of the current step were replaced from the host; the
// per-tile partials k_propagate would have left (same device function, same bits) ...
__global__ void __launch_bounds__(SMC_BLOCK)
k_f_partials(const FArgs av, const i64 ts, const int two_level)
{
    const FArgs& a = av;
    __shared__ double smd[SMC_SM];
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y;
    const i64 N = a.N;
    const double* lwp = f_lw(a, ts) + (i64)isl * N;
    double lw[F_IPT];
    const FOwn own = two_level ? f_own<true>(b, (int)threadIdx.x, N) : f_own<false>(b, (int)threadIdx.x, N);
#pragma unroll
    for (int k = 0; k < F_IPT; ++k) {
        const i64 n = f_own_idx(own, k);
        lw[k] = (n < N) ? smc_ldg(lwp + n) : -INFINITY;
    }
    const i64 o = (i64)isl * a.nparts;
    if (two_level) {
        u64 cx[4];
        const F2Tile r = a.strict_e ? f2_tile_weights_e(lw, cx) : f2_tile_weights(lw, cx);
        u64* cq = a.cq + (i64)isl * a.ncq;
        smc_st2g(cq + own.na, cx[0], cx[1]);
        smc_st2g(cq + own.nb, cx[2], cx[3]);
        if (threadIdx.x == 0) { a.pm[o + b] = r.K; a.ps[o + b] = r.S; a.pss[o + b] = r.SS; a.tq[o + b] = r.tb; }
        return;
    }
    const SmcLse r = f_tile_lse<F_IPT>(lw, smd);
    if (threadIdx.x == 0) {
        a.pm[o + b] = r.m;
        a.ps[o + b] = r.s;
        a.pss[o + b] = r.ss;
    }
}
// ... and, on the paths whose k_propagate finalises the step itself (flat CDF, one-launch small
// filter), the summary row of step ts and the record of step ts + 1, from those partials
__global__ void __launch_bounds__(SMC_BLOCK)
k_f_restate(const FArgs av, const i64 ts)
{
    const FArgs& a = av;
    __shared__ double smd[SMC_SM];
    const int isl = (int)blockIdx.x;
    const i64 o = (i64)isl * a.nparts;
    const SmcLse g = smc_lse_reduce_partials<false>(a.pm + o, a.ps + o, a.pss + o, a.nparts, smd);
    if (threadIdx.x == 0) {
        const double* row = a.summ + ((i64)isl * (a.T + 1) + ts) * SUMM_STRIDE;
        f_finalise_step(a, isl, ts, ts == 0, row[4] != 0.0, g, a.info + (i64)isl * INFO_STRIDE);
    }
}

// ---------------------------------------------------------------------------
// SMC^2 (smc_samplers.py:1038-1167), theta level on the device.  Every island is the particle
// filter of one theta-particle; after each time step the theta-weights pick up the islands'
// evidence increments, lw_theta[i] += log p(y_t | y_{0:t-1}, theta_i) (SMC2.logG, :1099-1120), and
// the theta-level ESS decides whether a resample-move step is due (core.py:181-183 applied to the
// outer SMC).  When it is, the kernel FREEZES the batch: it records the step at which to stop and
// pushes the device-resident time index of every island past T, so that the steps the host has
// already enqueued behind it return at once -- the host enqueues K steps at a time and looks at
// the stop record once per K, instead of synchronising after every step.
// One workgroup.  th: [0] stop step (0 = running), [1] theta-ESS, [2] steps accounted for,
// [3] log-mean-exp of lw_theta before the last resampling (evidence bookkeeping: unused here).
// ---------------------------------------------------------------------------
#define TH_STRIDE 8
// `gathered` (sharded population, smc_filter_theta_enable_sharded): the evidence increments of ALL Ng
// theta-particles, gathered from the ranks by the launch before (k_theta_pack + ncclAllGather); lwth then
// holds Ng replicated log-weights and the reductions run over them -- the same on every rank.
__global__ void __launch_bounds__(SMC_BLOCK)
k_theta_update(const FArgs av, double* lwth, double* th, double* ess_log, const double ess_min, const int two_level,
               const double* gathered, const int Ng)
{
    const FArgs& a = av;
    __shared__ double smd[SMC_SM];
    const int tid = (int)threadIdx.x, M = a.n_islands;
    if (th[0] != 0.0) return;                                  // frozen: waiting for the host
    const i64 t = (i64)th[2];                                  // the step just done
    if (t >= a.T) return;
    // did the step really run?  (the filter's own record says how many steps are done)
    const double done = two_level ? a.info2[0] : a.info[0];
    if ((i64)done < t + 1) return;
    SmcLse acc = smc_lse_empty();
    for (int i = tid; i < Ng; i += SMC_BLOCK) {
        const double inc = gathered ? gathered[i]
                                    : a.summ[((i64)i * (a.T + 1) + t) * SUMM_STRIDE + 2];      // loglt of step t
        double l = lwth[i] + inc;
        if (l != l) l = -INFINITY;
        lwth[i] = l;
        smc_lse_push(acc, l);
    }
    const SmcLse g = smc_lse_block(acc, smd);
    if (tid == 0) {
        const double ess = (g.s * g.s) / g.ss;
        th[1] = ess;
        th[2] = (double)(t + 1);
        ess_log[t] = ess;
        ess_log[a.T + t] = g.m + log(g.s / (double)Ng);        // log-mean of the theta weights after step t
                                                               // (the outer evidence, core.py:355-359)
    }
    const bool stop = !((g.s * g.s) / g.ss >= ess_min) && t + 1 < a.T;
    if (!stop) return;
    if (tid == 0) th[0] = (double)(t + 1);
    for (int i = tid; i < M; i += SMC_BLOCK) {                 // freeze: every kernel returns on t >= T
        a.info[(i64)i * INFO_STRIDE] = 1e18;
        a.info2[(i64)i * INFO_STRIDE] = 1e18;
    }
}
// sharded population: my islands' evidence increments of the step just done, in the send buffer of the
// all-gather (zeros when the batch is frozen or past T: k_theta_update then ignores what was gathered)
__global__ void __launch_bounds__(SMC_BLOCK)
k_theta_pack(const FArgs av, const double* th, double* send, const int two_level)
{
    const FArgs& a = av;
    const int tid = (int)threadIdx.x, M = a.n_islands;
    const i64 t = (i64)th[2];
    const double done = two_level ? a.info2[0] : a.info[0];
    const bool live = th[0] == 0.0 && t < a.T && (i64)done >= t + 1;
    for (int i = tid; i < M; i += SMC_BLOCK)
        send[i] = live ? a.summ[((i64)i * (a.T + 1) + t) * SUMM_STRIDE + 2] : 0.0;
}
// thaw: the time records back to step t (after the host has dealt with the stop)
__global__ void __launch_bounds__(SMC_BLOCK)
k_theta_thaw(const FArgs av, double* th, const double t)
{
    const FArgs& a = av;
    for (int i = (int)threadIdx.x; i < a.n_islands; i += SMC_BLOCK) {
        a.info[(i64)i * INFO_STRIDE] = t;
        a.info2[(i64)i * INFO_STRIDE] = t;
    }
    if (threadIdx.x == 0) { th[0] = 0.0; th[2] = t; }
}
// out[i][:] = in[src[i]][:] (src == null: identity) where keep == null or keep[i] != 0, for a
// per-island array of `words` 8-byte words per island; grid (chunks, islands)
__global__ void __launch_bounds__(SMC_BLOCK)
k_island_gather(const u64* in, u64* out, const i64* src, const unsigned char* keep, i64 words)
{
    const i64 i = (i64)blockIdx.y;
    if (keep && !keep[i]) return;
    const i64 s_ = src ? src[i] : i;
    for (i64 w = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x; w < words; w += (i64)gridDim.x * SMC_BLOCK)
        out[i * words + w] = in[s_ * words + w];
}

// pack / unpack island states for migration between GPUs: entry j of the pack buffer (stride
// `pstride` words, this array at word offset `off`) <-> island idx[j] of a per-island array
__global__ void __launch_bounds__(SMC_BLOCK)
k_island_pack(u64* arr, i64 words, u64* pack, i64 pstride, i64 off, const i64* idx, int unpack)
{
    const i64 j = (i64)blockIdx.y;
    u64* a = arr + idx[j] * words;
    u64* p = pack + j * pstride + off;
    for (i64 w = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x; w < words; w += (i64)gridDim.x * SMC_BLOCK) {
        if (unpack) a[w] = p[w];
        else p[w] = a[w];
    }
}

// W = exp(lw - m)/s for one island (SMC.W)
__global__ void __launch_bounds__(SMC_BLOCK)
k_f_write_W(const double* lw, i64 N, const double* row, double* W, const int kform)
{
    const double m = row[5], rs = row[6];     // kform (two-level path): row[5] = K, W = p 2^(k - K) / s
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i >= N) return;
    if (kform) {
        double k;
        double p = smc_expk(lw[i], k);
        const bool ok = lw[i] > -INFINITY;
        p = ok ? p : 0.0;
        k = ok ? k : -INFINITY;
        W[i] = smc_scale_pk(p, k, m) * rs;
    } else {
        W[i] = f_weight(lw[i], m, rs);
    }
}

// one backward step of the genealogy (smoothing.py:209-219): B_{s-1} = A_s[B_s]
// (A == nullptr: step s did not resample, A_s = arange)
__global__ void __launch_bounds__(SMC_BLOCK)
k_f_genealogy(const u32* A, const i64* Bs, i64 N, i64* Bprev)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < N) Bprev[i] = A ? (i64)A[Bs[i]] : Bs[i];
}
__global__ void __launch_bounds__(SMC_BLOCK)
k_f_iota(i64 N, i64* B)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < N) B[i] = i;
}

// ancestors as the ABI hands them out: int64 (resampling.py:503)
__global__ void __launch_bounds__(SMC_BLOCK)
k_f_widen(const u32* A, i64 N, i64* out)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < N) out[i] = (i64)A[i];
}

// ---------------------------------------------------------------------------
// Moments collector on the device (collectors.py:301-317 with the default
// FeynmanKac.default_moments = rs.wmean_and_var, resampling.py:320-338): weighted mean
// and variance of every component of X_t, written per step so that a run with
// collect=[Moments()] stays one asynchronous launch sequence.  Runs after k_propagate(t)
// (the record already says t+1).
// ---------------------------------------------------------------------------
#define F_MOM_CHUNK 4096
__global__ void __launch_bounds__(SMC_BLOCK)
k_f_moments_partials(const FArgs av)
{
    const FArgs& a = av;
    __shared__ double sm[SMC_SM];
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y;
    const i64 t = (i64)smc_uniform(smc_ldg((a.kform ? a.info2 : a.info) + (i64)isl * INFO_STRIDE)) - 1;   // step just done
    if (t < 0 || t >= a.T) return;
    const double* row = a.summ + ((i64)isl * (a.T + 1) + t) * SUMM_STRIDE;     // (two-level: k_flush2 ran first)
    const double m = smc_uniform(smc_ldg(row + 5)), rs = smc_uniform(smc_ldg(row + 6));
    const i64 N = a.N;
    const int d = a.dx;
    const double* X = f_X(a, t) + (i64)isl * N * d;
    const double* lw = f_lw(a, t) + (i64)isl * N;
    const i64 base = (i64)b * F_MOM_CHUNK;
    for (int c = 0; c < d; ++c) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        for (int k = 0; k < F_MOM_CHUNK / SMC_BLOCK; ++k) {
            const i64 i = base + (i64)k * SMC_BLOCK + threadIdx.x;
            if (i < N) {
                const double l = smc_ldg(lw + i);
                double w;
                if (a.kform) {                     // W = p 2^(k - K) / s, as k_f_write_W forms it
                    double k;
                    const double p = smc_expk(l, k);
                    w = (l > -INFINITY) ? smc_scale_pk(p, k, m) * rs : 0.0;
                } else {
                    w = f_weight(l, m, rs);
                }
                const double x = smc_ldg(X + i * d + c);
                a0 += w;
                a1 += w * x;
                a2 += w * (x * x);
            }
        }
        a0 = smc_block_sum(a0, sm);
        a1 = smc_block_sum(a1, sm);
        a2 = smc_block_sum(a2, sm);
        if (threadIdx.x == 0) {
            double* p = a.mpart + (((i64)isl * a.nmb + b) * d + c) * 3;
            p[0] = a0; p[1] = a1; p[2] = a2;
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(SMC_BLOCK)
k_f_moments_final(const FArgs av)
{
    const FArgs& a = av;
    const int isl = (int)blockIdx.x;
    const i64 t = (i64)smc_uniform(smc_ldg((a.kform ? a.info2 : a.info) + (i64)isl * INFO_STRIDE)) - 1;
    if (t < 0 || t >= a.T) return;
    const int d = a.dx;
    __shared__ double sm[SMC_SM];
    for (int c = 0; c < d; ++c) {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0;
        for (int b = (int)threadIdx.x; b < a.nmb; b += SMC_BLOCK) {
            const double* p = a.mpart + (((i64)isl * a.nmb + b) * d + c) * 3;
            t0 += p[0]; t1 += p[1]; t2 += p[2];
        }
        t0 = smc_block_sum(t0, sm);                            // fixed association order
        t1 = smc_block_sum(t1, sm);
        t2 = smc_block_sum(t2, sm);
        __syncthreads();
        if (threadIdx.x != 0) continue;
        // np.average(x, weights=W) = sum(W x) / sum(W)   (resampling.py:335-337)
        const double mean = t1 / t0, m2 = t2 / t0;
        double* o = a.mom + ((i64)isl * a.T + t) * 2 * d;
        o[c] = mean;
        o[d + c] = m2 - mean * mean;
    }
}

// one genealogical line through the kept history (smoothing.py:256-269
// extract_one_trajectory): n_{t-1} = A_t[n_t]; out (t_end, d) = X_t[n_t].  One lane.
__global__ void k_f_one_trajectory(const FArgs av, int isl, i64 n_last, i64 t_end,
                                   const double* rsflag /* (t_end, SUMM_STRIDE) rows */, double* out)
{
    const FArgs& a = av;
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    i64 n = n_last;
    for (i64 t = t_end - 1; t >= 0; --t) {
        const double* X = f_X(a, t) + (i64)isl * a.N * a.dx;
        for (int c = 0; c < a.dx; ++c) out[t * a.dx + c] = X[n * a.dx + c];
        if (t > 0 && rsflag[t * SUMM_STRIDE + 4] != 0.0)          // else A_t = arange (core.py:336)
            n = (i64)(f_A(a, t) + (i64)isl * a.N)[n];
    }
}
