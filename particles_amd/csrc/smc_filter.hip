// smc_filter.hip -- the fused on-device SMC step loop.
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
//       weight, the resample decision of step t (core.py:181-183), and -- if
//       resampling -- its tile's total of Q62 weights  W_i = exp(lw_i-m)/s.
//   k_move(t): one workgroup per tile of TILE consecutive parents.  Builds the
//       tile's exact CDF in LDS, derives the contiguous range of offspring it
//       owns (smc_resample.h), and for each offspring: parent search in LDS,
//       gather of the parent state from LDS, propagation  x = loc(xp)+scale*z
//       with a counted Philox normal (or a replayed draw), the weight
//       increment log G, and the online log-sum-exp partial of the new weights.
//
// The time index lives in device memory (ctl[0]/ctl[1], ping-ponged between
// the two kernels) so the same pair of launches -- or one hipGraph holding many
// pairs -- serves every step.
//
// HBM traffic per particle-step on a resampling step (d = 1):
//   k_prepare: read lw (8)            k_move: read lw, X (16), write X, lw, A (24)
// = 48 B moved vs 56 B "algorithmic" (SURVEY 8d): the normalised weights W are
// never materialised.
#include <vector>

#include "smc_internal.h"
#include "smc_resample.h"

#define F_IPT 4
#define F_TILE (SMC_BLOCK * F_IPT)
#define SUMM_STRIDE 8   /* ESS, log_mean, loglt, logLt, rs_flag, m, s, - */
#define PARAM_STRIDE 16

struct FArgs {
    i64 N, T;
    int ntiles, n_islands, scheme, rng_mode, island_offset;
    double ess_thresh;
    u64 seed;
    double *X0, *X1, *lw0, *lw1;
    i64* A;
    u64* Q;
    double *pm, *ps, *pss;
    double* summ;          // (n_islands, T+1, SUMM_STRIDE)
    const double* params;  // (n_islands, PARAM_STRIDE)
    const double* y;       // (T,)
    i64* ctl;              // [0] = t seen by k_prepare, [1] = t seen by k_move
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
template <int KIND>
__device__ __forceinline__ double m_init_scale(const double* p)
{
    return p[3];
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
        const double x = first ? m_init_loc<KIND>(p) + m_init_scale<KIND>(p) * z
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

// ---------------------------------------------------------------------------
// k_prepare
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(SMC_BLOCK)
k_prepare(FArgs a, int finalize_only)
{
    __shared__ double smd[SMC_NWAVE];
    __shared__ u64 smu[SMC_NWAVE];
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y;
    const i64 t = a.ctl[0];
    if (b == 0 && isl == 0 && threadIdx.x == 0) a.ctl[1] = t;
    if (t == 0) return;               // nothing to finalise; step 0 never resamples
    const i64 tp = t - 1;
    const double* pm = a.pm + (i64)isl * a.ntiles;
    const double* ps = a.ps + (i64)isl * a.ntiles;
    const double* pss = a.pss + (i64)isl * a.ntiles;
    const SmcLse r = smc_lse_reduce_partials(pm, ps, pss, a.ntiles, smd);
    const bool bad = !(r.m > -INFINITY) || !(r.m < INFINITY);
    const double ess = bad ? NAN : (r.s * r.s) / r.ss;                  // resampling.py:226
    const double log_mean = bad ? NAN : r.m + log(r.s / (double)a.N);   // resampling.py:224
    const bool flag = (t < a.T) && (ess < a.ess_thresh);                // core.py:181-183
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
        row[6] = bad ? NAN : r.s;
        if (t < a.T) row[SUMM_STRIDE + 4] = flag ? 1.0 : 0.0;
    }
    if (t >= a.T || finalize_only || !flag) return;
    // tile total of the Q62 weights of step t-1 (the parents of step t)
    const double* lw = ((tp & 1) ? a.lw1 : a.lw0) + (i64)isl * a.N;
    const i64 j0 = (i64)b * F_TILE + (i64)threadIdx.x * F_IPT;
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < F_IPT; ++i)
        if (j0 + i < a.N) s += smc_q62_w(exp(lw[j0 + i] - r.m) / r.s);
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
    __shared__ u64 smu[SMC_NWAVE];
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y;
    const i64 t = a.ctl[1];
    if (t >= a.T || t == 0) return;
    if (a.summ[((i64)isl * (a.T + 1) + t) * SUMM_STRIDE + 4] == 0.0) return;
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
    __shared__ u64 smu[SMC_NWAVE];
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y;
    const i64 t = a.ctl[1];
    if (t >= a.T || t == 0) return;
    if (a.summ[((i64)isl * (a.T + 1) + t) * SUMM_STRIDE + 4] == 0.0) return;
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
    pre = smc_block_sum_u64(pre, smu);
    all = smc_block_sum_u64(all, smu);
    u64 tot;
    u64 run = pre + smc_block_exscan_u64(tsum, smu, tot);
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
    __shared__ u64 sC[F_TILE];
    __shared__ double sX[F_TILE];
    __shared__ u64 smu[SMC_NWAVE];
    __shared__ double smd[SMC_NWAVE];
    __shared__ i64 sn[2];
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y;
    const i64 t = a.ctl[1];
    if (t >= a.T) return;
    if (b == 0 && isl == 0 && threadIdx.x == 0) a.ctl[0] = t + 1;

    const i64 N = a.N;
    const double* p = a.params + (i64)isl * PARAM_STRIDE;
    const double yt = a.y[t];
    const u32 gisl = (u32)(a.island_offset + isl);
    const int cur = (int)(t & 1);
    double* Xn = (cur ? a.X1 : a.X0) + (i64)isl * N;
    const double* Xo = (cur ? a.X0 : a.X1) + (i64)isl * N;
    double* lwn = (cur ? a.lw1 : a.lw0) + (i64)isl * N;
    const double* lwo = (cur ? a.lw0 : a.lw1) + (i64)isl * N;
    i64* A = a.A + (i64)isl * N;
    const double* zt = a.zt ? a.zt + ((i64)t * a.n_islands + isl) * N : nullptr;
    const double* row = a.summ + ((i64)isl * (a.T + 1) + t) * SUMM_STRIDE;
    const bool first = (t == 0);
    const bool resample = !first && row[4] != 0.0;

    SmcLse acc = smc_lse_empty();
    const i64 j0 = (i64)b * F_TILE;

    if (!resample) {
        // ---- element-wise step over the workgroup's own tile
        for (int k = 0; k < F_IPT / 2; ++k) {
            const i64 pr = j0 / 2 + (i64)k * SMC_BLOCK + threadIdx.x;
            const i64 n0 = 2 * pr;
            if (n0 >= N) continue;
            double z0, z1;
            if (zt) {
                z0 = zt[n0];
                z1 = (n0 + 1 < N) ? zt[n0 + 1] : 0.0;
            } else {
                smc_normal_pair(a.seed, (u32)pr, (u32)t, gisl, SMC_STREAM_NORMAL, z0, z1);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const i64 n = n0 + e;
                if (n >= N) break;
                double inc;
                const double xp = first ? 0.0 : Xo[n];
                const double x = m_step<KIND, FK>(p, first, yt, xp, e ? z1 : z0, inc);
                double lw = first ? inc : lwo[n] + inc;      // resampling.py:241-244
                if (lw != lw) lw = -INFINITY;                // resampling.py:220
                Xn[n] = x;
                lwn[n] = lw;
                if (!first) A[n] = n;                        // core.py:335
                smc_lse_push(acc, lw);
            }
        }
    } else {
        // ---- resample-move: this workgroup's tile of parents
        const double m = row[5 - SUMM_STRIDE], s = row[6 - SUMM_STRIDE];
        u64 wq[F_IPT];
#pragma unroll
        for (int i = 0; i < F_IPT; ++i) {
            const i64 j = j0 + (i64)threadIdx.x * F_IPT + i;
            const bool ok = j < N;
            wq[i] = ok ? smc_q62_w(exp(lwo[j] - m) / s) : 0ull;
            sX[threadIdx.x * F_IPT + i] = ok ? Xo[j] : 0.0;
        }
        u64 total;
        const u64 pre = smc_tile_cdf<F_IPT>(wq, a.Q + (i64)isl * a.ntiles, b, sC, smu, total);
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
                smc_philox(0u, su.t, su.island, SMC_STREAM_RESAMPLE, su.seed, x0, x1);
                su.u_sys = smc_u01_halfopen(x0);
            }
        }
        i64 n_lo, n_hi;
        smc_tile_outputs(su, b, a.ntiles, pre, total, sn, n_lo, n_hi);
        const int nvalid = (int)((N - j0 < F_TILE) ? (N - j0) : F_TILE);
        for (i64 pr = (n_lo >> 1) + threadIdx.x; 2 * pr < n_hi; pr += SMC_BLOCK) {
            const i64 n0 = 2 * pr;
            double s0, s1;
            smc_su_pair(su, pr, s0, s1);
            double z0, z1;
            if (zt) {
                z0 = (n0 < N) ? zt[n0] : 0.0;
                z1 = (n0 + 1 < N) ? zt[n0 + 1] : 0.0;
            } else {
                smc_normal_pair(a.seed, (u32)pr, (u32)t, gisl, SMC_STREAM_NORMAL, z0, z1);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const i64 n = n0 + e;
                if (n < n_lo || n >= n_hi) continue;
                int jl = smc_lower_bound_u64(sC, F_TILE, smc_q62_t(e ? s1 : s0));
                jl = jl < nvalid ? jl : nvalid - 1;
                double inc;
                const double x = m_step<KIND, FK>(p, false, yt, sX[jl], e ? z1 : z0, inc);
                double lw = inc;                             // weights reset, core.py:299-305
                if (lw != lw) lw = -INFINITY;
                A[n] = j0 + jl;
                Xn[n] = x;
                lwn[n] = lw;
                smc_lse_push(acc, lw);
            }
        }
    }
    const SmcLse r = smc_lse_block(acc, smd);
    if (threadIdx.x == 0) {
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
    const double m = row[5], s = row[6];
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < N) W[i] = exp(lw[i] - m) / s;
}

__global__ void __launch_bounds__(SMC_BLOCK)
k_f_gather1(const double* X, const i64* A, i64 N, double* Xp)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < N) Xp[i] = X[A[i]];
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
#define PROF_MAX 4096

struct smc_filter {
    smc_ctx* ctx;
    FArgs a;
    int kind, fk;
    i64 t_host;
    void* slab;            // one allocation holding every device array
    bool use_graph;
    hipGraphExec_t gexec;
    int graph_steps;
    bool prof;
    std::vector<hipEvent_t> ev;
    int prof_n;
    double* tmp;           // (N,) staging for W / Xp downloads
};

typedef void (*move_fn)(FArgs);

static void launch_move(smc_filter* f, dim3 grid)
{
    hipStream_t st = f->ctx->stream;
    if (f->kind == SMC_MODEL_LINGAUSS && f->fk == SMC_FK_BOOTSTRAP)
        SMC_LAUNCH((k_move<SMC_MODEL_LINGAUSS, SMC_FK_BOOTSTRAP>), grid, dim3(SMC_BLOCK), st, f->a);
    else if (f->kind == SMC_MODEL_LINGAUSS)
        SMC_LAUNCH((k_move<SMC_MODEL_LINGAUSS, SMC_FK_GUIDED>), grid, dim3(SMC_BLOCK), st, f->a);
    else
        SMC_LAUNCH((k_move<SMC_MODEL_STOCHVOL, SMC_FK_BOOTSTRAP>), grid, dim3(SMC_BLOCK), st, f->a);
}

static void enqueue_step(smc_filter* f, int k_prof)
{
    hipStream_t st = f->ctx->stream;
    const dim3 grid(f->a.ntiles, f->a.n_islands);
    if (k_prof >= 0) (void)hipEventRecord(f->ev[3 * k_prof], st);
    SMC_LAUNCH(k_prepare, grid, dim3(SMC_BLOCK), st, f->a, 0);
    if (f->a.scheme == SMC_MULTINOMIAL && !f->a.ut) {
        const dim3 g1(f->a.ntiles1, f->a.n_islands);
        SMC_LAUNCH(k_f_spacing_sums, g1, dim3(SMC_BLOCK), st, f->a);
        SMC_LAUNCH(k_f_spacing_write, g1, dim3(SMC_BLOCK), st, f->a);
    }
    if (k_prof >= 0) (void)hipEventRecord(f->ev[3 * k_prof + 1], st);
    launch_move(f, grid);
    if (k_prof >= 0) (void)hipEventRecord(f->ev[3 * k_prof + 2], st);
}

extern "C" {

int smc_filter_create(smc_ctx* ctx, const smc_model* model, const smc_filter_opts* o,
                      const double* y_host, smc_filter** out)
{
    SMC_REQUIRE(ctx && model && o && y_host && out, "null argument");
    SMC_REQUIRE(o->N > 0 && o->T > 0 && o->n_islands > 0, "N, T, n_islands must be positive");
    if (o->scheme != SMC_MULTINOMIAL && o->scheme != SMC_STRATIFIED &&
        o->scheme != SMC_SYSTEMATIC) {
        smc_set_error("%d is not a valid resampling scheme", o->scheme);
        return SMC_ERR_SCHEME;
    }
    SMC_REQUIRE(model->kind == SMC_MODEL_LINGAUSS || model->kind == SMC_MODEL_STOCHVOL,
                "fused filter: model kind must be LINGAUSS or STOCHVOL");
    SMC_REQUIRE(model->fk == SMC_FK_BOOTSTRAP ||
                    (model->fk == SMC_FK_GUIDED && model->kind == SMC_MODEL_LINGAUSS),
                "guided filter is available for LINGAUSS only");
    SMC_REQUIRE(model->params_host, "params_host is required");
    SMC_REQUIRE(o->N < ((i64)1 << 32), "N must be below 2^32");
    SMC_HIP_CHECK(hipSetDevice(ctx->device));

    smc_filter* f = new smc_filter();
    f->ctx = ctx;
    f->kind = model->kind;
    f->fk = model->fk;
    f->t_host = 0;
    f->use_graph = o->use_graph != 0;
    f->gexec = nullptr;
    f->graph_steps = 0;
    f->prof = false;
    f->prof_n = 0;
    FArgs& a = f->a;
    memset(&a, 0, sizeof a);
    a.N = o->N;
    a.T = o->T;
    a.n_islands = o->n_islands;
    a.ntiles = (int)((o->N + F_TILE - 1) / F_TILE);
    a.ntiles1 = (int)((o->N + 1 + F_TILE - 1) / F_TILE);
    a.scheme = o->scheme;
    a.rng_mode = o->rng_mode;
    a.island_offset = o->island_offset;
    a.ess_thresh = (double)o->N * o->ESSrmin;
    a.seed = o->seed;
    {
        int lg = 0;
        while (((i64)1 << lg) < o->N + 2) ++lg;
        a.spacing_scale = ldexp(1.0, 57 - lg);
    }
    const size_t M = (size_t)o->n_islands, N = (size_t)o->N, T = (size_t)o->T;
    const bool need_su = (o->scheme == SMC_MULTINOMIAL);
    // carve one slab
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o0 = off; off = smc_align_up(off + bytes, 256); return o0; };
    const size_t oX0 = carve(M * N * 8), oX1 = carve(M * N * 8);
    const size_t oL0 = carve(M * N * 8), oL1 = carve(M * N * 8);
    const size_t oA = carve(M * N * 8);
    const size_t oQ = carve(M * a.ntiles * 8);
    const size_t oPm = carve(M * a.ntiles * 8), oPs = carve(M * a.ntiles * 8),
                 oPss = carve(M * a.ntiles * 8);
    const size_t oSum = carve(M * (T + 1) * SUMM_STRIDE * 8);
    const size_t oPar = carve(M * PARAM_STRIDE * 8);
    const size_t oY = carve(T * 8);
    const size_t oCtl = carve(64);
    const size_t oSu = carve(need_su ? M * N * 8 : 8);
    const size_t oE = carve(need_su ? M * a.ntiles1 * 8 : 8);
    const size_t oTmp = carve(N * 8);
    void* slab = nullptr;
    hipError_t e = hipMalloc(&slab, off);
    if (e != hipSuccess) {
        smc_set_error("smc_filter_create: %zu bytes: %s", off, hipGetErrorString(e));
        delete f;
        return SMC_ERR_NOMEM;
    }
    f->slab = slab;
    char* base = (char*)slab;
    a.X0 = (double*)(base + oX0); a.X1 = (double*)(base + oX1);
    a.lw0 = (double*)(base + oL0); a.lw1 = (double*)(base + oL1);
    a.A = (i64*)(base + oA);
    a.Q = (u64*)(base + oQ);
    a.pm = (double*)(base + oPm); a.ps = (double*)(base + oPs); a.pss = (double*)(base + oPss);
    a.summ = (double*)(base + oSum);
    double* dpar = (double*)(base + oPar);
    double* dy = (double*)(base + oY);
    a.params = dpar;
    a.y = dy;
    a.ctl = (i64*)(base + oCtl);
    a.su = (double*)(base + oSu);
    a.E = (u64*)(base + oE);
    f->tmp = (double*)(base + oTmp);
    hipStream_t st = ctx->stream;
    SMC_HIP_CHECK(hipMemsetAsync(a.summ, 0, M * (T + 1) * SUMM_STRIDE * 8, st));
    SMC_HIP_CHECK(hipMemsetAsync(a.ctl, 0, 64, st));
    SMC_HIP_CHECK(hipMemsetAsync(a.A, 0, M * N * 8, st));
    SMC_HIP_CHECK(hipMemcpyAsync(dpar, model->params_host, M * PARAM_STRIDE * 8,
                                 hipMemcpyHostToDevice, st));
    SMC_HIP_CHECK(hipMemcpyAsync(dy, y_host, T * 8, hipMemcpyHostToDevice, st));
    SMC_HIP_CHECK(hipStreamSynchronize(st));
    *out = f;
    return SMC_OK;
}

int smc_filter_destroy(smc_filter* f)
{
    if (!f) return SMC_OK;
    (void)hipSetDevice(f->ctx->device);
    (void)hipStreamSynchronize(f->ctx->stream);
    if (f->gexec) (void)hipGraphExecDestroy(f->gexec);
    for (hipEvent_t e : f->ev) (void)hipEventDestroy(e);
    (void)hipFree(f->slab);
    delete f;
    return SMC_OK;
}

int smc_filter_set_replay(smc_filter* f, const double* z, const double* u)
{
    SMC_REQUIRE(f, "null filter");
    SMC_REQUIRE(f->t_host == 0, "replay tapes must be set before the first step");
    SMC_REQUIRE(z && u, "both tapes are required");
    f->a.zt = z;
    f->a.ut = u;
    f->a.ut_stride = (f->a.scheme == SMC_SYSTEMATIC) ? 1 : f->a.N;
    f->a.rng_mode = SMC_RNG_REPLAY;
    return SMC_OK;
}

int smc_filter_step(smc_filter* f, int64_t nsteps)
{
    SMC_REQUIRE(f, "null filter");
    SMC_REQUIRE(nsteps >= 0, "nsteps must be non-negative");
    SMC_HIP_CHECK(hipSetDevice(f->ctx->device));
    hipStream_t st = f->ctx->stream;
    i64 todo = nsteps;
    if (f->t_host + todo > f->a.T) todo = f->a.T - f->t_host;
    if (todo < 0) todo = 0;
    i64 done = 0;
#ifndef SMC_EMULATE
    const int GS = 25;
    if (f->use_graph && !f->prof && todo > 0) {
        if (!f->gexec) {   // captured once, on the first call (kernel arguments are final by then)
            hipGraph_t g = nullptr;
            if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                for (int k = 0; k < GS; ++k) enqueue_step(f, -1);
                if (hipStreamEndCapture(st, &g) == hipSuccess && g &&
                    hipGraphInstantiate(&f->gexec, g, nullptr, nullptr, 0) == hipSuccess) {
                    f->graph_steps = GS;
                } else {
                    f->gexec = nullptr;
                    f->use_graph = false;
                }
                if (g) (void)hipGraphDestroy(g);
            } else {
                f->use_graph = false;
            }
            (void)hipGetLastError();
        }
        while (f->gexec && todo - done >= f->graph_steps) {
            SMC_HIP_CHECK(hipGraphLaunch(f->gexec, st));
            done += f->graph_steps;
        }
    }
#endif
    for (; done < todo; ++done) {
        int kp = -1;
        if (f->prof && f->prof_n < PROF_MAX) kp = f->prof_n++;
        enqueue_step(f, kp);
    }
    // finalise the summaries of the last step run (idempotent)
    SMC_LAUNCH(k_prepare, dim3(f->a.ntiles, f->a.n_islands), dim3(SMC_BLOCK), st, f->a, 1);
    SMC_LAUNCH_CHECK();
    f->t_host += todo;
    return SMC_OK;
}

int smc_filter_sync(smc_filter* f)
{
    SMC_REQUIRE(f, "null filter");
    SMC_HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
    return SMC_OK;
}

int smc_filter_t(smc_filter* f, int64_t* t_out)
{
    SMC_REQUIRE(f && t_out, "null argument");
    *t_out = f->t_host;
    return SMC_OK;
}

int smc_filter_summaries(smc_filter* f, double* out_host)
{
    SMC_REQUIRE(f && out_host, "null argument");
    const i64 t = f->t_host, T = f->a.T;
    if (t == 0) return SMC_OK;
    std::vector<double> h((size_t)f->a.n_islands * (T + 1) * SUMM_STRIDE);
    SMC_HIP_CHECK(hipMemcpyAsync(h.data(), f->a.summ, h.size() * 8, hipMemcpyDeviceToHost,
                                 f->ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
    for (int i = 0; i < f->a.n_islands; ++i)
        for (i64 s = 0; s < t; ++s)
            for (int c = 0; c < SMC_SUMMARY_COLS; ++c)
                out_host[((size_t)i * t + s) * SMC_SUMMARY_COLS + c] =
                    h[((size_t)i * (T + 1) + s) * SUMM_STRIDE + c];
    return SMC_OK;
}

int smc_filter_logLt(smc_filter* f, double* out_host)
{
    SMC_REQUIRE(f && out_host, "null argument");
    const i64 t = f->t_host, T = f->a.T;
    for (int i = 0; i < f->a.n_islands; ++i) out_host[i] = 0.0;
    if (t == 0) return SMC_OK;
    for (int i = 0; i < f->a.n_islands; ++i) {
        const double* src = f->a.summ + ((size_t)i * (T + 1) + (t - 1)) * SUMM_STRIDE + 3;
        SMC_HIP_CHECK(hipMemcpyAsync(out_host + i, src, 8, hipMemcpyDeviceToHost, f->ctx->stream));
    }
    SMC_HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
    return SMC_OK;
}

int smc_filter_get(smc_filter* f, int field, int island, void* out_host)
{
    SMC_REQUIRE(f && out_host, "null argument");
    SMC_REQUIRE(island >= 0 && island < f->a.n_islands, "island out of range");
    const i64 t = f->t_host, N = f->a.N;
    if (t == 0) {
        smc_set_error("smc_filter_get: no step has run yet");
        return SMC_ERR_STATE;
    }
    hipStream_t st = f->ctx->stream;
    const int cur = (int)((t - 1) & 1);
    const double* X = (cur ? f->a.X1 : f->a.X0) + (size_t)island * N;
    const double* Xo = (cur ? f->a.X0 : f->a.X1) + (size_t)island * N;
    const double* lw = (cur ? f->a.lw1 : f->a.lw0) + (size_t)island * N;
    const i64* A = f->a.A + (size_t)island * N;
    const void* src = nullptr;
    const unsigned nb = (unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK);
    switch (field) {
    case SMC_FIELD_X: src = X; break;
    case SMC_FIELD_LW: src = lw; break;
    case SMC_FIELD_A:
        if (t < 2) { smc_set_error("smc_filter_get: A is undefined before step 1"); return SMC_ERR_STATE; }
        src = A;
        break;
    case SMC_FIELD_XP:
        if (t < 2) { smc_set_error("smc_filter_get: Xp is undefined before step 1"); return SMC_ERR_STATE; }
        SMC_LAUNCH(k_f_gather1, dim3(nb), dim3(SMC_BLOCK), st, Xo, A, N, f->tmp);
        src = f->tmp;
        break;
    case SMC_FIELD_W: {
        const double* row = f->a.summ + ((size_t)island * (f->a.T + 1) + (t - 1)) * SUMM_STRIDE;
        SMC_LAUNCH(k_f_write_W, dim3(nb), dim3(SMC_BLOCK), st, lw, N, row, f->tmp);
        src = f->tmp;
        break;
    }
    default:
        smc_set_error("smc_filter_get: unknown field %d", field);
        return SMC_ERR_INVALID;
    }
    SMC_LAUNCH_CHECK();
    SMC_HIP_CHECK(hipMemcpyAsync(out_host, src, (size_t)N * 8, hipMemcpyDeviceToHost, st));
    SMC_HIP_CHECK(hipStreamSynchronize(st));
    return SMC_OK;
}

int smc_filter_info(smc_filter* f, double* bytes_per_particle_step, int* kernels_per_step)
{
    SMC_REQUIRE(f, "null filter");
    if (bytes_per_particle_step) *bytes_per_particle_step = 16.0 * 1 + 40.0;   // SURVEY 8d, d = 1
    if (kernels_per_step) *kernels_per_step = (f->a.scheme == SMC_MULTINOMIAL && !f->a.ut) ? 4 : 2;
    return SMC_OK;
}

int smc_filter_profile(smc_filter* f, int enable)
{
    SMC_REQUIRE(f, "null filter");
    f->prof = enable != 0;
    f->prof_n = 0;
    if (f->prof && f->ev.empty()) {
        f->ev.resize(3 * PROF_MAX);
        for (auto& e : f->ev) SMC_HIP_CHECK(hipEventCreate(&e));
    }
    return SMC_OK;
}

int smc_filter_kernel_ms(smc_filter* f, double* move_ms_avg, double* prepare_ms_avg,
                         int64_t* n_samples)
{
    SMC_REQUIRE(f && move_ms_avg && prepare_ms_avg && n_samples, "null argument");
    SMC_HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
    double mv = 0.0, pr = 0.0;
    for (int k = 0; k < f->prof_n; ++k) {
        float a = 0.f, b = 0.f;
        SMC_HIP_CHECK(hipEventElapsedTime(&a, f->ev[3 * k], f->ev[3 * k + 1]));
        SMC_HIP_CHECK(hipEventElapsedTime(&b, f->ev[3 * k + 1], f->ev[3 * k + 2]));
        pr += a;
        mv += b;
    }
    *n_samples = f->prof_n;
    *move_ms_avg = f->prof_n ? mv / f->prof_n : 0.0;
    *prepare_ms_avg = f->prof_n ? pr / f->prof_n : 0.0;
    f->prof_n = 0;
    return SMC_OK;
}

}  // extern "C"
