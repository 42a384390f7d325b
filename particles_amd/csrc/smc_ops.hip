// smc_ops.hip -- stand-alone device operators behind the reference's L0 API:
// Weights / log-sum-exp (resampling.py:138-338), inverse_cdf and the three
// resampling schemes (resampling.py:477-610), the ancestor gather
// (core.py:332) and Normal / MvNormal rvs + logpdf (distributions.py:262-285,
// 888-969).  The fused step loop lives in smc_filter.hip and reuses the same
// device code (smc_device.h, smc_resample.h).
#include <vector>

#include "smc_internal.h"
#include "smc_resample.h"
#include "smc_seqx.h"

#define OPS_IPT 4
#define OPS_TILE (SMC_BLOCK * OPS_IPT)
#define LSE_CHUNK (SMC_BLOCK * 8)

// ===========================================================================
// a-4  Weights.__init__  (resampling.py:217-226)
// ===========================================================================
__global__ void __launch_bounds__(SMC_BLOCK)
k_lse_partials(double* lw, i64 N, int fix_nan, double* pm, double* ps, double* pss)
{
    __shared__ double sm[SMC_SM];
    const i64 base = (i64)blockIdx.x * LSE_CHUNK;
    SmcLse acc = smc_lse_empty();
    for (int k = 0; k < LSE_CHUNK / SMC_BLOCK; ++k) {
        const i64 i = base + (i64)k * SMC_BLOCK + threadIdx.x;
        if (i < N) {
            double v = lw[i];
            if (v != v) {                       // resampling.py:220
                v = -INFINITY;
                if (fix_nan) lw[i] = v;
            }
            smc_lse_push(acc, v);
        }
    }
    const SmcLse r = smc_lse_block(acc, sm);
    if (threadIdx.x == 0) {
        pm[blockIdx.x] = r.m;
        ps[blockIdx.x] = r.s;
        pss[blockIdx.x] = r.ss;
    }
}

// scal[0..4) = log_mean, ESS, m, s
__global__ void __launch_bounds__(SMC_BLOCK)
k_lse_finalize(const double* pm, const double* ps, const double* pss, int nparts, i64 N,
               double* scal)
{
    __shared__ double sm[SMC_SM];
    const SmcLse r = smc_lse_reduce_partials(pm, ps, pss, nparts, sm);
    if (threadIdx.x == 0) {
        double log_mean, ess, s = r.s;
        if (!(r.m > -INFINITY) || !(r.m < INFINITY)) {
            // all weights -inf (or a +inf): the reference yields NaN throughout
            log_mean = ess = s = NAN;
        } else {
            log_mean = r.m + log(r.s / (double)N);        // resampling.py:224
            ess = (r.s * r.s) / r.ss;                     // == 1/sum(W^2), :226
        }
        scal[0] = log_mean;
        scal[1] = ess;
        scal[2] = r.m;
        scal[3] = s;
    }
}

__global__ void __launch_bounds__(SMC_BLOCK)
k_lse_write_W(const double* lw, i64 N, const double* scal, double* W)
{
    const double m = scal[2], s = scal[3];
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < N) {
        double v = lw[i];
        if (v != v) v = -INFINITY;
        W[i] = smc_exp_nonpos(v - m) / s;                 // resampling.py:222,225
    }
}

static int lse_launch(smc_ctx* ctx, double* lw, i64 N, int fix_nan, double** scal_out)
{
    const int nparts = (int)((N + LSE_CHUNK - 1) / LSE_CHUNK);
    void* scr;
    int rc = smc_scratch(ctx, (size_t)(3 * nparts + 8) * sizeof(double), &scr);
    if (rc) return rc;
    double* pm = (double*)scr;
    double* ps = pm + nparts;
    double* pss = ps + nparts;
    double* scal = pss + nparts;
    SMC_LAUNCH(k_lse_partials, dim3(nparts), dim3(SMC_BLOCK), ctx->stream, lw, N, fix_nan, pm, ps,
               pss);
    SMC_LAUNCH(k_lse_finalize, dim3(1), dim3(SMC_BLOCK), ctx->stream, (const double*)pm,
               (const double*)ps, (const double*)pss, nparts, N, scal);
    SMC_LAUNCH_CHECK();
    *scal_out = scal;
    return SMC_OK;
}

extern "C" int smc_lse_normalise(smc_ctx* ctx, double* lw, int64_t N, double* W,
                                 double* out4_host)
{
    SMC_REQUIRE(ctx && lw && out4_host, "null argument");
    SMC_REQUIRE(N > 0, "N must be positive");
    double* scal;
    int rc = lse_launch(ctx, lw, N, 1, &scal);
    if (rc) return rc;
    if (W) {
        SMC_LAUNCH(k_lse_write_W, dim3((unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK)),
                   dim3(SMC_BLOCK), ctx->stream, (const double*)lw, (i64)N, (const double*)scal, W);
        SMC_LAUNCH_CHECK();
    }
    double h[4];
    SMC_HIP_CHECK(hipMemcpyAsync(h, scal, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 4; ++i) out4_host[i] = h[i];
    return SMC_OK;
}

// ===========================================================================
// weighted moments: log_mean_exp(v, W) (:291-317), wmean_and_var (:320-338)
// ===========================================================================
// out[0] = sum_i W_i f(v_i), out[1] = sum_i W_i f(v_i)^2 per column; one
// workgroup per column strip, deterministic tree.
__global__ void __launch_bounds__(SMC_BLOCK)
k_wsum_partials(const double* W, const double* X, i64 N, i64 d, int mode, double shift,
                double* part /* (nblk, d, 3) */)
{
    __shared__ double sm[SMC_SM];
    const i64 base = (i64)blockIdx.x * LSE_CHUNK;
    for (i64 c = 0; c < d; ++c) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        for (int k = 0; k < LSE_CHUNK / SMC_BLOCK; ++k) {
            const i64 i = base + (i64)k * SMC_BLOCK + threadIdx.x;
            if (i < N) {
                const double w = W[i];
                double x = X[i * d + c];
                if (mode == 1) x = exp(x - shift);
                a0 += w;
                a1 += w * x;
                a2 += w * (x * x);
            }
        }
        a0 = smc_block_sum(a0, sm);
        a1 = smc_block_sum(a1, sm);
        a2 = smc_block_sum(a2, sm);
        if (threadIdx.x == 0) {
            double* p = part + ((i64)blockIdx.x * d + c) * 3;
            p[0] = a0; p[1] = a1; p[2] = a2;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(SMC_BLOCK)
k_max_partials(const double* v, i64 N, double* pm)
{
    __shared__ double sm[SMC_SM];
    const i64 base = (i64)blockIdx.x * LSE_CHUNK;
    double m = -INFINITY;
    for (int k = 0; k < LSE_CHUNK / SMC_BLOCK; ++k) {
        const i64 i = base + (i64)k * SMC_BLOCK + threadIdx.x;
        if (i < N) m = fmax(m, v[i]);
    }
    m = smc_block_max(m, sm);
    if (threadIdx.x == 0) pm[blockIdx.x] = m;
}

static int wsum(smc_ctx* ctx, const double* W, const double* X, i64 N, i64 d, int mode,
                double shift, std::vector<double>& tot)
{
    const int nblk = (int)((N + LSE_CHUNK - 1) / LSE_CHUNK);
    void* scr;
    int rc = smc_scratch(ctx, (size_t)nblk * d * 3 * sizeof(double), &scr);
    if (rc) return rc;
    SMC_LAUNCH(k_wsum_partials, dim3(nblk), dim3(SMC_BLOCK), ctx->stream, W, X, N, d, mode, shift,
               (double*)scr);
    SMC_LAUNCH_CHECK();
    std::vector<double> h((size_t)nblk * d * 3);
    SMC_HIP_CHECK(hipMemcpyAsync(h.data(), scr, h.size() * sizeof(double), hipMemcpyDeviceToHost,
                                 ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    tot.assign((size_t)d * 3, 0.0);
    for (int b = 0; b < nblk; ++b)
        for (i64 c = 0; c < d * 3; ++c) tot[c] += h[(size_t)b * d * 3 + c];
    return SMC_OK;
}

extern "C" int smc_wmean_var(smc_ctx* ctx, const double* W, const double* X, int64_t N, int64_t d,
                             double* out_host)
{
    SMC_REQUIRE(ctx && W && X && out_host, "null argument");
    SMC_REQUIRE(N > 0 && d > 0, "bad shape");
    std::vector<double> t;
    int rc = wsum(ctx, W, X, N, d, 0, 0.0, t);
    if (rc) return rc;
    for (i64 c = 0; c < d; ++c) {
        // np.average(x, weights=W) = sum(W x)/sum(W)   (resampling.py:335-337)
        const double m = t[c * 3 + 1] / t[c * 3], m2 = t[c * 3 + 2] / t[c * 3];
        out_host[c] = m;
        out_host[d + c] = m2 - m * m;
    }
    return SMC_OK;
}

// partial sums of w (x_i - m_i)(x_j - m_j), i <= j, over a chunk of particles: one workgroup per chunk, thread k
// of pair (i, j) accumulates its particles (d <= 32: at most 528 pairs, two or three per thread)
__global__ void __launch_bounds__(SMC_BLOCK)
k_wcov_partials(const double* W, const double* X, i64 N, int d, const double* mean, double* part)
{
    __shared__ double sX[64 * 33];                             // 64 particles of the chunk at a time, centred
    __shared__ double sW[64];
    __shared__ double smean[32];
    const int tid = (int)threadIdx.x;
    if (tid < d) smean[tid] = mean[tid];
    const int npair = d * (d + 1) / 2;
    double acc[3] = {0.0, 0.0, 0.0};
    int pi[3], pj[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {                              // pair index -> (i, j), i <= j, row-major upper triangle
        int p = tid + r * SMC_BLOCK, i = 0;
        if (p >= npair) p = 0;
        while (p >= d - i) { p -= d - i; ++i; }
        pi[r] = i;
        pj[r] = i + p;
    }
    const i64 base = (i64)blockIdx.x * LSE_CHUNK;
    for (int c0 = 0; c0 < LSE_CHUNK; c0 += 64) {
        __syncthreads();
        for (int e = tid; e < 64 * d; e += SMC_BLOCK) {
            const int n = e / d, c = e - n * d;
            const i64 g = base + c0 + n;
            sX[n * 33 + c] = g < N ? X[g * d + c] - smean[c] : 0.0;
        }
        if (tid < 64) sW[tid] = base + c0 + tid < N ? W[base + c0 + tid] : 0.0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (tid + r * SMC_BLOCK >= npair) continue;
            double a = acc[r];
            for (int n = 0; n < 64; ++n) a += sW[n] * (sX[n * 33 + pi[r]] * sX[n * 33 + pj[r]]);
            acc[r] = a;
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
        if (tid + r * SMC_BLOCK < npair) part[(i64)blockIdx.x * npair + tid + r * SMC_BLOCK] = acc[r];
}

// (mean, cov) of weighted data: np.average(x, weights=W, axis=0) and np.cov(x.T, aweights=W, ddof=0)
// (resampling.py:341-358): out_host = mean (d) | cov (d, d) row-major.  d <= 32.
extern "C" int smc_wmean_cov(smc_ctx* ctx, const double* W, const double* X, int64_t N, int64_t d, double* out_host)
{
    SMC_REQUIRE(ctx && W && X && out_host, "null argument");
    SMC_REQUIRE(N > 0 && d > 0 && d <= 32, "bad shape (d <= 32)");
    std::vector<double> t;
    int rc = wsum(ctx, W, X, N, d, 0, 0.0, t);
    if (rc) return rc;
    const double wtot = t[0];
    for (i64 c = 0; c < d; ++c) out_host[c] = t[c * 3 + 1] / t[c * 3];
    const int nblk = (int)((N + LSE_CHUNK - 1) / LSE_CHUNK), npair = (int)(d * (d + 1) / 2);
    void* scr;
    rc = smc_scratch(ctx, (size_t)(nblk * npair + 32) * sizeof(double), &scr);
    if (rc) return rc;
    double* dmean = (double*)scr + (size_t)nblk * npair;
    SMC_HIP_CHECK(hipMemcpyAsync(dmean, out_host, (size_t)d * 8, hipMemcpyHostToDevice, ctx->stream));
    SMC_LAUNCH(k_wcov_partials, dim3(nblk), dim3(SMC_BLOCK), ctx->stream, W, X, (i64)N, (int)d, (const double*)dmean, (double*)scr);
    SMC_LAUNCH_CHECK();
    std::vector<double> h((size_t)nblk * npair);
    SMC_HIP_CHECK(hipMemcpyAsync(h.data(), scr, h.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    int p = 0;
    for (int i = 0; i < (int)d; ++i)
        for (int j = i; j < (int)d; ++j, ++p) {
            double a = 0.0;
            for (int b = 0; b < nblk; ++b) a += h[(size_t)b * npair + p];
            out_host[d + i * d + j] = out_host[d + j * d + i] = a / wtot;
        }
    return SMC_OK;
}

extern "C" int smc_log_wmean_exp(smc_ctx* ctx, const double* v, const double* W, int64_t N,
                                 double* out_host)
{
    SMC_REQUIRE(ctx && v && W && out_host, "null argument");
    SMC_REQUIRE(N > 0, "N must be positive");
    const int nblk = (int)((N + LSE_CHUNK - 1) / LSE_CHUNK);
    void* scr;
    int rc = smc_scratch(ctx, (size_t)nblk * sizeof(double), &scr);
    if (rc) return rc;
    SMC_LAUNCH(k_max_partials, dim3(nblk), dim3(SMC_BLOCK), ctx->stream, v, (i64)N, (double*)scr);
    SMC_LAUNCH_CHECK();
    std::vector<double> pm(nblk);
    SMC_HIP_CHECK(hipMemcpyAsync(pm.data(), scr, nblk * sizeof(double), hipMemcpyDeviceToHost,
                                 ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    double m = -INFINITY;
    for (double x : pm) m = x > m ? x : m;
    std::vector<double> t;
    rc = wsum(ctx, W, v, N, 1, 1, m, t);
    if (rc) return rc;
    *out_host = m + log(t[1] / t[0]);                     // resampling.py:312-317
    return SMC_OK;
}

// ===========================================================================
// a-5 / a-6  inverse_cdf and the resampling schemes
// ===========================================================================
__global__ void __launch_bounds__(SMC_BLOCK)
k_q62_tile_sums(const double* W, i64 N, u64* Q)
{
    __shared__ u64 sm[SMC_SM];
    const i64 j0 = (i64)blockIdx.x * OPS_TILE + (i64)threadIdx.x * OPS_IPT;
    u64 t = 0;
#pragma unroll
    for (int i = 0; i < OPS_IPT; ++i)
        if (j0 + i < N) t += smc_q62_w(W[j0 + i]);
    t = smc_block_sum_u64(t, sm);
    if (threadIdx.x == 0) Q[blockIdx.x] = t;
}

__device__ __forceinline__ SmcSu smc_su_prepare(SmcSu su)
{
    if (su.scheme == SMC_SYSTEMATIC_) {
        if (su.u) {
            su.u_sys = su.u[0];
        } else {
            u64 a, b;
            smc_philox_uniform(0u, su.t, su.island, SMC_STREAM_RESAMPLE, su.seed, a, b);
            su.u_sys = smc_u01_halfopen(a);
        }
    }
    return su;
}

__global__ void __launch_bounds__(SMC_BLOCK)
k_resample_tiles(const double* W, i64 N, SmcSu su_in, const u64* Q, int ntiles, i64* A)
{
    __shared__ u64 sC[OPS_TILE];
    __shared__ u64 sm[SMC_SM];
    __shared__ i64 sn[2];
    const SmcSu su = smc_su_prepare(su_in);
    const int b = (int)blockIdx.x;
    const i64 j0 = (i64)b * OPS_TILE;
    u64 wq[OPS_IPT];
#pragma unroll
    for (int i = 0; i < OPS_IPT; ++i) {
        const i64 j = j0 + (i64)threadIdx.x * OPS_IPT + i;
        wq[i] = (j < N) ? smc_q62_w(W[j]) : 0ull;
    }
    u64 total;
    const u64 pre = smc_tile_cdf<OPS_IPT>(wq, Q, b, sC, sm, total);
    i64 n_lo, n_hi;
    smc_tile_outputs(su, b, ntiles, pre, total, sn, n_lo, n_hi);
    const int nvalid = (int)((N - j0 < OPS_TILE) ? (N - j0) : OPS_TILE);
    for (i64 p = (n_lo >> 1) + threadIdx.x; 2 * p < n_hi; p += SMC_BLOCK) {
        double s0, s1;
        smc_su_pair(su, p, s0, s1);
        const i64 n0 = 2 * p, n1 = n0 + 1;
        if (n0 >= n_lo && n0 < n_hi) {
            int jl = smc_lower_bound_u64(sC, OPS_TILE, smc_q62_t(s0));
            jl = jl < nvalid ? jl : nvalid - 1;
            A[n0] = j0 + jl;
        }
        if (n1 >= n_lo && n1 < n_hi) {
            int jl = smc_lower_bound_u64(sC, OPS_TILE, smc_q62_t(s1));
            jl = jl < nvalid ? jl : nvalid - 1;
            A[n1] = j0 + jl;
        }
    }
}

static int resample_launch(smc_ctx* ctx, const double* W, i64 N, SmcSu su, i64* A)
{
    const int ntiles = (int)((N + OPS_TILE - 1) / OPS_TILE);
    void* scr;
    int rc = smc_scratch(ctx, (size_t)ntiles * sizeof(u64), &scr);
    if (rc) return rc;
    SMC_LAUNCH(k_q62_tile_sums, dim3(ntiles), dim3(SMC_BLOCK), ctx->stream, W, N, (u64*)scr);
    SMC_LAUNCH(k_resample_tiles, dim3(ntiles), dim3(SMC_BLOCK), ctx->stream, W, N, su,
               (const u64*)scr, ntiles, A);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

extern "C" int smc_inverse_cdf(smc_ctx* ctx, const double* su_dev, const double* W, int64_t M,
                               int64_t N, int64_t* A)
{
    SMC_REQUIRE(ctx && su_dev && W && A, "null argument");
    SMC_REQUIRE(M > 0 && N > 0, "M and N must be positive");
    SmcSu su{};                  // zeros, and the "no cached pair" mark of its default initialiser
    su.scheme = SMC_MULTINOMIAL_;       // "sorted uniforms given in memory"
    su.M = M;
    su.dM = (double)M;
    su.u = su_dev;
    return resample_launch(ctx, W, N, su, (i64*)A);
}

// S (n) <- sequential fp64 prefix sums of W (n); grid (1, islands) x 64 threads, arrays island-major
__global__ void __launch_bounds__(64)
k_seq_cdf(const double* W, const i64 n, double* S)
{
    const int lane = (int)threadIdx.x;
    const double* w = W + (i64)blockIdx.y * n;
    double* o = S + (i64)blockIdx.y * n;
    double s = 0.0;
    bool first = true;
    for (i64 c = 0; c < n; c += 64) {
        const i64 i = c + lane;
        const double wi = i < n ? w[i] : 0.0;
        double mine = 0.0;
        const int m = (int)(n - c < 64 ? n - c : 64);
        for (int k = 0; k < m; ++k) {
            const double wk = smc_readlane_f64(wi, k);
            s = first ? wk : s + wk;                            // s = W[0], then s += W[j]
            first = false;
            if (lane == k) mine = s;
        }
        if (i < n) o[i] = mine;
    }
}
__global__ void __launch_bounds__(SMC_BLOCK)
k_search_strict(const double* su, const double* S, i64 M, i64 N, i64* A)
{
    const i64 n = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (n < M) A[n] = smc_first_ge(S, N, su[n]);
}

// the two launches of smc_seqx.h on an array of weights (+ the pass that estimates the tiles' sums)
static void sqx_launch_array(hipStream_t st, const double* W, const i64 N, void* scr, SqxArgs& q, double** tsum_out = nullptr)
{
    q = sqx_carve(scr, N, 1);
    size_t used = 0;
    (void)sqx_carve(scr, N, 1, &used);
    double* tsum = (double*)((char*)scr + used);
    sqx_zero_counters(st, scr, N, 1);
    const SeqGate gate{nullptr, 0, 0, nullptr};
    SMC_LAUNCH(k_seq_tile_sums, dim3(q.ntiles, 1), dim3(SMC_BLOCK), st, W, N, tsum, gate);
    SMC_LAUNCH(k_sqx_classify, dim3(q.ntiles, 1), dim3(SMC_BLOCK), st, W, (const double*)tsum, q, gate);
    if (tsum_out) *tsum_out = tsum;
}
static size_t sqx_array_scratch(const i64 N) { return sqx_scratch_bytes(N, 1) + (size_t)((N + SEQ_TILE - 1) / SEQ_TILE) * 8 + 64; }

// S <- prefix sums of W as the reference's loop forms them (test hook: mode 0 the two-launch emulation of smc_seqx.h,
// mode 2 the tile walk of smc_seqsum.h, mode 1 the literal one-lane walk both must equal bit for bit).
// n_sequential_tiles: mode 2: tiles the walk did exactly; mode 0: -1 if the island took the exact path, else the number of
// exceptions the walk handled.
extern "C" int smc_seq_prefix_sums(smc_ctx* ctx, const double* W, int64_t N, double* S, int mode, int64_t* n_sequential_tiles)
{
    SMC_REQUIRE(ctx && W && S, "null argument");
    if (n_sequential_tiles) *n_sequential_tiles = (N + SEQ_TILE - 1) / SEQ_TILE;
    SMC_REQUIRE(N > 0 && N < ((int64_t)1 << 32), "N must be positive (and below 2^32)");
    SMC_HIP_CHECK(hipSetDevice(ctx->device));
    if (mode == 1) {
        SMC_LAUNCH(k_seq_cdf, dim3(1, 1), dim3(64), ctx->stream, W, (i64)N, S);
    } else if (mode == 2) {
        void* scr;
        int rc = smc_scratch(ctx, seq_scratch_bytes((i64)N, 1), &scr);
        if (rc) return rc;
        seq_tile_walk_launch(ctx->stream, W, (i64)N, 1, S, scr);
        if (n_sequential_tiles) {
            unsigned long long c = 0ull;
            SMC_HIP_CHECK(hipMemcpyAsync(&c, seq_nseq_ptr(scr, (i64)N, 1), 8, hipMemcpyDeviceToHost, ctx->stream));
            SMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            *n_sequential_tiles = (int64_t)c;
        }
    } else {
        void* scr;
        int rc = smc_scratch(ctx, sqx_array_scratch((i64)N), &scr);
        if (rc) return rc;
        SqxArgs q;
        sqx_launch_array(ctx->stream, W, (i64)N, scr, q);
        SMC_LAUNCH(k_sqx_fill, dim3(q.ntiles, 1), dim3(SMC_BLOCK), ctx->stream, q, S, SeqGate{nullptr, 0, 0, nullptr});
        if (n_sequential_tiles) {
            unsigned long long c[2] = {0ull, 0ull};
            SMC_HIP_CHECK(hipMemcpyAsync(c, q.ctr + 2, 16, hipMemcpyDeviceToHost, ctx->stream));
            SMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            *n_sequential_tiles = c[0] ? -(int64_t)c[0] : (int64_t)c[1];   // (< 0: the exact path ran, -why)
        }
    }
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

// inverse_cdf exactly as the reference computes it (sequential fp64 CDF, resampling.py:500-509): the sums by the two
// launches of smc_seqx.h, never written out -- every tile of parents finds its offspring among the sorted uniforms
extern "C" int smc_inverse_cdf_strict(smc_ctx* ctx, const double* su_dev, const double* W, int64_t M,
                                      int64_t N, int64_t* A)
{
    SMC_REQUIRE(ctx && su_dev && W && A, "null argument");
    SMC_REQUIRE(M > 0 && N > 0 && N < ((int64_t)1 << 32), "M and N must be positive (N below 2^32)");
    SMC_HIP_CHECK(hipSetDevice(ctx->device));
    void* scr;
    int rc = smc_scratch(ctx, sqx_array_scratch((i64)N), &scr);
    if (rc) return rc;
    SqxArgs q;
    sqx_launch_array(ctx->stream, W, (i64)N, scr, q);
    SMC_LAUNCH(k_sqx_search_sorted, dim3(q.ntiles, 1), dim3(SMC_BLOCK), ctx->stream, q, su_dev, (i64)M, (i64*)A);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

// ---- element-wise arithmetic for device-resident model code -------------------
// A user-defined Feynman-Kac model (core.py:108-197: M0 / M / logG written with numpy
// expressions on xp, x) keeps working when its arrays live in HBM: DeviceArray routes
// + - * / ** and the numpy ufuncs it is used with through this one kernel.  + - * / sqrt
// are IEEE-exact like numpy's; exp / log / sin / cos are the device libm's.
enum { EW_ADD = 0, EW_SUB, EW_MUL, EW_DIV, EW_RSUB, EW_RDIV, EW_NEG, EW_EXP, EW_LOG, EW_SQRT,
       EW_COS, EW_SIN, EW_ABS, EW_SQUARE, EW_POW, EW_MIN, EW_MAX, EW_ARCTAN, EW_COUNT };
__global__ void __launch_bounds__(SMC_BLOCK)
k_elementwise(int op, const double* a, i64 sa, const double* b, i64 sb, double alpha, i64 n,
              double* out)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i >= n) return;
    // stride 1: element i; 0: one value for all; -m: element i mod m (a row vector against (N, m))
    const double x = a[sa >= 0 ? i * sa : i % (-sa)];
    const double y = b ? b[sb >= 0 ? i * sb : i % (-sb)] : alpha;
    double r;
    switch (op) {
    case EW_ADD: r = x + y; break;
    case EW_SUB: r = x - y; break;
    case EW_MUL: r = x * y; break;
    case EW_DIV: r = x / y; break;
    case EW_RSUB: r = y - x; break;
    case EW_RDIV: r = y / x; break;
    case EW_NEG: r = -x; break;
    case EW_EXP: r = exp(x); break;
    case EW_LOG: r = log(x); break;
    case EW_SQRT: r = sqrt(x); break;
    case EW_COS: r = cos(x); break;
    case EW_SIN: r = sin(x); break;
    case EW_ABS: r = fabs(x); break;
    case EW_SQUARE: r = x * x; break;
    case EW_POW: r = pow(x, y); break;
    case EW_MIN: r = (x != x || y != y) ? NAN : fmin(x, y); break;      // np.minimum propagates NaN
    case EW_MAX: r = (x != x || y != y) ? NAN : fmax(x, y); break;
    default: r = atan(x); break;
    }
    out[i] = r;
}

extern "C" int smc_elementwise(smc_ctx* ctx, int op, const double* a, int64_t stride_a,
                               const double* b, int64_t stride_b, double alpha, int64_t n,
                               double* out)
{
    SMC_REQUIRE(ctx && a && out, "null argument");
    SMC_REQUIRE(op >= 0 && op < EW_COUNT, "unknown element-wise operation");
    SMC_REQUIRE(stride_a <= 1 && stride_b <= 1, "strides must be 1, 0 (broadcast) or -m (row vector)");
    if (n <= 0) return SMC_OK;
    SMC_LAUNCH(k_elementwise, dim3((unsigned)((n + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK),
               ctx->stream, op, a, (i64)stride_a, b, (i64)stride_b, alpha, (i64)n, out);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

// out (N,k) = X (N,d) @ M (d,k), M small and shared (np.dot(xp, F.T) of a model's PX / PY)
__global__ void __launch_bounds__(SMC_BLOCK)
k_rows_matmul(const double* X, i64 N, int d, const double* M, int k, double* out)
{
    __shared__ double sM[64 * 64];
    for (int i = (int)threadIdx.x; i < d * k; i += SMC_BLOCK) sM[i] = M[i];
    __syncthreads();
    const i64 idx = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (idx >= N * k) return;
    const i64 n = idx / k;
    const int j = (int)(idx - n * k);
    double acc = 0.0;
    for (int i = 0; i < d; ++i) acc = fma(X[n * d + i], sM[i * k + j], acc);
    out[idx] = acc;
}

extern "C" int smc_rows_matmul(smc_ctx* ctx, const double* X, int64_t N, int64_t d,
                               const double* M_host, int64_t k, double* out)
{
    SMC_REQUIRE(ctx && X && M_host && out, "null argument");
    SMC_REQUIRE(N > 0 && d > 0 && d <= 64 && k > 0 && k <= 64, "need 0 < d, k <= 64");
    void* scr;
    int rc = smc_scratch(ctx, (size_t)(d * k) * 8, &scr);
    if (rc) return rc;
    SMC_HIP_CHECK(hipMemcpyAsync(scr, M_host, (size_t)(d * k) * 8, hipMemcpyHostToDevice, ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));      // M_host may be a temporary
    const i64 total = (i64)N * k;
    SMC_LAUNCH(k_rows_matmul, dim3((unsigned)((total + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK),
               ctx->stream, X, (i64)N, (int)d, (const double*)scr, (int)k, out);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

__global__ void __launch_bounds__(SMC_BLOCK)
k_copy_strided(const double* src, i64 ss, double* dst, i64 ds, i64 n)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < n) dst[i * ds] = src[i * ss];
}

extern "C" int smc_copy_strided(smc_ctx* ctx, const double* src, int64_t src_stride, double* dst,
                                int64_t dst_stride, int64_t n)
{
    SMC_REQUIRE(ctx && src && dst, "null argument");
    SMC_REQUIRE(n > 0 && src_stride >= 1 && dst_stride >= 1, "n and the strides must be positive");
    SMC_LAUNCH(k_copy_strided, dim3((unsigned)((n + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK),
               ctx->stream, src, (i64)src_stride, dst, (i64)dst_stride, (i64)n);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

// ---- residual resampling (resampling.py:611-626) ----------------------------
// A[:sip] = arange(N).repeat(floor(M W)) is an inverse CDF on INTEGER weights with the
// thresholds n+1; the remaining M - sip draws are a multinomial on (M W - floor(M W)) /
// (M - sip), i.e. smc_inverse_cdf on that vector.
__device__ __forceinline__ u64 res_count(double w, double dM)
{
    const double c = floor(dM * w);                       // np.floor(M * W)
    return c > 0.0 ? (u64)c : 0ull;
}
__global__ void __launch_bounds__(SMC_BLOCK)
k_residual_counts(const double* W, i64 N, double dM, u64* Q)
{
    __shared__ u64 sm[SMC_SM];
    const i64 j0 = (i64)blockIdx.x * OPS_TILE + (i64)threadIdx.x * OPS_IPT;
    u64 t = 0;
#pragma unroll
    for (int i = 0; i < OPS_IPT; ++i)
        if (j0 + i < N) t += res_count(W[j0 + i], dM);
    t = smc_block_sum_u64(t, sm);
    if (threadIdx.x == 0) Q[blockIdx.x] = t;
}
__global__ void __launch_bounds__(SMC_BLOCK)
k_residual_split(const double* W, i64 N, double dM, double dsres, double* r)
{
    const i64 j = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (j < N) {
        const double mw = dM * W[j];
        r[j] = (mw - floor(mw)) / dsres;                  // res / sres
    }
}
__global__ void __launch_bounds__(SMC_BLOCK)
k_repeat_tiles(const double* W, i64 N, double dM, const u64* Q, i64* A)
{
    __shared__ u64 sC[OPS_TILE];
    __shared__ u64 sm[SMC_SM];
    const int b = (int)blockIdx.x;
    const i64 j0 = (i64)b * OPS_TILE;
    u64 c[OPS_IPT];
#pragma unroll
    for (int i = 0; i < OPS_IPT; ++i) {
        const i64 j = j0 + (i64)threadIdx.x * OPS_IPT + i;
        c[i] = (j < N) ? res_count(W[j], dM) : 0ull;
    }
    u64 total;
    const u64 pre = smc_tile_cdf<OPS_IPT>(c, Q, b, sC, sm, total);
    for (u64 n = pre + threadIdx.x; n < pre + total; n += SMC_BLOCK)       // its copies
        A[n] = j0 + smc_lower_bound_u64(sC, OPS_TILE, n + 1ull);
}

extern "C" int smc_residual_split(smc_ctx* ctx, const double* W, int64_t N, int64_t M,
                                  double* r_dev, int64_t* sip_host)
{
    SMC_REQUIRE(ctx && W && r_dev && sip_host, "null argument");
    SMC_REQUIRE(M > 0 && N > 0, "M and N must be positive");
    const int ntiles = (int)((N + OPS_TILE - 1) / OPS_TILE);
    void* scr;
    int rc = smc_scratch(ctx, (size_t)ntiles * sizeof(u64), &scr);
    if (rc) return rc;
    SMC_LAUNCH(k_residual_counts, dim3(ntiles), dim3(SMC_BLOCK), ctx->stream, W, (i64)N, (double)M,
               (u64*)scr);
    std::vector<u64> h(ntiles);
    SMC_HIP_CHECK(hipMemcpyAsync(h.data(), scr, (size_t)ntiles * 8, hipMemcpyDeviceToHost, ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    u64 sip = 0;
    for (u64 v : h) sip += v;
    *sip_host = (int64_t)sip;
    const int64_t sres = M - (int64_t)sip;
    if (sres > 0)
        SMC_LAUNCH(k_residual_split, dim3((unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK),
                   ctx->stream, W, (i64)N, (double)M, (double)sres, r_dev);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

extern "C" int smc_residual_ancestors(smc_ctx* ctx, const double* W, const double* r_dev, int64_t N,
                                      int64_t M, int64_t sip, const double* su_dev, int64_t* A)
{
    SMC_REQUIRE(ctx && W && A, "null argument");
    SMC_REQUIRE(M > 0 && N > 0 && sip >= 0 && sip <= M, "sizes out of range");
    const int ntiles = (int)((N + OPS_TILE - 1) / OPS_TILE);
    void* scr;
    int rc = smc_scratch(ctx, (size_t)ntiles * sizeof(u64), &scr);
    if (rc) return rc;
    SMC_LAUNCH(k_residual_counts, dim3(ntiles), dim3(SMC_BLOCK), ctx->stream, W, (i64)N, (double)M,
               (u64*)scr);
    SMC_LAUNCH(k_repeat_tiles, dim3(ntiles), dim3(SMC_BLOCK), ctx->stream, W, (i64)N, (double)M,
               (const u64*)scr, (i64*)A);
    SMC_LAUNCH_CHECK();
    if (M - sip > 0) {
        SMC_REQUIRE(r_dev && su_dev, "residual weights and sorted uniforms are required");
        return smc_inverse_cdf(ctx, su_dev, r_dev, M - sip, N, A + sip);
    }
    return SMC_OK;
}

// ---- SSP resampling (resampling.py:628-678, Gerber, Chopin & Whiteley 2019) ---
// The Srinivasan sampling process is a chain of N-1 pairwise steps, each depending on the
// previous one: there is nothing to run in parallel, so ONE lane walks it on the device
// (the weights stay where they are; N-1 dependent iterations, one fp64 division each).
// IEEE operations only, in the reference's order: same uniforms -> same offspring counts.
#define SSP_CHUNK 512
__global__ void __launch_bounds__(64)
k_ssp_counts(const double* W, const double* u, i64 N, double dM, i64* nr, i64* sum_out)
{
    // One wavefront: all lanes stage the next SSP_CHUNK uniforms and candidates (fractional
    // part and floor of M W) in LDS, lane 0 walks the chain on them.  The two active
    // particles live in registers; a particle's offspring number is final when it leaves
    // the active pair, so nr[] is written once per particle, never read.
    __shared__ double sU[SSP_CHUNK], sX[SSP_CHUNK];
    __shared__ i64 sF[SSP_CHUNK];
    const int lane = (int)threadIdx.x;
    i64 ii = 0, jj = 1, total = 0, fi = 0, fj = 0;
    double xi = 0.0, xj = 0.0;
    if (lane == 0) {
        const double m0 = dM * W[0], f0 = floor(m0);
        fi = (i64)f0; xi = m0 - f0;
        if (N > 1) { const double m1 = dM * W[1], f1 = floor(m1); fj = (i64)f1; xj = m1 - f1; }
    }
    for (i64 k0 = 0; k0 < N - 1; k0 += SSP_CHUNK) {
        const int cnt = (int)((N - 1 - k0 < SSP_CHUNK) ? (N - 1 - k0) : SSP_CHUNK);
        __syncthreads();
        for (int m = lane; m < cnt; m += 64) {
            sU[m] = u[k0 + m];
            const i64 n = k0 + m + 2;                       // the particle that enters after step k
            if (n < N) {
                const double mw = dM * W[n], fl = floor(mw);
                sX[m] = mw - fl;
                sF[m] = (i64)fl;
            } else {
                sX[m] = 0.0;
                sF[m] = 0;
            }
        }
        __syncthreads();
        if (lane == 0) {
#pragma unroll 4
            for (int m = 0; m < cnt; ++m) {
                const i64 k = k0 + m;
                double delta_i = fmin(xj, 1.0 - xi);            // increase i, decrease j
                const double delta_j = fmin(xi, 1.0 - xj);      // the opposite
                const double sum_delta = delta_i + delta_j;
                const double pj = sum_delta > 0.0 ? delta_i / sum_delta : 0.0;
                if (sU[m] < pj) {                               // swap so that we always increase i
                    const i64 ti = ii; ii = jj; jj = ti;
                    const i64 tf = fi; fi = fj; fj = tf;
                    const double tx = xi; xi = xj; xj = tx;
                    delta_i = delta_j;
                }
                if (xj < 1.0 - xi) {
                    xi += delta_i;
                    nr[jj] = fj;                                // j leaves with floor(M W_j) offspring
                    total += fj;
                    jj = k + 2; xj = sX[m]; fj = sF[m];
                } else {
                    xj -= delta_i;
                    nr[ii] = fi + 1;                            // i leaves with one more
                    total += fi + 1;
                    ii = k + 2; xi = sX[m]; fi = sF[m];
                }
            }
        }
    }
    if (lane == 0) {
        // the pair still active (one of the two indices is N when N >= 2: it never existed)
        const bool has_i = ii < N, has_j = N > 1 && jj < N;
        i64 tot = total + (has_i ? fi : 0) + (has_j ? fj : 0);
        if (N >= 2 && tot == (i64)dM - 1) {                     // round-off (resampling.py:669-673)
            const bool last_is_i = (jj == N);                   // last_ij = i if j == k + 2 else j
            const double xl = last_is_i ? xi : xj;
            if (xl > 0.99) { if (last_is_i) ++fi; else ++fj; ++tot; }
        }
        if (has_i) nr[ii] = fi;
        if (has_j) nr[jj] = fj;
        *sum_out = tot;
    }
}
__global__ void __launch_bounds__(SMC_BLOCK)
k_count_tile_sums(const i64* nr, i64 N, u64* Q)
{
    __shared__ u64 sm[SMC_SM];
    const i64 j0 = (i64)blockIdx.x * OPS_TILE + (i64)threadIdx.x * OPS_IPT;
    u64 t = 0;
#pragma unroll
    for (int i = 0; i < OPS_IPT; ++i)
        if (j0 + i < N) t += (u64)nr[j0 + i];
    t = smc_block_sum_u64(t, sm);
    if (threadIdx.x == 0) Q[blockIdx.x] = t;
}
__global__ void __launch_bounds__(SMC_BLOCK)
k_repeat_counts(const i64* nr, i64 N, const u64* Q, i64* A)         // arange(N).repeat(nr)
{
    __shared__ u64 sC[OPS_TILE];
    __shared__ u64 sm[SMC_SM];
    const int b = (int)blockIdx.x;
    const i64 j0 = (i64)b * OPS_TILE;
    u64 c[OPS_IPT];
#pragma unroll
    for (int i = 0; i < OPS_IPT; ++i) {
        const i64 j = j0 + (i64)threadIdx.x * OPS_IPT + i;
        c[i] = (j < N) ? (u64)nr[j] : 0ull;
    }
    u64 total;
    const u64 pre = smc_tile_cdf<OPS_IPT>(c, Q, b, sC, sm, total);
    for (u64 n = pre + threadIdx.x; n < pre + total; n += SMC_BLOCK)
        A[n] = j0 + smc_lower_bound_u64(sC, OPS_TILE, n + 1ull);
}

extern "C" int smc_resample_ssp(smc_ctx* ctx, const double* W, const double* u_dev, int64_t N,
                                int64_t M, int64_t* A)
{
    SMC_REQUIRE(ctx && W && A, "null argument");
    SMC_REQUIRE(N > 0 && M > 0, "M and N must be positive");
    SMC_REQUIRE(N == 1 || u_dev, "ssp needs N - 1 uniforms");
    const int ntiles = (int)((N + OPS_TILE - 1) / OPS_TILE);
    char* buf = nullptr;
    hipError_t e = hipMalloc((void**)&buf, (size_t)N * 8 + 8 + (size_t)ntiles * 8);
    if (e != hipSuccess) {
        smc_set_error("smc_resample_ssp: %zu bytes: %s", (size_t)N * 8, hipGetErrorString(e));
        return SMC_ERR_NOMEM;
    }
    i64* nr = (i64*)buf;
    i64* sum = (i64*)(buf + (size_t)N * 8);
    u64* Q = (u64*)(buf + (size_t)N * 8 + 8);
    SMC_LAUNCH(k_ssp_counts, dim3(1), dim3(64), ctx->stream, W, u_dev, (i64)N, (double)M, nr, sum);
    i64 total = 0;
    hipError_t rc = hipMemcpyAsync(&total, sum, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (rc == hipSuccess) rc = hipStreamSynchronize(ctx->stream);
    int ret = SMC_OK;
    if (rc != hipSuccess) {
        smc_set_error("smc_resample_ssp: %s", hipGetErrorString(rc));
        ret = SMC_ERR_HIP;
    } else if (total != M) {
        // the reference raises ValueError (resampling.py:674-676)
        smc_set_error("ssp resampling: wrong size for output");
        ret = SMC_ERR_INVALID;
    } else {
        SMC_LAUNCH(k_count_tile_sums, dim3(ntiles), dim3(SMC_BLOCK), ctx->stream, (const i64*)nr, (i64)N, Q);
        SMC_LAUNCH(k_repeat_counts, dim3(ntiles), dim3(SMC_BLOCK), ctx->stream, (const i64*)nr, (i64)N,
                   (const u64*)Q, (i64*)A);
        rc = hipStreamSynchronize(ctx->stream);
        if (rc != hipSuccess) { smc_set_error("smc_resample_ssp: %s", hipGetErrorString(rc)); ret = SMC_ERR_HIP; }
    }
    (void)hipFree(buf);
    return ret;
}

// ---- killing resampling (resampling.py:680-697) -----------------------------
// killed_i = u_i * max(W) >= W_i ; A = arange(N) ; A[killed] = multinomial(W, #killed)
__global__ void __launch_bounds__(SMC_BLOCK)
k_wmax_partials(const double* W, i64 N, double* part)
{
    __shared__ double sm[SMC_SM];
    double m = -INFINITY;
    for (i64 j = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x; j < N; j += (i64)gridDim.x * SMC_BLOCK)
        m = smc_max2(m, W[j]);
    m = smc_block_max(m, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = m;
}
__global__ void __launch_bounds__(SMC_BLOCK)
k_killing_flags(const double* W, const double* u, i64 N, const double* part, int nparts,
                unsigned char* killed, u64* Q)
{
    __shared__ u64 sm[SMC_SM];
    double wmax = part[0];
    for (int i = 1; i < nparts; ++i) wmax = smc_max2(wmax, part[i]);
    const i64 j0 = (i64)blockIdx.x * OPS_TILE + (i64)threadIdx.x * OPS_IPT;
    u64 t = 0;
#pragma unroll
    for (int i = 0; i < OPS_IPT; ++i)
        if (j0 + i < N) {
            const bool k = u[j0 + i] * wmax >= W[j0 + i];
            killed[j0 + i] = k ? 1 : 0;
            t += k ? 1ull : 0ull;
        }
    t = smc_block_sum_u64(t, sm);
    if (threadIdx.x == 0) Q[blockIdx.x] = t;
}
__global__ void __launch_bounds__(SMC_BLOCK)
k_flag_tile_sums(const unsigned char* killed, i64 N, u64* Q)
{
    __shared__ u64 sm[SMC_SM];
    const i64 j0 = (i64)blockIdx.x * OPS_TILE + (i64)threadIdx.x * OPS_IPT;
    u64 t = 0;
#pragma unroll
    for (int i = 0; i < OPS_IPT; ++i)
        if (j0 + i < N && killed[j0 + i]) t += 1ull;
    t = smc_block_sum_u64(t, sm);
    if (threadIdx.x == 0) Q[blockIdx.x] = t;
}
__global__ void __launch_bounds__(SMC_BLOCK)
k_killing_assign(const unsigned char* killed, const i64* Am, i64 N, const u64* Q, i64* A)
{
    __shared__ u64 sC[OPS_TILE];
    __shared__ u64 sm[SMC_SM];
    const int b = (int)blockIdx.x;
    const i64 j0 = (i64)b * OPS_TILE + (i64)threadIdx.x * OPS_IPT;
    u64 c[OPS_IPT];
#pragma unroll
    for (int i = 0; i < OPS_IPT; ++i) c[i] = (j0 + i < N && killed[j0 + i]) ? 1ull : 0ull;
    u64 total;
    smc_tile_cdf<OPS_IPT>(c, Q, b, sC, sm, total);         // inclusive ranks of the killed ones
#pragma unroll
    for (int i = 0; i < OPS_IPT; ++i)
        if (j0 + i < N)
            A[j0 + i] = c[i] ? Am[sC[threadIdx.x * OPS_IPT + i] - 1ull] : j0 + i;
}

extern "C" int smc_killing_split(smc_ctx* ctx, const double* W, const double* u_dev, int64_t N,
                                 unsigned char* killed_dev, int64_t* nkilled_host)
{
    SMC_REQUIRE(ctx && W && u_dev && killed_dev && nkilled_host, "null argument");
    SMC_REQUIRE(N > 0, "N must be positive");
    const int ntiles = (int)((N + OPS_TILE - 1) / OPS_TILE);
    const int nparts = ntiles < 64 ? ntiles : 64;
    void* scr;
    int rc = smc_scratch(ctx, (size_t)(ntiles + 64) * sizeof(u64), &scr);
    if (rc) return rc;
    double* part = (double*)scr;
    u64* Q = (u64*)scr + 64;
    SMC_LAUNCH(k_wmax_partials, dim3(nparts), dim3(SMC_BLOCK), ctx->stream, W, (i64)N, part);
    SMC_LAUNCH(k_killing_flags, dim3(ntiles), dim3(SMC_BLOCK), ctx->stream, W, u_dev, (i64)N,
               (const double*)part, nparts, killed_dev, Q);
    std::vector<u64> h(ntiles);
    SMC_HIP_CHECK(hipMemcpyAsync(h.data(), Q, (size_t)ntiles * 8, hipMemcpyDeviceToHost, ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    u64 nk = 0;
    for (u64 v : h) nk += v;
    *nkilled_host = (int64_t)nk;
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

extern "C" int smc_killing_ancestors(smc_ctx* ctx, const unsigned char* killed_dev, const int64_t* Am,
                                     int64_t N, int64_t* A)
{
    SMC_REQUIRE(ctx && killed_dev && A, "null argument");
    SMC_REQUIRE(N > 0, "N must be positive");
    const int ntiles = (int)((N + OPS_TILE - 1) / OPS_TILE);
    void* scr;
    int rc = smc_scratch(ctx, (size_t)ntiles * sizeof(u64), &scr);
    if (rc) return rc;
    // tile totals of the flags (k_residual-style sum over bytes)
    SMC_LAUNCH(k_flag_tile_sums, dim3(ntiles), dim3(SMC_BLOCK), ctx->stream, killed_dev, (i64)N, (u64*)scr);
    SMC_LAUNCH(k_killing_assign, dim3(ntiles), dim3(SMC_BLOCK), ctx->stream, killed_dev, (const i64*)Am,
               (i64)N, (const u64*)scr, (i64*)A);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

// ---- uniform_spacings on the device (resampling.py:512-537) ----------------
// e_n = -log(u_n), n = 0..M, in fixed point so that the running sums are
// exact and the resulting su is monotone:  su[n] = (e_0+..+e_n) / (e_0+..+e_M)
__device__ __forceinline__ u64 smc_spacing_q(u64 seed, u32 t, u32 island, i64 n, double scale)
{
    u64 a, b;
    smc_philox((u32)(n >> 1), t, island, SMC_STREAM_SPACINGS, seed, a, b);
    const double u = smc_u01_open((n & 1) ? b : a);
    return (u64)rint(-log(u) * scale);
}

__global__ void __launch_bounds__(SMC_BLOCK)
k_spacing_tile_sums(i64 M1, double scale, u64 seed, u32 t, u32 island, u64* E)
{
    __shared__ u64 sm[SMC_SM];
    const i64 n0 = (i64)blockIdx.x * OPS_TILE + (i64)threadIdx.x * OPS_IPT;
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < OPS_IPT; ++i)
        if (n0 + i < M1) s += smc_spacing_q(seed, t, island, n0 + i, scale);
    s = smc_block_sum_u64(s, sm);
    if (threadIdx.x == 0) E[blockIdx.x] = s;
}

__global__ void __launch_bounds__(SMC_BLOCK)
k_spacing_write(i64 M, double scale, u64 seed, u32 t, u32 island, const u64* E, int ntiles,
                double* su)
{
    __shared__ u64 sm[SMC_SM];
    const int b = (int)blockIdx.x;
    const i64 n0 = (i64)b * OPS_TILE + (i64)threadIdx.x * OPS_IPT;
    u64 q[OPS_IPT], tsum = 0;
#pragma unroll
    for (int i = 0; i < OPS_IPT; ++i) {
        q[i] = (n0 + i <= M) ? smc_spacing_q(seed, t, island, n0 + i, scale) : 0ull;
        tsum += q[i];
    }
    u64 pre = 0, all = 0;
    for (int i = (int)threadIdx.x; i < ntiles; i += SMC_BLOCK) {
        const u64 e = E[i];
        all += e;
        if (i < b) pre += e;
    }
    pre = smc_block_sum_u64(pre, sm);
    all = smc_block_sum_u64(all, sm);
    u64 tot;
    u64 run = smc_block_exscan_u64(tsum, sm, tot);
    const double dall = (double)all;
#pragma unroll
    for (int i = 0; i < OPS_IPT; ++i) {
        run += q[i];          // (the prefix inside a tile of 1024 saturates at 2^32 - 1: the filter kernels' contract, SP_OFF_MAX)
        if (n0 + i < M) su[n0 + i] = (double)(pre + (run < 0xFFFFFFFFull ? run : 0xFFFFFFFFull)) / dall;
    }
}

// the fixed-point scale of the exponential draws, q_n = rint(-log(u_n) 2^s): s = min(57 - ceil(log2(M + 2)), 21).
// 2^21 (any M below 2^36): a spacing is resolved to 2^-21 of the mean spacing -- 2^-43 of the unit interval at M = 2^22 --
// and the sum of a tile of 1024 draws, 2^31 +- 2^26, fits 32 bits: the one-pass kernel of the fused step stores
// 4 bytes per draw (smc_filter_kernels.h k_f_spacing_onepass; rounds 3-5 used s = 57 - lg and 8 bytes)
static double spacing_scale(i64 M)
{
    int lg = 0;
    while (((i64)1 << lg) < M + 2) ++lg;
    return ldexp(1.0, 57 - lg < 21 ? 57 - lg : 21);
}

extern "C" int smc_uniform_spacings(smc_ctx* ctx, int64_t M, uint64_t counter, double* su)
{
    SMC_REQUIRE(ctx && su, "null argument");
    SMC_REQUIRE(M > 0, "M must be positive");
    const int ntiles = (int)((M + 1 + OPS_TILE - 1) / OPS_TILE);
    // the tile sums live at the END of the scratch so a following
    // resample_launch (which uses the start) cannot alias them mid-flight
    void* scr;
    int rc = smc_scratch(ctx, (size_t)(2 * ntiles + 64) * sizeof(u64), &scr);
    if (rc) return rc;
    u64* E = (u64*)scr + ntiles + 32;
    const double scale = spacing_scale(M);
    const u32 t = (u32)counter, isl = (u32)(counter >> 32);
    SMC_LAUNCH(k_spacing_tile_sums, dim3(ntiles), dim3(SMC_BLOCK), ctx->stream, (i64)(M + 1), scale,
               (u64)ctx->seed, t, isl, E);
    SMC_LAUNCH(k_spacing_write, dim3(ntiles), dim3(SMC_BLOCK), ctx->stream, (i64)M, scale,
               (u64)ctx->seed, t, isl, (const u64*)E, ntiles, su);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

extern "C" int smc_resample(smc_ctx* ctx, int scheme, const double* W, int64_t N, int64_t M,
                            const double* u, uint64_t counter, int64_t* A)
{
    SMC_REQUIRE(ctx && W && A, "null argument");
    SMC_REQUIRE(M > 0 && N > 0, "M and N must be positive");
    if (scheme != SMC_MULTINOMIAL && scheme != SMC_STRATIFIED && scheme != SMC_SYSTEMATIC) {
        smc_set_error("%d is not a valid resampling scheme", scheme);  // resampling.py:477-481
        return SMC_ERR_SCHEME;
    }
    SmcSu su{};                  // zeros, and the "no cached pair" mark of its default initialiser
    su.scheme = scheme;
    su.M = M;
    su.dM = (double)M;
    su.u = u;
    su.seed = ctx->seed;
    su.t = (u32)counter;
    su.island = (u32)(counter >> 32);
    if (scheme == SMC_MULTINOMIAL && !u) {
        // draw the sorted uniforms on the device, then invert the CDF
        void* tmp;
        int rc = smc_malloc(ctx, (size_t)M * sizeof(double), &tmp);
        if (rc) return rc;
        rc = smc_uniform_spacings(ctx, M, counter, (double*)tmp);
        if (!rc) {
            su.u = (const double*)tmp;
            rc = resample_launch(ctx, W, N, su, (i64*)A);
        }
        int rc2 = smc_free(ctx, tmp);
        return rc ? rc : rc2;
    }
    return resample_launch(ctx, W, N, su, (i64*)A);
}

// ===========================================================================
// a-7  Xp = X[A]   (core.py:332)
// ===========================================================================
__global__ void __launch_bounds__(SMC_BLOCK)
k_gather(const double* X, const i64* A, i64 total, i64 d, double* Xp)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < total) {
        const i64 n = i / d, k = i - n * d;
        Xp[i] = X[A[n] * d + k];
    }
}

extern "C" int smc_gather(smc_ctx* ctx, const double* X, const int64_t* A, int64_t M, int64_t d,
                          double* Xp)
{
    SMC_REQUIRE(ctx && X && A && Xp, "null argument");
    SMC_REQUIRE(M > 0 && d > 0, "bad shape");
    const i64 total = M * d;
    SMC_LAUNCH(k_gather, dim3((unsigned)((total + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK),
               ctx->stream, X, (const i64*)A, total, (i64)d, Xp);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

// ===========================================================================
// a-2 / a-3  Normal.rvs / Normal.logpdf   (distributions.py:270-274)
// ===========================================================================
__global__ void __launch_bounds__(SMC_BLOCK)
k_normal_rvs(const double* loc, i64 ls, const double* scale, i64 ss, const double* z, u64 seed,
             u32 t, u32 island, i64 N, double* out)
{
    SMC_NTAB_LDS(s_ntab);
    smc_ntab_stage<SMC_BLOCK>(s_ntab, (int)threadIdx.x);
    __syncthreads();
    const i64 p = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;   // pair index
    const i64 n0 = 2 * p;
    if (n0 >= N) return;
    double z0, z1;
    if (z) {
        z0 = z[n0];
        z1 = (n0 + 1 < N) ? z[n0 + 1] : 0.0;
    } else {
        smc_normal_pair(s_ntab, seed, (u32)p, t, island, SMC_STREAM_NORMAL, z0, z1);
    }
    out[n0] = loc[n0 * ls] + scale[n0 * ss] * z0;               // loc + scale*z
    if (n0 + 1 < N) out[n0 + 1] = loc[(n0 + 1) * ls] + scale[(n0 + 1) * ss] * z1;
}

__global__ void __launch_bounds__(SMC_BLOCK)
k_normal_logpdf(const double* x, i64 xs, const double* loc, i64 ls, const double* scale, i64 ss,
                i64 N, double* out)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < N) {
        const double sc = scale[i * ss];
        const double y = (x[i * xs] - loc[i * ls]) / sc;
        out[i] = -(y * y) / 2.0 - SMC_C_NORM - log(sc);
    }
}

extern "C" int smc_normal_rvs(smc_ctx* ctx, const double* loc, int64_t loc_stride,
                              const double* scale, int64_t scale_stride, const double* z,
                              uint64_t counter, int64_t N, double* out)
{
    SMC_REQUIRE(ctx && loc && scale && out, "null argument");
    SMC_REQUIRE(N > 0, "N must be positive");
    const i64 pairs = (N + 1) / 2;
    SMC_LAUNCH(k_normal_rvs, dim3((unsigned)((pairs + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK),
               ctx->stream, loc, (i64)loc_stride, scale, (i64)scale_stride, z, (u64)ctx->seed,
               (u32)counter, (u32)(counter >> 32), (i64)N, out);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

extern "C" int smc_normal_logpdf(smc_ctx* ctx, const double* x, int64_t x_stride, const double* loc,
                                 int64_t loc_stride, const double* scale, int64_t scale_stride,
                                 int64_t N, double* out)
{
    SMC_REQUIRE(ctx && x && loc && scale && out, "null argument");
    SMC_REQUIRE(N > 0, "N must be positive");
    SMC_LAUNCH(k_normal_logpdf, dim3((unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK),
               ctx->stream, x, (i64)x_stride, loc, (i64)loc_stride, scale, (i64)scale_stride,
               (i64)N, out);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

// Poisson.logpdf (distributions.py:528-529): scipy.stats.poisson.logpmf with the generic
// rv_discrete guards (rate < 0 or NaN -> NaN; k < 0 or not an integer -> -inf)
__global__ void __launch_bounds__(SMC_BLOCK)
k_poisson_logpmf(const double* k, i64 ks, const double* rate, i64 rs, i64 N, double* out)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i >= N) return;
    const double kk = k[i * ks], mu = rate[i * rs];
    double r;
    if (!(mu >= 0.0) || kk != kk) r = NAN;
    else if (kk < 0.0 || floor(kk) != kk) r = -INFINITY;
    else {
        const double xl = (kk == 0.0) ? 0.0 : kk * log(mu);            // special.xlogy
        r = (xl - lgamma(kk + 1.0)) - mu;
    }
    out[i] = r;
}

extern "C" int smc_poisson_logpmf(smc_ctx* ctx, const double* k, int64_t k_stride, const double* rate,
                                  int64_t rate_stride, int64_t N, double* out)
{
    SMC_REQUIRE(ctx && k && rate && out, "null argument");
    SMC_REQUIRE(N > 0, "N must be positive");
    SMC_LAUNCH(k_poisson_logpmf, dim3((unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK),
               ctx->stream, k, (i64)k_stride, rate, (i64)rate_stride, (i64)N, out);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

__global__ void __launch_bounds__(SMC_BLOCK)
k_philox_fill(int normal, u64 seed, u32 t, u32 island, i64 n, double* out)
{
    SMC_NTAB_LDS(s_ntab);
    smc_ntab_stage<SMC_BLOCK>(s_ntab, (int)threadIdx.x);
    __syncthreads();
    const i64 p = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    const i64 n0 = 2 * p;
    if (n0 >= n) return;
    double v0, v1;
    if (normal) {
        smc_normal_pair(s_ntab, seed, (u32)p, t, island, SMC_STREAM_NORMAL, v0, v1);
    } else {
        u64 a, b;
        smc_philox((u32)p, t, island, SMC_STREAM_RESAMPLE, seed, a, b);
        v0 = smc_u01_halfopen(a);
        v1 = smc_u01_halfopen(b);
    }
    out[n0] = v0;
    if (n0 + 1 < n) out[n0 + 1] = v1;
}

static int philox_fill(smc_ctx* ctx, int normal, uint64_t counter, int64_t n, double* out)
{
    SMC_REQUIRE(ctx && out, "null argument");
    SMC_REQUIRE(n > 0, "n must be positive");
    const i64 pairs = (n + 1) / 2;
    SMC_LAUNCH(k_philox_fill, dim3((unsigned)((pairs + SMC_BLOCK - 1) / SMC_BLOCK)),
               dim3(SMC_BLOCK), ctx->stream, normal, (u64)ctx->seed, (u32)counter,
               (u32)(counter >> 32), (i64)n, out);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

extern "C" int smc_standard_normal(smc_ctx* ctx, uint64_t counter, int64_t n, double* out)
{
    return philox_fill(ctx, 1, counter, n, out);
}
extern "C" int smc_uniform(smc_ctx* ctx, uint64_t counter, int64_t n, double* out)
{
    return philox_fill(ctx, 0, counter, n, out);
}

// ===========================================================================
// a-8  MvNormal.rvs / logpdf   (distributions.py:946-969)
// ===========================================================================
#define MVN_MAXD 64

// out[n,i] = loc[n,i] + scale * sum_{k<=i} Z[n,k] L[i,k]        (:946-947)
__global__ void __launch_bounds__(SMC_BLOCK)
k_mvn_rvs(const double* loc, i64 loc_rows, double scale, const double* L, const double* z,
          u64 seed, u32 t, u32 island, i64 N, int d, double* out)
{
    __shared__ double sL[MVN_MAXD * MVN_MAXD];
    SMC_NTAB_LDS(s_ntab);
    for (int i = (int)threadIdx.x; i < d * d; i += SMC_BLOCK) sL[i] = L[i];
    smc_ntab_stage<SMC_BLOCK>(s_ntab, (int)threadIdx.x);
    __syncthreads();
    const i64 idx = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (idx >= N * d) return;
    const i64 n = idx / d;
    const int i = (int)(idx - n * d);
    const int hp = (d + 1) / 2;                 // Philox pairs per particle
    double acc = 0.0;
    for (int k = 0; k <= i; ++k) {
        double zk;
        if (z) {
            zk = z[n * d + k];
        } else {
            double z0, z1;
            smc_normal_pair(s_ntab, seed, (u32)(n * hp + (k >> 1)), t, island, SMC_STREAM_NORMAL, z0, z1);
            zk = (k & 1) ? z1 : z0;
        }
        acc += zk * sL[i * d + k];
    }
    const double lc = loc[(loc_rows == 1 ? 0 : n) * d + i];
    out[idx] = lc + scale * acc;
}

// one thread per particle; Linv (inverse of the lower Cholesky factor) in LDS:
// z = Linv (x-loc)/scale ; out = -0.5 |z|^2 - d log(scale) - sum log diag L - d*C
__global__ void __launch_bounds__(SMC_BLOCK)
k_mvn_logpdf(const double* x, i64 x_rows, const double* loc, i64 loc_rows, double scale,
             const double* Linv, double cst, i64 N, int d, double* out)
{
    __shared__ double sL[MVN_MAXD * MVN_MAXD];
    for (int i = (int)threadIdx.x; i < d * d; i += SMC_BLOCK) sL[i] = Linv[i];
    __syncthreads();
    const i64 n = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (n >= N) return;
    const double* xr = x + (x_rows == 1 ? 0 : n) * d;
    const double* lr = loc + (loc_rows == 1 ? 0 : n) * d;
    double q = 0.0;
    for (int i = 0; i < d; ++i) {
        double zi = 0.0;
        for (int k = 0; k <= i; ++k) zi += sL[i * d + k] * ((xr[k] - lr[k]) / scale);
        q += zi * zi;
    }
    out[n] = -0.5 * q - cst;
}

// upload a small host matrix into scratch (after `offset` bytes)
static int upload_small(smc_ctx* ctx, const double* h, size_t n, double* dst)
{
    SMC_HIP_CHECK(hipMemcpyAsync(dst, h, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SMC_OK;
}

extern "C" int smc_mvn_rvs(smc_ctx* ctx, const double* loc, int64_t loc_rows, double scale,
                           const double* L_host, const double* z, uint64_t counter, int64_t N,
                           int64_t d, double* out)
{
    SMC_REQUIRE(ctx && loc && L_host && out, "null argument");
    SMC_REQUIRE(N > 0 && d > 0 && d <= MVN_MAXD, "need 0 < d <= 64");
    SMC_REQUIRE(loc_rows == 1 || loc_rows == N, "loc must have 1 or N rows");
    void* scr;
    int rc = smc_scratch(ctx, (size_t)d * d * sizeof(double), &scr);
    if (rc) return rc;
    rc = upload_small(ctx, L_host, (size_t)(d * d), (double*)scr);
    if (rc) return rc;
    const i64 total = N * d;
    SMC_LAUNCH(k_mvn_rvs, dim3((unsigned)((total + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK),
               ctx->stream, loc, (i64)loc_rows, scale, (const double*)scr, z, (u64)ctx->seed,
               (u32)counter, (u32)(counter >> 32), (i64)N, (int)d, out);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

extern "C" int smc_mvn_logpdf(smc_ctx* ctx, const double* x, int64_t x_rows, const double* loc,
                              int64_t loc_rows, double scale, const double* L_host, int64_t N,
                              int64_t d, double* out)
{
    SMC_REQUIRE(ctx && x && loc && L_host && out, "null argument");
    SMC_REQUIRE(N > 0 && d > 0 && d <= MVN_MAXD, "need 0 < d <= 64");
    SMC_REQUIRE((x_rows == 1 || x_rows == N) && (loc_rows == 1 || loc_rows == N),
                "x / loc must have 1 or N rows");
    // Linv by forward substitution on the identity (distributions.py:952 solves
    // the same triangular system per particle)
    std::vector<double> Li((size_t)(d * d), 0.0);
    double halflogdet = 0.0;
    for (int64_t c = 0; c < d; ++c) {
        for (int64_t i = c; i < d; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int64_t k = c; k < i; ++k) s -= L_host[i * d + k] * Li[k * d + c];
            Li[i * d + c] = s / L_host[i * d + i];
        }
        halflogdet += log(L_host[c * d + c]);
    }
    const double cst = (double)d * log(scale) + halflogdet + (double)d * SMC_HALFLOG2PI;
    void* scr;
    int rc = smc_scratch(ctx, (size_t)d * d * sizeof(double), &scr);
    if (rc) return rc;
    rc = upload_small(ctx, Li.data(), (size_t)(d * d), (double*)scr);
    if (rc) return rc;
    SMC_LAUNCH(k_mvn_logpdf, dim3((unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK),
               ctx->stream, x, (i64)x_rows, loc, (i64)loc_rows, scale, (const double*)scr, cst,
               (i64)N, (int)d, out);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}
