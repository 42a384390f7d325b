/*
 * smc_hip.h -- C ABI of libsmc_hip.so: the MI355X (gfx950) SMC inner loop.
 *
 * The reference (nchopin/particles, pure Python) has no FFI; its extension
 * seams for this path are Python-level: SMC subclassing
 * (particles/core.py:369-383), the FeynmanKac API (core.py:145-197), the
 * resampling registry (resampling.py:445-481), ProbDist rvs/logpdf
 * (distributions.py:215-251) and rs.Weights (resampling.py:191-244).  Each
 * entry point below names the reference function it replaces; INTEGRATION.md
 * shows the ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - every function returns an smc_status (0 = OK); smc_last_error() gives
 *     the message of the last failure on the calling thread;
 *   - `double* / int64_t*` array arguments are DEVICE pointers obtained from
 *     smc_malloc unless the name ends in `_host`;
 *   - all work is enqueued on the context's own HIP stream; functions that
 *     return values to host memory synchronise that stream, the others do not;
 *   - arrays are C-contiguous fp64 (particles: (N,) or (N,d) row-major),
 *     ancestor indices are int64 (resampling.py:503).
 */
#ifndef SMC_HIP_H
#define SMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smc_ctx smc_ctx;        /* device + stream + scratch           */
typedef struct smc_filter smc_filter;  /* a fused on-device particle filter   */

enum smc_status {
    SMC_OK = 0,
    SMC_ERR_INVALID = 1,   /* bad argument (Python side raises ValueError)   */
    SMC_ERR_HIP = 2,       /* HIP runtime failure                             */
    SMC_ERR_NOMEM = 3,
    SMC_ERR_SCHEME = 4,    /* "<name> is not a valid resampling scheme"      */
    SMC_ERR_STATE = 5      /* call not valid in the object's current state   */
};

/* resampling.py:540-558, 599-603, 606-610 */
enum smc_scheme { SMC_MULTINOMIAL = 0, SMC_STRATIFIED = 1, SMC_SYSTEMATIC = 2 };

/* ---- context / memory ---------------------------------------------------- */
int smc_device_count(int* n_out);
int smc_ctx_create(int device, uint64_t seed, smc_ctx** out);
int smc_ctx_destroy(smc_ctx* ctx);
int smc_ctx_sync(smc_ctx* ctx);
int smc_ctx_seed(smc_ctx* ctx, uint64_t seed);        /* utils.py:209-213 seeder */
const char* smc_last_error(void);
const char* smc_version(void);
/* name, CU count, HBM bytes of the context's device */
int smc_ctx_device_info(smc_ctx* ctx, char* name_host, size_t name_len,
                        int* n_cu, uint64_t* hbm_bytes);

/* PCI bus id ("0000:05:00.0") of the context's device: what distinguishes the GPUs of one node
 * whatever ordinal a process sees them under (one-rank-per-GPU launches check that the ranks'
 * ids differ before they build the RCCL communicator) */
int smc_ctx_device_pci(smc_ctx* ctx, char* out_host, size_t len);

int smc_malloc(smc_ctx* ctx, size_t bytes, void** dptr_out);
int smc_free(smc_ctx* ctx, void* dptr);
int smc_memcpy_h2d(smc_ctx* ctx, void* dst, const void* src_host, size_t bytes);
int smc_memcpy_d2h(smc_ctx* ctx, void* dst_host, const void* src, size_t bytes);
int smc_memcpy_d2d(smc_ctx* ctx, void* dst, const void* src, size_t bytes);
int smc_memset(smc_ctx* ctx, void* dst, int byte, size_t bytes);

/* stream-ordered timers (hipEvent on the context's stream); ms between the two
 * most recent smc_timer_start / smc_timer_stop pairs */
int smc_timer_start(smc_ctx* ctx);
int smc_timer_stop(smc_ctx* ctx, float* ms_out);

/* ---- a-4: Weights.__init__ (resampling.py:217-226) -----------------------
 * NaN -> -inf IN PLACE in lw (like :220); W (may be NULL) <- exp(lw-max)/sum;
 * out4_host = {log_mean, ESS, max(lw), sum(exp(lw-max))}.  All-(-inf) input yields NaN outputs
 * like the reference.  The same kernel serves exp_and_normalise (:138),
 * log_sum_exp (:247), log_mean_exp (:291) and essl (:166) on the host side. */
int smc_lse_normalise(smc_ctx* ctx, double* lw, int64_t N, double* W,
                      double* out4_host);
/* log of the W-weighted mean of exp(v): log_mean_exp(v, W) (:291-317) */
int smc_log_wmean_exp(smc_ctx* ctx, const double* v, const double* W, int64_t N,
                      double* out_host);
/* wmean_and_var (resampling.py:320-338): out_host = mean[d], var[d] */
int smc_wmean_var(smc_ctx* ctx, const double* W, const double* X, int64_t N,
                  int64_t d, double* out_host);
/* resampling.wmean_and_cov (resampling.py:341-358): out_host = mean (d) | covariance (d, d) row-major of the weighted
 * data -- np.average(x, weights=W, axis=0), np.cov(x.T, aweights=W, ddof=0); X is (N, d) row-major, d <= 32. */
int smc_wmean_cov(smc_ctx* ctx, const double* W, const double* X, int64_t N, int64_t d, double* out_host);

/* ---- a-5: inverse_cdf (resampling.py:484-509) ----------------------------
 * su: M sorted points in [0,1]; A[n] = smallest j with su[n] <= CDF_j, clamped
 * to N-1.  The CDF is accumulated in exact 2^-62 fixed point (DESIGN.md
 * "Q62 contract"), which no summation order can change. */
int smc_inverse_cdf(smc_ctx* ctx, const double* su, const double* W, int64_t M,
                    int64_t N, int64_t* A);

/* The same with the reference's own CDF, literally: S_j accumulated left to right in fp64
 * (resampling.py:500-509: s += W[j]), A[n] = first j with su[n] <= S_j.  Given identical (su, W)
 * the ancestors ARE the reference's, bit for bit.  The N dependent roundings are reproduced in
 * parallel (csrc/smc_seqx.h: inside a binade the chain is an integer prefix sum; the few dozen elements
 * where it crosses a binade, or meets an exact tie, are walked exactly; every assumption is verified
 * before an ancestor is written, the CDF itself is never stored): tens of microseconds at N = 2^20
 * where the literal one-lane walk takes 43 ms.  W >= 0, N < 2^32. */
int smc_inverse_cdf_strict(smc_ctx* ctx, const double* su, const double* W, int64_t M,
                           int64_t N, int64_t* A);

/* S[j] = W[0] + W[1] + ... + W[j] rounded after every addition, as the reference's loop forms them
 * (resampling.py:500-509).  mode 0: the two-launch emulation smc_inverse_cdf_strict and the filter's strict mode use
 * (csrc/smc_seqx.h), written out; mode 2: the tile walk (csrc/smc_seqsum.h); mode 1: the literal one-lane walk.  All
 * agree bit for bit for any W >= 0 (tests compare them).  n_sequential_tiles (may be null): mode 0 -- the exceptions
 * the emulation walked, or -why (< 0) if it took its exact path (why: smc_filter_strict_stats); mode 2 -- tiles of 1024 the walk did exactly; mode 1 -- all. */
int smc_seq_prefix_sums(smc_ctx* ctx, const double* W, int64_t N, double* S, int mode, int64_t* n_sequential_tiles);

/* ---- a-6: rs.resampling(scheme, W, M) (resampling.py:477-481) -------------
 * u: the uniforms the scheme consumes, in the reference's order --
 *   systematic: 1 (rand(1), :609); stratified: M (rand(M), :602);
 *   multinomial: M SORTED uniforms, i.e. uniform_spacings(M) (:536-537)
 * or NULL to draw them on the device from the context's Philox stream
 * (`counter` selects the sub-stream, e.g. the time step). */
int smc_resample(smc_ctx* ctx, int scheme, const double* W, int64_t N, int64_t M,
                 const double* u, uint64_t counter, int64_t* A);
/* uniform_spacings(M) (:512-537) drawn on the device: su[0..M) sorted */
int smc_uniform_spacings(smc_ctx* ctx, int64_t M, uint64_t counter, double* su);

/* ---- a-7: Xp = X[A] (core.py:332) ---------------------------------------- */
int smc_gather(smc_ctx* ctx, const double* X, const int64_t* A, int64_t M,
               int64_t d, double* Xp);

/* ---- a-2 / a-3: Normal.rvs / Normal.logpdf (distributions.py:270-274) ------
 * loc/scale are device arrays read with the given element stride (0 = one
 * value broadcast).  z = standard normals to consume (replay) or NULL for the
 * Philox stream.  logpdf evaluates scipy's expression
 *   y=(x-loc)/scale ; -y*y/2 - 0.9189385332046727 - log(scale). */
int smc_normal_rvs(smc_ctx* ctx, const double* loc, int64_t loc_stride,
                   const double* scale, int64_t scale_stride, const double* z,
                   uint64_t counter, int64_t N, double* out);
int smc_normal_logpdf(smc_ctx* ctx, const double* x, int64_t x_stride,
                      const double* loc, int64_t loc_stride, const double* scale,
                      int64_t scale_stride, int64_t N, double* out);
/* Poisson.logpdf (distributions.py:528-529 -> scipy.stats.poisson.logpmf):
 *   xlogy(k, rate) - gammaln(k + 1) - rate,   xlogy(0, .) = 0;
 * k / rate are device arrays read with the given element stride (0 = broadcast). */
int smc_poisson_logpmf(smc_ctx* ctx, const double* k, int64_t k_stride, const double* rate,
                       int64_t rate_stride, int64_t N, double* out);
/* Normal.ppf (distributions.py:276-277 -> scipy.stats.norm.ppf): ndtri(u) * scale + loc with
 * Cephes' ndtri (the routine scipy.special.ndtri wraps); u, loc, scale strided as above.
 * SQMC's Gamma0 / Gamma (state_space_models.py:335-340). */
int smc_normal_ppf(smc_ctx* ctx, const double* u, int64_t u_stride, const double* loc,
                   int64_t loc_stride, const double* scale, int64_t scale_stride, int64_t N,
                   double* out);
/* standard normals / uniforms from the Philox stream (for the generic path) */
int smc_standard_normal(smc_ctx* ctx, uint64_t counter, int64_t n, double* out);
int smc_uniform(smc_ctx* ctx, uint64_t counter, int64_t n, double* out);

/* ---- a-8: MvNormal.rvs / logpdf (distributions.py:946-969) ----------------
 * loc: (N,d) rows (loc_rows = N) or one row broadcast (loc_rows = 1);
 * L_host: (d,d) lower Cholesky factor of cov, row-major, HOST memory;
 * rvs:    out = loc + scale * (Z @ L^T), Z (N,d) from z or Philox;
 * logpdf: -0.5*|L^-1 (x-loc)/scale|^2 - d*log(scale) - sum(log diag L) - d*C. */
int smc_mvn_rvs(smc_ctx* ctx, const double* loc, int64_t loc_rows, double scale,
                const double* L_host, const double* z, uint64_t counter,
                int64_t N, int64_t d, double* out);
int smc_mvn_logpdf(smc_ctx* ctx, const double* x, int64_t x_rows,
                   const double* loc, int64_t loc_rows, double scale,
                   const double* L_host, int64_t N, int64_t d, double* out);

/* ---- (b) arithmetic for device-resident model code -------------------------
 * out[i] = a[i*stride_a] (op) b[i*stride_b]  (b == NULL: the scalar alpha).  A user-defined
 * Feynman-Kac model (core.py:108-197) written with numpy expressions runs on arrays in HBM
 * through this entry (particles_amd.DeviceArray operators and ufuncs). */
enum smc_ew_op {
    SMC_EW_ADD = 0, SMC_EW_SUB, SMC_EW_MUL, SMC_EW_DIV, SMC_EW_RSUB, SMC_EW_RDIV, SMC_EW_NEG,
    SMC_EW_EXP, SMC_EW_LOG, SMC_EW_SQRT, SMC_EW_COS, SMC_EW_SIN, SMC_EW_ABS, SMC_EW_SQUARE,
    SMC_EW_POW, SMC_EW_MIN, SMC_EW_MAX, SMC_EW_ARCTAN
};
int smc_elementwise(smc_ctx* ctx, int op, const double* a, int64_t stride_a, const double* b,
                    int64_t stride_b, double alpha, int64_t n, double* out);
/* strides: 1 = element i, 0 = one value for all, -m = element i mod m (row vector against (N,m)).
 * out (N,k) = X (N,d) @ M_host (d,k), d, k <= 64: np.dot(xp, F.T) in a model's PX / PY
 * (kalman.py:339-346). */
int smc_rows_matmul(smc_ctx* ctx, const double* X, int64_t N, int64_t d, const double* M_host,
                    int64_t k, double* out);
/* dst[i*dst_stride] = src[i*src_stride], i < n (strides in elements, >= 1): a column of an (N,d)
 * array out (x[..., i], distributions.py:1102) or in (np.stack(cols, axis=1), :1106). */
int smc_copy_strided(smc_ctx* ctx, const double* src, int64_t src_stride, double* dst,
                     int64_t dst_stride, int64_t n);

/* ---- (f) SQMC building blocks (core.py:339-349, rqmc.py, hilbert.py:33-58) ---
 * smc_argsort: out[i] = index of the i-th smallest x (np.argsort(x); hilbert_sort for d = 1;
 *   stable, -0.0 before +0.0, NaNs last).
 * smc_sobol: the first N points of the d-dimensional Sobol' sequence (Joe-Kuo direction numbers,
 *   30 bits, Gray-code order: scipy.stats.qmc.Sobol(d, scramble=False) bit for bit), each
 *   coordinate XOR-ed with a 30-bit digital shift drawn from the Philox stream `counter`
 *   (randomised QMC; shift = 0 when scramble == 0), then rqmc.safe_generate's map
 *   0.5 + (1 - 1e-10) (u - 0.5) when safe != 0.  d <= 10.  out (N, d) row-major. */
int smc_argsort(smc_ctx* ctx, const double* x, int64_t N, int64_t* out);
/* Test access: the radix sort behind smc_argsort / smc_hilbert_sort / smc_wquantiles / the fused SQMC step takes four
 * passes over the 32 bits below the keys' highest varying bit plus a fix-up (instead of eight passes over all 64) from n
 * keys on (default and minimum: 8 193, the first size that takes more than one workgroup).  Process-wide; the results are the same permutation either way. */
int smc_debug_sort_window_min(long long n);
/* hilbert_sort (hilbert.py:33-58) of N vectors x (N, d), 2 <= d <= 16: standardise each
 * component (np.mean / np.std over the particles), logistic map to (0,1), scale to integers
 * below floor(2^(62/d)), Hilbert index of each point (Witham's codec, hilbert.py:61-292,
 * restated with integer operations only), argsort of the indices.  keys_out (N) receives the
 * Hilbert indices when not NULL. */
int smc_hilbert_sort(smc_ctx* ctx, const double* x, int64_t N, int32_t d, int64_t* out,
                     int64_t* keys_out);
/* hilbert_array (hilbert.py:13-30): Hilbert indices of N points with non-negative integer
 * coordinates xint (N, d) device int64, 1 <= d <= 16.  The index is accumulated in int64
 * that wraps, as the reference's does when d * bit_length(max coordinate) > 63 (d >= 4 on
 * hilbert_sort's grid); hilbert_sort orders the signed values like np.argsort. */
int smc_hilbert_array(smc_ctx* ctx, const int64_t* xint, int64_t N, int32_t d, int64_t* out);
int smc_sobol(smc_ctx* ctx, int64_t N, int32_t d, int32_t scramble, int32_t safe,
              uint64_t counter, double* out);
/* The same N points (N a power of two), rows in ascending order of the FIRST coordinate:
 * out = smc_sobol(...)[argsort(smc_sobol(...)[:, 0])], without the sort -- the order is known in
 * closed form (dimension 1 is the bit-reversed Gray code XOR the shift).  SQMC consumes its
 * points only through u[tau] with tau = argsort(u[:, 0]) (core.py:343-347). */
int smc_sobol_sorted(smc_ctx* ctx, int64_t N, int32_t d, int32_t scramble, int32_t safe,
                     uint64_t counter, double* out);

/* ---- (f) weighted quantiles (resampling.py:381-417 wquantiles) -------------
 * W (N), x (N,d) device; alphas_host (k) levels; out_host (d,k): for every column the
 * np.interp-olated alpha-quantiles of the weighted sample (argsort, cumsum of the weights
 * in that order, searchsorted, 2-point interpolation). */
int smc_wquantiles(smc_ctx* ctx, const double* W, const double* x, int64_t N, int64_t d,
                   const double* alphas_host, int k, double* out_host);

/* ---- (f) remaining schemes of rs_funcs ------------------------------------
 * residual (resampling.py:611-626), two calls because the reference draws
 * uniform_spacings(M - sip) AFTER it knows sip = sum floor(M W):
 *   split:      r_dev (N) <- (M W - floor(M W)) / (M - sip), *sip_host <- sip
 *   ancestors:  A[:sip] = arange(N).repeat(floor(M W));  A[sip:] = inverse_cdf(su_dev, r_dev)
 *               (su_dev: M - sip sorted uniforms; may be NULL when sip == M) */
int smc_residual_split(smc_ctx* ctx, const double* W, int64_t N, int64_t M,
                       double* r_dev, int64_t* sip_host);
int smc_residual_ancestors(smc_ctx* ctx, const double* W, const double* r_dev,
                           int64_t N, int64_t M, int64_t sip, const double* su_dev,
                           int64_t* A);
/* ssp (resampling.py:628-678): u_dev = the N - 1 uniforms the reference draws.  A
 * sequential process: one lane walks it on the device.  SMC_ERR_INVALID ("ssp resampling:
 * wrong size for output") where the reference raises ValueError. */
int smc_resample_ssp(smc_ctx* ctx, const double* W, const double* u_dev, int64_t N, int64_t M,
                     int64_t* A);
/* killing (resampling.py:680-697; M == N):
 *   split:      killed_dev[i] <- u_dev[i] * max(W) >= W[i], *nkilled_host <- their number
 *   ancestors:  A = arange(N); A[killed] = Am (the caller's multinomial(W, nkilled), device) */
int smc_killing_split(smc_ctx* ctx, const double* W, const double* u_dev, int64_t N,
                      unsigned char* killed_dev, int64_t* nkilled_host);
int smc_killing_ancestors(smc_ctx* ctx, const unsigned char* killed_dev, const int64_t* Am,
                          int64_t N, int64_t* A);

/* ---- a-1: the fused SMC step loop (core.py:369-383) ------------------------
 * One smc_filter holds `n_islands` independent particle filters of N particles
 * each (multiSMC runs / SMC^2 inner filters, core.py:431, smc_samplers.py:
 * 1110-1113) that advance in lock step on the device.  The model is one of
 * the closed family the reference's hot-path configs use. */
enum smc_model_kind {
    SMC_MODEL_LINGAUSS = 1,   /* kalman.py:397-452 (ToySSM = rho 1, sigmaX 1, sigma0 1) */
    SMC_MODEL_STOCHVOL = 2,   /* state_space_models.py:446-473 */
    SMC_MODEL_MVLINGAUSS = 3, /* kalman.py:296-361 */
    SMC_MODEL_GORDON = 4,     /* state_space_models.py:546-577 (bootstrap) */
    SMC_MODEL_THETALOGISTIC = 5, /* state_space_models.py:657-683 (bootstrap) */
    SMC_MODEL_SVLEVERAGE = 6, /* state_space_models.py:501-541 StochVolLeverage (bootstrap) */
    SMC_MODEL_DISCRETECOX = 7 /* state_space_models.py:611-630: Y_t | x ~ Poisson(exp(x)) (bootstrap) */
};
enum smc_fk_kind {
    SMC_FK_BOOTSTRAP = 0,     /* state_space_models.py:299-349 */
    SMC_FK_GUIDED = 1,        /* state_space_models.py:352-398 (model's own proposal) */
    SMC_FK_APF_BOOT = 3,      /* state_space_models.py:431-438 AuxiliaryBootstrap: the BOOTSTRAP step (proposal = the
                               * transition) + the auxiliary weights of SMC_FK_APF; STOCHVOL, LINGAUSS; same limits */
    SMC_FK_APF = 2            /* state_space_models.py:406-428 auxiliary PF: the guided step + the
                               * auxiliary weights of core.py:299-313 (resampling on lw + logeta,
                               * weights reset to log_mean_exp(logeta, W) - logeta[A]); STOCHVOL
                               * (Pitt & Shephard, :475-498) and LINGAUSS (kalman.py:448-452: the
                               * predictive density of y_{t+1}); N <= 1024 (the one-launch filter) or
                               * 1024 < N <= 2^30 (the two-level step); no moments, no rolling window */
};
enum smc_rng_mode {
    SMC_RNG_PHILOX = 0,       /* counter-based Philox4x32-10 per lane */
    SMC_RNG_REPLAY = 1        /* consume a tape of the reference's draws */
};

typedef struct smc_model {
    int32_t kind;             /* smc_model_kind */
    int32_t fk;               /* smc_fk_kind */
    int32_t dx, dy;           /* state / observation dimension (1 for univariate) */
    /* univariate parameters, one row of 16 per island (HOST, (n_islands,16)).
     * Derived constants are passed in (not recomputed) so that they carry the
     * caller's roundings (numpy's log / sqrt) and replay runs match bit for bit:
     *   LINGAUSS: 0 rho, 1 sigmaX, 2 sigmaY, 3 sigma0, 4 log(sigmaY),
     *             5 log(sigmaX), 6 log(sigma0), 7 sigmaX^2, 8 sigmaY^2,
     *             guided only (kalman.py:436-446): 9 sig2post, 10 sqrt(9),
     *             11 log(10), 12 sig2post0, 13 sqrt(12), 14 log(13);
     *             auxiliary filter (:448-452): 15 sqrt(sigmaX^2 + sigmaY^2)
     *   STOCHVOL: 0 mu, 1 rho, 2 sigma, 3 sigma/sqrt(1-rho^2), 4 (1-rho)*mu; guided / APF
     *             (:475-498): 5 log(sigma), 6 log(sig0), 7 0.5*sigma^2, 8 0.5*sig0^2, 9 0.5/sigma^2
     *   GORDON:   0 b, 1 sigmaX, 2 c, 3 sigma0 (2.0), 5 a;  aux_host[t] = d*cos(e*(t-1))
     *   THETALOGISTIC: 0 tau0, 1 sigmaX, 2 sigmaY, 3 sigma0 (1.0), 4 log(sigmaY),
     *             5 tau1, 6 tau2
     *   SVLEVERAGE: as STOCHVOL, plus 5 phi, 6 sqrt(1 - phi^2)
     *   DISCRETECOX: 0 mu, 1 phi, 2 sigma, 3 sigma/sqrt(1-phi^2);
     *             aux_host[t] = gammaln(y_t + 1) (the data-only term of the Poisson log-pmf) */
    const double* params_host;
    /* MVLINGAUSS (HOST, row-major): F(dx,dx) G(dy,dx) covX(dx,dx) covY(dy,dy)
     * mu0(dx) cov0(dx,dx); shared by all islands */
    const double *F_host, *G_host, *covX_host, *covY_host, *mu0_host, *cov0_host;
    /* per-step scalar of the model, (T,) HOST, or NULL (GORDON: additive term of the
     * transition mean, the caller's d*cos(e*(t-1)), entry 0 unused; DISCRETECOX: gammaln(y_t+1)) */
    const double* aux_host;
} smc_model;

typedef struct smc_filter_opts {
    int64_t N;                /* particles per island */
    int64_t T;                /* number of time steps (len(data)) */
    int32_t n_islands;
    int32_t scheme;           /* smc_scheme */
    double ESSrmin;           /* resample when ESS < N*ESSrmin (core.py:181-183) */
    uint64_t seed;            /* Philox key; island i uses counter word i */
    int32_t rng_mode;         /* smc_rng_mode */
    int32_t use_graph;        /* 1: replay the step loop from a hipGraph */
    int32_t island_offset;    /* global index of this filter's island 0 (multi-GPU sharding) */
    int32_t keep_history;     /* k >= 2: the k most recent steps stay resident in a ring of k slots
                               * (RollingParticleHistory, smoothing.py:186-207: k (8 dx + 12) B per particle);
                               * 1: X, A, lw of EVERY step stay resident -- the step loop writes step t
                               * into slot t of (T, n_islands, N[, dx]) arrays instead of alternating
                               * between two (ParticleHistory.save, smoothing.py:181-207, at no extra
                               * traffic); needs T*n_islands*N*(8 dx + 16) bytes of HBM */
    int32_t moments;          /* 1: the Moments collector on the device (collectors.py:301-317 with
                               * rs.wmean_and_var): weighted mean and variance of every component of
                               * X_t after every step, no host round trip (smc_filter_moments) */
    int32_t flags;            /* SMC_FLAG_* */
} smc_filter_opts;
/* MVLINGAUSS guided filter: evaluate the weight of the model's optimal proposal in its collapsed
 * form log G_t = log p(y_t | x_{t-1}) = log N(y_t; G F x_{t-1}, G covX G' + covY) (the three
 * terms of state_space_models.py:380-392 cancel to it analytically; S is the innovation
 * covariance of kalman.py:215-229) -- 44 instead of 72 matrix instructions per 16 particles.
 * Same particles; log-weights equal up to rounding, hence opt-in. */
#define SMC_FLAG_COLLAPSED_PROPOSAL 1
/* Univariate filters on the two-level step: ancestors by the reference's sequential fp64 CDF of the
 * filter's own normalised weights (what smc_filter_get(SMC_FIELD_W) returns for the step before) --
 * A_t == particles.resampling.inverse_cdf(su, W_{t-1}) bit for bit -- instead of the exact integer
 * CDF.  Costs a one-wavefront pass over the weights per resampling step (ms at N = 2^20). */
#define SMC_FLAG_STRICT_ANCESTORS 2
/* SQMC (particles.SMC(qmc=True), core.py:315-321, 339-349) as a fused loop: univariate Bootstrap / Guided
 * filters, N = 2^k >= 32 (N < 2048, and MVLINGAUSS with 2 <= d <= 9: the flat step -- eager launches, no history
 * slots, no moments; N >= 2048: the two-level step).  Every step sorts the particles (hilbert_sort = argsort for d = 1: the radix sort),
 * chooses ancestors by the inverse CDF of the weights in sorted order at the sorted first coordinates of a
 * scrambled Sobol' point set (known in closed form for N = 2^k: no second sort), and moves with the
 * inverse normal CDF of the second coordinates (Gamma = ProbDist.ppf).  Always resamples (core.py:340):
 * `scheme` and `ESSrmin` are ignored.  The points are smc_sobol's (see smc_filter_sqmc_points). */
#define SMC_FLAG_SQMC 4

/* Verification switches (bits 8 and up of opts.flags): each selects an ALTERNATIVE CODE PATH that must
 * give the same results (the same bits, or the documented near-tie differences between the two exact
 * CDFs) as the one a filter of that shape takes by default -- the parity suite and tools/fuzz_paths.py
 * run both and compare.  Nothing here changes what is computed; the library reads no environment
 * variable to choose a path (the Python layer maps its SMC_* test variables onto these bits). */
#define SMC_PATH_FLAT_CDF        (1 << 8)   /* flat Q62 CDF (k_ancestors [+ k_prepare]) where the two-level step applies */
#define SMC_PATH_TWO_LEVEL_MID   (1 << 9)   /* k_reduce2 in front of k_ancestors2 even on resident grids */
#define SMC_PATH_EXACT_COUNTS    (1 << 10)  /* form every c Q_b / t_b exactly (no fp64 band shortcut) */
#define SMC_PATH_FORCE_UNFUSED   (1 << 12)  /* flat step: k_prepare regardless of the grid size */
#define SMC_PATH_NO_SMALL        (1 << 13)  /* N <= 1024: the multi-kernel step instead of k_filter_small */
#define SMC_PATH_NO_HEAVY        (1 << 15)  /* no heavy-parent list */
#define SMC_PATH_NO_TK           (1 << 16)  /* normals never start on the host's time index */
#define SMC_PATH_SPACING_3PASS   (1 << 19)  /* uniform_spacings in three passes instead of one */
#define SMC_PATH_SPLIT_REDUCE     (1 << 24)  /* multinomial, one-pass spacings: k_reduce2 as a launch of its own */
#define SMC_PATH_NO_WIDE          (1 << 30)  /* resident two-level step: k_ancestors2 (one tile per workgroup) instead of k_ancestors2w */
#define SMC_PATH_NO_XCD_CHUNKS    (1 << 5)   /* two-level step: tile = workgroup index instead of contiguous runs of tiles per XCD */
#define SMC_PATH_STRICT_LITERAL   (1 << 6)   /* SMC_FLAG_STRICT_ANCESTORS: the sequential CDF by the literal one-lane walk, not its parallel emulation */
#define SMC_PATH_SQ_GATHER        (1 << 29)  /* SMC_FLAG_SQMC: gather the sorted log-weights where they could be recomputed */
#define SMC_PATH_MV_DENSE         (1 << 3)   /* MVLINGAUSS with diagonal G / covX / covY / cov0: the dense MFMA products all the same (the twin the
                                               element-wise form is checked against, bit for bit; bench.py's c4_dense leg) */
#define SMC_PATH_SP_TPW(n)        (((n) & 15) << 25)   /* one-pass spacings: n = 1, 2, 4, 8 tiles of draws per workgroup */
#define SMC_PATH_MV_CHUNKS(n)    (((n) & 15) << 20)   /* k_propagate_mv: n = 1, 2, 4, 8 chunks per workgroup */

/* y_host: data, (T, dy) row-major, shared by all islands. */
int smc_filter_create(smc_ctx* ctx, const smc_model* model,
                      const smc_filter_opts* opts, const double* y_host,
                      smc_filter** out);
int smc_filter_destroy(smc_filter* f);
/* SMC_FLAG_SQMC filters, before the first step: the run's points are those of smc_sobol /
 * smc_sobol_sorted (scramble = safe = 1) under a context seeded with point_seed -- point set `counter0`
 * (1 coordinate, rqmc.sobol(N, 1), core.py:317) for t = 0 and `counter0 + t` (2 coordinates,
 * core.py:341) for step t; island i takes counter word i << 32 on top.  Default: (opts.seed, 1). */
int smc_filter_sqmc_points(smc_filter* f, uint64_t point_seed, uint64_t counter0);
/* An independent copy of a filter in its current state: copy.deepcopy(pf) of the reference
 * (theta-level resampling of SMC^2 deep-copies every duplicated filter, smc_samplers.py:319-361).
 * Replay tapes are shared (caller-owned, read-only).  The copy's Philox streams are the source's
 * (same seed, island ids, time): smc_filter_reseed gives it draws of its own from the next step on. */
int smc_filter_clone(smc_filter* src, smc_filter** out);
int smc_filter_reseed(smc_filter* f, uint64_t seed);
/* Replay tapes (device): z (T, n_islands, N, dx) normals; u (T, n_islands, K)
 * with K = 1 (systematic: rand(1)), N (stratified: rand(N)) or N (multinomial:
 * the SORTED uniforms).  Slots of steps that do not resample are ignored. */
int smc_filter_set_replay(smc_filter* f, const double* z, const double* u);
/* Enqueue `nsteps` time steps (asynchronous).  Steps beyond T are no-ops. */
int smc_filter_step(smc_filter* f, int64_t nsteps);
int smc_filter_sync(smc_filter* f);
int smc_filter_t(smc_filter* f, int64_t* t_out);      /* steps enqueued so far */
/* Per-step summaries (collectors.py:278-295) of steps [0, t):
 * out_host (n_islands, t, 5) = ESS, log_mean_w, loglt, logLt, rs_flag. */
#define SMC_SUMMARY_COLS 5
int smc_filter_summaries(smc_filter* f, double* out_host);
int smc_filter_logLt(smc_filter* f, double* out_host /* n_islands */);
enum smc_state_field {
    SMC_FIELD_X = 0,   /* (N,dx) fp64   core.py SMC.X            */
    SMC_FIELD_XP = 1,  /* (N,dx) fp64   SMC.Xp = X_{t-1}[A]      */
    SMC_FIELD_A = 2,   /* (N,) int64    SMC.A                    */
    SMC_FIELD_LW = 3,  /* (N,) fp64     SMC.wgts.lw              */
    SMC_FIELD_W = 4    /* (N,) fp64     SMC.W                    */
};
/* Copy one island's field (state after the last enqueued step) to the host. */
int smc_filter_get(smc_filter* f, int field, int island, void* out_host);
/* Replace the particles (X_host, (N,dx) or NULL) and / or the log-weights (lw_host, (N,) or
 * NULL; univariate filters) of the step just done, island `island`: the counterpart of
 * assigning SMC.X / SMC.wgts between two steps (core.py:222-233 attributes; e.g. an MCMC
 * rejuvenation of the particles, smc_samplers.py:1129-1143).  ESS, log-mean weight and evidence
 * of that step, the resample decision and the CDF of the next are recomputed on the device. */
int smc_filter_set_state(smc_filter* f, int island, const double* X_host, const double* lw_host);
/* Algorithmic bytes moved per particle-step and kernel launches
 * per step, for roofline accounting. */
/* PMCMC move of SMC^2 (smc_samplers.py:1129-1143): where accept_host[i] != 0, island i of dst
 * takes over island i of src (same shapes, model kind and time index; one context). */
int smc_filter_copy_islands(smc_filter* dst, smc_filter* src, const unsigned char* accept_host);
/* Island migration: the state of whole filters (particles, log-weights, evidence history, step
 * record, parameters, CDF partials) packed into / restored from a contiguous device buffer of
 * n x smc_filter_island_bytes() bytes, entry j = island islands_host[j]. */
int smc_filter_island_bytes(smc_filter* f, int64_t* bytes);
/* A filter that has not stepped yet takes time index t; its state is then supplied by
 * smc_filter_unpack_islands (every island, before the next step).  Waste-free SMC^2
 * (smc_samplers.py:669-684) assembles its new population from the chains' batches this way. */
int smc_filter_fast_forward(smc_filter* f, int64_t t);
int smc_filter_pack_islands(smc_filter* f, const int64_t* islands_host, int n, void* pack_dev);
int smc_filter_unpack_islands(smc_filter* f, const int64_t* islands_host, int n, const void* pack_dev);
/* ---- SMC^2 (smc_samplers.py:1038-1167): the theta level on the device.  Every island is the
 * particle filter of one theta-particle.  Once enabled (before the first step), every time step
 * is followed by a one-workgroup kernel that adds the islands' evidence increments to the
 * theta log-weights (SMC2.logG, :1099-1120) and evaluates the theta-level ESS; when it drops
 * below ess_rmin * n_islands (core.py:181-183 for the outer SMC) the kernel records the step and
 * FREEZES the batch -- the steps already enqueued behind it do nothing -- so the caller can
 * enqueue many steps per synchronisation and deal with resample-move events when they occur:
 *   smc_filter_step(f, K); smc_filter_theta_state(f, lw, &stop, &done, ess);
 *   if (stop) { [resample thetas: smc_filter_permute_islands; PMCMC move: a second batch +
 *                smc_filter_copy_islands] ; smc_filter_theta_resume(f, NULL); }            */
int smc_filter_theta_enable(smc_filter* f, double ess_rmin);
/* The same for a theta-population SHARDED over the ranks of `comm` (multi-GPU SMC^2; every rank holds
 * n_islands consecutive theta-particles, rank r the global indices r M .. r M + M - 1).  The theta level
 * is replicated: each step is followed by one ncclAllGather of the ranks' evidence increments ENQUEUED ON
 * THE CONTEXT'S STREAM (N_theta x 8 bytes over xGMI, no host synchronisation) and the same one-workgroup
 * update on every rank -- same values, same order, same bits, hence the same ESS, the same decision and
 * the same freeze on every rank and for every world size.  smc_filter_theta_state / _resume / _logmeans
 * then exchange nranks x n_islands values.  Every rank makes the same calls in the same order. */
typedef struct smc_comm smc_comm;
int smc_filter_theta_enable_sharded(smc_filter* f, smc_comm* comm, double ess_rmin);
/* lw_theta_host (n_islands) or NULL; *stop_t = step at which the batch froze (0: running);
 * *steps_done = time steps accounted for in the theta weights; ess_host (steps_done) or NULL:
 * the theta-level ESS after every step. */
int smc_filter_theta_state(smc_filter* f, double* lw_theta_host, int64_t* stop_t, int64_t* steps_done,
                           double* ess_host);
/* out_host (steps_done): log of the mean theta weight after every step accounted for -- the outer
 * SMC's log_mean_w (core.py:351-359), from which the model's evidence increments follow */
int smc_filter_theta_logmeans(smc_filter* f, double* out_host, int64_t* steps_done);
/* New theta log-weights (NULL: zeros, i.e. after a theta-level resampling) and, if frozen, back
 * to the stop step: stepping continues from there. */
int smc_filter_theta_resume(smc_filter* f, const double* lw_theta_host);
/* opts.moments filters: out_host (n_islands, t, 2*dx) = per step the dx weighted means, then the
 * dx weighted variances of the particles (resampling.py:320-338). */
int smc_filter_moments(smc_filter* f, double* out_host);
/* theta-level resampling of whole filters (SMC^2, smc_samplers.py:319-361): island i
 * continues from the state of island src_host[i] (particles, log-weights, summaries, step
 * record, parameter row).  Random streams stay tied to the slot.  Not with keep_history. */
int smc_filter_permute_islands(smc_filter* f, const int64_t* src_host);
/* keep_history filters: the state of an earlier step, fields as smc_filter_get
 * (hist.X[step], hist.A[step], hist.wgts[step].lw / .W; smoothing.py:204-207). */
int smc_filter_history(smc_filter* f, int field, int64_t step, int island, void* out_host);
/* Multinomial filters in production (Philox) mode: out_host (N) = the sorted uniforms
 * uniform_spacings(N) (resampling.py:512-537) the resampling of step t (1 <= t < T) draws for this
 * island -- a function of (seed, island, t) only.  The step loop never writes them (the two-level
 * step regenerates the few it compares with); this entry is for inspection and the parity tests. */
int smc_filter_spacings(smc_filter* f, int64_t t, int island, double* out_host);
/* keep_history filters: genealogy of the current particles (compute_trajectories,
 * smoothing.py:209-219): out_host (t, N) int64, row t-1 = arange(N), row s-1 = A_s[row s]. */
int smc_filter_trajectories(smc_filter* f, int island, int64_t* out_host);
/* keep_history filters: one genealogical line (extract_one_trajectory, smoothing.py:256-269):
 * out_host (t, dx) = X_s[n_s] with n_{t-1} = n_last and n_{s-1} = A_s[n_s]. */
int smc_filter_one_trajectory(smc_filter* f, int island, int64_t n_last, double* out_host);
int smc_filter_info(smc_filter* f, double* bytes_per_particle_step,
                    int* kernels_per_step);
/* Average duration (ms) of the step's two parts -- the propagate kernel ("move") and the
 * resampling kernel(s) in front of it ("prepare") -- over the steps run since the last call,
 * measured with HIP events on the filter's stream when profiling is enabled with
 * smc_filter_profile(f, 1).  Steps are sampled in three kinds: the whole step, the interval up to
 * the launch of the propagate kernel, the interval from there to the end; move = whole - first,
 * prepare = whole - second, so the fixed cost of an event interval (~4 us) cancels and each
 * part is measured, not derived from the other. */
int smc_filter_profile(smc_filter* f, int enable);
/* the kernels one time step of this filter launches, e.g. "k_ancestors2+k_propagate"
 * (two-level CDF), "k_ancestors<fused>+k_propagate", "k_prepare+k_ancestors+k_propagate",
 * "k_filter_small" (whole T-loop in one launch), "...+k_propagate_mv"; NUL-terminated into out. */
int smc_filter_describe(smc_filter* f, char* out, size_t n);
/* Checkpoint / resume -- what pickling a particles.SMC object is to the reference (core.py:415-428 returns the SMC
 * objects of worker processes, utils.py:178-186; SURVEY 5 "checkpoint / resume").  The state is every device array of
 * the filter plus its host-side counters, as one host buffer of smc_filter_state_bytes bytes; smc_filter_load_state
 * accepts it into a filter created from the same (model, options, data) -- shape mismatches are SMC_ERR_INVALID -- and
 * the run then continues bit for bit.  (Replay tapes are the caller's arrays: set them again.) */
int smc_filter_state_bytes(smc_filter* f, int64_t* nbytes);
int smc_filter_save_state(smc_filter* f, void* out_host, int64_t nbytes);
int smc_filter_load_state(smc_filter* f, const void* in_host, int64_t nbytes);
/* SMC_FLAG_STRICT_ANCESTORS (the reference's sequential fp64 inverse_cdf, resampling.py:484-509): how the last
 * resampling step of `island` formed that CDF -- *exact_path = 0: the two-launch emulation (csrc/smc_seqx.h), its
 * assumptions verified; non-zero: the exact scan-until-exception path ran (milliseconds at N = 2^20), the value saying
 * why (1: more exceptions than the lists hold -- engineered ties, NaN / negative weights; 2: an integer step of the walk
 * left its binade; 4 / 8: a segment behind an exception / a tile's head expected another binade than the walk found);
 * *exceptions: the elements the emulation's serial walk handled.  Diagnostics: the results are the
 * reference's either way. */
int smc_filter_strict_stats(smc_filter* f, int32_t island, int64_t* exact_path, int64_t* exceptions);
int smc_filter_kernel_ms(smc_filter* f, double* move_ms_avg,
                         double* prepare_ms_avg, int64_t* n_samples);

/* ---- multi-GPU seam: multiSMC / SMC^2 islands (core.py:431, utils.py:158) ----
 * Independent runs shard across one process per GPU with no data-path
 * exchange; the only collective gathers the per-island log-evidences over RCCL
 * (xGMI).  Rank 0 obtains the id and distributes it out of band (the bench
 * uses the torch.distributed/gloo rendezvous the launcher provides). */
#define SMC_COMM_ID_BYTES 128
int smc_comm_unique_id(char* id_host /* SMC_COMM_ID_BYTES */);
int smc_comm_create(smc_ctx* ctx, int nranks, int rank, const char* id_host,
                    smc_comm** out);
/* recv (nranks*count) <- concatenation over ranks of send (count); blocking */
int smc_comm_allgather_f64(smc_comm* comm, const double* send, int64_t count,
                           double* recv);
/* ... enqueued on the context's stream without waiting for it */
int smc_comm_allgather_f64_async(smc_comm* comm, const double* send, int64_t count,
                                 double* recv);
int smc_comm_rank(smc_comm* comm, int32_t* rank, int32_t* nranks);
/* All-to-all of byte blocks between the ranks (island migration of multi-GPU SMC^2: a global
 * theta-resampling, smc_samplers.py:319-361, moves whole filters between GPUs): grouped
 * ncclSend / ncclRecv, one pair per peer over xGMI.  send / recv: device buffers; counts and
 * displacements (bytes, HOST, one per rank; the self block included). */
int smc_comm_alltoallv(smc_comm* comm, const void* send, const int64_t* scount, const int64_t* sdisp,
                       void* recv, const int64_t* rcount, const int64_t* rdisp);
int smc_comm_destroy(smc_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* SMC_HIP_H */
