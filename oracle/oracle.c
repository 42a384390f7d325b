/*
 * CPU oracle, C half -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain C restatement of the pieces of nchopin/particles' SMC hot path that
 * are too slow as Python loops, plus a restatement of the counter-based
 * generator the HIP path uses in production mode.  Nothing in particles_amd/
 * links or loads this file.  Paths below are relative to /root/reference.
 *
 * Build: make -C oracle   (-> oracle/_build/liboracle.so)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* particles/resampling.py:484-509 (numba inverse_cdf): sequential fp64 CDF,
 * strict '>' advance.  The reference has no bounds check (numba) / raises
 * IndexError (pure Python) when su[n] exceeds the accumulated total; we
 * return 1 in that case so the Python side can raise the same IndexError. */
int64_t orc_inverse_cdf_seq(const double *su, const double *W, int64_t M,
                            int64_t N, int64_t *A)
{
    int64_t j = 0;
    double s = W[0];
    for (int64_t n = 0; n < M; ++n) {
        while (su[n] > s) {
            ++j;
            if (j >= N) return 1;
            s += W[j];
        }
        A[n] = j;
    }
    return 0;
}

/* The fixed-point CDF contract of the HIP kernels (oracle/smc_oracle.py,
 * "Q62"): q_i = rint(W_i 2^62), C_j = sum q_i (exact), T_n = ceil(su_n 2^62),
 * A_n = smallest j with T_n <= C_j, clamped to N-1.  Boundary rule and clamp
 * follow resampling.py:505-508. */
static inline uint64_t q62_w(double w)
{
    return (w > 0.0) ? (uint64_t)rint(w * 4611686018427387904.0) : 0;
}
static inline uint64_t q62_t(double su)
{
    return (su > 0.0) ? (uint64_t)ceil(su * 4611686018427387904.0) : 0;
}

void orc_inverse_cdf_q62(const double *su, const double *W, int64_t M,
                         int64_t N, int64_t *A)
{
    int64_t j = 0;
    uint64_t c = q62_w(W[0]);
    for (int64_t n = 0; n < M; ++n) {
        uint64_t t = q62_t(su[n]);
        while (t > c && j < N - 1) {
            ++j;
            c += q62_w(W[j]);
        }
        A[n] = j;
    }
}

/* Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11; Random123). */
#define PH_M0 0xD2511F53u
#define PH_M1 0xCD9E8D57u
#define PH_W0 0x9E3779B9u
#define PH_W1 0xBB67AE85u

void orc_philox4x32_10(const uint32_t *ctr, const uint32_t *key, uint32_t *out)
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        if (r > 0) { k0 += PH_W0; k1 += PH_W1; }
        uint64_t p0 = (uint64_t)PH_M0 * c0, p1 = (uint64_t)PH_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void philox_u64_pair(uint64_t seed, uint32_t idx, uint32_t t,
                            uint32_t island, uint32_t stream, uint64_t *x01,
                            uint64_t *x23)
{
    uint32_t ctr[4] = {idx, t, island, stream};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    orc_philox4x32_10(ctr, key, o);
    *x01 = ((uint64_t)o[1] << 32) | o[0];
    *x23 = ((uint64_t)o[3] << 32) | o[2];
}

static inline double u01_open(uint64_t x)
{
    return ((double)(x >> 12) + 0.5) * 0x1.0p-52;
}
static inline double u01_halfopen(uint64_t x)
{
    return (double)(x >> 11) * 0x1.0p-53;
}

static void philox_normals(uint64_t seed, int64_t n, uint32_t t, double *z)
{
    const double twopi = 6.283185307179586476925286766559;
    for (int64_t p = 0; 2 * p < n; ++p) {
        uint64_t a, b;
        philox_u64_pair(seed, (uint32_t)p, t, 0, 0, &a, &b);
        double r = sqrt(-2.0 * log(u01_open(a)));
        double th = twopi * u01_open(b);
        z[2 * p] = r * cos(th);
        if (2 * p + 1 < n) z[2 * p + 1] = r * sin(th);
    }
}

/* Bootstrap filter for the univariate linear-Gaussian model with systematic
 * resampling, in the production (Philox) mode of the HIP path: the step is
 * particles/core.py:369-383 with Bootstrap (state_space_models.py:326-333),
 * LinearGauss (kalman.py:427-434), Weights (resampling.py:217-226), and the
 * Q62 CDF above.  Used for an end-to-end cross-check that does not depend on
 * a replay tape.  summ[t*4 + {0,1,2,3}] = ESS, log_mean, loglt, rs_flag. */
double orc_toy_filter_philox(const double *y, int64_t T, int64_t N, double rho,
                             double sigmaX, double sigmaY, double sigma0,
                             double ESSrmin, uint64_t seed, double *summ)
{
    const double C = 0.9189385332046727;
    double *X = malloc(sizeof(double) * N), *Xn = malloc(sizeof(double) * N);
    double *lw = malloc(sizeof(double) * N), *W = malloc(sizeof(double) * N);
    double *z = malloc(sizeof(double) * N), *su = malloc(sizeof(double) * N);
    int64_t *A = malloc(sizeof(int64_t) * N);
    double logLt = 0.0, ESS = 0.0, log_mean = 0.0, prev = 0.0;
    const double lsy = log(sigmaY);
    for (int64_t t = 0; t < T; ++t) {
        int rs = 0;
        philox_normals(seed, N, (uint32_t)t, z);
        if (t == 0) {
            for (int64_t n = 0; n < N; ++n) X[n] = 0.0 + sigma0 * z[n];
        } else {
            rs = ESS < (double)N * ESSrmin;
            if (rs) {
                uint64_t a, b;
                philox_u64_pair(seed, 0, (uint32_t)t, 0, 1, &a, &b);
                double u = u01_halfopen(a);
                for (int64_t n = 0; n < N; ++n) su[n] = (u + (double)n) / (double)N;
                orc_inverse_cdf_q62(su, W, N, N, A);
                for (int64_t n = 0; n < N; ++n) Xn[n] = rho * X[A[n]] + sigmaX * z[n];
            } else {
                for (int64_t n = 0; n < N; ++n) Xn[n] = rho * X[n] + sigmaX * z[n];
            }
            double *tmp = X; X = Xn; Xn = tmp;
        }
        double m = -INFINITY;
        for (int64_t n = 0; n < N; ++n) {
            double v = (y[t] - X[n]) / sigmaY;
            double inc = -(v * v) / 2.0 - C - lsy;
            lw[n] = (t == 0 || rs) ? inc : lw[n] + inc;
            if (isnan(lw[n])) lw[n] = -INFINITY;
            if (lw[n] > m) m = lw[n];
        }
        double s = 0.0, s2 = 0.0;
        for (int64_t n = 0; n < N; ++n) { W[n] = exp(lw[n] - m); s += W[n]; }
        for (int64_t n = 0; n < N; ++n) { W[n] /= s; s2 += W[n] * W[n]; }
        prev = log_mean;
        log_mean = m + log(s / (double)N);
        ESS = 1.0 / s2;
        double loglt = (t == 0 || rs) ? log_mean : log_mean - prev;
        logLt += loglt;
        if (summ) {
            summ[4 * t] = ESS; summ[4 * t + 1] = log_mean;
            summ[4 * t + 2] = loglt; summ[4 * t + 3] = (double)rs;
        }
    }
    free(X); free(Xn); free(lw); free(W); free(z); free(su); free(A);
    return logLt;
}

/* ==========================================================================
 * The two-level exact CDF contract of the fused step loop,
 * restated operation for operation so that the HIP kernels can be checked
 * BIT FOR BIT (tests: np.array_equal on ancestors at N = 2^12 .. 2^22).
 *
 * What the contract fixes (and this file therefore repeats):
 *   exp        exp(x) = p 2^k with k = rint(x log2 e) and p = P(x - k ln 2), P the degree-13
 *              polynomial below after a Cody-Waite reduction (the device evaluates the same
 *              IEEE operations; fma() is explicit, the file is built with -ffp-contract=off).
 *              Weights are carried as (p, k) pairs: no maximum is needed to form them.
 *   tile       1024 consecutive particles; K_b = max k_i, e_i = p_i 2^(k_i - K_b) (in [0, 1.42));
 *              partial (K_b, S_b, SS_b) = (K_b, sum e, sum e^2); summation tree: within each
 *              group of 256 consecutive particles slot l of 64 takes the elements 2l, 2l+1,
 *              128+2l, 129+2l in this order (e^2 by fma); a balanced binary tree over the 64
 *              slots; the 4 groups left to right
 *   island     K = max K_b; s = sum S_b 2^(K_b - K), ss = sum SS_b 2^(2 (K_b - K)) (exact
 *              scalings): per slot i of 256 the tiles 4i..4i+3 (and + 1024 c for every further
 *              chunk c) left to right, then the same 64-tree / 4-blocks order
 *   shares     Q_b = rint(min(S_b 2^(K_b-K) / s, 2) 2^52), G_b = sum_{b'<b} Q_b' (exact
 *              integers below 2^53: fp64 carries them)
 *   in a tile  q_i = rint(e_i 2^49), c_j = sum_{i<j} q_i, t_b = sum q_i  (< 2^60)
 *   offspring  parent j of tile b owns the offspring n with
 *              count(G_b + floor(c_j Q_b / t_b)) <= n < count(.. c_{j+1} ..),
 *              count(C) = #{n : ceil(fl(u_n + n) 2^(52-k)) <= C}, N = 2^k
 * Reference semantics being implemented: resampling.py:484-509, :599-610.
 * ========================================================================== */
static const double K_EXP[16] = {
    1.6059043836821613e-10, 2.0876756987868100e-09, 2.5052108385441720e-08,
    2.7557319223985888e-07, 2.7557319223985893e-06, 2.4801587301587302e-05,
    1.9841269841269841e-04, 1.3888888888888889e-03, 8.3333333333333332e-03,
    4.1666666666666664e-02, 1.6666666666666666e-01, 0.5,
    1.4426950408889634074, 6.93147180369123816490e-01, 1.90821492927058770002e-10, -745.2};

/* exp(x) = p 2^k: returns p, writes k (an integer-valued double; -inf and p = 0 for x = -inf) */
double orc_expk(double x, double *kout)
{
    if (!(x > -INFINITY)) { *kout = -INFINITY; return 0.0; }
    const double k = rint(x * K_EXP[12]);
    double r = fma(-k, K_EXP[13], x);
    r = fma(-k, K_EXP[14], r);
    double p = K_EXP[0];
    for (int i = 1; i < 12; ++i) p = fma(p, r, K_EXP[i]);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    *kout = k;
    return p;
}
/* p 2^(k - K): the weight relative to the reference exponent K */
static double scale_pk(double p, double k, double K)
{
    if (!(p > 0.0)) return 0.0;
    double d = k - K;
    if (!(d > -2000.0)) d = -2000.0;
    return ldexp(p, (int)d);
}
double orc_exp_nonpos(double x)          /* exp(x), x <= 0, as the other device paths form it */
{
    double k;
    const double p = orc_expk(x, &k);
    return (x < K_EXP[15]) ? 0.0 : ldexp(p, (int)k);
}
void orc_exp_nonpos_v(const double *x, int64_t n, double *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = orc_exp_nonpos(x[i]);
}
/* W-relative weights of a tile-less vector: e_i = p_i 2^(k_i - K) for a given K */
void orc_weights_pk(const double *lw, int64_t n, double K, double *out)
{
    for (int64_t i = 0; i < n; ++i) {
        double k;
        const double p = orc_expk(lw[i], &k);
        out[i] = scale_pk(p, k, K);
    }
}

/* balanced binary tree over 64 values (in place), pairs (2k, 2k+1) first */
static double tree64(double *v)
{
    for (int w = 64; w > 1; w >>= 1)
        for (int i = 0; i < w / 2; ++i) v[i] = v[2 * i] + v[2 * i + 1];
    return v[0];
}
/* 256 slot values -> ((t0 + t1) + t2) + t3, t_w = tree over slots 64w .. 64w+63 */
static double block_sum256(const double *slot)
{
    double r = 0.0;
    for (int w = 0; w < 4; ++w) {
        double v[64];
        memcpy(v, slot + 64 * w, sizeof v);
        const double t = tree64(v);
        r = w ? r + t : t;
    }
    return r;
}

/* partial (K_b, S_b, SS_b) of every aligned tile of 1024 log-weights; q (N, may be NULL): the
 * integer weights q_i = rint(e_i 2^49) */
void orc_tile_partials(const double *lw, int64_t N, double *pK, double *ps, double *pss, uint64_t *q)
{
    const int64_t nt = (N + 1023) / 1024;
    for (int64_t b = 0; b < nt; ++b) {
        double p[1024], k[1024], K = -INFINITY;
        for (int i = 0; i < 1024; ++i) {
            const int64_t j = b * 1024 + i;
            p[i] = orc_expk(j < N ? lw[j] : -INFINITY, &k[i]);
            if (k[i] > K) K = k[i];
        }
        double s1[256], s2[256];
        for (int th = 0; th < 256; ++th) {
            double a = 0.0, qq = 0.0;
            /* slot th holds the pairs (2l, 2l+1) and (128+2l, 128+2l+1) of its group of 256 */
            const int g0 = (th & ~63) * 4 + 2 * (th & 63);
            const int idx[4] = {g0, g0 + 1, g0 + 128, g0 + 129};
            for (int c = 0; c < 4; ++c) {
                const double e = scale_pk(p[idx[c]], k[idx[c]], K);
                a += e;
                qq = fma(e, e, qq);
                if (q && b * 1024 + idx[c] < N)
                    q[b * 1024 + idx[c]] = (uint64_t)rint(e * 562949953421312.0);   /* 2^49 */
            }
            s1[th] = a;
            s2[th] = qq;
        }
        pK[b] = K;
        ps[b] = block_sum256(s1);
        pss[b] = block_sum256(s2);
    }
}

/* island level: out = {K, s, ss, ESS, rs = 1/s}; Q, G (nt each; integer-valued doubles) may be NULL */
void orc_two_level_reduce(const double *pK, const double *ps, const double *pss, int64_t nt,
                          double *out, double *Q, double *G)
{
    double K = -INFINITY;
    for (int64_t b = 0; b < nt; ++b)
        if (pK[b] > K) K = pK[b];
    const int64_t nchunks = (nt + 1023) / 1024;
    double s1[256], s2[256];
    for (int th = 0; th < 256; ++th) {
        double a = 0.0, q = 0.0;
        for (int64_t c = 0; c < nchunks; ++c)
            for (int k = 0; k < 4; ++k) {
                const int64_t b = c * 1024 + 4 * th + k;
                if (b >= nt) continue;
                double d = pK[b] - K;
                if (!(d > -2000.0)) d = -2000.0;
                a = a + ldexp(ps[b], (int)d);
                q = q + ldexp(pss[b], 2 * (int)d);
            }
        s1[th] = a;
        s2[th] = q;
    }
    const double s = block_sum256(s1), ss = block_sum256(s2);
    const int bad = !(K > -INFINITY) || !(K < INFINITY);
    out[0] = K;
    out[1] = s;
    out[2] = ss;
    out[3] = bad ? NAN : (s * s) / ss;                 /* resampling.py:226 */
    out[4] = bad ? NAN : 1.0 / s;
    if (!Q) return;
    double g = 0.0;
    for (int64_t b = 0; b < nt; ++b) {
        double d = pK[b] - K;
        if (!(d > -2000.0)) d = -2000.0;
        const double w = ldexp(ps[b], (int)d) * out[4];
        const double qb = (w > 0.0) ? rint(fmin(w, 2.0) * 4503599627370496.0) : 0.0;   /* 2^52 */
        Q[b] = qb;
        G[b] = g;
        g += qb;                                       /* exact: integers below 2^53 */
    }
}

/* count(C) = #{ n < N : ceil(fl(u_n + n) 2^sh) <= C }, N = 2^k, sh = 52 - k
 * (systematic: u_n = u[0]; stratified: u_n = u[n]) */
static int64_t count_pow2(uint64_t C, const double *u, int stratified, int k, int64_t N)
{
    const int sh = 52 - k;
    const uint64_t nc = C >> sh;
    if (nc >= (uint64_t)N) return N;
    const double un = stratified ? u[nc] : u[0];
    const uint64_t T = (uint64_t)ceil((un + (double)(int64_t)nc) * ldexp(1.0, sh));
    return (int64_t)nc + (T <= C ? 1 : 0);
}

/* any N (not a power of two): su_n = fl(fl(u_n + n) / N), T_n = ceil(su_n 2^52) -- non-decreasing in n
 * (fl(u_n + n) lies in [n, n + 1], rounding and ceil are monotone): count(C) by bisection on the
 * definition (systematic: u_n = u[0]; stratified: u_n = u[n]) */
static int64_t count_general(uint64_t C, const double *u, int stratified, int64_t N)
{
    int64_t lo = 0, hi = N;                            /* T_n <= C on [0, lo), > C on [hi, N) */
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        const double un = stratified ? u[mid] : u[0];
        const double v = (un + (double)mid) / (double)N;
        const uint64_t T = (v > 0.0) ? (uint64_t)ceil(fmin(v, 2.0) * 4503599627370496.0) : 0;
        if (T <= C) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* multinomial: the N sorted uniforms su themselves (resampling.py:512-537), thresholds
 * T_n = ceil(su_n 2^52): count(C) = #{ n < N : T_n <= C } by bisection (T_n is non-decreasing) */
static int64_t count_sorted(uint64_t C, const double *su, int64_t N)
{
    int64_t lo = 0, hi = N;                            /* T_n <= C on [0, lo), > C on [hi, N) */
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        const double v = su[mid];
        const uint64_t T = (v > 0.0) ? (uint64_t)ceil(fmin(v, 2.0) * 4503599627370496.0) : 0;
        if (T <= C) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* The whole contract: ancestors A (N) from the log-weights of the parents.  scheme: 0
 * multinomial (u: the N sorted uniforms), 1 stratified (u: N uniforms), 2 systematic (u: 1
 * uniform).  red (5): K, s, ss, ESS, 1/s.
 * Returns 0, or 1 if N <= 1024 or N > 2^30 (the path does not apply). */
int orc_inverse_cdf_2level(const double *lw, int64_t N, int scheme, const double *u,
                           int64_t *A, double *red)
{
    int k = -1;                                        /* log2 N, or -1: the general counts */
    for (int i = 0; i < 40; ++i)
        if (((int64_t)1 << i) == N) k = i;
    if (N <= 1024 || N > ((int64_t)1 << 30)) return 1;
    const int64_t nt = (N + 1023) / 1024;              /* the last tile may be ragged: -inf beyond N */
    double *pK = malloc(sizeof(double) * nt), *ps = malloc(sizeof(double) * nt),
           *pss = malloc(sizeof(double) * nt), *Q = malloc(sizeof(double) * nt),
           *G = malloc(sizeof(double) * nt);
    uint64_t *q = malloc(sizeof(uint64_t) * N);
    orc_tile_partials(lw, N, pK, ps, pss, q);
    orc_two_level_reduce(pK, ps, pss, nt, red, Q, G);
    const int strat = scheme == 1;
    int64_t prev = 0;                                  /* first offspring of parent j - 1 */
    for (int64_t b = 0; b < nt; ++b) {
        uint64_t tb = 0;
        for (int i = 0; i < 1024 && b * 1024 + i < N; ++i) tb += q[b * 1024 + i];
        const uint64_t Qb = (uint64_t)Q[b], Gb = (uint64_t)G[b];
        uint64_t c = 0;
        for (int i = 0; i < 1024 && b * 1024 + i < N; ++i) {
            const int64_t j = b * 1024 + i;
            int64_t ns;
            if (j == 0) ns = 0;
            else {
                uint64_t pos;
                if (c == 0) pos = 0;
                else if (c >= tb) pos = Qb;
                else pos = (uint64_t)(((unsigned __int128)c * Qb) / tb);
                ns = scheme == 0 ? count_sorted(Gb + pos, u, N)
                   : (k >= 0 ? count_pow2(Gb + pos, u, strat, k, N) : count_general(Gb + pos, u, strat, N));
            }
            /* offspring prev .. ns-1 belong to parent j - 1 */
            for (int64_t n = prev; n < ns; ++n) A[n] = j - 1;
            if (ns > prev) prev = ns;
            c += q[j];
        }
    }
    for (int64_t n = prev; n < N; ++n) A[n] = N - 1;   /* resampling.py:505-508 clamp */
    free(pK); free(ps); free(pss); free(Q); free(G); free(q);
    return 0;
}
