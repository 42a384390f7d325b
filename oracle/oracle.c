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
