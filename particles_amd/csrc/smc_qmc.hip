// smc_qmc.hip -- building blocks of SQMC (particles/core.py:339-349): quasi-random points
// (rqmc.py: scipy.stats.qmc.Sobol behind safe_generate) and the inverse-CDF transform of a
// Normal kernel (FeynmanKac.Gamma0 / Gamma -> ProbDist.ppf, state_space_models.py:335-340,
// distributions.py:276-277).  The argsort of hilbert_sort (d = 1) lives in smc_sort.hip.
#include "smc_internal.h"
#include "smc_device.h"
#include "smc_qmc.h"

__global__ void __launch_bounds__(SMC_BLOCK)
k_normal_ppf(const double* u, i64 us, const double* loc, i64 ls, const double* scale, i64 ss, i64 N,
             double* out)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < N) out[i] = smc_ndtri(u[i * us]) * scale[i * ss] + loc[i * ls];     // scipy: _ppf(q) * scale + loc
}

extern "C" int smc_normal_ppf(smc_ctx* ctx, const double* u, int64_t u_stride, const double* loc,
                              int64_t loc_stride, const double* scale, int64_t scale_stride,
                              int64_t N, double* out)
{
    SMC_REQUIRE(ctx && u && loc && scale && out, "null argument");
    SMC_REQUIRE(N > 0, "N must be positive");
    SMC_LAUNCH(k_normal_ppf, dim3((unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK),
               ctx->stream, u, (i64)u_stride, loc, (i64)loc_stride, scale, (i64)scale_stride, (i64)N,
               out);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

// ---------------------------------------------------------------------------
// Sobol' points.  Point n (Gray-code order, as scipy's engine walks the sequence) is the XOR
// of the direction numbers v_k over the set bits k of gray(n) = n ^ (n >> 1): every point is
// independent of the others, one thread per (point, coordinate).
// ---------------------------------------------------------------------------
#define SOBOL_MAXD 10

struct SobolTable {
    u32 v[SOBOL_MAXD][SOBOL_BITS];
    u32 shift[SOBOL_MAXD];
};

// order_k < 0: row j holds point j.  order_k = log2 N >= 0: row j holds the point with the j-th
// smallest FIRST coordinate.  Dimension 1's direction numbers are v_b = 2^(29-b), so the top
// log2 N bits of x_0 are bitrev(gray(n)) ^ (the shift's top bits): the sorted order is known in
// closed form and SQMC's argsort(u[:, 0]) (core.py:343) needs no sort.
__global__ void __launch_bounds__(SMC_BLOCK)
k_sobol(const SobolTable tb, i64 N, int d, int safe, int order_k, double* out)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i >= N * d) return;
    const i64 row = i / d;
    const int c = (int)(i - row * d);
    u64 g;
    if (order_k < 0) {
        g = (u64)row ^ ((u64)row >> 1);
    } else if (order_k == 0) {
        g = 0;
    } else {
        const u32 top = (u32)row ^ (tb.shift[0] >> (SOBOL_BITS - order_k));
        g = (u64)(__brev(top) >> (32 - order_k));                         // gray(n) of that point
    }
    u32 x = tb.shift[c];
    for (int k = 0; k < SOBOL_BITS && g; ++k, g >>= 1)
        if (g & 1) x ^= tb.v[c][k];
    double u = (double)x * (1.0 / (double)(1u << SOBOL_BITS));
    if (safe) u = 0.5 + (1.0 - 1e-10) * (u - 0.5);                       // rqmc.py:9-13
    out[i] = u;
}

// Joe & Kuo's (s, a, m_1..m_s) for dimensions 2..10 (new-joe-kuo-6.21201, the table scipy ships)
static const int SOBOL_S[SOBOL_MAXD + 1] = {0, 0, 1, 2, 3, 3, 4, 4, 5, 5, 5};
static const int SOBOL_A[SOBOL_MAXD + 1] = {0, 0, 0, 1, 1, 2, 1, 4, 2, 4, 7};
static const int SOBOL_M[SOBOL_MAXD + 1][5] = {{0}, {0}, {1}, {1, 3}, {1, 3, 1}, {1, 1, 1},
                                               {1, 1, 3, 3}, {1, 3, 5, 13}, {1, 1, 5, 5, 17},
                                               {1, 1, 5, 5, 5}, {1, 1, 7, 11, 19}};

static int sobol_launch(smc_ctx* ctx, int64_t N, int32_t d, int32_t scramble, int32_t safe,
                        uint64_t counter, int order_k, double* out, const u64* seed_override = nullptr)
{
    SMC_REQUIRE(ctx && out, "null argument");
    SMC_REQUIRE(N > 0 && N <= ((int64_t)1 << SOBOL_BITS), "N must be in [1, 2^30]");
    SMC_REQUIRE(d >= 1 && d <= SOBOL_MAXD, "smc_sobol: 1 <= d <= 10");
    SobolTable tb;
    for (int dim = 1; dim <= d; ++dim) {
        u32 m[SOBOL_BITS];
        if (dim == 1) {
            for (int k = 0; k < SOBOL_BITS; ++k) m[k] = 1u;
        } else {
            const int s = SOBOL_S[dim], a = SOBOL_A[dim];
            for (int k = 0; k < s && k < SOBOL_BITS; ++k) m[k] = (u32)SOBOL_M[dim][k];
            for (int k = s; k < SOBOL_BITS; ++k) {
                u32 v = m[k - s] ^ (m[k - s] << s);
                for (int j = 1; j < s; ++j)
                    if ((a >> (s - 1 - j)) & 1) v ^= m[k - j] << j;
                m[k] = v;
            }
        }
        for (int k = 0; k < SOBOL_BITS; ++k) tb.v[dim - 1][k] = m[k] << (SOBOL_BITS - 1 - k);
        u32 sh = 0u;
        if (scramble) {                       // digital shift: one Philox word per coordinate
            sh = smc_sobol_shift(seed_override ? *seed_override : (u64)ctx->seed, counter, (u32)(dim - 1));
        }
        tb.shift[dim - 1] = sh;
    }
    for (int dim = d; dim < SOBOL_MAXD; ++dim) {
        tb.shift[dim] = 0u;
        for (int k = 0; k < SOBOL_BITS; ++k) tb.v[dim][k] = 0u;
    }
    const i64 tot = (i64)N * d;
    SMC_LAUNCH(k_sobol, dim3((unsigned)((tot + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK),
               ctx->stream, tb, (i64)N, (int)d, (int)safe, order_k, out);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

// smc_internal.h: the point set `counter` of the stream keyed by `seed` (scramble = safe = 1), rows in Gray-code
// order (sorted = 0) or sorted by the first coordinate (N = 2^k) -- the fused SQMC step of the multivariate filters
int smc_sobol_points(smc_ctx* ctx, unsigned long long seed, long long N, int d, unsigned long long counter, int sorted,
                     double* out)
{
    int k = -1;
    if (sorted) {
        SMC_REQUIRE(N > 0 && (N & (N - 1)) == 0, "smc_sobol_points: N must be a power of two");
        k = 0;
        while (((long long)1 << k) < N) ++k;
    }
    const u64 s = (u64)seed;
    return sobol_launch(ctx, N, d, 1, 1, counter, k, out, &s);
}

extern "C" int smc_sobol(smc_ctx* ctx, int64_t N, int32_t d, int32_t scramble, int32_t safe,
                         uint64_t counter, double* out)
{
    return sobol_launch(ctx, N, d, scramble, safe, counter, -1, out);
}

extern "C" int smc_sobol_sorted(smc_ctx* ctx, int64_t N, int32_t d, int32_t scramble, int32_t safe,
                                uint64_t counter, double* out)
{
    SMC_REQUIRE(N > 0 && (N & (N - 1)) == 0, "smc_sobol_sorted: N must be a power of two");
    int k = 0;
    while (((int64_t)1 << k) < N) ++k;
    return sobol_launch(ctx, N, d, scramble, safe, counter, k, out);
}
