// smc_math.h -- lean fp64 elementary functions for the ranges the SMC kernels
// actually use.  The step loop is bound by fp64 VALU issue on MI355X (4 cycles
// per wave instruction), and the general-purpose libm entry points spend a
// third of their instructions on argument ranges that cannot occur here, so:
//
//   smc_exp_nonpos(x)   exp(x) for x <= 0 (all our log-weight differences):
//                       result in [0,1], no overflow path
//   smc_sincospi_02(a)  sin(pi a), cos(pi a) for a in [0, 2] (Box-Muller angle)
//
// exp: <= 1 ulp; sincospi: absolute error < 2.3e-16, <= 2 ulp away from zeros
// (tests/test_math_accuracy.py checks them against libm through the
// emulator build).  Explicit fma() throughout: the translation unit is built
// with -ffp-contract=off.
#pragma once
#include "smc_platform.h"

#include <cmath>

// Polynomial coefficients live in constant memory: the compiler fetches them
// with scalar loads and feeds them to v_fma_f64 as SGPR operands.  Written as
// literals they would each cost a v_mov_b64 (fp64 VOP3 encodings cannot carry a
// 64-bit literal), i.e. double the instruction count of every polynomial.
#ifdef SMC_EMULATE
#define SMC_CONST static const
#else
#define SMC_CONST static __constant__ const
#endif
// p*r + K with K taken straight from an SGPR pair (VOP3 v_fma_f64).  Left to
// itself the compiler picks the two-address v_fmac_f64, whose addend must sit in
// the destination VGPR, and spends two v_mov_b32 per coefficient copying K there.
#ifdef SMC_EMULATE
#define SMC_FMA_K(p, r, K) fma((p), (r), (K))
#else
__device__ __forceinline__ double smc_fma_k(double p, double r, double K)
{
    double o;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(p), "v"(r), "s"(K));
    return o;
}
#define SMC_FMA_K(p, r, K) smc_fma_k((p), (r), (K))
#endif

SMC_CONST double smc_k_exp[16] = {
    1.6059043836821613e-10, 2.0876756987868100e-09, 2.5052108385441720e-08,
    2.7557319223985888e-07, 2.7557319223985893e-06, 2.4801587301587302e-05,
    1.9841269841269841e-04, 1.3888888888888889e-03, 8.3333333333333332e-03,
    4.1666666666666664e-02, 1.6666666666666666e-01, 0.5,
    1.4426950408889634074, 6.93147180369123816490e-01, 1.90821492927058770002e-10, -745.2};
SMC_CONST double smc_k_log[10] = {
    1.531383769920937332e-01, 2.222219843214978396e-01, 3.999999999940941908e-01,
    1.479819860511658591e-01, 1.818357216161805012e-01, 2.857142874366239149e-01,
    6.666666666666735130e-01, 0.70710678118654752440, 6.93147180369123816490e-01,
    1.90821492927058770002e-10};
SMC_CONST double smc_k_sc[18] = {
    2.8114572543455206e-15, -7.6471637318198164e-13, 1.6059043836821613e-10,
    -2.5052108385441720e-08, 2.7557319223985893e-06, -1.9841269841269841e-04,
    8.3333333333333332e-03, -1.6666666666666666e-01,
    -1.5619206968586225e-16, 4.7794773323873853e-14, -1.1470745597729725e-11,
    2.0876756987868100e-09, -2.7557319223985888e-07, 2.4801587301587302e-05,
    -1.3888888888888889e-03, 4.1666666666666664e-02,
    3.14159265358979311600e+00, 1.22464679914735317723e-16};

// exp(x) = p 2^k without forming 2^k: k = round(x / ln 2) (returned as an integer-valued double),
// p = P(x - k ln 2) in [0.7071, 1.4143].  The two-level CDF path carries weights as such pairs:
// a tile's (or the island's) reference is then a power of two, and rescaling a sum of weights to
// another reference is an exact ldexp instead of another exp (oracle.c orc_expk).
// x = -inf gives k = -inf and p = NaN: callers select (p, k) = (0, -inf) for it.
__host__ __device__ __forceinline__ double smc_expk(double x, double& k)
{
    const double* K = smc_k_exp;
    k = rint(x * K[12]);
    double r = fma(-k, K[13], x);                          // ln2 high part
    r = fma(-k, K[14], r);                                 // ln2 low part
    double p = K[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) p = SMC_FMA_K(p, r, K[i]);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return p;
}
// p 2^(k - K), K >= k: the weight relative to the reference exponent K (0 for p = 0; the
// exponent difference is clamped where the result is 0 anyway, and for -inf - -inf)
__host__ __device__ __forceinline__ double smc_scale_pk(double p, double k, double K)
{
    double d = k - K;
    d = (d > -2000.0) ? d : -2000.0;
    return ldexp(p, (int)d);
}

__host__ __device__ __forceinline__ double smc_exp_nonpos(double x)
{
    const double* K = smc_k_exp;
    // k = round(x / ln2), r = x - k ln2 (Cody-Waite, two-part ln2), |r| <= ln2/2
    const double k = rint(x * K[12]);
    double r = fma(-k, K[13], x);                          // ln2 high part
    r = fma(-k, K[14], r);                                 // ln2 low part
    // exp(r) by its Taylor polynomial of degree 13 (|r|^14/14! < 5e-18): 1/13! ... 1/2!
    double p = K[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) p = SMC_FMA_K(p, r, K[i]);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    // x < -745.2 underflows to 0 (also covers -inf, where r would be NaN)
    const double y = ldexp(p, (int)k);
    return (x < K[15]) ? 0.0 : y;
}

// log(x) for a positive, finite, normal x (the Box-Muller uniform lies in
// [2^-53, 1)).  Classic argument reduction x = 2^k m, m in [sqrt(1/2), sqrt 2),
// f = m-1, s = f/(2+f), log(1+f) = f - hfsq + s (hfsq + R(s^2)) with the
// degree-14 Remez polynomial R of fdlibm's e_log.c (Sun Microsystems, public
// algorithm); error < 1 ulp.  ~40 instructions against ~100 for the general
// libm entry point (which also handles 0, inf, NaN, subnormals).
__host__ __device__ __forceinline__ double smc_log_pos(double x)
{
    const double* K = smc_k_log;
    int e;
    double m = frexp(x, &e);                       // m in [0.5, 1)
    const bool lo = m < K[7];
    m = lo ? m + m : m;                            // [sqrt(1/2), sqrt 2)
    e = lo ? e - 1 : e;
    const double k = (double)e;
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * SMC_FMA_K(SMC_FMA_K(w * K[0], 1.0, K[1]) , w, K[2]);   // Lg6, Lg4, Lg2
    const double t2 = z * SMC_FMA_K(SMC_FMA_K(SMC_FMA_K(w * K[3], 1.0, K[4]), w, K[5]), w, K[6]);
    const double R = t1 + t2;
    const double hfsq = 0.5 * f * f;
    // k ln2_hi - ((hfsq - (s (hfsq+R) + k ln2_lo)) - f)
    return fma(k, K[8], -((hfsq - fma(s, hfsq + R, k * K[9])) - f));
}

__host__ __device__ __forceinline__ void smc_sincospi_02(double a, double* sn, double* cs)
{
    const double* K = smc_k_sc;
    // a = q/2 + r, q in {0..4}, |r| <= 1/4 ; x = pi r in [-pi/4, pi/4]
    const double qd = rint(a + a);
    const double r = fma(-0.5, qd, a);
    const double x = fma(r, K[16], r * K[17]);             // pi = hi + lo
    const double z = x * x;
    // sin x = x + x z S(z), S = Taylor up to x^17 ; cos x = 1 - z/2 + z^2 C(z) up to x^18
    double s = K[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s = SMC_FMA_K(s, z, K[i]);
    const double sx = fma(x * z, s, x);
    double c = K[8];
#pragma unroll
    for (int i = 9; i < 16; ++i) c = SMC_FMA_K(c, z, K[i]);
    const double cx = fma(z * z, c, fma(-0.5, z, 1.0));
    const int q = (int)qd;
    // rotate by q quarter turns: (sin, cos)(x + q pi/2)
    const bool swap = q & 1;
    const double s0 = swap ? cx : sx, c0 = swap ? sx : cx;
    *sn = (q & 2) ? -s0 : s0;
    *cs = ((q + 1) & 2) ? -c0 : c0;
}
