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

__host__ __device__ __forceinline__ double smc_exp_nonpos(double x)
{
    // k = round(x / ln2), r = x - k ln2 (Cody-Waite, two-part ln2), |r| <= ln2/2
    const double k = rint(x * 1.4426950408889634074);
    double r = fma(-k, 6.93147180369123816490e-01, x);     // ln2 high part
    r = fma(-k, 1.90821492927058770002e-10, r);            // ln2 low part
    // exp(r) by its Taylor polynomial of degree 13 (|r|^14/14! < 5e-18)
    double p = 1.6059043836821613e-10;                     // 1/13!
    p = fma(p, r, 2.0876756987868100e-09);                 // 1/12!
    p = fma(p, r, 2.5052108385441720e-08);                 // 1/11!
    p = fma(p, r, 2.7557319223985888e-07);                 // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);                 // 1/9!
    p = fma(p, r, 2.4801587301587302e-05);                 // 1/8!
    p = fma(p, r, 1.9841269841269841e-04);                 // 1/7!
    p = fma(p, r, 1.3888888888888889e-03);                 // 1/6!
    p = fma(p, r, 8.3333333333333332e-03);                 // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);                 // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);                 // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    // x < -745.2 underflows to 0 (also covers -inf, where r would be NaN)
    const double y = ldexp(p, (int)k);
    return (x < -745.2) ? 0.0 : y;
}

// log(x) for a positive, finite, normal x (the Box-Muller uniform lies in
// [2^-53, 1)).  Classic argument reduction x = 2^k m, m in [sqrt(1/2), sqrt 2),
// f = m-1, s = f/(2+f), log(1+f) = f - hfsq + s (hfsq + R(s^2)) with the
// degree-14 Remez polynomial R of fdlibm's e_log.c (Sun Microsystems, public
// algorithm); error < 1 ulp.  ~40 instructions against ~100 for the general
// libm entry point (which also handles 0, inf, NaN, subnormals).
__host__ __device__ __forceinline__ double smc_log_pos(double x)
{
    int e;
    double m = frexp(x, &e);                       // m in [0.5, 1)
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;                            // [sqrt(1/2), sqrt 2)
    e = lo ? e - 1 : e;
    const double k = (double)e;
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01),
                              3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                     2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t1 + t2;
    const double hfsq = 0.5 * f * f;
    // k ln2_hi - ((hfsq - (s (hfsq+R) + k ln2_lo)) - f)
    return fma(k, 6.93147180369123816490e-01,
               -((hfsq - fma(s, hfsq + R, k * 1.90821492927058770002e-10)) - f));
}

__host__ __device__ __forceinline__ void smc_sincospi_02(double a, double* sn, double* cs)
{
    // a = q/2 + r, q in {0..4}, |r| <= 1/4 ; x = pi r in [-pi/4, pi/4]
    const double qd = rint(a + a);
    const double r = fma(-0.5, qd, a);
    const double x = fma(r, 3.14159265358979311600e+00, r * 1.22464679914735317723e-16);
    const double z = x * x;
    // sin x = x + x z S(z), S = Taylor up to x^17 ; cos x = 1 - z/2 + z^2 C(z) up to x^18
    double s = 2.8114572543455206e-15;                     //  1/17!
    s = fma(s, z, -7.6471637318198164e-13);                // -1/15!
    s = fma(s, z, 1.6059043836821613e-10);                 //  1/13!
    s = fma(s, z, -2.5052108385441720e-08);                // -1/11!
    s = fma(s, z, 2.7557319223985893e-06);                 //  1/9!
    s = fma(s, z, -1.9841269841269841e-04);                // -1/7!
    s = fma(s, z, 8.3333333333333332e-03);                 //  1/5!
    s = fma(s, z, -1.6666666666666666e-01);                // -1/3!
    const double sx = fma(x * z, s, x);
    double c = -1.5619206968586225e-16;                    // -1/18!
    c = fma(c, z, 4.7794773323873853e-14);                 //  1/16!
    c = fma(c, z, -1.1470745597729725e-11);                // -1/14!
    c = fma(c, z, 2.0876756987868100e-09);                 //  1/12!
    c = fma(c, z, -2.7557319223985888e-07);                // -1/10!
    c = fma(c, z, 2.4801587301587302e-05);                 //  1/8!
    c = fma(c, z, -1.3888888888888889e-03);                // -1/6!
    c = fma(c, z, 4.1666666666666664e-02);                 //  1/4!
    const double cx = fma(z * z, c, fma(-0.5, z, 1.0));
    const int q = (int)qd;
    // rotate by q quarter turns: (sin, cos)(x + q pi/2)
    const bool swap = q & 1;
    const double s0 = swap ? cx : sx, c0 = swap ? sx : cx;
    *sn = (q & 2) ? -s0 : s0;
    *cs = ((q + 1) & 2) ? -c0 : c0;
}
