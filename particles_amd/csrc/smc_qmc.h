// smc_qmc.h -- device pieces shared by the stand-alone SQMC operators (smc_qmc.hip) and the fused SQMC
// step of the filter (smc_filter_sqmc.h): the inverse normal CDF behind ProbDist.ppf and the first two
// coordinates of the Sobol' sequence.
#pragma once
#include "smc_device.h"

// ---------------------------------------------------------------------------
// ndtri: Cephes' inverse of the standard normal CDF -- the routine behind
// scipy.special.ndtri / scipy.stats.norm.ppf -- restated: central region by a
// rational function of (y - 1/2)^2, tails by rational functions of
// 1/sqrt(-2 log y).  Only IEEE + - * / in the central region (bit-identical to
// scipy there); the tails add the device's log / sqrt (<= 1 ulp each).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double ndtri_polevl(double x, const double* c, int n)
{
    double a = c[0];
    for (int i = 1; i <= n; ++i) a = a * x + c[i];
    return a;
}
__device__ __forceinline__ double ndtri_p1evl(double x, const double* c, int n)
{
    double a = x + c[0];
    for (int i = 1; i < n; ++i) a = a * x + c[i];
    return a;
}
// scipy.special.ndtri (Cephes ndtri.c), in two pieces so that a workgroup can evaluate the TAIL branch -- 27 % of
// uniform arguments, two logarithms, a square root and three divisions: ten times the centre's cost -- on lanes it has
// packed with tail arguments (k_sq_permute) instead of on every wave that holds one.  smc_ndtri(y0) is the same
// arithmetic in the same order as before the split.
//   smc_ndtri_centre(y0, tail): the result for y0 outside (0, 1), at its ends and in the central region
//                               (exp(-2) < y0 <= 1 - exp(-2)); tail = true and an unspecified value otherwise
//   smc_ndtri_tail(y0):         the result where smc_ndtri_centre said tail
__device__ inline double smc_ndtri_centre(const double y0, bool& tail)
{
    const double P0[5] = {-5.99633501014107895267E1, 9.80010754185999661536E1,
                          -5.66762857469070293439E1, 1.39312609387279679503E1,
                          -1.23916583867381258016E0};
    const double Q0[8] = {1.95448858338141759834E0, 4.67627912898881538453E0,
                          8.63602421390890590575E1, -2.25462687854119370527E2,
                          2.00260212380060660359E2, -8.20372256168333339912E1,
                          1.59056225126211695515E1, -1.18331621121330003142E0};
    const double s2pi = 2.50662827463100050242E0, em2 = 0.13533528323661269189;   // sqrt(2 pi), exp(-2)
    tail = false;
    if (y0 == 0.0) return -INFINITY;
    if (y0 == 1.0) return INFINITY;
    if (!(y0 > 0.0 && y0 < 1.0)) return NAN;
    double y = y0;
    if (y > 1.0 - em2) y = 1.0 - y;
    if (y > em2) {
        y = y - 0.5;
        const double y2 = y * y;
        const double x = y + y * (y2 * ndtri_polevl(y2, P0, 4) / ndtri_p1evl(y2, Q0, 8));
        return x * s2pi;
    }
    tail = true;
    return 0.0;
}
__device__ inline double smc_ndtri_tail(const double y0)
{
    const double P1[9] = {4.05544892305962419923E0, 3.15251094599893866154E1,
                          5.71628192246421288162E1, 4.40805073893200834700E1,
                          1.46849561928858024014E1, 2.18663306850790267539E0,
                          -1.40256079171354495875E-1, -3.50424626827848203418E-2,
                          -8.57456785154685413611E-4};
    const double Q1[8] = {1.57799883256466749731E1, 4.53907635128879210584E1,
                          4.13172038254672030440E1, 1.50425385692907503408E1,
                          2.50464946208309415979E0, -1.42182922854787788574E-1,
                          -3.80806407691578277194E-2, -9.33259480895457427372E-4};
    const double P2[9] = {3.23774891776946035970E0, 6.91522889068984211695E0,
                          3.93881025292474443415E0, 1.33303460815807542389E0,
                          2.01485389549179081538E-1, 1.23716634817820021358E-2,
                          3.01581553508235416007E-4, 2.65806974686737550832E-6,
                          6.23974539184983293730E-9};
    const double Q2[8] = {6.02427039364742014255E0, 3.67983563856160859403E0,
                          1.37702099489081330271E0, 2.16236993594496635890E-1,
                          1.34204006088543189037E-2, 3.28014464682127739104E-4,
                          2.89247864745380683936E-6, 6.79019408009981274425E-9};
    const double em2 = 0.13533528323661269189;
    bool neg = true;
    double y = y0;
    if (y > 1.0 - em2) { y = 1.0 - y; neg = false; }
    double x = sqrt(-2.0 * log(y));
    const double x0 = x - log(x) / x;
    const double z = 1.0 / x;
    const double x1 = (x < 8.0) ? z * ndtri_polevl(z, P1, 8) / ndtri_p1evl(z, Q1, 8)
                                : z * ndtri_polevl(z, P2, 8) / ndtri_p1evl(z, Q2, 8);
    x = x0 - x1;
    return neg ? -x : x;
}
__device__ inline double smc_ndtri(double y0)
{
    bool tail;
    const double c = smc_ndtri_centre(y0, tail);
    return tail ? smc_ndtri_tail(y0) : c;
}


#define SOBOL_BITS 30

// rqmc.py:9-13 safe_generate: points pulled away from 0 and 1 by a relative 1e-10
__device__ __forceinline__ double smc_sobol_safe(const u32 x)
{
    const double u = (double)x * (1.0 / (double)(1u << SOBOL_BITS));
    return 0.5 + (1.0 - 1e-10) * (u - 0.5);
}
// Coordinates 1 and 2 of the Sobol' point whose Gray code is g, digitally shifted: dimension 1 has the
// direction numbers v_k = 2^(29-k), dimension 2 (Joe & Kuo: s = 1, a = 0, m_1 = 1) m_k = m_{k-1} ^ (m_{k-1} << 1),
// v_k = m_k 2^(29-k) -- the same integers sobol_launch (smc_qmc.hip) tabulates.
__device__ __forceinline__ void smc_sobol2(u32 g, const u32 sh0, const u32 sh1, u32& x0, u32& x1)
{
    x0 = sh0;
    x1 = sh1;
    u32 m = 1u;
    for (int k = 0; k < SOBOL_BITS && g; ++k, g >>= 1) {
        if (g & 1u) { x0 ^= 1u << (SOBOL_BITS - 1 - k); x1 ^= m << (SOBOL_BITS - 1 - k); }
        m ^= m << 1;
    }
}
// Gray code of the point with the `row`-th smallest FIRST coordinate among the first 2^k points (k >= 1):
// the top k bits of x_0 are bitrev(gray(n)) ^ (the shift's top bits), so the sorted order is known in
// closed form (k_sobol's order_k branch)
__device__ __forceinline__ u32 smc_sobol_sorted_gray(const u32 row, const u32 sh0, const int k)
{
    const u32 top = row ^ (sh0 >> (SOBOL_BITS - k));
    return __brev(top) >> (32 - k);
}
// the digital shift of coordinate c for the point set `counter` (sobol_launch: one Philox word per coordinate)
__host__ __device__ __forceinline__ u32 smc_sobol_shift(const u64 seed, const u64 counter, const u32 c)
{
    u64 x0, x1;
    smc_philox(c, (u32)counter, (u32)(counter >> 32), SMC_STREAM_RESAMPLE, seed, x0, x1);
    return (u32)(x0 >> (64 - SOBOL_BITS));
}
