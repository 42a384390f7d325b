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

// The coefficient tables are used by functions the host can call too (the debug entry points of smc_api.hip).  A
// static device variable that a __host__ __device__ function names is externalised by the compiler, and an external
// symbol of a -fPIC code object is addressed through the GOT: one more dependent scalar load in front of every
// s_load of a coefficient (k_propagate: three such chains on its common path).  So each table exists twice -- a
// device copy that only a __device__ accessor names, a host copy -- and SMC_K(name) picks by compilation pass.
#ifdef SMC_EMULATE
#define SMC_K_TABLE(name, n, ...) static const double name[n] = {__VA_ARGS__};
#define SMC_K(name) name
#else
#define SMC_K_TABLE(name, n, ...)                                          \
    static __constant__ const double name##_dev[n] = {__VA_ARGS__};        \
    static const double name##_host[n] = {__VA_ARGS__};                    \
    __device__ __forceinline__ const double* name##_ptr() { return name##_dev; }
#if defined(__HIP_DEVICE_COMPILE__)
#define SMC_K(name) name##_ptr()
#else
#define SMC_K(name) name##_host
#endif
#endif

SMC_K_TABLE(smc_k_exp, 16,
    1.6059043836821613e-10, 2.0876756987868100e-09, 2.5052108385441720e-08,
    2.7557319223985888e-07, 2.7557319223985893e-06, 2.4801587301587302e-05,
    1.9841269841269841e-04, 1.3888888888888889e-03, 8.3333333333333332e-03,
    4.1666666666666664e-02, 1.6666666666666666e-01, 0.5,
    1.4426950408889634074, 6.93147180369123816490e-01, 1.90821492927058770002e-10, -745.2)
SMC_K_TABLE(smc_k_log, 10,
    1.531383769920937332e-01, 2.222219843214978396e-01, 3.999999999940941908e-01,
    1.479819860511658591e-01, 1.818357216161805012e-01, 2.857142874366239149e-01,
    6.666666666666735130e-01, 0.70710678118654752440, 6.93147180369123816490e-01,
    1.90821492927058770002e-10)
SMC_K_TABLE(smc_k_sc, 18,
    2.8114572543455206e-15, -7.6471637318198164e-13, 1.6059043836821613e-10,
    -2.5052108385441720e-08, 2.7557319223985893e-06, -1.9841269841269841e-04,
    8.3333333333333332e-03, -1.6666666666666666e-01,
    -1.5619206968586225e-16, 4.7794773323873853e-14, -1.1470745597729725e-11,
    2.0876756987868100e-09, -2.7557319223985888e-07, 2.4801587301587302e-05,
    -1.3888888888888889e-03, 4.1666666666666664e-02,
    3.14159265358979311600e+00, 1.22464679914735317723e-16)

// exp(x) = p 2^k without forming 2^k: k = round(x / ln 2) (returned as an integer-valued double),
// p = P(x - k ln 2) in [0.7071, 1.4143].  The two-level CDF path carries weights as such pairs:
// a tile's (or the island's) reference is then a power of two, and rescaling a sum of weights to
// another reference is an exact ldexp instead of another exp (oracle.c orc_expk).
// x = -inf gives k = -inf and p = NaN: callers select (p, k) = (0, -inf) for it.
__host__ __device__ __forceinline__ double smc_expk(double x, double& k)
{
    const double* K = SMC_K(smc_k_exp);
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
    const double* K = SMC_K(smc_k_exp);
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
    const double* K = SMC_K(smc_k_log);
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

// ---------------------------------------------------------------------------
// Box-Muller on two 52-bit uniforms, table-driven (round 3).  The two transcendental pieces of a
// normal pair -- log of the radius uniform, sin/cos of the angle -- were 127 of the 185 VALU
// instructions of a pair (polynomials of degree 14 and 2 x 8, a true division, range selects).
// Both arguments are known bit patterns, so the top bits index a small table and the remaining
// bits feed a short polynomial:
//   log u,  u = (k + 1/2) 2^-52:  u = m 2^e, node j = round(128 (m - 1)), r = m RN(1/c_j) - 1 (one
//       fma, |r| <= 2^-8), log u = e' ln2 + logc_j + log1p(r) with logc_j = -log RN(1/c_j) (an
//       identity for ANY invc), nodes above sqrt 2 filed under the next exponent so that nothing
//       cancels as u -> 1 (last node: invc = 1/2, logc = 0 exactly).  <= 3 ulp.
//   sin, cos of 2 pi u, u = (k + 1/2) 2^-52: top 8 bits = one of 256 centre angles (table), the
//       44 bits below = the offset x in (-pi/256, pi/256): sin x, cos x - 1 by 3 terms each, one
//       angle addition.  Absolute error < 2e-16; no quadrant logic.
//   sqrt(s), s = -2 log u > 0: v_rsq_f64 + one Goldschmidt step + one correction.  <= 1 ulp.
// tools/gen_normal_tables.py makes the tables (smc_normal_tab.h, 6 KB); kernels stage them in LDS
// (smc_ntab_stage) -- the two lookups of a pair are lane-random 16-byte reads.
// ---------------------------------------------------------------------------
#define SMC_NTAB_DECL SMC_CONST __attribute__((aligned(16)))
#include "smc_normal_tab.h"

// one table entry: 16 bytes, read with one ds_read_b128
struct __attribute__((aligned(16))) SmcD2 {
    double x, y;
};
// stage the tables in LDS (SMC_NTAB_N entries): every thread of the workgroup calls, a barrier
// must follow before the first smc_bm_pair.  Two phases so that a kernel can have its own first
// loads in flight between them: smc_ntab_fetch issues the (L2-resident) table loads into
// registers, smc_ntab_store writes them to LDS.  NTH threads; 385 entries: at most NE per thread.
template <int NTH>
struct SmcNtabRegs {
    static constexpr int NE = (SMC_NTAB_N + NTH - 1) / NTH;
    SmcD2 e[NE];
};
template <int NTH>
__device__ __forceinline__ void smc_ntab_fetch(SmcNtabRegs<NTH>& r, const int tid)
{
    const SmcD2* src = reinterpret_cast<const SmcD2*>(smc_ntab);
#pragma unroll
    for (int k = 0; k < SmcNtabRegs<NTH>::NE; ++k) {
        const int i = tid + k * NTH;
        r.e[k] = src[i < SMC_NTAB_N ? i : SMC_NTAB_N - 1];       // (unconditional: clamped address)
    }
}
// (the LDS array has SMC_NTAB_LDS_N >= NE * NTH entries: loads and stores are unconditional -- behind
//  a condition the compiler sinks each load next to its store and waits for everything in between)
#define SMC_NTAB_LDS_N 512
template <int NTH>
__device__ __forceinline__ void smc_ntab_store(const SmcNtabRegs<NTH>& r, SmcD2* lds, const int tid)
{
    static_assert(SmcNtabRegs<NTH>::NE * NTH <= SMC_NTAB_LDS_N, "LDS table too small for this workgroup size");
#pragma unroll
    for (int k = 0; k < SmcNtabRegs<NTH>::NE; ++k) lds[tid + k * NTH] = r.e[k];
}
template <int NTH>
__device__ __forceinline__ void smc_ntab_stage(SmcD2* lds, const int tid)
{
    SmcNtabRegs<NTH> r;
    smc_ntab_fetch<NTH>(r, tid);
    smc_ntab_store<NTH>(r, lds, tid);
}

SMC_K_TABLE(smc_k_bm, 16,
    1.0 / 7.0, -1.0 / 6.0, 0.2, -0.25, 1.0 / 3.0, -0.5,                  // log1p
    6.93147180369123816490e-01, 1.90821492927058770002e-10,              // ln2 hi, lo
    0.024543692606170259675,                                             // 2 pi / 256
    1.0 / 120.0, -1.0 / 6.0,                                             // sin x - x
    -1.0 / 720.0, 1.0 / 24.0, -0.5,                                      // cos x - 1
    0x1.fffffffffffffp-1, 0x1.7ffffffffffcp+0)                          // 1 - 2^-53, 1.5 - 2^-45

#ifdef SMC_EMULATE
__host__ __device__ __forceinline__ double smc_frexp_m(double x, int& e) { return frexp(x, &e); }
__host__ __device__ __forceinline__ double smc_rsq(double x) { return 1.0 / sqrt(x); }
#else
__device__ __forceinline__ double smc_frexp_m(double x, int& e)
{
    e = __builtin_amdgcn_frexp_exp(x);
    return __builtin_amdgcn_frexp_mant(x);
}
__device__ __forceinline__ double smc_rsq(double x) { return __builtin_amdgcn_rsq(x); }
#endif

// log u for u = ((a >> 12) + 1/2) 2^-52 in (0, 1): < 0, <= 3 ulp; tab: the staged tables (LDS)
__host__ __device__ __forceinline__ double smc_log_u52(const SmcD2* tab, const u64 a)
{
    const double* K = SMC_K(smc_k_bm);
    const double D = __longlong_as_double((long long)((a >> 12) | 0x3FF0000000000000ull));   // 1 + k 2^-52
    const double u = D - K[14];                                 // (k + 1/2) 2^-52, exact
    int e;
    const double mm = smc_frexp_m(u, e);                        // u = mm 2^e, mm in [1/2, 1)
    const u32 hi = (u32)((u64)__double_as_longlong(mm) >> 32);
    const u32 j = (((hi >> 12) & 0xFFu) + 1u) >> 1;             // nearest of 129 nodes of 2 mm in [1, 2]
    const SmcD2 lt = tab[j];
    const double invc2 = lt.x, logc = lt.y;
    const double r = fma(mm, invc2, -1.0);
    const double ed = (double)(e - 1 + (j >= (u32)SMC_NTAB_JUP ? 1 : 0));
    double p = SMC_FMA_K(r, K[0], K[1]);
    p = SMC_FMA_K(p, r, K[2]);
    p = SMC_FMA_K(p, r, K[3]);
    p = SMC_FMA_K(p, r, K[4]);
    p = SMC_FMA_K(p, r, K[5]);
    const double lp = fma(p, r * r, r);                         // log1p(r)
    return fma(ed, K[6], logc) + fma(ed, K[7], lp);
}

// (z0, z1) = sqrt(-2 log u1) (cos, sin)(2 pi u2), u1 = ((a >> 12) + 1/2) 2^-52, u2 likewise from b;
// tab: the staged tables (LDS)
__host__ __device__ __forceinline__ void smc_bm_pair(const SmcD2* tab, const u64 a, const u64 b,
                                                     double& z0, double& z1)
{
    const double* K = SMC_K(smc_k_bm);
    const double L = smc_log_u52(tab, a);                       // log u1 < 0
    // ---- radius sqrt(-2 L)
    const double s = -2.0 * L;
    const double y = smc_rsq(s);
    double g = s * y, h = 0.5 * y;
    const double e1 = fma(-h, g, 0.5);
    g = fma(g, e1, g);
    h = fma(h, e1, h);
    g = fma(fma(-g, g, s), h, g);
    // ---- angle 2 pi u2 = 2 pi (i + 1/2 + rho) / 256
    const u64 kb = b >> 12;
    const u32 i = (u32)(kb >> 44);
    const double D2 = __longlong_as_double((long long)(((kb << 8) & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
    const double x = (D2 - K[15]) * K[8];                       // offset from the centre angle, |x| < pi/256
    const double z = x * x;
    const double sx = fma(x * z, SMC_FMA_K(z, K[9], K[10]), x);                 // sin x
    const double dc = z * SMC_FMA_K(SMC_FMA_K(z, K[11], K[12]), z, K[13]);      // cos x - 1
    const SmcD2 st = tab[SMC_NTAB_SC + i];
    const double si = st.x, ci = st.y;
    const double sn = fma(ci, sx, fma(si, dc, si));
    const double cs = fma(-si, sx, fma(ci, dc, ci));
    z0 = g * cs;
    z1 = g * sn;
}

__host__ __device__ __forceinline__ void smc_sincospi_02(double a, double* sn, double* cs)
{
    const double* K = SMC_K(smc_k_sc);
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
