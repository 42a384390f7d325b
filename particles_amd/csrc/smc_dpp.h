// smc_dpp.h -- wave64 scans / reductions on the DPP data path (v_mov_b32_dpp
// row_shr / row_bcast), instead of ds_bpermute round trips through the LDS
// hardware.
//
// Sequence (Kogge-Stone inside each row of 16 lanes, then row broadcasts):
//   row_shr:1, row_shr:2, row_shr:4, row_shr:8, row_bcast:15 (rows 1,3),
//   row_bcast:31 (rows 2,3)
// after which lane l holds op(x_0..x_l) and lane 63 the wave total.  Lanes
// without a source read the operation's identity straight from the DPP
// controls (bound_ctrl zero-fill for sums, their own value for max; row_mask
// for the broadcasts), so a step is two DPP moves and one op per 64-bit value
// -- no compare/select.
#pragma once
#include "smc_platform.h"

#define SMC_DPP_ROW_SHR(n) (0x110 + (n))
#define SMC_DPP_ROW_BCAST15 0x142
#define SMC_DPP_ROW_BCAST31 0x143
#define SMC_DPP_WAVE_SHR1 0x138     /* lane l <- lane l-1 across the whole wave */

// smc_dpp<CTRL, ROW_MASK, ZERO>(old, v): the value of v in the source lane that
// CTRL designates; lanes whose row is not in ROW_MASK keep `old`; lanes whose
// source does not exist get 0 if ZERO (bound_ctrl) else `old`.
#ifdef SMC_EMULATE
template <int CTRL, int ROW_MASK, bool ZERO>
inline unsigned smc_dpp(unsigned old, unsigned v)
{
    const int l = emu_lane();
    int src = -1;
    if (CTRL >= 0x111 && CTRL <= 0x11F) {
        const int n = CTRL - 0x110;
        src = ((l & 15) >= n) ? l - n : -1;
    } else if (CTRL == SMC_DPP_ROW_BCAST15) {
        src = (l >= 16) ? (l & ~15) - 1 : -1;
    } else if (CTRL == SMC_DPP_ROW_BCAST31) {
        src = (l >= 32) ? 31 : -1;
    } else if (CTRL == SMC_DPP_WAVE_SHR1) {
        src = (l >= 1) ? l - 1 : -1;
    }
    const unsigned got = hipemu::exchange(v, emu_wbase() + (src < 0 ? l : src));
    if (!((ROW_MASK >> (l >> 4)) & 1)) return old;
    if (src < 0) return ZERO ? 0u : old;
    return got;
}
inline unsigned smc_readlane(unsigned v, int lane) { return hipemu::exchange(v, emu_wbase() + lane); }
#else
template <int CTRL, int ROW_MASK, bool ZERO>
__device__ __forceinline__ unsigned smc_dpp(unsigned old, unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xf, ZERO);
}
__device__ __forceinline__ unsigned smc_readlane(unsigned v, int lane)
{
    return (unsigned)__builtin_amdgcn_readlane((int)v, lane);
}
#endif

template <int CTRL>
__device__ __forceinline__ unsigned smc_mov_dpp(unsigned v)      // lanes without source keep v
{
    return smc_dpp<CTRL, 0xf, false>(v, v);
}
__device__ __forceinline__ u64 smc_readlane64(u64 v, int lane)
{
    const unsigned lo = smc_readlane((unsigned)v, lane), hi = smc_readlane((unsigned)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ double smc_readlane_f64(double v, int lane)
{
    return __longlong_as_double((long long)smc_readlane64((u64)__double_as_longlong(v), lane));
}

// 64-bit DPP move; OWN: lanes without a source see their own value (identity
// of max), otherwise 0 (identity of +)
template <int CTRL, int ROW_MASK, bool OWN>
__device__ __forceinline__ u64 smc_dpp64(u64 v)
{
    const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    const unsigned l2 = OWN ? smc_dpp<CTRL, ROW_MASK, false>(lo, lo) : smc_dpp<CTRL, ROW_MASK, true>(0u, lo);
    const unsigned h2 = OWN ? smc_dpp<CTRL, ROW_MASK, false>(hi, hi) : smc_dpp<CTRL, ROW_MASK, true>(0u, hi);
    return ((u64)h2 << 32) | l2;
}
template <int CTRL, int ROW_MASK, bool OWN>
__device__ __forceinline__ double smc_dpp_f64(double v)
{
    return __longlong_as_double(
        (long long)smc_dpp64<CTRL, ROW_MASK, OWN>((u64)__double_as_longlong(v)));
}

// gfx950's lane swaps (v_permlane16_swap_b32 / v_permlane32_swap_b32): ONE VALU instruction trades rows (16 lanes)
// 1 and 3 of `a` with rows 0 and 2 of `b` (SWAP16), or the upper 32 lanes of `a` with the lower 32 of `b` (SWAP32) --
// both directions at once, no LDS round trip (ds_bpermute) and no select afterwards:
//   swap16:  a' = [a.r0, b.r0, a.r2, b.r2]   b' = [a.r1, b.r1, a.r3, b.r3]
//   swap32:  a' = [a.lo, b.lo]               b' = [a.hi, b.hi]
#ifdef SMC_EMULATE
inline void smc_swap16(unsigned& a, unsigned& b)
{
    const int l = emu_lane();
    const unsigned af = hipemu::exchange(a, emu_wbase() + (l ^ 16)), bf = hipemu::exchange(b, emu_wbase() + (l ^ 16));
    if ((l >> 4) & 1) a = bf; else b = af;
}
inline void smc_swap32(unsigned& a, unsigned& b)
{
    const int l = emu_lane();
    const unsigned af = hipemu::exchange(a, emu_wbase() + (l ^ 32)), bf = hipemu::exchange(b, emu_wbase() + (l ^ 32));
    if (l & 32) a = bf; else b = af;
}
#else
__device__ __forceinline__ void smc_swap16(unsigned& a, unsigned& b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
__device__ __forceinline__ void smc_swap32(unsigned& a, unsigned& b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
#endif
__device__ __forceinline__ void smc_swap16_f64(double& a, double& b)
{
    const u64 ua = (u64)__double_as_longlong(a), ub = (u64)__double_as_longlong(b);
    unsigned al = (unsigned)ua, ah = (unsigned)(ua >> 32), bl = (unsigned)ub, bh = (unsigned)(ub >> 32);
    smc_swap16(al, bl);
    smc_swap16(ah, bh);
    a = __longlong_as_double((long long)(((u64)ah << 32) | al));
    b = __longlong_as_double((long long)(((u64)bh << 32) | bl));
}
__device__ __forceinline__ void smc_swap32_f64(double& a, double& b)
{
    const u64 ua = (u64)__double_as_longlong(a), ub = (u64)__double_as_longlong(b);
    unsigned al = (unsigned)ua, ah = (unsigned)(ua >> 32), bl = (unsigned)ub, bh = (unsigned)(ub >> 32);
    smc_swap32(al, bl);
    smc_swap32(ah, bh);
    a = __longlong_as_double((long long)(((u64)ah << 32) | al));
    b = __longlong_as_double((long long)(((u64)bh << 32) | bl));
}
// sum over the 4 lanes l, l ^ 16, l ^ 32, l ^ 48 (one lane of every row), every one of them gets it:
// (v + v[l ^ 16]) + the same of l ^ 32 -- the tree of two __shfl_xor steps (additions commute: same bits)
__device__ __forceinline__ double smc_sum_rows(double v)
{
    double a = v, b = v;
    smc_swap16_f64(a, b);        // a = the even row's value of the pair, b = the odd row's
    a = a + b;
    b = a;
    smc_swap32_f64(a, b);        // a = the lower half's pair sum, b = the upper half's
    return a + b;
}

// inclusive scans over the 64 lanes of a wave (all lanes must be active)
__device__ __forceinline__ u64 smc_wave_scan_add_u64(u64 v)
{
    v += smc_dpp64<SMC_DPP_ROW_SHR(1), 0xf, false>(v);
    v += smc_dpp64<SMC_DPP_ROW_SHR(2), 0xf, false>(v);
    v += smc_dpp64<SMC_DPP_ROW_SHR(4), 0xf, false>(v);
    v += smc_dpp64<SMC_DPP_ROW_SHR(8), 0xf, false>(v);
    v += smc_dpp64<SMC_DPP_ROW_BCAST15, 0xa, false>(v);
    v += smc_dpp64<SMC_DPP_ROW_BCAST31, 0xc, false>(v);
    return v;
}
__device__ __forceinline__ double smc_wave_scan_add_f64(double v)
{
    v = v + smc_dpp_f64<SMC_DPP_ROW_SHR(1), 0xf, false>(v);
    v = v + smc_dpp_f64<SMC_DPP_ROW_SHR(2), 0xf, false>(v);
    v = v + smc_dpp_f64<SMC_DPP_ROW_SHR(4), 0xf, false>(v);
    v = v + smc_dpp_f64<SMC_DPP_ROW_SHR(8), 0xf, false>(v);
    v = v + smc_dpp_f64<SMC_DPP_ROW_BCAST15, 0xa, false>(v);
    v = v + smc_dpp_f64<SMC_DPP_ROW_BCAST31, 0xc, false>(v);
    return v;
}
// two independent scans, step by step side by side: a DPP move must wait two cycles for the VALU result it reads,
// and each chain alone is six such dependent steps -- written (and, with the scheduler kept out, issued) alternately
// each chain's add covers the other's wait.  Same operations per chain, hence the same bits as two calls above.
#define SMC_SCAN2_STEP(OP, CTRL, MASK)                     \
    {                                                      \
        const auto ta = OP<CTRL, MASK, false>(a);          \
        const auto tb = OP<CTRL, MASK, false>(b);          \
        a = a + ta;                                        \
        b = b + tb;                                        \
    }
__device__ __forceinline__ void smc_wave_scan_add_f64x2(double& a, double& b)
{
    SMC_SCAN2_STEP(smc_dpp_f64, SMC_DPP_ROW_SHR(1), 0xf)
    SMC_SCAN2_STEP(smc_dpp_f64, SMC_DPP_ROW_SHR(2), 0xf)
    SMC_SCAN2_STEP(smc_dpp_f64, SMC_DPP_ROW_SHR(4), 0xf)
    SMC_SCAN2_STEP(smc_dpp_f64, SMC_DPP_ROW_SHR(8), 0xf)
    SMC_SCAN2_STEP(smc_dpp_f64, SMC_DPP_ROW_BCAST15, 0xa)
    SMC_SCAN2_STEP(smc_dpp_f64, SMC_DPP_ROW_BCAST31, 0xc)
}
#undef SMC_SCAN2_STEP
// two inclusive scans of values below 2^51 as four scans of limbs (26 low bits, 25 high): 64 limbs sum to less than
// 2^32, so no carry ever leaves a limb on the way, each step of each limb is ONE v_add_u32 with a DPP operand, and
// (high << 26) + low puts the exact sums back together -- 12 instructions per scan instead of 24 and none of the
// waits between a 64-bit add and the move that reads it.  Integer arithmetic: the same values as the 64-bit scans.
__device__ __forceinline__ unsigned smc_wave_scan_add_u32(unsigned v)
{
    v += smc_dpp<SMC_DPP_ROW_SHR(1), 0xf, true>(0u, v);
    v += smc_dpp<SMC_DPP_ROW_SHR(2), 0xf, true>(0u, v);
    v += smc_dpp<SMC_DPP_ROW_SHR(4), 0xf, true>(0u, v);
    v += smc_dpp<SMC_DPP_ROW_SHR(8), 0xf, true>(0u, v);
    v += smc_dpp<SMC_DPP_ROW_BCAST15, 0xa, true>(0u, v);
    v += smc_dpp<SMC_DPP_ROW_BCAST31, 0xc, true>(0u, v);
    return v;
}
__device__ __forceinline__ void smc_wave_scan_add_u51x2(u64& a, u64& b)
{
    unsigned al = (unsigned)a & 0x3FFFFFFu, ah = (unsigned)(a >> 26);
    unsigned bl = (unsigned)b & 0x3FFFFFFu, bh = (unsigned)(b >> 26);
    al = smc_wave_scan_add_u32(al);
    ah = smc_wave_scan_add_u32(ah);
    bl = smc_wave_scan_add_u32(bl);
    bh = smc_wave_scan_add_u32(bh);
    a = ((u64)ah << 26) + al;
    b = ((u64)bh << 26) + bl;
}
// the two sums of a wave (every lane receives them)
__device__ __forceinline__ void smc_wave_sum2(double& a, double& b)
{
    smc_wave_scan_add_f64x2(a, b);
    a = smc_readlane_f64(a, 63);
    b = smc_readlane_f64(b, 63);
}
// max of two doubles in one instruction (fmax() canonicalises both operands
// first: three v_max_f64); NaNs never reach the reductions (sanitised to -inf)
__device__ __forceinline__ double smc_max2(double a, double b)
{
#ifdef SMC_EMULATE
    return a > b ? a : b;
#else
    double o;
    asm("v_max_f64 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
    return o;
#endif
}
__device__ __forceinline__ double smc_wave_scan_max_f64(double v)
{
    v = smc_max2(v, smc_dpp_f64<SMC_DPP_ROW_SHR(1), 0xf, true>(v));
    v = smc_max2(v, smc_dpp_f64<SMC_DPP_ROW_SHR(2), 0xf, true>(v));
    v = smc_max2(v, smc_dpp_f64<SMC_DPP_ROW_SHR(4), 0xf, true>(v));
    v = smc_max2(v, smc_dpp_f64<SMC_DPP_ROW_SHR(8), 0xf, true>(v));
    v = smc_max2(v, smc_dpp_f64<SMC_DPP_ROW_BCAST15, 0xa, true>(v));
    v = smc_max2(v, smc_dpp_f64<SMC_DPP_ROW_BCAST31, 0xc, true>(v));
    return v;
}
__device__ __forceinline__ unsigned smc_wave_scan_max_u32(unsigned v)
{
    unsigned t;
    t = smc_dpp<SMC_DPP_ROW_SHR(1), 0xf, true>(0u, v); v = v > t ? v : t;
    t = smc_dpp<SMC_DPP_ROW_SHR(2), 0xf, true>(0u, v); v = v > t ? v : t;
    t = smc_dpp<SMC_DPP_ROW_SHR(4), 0xf, true>(0u, v); v = v > t ? v : t;
    t = smc_dpp<SMC_DPP_ROW_SHR(8), 0xf, true>(0u, v); v = v > t ? v : t;
    t = smc_dpp<SMC_DPP_ROW_BCAST15, 0xa, true>(0u, v); v = v > t ? v : t;
    t = smc_dpp<SMC_DPP_ROW_BCAST31, 0xc, true>(0u, v); v = v > t ? v : t;
    return v;
}

// reductions: every lane receives the wave's result (order of operations fixed)
__device__ __forceinline__ u64 smc_wave_sum_u64(u64 v)
{
    return smc_readlane64(smc_wave_scan_add_u64(v), 63);
}
__device__ __forceinline__ double smc_wave_sum(double v)
{
    return smc_readlane_f64(smc_wave_scan_add_f64(v), 63);
}
__device__ __forceinline__ double smc_wave_max(double v)
{
    return smc_readlane_f64(smc_wave_scan_max_f64(v), 63);
}
