// smc_dpp.h -- wave64 scans / reductions on the DPP data path (v_mov_b32_dpp
// row_shr / row_bcast), instead of ds_bpermute round trips through the LDS
// hardware: a 6-step inclusive scan costs ~6 x (2 DPP moves + 1 op) VALU
// instructions and no LDS latency.
//
// Sequence (Kogge-Stone inside each row of 16 lanes, then row broadcasts):
//   row_shr:1, row_shr:2, row_shr:4, row_shr:8, row_bcast:15, row_bcast:31
// after which lane l holds op(x_0..x_l) and lane 63 the wave total.
#pragma once
#include "smc_platform.h"

#define SMC_DPP_ROW_SHR(n) (0x110 + (n))
#define SMC_DPP_ROW_BCAST15 0x142
#define SMC_DPP_ROW_BCAST31 0x143
#define SMC_DPP_WAVE_SHR1 0x138     /* lane l <- lane l-1 across the whole wave */

#ifdef SMC_EMULATE
// emulator: source lane of each control code (guards in the callers make the
// value delivered to lanes without a valid source irrelevant)
template <int CTRL>
inline unsigned smc_mov_dpp(unsigned v)
{
    const int l = emu_lane();
    int src = l;
    if (CTRL >= 0x111 && CTRL <= 0x11F) {
        const int n = CTRL - 0x110;
        src = ((l & 15) >= n) ? l - n : l;
    } else if (CTRL == SMC_DPP_ROW_BCAST15) {
        src = (l >= 16) ? (l & ~15) - 1 : l;
    } else if (CTRL == SMC_DPP_ROW_BCAST31) {
        src = (l >= 32) ? 31 : l;
    } else if (CTRL == SMC_DPP_WAVE_SHR1) {
        src = (l >= 1) ? l - 1 : l;
    }
    return hipemu::exchange(v, emu_wbase() + src);
}
inline unsigned smc_readlane(unsigned v, int lane) { return hipemu::exchange(v, emu_wbase() + lane); }
#else
template <int CTRL>
__device__ __forceinline__ unsigned smc_mov_dpp(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned smc_readlane(unsigned v, int lane)
{
    return (unsigned)__builtin_amdgcn_readlane((int)v, lane);
}
#endif

template <int CTRL>
__device__ __forceinline__ u64 smc_mov_dpp64(u64 v)
{
    const unsigned lo = smc_mov_dpp<CTRL>((unsigned)v), hi = smc_mov_dpp<CTRL>((unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}
template <int CTRL>
__device__ __forceinline__ double smc_mov_dpp_f64(double v)
{
    return __longlong_as_double((long long)smc_mov_dpp64<CTRL>((u64)__double_as_longlong(v)));
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

struct SmcOpAddU64 { __device__ __forceinline__ u64 operator()(u64 a, u64 b) const { return a + b; } };
struct SmcOpMaxU32 { __device__ __forceinline__ unsigned operator()(unsigned a, unsigned b) const { return a > b ? a : b; } };
struct SmcOpAddF64 { __device__ __forceinline__ double operator()(double a, double b) const { return a + b; } };
struct SmcOpMaxF64 { __device__ __forceinline__ double operator()(double a, double b) const { return fmax(a, b); } };

#define SMC_DPP_SCAN_BODY(MOV)                                                        \
    const int l = (int)(threadIdx.x & 63u);                                            \
    { auto t = op(MOV<SMC_DPP_ROW_SHR(1)>(v), v); if ((l & 15) >= 1) v = t; }          \
    { auto t = op(MOV<SMC_DPP_ROW_SHR(2)>(v), v); if ((l & 15) >= 2) v = t; }          \
    { auto t = op(MOV<SMC_DPP_ROW_SHR(4)>(v), v); if ((l & 15) >= 4) v = t; }          \
    { auto t = op(MOV<SMC_DPP_ROW_SHR(8)>(v), v); if ((l & 15) >= 8) v = t; }          \
    { auto t = op(MOV<SMC_DPP_ROW_BCAST15>(v), v); if ((l & 31) >= 16) v = t; }        \
    { auto t = op(MOV<SMC_DPP_ROW_BCAST31>(v), v); if (l >= 32) v = t; }               \
    return v;

// inclusive scans over the 64 lanes of a wave (all lanes must be active)
template <class Op>
__device__ __forceinline__ u64 smc_wave_scan_u64(u64 v, Op op) { SMC_DPP_SCAN_BODY(smc_mov_dpp64) }
template <class Op>
__device__ __forceinline__ unsigned smc_wave_scan_u32(unsigned v, Op op) { SMC_DPP_SCAN_BODY(smc_mov_dpp) }
template <class Op>
__device__ __forceinline__ double smc_wave_scan_f64(double v, Op op) { SMC_DPP_SCAN_BODY(smc_mov_dpp_f64) }

// reductions: every lane receives the wave's result (order of operations fixed)
__device__ __forceinline__ u64 smc_wave_sum_u64(u64 v)
{
    return smc_readlane64(smc_wave_scan_u64(v, SmcOpAddU64()), 63);
}
__device__ __forceinline__ double smc_wave_sum(double v)
{
    return smc_readlane_f64(smc_wave_scan_f64(v, SmcOpAddF64()), 63);
}
__device__ __forceinline__ double smc_wave_max(double v)
{
    return smc_readlane_f64(smc_wave_scan_f64(v, SmcOpMaxF64()), 63);
}
