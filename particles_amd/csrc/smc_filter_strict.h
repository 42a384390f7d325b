// smc_filter_strict.h -- SMC_FLAG_STRICT_ANCESTORS on the two-level step: the reference's own inverse_cdf
// (resampling.py:484-509: the CDF accumulated sequentially in fp64, the strict `>` advance) on the filter's own
// normalised weights, as TWO launches between two k_propagate launches (smc_seqx.h):
//
//   k_strict_classify<MID>(t)  one workgroup per tile of 1024 parents.  Unless k_reduce2 ran (MID: more than 1024
//       tiles, multinomial), every workgroup reduces the island's log-sum-exp partials itself -- K, s, ESS, the
//       decision, exactly k_ancestors2's operations -- and workgroup 0 writes the step record and the summary row.
//       W_j = p_j 2^(k_j - K) / s (the values smc_filter_get(SMC_FIELD_W) returns) is formed in registers and never
//       stored; the estimate of the running sum in front of the tile comes from the same partials (S_b 2^(K_b - K) / s:
//       within a few ulps of the tile's true sum).  Classification, the exception lists, and -- in the last workgroup
//       of the island to finish -- the walk, the verification and the tiles' headers: sqx_classify_tile / sqx_chain.
//   k_strict_search(t)         one workgroup per tile of parents: the tile's sums staged in LDS, its range of
//       offspring, a bisection per offspring: A_t (sqx_search_tile).
//
// The step is then k_propagate -> k_strict_classify -> k_strict_search -> k_propagate: three dependent launches
// (round 4: eight).  Filters of fewer than two tiles, and the flat-CDF test paths, keep the materialised form
// (k_strict_W -> smc_seqx.h on the array -> k_sqx_fill -> k_strict_search_S).
#pragma once
#include "smc_filter_kernels.h"
#include "smc_seqx.h"

// the filter's normalised weights of step t - 1, formed on the fly
struct SqxSrcFilter {
    const double* lw;
    double K, rs;
    int kform;
    i64 n;
    __device__ __forceinline__ double weight(const double l) const
    {
        if (kform) {
            double kk;
            double p = smc_expk(l, kk);
            const bool ok = l > -INFINITY;
            p = ok ? p : 0.0;
            kk = ok ? kk : -INFINITY;
            return smc_scale_pk(p, kk, K) * rs;
        }
        return f_weight(l, K, rs);
    }
    __device__ __forceinline__ void load4(const i64 i0, double (&w)[4]) const
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (i0 + k < n) ? weight(lw[i0 + k]) : 0.0;
    }
};

// Returns whether step t resamples (the same answer in every workgroup of the island); t_out = t.
template <bool MID>
__device__ __forceinline__ bool strict_classify_part(const FArgs& a, const SqxArgs& q, int& b_out, i64& t_out, u64* lds8k)
{
    __shared__ double s_max[SMC_NWAVE];
    __shared__ double s_esc[SMC_NWAVE];
    __shared__ double s_sum[3 * SMC_NWAVE];
    // (workgroups are dispatched in index order and the island's chain starts when the LAST of them has taken its ticket:
    //  the two tiles that take longest -- the head of the array, a dozen exceptions, and its end, inside the margin of the
    //  binade edge at 1.0 -- go first)
    const int bx = (int)blockIdx.x;
    const int b = bx == 0 ? 0 : (bx == 1 ? q.ntiles - 1 : bx - 1);
    const int isl = (int)blockIdx.y, tid = (int)threadIdx.x;
    const int lane = smc_lane(), wave = smc_wave();
    SQX_STAMP(q, b, 0);
    double* info = a.info + (i64)isl * INFO_STRIDE;
    const double r0 = smc_ldg(MID ? info : a.info2 + (i64)isl * INFO_STRIDE);
    const double r1 = MID ? smc_ldg(info + 1) : 0.0, r3 = MID ? smc_ldg(info + 3) : 0.0, r4 = MID ? smc_ldg(info + 4) : 0.0;
    const i64 o = (i64)isl * a.nparts;
    double pm4[4], ps4[4], pss4[4];
    if (!MID) {
        const bool pvec = (a.nparts & 3) == 0;
        f_load4<double>(a.pm + o, (i64)tid * 4, a.nparts, pvec, -INFINITY, pm4);
        f_load4<double>(a.ps + o, (i64)tid * 4, a.nparts, pvec, 0.0, ps4);
        f_load4<double>(a.pss + o, (i64)tid * 4, a.nparts, pvec, 0.0, pss4);
    }
    const double Kb_v = smc_ldg(a.pm + o + b);                 // this tile's own exponent: the scale of the in-tile estimate
    const double Gb_v = MID ? smc_ldg(reinterpret_cast<const double*>(a.Qpre) + (i64)isl * a.ntiles + b) : 0.0;
    // the tile's weights on the tile's own scale, e_j = p_j 2^(k_j - K_b): k_propagate left them where the default step keeps
    // the integer CDF (FArgs::strict_e) -- no slot arithmetic (one array, rewritten every step), no exponential here
    const i64 i0 = (i64)b * SEQ_TILE + (i64)tid * 4;
    u64 eb[4];
    {
        const u64* ce = a.cq + (i64)isl * a.ncq + i0;          // (padded to whole tiles: in bounds; 0 beyond N)
        smc_ld2g(ce, eb[0], eb[1]);
        smc_ld2g(ce + 2, eb[2], eb[3]);
    }
    const i64 t = (i64)smc_uniform(r0);
    b_out = b;
    t_out = t;
    if (t >= a.T) {
        if (!MID && b == 0 && tid == 0) info[0] = (double)t;   // k_propagate returns on it
        return false;
    }
    if (t == 0) return false;                                  // the host wrote the record of step 0
    if (MID && smc_uniform(r1) == 0.0) return false;           // k_reduce2: step t does not resample
    F2RecIn rin = {0.0, 0.0, 0.0, 0.0, 0.0};                   // (what the record's writer reads: on its way during the reduction)
    if (!MID && b == 0 && tid == 0) rin = f2_record_loads(a, isl, t);
    const double Kb = smc_uniform(Kb_v);
    double e4[4], esum = 0.0;
    bool deep = false;                                         // a subnormal e: 2^-1022 and more below the tile's maximum
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        e4[k] = (i0 + k < a.N) ? __longlong_as_double((long long)eb[k]) : 0.0;
        deep = deep || (e4[k] != 0.0 && e4[k] < 0x1.0p-1022);
        esum += e4[k];
    }
    const double einc = smc_wave_scan_add_f64(esum);
    // (the sum in front of this thread: the left neighbour's inclusive one -- NOT einc - esum, which cancels wherever the
    //  weights rise steeply, the head of every array, and then sends the step to the exact path)
    const double eexc = smc_dpp_f64<SMC_DPP_WAVE_SHR1, 0xf, false>(einc);
    double K, rs, before = 0.0;
    F2Red rec;                                                 // (!MID: the step's record, written by tile 0's workgroup at its end)
    rec.K = rec.s = rec.ss = rec.ess = rec.rs = rec.log_mean = 0.0;
    rec.bad = false;
    if (MID) {
        K = smc_uniform(r3);
        rs = smc_uniform(r4);
        if (lane == 63) s_esc[wave] = einc;
        // the estimate in front of this tile: k_reduce2 left the tiles' fractions of the normalised sum and their prefixes
        // (FArgs::strict_e: plain doubles, positive terms added forwards -- relative error below ntiles 2^-53 however
        //  little of the mass lies in front of the tile; the default step's integer shares of 2^52 are accurate to
        //  2^-53 of the TOTAL, which is nothing in front of the first heavy particle of a collapsed population)
        before = smc_uniform(Gb_v);
        __syncthreads();                                       // (s_esc is complete)
    } else {
        // ---- all partials -> K, (s, ss), ESS, the decision: k_ancestors2's operations, hence its bits; the estimate of
        // the sum in front of this tile (the shares of the tiles before it) rides in the second exchange
        double tm = smc_max2(smc_max2(pm4[0], pm4[1]), smc_max2(pm4[2], pm4[3]));
        tm = smc_wave_max(tm);
        if (lane == 0) s_max[wave] = tm;
        if (lane == 63) s_esc[wave] = einc;
        __syncthreads();
        F2Red r;
        r.K = s_max[0];
#pragma unroll
        for (int w = 1; w < SMC_NWAVE; ++w) r.K = smc_max2(r.K, s_max[w]);
        double v4[4], s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double w;
            f2_rescale(pm4[k], r.K, ps4[k], pss4[k], v4[k], w);
            s1 = s1 + v4[k];
            s2 = s2 + w;
            before += (tid * 4 + k < b) ? v4[k] : 0.0;
        }
        s1 = smc_wave_sum(s1);
        s2 = smc_wave_sum(s2);
        before = smc_wave_sum(before);
        if (lane == 0) { s_sum[wave] = s1; s_sum[SMC_NWAVE + wave] = s2; s_sum[2 * SMC_NWAVE + wave] = before; }
        __syncthreads();
        s1 = s_sum[0];
        s2 = s_sum[SMC_NWAVE];
        before = s_sum[2 * SMC_NWAVE];
#pragma unroll
        for (int w = 1; w < SMC_NWAVE; ++w) {
            s1 = s1 + s_sum[w];
            s2 = s2 + s_sum[SMC_NWAVE + w];
            before = before + s_sum[2 * SMC_NWAVE + w];
        }
        r.s = s1;
        r.ss = s2;
        f2_finish(a, r);
        const bool resample = r.ess < a.ess_thresh;            // core.py:181-183 (t < T here)
        // (the record is read by the NEXT launch: its stores go out behind the tile's ticket, not in front of the barriers
        //  on the way to it -- tile 0 carries the head of the array, a dozen exceptions, and is among the last to arrive)
        if (!resample) {
            if (b == 0 && tid == 0) f2_write_record(a, isl, t, r, false, rin);
            return false;
        }
        rec = r;
        K = r.K;
        rs = r.rs;
        before = before * rs;
    }
    // the weights themselves (the values smc_filter_get(SMC_FIELD_W) returns): p 2^(k - K) / s = e 2^(K_b - K) / s -- the
    // second scaling is exact (or rounds the same exact value once) as long as e is a normal number; a subnormal e has
    // already lost bits, and such a particle goes back to its log-weight (never in practice: 700 nats below its tile's best)
    const SqxSrcFilter src{f_lw(a, t - 1) + (i64)isl * a.N, K, rs, a.kform, a.N};
    double dsc = Kb - K;                                       // (<= 0: K is the maximum of the tiles' exponents)
    dsc = (dsc > -2000.0) ? dsc : -2000.0;
    double w4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w4[k] = ldexp(e4[k], (int)dsc) * rs;
    if (deep) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (e4[k] != 0.0 && e4[k] < 0x1.0p-1022) w4[k] = src.weight(src.lw[i0 + k]);
    }
    double ebase = 0.0;
#pragma unroll
    for (int w = 0; w < SMC_NWAVE; ++w) ebase += (w < wave) ? s_esc[w] : 0.0;
    const double run0 = before + ldexp(ebase + eexc, (int)dsc) * rs;
    SQX_STAMP(q, b, 1);
    const bool last = sqx_classify_tile(w4, run0, isl, b, q);
    if (!MID && b == 0 && tid == 0) f2_write_record(a, isl, t, rec, true, rin);
    if (last) sqx_chain<SqxSrcFilter>(src, isl, q, lds8k);
    return true;
}
template <bool MID>
__global__ void __launch_bounds__(SMC_BLOCK)
k_strict_classify(const FArgs av, const SqxArgs q)
{
    __shared__ u64 c_Pt[SEQ_TILE];
    int b;
    i64 t;
    (void)strict_classify_part<MID>(av, q, b, t, c_Pt);
}

__device__ __forceinline__ void strict_search_part(const FArgs& a, const SqxArgs& q, const int b, double* sS)
{
    const int isl = (int)blockIdx.y;
    SQX_STAMP(q, q.ntiles + 8 + b, 0);
    const double* info = a.info + (i64)isl * INFO_STRIDE;
    // every load of the common path in one go: the record, what the tile's sums are staged from, the spacings' total
    const double r0 = smc_ldg(info), r1 = smc_ldg(info + 1);
    const SqxStage ld = sqx_stage_load(q, isl, b);
    const bool zform = a.scheme == SMC_MULTINOMIAL_ && !a.ut && a.sp_tpw;
    const u64 zall = zform ? smc_ldg(a.E + (i64)isl * (a.ntiles1 + 1) + a.ntiles1) : 0ull;
    const i64 t = (i64)smc_uniform(r0);
    if (t >= a.T || t == 0 || smc_uniform(r1) == 0.0) return;
    SmcSu su;
    u64 Us;
    f2_su(a, isl, t, su, Us);
    if (a.log2N >= 0) su.rM = 1.0 / su.dM;                     // (N = 2^k: the division by N is an exact scaling)
    if (zform) {
        // one-pass uniform_spacings (k_f_spacing_onepass): the integer prefix sums; the look-back words re-armed
        su.zo = reinterpret_cast<const u32*>(a.su) + (i64)isl * a.N * 2;
        su.zE = a.E + (i64)isl * (a.ntiles1 + 1);
        su.dall = (double)smc_uniform_u64(zall);
        if (b < a.sp_nwg && threadIdx.x == 0) a.sst[(i64)isl * a.sp_nwg + b] = 0ull;
    }
    SQX_STAMP(q, q.ntiles + 8 + b, 1);
    const double S_start = sqx_stage_tile(q, isl, b, ld, sS);
    SQX_STAMP(q, q.ntiles + 8 + b, 2);
    sqx_search_tile<u32>(q, b, su, sS, S_start, f_A(a, t) + (i64)isl * a.N);
}
__global__ void __launch_bounds__(SMC_BLOCK)
k_strict_search(const FArgs av, const SqxArgs q)
{
    __shared__ double sS[SEQ_TILE];
    strict_search_part(av, q, (int)blockIdx.x, sS);
}
