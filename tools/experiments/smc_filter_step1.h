// smc_filter_step1.h -- the step of the two-level path as ONE launch (k_step1): what k_ancestors2(t) and
// k_propagate(t) do in two, for the grids every workgroup of which reduces the partials itself (<= 1024 whole
// tiles per island), systematic / stratified, Bootstrap / Guided.
//
// k_ancestors2 is PARENT-major (workgroup b owns parent tile b and scatters its offspring, wherever they fall):
// its A has to travel through memory to the k_propagate workgroups that own the offspring.  Here workgroup b
// owns OFFSPRING tile b from the weights to the new particles -- the same contract read the other way round:
//
//   ns(j) = count(P_j),  P_j = G_b + floor(c_j Q_b / t_b),  count(C) = #{ n : T_n <= C },  T non-decreasing
//   A_n = max{ j : ns(j) <= n } = #{ j >= 1 : count(P_j) <= n } = #{ j >= 1 : P_j < T_n }
//       = the LARGEST (b, i) with  G_b < T_n  and  c_i Q_b < (T_n - G_b) t_b   (c_i = 0: always)
//
// (oracle.c orc_inverse_cdf_2level; count(P) > n <=> T_n <= P).  Per workgroup: all partials -> K, s, ss, the
// decision, the shares Q_b and their prefix G (LDS, exact integers); the thresholds of its 1024 offspring in
// closed form; the parent tile of each by bisection over G; the integer CDFs of the <= F1_STAGE parent tiles its
// offspring fall into staged in LDS as doubles; the parent inside the tile by bisection on c_i < x,
// x = (T_n - G_b) t_b / Q_b evaluated in fp64 with a band that covers its roundings (3 of x, 1 of each c_i, all
// below 2^-50 x + 2^7) -- a probe inside the band, or a parent tile outside the staged window, sends that offspring
// to the exact integer search on the stored CDF (smc_muldiv_floor: the oracle's own expression).  Then the move:
// gather, normals, m_step, the tile's partial and integer CDF for the next step.
//
// Nothing written by a launch is read by the same launch: the partials (K_b, S_b, SS_b, t_b) and the tiles' CDFs
// are double-buffered by the parity of t (f1_view; k_flush2 / k_f_partials pick the buffer the same way), X and
// lw by their slots.  The count of steps done (info2[0] = t + 1, written by workgroup 0 at its end) may be read
// by a late workgroup of the same launch as t + 1: the host passes the parity of t with every launch (FArgs::pp,
// static inside a captured graph), which settles it.
#pragma once
#include "smc_filter_kernels.h"

#define F1_STAGE 3                      /* parent tiles staged per workgroup (24 KB as doubles) */

__device__ __forceinline__ FArgs f1_view(const FArgs& a, const int q)
{
    FArgs v = a;
    if (q) { v.pm = a.pmB; v.ps = a.psB; v.pss = a.pssB; v.tq = a.tqB; v.cq = a.cqB; }
    return v;
}

// the exact route: largest i in [0, 1024) with pos(c_i) < D, pos as the contract states it
__device__ __attribute__((noinline)) int f1_exact_parent(const u64* cqt, const u64 tb, const u64 Qb, const u64 D)
{
    int pos = 0;
#pragma unroll 1
    for (int step = F_TILE / 2; step; step >>= 1) {
        const u64 c = smc_ldg(cqt + pos + step);
        if (smc_muldiv_floor(c, Qb, tb) < D) pos += step;
    }
    return pos;
}

template <int KIND, int FK>
__global__ void __launch_bounds__(SMC_BLOCK)
k_step1(const FArgs av)
{
    const FArgs& a = av;
    constexpr int OPT = 4;
    // (one area: the Box-Muller tables until the step's normals are formed, the staged tiles afterwards -- at
    //  40 KB a CU holds four workgroups, and the 1024 of N = 2^20 are resident at once)
    __shared__ __attribute__((aligned(16))) double s_raw[F1_STAGE * F_TILE];
    static_assert(sizeof(SmcD2) * SMC_NTAB_LDS_N <= sizeof(double) * F1_STAGE * F_TILE, "tables fit the staging area");
    SmcD2* s_ntab = reinterpret_cast<SmcD2*>(s_raw);
    double* sC = s_raw;                                 // the staged tiles' integer CDFs
    __shared__ double s_max[SMC_NWAVE];
    __shared__ double s_sum[2 * SMC_NWAVE];
    __shared__ double s_x[SMC_SM];
    __shared__ double sG[F_TILE + 4];                   // G_b, b = 0 .. ntiles (the last: all shares)
    __shared__ double sTb[F1_STAGE];
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y, tid = (int)threadIdx.x;
    const int lane = smc_lane(), wave = smc_wave();
    const i64 N = a.N;
    const int q = a.pp;
    const FOwn own = f_own<true>(b, tid, N);
    const double r0 = smc_ldg(a.info2 + (i64)isl * INFO_STRIDE);
    // ---- requests first: the partials of step t - 1 (parity q ^ 1), the Box-Muller tables
    const FArgs rd = f1_view(a, q ^ 1);
    const i64 o = (i64)isl * a.nparts;
    double pm4[4], ps4[4], pss4[4];
    {
        const bool pvec = (a.nparts & 3) == 0;
        f_load4<double>(rd.pm + o, (i64)tid * 4, a.nparts, pvec, -INFINITY, pm4);
        f_load4<double>(rd.ps + o, (i64)tid * 4, a.nparts, pvec, 0.0, ps4);
        f_load4<double>(rd.pss + o, (i64)tid * 4, a.nparts, pvec, 0.0, pss4);
    }
    SmcNtabRegs<SMC_BLOCK> ntr;
    smc_ntab_fetch<SMC_BLOCK>(ntr, tid);
    const u32 gisl = (u32)(a.island_offset + isl);
    const bool spec_z = a.tk >= 0 && !a.zt;
    u64 pa0 = 0ull, pa1 = 0ull, pb0 = 0ull, pb1 = 0ull;
    if (spec_z) {                                   // (the Philox calls need no table: they run under the loads)
        smc_normal_bits(a.seed, (u32)(own.na >> 1), (u32)a.tk, gisl, SMC_STREAM_NORMAL, pa0, pa1);
        smc_normal_bits(a.seed, (u32)(own.nb >> 1), (u32)a.tk, gisl, SMC_STREAM_NORMAL, pb0, pb1);
    }
    smc_ntab_store<SMC_BLOCK>(ntr, s_ntab, tid);
    if (!(smc_uniform(r0) < 1e17)) return;          // records frozen (k_theta_update)
    i64 t = (i64)smc_uniform(r0);
    if ((int)(t & 1) != q) t -= 1;                  // (a late workgroup may see workgroup 0's t + 1)
    if (t >= a.T || t < 0) return;
    const bool first = (t == 0);
    const double* p = a.params + (i64)isl * PARAM_STRIDE;
    const double yt = a.y[t * a.dy];
    const double aux = (m_has_aux<KIND>() && a.aux) ? a.aux[t] : 0.0;
    double* Xn = f_X(a, t) + (i64)isl * N;
    double* lwn = f_lw(a, t) + (i64)isl * N;
    const double* Xo = first ? Xn : f_X(a, t - 1) + (i64)isl * N;
    const double* lwo = first ? lwn : f_lw(a, t - 1) + (i64)isl * N;
    const double* zt = a.zt ? a.zt + ((i64)t * a.zt_ts + (i64)isl * N) : nullptr;

    // ---- all partials -> K, (s, ss), ESS, the decision (f2_reduce_island's operations, as in k_ancestors2)
    bool resample = false;
    double v4[4] = {0.0, 0.0, 0.0, 0.0};
    F2Red r;
    r.rs = 0.0;
    {
        double tm = smc_max2(smc_max2(pm4[0], pm4[1]), smc_max2(pm4[2], pm4[3]));
        tm = smc_wave_max(tm);
        if (lane == 0) s_max[wave] = tm;
        __syncthreads();                                       // (1) also: the tables are in LDS
        r.K = s_max[0];
#pragma unroll
        for (int w = 1; w < SMC_NWAVE; ++w) r.K = smc_max2(r.K, s_max[w]);
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double w;
            f2_rescale(pm4[k], r.K, ps4[k], pss4[k], v4[k], w);
            s1 = s1 + v4[k];
            s2 = s2 + w;
        }
        s1 = smc_wave_sum(s1);
        s2 = smc_wave_sum(s2);
        if (lane == 0) { s_sum[wave] = s1; s_sum[SMC_NWAVE + wave] = s2; }
        __syncthreads();                                       // (2)
        s1 = s_sum[0];
        s2 = s_sum[SMC_NWAVE];
#pragma unroll
        for (int w = 1; w < SMC_NWAVE; ++w) { s1 = s1 + s_sum[w]; s2 = s2 + s_sum[SMC_NWAVE + w]; }
        r.s = s1;
        r.ss = s2;
        f2_finish(a, r);
        if (!first) {
            resample = r.ess < a.ess_thresh;                   // core.py:181-183
            if (b == 0 && tid == 0) f2_write_record(a, isl, t, r, resample);
        }
    }
    // ---- standard normals: one Philox call per (even, odd) pair (started on the host's t above), or the tape
    double z[OPT];
    if (zt) {
        smc_ld2g(zt + own.na, z[0], z[1]);
        smc_ld2g(zt + own.nb, z[2], z[3]);
    } else if (spec_z && t == a.tk) {
        smc_normal_from_bits(s_ntab, pa0, pa1, z[0], z[1]);
        smc_normal_from_bits(s_ntab, pb0, pb1, z[2], z[3]);
    } else {
        smc_normal_pair(s_ntab, a.seed, (u32)(own.na >> 1), (u32)t, gisl, SMC_STREAM_NORMAL, z[0], z[1]);
        smc_normal_pair(s_ntab, a.seed, (u32)(own.nb >> 1), (u32)t, gisl, SMC_STREAM_NORMAL, z[2], z[3]);
    }
    double xp[OPT] = {0.0, 0.0, 0.0, 0.0}, lwp[OPT] = {0.0, 0.0, 0.0, 0.0};
    if (resample) {
        // ---- the shares Q_b of the 2^52 scale and their prefix (integers below 2^53: exact in fp64)
        double Qk[4], q4 = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            Qk[k] = (tid * 4 + k < a.nparts) ? f2_share(v4[k], r.rs) : 0.0;
            q4 += Qk[k];
        }
        double all;
        double ex = smc_block_exscan_f64(q4, s_x, all);
#pragma unroll
        for (int k = 0; k < 4; ++k) { sG[tid * 4 + k] = ex; ex += Qk[k]; }
        if (tid == SMC_BLOCK - 1) sG[F_TILE] = ex;
        // ---- the thresholds of my offspring: T_n = ceil(fl(fl(u_n + n) / N) 2^52)
        SmcSu su;
        u64 Us;
        f2_su(a, isl, t, su, Us);
        double Td[OPT];
#pragma unroll
        for (int k = 0; k < OPT; ++k) {
            const i64 n = f_own_idx(own, k);
            const double un = (a.scheme == SMC_SYSTEMATIC_) ? su.u_sys : smc_strat_u(su, (u64)n);
            Td[k] = (double)f2_t52_div(un + (double)n, su.dM);
        }
        __syncthreads();                                       // (3) sG
        // ---- the parent tiles of the workgroup's first and last offspring (every thread the same two searches:
        // LDS broadcasts), then each offspring's own among them: the largest bb with G_bb < T_n
        const int nt = a.ntiles;
        int blo = 0, bhi = 0;
        {
            const i64 n0 = (i64)b * F_TILE, n1 = n0 + F_TILE - 1;
            const double u0 = (a.scheme == SMC_SYSTEMATIC_) ? su.u_sys : smc_strat_u(su, (u64)n0);
            const double u1 = (a.scheme == SMC_SYSTEMATIC_) ? su.u_sys : smc_strat_u(su, (u64)n1);
            const double T0 = (double)f2_t52_div(u0 + (double)n0, su.dM), T1 = (double)f2_t52_div(u1 + (double)n1, su.dM);
#pragma unroll
            for (int step = F_TILE / 2; step; step >>= 1) {
                const int c0 = blo + step, c1 = bhi + step;
                if (c0 < nt && sG[c0] < T0) blo = c0;
                if (c1 < nt && sG[c1] < T1) bhi = c1;
            }
        }
        int bt[OPT];
#pragma unroll
        for (int k = 0; k < OPT; ++k) {
            int pos = blo;
            if (bhi - blo <= 4) {
#pragma unroll
                for (int s2 = 1; s2 <= 4; ++s2)
                    if (blo + s2 <= bhi && sG[blo + s2] < Td[k]) pos = blo + s2;
            } else {
                pos = 0;
#pragma unroll 1
                for (int step = F_TILE / 2; step; step >>= 1) {
                    const int c = pos + step;
                    if (c < nt && sG[c] < Td[k]) pos = c;
                }
            }
            bt[k] = pos;
        }
        int wt[F1_STAGE];
#pragma unroll
        for (int s = 0; s < F1_STAGE; ++s) wt[s] = blo + s < nt ? blo + s : nt - 1;
        if (bhi > wt[F1_STAGE - 1]) wt[F1_STAGE - 1] = bhi;
        const u64* cqi = rd.cq + (i64)isl * a.ncq;
        {
            u64 c0[F1_STAGE][2], c1[F1_STAGE][2];
#pragma unroll
            for (int s = 0; s < F1_STAGE; ++s) {
                smc_ld2g(cqi + (i64)wt[s] * F_TILE + 2 * tid, c0[s][0], c0[s][1]);
                smc_ld2g(cqi + (i64)wt[s] * F_TILE + 2 * SMC_BLOCK + 2 * tid, c1[s][0], c1[s][1]);
            }
            if (tid < F1_STAGE) {
                int w = wt[0];
#pragma unroll
                for (int s = 1; s < F1_STAGE; ++s) w = (tid == s) ? wt[s] : w;
                sTb[tid] = (double)smc_ldg(rd.tq + o + w);
            }
#pragma unroll
            for (int s = 0; s < F1_STAGE; ++s) {
                sC[s * F_TILE + 2 * tid] = (double)c0[s][0];
                sC[s * F_TILE + 2 * tid + 1] = (double)c0[s][1];
                sC[s * F_TILE + 2 * SMC_BLOCK + 2 * tid] = (double)c1[s][0];
                sC[s * F_TILE + 2 * SMC_BLOCK + 2 * tid + 1] = (double)c1[s][1];
            }
        }
        __syncthreads();                                       // (5)
        u32 an[OPT];
        int prev_pos = 0;
        bool prev_exact = true;
#pragma unroll
        for (int k = 0; k < OPT; ++k) {
            const int bb = bt[k];
            int slot = -1;
#pragma unroll
            for (int s = F1_STAGE - 1; s >= 0; --s) slot = (wt[s] == bb) ? s : slot;
            const double Gb = sG[bb], Qb = sG[bb + 1] - Gb, D = Td[k] - Gb;
            int pos = 0;
            bool exact = slot < 0;
            if (!(D > 0.0)) {
                pos = 0;                                       // T_n = 0: parent 0 (ns(0) = 0)
                exact = false;
            } else if (!(Qb > 0.0)) {
                pos = F_TILE - 1;                              // a last tile without a share: the clamp
                exact = false;
            } else if (slot >= 0) {
                const double x = D * (sTb[slot] / Qb);
                const double band = 128.0 + x * 0x1.0p-50;
                const double xl = x - band, xh = x + band;
                const double* Cs = sC + slot * F_TILE;
                // (the second offspring of a pair starts where the first ended: its parent is that one or one
                //  of the next few -- a short walk instead of ten probes)
                bool walked = false;
                if ((k & 1) && bt[k - 1] == bb && !prev_exact) {
                    pos = prev_pos;
                    bool go = true;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const int c1 = pos + 1;
                        const bool in = go && c1 < F_TILE;
                        const double c = Cs[in ? c1 : pos];
                        exact = exact || (in && c >= xl && c <= xh);
                        go = in && c < xl;
                        pos = go ? c1 : pos;
                    }
                    walked = !go;                              // (still advancing: the full search decides)
                    if (go) pos = 0;
                }
                if (!walked) {
#pragma unroll
                    for (int step = F_TILE / 2; step; step >>= 1) {
                        const double c = Cs[pos + step];
                        exact = exact || (c >= xl && c <= xh);
                        if (c < xl) pos += step;
                    }
                }
            }
            if (exact)
                pos = f1_exact_parent(cqi + (i64)bb * F_TILE, smc_ldg(rd.tq + o + bb), (u64)Qb, (u64)D);
            prev_pos = pos;
            prev_exact = exact;
            an[k] = (u32)(bb * F_TILE + pos);
        }
        u32* Aw = f_A(a, t) + (i64)isl * N;
        smc_st2g(Aw + own.na, an[0], an[1]);
        smc_st2g(Aw + own.nb, an[2], an[3]);
#pragma unroll
        for (int k = 0; k < OPT; ++k) xp[k] = smc_ldg(Xo + an[k]);                 // core.py:332
    } else if (!first) {
        smc_ld2g(Xo + own.na, xp[0], xp[1]);
        smc_ld2g(lwo + own.na, lwp[0], lwp[1]);
        smc_ld2g(Xo + own.nb, xp[2], xp[3]);
        smc_ld2g(lwo + own.nb, lwp[2], lwp[3]);
    }
    double xn[OPT], lw[OPT];
#pragma unroll
    for (int k = 0; k < OPT; ++k) {
        double inc;
        xn[k] = m_step<KIND, FK>(p, first, yt, aux, xp[k], z[k], inc);
        double l = (first || resample) ? inc : lwp[k] + inc;                       // resampling.py:241-244
        if (l != l) l = -INFINITY;                                               // resampling.py:220
        lw[k] = l;
    }
    if (a.nt) {
        smc_st2g_nt(Xn + own.na, xn[0], xn[1]);
        smc_st2g_nt(Xn + own.nb, xn[2], xn[3]);
        smc_st2g_nt(lwn + own.na, lw[0], lw[1]);
        smc_st2g_nt(lwn + own.nb, lw[2], lw[3]);
    } else {
        smc_st2g(Xn + own.na, xn[0], xn[1]);
        smc_st2g(Xn + own.nb, xn[2], xn[3]);
        smc_st2g(lwn + own.na, lw[0], lw[1]);
        smc_st2g(lwn + own.nb, lw[2], lw[3]);
    }
    // ---- the tile's partial and integer CDF, for step t + 1 (parity q)
    const FArgs wr = f1_view(a, q);
    u64 cx[4];
    const F2Tile rt = f2_tile_weights(lw, cx);
    u64* cqo = wr.cq + (i64)isl * a.ncq;
    if (a.nt) { smc_st2g_nt(cqo + own.na, cx[0], cx[1]); smc_st2g_nt(cqo + own.nb, cx[2], cx[3]); }
    else { smc_st2g(cqo + own.na, cx[0], cx[1]); smc_st2g(cqo + own.nb, cx[2], cx[3]); }
    if (tid == 0) {
        wr.pm[o + b] = rt.K;
        wr.ps[o + b] = rt.S;
        wr.pss[o + b] = rt.SS;
        wr.tq[o + b] = rt.tb;
        if (b == 0) a.info2[(i64)isl * INFO_STRIDE] = (double)(t + 1);
    }
}
