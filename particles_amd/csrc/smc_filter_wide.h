// smc_filter_wide.h -- k_ancestors2w: the resident two-level resampling launch (every workgroup
// reduces the island's partials itself: k_ancestors2<MID = false>) with TPW tiles of parents per
// workgroup instead of one.
//
// Why.  With one tile per workgroup all 4 waves of all 4 workgroups of a CU reduce the same <= 1024
// partials at the same time: 16 waves x ~230 VALU instructions per CU for one result, and since a
// SIMD issues one wave at a time the phase lasts 4 x 230 issue slots -- 1.6 us of C2's 8.5 us launch
// (k_ancestors2<MID = true> behind a k_reduce2 that did the reduction once: 6.9 us; profiles/r12b).
// Here the workgroup has TPW x 256 threads, its FIRST 256 (waves 0..3: one per SIMD) reduce the partials
// with exactly the operations of k_ancestors2 -- same loads, same association order, same bits --
// while the other waves wait at the barriers without issuing anything, and everybody picks the
// result up from LDS.  From there on a thread does for its tile what k_ancestors2's thread does.
//
// Control flow is uniform over the workgroup (barriers are workgroup-wide): a tile that needs a
// second scatter pass or owns a heavy parent takes ALL tiles of the workgroup through the pass loop
// (`slow`, decided from the tiles' shares, which every thread holds).
//
// Contract, results and the heavy-parent protocol with k_propagate are k_ancestors2's; no k_reduce2 in
// front (a.ntiles <= 1024).  POW2 = true: N = 2^k, systematic / stratified with closed-form integer counts,
// a.ntiles % TPW == 0.  POW2 = false (systematic): any N -- the general counts, any number of tiles (the last
// workgroup of a run of tiles may hold fewer than TPW), a ragged last tile.
#pragma once

template <int TPW, int SCH, bool POW2 = true>
__global__ void __launch_bounds__(SMC_BLOCK * TPW)
k_ancestors2w(const double* __restrict__ pre_info2, const double* __restrict__ pre_pm, const double* __restrict__ pre_ps,
              const double* __restrict__ pre_pss, const u64* __restrict__ pre_cq, const i64 pre_N, const int pre_geom,
              const FArgs av)
{
    // pre_geom: ntiles (== nparts on this path) | xcd_chunks << 30
    // (the leading arguments repeat fields of the block: scalar kernel arguments in front can be PRELOADED into SGPRs
    //  by the command processor -- -amdgpu-kernarg-preload-count -- so the launch's first loads need not wait for a
    //  scalar load of their own addresses from the kernarg segment; a firmware that does not preload runs the
    //  compiler's prologue, which fetches them the old way)
    static_assert(TPW == 2 || TPW == 4, "tiles per workgroup");
    static_assert(SCH == SMC_SYSTEMATIC_ || SCH == SMC_STRATIFIED_, "closed-form counts");
    static_assert(POW2 || SCH == SMC_SYSTEMATIC_, "general N: the systematic scheme");
    const FArgs& a = av;
    constexpr int WIN = 2 * F_PASS;                                        // offspring per pass
    __shared__ __attribute__((aligned(16))) u32 sP_all[TPW][WIN];
    __shared__ double s_max[SMC_NWAVE];                                    // one area per exchange
    __shared__ double s_sum[2 * SMC_NWAVE];
    __shared__ double s_wt[SMC_NWAVE];                                     // the shares held by each reducing wave
    __shared__ double s_before[TPW], s_Qb[TPW];                            // per tile: shares before it within its wave; its own
    __shared__ int s_dec;
    __shared__ u32 s_mx_all[TPW][2 * SMC_NWAVE];
    __shared__ i64 s_n_all[TPW][2];
    __shared__ int s_more[TPW];
    __shared__ i64 sH_all[TPW][2 * F_HLOC];
    __shared__ unsigned sHn_all[TPW];
    const int st = (int)threadIdx.x / SMC_BLOCK;                           // which tile of the workgroup
    const int tid = (int)threadIdx.x % SMC_BLOCK;                          // thread of the tile
    const int lane = (int)(threadIdx.x & 63u), wave = tid >> 6;            // wave of the tile
    // which tiles: tile b is written by workgroup b of k_propagate and read by workgroup b of the next one, i.e. on
    // XCD b % 8 (workgroups go round the 8 XCDs, each with an L2 of its own) -- a workgroup here takes TPW tiles
    // of ITS XCD (b = x, x + 8, ... within a group of 8 TPW tiles) where the grid allows, so that the integer CDF it
    // reads and the ancestors it writes stay in that L2 (consecutive tiles: C2 19.7 us per step against 18.1 with
    // one tile per workgroup, profiles/r12c)
    const int ntiles = pre_geom & 0x3FFFFFFF;
    const bool xcd_chunks = (pre_geom >> 30) != 0;
    const bool xcd_map = (ntiles % (8 * TPW)) == 0;
    const int bx = (int)blockIdx.x;
    // (xcd_chunks -- f_tile_xcd: XCD x owns a contiguous run of tiles -- : TPW consecutive tiles of this XCD's run)
    int b_first = xcd_chunks ? (bx & 7) * (ntiles >> 3) + (bx >> 3) * TPW
                             : (xcd_map ? (bx / 8) * (8 * TPW) + (bx % 8) : bx * TPW);
    int b_step = xcd_chunks ? 1 : (xcd_map ? 8 : 1);
    int n_here = TPW;                                                      // this workgroup's tiles
    if (!POW2) {
        // any number of tiles (POW2 = false: general N): consecutive tiles of the run f_tile_xcd gives this XCD (or of
        // the whole range), the last workgroup of a run holding fewer than TPW -- its other waves go through the
        // barriers with a copy of the first tile's data and write nothing
        b_step = 1;
        if (xcd_chunks) {
            const int c = bx & 7, base = ntiles >> 3, rem = ntiles & 7;
            const int len = base + (c < rem ? 1 : 0), q0 = (bx >> 3) * TPW;
            b_first = c * base + (c < rem ? c : rem) + q0;
            n_here = len - q0;
        } else {
            b_first = bx * TPW;
            n_here = ntiles - b_first;
        }
        if (n_here <= 0) return;                                           // (uniform: before any barrier)
        n_here = n_here > TPW ? TPW : n_here;
    }
    const bool tile_ok = POW2 || st < n_here;
    const int b = tile_ok ? b_first + st * b_step : b_first, isl = (int)blockIdx.y;
    const bool red = st == 0;                                              // the reducing waves
    u32* sP = sP_all[st];
    u32* s_mx = s_mx_all[st];
    const i64 N = pre_N;
    F_STAMP_A(0);
    const i64 j0 = (i64)b * F_TILE;
    const i64 jt = j0 + (i64)tid * F_IPT;
    double* info = a.info + (i64)isl * INFO_STRIDE;
    const double r0 = smc_ldg(pre_info2 + (i64)isl * INFO_STRIDE);
    const i64 o = (i64)isl * ntiles;                                       // (nparts == ntiles: one partial per tile)
    double pm4[4] = {0.0, 0.0, 0.0, 0.0}, ps4[4] = {0.0, 0.0, 0.0, 0.0}, pss4[4] = {0.0, 0.0, 0.0, 0.0};
    // the tile's integer CDF: this thread's 4 positions and the next thread's first
    const u64* cq = pre_cq + (i64)isl * ((i64)ntiles * F_TILE);            // (ncq = ntiles x 1024; N = 2^k: == N)
    u64 cx[F_IPT + 1];
    smc_ld2g(cq + jt, cx[0], cx[1]);
    smc_ld2g(cq + jt + 2, cx[2], cx[3]);
    cx[4] = (tid < SMC_BLOCK - 1) ? smc_ldg(cq + jt + 4) : 0ull;
    if (red) {
        const bool pvec = (ntiles & 3) == 0;
        f_load4<double>(pre_pm + o, (i64)tid * 4, ntiles, pvec, -INFINITY, pm4);
        f_load4<double>(pre_ps + o, (i64)tid * 4, ntiles, pvec, 0.0, ps4);
        f_load4<double>(pre_pss + o, (i64)tid * 4, ntiles, pvec, 0.0, pss4);
    }
    const u64 tb_raw = smc_ldg(a.tq + o + b);      // (the first use of the argument block: its scalar loads have had the time of the requests above)
    // the scatter window of the first pass, while the loads are on their way
    *reinterpret_cast<uint4*>(&sP[tid * 4]) = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(&sP[F_PASS + tid * 4]) = make_uint4(0u, 0u, 0u, 0u);
    const i64 t = (i64)smc_uniform(r0);
    if (t >= a.T) {
        if (blockIdx.x == 0 && threadIdx.x == 0) info[0] = (double)t;     // k_propagate returns on it
        return;
    }
    if (t == 0) return;                                        // the host wrote the record of step 0
    F2RecIn rin = {0.0, 0.0, 0.0, 0.0, 0.0};                   // (what the record's writer reads: on its way during the reduction)
    if (blockIdx.x == 0 && tid == 0) rin = f2_record_loads(a, isl, t);
    F_STAMP_A(1);
    SmcSu su;                                                  // (the step's uniform: one Philox call,
    u64 Us;                                                    //  all inputs uniform: scalar unit)
    f2_su(a, isl, t, su, Us, SCH);
    // ---- all partials -> K, (s, ss), ESS, the decision (k_ancestors2's operations, f2_reduce_island's bits)
    if (red) {
        double tm = smc_max2(smc_max2(pm4[0], pm4[1]), smc_max2(pm4[2], pm4[3]));
        tm = smc_wave_max(tm);
        if (lane == 0) s_max[wave] = tm;
    }
    __syncthreads();                                           // (1) also: sP zeroed
    F_STAMP_A(2);
    double v4[4] = {0.0, 0.0, 0.0, 0.0};
    F2Red r;
    r.K = 0.0;
    if (red) {
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
        smc_wave_sum2(s1, s2);
        if (lane == 0) { s_sum[wave] = s1; s_sum[SMC_NWAVE + wave] = s2; }
    }
    __syncthreads();                                           // (2)
    if (red) {
        double s1 = s_sum[0], s2 = s_sum[SMC_NWAVE];
#pragma unroll
        for (int w = 1; w < SMC_NWAVE; ++w) { s1 = s1 + s_sum[w]; s2 = s2 + s_sum[SMC_NWAVE + w]; }
        r.s = s1;
        r.ss = s2;
        f2_finish(a, r);
        const bool resample = r.ess < a.ess_thresh;            // core.py:181-183 (t < T here)
        if (blockIdx.x == 0 && tid == 0) f2_write_record(a, isl, t, r, resample, rin);
        if (tid == 0) s_dec = resample ? 1 : 0;
        // ---- the shares Q_b of this workgroup's tiles and the shares before each of them: ONE scan of the
        // threads' sums (integers below 2^53: every association gives the same doubles), from which the thread
        // that holds tile b's partial (tid == b / 4) reads off what lies in front of b within its wave; the
        // waves' totals complete it on the other side of the barrier
        double Q4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) Q4[k] = (tid * 4 + k < a.nparts) ? f2_share(v4[k], r.rs) : 0.0;
        const double q01 = Q4[0] + Q4[1], q012 = q01 + Q4[2], qsum = q012 + Q4[3];
        const double incl = smc_wave_scan_add_f64(qsum);
        if (lane == 63) s_wt[wave] = incl;
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int bq = (POW2 || q < n_here) ? b_first + q * b_step : b_first;
            if (tid == (bq >> 2)) {
                const int kk = bq & 3;
                s_before[q] = (incl - qsum) + (kk == 0 ? 0.0 : (kk == 1 ? Q4[0] : (kk == 2 ? q01 : q012)));
                s_Qb[q] = kk == 0 ? Q4[0] : (kk == 1 ? Q4[1] : (kk == 2 ? Q4[2] : Q4[3]));
            }
        }
    }
    __syncthreads();                                           // (3)
    F_STAMP_A(3);
    if (!s_dec) return;
    double Gd = s_before[st];
    const int wave_b = b >> 8;                                 // the wave of the thread that holds tile b's partial
#pragma unroll
    for (int w = 0; w < SMC_NWAVE - 1; ++w) Gd = Gd + (w < wave_b ? s_wt[w] : 0.0);
    double Qall[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) Qall[k] = s_Qb[k];
    double Qd = Qall[0];
#pragma unroll
    for (int k = 1; k < TPW; ++k) Qd = (k == st) ? Qall[k] : Qd;
    const u64 tb = smc_uniform_u64(tb_raw);
    if (tid == SMC_BLOCK - 1) cx[4] = tb;
    // ---- first offspring of each parent; the tile's range [n_lo, n_hi)
    const u64 Gb = (u64)Gd, Qb = (u64)Qd;
    i64 ns[F_IPT + 1], n_lo, n_hi;
    const double down = (double)N * 0x1.0p-52;            // offspring per unit of the 2^52 scale (2^-sh for N = 2^k)
    F2Fast f;
    f.Gb = Gb; f.Qb = Qb; f.tb = tb;
    f.Gd = Gd * down;
    f.r = tb ? (Qd / (double)tb) * down : 0.0;
    f.u = SCH == SMC_SYSTEMATIC_ ? su.u_sys : 0.0;
    f.dN = (double)N;
    f.eps = a.exact_counts ? 2.0 : f.dN * 0x1.0p-49;                 // (2.0: always the exact route)
    f.one_m_eps = 1.0 - f.eps;
    if (SCH == SMC_SYSTEMATIC_) {
#pragma unroll
        for (int i = 0; i <= F_IPT; ++i) {
            const i64 j = jt + i;
            ns[i] = (j == 0) ? 0 : (j >= N ? N : f2_ns_sys<POW2>(a, su, Us, f, cx[i]));
        }
        if (!POW2) {                               // (general N: the counts the fp64 shortcut left open, formed exactly)
            unsigned need = 0u;
#pragma unroll
            for (int i = 0; i <= F_IPT; ++i) need |= (ns[i] < 0) ? (1u << i) : 0u;
            if (need) {
                const u64 c7[7] = {cx[0], cx[1], cx[2], cx[3], cx[4], 0ull, tb};
                i64 n7[7] = {ns[0], ns[1], ns[2], ns[3], ns[4], 0, 0};
                f2_resolve_general(su, need, Gb, Qb, tb, c7, n7);
#pragma unroll
                for (int i = 0; i <= F_IPT; ++i) ns[i] = n7[i];
            }
        }
    } else {
        // the stratified uniforms of each tile's offspring staged in LDS (k_ancestors2: F2_SU_PAIRS Philox calls per
        // tile instead of one per boundary); every tile of the workgroup must fit its window -- the barrier is shared
        __shared__ __attribute__((aligned(16))) double sU_all[TPW][2 * F2_SU_PAIRS];
        double* sU = sU_all[st];
        const double nfirst = floor(f.Gd * 0.5) * 2.0;                     // first staged offspring (even)
        bool stage = !su.u;
#pragma unroll
        for (int k = 0; k < TPW; ++k) stage = stage && (Qall[k] * down + 4.0 <= (double)(2 * F2_SU_PAIRS - 2));
        if (stage) {
            const u32 p0 = (u32)(nfirst * 0.5);
#pragma unroll
            for (int r = 0; r < (F2_SU_PAIRS + SMC_BLOCK - 1) / SMC_BLOCK; ++r) {
                const int q = tid + r * SMC_BLOCK;
                if (q < F2_SU_PAIRS && nfirst < f.dN) {
                    u64 xa, xb;
                    smc_philox(p0 + (u32)q, su.t, su.island, SMC_STREAM_RESAMPLE, su.seed, xa, xb);
                    double2 uu;
                    uu.x = smc_u01_halfopen(xa);
                    uu.y = smc_u01_halfopen(xb);
                    *reinterpret_cast<double2*>(&sU[2 * q]) = uu;
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i <= F_IPT; ++i) {
                const i64 j = jt + i;
                ns[i] = (j == 0) ? 0 : (j >= N ? N : f2_ns_strat_lds(a, su, Us, f, cx[i], sU, nfirst));
            }
        } else {
#pragma unroll
            for (int i = 0; i <= F_IPT; ++i) {
                const i64 j = jt + i;
                ns[i] = (j == 0) ? 0 : (j >= N ? N : f2_ns_strat(a, su, Us, f, cx[i]));
            }
        }
    }
    {
        // the tile's range [n_lo, n_hi): its first thread's first count and its last thread's last (positions 0 and
        // t_b of the tile's CDF) -- handed round through LDS; evaluating them in every thread was 2 of its 7 counts
        i64* s_n = s_n_all[st];
        if (tid == 0) s_n[0] = ns[0];
        if (tid == SMC_BLOCK - 1) s_n[1] = ns[F_IPT];
        __syncthreads();
        n_lo = s_n[0];
        n_hi = s_n[1];
    }
    F_STAMP_A(4);
    u32* A = f_A(a, t) + (i64)isl * N;
    // (32-bit arithmetic from here on: N <= 2^30 on this path, offspring indices fit)
    u32 nsu[F_IPT + 1];
#pragma unroll
    for (int i = 0; i <= F_IPT; ++i) nsu[i] = (u32)ns[i];
    const u32 lo = (u32)n_lo, hi = (u32)n_hi;
    const u32 jb = (u32)j0;
    // one scatter pass over the window [pb, pb + WIN): every parent writes its index at its first
    // offspring's slot (sP zeroed), a running maximum over the slots gives each offspring its parent.
    // Barriers inside: called by every thread of the workgroup, `act` says whether this tile takes part.
    auto pass = [&](const u32 pb, const bool act) {
        int rel[F_IPT + 1];
#pragma unroll
        for (int i = 0; i <= F_IPT; ++i) {
            const int d = (int)(nsu[i] - pb);
            rel[i] = d < 0 ? 0 : (d > WIN ? WIN : d);
        }
        if (act) {
#pragma unroll
            for (int i = 0; i < F_IPT; ++i)
                if (rel[i] < rel[i + 1]) sP[rel[i]] = (u32)(tid * F_IPT + i);
        }
        const bool two = act && hi > pb + F_PASS;  // does the second half of the window hold offspring?
        __syncthreads();
        const uint4 v = *reinterpret_cast<const uint4*>(&sP[tid * 4]);
        uint4 w = make_uint4(0u, 0u, 0u, 0u);
        if (two) w = *reinterpret_cast<const uint4*>(&sP[F_PASS + tid * 4]);
        const u32 m0 = v.x, m1 = m0 > v.y ? m0 : v.y, m2 = m1 > v.z ? m1 : v.z, m3 = m2 > v.w ? m2 : v.w;
        const u32 k0 = w.x, k1 = k0 > w.y ? k0 : w.y, k2 = k1 > w.z ? k1 : w.z, k3 = k2 > w.w ? k2 : w.w;
        const u32 inc1 = smc_wave_scan_max_u32(m3);
        u32 ex1 = smc_mov_dpp<SMC_DPP_WAVE_SHR1>(inc1);
        if (lane == 0) ex1 = 0u;
        u32 inc2 = 0u, ex2 = 0u;
        if (two) {
            inc2 = smc_wave_scan_max_u32(k3);
            ex2 = smc_mov_dpp<SMC_DPP_WAVE_SHR1>(inc2);
            if (lane == 0) ex2 = 0u;
        }
        if (lane == 63) { s_mx[wave] = inc1; s_mx[SMC_NWAVE + wave] = inc2; }
        __syncthreads();
        u32 all1 = 0u;
#pragma unroll
        for (int ww = 0; ww < SMC_NWAVE; ++ww) {
            const u32 x1 = s_mx[ww], x2 = s_mx[SMC_NWAVE + ww];
            all1 = all1 > x1 ? all1 : x1;
            if (ww < wave) {
                ex1 = ex1 > x1 ? ex1 : x1;
                ex2 = ex2 > x2 ? ex2 : x2;
            }
        }
        ex2 = ex2 > all1 ? ex2 : all1;             // the second half continues the first
        if (act) {
            const u32 n0 = pb + (u32)tid * 4u;
            const u32 a32[4] = {jb + (m0 > ex1 ? m0 : ex1), jb + (m1 > ex1 ? m1 : ex1),
                                jb + (m2 > ex1 ? m2 : ex1), jb + (m3 > ex1 ? m3 : ex1)};
            if (n0 >= lo && n0 + 3u < hi) {                                             // core.py:329
                if (a.nt & 8) smc_st4g_nt(A + n0, a32);
                else smc_st4g(A + n0, a32);
            } else {
#pragma unroll
                for (u32 i = 0; i < 4u; ++i)
                    if (n0 + i >= lo && n0 + i < hi) smc_stg(A + n0 + i, a32[i]);
            }
        }
        if (two) {
            const u32 n0 = pb + F_PASS + (u32)tid * 4u;
            const u32 a32[4] = {jb + (k0 > ex2 ? k0 : ex2), jb + (k1 > ex2 ? k1 : ex2),
                                jb + (k2 > ex2 ? k2 : ex2), jb + (k3 > ex2 ? k3 : ex2)};
            if (n0 + 3u < hi) {                    // (n0 >= lo: the second half starts 1024 past it)
                if (a.nt & 8) smc_st4g_nt(A + n0, a32);
                else smc_st4g(A + n0, a32);
            } else {
#pragma unroll
                for (u32 i = 0; i < 4u; ++i)
                    if (n0 + i < hi) smc_stg(A + n0 + i, a32[i]);
            }
        }
    };
    // does any tile of the workgroup need the pass loop (more than one window, or a heavy parent)?  A tile's
    // offspring number at most Q_b N 2^-52 + 2 (count(C) = floor(C N 2^-52) + [0 or 1]); decided from the
    // shares, which every thread holds: uniform over the workgroup without an exchange
    bool slow = false;
#pragma unroll
    for (int k = 0; k < TPW; ++k) slow = slow || !(Qall[k] * down + 2.0 <= (double)(WIN - 4));
    if (!slow) {                                   // one window per tile (an empty tile goes through the barriers)
        const bool act = tile_ok && lo < hi;
        pass(act ? (lo & ~3u) : 0u, act);
        F_STAMP_A(5);
        return;
    }
    // ---- the general case, all tiles together.  Heavy parents (>= 2048 offspring): registered, their
    // whole blocks left to k_propagate (f_register_heavy's protocol, one list per tile of the workgroup)
    i64* sH = sH_all[st];
    int nH = 0;
    if (a.hcnt) {
        const bool cand = tile_ok && n_hi - n_lo >= 2 * (i64)F_TILE;
        unsigned* hcnt = a.hcnt + (i64)isl * 2 + (t & 1);
        i64* hlist = a.hlist + ((i64)isl * 2 + (t & 1)) * F_HMAX * 3;
        if (tid == 0) sHn_all[st] = 0u;
        __syncthreads();
        if (cand) {
            for (int i = 0; i < F_IPT; ++i) {
                const i64 bs = ((ns[i] + F_TILE - 1) / F_TILE) * F_TILE, be = (ns[i + 1] / F_TILE) * F_TILE;
                if (ns[i + 1] - ns[i] >= 2 * (i64)F_TILE && be > bs) {
                    const unsigned g = atomicAdd(hcnt, 1u);
                    if (g < F_HMAX) {
                        const unsigned k = atomicAdd(&sHn_all[st], 1u);
                        i64* e = hlist + (i64)g * 3;
                        if (k < F_HLOC) {
                            e[0] = bs; e[1] = be; e[2] = jt + i;
                            sH[2 * k] = bs; sH[2 * k + 1] = be;
                        } else {                   // no room here: the entry stays harmless (empty)
                            e[0] = 0; e[1] = 0; e[2] = 0;
                        }
                    }
                }
            }
        }
        __syncthreads();
        nH = (int)(sHn_all[st] < F_HLOC ? sHn_all[st] : F_HLOC);
    }
    u32 pb = lo & ~3u;
    bool first_pass = true;
    for (;;) {
        // this tile's next window: the passes that lie wholly inside a registered parent's blocks are left out
        bool act = tile_ok && pb < hi;
        while (act && nH) {
            const i64 w_lo = pb > lo ? pb : lo, w_hi = pb + WIN < hi ? pb + WIN : hi;
            i64 jump = 0;                          // passes to leave out, this one included
            for (int k = 0; k < nH; ++k)
                if (sH[2 * k] <= w_lo && w_hi <= sH[2 * k + 1]) {
                    const i64 whole = (sH[2 * k + 1] - (i64)pb) / WIN;      // passes that end inside the blocks
                    jump = whole > 1 ? whole : 1;
                }
            if (!jump) break;
            pb += (u32)jump * WIN;
            act = pb < hi;
        }
        __syncthreads();                           // previous pass has read sP and the votes
        if (!first_pass) {
            *reinterpret_cast<uint4*>(&sP[tid * 4]) = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(&sP[F_PASS + tid * 4]) = make_uint4(0u, 0u, 0u, 0u);
        }
        if (tid == 0) s_more[st] = act ? 1 : 0;
        __syncthreads();
        bool any = false;
#pragma unroll
        for (int k = 0; k < TPW; ++k) any = any || s_more[k] != 0;
        if (!any) break;
        pass(pb, act);
        first_pass = false;
        if (act) pb += WIN;
    }
}
