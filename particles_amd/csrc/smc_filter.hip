// smc_filter.hip -- host side of the fused on-device SMC step loop
// (smc_filter_* of include/smc_hip.h).  The kernels and the description of the
// two-kernel step are in smc_filter_kernels.h.
#include <type_traits>
#include <vector>

#include "smc_filter_mv.h"
#include "smc_filter_small.h"
#include "smc_filter_sqmc.h"
#include "smc_filter_wide.h"
#include "smc_filter_strict.h"

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
#define PROF_MAX 4096

struct smc_filter {
    smc_ctx* ctx;
    FArgs a;               // the argument block: passed BY VALUE (kernarg segment: one scalar-load hop,
                           // and the compiler knows its pointers address global memory)
    int kind, fk;
    i64 t_host;
    void* slab;            // one allocation holding every device array
    size_t slab_bytes;
    bool use_graph;
    bool fused;            // k_ancestors<true> (no k_prepare launch)
    bool two_level;        // k_ancestors2 + tail-free k_propagate (two-level CDF, no intra-launch exchange)
    bool reduce_narrow;    // SMC_PATH_NO_WIDE: k_reduce2 also where k_reduce2w would run (the A/B the tests compare)
    bool two_level_mid;    // ... with k_reduce2 in front (grids too large for every workgroup to reduce)
    int ragged;            // two-level step with N not a multiple of the tile: 1 (N even), 2 (N odd), else 0
                           // (k_propagate<.., RAGGED>)
    bool mv_collapsed;     // MVLINGAUSS guided: log G = log p(y_t | x_{t-1}) in one product (opts.flags)
    bool strict;           // SMC_FLAG_STRICT_ANCESTORS: sequential fp64 CDF of the filter's weights
    double* strict_ws;     // (n_islands, N) W | (n_islands, N) S | scratch of smc_seqsum.h
    bool strict_literal;   // SMC_PATH_STRICT_LITERAL: S by the one-lane walk, in place
    bool sq_plan_zero;     // SQMC: the sort workspace's plan words are zero (smc_rs_sort_ws leaves them so)
    // SMC_FLAG_SQMC (smc_filter_sqmc.h): the point stream, the tape of ndtri(second coordinate), the
    // sort's workspace and -- more than one island -- the islands' permutations
    u64 sp_epoch;          // launches of the merged spacings + reduction kernel so far (see FArgs::sp_epoch)
    bool sp_merge;
    bool flush_pending;    // two-level step: the summary row of the last step enqueued is still to be written (k_flush2 on demand)
    bool sqmc, sq_gather;
    bool sq_flat;          // SQMC on the flat step (multivariate filters; univariate ones below two tiles)
    u64 sq_seed, sq_ctr0;
    double* sq_z;
    u64* sq_perm;
    void* sq_ws;
    i64 perm_t;            // t_host at the last smc_filter_permute_islands (A / Xp undefined there)
    hipGraphExec_t gexec[3];   // captured step sequences of F_GRAPH_SIZES steps (even: see enqueue_step)
    bool graph_failed;
    bool prof;
    std::vector<hipEvent_t> ev;
    int prof_n;
    bool no_tk;            // SMC_PATH_NO_TK: kernels never start on the host's time index (A/B)
    bool no_small;         // SMC_PATH_NO_SMALL
    int wide_tpw;          // k_ancestors2w: tiles per workgroup (0: k_ancestors2, one tile per workgroup)
    double* tmp;           // (N,) staging for W / Xp downloads
    double* ll_stage;      // (n_islands,) staging for smc_filter_logLt: PINNED host memory the
                           // collect kernel writes straight into (no copy engine, no staging)
    // SMC^2 theta level (smc_filter_theta_enable): theta log-weights, stop record, ESS log
    double *lwth, *th, *th_ess;
    double th_ess_min;
    void* th_buf;
    // ... of a population sharded over ranks (smc_filter_theta_enable_sharded): the theta level is
    // replicated -- th_n = nranks x n_islands log-weights on every rank -- and fed by an all-gather of the
    // ranks' evidence increments enqueued behind every step
    smc_comm* th_comm;
    int th_n;
    double *th_send, *th_recv;
};

typedef void (*move_fn)(FArgs);

// the island's reduction as a launch of its own: islands of 1025 .. 4096 tiles by a workgroup of 1024 threads (k_reduce2w:
// the same bits), anything else -- and the APF's two sets of partials -- by k_reduce2
static void launch_reduce2(smc_filter* f, hipStream_t st)
{
    const int nchunks = (f->a.nparts + 4 * SMC_BLOCK - 1) / (4 * SMC_BLOCK);
    if (nchunks >= 2 && nchunks <= 4 && !f->a.pm2 && !f->reduce_narrow)
        SMC_LAUNCH(k_reduce2w, dim3(f->a.n_islands), dim3(4 * SMC_BLOCK), st, f->a);
    else
        SMC_LAUNCH(k_reduce2, dim3(f->a.n_islands), dim3(SMC_BLOCK), st, f->a);
}

#define F_OPT 4     /* new particles per thread of k_propagate */

static void launch_propagate(smc_filter* f)
{
    hipStream_t st = f->ctx->stream;
    const dim3 grid(f->a.nparts, f->a.n_islands);
    if (f->kind == SMC_MODEL_MVLINGAUSS) {
#define MV_CASE(FKV, DPV, COLLV)                                                            \
    if (f->fk == FKV && f->a.dp == DPV && f->mv_collapsed == COLLV) {                       \
        if (f->a.dx == DPV && f->a.mv_diag)                                                 \
            SMC_LAUNCH((k_propagate_mv<FKV, DPV, true, COLLV, true>), grid, dim3(SMC_BLOCK), st,  \
                       f->a, f->a.mvc);                                                     \
        else if (f->a.mv_diag)                                                              \
            SMC_LAUNCH((k_propagate_mv<FKV, DPV, false, COLLV, true>), grid, dim3(SMC_BLOCK), st, \
                       f->a, f->a.mvc);                                                     \
        else if (f->a.dx == DPV)                                                            \
            SMC_LAUNCH((k_propagate_mv<FKV, DPV, true, COLLV>), grid, dim3(SMC_BLOCK), st,  \
                       f->a, f->a.mvc);                                                     \
        else                                                                                \
            SMC_LAUNCH((k_propagate_mv<FKV, DPV, false, COLLV>), grid, dim3(SMC_BLOCK), st, \
                       f->a, f->a.mvc);                                                     \
        return;                                                                             \
    }
        MV_CASE(SMC_FK_BOOTSTRAP, 16, false) MV_CASE(SMC_FK_BOOTSTRAP, 32, false)
        MV_CASE(SMC_FK_GUIDED, 16, false) MV_CASE(SMC_FK_GUIDED, 32, false)
        MV_CASE(SMC_FK_GUIDED, 16, true) MV_CASE(SMC_FK_GUIDED, 32, true)
        MV_CASE(SMC_FK_APF, 16, false) MV_CASE(SMC_FK_APF, 32, false)
#undef MV_CASE
        return;
    }
// (k_propagate's leading arguments: the fields its first loads are addressed with, preloaded into SGPRs)
#define P_LEAD f->a.A, f->a.info, f->a.params, f->a.hcnt, f->a.N, f->a.ntiles | (f->a.xcd_chunks ? 1 << 30 : 0)
#define P_CASE(KINDV, FKV)                                                                    \
    if (f->kind == KINDV && f->fk == FKV) {                                                   \
        if (f->two_level && f->ragged == 1 && f->a.par >= 0)                                  \
            SMC_LAUNCH((k_propagate<KINDV, FKV, F_OPT, true, false, 1>), grid, dim3(SMC_BLOCK), st, P_LEAD, f->a);  \
        else if (f->two_level && f->ragged == 1)                                              \
            SMC_LAUNCH((k_propagate<KINDV, FKV, F_OPT, false, false, 1>), grid, dim3(SMC_BLOCK), st, P_LEAD, f->a); \
        else if (f->two_level && f->ragged == 2 && f->a.par >= 0)                             \
            SMC_LAUNCH((k_propagate<KINDV, FKV, F_OPT, true, false, 2>), grid, dim3(SMC_BLOCK), st, P_LEAD, f->a);  \
        else if (f->two_level && f->ragged == 2)                                              \
            SMC_LAUNCH((k_propagate<KINDV, FKV, F_OPT, false, false, 2>), grid, dim3(SMC_BLOCK), st, P_LEAD, f->a); \
        else if (f->two_level && f->a.par >= 0)                                               \
            SMC_LAUNCH((k_propagate<KINDV, FKV, F_OPT, true, false>), grid, dim3(SMC_BLOCK), st, P_LEAD, f->a);  \
        else if (f->two_level)                                                                \
            SMC_LAUNCH((k_propagate<KINDV, FKV, F_OPT, false, false>), grid, dim3(SMC_BLOCK), st, P_LEAD, f->a); \
        else if (f->a.par >= 0)                                                               \
            SMC_LAUNCH((k_propagate<KINDV, FKV, F_OPT, true>), grid, dim3(SMC_BLOCK), st, P_LEAD, f->a);  \
        else                                                                                  \
            SMC_LAUNCH((k_propagate<KINDV, FKV, F_OPT, false>), grid, dim3(SMC_BLOCK), st, P_LEAD, f->a); \
        return;                                                                               \
    }
    P_CASE(SMC_MODEL_LINGAUSS, SMC_FK_BOOTSTRAP)
    P_CASE(SMC_MODEL_LINGAUSS, SMC_FK_GUIDED)
    P_CASE(SMC_MODEL_STOCHVOL, SMC_FK_BOOTSTRAP)
    P_CASE(SMC_MODEL_STOCHVOL, SMC_FK_GUIDED)
    P_CASE(SMC_MODEL_STOCHVOL, SMC_FK_APF)              // (tail-free two-level launches only: create checks)
    P_CASE(SMC_MODEL_LINGAUSS, SMC_FK_APF)
    P_CASE(SMC_MODEL_STOCHVOL, SMC_FK_APF_BOOT)
    P_CASE(SMC_MODEL_LINGAUSS, SMC_FK_APF_BOOT)
    P_CASE(SMC_MODEL_GORDON, SMC_FK_BOOTSTRAP)
    P_CASE(SMC_MODEL_THETALOGISTIC, SMC_FK_BOOTSTRAP)
    P_CASE(SMC_MODEL_SVLEVERAGE, SMC_FK_BOOTSTRAP)
    P_CASE(SMC_MODEL_DISCRETECOX, SMC_FK_BOOTSTRAP)
#undef P_CASE
#undef P_LEAD
}

static void launch_onepass(const FArgs& a, const dim3 gw, hipStream_t st)
{
    switch (a.sp_tpw) {
    case 1: SMC_LAUNCH(k_f_spacing_onepass<1>, gw, dim3(SMC_BLOCK), st, a); break;
    case 2: SMC_LAUNCH(k_f_spacing_onepass<2>, gw, dim3(SMC_BLOCK), st, a); break;
    case 4: SMC_LAUNCH(k_f_spacing_onepass<4>, gw, dim3(SMC_BLOCK), st, a); break;
    default: SMC_LAUNCH(k_f_spacing_onepass<8>, gw, dim3(SMC_BLOCK), st, a); break;
    }
}

// one time step: [k_prepare, (spacings), k_ancestors] do nothing unless the step
// resamples (decided on the device by the previous k_propagate), then k_propagate
static void enqueue_step(smc_filter* f, int k_prof, i64 t, bool t_known = true)
{
    // the host knows the time index of every step it enqueues; inside a replayed graph
    // only its parity is static (graphs hold an even number of steps and start at even t)
    f->a.par = f->a.hist ? -1 : (int)(t & 1);
    f->a.tk = (t_known && !f->no_tk) ? t : -1;
    hipStream_t st = f->ctx->stream;
    const dim3 grid(f->a.ntiles, f->a.n_islands);
    if (k_prof >= 0) (void)hipEventRecord(f->ev[3 * k_prof], st);
    if (f->sqmc && !f->sq_flat) {
        // smc_filter_sqmc.h.  The step's normals come from a tape that is ONE step's buffer (zt_ts = 0),
        // written by k_sq_init / k_sq_permute just before; the thresholds are a function of n (f2_sq_T)
        FArgs& a = f->a;
        a.zt = f->sq_z;
        a.zt_ts = 0;
        a.sq_seed = f->sq_seed;
        a.sq_ctr = f->sq_ctr0;
        // (inside a captured graph only the parity of t is static: replays start at t >= 2, see smc_filter_step)
        if (t_known && t == 0) {
            SMC_LAUNCH(k_sq_init, dim3((unsigned)((a.N + SMC_BLOCK - 1) / SMC_BLOCK), a.n_islands), dim3(SMC_BLOCK), st,
                       f->a, f->sq_z, f->sq_seed, f->sq_ctr0);
        } else {
            // bootstrap filters whose weight depends on the new particle only: the sorted weights are recomputed
            // from the sorted keys (k_sq_permute<.., true>); SMC_PATH_SQ_GATHER (A/B) and the others gather them
            // (one island, N beyond the one-workgroup sort: the workspace then holds the key images of that island)
            const bool recompute = f->fk == SMC_FK_BOOTSTRAP && f->kind != SMC_MODEL_SVLEVERAGE && !f->sq_gather &&
                                   a.n_islands == 1 && a.N > 2048;
            const u64 *perm = f->sq_perm, *skeys = nullptr;
            for (int i = 0; i < a.n_islands; ++i) {        // h_order = argsort(X_{t-1}) (hilbert.py:52-54, d = 1)
                u64 *k0 = nullptr, *v0 = nullptr;
                (void)smc_rs_sort_ws(f->ctx, f_X(a, t - 1) + (i64)i * a.N, nullptr, a.N, 0, f->sq_ws, recompute ? &k0 : nullptr, &v0, f->sq_plan_zero);
                f->sq_plan_zero = true;               // (every sort leaves its plan words zeroed for the next)
                if (a.n_islands == 1) { perm = v0; skeys = k0; }
                else (void)hipMemcpyAsync(f->sq_perm + (i64)i * a.N, v0, (size_t)a.N * 8, hipMemcpyDeviceToDevice, st);
            }
#define SQ_CASE(KINDV)                                                                                           \
    if (f->kind == KINDV) {                                                                                      \
        if (recompute) SMC_LAUNCH((k_sq_permute<KINDV, true>), grid, dim3(SMC_BLOCK), st, f->a, perm, skeys, f->sq_z, \
                                  f->sq_seed, f->sq_ctr0);                                                       \
        else SMC_LAUNCH((k_sq_permute<KINDV, false>), grid, dim3(SMC_BLOCK), st, f->a, perm, skeys, f->sq_z,      \
                        f->sq_seed, f->sq_ctr0);                                                                 \
    }
            SQ_CASE(SMC_MODEL_LINGAUSS) SQ_CASE(SMC_MODEL_STOCHVOL) SQ_CASE(SMC_MODEL_GORDON)
            SQ_CASE(SMC_MODEL_THETALOGISTIC) SQ_CASE(SMC_MODEL_SVLEVERAGE) SQ_CASE(SMC_MODEL_DISCRETECOX)
#undef SQ_CASE
            launch_reduce2(f, st);
            f->a.sq_perm = perm;                     // (A = h_order[sorted position]: composed where the ancestors are stored)
            SMC_LAUNCH((k_ancestors2<true, true, true, false, true>), grid, dim3(SMC_BLOCK), st, f->a);
        }
        if (k_prof >= 0 && (k_prof % 3)) (void)hipEventRecord(f->ev[3 * k_prof + 1], st);
        launch_propagate(f);
        if (k_prof >= 0) (void)hipEventRecord(f->ev[3 * k_prof + 2], st);
        if (a.mom) {
            SMC_LAUNCH(k_flush2, dim3(a.n_islands), dim3(SMC_BLOCK), st, f->a);
            SMC_LAUNCH(k_f_moments_partials, dim3(a.nmb, a.n_islands), dim3(SMC_BLOCK), st, f->a);
            SMC_LAUNCH(k_f_moments_final, dim3(a.n_islands), dim3(SMC_BLOCK), st, f->a);
        }
        return;
    }
    if (f->strict) {
        // decision + normalisation of step t-1 (two-level: by k_strict_classify's workgroups themselves, or k_reduce2
        // beyond 1024 tiles / for multinomial draws / the literal walk; flat: k_propagate's tail did it), the
        // sequential CDF of W_{t-1}, the searches
        // (multinomial, Philox draws: uniform_spacings in one pass, the island's reduction as its workgroup 0 where the
        //  grid fits -- as on the default path; the literal walk and one-tile filters: one workgroup per island)
        const bool sp1 = f->a.scheme == SMC_MULTINOMIAL && !f->a.ut && f->a.sp_tpw;
        const bool merge = sp1 && f->sp_merge && t_known;
        f->a.sp_epoch = merge ? ++f->sp_epoch : 0ull;
        if (f->two_level && (f->two_level_mid || f->strict_literal) && !merge)
            launch_reduce2(f, st);
        if (sp1) {
            const dim3 gw(f->a.sp_nwg + (merge ? 1 : 0), f->a.n_islands);
            switch (f->a.sp_tpw) {
            case 1: SMC_LAUNCH(k_f_spacing_onepass<1>, gw, dim3(SMC_BLOCK), st, f->a); break;
            case 2: SMC_LAUNCH(k_f_spacing_onepass<2>, gw, dim3(SMC_BLOCK), st, f->a); break;
            case 4: SMC_LAUNCH(k_f_spacing_onepass<4>, gw, dim3(SMC_BLOCK), st, f->a); break;
            default: SMC_LAUNCH(k_f_spacing_onepass<8>, gw, dim3(SMC_BLOCK), st, f->a); break;
            }
        } else if (f->a.scheme == SMC_MULTINOMIAL && !f->a.ut) {
            for (int i = 0; i < f->a.n_islands; ++i)
                SMC_LAUNCH(k_f_spacings_step, dim3(1), dim3(SMC_BLOCK), st, f->a, i, f->a.su + (size_t)i * f->a.N);
        }
        const unsigned nb = (unsigned)((f->a.N + 1023) / 1024);
        const dim3 gt(nb, f->a.n_islands);
        // (strict_ws: (n_islands, N) W | (n_islands, N) S | the scratch of smc_seqx.h | (n_islands, ntiles) tile sums)
        double* S = f->strict_ws + (size_t)f->a.n_islands * f->a.N;
        void* sqx_scr = (void*)(S + (size_t)f->a.n_islands * f->a.N);
        SqxArgs q = sqx_carve(sqx_scr, f->a.N, f->a.n_islands);
#ifdef SMC_TRACE
        q.trace = f->a.trace + (size_t)f->a.n_islands * (f->a.nparts + f->a.ntiles) * 8;
#endif
        const SeqGate gate{f->a.info, INFO_STRIDE, f->a.T, nullptr};
        if (f->strict_literal) {
            // the definition: W written out, ONE lane adding it up in place (44 ms at N = 2^20), a search per offspring
            SMC_LAUNCH(k_strict_W, gt, dim3(SMC_BLOCK), st, f->a, f->strict_ws, (double*)nullptr);
            SMC_LAUNCH(k_strict_cdf, dim3(1, f->a.n_islands), dim3(64), st, f->a, f->strict_ws);
            SMC_LAUNCH(k_strict_search_S, dim3((unsigned)((f->a.N / 2 + SMC_BLOCK) / SMC_BLOCK), f->a.n_islands), dim3(SMC_BLOCK), st,
                       f->a, (const double*)f->strict_ws, (const double*)f->a.su);
        } else if (f->two_level) {
            // the same doubles in two launches, S never written (smc_filter_strict.h)
            if (f->two_level_mid) SMC_LAUNCH(k_strict_classify<true>, gt, dim3(SMC_BLOCK), st, f->a, q);
            else SMC_LAUNCH(k_strict_classify<false>, gt, dim3(SMC_BLOCK), st, f->a, q);
            SMC_LAUNCH(k_strict_search, gt, dim3(SMC_BLOCK), st, f->a, q);
        } else {
            // one tile (or a flat test path): W materialised, the two launches of smc_seqx.h on the array, S written
            size_t used = 0;
            (void)sqx_carve(sqx_scr, f->a.N, f->a.n_islands, &used);
            double* tsum = (double*)((char*)sqx_scr + used);
            SMC_LAUNCH(k_strict_W, gt, dim3(SMC_BLOCK), st, f->a, f->strict_ws, tsum);
            SMC_LAUNCH(k_sqx_classify, gt, dim3(SMC_BLOCK), st, (const double*)f->strict_ws, (const double*)tsum, q, gate);
            SMC_LAUNCH(k_sqx_fill, gt, dim3(SMC_BLOCK), st, q, S, gate);
            SMC_LAUNCH(k_strict_search_S, dim3((unsigned)((f->a.N / 2 + SMC_BLOCK) / SMC_BLOCK), f->a.n_islands), dim3(SMC_BLOCK), st,
                       f->a, (const double*)S, (const double*)f->a.su);
        }
        if (k_prof >= 0 && (k_prof % 3)) (void)hipEventRecord(f->ev[3 * k_prof + 1], st);
        launch_propagate(f);
        if (k_prof >= 0) (void)hipEventRecord(f->ev[3 * k_prof + 2], st);
        if (f->a.mom) {
            if (f->two_level) SMC_LAUNCH(k_flush2, dim3(f->a.n_islands), dim3(SMC_BLOCK), st, f->a);
            SMC_LAUNCH(k_f_moments_partials, dim3(f->a.nmb, f->a.n_islands), dim3(SMC_BLOCK), st, f->a);
            SMC_LAUNCH(k_f_moments_final, dim3(f->a.n_islands), dim3(SMC_BLOCK), st, f->a);
        }
        return;
    }
    if (f->two_level) {
        if (f->two_level_mid && f->a.scheme == SMC_MULTINOMIAL) {
            // (the island's reduction: a launch of its own, or workgroup 0 of the one-pass spacings kernel)
            const bool merge = f->sp_merge && !f->a.ut && f->a.sp_tpw && t_known;
            f->a.sp_epoch = merge ? ++f->sp_epoch : 0ull;
            if (!merge) launch_reduce2(f, st);
            if (!f->a.ut) {
                // production mode: uniform_spacings in two passes over the same draws (tile sums, their
                // prefixes, the uniforms written once); k_ancestors2 finds its window through the prefixes
                const dim3 g1(f->a.ntiles1, f->a.n_islands);
                if (f->a.sp_tpw) {                 // one pass (decoupled look-back): every workgroup resident
                    const dim3 gw(f->a.sp_nwg + (merge ? 1 : 0), f->a.n_islands);
                    switch (f->a.sp_tpw) {
                    case 1: SMC_LAUNCH(k_f_spacing_onepass<1>, gw, dim3(SMC_BLOCK), st, f->a); break;
                    case 2: SMC_LAUNCH(k_f_spacing_onepass<2>, gw, dim3(SMC_BLOCK), st, f->a); break;
                    case 4: SMC_LAUNCH(k_f_spacing_onepass<4>, gw, dim3(SMC_BLOCK), st, f->a); break;
                    default: SMC_LAUNCH(k_f_spacing_onepass<8>, gw, dim3(SMC_BLOCK), st, f->a); break;
                    }
                } else {
                    SMC_LAUNCH(k_f_spacing_sums, g1, dim3(SMC_BLOCK), st, f->a);
                    SMC_LAUNCH(k_f_spacing_scan, dim3(f->a.n_islands), dim3(SMC_BLOCK), st, f->a);
                    SMC_LAUNCH(k_f_spacing_write, g1, dim3(SMC_BLOCK), st, f->a);
                }
                SMC_LAUNCH((k_ancestors2<true, true, true, true>), grid, dim3(SMC_BLOCK), st, f->a);
            } else {
                SMC_LAUNCH((k_ancestors2<true, true>), grid, dim3(SMC_BLOCK), st, f->a);
            }
        } else {
            // closed-form counts: one instantiation per scheme (see k_ancestors2's SCH)
            if (f->two_level_mid) launch_reduce2(f, st);
#ifdef SMC_NO_SCHEME_SPLIT                         /* (A/B builds: tools/build_ablations.sh) */
#define A2_CASE(MIDV, POW2V, SCHV) SMC_LAUNCH((k_ancestors2<MIDV, false, POW2V>), grid, dim3(SMC_BLOCK), st, f->a)
#else
#define A2_CASE(MIDV, POW2V, SCHV) SMC_LAUNCH((k_ancestors2<MIDV, false, POW2V, false, false, SCHV>), grid, dim3(SMC_BLOCK), st, f->a)
#endif
            const bool sys = f->a.scheme == SMC_SYSTEMATIC, p2 = f->a.log2N >= 0;
            if (f->two_level_mid) {
                if (p2) { if (sys) A2_CASE(true, true, SMC_SYSTEMATIC_); else A2_CASE(true, true, SMC_STRATIFIED_); }
                else { if (sys) A2_CASE(true, false, SMC_SYSTEMATIC_); else A2_CASE(true, false, SMC_STRATIFIED_); }
            } else if (!p2 && f->wide_tpw) {
                // resident grid, any N, systematic: two tiles per workgroup over the runs of tiles f_tile_xcd gives the XCDs
                const int per_run = (f->a.ntiles + 7) / 8;
                const dim3 gridw(f->a.xcd_chunks ? 8 * ((per_run + 1) / 2) : (f->a.ntiles + 1) / 2, f->a.n_islands);
                SMC_LAUNCH((k_ancestors2w<2, SMC_SYSTEMATIC_, false>), gridw, dim3(SMC_BLOCK * 2), st, f->a.info2, f->a.pm, f->a.ps,
                           f->a.pss, f->a.cq, f->a.N, f->a.ntiles | (f->a.xcd_chunks ? 1 << 30 : 0), f->a);
            } else if (p2 && f->wide_tpw) {
                // resident grid, N = 2^k: TPW tiles per workgroup, the partials reduced by its first 4 waves only
                const dim3 gridw(f->a.ntiles / f->wide_tpw, f->a.n_islands);
                // (leading arguments: the fields the kernel's first loads are addressed with, preloaded into SGPRs)
#define W_LEAD f->a.info2, f->a.pm, f->a.ps, f->a.pss, f->a.cq, f->a.N, f->a.ntiles | (f->a.xcd_chunks ? 1 << 30 : 0)
                if (sys) SMC_LAUNCH((k_ancestors2w<2, SMC_SYSTEMATIC_>), gridw, dim3(SMC_BLOCK * 2), st, W_LEAD, f->a);
                else SMC_LAUNCH((k_ancestors2w<2, SMC_STRATIFIED_>), gridw, dim3(SMC_BLOCK * 2), st, W_LEAD, f->a);
            } else {
                if (p2) { if (sys) A2_CASE(false, true, SMC_SYSTEMATIC_); else A2_CASE(false, true, SMC_STRATIFIED_); }
                else { if (sys) A2_CASE(false, false, SMC_SYSTEMATIC_); else A2_CASE(false, false, SMC_STRATIFIED_); }
            }
#undef A2_CASE
        }
        if (k_prof >= 0 && (k_prof % 3)) (void)hipEventRecord(f->ev[3 * k_prof + 1], st);
        launch_propagate(f);
        if (k_prof >= 0) (void)hipEventRecord(f->ev[3 * k_prof + 2], st);
        if (f->a.mom) {      // device-side Moments: the row of the step just done (K, 1/s) first
            SMC_LAUNCH(k_flush2, dim3(f->a.n_islands), dim3(SMC_BLOCK), st, f->a);
            SMC_LAUNCH(k_f_moments_partials, dim3(f->a.nmb, f->a.n_islands), dim3(SMC_BLOCK), st, f->a);
            SMC_LAUNCH(k_f_moments_final, dim3(f->a.n_islands), dim3(SMC_BLOCK), st, f->a);
        }
        return;
    }
    const bool fused = f->fused;
    if (f->sq_flat) {
        // SQMC of a multivariate filter, or of a univariate one of fewer than two tiles (smc_filter_sqmc.h):
        // Hilbert order (d = 1: the radix argsort), the step's points, the tapes; the flat step then runs as it
        // is -- its multinomial search over the sorted uniforms in a.su, its propagate kernel fed from the tape
        // of ndtri values
        FArgs& a = f->a;
        const int d = a.dx;
        const unsigned nb = (unsigned)((a.N + SMC_BLOCK - 1) / SMC_BLOCK);
        double* U = (double*)f->sq_ws;
        double* lws = U + (size_t)a.N * (d + 1);
        a.zt = f->sq_z;
        a.zt_ts = 0;
        for (int i = 0; i < a.n_islands; ++i) {
            const u64 ctr = f->sq_ctr0 + (u64)t + ((u64)(u32)(a.island_offset + i) << 32);
            if (t == 0) {
                (void)smc_sobol_points(f->ctx, f->sq_seed, a.N, d, ctr, 0, U);
                SMC_LAUNCH(k_sqmv_tapes, dim3(nb), dim3(SMC_BLOCK), st, f->a, i, (const i64*)nullptr, (const double*)U, d,
                           lws, f->sq_z);
            } else {
                i64* perm = (i64*)f->sq_perm + (size_t)i * a.N;
                if (d == 1) {                      // hilbert.py:52-54: argsort
                    u64* v0 = nullptr;
                    (void)smc_rs_sort_ws(f->ctx, f_X(a, t - 1) + (size_t)i * a.N, nullptr, a.N, 0, (void*)(lws + a.N), nullptr, &v0, f->sq_plan_zero);
                    f->sq_plan_zero = true;
                    (void)hipMemcpyAsync(perm, v0, (size_t)a.N * 8, hipMemcpyDeviceToDevice, st);
                } else {
                    (void)smc_hilbert_sort(f->ctx, f_X(a, t - 1) + (size_t)i * a.N * d, a.N, d, (int64_t*)perm, nullptr);
                }
                (void)smc_sobol_points(f->ctx, f->sq_seed, a.N, d + 1, ctr, 1, U);
                SMC_LAUNCH(k_sqmv_tapes, dim3(nb), dim3(SMC_BLOCK), st, f->a, i, (const i64*)perm, (const double*)U, d + 1,
                           lws, f->sq_z);
                (void)hipMemcpyAsync(f_lw(a, t - 1) + (size_t)i * a.N, lws, (size_t)a.N * 8, hipMemcpyDeviceToDevice, st);
            }
        }
    }
    if (f->kind == SMC_MODEL_MVLINGAUSS && f->fk == SMC_FK_APF) {
        // auxiliary weights of step t (core.py:307-313) before its resampling: smc_filter_mv.h
        const dim3 gp(f->a.nparts, f->a.n_islands);
        if (f->a.dp == 16 && f->a.dx == 16) SMC_LAUNCH((k_mv_aux<16, true>), gp, dim3(SMC_BLOCK), st, f->a, f->a.mvc);
        else if (f->a.dp == 16) SMC_LAUNCH((k_mv_aux<16, false>), gp, dim3(SMC_BLOCK), st, f->a, f->a.mvc);
        else if (f->a.dx == 32) SMC_LAUNCH((k_mv_aux<32, true>), gp, dim3(SMC_BLOCK), st, f->a, f->a.mvc);
        else SMC_LAUNCH((k_mv_aux<32, false>), gp, dim3(SMC_BLOCK), st, f->a, f->a.mvc);
        SMC_LAUNCH(k_mv_aux_restate, dim3(f->a.n_islands), dim3(SMC_BLOCK), st, f->a);
    }
    if (!fused) SMC_LAUNCH(k_prepare, grid, dim3(SMC_BLOCK), st, f->a);
    if (f->a.scheme == SMC_MULTINOMIAL && !f->a.ut && !f->sqmc) {
        const dim3 g1(f->a.ntiles1, f->a.n_islands);
        SMC_LAUNCH(k_f_spacing_sums, g1, dim3(SMC_BLOCK), st, f->a);
        SMC_LAUNCH(k_f_spacing_scan, dim3(f->a.n_islands), dim3(SMC_BLOCK), st, f->a);
        SMC_LAUNCH(k_f_spacing_write, g1, dim3(SMC_BLOCK), st, f->a);
    }
    if (fused && f->a.par >= 0) SMC_LAUNCH((k_ancestors<true, true>), grid, dim3(SMC_BLOCK), st, f->a);
    else if (fused) SMC_LAUNCH((k_ancestors<true, false>), grid, dim3(SMC_BLOCK), st, f->a);
    else SMC_LAUNCH((k_ancestors<false, false>), grid, dim3(SMC_BLOCK), st, f->a);
    if (f->sq_flat && t > 0)                       // A <- h_order[A] (core.py:344)
        for (int i = 0; i < f->a.n_islands; ++i)
            SMC_LAUNCH(k_sqmv_compose, dim3((unsigned)((f->a.N + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK), st, f->a, i,
                       (const i64*)f->sq_perm + (size_t)i * f->a.N);
    // samples come in three kinds (k mod 3): 0 times the whole step, 1 the interval [start,
    // resampling kernels done], 2 the interval [resampling kernels done, end].  Every event
    // interval carries the same ~4 us of marker processing on MI355X (tools/micro/events.hip),
    // which cancels in propagate = whole - first part, resampling = whole - second part: both
    // kernels are MEASURED (smc_filter_kernel_ms), neither is derived from the step time.
    if (k_prof >= 0 && (k_prof % 3)) (void)hipEventRecord(f->ev[3 * k_prof + 1], st);
    launch_propagate(f);
    if (k_prof >= 0) (void)hipEventRecord(f->ev[3 * k_prof + 2], st);
    if (f->a.mom) {
        SMC_LAUNCH(k_f_moments_partials, dim3(f->a.nmb, f->a.n_islands), dim3(SMC_BLOCK), st, f->a);
        SMC_LAUNCH(k_f_moments_final, dim3(f->a.n_islands), dim3(SMC_BLOCK), st, f->a);
    }
}

extern "C" {

static int filter_fetch(smc_filter* f, int field, i64 s, int island, void* out_host);
static void flush_rows(smc_filter* f);

int smc_filter_create(smc_ctx* ctx, const smc_model* model, const smc_filter_opts* o,
                      const double* y_host, smc_filter** out)
{
    SMC_REQUIRE(ctx && model && o && y_host && out, "null argument");
    SMC_REQUIRE(o->N > 0 && o->T > 0 && o->n_islands > 0, "N, T, n_islands must be positive");
    if (o->scheme != SMC_MULTINOMIAL && o->scheme != SMC_STRATIFIED &&
        o->scheme != SMC_SYSTEMATIC) {
        smc_set_error("%d is not a valid resampling scheme", o->scheme);
        return SMC_ERR_SCHEME;
    }
    const bool mv = model->kind == SMC_MODEL_MVLINGAUSS;
    SMC_REQUIRE(model->kind == SMC_MODEL_LINGAUSS || model->kind == SMC_MODEL_STOCHVOL || mv ||
                    model->kind == SMC_MODEL_GORDON || model->kind == SMC_MODEL_THETALOGISTIC ||
                    model->kind == SMC_MODEL_SVLEVERAGE || model->kind == SMC_MODEL_DISCRETECOX,
                "fused filter: unknown model kind");
    SMC_REQUIRE(model->fk == SMC_FK_BOOTSTRAP ||
                    (model->fk == SMC_FK_GUIDED &&
                     (model->kind == SMC_MODEL_LINGAUSS || model->kind == SMC_MODEL_STOCHVOL || mv)) ||
                    (model->fk == SMC_FK_APF && (model->kind == SMC_MODEL_STOCHVOL || model->kind == SMC_MODEL_LINGAUSS || mv)) ||
                    (model->fk == SMC_FK_APF_BOOT && (model->kind == SMC_MODEL_STOCHVOL || model->kind == SMC_MODEL_LINGAUSS)),
                "guided filter: LINGAUSS, STOCHVOL, MVLINGAUSS; auxiliary filter: STOCHVOL, LINGAUSS, MVLINGAUSS; "
                "auxiliary bootstrap filter: STOCHVOL, LINGAUSS");
    {
        const bool big = o->N > F_TILE && o->N <= ((int64_t)1 << 30);
        if (!mv && f_is_apf(model->fk) && ((o->N > F_TILE && !big) || (!big && (o->moments || o->keep_history >= 2)))) {
            smc_set_error("the auxiliary particle filter is fused for N <= 1024 (the one-launch filter: no moments, no "
                          "rolling window) and for 1024 < N <= 2^30 (the two-level step)");
            return SMC_ERR_INVALID;
        }
    }
    SMC_REQUIRE(model->kind != SMC_MODEL_GORDON || model->aux_host,
                "GORDON needs aux_host (d*cos(e*(t-1)) per step)");
    SMC_REQUIRE(model->kind != SMC_MODEL_DISCRETECOX || model->aux_host,
                "DISCRETECOX needs aux_host (gammaln(y_t + 1) per step)");
    SMC_REQUIRE(mv || model->params_host, "params_host is required");
    int dxm = 1, dym = 1, dpm = 1;
    std::vector<double> mvc_host;
    bool mv_diag = false;
    if (mv) {
        dxm = model->dx; dym = model->dy;
        SMC_REQUIRE(dxm >= 1 && dxm <= 32 && dym >= 1 && dym <= dxm,
                    "MVLINGAUSS needs 1 <= dy <= dx <= 32");
        SMC_REQUIRE(model->F_host && model->G_host && model->covX_host && model->covY_host &&
                        model->mu0_host && model->cov0_host, "MVLINGAUSS matrices are required");
        dpm = dxm <= 16 ? 16 : 32;              // padded to whole 16x16 MFMA blocks
        if (!mv_build_constants(model, model->fk, dpm, o->T, y_host, mvc_host, &mv_diag)) {
            // same failure as MvNormal.__init__ (distributions.py:935-940)
            smc_set_error("MvNormal: argument cov must be a (d, d) pos. definite matrix");
            return SMC_ERR_INVALID;
        }
    }
    SMC_REQUIRE(o->N < ((i64)1 << 32), "N must be below 2^32");
    SMC_HIP_CHECK(hipSetDevice(ctx->device));

    smc_filter* f = new smc_filter();
    f->ctx = ctx;
    f->kind = model->kind;
    f->fk = model->fk;
    f->t_host = 0;
    f->use_graph = o->use_graph != 0;
    f->gexec[0] = f->gexec[1] = f->gexec[2] = nullptr;
    f->graph_failed = false;
    f->prof = false;
    f->prof_n = 0;
    f->perm_t = -1;
    f->lwth = f->th = f->th_ess = nullptr;
    f->th_buf = nullptr;
    f->th_ess_min = 0.0;
    f->th_comm = nullptr;
    f->th_n = 0;
    f->th_send = f->th_recv = nullptr;
    f->mv_collapsed = mv && model->fk == SMC_FK_GUIDED && (o->flags & SMC_FLAG_COLLAPSED_PROPOSAL);
    f->strict = (o->flags & SMC_FLAG_STRICT_ANCESTORS) != 0;
    f->strict_literal = (o->flags & SMC_PATH_STRICT_LITERAL) != 0;
    f->no_small = (o->flags & SMC_PATH_NO_SMALL) != 0;
    f->strict_ws = nullptr;
    f->sqmc = (o->flags & SMC_FLAG_SQMC) != 0;
    f->sq_flat = false;
    f->sq_gather = (o->flags & SMC_PATH_SQ_GATHER) != 0;
    f->sq_seed = o->seed;
    f->sq_ctr0 = 1;
    f->sq_z = nullptr;
    f->sq_perm = nullptr;
    f->sq_ws = nullptr;
    if (f->sqmc) {
        // core.py:339-349 as a fused loop.  Univariate Normal kernels (Gamma = ppf), N = 2^k >= 2 tiles: the
        // two-level step (the sorted Sobol' order in closed form).  MVLINGAUSS (2 <= d <= 9: d + 1 Sobol'
        // coordinates) and univariate filters of N = 2^k < 2 tiles: the flat step behind the Hilbert sort / the
        // argsort, N >= 32, no history slots (the sorted weights take the slot).
        bool pow2 = false;
        for (int k = 5; k <= 30; ++k) pow2 = pow2 || (((i64)1 << k) == o->N);
        f->sq_flat = mv || o->N < 2 * F_TILE;
        const bool fk_ok = model->fk == SMC_FK_BOOTSTRAP || model->fk == SMC_FK_GUIDED;
        const bool mv_ok = !f->sq_flat || ((!mv || (model->dx >= 2 && model->dx <= 9)) && !o->keep_history && !o->moments &&
                                           !(o->flags & SMC_FLAG_COLLAPSED_PROPOSAL));
        if (!fk_ok || !pow2 || !mv_ok || f->strict || o->rng_mode != SMC_RNG_PHILOX || (f->sq_flat && o->use_graph) ||
            (o->flags & (SMC_PATH_FLAT_CDF | SMC_PATH_FORCE_UNFUSED))) {
            smc_set_error("SMC_FLAG_SQMC: Bootstrap / Guided filters, N = 2^k with 5 <= k <= 30, Philox mode; MVLINGAUSS "
                          "(2 <= d <= 9) and univariate models with k <= 10: eager launches, no history slots, no moments");
            delete f;
            return SMC_ERR_INVALID;
        }
    }
    const int scheme = f->sqmc ? (int)SMC_MULTINOMIAL : (int)o->scheme;     // (sorted uniforms from a tape)
    if (f->strict && (mv || f_is_apf(model->fk) || o->N >= ((i64)1 << 32))) {
        smc_set_error("SMC_FLAG_STRICT_ANCESTORS: univariate Bootstrap / Guided filters");
        delete f;
        return SMC_ERR_INVALID;
    }
    FArgs& a = f->a;
    memset(&a, 0, sizeof a);
    a.N = o->N;
    a.T = o->T;
    a.n_islands = o->n_islands;
    a.ntiles = (int)((o->N + F_TILE - 1) / F_TILE);
    a.ntiles1 = (int)((o->N + 1 + F_TILE - 1) / F_TILE);
    a.scheme = scheme;
    a.rng_mode = o->rng_mode;
    a.island_offset = o->island_offset;
    a.ess_thresh = f->sqmc ? INFINITY : (double)o->N * o->ESSrmin;     // (SQMC always resamples, core.py:340)
    a.seed = o->seed;
    a.log2N = -1;
    for (int k = 0; k < 62; ++k)
        if (((i64)1 << k) == o->N) a.log2N = k;
    {
        int lg = 0;
        while (((i64)1 << lg) < o->N + 2) ++lg;
        a.spacing_scale = ldexp(1.0, 57 - lg < 21 ? 57 - lg : 21);        // (see smc_ops.hip spacing_scale)
    }
    const size_t M = (size_t)o->n_islands, N = (size_t)o->N, T = (size_t)o->T;
    const bool need_su = (scheme == SMC_MULTINOMIAL);
    // carve one slab
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o0 = off; off = smc_align_up(off + bytes, 256); return o0; };
    a.dx = dxm; a.dy = dym; a.dp = dpm;
    SMC_REQUIRE(o->keep_history >= 0, "keep_history must be 0, 1 or a window length >= 2");
    a.hist = o->keep_history;                      // 0 / 1 (whole history) / k >= 2 (rolling window)
    if (a.hist >= 2 && (size_t)a.hist >= T) a.hist = 1;      // a window as long as the run: all of it
    const size_t nslots = a.hist == 1 ? T : (a.hist >= 2 ? (size_t)a.hist : 2);
    a.xslot = (i64)(M * N * dxm);
    a.lslot = (i64)(M * N);
    // (+ one tile of padding each: the ragged last tile's loads are unconditional, k_propagate<RAGGED = 1>)
    const size_t oX0 = carve(nslots * M * N * dxm * 8 + F_TILE * dxm * 8);
    const size_t oL0 = carve(nslots * M * N * 8 + F_TILE * 8);
    const size_t nA = (a.hist ? nslots : 1) * M * N + F_TILE;
    const size_t oA = carve(nA * 4);
    // (published tile totals pay off only while every workgroup of the launch is resident)
    f->fused = (i64)a.ntiles * (i64)M <= F_DIRECT_PREFIX_MAX;
    if (o->flags & SMC_PATH_FORCE_UNFUSED) f->fused = false;  // tests: the k_prepare path at any size
    const size_t oQ = carve(M * a.ntiles * 8);
    const size_t oQpre = carve(M * a.ntiles * 8);
    // MV: a workgroup stages the step's matrices in LDS once and then walks
    // mv_chunks chunks of 256 particles (2 workgroups per CU when N allows)
    a.mv_chunks = 1;
    a.mv_diag = (mv && mv_diag && !(o->flags & SMC_PATH_MV_DENSE)) ? 1 : 0;
    if (mv) {
        // 8 by default, halved until the grid has at least 512 workgroups (element-wise form: 2 workgroups per CU) or
        // 1024 (dense form: 3 per CU fit); SMC_PATH_MV_CHUNKS(1|2|4|8) (tests) is taken as given, so that the
        // multi-chunk prefetch loop is audited at small N too
        const int forced = (o->flags >> 20) & 15;
        if (forced == 1 || forced == 2 || forced == 4 || forced == 8) a.mv_chunks = forced;
        else {
            const i64 min_grid = a.mv_diag ? 512 : 1024;
            a.mv_chunks = 8;
            while (a.mv_chunks > 1 && (i64)N / (SMC_BLOCK * a.mv_chunks) < min_grid) a.mv_chunks >>= 1;
        }
    }
    const i64 per_wg = mv ? (i64)SMC_BLOCK * a.mv_chunks : (i64)SMC_BLOCK * F_OPT;
    a.nparts = (int)((o->N + per_wg - 1) / per_wg);
    const size_t oPm = carve(M * a.nparts * 8), oPs = carve(M * a.nparts * 8),
                 oPss = carve(M * a.nparts * 8);
    const size_t oSum = carve(M * (T + 1) * SUMM_STRIDE * 8);
    const size_t oPar = carve(M * PARAM_STRIDE * 8);
    const size_t oY = carve(T * dym * 8);
    const size_t oAux = carve(T * 8);
    const size_t oMvc = carve(mvc_host.size() * 8 + 8);
    const size_t oCtl = carve(M * 2 * F_CNT_WORDS * sizeof(unsigned));
    const size_t oSpart = carve(M * 96 * 8);
    const size_t oInfo = carve(M * INFO_STRIDE * 8);
    const size_t oInfo2 = carve(M * INFO_STRIDE * 8);
    // two-level CDF: closed-form offspring counts (N = 2^k, systematic / stratified), at most
    // 1024 tiles per island (4 partials per thread), at least 2 (below, the one-workgroup filter)
    // (multinomial: the counts are searches over the sorted uniforms -- the tape's, or the exponential
    //  spacings drawn between k_reduce2, which decides the step, and k_ancestors2)
    // (any N >= 2 tiles: N = 2^k counts in closed form with integers, other N with the general counts)
    f->two_level = !mv && o->N <= ((int64_t)1 << 30) && a.ntiles >= 2 &&
                   !(o->flags & (SMC_PATH_FLAT_CDF | SMC_PATH_FORCE_UNFUSED));
    // every workgroup reduces the partials itself while the launch is resident and an island has
    // at most 1024 tiles (4 per thread); otherwise one workgroup per island does it first
    const bool apf2 = !mv && f_is_apf(model->fk) && o->N > F_TILE;      // APF on the two-level step: k_reduce2
    if (apf2 && !f->two_level) {                                      // forms its two reductions
        smc_set_error("the auxiliary particle filter beyond N = 1024 runs on the two-level step only");
        delete f;
        return SMC_ERR_INVALID;
    }
    f->two_level_mid = f->two_level && (!f->fused || a.ntiles > 1024 || (o->flags & SMC_PATH_TWO_LEVEL_MID) ||
                                        scheme == SMC_MULTINOMIAL || apf2);
    // resident grids of N = 2^k: 2 tiles per workgroup (smc_filter_wide.h; C2, same box: 17.6 us per step, 4 tiles 18.3,
    // one tile -- k_ancestors2 -- 18.1: profiles/r12d); SMC_PATH_NO_WIDE keeps the one-tile kernel testable at these sizes
    f->wide_tpw = 0;
    f->reduce_narrow = (o->flags & SMC_PATH_NO_WIDE) != 0;
    if (f->two_level && !f->two_level_mid && a.log2N >= 0 && !(o->flags & SMC_PATH_NO_WIDE) && !f->strict && !f->sqmc) {
        f->wide_tpw = (a.ntiles % 2) == 0 ? 2 : 0;
    }
    // ... and of any other N under the systematic scheme (k_ancestors2w<2, .., POW2 = false>: the general counts, any
    // number of tiles -- the last workgroup of a run may hold one tile)
    if (f->two_level && !f->two_level_mid && a.log2N < 0 && scheme == SMC_SYSTEMATIC && a.ntiles >= 2 &&
        !(o->flags & SMC_PATH_NO_WIDE) && !f->strict && !f->sqmc)
        f->wide_tpw = 2;
    // consecutive tiles on one XCD (f_tile_xcd): any number of tiles with one tile per resampling workgroup; with
    // k_ancestors2w whole multiples of 8 x (its tiles per workgroup) -- else the wide kernel keeps its own XCD-strided map
    a.xcd_chunks = (f->two_level && !mv && !f->strict && !f->sqmc && !(o->flags & SMC_PATH_NO_XCD_CHUNKS) && a.ntiles >= 16 &&
                    (!f->wide_tpw || a.log2N < 0 || a.ntiles % (8 * f->wide_tpw) == 0)) ? 1 : 0;
    // (measured and kept out, round 4: the reduction MERGED into the resampling launch on grids beyond 2048 workgroups --
    //  (a) every workgroup of k_ancestors2w reducing: C5 99.2 us per step (2 tiles per workgroup) / 117.2 (4) against 92.9
    //  behind k_reduce2 (r12h); (b) workgroup 0 of k_ancestors2 reducing, the others waiting for its word with their
    //  loads in flight and reading (G_b, Q_b) past their L2: C3 60.7 against 57.6 us, C5 159 against 99 -- a dependent
    //  global round trip in every workgroup costs more than the launch it saves (r12j))
    // (SQMC: k_ancestors2 counts in SORTED positions; a heavy parent's blocks would be filled with that index)
    const bool heavy_list = !mv && !f->strict && !f->sqmc && !(o->flags & SMC_PATH_NO_HEAVY);
    // (history slots are written step by step: the lanes beyond N of a slot would read indices nobody
    //  initialised -- every access tests its index there as well)
    a.kform = f->two_level ? 1 : 0;
    a.strict_e = (f->strict && f->two_level && !f->strict_literal) ? 1 : 0;
    f->ragged = (f->two_level && (o->N % F_TILE) != 0) ? (((o->N & 1) || a.hist) ? 2 : 1) : 0;
    a.ncq = (i64)a.ntiles * F_TILE;
    const size_t oCq = carve(f->two_level ? M * (size_t)a.ncq * 8 : 8);
    const size_t oTq = carve(f->two_level ? M * a.ntiles * 8 : 8);
    const size_t oP2 = carve(apf2 ? 3 * M * a.nparts * 8 : 8);
    const size_t oHcnt = carve(heavy_list ? M * 2 * sizeof(unsigned) : 8);
    const size_t oHlist = carve(heavy_list ? M * 2 * F_HMAX * 3 * 8 : 8);
    const size_t oSu = carve(need_su ? M * N * 8 + 16 : 8);
    const size_t oE = carve(need_su ? M * (a.ntiles1 + 1) * 8 : 8);
    // one-pass uniform_spacings (two-level step, Philox draws): 1, 2, 4 or 8 tiles of draws per workgroup,
    // the fewest that keep the whole launch resident (<= 1024 workgroups: half of what the chip holds);
    // more islands than that: the three-pass form
    a.sp_tpw = a.sp_nwg = 0;
    bool merge_fits = false;
    if (need_su && f->two_level && !(f->strict && f->strict_literal) && !f->sqmc && !(o->flags & SMC_PATH_SPACING_3PASS))
        for (int tpw = 1; tpw <= 8 && !a.sp_tpw; tpw *= 2) {
            const int forced_tpw = (o->flags >> 25) & 15;          // SMC_PATH_SP_TPW (A/B: workgroups wait for
            if (forced_tpw && tpw != forced_tpw) continue;         //  lower-numbered ones only, dispatch is in order)
            const i64 nwg = (a.ntiles + tpw - 1) / tpw;
            int per_cu = 0;                        // workgroups of this instantiation a CU holds at once
#ifdef SMC_EMULATE
            per_cu = 4;
            (void)per_cu;
#else
            const void* fn = tpw == 1 ? (const void*)k_f_spacing_onepass<1> : tpw == 2 ? (const void*)k_f_spacing_onepass<2>
                           : tpw == 4 ? (const void*)k_f_spacing_onepass<4> : (const void*)k_f_spacing_onepass<8>;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, SMC_BLOCK, 0) != hipSuccess) per_cu = 0;
            (void)hipGetLastError();
#endif
            // (a margin of one workgroup per CU: the occupancy API is optimistic near register-file edges)
#ifdef SMC_EMULATE
            const i64 cap = 1024;                  // (workgroups run one after the other, in order)
#else
            const i64 cap = (i64)(per_cu > 1 ? per_cu - 1 : 0) * ctx->n_cu;
#endif
            if ((nwg * (i64)M <= cap && nwg * (i64)M <= 1024) || (forced_tpw && nwg <= 1024)) {
                a.sp_tpw = tpw;
                a.sp_nwg = (int)nwg;
                // one workgroup more per island when the reduction rides along (see sp_merge): it is the first
                // of the launch and waits for nobody, the others wait for lower-numbered ones only -- dispatch is
                // in order, so the margin kept above is not needed for it, the chip's nominal capacity is
#ifdef SMC_EMULATE
                merge_fits = true;
#else
                merge_fits = (nwg + 1) * (i64)M <= (i64)per_cu * ctx->n_cu || forced_tpw;
#endif
            }
        }
    const size_t oSst = carve(a.sp_tpw ? M * a.sp_nwg * 8 : 8);
    const size_t oSdec = carve(M * 8);
    f->sp_epoch = 0;
    f->flush_pending = false;
    // (inside a replayed graph the argument block -- the epoch with it -- is frozen: separate launches there)
    f->sp_merge = a.sp_tpw && merge_fits && !o->use_graph && !(o->flags & SMC_PATH_SPLIT_REDUCE);
    f->sq_plan_zero = false;
    const size_t oTmp = carve(N * dxm * 8);
    const size_t oStrict = carve(f->strict ? 2 * M * N * 8 + sqx_scratch_bytes((i64)N, (int)M) + M * a.ntiles * 8 + 64 : 8);
    if (f->sqmc && !f->sq_flat && !f->two_level) {
        smc_set_error("SMC_FLAG_SQMC needs the two-level step");
        delete f;
        return SMC_ERR_INVALID;
    }
    const bool apf_mv = mv && model->fk == SMC_FK_APF;
    const size_t oEta = carve(apf_mv ? 2 * M * N * 8 : 8);
    const size_t oSqZ = carve(f->sqmc ? M * N * dxm * 8 : 8);
    const size_t oSqPerm = carve(f->sqmc && (M > 1 || f->sq_flat) ? M * N * 8 : 8);
    // (two-level: the sort's workspace; flat: the step's points (N, d + 1), a row of sorted log-weights and -- d = 1
    // -- the sort's workspace behind them)
    const size_t oSqWs = carve(!f->sqmc ? 8 : f->sq_flat ? (N * (dxm + 1) + N) * 8 + (mv ? 0 : smc_rs_ws_bytes((i64)N))
                                                          : smc_rs_ws_bytes((i64)N));
    a.nmb = (int)((o->N + F_MOM_CHUNK - 1) / F_MOM_CHUNK);
    const size_t oMom = carve(o->moments ? M * T * 2 * dxm * 8 : 8);
    const size_t oMpart = carve(o->moments ? M * a.nmb * dxm * 3 * 8 : 8);
    const size_t oTrace = carve(M * (size_t)(2 * a.ntiles + 8) * 8 * 8 + (size_t)(2 * a.ntiles + 16) * 8 * 8);
    // the slab comes from the context's pool (smc_malloc: blocks recycled by exact size, ordered on
    // the context's one stream): a PMMH chain or the PMCMC moves of SMC^2 create and destroy a filter
    // of the same shape per proposal, and hipMalloc / hipFree of tens of MB cost milliseconds each
    void* slab = nullptr;
    if (smc_malloc(ctx, off, &slab) != SMC_OK) {
        delete f;
        return SMC_ERR_NOMEM;
    }
    f->slab = slab;
    f->slab_bytes = off;
    // (from here on a failing HIP call must not leak the slab and the struct)
#define F_CREATE_CHECK(expr)                                                                   \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            smc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,     \
                          __LINE__);                                                           \
            (void)smc_free(ctx, slab);                                                         \
            if (f->ll_stage) (void)hipHostFree(f->ll_stage);                                   \
            delete f;                                                                          \
            return SMC_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)
    char* base = (char*)slab;
    a.X = (double*)(base + oX0);
    a.lw = (double*)(base + oL0);
    a.A = (u32*)(base + oA);
    a.Q = (u64*)(base + oQ);
    a.Qpre = (u64*)(base + oQpre);
    a.pm = (double*)(base + oPm); a.ps = (double*)(base + oPs); a.pss = (double*)(base + oPss);
    a.summ = (double*)(base + oSum);
    double* dpar = (double*)(base + oPar);
    double* dy = (double*)(base + oY);
    a.params = dpar;
    a.y = dy;
    a.cnt = (unsigned*)(base + oCtl);
    a.cq = (u64*)(base + oCq);
    a.tq = (u64*)(base + oTq);
    a.spart = (double*)(base + oSpart);
    a.info = (double*)(base + oInfo);
    a.info2 = (double*)(base + oInfo2);
    if (heavy_list) {
        a.hcnt = (unsigned*)(base + oHcnt);
        a.hlist = (i64*)(base + oHlist);
        F_CREATE_CHECK(hipMemsetAsync(a.hcnt, 0, M * 2 * sizeof(unsigned), ctx->stream));
    }
    a.exact_counts = (o->flags & SMC_PATH_EXACT_COUNTS) ? 1 : 0;
    f->no_tk = (o->flags & SMC_PATH_NO_TK) != 0;
    a.tk = -1;
    // streaming stores pay while a launch is short (its end-of-kernel write-back shows): C2 +8 %;
    // on the large grids they cost 2 % (C5)
    a.nt = ((i64)a.ntiles * (i64)M <= F_DIRECT_PREFIX_MAX && !mv) ? 15 : 0;
    // (bits: 1 X, 2 lw, 4 the tile CDF, 8 A.  Measured at C2, r12f: any mask that streams lw -- written every step, read
    //  only on the steps that do not resample -- is as fast as streaming everything, 17.63 us; none: 19.31.  With
    //  consecutive tiles per XCD, r12w: 15: 17.4, X plain 17.5, X and the tile CDF plain 18.0, lw only 18.3)
    a.pm2 = a.ps2 = a.pss2 = nullptr;
    if (apf2) {
        a.pm2 = (double*)(base + oP2);
        a.ps2 = a.pm2 + M * a.nparts;
        a.pss2 = a.ps2 + M * a.nparts;
    }
    a.su = (double*)(base + oSu);
    a.E = (u64*)(base + oE);
    a.sst = (u64*)(base + oSst);
    if (a.sp_tpw) F_CREATE_CHECK(hipMemsetAsync(a.sst, 0, M * a.sp_nwg * 8, ctx->stream));
    a.sdec = (u64*)(base + oSdec);
    F_CREATE_CHECK(hipMemsetAsync(a.sdec, 0, M * 8, ctx->stream));
    f->tmp = (double*)(base + oTmp);
    f->strict_ws = (double*)(base + oStrict);
    if (f->strict)         // (the counters of smc_seqx.h: zero once, re-armed by its passes)
        sqx_zero_counters(ctx->stream, (void*)(f->strict_ws + 2 * M * N), (i64)N, (int)M);
    if (apf_mv) {
        a.eta = (double*)(base + oEta);
        a.lwsv = a.eta + M * N;
    }
    if (f->sqmc) {
        f->sq_z = (double*)(base + oSqZ);
        f->sq_perm = (u64*)(base + oSqPerm);
        f->sq_ws = base + oSqWs;
    }
    f->ll_stage = nullptr;
    {
        auto it = ctx->pinned.find(M * 8);
        if (it != ctx->pinned.end() && !it->second.empty()) {
            f->ll_stage = (double*)it->second.back();
            it->second.pop_back();
        } else if (hipHostMalloc((void**)&f->ll_stage, M * 8, hipHostMallocMapped) != hipSuccess) {
            f->ll_stage = nullptr;
        }
    }
    (void)hipGetLastError();
    if (o->moments) {
        a.mom = (double*)(base + oMom);
        a.mpart = (double*)(base + oMpart);
    }
    a.trace = (u64*)(base + oTrace);
    hipStream_t st = ctx->stream;
    F_CREATE_CHECK(hipMemsetAsync(a.summ, 0, M * (T + 1) * SUMM_STRIDE * 8, st));
    F_CREATE_CHECK(hipMemsetAsync(a.cnt, 0, M * 2 * F_CNT_WORDS * sizeof(unsigned), st));
    F_CREATE_CHECK(hipMemsetAsync(a.Q, 0, M * a.ntiles * 8, st));
    {   // step record of t = 0: {t, rs_flag, y_0, m, 1/s}
        std::vector<double> h(M * INFO_STRIDE, 0.0);
        for (size_t i = 0; i < M; ++i) {
            h[i * INFO_STRIDE + 2] = y_host[0];
            h[i * INFO_STRIDE + 5] = model->aux_host ? model->aux_host[0] : 0.0;
        }
        F_CREATE_CHECK(hipMemcpyAsync(a.info, h.data(), h.size() * 8, hipMemcpyHostToDevice, st));
        F_CREATE_CHECK(hipMemsetAsync(a.info2, 0, M * INFO_STRIDE * 8, st));
        F_CREATE_CHECK(hipStreamSynchronize(st));
    }
    F_CREATE_CHECK(hipMemsetAsync(a.A, 0, (M * N + F_TILE) * 4, st));     // (+ the padding tile: valid indices)
    std::vector<double> par_host(M * PARAM_STRIDE, 0.0);
    if (!mv) {       // the host's rows + the correctly rounded reciprocals smc_div_c works with
        for (size_t i = 0; i < M; ++i)
            for (int k = 0; k < PARAM_HOST; ++k) {
                const double v = model->params_host[i * PARAM_HOST + k];
                par_host[i * PARAM_STRIDE + k] = v;
                const double av = v < 0 ? -v : v;
                par_host[i * PARAM_STRIDE + PARAM_HOST + k] = (av > 1e-20 && av < 1e20) ? 1.0 / v : 0.0;
            }
        F_CREATE_CHECK(hipMemcpyAsync(dpar, par_host.data(), M * PARAM_STRIDE * 8,
                                     hipMemcpyHostToDevice, st));
    }
    a.mvc = (const double*)(base + oMvc);
    if (mv)
        F_CREATE_CHECK(hipMemcpyAsync((void*)a.mvc, mvc_host.data(), mvc_host.size() * 8,
                                     hipMemcpyHostToDevice, st));
    F_CREATE_CHECK(hipMemcpyAsync(dy, y_host, T * dym * 8, hipMemcpyHostToDevice, st));
    if (model->aux_host) {
        a.aux = (const double*)(base + oAux);
        F_CREATE_CHECK(hipMemcpyAsync((void*)a.aux, model->aux_host, T * 8, hipMemcpyHostToDevice, st));
    }
    F_CREATE_CHECK(hipStreamSynchronize(st));
    F_CREATE_CHECK(hipStreamSynchronize(st));
#undef F_CREATE_CHECK
    *out = f;
    return SMC_OK;
}

#ifdef SMC_TRACE
// debug builds only (tools/trace_step.py): per-workgroup phase stamps of the last step
int smc_debug_trace(smc_filter* f, uint64_t* out_host)
{
    // (n_islands, nparts, 8) stamps of k_propagate, then (n_islands, ntiles, 8) of k_ancestors
    SMC_HIP_CHECK(hipMemcpyAsync(out_host, f->a.trace,
                                 (size_t)f->a.n_islands * (f->a.nparts + f->a.ntiles) * 64,
                                 hipMemcpyDeviceToHost, f->ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
    return SMC_OK;
}
// ... of the strict step's two launches (island 0): (2 ntiles + 8, 8) stamps -- classify per tile, the chain, search per tile
int smc_debug_trace_strict(smc_filter* f, uint64_t* out_host)
{
    SMC_HIP_CHECK(hipMemcpyAsync(out_host, f->a.trace + (size_t)f->a.n_islands * (f->a.nparts + f->a.ntiles) * 8,
                                 (size_t)(2 * f->a.ntiles + 8) * 64, hipMemcpyDeviceToHost, f->ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
    return SMC_OK;
}
#endif

int smc_filter_destroy(smc_filter* f)
{
    if (!f) return SMC_OK;
    (void)hipSetDevice(f->ctx->device);
    (void)hipStreamSynchronize(f->ctx->stream);
    for (hipGraphExec_t g : f->gexec)
        if (g) (void)hipGraphExecDestroy(g);
    for (hipEvent_t e : f->ev) (void)hipEventDestroy(e);
    (void)smc_free(f->ctx, f->slab);
    if (f->ll_stage) {
        auto& v = f->ctx->pinned[(size_t)f->a.n_islands * 8];
        if (v.size() < 4) v.push_back(f->ll_stage);
        else (void)hipHostFree(f->ll_stage);
    }
    if (f->th_buf) (void)hipFree(f->th_buf);
    delete f;
    return SMC_OK;
}

// An independent copy of a filter in its current state: what copy.deepcopy(pf) is to the reference
// (smc_samplers.py:319-361: theta-level resampling of SMC^2 deep-copies every duplicated filter).
// One slab allocation, one device-to-device copy, every slab pointer of the argument block re-based;
// replay tapes (caller-owned, read-only) are shared.  Philox streams are a function of
// (seed, island id, particle, t): a clone continues with the SAME draws as its source -- exactly
// what deepcopy of a filter plus numpy's global generator gives the reference NOT; callers that
// need the copies to diverge re-seed them (smc_filter_reseed).
int smc_filter_clone(smc_filter* src, smc_filter** out)
{
    if (src) { (void)hipSetDevice(src->ctx->device); flush_rows(src); }
    SMC_REQUIRE(src && out, "null argument");
    smc_ctx* ctx = src->ctx;
    SMC_HIP_CHECK(hipSetDevice(ctx->device));
    void* slab = nullptr;
    if (smc_malloc(ctx, src->slab_bytes, &slab) != SMC_OK) return SMC_ERR_NOMEM;
    smc_filter* f = new smc_filter(*src);
    f->slab = slab;
    f->gexec[0] = f->gexec[1] = f->gexec[2] = nullptr;      // graphs hold the source's addresses
    f->graph_failed = false;
    f->prof = false;
    f->prof_n = 0;
    f->ev.clear();
    f->ll_stage = nullptr;
    f->th_buf = nullptr;
    f->lwth = f->th = f->th_ess = nullptr;
    const char* s0 = (const char*)src->slab;
    const ptrdiff_t delta = (char*)slab - s0;
    auto rebase = [&](auto*& p) {
        const char* c = (const char*)p;
        if (c && c >= s0 && c < s0 + src->slab_bytes) p = (std::remove_reference_t<decltype(p)>)((char*)p + delta);
    };
    FArgs& a = f->a;
    rebase(a.X); rebase(a.lw); rebase(a.A); rebase(a.Q); rebase(a.Qpre); rebase(a.pm); rebase(a.ps); rebase(a.pss);
    rebase(a.cq); rebase(a.tq); rebase(a.cnt); rebase(a.spart); rebase(a.summ); rebase(a.params); rebase(a.y);
    rebase(a.mom); rebase(a.mpart); rebase(a.aux); rebase(a.info); rebase(a.hcnt); rebase(a.hlist); rebase(a.info2);
    rebase(a.su); rebase(a.E); rebase(a.sst); rebase(a.mvc); rebase(a.trace); rebase(a.pm2); rebase(a.ps2); rebase(a.pss2);
    rebase(a.eta); rebase(a.lwsv); rebase(a.sdec);
    rebase(f->tmp); rebase(f->strict_ws); rebase(f->sq_z); rebase(f->sq_perm); rebase(f->sq_ws);
    hipStream_t st = ctx->stream;
    hipError_t e = hipMemcpyAsync(slab, src->slab, src->slab_bytes, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && src->th_buf) {
        const size_t M = (size_t)a.n_islands, T = (size_t)a.T, Ng = (size_t)src->th_n;
        const size_t nb = (Ng + TH_STRIDE + 2 * T + (src->th_comm ? M + Ng : 0)) * 8;
        e = hipMalloc(&f->th_buf, nb);
        if (e == hipSuccess) e = hipMemcpyAsync(f->th_buf, src->th_buf, nb, hipMemcpyDeviceToDevice, st);
        f->lwth = (double*)f->th_buf;
        f->th = f->lwth + Ng;
        f->th_ess = f->th + TH_STRIDE;
        f->th_send = src->th_comm ? f->th_ess + 2 * T : nullptr;
        f->th_recv = src->th_comm ? f->th_send + M : nullptr;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        smc_set_error("smc_filter_clone: %s", hipGetErrorString(e));
        if (f->th_buf) (void)hipFree(f->th_buf);
        (void)smc_free(ctx, slab);
        delete f;
        return SMC_ERR_HIP;
    }
    if (hipHostMalloc((void**)&f->ll_stage, (size_t)a.n_islands * 8, hipHostMallocMapped) != hipSuccess)
        f->ll_stage = nullptr;
    (void)hipGetLastError();
    *out = f;
    return SMC_OK;
}

// ---- checkpoint / resume (pickling of a device filter: core.py:415-428 returns SMC objects from worker processes,
// utils.py:178-186).  The state of a filter is its slab -- every device array, no pointers inside -- plus a handful of
// host words; a filter created from the same (model, options, data) has the same layout, so the slab of one can be
// loaded into the other and the run continues bit for bit.
struct FStateHeader {
    u64 magic, slab_bytes;
    i64 N, T, t_host, perm_t;
    int n_islands, scheme, kind, fk, hist, dx, flags_strict, flags_sqmc;
    u64 seed, sp_epoch, sq_seed, sq_ctr0;
    int flush_pending, island_offset;
};
#define F_STATE_MAGIC 0x534d435f53544131ull    /* "SMC_STA1" */
int smc_filter_state_bytes(smc_filter* f, int64_t* nbytes)
{
    SMC_REQUIRE(f && nbytes, "null argument");
    SMC_REQUIRE(!f->th_buf, "a filter with a theta level (SMC^2) is checkpointed by its owner, not as a filter");
    *nbytes = (int64_t)(sizeof(FStateHeader) + f->slab_bytes);
    return SMC_OK;
}
int smc_filter_save_state(smc_filter* f, void* out_host, int64_t nbytes)
{
    SMC_REQUIRE(f && out_host, "null argument");
    SMC_REQUIRE(!f->th_buf, "a filter with a theta level (SMC^2) is checkpointed by its owner, not as a filter");
    SMC_REQUIRE(nbytes == (int64_t)(sizeof(FStateHeader) + f->slab_bytes), "buffer size: smc_filter_state_bytes");
    SMC_HIP_CHECK(hipSetDevice(f->ctx->device));
    flush_rows(f);
    FStateHeader h;
    memset(&h, 0, sizeof h);
    h.magic = F_STATE_MAGIC; h.slab_bytes = f->slab_bytes;
    h.N = f->a.N; h.T = f->a.T; h.t_host = f->t_host; h.perm_t = f->perm_t;
    h.n_islands = f->a.n_islands; h.scheme = f->a.scheme; h.kind = f->kind; h.fk = f->fk; h.hist = f->a.hist; h.dx = f->a.dx;
    h.flags_strict = f->strict ? 1 : 0; h.flags_sqmc = f->sqmc ? 1 : 0;
    h.seed = f->a.seed; h.sp_epoch = f->sp_epoch; h.sq_seed = f->sq_seed; h.sq_ctr0 = f->sq_ctr0;
    h.flush_pending = f->flush_pending ? 1 : 0; h.island_offset = f->a.island_offset;
    memcpy(out_host, &h, sizeof h);
    SMC_HIP_CHECK(hipMemcpyAsync((char*)out_host + sizeof h, f->slab, f->slab_bytes, hipMemcpyDeviceToHost, f->ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
    return SMC_OK;
}
int smc_filter_load_state(smc_filter* f, const void* in_host, int64_t nbytes)
{
    SMC_REQUIRE(f && in_host, "null argument");
    SMC_REQUIRE(!f->th_buf, "a filter with a theta level (SMC^2) is checkpointed by its owner, not as a filter");
    SMC_REQUIRE(nbytes >= (int64_t)sizeof(FStateHeader), "truncated state");
    FStateHeader h;
    memcpy(&h, in_host, sizeof h);
    SMC_REQUIRE(h.magic == F_STATE_MAGIC, "not a filter state (magic)");
    SMC_REQUIRE(h.slab_bytes == f->slab_bytes && nbytes == (int64_t)(sizeof h + h.slab_bytes) && h.N == f->a.N && h.T == f->a.T &&
                    h.n_islands == f->a.n_islands && h.scheme == f->a.scheme && h.kind == f->kind && h.fk == f->fk &&
                    h.hist == f->a.hist && h.dx == f->a.dx && h.flags_strict == (f->strict ? 1 : 0) &&
                    h.flags_sqmc == (f->sqmc ? 1 : 0),
                "the state belongs to a filter of another shape (model, N, T, islands, scheme, history or flags differ)");
    SMC_HIP_CHECK(hipSetDevice(f->ctx->device));
    SMC_HIP_CHECK(hipMemcpyAsync(f->slab, (const char*)in_host + sizeof h, f->slab_bytes, hipMemcpyHostToDevice, f->ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
    f->t_host = h.t_host; f->perm_t = h.perm_t;
    f->a.seed = h.seed; f->sp_epoch = h.sp_epoch; f->sq_seed = h.sq_seed; f->sq_ctr0 = h.sq_ctr0;
    f->flush_pending = h.flush_pending != 0;
    f->a.island_offset = h.island_offset;
    for (hipGraphExec_t& g : f->gexec)          // captured launches carry the old key / counters by value
        if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
    return SMC_OK;
}

// New Philox key for the steps still to run (a clone that must not repeat its source's draws).
int smc_filter_reseed(smc_filter* f, uint64_t seed)
{
    SMC_REQUIRE(f, "null filter");
    f->a.seed = seed;
    for (hipGraphExec_t& g : f->gexec)          // captured launches carry the old key by value
        if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
    return SMC_OK;
}

// SMC_FLAG_SQMC: which points the run uses -- the stand-alone operator's stream (smc_sobol with
// scramble = safe = 1 under a context seeded with `point_seed`): point set counter0 at t = 0 (one
// coordinate), counter0 + t at step t (two coordinates, sorted by the first).  Default: the filter's own
// seed, counter0 = 1.
int smc_filter_sqmc_points(smc_filter* f, uint64_t point_seed, uint64_t counter0)
{
    SMC_REQUIRE(f, "null filter");
    SMC_REQUIRE(f->sqmc, "the filter was not created with SMC_FLAG_SQMC");
    SMC_REQUIRE(f->t_host == 0, "the point stream must be chosen before the first step");
    f->sq_seed = point_seed;
    f->sq_ctr0 = counter0;
    return SMC_OK;
}

int smc_filter_set_replay(smc_filter* f, const double* z, const double* u)
{
    SMC_REQUIRE(f, "null filter");
    SMC_REQUIRE(!f->sqmc, "SMC_FLAG_SQMC filters generate their points on the device");
    SMC_REQUIRE(f->t_host == 0, "replay tapes must be set before the first step");
    SMC_REQUIRE(z && u, "both tapes are required");
    f->a.zt = z;
    f->a.zt_ts = (i64)f->a.n_islands * f->a.N * f->a.dx;
    f->a.ut = u;
    f->a.ut_stride = (f->a.scheme == SMC_SYSTEMATIC) ? 1 : f->a.N;
    f->a.rng_mode = SMC_RNG_REPLAY;
    return SMC_OK;      // (the argument block travels by value with every launch)
}

// N <= 1024, univariate, no per-step side kernels: one persistent workgroup per island runs the
// requested steps in a single launch (smc_filter_small.h)
static bool small_filter_ok(const smc_filter* f)
{
    return f->a.N <= F_TILE && f->kind != SMC_MODEL_MVLINGAUSS && !f->a.mom && !f->prof && !f->strict &&
           !(f->a.scheme == SMC_MULTINOMIAL && !f->a.ut) && !f->no_small && !f->sqmc;
}

static void launch_small(smc_filter* f, int nsteps)
{
    hipStream_t st = f->ctx->stream;
    const dim3 grid(1, f->a.n_islands);
    f->a.par = -1;
#define S_CASE(KINDV, FKV)                                                                       \
    if (f->kind == KINDV && f->fk == FKV) {                                                      \
        if (f->a.N <= 256)                                                                       \
            SMC_LAUNCH((k_filter_small<KINDV, FKV, 64>), grid, dim3(64), st, f->a, nsteps);      \
        else                                                                                     \
            SMC_LAUNCH((k_filter_small<KINDV, FKV, SMC_BLOCK>), grid, dim3(SMC_BLOCK), st, f->a, \
                       nsteps);                                                                  \
        return;                                                                                  \
    }
    S_CASE(SMC_MODEL_LINGAUSS, SMC_FK_BOOTSTRAP)
    S_CASE(SMC_MODEL_LINGAUSS, SMC_FK_GUIDED)
    S_CASE(SMC_MODEL_STOCHVOL, SMC_FK_BOOTSTRAP)
    S_CASE(SMC_MODEL_STOCHVOL, SMC_FK_GUIDED)
    S_CASE(SMC_MODEL_STOCHVOL, SMC_FK_APF)
    S_CASE(SMC_MODEL_LINGAUSS, SMC_FK_APF)
    S_CASE(SMC_MODEL_STOCHVOL, SMC_FK_APF_BOOT)
    S_CASE(SMC_MODEL_LINGAUSS, SMC_FK_APF_BOOT)
    S_CASE(SMC_MODEL_GORDON, SMC_FK_BOOTSTRAP)
    S_CASE(SMC_MODEL_THETALOGISTIC, SMC_FK_BOOTSTRAP)
    S_CASE(SMC_MODEL_SVLEVERAGE, SMC_FK_BOOTSTRAP)
    S_CASE(SMC_MODEL_DISCRETECOX, SMC_FK_BOOTSTRAP)
#undef S_CASE
}

// The summary row of step t (ESS, evidence terms, the (K, 1/s) its weights are normalised with) is written by the
// reduction that OPENS step t + 1; behind the last step enqueued nobody has done it yet.  k_flush2 does -- when
// somebody asks for results (every accessor below calls this), not at the end of every smc_filter_step call: a
// caller that steps in pieces (20 steps, sync, 20 steps ...) paid a launch per piece for a row the next piece's first
// kernel writes anyway.
static void flush_rows(smc_filter* f)
{
    if (f && f->flush_pending) {
        f->flush_pending = false;
        SMC_LAUNCH(k_flush2, dim3(f->a.n_islands), dim3(SMC_BLOCK), f->ctx->stream, f->a);
    }
}

int smc_filter_step(smc_filter* f, int64_t nsteps)
{
    SMC_REQUIRE(f, "null filter");
    SMC_REQUIRE(nsteps >= 0, "nsteps must be non-negative");
    SMC_HIP_CHECK(hipSetDevice(f->ctx->device));
    hipStream_t st = f->ctx->stream;
    i64 todo = nsteps;
    if (f->t_host + todo > f->a.T) todo = f->a.T - f->t_host;
    if (todo < 0) todo = 0;
    if (todo > 0 && f_is_apf(f->fk) && f->kind != SMC_MODEL_MVLINGAUSS && !small_filter_ok(f) && !f->a.pm2) {
        smc_set_error("the auxiliary particle filter with N <= 1024 runs on the one-launch filter only (no "
                      "profiling, no Philox multinomial)");
        return SMC_ERR_STATE;
    }
    if (todo > 0 && f->lwth) {
        // theta level on: the theta-weights are updated behind every step (and may freeze the batch)
        const bool small = small_filter_ok(f);
        for (i64 k = 0; k < todo; ++k) {
            if (small) launch_small(f, 1);
            else {
                enqueue_step(f, -1, f->t_host + k);
                if (f->two_level) SMC_LAUNCH(k_flush2, dim3(f->a.n_islands), dim3(SMC_BLOCK), st, f->a);
            }
            if (f->th_comm) {
                // sharded population: my filters' increments -> all ranks' (ncclAllGather on this stream, no
                // host round trip) -> the replicated theta level, the same arithmetic on every rank
                SMC_LAUNCH(k_theta_pack, dim3(1), dim3(SMC_BLOCK), st, f->a, (const double*)f->th, f->th_send,
                           f->two_level ? 1 : 0);
                const int rc = smc_comm_allgather_f64_async(f->th_comm, f->th_send, f->a.n_islands, f->th_recv);
                if (rc) return rc;
            }
            SMC_LAUNCH(k_theta_update, dim3(1), dim3(SMC_BLOCK), st, f->a, f->lwth, f->th, f->th_ess,
                       f->th_ess_min, f->two_level ? 1 : 0, (const double*)(f->th_comm ? f->th_recv : nullptr),
                       f->th_comm ? f->th_n : f->a.n_islands);
        }
        SMC_LAUNCH_CHECK();
        f->t_host += todo;
        return SMC_OK;
    }
    if (todo > 0 && small_filter_ok(f)) {
        launch_small(f, (int)(todo > 0x7fffffff ? 0x7fffffff : todo));
        SMC_LAUNCH_CHECK();
        f->t_host += todo;
        return SMC_OK;
    }
    i64 done = 0;
#ifndef SMC_EMULATE
    // hipGraphs of 24, 8 and 2 steps (even sizes, entered at even t: the slot parity is baked into
    // the nodes), captured on first use; any request of two or more steps replays them greedily
    static const int F_GRAPH_SIZES[3] = {24, 8, 2};
    if (f->use_graph && !f->prof && !f->graph_failed && todo >= 2) {
        // graphs start at even t (SQMC: at even t >= 2 -- step 0 has no sort, the captured sequence always does)
        while (done < todo && (((f->t_host + done) & 1) || (f->sqmc && f->t_host + done < 2))) {
            enqueue_step(f, -1, f->t_host + done);
            ++done;
        }
        for (int gi = 0; gi < 3 && !f->graph_failed; ++gi) {
            const int GS = F_GRAPH_SIZES[gi];
            if (todo - done < GS) continue;
            if (!f->gexec[gi]) {   // (kernel arguments are final by the first step call)
                hipGraph_t g = nullptr;
                bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
                if (ok) {
                    for (int k = 0; k < GS; ++k) enqueue_step(f, -1, k, false);
                    ok = hipStreamEndCapture(st, &g) == hipSuccess && g &&
                         hipGraphInstantiate(&f->gexec[gi], g, nullptr, nullptr, 0) == hipSuccess;
                }
                if (g) (void)hipGraphDestroy(g);
                (void)hipGetLastError();
                if (!ok) { f->gexec[gi] = nullptr; f->graph_failed = true; break; }
            }
            while (todo - done >= GS) {
                SMC_HIP_CHECK(hipGraphLaunch(f->gexec[gi], st));
                done += GS;
            }
        }
    }
#endif
    for (; done < todo; ++done) {
        int kp = -1;
        if (f->prof && f->prof_n < PROF_MAX) kp = f->prof_n++;
        enqueue_step(f, kp, f->t_host + done);
    }
    if (f->two_level && todo > 0) f->flush_pending = true;      // summary row of the last step: flush_rows, on demand
    SMC_LAUNCH_CHECK();
    f->t_host += todo;
    return SMC_OK;
}

int smc_filter_sync(smc_filter* f)
{
    SMC_REQUIRE(f, "null filter");
    SMC_HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
    return SMC_OK;
}

int smc_filter_t(smc_filter* f, int64_t* t_out)
{
    SMC_REQUIRE(f && t_out, "null argument");
    *t_out = f->t_host;
    return SMC_OK;
}

int smc_filter_summaries(smc_filter* f, double* out_host)
{
    if (f) { (void)hipSetDevice(f->ctx->device); flush_rows(f); }
    SMC_REQUIRE(f && out_host, "null argument");
    const i64 t = f->t_host, T = f->a.T;
    if (t == 0) return SMC_OK;
    std::vector<double> h((size_t)f->a.n_islands * (T + 1) * SUMM_STRIDE);
    SMC_HIP_CHECK(hipMemcpyAsync(h.data(), f->a.summ, h.size() * 8, hipMemcpyDeviceToHost,
                                 f->ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
    for (int i = 0; i < f->a.n_islands; ++i)
        for (i64 s = 0; s < t; ++s)
            for (int c = 0; c < SMC_SUMMARY_COLS; ++c)
                out_host[((size_t)i * t + s) * SMC_SUMMARY_COLS + c] =
                    h[((size_t)i * (T + 1) + s) * SUMM_STRIDE + c];
    return SMC_OK;
}

// gather logLt of step t-1 of every island into one staging array (one D2H copy, one sync)
__global__ void k_f_collect_logLt(const double* summ, i64 T, i64 t, int n, double* out)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) out[i] = summ[((i64)i * (T + 1) + (t - 1)) * SUMM_STRIDE + 3];
}

int smc_filter_logLt(smc_filter* f, double* out_host)
{
    if (f) { (void)hipSetDevice(f->ctx->device); flush_rows(f); }
    SMC_REQUIRE(f && out_host, "null argument");
    const i64 t = f->t_host, T = f->a.T;
    const int M = f->a.n_islands;
    for (int i = 0; i < M; ++i) out_host[i] = 0.0;
    if (t == 0) return SMC_OK;
    hipStream_t st = f->ctx->stream;
    if (f->ll_stage) {
        // the kernel writes into pinned host memory: one launch + one (spinning) stream sync
        SMC_LAUNCH(k_f_collect_logLt, dim3((M + 255) / 256), dim3(256), st, f->a.summ, T, t, M, f->ll_stage);
        SMC_LAUNCH_CHECK();
        SMC_HIP_CHECK(hipStreamSynchronize(st));
        memcpy(out_host, f->ll_stage, (size_t)M * 8);
        return SMC_OK;
    }
    for (int i = 0; i < M; ++i)
        SMC_HIP_CHECK(hipMemcpyAsync(out_host + i, f->a.summ + ((size_t)i * (T + 1) + (t - 1)) * SUMM_STRIDE + 3, 8,
                                     hipMemcpyDeviceToHost, st));
    SMC_HIP_CHECK(hipStreamSynchronize(st));
    return SMC_OK;
}

int smc_filter_get(smc_filter* f, int field, int island, void* out_host)
{
    if (f) { (void)hipSetDevice(f->ctx->device); flush_rows(f); }
    SMC_REQUIRE(f && out_host, "null argument");
    SMC_REQUIRE(island >= 0 && island < f->a.n_islands, "island out of range");
    const i64 t = f->t_host;
    if (t == 0) {
        smc_set_error("smc_filter_get: no step has run yet");
        return SMC_ERR_STATE;
    }
    return filter_fetch(f, field, t - 1, island, out_host);
}

/* state of step s (the filter's current one, or -- keep_history -- any earlier one) */
static int filter_fetch(smc_filter* f, int field, i64 s, int island, void* out_host)
{
    const i64 N = f->a.N;
    const i64 t = s + 1;
    hipStream_t st = f->ctx->stream;
    const int dx = f->a.dx;
    const double* X = f_X(f->a, s) + (size_t)island * N * dx;
    const double* Xo = f_X(f->a, s - 1) + (size_t)island * N * dx;
    size_t nbytes = (size_t)N * 8;
    const double* lw = f_lw(f->a, s) + (size_t)island * N;
    const u32* A = f_A(f->a, s) + (size_t)island * N;
    const void* src = nullptr;
    const unsigned nb = (unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK);
    switch (field) {
    case SMC_FIELD_X: src = X; nbytes *= dx; break;
    case SMC_FIELD_LW: src = lw; break;
    case SMC_FIELD_A:
    case SMC_FIELD_XP: {
        if (t < 2) { smc_set_error("smc_filter_get: A / Xp are undefined before step 1"); return SMC_ERR_STATE; }
        if (f->perm_t == f->t_host && s == f->t_host - 1) {
            smc_set_error("smc_filter_get: A / Xp are undefined right after smc_filter_permute_islands");
            return SMC_ERR_STATE;
        }
        // did the last step resample?  (core.py:329-336: else A = arange(N), Xp = X)
        double flag = 0.0;
        SMC_HIP_CHECK(hipMemcpyAsync(&flag, f->a.summ + ((size_t)island * (f->a.T + 1) + (t - 1)) * SUMM_STRIDE + 4,
                                     8, hipMemcpyDeviceToHost, st));
        SMC_HIP_CHECK(hipStreamSynchronize(st));
        if (flag == 0.0) {
            if (field == SMC_FIELD_A) {
                int64_t* o = (int64_t*)out_host;
                for (i64 i = 0; i < N; ++i) o[i] = i;
                return SMC_OK;
            }
            src = Xo;
            nbytes *= dx;
        } else if (field == SMC_FIELD_A) {
            SMC_LAUNCH(k_f_widen, dim3(nb), dim3(SMC_BLOCK), st, A, N, (i64*)f->tmp);
            src = f->tmp;
        } else {
            SMC_LAUNCH(k_f_gather_rows, dim3((unsigned)((N * dx + SMC_BLOCK - 1) / SMC_BLOCK)),
                       dim3(SMC_BLOCK), st, Xo, A, N, dx, f->tmp);
            src = f->tmp;
            nbytes *= dx;
        }
        break;
    }
    case SMC_FIELD_W: {
        const double* row = f->a.summ + ((size_t)island * (f->a.T + 1) + (t - 1)) * SUMM_STRIDE;
        SMC_LAUNCH(k_f_write_W, dim3(nb), dim3(SMC_BLOCK), st, lw, N, row, f->tmp, f->two_level ? 1 : 0);
        src = f->tmp;
        break;
    }
    default:
        smc_set_error("smc_filter_get: unknown field %d", field);
        return SMC_ERR_INVALID;
    }
    SMC_LAUNCH_CHECK();
    SMC_HIP_CHECK(hipMemcpyAsync(out_host, src, nbytes, hipMemcpyDeviceToHost, st));
    SMC_HIP_CHECK(hipStreamSynchronize(st));
    return SMC_OK;
}

// Replace the particles and / or log-weights of the step just done (island `island`) by host
// arrays: what a caller does that moves particles itself between two steps (an MCMC
// rejuvenation of the x-particles, importing a particle system; the reference's counterpart is
// assigning pf.X / pf.wgts, core.py:222-233).  Everything the next step derives from the
// log-weights (ESS, log-mean weight and evidence of this step, the resample decision and the
// CDF of the next) is recomputed on the device as if the step had produced these values.
int smc_filter_set_state(smc_filter* f, int island, const double* X_host, const double* lw_host)
{
    if (f) { (void)hipSetDevice(f->ctx->device); flush_rows(f); }
    SMC_REQUIRE(f, "null filter");
    SMC_REQUIRE(island >= 0 && island < f->a.n_islands, "island out of range");
    if (f->t_host == 0) {
        smc_set_error("smc_filter_set_state: no step has run yet");
        return SMC_ERR_STATE;
    }
    if (f->kind == SMC_MODEL_MVLINGAUSS && lw_host) {
        smc_set_error("smc_filter_set_state: log-weights of a multivariate filter cannot be replaced");
        return SMC_ERR_STATE;
    }
    if (f_is_apf(f->fk)) {
        // the next step of an auxiliary filter resamples on lw + logeta(X) and resets the weights to a
        // constant formed from both (core.py:299-313): replacing X or lw alone would leave the record's
        // auxiliary normalisation and reset constant stale (one-launch filter and two-level step alike)
        smc_set_error("smc_filter_set_state: not available for the auxiliary particle filter");
        return SMC_ERR_STATE;
    }
    SMC_HIP_CHECK(hipSetDevice(f->ctx->device));
    hipStream_t st = f->ctx->stream;
    const i64 N = f->a.N, ts = f->t_host - 1;
    if (X_host)
        SMC_HIP_CHECK(hipMemcpyAsync(f_X(f->a, ts) + (size_t)island * N * f->a.dx, X_host,
                                     (size_t)N * f->a.dx * 8, hipMemcpyHostToDevice, st));
    if (lw_host) {
        SMC_HIP_CHECK(hipMemcpyAsync(f_lw(f->a, ts) + (size_t)island * N, lw_host, (size_t)N * 8,
                                     hipMemcpyHostToDevice, st));
        SMC_LAUNCH(k_f_partials, dim3(f->a.nparts, f->a.n_islands), dim3(SMC_BLOCK), st, f->a, ts,
                   f->two_level ? 1 : 0);
        if (f->two_level) SMC_LAUNCH(k_flush2, dim3(f->a.n_islands), dim3(SMC_BLOCK), st, f->a);
        else SMC_LAUNCH(k_f_restate, dim3(f->a.n_islands), dim3(SMC_BLOCK), st, f->a, ts);
    }
    SMC_LAUNCH_CHECK();
    SMC_HIP_CHECK(hipStreamSynchronize(st));
    return SMC_OK;
}

// The per-island arrays that make up the state of a filter at step t-1 (what theta-level
// resampling and the PMCMC move transport): (pointer, 8-byte words per island)
struct IslandArray { void* p; i64 words; };
static int island_arrays(smc_filter* f, i64 t, IslandArray* out)
{
    const FArgs& a = f->a;
    const i64 N = a.N, T = a.T;
    int n = 0;
    out[n++] = {f_X(a, t - 1), N * a.dx};
    out[n++] = {f_lw(a, t - 1), N};
    out[n++] = {a.summ, (T + 1) * SUMM_STRIDE};
    out[n++] = {a.info, INFO_STRIDE};
    if (f->kind != SMC_MODEL_MVLINGAUSS) out[n++] = {(void*)a.params, PARAM_STRIDE};
    if (f->two_level) {
        out[n++] = {a.pm, a.nparts};
        out[n++] = {a.ps, a.nparts};
        out[n++] = {a.pss, a.nparts};
        out[n++] = {a.tq, a.nparts};
        out[n++] = {a.info2, INFO_STRIDE};
        out[n++] = {a.cq, a.ncq};
    }
    return n;
}

// theta-level resampling of whole filters (SMC^2: smc_samplers.py:319-361 FancyList
// deep copies): island i continues from the state of island src[i] -- particles,
// log-weights, per-step summaries, step record and parameter row move together; the
// Philox streams stay tied to the SLOT (two copies of one island evolve independently).
// One gather kernel + one contiguous copy per array (a dozen launches, whatever n_islands).
int smc_filter_permute_islands(smc_filter* f, const int64_t* src_host)
{
    if (f) { (void)hipSetDevice(f->ctx->device); flush_rows(f); }
    SMC_REQUIRE(f && src_host, "null argument");
    if (f->a.pm2) {
        smc_set_error("whole-island moves are not available for the auxiliary filter on the two-level step");
        return SMC_ERR_STATE;
    }
    if (f->a.hist) {
        smc_set_error("smc_filter_permute_islands: not available with keep_history");
        return SMC_ERR_STATE;
    }
    const int M = f->a.n_islands;
    const i64 t = f->t_host;
    for (int i = 0; i < M; ++i)
        if (src_host[i] < 0 || src_host[i] >= M) {
            smc_set_error("smc_filter_permute_islands: source %lld out of range", (long long)src_host[i]);
            return SMC_ERR_INVALID;
        }
    if (t == 0 || M == 1) return SMC_OK;
    SMC_HIP_CHECK(hipSetDevice(f->ctx->device));
    hipStream_t st = f->ctx->stream;
    IslandArray arr[16];
    const int na = island_arrays(f, t, arr);
    i64 maxw = 0;
    for (int k = 0; k < na; ++k) maxw = arr[k].words > maxw ? arr[k].words : maxw;
    char* tmp = nullptr;
    const size_t stage = (size_t)M * maxw * 8, idxb = (size_t)M * 8;
    hipError_t e = hipMalloc((void**)&tmp, stage + idxb);
    if (e != hipSuccess) {
        smc_set_error("smc_filter_permute_islands: %zu bytes: %s", stage + idxb, hipGetErrorString(e));
        return SMC_ERR_NOMEM;
    }
    i64* src = (i64*)(tmp + stage);
    hipError_t rc = hipMemcpyAsync(src, src_host, idxb, hipMemcpyHostToDevice, st);
    for (int k = 0; k < na && rc == hipSuccess; ++k) {
        const i64 w = arr[k].words;
        const unsigned chunks = (unsigned)((w + SMC_BLOCK - 1) / SMC_BLOCK > 64 ? 64 : (w + SMC_BLOCK - 1) / SMC_BLOCK);
        SMC_LAUNCH(k_island_gather, dim3(chunks, M), dim3(SMC_BLOCK), st, (const u64*)arr[k].p, (u64*)tmp,
                   (const i64*)src, (const unsigned char*)nullptr, w);
        rc = hipMemcpyAsync(arr[k].p, tmp, (size_t)M * w * 8, hipMemcpyDeviceToDevice, st);
    }
    if (rc == hipSuccess) rc = hipGetLastError();
    if (rc == hipSuccess) rc = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    SMC_HIP_CHECK(rc);
    f->perm_t = t;
    return SMC_OK;
}

// PMCMC move of SMC^2 (smc_samplers.py:1129-1143): a second batch of filters was run on the
// proposed thetas; where the proposal is accepted, island i of `dst` takes over island i of `src`.
int smc_filter_copy_islands(smc_filter* dst, smc_filter* src, const unsigned char* accept_host)
{
    if (dst) { (void)hipSetDevice(dst->ctx->device); flush_rows(dst); }
    if (src) flush_rows(src);
    SMC_REQUIRE(dst && src && accept_host, "null argument");
    SMC_REQUIRE(dst->ctx == src->ctx, "both filters must live on one context");
    const FArgs &a = dst->a, &b = src->a;
    if (a.pm2 || b.pm2) {
        smc_set_error("whole-island moves are not available for the auxiliary filter on the two-level step");
        return SMC_ERR_STATE;
    }
    if (a.hist || b.hist) {
        smc_set_error("smc_filter_copy_islands: not available with keep_history");
        return SMC_ERR_STATE;
    }
    SMC_REQUIRE(a.N == b.N && a.T == b.T && a.n_islands == b.n_islands && a.dx == b.dx &&
                    dst->kind == src->kind && dst->fk == src->fk && dst->t_host == src->t_host &&
                    dst->two_level == src->two_level,
                "the two filters must have the same shape, model kind and time index");
    const i64 t = dst->t_host;
    if (t == 0) return SMC_OK;
    SMC_HIP_CHECK(hipSetDevice(dst->ctx->device));
    hipStream_t st = dst->ctx->stream;
    const int M = a.n_islands;
    unsigned char* mask = nullptr;
    SMC_HIP_CHECK(hipMalloc((void**)&mask, (size_t)M));
    hipError_t rc = hipMemcpyAsync(mask, accept_host, (size_t)M, hipMemcpyHostToDevice, st);
    IslandArray da[16], sa[16];
    const int na = island_arrays(dst, t, da);
    (void)island_arrays(src, t, sa);
    for (int k = 0; k < na && rc == hipSuccess; ++k) {
        const i64 w = da[k].words;
        const unsigned chunks = (unsigned)((w + SMC_BLOCK - 1) / SMC_BLOCK > 64 ? 64 : (w + SMC_BLOCK - 1) / SMC_BLOCK);
        SMC_LAUNCH(k_island_gather, dim3(chunks, M), dim3(SMC_BLOCK), st, (const u64*)sa[k].p, (u64*)da[k].p,
                   (const i64*)nullptr, (const unsigned char*)mask, w);
    }
    if (rc == hipSuccess) rc = hipGetLastError();
    if (rc == hipSuccess) rc = hipStreamSynchronize(st);
    (void)hipFree(mask);
    SMC_HIP_CHECK(rc);
    dst->perm_t = t;
    return SMC_OK;
}

// ---- island migration between GPUs (multi-GPU SMC^2: theta-particles sharded over ranks, a
// global theta-resampling moves whole filters; smc_comm_alltoallv carries the packed states) ----
int smc_filter_island_bytes(smc_filter* f, int64_t* bytes)
{
    SMC_REQUIRE(f && bytes, "null argument");
    IslandArray arr[16];
    const int na = island_arrays(f, f->t_host > 0 ? f->t_host : 1, arr);
    i64 w = 0;
    for (int k = 0; k < na; ++k) w += arr[k].words;
    *bytes = w * 8;
    return SMC_OK;
}

static int island_pack(smc_filter* f, const int64_t* islands_host, int n, void* pack_dev, int unpack)
{
    SMC_REQUIRE(f && (n == 0 || (islands_host && pack_dev)), "null argument");
    if (f->a.pm2) {
        smc_set_error("whole-island moves are not available for the auxiliary filter on the two-level step");
        return SMC_ERR_STATE;
    }
    if (f->a.hist) {
        smc_set_error("island migration is not available with keep_history");
        return SMC_ERR_STATE;
    }
    if (n == 0 || f->t_host == 0) return SMC_OK;
    for (int j = 0; j < n; ++j)
        SMC_REQUIRE(islands_host[j] >= 0 && islands_host[j] < f->a.n_islands, "island out of range");
    SMC_HIP_CHECK(hipSetDevice(f->ctx->device));
    hipStream_t st = f->ctx->stream;
    IslandArray arr[16];
    const int na = island_arrays(f, f->t_host, arr);
    i64 stride = 0;
    for (int k = 0; k < na; ++k) stride += arr[k].words;
    i64* idx = nullptr;
    SMC_HIP_CHECK(hipMalloc((void**)&idx, (size_t)n * 8));
    hipError_t rc = hipMemcpyAsync(idx, islands_host, (size_t)n * 8, hipMemcpyHostToDevice, st);
    i64 off = 0;
    for (int k = 0; k < na && rc == hipSuccess; ++k) {
        const i64 w = arr[k].words;
        const unsigned chunks = (unsigned)((w + SMC_BLOCK - 1) / SMC_BLOCK > 64 ? 64 : (w + SMC_BLOCK - 1) / SMC_BLOCK);
        SMC_LAUNCH(k_island_pack, dim3(chunks, n), dim3(SMC_BLOCK), st, (u64*)arr[k].p, w, (u64*)pack_dev, stride,
                   off, (const i64*)idx, unpack);
        off += w;
    }
    if (rc == hipSuccess) rc = hipGetLastError();
    if (rc == hipSuccess) rc = hipStreamSynchronize(st);
    (void)hipFree(idx);
    SMC_HIP_CHECK(rc);
    if (unpack) f->perm_t = f->t_host;
    return SMC_OK;
}
int smc_filter_pack_islands(smc_filter* f, const int64_t* islands_host, int n, void* pack_dev)
{
    if (f) { (void)hipSetDevice(f->ctx->device); flush_rows(f); }
    return island_pack(f, islands_host, n, pack_dev, 0);
}
int smc_filter_unpack_islands(smc_filter* f, const int64_t* islands_host, int n, const void* pack_dev)
{
    if (f) { (void)hipSetDevice(f->ctx->device); flush_rows(f); }
    return island_pack(f, islands_host, n, (void*)pack_dev, 1);
}

// A filter that has not stepped yet takes the time index t: its state at t - 1 is then whatever
// smc_filter_unpack_islands puts there (waste-free SMC^2 assembles its new population from the chains'
// batches this way: smc_samplers.py:669-684 keeps every intermediate state WITH its particle filter).
// Islands that receive no state must not be stepped.
int smc_filter_fast_forward(smc_filter* f, int64_t t)
{
    if (f) { (void)hipSetDevice(f->ctx->device); flush_rows(f); }
    SMC_REQUIRE(f, "null filter");
    SMC_REQUIRE(f->t_host == 0, "smc_filter_fast_forward: the filter has already stepped");
    SMC_REQUIRE(t >= 0 && t <= f->a.T, "smc_filter_fast_forward: t must be in [0, T]");
    SMC_REQUIRE(!f->a.hist && !f->sqmc && !f->lwth, "smc_filter_fast_forward: no history slots, SQMC or theta level");
    f->t_host = t;
    f->perm_t = t;
    return SMC_OK;
}

// ---- SMC^2: the theta level (see k_theta_update) -------------------------------------------
static int theta_enable(smc_filter* f, double ess_rmin, smc_comm* comm)
{
    if (f) { (void)hipSetDevice(f->ctx->device); flush_rows(f); }
    SMC_REQUIRE(f, "null filter");
    SMC_REQUIRE(!f->a.hist && !f->a.mom, "the theta level is not available with keep_history / moments");
    SMC_REQUIRE(!f->a.pm2, "the theta level is not available for the auxiliary filter on the two-level step");
    SMC_REQUIRE(!comm || comm->ctx == f->ctx, "the communicator belongs to another context");
    SMC_HIP_CHECK(hipSetDevice(f->ctx->device));
    const size_t M = (size_t)f->a.n_islands, T = (size_t)f->a.T;
    const size_t Ng = comm ? M * (size_t)comm->nranks : M;
    const size_t nb = (Ng + TH_STRIDE + 2 * T + (comm ? M + Ng : 0)) * 8;
    if (f->th_buf && (size_t)f->th_n != Ng) { (void)hipFree(f->th_buf); f->th_buf = nullptr; }
    if (!f->th_buf) SMC_HIP_CHECK(hipMalloc(&f->th_buf, nb));
    SMC_HIP_CHECK(hipMemsetAsync(f->th_buf, 0, nb, f->ctx->stream));
    f->lwth = (double*)f->th_buf;
    f->th = f->lwth + Ng;
    f->th_ess = f->th + TH_STRIDE;
    f->th_comm = comm;
    f->th_n = (int)Ng;
    f->th_send = comm ? f->th_ess + 2 * T : nullptr;
    f->th_recv = comm ? f->th_send + M : nullptr;
    f->th_ess_min = ess_rmin * (double)Ng;
    // (enabled on a batch that has already run t steps -- the exchange step: the theta weights
    //  start at zero, or at what smc_filter_theta_resume sets, and account for steps >= t)
    const double done = (double)f->t_host;
    SMC_HIP_CHECK(hipMemcpyAsync(f->th + 2, &done, 8, hipMemcpyHostToDevice, f->ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
    return SMC_OK;
}
int smc_filter_theta_enable(smc_filter* f, double ess_rmin) { return theta_enable(f, ess_rmin, nullptr); }

// The theta level of a population sharded over the ranks of `comm` (every rank: the same number of
// islands, rank r holds the global theta indices r M .. r M + M - 1): replicated on every rank, fed behind
// every step by ONE ncclAllGather of the ranks' evidence increments enqueued on the context's stream --
// the same arithmetic on the same N_theta values in the same order everywhere, hence the same decisions
// on every rank and for every world size, and no host synchronisation per step.  smc_filter_theta_state /
// _resume / _logmeans then speak of all nranks x n_islands theta-particles.  Every rank must make the same
// calls in the same order (smc_filter_step enqueues a collective per step).
int smc_filter_theta_enable_sharded(smc_filter* f, smc_comm* comm, double ess_rmin)
{
    SMC_REQUIRE(comm, "null communicator");
    return theta_enable(f, ess_rmin, comm);
}

int smc_filter_theta_state(smc_filter* f, double* lw_theta_host, int64_t* stop_t, int64_t* steps_done,
                           double* ess_host)
{
    if (f) { (void)hipSetDevice(f->ctx->device); flush_rows(f); }
    SMC_REQUIRE(f && f->lwth, "the theta level is not enabled");
    hipStream_t st = f->ctx->stream;
    double th[TH_STRIDE];
    SMC_HIP_CHECK(hipMemcpyAsync(th, f->th, sizeof th, hipMemcpyDeviceToHost, st));
    if (lw_theta_host)
        SMC_HIP_CHECK(hipMemcpyAsync(lw_theta_host, f->lwth, (size_t)f->th_n * 8, hipMemcpyDeviceToHost, st));
    SMC_HIP_CHECK(hipStreamSynchronize(st));
    if (stop_t) *stop_t = (int64_t)th[0];
    if (steps_done) *steps_done = (int64_t)th[2];
    if (ess_host && th[2] > 0) {
        SMC_HIP_CHECK(hipMemcpyAsync(ess_host, f->th_ess, (size_t)th[2] * 8, hipMemcpyDeviceToHost, st));
        SMC_HIP_CHECK(hipStreamSynchronize(st));
    }
    return SMC_OK;
}

// log-mean of the theta weights after every step accounted for (the outer SMC's log_mean_w,
// core.py:351-359: its differences between resamplings are the evidence increments of the model)
int smc_filter_theta_logmeans(smc_filter* f, double* out_host, int64_t* steps_done)
{
    if (f) { (void)hipSetDevice(f->ctx->device); flush_rows(f); }
    SMC_REQUIRE(f && f->lwth && out_host, "the theta level is not enabled, or null output");
    hipStream_t st = f->ctx->stream;
    double th[TH_STRIDE];
    SMC_HIP_CHECK(hipMemcpyAsync(th, f->th, sizeof th, hipMemcpyDeviceToHost, st));
    SMC_HIP_CHECK(hipStreamSynchronize(st));
    if (steps_done) *steps_done = (int64_t)th[2];
    if (th[2] > 0) {
        SMC_HIP_CHECK(hipMemcpyAsync(out_host, f->th_ess + f->a.T, (size_t)th[2] * 8, hipMemcpyDeviceToHost, st));
        SMC_HIP_CHECK(hipStreamSynchronize(st));
    }
    return SMC_OK;
}

// After a stop: new theta log-weights (null: zeros -- the outer resampling, core.py:299-305),
// the time records back to the stop step, the host's step count with them; the filter then
// continues from there.  Also valid when nothing stopped (lw_theta replaced, time unchanged).
int smc_filter_theta_resume(smc_filter* f, const double* lw_theta_host)
{
    if (f) { (void)hipSetDevice(f->ctx->device); flush_rows(f); }
    SMC_REQUIRE(f && f->lwth, "the theta level is not enabled");
    SMC_HIP_CHECK(hipSetDevice(f->ctx->device));
    hipStream_t st = f->ctx->stream;
    double th[TH_STRIDE];
    SMC_HIP_CHECK(hipMemcpyAsync(th, f->th, sizeof th, hipMemcpyDeviceToHost, st));
    SMC_HIP_CHECK(hipStreamSynchronize(st));
    if (lw_theta_host)
        SMC_HIP_CHECK(hipMemcpyAsync(f->lwth, lw_theta_host, (size_t)f->th_n * 8, hipMemcpyHostToDevice, st));
    else
        SMC_HIP_CHECK(hipMemsetAsync(f->lwth, 0, (size_t)f->th_n * 8, st));
    if (th[0] != 0.0) {
        SMC_LAUNCH(k_theta_thaw, dim3(1), dim3(SMC_BLOCK), st, f->a, f->th, th[0]);
        SMC_LAUNCH_CHECK();
        f->t_host = (i64)th[0];
    }
    SMC_HIP_CHECK(hipStreamSynchronize(st));
    return SMC_OK;
}

int smc_filter_moments(smc_filter* f, double* out_host)
{
    if (f) { (void)hipSetDevice(f->ctx->device); flush_rows(f); }
    SMC_REQUIRE(f && out_host, "null argument");
    if (!f->a.mom) {
        smc_set_error("smc_filter_moments: the filter was created without opts.moments");
        return SMC_ERR_STATE;
    }
    const i64 t = f->t_host, T = f->a.T;
    const size_t w = (size_t)2 * f->a.dx;
    for (int i = 0; i < f->a.n_islands && t > 0; ++i)
        SMC_HIP_CHECK(hipMemcpyAsync(out_host + (size_t)i * t * w, f->a.mom + (size_t)i * T * w,
                                     (size_t)t * w * 8, hipMemcpyDeviceToHost, f->ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
    return SMC_OK;
}

int smc_filter_history(smc_filter* f, int field, int64_t step, int island, void* out_host)
{
    if (f) { (void)hipSetDevice(f->ctx->device); flush_rows(f); }
    SMC_REQUIRE(f && out_host, "null argument");
    SMC_REQUIRE(island >= 0 && island < f->a.n_islands, "island out of range");
    if (!f->a.hist) {
        smc_set_error("smc_filter_history: the filter was created without keep_history");
        return SMC_ERR_STATE;
    }
    if (step < 0 || step >= f->t_host) {
        smc_set_error("smc_filter_history: step %lld has not run (t = %lld)", (long long)step,
                      (long long)f->t_host);
        return SMC_ERR_STATE;
    }
    if (f->a.hist >= 2) {          // rolling window: the f->a.hist most recent steps are resident
        const i64 oldest = f->t_host - f->a.hist;
        const bool needs_prev = field == SMC_FIELD_XP;         // Xp gathers from step - 1
        if (step < oldest || (needs_prev && step - 1 < oldest && step > 0)) {
            smc_set_error("smc_filter_history: step %lld has left the rolling window of %d steps",
                          (long long)step, f->a.hist);
            return SMC_ERR_STATE;
        }
    }
    return filter_fetch(f, field, step, island, out_host);
}

int smc_filter_spacings(smc_filter* f, int64_t t, int island, double* out_host)
{
    if (f) { (void)hipSetDevice(f->ctx->device); flush_rows(f); }
    SMC_REQUIRE(f && out_host, "null argument");
    SMC_REQUIRE(island >= 0 && island < f->a.n_islands, "island out of range");
    SMC_REQUIRE(t >= 1 && t < f->a.T, "resampling happens at steps 1 .. T - 1");
    if (f->a.scheme != SMC_MULTINOMIAL || f->a.ut || f->kind == SMC_MODEL_MVLINGAUSS) {
        smc_set_error("smc_filter_spacings: a univariate multinomial filter in production (Philox) mode only");
        return SMC_ERR_STATE;
    }
    SMC_HIP_CHECK(hipSetDevice(f->ctx->device));
    hipStream_t st = f->ctx->stream;
    SMC_LAUNCH(k_f_spacings_out, dim3(1), dim3(SMC_BLOCK), st, f->a, (i64)t, island, f->tmp);
    SMC_LAUNCH_CHECK();
    SMC_HIP_CHECK(hipMemcpyAsync(out_host, f->tmp, (size_t)f->a.N * 8, hipMemcpyDeviceToHost, st));
    SMC_HIP_CHECK(hipStreamSynchronize(st));
    return SMC_OK;
}

int smc_filter_trajectories(smc_filter* f, int island, int64_t* out_host)
{
    if (f) { (void)hipSetDevice(f->ctx->device); flush_rows(f); }
    SMC_REQUIRE(f && out_host, "null argument");
    SMC_REQUIRE(island >= 0 && island < f->a.n_islands, "island out of range");
    if (!f->a.hist) {
        smc_set_error("smc_filter_trajectories: the filter was created without keep_history");
        return SMC_ERR_STATE;
    }
    const i64 t = f->t_host, N = f->a.N, T = f->a.T;
    if (t == 0) {
        smc_set_error("smc_filter_trajectories: no step has run yet");
        return SMC_ERR_STATE;
    }
    hipStream_t st = f->ctx->stream;
    // which steps resampled (A_s = arange otherwise, core.py:336)
    std::vector<double> rows((size_t)t * SUMM_STRIDE);
    SMC_HIP_CHECK(hipMemcpyAsync(rows.data(), f->a.summ + (size_t)island * (T + 1) * SUMM_STRIDE,
                                 rows.size() * 8, hipMemcpyDeviceToHost, st));
    SMC_HIP_CHECK(hipStreamSynchronize(st));
    // rolling window: the genealogy of the resident steps only (RollingParticleHistory,
    // smoothing.py:209-219 over its deque); out_host then holds min(t, window) rows
    const i64 first = (f->a.hist >= 2 && t > f->a.hist) ? t - f->a.hist : 0;
    const i64 nrows = t - first;
    i64* B = nullptr;
    hipError_t e = hipMalloc((void**)&B, (size_t)nrows * N * 8);
    if (e != hipSuccess) {
        smc_set_error("smc_filter_trajectories: %zu bytes: %s", (size_t)nrows * N * 8, hipGetErrorString(e));
        return SMC_ERR_NOMEM;
    }
    const dim3 grid((unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK));
    SMC_LAUNCH(k_f_iota, grid, dim3(SMC_BLOCK), st, N, B + (size_t)(nrows - 1) * N);
    for (i64 s = t - 1; s >= first + 1; --s) {          // smoothing.py:213-216
        const u32* A = rows[(size_t)s * SUMM_STRIDE + 4] != 0.0 ? f_A(f->a, s) + (size_t)island * N : nullptr;
        SMC_LAUNCH(k_f_genealogy, grid, dim3(SMC_BLOCK), st, A, (const i64*)(B + (size_t)(s - first) * N), N,
                   B + (size_t)(s - first - 1) * N);
    }
    hipError_t e2 = hipMemcpyAsync(out_host, B, (size_t)nrows * N * 8, hipMemcpyDeviceToHost, st);
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
    (void)hipFree(B);
    SMC_HIP_CHECK(e2);
    return SMC_OK;
}

int smc_filter_one_trajectory(smc_filter* f, int island, int64_t n_last, double* out_host)
{
    if (f) { (void)hipSetDevice(f->ctx->device); flush_rows(f); }
    SMC_REQUIRE(f && out_host, "null argument");
    SMC_REQUIRE(island >= 0 && island < f->a.n_islands, "island out of range");
    if (f->a.hist != 1) {
        smc_set_error("smc_filter_one_trajectory: the filter was created without keep_history = 1");
        return SMC_ERR_STATE;
    }
    const i64 t = f->t_host;
    SMC_REQUIRE(t > 0 && n_last >= 0 && n_last < f->a.N, "no step has run, or particle index out of range");
    hipStream_t st = f->ctx->stream;
    double* buf = nullptr;
    hipError_t e = hipMalloc((void**)&buf, (size_t)t * f->a.dx * 8);
    if (e != hipSuccess) {
        smc_set_error("smc_filter_one_trajectory: %s", hipGetErrorString(e));
        return SMC_ERR_NOMEM;
    }
    const double* rows = f->a.summ + (size_t)island * (f->a.T + 1) * SUMM_STRIDE;
    SMC_LAUNCH(k_f_one_trajectory, dim3(1), dim3(64), st, f->a, island, (i64)n_last, t, rows, buf);
    hipError_t rc = hipMemcpyAsync(out_host, buf, (size_t)t * f->a.dx * 8, hipMemcpyDeviceToHost, st);
    if (rc == hipSuccess) rc = hipStreamSynchronize(st);
    (void)hipFree(buf);
    SMC_HIP_CHECK(rc);
    return SMC_OK;
}

int smc_filter_info(smc_filter* f, double* bytes_per_particle_step, int* kernels_per_step)
{
    SMC_REQUIRE(f, "null filter");
    if (bytes_per_particle_step) *bytes_per_particle_step = 16.0 * f->a.dx + 40.0;
    if (kernels_per_step) *kernels_per_step = (f->a.scheme == SMC_MULTINOMIAL && !f->a.ut) ? 5 : 3;
    return SMC_OK;
}

int smc_filter_profile(smc_filter* f, int enable)
{
    SMC_REQUIRE(f, "null filter");
    f->prof = enable != 0;
    f->prof_n = 0;
    if (f->prof && f->ev.empty()) {
        f->ev.resize(3 * PROF_MAX);
        for (auto& e : f->ev) SMC_HIP_CHECK(hipEventCreate(&e));
    }
    return SMC_OK;
}

int smc_filter_strict_stats(smc_filter* f, int32_t island, int64_t* exact_path, int64_t* exceptions)
{
    SMC_REQUIRE(f && exact_path && exceptions, "null argument");
    SMC_REQUIRE(f->strict && !f->strict_literal, "not a strict_ancestors filter");
    SMC_REQUIRE(island >= 0 && island < f->a.n_islands, "island out of range");
    const SqxArgs q = sqx_carve((void*)(f->strict_ws + 2 * (size_t)f->a.n_islands * f->a.N), f->a.N, f->a.n_islands);
    unsigned long long c[2] = {0ull, 0ull};
    SMC_HIP_CHECK(hipMemcpyAsync(c, q.ctr + (size_t)island * 4 + 2, 16, hipMemcpyDeviceToHost, f->ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
    *exact_path = (int64_t)c[0];
    *exceptions = (int64_t)c[1];
    return SMC_OK;
}

int smc_filter_describe(smc_filter* f, char* out, size_t n)
{
    SMC_REQUIRE(f && out && n > 0, "null argument");
    const bool mv = f->kind == SMC_MODEL_MVLINGAUSS;
    std::string s;
    if (small_filter_ok(f)) s = "k_filter_small";
    else if (f->sq_flat) {
        s = std::string(mv ? "smc_hilbert_sort" : "k_rs_sort") + "+k_sobol+k_sqmv_tapes+" + (f->fused ? "k_ancestors<fused>" : "k_prepare+k_ancestors") +
            "+k_sqmv_compose+" + (mv ? "k_propagate_mv [mv_chunks=" + std::to_string(f->a.mv_chunks) + "]" : std::string("k_propagate"));
    } else if (f->sqmc) {
        s = "k_rs_sort+k_sq_permute+k_reduce2+k_ancestors2+k_propagate";
        if (f->a.mom) s += "+k_f_moments_partials+k_f_moments_final";
    } else if (f->strict) {
        if (f->strict_literal) s = std::string(f->two_level ? "k_reduce2+" : "") + "k_strict_W+k_strict_cdf+k_strict_search_S+k_propagate";
        else if (f->two_level) s = std::string(f->two_level_mid ? "k_reduce2+" : "") +
                                   std::string("k_strict_classify+k_strict_search") + "+k_propagate";
        else s = "k_strict_W+k_sqx_classify+k_sqx_fill+k_strict_search_S+k_propagate";
    } else {
        if (f->two_level_mid) s = "k_reduce2+k_ancestors2";
        else if (f->two_level) s = f->wide_tpw ? "k_ancestors2w" : "k_ancestors2";
        else if (f->fused) s = "k_ancestors<fused>";
        else s = "k_prepare+k_ancestors";
        if (f->a.scheme == SMC_MULTINOMIAL && !f->a.ut)
            s = (f->a.sp_tpw ? "k_f_spacing_onepass+" : "k_f_spacing_sums+k_f_spacing_scan+k_f_spacing_write+") + s;
        if (f->sp_merge && f->a.scheme == SMC_MULTINOMIAL && !f->a.ut && f->two_level_mid) {
            const size_t p = s.find("+k_reduce2");
            if (p != std::string::npos) s.replace(0, p + 10, "k_f_spacing_onepass<with k_reduce2>");
        }
        if (mv && f->fk == SMC_FK_APF) s = "k_mv_aux+k_mv_aux_restate+" + s;
        s += mv ? (f->mv_collapsed ? "+k_propagate_mv<collapsed>" : "+k_propagate_mv") : "+k_propagate";
        if (mv) s += " [mv_chunks=" + std::to_string(f->a.mv_chunks) + "]" + (f->a.mv_diag ? " [diagonal factors]" : "");
        if (f->a.mom) s += "+k_f_moments_partials+k_f_moments_final";
    }
    {   // (islands of 1025 .. 4096 tiles: the reduction's launch is the 1024-thread kernel, launch_reduce2)
        const int nchunks = (f->a.nparts + 4 * SMC_BLOCK - 1) / (4 * SMC_BLOCK);
        const size_t p = s.find("k_reduce2+");
        if (p != std::string::npos && nchunks >= 2 && nchunks <= 4 && !f->a.pm2 && !f->reduce_narrow) s.replace(p, 10, "k_reduce2w+");
    }
    snprintf(out, n, "%s", s.c_str());
    return SMC_OK;
}

int smc_filter_kernel_ms(smc_filter* f, double* move_ms_avg, double* prepare_ms_avg,
                         int64_t* n_samples)
{
    SMC_REQUIRE(f && move_ms_avg && prepare_ms_avg && n_samples, "null argument");
    SMC_HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
    double whole = 0.0, pre = 0.0, post = 0.0;
    int nw = 0, np = 0, nq = 0;
    for (int k = 0; k < f->prof_n; ++k) {
        float a = 0.f;
        if (k % 3 == 1) {
            SMC_HIP_CHECK(hipEventElapsedTime(&a, f->ev[3 * k], f->ev[3 * k + 1]));
            pre += a; ++np;
        } else if (k % 3 == 2) {
            SMC_HIP_CHECK(hipEventElapsedTime(&a, f->ev[3 * k + 1], f->ev[3 * k + 2]));
            post += a; ++nq;
        } else {
            SMC_HIP_CHECK(hipEventElapsedTime(&a, f->ev[3 * k], f->ev[3 * k + 2]));
            whole += a; ++nw;
        }
    }
    *n_samples = f->prof_n;
    const double w = nw ? whole / nw : 0.0, p = np ? pre / np : 0.0, q = nq ? post / nq : 0.0;
    *move_ms_avg = (nw && np) ? w - p : 0.0;
    *prepare_ms_avg = (nw && nq) ? w - q : 0.0;
    f->prof_n = 0;
    return SMC_OK;
}

}  // extern "C"
