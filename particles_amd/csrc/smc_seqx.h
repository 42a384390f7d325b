// smc_seqx.h -- the reference's sequential fp64 CDF (resampling.py:500-509) in TWO launches, and the searches
// against it without ever writing it to memory.
//
// What is computed is what smc_seqsum.h describes: S_0 = W_0, S_j = fl(S_{j-1} + W_j) -- every rounding of the
// reference's loop -- from the observation that, while the running sum stays inside one binade, the chain is an integer
// sum on that binade's grid.  What changed (round 5) is how the work is cut:
//
//   launch 1, "classify" (one workgroup per tile of 1024 weights):
//     * an ESTIMATE of the running sum in front of every element (the tiles' sums come from the caller: the filter's
//       log-sum-exp partials, or a pass over W), within `margin` = (8192 + 8 ntiles) 2^-53 relative of the truth;
//     * every element becomes REGULAR on a grid (the estimate puts the sum before and after it inside one binade E,
//       margin included; r = W rounded to that grid, no tie), ZERO (adds nothing on any grid), EDGE-TINY (the estimate
//       is within the margin of a binade edge 2^(E+1-1023) but W is below half a unit of the FINER grid: it adds
//       nothing on either -- this is what keeps normalised weights, whose sum ends within a few ulps of 1.0, and
//       degenerate weight vectors on the fast path) or an EXCEPTION (binade crossings, ties, the head of the array);
//     * per element the tile-local inclusive prefix Pin_j of the roundings is stored (8 bytes), per tile the total,
//       per SEGMENT between two exceptions the interval of binades its elements accept (max of the lower, min of the
//       upper bounds), per exception (index, Pin, W, its tail segment's interval);
//     * the LAST workgroup to finish (tickets) then does the serial part for the island: prefix of the tile totals, the
//       exceptions sorted by index, and ONE thread walking them -- between two exceptions the chain is the integer
//       difference of P added on the grid of s (checked: s normal, the integer stays below 2^53), at the exception
//       s <- s + W with the hardware's own addition, whatever the rounding case; then, in parallel, every segment's
//       interval is VERIFIED against the binade the walk actually found in front of it, and every tile gets its header
//       (E_b, I_b): S_j = (I_b + Pin_j) 2^(E_b - 1075) for the elements in front of the tile's first exception.
//     Any violation (or more exceptions than the lists hold) sends the island down the exact path: the same workgroup
//     does every tile with seq_tile_block_exact (smc_seqsum.h) and writes S out -- milliseconds, bit-identical, and
//     reached only by inputs built for it (hundreds of engineered ties, NaN / negative weights).
//   launch 2, "search" (one workgroup per tile of parents): stages the tile's S_j in LDS from (header, Pin, the
//     tile's exceptions), finds the range of offspring the tile owns -- n with S_start < su_n <= S_end, a count with a
//     closed-form guess fixed by the definition for systematic / stratified draws, a 256-ary search of the sorted
//     uniforms for multinomial ones -- and gives each its ancestor by bisection in LDS: A_n = first j with
//     su_n <= S_j, the reference's strict `>` advance.  S never exists in HBM.  (k_sqx_fill writes it, for
//     smc_seq_prefix_sums and the tests.)
//
// Two things about this hardware that shaped the code: (1) vmcnt counts STORES as well as loads and retires in order --
// the next wait for ANY load (or an explicit drain) also waits for every store issued before it, and a store that has to
// reach memory (agent scope: the tiles' reports, the fused launch's words) takes about a microsecond to do so; such
// stores are therefore issued behind the last load their wave waits for (barriers themselves wait for LDS only);
// (2) one wave alone issues an instruction every few cycles at best -- the walk is as long as its
// instruction count, which is why it is done in floating point (sqx_chain).
//
// Why the result is the reference's, not an approximation: the estimate only PROPOSES a grid per element; everything
// the fast path assumes -- the binade of the true sum in front of every run of regular elements, the integer staying
// inside that binade to the end of the run -- is checked against the exactly walked sums before any ancestor is
// written, and the exact path redoes the island otherwise.
#pragma once
#include "smc_seqsum.h"

#define SQX_CAP 256                    /* exceptions per island the walk takes (8 KB of LDS in the chain's workgroup): a few dozen occur.
                                          (Round 6 tried 1024 -- four slots per thread of the chain: a likelihood over SORTED states
                                          rises through hundreds of binades -- and took it back: the classify kernel went from 4 to 3
                                          waves per SIMD, 37.3 -> 39.8 us per C2 step, and that shape's exceptions grow with N, every
                                          element of the rising flank being one: at 2^16 they overflow 1024 as well.) */
#define SQX_XPT (SQX_CAP / SMC_BLOCK)  /* exception slots per thread of the chain */
#define SQX_TCAP 254                   /* ... per tile */
#define SQX_CNT_STRIDE 16
#define SQX_CNT_WORDS (34 * SQX_CNT_STRIDE)

struct SqxArgs {
    i64 n;
    int ntiles;
    double margin;
    u64* Pin;                          // (islands, ntiles * 1024) tile-local inclusive prefixes of the roundings
    u64* Rt;                           // (islands, ntiles) tile totals
    u64* hseg;                         // (islands, ntiles) head segment: max lower bound | min upper bound << 16
    u64* Pt;                           // (islands, ntiles + 1) exclusive prefix of Rt, [ntiles] = total
    u64* xraw;                         // (islands, SQX_CAP, 4) exceptions as the tiles append them
    u64* xs;                           // (islands, SQX_CAP, 8) sorted: index, S bits, Pin
    int* xfirst;                       // (islands, ntiles + 1) first sorted exception at or behind each tile's start
    int* hE;                           // (islands, ntiles) header: biased exponent ...
    u64* hI;                           // ... and integer (implicit bit included) of the sum in front of the tile
    u64* ctr;                          // (islands, 4) exceptions appended | overflow | mode of the last run | its exceptions
    unsigned* tick;                    // (islands, SQX_CNT_WORDS) completion tickets
    double* Sfull;                     // (islands, n) the exact path's sums (mode 1)
    u64* trace;                        // SMC_TRACE builds: (2 ntiles + 8, 8) shader-clock stamps (rows: classify per tile,
                                       // 8 rows of the chain, search per tile), else null
};
#ifdef SMC_TRACE
#define SQX_STAMP(q, row, k) do { if ((q).trace && threadIdx.x == 0 && blockIdx.y == 0) (q).trace[(i64)(row) * 8 + (k)] = (u64)wall_clock64(); } while (0)
#else
#define SQX_STAMP(q, row, k) do { } while (0)
#endif
static inline SqxArgs sqx_carve(void* scratch, const i64 n, const int islands, size_t* bytes = nullptr, size_t* counters_at = nullptr,
                                size_t* counters_bytes = nullptr)
{
    const size_t nt = (size_t)((n + SEQ_TILE - 1) / SEQ_TILE), M = (size_t)islands;
    SqxArgs q;
    q.n = n;
    q.ntiles = (int)nt;
    q.margin = ldexp((double)(8192 + 8 * (i64)nt), -53);
    q.trace = nullptr;
    char* p = (char*)scratch;
    q.Pin = (u64*)p; p += M * nt * SEQ_TILE * 8;
    q.Sfull = (double*)p; p += M * ((size_t)n + 8) * 8;
    q.Rt = (u64*)p; p += M * nt * 8;
    q.hseg = (u64*)p; p += M * nt * 8;
    q.Pt = (u64*)p; p += M * (nt + 1) * 8;
    q.hI = (u64*)p; p += M * nt * 8;
    q.xraw = (u64*)p; p += M * SQX_CAP * 32;
    q.xs = (u64*)p; p += M * SQX_CAP * 64;
    if (counters_at) *counters_at = (size_t)(p - (char*)scratch);
    q.ctr = (u64*)p; p += M * 4 * 8;
    q.tick = (unsigned*)p; p += M * SQX_CNT_WORDS * 4;
    if (counters_bytes) *counters_bytes = M * (4 * 8 + SQX_CNT_WORDS * 4);
    q.xfirst = (int*)p; p += M * (nt + 1) * 4;
    q.hE = (int*)p; p += M * nt * 4;
    if (bytes) *bytes = (size_t)(p - (char*)scratch) + 64;
    return q;
}

// (the counters -- ctr, tick: one contiguous block -- must be zero before the first launch; the passes re-arm them)
static inline size_t sqx_scratch_bytes(const i64 n, const int islands)
{
    size_t b = 0;
    (void)sqx_carve((void*)0, n, islands, &b);
    return b;
}
static inline void sqx_zero_counters(hipStream_t st, void* scratch, const i64 n, const int islands)
{
    size_t at = 0, nb = 0;
    (void)sqx_carve(scratch, n, islands, nullptr, &at, &nb);
    (void)hipMemsetAsync((char*)scratch + at, 0, nb, st);
}
// ---- where the weights come from --------------------------------------------------------------------------------
// an array (the stand-alone operators)
struct SqxSrcArray {
    const double* W;
    i64 n;
    __device__ __forceinline__ void load4(const i64 i0, double (&w)[4]) const
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (i0 + k < n) ? W[i0 + k] : 0.0;
    }
};

// seq_round_to_grid with fewer instructions: (M + half) >> sh instead of quotient, remainder and compare -- the same
// integer except at an exact tie, which is flagged (and then an exception) either way
__device__ __forceinline__ u64 sqx_round_to_grid(const double W, const int Es, bool& tie, bool& big)
{
    const u64 bits = (u64)__double_as_longlong(W);
    const int e = (int)(bits >> 52);                           // (a sign bit makes e >= 0x800: `big` below)
    const u64 M = (bits & 0x000FFFFFFFFFFFFFull) | (e ? 0x0010000000000000ull : 0ull);
    int sh = Es - (e ? e : 1);
    tie = false;
    big = sh < 0 || e >= 0x7ff;                                // (inf / NaN / negative: never regular)
    if (big) return 0ull;
    if (sh == 0) return M;
    sh = sh > 63 ? 63 : sh;
    const u64 half = 1ull << (sh - 1);
    const u64 t = M + half;
    tie = (t & ((half << 1) - 1ull)) == 0ull;
    return t >> sh;
}
__device__ __forceinline__ u64 sqx_mant(const double s)       // the integer of s on its own grid (implicit bit included)
{
    const u64 b = (u64)__double_as_longlong(s);
    return (b & 0x000FFFFFFFFFFFFFull) | (((b >> 52) & 0x7ffull) ? 0x0010000000000000ull : 0ull);
}
__device__ __forceinline__ double sqx_pack(const int E, const u64 Iv)
{
    return __longlong_as_double((long long)(((u64)E << 52) | (Iv & 0x000FFFFFFFFFFFFFull)));
}
__device__ __forceinline__ bool sqx_last_block(unsigned* cnt, const int b, const int nblocks, int* s_flag)
{
    if (threadIdx.x == 0) {
        smc_drain_stores();
        const int shards = nblocks >= 64 ? 32 : 1;
        const int s = b & (shards - 1);
        const int size_s = nblocks / shards + (s < nblocks % shards ? 1 : 0);
        bool last = atomicAdd(cnt + (1 + s) * SQX_CNT_STRIDE, 1u) == (unsigned)(size_s - 1);
        if (last) last = atomicAdd(cnt, 1u) == (unsigned)(shards - 1);
        if (last)
            for (int i = 0; i <= shards; ++i) cnt[i * SQX_CNT_STRIDE] = 0u;
        *s_flag = last;
    }
    __syncthreads();
    return *s_flag != 0;
}


// ---- launch 1, per tile ------------------------------------------------------------------------------------------
// w: this thread's weights 4 tid .. 4 tid + 3 of tile b (0 beyond n); run0: the estimate of the running sum in front of
// the thread's first element.  Takes the workgroup's completion ticket (the Pin stores -- read by the next launch only --
// are issued behind it, so that the ticket does not wait for them) and returns whether this workgroup is the island's last.
__device__ __forceinline__ bool sqx_classify_tile(const double (&w)[4], const double run0, const int isl, const int b, const SqxArgs& q)
{
    __shared__ int s_flag;
    __shared__ u64 smu[SMC_NWAVE];
    __shared__ u32 smx[SMC_NWAVE];
    __shared__ u32 s_lo[SQX_TCAP + 2], s_hi[SQX_TCAP + 2];
    __shared__ u32 s_base;
    const int tid = (int)threadIdx.x, lane = smc_lane(), wave = smc_wave();
    const i64 j0 = (i64)b * SEQ_TILE, i0 = j0 + (i64)tid * 4;
    for (int i = tid; i < SQX_TCAP + 2; i += SMC_BLOCK) { s_lo[i] = 0u; s_hi[i] = 2047u; }      // (armed by the barrier below)
    const double dn = 1.0 - q.margin, up = 1.0 + q.margin;
    u64 r[4], rsum = 0ull;
    u32 aLo[4], aHi[4], nx = 0u;
    bool exc[4];
    double hi4[4];
    hi4[0] = run0 + w[0];
    hi4[1] = hi4[0] + w[1];
    hi4[2] = hi4[1] + w[2];
    hi4[3] = hi4[2] + w[3];
    // the common case decided once per thread: the estimate puts the sum in front of the thread's first element and
    // behind its last inside ONE binade (margins included) -- then every element of the thread sees that grid (the
    // estimate is monotone inside a thread), and what is left per element is the rounding itself
    const int E_a = seq_bexp(run0 * dn), E_b = seq_bexp(hi4[3] * up);
    if (run0 > 0.0 && E_a == E_b && E_a >= 1 && E_b < 0x7fe) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bool tie, big;
            const u64 rr = sqx_round_to_grid(w[k], E_a, tie, big);
            const bool zero = (u64)__double_as_longlong(w[k]) == 0ull;          // (+0.0: beyond n as well)
            exc[k] = i0 + k < q.n && !zero && (tie || big);
            r[k] = (zero || exc[k]) ? 0ull : rr;
            aLo[k] = (zero || exc[k]) ? 0u : (u32)E_a;
            aHi[k] = (zero || exc[k]) ? 2047u : (u32)E_a;
            rsum += r[k];
            nx += exc[k] ? 1u : 0u;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double lo = k ? hi4[k - 1] : run0, hi = hi4[k];
            const int E_lo = seq_bexp(lo * dn), E_hi = seq_bexp(hi * up);
            bool tie, big;
            const u64 rr = sqx_round_to_grid(w[k], E_lo >= 1 ? E_lo : 1, tie, big);
            const bool inside = i0 + k < q.n;
            const bool zero = (u64)__double_as_longlong(w[k]) == 0ull;
            const bool grid_ok = lo > 0.0 && E_lo >= 1 && E_hi < 0x7fe && !tie && !big;
            const bool regular = grid_ok && E_lo == E_hi;
            const bool edge = grid_ok && E_hi == E_lo + 1 && rr == 0ull;        // below half a unit of the finer grid
            exc[k] = inside && !zero && !regular && !edge;
            r[k] = (regular && !zero) ? rr : 0ull;
            aLo[k] = (zero || exc[k]) ? 0u : (u32)E_lo;
            aHi[k] = (zero || exc[k]) ? 2047u : (u32)E_hi;
            rsum += r[k];
            nx += exc[k] ? 1u : 0u;
        }
    }
    SQX_STAMP(q, b, 2);
    // tile-local prefixes of the roundings and of the exception count: one exchange
    const u64 rinc = smc_wave_scan_add_u64(rsum);
    const u32 xinc = smc_wave_scan_add_u32(nx);
    if (lane == 63) { smu[wave] = rinc; smx[wave] = xinc; }
    __syncthreads();
    u64 rbase = 0ull, rtot = 0ull;
    u32 xbase = 0u, xtot = 0u;
#pragma unroll
    for (int ww = 0; ww < SMC_NWAVE; ++ww) {
        if (ww < wave) { rbase += smu[ww]; xbase += smx[ww]; }
        rtot += smu[ww];
        xtot += smx[ww];
    }
    u64 Pin[4];
    u32 seg[4];
    {
        u64 p = rbase + rinc - rsum;
        u32 s = xbase + xinc - nx;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            p += r[k];
            Pin[k] = p;
            seg[k] = s;                                         // exceptions in front of element k
            s += exc[k] ? 1u : 0u;
        }
    }
    SQX_STAMP(q, b, 3);
    const bool tile_over = xtot > (u32)SQX_TCAP;
    // (the tile's slots in the island's exception list: a global atomic with a return value -- issued here, consumed
    //  after the segments' intervals are formed, so that its round trip overlaps them)
    u32 base_t0 = 0u;
    if (tid == 0 && xtot) base_t0 = (u32)atomicAdd(reinterpret_cast<unsigned long long*>(q.ctr + (i64)isl * 4), (unsigned long long)xtot);
    // the segments' intervals: slot = exceptions in front of the element (an exception's own slot is neutral).
    // A wave without an exception (nearly all of them) holds ONE segment: its 64 lanes' bounds are reduced on the DPP
    // path and one lane updates the slot -- 256 same-address LDS atomics per tile cost 1 us of the tile's 8.
    if (!tile_over) {
        const bool wave_plain = __ballot(nx != 0u ? 1 : 0) == 0ull;
        if (wave_plain) {
            u32 cl = aLo[0], ch = aHi[0];
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                cl = cl > aLo[k] ? cl : aLo[k];
                ch = ch < aHi[k] ? ch : aHi[k];
            }
            const u32 wl = smc_wave_scan_max_u32(cl), wh = smc_wave_scan_max_u32(2047u - ch);
            if (lane == 63 && (wl != 0u || wh != 0u)) { atomicMax(&s_lo[seg[0]], wl); atomicMin(&s_hi[seg[0]], 2047u - wh); }
        } else {
            u32 cl = aLo[0], ch = aHi[0], cs = seg[0] + (exc[0] ? 1u : 0u);
            // (an exception closes its segment: what follows it accumulates in slot seg + 1)
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const u32 sk = seg[k] + (exc[k] ? 1u : 0u);
                if (sk != cs) {
                    if (cl != 0u || ch != 2047u) { atomicMax(&s_lo[cs], cl); atomicMin(&s_hi[cs], ch); }
                    cl = aLo[k]; ch = aHi[k]; cs = sk;
                } else {
                    cl = cl > aLo[k] ? cl : aLo[k];
                    ch = ch < aHi[k] ? ch : aHi[k];
                }
            }
            if (cl != 0u || ch != 2047u) { atomicMax(&s_lo[cs], cl); atomicMin(&s_hi[cs], ch); }
        }
    }
    if (tid == 0) s_base = base_t0;
    __syncthreads();                                           // (the slots are final, s_base is set)
    // (every store that must reach memory before the ticket, issued together: the drain in front of the ticket then
    //  covers one round trip to memory, not two in a row)
    const u32 base = s_base;
    if (tid == 0) {
        if (tile_over || base + xtot > (u32)SQX_CAP) smc_st_agent(q.ctr + (i64)isl * 4 + 1, 1ull);
        smc_st_agent(q.Rt + (i64)isl * q.ntiles + b, rtot);
        smc_st_agent(q.hseg + (i64)isl * q.ntiles + b, (u64)s_lo[0] | ((u64)s_hi[0] << 16));
    }
    if (!tile_over && base + xtot <= (u32)SQX_CAP) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (exc[k]) {
                u64* e = q.xraw + ((i64)isl * SQX_CAP + base + seg[k]) * 4;
                smc_st_agent(e + 0, (u64)(i0 + k));
                smc_st_agent(e + 1, Pin[k]);
                smc_st_agent(e + 2, (u64)__double_as_longlong(w[k]));
                smc_st_agent(e + 3, (u64)s_lo[seg[k] + 1] | ((u64)s_hi[seg[k] + 1] << 16));
            }
    }
    SQX_STAMP(q, b, 4);
    if (xtot) {                                                // (uniform: the exceptions' words are other threads' stores --
        smc_drain_stores();                                    //  landed before thread 0 takes the ticket)
        __syncthreads();
    }
    const bool last = sqx_last_block(q.tick + (i64)isl * SQX_CNT_WORDS, b, q.ntiles, &s_flag);
    smc_st2g(q.Pin + (i64)isl * q.ntiles * SEQ_TILE + i0, Pin[0], Pin[1]);
    smc_st2g(q.Pin + (i64)isl * q.ntiles * SEQ_TILE + i0 + 2, Pin[2], Pin[3]);
    SQX_STAMP(q, b, 5);
    return last;
}

// ---- launch 1, the island's serial part: the last workgroup to arrive ----------------------------------------------
// One workgroup, on the critical path of the step: every global load it needs that does not depend on another is issued
// up front (the counters, the first 1024 tiles' totals and head intervals, every slot of the exception list), so that the
// whole function pays about three memory latencies; the walk reads (dP, W) pairs from LDS, branch-free.
// c_Pt: 8 KB of LDS the caller can spare during the chain (the first 1024 tiles' offsets; larger islands keep the rest in memory)
template <class Src>
__device__ __forceinline__ void sqx_chain(const Src& src, const int isl, const SqxArgs& q, u64* c_Pt)
{
    __shared__ u32 c_j[SQX_CAP];
    __shared__ u32 c_acc[SQX_CAP];
    __shared__ u64 c_P[SQX_CAP];                               // global P in front of each exception
    __shared__ double c_w[SQX_CAP];
    __shared__ double c_S[SQX_CAP];
    __shared__ u64 smu[SMC_SM];
    __shared__ int c_ok, s_idx;                                // c_ok: 1, or 0 with c_why saying which assumption failed
    __shared__ unsigned c_why;
    __shared__ double s_tmp;
    static_assert(SQX_CAP % SMC_BLOCK == 0, "whole exception slots per thread");
    const int tid = (int)threadIdx.x, ntiles = q.ntiles;
    u64* ctr = q.ctr + (i64)isl * 4;
    u64* Pt = q.Pt + (i64)isl * (ntiles + 1);
    SQX_STAMP(q, ntiles, 0);
    // ---- every independent load, at once
    const u64 cnt64 = smc_ld_agent(ctr), ovf = smc_ld_agent(ctr + 1);
    // (slot x of thread tid: exception x 256 + tid of the unsorted list; slot 0 -- all there is, as a rule -- requested
    //  here with everything else, the others when the count says there are any)
    u64 e0[SQX_XPT], e1[SQX_XPT], e2[SQX_XPT], e3[SQX_XPT];
    {
        const u64* ex = q.xraw + ((i64)isl * SQX_CAP + tid) * 4;
        e0[0] = smc_ld_agent(ex); e1[0] = smc_ld_agent(ex + 1); e2[0] = smc_ld_agent(ex + 2); e3[0] = smc_ld_agent(ex + 3);
    }
    u64 rt0[4], hs0[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int b = tid * 4 + k;
        rt0[k] = b < ntiles ? smc_ld_agent(q.Rt + (i64)isl * ntiles + b) : 0ull;
        const int bh = k * SMC_BLOCK + tid;                    // (the headers' mapping: tile 256 k + tid, see below)
        hs0[k] = bh < ntiles ? smc_ld_agent(q.hseg + (i64)isl * ntiles + bh) : (2047ull << 16);
    }
    __syncthreads();                                           // (every thread has read the counters)
    SQX_STAMP(q, ntiles, 1);
    if (tid == 0) { smc_st_agent(ctr, 0ull); smc_st_agent(ctr + 1, 0ull); c_ok = 1; c_why = 0u; }   // (re-armed for the next launch)
    bool slow = ovf != 0ull || cnt64 > (u64)SQX_CAP;
    if (slow && tid == 0) c_why = 1u;                          // more exceptions than the lists hold
    const int cnt = slow ? 0 : (int)cnt64;
#pragma unroll
    for (int x = 1; x < SQX_XPT; ++x) {
        e0[x] = e1[x] = e2[x] = e3[x] = 0ull;
        if (x * SMC_BLOCK + tid < cnt) {
            const u64* ex = q.xraw + ((i64)isl * SQX_CAP + x * SMC_BLOCK + tid) * 4;
            e0[x] = smc_ld_agent(ex); e1[x] = smc_ld_agent(ex + 1); e2[x] = smc_ld_agent(ex + 2); e3[x] = smc_ld_agent(ex + 3);
        }
    }
    if (!slow) {
        // ---- P offsets of the tiles: thread tid owns tiles 4 tid .. 4 tid + 3 of every chunk of 1024
        u64 carry = 0ull;
        for (int c0 = 0; c0 < ntiles; c0 += 4 * SMC_BLOCK) {
            u64 rt[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int b = c0 + tid * 4 + k;
                rt[k] = c0 == 0 ? rt0[k] : (b < ntiles ? smc_ld_agent(q.Rt + (i64)isl * ntiles + b) : 0ull);
            }
            u64 tot;
            u64 run = carry + smc_block_exscan_u64((rt[0] + rt[1]) + (rt[2] + rt[3]), smu, tot);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int b = c0 + tid * 4 + k;
                if (c0 == 0) c_Pt[tid * 4 + k] = run;
                else if (b < ntiles) Pt[b] = run;
                run += rt[k];
            }
            carry += tot;
            __syncthreads();
        }
        SQX_STAMP(q, ntiles, 2);
        // ---- the exceptions in order of index (rank by counting: a few dozen of them)
#pragma unroll
        for (int x = 0; x < SQX_XPT; ++x)
            if (x * SMC_BLOCK + tid < cnt) c_j[x * SMC_BLOCK + tid] = (u32)e0[x];
        __syncthreads();                                       // (c_j complete; Pt visible to the workgroup)
        int rank[SQX_XPT];
        u64 Pg[SQX_XPT];
#pragma unroll
        for (int x = 0; x < SQX_XPT; ++x) {
            rank[x] = 0;
            Pg[x] = 0ull;
            if (x * SMC_BLOCK + tid < cnt) {
                const int xb = (int)(e0[x] >> 10);
                Pg[x] = e1[x] + (xb < 4 * SMC_BLOCK ? c_Pt[xb] : Pt[xb]);
                const u32 mine = (u32)e0[x];
                for (int m = 0; m < cnt; ++m) rank[x] += c_j[m] < mine ? 1 : 0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < SQX_XPT; ++x)
            if (x * SMC_BLOCK + tid < cnt) {
                c_j[rank[x]] = (u32)e0[x];
                c_P[rank[x]] = Pg[x];
                c_w[rank[x]] = __longlong_as_double((long long)e2[x]);
                c_acc[rank[x]] = (u32)e3[x];
                // (the sorted list goes to memory behind the verdict: vmcnt counts stores too, and any wait for a load between
                //  here and there would also wait for these stores to land)
            }
        __syncthreads();
        SQX_STAMP(q, ntiles, 3);
        // ---- the walk (one thread): a run of regular elements is an integer added on the grid of s, an exception the
        // hardware's own addition (resampling.py:506-508).  pack(bexp(s), mant(s) + 0) == s for every s >= 0: no branch
        // on dP; the checks apply where dP != 0 and are collected off the chain of s.
        if (cnt <= 64) {
            // up to 64 exceptions (the rule: a few dozen): lane l of wave 0 holds exception l's (dP, W) in registers and
            // the wave walks in lockstep -- v_readlane feeds the chain of s, nothing on it waits for LDS
            if (tid < 64) {
                const int lane = tid;
                const u64 Pl = lane < cnt ? c_P[lane] : 0ull;
                const u64 dPl = Pl - ((lane > 0 && lane < cnt) ? c_P[lane - 1] : 0ull);
                const double wl = lane < cnt ? c_w[lane] : 0.0;
                const double dPd = (double)dPl;                // (exact below 2^53; beyond, the check below refuses the step)
                double s = 0.0, Sl = 0.0;
#ifndef SMC_EMULATE
                asm volatile("" : "+v"(s));                    // (a vector value: see the scalar-register note below)
#endif
                // (a single wave issues every instruction of the loop back to back: the integer form of the step --
                //  unpack, 64-bit add, repack, 28 instructions -- cost 56 ns per exception.  The same value in floating
                //  point: the run's integer times the unit in the last place of s, added to s -- EXACT whenever the step is
                //  legitimate (the integer stays below 2^53 on the grid of a normal s: the product and the sum are then
                //  representable), and where it is not the result does not matter: the checks of the walk's assumptions
                //  are made afterwards, by every lane for its own exception from the sum its left neighbour recorded,
                //  on the integers.  ulp(s) = (s with its fraction cleared) 2^-52, a multiplication the hardware does
                //  exactly down to the subnormals; s = 0 gives 0 and W[0] starts the chain as 0 + W[0] = W[0].)
                for (int i = 0; i < cnt; ++i) {
                    const double dP = smc_readlane_f64(dPd, i);
                    const double w = smc_readlane_f64(wl, i);
                    const double g = __longlong_as_double(__double_as_longlong(s) & 0x7ff0000000000000ll) * 0x1.0p-52;
                    s = fma(dP, g, s) + w;
                    Sl = lane == i ? s : Sl;
                }
                if (lane < cnt) c_S[lane] = Sl;
                // lane i: was the integer step in front of exception i legitimate?  (the sum in front of it: lane i - 1's)
                const double sprev = smc_dpp_f64<SMC_DPP_WAVE_SHR1, 0xf, false>(Sl);   // (lane 0: 0.0, nothing summed yet)
                bool bad = false;
                if (lane < cnt && dPl != 0ull) {
                    const int Ep = seq_bexp(sprev);
                    bad = Ep < 1 || Ep >= 0x7ff || sqx_mant(sprev) + dPl >= (1ull << 53);
                }
                const u64 dP = carry - (cnt ? smc_readlane64(Pl, cnt - 1) : 0ull);   // the run behind the last exception
                const int Es = seq_bexp(s);
                bad = bad || (dP != 0ull && !(Es >= 1 && Es < 0x7ff && sqx_mant(s) + dP < (1ull << 53)));
                if (bad) { c_ok = 0; atomicOr(&c_why, 2u); }   // the walk: an integer step left its binade
            }
        } else if (tid == 0) {
            double s = 0.0;
#ifndef SMC_EMULATE
            // (only lane 0 is here, and the compiler would keep the chain of s in scalar registers -- a
            //  v_readfirstlane and a handful of SALU operations between two v_add_f64, 95 ns per exception;
            //  as a vector value it is nine dependent VALU operations)
            asm volatile("" : "+v"(s));
#endif
            u64 Pprev = 0ull;
            u32 viol = 0u;
            const bool head = cnt > 0 && c_j[0] == 0u;
            u64 Pn = cnt > 0 ? c_P[0] : 0ull;                  // (the next exception's P and W: read one step ahead,
            double wn = cnt > 0 ? c_w[0] : 0.0;                //  so that no LDS latency sits on the chain of s)
            for (int i = 0; i < cnt; ++i) {
                const u64 Pi = Pn;
                const double w = wn;
                const int i1 = i + 1 < cnt ? i + 1 : i;
                Pn = c_P[i1];
                wn = c_w[i1];
                const u64 dP = Pi - Pprev;
                Pprev = Pi;
                const u64 sb = (u64)__double_as_longlong(s);
                const u32 hi = (u32)(sb >> 32), Es = hi >> 20;
                const u32 mh = (hi & 0xfffffu) | ((Es < 1u ? Es : 1u) << 20);
                const u64 Iv = (((u64)mh << 32) | (u32)sb) + dP;
                const u32 ih = (u32)(Iv >> 32);
                viol |= (dP != 0ull && (Es - 1u >= 0x7feu || (ih >> 21) != 0u)) ? 1u : 0u;
                s = __longlong_as_double((long long)(((u64)((Es << 20) | (ih & 0xfffffu)) << 32) | (u32)Iv)) + w;
                if (i == 0 && head) s = w;
                c_S[i] = s;
            }
            const u64 dP = carry - Pprev;                      // the run behind the last exception
            const int Es = seq_bexp(s);
            const bool ok = viol == 0u && (dP == 0ull || (Es >= 1 && Es < 0x7ff && sqx_mant(s) + dP < (1ull << 53)));
            if (!ok) { c_ok = 0; atomicOr(&c_why, 2u); }
        }
        __syncthreads();
        SQX_STAMP(q, ntiles, 4);
        // ---- verification of every segment against the binade the walk found in front of it; the tiles' headers
        bool bad = false;
        for (int xs_ = tid; xs_ < cnt; xs_ += SMC_BLOCK) {
            const u32 e = (u32)seq_bexp(c_S[xs_]);
            const bool b1 = e < (c_acc[xs_] & 0xffffu) || e > (c_acc[xs_] >> 16);
            bad = bad || b1;
            if (b1) atomicOr(&c_why, 4u);                      // the segment behind an exception expected another binade
        }
        // (measured and dropped, r14: waves 1-3 looking their tiles' first exceptions up WHILE lane 0 walks -- their LDS
        //  traffic sits in front of the walk's reads, and three waves do four waves' work afterwards: 20.8 against 18.5 us)
        // Thread tid takes tiles c0 + 256 k + tid, k = 0 .. 3: the lanes of a wave then write consecutive words of the header
        // arrays (whole cache lines).  The fused launch keeps the words in registers until every check has passed.
        u32 w0[4] = {0u, 0u, 0u, 0u}, w1[4] = {0u, 0u, 0u, 0u}, w2[4] = {0u, 0u, 0u, 0u};
        for (int c0 = 0; c0 < ntiles; c0 += 4 * SMC_BLOCK) {
            u64 hs[4], pt[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int b = c0 + k * SMC_BLOCK + tid;
                hs[k] = c0 == 0 ? hs0[k] : (b < ntiles ? smc_ld_agent(q.hseg + (i64)isl * ntiles + b) : (2047ull << 16));
                pt[k] = c0 == 0 ? c_Pt[k * SMC_BLOCK + tid] : (b < ntiles ? Pt[b] : 0ull);
            }
            // first exception with index >= each tile's start: four branch-free lower bounds side by side (the same
            // number of rounds for all, the LDS reads of a round independent)
            int lo[4] = {0, 0, 0, 0};
            u32 key[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) key[k] = (u32)(c0 + k * SMC_BLOCK + tid) << 10;      // (N < 2^32)
            for (int n = cnt; n > 1;) {
                const int half = n >> 1;
#pragma unroll
                for (int k = 0; k < 4; ++k) lo[k] = c_j[lo[k] + half - 1] < key[k] ? lo[k] + half : lo[k];
                n -= half;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int b = c0 + k * SMC_BLOCK + tid;
                if (b >= ntiles) continue;
                if (cnt > 0) lo[k] += c_j[lo[k]] < key[k] ? 1 : 0;
                const int qx = lo[k] - 1;
                const double sb = qx < 0 ? 0.0 : c_S[qx];
                const u32 e = (u32)seq_bexp(sb);
                const bool badh = e < (u32)(hs[k] & 0xffffull) || e > (u32)(hs[k] >> 16);
                if (badh) atomicOr(&c_why, 8u);                // a tile's head segment expected another binade
                bad = bad || badh;
                const u64 hI = sqx_mant(sb) + (pt[k] - (qx < 0 ? 0ull : c_P[qx]));
                if (c0 == 0) {                                 // (the first 1024 tiles' headers wait in registers for the verdict)
                    w0[k] = (e << 16) | (u32)lo[k];
                    w1[k] = (u32)hI;
                    w2[k] = (u32)(hI >> 32);
                } else {                                       // (larger islands, two launches: straight to memory)
                    q.xfirst[(i64)isl * (ntiles + 1) + b] = lo[k];
                    q.hE[(i64)isl * ntiles + b] = (int)e;
                    q.hI[(i64)isl * ntiles + b] = hI;
                }
            }
        }
        if (bad) c_ok = 0;                                     // (benign race: every writer stores 0)
        __syncthreads();
        slow = c_ok == 0;
        if (!slow) {
            if (tid == 0) q.xfirst[(i64)isl * (ntiles + 1) + ntiles] = cnt;
#pragma unroll
            for (int x = 0; x < SQX_XPT; ++x)
                if (x * SMC_BLOCK + tid < cnt) {               // (this thread's exceptions: slots `rank` of the sorted list)
                    u64* o = q.xs + ((i64)isl * SQX_CAP + rank[x]) * 8;
                    o[0] = e0[x] & 0xffffffffull;
                    o[1] = (u64)__double_as_longlong(c_S[rank[x]]);
                    o[2] = e1[x];
                }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int b = k * SMC_BLOCK + tid;
                if (b >= ntiles) continue;
                q.xfirst[(i64)isl * (ntiles + 1) + b] = (int)(w0[k] & 0xffffu);
                q.hE[(i64)isl * ntiles + b] = (int)(w0[k] >> 16);
                q.hI[(i64)isl * ntiles + b] = (u64)w1[k] | ((u64)w2[k] << 32);
            }
        }
    }
    if (slow) {
        // ---- the exact path: every tile by the workgroup, scan-until-exception (smc_seqsum.h)
        double* So = q.Sfull + (i64)isl * (q.n + 8);
        double s = 0.0;
        bool first = true;
        for (i64 lo = 0; lo < q.n; lo += SEQ_TILE) {
            const int m_all = (int)(lo + SEQ_TILE < q.n ? SEQ_TILE : q.n - lo);
            double w4[4], o4[4] = {0.0, 0.0, 0.0, 0.0};
            src.load4(lo + (i64)tid * 4, w4);
            s = seq_tile_block_exact(w4, o4, m_all, s, first, smu, &s_idx, &s_tmp);
            first = false;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (tid * 4 + k < m_all) So[lo + tid * 4 + k] = o4[k];
        }
    }
    __syncthreads();
    if (tid == 0) { ctr[2] = slow ? (u64)(c_why ? c_why : 16u) : 0ull; ctr[3] = cnt64; }   // (mode word: 0, or why the exact path ran)
    SQX_STAMP(q, ntiles, 5);
}

// ---- launch 2: a tile's sums in LDS ------------------------------------------------------------------------------
// What a workgroup needs of tile b, requested in one go (sqx_stage_load: nothing here depends on the step record, so
// the filter's kernel issues these loads together with the record's) ...
struct SqxStage {
    u64 mode, hI, Pin[4];
    int hE, xf0, xf1;
};
__device__ __forceinline__ SqxStage sqx_stage_load(const SqxArgs& q, const int isl, const int b)
{
    SqxStage st;
    const i64 i0 = (i64)b * SEQ_TILE + (i64)threadIdx.x * 4;
    st.mode = smc_ldg(q.ctr + (i64)isl * 4 + 2);
    st.hE = q.hE[(i64)isl * q.ntiles + b];
    st.hI = smc_ldg(q.hI + (i64)isl * q.ntiles + b);
    st.xf0 = q.xfirst[(i64)isl * (q.ntiles + 1) + b];
    st.xf1 = q.xfirst[(i64)isl * (q.ntiles + 1) + b + 1];
    smc_ld2g(q.Pin + (i64)isl * q.ntiles * SEQ_TILE + i0, st.Pin[0], st.Pin[1]);
    smc_ld2g(q.Pin + (i64)isl * q.ntiles * SEQ_TILE + i0 + 2, st.Pin[2], st.Pin[3]);
    return st;
}
// ... and sS[0 .. 1023] <- S_j of tile b (beyond n: the last sum); returns the sum in front of the tile (-inf for
// tile 0).  Every thread must call; ends with a barrier.
__device__ __forceinline__ double sqx_stage_tile(const SqxArgs& q, const int isl, const int b, const SqxStage& ld, double* sS)
{
    __shared__ u64 s_x[3 * 64];                                // a batch of the tile's exceptions (index, S bits, Pin)
    const int tid = (int)threadIdx.x;
    const i64 j0 = (i64)b * SEQ_TILE, i0 = j0 + (i64)tid * 4;
    const u64 mode = smc_uniform_u64(ld.mode);
    double S_start;
    if (mode != 0ull) {
        const double* Sf = q.Sfull + (i64)isl * (q.n + 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) sS[tid * 4 + k] = Sf[i0 + k < q.n ? i0 + k : q.n - 1];
        S_start = b ? Sf[j0 - 1] : -INFINITY;
    } else {
        const int hE = (int)smc_uniform_u64((u64)ld.hE);
        const u64 hI = smc_uniform_u64(ld.hI);
        const int xf0 = (int)smc_uniform_u64((u64)ld.xf0), xf1 = (int)smc_uniform_u64((u64)ld.xf1);
        int bE[4];
        u64 bI[4], bP[4];
        double fin[4];
        bool isfin[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { bE[k] = hE; bI[k] = hI; bP[k] = 0ull; fin[k] = 0.0; isfin[k] = false; }
        for (int e0 = xf0; e0 < xf1; e0 += 64) {               // (uniform trip counts: the tile's exceptions, in order)
            const int ne = xf1 - e0 < 64 ? xf1 - e0 : 64;
            __syncthreads();
            if (tid < 3 * ne) {
                s_x[tid] = smc_ldg(q.xs + ((i64)isl * SQX_CAP + e0 + tid / 3) * 8 + tid % 3);
            }
            __syncthreads();
            for (int e = 0; e < ne; ++e) {
                const i64 jx = (i64)s_x[3 * e];
                const double Sx = __longlong_as_double((long long)s_x[3 * e + 1]);
                const u64 Px = s_x[3 * e + 2];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (i0 + k > jx) { bE[k] = seq_bexp(Sx); bI[k] = sqx_mant(Sx); bP[k] = Px; }
                    if (i0 + k == jx) { fin[k] = Sx; isfin[k] = true; }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) sS[tid * 4 + k] = isfin[k] ? fin[k] : sqx_pack(bE[k], bI[k] + (ld.Pin[k] - bP[k]));
        S_start = b ? sqx_pack(hE, hI) : -INFINITY;
    }
    __syncthreads();
    return S_start;
}

// #{ n : su_n <= x } for the closed-form schemes: a guess, fixed with the definition (su_n as the reference forms it)
__device__ inline i64 sqx_count_le(const SmcSu& s, const double x)
{
    if (!(x >= 0.0)) return 0;
    i64 lo = 0, hi = s.M;       // invariant: su_n <= x on [0, lo), su_n > x on [hi, M)
    const double r = x * s.dM - (s.scheme == SMC_SYSTEMATIC_ ? s.u_sys : 0.0);
    i64 g = (r < 0.0) ? 0 : (r >= s.dM ? s.M : (i64)r + 1);
    g = g > s.M ? s.M : g;
    int it = 0;
    while (g < s.M && it < 4 && smc_su_at(s, g) <= x) { ++g; ++it; }
    if (it < 4) {
        int jt = 0;
        while (g > 0 && jt < 4 && !(smc_su_at(s, g - 1) <= x)) { --g; ++jt; }
        if (jt < 4) return g;
        hi = g;
    } else {
        lo = g;
    }
    while (lo < hi) {
        const i64 mid = lo + ((hi - lo) >> 1);
        if (smc_su_at(s, mid) <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// the same by a whole wave (every lane calls, x uniform across the wave): lane l evaluates the definition at
// guess - 32 + l -- one division per lane, side by side -- and a ballot counts; the sequential form above only if
// the window does not bracket the answer (it does: the guess is off by a unit at most)
__device__ __forceinline__ i64 sqx_count_le_wave(const SmcSu& s, const double x)
{
    if (!(x >= 0.0)) return 0;
    const double r = x * s.dM - (s.scheme == SMC_SYSTEMATIC_ ? s.u_sys : 0.0);
    i64 g = (r < 0.0) ? 0 : (r >= s.dM ? s.M : (i64)r + 1);
    g = g > s.M ? s.M : g;
    const i64 n = g - 32 + smc_lane();
    const bool in = n >= 0 && n < s.M;
    const bool le = in ? smc_su_at(s, in ? n : 0) <= x : n < 0;
    const u64 mask = __ballot(le ? 1 : 0);
    const bool bracketed = (mask & 1ull) != 0ull && (mask >> 63) == 0ull;      // true at the low end, false at the high end
    if (bracketed && (mask & (mask + 1ull)) == 0ull) return g - 32 + (i64)__popcll(mask);
    return sqx_count_le(s, x);
}
// the same for sorted uniforms in memory, two thresholds at once: a 256-ary search by the workgroup (three rounds at
// M = 2^22 instead of 22 dependent loads).  Every thread must call; the results are the same in every thread.
__device__ __forceinline__ void sqx_count_le_sorted2(const SmcSu& su, const double xa, const double xb, i64& na, i64& nb)
{
    const i64 M = su.M;
    __shared__ u32 s_c[2 * SMC_NWAVE];
    const int tid = (int)threadIdx.x, lane = smc_lane(), wave = smc_wave();
    i64 lo[2] = {0, 0}, hi[2] = {M, M};
    const double x[2] = {xa, xb};
    if (!(xa >= 0.0)) hi[0] = 0;
    if (!(xb >= 0.0)) hi[1] = 0;
    while (lo[0] < hi[0] || lo[1] < hi[1]) {
        u32 c[2];
        i64 chunk[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const i64 span = hi[v] - lo[v];
            chunk[v] = span > 0 ? (span + SMC_BLOCK - 1) / SMC_BLOCK : 1;
            const i64 idx = lo[v] + (i64)tid * chunk[v];
            const bool t = idx < hi[v] && smc_su_at(su, idx) <= x[v];
            c[v] = (u32)__popcll(__ballot(t ? 1 : 0));
        }
        __syncthreads();
        if (lane == 0) { s_c[wave] = c[0]; s_c[SMC_NWAVE + wave] = c[1]; }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            u32 tot = 0u;
#pragma unroll
            for (int ww = 0; ww < SMC_NWAVE; ++ww) tot += s_c[v * SMC_NWAVE + ww];
            if (lo[v] >= hi[v]) continue;
            if (tot == 0u) { hi[v] = lo[v]; continue; }
            const i64 nlo = lo[v] + (i64)(tot - 1u) * chunk[v] + 1;
            const i64 nhi = lo[v] + (i64)tot * chunk[v];
            hi[v] = nhi < hi[v] ? nhi : hi[v];
            lo[v] = nlo;
        }
    }
    na = lo[0];
    nb = lo[1];
}
// first k in [0, m) with x <= S[k] (m - 1 if none), for eight thresholds side by side: branch-free lower bounds with
// the same number of rounds, the eight LDS reads of a round independent of one another
__device__ __forceinline__ void sqx_first_ge_lds8(const double* S, const int m, const double (&x)[8], int (&out)[8])
{
    int lo[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // (measured, r14: the ten rounds of a whole tile unrolled with compile-time steps -- immediate offsets in the LDS
    //  reads -- run 3.2 us against 2.1 for this loop: the unrolled form serialises on its waits)
    for (int n = m; n > 1;) {
        const int half = n >> 1;
#pragma unroll
        for (int k = 0; k < 8; ++k) lo[k] = S[lo[k] + half - 1] < x[k] ? lo[k] + half : lo[k];
        n -= half;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        lo[k] += S[lo[k]] < x[k] ? 1 : 0;
        out[k] = lo[k] < m ? lo[k] : m - 1;
    }
}
// ---- systematic draws, N < 2^30: the offspring of a tile WITHOUT a search.  su_n = fl(fl(u + n) / N) is monotone in
// n, so parent j owns the offspring n_{j-1} <= n < n_j with n_j = #{ n : su_n <= S_j } -- a count with a closed-form
// guess, fixed with the definition itself (two evaluations: four instructions each) -- and the tile's ancestors are
// one scatter of the parents' indices at their first offspring plus a running maximum over a window of offspring
// (k_ancestors2's shape): ~140 vector instructions per thread where eight bisections of ten rounds took ~480.
// Branch-free: with r = S N - u the count is floor(r) + 1 up to ONE unit either way -- r and the definition's two
// roundings are each within N 2^-51 <= 2^-21 of their exact values for N < 2^30 -- so two evaluations of the definition,
// at g - 1 and at g, decide among g - 1, g and g + 1.
__device__ __forceinline__ int sqx_sys_count(const SmcSu& su, const int N, const double S)
{
    const double r = S * su.dM - su.u_sys;
    int g = r < 0.0 ? 0 : (r >= su.dM ? N : (int)r + 1);       // (S = -inf, NaN: 0)
    g = g > N ? N : g;
    const bool le0 = g < 1 || smc_su_div(su, su.u_sys + (double)(g - 1)) <= S;
    const bool le1 = g < N && smc_su_div(su, su.u_sys + (double)g) <= S;
    const int c = g - 1 + (le0 ? 1 : 0) + (le1 ? 1 : 0);
    return S >= 0.0 ? c : 0;
}
template <class AT>
__device__ __forceinline__ void sqx_scatter_tile(const SqxArgs& q, const int b, const SmcSu& su, const double* sS, const double S_start, AT* A)
{
    constexpr int WIN = 8 * SMC_BLOCK;
    __shared__ __attribute__((aligned(16))) u32 sP[WIN];
    __shared__ u32 s_wm[SMC_NWAVE];
    __shared__ u32 s_carry;
    const int tid = (int)threadIdx.x, lane = smc_lane(), wave = smc_wave();
    const i64 j0 = (i64)b * SEQ_TILE;
    const int m_all = (int)(j0 + SEQ_TILE < q.n ? SEQ_TILE : q.n - j0);
    const int N = (int)su.M;
    const bool last = b == q.ntiles - 1;
    // n_j of this thread's four parents and of the one in front of them
    int nj[5];
    {
        const int l0 = tid * 4;
        const double Sp = l0 == 0 ? S_start : sS[l0 - 1];
        nj[0] = (b == 0 && l0 == 0) ? 0 : sqx_sys_count(su, N, Sp);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int l = l0 + k;
            nj[k + 1] = (last && l >= m_all - 1) ? N : sqx_sys_count(su, N, sS[l < m_all ? l : m_all - 1]);
        }
    }
    // the tile's range: the first thread's nj[0], the last thread's nj[4]
    if (tid == 0) sP[0] = (u32)nj[0];
    if (tid == SMC_BLOCK - 1) sP[1] = (u32)nj[4];
    __syncthreads();
    const int n_lo = (int)sP[0], n_hi = (int)sP[1];
    __syncthreads();
    SQX_STAMP(q, q.ntiles + 8 + b, 3);
    for (int base = n_lo; base < n_hi; base += WIN) {
        *reinterpret_cast<uint4*>(&sP[tid * 8]) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(&sP[tid * 8 + 4]) = make_uint4(0u, 0u, 0u, 0u);
        if (tid == 0) s_carry = 0u;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (nj[k + 1] <= nj[k]) continue;                      // no offspring
            const int first = nj[k] - base;
            if (first >= 0 && first < WIN) sP[first] = (u32)(tid * 4 + k + 1);
            else if (first < 0 && nj[k + 1] > base) s_carry = (u32)(tid * 4 + k + 1);   // the parent that straddles the window's start
        }
        __syncthreads();
        u32 v[8];
        {
            const uint4 a4 = *reinterpret_cast<const uint4*>(&sP[tid * 8]), b4 = *reinterpret_cast<const uint4*>(&sP[tid * 8 + 4]);
            v[0] = a4.x; v[1] = a4.y; v[2] = a4.z; v[3] = a4.w; v[4] = b4.x; v[5] = b4.y; v[6] = b4.z; v[7] = b4.w;
        }
#pragma unroll
        for (int i = 1; i < 8; ++i) v[i] = v[i] > v[i - 1] ? v[i] : v[i - 1];
        const u32 inc = smc_wave_scan_max_u32(v[7]);
        u32 ex = smc_mov_dpp<SMC_DPP_WAVE_SHR1>(inc);
        if (lane == 0) ex = 0u;
        if (lane == 63) s_wm[wave] = inc;
        __syncthreads();
        u32 pre = s_carry;
        for (int w = 0; w < wave; ++w) pre = pre > s_wm[w] ? pre : s_wm[w];
        pre = pre > ex ? pre : ex;
        const int lim = n_hi - base;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const u32 m = v[i] > pre ? v[i] : pre;
            if (tid * 8 + i < lim) A[(i64)base + tid * 8 + i] = (AT)(j0 + (i64)m - 1);
        }
        __syncthreads();
    }
    SQX_STAMP(q, q.ntiles + 8 + b, 4);
}

// the offspring of tile b: n with S_start < su_n <= S_end (the last tile takes every offspring left: the reference
// would run off the end of W there), each written with its ancestor.  AT: u32 (the filter) or i64 (the operator).
// sS: the tile's sums, staged (sqx_stage_tile).
template <class AT>
__device__ __forceinline__ void sqx_search_tile(const SqxArgs& q, const int b, const SmcSu& su, const double* sS, const double S_start, AT* A)
{
    if (su.scheme == SMC_SYSTEMATIC_ && su.M < ((i64)1 << 30) && q.n < ((i64)1 << 30)) {
        sqx_scatter_tile<AT>(q, b, su, sS, S_start, A);            // (uniform branch: no search at all)
        return;
    }
    __shared__ i64 s_n[2];
    const int tid = (int)threadIdx.x;
    const i64 j0 = (i64)b * SEQ_TILE;
    const int m_all = (int)(j0 + SEQ_TILE < q.n ? SEQ_TILE : q.n - j0);
    const double S_end = sS[m_all - 1];
    const bool last = b == q.ntiles - 1;
    i64 n_lo, n_hi;
    if (su.scheme == SMC_MULTINOMIAL_) {
        sqx_count_le_sorted2(su, S_start, S_end, n_lo, n_hi);
    } else {
        // wave 0: the offspring in front of the tile; wave 1: those up to its end
        if (smc_wave() < 2) {
            const i64 c = sqx_count_le_wave(su, smc_wave() == 0 ? S_start : S_end);
            if (smc_lane() == 0) s_n[smc_wave()] = c;
        }
        __syncthreads();
        n_lo = s_n[0];
        n_hi = s_n[1];
    }
    if (b == 0) n_lo = 0;
    if (last) n_hi = su.M;
    SQX_STAMP(q, q.ntiles + 8 + b, 3);
    // passes of 2048 offspring, two groups of four consecutive ones per thread (pairs: a Philox call each under
    // stratified draws), the eight bisections of a thread side by side
    for (i64 n0 = (n_lo & ~(i64)3) + (i64)tid * 4; n0 < n_hi; n0 += 8 * SMC_BLOCK) {
        double x[8];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const i64 ng = n0 + (i64)g * 4 * SMC_BLOCK;
            if (su.scheme == SMC_MULTINOMIAL_) {
#pragma unroll
                for (int k = 0; k < 4; ++k) x[4 * g + k] = ng + k < su.M ? smc_su_at(su, ng + k) : 2.0;
            } else if (su.scheme == SMC_SYSTEMATIC_ && su.M < ((i64)1 << 30)) {
                // fl(fl(u + n) / M) with a 32-bit conversion of n (the same double as the 64-bit one)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    x[4 * g + k] = ng + k < su.M ? smc_su_div(su, su.u_sys + (double)(int)(ng + k)) : 2.0;
            } else {
                smc_su_pair(su, ng >> 1, x[4 * g], x[4 * g + 1]);
                smc_su_pair(su, (ng >> 1) + 1, x[4 * g + 2], x[4 * g + 3]);
            }
        }
        SQX_STAMP(q, q.ntiles + 8 + b, 5);
        int a8[8];
        sqx_first_ge_lds8(sS, m_all, x, a8);
        SQX_STAMP(q, q.ntiles + 8 + b, 6);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const i64 n = n0 + (i64)g * 4 * SMC_BLOCK + k;
                if (n >= n_lo && n < n_hi) A[n] = (AT)(j0 + a8[4 * g + k]);
            }
    }
    SQX_STAMP(q, q.ntiles + 8 + b, 4);
}

// ---- the stand-alone operators' kernels (W an array; the tiles' sums by k_seq_tile_sums) ------------------------------
static __global__ void __launch_bounds__(SMC_BLOCK)
k_sqx_classify(const double* W, const double* tsum, const SqxArgs q, const SeqGate gate)
{
    __shared__ double smd[SMC_SM];
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y, tid = (int)threadIdx.x;
    if (!seq_gate_open(gate, isl)) return;
    const double* ts = tsum + (i64)isl * q.ntiles;
    double before = 0.0;
    for (int j = tid; j < b; j += SMC_BLOCK) before += ts[j];
    before = smc_block_sum(before, smd);
    __syncthreads();
    const SqxSrcArray src{W + (i64)isl * q.n, q.n};
    double w4[4], tot;
    src.load4((i64)b * SEQ_TILE + (i64)tid * 4, w4);
    const double run0 = before + smc_block_exscan_pos_f64((w4[0] + w4[1]) + (w4[2] + w4[3]), smd, tot);
    __syncthreads();
    __shared__ u64 c_Pt[SEQ_TILE];
    if (sqx_classify_tile(w4, run0, isl, b, q)) sqx_chain(src, isl, q, c_Pt);
}
static __global__ void __launch_bounds__(SMC_BLOCK)
k_sqx_fill(const SqxArgs q, double* S, const SeqGate gate)
{
    __shared__ double sS[SEQ_TILE];
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y, tid = (int)threadIdx.x;
    if (!seq_gate_open(gate, isl)) return;
    const SqxStage ld = sqx_stage_load(q, isl, b);
    (void)sqx_stage_tile(q, isl, b, ld, sS);
    const i64 i0 = (i64)b * SEQ_TILE + (i64)tid * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < q.n) S[(i64)isl * q.n + i0 + k] = sS[tid * 4 + k];
}
static __global__ void __launch_bounds__(SMC_BLOCK)
k_sqx_search_sorted(const SqxArgs q, const double* su_sorted, const i64 M, i64* A)
{
    SmcSu su;
    su.scheme = SMC_MULTINOMIAL_;
    su.M = M;
    su.dM = (double)M;
    su.u = su_sorted;
    su.u_sys = 0.0;
    su.seed = 0ull;
    su.t = 0u;
    su.island = 0u;
    __shared__ double sS[SEQ_TILE];
    const SqxStage ld = sqx_stage_load(q, (int)blockIdx.y, (int)blockIdx.x);
    const double S_start = sqx_stage_tile(q, (int)blockIdx.y, (int)blockIdx.x, ld, sS);
    sqx_search_tile<i64>(q, (int)blockIdx.x, su, sS, S_start, A);
}
