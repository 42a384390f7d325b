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
// Why the result is the reference's, not an approximation: the estimate only PROPOSES a grid per element; everything
// the fast path assumes -- the binade of the true sum in front of every run of regular elements, the integer staying
// inside that binade to the end of the run -- is checked against the exactly walked sums before any ancestor is
// written, and the exact path redoes the island otherwise.
#pragma once
#include "smc_seqsum.h"

#define SQX_CAP 512                    /* exceptions per island the walk takes (16 KB of LDS) */
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
    u64* xs;                           // (islands, SQX_CAP, 4) sorted: index, S bits, Pin, -
    int* xfirst;                       // (islands, ntiles + 1) first sorted exception at or behind each tile's start
    int* hE;                           // (islands, ntiles) header: biased exponent ...
    u64* hI;                           // ... and integer (implicit bit included) of the sum in front of the tile
    u64* ctr;                          // (islands, 4) exceptions appended | overflow | mode of the last run | its exceptions
    unsigned* tick;                    // (islands, SQX_CNT_WORDS) completion tickets
    double* Sfull;                     // (islands, n) the exact path's sums (mode 1)
};
static inline SqxArgs sqx_carve(void* scratch, const i64 n, const int islands, size_t* bytes = nullptr, size_t* counters_at = nullptr,
                                size_t* counters_bytes = nullptr)
{
    const size_t nt = (size_t)((n + SEQ_TILE - 1) / SEQ_TILE), M = (size_t)islands;
    SqxArgs q;
    q.n = n;
    q.ntiles = (int)nt;
    q.margin = ldexp((double)(8192 + 8 * (i64)nt), -53);
    char* p = (char*)scratch;
    q.Pin = (u64*)p; p += M * nt * SEQ_TILE * 8;
    q.Sfull = (double*)p; p += M * ((size_t)n + 8) * 8;
    q.Rt = (u64*)p; p += M * nt * 8;
    q.hseg = (u64*)p; p += M * nt * 8;
    q.Pt = (u64*)p; p += M * (nt + 1) * 8;
    q.hI = (u64*)p; p += M * nt * 8;
    q.xraw = (u64*)p; p += M * SQX_CAP * 32;
    q.xs = (u64*)p; p += M * SQX_CAP * 32;
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
// `before`: the estimate of the sum in front of tile b (the same value in every thread)
template <class Src>
__device__ __forceinline__ void sqx_classify_tile(const Src& src, const int isl, const int b, const double before, const SqxArgs& q)
{
    __shared__ double smd[SMC_SM];
    __shared__ u64 smu[SMC_NWAVE];
    __shared__ u32 smx[SMC_NWAVE];
    __shared__ u32 s_lo[SQX_TCAP + 2], s_hi[SQX_TCAP + 2];
    __shared__ u32 s_base;
    const int tid = (int)threadIdx.x, lane = smc_lane(), wave = smc_wave();
    const i64 j0 = (i64)b * SEQ_TILE, i0 = j0 + (i64)tid * 4;
    double w[4];
    src.load4(i0, w);
    for (int i = tid; i < SQX_TCAP + 2; i += SMC_BLOCK) { s_lo[i] = 0u; s_hi[i] = 2047u; }
    double tot;
    double run = before + smc_block_exscan_f64((w[0] + w[1]) + (w[2] + w[3]), smd, tot);   // (barrier inside: slots armed)
    const double dn = 1.0 - q.margin, up = 1.0 + q.margin;
    u64 r[4], rsum = 0ull;
    u32 aLo[4], aHi[4], nx = 0u;
    bool exc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double lo = run, hi = run + w[k];
        run = hi;
        const int E_lo = seq_bexp(lo * dn), E_hi = seq_bexp(hi * up);
        bool tie, big;
        const u64 rr = seq_round_to_grid(w[k], E_lo >= 1 ? E_lo : 1, tie, big);
        const bool inside = i0 + k < q.n;
        const bool zero = (u64)__double_as_longlong(w[k]) == 0ull;              // (+0.0: beyond n as well)
        const bool grid_ok = lo > 0.0 && E_lo >= 1 && E_hi < 0x7fe && !tie && !big;
        const bool regular = grid_ok && E_lo == E_hi;
        const bool edge = grid_ok && E_hi == E_lo + 1 && rr == 0ull;            // below half a unit of the finer grid
        exc[k] = inside && !zero && !regular && !edge;
        r[k] = (regular && !zero) ? rr : 0ull;
        aLo[k] = (zero || exc[k]) ? 0u : (u32)E_lo;
        aHi[k] = (zero || exc[k]) ? 2047u : (u32)E_hi;
        rsum += r[k];
        nx += exc[k] ? 1u : 0u;
    }
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
    smc_st2g(q.Pin + (i64)isl * q.ntiles * SEQ_TILE + i0, Pin[0], Pin[1]);
    smc_st2g(q.Pin + (i64)isl * q.ntiles * SEQ_TILE + i0 + 2, Pin[2], Pin[3]);
    const bool tile_over = xtot > (u32)SQX_TCAP;
    // the segments' intervals: slot = exceptions in front of the element (an exception's own slot is neutral)
    if (!tile_over) {
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
    if (tid == 0) {
        u32 base = 0u;
        if (xtot) base = (u32)atomicAdd(reinterpret_cast<unsigned long long*>(q.ctr + (i64)isl * 4), (unsigned long long)xtot);
        s_base = base;
        if (tile_over || base + xtot > (u32)SQX_CAP) smc_st_agent(q.ctr + (i64)isl * 4 + 1, 1ull);
        smc_st_agent(q.Rt + (i64)isl * q.ntiles + b, rtot);
    }
    __syncthreads();                                           // (the slots are final, s_base is set)
    if (tid == 0) smc_st_agent(q.hseg + (i64)isl * q.ntiles + b, (u64)s_lo[0] | ((u64)s_hi[0] << 16));
    const u32 base = s_base;
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
}

// ---- launch 1, the island's serial part: the last workgroup to arrive ----------------------------------------------
template <class Src>
__device__ __forceinline__ void sqx_chain(const Src& src, const int isl, const SqxArgs& q)
{
    __shared__ u32 c_j[SQX_CAP];
    __shared__ u32 c_acc[SQX_CAP];
    __shared__ u64 c_P[SQX_CAP];
    __shared__ double c_w[SQX_CAP];
    __shared__ double c_S[SQX_CAP];
    __shared__ u64 smu[SMC_SM];
    __shared__ int c_ok, s_idx;
    __shared__ double s_tmp;
    const int tid = (int)threadIdx.x, ntiles = q.ntiles;
    u64* ctr = q.ctr + (i64)isl * 4;
    const u64 cnt64 = smc_ld_agent(ctr), ovf = smc_ld_agent(ctr + 1);
    __syncthreads();
    if (tid == 0) { smc_st_agent(ctr, 0ull); smc_st_agent(ctr + 1, 0ull); c_ok = 1; }   // (re-armed for the next launch)
    bool slow = ovf != 0ull || cnt64 > (u64)SQX_CAP;
    const int cnt = slow ? 0 : (int)cnt64;
    u64* Pt = q.Pt + (i64)isl * (ntiles + 1);
    if (!slow) {
        // ---- P offsets of the tiles
        u64 carry = 0ull;
        for (int b0 = 0; b0 < ntiles; b0 += SMC_BLOCK) {
            const int b = b0 + tid;
            u64 tot;
            const u64 pre = smc_block_exscan_u64(b < ntiles ? smc_ld_agent(q.Rt + (i64)isl * ntiles + b) : 0ull, smu, tot);
            if (b < ntiles) Pt[b] = carry + pre;
            carry += tot;
            __syncthreads();
        }
        if (tid == 0) Pt[ntiles] = carry;
        // ---- the exceptions in order of index (each thread: up to SQX_CAP / 256 of them; rank by counting)
        constexpr int PER = SQX_CAP / SMC_BLOCK;
        u64 e0[PER], e1[PER], e2[PER], e3[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * SMC_BLOCK;
            e0[k] = e1[k] = e2[k] = e3[k] = 0ull;
            if (i < cnt) {
                const u64* e = q.xraw + ((i64)isl * SQX_CAP + i) * 4;
                e0[k] = smc_ld_agent(e);
                e1[k] = smc_ld_agent(e + 1);
                e2[k] = smc_ld_agent(e + 2);
                e3[k] = smc_ld_agent(e + 3);
                c_j[i] = (u32)e0[k];
            }
        }
        __syncthreads();                                       // (c_j complete; Pt visible to the workgroup)
        int rank[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * SMC_BLOCK;
            rank[k] = 0;
            if (i < cnt) {
                const u32 mine = (u32)e0[k];
                for (int m = 0; m < cnt; ++m) rank[k] += c_j[m] < mine ? 1 : 0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * SMC_BLOCK;
            if (i < cnt) {
                const int d = rank[k];
                c_j[d] = (u32)e0[k];
                c_P[d] = e1[k] + Pt[(int)(e0[k] >> 10)];
                c_w[d] = __longlong_as_double((long long)e2[k]);
                c_acc[d] = (u32)e3[k];
            }
        }
        __syncthreads();
        // ---- the walk (one thread): a run of regular elements is an integer added on the grid of s, an exception the
        // hardware's own addition (resampling.py:506-508)
        if (tid == 0) {
            double s = 0.0;
            u64 Pprev = 0ull;
            bool ok = true;
            for (int i = 0; i < cnt; ++i) {
                const u64 dP = c_P[i] - Pprev;
                if (dP != 0ull) {
                    const int Es = seq_bexp(s);
                    const u64 Iv = sqx_mant(s) + dP;
                    ok = ok && Es >= 1 && Es < 0x7ff && Iv < (1ull << 53);
                    s = sqx_pack(Es, Iv);
                }
                s = c_j[i] == 0u ? c_w[i] : s + c_w[i];        // (s = W[0] starts the chain)
                c_S[i] = s;
                Pprev = c_P[i];
            }
            const u64 dP = Pt[ntiles] - Pprev;                 // the run behind the last exception
            if (dP != 0ull) {
                const int Es = seq_bexp(s);
                ok = ok && Es >= 1 && Es < 0x7ff && sqx_mant(s) + dP < (1ull << 53);
            }
            if (!ok) c_ok = 0;
        }
        __syncthreads();
        // ---- verification of every segment against the binade the walk found in front of it; the tiles' headers
        bool bad = false;
        for (int i = tid; i < cnt; i += SMC_BLOCK) {
            const u32 e = (u32)seq_bexp(c_S[i]);
            bad = bad || e < (c_acc[i] & 0xffffu) || e > (c_acc[i] >> 16);
            u64* o = q.xs + ((i64)isl * SQX_CAP + i) * 4;
            o[0] = (u64)c_j[i];
            o[1] = (u64)__double_as_longlong(c_S[i]);
            o[2] = c_P[i] - Pt[(int)(c_j[i] >> 10)];
        }
        for (int b = tid; b <= ntiles; b += SMC_BLOCK) {
            const u64 lo_j = (u64)b * SEQ_TILE;
            int lo = 0, hi = cnt;                              // first exception with index >= the tile's start
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if ((u64)c_j[mid] < lo_j) lo = mid + 1; else hi = mid;
            }
            q.xfirst[(i64)isl * (ntiles + 1) + b] = lo;
            if (b == ntiles) break;
            const int qx = lo - 1;
            const double sb = qx < 0 ? 0.0 : c_S[qx];
            const u32 e = (u32)seq_bexp(sb);
            const u64 hs = smc_ld_agent(q.hseg + (i64)isl * ntiles + b);
            bad = bad || e < (u32)(hs & 0xffffull) || e > (u32)(hs >> 16);
            q.hE[(i64)isl * ntiles + b] = (int)e;
            q.hI[(i64)isl * ntiles + b] = sqx_mant(sb) + (Pt[b] - (qx < 0 ? 0ull : c_P[qx]));
        }
        if (bad) c_ok = 0;                                     // (benign race: every writer stores 0)
        __syncthreads();
        slow = c_ok == 0;
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
    if (tid == 0) { ctr[2] = slow ? 1ull : 0ull; ctr[3] = cnt64; }
}

// ---- launch 2: a tile's sums in LDS ------------------------------------------------------------------------------
// sS[0 .. 1023] <- S_j of tile b (beyond n: the last sum); returns the sum in front of the tile (-inf for tile 0).
// Every thread must call; ends with a barrier.
__device__ __forceinline__ double sqx_stage_tile(const SqxArgs& q, const int isl, const int b, double* sS)
{
    const int tid = (int)threadIdx.x;
    const i64 j0 = (i64)b * SEQ_TILE, i0 = j0 + (i64)tid * 4;
    const u64 mode = smc_uniform_u64(smc_ldg(q.ctr + (i64)isl * 4 + 2));
    double S_start;
    if (mode != 0ull) {
        const double* Sf = q.Sfull + (i64)isl * (q.n + 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) sS[tid * 4 + k] = Sf[i0 + k < q.n ? i0 + k : q.n - 1];
        S_start = b ? Sf[j0 - 1] : -INFINITY;
    } else {
        const int hE = q.hE[(i64)isl * q.ntiles + b];
        const u64 hI = smc_ldg(q.hI + (i64)isl * q.ntiles + b);
        const int xf0 = (int)smc_uniform_u64((u64)q.xfirst[(i64)isl * (q.ntiles + 1) + b]);
        const int xf1 = (int)smc_uniform_u64((u64)q.xfirst[(i64)isl * (q.ntiles + 1) + b + 1]);
        u64 Pin[4];
        smc_ld2g(q.Pin + (i64)isl * q.ntiles * SEQ_TILE + i0, Pin[0], Pin[1]);
        smc_ld2g(q.Pin + (i64)isl * q.ntiles * SEQ_TILE + i0 + 2, Pin[2], Pin[3]);
        int bE[4];
        u64 bI[4], bP[4];
        double fin[4];
        bool isfin[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { bE[k] = hE; bI[k] = hI; bP[k] = 0ull; fin[k] = 0.0; isfin[k] = false; }
        for (int e = xf0; e < xf1; ++e) {                      // (uniform trip count: the tile's exceptions, in order)
            const u64* x = q.xs + ((i64)isl * SQX_CAP + e) * 4;
            const i64 jx = (i64)smc_uniform_u64(smc_ldg(x));
            const double Sx = __longlong_as_double((long long)smc_uniform_u64(smc_ldg(x + 1)));
            const u64 Px = smc_uniform_u64(smc_ldg(x + 2));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (i0 + k > jx) { bE[k] = seq_bexp(Sx); bI[k] = sqx_mant(Sx); bP[k] = Px; isfin[k] = false; }
                if (i0 + k == jx) { fin[k] = Sx; isfin[k] = true; }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) sS[tid * 4 + k] = isfin[k] ? fin[k] : sqx_pack(bE[k], bI[k] + (Pin[k] - bP[k]));
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
// the same for sorted uniforms in memory, two thresholds at once: a 256-ary search by the workgroup (three rounds at
// M = 2^22 instead of 22 dependent loads).  Every thread must call; the results are the same in every thread.
__device__ __forceinline__ void sqx_count_le_sorted2(const double* su, const i64 M, const double xa, const double xb, i64& na, i64& nb)
{
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
            const bool t = idx < hi[v] && su[idx] <= x[v];
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
// first k in [0, m) with x <= S[k]; m - 1 if none
__device__ __forceinline__ int sqx_first_ge_lds(const double* S, const int m, const double x)
{
    int lo = 0, hi = m;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (S[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo < m ? lo : m - 1;
}
// the offspring of tile b: n with S_start < su_n <= S_end (the last tile takes every offspring left: the reference
// would run off the end of W there), each written with its ancestor.  AT: u32 (the filter) or i64 (the operator).
template <class AT>
__device__ __forceinline__ void sqx_search_tile(const SqxArgs& q, const int isl, const int b, const SmcSu& su, AT* A)
{
    __shared__ double sS[SEQ_TILE];
    const int tid = (int)threadIdx.x;
    const i64 j0 = (i64)b * SEQ_TILE;
    const int m_all = (int)(j0 + SEQ_TILE < q.n ? SEQ_TILE : q.n - j0);
    const double S_start = sqx_stage_tile(q, isl, b, sS);
    const double S_end = sS[m_all - 1];
    const bool last = b == q.ntiles - 1;
    i64 n_lo, n_hi;
    if (su.scheme == SMC_MULTINOMIAL_) {
        sqx_count_le_sorted2(su.u, su.M, S_start, S_end, n_lo, n_hi);
    } else {
        n_lo = b ? sqx_count_le(su, S_start) : 0;
        n_hi = last ? su.M : sqx_count_le(su, S_end);
    }
    if (b == 0) n_lo = 0;
    if (last) n_hi = su.M;
    if (su.scheme == SMC_MULTINOMIAL_) {
        for (i64 n = n_lo + tid; n < n_hi; n += SMC_BLOCK) A[n] = (AT)(j0 + sqx_first_ge_lds(sS, m_all, su.u[n]));
        return;
    }
    for (i64 p = (n_lo >> 1) + tid; 2 * p < n_hi; p += SMC_BLOCK) {      // pairs (2p, 2p + 1): one Philox call
        double s0, s1;
        smc_su_pair(su, p, s0, s1);
        if (2 * p >= n_lo) A[2 * p] = (AT)(j0 + sqx_first_ge_lds(sS, m_all, s0));
        if (2 * p + 1 < n_hi) A[2 * p + 1] = (AT)(j0 + sqx_first_ge_lds(sS, m_all, s1));
    }
}

// ---- the stand-alone operators' kernels (W an array; the tiles' sums by k_seq_tile_sums) ------------------------------
static __global__ void __launch_bounds__(SMC_BLOCK)
k_sqx_classify(const double* W, const double* tsum, const SqxArgs q, const SeqGate gate)
{
    __shared__ double smd[SMC_SM];
    __shared__ int s_flag;
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y, tid = (int)threadIdx.x;
    if (!seq_gate_open(gate, isl)) return;
    const double* ts = tsum + (i64)isl * q.ntiles;
    double before = 0.0;
    for (int j = tid; j < b; j += SMC_BLOCK) before += ts[j];
    before = smc_block_sum(before, smd);
    __syncthreads();
    const SqxSrcArray src{W + (i64)isl * q.n, q.n};
    sqx_classify_tile(src, isl, b, before, q);
    if (sqx_last_block(q.tick + (i64)isl * SQX_CNT_WORDS, b, q.ntiles, &s_flag)) sqx_chain(src, isl, q);
}
static __global__ void __launch_bounds__(SMC_BLOCK)
k_sqx_fill(const SqxArgs q, double* S, const SeqGate gate)
{
    __shared__ double sS[SEQ_TILE];
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y, tid = (int)threadIdx.x;
    if (!seq_gate_open(gate, isl)) return;
    (void)sqx_stage_tile(q, isl, b, sS);
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
    sqx_search_tile<i64>(q, (int)blockIdx.y, (int)blockIdx.x, su, A);
}
