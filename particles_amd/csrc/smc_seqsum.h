// smc_seqsum.h -- the reference's SEQUENTIAL fp64 prefix sums, computed in parallel, bit for bit.
//
// `inverse_cdf` of the reference (resampling.py:500-509) walks  s = W[0];  s += W[j]  in fp64: the N roundings
// are a chain, and reproducing the reference's ancestors exactly -- not up to near-ties -- means reproducing
// every s_j.  A literal restatement is one lane adding N numbers (smc_resample.h "STRICT", k_seq_cdf: 2.5 ms at
// N = 2^20).  But the chain is almost everywhere an INTEGER sum in disguise:
//
//   while s stays inside one binade [2^k, 2^(k+1)) its values lie on the grid g = 2^(k-52), s = I g with an
//   integer 2^52 <= I < 2^53, and for W >= 0   fl(s + W) = (I + R(W)) g   with  R(W) = W / g rounded to the nearest
//   integer -- a function of W alone, except for an exact tie (W / g = m + 1/2), where the parity of I + m decides.
//
// So: split the array into tiles of 1024; a tile is CLEAN for binade k when the running sum provably enters and
// leaves it inside that binade (decided from a parallel fp64 estimate of the prefix sums, with a margin far above
// the estimate's error), none of its elements is a tie on that grid and none is larger than the grid's binade.
// A clean tile adds the integer T_b = sum R(W_i) -- an ordinary parallel sum.  ONE workgroup then walks the tiles
// (k_seq_chain): runs of clean tiles of the current binade are prefix-summed 256 at a time and VERIFIED (the tile
// must start in its binade and end below 2^(k+1): the sums are monotone, so everything in between is inside);
// the first tile that fails -- a binade crossing, a tie, a wrong guess -- is done exactly by one wave with the same
// idea at wave granularity (seq_tile_wave_exact: integer scans of 64 elements, the hardware's own addition at every
// exception); a dozen or two tiles per call (s doubles log2(N) times), and the walk resumes.
// A last parallel pass writes S_j = (I_b + prefix_j) g of the clean tiles.  Every S_j is the reference's, for any
// W >= 0 (ties, zeros, subnormals, one element holding all the mass); NaN or negative weights make every tile
// fail its verification and the whole array goes element by element -- slow, still the reference's values.
//
// Cost at N = 2^20: 0.2 - 0.8 ms instead of 43 (profiles/r12*_strict*).  Since round 5 this tile walk is the
// definition-level form the tests hold the production path against (smc_seq_prefix_sums mode 2); the production path --
// smc_inverse_cdf_strict (smc_ops.hip) and the filter's SMC_FLAG_STRICT_ANCESTORS (smc_filter.hip) -- is smc_seqx.h: the
// same idea at ELEMENT granularity in two launches, with seq_tile_block_exact below as its exact fallback.
#pragma once

#define SEQ_TILE 1024                  /* elements per tile: 4 per thread of a 256-thread workgroup */
#define SEQ_NOT_CLEAN (-100000)

// optional gate of a launch (the filter's step loop): the per-island step record -- return unless step t resamples
struct SeqGate {
    const double* info;                // null: always run
    int stride;
    i64 T;
    const unsigned* only_if;           // (n_islands) or null: run only where the word is non-zero (the fallback passes)
};
__device__ __forceinline__ bool seq_gate_open(const SeqGate& g, const int isl)
{
    if (g.only_if && smc_uniform_u64((u64)g.only_if[isl]) == 0ull) return false;
    if (!g.info) return true;
    const double* r = g.info + (i64)isl * g.stride;
    const i64 t = (i64)smc_uniform(r[0]);
    return !(t >= g.T || t == 0 || smc_uniform(r[1]) == 0.0);
}

// biased exponent of a finite positive double (0 for zero / subnormal)
__device__ __forceinline__ int seq_bexp(const double x)
{
    return (int)(((u64)__double_as_longlong(x) >> 52) & 0x7ffull);
}
// W on the grid of a running sum with biased exponent Es (Es >= 1): the integer R(W) it adds, or flags
//   tie: W / g is exactly m + 1/2 (the parity of the running integer decides);  big: W's binade lies above Es
__device__ __forceinline__ u64 seq_round_to_grid(const double W, const int Es, bool& tie, bool& big)
{
    const u64 bits = (u64)__double_as_longlong(W);
    const int e = (int)((bits >> 52) & 0x7ffull);
    const u64 M = (bits & 0x000FFFFFFFFFFFFFull) | (e ? 0x0010000000000000ull : 0ull);
    const int ee = e ? e : 1;
    int sh = Es - ee;
    tie = false;
    big = sh < 0 || e == 0x7ff || (bits >> 63) != 0ull;       // (inf / NaN / negative: never clean)
    if (big) return 0ull;
    if (sh == 0) return M;
    sh = sh > 63 ? 63 : sh;
    const u64 m = M >> sh;
    const u64 rem = M & ((1ull << sh) - 1ull), half = 1ull << (sh - 1);
    tie = rem == half;
    return m + (rem > half ? 1ull : 0ull);
}

// ---- pass 1: fp64 tile sums (any order: an ESTIMATE of where the running sum is, error << the margin below)
static __global__ void __launch_bounds__(SMC_BLOCK)
k_seq_tile_sums(const double* W, const i64 n, double* tsum, const SeqGate gate)
{
    __shared__ double smd[SMC_SM];
    const int isl = (int)blockIdx.y, b = (int)blockIdx.x, tid = (int)threadIdx.x;
    if (!seq_gate_open(gate, isl)) return;
    const double* w = W + (i64)isl * n;
    const i64 i0 = (i64)b * SEQ_TILE + (i64)tid * 4;
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) v += (i0 + k < n) ? w[i0 + k] : 0.0;
    v = smc_block_sum(v, smd);
    if (tid == 0) tsum[(i64)isl * gridDim.x + b] = v;
}

// ---- pass 2: which binade (if any) tile b is clean for, and the integer it adds there.
// Estimate of the running sum before / after the tile: the fp64 sums of the tiles before it (every workgroup adds
// them up itself: <= ntiles loads, 4 tiles per thread and round).  Margin 2^-20 relative: the estimate (tile sums in
// tree order, then a chain of <= ntiles additions) is within (1024 + ntiles) 2^-53 of the reference's running sum.
static __global__ void __launch_bounds__(SMC_BLOCK)
k_seq_tile_classify(const double* W, const i64 n, const double* tsum, int* tk, u64* tT, const SeqGate gate)
{
    __shared__ double smd[SMC_SM];
    __shared__ u64 smu[SMC_SM];
    __shared__ int s_bad;
    const int isl = (int)blockIdx.y, b = (int)blockIdx.x, tid = (int)threadIdx.x, ntiles = (int)gridDim.x;
    if (!seq_gate_open(gate, isl)) return;
    const double* ts = tsum + (i64)isl * ntiles;
    double before = 0.0;
    for (int j = tid; j < b; j += SMC_BLOCK) before += ts[j];
    if (tid == 0) s_bad = 0;
    before = smc_block_sum(before, smd);                       // (barrier inside: s_bad is set)
    const double after = before + ts[b];
    const int k_lo = seq_bexp(before * (1.0 - 0x1.0p-20)), k_hi = seq_bexp(after * (1.0 + 0x1.0p-20));
    const bool guess = b > 0 && k_lo == k_hi && k_lo >= 1 && k_lo < 0x7fe && before > 0.0;
    const double* w = W + (i64)isl * n;
    const i64 i0 = (i64)b * SEQ_TILE + (i64)tid * 4;
    u64 sum = 0ull;
    bool bad = false;
    if (guess) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bool tie, big;
            const u64 r = seq_round_to_grid((i0 + k < n) ? w[i0 + k] : 0.0, k_lo, tie, big);
            sum += r;
            bad = bad || tie || big;
        }
    }
    if (bad) s_bad = 1;
    sum = smc_block_sum_u64(sum, smu);                         // (barrier inside: s_bad is final)
    if (tid == 0) {
        const bool clean = guess && !s_bad && sum < (1ull << 53);
        tk[(i64)isl * ntiles + b] = clean ? k_lo : SEQ_NOT_CLEAN;
        tT[(i64)isl * ntiles + b] = clean ? sum : 0ull;
    }
}

// A tile the walk cannot take on trust, done exactly by the whole workgroup: the same idea one level down.  With the
// running sum s in binade Es every thread rounds its 4 elements to that grid and the workgroup scans; everything up
// to the first EXCEPTION -- a tie, an element above the binade, the sum reaching 2^(k+1), s not a normal number -- is
// exact as an integer sum; the exceptional element is added by the hardware (s + W: the reference's own operation,
// whatever the rounding case), and the scan resumes behind it on the grid s is on now.  A tile holds a handful of
// exceptions (one binade crossing, rarely a tie; the first tile a dozen: s = W[0] doubles ten times in it).
// `first`: s = W[0] starts the chain (resampling.py:506).  w4: this thread's elements 4 tid .. 4 tid + 3 of the tile;
// out4: their sums.  Returns the sum behind the tile (the same in every thread).
__device__ __forceinline__ double seq_tile_block_exact(const double (&w4)[4], double (&out4)[4], const int m_all, double s,
                                                       bool first, u64* smu, int* s_idx, double* s_val)
{
    const int tid = (int)threadIdx.x;
    int pos = 0;                                               // elements below pos are done (the same in every thread)
    while (pos < m_all) {
        if (tid == 0) *s_idx = m_all;
        __syncthreads();
        if (first || s == 0.0) {
            // nothing summed yet (or only zeros): 0 + W = W exactly -- skip to the first non-zero element
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = tid * 4 + k;
                if (i >= pos && i < m_all && w4[k] != 0.0) atomicMin(s_idx, i);
            }
            __syncthreads();
            const int f0 = *s_idx;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = tid * 4 + k;
                if (i >= pos && i < f0) out4[k] = 0.0;
                if (i == f0) { out4[k] = w4[k]; *s_val = w4[k]; }
            }
            __syncthreads();
            s = f0 < m_all ? *s_val : 0.0;
            first = false;
            pos = f0 + 1;
            continue;
        }
        const int Es = seq_bexp(s);
        const bool normal = Es >= 1 && Es < 0x7ff;
        const u64 I = ((u64)__double_as_longlong(s) & 0x000FFFFFFFFFFFFFull) | 0x0010000000000000ull;
        u64 r[4], sum = 0ull;
        bool bad[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid * 4 + k;
            bool tie, big;
            r[k] = seq_round_to_grid(w4[k], normal ? Es : 1, tie, big);
            const bool active = i >= pos && i < m_all;
            r[k] = active ? r[k] : 0ull;
            bad[k] = active && (tie || big || !normal);
            sum += r[k];
        }
        u64 total;
        u64 run = I + smc_block_exscan_u64(sum, smu, total);
        double cand[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid * 4 + k;
            run += r[k];
            cand[k] = __longlong_as_double((long long)(((u64)Es << 52) | (run & 0x000FFFFFFFFFFFFFull)));
            if (i >= pos && i < m_all && (bad[k] || run >= (1ull << 53))) atomicMin(s_idx, i);
        }
        __syncthreads();
        const int f = *s_idx;                                  // first exception (m_all: none)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid * 4 + k;
            if (i >= pos && i < f) out4[k] = cand[k];
            if (i == f - 1 && f - 1 >= pos) *s_val = cand[k];  // the sum in front of the exception
        }
        __syncthreads();
        if (f - 1 >= pos) s = *s_val;
        __syncthreads();
        if (f < m_all) {                                       // the exception: the hardware's own addition
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (tid * 4 + k == f) { out4[k] = s + w4[k]; *s_val = out4[k]; }
            __syncthreads();
            s = *s_val;
        }
        pos = f + 1;
    }
    __syncthreads();
    return s;
}

// ---- pass 3: the walk.  One workgroup per island.  tstart[b]: the running sum BEFORE tile b (exact), for the clean
// tiles that verified; tk[b] is rewritten to SEQ_NOT_CLEAN for every tile done element by element (its S is final).
static __global__ void __launch_bounds__(SMC_BLOCK)
k_seq_chain(const double* W, const i64 n, const int ntiles, int* tk, const u64* tT, double* tstart, double* S,
            unsigned long long* nseq, const SeqGate gate)
{
    __shared__ u64 smu[SMC_SM];
    __shared__ int s_first;
    __shared__ double s_s;
    __shared__ double s_tmp;
    const int isl = (int)blockIdx.x, tid = (int)threadIdx.x;
    if (!seq_gate_open(gate, isl)) return;
    const double* w = W + (i64)isl * n;
    double* So = S + (i64)isl * n;
    int* k_of = tk + (i64)isl * ntiles;
    const u64* T_of = tT + (i64)isl * ntiles;
    double* st = tstart + (i64)isl * ntiles;
    unsigned long long nexact = 0ull;
    double s = 0.0;                                            // (every thread holds the running sum)
    bool started = false;
    for (int b0 = 0; b0 < ntiles; b0 += SMC_BLOCK) {
        // a chunk of 256 tiles: their classification is read once, the rounds below work on registers
        const int chunk = ntiles - b0 < SMC_BLOCK ? ntiles - b0 : SMC_BLOCK;
        const int b = b0 + tid;
        const int kb = tid < chunk ? k_of[b] : SEQ_NOT_CLEAN;
        const u64 Tb_all = tid < chunk ? T_of[b] : 0ull;
        int done = 0;                                          // tiles of the chunk behind us (the same in every thread)
        while (done < chunk) {
            // ---- speculate: the tiles done .. chunk - 1 are clean for the binade s is in
            const int Es = seq_bexp(s);
            const u64 I = ((u64)__double_as_longlong(s) & 0x000FFFFFFFFFFFFFull) | 0x0010000000000000ull;
            const bool open = tid >= done && tid < chunk;
            const bool mine = started && open && kb == Es && Es >= 1;
            const u64 Tb = mine ? Tb_all : 0ull;
            u64 total;
            const u64 pre = smc_block_exscan_u64(Tb, smu, total);       // integers added by the open tiles before mine
            const bool ok = mine && I + pre + Tb < (1ull << 53);        // starts on the grid of binade Es, ends inside it
            if (tid == 0) s_first = chunk;
            __syncthreads();
            if (open && !ok) atomicMin(&s_first, tid);
            __syncthreads();
            const int f = s_first;                             // first open tile that is not taken on trust
            if (open && tid < f) st[b] = __longlong_as_double((long long)(((u64)Es << 52) | ((I + pre) & 0x000FFFFFFFFFFFFFull)));
            if (f > done && tid == f - 1)                      // the sum behind the last verified tile
                s_s = __longlong_as_double((long long)(((u64)Es << 52) | ((I + pre + Tb) & 0x000FFFFFFFFFFFFFull)));
            __syncthreads();
            if (f > done) s = s_s;
            done = f;
            if (done < chunk) {
                // ---- tile b0 + done exactly, by the whole workgroup
                const int bx = b0 + done;
                const i64 lo = (i64)bx * SEQ_TILE;
                const int m_all = (int)(lo + SEQ_TILE < n ? SEQ_TILE : n - lo);
                double w4[4], o4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k = 0; k < 4; ++k) w4[k] = tid * 4 + k < m_all ? w[lo + tid * 4 + k] : 0.0;
                s = seq_tile_block_exact(w4, o4, m_all, s, !started, smu, &s_first, &s_tmp);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (tid * 4 + k < m_all) So[lo + tid * 4 + k] = o4[k];
                if (tid == 0) k_of[bx] = SEQ_NOT_CLEAN;
                started = true;
                nexact += 1ull;
                done += 1;
                __syncthreads();
            }
        }
    }
    if (tid == 0) nseq[isl] = nexact;                          // (diagnostic: tiles done by the exact wave)
}

// ---- pass 4: S of the clean tiles, S_j = (I_b + sum_{i <= j} R(W_i)) g
static __global__ void __launch_bounds__(SMC_BLOCK)
k_seq_fill(const double* W, const i64 n, const int* tk, const double* tstart, double* S, const SeqGate gate)
{
    __shared__ u64 smu[SMC_SM];
    const int isl = (int)blockIdx.y, b = (int)blockIdx.x, tid = (int)threadIdx.x, ntiles = (int)gridDim.x;
    if (!seq_gate_open(gate, isl)) return;
    const int Es = tk[(i64)isl * ntiles + b];
    if (Es == SEQ_NOT_CLEAN) return;                           // written by the walk
    const double s0 = tstart[(i64)isl * ntiles + b];
    const u64 I0 = ((u64)__double_as_longlong(s0) & 0x000FFFFFFFFFFFFFull) | 0x0010000000000000ull;
    const double* w = W + (i64)isl * n;
    double* So = S + (i64)isl * n;
    const i64 i0 = (i64)b * SEQ_TILE + (i64)tid * 4;
    u64 r[4], sum = 0ull;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        bool tie, big;
        r[k] = seq_round_to_grid((i0 + k < n) ? w[i0 + k] : 0.0, Es, tie, big);
        sum += r[k];
    }
    u64 total;
    u64 run = I0 + smc_block_exscan_u64(sum, smu, total);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        run += r[k];
        if (i0 + k < n) So[i0 + k] = __longlong_as_double((long long)(((u64)Es << 52) | (run & 0x000FFFFFFFFFFFFFull)));
    }
}

// scratch the tile walk needs besides S, per island: per tile 3 x 8 + 4 bytes, a counter
static inline size_t seq_scratch_bytes(const i64 n, const int islands)
{
    const size_t ntiles = (size_t)((n + SEQ_TILE - 1) / SEQ_TILE);
    return (size_t)islands * (ntiles * 32 + 16) + 64;
}
struct SeqScratch {
    double *tsum, *tstart;
    u64* tT;
    int* tk;
    unsigned long long* nseq;
};
static inline SeqScratch seq_scratch_carve(void* scratch, const i64 n, const int islands)
{
    const size_t nt = (size_t)islands * (size_t)((n + SEQ_TILE - 1) / SEQ_TILE);
    SeqScratch q;
    char* p = (char*)scratch;
    q.tsum = (double*)p; p += nt * 8;
    q.tstart = (double*)p; p += nt * 8;
    q.tT = (u64*)p; p += nt * 8;
    q.nseq = (unsigned long long*)p; p += (size_t)islands * 8;
    q.tk = (int*)p;
    return q;
}
// (where the walk's count of tiles it did exactly sits in the scratch: one u64 per island)
static inline const unsigned long long* seq_nseq_ptr(const void* scratch, const i64 n, const int islands)
{
    return seq_scratch_carve((void*)scratch, n, islands).nseq;
}
// S <- the reference's sequential fp64 prefix sums of W (both (islands, n), S may not alias W) by the TILE WALK: the
// definition-level parallel form the two-launch path of smc_seqx.h is tested against (smc_seq_prefix_sums mode 2).
static inline void seq_tile_walk_launch(hipStream_t st, const double* W, const i64 n, const int islands, double* S, void* scratch)
{
    const int ntiles = (int)((n + SEQ_TILE - 1) / SEQ_TILE);
    const SeqScratch q = seq_scratch_carve(scratch, n, islands);
    const SeqGate gate{nullptr, 0, 0, nullptr};
    SMC_LAUNCH(k_seq_tile_sums, dim3(ntiles, islands), dim3(SMC_BLOCK), st, W, n, q.tsum, gate);
    SMC_LAUNCH(k_seq_tile_classify, dim3(ntiles, islands), dim3(SMC_BLOCK), st, W, n, (const double*)q.tsum, q.tk, q.tT, gate);
    SMC_LAUNCH(k_seq_chain, dim3(islands), dim3(SMC_BLOCK), st, W, n, ntiles, q.tk, (const u64*)q.tT, q.tstart, S, q.nseq, gate);
    SMC_LAUNCH(k_seq_fill, dim3(ntiles, islands), dim3(SMC_BLOCK), st, W, n, (const int*)q.tk, (const double*)q.tstart, S, gate);
}
