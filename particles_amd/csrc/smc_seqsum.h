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
// Cost at N = 2^20: ~50 us instead of 2.5 ms (profiles/r12*_strict*).  Used by smc_inverse_cdf_strict
// (smc_ops.hip) and by the filter's SMC_FLAG_STRICT_ANCESTORS path (smc_filter.hip).
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

// =====================================================================================================================
// The same at ELEMENT granularity -- the fast path; the tile walk above is its fallback.
// An element j is REGULAR when the estimate puts the running sum before AND after it inside one binade E_j (margin as
// above) and it is neither a tie nor above that binade; everything else -- the few elements at which the sum changes
// binade, the first elements of the array, ties -- is an EXCEPTION, a few dozen per call.  Consecutive regular elements
// share their grid (E_{j+1} = E_j: see DESIGN 5.4), so between two exceptions the chain is an integer sum
// P[j] = sum_{i < j} r_i of the regular elements' roundings -- one parallel scan over the whole array -- and the SERIAL
// part shrinks to the exceptions themselves: one thread walks the sorted list, s <- (I(s) + P-difference) on the grid,
// then s <- s + W_x with the hardware's own addition.  k_seq_elem_fill writes S_j = (I(S_x) + P[j+1] - P[x+1]) g for the
// regular elements behind exception x and VERIFIES what the walk assumed (the base's binade is E_j, the integer stays
// below 2^53): any violation, or more exceptions than the list holds, raises `need_fallback` and k_seq_fallback redoes
// the array.  Cost at N = 2^20: four short launches (+ the fallback's, which returns at once).
#define SEQ_E_ANY (-2)                 /* SeqElem::E of a zero element: regular on whatever grid the sum is on */
#define SEQ_XCAP 2048                  /* exceptions the list holds per island (48 KB of LDS in the walk): more than a
                                          sum can cross binades (2046) -- only engineered ties or NaN weights overflow it */
struct SeqX {                          // one exception: its index and value, P in front of it (tile-local, then global)
    i64 j;
    u64 P;
    double w;
};
// what a thread knows about its 4 elements of tile b (the same code in the classify and the fill pass: same bits)
struct SeqElem {
    double w[4];
    int E[4];                          // grid (biased exponent) of a regular element; -1: exception
    u64 r[4];                          // its rounding (0 for exceptions)
    u64 Pex[4];                        // tile-local exclusive prefix of r at each element
    u64 rtot;                          // the tile's sum of r
};
__device__ __forceinline__ void seq_elem_eval(const double* w, const i64 n, const double* ts, const int b, SeqElem& e,
                                              double* smd, u64* smu)
{
    const int tid = (int)threadIdx.x;
    double before = 0.0;
    for (int j = tid; j < b; j += SMC_BLOCK) before += ts[j];
    before = smc_block_sum(before, smd);
    const i64 i0 = (i64)b * SEQ_TILE + (i64)tid * 4;
    double mysum = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        e.w[k] = (i0 + k < n) ? w[i0 + k] : 0.0;
        mysum += e.w[k];
    }
    __syncthreads();                                           // (smd is reused)
    double tot;
    double run = before + smc_block_exscan_f64(mysum, smd, tot);   // estimate of the running sum in front of my elements
    u64 rs = 0ull;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double lo = run, hi = run + e.w[k];
        run = hi;
        const int E_lo = seq_bexp(lo * (1.0 - 0x1.0p-20)), E_hi = seq_bexp(hi * (1.0 + 0x1.0p-20));
        bool tie, big;
        const u64 r = seq_round_to_grid(e.w[k], E_lo >= 1 ? E_lo : 1, tie, big);
        const bool regular = i0 + k < n && lo > 0.0 && E_lo == E_hi && E_lo >= 1 && E_lo < 0x7fe && !tie && !big;
        // (a zero adds nothing on any grid: regular wherever the sum is -- SEQ_E_ANY; this keeps a collapsed weight
        //  vector, zeros around one mass sitting exactly on a power of two, on the fast path)
        const bool zero = i0 + k < n && e.w[k] == 0.0 && (u64)__double_as_longlong(e.w[k]) == 0ull;
        e.E[k] = zero ? SEQ_E_ANY : (regular ? E_lo : -1);
        e.r[k] = (regular && !zero) ? r : 0ull;
        rs += e.r[k];
    }
    u64 pre = smc_block_exscan_u64(rs, smu, e.rtot);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        e.Pex[k] = pre;
        pre += e.r[k];
    }
}
static __global__ void __launch_bounds__(SMC_BLOCK)
k_seq_elem_classify(const double* W, const i64 n, const double* tsum, u64* Rt, SeqX* xlist, unsigned* xcount, unsigned* need_fallback,
                    const SeqGate gate)
{
    __shared__ double smd[SMC_SM];
    __shared__ u64 smu[SMC_SM];
    const int isl = (int)blockIdx.y, b = (int)blockIdx.x, tid = (int)threadIdx.x, ntiles = (int)gridDim.x;
    if (!seq_gate_open(gate, isl)) return;
    SeqElem e;
    seq_elem_eval(W + (i64)isl * n, n, tsum + (i64)isl * ntiles, b, e, smd, smu);
    if (tid == 0) Rt[(i64)isl * ntiles + b] = e.rtot;
    const i64 i0 = (i64)b * SEQ_TILE + (i64)tid * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k < n && e.E[k] == -1) {
            const unsigned slot = atomicAdd(xcount + isl, 1u);
            if (slot < SEQ_XCAP) {
                SeqX x;
                x.j = i0 + k;
                x.P = e.Pex[k];
                x.w = e.w[k];
                xlist[(i64)isl * SEQ_XCAP + slot] = x;
            }
        }
    }
}
// one workgroup per island: the tiles' P offsets, the exceptions in order, the walk over them, every tile's base
static __global__ void __launch_bounds__(SMC_BLOCK)
k_seq_elem_chain(const double* W, const i64 n, const int ntiles, const u64* Rt, u64* Pt, SeqX* xlist, unsigned* xcount,
                 unsigned* need_fallback, int* tbase, double* S, const SeqGate gate)
{
    __shared__ u64 smu[SMC_SM];
    __shared__ SeqX sx[SEQ_XCAP];
    const int isl = (int)blockIdx.x, tid = (int)threadIdx.x;
    if (!seq_gate_open(gate, isl)) return;
    double* So = S + (i64)isl * n;
    u64* Pto = Pt + (i64)isl * ntiles;
    SeqX* xl = xlist + (i64)isl * SEQ_XCAP;
    const unsigned cnt = xcount[isl];
    __syncthreads();
    if (tid == 0) { xcount[isl] = 0u; need_fallback[isl] = cnt > SEQ_XCAP ? 1u : 0u; }     // (the counter: re-armed)
    if (cnt > SEQ_XCAP) return;                                // too many: the tile walk does the array
    // ---- P offsets of the tiles
    u64 carry = 0ull;
    for (int b0 = 0; b0 < ntiles; b0 += SMC_BLOCK) {
        const int b = b0 + tid;
        u64 tot;
        const u64 pre = smc_block_exscan_u64(b < ntiles ? Rt[(i64)isl * ntiles + b] : 0ull, smu, tot);
        if (b < ntiles) Pto[b] = carry + pre;
        carry += tot;
        __syncthreads();
    }
    // ---- the exceptions, sorted by index (bitonic, padded with +inf keys), P made global
    int m = 1;
    while (m < (int)cnt) m <<= 1;
    for (int i = tid; i < m; i += SMC_BLOCK) {
        SeqX x;
        if (i < (int)cnt) {
            x = xl[i];
            x.P += Pto[x.j / SEQ_TILE];
        } else {
            x.j = (i64)0x7fffffffffffffffll; x.P = 0ull; x.w = 0.0;
        }
        sx[i] = x;
    }
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < m; i += SMC_BLOCK) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const SeqX a = sx[i], c = sx[l];
                    if ((a.j > c.j) == up) { sx[i] = c; sx[l] = a; }
                }
            }
            __syncthreads();
        }
    // ---- the walk (one thread): segment of regular elements as an integer sum, the exception with the hardware's addition
    if (tid == 0) {
        double s = 0.0;
        u64 Pprev = 0ull;
        bool ok = true;                                        // (s = 0 + W[0] = W[0] starts the chain, resampling.py:506)
        for (int i = 0; ok && i < (int)cnt; ++i) {
            const i64 j = sx[i].j;
            const u64 dP = sx[i].P - Pprev;
            if (dP != 0ull) {                                  // regular non-zero elements in between: on the grid of s
                const int Es = seq_bexp(s);
                const u64 I = ((u64)__double_as_longlong(s) & 0x000FFFFFFFFFFFFFull) | 0x0010000000000000ull;
                const u64 Iv = I + dP;
                ok = Es >= 1 && Es < 0x7ff && Iv < (1ull << 53);
                s = __longlong_as_double((long long)(((u64)Es << 52) | (Iv & 0x000FFFFFFFFFFFFFull)));
            }
            s = s + sx[i].w;                                   // resampling.py:508, the hardware's own addition
            So[j] = s;
            Pprev = sx[i].P;
        }
        if (!ok) need_fallback[isl] = 1u;
    }
    __syncthreads();
    // ---- every tile's base: the last exception in front of it (index into the sorted list, written back for the fill)
    for (int i = tid; i < (int)cnt; i += SMC_BLOCK) xl[i] = sx[i];
    for (int b = tid; b < ntiles; b += SMC_BLOCK) {
        const i64 lo_j = (i64)b * SEQ_TILE;
        int lo = 0, hi = (int)cnt;                             // first exception with j >= lo_j
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sx[mid].j < lo_j) lo = mid + 1; else hi = mid;
        }
        tbase[(i64)isl * ntiles + b] = lo - 1;                 // (-1 for tile 0: it starts with exception 0)
    }
}
static __global__ void __launch_bounds__(SMC_BLOCK)
k_seq_elem_fill(const double* W, const i64 n, const double* tsum, const u64* Pt, const SeqX* xlist, unsigned* need_fallback,
                const int* tbase, double* S, const SeqGate gate)
{
    __shared__ double smd[SMC_SM];
    __shared__ u64 smu[SMC_SM];
    __shared__ u32 s_mx[SMC_NWAVE];
    __shared__ u64 s_P[SEQ_TILE];
    const int isl = (int)blockIdx.y, b = (int)blockIdx.x, tid = (int)threadIdx.x, ntiles = (int)gridDim.x;
    if (!seq_gate_open(gate, isl)) return;
    if (smc_uniform_u64((u64)need_fallback[isl]) != 0ull) return;      // (the walk gave up: the tile passes write S)
    SeqElem e;
    seq_elem_eval(W + (i64)isl * n, n, tsum + (i64)isl * ntiles, b, e, smd, smu);
    double* So = S + (i64)isl * n;
    const SeqX* xl = xlist + (i64)isl * SEQ_XCAP;
    const u64 Ptb = Pt[(i64)isl * ntiles + b];
    const int base0 = tbase[(i64)isl * ntiles + b];
    // last exception at or in front of each element, inside the tile: a running maximum over (local index + 1)
    u32 mine = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        s_P[tid * 4 + k] = e.Pex[k];
        if (e.E[k] == -1 && (i64)b * SEQ_TILE + tid * 4 + k < n) mine = (u32)(tid * 4 + k + 1);
    }
    const u32 inc = smc_wave_scan_max_u32(mine);
    u32 ex = smc_mov_dpp<SMC_DPP_WAVE_SHR1>(inc);
    if (smc_lane() == 0) ex = 0u;
    if (smc_lane() == 63) s_mx[smc_wave()] = inc;
    __syncthreads();                                           // (s_P is written too)
    for (int ww = 0; ww < smc_wave(); ++ww) ex = ex > s_mx[ww] ? ex : s_mx[ww];
    u32 last = ex;                                             // exceptions in front of this thread's elements
    bool bad = false;
    const i64 i0 = (i64)b * SEQ_TILE + (i64)tid * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k >= n) continue;
        if (e.E[k] == -1) { last = (u32)(tid * 4 + k + 1); continue; }   // its S is the walk's
        double baseS = 0.0;                                    // (nothing in front: the sum is still zero)
        u64 baseP = 0ull;
        if (last) {                                            // an exception inside the tile
            baseS = So[(i64)b * SEQ_TILE + (last - 1u)];
            baseP = Ptb + s_P[last - 1u];
        } else if (base0 >= 0) {
            baseS = So[xl[base0].j];                           // (the walk wrote the exception's own sum)
            baseP = xl[base0].P;
        }
        const u64 dP = Ptb + e.Pex[k] + e.r[k] - baseP;
        if (dP == 0ull) {                                      // only zeros since the base: its value, whatever it is
            bad = bad || (e.E[k] != SEQ_E_ANY && e.E[k] != seq_bexp(baseS));
            So[i0 + k] = baseS;
            continue;
        }
        const int Es = seq_bexp(baseS);
        const u64 I = ((u64)__double_as_longlong(baseS) & 0x000FFFFFFFFFFFFFull) | 0x0010000000000000ull;
        const u64 Iv = I + dP;
        bad = bad || Es < 1 || Es >= 0x7ff || (e.E[k] != SEQ_E_ANY && Es != e.E[k]) || Iv >= (1ull << 53);
        So[i0 + k] = __longlong_as_double((long long)(((u64)Es << 52) | (Iv & 0x000FFFFFFFFFFFFFull)));
    }
    if (bad) need_fallback[isl] = 1u;
}

// ---- the fallback as ONE launch (the filter's step loop pays a launch for it every step, needed or not): one
// workgroup per island does every tile exactly with seq_tile_block_exact -- a clean tile is one scan, an exceptional
// one a scan per exception.  ~3 us per tile: milliseconds at N = 2^20, but only engineered ties, NaN or negative
// weights get here (SEQ_XCAP holds more exceptions than a sum can cross binades).  The tile walk above (three
// launches, 0.2 - 0.8 ms) stays as smc_seq_prefix_sums' mode 2.
static __global__ void __launch_bounds__(SMC_BLOCK)
k_seq_fallback(const double* W, const i64 n, double* S, const SeqGate gate)
{
    __shared__ u64 smu[SMC_SM];
    __shared__ int s_idx;
    __shared__ double s_tmp;
    const int isl = (int)blockIdx.x, tid = (int)threadIdx.x;
    if (!seq_gate_open(gate, isl)) return;
    const double* w = W + (i64)isl * n;
    double* So = S + (i64)isl * n;
    double s = 0.0;
    bool first = true;
    for (i64 lo = 0; lo < n; lo += SEQ_TILE) {
        const int m_all = (int)(lo + SEQ_TILE < n ? SEQ_TILE : n - lo);
        double w4[4], o4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 4; ++k) w4[k] = tid * 4 + k < m_all ? w[lo + tid * 4 + k] : 0.0;
        s = seq_tile_block_exact(w4, o4, m_all, s, first, smu, &s_idx, &s_tmp);
        first = false;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tid * 4 + k < m_all) So[lo + tid * 4 + k] = o4[k];
    }
}

// scratch the passes need besides S, per island: per tile 4 x 8 + 2 x 8 bytes, the exception list, three words
static inline size_t seq_scratch_bytes(const i64 n, const int islands)
{
    const size_t ntiles = (size_t)((n + SEQ_TILE - 1) / SEQ_TILE);
    return (size_t)islands * (ntiles * 48 + SEQ_XCAP * sizeof(SeqX) + 32);
}
struct SeqScratch {
    double *tsum, *tstart;
    u64 *tT, *Rt, *Pt;
    int *tk, *tbase;
    SeqX* xlist;
    unsigned long long* nseq;
    unsigned *xcount, *need;
};
static inline SeqScratch seq_scratch_carve(void* scratch, const i64 n, const int islands)
{
    const size_t nt = (size_t)islands * (size_t)((n + SEQ_TILE - 1) / SEQ_TILE);
    SeqScratch q;
    char* p = (char*)scratch;
    q.tsum = (double*)p; p += nt * 8;
    q.tstart = (double*)p; p += nt * 8;
    q.tT = (u64*)p; p += nt * 8;
    q.Rt = (u64*)p; p += nt * 8;
    q.Pt = (u64*)p; p += nt * 8;
    q.tk = (int*)p; p += nt * 4;
    q.tbase = (int*)p; p += nt * 4;
    q.xlist = (SeqX*)p; p += (size_t)islands * SEQ_XCAP * sizeof(SeqX);
    q.nseq = (unsigned long long*)p; p += (size_t)islands * 8;
    q.xcount = (unsigned*)p; p += (size_t)islands * 4;
    q.need = (unsigned*)p;
    return q;
}
// (where the walk's count of tiles it did exactly sits in the scratch: one u64 per island; the fallback flag)
static inline const unsigned long long* seq_nseq_ptr(const void* scratch, const i64 n, const int islands)
{
    return seq_scratch_carve((void*)scratch, n, islands).nseq;
}
static inline const unsigned* seq_need_ptr(const void* scratch, const i64 n, const int islands)
{
    return seq_scratch_carve((void*)scratch, n, islands).need;
}
// S <- the reference's sequential fp64 prefix sums of W (both (islands, n), S may not alias W).  `scratch`: seq_scratch_bytes;
// zero_counters = false: the caller zeroed it once (the exception counters are re-armed by the passes themselves).
// tiles_only: the tile walk alone (tests: the fallback must give the same bits)
static inline void seq_prefix_sums_launch(hipStream_t st, const double* W, const i64 n, const int islands, double* S, void* scratch,
                                          const SeqGate gate, const bool tiles_only = false, const bool zero_counters = true,
                                          const bool have_tile_sums = false)
{
    const int ntiles = (int)((n + SEQ_TILE - 1) / SEQ_TILE);
    const SeqScratch q = seq_scratch_carve(scratch, n, islands);
    if (zero_counters) (void)hipMemsetAsync(q.xcount, 0, (size_t)islands * 8, st);       // (xcount and need: adjacent)
    // (have_tile_sums: the caller's kernel that wrote W left the tiles' fp64 sums in the scratch's first array)
    if (!have_tile_sums) SMC_LAUNCH(k_seq_tile_sums, dim3(ntiles, islands), dim3(SMC_BLOCK), st, W, n, q.tsum, gate);
    SeqGate fb = gate;
    if (!tiles_only) {
        SMC_LAUNCH(k_seq_elem_classify, dim3(ntiles, islands), dim3(SMC_BLOCK), st, W, n, (const double*)q.tsum, q.Rt, q.xlist,
                   q.xcount, q.need, gate);
        SMC_LAUNCH(k_seq_elem_chain, dim3(islands), dim3(SMC_BLOCK), st, W, n, ntiles, (const u64*)q.Rt, q.Pt, q.xlist, q.xcount,
                   q.need, q.tbase, S, gate);
        SMC_LAUNCH(k_seq_elem_fill, dim3(ntiles, islands), dim3(SMC_BLOCK), st, W, n, (const double*)q.tsum, (const u64*)q.Pt,
                   (const SeqX*)q.xlist, q.need, (const int*)q.tbase, S, gate);
        fb.only_if = q.need;                                   // only where the fast path gave up
        SMC_LAUNCH(k_seq_fallback, dim3(islands), dim3(SMC_BLOCK), st, W, n, S, fb);
        return;
    }
    SMC_LAUNCH(k_seq_tile_classify, dim3(ntiles, islands), dim3(SMC_BLOCK), st, W, n, (const double*)q.tsum, q.tk, q.tT, fb);
    SMC_LAUNCH(k_seq_chain, dim3(islands), dim3(SMC_BLOCK), st, W, n, ntiles, q.tk, (const u64*)q.tT, q.tstart, S, q.nseq, fb);
    SMC_LAUNCH(k_seq_fill, dim3(ntiles, islands), dim3(SMC_BLOCK), st, W, n, (const int*)q.tk, (const double*)q.tstart, S, fb);
}
