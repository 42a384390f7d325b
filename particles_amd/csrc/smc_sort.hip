// smc_sort.hip -- device sorts of the path's callers: np.argsort of the first QMC coordinate and
// the Hilbert sort of SQMC (particles/core.py:339-349, hilbert.py:33-58), weighted quantiles
// (particles/resampling.py:381-417 wquantiles: argsort, running sum of the weights in that order,
// searchsorted + np.interp between the two neighbours of each level).
//
// The sort is a hand-written stable LSD radix sort for gfx950 (wave64), 8 bits per pass over
// order-preserving 64-bit images of the keys (fp64: sign flip / complement; int64: sign flip),
// (key, 64-bit payload) pairs, three kernels per pass:
//   k_rs_hist     per tile of 2048 keys: digit counts through LDS atomics -> hist[256][tiles]
//   k_rs_scan     one workgroup per digit: prefix sum of its row (keys of this digit in earlier tiles)
//                 and the row's total; the scatter adds the totals of the smaller digits
//                 -> the first output slot of every (digit, tile)
//   k_rs_scatter  per tile: a wave takes 512 consecutive keys in 8 chunks of 64; the rank of a key
//                 among the EARLIER keys of its digit is (keys of that digit in the wave's
//                 earlier chunks: an LDS counter) + (lanes below it with the same digit: 8
//                 ballots build the mask of equal digits, one popcount ranks) -- no sorting
//                 network, no atomics, stable by construction; the tile is laid out in sorted
//                 order in LDS (32 KB) and written out slot by slot, so that consecutive lanes
//                 write the consecutive 8-byte slots of a digit's run.
// HBM traffic per pass: read keys twice, read payload once, write both: 40 B per element (the first pass
// reads the caller's keys and generates the index payload; an argsort's last pass writes no keys).
// Beyond one workgroup's 8192 keys the eight passes become FOUR plus a fix-up (round 5): the keys' images differ only below
// their highest varying bit (one pass of maxima: k_rs_minmax); the 32 bits from there down are sorted by four stable
// passes, after which keys that agree on those 32 bits -- and on everything above -- sit together in input order, and
// k_rs_fix orders each such group by the remaining low bits (groups are pairs, rarely: N^2 / 2^33 expected collisions
// over the range of the data; runs of fully equal keys need nothing and may be any length).  A group reaching more than 64
// keys to either side of one of its members, with different low bits (a cluster 2^-32 of the data's range wide) raises a flag and ONE workgroup redoes the
// sort with all eight passes (k_rs_fallback: milliseconds; correct for any input).  Same permutation as before: a stable
// sort by the full key.
// The running sum of wquantiles is a three-kernel scan (tile sums, scan of the sums, apply).
// Everything also runs under the fiber emulator (tests/emu), so the CPU suite exercises it.
#include "smc_internal.h"
#include "smc_device.h"
#include <atomic>
#include <vector>

#define RS_TILE 2048
#define RS_SEG (RS_TILE / SMC_NWAVE)       /* keys per wave */
#define RS_CH (RS_SEG / 64)                /* chunks of 64 per wave */

#ifdef SMC_EMULATE
__device__ inline u64 smc_ballot(bool p) { return __ballot(p ? 1 : 0); }
__device__ inline void smc_wave_lockstep() { emu_wave_sync(); }
#else
__device__ __forceinline__ u64 smc_ballot(bool p) { return __ballot(p ? 1 : 0); }
__device__ __forceinline__ void smc_wave_lockstep() {}     // a wavefront executes in lock step
#endif

enum { RS_KEY_F64 = 0, RS_KEY_I64 = 1 };
// order-preserving map onto unsigned 64-bit integers
__host__ __device__ __forceinline__ u64 rs_encode(u64 bits, int kind)
{
    if (kind == RS_KEY_I64) return bits ^ 0x8000000000000000ull;
    return (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);      // fp64: -x < +x, NaN (+) last
}
__host__ __device__ __forceinline__ u64 rs_decode(u64 e, int kind)
{
    if (kind == RS_KEY_I64) return e ^ 0x8000000000000000ull;
    return (e >> 63) ? (e & 0x7fffffffffffffffull) : ~e;
}

// plan[0] = max of the key images, plan[1] = max of their complements (= ~min), plan[2] = fix-up overflow flag; all zero
// between sorts.  The sorted window: 32 bits from the highest bit in which any two keys differ.
// (Measured and dropped: the maxima spread over 32 slots each, to take the 1024 atomics off two words -- k_rs_minmax 10.6
//  -> 7.8 us, but the nine kernels that then combine 64 words in their prologue lost 1.3 - 2 us EACH.)
#define RS_PLAN_WORDS 3
__device__ __forceinline__ int rs_window_shift(const u64* plan)
{
    const u64 x = plan[0] ^ ~plan[1];
    const int top = x ? 63 - __clzll((long long)x) : 0;
    return top > 31 ? top - 31 : 0;
}
__device__ __forceinline__ int rs_pass_shift(const int shift, const u64* plan) { return plan ? rs_window_shift(plan) + shift : shift; }

// workgroups of 1024 threads, 8192 keys each: a quarter of the atomics on the two words for the same bytes in flight
#define RS_MM_BLOCK 1024
#define RS_MM_TILE (8 * RS_MM_BLOCK)
__global__ void __launch_bounds__(RS_MM_BLOCK)
k_rs_minmax(const u64* keys, i64 N, int kind, u64* plan)
{
    __shared__ u64 s_m[2 * (RS_MM_BLOCK / 64)];
    const int tid = (int)threadIdx.x, wave = tid >> 6;
    const i64 base = (i64)blockIdx.x * RS_MM_TILE;
    u64 mx = 0ull, mn = 0ull;
#pragma unroll
    for (int c = 0; c < RS_MM_TILE / RS_MM_BLOCK; ++c) {
        const i64 i = base + (i64)c * RS_MM_BLOCK + tid;
        if (i < N) {
            const u64 e = rs_encode(keys[i], kind);
            mx = e > mx ? e : mx;
            mn = ~e > mn ? ~e : mn;
        }
    }
    // (an inclusive max-scan over the wave on the DPP path: lane 63 holds the wave's maximum)
#define RS_MAX_STEP(CTRL, MASK)                                     \
    {                                                               \
        const u64 a = smc_dpp64<CTRL, MASK, true>(mx), b = smc_dpp64<CTRL, MASK, true>(mn); \
        mx = a > mx ? a : mx;                                       \
        mn = b > mn ? b : mn;                                       \
    }
    RS_MAX_STEP(SMC_DPP_ROW_SHR(1), 0xf)
    RS_MAX_STEP(SMC_DPP_ROW_SHR(2), 0xf)
    RS_MAX_STEP(SMC_DPP_ROW_SHR(4), 0xf)
    RS_MAX_STEP(SMC_DPP_ROW_SHR(8), 0xf)
    RS_MAX_STEP(SMC_DPP_ROW_BCAST15, 0xa)
    RS_MAX_STEP(SMC_DPP_ROW_BCAST31, 0xc)
#undef RS_MAX_STEP
    if ((tid & 63) == 63) { s_m[wave] = mx; s_m[RS_MM_BLOCK / 64 + wave] = mn; }
    __syncthreads();
    if (tid < 2) {
        u64 m = s_m[tid * (RS_MM_BLOCK / 64)];
        for (int w = 1; w < RS_MM_BLOCK / 64; ++w) m = s_m[tid * (RS_MM_BLOCK / 64) + w] > m ? s_m[tid * (RS_MM_BLOCK / 64) + w] : m;
        atomicMax(reinterpret_cast<unsigned long long*>(plan + tid), (unsigned long long)m);
    }
}

__global__ void __launch_bounds__(SMC_BLOCK)
k_rs_decode(const u64* ek, i64 N, int kind, u64* keys)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < N) keys[i] = rs_decode(ek[i], kind);
}

// hist[digit][tile] (digit-major: the scan below is then a plain prefix sum of one flat array per
// digit row).  RAW (the first pass): the caller's keys, mapped to their sortable images on the fly -- no
// encode pass, no copy of the input.
// (Rounds 1-2 also accumulated the 256 digit totals of the whole input here with global atomics: 512
//  workgroups x 256 atomics on one kilobyte cost 9 of the kernel's 14 us at N = 2^20; the scan's
//  workgroups now leave their row totals and the scatter adds up the smaller digits' itself.)
// the digit counts of tile `tile` into h[256] (LDS, zeroed by the caller, a barrier behind it)
template <bool RAW>
__device__ __forceinline__ void rs_tile_hist(const u64* keys, i64 N, int shift, int kind, int tile, unsigned* h)
{
    const int tid = (int)threadIdx.x;
    const i64 base = (i64)tile * RS_TILE;
    const int lane = smc_lane();
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    u64 kk[RS_TILE / SMC_BLOCK];
#pragma unroll
    for (int c = 0; c < RS_TILE / SMC_BLOCK; ++c) {
        const i64 i = base + (i64)c * SMC_BLOCK + tid;
        kk[c] = (i < N) ? keys[i] : 0ull;
    }
#pragma unroll
    for (int c = 0; c < RS_TILE / SMC_BLOCK; ++c) {
        const i64 i = base + (i64)c * SMC_BLOCK + tid;
        const bool valid = i < N;
        const u64 e = RAW ? rs_encode(kk[c], kind) : kk[c];
        const unsigned dg = valid ? (unsigned)(e >> shift) & 255u : 0u;
        // one LDS atomic per distinct digit of the wave (the exponent bytes of fp64 keys take a
        // handful of values: per-key atomics would serialise on them)
        u64 mask = smc_ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const bool mine = (dg >> bit) & 1u;
            const u64 bb = smc_ballot(mine);
            mask &= mine ? bb : ~bb;
        }
        if (valid && (mask & lt) == 0ull) atomicAdd(&h[dg], (unsigned)__popcll(mask));
    }
}

template <bool RAW>
__global__ void __launch_bounds__(SMC_BLOCK)
k_rs_hist(const u64* keys, i64 N, int shift0, int kind, unsigned* hist, int ntiles, const u64* plan)
{
    __shared__ unsigned h[256];
    const int tid = (int)threadIdx.x;
    const int shift = rs_pass_shift(shift0, plan);
    h[tid] = 0u;
    __syncthreads();
    rs_tile_hist<RAW>(keys, N, shift, kind, (int)blockIdx.x, h);
    __syncthreads();
    hist[(i64)tid * ntiles + blockIdx.x] = h[tid];
}

// one workgroup per digit: the exclusive prefix sum of its row in place (keys with this digit in earlier
// tiles) and the row's total; the scatter adds the totals of the smaller digits
__global__ void __launch_bounds__(SMC_BLOCK)
k_rs_scan(unsigned* hist, unsigned* rowtot, int ntiles)
{
    __shared__ u64 smu[SMC_SM];
    const int d = (int)blockIdx.x, tid = (int)threadIdx.x;
    u64 carry = 0ull;
    unsigned* row = hist + (i64)d * ntiles;
    for (int w0 = 0; w0 < ntiles; w0 += SMC_BLOCK) {
        const int w = w0 + tid;
        const u64 c = (w < ntiles) ? (u64)row[w] : 0ull;
        __syncthreads();
        u64 tot;
        const u64 ex = smc_block_exscan_u64(c, smu, tot);
        if (w < ntiles) row[w] = (unsigned)(carry + ex);
        carry += tot;
    }
    if (tid == 0) rowtot[d] = (unsigned)carry;
}

// RAW (the first pass): the caller's keys and -- vals null -- the index as payload.  okeys null (a last
// pass whose caller wants the permutation only): the keys are not written.
// MODE 0: offs = the scanned histogram, rowtot = the digit totals (k_rs_scan).  Below that the passes are bound by
// the latency chains of their launches (24 kernels of 5 - 9 us for the sort of 2^13 .. 2^17 keys), so one kernel
// fewer per pass: MODE 1 (<= RS_FEW tiles): offs = the RAW histogram, thread d adds up digit d's short row itself.
// (Measured and dropped: no histogram kernel either, every workgroup counting the digits of ALL <= 8 tiles -- the
//  eight dependent tile reads cost more than the launch they save: 0.207 ms per SQMC step at N = 2^14 against
//  0.15.)
#define RS_FEW 64
// MODE 2 (k_rs_fallback: one workgroup walking the tiles in order): my_first = the first output slot of (digit tid, this
// tile), handed in; returns the tile's count of digit tid.
template <bool RAW, int MODE>
__device__ __forceinline__ unsigned rs_scatter_tile(const u64* keys, const u64* vals, i64 N, int shift, int kind, const unsigned* offs,
                                                    const unsigned* rowtot, int ntiles, u64* okeys, u64* ovals, const int tile,
                                                    const unsigned my_first = 0u)
{
    __shared__ unsigned cnt[SMC_NWAVE][256];
    __shared__ unsigned start[256];              // first slot of each digit in the tile's sorted order
    __shared__ unsigned goff[256];               // first output slot of (digit, this tile)
    __shared__ u64 sk[RS_TILE], sv[RS_TILE];     // the tile in sorted order
    __shared__ u64 smu[SMC_SM];
    const int tid = (int)threadIdx.x, lane = smc_lane(), wave = smc_wave();
    if (MODE == 2) __syncthreads();              // (the previous tile's write-out has read sk / sv / goff / start)
#pragma unroll
    for (int w = 0; w < SMC_NWAVE; ++w) cnt[w][tid] = 0u;
    unsigned my_off = 0u, my_tot = 0u;
    if constexpr (MODE == 0) {
        my_off = offs[(i64)tid * ntiles + tile];
        my_tot = rowtot[tid];
    } else if constexpr (MODE == 1) {
        const unsigned* row = offs + (i64)tid * ntiles;
        for (int w = 0; w < ntiles; ++w) {
            const unsigned c = row[w];
            my_off += (w < tile) ? c : 0u;
            my_tot += c;
        }
    }
    const i64 tile0 = (i64)tile * RS_TILE;
    const i64 base = tile0 + (i64)wave * RS_SEG;
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));       // lanes below this one
    u64 k[RS_CH], v[RS_CH];
    unsigned lrank[RS_CH];
#pragma unroll
    for (int c = 0; c < RS_CH; ++c) {
        const i64 i = base + (i64)c * 64 + lane;
        const bool valid = i < N;
        k[c] = valid ? keys[i] : ~0ull;
        v[c] = (RAW && !vals) ? (u64)i : (valid ? vals[i] : 0ull);
    }
    if constexpr (MODE == 2) {
        goff[tid] = my_first;
    } else {   // keys with a smaller digit anywhere in the input
        u64 all;
        const u64 lower = smc_block_exscan_u64((u64)my_tot, smu, all);
        goff[tid] = (unsigned)lower + my_off;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < RS_CH; ++c) {
        const bool valid = base + (i64)c * 64 + lane < N;
        if (RAW) k[c] = valid ? rs_encode(k[c], kind) : ~0ull;
        const unsigned dg = (unsigned)(k[c] >> shift) & 255u;
        u64 mask = smc_ballot(valid);                // lanes holding a key with THIS lane's digit
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const bool mine = (dg >> bit) & 1u;
            const u64 bb = smc_ballot(mine);
            mask &= mine ? bb : ~bb;
        }
        const unsigned before = cnt[wave][dg];       // keys of this digit in the wave's earlier chunks
        smc_wave_lockstep();
        lrank[c] = before + (unsigned)__popcll(mask & lt);
        if (valid && (mask & lt) == 0ull) cnt[wave][dg] = before + (unsigned)__popcll(mask);
        smc_wave_lockstep();
    }
    __syncthreads();
    unsigned tile_count = 0u;
    {   // digit tid: its slots in the tile's sorted order start after all smaller digits; per
        // wave, after the earlier waves' keys of the same digit
        unsigned c4[SMC_NWAVE], tot = 0u;
#pragma unroll
        for (int w = 0; w < SMC_NWAVE; ++w) { c4[w] = cnt[w][tid]; tot += c4[w]; }
        tile_count = tot;
        u64 all;
        unsigned run = (unsigned)smc_block_exscan_u64((u64)tot, smu, all);
        start[tid] = run;
#pragma unroll
        for (int w = 0; w < SMC_NWAVE; ++w) { cnt[w][tid] = run; run += c4[w]; }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < RS_CH; ++c) {
        if (base + (i64)c * 64 + lane < N) {
            const unsigned dg = (unsigned)(k[c] >> shift) & 255u;
            const unsigned p = cnt[wave][dg] + lrank[c];
            sk[p] = k[c];
            sv[p] = v[c];
        }
    }
    __syncthreads();
    // out in sorted order: consecutive threads write consecutive slots of a digit's run
    const i64 nin = (N - tile0 < RS_TILE) ? (N - tile0) : RS_TILE;
#pragma unroll
    for (int c = 0; c < RS_TILE / SMC_BLOCK; ++c) {
        const int j = c * SMC_BLOCK + tid;
        if (j < nin) {
            const u64 kk = sk[j];
            const unsigned dg = (unsigned)(kk >> shift) & 255u;
            const i64 pos = (i64)goff[dg] + (j - (int)start[dg]);
            if (okeys) okeys[pos] = kk;
            ovals[pos] = sv[j];
        }
    }
    return tile_count;
}
template <bool RAW, int MODE>
__global__ void __launch_bounds__(SMC_BLOCK)
k_rs_scatter(const u64* keys, const u64* vals, i64 N, int shift0, int kind, const unsigned* offs,
             const unsigned* rowtot, int ntiles, u64* okeys, u64* ovals, const u64* plan)
{
    (void)rs_scatter_tile<RAW, MODE>(keys, vals, N, rs_pass_shift(shift0, plan), kind, offs, rowtot, ntiles, okeys, ovals, (int)blockIdx.x);
}

// ---- the fix-up behind the four window passes: keys that agree on the window (and above) sit together in input order;
// within such a group, order by the low bits (stable).  new position = i - #{j < i in the group: low_j > low_i}
//                                                                      + #{j > i in the group: low_j < low_i}.
// A thread looks RS_GCAP keys to either side; a group that reaches further is left where it is if nothing in sight has
// other low bits (a run of equal keys, any length) -- the keys that DO differ sit in the same group, see the same long
// group from where they are and raise the flag.
#define RS_GCAP 64
// (round 6: the tile's keys go through LDS with one key of halo on either side -- each key is read from memory ONCE, by
//  16-byte accesses, and nearly every pair leaves by ONE 16-byte store; the version that read every key three times with
//  8-byte accesses behind exec-mask branches took 22 us of the SQMC step at N = 2^20, profiles/r15_sqmc_step_trace.txt)
__global__ void __launch_bounds__(SMC_BLOCK)
k_rs_fix(const u64* keys, const u64* vals, i64 N, u64* okeys, u64* ovals, u64* plan)
{
    __shared__ u64 sk[RS_TILE + 2];
    constexpr int NP = RS_TILE / (2 * SMC_BLOCK);              // pairs per thread
    const int tid = (int)threadIdx.x;
    const i64 tile0 = (i64)blockIdx.x * RS_TILE;
    const int sh = rs_window_shift(plan);
    const u64 lowmask = sh ? ((1ull << sh) - 1ull) : 0ull;
    const bool even = (N & 1) == 0;                            // (pairs are 16-byte aligned in all four arrays)
    u64 k[2 * NP], v[2 * NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const i64 i = tile0 + (i64)c * 2 * SMC_BLOCK + 2 * tid;
        k[2 * c] = k[2 * c + 1] = v[2 * c] = v[2 * c + 1] = 0ull;
        if (even && i + 1 < N) {
            smc_ld2g(keys + i, k[2 * c], k[2 * c + 1]);
            smc_ld2g(vals + i, v[2 * c], v[2 * c + 1]);
        } else {
            if (i < N) { k[2 * c] = keys[i]; v[2 * c] = vals[i]; }
            if (i + 1 < N) { k[2 * c + 1] = keys[i + 1]; v[2 * c + 1] = vals[i + 1]; }
        }
    }
    if (tid == 0) sk[0] = tile0 > 0 ? keys[tile0 - 1] : 0ull;
    if (tid == 1) sk[RS_TILE + 1] = tile0 + RS_TILE < N ? keys[tile0 + RS_TILE] : 0ull;
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const int j = c * 2 * SMC_BLOCK + 2 * tid;
        sk[1 + j] = k[2 * c];
        sk[2 + j] = k[2 * c + 1];
    }
    __syncthreads();
    bool over = false;
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const int j0 = c * 2 * SMC_BLOCK + 2 * tid;
        i64 pos[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = j0 + e;
            const i64 i = tile0 + j;
            pos[e] = i;
            if (i >= N) continue;
            const u64 kk = k[2 * c + e];
            const u64 hi = kk >> sh, low = kk & lowmask;
            const bool left = i > 0 && (sk[j] >> sh) == hi, right = i + 1 < N && (sk[j + 2] >> sh) == hi;
            if (sh && (left || right)) {
                int before = 0, after = 0;
                bool foundL = !left, foundR = !right, other = false;
                for (int s_ = 1; left && s_ <= RS_GCAP; ++s_) {
                    if (i - s_ < 0) { foundL = true; break; }
                    const u64 q = keys[i - s_];
                    if ((q >> sh) != hi) { foundL = true; break; }
                    const u64 ql = q & lowmask;
                    before += ql > low ? 1 : 0;
                    other = other || ql != low;
                }
                for (int s_ = 1; right && s_ <= RS_GCAP; ++s_) {
                    if (i + s_ >= N) { foundR = true; break; }
                    const u64 q = keys[i + s_];
                    if ((q >> sh) != hi) { foundR = true; break; }
                    const u64 ql = q & lowmask;
                    after += ql < low ? 1 : 0;
                    other = other || ql != low;
                }
                if (foundL && foundR) pos[e] = i - before + after;
                else over = over || other;
            }
        }
        const i64 i0 = tile0 + j0;
        if (even && i0 + 1 < N && pos[0] == i0 && pos[1] == i0 + 1) {
            smc_st2g(okeys + i0, k[2 * c], k[2 * c + 1]);
            smc_st2g(ovals + i0, v[2 * c], v[2 * c + 1]);
        } else {
            if (i0 < N) { okeys[pos[0]] = k[2 * c]; ovals[pos[0]] = v[2 * c]; }
            if (i0 + 1 < N) { okeys[pos[1]] = k[2 * c + 1]; ovals[pos[1]] = v[2 * c + 1]; }
        }
    }
    if (over) plan[2] = 1ull;                   // (benign race: every writer stores 1)
}

// ---- the whole sort by ONE workgroup, eight passes over the full keys: run only when k_rs_fix raised its flag (the same
// launch returns at once otherwise).  Pass p reads what pass p - 1 wrote; the result lands where k_rs_fix's would have.
__global__ void __launch_bounds__(SMC_BLOCK)
k_rs_fallback(const u64* keys, const u64* vals, i64 N, int kind, u64* kA, u64* vA, u64* kB, u64* vB, u64* plan)
{
    __shared__ unsigned h[256];
    __shared__ u64 smu[SMC_SM];
    const int tid = (int)threadIdx.x;
    const bool needed = plan[2] != 0ull;
    __syncthreads();
    if (tid < RS_PLAN_WORDS) plan[tid] = 0ull;                             // (the last launch of a sort leaves the plan ready for the next)
    if (!needed) return;
    const int ntiles = (int)((N + RS_TILE - 1) / RS_TILE);
    const u64 *ks = keys, *vs = vals;
    u64 *kd = kA, *vd = vA;
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 8 * pass;
        h[tid] = 0u;
        __syncthreads();
        for (int t = 0; t < ntiles; ++t) {
            if (pass == 0) rs_tile_hist<true>(ks, N, shift, kind, t, h);
            else rs_tile_hist<false>(ks, N, shift, kind, t, h);
        }
        __syncthreads();
        u64 all;
        unsigned first = (unsigned)smc_block_exscan_u64((u64)h[tid], smu, all);   // keys with a smaller digit
        for (int t = 0; t < ntiles; ++t) {
            const unsigned c = pass == 0 ? rs_scatter_tile<true, 2>(ks, vs, N, shift, kind, nullptr, nullptr, ntiles, kd, vd, t, first)
                                         : rs_scatter_tile<false, 2>(ks, vs, N, shift, kind, nullptr, nullptr, ntiles, kd, vd, t, first);
            first += c;
        }
        __threadfence();
        __syncthreads();
        ks = kd; vs = vd;
        if (kd == kA) { kd = kB; vd = vB; } else { kd = kA; vd = vA; }
    }
}

// N <= TILE: the whole sort in ONE launch by one workgroup -- the 8 passes of the same ranking scheme with the
// tile ping-ponging between registers and LDS (24 launches of ~9 us each, bound by their own latency chains,
// would otherwise be the cost of sorting a few thousand keys: SQMC at small N).  TILE = 2048 (any payload; DECODED
// keys out -- the operators' entry points write straight to the caller's arrays); TILE = 4096 / 8192 (PV = u32:
// argsort only, the index as payload held in 32 bits so that the tile fits the CU's LDS; key IMAGES out, the
// contract of smc_rs_sort_ws beyond 2048 keys).
template <int TILE, typename PV>
__global__ void __launch_bounds__(SMC_BLOCK)
k_rs_small(const u64* keys, const u64* vals, i64 N, int kind, u64* okeys, u64* ovals)
{
    constexpr int SEG = TILE / SMC_NWAVE, CH = SEG / 64;
    constexpr bool IMAGES = TILE > RS_TILE;
    __shared__ unsigned cnt[SMC_NWAVE][256];
    __shared__ u64 sk[TILE];
    __shared__ PV sv[TILE];
    __shared__ u64 smu[SMC_SM];
    const int tid = (int)threadIdx.x, lane = smc_lane(), wave = smc_wave();
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int base = wave * SEG;
    u64 k[CH];
    PV v[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = base + c * 64 + lane;
        const bool valid = i < N;
        k[c] = valid ? rs_encode(keys[i], kind) : ~0ull;
        v[c] = valid ? ((sizeof(PV) == 8 && vals) ? (PV)vals[i] : (PV)i) : (PV)0;
    }
    for (int shift = 0; shift < 64; shift += 8) {
#pragma unroll
        for (int w = 0; w < SMC_NWAVE; ++w) cnt[w][tid] = 0u;
        __syncthreads();
        unsigned lrank[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const bool valid = base + c * 64 + lane < N;
            const unsigned dg = (unsigned)(k[c] >> shift) & 255u;
            u64 mask = smc_ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool mine = (dg >> bit) & 1u;
                const u64 bb = smc_ballot(mine);
                mask &= mine ? bb : ~bb;
            }
            const unsigned before = cnt[wave][dg];
            smc_wave_lockstep();
            lrank[c] = before + (unsigned)__popcll(mask & lt);
            if (valid && (mask & lt) == 0ull) cnt[wave][dg] = before + (unsigned)__popcll(mask);
            smc_wave_lockstep();
        }
        __syncthreads();
        {
            unsigned c4[SMC_NWAVE], tot = 0u;
#pragma unroll
            for (int w = 0; w < SMC_NWAVE; ++w) { c4[w] = cnt[w][tid]; tot += c4[w]; }
            u64 all;
            unsigned run = (unsigned)smc_block_exscan_u64((u64)tot, smu, all);
#pragma unroll
            for (int w = 0; w < SMC_NWAVE; ++w) { cnt[w][tid] = run; run += c4[w]; }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (base + c * 64 + lane < N) {
                const unsigned dg = (unsigned)(k[c] >> shift) & 255u;
                const unsigned p = cnt[wave][dg] + lrank[c];
                sk[p] = k[c];
                sv[p] = v[c];
            }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CH; ++c) {                 // back into registers, in the new order
            const int i = base + c * 64 + lane;
            k[c] = i < N ? sk[i] : ~0ull;
            v[c] = i < N ? sv[i] : (PV)0;
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = base + c * 64 + lane;
        if (i < N) {
            if (okeys) okeys[i] = IMAGES ? k[c] : rs_decode(k[c], kind);
            if (ovals) ovals[i] = (u64)v[c];
        }
    }
}
#define RS_ONE_WG 8192      /* argsorts up to here: one workgroup (k_rs_small<4096 / 8192, u32>) */

// Workspace form (smc_internal.h): the caller owns `ws` (smc_rs_ws_bytes(N) bytes, reusable from call to
// call in stream order); the sorted payloads (argsort: the permutation, as 64-bit words) and the sorted
// key IMAGES (rs_encode) are left inside it -- no copy-out.  The fused SQMC step of the filter sorts
// through this entry once per time step.
// (tests: the window path from smaller sizes on -- smc_debug_sort_window_min)
static std::atomic<long long> g_rs_window_min{RS_ONE_WG + 1};      // (every sort that takes more than one workgroup: 86 -> 60 us at 2^14, 171 -> 121 at 2^20)
extern "C" int smc_debug_sort_window_min(long long n)
{
    g_rs_window_min = n > RS_ONE_WG ? n : RS_ONE_WG + 1;
    return SMC_OK;
}
size_t smc_rs_ws_bytes(i64 N)
{
    const size_t ntiles = (size_t)((N + RS_TILE - 1) / RS_TILE);
    return 4 * (size_t)N * 8 + ntiles * 256 * 4 + 256 * 4 + 256 + 1024;     // (... + the plan words of the four-pass form)
}
// plan_is_zero: the caller has sorted through this workspace before -- every sort leaves the plan words zeroed -- which
// saves the fill launch per sort.
int smc_rs_sort_ws(smc_ctx* ctx, const void* keys, const void* vals, i64 N, int kind, void* ws,
                   u64** sorted_keys, u64** sorted_vals, bool plan_is_zero)
{
    hipStream_t st = ctx->stream;
    const size_t nb = (size_t)N * 8;
    u64* k0 = (u64*)ws;
    u64* v0 = (u64*)((char*)ws + nb);
    u64* k1 = (u64*)((char*)ws + 2 * nb);
    u64* v1 = (u64*)((char*)ws + 3 * nb);
    int rc = SMC_OK;
    if (N <= RS_TILE) {
        // (k_rs_small hands out DECODED keys: callers of this branch that want images re-encode)
        SMC_LAUNCH((k_rs_small<RS_TILE, u64>), dim3(1), dim3(SMC_BLOCK), st, (const u64*)keys, (const u64*)vals, N, kind, k0, v0);
        if (hipGetLastError() != hipSuccess) rc = SMC_ERR_HIP;
        if (sorted_keys) *sorted_keys = k0;
        if (sorted_vals) *sorted_vals = v0;
    } else if (N <= RS_ONE_WG && !vals) {
        if (N <= 4096) SMC_LAUNCH((k_rs_small<4096, u32>), dim3(1), dim3(SMC_BLOCK), st, (const u64*)keys, (const u64*)nullptr, N, kind, k0, v0);
        else SMC_LAUNCH((k_rs_small<8192, u32>), dim3(1), dim3(SMC_BLOCK), st, (const u64*)keys, (const u64*)nullptr, N, kind, k0, v0);
        if (hipGetLastError() != hipSuccess) rc = SMC_ERR_HIP;
        if (sorted_keys) *sorted_keys = k0;
        if (sorted_vals) *sorted_vals = v0;
    } else if (N >= g_rs_window_min) {
        // four passes over the 32 bits below the keys' highest varying bit, the low bits by k_rs_fix (see the top of the file)
        const int ntiles = (int)((N + RS_TILE - 1) / RS_TILE);
        unsigned* hist = (unsigned*)((char*)ws + 4 * nb);
        unsigned* rowtot = hist + (size_t)ntiles * 256;
        u64* plan = (u64*)(rowtot + 256);
        const int mode = ntiles <= RS_FEW ? 1 : 0;
        if (!plan_is_zero && hipMemsetAsync(plan, 0, RS_PLAN_WORDS * 8, st) != hipSuccess) rc = SMC_ERR_HIP;
        SMC_LAUNCH(k_rs_minmax, dim3((unsigned)((N + RS_MM_TILE - 1) / RS_MM_TILE)), dim3(RS_MM_BLOCK), st, (const u64*)keys, N, kind, plan);
        const u64 *ks = (const u64*)keys, *vs = (const u64*)vals;
        u64 *kd = k1, *vd = v1;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 8 * pass;
            if (pass == 0) SMC_LAUNCH(k_rs_hist<true>, dim3(ntiles), dim3(SMC_BLOCK), st, ks, N, shift, kind, hist, ntiles, (const u64*)plan);
            else SMC_LAUNCH(k_rs_hist<false>, dim3(ntiles), dim3(SMC_BLOCK), st, ks, N, shift, kind, hist, ntiles, (const u64*)plan);
            if (mode == 0) SMC_LAUNCH(k_rs_scan, dim3(256), dim3(SMC_BLOCK), st, hist, rowtot, ntiles);
#define RS_SCATTER(RAWV, MODEV)                                                                                     \
    SMC_LAUNCH((k_rs_scatter<RAWV, MODEV>), dim3(ntiles), dim3(SMC_BLOCK), st, ks, vs, N, shift, kind,               \
               (const unsigned*)hist, (const unsigned*)rowtot, ntiles, kd, vd, (const u64*)plan)
            if (pass == 0) { if (mode == 1) RS_SCATTER(true, 1); else RS_SCATTER(true, 0); }
            else { if (mode == 1) RS_SCATTER(false, 1); else RS_SCATTER(false, 0); }
#undef RS_SCATTER
            ks = kd; vs = vd;
            if (kd == k1) { kd = k0; vd = v0; } else { kd = k1; vd = v1; }
        }
        // (four passes: the window-sorted pairs are in (k0, v0); the fix-up -- and the fallback's eighth pass -- write (k1, v1))
        SMC_LAUNCH(k_rs_fix, dim3(ntiles), dim3(SMC_BLOCK), st, (const u64*)k0, (const u64*)v0, N, k1, v1, plan);
        SMC_LAUNCH(k_rs_fallback, dim3(1), dim3(SMC_BLOCK), st, (const u64*)keys, (const u64*)vals, N, kind, k0, v0, k1, v1, plan);
        if (hipGetLastError() != hipSuccess) rc = SMC_ERR_HIP;
        if (sorted_keys) *sorted_keys = k1;
        if (sorted_vals) *sorted_vals = v1;
    } else {
        const int ntiles = (int)((N + RS_TILE - 1) / RS_TILE);
        unsigned* hist = (unsigned*)((char*)ws + 4 * nb);
        unsigned* rowtot = hist + (size_t)ntiles * 256;           // 256 row totals
        // pass 0 reads the caller's arrays (encoded on the fly, payload = index unless given) into the
        // second pair; 8 passes: the result is back in the first pair.  The last pass writes no keys
        // unless the caller asked for them.
        const u64 *ks = (const u64*)keys, *vs = (const u64*)vals;
        u64 *kd = k1, *vd = v1;
        const int mode = ntiles <= RS_FEW ? 1 : 0;
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 8 * pass;
            if (pass == 0) SMC_LAUNCH(k_rs_hist<true>, dim3(ntiles), dim3(SMC_BLOCK), st, ks, N, shift, kind, hist, ntiles, (const u64*)nullptr);
            else SMC_LAUNCH(k_rs_hist<false>, dim3(ntiles), dim3(SMC_BLOCK), st, ks, N, shift, kind, hist, ntiles, (const u64*)nullptr);
            if (mode == 0) SMC_LAUNCH(k_rs_scan, dim3(256), dim3(SMC_BLOCK), st, hist, rowtot, ntiles);
            u64* ko = (pass == 7 && !sorted_keys) ? nullptr : kd;
#define RS_SCATTER(RAWV, MODEV)                                                                                     \
    SMC_LAUNCH((k_rs_scatter<RAWV, MODEV>), dim3(ntiles), dim3(SMC_BLOCK), st, ks, vs, N, shift, kind,               \
               (const unsigned*)hist, (const unsigned*)rowtot, ntiles, ko, vd, (const u64*)nullptr)
            if (pass == 0) { if (mode == 1) RS_SCATTER(true, 1); else RS_SCATTER(true, 0); }
            else { if (mode == 1) RS_SCATTER(false, 1); else RS_SCATTER(false, 0); }
#undef RS_SCATTER
            ks = kd; vs = vd;
            if (kd == k1) { kd = k0; vd = v0; } else { kd = k1; vd = v1; }
        }
        if (hipGetLastError() != hipSuccess) rc = SMC_ERR_HIP;
        if (sorted_keys) *sorted_keys = k0;                       // (8 passes: back in the first pair)
        if (sorted_vals) *sorted_vals = v0;
    }
#ifdef SMC_EMULATE
    if (hipStreamSynchronize(st) != hipSuccess) rc = SMC_ERR_HIP;
#endif
    return rc;
}

// Stable sort of (key, payload) pairs by key: keys (N) 64-bit patterns of `kind`, vals (N) 64-bit
// payloads or null (payload = index: argsort).  out_keys / out_vals (N each, either may be null).
// Scratch from the context's pool (recycled in stream order).  N < 2^32.
static int rs_sort_pairs(smc_ctx* ctx, const void* keys, const void* vals, i64 N, int kind, void* out_keys,
                         void* out_vals)
{
    hipStream_t st = ctx->stream;
    if (N <= RS_TILE) {
        SMC_LAUNCH((k_rs_small<RS_TILE, u64>), dim3(1), dim3(SMC_BLOCK), st, (const u64*)keys, (const u64*)vals, N, kind,
                   (u64*)out_keys, (u64*)out_vals);
        int rc1 = hipGetLastError() == hipSuccess ? SMC_OK : SMC_ERR_HIP;
#ifdef SMC_EMULATE
        if (hipStreamSynchronize(st) != hipSuccess) rc1 = SMC_ERR_HIP;
#endif
        return rc1;
    }
    void* buf = nullptr;
    if (smc_malloc(ctx, smc_rs_ws_bytes(N), &buf) != SMC_OK) return SMC_ERR_NOMEM;
    u64 *k0 = nullptr, *v0 = nullptr;
    int rc = smc_rs_sort_ws(ctx, keys, vals, N, kind, buf, out_keys ? &k0 : nullptr, &v0);
    const dim3 ge((unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK));
    if (out_keys) SMC_LAUNCH(k_rs_decode, ge, dim3(SMC_BLOCK), st, (const u64*)k0, N, kind, (u64*)out_keys);
    if (out_vals && hipMemcpyAsync(out_vals, v0, (size_t)N * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = SMC_ERR_HIP;
    if (hipGetLastError() != hipSuccess) rc = SMC_ERR_HIP;
#ifdef SMC_EMULATE
    if (hipStreamSynchronize(st) != hipSuccess) rc = SMC_ERR_HIP;
#endif
    (void)smc_free(ctx, buf);
    return rc;
}

// ---- inclusive running sum of N doubles (np.cumsum's role in wquantiles): tile sums, scan of
// the sums by one workgroup, apply.  Fixed association order (tile by tile, lanes as a DPP scan).
#define SC_TILE (SMC_BLOCK * 4)
__global__ void __launch_bounds__(SMC_BLOCK)
k_sc_tile(const double* x, i64 N, double* out, double* tsum, const double* tpre)
{
    __shared__ double sm[SMC_SM];
    const i64 i0 = (i64)blockIdx.x * SC_TILE + (i64)threadIdx.x * 4;
    double v[4], s = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = (i0 + k < N) ? x[i0 + k] : 0.0; s += v[k]; }
    const double inc = smc_wave_scan_add_f64(s);
    __syncthreads();
    if (smc_lane() == 63) sm[smc_wave()] = inc;
    __syncthreads();
    double base = tpre ? tpre[blockIdx.x] : 0.0, tot = 0.0;
#pragma unroll
    for (int w = 0; w < SMC_NWAVE; ++w) {
        if (w < smc_wave()) base += sm[w];
        tot += sm[w];
    }
    if (!tpre) {
        if (threadIdx.x == 0) tsum[blockIdx.x] = tot;
        return;
    }
    double run = base + inc - s;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        run += v[k];
        if (i0 + k < N) out[i0 + k] = run;
    }
}
__global__ void __launch_bounds__(SMC_BLOCK)
k_sc_sums(double* tsum, int nt)                 // exclusive scan of the tile sums, in place, one workgroup
{
    __shared__ double sm[SMC_SM];
    double carry = 0.0;
    for (int b0 = 0; b0 < nt; b0 += SMC_BLOCK) {
        const int i = b0 + (int)threadIdx.x;
        const double v = i < nt ? tsum[i] : 0.0;
        const double inc = smc_wave_scan_add_f64(v);
        __syncthreads();
        if (smc_lane() == 63) sm[smc_wave()] = inc;
        __syncthreads();
        double base = carry, tot = 0.0;
#pragma unroll
        for (int w = 0; w < SMC_NWAVE; ++w) {
            if (w < smc_wave()) base += sm[w];
            tot += sm[w];
        }
        if (i < nt) tsum[i] = base + inc - v;
        carry += tot;
    }
}
static void sc_inclusive_sum(smc_ctx* ctx, const double* x, i64 N, double* out, double* tsum)
{
    const int nt = (int)((N + SC_TILE - 1) / SC_TILE);
    hipStream_t st = ctx->stream;
    SMC_LAUNCH(k_sc_tile, dim3(nt), dim3(SMC_BLOCK), st, x, N, out, tsum, (const double*)nullptr);
    SMC_LAUNCH(k_sc_sums, dim3(1), dim3(SMC_BLOCK), st, tsum, nt);
    SMC_LAUNCH(k_sc_tile, dim3(nt), dim3(SMC_BLOCK), st, x, N, out, tsum, (const double*)tsum);
}

__global__ void k_column(const double* x, i64 N, i64 d, i64 col, double* out)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < N) out[i] = x[i * d + col];
}

// level a -> n = first index with cw[n] >= a (np.searchsorted, side='left'),
// prev = clip(n - 1, 0, N - 2); out[4 j ..] = cw[prev], cw[prev+1], xs[prev], xs[prev+1]
__global__ void k_quantile_probe(const double* cw, const double* xs, i64 N, const double* alphas,
                                 int k, double* out)
{
    const int j = (int)(blockIdx.x * SMC_BLOCK + threadIdx.x);
    if (j >= k) return;
    const double a = alphas[j];
    i64 lo = 0, len = N;
    while (len > 0) {
        const i64 half = len >> 1;
        const bool less = cw[lo + half] < a;
        lo = less ? lo + half + 1 : lo;
        len = less ? len - half - 1 : half;
    }
    i64 prev = lo - 1;
    prev = prev < 0 ? 0 : (prev > N - 2 ? N - 2 : prev);
    out[4 * j + 0] = cw[prev];
    out[4 * j + 1] = cw[prev + 1];
    out[4 * j + 2] = xs[prev];
    out[4 * j + 3] = xs[prev + 1];
}

__global__ void __launch_bounds__(SMC_BLOCK)
k_iota_i64(i64 N, i64* out)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < N) out[i] = i;
}

// np.argsort(x) (hilbert.py:52-54 for d = 1; core.py:342 argsort of the first QMC coordinate):
// radix sort of (x, index) pairs; stable, like np.argsort(kind="stable") on ties
extern "C" int smc_argsort(smc_ctx* ctx, const double* x, int64_t N, int64_t* out)
{
    SMC_REQUIRE(ctx && x && out, "null argument");
    SMC_REQUIRE(N > 0 && N < ((int64_t)1 << 31), "N must be in [1, 2^31)");
    SMC_HIP_CHECK(hipSetDevice(ctx->device));
    const int rc = rs_sort_pairs(ctx, x, nullptr, (i64)N, RS_KEY_F64, nullptr, out);
    if (rc == SMC_ERR_HIP) smc_set_error("smc_argsort: HIP error: %s", hipGetErrorString(hipGetLastError()));
    return rc;
}

// ---------------------------------------------------------------------------
// Hilbert sort (hilbert.py:33-58).  Witham's coordinate codec restated per point with
// integer operations: the coordinates' bits are cut into chunks of d bits (one bit of every
// coordinate, coordinate 0 highest), most significant chunk first; each chunk is decoded
// through a Gray code rotated / reflected by the (start, end) corners of the current sub-cube.
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ i64 hb_gray_encode(i64 bn) { return bn ^ (bn / 2); }      // :198-203
__host__ __device__ __forceinline__ i64 hb_gray_decode(i64 n)                                  // :206-215
{
    i64 sh = 1;
    for (;;) {
        const i64 div = n >> sh;
        n ^= div;
        if (div <= 1) return n;
        sh <<= 1;
    }
}
__host__ __device__ __forceinline__ i64 hb_encode_travel(i64 start, i64 end, i64 mask, i64 i)  // :227-236
{
    const i64 travel_bit = start ^ end, modulus = mask + 1;
    const i64 g = hb_gray_encode(i) * (travel_bit * 2);
    return ((g | (g / modulus)) & mask) ^ start;
}
__host__ __device__ __forceinline__ i64 hb_decode_travel(i64 start, i64 end, i64 mask, i64 g)  // :239-244
{
    const i64 travel_bit = start ^ end, modulus = mask + 1;
    const i64 rg = (g ^ start) * (modulus / (travel_bit * 2));
    return hb_gray_decode((rg | (rg / modulus)) & mask);
}
// Hilbert_to_int (hilbert.py:79-91) of one point with d coordinates c[0..d)
__host__ __device__ inline i64 hb_hilbert_to_int(const i64* c, int d)
{
    i64 biggest = 0;
    for (int k = 0; k < d; ++k) biggest = c[k] > biggest ? c[k] : biggest;
    int nchunks = 0;                                    // ceil(log2(biggest + 1)), at least 1 (:149-156)
    while ((biggest >> nchunks) != 0) ++nchunks;
    if (nchunks < 1) nchunks = 1;
    const i64 mask = ((i64)1 << d) - 1;
    i64 start = 0;                                      // initial_start_end (:94-99)
    int e = (-nchunks - 1) % d;
    if (e < 0) e += d;                                  // Python's %: 0 <= e < d
    i64 end = (i64)1 << e;
    i64 z = 0;
    for (int j = 0; j < nchunks; ++j) {
        // coord chunk j: bit (nchunks-1-j) of every coordinate, coordinate 0 highest (transpose_bits)
        i64 chunk = 0;
        const int bit = nchunks - 1 - j;
        for (int k = 0; k < d; ++k) chunk = chunk * 2 + ((c[k] >> bit) & 1);
        const i64 i = hb_decode_travel(start, end, mask, chunk);
        // pack_index (:134-140), in int64 that wraps like numba's: from d = 4 on, d * nchunks
        // exceeds 63 bits for the largest coordinates and the reference's index goes negative
        z = (i64)(((u64)z << d) + (u64)i);
        // child_start_end (:287-292)
        i64 start_i = (i - 1) & ~(i64)1;
        if (start_i < 0) start_i = 0;
        i64 end_i = (i + 1) | 1;
        if (end_i > mask) end_i = mask;
        const i64 cs = hb_encode_travel(start, end, mask, start_i);
        const i64 ce = hb_encode_travel(start, end, mask, end_i);
        start = cs;
        end = ce;
    }
    return z;
}

#define HB_MAXD 16
// per-column sums (pass 0: of x; pass 1: of |x - mean|^2), one workgroup per column chunk
__global__ void __launch_bounds__(SMC_BLOCK)
k_hb_colsum(const double* x, i64 N, int d, const double* mean, double* part)
{
    __shared__ double sm[SMC_SM];
    const int c = (int)blockIdx.y;
    double acc = 0.0;
    for (i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x; i < N; i += (i64)gridDim.x * SMC_BLOCK) {
        const double v = x[i * d + c];
        if (mean) { const double r = v - mean[c]; acc += r * r; }
        else acc += v;
    }
    acc = smc_block_sum(acc, sm);
    if (threadIdx.x == 0) part[(i64)c * gridDim.x + blockIdx.x] = acc;
}
__global__ void k_hb_colfinal(const double* part, int nb, i64 N, int d, int pass, double* stat)
{
    const int c = (int)threadIdx.x;
    if (c >= d) return;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += part[(i64)c * nb + b];
    stat[pass * HB_MAXD + c] = pass ? sqrt(s / (double)N) : s / (double)N;     // np.std / np.mean
}
__global__ void __launch_bounds__(SMC_BLOCK)
k_hb_keys(const double* x, i64 N, int d, const double* stat, double maxint, i64* keys)
{
    const i64 n = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (n >= N) return;
    i64 c[HB_MAXD];
    for (int k = 0; k < d; ++k) {
        const double s = (x[n * d + k] - stat[k]) / stat[HB_MAXD + k];
        const double xs = 1.0 / (1.0 + exp(-s));                                 // invlogit (:9-10)
        c[k] = (i64)floor(xs * maxint);                                          // :56-57
    }
    keys[n] = hb_hilbert_to_int(c, d);
}

__global__ void __launch_bounds__(SMC_BLOCK)
k_hb_index(const i64* xint, i64 N, int d, i64* out)
{
    const i64 n = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (n >= N) return;
    i64 c[HB_MAXD];
    for (int k = 0; k < d; ++k) c[k] = xint[n * d + k];
    out[n] = hb_hilbert_to_int(c, d);
}

extern "C" int smc_hilbert_array(smc_ctx* ctx, const int64_t* xint, int64_t N, int32_t d, int64_t* out)
{
    SMC_REQUIRE(ctx && xint && out, "null argument");
    SMC_REQUIRE(N > 0 && d >= 1 && d <= HB_MAXD, "smc_hilbert_array: N > 0, 1 <= d <= 16");
    SMC_LAUNCH(k_hb_index, dim3((unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK),
               ctx->stream, (const i64*)xint, (i64)N, (int)d, (i64*)out);
    SMC_LAUNCH_CHECK();
    return SMC_OK;
}

extern "C" int smc_hilbert_sort(smc_ctx* ctx, const double* x, int64_t N, int32_t d, int64_t* out,
                                int64_t* keys_out)
{
    SMC_REQUIRE(ctx && x && out, "null argument");
    SMC_REQUIRE(N > 0 && N < ((int64_t)1 << 31), "N must be in [1, 2^31)");
    SMC_REQUIRE(d >= 2 && d <= HB_MAXD, "smc_hilbert_sort: 2 <= d <= 16 (d = 1: smc_argsort)");
    hipStream_t st = ctx->stream;
    SMC_HIP_CHECK(hipSetDevice(ctx->device));
    const int nb = (int)((N + 4 * SMC_BLOCK - 1) / (4 * SMC_BLOCK) > 256 ? 256
                                                                          : (N + 4 * SMC_BLOCK - 1) / (4 * SMC_BLOCK));
    const size_t nbytes = (size_t)N * 8;
    void* buf = nullptr;
    const size_t small = (size_t)(HB_MAXD * nb + 2 * HB_MAXD) * 8;
    if (smc_malloc(ctx, nbytes + small, &buf) != SMC_OK) return SMC_ERR_NOMEM;
    i64* keys = (i64*)buf;
    double* part = (double*)((char*)buf + nbytes);
    double* stat = part + (size_t)HB_MAXD * nb;
    const double maxint = floor(pow(2.0, 62.0 / (double)d));                     // :55
    SMC_LAUNCH(k_hb_colsum, dim3(nb, d), dim3(SMC_BLOCK), st, x, (i64)N, (int)d, (const double*)nullptr, part);
    SMC_LAUNCH(k_hb_colfinal, dim3(1), dim3(64), st, (const double*)part, nb, (i64)N, (int)d, 0, stat);
    SMC_LAUNCH(k_hb_colsum, dim3(nb, d), dim3(SMC_BLOCK), st, x, (i64)N, (int)d, (const double*)stat, part);
    SMC_LAUNCH(k_hb_colfinal, dim3(1), dim3(64), st, (const double*)part, nb, (i64)N, (int)d, 1, stat);
    SMC_LAUNCH(k_hb_keys, dim3((unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK), st, x,
               (i64)N, (int)d, (const double*)stat, maxint, keys);
    // signed keys: np.argsort of the (possibly wrapped) int64 indices
    int rc = rs_sort_pairs(ctx, keys, nullptr, (i64)N, RS_KEY_I64, nullptr, out);
    if (rc == SMC_OK && keys_out &&
        hipMemcpyAsync(keys_out, keys, nbytes, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = SMC_ERR_HIP;
#ifdef SMC_EMULATE
    if (hipStreamSynchronize(st) != hipSuccess) rc = SMC_ERR_HIP;
#endif
    if (rc == SMC_ERR_HIP) smc_set_error("smc_hilbert_sort: HIP error: %s", hipGetErrorString(hipGetLastError()));
    (void)smc_free(ctx, buf);
    return rc;
}

extern "C" int smc_wquantiles(smc_ctx* ctx, const double* W, const double* x, int64_t N, int64_t d,
                              const double* alphas_host, int k, double* out_host)
{
    SMC_REQUIRE(ctx && W && x && alphas_host && out_host, "null argument");
    SMC_REQUIRE(N >= 2 && d >= 1 && k >= 1, "wquantiles needs N >= 2, d >= 1, k >= 1");
    hipStream_t st = ctx->stream;
    SMC_HIP_CHECK(hipSetDevice(ctx->device));
    // buffers: column, sorted keys, sorted weights, running sums, levels, probes
    double* buf = nullptr;
    const size_t nb = (size_t)N * 8;
    hipError_t e = hipMalloc((void**)&buf, 4 * nb + (size_t)k * 8 * 5);
    if (e != hipSuccess) {
        smc_set_error("smc_wquantiles: %zu bytes: %s", 4 * nb, hipGetErrorString(e));
        return SMC_ERR_NOMEM;
    }
    double *col = buf, *xs = buf + N, *ws = buf + 2 * N, *cw = buf + 3 * N;
    double* al = buf + 4 * N;
    double* pr = al + k;
    int rc = SMC_OK;
    void* tmp = nullptr;
    do {
        if (hipMemcpyAsync(al, alphas_host, (size_t)k * 8, hipMemcpyHostToDevice, st) != hipSuccess) { rc = SMC_ERR_HIP; break; }
        if (hipMalloc(&tmp, (size_t)((N + SC_TILE - 1) / SC_TILE) * 8 + 8) != hipSuccess) { rc = SMC_ERR_NOMEM; break; }
        std::vector<double> probes((size_t)4 * k);
        for (i64 c = 0; c < d && rc == SMC_OK; ++c) {
            const double* keys = x;
            if (d > 1) {
                SMC_LAUNCH(k_column, dim3((unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK), st,
                           x, (i64)N, (i64)d, c, col);
                keys = col;
            }
            // (x, W) pairs sorted by x, then the running sum of the weights in that order
            rc = rs_sort_pairs(ctx, keys, W, (i64)N, RS_KEY_F64, xs, ws);
            if (rc != SMC_OK) break;
            sc_inclusive_sum(ctx, ws, (i64)N, cw, (double*)tmp);
            SMC_LAUNCH(k_quantile_probe, dim3((unsigned)((k + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK), st,
                       (const double*)cw, (const double*)xs, (i64)N, (const double*)al, k, pr);
            if (hipMemcpyAsync(probes.data(), pr, (size_t)4 * k * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) { rc = SMC_ERR_HIP; break; }
            for (int j = 0; j < k; ++j) {                      // np.interp on 2 points
                const double a = alphas_host[j], c0 = probes[4 * j], c1 = probes[4 * j + 1],
                             x0 = probes[4 * j + 2], x1 = probes[4 * j + 3];
                double q;
                if (a <= c0) q = x0;
                else if (a >= c1) q = x1;
                else q = ((x1 - x0) / (c1 - c0)) * (a - c0) + x0;
                out_host[c * k + j] = q;
            }
        }
    } while (0);
    if (rc == SMC_ERR_HIP) smc_set_error("smc_wquantiles: HIP error: %s", hipGetErrorString(hipGetLastError()));
    if (tmp) (void)hipFree(tmp);
    (void)hipFree(buf);
    return rc;
}
