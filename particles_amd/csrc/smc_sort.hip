// smc_sort.hip -- weighted quantiles (particles/resampling.py:381-417 wquantiles,
// _wquantiles): argsort of the particles, running sum of the weights in that
// order, searchsorted + np.interp between the two neighbours of each level.
//
// Not on the per-step path (a collector): the sort and the
// scan are rocPRIM's through hipCUB (device-wide radix sort of (x, W) pairs,
// inclusive sum); the search runs in one small kernel and the 2-point
// interpolation of np.interp on the host from 4 numbers per level.
#include "smc_internal.h"
#include "smc_device.h"
#include <vector>
#ifdef SMC_EMULATE
#include <algorithm>
#include <numeric>
#else
#include <hipcub/hipcub.hpp>
#endif

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
// device-wide radix sort of (x, index) pairs
extern "C" int smc_argsort(smc_ctx* ctx, const double* x, int64_t N, int64_t* out)
{
    SMC_REQUIRE(ctx && x && out, "null argument");
    SMC_REQUIRE(N > 0 && N < ((int64_t)1 << 31), "N must be in [1, 2^31)");
    hipStream_t st = ctx->stream;
    SMC_HIP_CHECK(hipSetDevice(ctx->device));
#ifdef SMC_EMULATE
    {   // test infrastructure: host sort
        SMC_HIP_CHECK(hipStreamSynchronize(st));
        std::vector<i64> o((size_t)N);
        std::iota(o.begin(), o.end(), 0);
        std::stable_sort(o.begin(), o.end(), [&](i64 a, i64 b) { return x[a] < x[b]; });
        for (i64 i = 0; i < N; ++i) out[i] = o[(size_t)i];
        return SMC_OK;
    }
#else
    void* buf = nullptr;
    size_t tb = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, x, (double*)nullptr, (const i64*)nullptr,
                                             (i64*)nullptr, (int)N, 0, 64, st);
    const size_t nb = (size_t)N * 8;
    // scratch from the context's pool: recycled in stream order, no sync, no hipMalloc per call
    if (smc_malloc(ctx, 2 * nb + (tb ? tb : 8), &buf) != SMC_OK) return SMC_ERR_NOMEM;
    double* ks = (double*)buf;
    i64* idx = (i64*)((char*)buf + nb);
    void* tmp = (char*)buf + 2 * nb;
    SMC_LAUNCH(k_iota_i64, dim3((unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK), st,
               (i64)N, idx);
    int rc = SMC_OK;
    if (hipcub::DeviceRadixSort::SortPairs(tmp, tb, x, ks, (const i64*)idx, (i64*)out, (int)N, 0, 64,
                                           st) != hipSuccess) {
        smc_set_error("smc_argsort: HIP error: %s", hipGetErrorString(hipGetLastError()));
        rc = SMC_ERR_HIP;
    }
    (void)smc_free(ctx, buf);
    return rc;
#endif
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
    size_t tb = 0;
#ifndef SMC_EMULATE
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (const i64*)nullptr, (i64*)nullptr,
                                             (const i64*)nullptr, (i64*)nullptr, (int)N, 0, 64, st);
#endif
    void* buf = nullptr;
    const size_t small = (size_t)(HB_MAXD * nb + 2 * HB_MAXD) * 8;
    if (smc_malloc(ctx, 3 * nbytes + small + (tb ? tb : 8), &buf) != SMC_OK) return SMC_ERR_NOMEM;
    i64* keys = (i64*)buf;
    i64* ks = (i64*)((char*)buf + nbytes);
    i64* idx = (i64*)((char*)buf + 2 * nbytes);
    double* part = (double*)((char*)buf + 3 * nbytes);
    double* stat = part + (size_t)HB_MAXD * nb;
    void* tmp = (char*)buf + 3 * nbytes + small;
    const double maxint = floor(pow(2.0, 62.0 / (double)d));                     // :55
    SMC_LAUNCH(k_hb_colsum, dim3(nb, d), dim3(SMC_BLOCK), st, x, (i64)N, (int)d, (const double*)nullptr, part);
    SMC_LAUNCH(k_hb_colfinal, dim3(1), dim3(64), st, (const double*)part, nb, (i64)N, (int)d, 0, stat);
    SMC_LAUNCH(k_hb_colsum, dim3(nb, d), dim3(SMC_BLOCK), st, x, (i64)N, (int)d, (const double*)stat, part);
    SMC_LAUNCH(k_hb_colfinal, dim3(1), dim3(64), st, (const double*)part, nb, (i64)N, (int)d, 1, stat);
    SMC_LAUNCH(k_hb_keys, dim3((unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK), st, x,
               (i64)N, (int)d, (const double*)stat, maxint, keys);
    int rc = SMC_OK;
#ifdef SMC_EMULATE
    {
        (void)ks; (void)idx; (void)tmp;
        std::vector<i64> o((size_t)N);
        std::iota(o.begin(), o.end(), 0);
        std::stable_sort(o.begin(), o.end(), [&](i64 a, i64 b) { return keys[a] < keys[b]; });
        for (i64 i = 0; i < N; ++i) out[i] = o[(size_t)i];
    }
#else
    SMC_LAUNCH(k_iota_i64, dim3((unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK), st,
               (i64)N, idx);
    // signed keys: np.argsort of the (possibly wrapped) int64 indices
    if (hipcub::DeviceRadixSort::SortPairs(tmp, tb, (const i64*)keys, (i64*)ks, (const i64*)idx,
                                           (i64*)out, (int)N, 0, 64, st) != hipSuccess) rc = SMC_ERR_HIP;
#endif
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
#ifndef SMC_EMULATE
        size_t tb_sort = 0, tb_scan = 0;
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb_sort, col, xs, W, ws, (int)N, 0, 64, st);
        (void)hipcub::DeviceScan::InclusiveSum(nullptr, tb_scan, ws, cw, (int)N, st);
        const size_t tb = tb_sort > tb_scan ? tb_sort : tb_scan;
        if (hipMalloc(&tmp, tb ? tb : 8) != hipSuccess) { rc = SMC_ERR_NOMEM; break; }
#endif
        std::vector<double> probes((size_t)4 * k);
        for (i64 c = 0; c < d && rc == SMC_OK; ++c) {
            const double* keys = x;
            if (d > 1) {
                SMC_LAUNCH(k_column, dim3((unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK), st,
                           x, (i64)N, (i64)d, c, col);
                keys = col;
            }
#ifdef SMC_EMULATE
            {   // test infrastructure: host sort (np.argsort + np.cumsum)
                std::vector<i64> o((size_t)N);
                std::iota(o.begin(), o.end(), 0);
                std::stable_sort(o.begin(), o.end(), [&](i64 a, i64 b) { return keys[a] < keys[b]; });
                double run = 0.0;
                for (i64 i = 0; i < N; ++i) { xs[i] = keys[o[i]]; ws[i] = W[o[i]]; run += ws[i]; cw[i] = run; }
            }
#else
            size_t tb1 = tb_sort, tb2 = tb_scan;
            if (hipcub::DeviceRadixSort::SortPairs(tmp, tb1, keys, xs, W, ws, (int)N, 0, 64, st) != hipSuccess ||
                hipcub::DeviceScan::InclusiveSum(tmp, tb2, ws, cw, (int)N, st) != hipSuccess) {
                rc = SMC_ERR_HIP;
                break;
            }
#endif
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
