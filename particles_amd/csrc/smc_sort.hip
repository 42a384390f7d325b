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
    hipError_t e = hipMalloc(&buf, 2 * nb + (tb ? tb : 8));
    if (e != hipSuccess) {
        smc_set_error("smc_argsort: %zu bytes: %s", 2 * nb + tb, hipGetErrorString(e));
        return SMC_ERR_NOMEM;
    }
    double* ks = (double*)buf;
    i64* idx = (i64*)((char*)buf + nb);
    void* tmp = (char*)buf + 2 * nb;
    SMC_LAUNCH(k_iota_i64, dim3((unsigned)((N + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK), st,
               (i64)N, idx);
    int rc = SMC_OK;
    if (hipcub::DeviceRadixSort::SortPairs(tmp, tb, x, ks, (const i64*)idx, (i64*)out, (int)N, 0, 64,
                                           st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        smc_set_error("smc_argsort: HIP error: %s", hipGetErrorString(hipGetLastError()));
        rc = SMC_ERR_HIP;
    }
    (void)hipFree(buf);
    return rc;
#endif
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
