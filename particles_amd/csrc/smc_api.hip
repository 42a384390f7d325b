// smc_api.hip -- context, memory and error plumbing of libsmc_hip.so
#include <cstdarg>

#include "smc_internal.h"

static thread_local char g_err[512] = "";

void smc_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// A host-to-device copy of a few words (a distribution's scalar parameters, the observation of the step: what the
// template-method step of a user-defined model uploads three times per time step) travels IN THE KERNEL ARGUMENTS of a
// one-wave launch: captured when the launch is enqueued -- the caller's buffer is free on return, as the contract of
// smc_memcpy_h2d says -- and ordered on the stream like everything else, with no hipStreamSynchronize (the pageable
// copy below needs one, and a synchronisation per upload made the host wait for the device three times per step:
// 143 -> 96 us per step of the `generic_model` bench leg).
struct SmcSmallCopy {
    u64 w[32];
};
__global__ void k_put_small(u64* dst, const SmcSmallCopy v, const int nwords)
{
    const int i = (int)threadIdx.x;
    if (i < nwords) dst[i] = v.w[i];
}

extern "C" {

const char* smc_last_error(void) { return g_err; }

const char* smc_version(void)
{
#ifdef SMC_EMULATE
    return "smc_hip 0.1 (CPU EMULATOR - test build, not the product)";
#else
    return "smc_hip 0.1 (gfx950)";
#endif
}

int smc_device_count(int* n_out)
{
    SMC_REQUIRE(n_out, "null output");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *n_out = n;
    return SMC_OK;
}

int smc_ctx_create(int device, uint64_t seed, smc_ctx** out)
{
    SMC_REQUIRE(out, "null output");
    int n = 0;
    SMC_HIP_CHECK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) {
        smc_set_error("smc_ctx_create: device %d out of range (%d visible)", device, n);
        return SMC_ERR_INVALID;
    }
    SMC_HIP_CHECK(hipSetDevice(device));
#ifndef SMC_EMULATE
    // host threads waiting on the stream spin instead of sleeping: a filter's step loop is read
    // back in units of a few hundred microseconds and a sleeping waiter adds tens of them
    // (SMC_SYNC_YIELD=1 restores the runtime's default)
    if (!getenv("SMC_SYNC_YIELD")) (void)hipSetDeviceFlags(hipDeviceScheduleSpin);
    (void)hipGetLastError();
#endif
    smc_ctx* c = new smc_ctx();
    c->device = device;
    c->seed = seed;
    c->scratch = nullptr;
    c->scratch_bytes = 0;
    c->pooled_bytes = 0;
    hipDeviceProp_t prop;
    SMC_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount;
    SMC_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    SMC_HIP_CHECK(hipEventCreate(&c->ev0));
    SMC_HIP_CHECK(hipEventCreate(&c->ev1));
    *out = c;
    return SMC_OK;
}

int smc_ctx_destroy(smc_ctx* ctx)
{
    if (!ctx) return SMC_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    for (auto& kv : ctx->pool)
        for (void* p : kv.second) (void)hipFree(p);
    for (auto& kv : ctx->pinned)
        for (void* p : kv.second) (void)hipHostFree(p);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SMC_OK;
}

int smc_ctx_sync(smc_ctx* ctx)
{
    SMC_REQUIRE(ctx, "null context");
    SMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SMC_OK;
}

int smc_ctx_seed(smc_ctx* ctx, uint64_t seed)
{
    SMC_REQUIRE(ctx, "null context");
    ctx->seed = seed;
    return SMC_OK;
}

int smc_ctx_device_info(smc_ctx* ctx, char* name_host, size_t name_len, int* n_cu,
                        uint64_t* hbm_bytes)
{
    SMC_REQUIRE(ctx, "null context");
    hipDeviceProp_t prop;
    SMC_HIP_CHECK(hipGetDeviceProperties(&prop, ctx->device));
    if (name_host && name_len) snprintf(name_host, name_len, "%s", prop.name);
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)prop.totalGlobalMem;
    return SMC_OK;
}

int smc_ctx_device_pci(smc_ctx* ctx, char* out_host, size_t len)
{
    SMC_REQUIRE(ctx && out_host && len >= 16, "null argument or buffer below 16 bytes");
    SMC_HIP_CHECK(hipDeviceGetPCIBusId(out_host, (int)len, ctx->device));
    return SMC_OK;
}

int smc_malloc(smc_ctx* ctx, size_t bytes, void** dptr_out)
{
    SMC_REQUIRE(ctx && dptr_out, "null argument");
    SMC_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t sz = smc_align_up(bytes ? bytes : 1, 256);
    void* p = nullptr;
    auto it = ctx->pool.find(sz);
    if (it != ctx->pool.end() && !it->second.empty()) {
        p = it->second.back();
        it->second.pop_back();
        ctx->pooled_bytes -= sz;
    } else {
        hipError_t e = hipMalloc(&p, sz);
        if (e != hipSuccess && ctx->pooled_bytes) {          // give the cached blocks back and retry
            (void)hipStreamSynchronize(ctx->stream);
            for (auto& kv : ctx->pool) {
                for (void* q : kv.second) (void)hipFree(q);
                kv.second.clear();
            }
            ctx->pooled_bytes = 0;
            e = hipMalloc(&p, sz);
        }
        if (e != hipSuccess) {
            smc_set_error("smc_malloc: %zu bytes: %s", bytes, hipGetErrorString(e));
            return SMC_ERR_NOMEM;
        }
    }
    ctx->live[p] = sz;
    *dptr_out = p;
    return SMC_OK;
}

int smc_free(smc_ctx* ctx, void* dptr)
{
    SMC_REQUIRE(ctx, "null context");
    if (!dptr) return SMC_OK;
    auto it = ctx->live.find(dptr);
    if (it == ctx->live.end()) {
        smc_set_error("smc_free: %p was not allocated by smc_malloc on this context", dptr);
        return SMC_ERR_INVALID;
    }
    const size_t sz = it->second;
    ctx->live.erase(it);
    if (ctx->pooled_bytes + sz <= SMC_POOL_MAX_BYTES) {
        ctx->pool[sz].push_back(dptr);
        ctx->pooled_bytes += sz;
        return SMC_OK;
    }
    SMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    SMC_HIP_CHECK(hipFree(dptr));
    return SMC_OK;
}

int smc_memcpy_h2d(smc_ctx* ctx, void* dst, const void* src_host, size_t bytes)
{
    SMC_REQUIRE(ctx && (bytes == 0 || (dst && src_host)), "null argument");
    if (!bytes) return SMC_OK;
    if (bytes <= sizeof(SmcSmallCopy) && bytes % 8 == 0 && ((uintptr_t)dst & 7) == 0) {
        SmcSmallCopy v;
        memcpy(v.w, src_host, bytes);
        SMC_LAUNCH(k_put_small, dim3(1), dim3(64), ctx->stream, (u64*)dst, v, (int)(bytes / 8));
        SMC_LAUNCH_CHECK();
        return SMC_OK;
    }
    SMC_HIP_CHECK(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    // pageable host memory: the source may be reused as soon as we return
    SMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SMC_OK;
}

int smc_memcpy_d2h(smc_ctx* ctx, void* dst_host, const void* src, size_t bytes)
{
    SMC_REQUIRE(ctx && (bytes == 0 || (dst_host && src)), "null argument");
    if (!bytes) return SMC_OK;
    SMC_HIP_CHECK(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SMC_OK;
}

int smc_memcpy_d2d(smc_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    SMC_REQUIRE(ctx && (bytes == 0 || (dst && src)), "null argument");
    if (!bytes) return SMC_OK;
    SMC_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return SMC_OK;
}

int smc_memset(smc_ctx* ctx, void* dst, int byte, size_t bytes)
{
    SMC_REQUIRE(ctx && (bytes == 0 || dst), "null argument");
    if (!bytes) return SMC_OK;
    SMC_HIP_CHECK(hipMemsetAsync(dst, byte, bytes, ctx->stream));
    return SMC_OK;
}

int smc_timer_start(smc_ctx* ctx)
{
    SMC_REQUIRE(ctx, "null context");
    SMC_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
    return SMC_OK;
}

int smc_timer_stop(smc_ctx* ctx, float* ms_out)
{
    SMC_REQUIRE(ctx && ms_out, "null argument");
    SMC_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
    SMC_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    SMC_HIP_CHECK(hipEventElapsedTime(ms_out, ctx->ev0, ctx->ev1));
    return SMC_OK;
}

}  // extern "C"

int smc_scratch(smc_ctx* ctx, size_t bytes, void** out)
{
    if (bytes > ctx->scratch_bytes) {
        SMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) SMC_HIP_CHECK(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        size_t want = smc_align_up(bytes + bytes / 4, 1 << 16);
        hipError_t e = hipMalloc(&ctx->scratch, want);
        if (e != hipSuccess) {
            smc_set_error("scratch allocation of %zu bytes failed: %s", want,
                          hipGetErrorString(e));
            return SMC_ERR_NOMEM;
        }
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return SMC_OK;
}

#ifdef SMC_EMULATE
// test hooks (emulator build only): evaluate the lean math routines on the host
#include "smc_math.h"
extern "C" void smc_test_exp_nonpos(const double* x, int64_t n, double* out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = smc_exp_nonpos(x[i]);
}
extern "C" void smc_test_log_pos(const double* x, int64_t n, double* out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = smc_log_pos(x[i]);
}
extern "C" void smc_test_bm_pair(const uint64_t* a, const uint64_t* b, int64_t n, double* z0, double* z1)
{
    for (int64_t i = 0; i < n; ++i) smc_bm_pair(reinterpret_cast<const SmcD2*>(smc_ntab), a[i], b[i], z0[i], z1[i]);
}
extern "C" void smc_test_sincospi_02(const double* a, int64_t n, double* s, double* c)
{
    for (int64_t i = 0; i < n; ++i) smc_sincospi_02(a[i], &s[i], &c[i]);
}
#endif
