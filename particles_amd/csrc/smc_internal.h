// smc_internal.h -- host-side internals of libsmc_hip (context, errors, scratch)
#pragma once
#include "smc_platform.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/smc_hip.h"

struct smc_ctx {
    int device;
    u64 seed;
    hipStream_t stream;
    void* scratch;          // grow-only device scratch for the stand-alone ops
    size_t scratch_bytes;
    hipEvent_t ev0, ev1;
    int n_cu;
    // smc_malloc / smc_free recycle blocks by exact size: every user of a context is ordered
    // on its one stream, so a freed block can be handed out again without synchronising (the
    // temporaries of device-resident model code come and go at kernel-launch rate)
    std::unordered_map<size_t, std::vector<void*>> pool;
    std::unordered_map<void*, size_t> live;
    size_t pooled_bytes;
    // pinned (mapped) host staging blocks of destroyed filters, by size: hipHostMalloc / hipHostFree
    // cost about a millisecond each and a PMMH chain makes a filter per proposal
    std::unordered_map<size_t, std::vector<void*>> pinned;
};
#define SMC_POOL_MAX_BYTES ((size_t)8 << 30)

void smc_set_error(const char* fmt, ...);

// communicator of the sharded runs (smc_comm.hip): one RCCL communicator per process
struct smc_comm {
    smc_ctx* ctx;
    void* nccl;      // ncclComm_t; null: the emulator's one-rank stub
    int nranks, rank;
};

#define SMC_HIP_CHECK(expr)                                                        \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                    \
            smc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                          __FILE__, __LINE__);                                     \
            return SMC_ERR_HIP;                                                    \
        }                                                                          \
    } while (0)

#define SMC_REQUIRE(cond, msg)                                 \
    do {                                                       \
        if (!(cond)) {                                         \
            smc_set_error("%s: %s", __func__, msg);            \
            return SMC_ERR_INVALID;                            \
        }                                                      \
    } while (0)

#define SMC_LAUNCH_CHECK() SMC_HIP_CHECK(hipGetLastError())

// device scratch of at least `bytes` (256-byte aligned); contents undefined
int smc_scratch(smc_ctx* ctx, size_t bytes, void** out);

// radix sort of (64-bit key, 64-bit payload) pairs in a caller-owned workspace (smc_sort.hip); kind 0: the
// keys are fp64 bit patterns, 1: int64.  vals null: payload = index (argsort).  The results stay in `ws`.
size_t smc_rs_ws_bytes(long long N);
int smc_rs_sort_ws(smc_ctx* ctx, const void* keys, const void* vals, long long N, int kind, void* ws,
                   unsigned long long** sorted_keys, unsigned long long** sorted_vals, bool plan_is_zero = false);

// the scrambled Sobol' point set `counter` of the stream keyed by `seed` (smc_qmc.hip), (N, d) row-major
int smc_sobol_points(smc_ctx* ctx, unsigned long long seed, long long N, int d, unsigned long long counter, int sorted,
                     double* out);
// hilbert_sort (smc_sort.hip; the C-ABI entry, declared here for the fused SQMC step)
extern "C" int smc_hilbert_sort(smc_ctx* ctx, const double* x, int64_t N, int32_t d, int64_t* out, int64_t* keys_out);

static inline size_t smc_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
