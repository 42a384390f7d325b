/*
 * hip_emu.h -- a tiny CPU emulator of the HIP constructs the smc kernels use.
 *
 * TEST INFRASTRUCTURE ONLY.  The build container has no GPU, so the kernel
 * sources under particles_amd/csrc are additionally compiled with g++
 * -DSMC_EMULATE against this header into tests/emu/_build/libsmc_emu.so, and
 * the CPU test-suite drives the very same kernel code through the same C ABI
 * to catch indexing / divergent-barrier / logic bugs before a GPU run.  It is
 * never a fallback for the product: particles_amd only loads libsmc_hip.so
 * unless a test explicitly points it at the emulator.
 *
 * Model: one workgroup at a time; every thread of the workgroup is a ucontext
 * fiber; __syncthreads() and wave shuffles are counting barriers at which the
 * fibers are switched round-robin.  A barrier that not all live threads of
 * the workgroup (or wave) reach is reported as an error -- on hardware it
 * would be a hang.  Wavefront = 64 lanes.
 */
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace hipemu {
constexpr int WAVE = 64;
constexpr int MAXT = 1024;
constexpr size_t STACK = 256 * 1024;

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = true;
};

struct State {
    dim3 grid, block, bidx;
    int nthreads = 0, cur = 0, live = 0;
    Fiber fib[MAXT];
    ucontext_t sched;
    // block barrier
    long progress = 0;
    int bar_count = 0;
    long bar_gen = 0;
    // per-wave barrier + exchange
    int wbar_count[MAXT / WAVE] = {0};
    long wbar_gen[MAXT / WAVE] = {0};
    int wave_live[MAXT / WAVE] = {0};
    uint64_t xch[MAXT];
    double xa[MAXT], xb[MAXT];      // MFMA operands
    std::function<void()> body;
    bool error = false;
    alignas(16) char dyn_smem[160 * 1024];
};
inline State& S() { static State s; return s; }

inline void yield_() {
    State& s = S();
    swapcontext(&s.fib[s.cur].ctx, &s.sched);
}

inline void fiber_main() {
    State& s = S();
    s.body();
    s.fib[s.cur].done = true;
    s.progress++;
    s.live--;
    s.wave_live[s.cur / WAVE]--;
    swapcontext(&s.fib[s.cur].ctx, &s.sched);
}

inline void run_block() {
    State& s = S();
    s.live = s.nthreads;
    s.bar_count = 0;
    int nw = (s.nthreads + WAVE - 1) / WAVE;
    for (int w = 0; w < nw; ++w) {
        s.wbar_count[w] = 0;
        int hi = (w + 1) * WAVE < s.nthreads ? (w + 1) * WAVE : s.nthreads;
        s.wave_live[w] = hi - w * WAVE;
    }
    for (int t = 0; t < s.nthreads; ++t) {
        Fiber& f = s.fib[t];
        if (!f.stack) f.stack = (char*)malloc(STACK);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link = nullptr;
        f.done = false;
        makecontext(&f.ctx, (void (*)())fiber_main, 0);
    }
    long spins = 0;
    while (s.live > 0) {
        bool progressed = false;
        for (int t = 0; t < s.nthreads; ++t) {
            if (s.fib[t].done) continue;
            s.cur = t;
            long p0 = s.progress;
            swapcontext(&s.sched, &s.fib[t].ctx);
            if (s.progress != p0) progressed = true;
        }
        // a full round with nobody finishing and no barrier released repeatedly
        // means some threads wait at a barrier others never reach
        spins = progressed ? 0 : spins + 1;
        if (spins > 100000) {
            fprintf(stderr, "hipemu: DEADLOCK (divergent barrier) in block (%u,%u)\n",
                    s.bidx.x, s.bidx.y);
            s.error = true;
            abort();
        }
    }
}

inline void block_barrier() {
    State& s = S();
    long g = s.bar_gen;
    if (++s.bar_count >= s.live) {
        s.bar_count = 0;
        s.bar_gen++;
        s.progress++;
        return;
    }
    while (s.bar_gen == g) {
        // threads that exited no longer count (as on hardware)
        if (s.bar_count >= s.live) { s.bar_count = 0; s.bar_gen++; s.progress++; break; }
        yield_();
    }
}

inline void wave_barrier() {
    State& s = S();
    int w = s.cur / WAVE;
    long g = s.wbar_gen[w];
    if (++s.wbar_count[w] >= s.wave_live[w]) {
        s.wbar_count[w] = 0;
        s.wbar_gen[w]++;
        s.progress++;
        return;
    }
    long spins = 0;
    while (s.wbar_gen[w] == g) {
        if (s.wbar_count[w] >= s.wave_live[w]) { s.wbar_count[w] = 0; s.wbar_gen[w]++; s.progress++; break; }
        yield_();
        if (++spins > 10000000) {
            fprintf(stderr, "hipemu: wave shuffle not reached by all lanes\n");
            abort();
        }
    }
}

template <typename T>
inline T exchange(T v, int src_lane_abs) {
    static_assert(sizeof(T) <= 8, "shuffle of >8 bytes");
    State& s = S();
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    s.xch[s.cur] = raw;
    wave_barrier();
    uint64_t r = s.xch[src_lane_abs];
    wave_barrier();
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}

// v_mfma_f64_16x16x4_f64: D(16x16) = C + A(16x4) B(4x16) over the wavefront.
// Lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; it owns
// D[(l >> 4) + 4 r][l & 15], r = 0..3 (cdna_hip_programming.md, f64 MFMA layout).
template <typename V4>
inline V4 mfma_f64_16x16x4(double a, double b, V4 c) {
    State& s = S();
    const int base = s.cur - s.cur % WAVE, l = s.cur % WAVE;
    s.xa[s.cur] = a;
    s.xb[s.cur] = b;
    wave_barrier();
    V4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) + 4 * r, col = l & 15;
        double acc = c[r];
        for (int k = 0; k < 4; ++k) acc = std::fma(s.xa[base + row + 16 * k], s.xb[base + col + 16 * k], acc);
        d[r] = acc;
    }
    wave_barrier();
    return d;
}

template <typename F>
inline void launch(dim3 grid, dim3 block, F&& f) {
    State& s = S();
    s.grid = grid;
    s.block = block;
    s.nthreads = block.x * block.y * block.z;
    if (s.nthreads > MAXT) { fprintf(stderr, "hipemu: block too large\n"); abort(); }
    s.body = f;
    for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) {
            s.bidx = dim3(bx, by, 0);
            run_block();
        }
}
}  // namespace hipemu

// ---- built-in variables --------------------------------------------------
struct EmuThreadIdx {
    struct X { operator unsigned() const { return hipemu::S().cur % hipemu::S().block.x; } } x;
    struct Y { operator unsigned() const { return hipemu::S().cur / hipemu::S().block.x; } } y;
};
struct EmuBlockIdx {
    struct X { operator unsigned() const { return hipemu::S().bidx.x; } } x;
    struct Y { operator unsigned() const { return hipemu::S().bidx.y; } } y;
};
struct EmuBlockDim {
    struct X { operator unsigned() const { return hipemu::S().block.x; } } x;
};
struct EmuGridDim {
    struct X { operator unsigned() const { return hipemu::S().grid.x; } } x;
    struct Y { operator unsigned() const { return hipemu::S().grid.y; } } y;
};
static EmuThreadIdx threadIdx;
static EmuBlockIdx blockIdx;
static EmuBlockDim blockDim;
static EmuGridDim gridDim;

#define hipemu_mfma_f64_16x16x4(a, b, c) hipemu::mfma_f64_16x16x4((a), (b), (c))
inline void __syncthreads() { hipemu::block_barrier(); }
inline void __threadfence() {}

inline int emu_lane() { return hipemu::S().cur % hipemu::WAVE; }
inline int emu_wbase() { return hipemu::S().cur - emu_lane(); }

template <typename T>
inline T __shfl(T v, int lane, int width = 64) {
    (void)width;
    return hipemu::exchange(v, emu_wbase() + (lane & 63));
}
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
    (void)width;
    return hipemu::exchange(v, emu_wbase() + ((emu_lane() ^ mask) & 63));
}
template <typename T>
inline T __shfl_up(T v, unsigned d, int width = 64) {
    (void)width;
    int l = emu_lane();
    int src = l >= (int)d ? l - (int)d : l;
    return hipemu::exchange(v, emu_wbase() + src);
}
template <typename T>
inline T __shfl_down(T v, unsigned d, int width = 64) {
    (void)width;
    int l = emu_lane();
    int src = l + (int)d < 64 ? l + (int)d : l;
    // lanes beyond the live part of a partial wave do not exist: clamp
    if (emu_wbase() + src >= hipemu::S().nthreads) src = l;
    return hipemu::exchange(v, emu_wbase() + src);
}

// wave-wide ballot: bit l = predicate of lane l (lanes that do not exist contribute 0)
inline unsigned long long __ballot(int pred) {
    hipemu::State& s = hipemu::S();
    s.xch[s.cur] = pred ? 1ull : 0ull;
    hipemu::wave_barrier();
    unsigned long long m = 0;
    const int base = emu_wbase();
    for (int l = 0; l < 64 && base + l < s.nthreads; ++l)
        if (s.xch[base + l]) m |= 1ull << l;
    hipemu::wave_barrier();
    return m;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline unsigned int __brev(unsigned int v)
{
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(v);
}
inline void emu_wave_sync() { hipemu::wave_barrier(); }

struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(16) double2 { double x, y; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

inline unsigned __umulhi(unsigned a, unsigned b) {
    return (unsigned)(((unsigned long long)a * b) >> 32);
}
inline void sincospi(double x, double* s, double* c) {
    *s = sin(M_PI * x);
    *c = cos(M_PI * x);
}
inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) {
    return (unsigned long long)(((unsigned __int128)a * b) >> 64);
}
inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p += v; return o; }
inline int atomicMin(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }
inline unsigned atomicMin(unsigned* p, unsigned v) { unsigned o = *p; if (v < o) *p = v; return o; }
inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }
inline unsigned atomicOr(unsigned* p, unsigned v) { unsigned o = *p; *p |= v; return o; }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
    unsigned long long o = *p; *p += v; return o;
}

// ---- runtime API subset ----------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
#define hipSuccess 0
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
#define hipStreamNonBlocking 0
#define hipStreamCaptureModeThreadLocal 0
struct hipDeviceProp_t {
    char name[256];
    int multiProcessorCount;
    size_t totalGlobalMem;
};
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
// (SMC_EMU_NDEV: how many devices the emulator pretends to see -- the 8-rank launch line of bench.py gives every rank
//  its own, tests/test_distributed_cpu.py)
inline hipError_t hipGetDeviceCount(int* n) { const char* e = getenv("SMC_EMU_NDEV"); *n = e && atoi(e) > 0 ? atoi(e) : 1; return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    snprintf(p->name, sizeof p->name, "hipemu (CPU fibers)");
    p->multiProcessorCount = 1;
    p->totalGlobalMem = 0;
    return 0;
}
inline hipError_t hipDeviceGetPCIBusId(char* out, int len, int dev) {
    snprintf(out, (size_t)len, "emu0:%02x:00.0", dev); return 0;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, int) { *s = nullptr; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipFree(void* p) { free(p); return 0; }
#define hipHostMallocMapped 0x2
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipHostFree(void* p) { free(p); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) {
    memcpy(d, s, n); return 0;
}
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return 1; }  // graphs: unsupported
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*) { return 1; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return 1; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return 1; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return 0; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return 0; }

#define SMC_LAUNCH(kernel, grid, block, stream, ...) \
    hipemu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
