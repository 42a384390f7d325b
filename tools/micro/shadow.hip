// Microbenchmark (perf experiments only): does VALU work issued BETWEEN the fp64 MFMAs of ONE wave run in the
// shadow of the matrix instruction on MI355X?  Each loop iteration is 8 x [v_mfma_f64_16x16x4_f64 ; k fillers] with
// the fillers independent of the MFMAs (inline asm: nothing is reordered).  Fillers: 0 = v_xor_b32 / v_add_u32
// (full rate), 1 = v_fma_f64, 2 = v_mad_u64_u32 (Philox's multiply), 3 = v_mul_lo_u32.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/shadow.hip -o /tmp/shadow && /tmp/shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

#define FILL_INT(n) asm volatile("v_xor_b32 %0, %0, %1\n v_add_u32 %1, %1, %0" : "+v"(u##n), "+v"(w##n));
#define FILL_FMA(n) asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %0, %0" : "+v"(c##n), "+v"(e##n));
#define FILL_MAD(n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_xor_b32 %1, %1, %2" : "+v"(m##n), "+v"(u##n) : "v"(w##n) : "vcc");
#define FILL_MUL(n) asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %0" : "+v"(u##n), "+v"(w##n));

template <int KIND>
__device__ __forceinline__ void fill2(unsigned& u0, unsigned& w0, double& c0, double& e0, unsigned long long& m0)
{
    if (KIND == 0) { FILL_INT(0) }
    if (KIND == 1) { FILL_FMA(0) }
    if (KIND == 2) { FILL_MAD(0) }
    if (KIND == 3) { FILL_MUL(0) }
}

// K2 = pairs of filler instructions behind every MFMA; NOMFMA: the fillers alone
template <int KIND, int K2, bool MFMA>
__global__ void __launch_bounds__(1024) k(double* out, int iters)
{
    double x = threadIdx.x * 1e-3, y = 1.0 + 1e-9 * threadIdx.x;
    v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    unsigned u[4] = {threadIdx.x, threadIdx.x * 3u, threadIdx.x * 5u, 7u}, w[4] = {1, 2, 3, 4};
    double c[4] = {x, x + 1, x + 2, x + 3}, e[4] = {y, y, y, y};
    unsigned long long m[4] = {1, 2, 3, 4};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (MFMA) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
#pragma unroll
            for (int q = 0; q < K2; ++q) fill2<KIND>(u[q & 3], w[q & 3], c[q & 3], e[q & 3], m[q & 3]);
            if (MFMA) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
#pragma unroll
            for (int q = 0; q < K2; ++q) fill2<KIND>(u[q & 3], w[q & 3], c[q & 3], e[q & 3], m[q & 3]);
            if (MFMA) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y));
#pragma unroll
            for (int q = 0; q < K2; ++q) fill2<KIND>(u[q & 3], w[q & 3], c[q & 3], e[q & 3], m[q & 3]);
            if (MFMA) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(a3) : "v"(x), "v"(y));
#pragma unroll
            for (int q = 0; q < K2; ++q) fill2<KIND>(u[q & 3], w[q & 3], c[q & 3], e[q & 3], m[q & 3]);
        }
    }
    double s = a0[0] + a1[1] + a2[2] + a3[3];
    for (int q = 0; q < 4; ++q) s += (double)(u[q] ^ w[q]) + c[q] + e[q] + (double)m[q];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int KIND, int K2, bool MFMA>
static double run(double* d, int iters, int waves_per_simd)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 256 * waves_per_simd;
    k<KIND, K2, MFMA><<<256, threads>>>(d, 100);
    hipEventRecord(e0);
    k<KIND, K2, MFMA><<<256, threads>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    // cycles (2.4 GHz nominal) per MFMA slot of ONE wave
    return ms * 1e-3 * 2.4e9 / (8.0 * iters);
}

template <int KIND>
static void table(double* d, const char* name)
{
    const int iters = 20000;
    printf("filler %-14s  cycles per [MFMA + 2k fillers] slot of one wave (2.4 GHz nominal); 'alone' = the fillers without the MFMA\n", name);
    printf("  waves/SIMD |   k=0 |  2k=4  alone |  2k=8  alone | 2k=16  alone | 2k=32  alone | 2k=64  alone\n");
    for (int w = 1; w <= 4; ++w) {
        printf("  %10d | %5.0f | %5.0f %6.0f | %5.0f %6.0f | %5.0f %6.0f | %5.0f %6.0f | %5.0f %6.0f\n", w,
               run<KIND, 0, true>(d, iters, w),
               run<KIND, 2, true>(d, iters, w), run<KIND, 2, false>(d, iters, w),
               run<KIND, 4, true>(d, iters, w), run<KIND, 4, false>(d, iters, w),
               run<KIND, 8, true>(d, iters, w), run<KIND, 8, false>(d, iters, w),
               run<KIND, 16, true>(d, iters, w), run<KIND, 16, false>(d, iters, w),
               run<KIND, 32, true>(d, iters, w), run<KIND, 32, false>(d, iters, w));
    }
}

int main()
{
    double* d; hipMalloc(&d, 256 * 1024 * 8);
    table<0>(d, "xor/add u32");
    table<1>(d, "v_fma_f64");
    table<2>(d, "mad_u64_u32+xor");
    table<3>(d, "mul_lo/hi_u32");
    return 0;
}
