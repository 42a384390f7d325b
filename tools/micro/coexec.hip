// Microbenchmark (perf experiments only): do fp64 MFMA and fp64 VALU instructions
// from different wavefronts of one SIMD execute concurrently on MI355X?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/coexec.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: all waves MFMA, 1: all waves VALU fma, 2: even waves MFMA / odd waves VALU,
                      // 3: all waves VALU int (v_xor/v_add), 4: even MFMA / odd int
__global__ void __launch_bounds__(512) k(double* out, int iters)
{
    const int wave = threadIdx.x >> 6;
    const bool mf = MODE == 0 || ((MODE == 2 || MODE == 4 || MODE == 5) && (wave & 1) == 0);
    if ((MODE == 5 && (wave & 1)) || (MODE == 6 && !(wave & 1))) return;
    const bool iv = MODE == 3 || (MODE == 4 && (wave & 1));
    double x = threadIdx.x * 1e-3, y = 1.0 + 1e-9 * threadIdx.x;
    if (mf) {
        v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    } else if (iv) {
        unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7, u4 = 1, u5 = 2, u6 = 3, u7 = 4;
        for (int i = 0; i < iters * 2; ++i) {    // 64 int ops per iteration
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                u0 = (u0 ^ u4) + 0x9E3779B9u; u1 = (u1 ^ u5) + 0x7F4A7C15u; u2 = (u2 ^ u6) + u0; u3 = (u3 ^ u7) + u1;
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = (double)(u0 ^ u1 ^ u2 ^ u3);
    } else {
        double c0 = x, c1 = x + 1, c2 = x + 2, c3 = x + 3, c4 = x + 4, c5 = x + 5, c6 = x + 6, c7 = x + 7;
        for (int i = 0; i < iters; ++i) {    // 64 v_fma_f64 per iteration = the FMAs of 4 MFMAs
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                c0 = __builtin_fma(c0, y, x); c1 = __builtin_fma(c1, y, x); c2 = __builtin_fma(c2, y, x);
                c3 = __builtin_fma(c3, y, x); c4 = __builtin_fma(c4, y, x); c5 = __builtin_fma(c5, y, x);
                c6 = __builtin_fma(c6, y, x); c7 = __builtin_fma(c7, y, x);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    }
}

template <int MODE>
static float run(double* d, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 512>>>(d, 10);
    hipEventRecord(e0);
    k<MODE><<<256, 512>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    double* d; hipMalloc(&d, 256 * 512 * 8);
    const int iters = 100000;
    // one workgroup of 8 waves per CU: 2 waves per SIMD
    const float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters), t4 = run<4>(d, iters);
    const double mf = 2.0 * iters * 4 * 64.0;   // cycles if one 16x16x4 f64 MFMA holds the pipe 64 cycles
    const float t5 = run<5>(d, iters), t6 = run<6>(d, iters);
    printf("1 wave/SIMD MFMA %8.3f ms ; 1 wave/SIMD FMA64 %8.3f ms\n", t5, t6);
    printf("all MFMA   %8.3f ms  (%.0f cycles/MFMA at 2.4 GHz)\n", t0, t0 * 1e-3 * 2.4e9 / (2.0 * iters * 4));
    printf("all FMA64  %8.3f ms  (%.2f cycles/v_fma_f64)\n", t1, t1 * 1e-3 * 2.4e9 / (2.0 * iters * 64));
    printf("MFMA+FMA64 %8.3f ms  (no overlap would be %.3f, full overlap %.3f)\n", t2, t5 + t6, (t5 > t6 ? t5 : t6));
    printf("all INT    %8.3f ms\n", t3);
    printf("MFMA+INT   %8.3f ms  (no overlap would be %.3f, full overlap %.3f)\n", t4, t5 + 0.5 * t3, (t5 > 0.5 * t3 ? t5 : 0.5 * t3));
    (void)mf;
    return 0;
}
