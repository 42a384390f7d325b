// Microbenchmark (perf experiments only): wavefronts that ALTERNATE between a
// block of dependent fp64 MFMAs and a block of VALU work, as k_propagate_mv does.
// Does the SIMD overlap the MFMA phase of one wave with the VALU phase of another?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NM, int NV, int NACC, int DEP = 1>   // NM MFMAs then NV v_fma_f64 per phase pair
__global__ void __launch_bounds__(256) k(double* out, int iters)
{
    double x = threadIdx.x * 1e-3, y = 1.0 + 1e-9 * threadIdx.x;
    v4d a[NACC];
    for (int i = 0; i < NACC; ++i) a[i] = (v4d){0, 0, 0, 0};
    double c0 = x, c1 = x + 1, c2 = x + 2, c3 = x + 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m)
            a[m % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(DEP == 1 ? c0 : x, y, a[m % NACC], 0, 0, 0);
        if (NM && DEP == 1) { c0 += a[0][0]; c1 += a[NACC - 1][1]; }          // VALU phase depends on the products
#pragma unroll
        for (int r = 0; r < NV / 4; ++r) {
            c0 = __builtin_fma(c0, y, x); c1 = __builtin_fma(c1, y, x);
            c2 = __builtin_fma(c2, y, x); c3 = __builtin_fma(c3, y, x);
        }
        c0 = c0 * 1e-30 + c2 * 1e-30 + c3 * 1e-30 + c1 * 1e-30;   // and the next products on it
        if constexpr (DEP == 2 && NM > 0) {          // independent streams + an explicit 1 MFMA : NV/NM VALU interleave
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, NM ? NV / NM : 1, 0);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3 + a[0][0];
}

template <int NM, int NV, int NACC, int DEP = 1>
static float run(double* d, int iters, int wg_per_cu)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NM, NV, NACC, DEP><<<256 * wg_per_cu, 256>>>(d, 10);
    (void)hipEventRecord(e0);
    k<NM, NV, NACC, DEP><<<256 * wg_per_cu, 256>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    double* d; (void)hipMalloc(&d, 256 * 8 * 256 * 8);
    const int iters = 20000;
    for (int w = 1; w <= 4; ++w) {     // w workgroups of 4 waves per CU = w waves per SIMD
        const float tm = run<16, 0, 2>(d, iters, w), tv = run<0, 256, 2>(d, iters, w), tb = run<16, 256, 2>(d, iters, w);
        const float tm4 = run<16, 0, 4>(d, iters, w), tb4 = run<16, 256, 4>(d, iters, w);
        const float ti = run<16, 256, 2, 0>(d, iters, w), tg = run<16, 256, 2, 2>(d, iters, w), tg1 = run<16, 128, 2, 2>(d, iters, w), tv1 = run<0, 128, 2>(d, iters, w);
        printf("   independent streams: compiler order %7.3f  sched_group 1:16 %7.3f ;  NV=128: VALU-only %7.3f sched_group 1:8 %7.3f\n", ti, tg, tv1, tg1);
        printf("%d waves/SIMD: MFMA-only %7.3f (4 acc %7.3f)  VALU-only %7.3f  both %7.3f (4 acc %7.3f)   sum %7.3f  max %7.3f\n",
               w, tm, tm4, tv, tb, tb4, tm + tv, tm > tv ? tm : tv);
    }
    return 0;
}
