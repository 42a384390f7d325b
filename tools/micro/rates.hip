// Microbenchmark (perf experiments only): issue cost, in shader cycles per wavefront instruction, of the
// instructions the normal generator is made of on MI355X -- fp64 fma / mul / add, the 32-bit integer
// multiplies of Philox (v_mad_u64_u32, v_mul_lo_u32, v_mul_hi_u32), 32-bit logic, the fp64
// transcendentals (v_rcp_f64, v_rsq_f64), conversions, v_ldexp_f64 and a random 16-byte LDS read.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/rates.hip -o /tmp/rates && /tmp/rates
// Each kernel runs NI independent dependency chains of one instruction, unrolled, for `iters`
// iterations; cycles = s_memtime difference of wave 0 / instructions issued by that wave.  Reported
// for 1, 2 and 4 waves per SIMD (blocks of 256, 512, 1024 threads; one block per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define NI 8
#define UNROLL 8

enum Op { FMA64, MUL64, ADD64, MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, XOR32, ADD32, RCP64, RSQ64, LDEXP64, CVT_F64_U32,
          FMA32, MUL_U24, LDS_B128, RNDNE64, CNDMASK, NOPS };
static const char* NAMES[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32",
                              "v_xor_b32", "v_add_u32", "v_rcp_f64", "v_rsq_f64", "v_ldexp_f64", "v_cvt_f64_u32",
                              "v_fma_f32", "v_mul_u32_u24", "ds_read_b128 (random)", "v_rndne_f64", "v_cndmask_b32"};

template <int OP>
__global__ void __launch_bounds__(1024) k(double* out, long long* cyc, int iters)
{
    __shared__ double tab[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) tab[i] = 1.0 + i * 1e-6;
    __syncthreads();
    double d[NI];
    unsigned u[NI];
    unsigned long long w[NI];
    float f[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        d[i] = 1.0 + 1e-3 * (threadIdx.x + i);
        u[i] = threadIdx.x * 2654435761u + i;
        w[i] = u[i];
        f[i] = 1.0f + 1e-3f * i;
    }
    const double y = 1.0 + 1e-12 * threadIdx.x, x = 1e-9;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < UNROLL; ++r) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(y), "v"(x));
                if (OP == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(y));
                if (OP == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(x));
                if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(w[i]) : "v"(u[i]), "v"(0xD2511F53u) : "s10", "s11");
                if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(0xD2511F53u));
                if (OP == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(0xD2511F53u));
                if (OP == XOR32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(0x9E3779B9u));
                if (OP == ADD32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(0x9E3779B9u));
                if (OP == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
                if (OP == RSQ64) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));
                if (OP == LDEXP64) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(0));
                if (OP == CVT_F64_U32) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(d[i]) : "v"(u[i]));
                if (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(1.0001f), "v"(1e-9f));
                if (OP == MUL_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(0x511F53u));
                if (OP == RNDNE64) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i]));
                if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(0x9E3779B9u) : "vcc");
                if (OP == LDS_B128) {
                    // address from the previous value: a dependent, lane-random 16-byte read
                    const unsigned a = ((u[i] >> 7) & 1023u) * 16u;
                    double2 v = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(tab) + a);
                    u[i] = u[i] * 1664525u + (unsigned)__double_as_longlong(v.x) + 1013904223u;
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NI; ++i) s += d[i] + (double)u[i] + (double)w[i] + (double)f[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(double* d, long long* c, int ncu)
{
    const int iters = 2000;
    printf("%-24s", NAMES[OP]);
    for (int waves = 1; waves <= 4; waves *= 2) {
        const int threads = 256 * waves;
        k<OP><<<ncu, threads>>>(d, c, 10);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<OP><<<ncu, threads>>>(d, c, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(ncu);
        hipMemcpy(h.data(), c, ncu * 8, hipMemcpyDeviceToHost);
        double cy = 0; for (auto v : h) cy += (double)v; cy /= ncu;
        const double ninst = (double)iters * UNROLL * NI;
        // s_memtime ticks at 100 MHz on this part?  report both: ticks per instruction and wall ns per
        // instruction per SIMD (wall / (instructions per wave * waves per SIMD))
        printf("  %dw/SIMD: %7.2f ticks/inst %7.3f ns/inst/SIMD", waves, cy / ninst, ms * 1e6 / (ninst * waves));
    }
    printf("\n");
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.name, ncu, p.clockRate);
    double* d; hipMalloc(&d, (size_t)ncu * 1024 * 8);
    long long* c; hipMalloc(&c, ncu * 8);
    run<FMA64>(d, c, ncu); run<MUL64>(d, c, ncu); run<ADD64>(d, c, ncu); run<MAD_U64_U32>(d, c, ncu);
    run<MUL_LO_U32>(d, c, ncu); run<MUL_HI_U32>(d, c, ncu); run<XOR32>(d, c, ncu); run<ADD32>(d, c, ncu);
    run<RCP64>(d, c, ncu); run<RSQ64>(d, c, ncu); run<LDEXP64>(d, c, ncu); run<CVT_F64_U32>(d, c, ncu);
    run<FMA32>(d, c, ncu); run<MUL_U24>(d, c, ncu); run<LDS_B128>(d, c, ncu); run<RNDNE64>(d, c, ncu);
    run<CNDMASK>(d, c, ncu);
    return 0;
}
