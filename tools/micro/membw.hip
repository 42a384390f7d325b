// What do the memory access patterns of the headline step cost on MI355X when nothing is computed?
// One launch = 1024 workgroups x 256 threads over N = 2^20 elements, 4 elements per thread as two
// 16-byte accesses per array (k_propagate's shape at C2); K launches back to back on one stream
// between two events, the working set (8 .. 56 MB) resident in the Infinity Cache like the filter's.
//
//     hipcc --offload-arch=gfx950 -O3 tools/micro/membw.hip -o tools/micro/_build/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ void st2(double* p, double a, double b)
{
    d2 v = {a, b};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<d2*>(p));
    else *reinterpret_cast<d2*>(p) = v;
}
__device__ __forceinline__ void ld2(const double* p, double& a, double& b)
{
    const d2 v = *reinterpret_cast<const d2*>(p);
    a = v.x; b = v.y;
}
__device__ __forceinline__ void own(long long& na, long long& nb)
{
    const long long wb = ((long long)blockIdx.x * 256 + (threadIdx.x & ~63)) * 4;
    na = wb + 2 * (threadIdx.x & 63);
    nb = na + 128;
}

// NR arrays read (sequential), NW arrays written; GATHER: one more array read through 32-bit indices
template <int NR, int NW, bool GATHER, bool NT>
__global__ void __launch_bounds__(256)
k_mem(const double* const* rd, double* const* wr, const unsigned* A, const double* X, double* sink)
{
    long long na, nb;
    own(na, nb);
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        double a, b, c, d;
        ld2(rd[r] + na, a, b);
        ld2(rd[r] + nb, c, d);
        acc += a + b + c + d;
    }
    if (GATHER) {
        const u2 i0 = *reinterpret_cast<const u2*>(A + na), i1 = *reinterpret_cast<const u2*>(A + nb);
        acc += X[i0.x] + X[i0.y] + X[i1.x] + X[i1.y];
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        st2<NT>(wr[w] + na, acc + w, acc);
        st2<NT>(wr[w] + nb, acc, acc - w);
    }
    if (NW == 0 && acc == 1.2345e301) sink[0] = acc;       // (keeps the loads)
}

template <int NR, int NW, bool GATHER, bool NT>
static void run(const char* name, hipStream_t st, const double* const* rd, double* const* wr, const unsigned* A,
                const double* X, double* sink, double mbytes)
{
    const int K = 400;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < K; ++i) k_mem<NR, NW, GATHER, NT><<<1024, 256, 0, st>>>(rd, wr, A, X, sink);
        (void)hipEventRecord(e1, st);
        (void)hipStreamSynchronize(st);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double us = 1e3 * best / K;
    printf("%-58s %7.2f us/launch  %6.1f MB  %6.2f TB/s\n", name, us, mbytes, mbytes / us);
}

int main()
{
    const long long N = 1 << 20;
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    std::vector<double*> bufs(8);
    for (auto& b : bufs) { (void)hipMalloc(&b, N * 8); (void)hipMemset(b, 0, N * 8); }
    unsigned* A;
    (void)hipMalloc(&A, N * 4);
    {   // monotone ancestors with runs, like a systematic draw: A_n = n + a slow drift, clamped
        std::vector<unsigned> h(N);
        for (long long n = 0; n < N; ++n) {
            long long v = n + (long long)(700.0 * __builtin_sin(6.283 * n / (double)N)) - (n % 3 == 0);
            h[n] = (unsigned)(v < 0 ? 0 : (v >= N ? N - 1 : v));
        }
        (void)hipMemcpy(A, h.data(), N * 4, hipMemcpyHostToDevice);
    }
    const double** rd; double** wr; double* sink;
    (void)hipMalloc(&rd, 8 * sizeof(double*));
    (void)hipMalloc(&wr, 8 * sizeof(double*));
    (void)hipMalloc(&sink, 8);
    const double* hr[4] = {bufs[0], bufs[1], bufs[2], bufs[3]};
    double* hw[4] = {bufs[4], bufs[5], bufs[6], bufs[7]};
    (void)hipMemcpy(rd, hr, sizeof hr, hipMemcpyHostToDevice);
    (void)hipMemcpy(wr, hw, sizeof hw, hipMemcpyHostToDevice);
    hipDeviceProp_t pr;
    (void)hipGetDeviceProperties(&pr, 0);
    printf("%s, %d CUs; N = 2^20 elements per array, 1024 workgroups x 256 threads, K = 400 launches back to back\n",
           pr.gcnArchName, pr.multiProcessorCount);
    run<0, 0, false, false>("empty kernel", st, rd, wr, A, bufs[0], sink, 0.0);
    run<1, 0, false, false>("read 1 array (8 B/elem)", st, rd, wr, A, bufs[0], sink, 8.39);
    run<3, 0, false, false>("read 3 arrays", st, rd, wr, A, bufs[0], sink, 25.17);
    run<0, 1, false, false>("write 1 array", st, rd, wr, A, bufs[0], sink, 8.39);
    run<0, 1, false, true>("write 1 array, nt", st, rd, wr, A, bufs[0], sink, 8.39);
    run<0, 3, false, false>("write 3 arrays", st, rd, wr, A, bufs[0], sink, 25.17);
    run<0, 3, false, true>("write 3 arrays, nt", st, rd, wr, A, bufs[0], sink, 25.17);
    run<1, 1, false, true>("copy 1 -> 1, nt", st, rd, wr, A, bufs[0], sink, 16.78);
    run<0, 0, true, false>("A (4 B) + gather X[A]", st, rd, wr, A, bufs[0], sink, 12.58);
    run<0, 3, true, true>("k_propagate's pattern: A + gather, write 3 arrays nt", st, rd, wr, A, bufs[0], sink, 37.75);
    run<0, 3, true, false>("k_propagate's pattern, plain stores", st, rd, wr, A, bufs[0], sink, 37.75);
    run<1, 1, false, true>("k_ancestors2's pattern: read 8 B, write 8 B (A is 4)", st, rd, wr, A, bufs[0], sink, 16.78);
    return 0;
}
