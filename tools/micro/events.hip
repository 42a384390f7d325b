// What does a HIP-event interval around one kernel launch include on MI355X?
// (perf diagnostics for bench.py's kernel timing)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty() {}
__global__ void spin(long long cycles, int* out)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}
int main()
{
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const int R = 200;
    hipEvent_t ev[3 * R];
    for (auto& e : ev) (void)hipEventCreate(&e);
    int* d; (void)hipMalloc(&d, 4);
    for (int mode = 0; mode < 4; ++mode) {
        for (int w = 0; w < 2; ++w) {        // warm-up + measured
            for (int i = 0; i < R; ++i) {
                (void)hipEventRecord(ev[3 * i], st);
                if (mode == 1) empty<<<1024, 256, 0, st>>>();
                if (mode >= 2) spin<<<1024, 256, 0, st>>>(mode == 2 ? 1000 : 2000, d);   // 100 MHz clock: 10 / 20 us
                (void)hipEventRecord(ev[3 * i + 1], st);
                spin<<<1024, 256, 0, st>>>(500, d);
                (void)hipEventRecord(ev[3 * i + 2], st);
            }
            (void)hipStreamSynchronize(st);
        }
        double a = 0, b = 0;
        for (int i = 0; i < R; ++i) {
            float x, y;
            (void)hipEventElapsedTime(&x, ev[3 * i], ev[3 * i + 1]);
            (void)hipEventElapsedTime(&y, ev[3 * i + 1], ev[3 * i + 2]);
            a += x; b += y;
        }
        const char* names[] = {"nothing", "empty kernel (1024 WG)", "10 us spin kernel", "20 us spin kernel"};
        printf("%-26s between events: %7.2f us   (following 5 us spin kernel: %7.2f us)\n", names[mode], 1e3 * a / R, 1e3 * b / R);
    }
    return 0;
}
