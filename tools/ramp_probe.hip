// Measurement: the shader clock over the first launches of a fresh process (why bench.py's `value` right after a short warm-up is lower
// than `value_steady`).  A ~65 us f64 kernel on all 256 CUs, launched back to back; every launch records, per workgroup, the shader cycles
// (clock64 = s_memtime) and the constant 100 MHz ticks (wall_clock64) it ran for: cycles / ticks x 100 = the clock in MHz during that launch;
// the host clock gives the time per launch.  Build and run on the GPU box:
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/ramp_probe.hip -o /tmp/ramp_probe && /tmp/ramp_probe [idle_ms_before_start]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ __launch_bounds__(512) void k(double *out, unsigned long long *clk, int iters)
{
    double a[16], b[16], d[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        a[i] = (double)(threadIdx.x + i) * 1e-3 + 1.0;
        b[i] = 1.0 + (double)(threadIdx.x * 16 + i) * 1e-9;
        d[i] = (double)(threadIdx.x + 3 * i) * 1e-12;
    }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = a[i] * b[i] + d[i];
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    double r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r += a[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
int main(int argc, char **argv)
{
    const int idle_ms = argc > 1 ? atoi(argv[1]) : 0;
    const int N = 6000, iters = 2200;   // ~65 us per launch at 2.4 GHz
    double *out;
    unsigned long long *clk;
    (void)hipMalloc(&out, 8 * 256 * 512);
    (void)hipMalloc(&clk, 16 * (size_t)N);
    (void)hipDeviceSynchronize();
    std::this_thread::sleep_for(std::chrono::milliseconds(idle_ms));
    std::vector<double> t_host(N);
    const auto t0 = std::chrono::steady_clock::now();
    // in batches of 20 launches so that the host clock can be read without stalling the queue for long
    for (int i = 0; i < N; i++) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, clk + 2 * i, iters);
        if (i % 20 == 19) {
            (void)hipDeviceSynchronize();
            const double t = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            for (int j = i - 19; j <= i; j++) t_host[j] = t;
        }
    }
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * (size_t)N);
    (void)hipMemcpy(h.data(), clk, 16 * (size_t)N, hipMemcpyDeviceToHost);
    printf("idle before the first launch: %d ms.  launch: shader MHz during it, its duration by the 100 MHz counter, host time since the first launch\n", idle_ms);
    for (int i : {0, 1, 2, 4, 9, 19, 39, 79, 159, 319, 639, 999, 1499, 1999, 2999, 3999, 5999})
        printf("launch %5d: %6.0f MHz  %7.2f us  (t = %8.2f ms)\n", i + 1, (double)h[2 * i] / (double)h[2 * i + 1] * 100.0, (double)h[2 * i + 1] / 100.0,
               t_host[(i / 20) * 20 + 19] / 1e3);   // (the host clock is read at the end of every batch of 20 launches)
    return 0;
}
