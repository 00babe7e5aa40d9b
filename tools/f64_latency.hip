// Measurement: cycles per unfused f64 wave-instruction on gfx950 at 2 waves per SIMD as a function of the number of independent
// dependency chains per wave (distance between dependent instructions), i.e. how much instruction-level parallelism the frame
// loop's butterflies must expose.  hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/f64_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCH>
__global__ __launch_bounds__(512) void k(double *out, unsigned long long *clk, int iters)
{
    double a[NCH], b[NCH], d[NCH];
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        a[i] = (double)(threadIdx.x + i) * 1e-3 + 1.0;
        b[i] = 1.0 + (double)(threadIdx.x * 16 + i) * 1e-9;
        d[i] = (double)(threadIdx.x + 3 * i) * 1e-12;
    }
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 16 / NCH; rep++) {
#pragma unroll
            for (int i = 0; i < NCH; i++) a[i] = a[i] * b[i];
#pragma unroll
            for (int i = 0; i < NCH; i++) a[i] = a[i] + d[i];
        }
    }
    const unsigned long long c1 = clock64();
    double r = 0;
#pragma unroll
    for (int i = 0; i < NCH; i++) r += a[i] + b[i] + d[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}
template <int NCH>
void run(double *out, unsigned long long *clk)
{
    unsigned long long h[8];
    const int iters = 20000;
    for (int rep = 0; rep < 2; rep++) {
        k<NCH><<<8, 512>>>(out, clk, iters);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    double s = 0;
    for (int b = 0; b < 8; b++) s += h[b];
    printf("%2d independent chains: %.2f cycles per wave-instruction per SIMD (2 waves/SIMD)\n", NCH, s / 8 / (2.0 * iters * 32));
}
int main()
{
    double *out; unsigned long long *clk;
    (void)hipMalloc(&out, 8 * 8 * 512); (void)hipMalloc(&clk, 64);
    run<1>(out, clk); run<2>(out, clk); run<4>(out, clk); run<8>(out, clk); run<16>(out, clk);
    return 0;
}
