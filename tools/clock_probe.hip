// Measurement: effective shader clock under an f64 VALU load on 8 vs 256 CUs (clock64 = s_memtime shader cycles,
// wall_clock64 = constant 100 MHz counter).  Build on the GPU box: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512) void k(double *out, unsigned long long *clk, int iters, double c)
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
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) a[i] = a[i] * c + 1e-9;            // one VGPR pair + scalar / literal operand per instruction
            else a[i] = a[i] * b[i] + d[i];                   // two VGPR pairs per instruction, as in a butterfly
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    double r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r += a[i] + b[i] + d[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
int main()
{
    double *out; unsigned long long *clk, h[512];
    (void)hipMalloc(&out, 8 * 256 * 512); (void)hipMalloc(&clk, sizeof h);
    for (int mode = 0; mode < 2; mode++)
    for (int blocks : {8, 256}) {
        for (int rep = 0; rep < 2; rep++) {
            if (mode == 0) k<0><<<blocks, 512>>>(out, clk, 50000, 1.0000001);
            else k<1><<<blocks, 512>>>(out, clk, 50000, 1.0000001);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
        double sc = 0, sw = 0;
        for (int b = 0; b < blocks; b++) { sc += h[2 * b]; sw += h[2 * b + 1]; }
        printf("mode %d blocks=%3d: %.0f MHz, %.2f ms, %.2f cycles per wave-instruction per SIMD (2 waves/SIMD, 32 f64 instr per iteration)\n", mode, blocks,
               (sc / blocks) / (sw / blocks) * 100.0, sw / blocks / 1e5, (sc / blocks) / (2.0 * 50000 * 32));
    }
    return 0;
}
