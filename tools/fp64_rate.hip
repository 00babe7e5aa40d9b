// Micro-benchmark: sustained f64 VALU rate on gfx950 for fused vs. unfused multiply/add streams (hipcc -O3 -ffp-contract=off).
// Build on the GPU box: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/fp64_rate.hip -o /tmp/fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters, double c, double s)
{
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = (double)(threadIdx.x + i) * 1e-3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) a[i] = __builtin_fma(a[i], c, s);          // 1 fused op
            else if (MODE == 1) a[i] = a[i] * c + s;                   // v_mul_f64 + v_add_f64
            else if (MODE == 2) a[i] = a[i] + s;                       // v_add_f64
            else a[i] = a[i] * c;                                      // v_mul_f64
        }
    }
    double r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
void run(const char *name, int blocks, double ops_per_iter)
{
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 100, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 /*waves*/ * iters * 16 * ops_per_iter;   // wave-instructions
    const double per_simd_cycle = instr / (1024.0 * 2.4e9 * ms * 1e-3);
    printf("%-22s blocks=%5d  %.3f ms  %.2f Tinstr-lanes/s  %.3f wave-instr per SIMD-cycle@2.4GHz (=> %.1f cycles per wave-instr)\n", name, blocks, ms,
           instr * 64 / (ms * 1e-3) / 1e12, per_simd_cycle, 1.0 / per_simd_cycle);
    hipFree(out);
}

int main()
{
    for (int blocks : {256, 512, 1024, 2048}) {
        run<0>("fma_f64", blocks, 1);
        run<1>("mul_f64+add_f64", blocks, 2);
        run<2>("add_f64", blocks, 1);
        run<3>("mul_f64", blocks, 1);
    }
    return 0;
}
