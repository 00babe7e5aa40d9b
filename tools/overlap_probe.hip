// Measurement: do f64 VALU work and LDS exchange traffic of different waves of one CU overlap on gfx950?
// One 512-thread workgroup per CU (2 waves per SIMD) or 768 (3 per SIMD).  Per "frame" a wave runs PASSES passes of
// OPS independent-chain f64 instructions and, between passes, an exchange of 16 ds_write_b64 + 16 ds_read_b64 per component
// (two components) through a wave-private padded buffer, as the frame loop does.
//   hipcc -O3 --offload-arch=gfx950 tools/overlap_probe.hip -o /tmp/overlap_probe && /tmp/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>   // 0: VALU + LDS (the frame loop's shape), 1: VALU only, 2: LDS only
__global__ void k(double *out, unsigned long long *clk, int frames, int ops_per_pass)
{
    extern __shared__ double lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *buf = lds + wave * (1024 + 64);
    double v[16], w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        v[i] = 1.0 + (double)(threadIdx.x + i) * 1e-9;
        w[i] = 1.0 + (double)(lane * 16 + i) * 1e-12;
    }
    __syncthreads();
    const unsigned long long c0 = clock64();
    for (int f = 0; f < frames; f++) {
#pragma unroll 1
        for (int p = 0; p < 3; p++) {
            if (MODE != 2) {
                for (int it = 0; it < ops_per_pass / 32; it++) {
#pragma unroll
                    for (int i = 0; i < 16; i++) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
#pragma unroll
                    for (int i = 0; i < 16; i++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
                }
            }
            if (MODE != 1 && p < 2) {
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    double *wr = buf + lane * 17, *rd = buf + lane + (lane >> 4);
#pragma unroll
                    for (int i = 0; i < 16; i++) wr[i] = v[i];
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int i = 0; i < 16; i++) v[i] = rd[i * 68];
                    asm volatile("" ::: "memory");
                }
            }
        }
    }
    __syncthreads();
    const unsigned long long c1 = clock64();
    double r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

template <int MODE>
void run(const char *name, double *out, unsigned long long *clk, int threads)
{
    static unsigned long long h[256];
    const int frames = 64, ops = 256;   // 3 passes x 256 = 768 f64 instructions per frame
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    (void)hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        k<MODE><<<256, threads, (threads / 64) * (1024 + 64) * 8>>>(out, clk, frames, ops);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    (void)hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    double s = 0;
    for (int b = 0; b < 256; b++) s += h[b];
    const int wps = threads / 256;
    printf("%-10s %4d threads: %8.0f cycles per frame per SIMD-wave slot (%.1f us kernel; %d frames per wave)\n", name, threads,
           s / 256 / frames, ms * 1e3, frames);
    (void)wps;
}

int main()
{
    double *out; unsigned long long *clk;
    (void)hipMalloc(&out, 8 * 256 * 1024); (void)hipMalloc(&clk, 8 * 256);
    for (int threads : {256, 512, 768, 1024}) {
        run<1>("valu only", out, clk, threads);
        run<2>("lds only", out, clk, threads);
        run<0>("valu+lds", out, clk, threads);
    }
    return 0;
}
