// Measurement: issue cost (cycles per wave-instruction per SIMD) of the VALU instructions the frame loop is made of, at 1-4 waves
// per SIMD on every CU.  16 independent chains per wave, so dependency latency does not show.
// Build and run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/op_cost.hip -o /tmp/op_cost && /tmp/op_cost
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAINS 16
enum { MUL64, ADD64, MULADD64, FMA64, MULS64, MAX64, CVT64_32, CVT32_64, FMA32, MUL32, LOG32, FLR32, FRACT32, MED332, SWAP32, CNDMASK, ADDU32, CMPF32, CMPF64,
       CVTU32, MED3U32, MAX3F32, MAXSDWA, ADDSDWA, MOVSDWA, LSHLSDWA, ADDLSHL, LSHLOR, LSHL32, PKFMA32, READFL, NOPS };
static const char *names[] = {"v_mul_f64", "v_add_f64", "mul+add f64", "v_fma_f64", "v_mul_f64 sgpr", "v_max_f64", "v_cvt_f64_f32", "v_cvt_f32_f64",
                              "v_fma_f32", "v_mul_f32", "v_log_f32", "v_cvt_flr_i32_f32", "v_fract_f32", "v_med3_f32", "v_permlane32_swap",
                              "v_cndmask_b32", "v_add_u32", "v_cmp_gt_f32", "v_cmp_ge_f64",
                              // round 5: what a fixed-point epilogue would be made of, and the instructions the census changed
                              "v_cvt_u32_f32", "v_med3_u32", "v_max3_f32", "v_max_u32_sdwa W0", "v_add_u32_sdwa W1", "v_mov_b32_sdwa B",
                              "v_lshlrev_sdwa B1", "v_add_lshl_u32", "v_lshl_or_b32", "v_lshlrev_b32", "v_pk_fma_f32", "v_readfirstlane"};

template <int OP>
__global__ void k(double *out, unsigned long long *clk, int iters, double sc)
{
    double a[CHAINS], b[CHAINS];
    float f[CHAINS], g[CHAINS];
    unsigned u[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; i++) {
        a[i] = 1.0 + (double)(threadIdx.x + i) * 1e-9;
        b[i] = 1.0 + (double)(threadIdx.x * 16 + i) * 1e-12;
        f[i] = 1.0f + (float)(threadIdx.x + i) * 1e-6f;
        g[i] = 1.0f + (float)i * 1e-7f;
        u[i] = threadIdx.x + i;
    }
    __syncthreads();
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 2; rep++) {
#pragma unroll
            for (int i = 0; i < CHAINS; i++) {
                if (OP == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == MULADD64) {
                    if (rep == 0) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                    else asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                }
                if (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == MULS64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "s"(sc));
                if (OP == MAX64) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == CVT64_32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
                if (OP == CVT32_64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
                if (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(g[i]));
                if (OP == MUL32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g[i]));
                if (OP == LOG32) asm volatile("v_log_f32 %0, %1" : "=v"(f[i]) : "v"(g[i]));
                if (OP == FLR32) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(u[i]) : "v"(g[i]));
                if (OP == FRACT32) asm volatile("v_fract_f32 %0, %1" : "=v"(f[i]) : "v"(g[i]));
                if (OP == MED332) asm volatile("v_med3_f32 %0, %0, %1, 1.0" : "+v"(f[i]) : "v"(g[i]));
                if (OP == SWAP32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 1) % CHAINS]));
                if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(f[i]));
                if (OP == ADDU32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(f[i]));
                if (OP == CMPF32) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(f[i]), "v"(g[i]) : "vcc");
                if (OP == CMPF64) asm volatile("v_cmp_ge_f64 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");
                if (OP == CVTU32) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[i]) : "v"(g[i]));
                if (OP == MED3U32) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(f[i]), "v"(g[i]));
                if (OP == MAX3F32) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(g[i]), "v"(g[(i + 1) % CHAINS]));
                if (OP == MAXSDWA) asm volatile("v_max_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(u[i]) : "v"(f[i]));
                if (OP == ADDSDWA) asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(u[i]) : "v"(f[i]), "v"(g[i]));
                if (OP == MOVSDWA) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(u[i]) : "v"(f[i]));
                if (OP == LSHLSDWA) asm volatile("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(u[i]) : "v"(f[i]));
                if (OP == ADDLSHL) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(u[i]) : "v"(f[i]));
                if (OP == LSHLOR) asm volatile("v_lshl_or_b32 %0, %1, 8, %0" : "+v"(u[i]) : "v"(f[i]));
                if (OP == LSHL32) asm volatile("v_lshlrev_b32 %0, 2, %1" : "=v"(u[i]) : "v"(f[i]));
                if (OP == PKFMA32) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == READFL) { unsigned sx; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sx) : "v"(u[i])); asm volatile("" ::"s"(sx)); }
            }
        }
    }
    __syncthreads();   // the slowest wave of the workgroup ends the measurement (issue arbitration favours the oldest wave)
    const unsigned long long c1 = clock64();
    double r = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) r += a[i] + b[i] + (double)f[i] + (double)g[i] + (double)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

template <int OP>
void run(double *out, unsigned long long *clk, int blocks)
{
    static unsigned long long h[1024];
    const int iters = 20000;
    printf("%-20s", names[OP]);
    for (int threads = 256; threads <= 1024; threads += 256) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            k<OP><<<blocks, threads>>>(out, clk, iters, 1.0000001);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        (void)hipMemcpy(h, clk, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
        double s = 0;
        for (int b = 0; b < blocks; b++) s += h[b];
        const int wps = threads / 256;
        // second figure: the same from the kernel's wall time at 2.4 GHz (launch overhead included)
        printf("  %dw: %5.2f (%5.2f)", wps, s / blocks / ((double)wps * iters * 2 * CHAINS), ms * 1e-3 * 2.4e9 / ((double)wps * iters * 2 * CHAINS));
    }
    printf("\n");
}

template <int OP>
void run_all(double *out, unsigned long long *clk, int blocks)
{
    if constexpr (OP < NOPS) {
        run<OP>(out, clk, blocks);
        run_all<OP + 1>(out, clk, blocks);
    }
}

int main()
{
    double *out; unsigned long long *clk;
    (void)hipMalloc(&out, 8 * 256 * 1024); (void)hipMalloc(&clk, 8 * 1024);
    for (int blocks : {8, 256}) {
        printf("---- %d blocks (one per CU), cycles per wave-instruction per SIMD ----\n", blocks);
        run_all<0>(out, clk, blocks);
    }
    return 0;
}
