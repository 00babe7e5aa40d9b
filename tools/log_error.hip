// Measurement: error of v_log_f32 (log2) on gfx950 over every positive normal f32, against log2 in f64, in units of the ulp of the
// f32 result and as an absolute error; maxima per binade of |result|.  The margins of k_frames' index test rest on it (sp_host.cpp).
//   hipcc -O3 --offload-arch=gfx950 tools/log_error.hip -o /tmp/log_error && /tmp/log_error
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double *max_ulp, double *max_abs)
{
    __shared__ double s_ulp[9], s_abs[9];
    if (threadIdx.x < 9) { s_ulp[threadIdx.x] = 0; s_abs[threadIdx.x] = 0; }
    __syncthreads();
    double lu[9] = {0}, la[9] = {0};
    const unsigned long long total = 0x7f800000ull - 0x00800000ull;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned bits = (unsigned)(0x00800000ull + i);
        const float x = __uint_as_float(bits);
        float r;
        asm volatile("v_log_f32 %0, %1" : "=v"(r) : "v"(x));
        const double ref = log2((double)x);
        const double aerr = fabs((double)r - ref);
        int e; frexp(ref == 0.0 ? 1e-300 : ref, &e);                 // |ref| in [2^(e-1), 2^e)
        const double ulp = ldexp(1.0, (e < -125 ? -125 : e) - 24);
        int b = e < 0 ? 0 : (e > 8 ? 8 : e);                          // binade of |result|: <1, [1,2), [2,4) ... [64,128), [128,..)
        const double u = aerr / ulp;
        if (u > lu[b]) lu[b] = u;
        if (aerr > la[b]) la[b] = aerr;
    }
    for (int b = 0; b < 9; b++) {
        atomicMax((unsigned long long *)&s_ulp[b], (unsigned long long)__double_as_longlong(lu[b]));
        atomicMax((unsigned long long *)&s_abs[b], (unsigned long long)__double_as_longlong(la[b]));
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        atomicMax((unsigned long long *)&max_ulp[threadIdx.x], (unsigned long long)__double_as_longlong(s_ulp[threadIdx.x]));
        atomicMax((unsigned long long *)&max_abs[threadIdx.x], (unsigned long long)__double_as_longlong(s_abs[threadIdx.x]));
    }
}
int main()
{
    double *d, h[18];
    (void)hipMalloc(&d, sizeof h); (void)hipMemset(d, 0, sizeof h);
    k<<<4096, 256>>>(d, d + 9);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[9] = {"|L| < 1", "[1,2)", "[2,4)", "[4,8)", "[8,16)", "[16,32)", "[32,64)", "[64,128)", ">= 128"};
    for (int b = 0; b < 9; b++) printf("v_log_f32 result %-9s max error %.3f ulp of the result, %.3e absolute\n", names[b], h[b], h[9 + b]);
    return 0;
}
