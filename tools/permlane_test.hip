// Prints what v_permlane16_swap / v_permlane32_swap do on gfx950 (used by the register transpose in sp_frame_parts.h, exchange_permlane).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *y)
{
    unsigned a = threadIdx.x, b = threadIdx.x + 1000;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    y[threadIdx.x] = r[0]; y[threadIdx.x + 64] = r[1]; y[threadIdx.x + 128] = q[0]; y[threadIdx.x + 192] = q[1];
}
int main()
{
    unsigned *d, h[256];
    (void)hipMalloc(&d, sizeof h);
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[4] = {"p16 new a", "p16 new b", "p32 new a", "p32 new b"};
    for (int v = 0; v < 4; v++) {
        printf("%s:", names[v]);
        for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[v * 64 + l]);
        printf("\n");
    }
    return 0;
}
