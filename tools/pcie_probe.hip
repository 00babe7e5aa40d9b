// pcie_probe.hip - what the host link gives a config-2 message (128 MiB of samples in, 64 MiB of image out), piece by piece:
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pcie_probe tools/pcie_probe.hip && /tmp/pcie_probe
// (a) the samples in one copy and in k chunks, page-locked source; (b) the image out in one copy, as k column bands (pitched copies: n rows
// of 4 * W / k bytes each, the spectrogram layout's chunks) and as k row bands (contiguous: the waterfall layout's); (c) both directions at
// once on two streams, the way sp_render overlaps them.  Prints milliseconds and GB/s; the best of (c) is the floor of a message.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    const size_t in_bytes = (size_t)128 << 20, n = 1024, W = 16384, out_bytes = 4 * n * W;
    void *h_in, *h_out, *d_in, *d_out;
    CK(hipHostMalloc(&h_in, in_bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&h_out, out_bytes, hipHostMallocDefault));
    CK(hipMalloc(&d_in, in_bytes));
    CK(hipMalloc(&d_out, out_bytes));
    hipStream_t s_in, s_out;
    CK(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
    auto in_chunks = [&](int k) {
        for (int c = 0; c < k; c++) CK(hipMemcpyAsync((char *)d_in + in_bytes / k * c, (char *)h_in + in_bytes / k * c, in_bytes / k, hipMemcpyHostToDevice, s_in));
    };
    auto out_cols = [&](int k) {
        for (int c = 0; c < k; c++)
            CK(hipMemcpy2DAsync((char *)h_out + 4 * W / k * c, 4 * W, (char *)d_out + 4 * W / k * c, 4 * W, 4 * W / k, n, hipMemcpyDeviceToHost, s_out));
    };
    auto out_rows = [&](int k) {
        for (int c = 0; c < k; c++) CK(hipMemcpyAsync((char *)h_out + out_bytes / k * c, (char *)d_out + out_bytes / k * c, out_bytes / k, hipMemcpyDeviceToHost, s_out));
    };
    auto timed = [&](const char *what, int k, double bytes, auto &&f) {
        double best = 1e30;
        for (int r = 0; r < 6; r++) {
            CK(hipDeviceSynchronize());
            const double t0 = now_ms();
            f();
            CK(hipStreamSynchronize(s_in));
            CK(hipStreamSynchronize(s_out));
            const double t = now_ms() - t0;
            if (r && t < best) best = t;
        }
        printf("%-46s k=%2d  %7.3f ms  %6.1f GB/s\n", what, k, best, bytes / best / 1e6);
    };
    // (d) a sparse request's pitched upload: 2048 rows of 8.5 KiB, 64 KiB apart in the capture (page-locked and pageable), in k calls
    {
        void *h_page = malloc(in_bytes);
        memset(h_page, 1, in_bytes);
        const size_t rows = 2048, row = 8704, spitch = 65536, dpitch = 8704;
        for (void *src : {h_in, h_page})
            for (int k : {1, 4, 16})
                timed(src == h_in ? "pitched upload, page-locked (17 MiB)" : "pitched upload, pageable (17 MiB)", k, (double)(rows * row), [&] {
                    for (int c = 0; c < k; c++)
                        CK(hipMemcpy2DAsync((char *)d_in + dpitch * (rows / k) * c, dpitch, (char *)src + spitch * (rows / k) * c, spitch, row, rows / k,
                                            hipMemcpyHostToDevice, s_in));
                });
        timed("the same bytes in one linear copy, page-locked", 1, (double)(rows * row), [&] { CK(hipMemcpyAsync(d_in, h_in, rows * row, hipMemcpyHostToDevice, s_in)); });
        timed("the same bytes in one linear copy, pageable", 1, (double)(rows * row), [&] { CK(hipMemcpyAsync(d_in, h_page, rows * row, hipMemcpyHostToDevice, s_in)); });
        timed("8 MiB image out, one copy", 1, (double)(8 << 20), [&] { CK(hipMemcpyAsync(h_out, d_out, 8 << 20, hipMemcpyDeviceToHost, s_out)); });
        free(h_page);
    }
    for (int k : {1, 4, 8, 16}) timed("samples in (128 MiB, page-locked)", k, (double)in_bytes, [&] { in_chunks(k); });
    for (int k : {1, 4, 8, 16}) timed("image out as column bands (pitched)", k, (double)out_bytes, [&] { out_cols(k); });
    for (int k : {1, 4, 8, 16}) timed("image out as row bands (contiguous)", k, (double)out_bytes, [&] { out_rows(k); });
    for (int k : {1, 4, 8, 16}) timed("both ways, column bands", k, (double)(in_bytes + out_bytes), [&] { in_chunks(k); out_cols(k); });
    for (int k : {1, 4, 8, 16}) timed("both ways, row bands", k, (double)(in_bytes + out_bytes), [&] { in_chunks(k); out_rows(k); });
    return 0;
}
