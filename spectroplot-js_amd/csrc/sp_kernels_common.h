// sp_kernels_common.h — device-side structures and helpers shared by the frame-loop kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "sp_formats.h"

namespace spk {

// Everything one launch of the frame loop (lib/worker.js:68-137) needs; passed by value.
struct FrameArgs {
    const uint8_t *bytes;        // raw capture in HBM
    int64_t nbytes;
    int64_t nelem;               // length of the reference's typed view over the buffer
    double stride;               // (sampleCount - n) / (width - 1)                          worker.js:50
    int32_t n, levels, width;
    int32_t channel_mode, waterfall, lut_len;
    int32_t in_bounds;           // every frame lies inside the buffer: unchecked loads are safe
    int32_t frame0, x_end;       // frames [frame0, x_end) of the image are rendered by this launch (sp_render renders in chunks)
    int32_t sample_width;        // bytes per complex sample
    const double *window;        // [n]
    const double *cos_t;         // [n/2]
    const double *sin_t;         // [n/2]
    const double *gray_edge;     // [lut_len]
    const double *cb_edge;       // [1001]
    const uint32_t *lut_rgba;    // [lut_len] packed r | g<<8 | b<<16 | 255<<24
    float gray_a, gray_b, cb_a, cb_b;
    // k_frames epilogue (sp_host.h Thresholds): t = a + b*log2(abs2) lowered by the margin m, risky when fract(t) >= thr
    float g2_a, g2_b, g2_thr, g2_m;
    float c2_a, c2_b, c2_thr, c2_m, c2_lo, c2_hi;
    uint8_t *rgba;               // [4*width*n] or nullptr
    unsigned long long *c_hist;  // [lut_len]   accumulators (zero before every launch, k_finish_frames moves them out)
    unsigned long long *cb_hist; // [1000]
    double *frame_min;           // [width] min over the frame of abs2 (NaN ignored), +inf if none
    double *frame_max;           // [width] max over the frame of abs2 (NaN ignored), 0 if none
    unsigned long long *mm_acc;  // [2] bit patterns of the min / max of abs2 over all frames ({+inf, 0} before every launch)
    double *scratch;             // scratch kernel only: gridDim.x * 2 * n doubles
    int32_t cells;               // k_frames: merged histogram cells in use (sp_host.h Thresholds)
    int32_t rgba_fast;           // rgba 16-byte aligned, width a multiple of 4 and < 2^24, image below 4 GiB
    // k_frames produces the request's side outputs itself (the scratch kernel leaves them to k_finish_frames): gauges per group of
    // frames; every workgroup adds its share of the histograms and of the dBfs range to the reply                worker.js:124-155
    int32_t first;               // this launch starts its request (sp_render's chunks: only the first one does): workgroup 0 clears the reply
    uint32_t seq;                // the request's number (never 0) ...
    unsigned int *flag;          // ... which workgroup 0 stores here once the reply's histograms are zero and its range (0, -200)
    uint8_t *gauge_mins, *gauge_maxs, *gauge_amps;   // [width] each, or nullptr
    double block_norm_db, gain, range;
    unsigned long long *out_c, *out_cb;              // the reply's histograms (device memory), or nullptr
    double *out_minmax;          // [2] or nullptr
    const uint16_t *cell_g, *cell_l;                 // merged-cell ranges per colour index / level (sp_host.h Thresholds)
};

// store into a Uint8ClampedArray: round half to even, clamp, NaN -> 0
__device__ inline uint8_t clamp_u8(double v)
{
    if (!(v > 0.0)) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)rint(v);
}

// d = dBfs - gain of one |X|^2 value, the reference's operation order                     worker.js:100,124-125
__device__ inline double d_of_abs2(double abs2, double block_norm_db, double gain) { return (5 * spjs::log10(abs2) + block_norm_db + gain) - gain; }

// frame start: ~~(0.5 + stride * x)                                                          worker.js:72
__host__ __device__ inline int32_t frame_start(double stride, int32_t x) { return spjs::to_int32(0.5 + stride * (double)x); }

// The same for launches whose frames all lie inside the buffer (sp_api.hip: in_bounds requires 0 <= 0.5 + stride * x < 2^31 - 1 for every
// frame, and passes stride = 0 for a one-frame image): ToInt32 is then plain truncation - one v_cvt_i32_f64 instead of the ~18
// instructions and two branches of the general conversion.
__device__ inline int32_t frame_start_in_bounds(double stride, int32_t x) { return (int32_t)(0.5 + stride * (double)x); }

// Exact colour index from the edge table: number of edges 1..lut_len-1 that are <= abs2 (NaN -> 0).
__device__ inline int32_t gray_exact(const double *edge, int32_t lut_len, double abs2)
{
    int32_t lo = 0, hi = lut_len - 1;   // answer in [lo, hi]
    while (lo < hi) {
        const int32_t mid = (lo + hi + 1) >> 1;
        if (abs2 >= edge[mid]) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// Exact centi-bel bin: -1 means "dropped" (negative key in the reference).
__device__ inline int32_t cb_bin_exact(const double *edge, double abs2)
{
    if (!(abs2 > 0.0) || abs2 == spjs::inf()) return 0;   // -inf / +inf / NaN dB: ToInt32 gives 0
    int32_t lo = 0, hi = SP_CB_HIST_SIZE;
    while (lo < hi) {
        const int32_t mid = (lo + hi + 1) >> 1;
        if (abs2 >= edge[mid]) lo = mid;
        else hi = mid - 1;
    }
    return SP_CB_HIST_SIZE - 1 - lo;
}

// image byte offset of bin i of frame x                                                       worker.js:90,115-117
__device__ inline size_t pixel_offset(int32_t n, int32_t width, int32_t waterfall, int32_t x, int32_t i)
{
    const int32_t half = n >> 1;
    const int32_t y = i <= half ? half - i : half + n - i;
    return waterfall ? ((size_t)n * (size_t)(width - 1 - x) + (size_t)(n - 1 - y)) * 4
                     : ((size_t)x + (size_t)width * (size_t)y) * 4;
}

// fmin / fmax that ignore NaN operands (v_min_f64 / v_max_f64 semantics)
__device__ inline double min_nn(double a, double b) { return fmin(a, b); }
__device__ inline double max_nn(double a, double b) { return fmax(a, b); }

}  // namespace spk
