// sp_kernel_scratch.h — the portable frame-loop kernel: one workgroup per frame, the frame kept in an HBM/L2
// scratch slab, one radix-2 stage per barrier, or four from n = 4096 (a thread takes the 16 positions that differ in the
// four stage bits through all four stages in registers: a quarter of the slab traffic).  It handles
// every (format, n, width, stride, layout) the library accepts, including out-of-range frames and n beyond what fits in
// LDS.  It is the fallback behind the frame loop (sp_kernel_frames.h) and the in-library cross-check for it - written
// with run-time loops and none of that kernel's templates - not the fast path.
#pragma once

#include "sp_kernels_common.h"

namespace spk {

constexpr int kScratchThreads = 256;

__device__ inline uint32_t bit_reverse(uint32_t x, int bits) { return __brev(x) >> (32 - bits); }

// Stages s0 .. s0+G-1 of the radix-2 decimation in time (fft_nayuki.js:72-88) on the slab: an item is the 2^G positions that differ
// in bits [s0-1, s0-1+G); every butterfly of those stages inside the item pairs two of them, in the reference's operation order.
template <int G>
__device__ inline void scratch_stages(double *re, double *im, const double *__restrict__ cos_t, const double *__restrict__ sin_t, int n,
                                      int levels, int s0, int tid)
{
    constexpr int P = 1 << G;
    const int lowmask = (1 << (s0 - 1)) - 1;
    for (int b = tid; b < (n >> G); b += kScratchThreads) {
        const int low0 = b & lowmask;
        const int base = ((b >> (s0 - 1)) << (s0 - 1 + G)) | low0;
        double r[P], q[P];
#pragma unroll
        for (int e = 0; e < P; e++) {
            r[e] = re[base | (e << (s0 - 1))];
            q[e] = im[base | (e << (s0 - 1))];
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int s = s0 + g;                                   // this stage: size 2^s, partner distance 2^(s-1) = item bit g
#pragma unroll
            for (int e0 = 0; e0 < P; e0++) {
                if (e0 & (1 << g)) continue;
                const int e1 = e0 | (1 << g);
                // k = (position bits below bit s-1) * (n / size)                                   fft_nayuki.js:76-78
                const int k = (low0 | ((e0 & ((1 << g) - 1)) << (s0 - 1))) << (levels - s);
                const double c = cos_t[k], sn = sin_t[k];
                const double tpre = r[e1] * c + q[e1] * sn;
                const double tpim = -r[e1] * sn + q[e1] * c;
                r[e1] = r[e0] - tpre;
                q[e1] = q[e0] - tpim;
                r[e0] = r[e0] + tpre;
                q[e0] = q[e0] + tpim;
            }
        }
#pragma unroll
        for (int e = 0; e < P; e++) {
            re[base | (e << (s0 - 1))] = r[e];
            im[base | (e << (s0 - 1))] = q[e];
        }
    }
}

// WIDE: four stages per barrier (n >= 4096, where every thread still has an item); the narrow instance keeps the plain one-stage
// loop and its small register footprint (the latency-bound small sizes need the occupancy: n = 32 ran 1.8 x slower in one kernel
// with the 16-point items).
template <int FMT, bool WIDE>
__global__ __launch_bounds__(kScratchThreads) void k_scratch_radix2(const FrameArgs a)
{
    __shared__ unsigned int s_c_hist[SP_MAX_LUT];
    __shared__ unsigned int s_cb_hist[SP_CB_HIST_SIZE];
    __shared__ double s_red[2 * (kScratchThreads / 64)];

    const int tid = threadIdx.x;
    const int n = a.n;
    double *re = a.scratch + (size_t)blockIdx.x * 2 * (size_t)n;
    double *im = re + n;
    const spfmt::View view{a.bytes, a.nbytes, a.nelem};

    for (int i = tid; i < a.lut_len; i += kScratchThreads) s_c_hist[i] = 0;
    for (int i = tid; i < SP_CB_HIST_SIZE; i += kScratchThreads) s_cb_hist[i] = 0;
    __syncthreads();

    double blk_mn = spjs::inf(), blk_mx = 0.0;   // thread 0: over this workgroup's frames
    for (int x = a.frame0 + blockIdx.x; x < a.x_end; x += gridDim.x) {
        const int64_t start = frame_start(a.stride, x);

        // decode + taper, stored in bit-reversed order (fft_nayuki.js:57-69)            worker.js:70-75
        for (int k = tid; k < n; k += kScratchThreads) {
            double vi, vq;
            if (a.in_bounds) {
                spfmt::sample_fast<FMT>(a.bytes, start + k, vi, vq);
            } else {
                vi = spfmt::sample_checked<FMT>(view, start + k, 0);
                vq = spfmt::sample_checked<FMT>(view, start + k, 1);
            }
            const double w = a.window[k];
            const uint32_t j = a.levels ? bit_reverse((uint32_t)k, a.levels) : 0;
            re[j] = w * vi;
            im[j] = w * vq;
        }
        __syncthreads();

        // radix-2 decimation in time, same butterfly arithmetic as fft_nayuki.js:72-88, up to four stages between barriers
        if constexpr (!WIDE) {
            for (int s = 1; s <= a.levels; s++) {
                const int half = 1 << (s - 1);
                for (int b = tid; b < (n >> 1); b += kScratchThreads) {
                    const int lowbits = b & (half - 1);
                    const int j = ((b >> (s - 1)) << s) | lowbits;
                    const int l = j + half;
                    const int k = lowbits << (a.levels - s);
                    const double c = a.cos_t[k], sn = a.sin_t[k];
                    const double rl = re[l], il = im[l];
                    const double tpre = rl * c + il * sn;
                    const double tpim = -rl * sn + il * c;
                    const double rj = re[j], ij = im[j];
                    re[l] = rj - tpre;
                    im[l] = ij - tpim;
                    re[j] = rj + tpre;
                    im[j] = ij + tpim;
                }
                __syncthreads();
            }
        } else {
            for (int s = 1; s <= a.levels; s += 4) {
                const int left = a.levels - s + 1;
                if (left >= 4) scratch_stages<4>(re, im, a.cos_t, a.sin_t, n, a.levels, s, tid);
                else if (left == 3) scratch_stages<3>(re, im, a.cos_t, a.sin_t, n, a.levels, s, tid);
                else if (left == 2) scratch_stages<2>(re, im, a.cos_t, a.sin_t, n, a.levels, s, tid);
                else scratch_stages<1>(re, im, a.cos_t, a.sin_t, n, a.levels, s, tid);
                __syncthreads();
            }
        }

        if (a.channel_mode) {   // fft_nayuki.js:103-119
            for (int i = 1 + tid; i < (n >> 1); i += kScratchThreads) {
                const double ra = re[i], rb = re[n - i], ia = im[i], ib = im[n - i];
                re[i] = 0.5 * (ra + rb);
                im[i] = 0.5 * (ia - ib);
                re[n - i] = 0.5 * (ia + ib);
                im[n - i] = 0.5 * (-ra + rb);
            }
            if (tid == 0) {
                im[0] = 0.0;
                re[n >> 1] = 0.0;   // = imag[0], already zeroed in the reference
                im[n >> 1] = 0.0;
            }
            __syncthreads();
        }

        // |X|^2 -> indices -> histograms -> RGBA                                          worker.js:85-122
        double mn = spjs::inf(), mx = 0.0;
        for (int i = tid; i < n; i += kScratchThreads) {
            const double r = re[i], q = im[i];
            const double abs2 = r * r + q * q;
            mn = min_nn(mn, abs2);
            mx = max_nn(mx, abs2);
            const int gray = gray_exact(a.gray_edge, a.lut_len, abs2);
            const int bin = cb_bin_exact(a.cb_edge, abs2);
            atomicAdd(&s_c_hist[gray], 1u);
            if (bin >= 0) atomicAdd(&s_cb_hist[bin], 1u);
            if (a.rgba) *(uint32_t *)(a.rgba + pixel_offset(n, a.width, a.waterfall, x, i)) = a.lut_rgba[gray];
        }
        for (int off = 32; off > 0; off >>= 1) {
            mn = min_nn(mn, __shfl_xor(mn, off));
            mx = max_nn(mx, __shfl_xor(mx, off));
        }
        if ((tid & 63) == 0) {
            s_red[2 * (tid >> 6)] = mn;
            s_red[2 * (tid >> 6) + 1] = mx;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < kScratchThreads / 64; w++) {
                mn = min_nn(mn, s_red[2 * w]);
                mx = max_nn(mx, s_red[2 * w + 1]);
            }
            a.frame_min[x] = mn;
            a.frame_max[x] = mx;
            blk_mn = min_nn(blk_mn, mn);
            blk_mx = max_nn(blk_mx, mx);
        }
        __syncthreads();   // scratch slab and s_red are reused by the next frame
    }

    if (tid == 0) {   // abs2 >= +0 and never NaN here: the order of the doubles is the order of their bit patterns
        atomicMin(&a.mm_acc[0], (unsigned long long)__double_as_longlong(blk_mn));
        atomicMax(&a.mm_acc[1], (unsigned long long)__double_as_longlong(blk_mx));
    }
    for (int i = tid; i < a.lut_len; i += kScratchThreads)
        if (s_c_hist[i]) atomicAdd(&a.c_hist[i], (unsigned long long)s_c_hist[i]);
    for (int i = tid; i < SP_CB_HIST_SIZE; i += kScratchThreads)
        if (s_cb_hist[i]) atomicAdd(&a.cb_hist[i], (unsigned long long)s_cb_hist[i]);
}

// ---- per-frame side outputs (worker.js:124-136) and the global dBfs range (worker.js:124-125) -----------

struct FinishArgs {
    const uint8_t *bytes;
    int64_t nbytes, nelem;
    double stride;
    int32_t n, width, format;
    double block_norm_db, gain, range;
    const double *frame_min, *frame_max;
    uint8_t *gauge_mins, *gauge_maxs, *gauge_amps;
    unsigned long long *mm_acc;   // [2] bit patterns of min / max |X|^2 over all frames; reset to {+inf, 0} here
    double *out_minmax;       // [2] or nullptr
    // histograms: the scratch kernel adds into context-owned accumulators (zero before and after every request);
    // this kernel moves them to the caller's arrays, so a reply always holds the counts of its own request
    int32_t lut_len;
    unsigned long long *acc_c, *acc_cb;
    unsigned long long *out_c, *out_cb;   // or nullptr
};

__device__ inline double centre_sample(int fmt, const spfmt::View &v, int64_t pos, int c)
{
    switch (fmt) {
#define SP_CASE(F) case F: return spfmt::sample_checked<F>(v, pos, c);
        SP_CASE(SP_FMT_CU4) SP_CASE(SP_FMT_CS4) SP_CASE(SP_FMT_CS8) SP_CASE(SP_FMT_CU12) SP_CASE(SP_FMT_CS12)
        SP_CASE(SP_FMT_CU16) SP_CASE(SP_FMT_CS16) SP_CASE(SP_FMT_CU32) SP_CASE(SP_FMT_CS32) SP_CASE(SP_FMT_CU64)
        SP_CASE(SP_FMT_CS64) SP_CASE(SP_FMT_CF32) SP_CASE(SP_FMT_CF64)
#undef SP_CASE
    default: return spfmt::sample_checked<SP_FMT_CU8>(v, pos, c);
    }
}

constexpr int kFinishThreads = 256;

// The side outputs behind the scratch kernel (k_frames produces its own, sp_kernel_frames.h).
// Grid: 3 * ceil(width / kFinishThreads) workgroups (at least enough threads for the histograms) + 1.  The three software log10
// evaluations a frame needs are independent, so they run in different workgroups (role = block % 3) instead of as one long
// dependent chain per thread.
__global__ __launch_bounds__(kFinishThreads) void k_finish_frames(const FinishArgs a)
{
    const int fb = (int)blockIdx.x;
    const int role = fb % 3;
    const int x = (fb / 3) * kFinishThreads + threadIdx.x;
    if (x < a.width) {
        // d is monotone in abs2, so the frame's extreme d values come from its extreme abs2 values
        if (role == 0) {
            if (a.gauge_mins) {
                const double dlo = d_of_abs2(a.frame_min[x], a.block_norm_db, a.gain);
                double fmin = 0.0;   // worker.js:82
                if (dlo < fmin) fmin = dlo;
                a.gauge_mins[x] = clamp_u8(0.5 + (a.range + fmin) * 256 / a.range);
            }
        } else if (role == 1) {
            if (a.gauge_maxs) {
                const double dhi = d_of_abs2(a.frame_max[x], a.block_norm_db, a.gain);
                double fmax = -200.0;   // worker.js:83
                if (dhi > fmax) fmax = dhi;
                a.gauge_maxs[x] = clamp_u8(0.5 + (a.range + fmax) * 256 / a.range);
            }
        } else if (a.gauge_amps) {
            const spfmt::View v{a.bytes, a.nbytes, a.nelem};
            const int64_t mid = (int64_t)frame_start(a.stride, x) + (a.n >> 1);
            const double ci = centre_sample(a.format, v, mid, 0), cq = centre_sample(a.format, v, mid, 1);
            const double amp = 5 * spjs::log10(ci * ci + cq * cq) + a.gain;
            a.gauge_amps[x] = clamp_u8(0.5 + (a.range + amp) * 256 / a.range);
        }
    }
    {
        // histograms: accumulators -> reply, accumulators back to zero
        const int gi = blockIdx.x * kFinishThreads + threadIdx.x;
        if (gi < a.lut_len) {
            const unsigned long long v = a.acc_c[gi];
            if (a.out_c) a.out_c[gi] = v;
            a.acc_c[gi] = 0ull;
        }
        if (gi < SP_CB_HIST_SIZE) {
            const unsigned long long v = a.acc_cb[gi];
            if (a.out_cb) a.out_cb[gi] = v;
            a.acc_cb[gi] = 0ull;
        }
    }
    // global dBfs range (worker.js:35-36,124-125): min over frames of min(0, d(frame_min)) = min(0, d(min over frames)),
    // by the same monotonicity; the frame-loop kernels left the extreme |X|^2 of the whole launch in mm_acc
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x < 2) {   // the last workgroup serves no frames and no histogram (sp_api.hip)
        const double v = __longlong_as_double((long long)a.mm_acc[threadIdx.x]);
        const double d = d_of_abs2(v, a.block_norm_db, a.gain);
        if (threadIdx.x == 0) {
            if (a.out_minmax) a.out_minmax[0] = d < 0.0 ? d : 0.0;
            a.mm_acc[0] = 0x7ff0000000000000ull;
        } else {
            if (a.out_minmax) a.out_minmax[1] = d > -200.0 ? d : -200.0;
            a.mm_acc[1] = 0ull;
        }
    }
}

}  // namespace spk
