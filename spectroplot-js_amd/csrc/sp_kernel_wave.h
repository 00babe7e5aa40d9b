// sp_kernel_wave.h — the barrier-free frame-loop kernel for 64 <= n <= 1024 plus the colourise / transpose kernel.
//
// Why two kernels.  At n <= 1024 a frame fits one wavefront (16 points per lane), and the only LDS a frame needs is its
// 8.5 KiB exchange buffer.  Keeping the image transposition out of this kernel removes every workgroup barrier and the
// 32 KiB colour-index tile, so 12 waves (3 per SIMD) fit a CU instead of 8, and a wave that waits on LDS or HBM is
// covered by two others instead of one.  The price is a one-byte-per-pixel colour-index plane between the kernels
// (1/12 of the algorithmic traffic of the cf32 configuration); it is written and re-read in chunks small enough to
// stay in the 256 MiB Infinity Cache.
//
//   k_wave_r16   input bytes -> taper -> DFT (same register passes and arithmetic as sp_kernel_lds.h) -> |X|^2 ->
//                colour index byte per bin, centi-bel histogram, per-frame min / max of |X|^2
//   k_colorize   colour-index plane -> RGBA through the LUT, transposed for the spectrogram layout (64 x 64 tiles,
//                256-byte row segments) or rotated for the waterfall layout; colour histogram
#pragma once

#include "sp_kernel_lds.h"

namespace spk {

constexpr int kWaveThreads = 768;            // 12 waves = 3 per SIMD
constexpr int kWaveMinLog2 = 6, kWaveMaxLog2 = 10;

__host__ __device__ inline bool wave_kernel_supports(int n) { return n >= (1 << kWaveMinLog2) && n <= (1 << kWaveMaxLog2) && (n & (n - 1)) == 0; }

struct WaveLayout {
    int off_tw, off_win, off_gedge, off_cbedge, off_cbhist, off_trash, off_pack, total;
};

__host__ __device__ inline WaveLayout wave_layout(int n, int lut_len)
{
    WaveLayout l;
    const int fpb = kWaveThreads * 16 / n;
    int o = fpb * (n + n / 16) * 8;
    l.off_tw = o;     o += lds_tw_entries(n) * 16;
    l.off_win = o;    o += n * 8;                       // taper, kept out of the register file (3 waves per SIMD)
    l.off_gedge = o;  o += lut_len * 8;
    l.off_cbedge = o; o += (SP_CB_HIST_SIZE + 1) * 8;
    l.off_cbhist = o; o += SP_CB_HIST_SIZE * 4;
    l.off_trash = o;  o += kWaveThreads * 4;
    l.off_pack = o;   o += fpb * n;                     // colour-index bytes of a frame, re-read as dwords
    l.total = (o + 15) & ~15;
    return l;
}

template <int LOG2N, bool CH>
__global__ __launch_bounds__(kWaveThreads) void k_wave_r16(const FrameArgs a, const int format, const double2 *__restrict__ stage_tw,
                                                           uint8_t *__restrict__ gray, const int frame_begin, const int frame_end)
{
    constexpr int N = 1 << LOG2N;
    constexpr int T = N / 16;                       // threads per frame (<= 64: a frame never leaves its wave)
    constexpr int FPW = 64 / T;                     // frames per wave
    constexpr int FPB = kWaveThreads / T;           // frames per workgroup pass
    constexpr int NPASS = (LOG2N + 3) / 4;
    static_assert(T <= 64, "one frame per wave at most");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const WaveLayout lay = wave_layout(N, a.lut_len);
    double *s_xch = (double *)smem;
    double2 *s_tw = (double2 *)(smem + lay.off_tw);
    double *s_win = (double *)(smem + lay.off_win);
    double *s_gedge = (double *)(smem + lay.off_gedge);
    double *s_cbedge = (double *)(smem + lay.off_cbedge);
    unsigned int *s_cbhist = (unsigned int *)(smem + lay.off_cbhist);
    unsigned int *const trash = (unsigned int *)(smem + lay.off_trash) + threadIdx.x;

    const int tid = threadIdx.x;
    const int fs = tid / T;                         // frame slot within the workgroup
    const int tl = tid % T;                         // thread within the frame
    double *xbuf = s_xch + fs * (N + N / 16);
    unsigned char *const pack = smem + lay.off_pack + fs * N;
    const int cmax = a.lut_len - 1;

    for (int i = tid; i < a.lut_len; i += kWaveThreads) s_gedge[i] = a.gray_edge[i];
    for (int i = tid; i <= SP_CB_HIST_SIZE; i += kWaveThreads) s_cbedge[i] = a.cb_edge[i];
    for (int i = tid; i < SP_CB_HIST_SIZE; i += kWaveThreads) s_cbhist[i] = 0;
    for (int i = tid; i < lds_tw_entries(N); i += kWaveThreads) s_tw[i] = stage_tw[16 + i];

    for (int i = tid; i < N; i += kWaveThreads) s_win[i] = a.window[i];
    // this thread's 16 taper coefficients sit at s_win[rev4(e)*T + rev(tl)]
    const double *const wbase = s_win + (int)(__brev((unsigned)tl) >> (32 - (LOG2N - 4)));
    __syncthreads();   // the only workgroup barrier before the final histogram flush

    const spfmt::View view{a.bytes, a.nbytes, a.nelem};
    uint32_t pf_word = 0;
    unsigned int cnt_cb_last = 0, cnt_cb0 = 0;
    const float gray_a = a.gray_a, gray_b = a.gray_b, cb_a = a.cb_a, cb_b = a.cb_b;
    const float gc_hi = (float)(cmax - 1);

    // exchange addresses of this thread: one base per window, immediates per register
    constexpr int WS1 = LOG2N >= 8 ? 4 : LOG2N - 4;
    constexpr int E1 = LOG2N >= 8 ? 8 : LOG2N;
    constexpr int WS2 = LOG2N - 4;
    double *const b0 = xbuf + pad_idx(win_pos(tl, 0, 0));
    double *const b1 = xbuf + pad_idx(win_pos(tl, 0, WS1));
    double *const b2 = xbuf + pad_idx(win_pos(tl, 0, WS2));

    // consecutive frame slots of the whole grid take consecutive frames: the chip sweeps the capture front to back
    const int step = gridDim.x * FPB;
    const int wave_first = frame_begin + blockIdx.x * FPB + (tid >> 6) * FPW;   // wave-uniform
    for (int xb = wave_first; xb < frame_end; xb += step) {
        const int xr = xb + (fs % FPW);
        const bool live = xr < frame_end;
        const int x = live ? xr : frame_end - 1;    // surplus slots recompute the last frame and discard it
        const int64_t start = frame_start(a.stride, x);

        // touch the cache lines of the frame this slot processes next, so that its loads hit L2 instead of HBM
        asm volatile("" ::"v"(pf_word));
        if (a.in_bounds && xr + step < frame_end && !(a.dbg & 32)) {
            const int lines = (N * a.sample_width + 127) >> 7;
            const int64_t nb = (int64_t)frame_start(a.stride, xr + step) * a.sample_width;
            for (int l = tl; l < lines; l += T) pf_word = *(const uint32_t *)(a.bytes + ((nb + (int64_t)l * 128) & ~(int64_t)3));
        }

        double re[16], im[16];
        double win[16];
#pragma unroll
        for (int e = 0; e < 16; e++) win[e] = wbase[rev4(e) * T];
        if (a.dbg & 8) {   // ablation: no input loads
#pragma unroll
            for (int e = 0; e < 16; e++) { re[e] = win[e] * (double)(tl + e); im[e] = win[e] * (double)(x & 255); }
        } else
        switch (format) {
#define SP_CASE(F) case F: load_frame<F>(a, view, start, tl, T, LOG2N, win, re, im); break;
            SP_CASE(SP_FMT_CU4) SP_CASE(SP_FMT_CS4) SP_CASE(SP_FMT_CU8) SP_CASE(SP_FMT_CS8) SP_CASE(SP_FMT_CU12)
            SP_CASE(SP_FMT_CS12) SP_CASE(SP_FMT_CU16) SP_CASE(SP_FMT_CS16) SP_CASE(SP_FMT_CU32) SP_CASE(SP_FMT_CS32)
            SP_CASE(SP_FMT_CU64) SP_CASE(SP_FMT_CS64) SP_CASE(SP_FMT_CF32)
#undef SP_CASE
        default: load_frame<SP_FMT_CF64>(a, view, start, tl, T, LOG2N, win, re, im); break;
        }

        // ---- DFT -------------------------------------------------------------------------------------------------
        if (!(a.dbg & 1)) fft_pass<0, 1, 4>(re, im, tl, s_tw, stage_tw);
        if (!(a.dbg & 2)) {
            exchange<0, WS1, false>(re, b0, b1);
            exchange<0, WS1, false>(im, b0, b1);
        }
        if (!(a.dbg & 1)) fft_pass<WS1, 5, E1>(re, im, tl, s_tw, stage_tw);
        if constexpr (NPASS >= 3) {
            if (!(a.dbg & 2)) {
                exchange<WS1, WS2, false>(re, b1, b2);
                exchange<WS1, WS2, false>(im, b1, b2);
            }
            if (!(a.dbg & 1)) fft_pass<WS2, 9, LOG2N>(re, im, tl, s_tw, stage_tw);
        }
        // register e of thread tl now holds bin i = tl + e*T

        if constexpr (CH) {   // fft_nayuki.js:103-119, partner bin n-i fetched through LDS
            double pr[16], pi[16];
            frame_sync<false>();
#pragma unroll
            for (int e = 0; e < 16; e++) xbuf[pad_idx(tl + e * T)] = re[e];
            frame_sync<false>();
#pragma unroll
            for (int e = 0; e < 16; e++) pr[e] = xbuf[pad_idx((N - (tl + e * T)) & (N - 1))];
            frame_sync<false>();
#pragma unroll
            for (int e = 0; e < 16; e++) xbuf[pad_idx(tl + e * T)] = im[e];
            frame_sync<false>();
#pragma unroll
            for (int e = 0; e < 16; e++) pi[e] = xbuf[pad_idx((N - (tl + e * T)) & (N - 1))];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int i = tl + e * T;
                const double orr = re[e], oi = im[e];
                if (i == 0) {
                    im[e] = 0.0;
                } else if (i == N / 2) {
                    re[e] = 0.0;
                    im[e] = 0.0;
                } else if (i < N / 2) {
                    re[e] = 0.5 * (orr + pr[e]);
                    im[e] = 0.5 * (oi - pi[e]);
                } else {
                    re[e] = 0.5 * (pi[e] + oi);
                    im[e] = 0.5 * (-pr[e] + orr);
                }
            }
        }

        // ---- |X|^2 -> colour index + centi-bel bin (see sp_kernel_lds.h for the first-guess / exact-edge scheme) ----
        double mn = spjs::inf(), mx = 0.0;
        if (a.dbg & 4) {   // ablation: no classification
            mn = re[0] + im[5];
            mx = re[3] + im[9];
#pragma unroll
            for (int e = 0; e < 16; e++) pack[tl + e * T] = (unsigned char)(re[e] > im[e] ? 1 : 2);
        } else
#pragma unroll
        for (int h = 0; h < 2; h++) {
            double abs2[8], ge[8], ce[8];
            int gc[8], lc[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int e = h * 8 + k;
                abs2[k] = re[e] * re[e] + im[e] * im[e];                           // worker.js:92
                mn = min_nn(mn, abs2[k]);
                mx = max_nn(mx, abs2[k]);
                int ex;
                const double mant = frexp(abs2[k], &ex);
                const float l2 = (float)ex + __log2f((float)mant);
                gc[k] = floor_to_int(__builtin_amdgcn_fmed3f(fmaf(gray_b, l2, gray_a), 0.0f, gc_hi));
                lc[k] = floor_to_int(__builtin_amdgcn_fmed3f(fmaf(cb_b, l2, cb_a), 0.0f, (float)(SP_CB_HIST_SIZE - 1)));
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                ge[k] = s_gedge[gc[k] + 1];
                ce[k] = s_cbedge[lc[k] + 1];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int e = h * 8 + k;
                const int gr = gc[k] + (abs2[k] >= ge[k] ? 1 : 0);
                const int lv = lc[k] + (abs2[k] >= ce[k] ? 1 : 0);
                const bool special = !(abs2[k] > 0.0) || abs2[k] == spjs::inf();   // ToInt32(+-inf / NaN) = 0  worker.js:105
                pack[tl + e * T] = (unsigned char)gr;
                const bool l0 = lv == 0, lx = lv == SP_CB_HIST_SIZE;
                cnt_cb0 += (live && special) ? 1u : 0u;
                cnt_cb_last += (live && !special && l0) ? 1u : 0u;
                if (!(a.dbg & 16)) atomicAdd((special || l0 || lx || !live) ? trash : &s_cbhist[SP_CB_HIST_SIZE - 1 - lv], 1u);
            }
        }
        // colour indices leave as dwords (sub-dword global stores are not combined by the memory pipeline)
        frame_sync<false>();
        if (live && !(a.dbg & 64)) {
            uint32_t *grow = (uint32_t *)(gray + (size_t)(x - frame_begin) * N);
#pragma unroll
            for (int k = 0; k < 4; k++) grow[tl + k * T] = ((const uint32_t *)pack)[tl + k * T];
        }
        frame_sync<false>();
#pragma unroll
        for (int off = T / 2; off > 0; off >>= 1) {
            mn = min_nn(mn, __shfl_xor(mn, off));
            mx = max_nn(mx, __shfl_xor(mx, off));
        }
        if (tl == 0 && live && !(a.dbg & 64)) {
            a.frame_min[x] = mn;
            a.frame_max[x] = mx;
        }
    }

    if (cnt_cb_last) atomicAdd(&s_cbhist[SP_CB_HIST_SIZE - 1], cnt_cb_last);
    if (cnt_cb0) atomicAdd(&s_cbhist[0], cnt_cb0);
    __syncthreads();
    for (int i = tid; i < SP_CB_HIST_SIZE; i += kWaveThreads)
        if (s_cbhist[i]) atomicAdd(&a.cb_hist[i], (unsigned long long)s_cbhist[i]);
}

// ---- colour-index plane -> RGBA --------------------------------------------------------------------------------------

struct ColorizeArgs {
    const uint8_t *gray;          // [frames][n], frame-major
    uint8_t *rgba;                // full image
    const uint32_t *lut_rgba;     // [lut_len]
    unsigned long long *c_hist;   // [lut_len] accumulated
    int32_t n, width, waterfall, lut_len;
    int32_t frame_begin, frame_end;
};

constexpr int kColorThreads = 256;
constexpr int kColorFrames = 64;              // tile: 64 frames x 64 bins
constexpr int kColorBins = 64;
constexpr int kColorPitch = kColorBins + 4;   // bytes; 17 dwords: frame quads land on different banks

__global__ __launch_bounds__(kColorThreads) void k_colorize(const ColorizeArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char tile[kColorFrames * kColorPitch];
    __shared__ unsigned int s_lut[256];
    __shared__ unsigned int s_hist[256];

    const int t = threadIdx.x;
    const int n = a.n;
    const int cmax = a.lut_len - 1;
    s_lut[t] = t < a.lut_len ? a.lut_rgba[t] : 0;
    s_hist[t] = 0;
    unsigned int c0 = 0, cm = 0;

    const int tiles_x = (a.frame_end - a.frame_begin + kColorFrames - 1) / kColorFrames;
    const int tiles_y = n / kColorBins;
    const int f = t >> 2, seg = t & 3;
    // a workgroup walks down the bins of one 64-frame column block before moving on (row segments of neighbouring
    // tiles share cache lines of the image only in x, so this order keeps its writes in one stripe)
    for (int tile_id = blockIdx.x; tile_id < tiles_x * tiles_y; tile_id += gridDim.x) {
        const int x0 = a.frame_begin + (tile_id / tiles_y) * kColorFrames;
        const int i0 = (tile_id % tiles_y) * kColorBins;
        __syncthreads();   // previous tile fully consumed (also orders the table initialisation)

        // load: 4 threads per frame, 16 colour indices each
        uint4 v = make_uint4(0, 0, 0, 0);
        const bool have = x0 + f < a.frame_end;
        if (have) v = *(const uint4 *)(a.gray + (size_t)(x0 + f - a.frame_begin) * n + i0 + seg * 16);
        uint32_t *trow = (uint32_t *)(tile + f * kColorPitch + seg * 16);
        trow[0] = v.x; trow[1] = v.y; trow[2] = v.z; trow[3] = v.w;

        // colour histogram (worker.js:113): the clipped ends are counted in registers, the rest with LDS atomics
        if (a.c_hist && have) {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int g = (w[k >> 2] >> (8 * (k & 3))) & 0xff;
                c0 += g == 0 ? 1u : 0u;
                cm += g == cmax ? 1u : 0u;
                if (g != 0 && g != cmax) atomicAdd(&s_hist[g], 1u);
            }
        }
        __syncthreads();
        if (!a.rgba) continue;

        if (!a.waterfall) {
            // image = n rows x width columns, row y holds bin (n/2 - y) mod n; 16 threads write one 256-byte row segment
            const int fq = t & 15;
            const int xa = x0 + 4 * fq;
            uint32_t g4[4][4];
#pragma unroll
            for (int pass = 0; pass < 4; pass++)
#pragma unroll
                for (int k = 0; k < 4; k++) g4[pass][k] = tile[(4 * fq + k) * kColorPitch + pass * 16 + (t >> 4)];
#pragma unroll
            for (int pass = 0; pass < 4; pass++) {
                const int i = i0 + pass * 16 + (t >> 4);
                const int y = (n / 2 - i) & (n - 1);
                uint32_t px[4];
#pragma unroll
                for (int k = 0; k < 4; k++) px[k] = s_lut[g4[pass][k]];
                uint8_t *dst = a.rgba + ((size_t)y * (size_t)a.width + (size_t)xa) * 4;
                if (xa + 3 < a.frame_end && (((size_t)dst & 15) == 0)) {
                    *(uint4 *)dst = make_uint4(px[0], px[1], px[2], px[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (xa + k < a.frame_end) ((uint32_t *)dst)[k] = px[k];
                }
            }
        } else {
            // image = width rows x n columns; frame x is row width-1-x, bin i is column (i + n/2 - 1) mod n
#pragma unroll 4
            for (int it = t; it < kColorFrames * kColorBins; it += kColorThreads) {
                const int ff = it >> 6, r = it & 63;
                const int xa = x0 + ff;
                if (xa >= a.frame_end) continue;
                const int col = (i0 + r + n / 2 - 1) & (n - 1);
                *(uint32_t *)(a.rgba + ((size_t)(a.width - 1 - xa) * n + (size_t)col) * 4) = s_lut[tile[ff * kColorPitch + r]];
            }
        }
    }
    if (c0) atomicAdd(&s_hist[0], c0);
    if (cm) atomicAdd(&s_hist[cmax], cm);
    __syncthreads();
    if (a.c_hist && s_hist[t]) atomicAdd(&a.c_hist[t], (unsigned long long)s_hist[t]);
}

// Host-side launch of the pair over [0, width) in chunks whose colour-index plane stays cache resident.
inline int launch_wave(const FrameArgs &a, int format, const double2 *stage_tw, uint8_t *gray, size_t gray_capacity, int cu_count,
                       hipStream_t stream)
{
    if (!wave_kernel_supports(a.n) || a.lut_len > kLdsMaxLut || a.lut_len < 2 || !(a.gray_b <= kLdsMaxGrayB)) return SP_ERR_UNSUPPORTED;
    const int n = a.n;
    const WaveLayout lay = wave_layout(n, a.lut_len);
    if (lay.total > 160 * 1024) return SP_ERR_UNSUPPORTED;
    long long chunk = (long long)(gray_capacity / (size_t)n) & ~63ll;
    if (chunk < 64) return SP_ERR_NOMEM;
    const int fpb = kWaveThreads * 16 / n;

    for (long long begin = 0; begin < a.width; begin += chunk) {
        const int fb = (int)begin;
        const int fe = (int)((begin + chunk < a.width) ? begin + chunk : a.width);
        const int frames = fe - fb;
        int grid = (frames + fpb - 1) / fpb;
        if (grid > cu_count) grid = cu_count;
#define SP_LAUNCH_CH(L, C)                                                                                                \
    {                                                                                                                    \
        static bool attr_set = false;                                                                                    \
        if (!attr_set) {                                                                                                 \
            if (hipFuncSetAttribute((const void *)k_wave_r16<L, C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) \
                != hipSuccess)                                                                                           \
                return SP_ERR_HIP;                                                                                       \
            attr_set = true;                                                                                             \
        }                                                                                                                \
        hipLaunchKernelGGL((k_wave_r16<L, C>), dim3((unsigned)grid), dim3(kWaveThreads), (size_t)lay.total, stream, a,   \
                           format, stage_tw, gray, fb, fe);                                                              \
    }
#define SP_LAUNCH(L)                                                                                                     \
    case L:                                                                                                              \
        if (a.channel_mode) SP_LAUNCH_CH(L, true) else SP_LAUNCH_CH(L, false)                                            \
        break;
        switch (a.levels) {
            SP_LAUNCH(6) SP_LAUNCH(7) SP_LAUNCH(8) SP_LAUNCH(9) SP_LAUNCH(10)
        default: return SP_ERR_UNSUPPORTED;
        }
#undef SP_LAUNCH
#undef SP_LAUNCH_CH
        ColorizeArgs c{};
        c.gray = gray;
        c.rgba = a.rgba;
        c.lut_rgba = a.lut_rgba;
        c.c_hist = a.c_hist;
        c.n = n;
        c.width = a.width;
        c.waterfall = a.waterfall;
        c.lut_len = a.lut_len;
        c.frame_begin = fb;
        c.frame_end = fe;
        {
            const long long tiles = (long long)((frames + kColorFrames - 1) / kColorFrames) * (n / kColorBins);
            const long long cgrid = tiles < 8ll * cu_count ? tiles : 8ll * cu_count;
            hipLaunchKernelGGL(k_colorize, dim3((unsigned)cgrid), dim3(kColorThreads), 0, stream, c);
        }
    }
    return SP_OK;
}

}  // namespace spk
