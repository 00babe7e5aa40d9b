// sp_kernel_wave.h — the barrier-free frame-loop kernel for 64 <= n <= 1024 (gfx950).
//
// At n <= 1024 a frame fits one wavefront (16 points per lane), so nothing in the frame loop needs a workgroup barrier.
// Phase stamps of the group-synchronous kernel (sp_kernel_lds.h) showed a quarter of every wave's time parked at the group
// barrier or in the lock-step tile write-out; here each wave owns a strip of 4 consecutive frames instead:
//
//   * same register passes, exchanges and epilogue arithmetic as sp_kernel_lds.h (bit-identical results);
//   * colour indices of the strip go to a private LDS tile [bin][4 frames] (one byte each, conflict-free), and when the
//     strip is complete the wave itself writes it through the RGBA LUT: one 16-byte store per image row (4 frames x RGBA).
//     The 8 waves of a workgroup own 8 adjacent strips, so together they fill 128-byte row segments that merge in the
//     XCD's L2 before they leave for HBM;
//   * waves drift apart freely, so one wave's write-out or HBM wait overlaps the others' butterflies.
#pragma once

#include "sp_kernel_lds.h"

namespace spk {

constexpr int kWaveThreads = 512;            // 8 waves; LDS (68 KiB of exchange buffers + tiles + tables) allows one workgroup per CU
constexpr int kWaveMinLog2 = 6, kWaveMaxLog2 = 10;

__host__ __device__ inline bool wave_kernel_supports(int n) { return n >= (1 << kWaveMinLog2) && n <= (1 << kWaveMaxLog2) && (n & (n - 1)) == 0; }

// frames per strip: 4 (16-byte row segments), or one round's worth when a wave holds more than 4 frames
__host__ __device__ inline constexpr int wave_strip_frames(int n) { return (64 * 16 / n) > 4 ? (64 * 16 / n) : 4; }

struct WaveLayout {
    int off_tw, off_gedge, off_cbedge, off_tile, off_lut, off_chist, off_cbhist, off_trash, total;
};

__host__ __device__ inline WaveLayout wave_layout(int n, int lut_len)
{
    WaveLayout l;
    const int fpb = kWaveThreads * 16 / n;
    int o = fpb * (n + n / 16) * 8;
    l.off_tw = o;     o += lds_tw_entries(n) * 16;
    l.off_gedge = o;  o += lut_len * 8;
    l.off_cbedge = o; o += (SP_CB_HIST_SIZE + 1) * 8;
    l.off_tile = o;   o += (kWaveThreads / 64) * n * wave_strip_frames(n);
    o = (o + 15) & ~15;
    l.off_lut = o;    o += lut_len * 4;
    l.off_chist = o;  o += lut_len * 4;
    l.off_cbhist = o; o += SP_CB_HIST_SIZE * 4;
    l.off_trash = o;  o += kWaveThreads * 4;
    l.total = (o + 15) & ~15;
    return l;
}

template <int LOG2N, bool CH>
__global__ __launch_bounds__(kWaveThreads) void k_wave_r16(const FrameArgs a, const int format, const double2 *__restrict__ stage_tw,
                                                           const int strips)
{
    constexpr int N = 1 << LOG2N;
    constexpr int T = N / 16;                       // threads per frame (<= 64: a frame never leaves its wave)
    constexpr int FPW = 64 / T;                     // frames per wave and round
    constexpr int SF = wave_strip_frames(N);        // frames per strip
    constexpr int RPS = SF / FPW;                   // rounds per strip
    constexpr int NPASS = (LOG2N + 3) / 4;
    constexpr int WAVES = kWaveThreads / 64;
    static_assert(T <= 64, "one frame per wave at most");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const WaveLayout lay = wave_layout(N, a.lut_len);
    double *s_xch = (double *)smem;
    double2 *s_tw = (double2 *)(smem + lay.off_tw);
    double *s_gedge = (double *)(smem + lay.off_gedge);
    double *s_cbedge = (double *)(smem + lay.off_cbedge);
    unsigned int *s_lut = (unsigned int *)(smem + lay.off_lut);
    unsigned int *s_chist = (unsigned int *)(smem + lay.off_chist);
    unsigned int *s_cbhist = (unsigned int *)(smem + lay.off_cbhist);
    unsigned int *const trash = (unsigned int *)(smem + lay.off_trash) + threadIdx.x;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int fs = tid / T;                         // frame slot within the workgroup
    const int fw = lane / T;                        // frame slot within the wave
    const int tl = tid % T;                         // thread within the frame
    double *xbuf = s_xch + fs * (N + N / 16);
    unsigned char *const tile = smem + lay.off_tile + wave * (N * SF);   // [bin][SF frames]
    const int cmax = a.lut_len - 1;

    {
        // request tables -> LDS, all global loads issued before the first LDS store
        constexpr int NTW = lds_tw_entries(N);
        constexpr int TWK = (NTW + kWaveThreads - 1) / kWaveThreads;
        double2 tw_r[TWK > 0 ? TWK : 1];
#pragma unroll
        for (int k = 0; k < TWK; k++) {
            const int i = tid + k * kWaveThreads;
            tw_r[k] = i < NTW ? stage_tw[i] : make_double2(0.0, 0.0);
        }
        const double ge_r = tid < a.lut_len ? a.gray_edge[tid] : 0.0;
        const unsigned int lut_r = tid < a.lut_len ? a.lut_rgba[tid] : 0u;
        double cb_r[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = tid + k * kWaveThreads;
            cb_r[k] = i <= SP_CB_HIST_SIZE ? a.cb_edge[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < TWK; k++) {
            const int i = tid + k * kWaveThreads;
            if (i < NTW) s_tw[i] = tw_r[k];
        }
        if (tid < a.lut_len) {
            s_gedge[tid] = ge_r;
            s_lut[tid] = lut_r;
            s_chist[tid] = 0;
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = tid + k * kWaveThreads;
            if (i <= SP_CB_HIST_SIZE) s_cbedge[i] = cb_r[k];
            if (i < SP_CB_HIST_SIZE) s_cbhist[i] = 0;
        }
    }
    double win[16];
    {
        const int sidx = (int)(__brev((unsigned)tl) >> (32 - (LOG2N - 4)));
#pragma unroll
        for (int e = 0; e < 16; e++) win[e] = a.window[rev4(e) * T + sidx];
    }
    __syncthreads();   // the only workgroup barrier before the final histogram flush

    const spfmt::View view{a.bytes, a.nbytes, a.nelem};
    uint32_t pf_word = 0;
    unsigned int cnt_g0 = 0, cnt_gmax = 0, cnt_cb_last = 0, cnt_cb0 = 0;
    const float gray_a = a.gray_a, gray_b = a.gray_b, cb_a = a.cb_a, cb_b = a.cb_b;
    const float gc_hi = (float)(cmax - 1);

    constexpr int WS1 = LOG2N >= 8 ? 4 : LOG2N - 4;
    constexpr int E1 = LOG2N >= 8 ? 8 : LOG2N;
    constexpr int WS2 = LOG2N - 4;
    double *const b0 = xbuf + pad_idx(win_pos(tl, 0, 0));
    double *const b1 = xbuf + pad_idx(win_pos(tl, 0, WS1));

    // strip s = frames [s*SF, (s+1)*SF); the waves of a workgroup take adjacent strips
    const int strip_step = gridDim.x * WAVES;
    for (int strip = blockIdx.x * WAVES + wave; strip < strips; strip += strip_step) {
        const int xs = strip * SF;
#pragma unroll 1
        for (int r = 0; r < RPS; r++) {
            const int f_in_strip = r * FPW + fw;
            const int xr = xs + f_in_strip;
            const bool live = xr < a.width;
            const int x = live ? xr : a.width - 1;  // surplus slots recompute the last frame and discard it
            const int64_t start = frame_start(a.stride, x);

            // touch the cache lines of the frame this slot processes next, so that its loads hit L2 instead of HBM
            asm volatile("" ::"v"(pf_word));
            {
                const int xn = (r + 1 < RPS) ? xr + FPW : (strip + strip_step) * SF + fw;
                if (a.in_bounds && xn < a.width) {
                    const int lines = (N * a.sample_width + 127) >> 7;
                    const int64_t nb = (int64_t)frame_start(a.stride, xn) * a.sample_width;
                    for (int l = tl; l < lines; l += T) pf_word = *(const uint32_t *)(a.bytes + ((nb + (int64_t)l * 128) & ~(int64_t)3));
                }
            }

            double re[16], im[16];
            switch (format) {
#define SP_CASE(F) case F: load_frame<F>(a, view, start, tl, T, LOG2N, win, re, im); break;
                SP_CASE(SP_FMT_CU4) SP_CASE(SP_FMT_CS4) SP_CASE(SP_FMT_CU8) SP_CASE(SP_FMT_CS8) SP_CASE(SP_FMT_CU12)
                SP_CASE(SP_FMT_CS12) SP_CASE(SP_FMT_CU16) SP_CASE(SP_FMT_CS16) SP_CASE(SP_FMT_CU32) SP_CASE(SP_FMT_CS32)
                SP_CASE(SP_FMT_CU64) SP_CASE(SP_FMT_CS64) SP_CASE(SP_FMT_CF32)
#undef SP_CASE
            default: load_frame<SP_FMT_CF64>(a, view, start, tl, T, LOG2N, win, re, im); break;
            }

            // ---- DFT ---------------------------------------------------------------------------------------------
            fft_pass<0, 1, 4>(re, im, tl, s_tw, stage_tw);
            exchange<0, WS1, false>(re, b0, b1);
            exchange<0, WS1, false>(im, b0, b1);
            fft_pass<WS1, 5, E1>(re, im, tl, s_tw, stage_tw);
            if constexpr (NPASS >= 3) {
                exchange_permlane<LOG2N>(re);       // v_permlane16_swap / v_permlane32_swap, no LDS
                exchange_permlane<LOG2N>(im);
                fft_pass<WS2, 9, LOG2N>(re, im, tl, s_tw, stage_tw);
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

            // ---- |X|^2 -> colour index + centi-bel bin (first guess + exact edge, see sp_kernel_lds.h) -----------------
            double mn = spjs::inf(), mx = 0.0;
            unsigned char *const tcol = tile + f_in_strip;
            double abs2[16];
            int gc[16], lc[16];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                abs2[e] = re[e] * re[e] + im[e] * im[e];                           // worker.js:92
                mn = min_nn(mn, abs2[e]);
                mx = max_nn(mx, abs2[e]);
                int ex;
                const double mant = frexp(abs2[e], &ex);
                const float l2 = (float)ex + __log2f((float)mant);
                gc[e] = floor_to_int(__builtin_amdgcn_fmed3f(fmaf(gray_b, l2, gray_a), 0.0f, gc_hi));
                lc[e] = floor_to_int(__builtin_amdgcn_fmed3f(fmaf(cb_b, l2, cb_a), 0.0f, (float)(SP_CB_HIST_SIZE - 1)));
            }
            double ge[16], ce[16];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                ge[e] = s_gedge[gc[e] + 1];
                ce[e] = s_cbedge[lc[e] + 1];
            }
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int gr = gc[e] + (abs2[e] >= ge[e] ? 1 : 0);
                const int lv = lc[e] + (abs2[e] >= ce[e] ? 1 : 0);
                const bool special = !(abs2[e] > 0.0) || abs2[e] == spjs::inf();   // ToInt32(+-inf / NaN) = 0   worker.js:105
                tcol[(tl + e * T) * SF] = (unsigned char)gr;
                const bool g0 = gr == 0, gm = gr == cmax;
                cnt_g0 += (live && g0) ? 1u : 0u;
                cnt_gmax += (live && gm) ? 1u : 0u;
                atomicAdd((g0 || gm || !live) ? trash : &s_chist[gr], 1u);
                const bool l0 = lv == 0, lx = lv == SP_CB_HIST_SIZE;
                cnt_cb0 += (live && special) ? 1u : 0u;
                cnt_cb_last += (live && !special && l0) ? 1u : 0u;
                atomicAdd((special || l0 || lx || !live) ? trash : &s_cbhist[SP_CB_HIST_SIZE - 1 - lv], 1u);
            }
#pragma unroll
            for (int off = T / 2; off > 0; off >>= 1) {
                mn = min_nn(mn, __shfl_xor(mn, off));
                mx = max_nn(mx, __shfl_xor(mx, off));
            }
            if (tl == 0 && live) {
                a.frame_min[x] = mn;
                a.frame_max[x] = mx;
            }
        }

        // ---- the wave writes its own strip: tile [bin][SF frames] -> RGBA ------------------------------------------------
        frame_sync<false>();
        if (a.rgba) {
            if (!a.waterfall) {
                // image = n rows x width columns, row y holds bin (n/2 - y) mod n                   worker.js:90,117
                // dword c of the tile = 4 consecutive frames of one bin = one 16-byte store
                constexpr int CHUNKS = N * SF / 4;
#pragma unroll 4
                for (int c = lane; c < CHUNKS; c += 64) {
                    const uint32_t g4 = ((const uint32_t *)tile)[c];
                    const int i = c / (SF / 4), q = c % (SF / 4);
                    const int y = (N / 2 - i) & (N - 1);
                    const int xa = xs + 4 * q;
                    const uint32_t p0 = s_lut[g4 & 0xff], p1 = s_lut[(g4 >> 8) & 0xff], p2 = s_lut[(g4 >> 16) & 0xff], p3 = s_lut[g4 >> 24];
                    uint8_t *dst = a.rgba + ((size_t)y * (size_t)a.width + (size_t)xa) * 4;
                    if (xa + 3 < a.width && (((size_t)dst & 15) == 0)) {
                        *(uint4 *)dst = make_uint4(p0, p1, p2, p3);
                    } else {
                        if (xa < a.width) ((uint32_t *)dst)[0] = p0;
                        if (xa + 1 < a.width) ((uint32_t *)dst)[1] = p1;
                        if (xa + 2 < a.width) ((uint32_t *)dst)[2] = p2;
                        if (xa + 3 < a.width) ((uint32_t *)dst)[3] = p3;
                    }
                }
            } else {
                // image = width rows x n columns; frame x is row width-1-x, bin i is column (i + n/2 - 1) mod n
#pragma unroll 1
                for (int f = 0; f < SF; f++) {
                    const int xa = xs + f;
                    if (xa >= a.width) break;
                    uint8_t *rowp = a.rgba + (size_t)(a.width - 1 - xa) * N * 4;
#pragma unroll 4
                    for (int c4 = lane * 4; c4 < N; c4 += 256) {
                        uint32_t px[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) px[k] = s_lut[tile[((c4 + k + N / 2 + 1) & (N - 1)) * SF + f]];
                        *(uint4 *)(rowp + (size_t)c4 * 4) = make_uint4(px[0], px[1], px[2], px[3]);
                    }
                }
            }
        }
        frame_sync<false>();
    }

    if (cnt_g0) atomicAdd(&s_chist[0], cnt_g0);
    if (cnt_gmax) atomicAdd(&s_chist[cmax], cnt_gmax);
    if (cnt_cb_last) atomicAdd(&s_cbhist[SP_CB_HIST_SIZE - 1], cnt_cb_last);
    if (cnt_cb0) atomicAdd(&s_cbhist[0], cnt_cb0);
    __syncthreads();
    for (int i = tid; i < a.lut_len; i += kWaveThreads)
        if (s_chist[i]) atomicAdd(&a.c_hist[i], (unsigned long long)s_chist[i]);
    for (int i = tid; i < SP_CB_HIST_SIZE; i += kWaveThreads)
        if (s_cbhist[i]) atomicAdd(&a.cb_hist[i], (unsigned long long)s_cbhist[i]);
}

// Host-side launch.
inline int launch_wave(const FrameArgs &a, int format, const double2 *stage_tw, int cu_count, hipStream_t stream)
{
    if (!wave_kernel_supports(a.n) || a.lut_len > kLdsMaxLut || a.lut_len < 2 || !(a.gray_b <= kLdsMaxGrayB)) return SP_ERR_UNSUPPORTED;
    const int n = a.n;
    const WaveLayout lay = wave_layout(n, a.lut_len);
    if (lay.total > 160 * 1024) return SP_ERR_UNSUPPORTED;
    const int sf = wave_strip_frames(n);
    const int strips = (a.width + sf - 1) / sf;
    const int waves = kWaveThreads / 64;
    int grid = (strips + waves - 1) / waves;
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
                           format, stage_tw, strips);                                                                    \
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
    return SP_OK;
}

}  // namespace spk
