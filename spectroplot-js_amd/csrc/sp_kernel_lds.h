// sp_kernel_lds.h — the fast frame-loop kernel for 64 <= n <= 8192 (gfx950).
//
// Shape of the work (lib/worker.js:68-137 per frame): n samples in, n RGBA pixels out, an n-point DFT in between whose
// butterfly graph and f64 operation order are fixed by bit-exactness with lib/fft_nayuki.js:54-96.  There is no dense
// contraction here, so no MFMA: the kernel is f64-VALU work fed from HBM through registers, with LDS used only to
// re-distribute points between register passes and to transpose the colour indices for coalesced image stores.
//
// Decomposition
//   * 16 points per thread, T = n/16 threads per frame (one wave = one frame at n = 1024).  A 512-thread workgroup
//     runs 8192/n frames at a time.
//   * The radix-2 DIT graph is executed as register passes of <= 4 consecutive stages.  During a pass a thread owns
//     the 16 positions that differ in a 4-bit "window" [ws, ws+4) of the position index; every butterfly of those
//     stages pairs two of its own registers.  Between passes the 16 values go through LDS (real parts, then
//     imaginary parts, through the same padded buffer) and come back under the next window.  The arithmetic of each
//     butterfly is exactly the reference's: 4 multiplies, 2 adds for the twiddle product, 4 adds/subtracts, each
//     rounded on its own (compile with -ffp-contract=off).
//   * Twiddles: per-stage tables, in LDS for stages <= 10 (lane-uniform broadcast reads in the first pass, consecutive
//     across lanes later) and in L2 above; the reads of a pass go out as one batch ahead of its re-distribution.
//     n = 512 / 1024 do their second re-distribution in registers (v_permlane16_swap / v_permlane32_swap).
//   * Input: thread tl of a frame owns positions 16*tl + e, i.e. samples rev4(e)*T + rev(tl): every load instruction
//     of a frame covers one contiguous run of T samples (whole cache lines), permuted across lanes.  1-, 2-, 3-, 4-
//     and 8-byte samples are requested one frame ahead into registers.
//   * Epilogue: |X|^2 in f64 -> colour index and centi-bel bin by a v_log_f32 first guess corrected against exact
//     edge tables in LDS (no f64 log per pixel) -> LDS histograms (clipped colour indices counted per wave on the
//     scalar unit, rare centi-bel ends on a slow path) -> one byte per pixel into an LDS tile [frame][bin]; frame
//     extremes of |X|^2 by LDS integer atomics on the bit patterns.
//   * After F frames the workgroup writes the tile out through the RGBA LUT as 16-byte stores (128-byte row segments
//     in spectrogram layout, whole rows in waterfall layout), between the decode and the epilogue of the next
//     group's first frame.  Histograms and the launch's extreme |X|^2 go to context accumulators at the end; the
//     finish kernel (sp_kernel_scratch.h) turns per-frame extremes into gauges and moves the accumulators to the reply.
#pragma once

#include "sp_kernels_common.h"

// Compile-time tunables (tools/build_variant.sh builds a library variant with other values; tools/ab_kernel.sh compares
// variants on one GPU in one call).  The defaults are the measured best for config 2 on MI355X.
#ifndef SP_STAGED_TW
#define SP_STAGED_TW 0     // 1: twiddles read stage by stage everywhere (default: one batch per pass, except the generic loaders)
#endif
#ifndef SP_DRAIN_PARTS
#define SP_DRAIN_PARTS 2   // slices of a group's write-out around the passes of the next frame (1..4)
#endif
#ifndef SP_EPI_CHUNK
#define SP_EPI_CHUNK 8     // bins whose edge reads are in flight together (4, 8, 16)
#endif
#ifndef SP_HIST_COPIES
#define SP_HIST_COPIES 1   // LDS histogram copies dealt over lanes (1, 2)
#endif
#ifndef SP_ASM_READS
#define SP_ASM_READS 1     // exchange reads as single ds_read_b64 (the compiler pairs them into half-rate ds_read2_b64)
#endif

namespace spk {

constexpr int kLdsThreads = 512;
constexpr int kLdsMinLog2 = 6, kLdsMaxLog2 = 13;
constexpr int kLdsMaxLut = 256;      // colour indices travel through a byte tile
constexpr float kLdsMaxGrayB = 2000.0f;   // first-guess slope bound that keeps the one-compare correction exact
constexpr int kHistCopies = SP_HIST_COPIES;   // LDS histogram copies (lanes are dealt over them): fewer same-word atomics
constexpr int kTilePad = 4;   // tile row pitch = n + 4 bytes: conflict-free dword reads across 8 frame quads

__host__ __device__ inline bool lds_kernel_supports(int n)
{
    return n >= (1 << kLdsMinLog2) && n <= (1 << kLdsMaxLog2) && (n & (n - 1)) == 0;
}

// frames per output group (tile height)
__host__ __device__ inline int lds_group_frames(int n, int want)
{
    const int fpb = kLdsThreads * 16 / n;   // frames per round
    int f = 32768 / n;
    if (f > want) f = want;
    if (f < fpb) f = fpb;
    if (f < 4) f = 4;
    return f;
}

// Stages 1..kLdsTwMaxStage keep their twiddle tables in LDS (table of stage s = entries [2^(s-1), 2^s) of stage_tw);
// stages 1-4 have lane-uniform twiddles (broadcast LDS reads: keeping them in SGPRs cost ~100 spilled SGPRs and the
// v_readlane traffic that goes with them) and stages above kLdsTwMaxStage read HBM/L2.
constexpr int kLdsTwMaxStage = 10;
__host__ __device__ inline constexpr int lds_tw_entries(int n)
{
    return n < (1 << kLdsTwMaxStage) ? n : (1 << kLdsTwMaxStage);   // entries [0, top): stage s at [2^(s-1), 2^s)
}

__host__ __device__ inline constexpr bool lds_win_in_lds(int n) { return n <= 1024; }

// The T threads of a frame fold their extreme |X|^2 into LDS with atomics; lanes are spread over this many slots per frame so
// that at most 4 lanes of a wave meet on one word (64 on one word cost ~5 % of the kernel).
__host__ __device__ inline constexpr int lds_mm_slots(int n) { return n / 64 < 1 ? 1 : (n / 64 > 16 ? 16 : n / 64); }

struct LdsLayout {
    int xch_doubles;   // exchange buffer (all frames of a round)
    int tile_bytes;
    int off_tile, off_lut, off_chist, off_cbhist, off_gedge, off_cbedge, off_mm, off_tw, off_trash, off_win, total;
};

__host__ __device__ inline LdsLayout lds_layout(int n, int lut_len, int group_frames)
{
    LdsLayout l;
    const int fpb = kLdsThreads * 16 / n;
    l.xch_doubles = fpb * (n + n / 16);
    l.tile_bytes = group_frames * (n + kTilePad);
    int o = l.xch_doubles * 8;
    l.off_tw = o;     o += lds_tw_entries(n) * 16;     // per-stage twiddles of stages 5..10 (16-byte aligned)
    l.off_gedge = o;  o += lut_len * 8;
    l.off_cbedge = o; o += (SP_CB_HIST_SIZE + 1) * 8;
    o = (o + 15) & ~15;
    l.off_mm = o;     o += group_frames * lds_mm_slots(n) * 2 * 8;   // per frame and lane slot {min, max} of |X|^2 (bit patterns)
    l.off_tile = o;   o += (l.tile_bytes + 15) & ~15;
    l.off_lut = o;    o += lut_len * 4;
    l.off_chist = o;  o += lut_len * 4 * kHistCopies;
    l.off_cbhist = o; o += SP_CB_HIST_SIZE * 4 * kHistCopies;
    l.off_trash = o;  o += kLdsThreads * 4;
    o = (o + 7) & ~7;
    l.off_win = o;    o += lds_win_in_lds(n) ? n * 8 : 0;   // taper (n <= 1024: frees 32 VGPRs for the input prefetch)
    l.total = (o + 15) & ~15;
    return l;
}

// floor(x) as int32 in one instruction (callers clamp first; a NaN input does not give 0 on gfx950)
__device__ inline int floor_to_int(float x)
{
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

__device__ inline constexpr int rev4(int e) { return ((e & 1) << 3) | ((e & 2) << 1) | ((e & 4) >> 1) | ((e & 8) >> 3); }

// position owned by (thread tl, register e) under window start ws
__device__ inline int win_pos(int tl, int e, int ws) { return (tl & ((1 << ws) - 1)) | (e << ws) | ((tl >> ws) << (ws + 4)); }
__device__ inline int pad_idx(int p) { return p + (p >> 4); }
// pad_idx(win_pos(tl, e, ws)) == pad_idx(win_pos(tl, 0, ws)) + win_off(e, ws): the per-register part is a compile-time
// constant, so every LDS access of an exchange is one base register plus an immediate offset
__device__ inline constexpr int win_off(int e, int ws) { return (e << ws) + ((e << ws) >> 4); }

// Twiddles of one register pass: stages S0..S1 inside window [WS, WS+4); stage s needs 2^(s-1-WS) of them per thread.
// (MAXS: the last stage whose table is in LDS; k_frames keeps fewer stages there for n >= 4096)
template <int WS, int S0, int S1, int MAXS = kLdsTwMaxStage>
struct PassTw {
    static constexpr int count = (1 << (S1 - WS)) - (1 << (S0 - 1 - WS));
    double2 w[count > 0 ? count : 1];
};

// tw_lds: LDS copy of stage_tw[0 .. min(n, 1024)), tw_glb: the full table in HBM/L2.  Issued as one batch well before the
// pass (ahead of the re-distribution that precedes it), so no butterfly waits on an LDS round trip.
template <int WS, int S0, int S1, int MAXS>
__device__ inline void load_pass_tw(PassTw<WS, S0, S1, MAXS> &t, int tl, const double2 *__restrict__ tw_lds, const double2 *__restrict__ tw_glb)
{
    const int tl_low = tl & ((1 << WS) - 1);
    int k = 0;
#pragma unroll
    for (int s = S0; s <= S1; s++) {
        const int u = (s - 1) - WS;
        const int half = 1 << (s - 1);
#pragma unroll
        for (int j = 0; j < (1 << u); j++) {
            // twiddle index within the stage = position bits below bit s-1     fft_nayuki.js:76-78 (k = j * tablestep)
            const int m = tl_low | (j << WS);
            t.w[k++] = s <= MAXS ? tw_lds[half + m] : tw_glb[half + m];
        }
    }
}

// One register pass: stages S0..S1 (1-based; stage s has size 2^s) inside window [WS, WS+4).
// TRIV: the butterflies of the first pass whose twiddle is entry 0 of its stage, (cos 0, sin 0) = (1, 0) exactly, skip the
// four products: x*1 + y*0 = x bit for bit when x and y are finite (only the sign of a zero can differ, and no output depends
// on it); with an infinite or NaN y the product y*0 is NaN, so the caller enables TRIV only for frames without such samples.
template <int WS, int S0, int S1, bool TRIV = false, int MAXS>
__device__ inline void fft_pass(double (&re)[16], double (&im)[16], const PassTw<WS, S0, S1, MAXS> &t)
{
    int base = 0;
#pragma unroll
    for (int s = S0; s <= S1; s++) {
        const int u = (s - 1) - WS;          // bit of the register index toggled by this stage
#pragma unroll
        for (int e0 = 0; e0 < 16; e0++) {
            if (e0 & (1 << u)) continue;
            const int e1 = e0 | (1 << u);
            const double2 w = t.w[base + (e0 & ((1 << u) - 1))];
            const double c = w.x, sn = w.y;
            const double rl = re[e1], il = im[e1];
#ifdef SP_ABL_NOBF
            asm volatile("" ::"v"(c), "v"(sn));
            continue;
#endif
            const bool unit = TRIV && WS == 0 && (e0 & ((1 << u) - 1)) == 0;   // compile-time after unrolling
            const double tpre = unit ? rl : rl * c + il * sn;          // fft_nayuki.js:80
            const double tpim = unit ? il : il * c - rl * sn;          // fft_nayuki.js:81  (-rl*sn + il*c)
            const double rj = re[e0], ij = im[e0];
            re[e1] = rj - tpre;
            im[e1] = ij - tpim;
            re[e0] = rj + tpre;
            im[e0] = ij + tpim;
        }
        base += 1 << u;
    }
}

// Workgroup barrier for data that travels through LDS only.  __syncthreads() also waits for every outstanding global-memory
// operation of the wave (s_waitcnt vmcnt(0)): here that would be the in-flight input prefetch at every exchange of the large-n
// variants and, after a write-out, the whole HBM store burst.  Nothing in this kernel hands global data from wave to wave.
__device__ inline void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool BLOCK_SYNC>
__device__ inline void frame_sync()
{
    if constexpr (BLOCK_SYNC) {
        lds_barrier();
    } else {
        // The frame lives in one wave and the LDS executes a wave's operations in order, so the hardware needs nothing;
        // only the compiler must not move LDS accesses across this point.  (A wavefront-scope fence is NOT used here:
        // hipcc lowers it to s_waitcnt vmcnt(0), which would stall every exchange on the outstanding HBM loads.)
        asm volatile("" ::: "memory");
    }
}

// How the threads of one frame meet around a re-distribution: frame_sync<BLOCK_SYNC>() as a function object (the default), or
// a kernel's own (k_frames: the waves of a frame meet through an LDS counter).
// A sync has two halves: arrive() announces "everything I issued so far may be relied on", wait() blocks until every partner has
// announced; operator() is both.  Where a wave's reads are followed by a pass of arithmetic before the buffer is written again, the
// exchange announces right after the reads and only waits before the writes (the default object cannot split: its arrive() does nothing
// and its wait() is the whole sync).
template <bool BLOCK_SYNC>
struct FrameSyncDefault {
    __device__ inline void operator()() const { frame_sync<BLOCK_SYNC>(); }
    __device__ inline void arrive() const {}
    __device__ inline void wait() const { frame_sync<BLOCK_SYNC>(); }
};

// Re-distribute the 16 values of every thread from window WS_FROM to window WS_TO through the frame's LDS buffer.
// from / to point at this thread's element 0 under the respective window.
// SECOND: the imaginary parts, which follow the real parts through the same buffer: the sync before their writes comes right after
// the reads of the real parts (nothing to split); the sync before the writes of the real parts was announced after the previous
// re-distribution's last reads.
template <int WS_FROM, int WS_TO, bool BLOCK_SYNC, bool SECOND, class SYNC = FrameSyncDefault<BLOCK_SYNC>>
__device__ inline void exchange(double (&v)[16], double *from, const double *to, SYNC &&sync = SYNC())   // from / to alias: no __restrict__
{
#ifdef SP_ABL_NOEXCH
    return;
#endif
    if constexpr (SECOND) sync();   // previous readers are done with the buffer
    else sync.wait();
#pragma unroll
    for (int e = 0; e < 16; e++) from[win_off(e, WS_FROM)] = v[e];
    sync();
#if SP_ASM_READS
    // one ds_read_b64 per value: the compiler pairs them into ds_read2_b64, which the LDS serves at half the rate per byte
    const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) double *)to;
#pragma unroll
    for (int e = 0; e < 16; e++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[e]) : "v"(addr), "n"(win_off(e, WS_TO) * 8));
    // the caller waits for them with exchange_wait() once both components are on their way
#else
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = to[win_off(e, WS_TO)];
#endif
    if constexpr (SECOND) sync.arrive();   // (a wave's LDS operations execute in order: the announcement follows the reads)
}

// The compiler does not see the reads of exchange() when they are written as asm (SP_X_ASM_READS): every value passes through
// this wait before it is used.
__device__ inline void exchange_wait(double (&re)[16], double (&im)[16])
{
#if SP_ASM_READS
#define SP_W8(v, o) "+v"(v[o]), "+v"(v[o + 1]), "+v"(v[o + 2]), "+v"(v[o + 3]), "+v"(v[o + 4]), "+v"(v[o + 5]), "+v"(v[o + 6]), "+v"(v[o + 7])
    asm volatile("s_waitcnt lgkmcnt(0)" : SP_W8(re, 0), SP_W8(re, 8)::"memory");
    asm volatile("" : SP_W8(im, 0), SP_W8(im, 8)::"memory");
#undef SP_W8
#endif
}

// ---- second re-distribution without LDS (n = 512, 1024; one frame per wave or half-wave) ----------------------------
// After the pass over window [4,8) the stage bits of the last pass sit in the lane index (p8, p9 = lane bits 4, 5) and
// two of the register-index bits (p4, p5) must become lane bits.  That is a transpose between register pairs and the
// 16-lane rows / 32-lane halves of the wavefront, which gfx950 does in the VALU: v_permlane16_swap exchanges the odd rows
// of one register with the even rows of another, v_permlane32_swap the upper half of one with the lower half of another.
__device__ inline void swap_rows16(double &a, double &b)
{
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ inline void swap_halves32(double &a, double &b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}

// window [4,8) -> window [LOG2N-4, LOG2N) for LOG2N = 9, 10, in registers
template <int LOG2N>
__device__ inline void exchange_permlane(double (&v)[16])
{
    static_assert(LOG2N == 9 || LOG2N == 10, "only the 512- and 1024-point layouts put the stage bits in lane bits 4/5");
#pragma unroll
    for (int e = 0; e < 16; e += 2) swap_rows16(v[e], v[e + 1]);              // register bit 0 (p4) <-> lane bit 4 (p8)
    if constexpr (LOG2N == 10) {
#pragma unroll
        for (int e = 0; e < 16; e++)
            if (!(e & 2)) swap_halves32(v[e], v[e | 2]);                       // register bit 1 (p5) <-> lane bit 5 (p9)
    }
    // rename registers to the standard order of the new window: index bits (p[L-4] .. p[L-1])
    double t[16];
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int std_idx = LOG2N == 10 ? (((e >> 2) & 3) | ((e & 3) << 2))    // ours (p8,p9,p6,p7) -> (p6,p7,p8,p9)
                                        : ((e >> 1) | ((e & 1) << 3));          // ours (p8,p5,p6,p7) -> (p5,p6,p7,p8)
        t[std_idx] = v[e];
    }
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = t[e];
}

template <int FMT>
__device__ inline void load_frame(const FrameArgs &a, const spfmt::View &view, int64_t start, int tl, int T, int levels,
                                  const double (&win)[16], double (&re)[16], double (&im)[16])
{
    const int sidx = (int)(__brev((unsigned)tl) >> (32 - (levels - 4)));   // rev_{L-4}(tl)
    if (a.in_bounds) {
        // 16 independent loads issued back to back, decoded afterwards
        double vi[16], vq[16];
#pragma unroll
        for (int e = 0; e < 16; e++) spfmt::sample_fast<FMT>(a.bytes, start + rev4(e) * T + sidx, vi[e], vq[e]);
#pragma unroll
        for (int e = 0; e < 16; e++) {
            re[e] = win[e] * vi[e];                                            // worker.js:73-74
            im[e] = win[e] * vq[e];
        }
    } else {
#pragma unroll 1
        for (int e = 0; e < 16; e++) {
            const int64_t pos = start + rev4(e) * T + sidx;
            // runtime register index: keep this rare path small (it is only taken when frames leave the buffer)
            const double vi = spfmt::sample_checked<FMT>(view, pos, 0);
            const double vq = spfmt::sample_checked<FMT>(view, pos, 1);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k == e) {
                    re[k] = win[k] * vi;
                    im[k] = win[k] * vq;
                }
            }
        }
    }
}

// ---- next-frame input prefetch into registers (2-, 4- and 8-byte samples, frames inside the buffer) --------------------
// The raw words of the NEXT frame are requested right after the current frame has been decoded and are consumed one frame
// later, so HBM latency is covered by a whole frame of butterflies; no other vector-memory load sits inside the frame loop
// at n <= 1024 (twiddles and tables are in LDS), so nothing forces an early wait on them.
// 3-byte samples are fetched as one unaligned dword; `back` (0 or 1) moves the load one byte down for a frame that ends with the
// buffer, whose last dword would otherwise reach one byte past it (the decode shifts such words right by 8).
template <int BYTES>
__device__ inline void issue_raw(const uint8_t *__restrict__ base, int64_t start, int T, int sidx, uint32_t (&lo)[16],
                                 uint32_t (&hi)[BYTES == 8 ? 16 : 1], int back = 0)
{
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const uint8_t *p = base + (start + rev4(e) * T + sidx) * BYTES;
        if constexpr (BYTES == 3) {
            lo[e] = *(const uint32_t *)(p - back);
        } else if constexpr (BYTES == 1) {
            lo[e] = *p;
        } else if constexpr (BYTES == 2) {
            lo[e] = *(const uint16_t *)p;
        } else if constexpr (BYTES == 4) {
            lo[e] = *(const uint32_t *)p;
        } else {
            // keep the two words of a sample in one 64-bit load result (an even-aligned register pair)
#ifdef SP_X_NT_LOAD
            const unsigned long long w = __builtin_nontemporal_load((const unsigned long long *)p);
#else
            const unsigned long long w = *(const unsigned long long *)p;
#endif
            lo[e] = (uint32_t)w;
            hi[e] = (uint32_t)(w >> 32);
        }
    }
}

// Returns true if the frame may hold infinities or NaNs (float formats; integer formats never do).
template <int FMT, int NHI>
__device__ inline bool decode_frame(const uint32_t (&lo)[16], const uint32_t (&hi)[NHI], const double (&win)[16], double (&re)[16],
                                    double (&im)[16], int shift = 0)
{
#pragma unroll
    for (int e = 0; e < 16; e++) {
        double vi, vq;
        spfmt::decode_raw<FMT>((FMT == SP_FMT_CU12 || FMT == SP_FMT_CS12) ? lo[e] >> shift : lo[e], hi[NHI == 16 ? e : 0], vi, vq);
        re[e] = win[e] * vi;                                                   // worker.js:73-74
        im[e] = win[e] * vq;
    }
    // A float capture may hold them, and telling costs as much as the products it would save (32 v_cmp_class per lane-frame
    // plus a second copy of the first pass: measured 28 % slower), so float frames keep the full butterflies.
    return FMT == SP_FMT_CF32 || FMT == SP_FMT_CF64;
}


// A register pass whose twiddles are read stage by stage instead of as one batch ahead of the re-distribution.
template <int WS, int S0, int S1, int MAXS = kLdsTwMaxStage>
__device__ inline void fft_pass_staged(double (&re)[16], double (&im)[16], int tl, const double2 *__restrict__ tw_lds,
                                       const double2 *__restrict__ tw_glb)
{
    if constexpr (S0 <= S1) {
        PassTw<WS, S0, S0, MAXS> t;
        load_pass_tw(t, tl, tw_lds, tw_glb);
        fft_pass<WS, S0, S0>(re, im, t);
        asm volatile("" ::: "memory");   // keep the next stage's reads behind this stage: at most 8 twiddles are live
        fft_pass_staged<WS, S0 + 1, S1, MAXS>(re, im, tl, tw_lds, tw_glb);
    }
}


template <int LOG2N, bool CH, int PFB>   // PFB: bytes per sample of the register-prefetch path (1, 2, 3, 4, 8) or 0 = no prefetch
__global__ __launch_bounds__(kLdsThreads) void k_lds_r16(const FrameArgs a, const int format, const double2 *__restrict__ stage_tw,
                                                         const int group_frames, const int groups)
{
    constexpr int N = 1 << LOG2N;
    constexpr int T = N / 16;                       // threads per frame
    constexpr int FPB = kLdsThreads / T;            // frames per round
    constexpr bool BLOCK_SYNC = T > 64;
    constexpr int NPASS = (LOG2N + 3) / 4;
    // Twiddles of a pass are read as one batch ahead of its re-distribution, except in the generic-loader variants, whose
    // format switch and bounds-checked loads leave no registers for the batch (SP_STAGED_TW: experiment switch).
    constexpr bool STAGED = (SP_STAGED_TW != 0 && LOG2N > 0) || PFB == 0;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LdsLayout lay = lds_layout(N, a.lut_len, group_frames);
    double *s_xch = (double *)smem;
    double2 *s_tw = (double2 *)(smem + lay.off_tw);
    double *s_gedge = (double *)(smem + lay.off_gedge);
    double *s_cbedge = (double *)(smem + lay.off_cbedge);
    unsigned long long *s_mm = (unsigned long long *)(smem + lay.off_mm);
    unsigned char *s_tile = smem + lay.off_tile;
    unsigned int *s_lut = (unsigned int *)(smem + lay.off_lut);
    unsigned int *s_chist = (unsigned int *)(smem + lay.off_chist);
    unsigned int *s_cbhist = (unsigned int *)(smem + lay.off_cbhist);
    unsigned int *const my_chist = s_chist + (threadIdx.x & (kHistCopies - 1)) * a.lut_len;
    unsigned int *const my_cbhist = s_cbhist + (threadIdx.x & (kHistCopies - 1)) * SP_CB_HIST_SIZE;
    unsigned int *const trash = (unsigned int *)(smem + lay.off_trash) + threadIdx.x;

    const int tid = threadIdx.x;
    const int fs = tid / T;                         // frame slot within a round
    const int tl = tid % T;                         // thread within the frame
    double *xbuf = s_xch + fs * (N + N / 16);
    const int tile_pitch = N + kTilePad;
    const int cmax = a.lut_len - 1;

    // groups are dealt so that workgroups sharing an XCD (blockIdx % 8) own neighbouring groups
    const int xcd = blockIdx.x & 7, lane_in_xcd = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int chunk = (groups + 7) >> 3;
    const int g_end = min(groups, (xcd + 1) * chunk);

    // register prefetch is used for the common sample widths when every frame lies inside the buffer
    // (PFB is chosen by the host: frames inside the buffer and 2-, 4- or 8-byte samples)
    constexpr bool PF = PFB != 0;
    const int sidx_pf = (int)(__brev((unsigned)tl) >> (32 - (LOG2N - 4)));
    const int rounds = group_frames / FPB;
    uint32_t raw_lo[PF ? 16 : 1], raw_hi[PFB == 8 ? 16 : 1];
    int raw_back = 0;   // 3-byte samples: 1 if the words now in raw_lo were fetched one byte low (frame ending with the buffer)
    auto request = [&](int xq) {
        if constexpr (PF) {
        const int xc = xq < a.x_end ? xq : a.x_end - 1;
        const int64_t st = frame_start(a.stride, xc);
        if constexpr (PFB == 3) raw_back = (st + N) * 3 + 1 > a.nbytes ? 1 : 0;
        issue_raw<PFB>(a.bytes, st, T, sidx_pf, raw_lo, raw_hi, raw_back);
        }
    };
    // the first frame's samples are requested before the tables: one HBM round trip for both
    if (PF && xcd * chunk + lane_in_xcd < g_end) request(a.frame0 + (xcd * chunk + lane_in_xcd) * group_frames + fs);

    {
        // request tables -> LDS: every global load is issued before the first LDS store, so the prologue costs one memory
        // latency instead of one per table (it is a visible share of a launch at the 16 k-frame size of config 2)
        constexpr int NTW = lds_tw_entries(N);
        constexpr int TWK = (NTW + kLdsThreads - 1) / kLdsThreads;
        double2 tw_r[TWK > 0 ? TWK : 1];
#pragma unroll
        for (int k = 0; k < TWK; k++) {
            const int i = tid + k * kLdsThreads;
            tw_r[k] = i < NTW ? stage_tw[i] : make_double2(0.0, 0.0);
        }
        const double ge_r = tid < a.lut_len ? a.gray_edge[tid] : 0.0;            // lut_len <= 256 < kLdsThreads
        const unsigned int lut_r = tid < a.lut_len ? a.lut_rgba[tid] : 0u;
        double cb_r[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = tid + k * kLdsThreads;
            cb_r[k] = i <= SP_CB_HIST_SIZE ? a.cb_edge[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < TWK; k++) {
            const int i = tid + k * kLdsThreads;
            if (i < NTW) s_tw[i] = tw_r[k];
        }
        if (tid < a.lut_len) {
            s_gedge[tid] = ge_r;
            s_lut[tid] = lut_r;
            for (int c = 0; c < kHistCopies; c++) s_chist[tid + c * a.lut_len] = 0;
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = tid + k * kLdsThreads;
            if (i <= SP_CB_HIST_SIZE) s_cbedge[i] = cb_r[k];
            if (i < SP_CB_HIST_SIZE)
                for (int c = 0; c < kHistCopies; c++) s_cbhist[i + c * SP_CB_HIST_SIZE] = 0;
        }
    }

    // taper: in LDS for n <= 1024 (read per frame), in registers for the whole launch otherwise
    constexpr bool WIN_LDS = lds_win_in_lds(N);
    double *s_win = (double *)(smem + lay.off_win);
    const double *const wbase = s_win + (int)(__brev((unsigned)tl) >> (32 - (LOG2N - 4)));
    double win_reg[WIN_LDS ? 1 : 16];
    if constexpr (WIN_LDS) {
        for (int i = tid; i < N; i += kLdsThreads) s_win[i] = a.window[i];
    } else {
        const int sidx = (int)(__brev((unsigned)tl) >> (32 - (LOG2N - 4)));
#pragma unroll
        for (int e = 0; e < 16; e++) win_reg[e] = a.window[rev4(e) * T + sidx];
    }
    constexpr int MMS = lds_mm_slots(N);
    for (int i = tid; i < group_frames * MMS; i += kLdsThreads) {
        s_mm[2 * i] = 0x7ff0000000000000ull;
        s_mm[2 * i + 1] = 0ull;
    }
    lds_barrier();

    const spfmt::View view{a.bytes, a.nbytes, a.nelem};
    uint32_t pf_word = 0;
    // clipped colour indices and the end bins of the centi-bel histogram are counted in per-wave (scalar) registers
    // (they dominate typical images and would serialise as same-address LDS atomics)
    unsigned int cnt_g0 = 0, cnt_gmax = 0, cnt_cb_last = 0, cnt_cb0 = 0;
    const float gray_a = a.gray_a, gray_b = a.gray_b, cb_a = a.cb_a, cb_b = a.cb_b;
    const float gc_hi = (float)(cmax - 1);

    unsigned long long blk_mn = 0x7ff0000000000000ull, blk_mx = 0ull;   // threads < group_frames: over their frames
    // ---- a finished group: per-frame extremes to HBM, tile -> RGBA (between two workgroup barriers) ---------------------------
    // part / nparts: the write-out can be issued in slices between the passes of the next frame, so that the stores drain
    // into HBM while the SIMDs compute (the extremes go with slice 0)
    auto drain = [&](const int x0, const int part, const int nparts) {
        if (part == 0 && tid < group_frames) {
            if (x0 + tid < a.x_end) {
                unsigned long long bmn = 0x7ff0000000000000ull, bmx = 0ull;
#pragma unroll
                for (int k = 0; k < MMS; k++) {
                    const ulonglong2 v = *(const ulonglong2 *)(s_mm + 2 * (tid * MMS + k));
                    bmn = v.x < bmn ? v.x : bmn;
                    bmx = v.y > bmx ? v.y : bmx;
                }
                a.frame_min[x0 + tid] = __longlong_as_double((long long)bmn);
                a.frame_max[x0 + tid] = __longlong_as_double((long long)bmx);
                blk_mn = bmn < blk_mn ? bmn : blk_mn;
                blk_mx = bmx > blk_mx ? bmx : blk_mx;
            }
#pragma unroll
            for (int k = 0; k < MMS; k++)   // +inf, 0: ready for the next group (barrier below)
                *(ulonglong2 *)(s_mm + 2 * (tid * MMS + k)) = make_ulonglong2(0x7ff0000000000000ull, 0ull);
        }

        // ---- tile -> RGBA -------------------------------------------------------------------------------------
        if (a.rgba) {
            if (!a.waterfall) {
                // spectrogram: image is n rows x width columns; row y holds bin (n/2 - y) mod n            worker.js:90,117
                // a thread owns 4 consecutive bins x 4 consecutive frames; 8 threads cover 32 frames = one 128-byte run
                const int quads = group_frames / 4;                 // frame quads per row
                const int items = (N / 4) * quads;
                // two items per thread and iteration: all tile reads first, then the LUT reads, then the stores, so the
                // LDS latencies of the write-out overlap instead of adding up
                for (int it0 = tid + part * 2 * kLdsThreads; it0 < items; it0 += nparts * 2 * kLdsThreads) {
                    uint32_t gb[2][4];
                    int i0v[2], xav[2];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int it = it0 + u * kLdsThreads;
                        const int itc = it < items ? it : it0;
                        const int fq = itc % quads, bq = itc / quads;
                        i0v[u] = bq * 4;
                        xav[u] = it < items ? x0 + fq * 4 : a.x_end;      // past the image: nothing is stored
#pragma unroll
                        for (int k = 0; k < 4; k++) gb[u][k] = *(const uint32_t *)(s_tile + (fq * 4 + k) * tile_pitch + i0v[u]);
                    }
                    uint32_t px[2][4][4];
#pragma unroll
                    for (int u = 0; u < 2; u++)
#pragma unroll
                        for (int j = 0; j < 4; j++)
#pragma unroll
                            for (int k = 0; k < 4; k++) px[u][j][k] = s_lut[(gb[u][k] >> (8 * j)) & 0xff];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int xa = xav[u];
                        if (xa >= a.x_end) continue;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int i = i0v[u] + j;
                            const int y = (N / 2 - i) & (N - 1);
                            uint8_t *dst = a.rgba + ((size_t)y * (size_t)a.width + (size_t)xa) * 4;
                            if (xa + 3 < a.x_end && (((size_t)dst & 15) == 0)) {
                                *(uint4 *)dst = make_uint4(px[u][j][0], px[u][j][1], px[u][j][2], px[u][j][3]);
                            } else {
#pragma unroll
                                for (int k = 0; k < 4; k++)
                                    if (xa + k < a.x_end) ((uint32_t *)dst)[k] = px[u][j][k];
                            }
                        }
                    }
                }
            } else {
                // waterfall: image is width rows x n columns; frame x is row width-1-x, bin i is column (i + n/2 - 1) mod n
                const int items = group_frames * (N / 4);
                for (int it = tid + part * kLdsThreads; it < items; it += nparts * kLdsThreads) {
                    const int c4 = (it % (N / 4)) * 4, f = it / (N / 4);
                    const int xa = x0 + f;
                    if (xa >= a.x_end) continue;
                    const unsigned char *row = s_tile + f * tile_pitch;
                    uint32_t px[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) px[k] = s_lut[row[(c4 + k + N / 2 + 1) & (N - 1)]];
                    uint8_t *dst = a.rgba + ((size_t)(a.width - 1 - xa) * N + (size_t)c4) * 4;
                    *(uint4 *)dst = make_uint4(px[0], px[1], px[2], px[3]);
                }
            }
        }
    };
    int drain_x0 = -1;   // first frame of the group whose tile is still waiting for its write-out
    for (int g = xcd * chunk + lane_in_xcd; g < g_end; g += per_xcd) {
        const int x0 = a.frame0 + g * group_frames;
        for (int r = 0; r < rounds; r++) {
            const int fr = r * FPB + fs;            // frame within the group
            const int xr = x0 + fr;
            const bool live = xr < a.x_end;
            const int x = live ? xr : a.x_end - 1;  // surplus slots recompute the last frame and discard it
            const int64_t start = frame_start(a.stride, x);

            double re[16], im[16];
            double win[16];
            bool nonfinite = true;   // wave-uniform; only the register-load variants can rule it out
#pragma unroll
            for (int e = 0; e < 16; e++) win[e] = WIN_LDS ? wbase[rev4(e) * T] : win_reg[WIN_LDS ? 0 : e];
            PassTw<0, 1, STAGED ? 0 : 4> tw0;
            // lane-uniform (broadcast reads); in flight during the decode where the registers allow it
            constexpr bool TW0_LATE = PFB != 8;
            if constexpr (!STAGED && !TW0_LATE) load_pass_tw(tw0, tl, s_tw, stage_tw);
            // the frame this slot processes next
            const int xn = (r + 1 < rounds) ? xr + FPB : (g + per_xcd < g_end ? a.frame0 + (g + per_xcd) * group_frames + fs : -1);
            if constexpr (PF) {
                if constexpr (PFB == 1) {
                    if (format == SP_FMT_CU4) nonfinite = decode_frame<SP_FMT_CU4, 1>(raw_lo, raw_hi, win, re, im);
                    else nonfinite = decode_frame<SP_FMT_CS4, 1>(raw_lo, raw_hi, win, re, im);
                } else if constexpr (PFB == 3) {
                    if (format == SP_FMT_CU12) nonfinite = decode_frame<SP_FMT_CU12, 1>(raw_lo, raw_hi, win, re, im, 8 * raw_back);
                    else nonfinite = decode_frame<SP_FMT_CS12, 1>(raw_lo, raw_hi, win, re, im, 8 * raw_back);
                } else if constexpr (PFB == 2) {
                    if (format == SP_FMT_CU8) nonfinite = decode_frame<SP_FMT_CU8, 1>(raw_lo, raw_hi, win, re, im);
                    else nonfinite = decode_frame<SP_FMT_CS8, 1>(raw_lo, raw_hi, win, re, im);
                } else if constexpr (PFB == 4) {
                    if (format == SP_FMT_CU16) nonfinite = decode_frame<SP_FMT_CU16, 1>(raw_lo, raw_hi, win, re, im);
                    else nonfinite = decode_frame<SP_FMT_CS16, 1>(raw_lo, raw_hi, win, re, im);
                } else {
                    if (format == SP_FMT_CU32) nonfinite = decode_frame<SP_FMT_CU32, 16>(raw_lo, raw_hi, win, re, im);
                    else if (format == SP_FMT_CS32) nonfinite = decode_frame<SP_FMT_CS32, 16>(raw_lo, raw_hi, win, re, im);
                    else nonfinite = decode_frame<SP_FMT_CF32, 16>(raw_lo, raw_hi, win, re, im);
                }
                if (xn >= 0) request(xn);           // in flight during this frame's butterflies
            } else {
            // Touch the cache lines of the frame this slot processes next, so that its loads hit L2 instead of HBM.
            asm volatile("" ::"v"(pf_word));   // the previous touch has long landed; this only keeps the load alive
            if (a.in_bounds && xn >= 0 && xn < a.x_end) {
                const int lines = (N * a.sample_width + 127) >> 7;
                const int64_t nb = (int64_t)frame_start(a.stride, xn) * a.sample_width;
                for (int l = tl; l < lines; l += T) pf_word = *(const uint32_t *)(a.bytes + ((nb + (int64_t)l * 128) & ~(int64_t)3));
            }

            switch (format) {
#define SP_CASE(F) case F: load_frame<F>(a, view, start, tl, T, LOG2N, win, re, im); break;
                SP_CASE(SP_FMT_CU4) SP_CASE(SP_FMT_CS4) SP_CASE(SP_FMT_CU8) SP_CASE(SP_FMT_CS8) SP_CASE(SP_FMT_CU12)
                SP_CASE(SP_FMT_CS12) SP_CASE(SP_FMT_CU16) SP_CASE(SP_FMT_CS16) SP_CASE(SP_FMT_CU32) SP_CASE(SP_FMT_CS32)
                SP_CASE(SP_FMT_CU64) SP_CASE(SP_FMT_CS64) SP_CASE(SP_FMT_CF32)
#undef SP_CASE
            default: load_frame<SP_FMT_CF64>(a, view, start, tl, T, LOG2N, win, re, im); break;
            }
            }

            // ---- DFT: register passes with LDS re-distribution in between ---------------------------------
            // Twiddles above stage 10 are re-read every frame (L2 hits, consecutive across lanes); hiding the offset from
            // the optimiser keeps it from hoisting loop-invariant twiddle registers out of the frame loop.
            unsigned tw_off = 0;
            asm volatile("" : "+s"(tw_off));
            const double2 *tw = stage_tw + tw_off;   // still a global-memory pointer for the compiler (no flat loads)
            if (SP_DRAIN_PARTS > 1 && drain_x0 >= 0) {   // every wave has finished the previous group: first slice of its write-out
                lds_barrier();
                drain(drain_x0, 0, SP_DRAIN_PARTS);
            }
            if constexpr (STAGED) {
                fft_pass_staged<0, 1, 4>(re, im, tl, s_tw, stage_tw);
            } else {
                if constexpr (TW0_LATE) load_pass_tw(tw0, tl, s_tw, stage_tw);
                // (not in the 8-byte variants: cf32 needs the full butterflies, and a second copy of the pass for cu32 / cs32
                // costs the cf32 path its registers)
                if (PFB != 8 && !nonfinite) fft_pass<0, 1, 4, true>(re, im, tw0);   // 15 of the 32 butterflies without their products
                else
                fft_pass<0, 1, 4>(re, im, tw0);
            }
            if (SP_DRAIN_PARTS >= 3 && drain_x0 >= 0) drain(drain_x0, 1, SP_DRAIN_PARTS);
            if constexpr (NPASS >= 2) {
                constexpr int WS1 = LOG2N >= 8 ? 4 : LOG2N - 4;
                constexpr int E1 = LOG2N >= 8 ? 8 : LOG2N;
                double *const b0 = xbuf + pad_idx(win_pos(tl, 0, 0)), *const b1 = xbuf + pad_idx(win_pos(tl, 0, WS1));
                PassTw<WS1, 5, STAGED ? 4 : E1> tw1;
                if constexpr (!STAGED) load_pass_tw(tw1, tl, s_tw, tw);
                exchange<0, WS1, BLOCK_SYNC, false>(re, b0, b1);
                exchange<0, WS1, BLOCK_SYNC, true>(im, b0, b1);
                exchange_wait(re, im);
                if constexpr (STAGED) fft_pass_staged<WS1, 5, E1>(re, im, tl, s_tw, tw);
                else fft_pass<WS1, 5, E1>(re, im, tw1);
                if (SP_DRAIN_PARTS >= 4 && drain_x0 >= 0) drain(drain_x0, 2, SP_DRAIN_PARTS);
                if constexpr (NPASS >= 3) {
                    constexpr int WS2 = LOG2N >= 12 ? 8 : LOG2N - 4;
                    constexpr int E2 = LOG2N >= 12 ? 12 : LOG2N;
                    double *const b2 = xbuf + pad_idx(win_pos(tl, 0, WS2));
                    PassTw<WS2, 9, STAGED ? 8 : E2> tw2;
                    if constexpr (!STAGED) load_pass_tw(tw2, tl, s_tw, tw);
                    if constexpr (LOG2N == 9 || LOG2N == 10) {
                        exchange_permlane<LOG2N>(re);      // no LDS: v_permlane16_swap / v_permlane32_swap
                        exchange_permlane<LOG2N>(im);
                    } else {
                        exchange<WS1, WS2, BLOCK_SYNC, false>(re, b1, b2);
                        exchange<WS1, WS2, BLOCK_SYNC, true>(im, b1, b2);
                exchange_wait(re, im);
                    }
                    if constexpr (STAGED) fft_pass_staged<WS2, 9, E2>(re, im, tl, s_tw, tw);
                    else fft_pass<WS2, 9, E2>(re, im, tw2);
                    if constexpr (NPASS >= 4) {
                        constexpr int WS3 = LOG2N - 4;
                        double *const b3 = xbuf + pad_idx(win_pos(tl, 0, WS3));
                        PassTw<WS3, 13, LOG2N> tw3;
                        load_pass_tw(tw3, tl, s_tw, tw);
                        exchange<WS2, WS3, BLOCK_SYNC, false>(re, b2, b3);
                        exchange<WS2, WS3, BLOCK_SYNC, true>(im, b2, b3);
                exchange_wait(re, im);
                        fft_pass<WS3, 13, LOG2N>(re, im, tw3);
                    }
                }
            }
            // now register e of thread tl holds bin i = tl + e*T

            if constexpr (CH) {   // fft_nayuki.js:103-119, partner bin n-i fetched through LDS
                // One component at a time, so only 16 partner values are live: the real parts give 0.5*(re[i] + re[n-i]) below
                // n/2 and 0.5*(-re[n-i] + re[i]) above it, the imaginary parts 0.5*(im[i] - im[n-i]) and 0.5*(im[n-i] + im[i]).
                // Above n/2 the two results land in swapped registers (the reference's real part sits in im[e]); only
                // re*re + im*im is taken from here on, and that sum does not depend on the order.
                double pp[16];
                frame_sync<BLOCK_SYNC>();
#pragma unroll
                for (int e = 0; e < 16; e++) xbuf[pad_idx(tl + e * T)] = re[e];
                frame_sync<BLOCK_SYNC>();
#pragma unroll
                for (int e = 0; e < 16; e++) pp[e] = xbuf[pad_idx((N - (tl + e * T)) & (N - 1))];
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int i = tl + e * T;
                    const double orr = re[e];
                    if (i == 0) {
                        // bin 0 keeps its real part
                    } else if (i == N / 2) {
                        re[e] = 0.0;
                    } else if (i < N / 2) {
                        re[e] = 0.5 * (orr + pp[e]);
                    } else {
                        re[e] = 0.5 * (-pp[e] + orr);
                    }
                }
                frame_sync<BLOCK_SYNC>();
#pragma unroll
                for (int e = 0; e < 16; e++) xbuf[pad_idx(tl + e * T)] = im[e];
                frame_sync<BLOCK_SYNC>();
#pragma unroll
                for (int e = 0; e < 16; e++) pp[e] = xbuf[pad_idx((N - (tl + e * T)) & (N - 1))];
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int i = tl + e * T;
                    const double oi = im[e];
                    if (i == 0 || i == N / 2) {
                        im[e] = 0.0;
                    } else if (i < N / 2) {
                        im[e] = 0.5 * (oi - pp[e]);
                    } else {
                        im[e] = 0.5 * (pp[e] + oi);
                    }
                }
            }

            if (drain_x0 >= 0) {   // the previous group's tile: last slice, then the tile is free again
                if (SP_DRAIN_PARTS == 1) lds_barrier();
                drain(drain_x0, SP_DRAIN_PARTS - 1, SP_DRAIN_PARTS);
                lds_barrier();
                drain_x0 = -1;
            }
            // ---- |X|^2 -> indices ---------------------------------------------------------------------------
            // The first guess floor(a + b*log2(abs2)) is biased half a step low, so the exact index is the guess or
            // the guess + 1; one comparison against the exact edge decides (edges: sp_host.h Thresholds).
            double mn = spjs::inf(), mx = 0.0;
            unsigned char *trow = s_tile + fr * tile_pitch;
            // Surplus slots (recomputed last frame) skip the whole epilogue: one branch per frame instead of a mask per bin.
            // Two halves of 8 bins keep the epilogue's temporaries (|X|^2, guesses, edges) at 64 VGPRs instead of 128.
            // s_cbhist is indexed by level (= 999 - bin): the flush reverses it.
            if (live) {
            // all edge reads of a chunk (the whole frame by default) are in flight before the first comparison needs one
            constexpr int EC = SP_EPI_CHUNK;
            constexpr int SC = EC < 8 ? EC : 8;   // bins per classification batch (one rare-path test each)
#pragma unroll
            for (int c = 0; c < 16 / EC; c++) {
            double abs2_all[EC], ge_all[EC], ce_all[EC];
            int gc_all[EC], lc_all[EC];
#pragma unroll
            for (int ee = 0; ee < EC; ee++) {
                const int e = c * EC + ee;
                abs2_all[ee] = re[e] * re[e] + im[e] * im[e];                      // worker.js:92
                mn = min_nn(mn, abs2_all[ee]);
                mx = max_nn(mx, abs2_all[ee]);
                // f32 range is enough: the host only selects this kernel when every edge lies in [2^-100, 2^100], so a
                // |X|^2 that under- or overflows f32 is clipped either way (sp_api.hip plan_lds_capable)
                const float l2 = __log2f((float)abs2_all[ee]);
                // v_med3_f32 clamps and turns a NaN guess into 0 (NaN abs2 must end at index 0: every comparison is false)
                gc_all[ee] = floor_to_int(__builtin_amdgcn_fmed3f(fmaf(gray_b, l2, gray_a), 0.0f, gc_hi));
                lc_all[ee] = floor_to_int(__builtin_amdgcn_fmed3f(fmaf(cb_b, l2, cb_a), 0.0f, (float)(SP_CB_HIST_SIZE - 1)));
                ge_all[ee] = s_gedge[gc_all[ee] + 1];
                ce_all[ee] = s_cbedge[lc_all[ee] + 1];
            }
#pragma unroll
            for (int h = 0; h < EC / SC; h++) {
                const double *abs2 = abs2_all + h * SC, *ge = ge_all + h * SC, *ce = ce_all + h * SC;
                const int *gc = gc_all + h * SC, *lc = lc_all + h * SC;
                // colour index: clipped values (typical images are full of them) are counted per wave with s_bcnt1 on the
                // compare masks instead of hammering one LDS word; their atomic goes to a per-lane trash word
                int lvm1[SC];                // centi-bel level - 1
                unsigned int lv_span = 0;    // max over the half of (level - 1) as unsigned: >= 999 iff a level is 0 or 1000
#pragma unroll
                for (int k = 0; k < SC; k++) {
                    const int e = c * EC + h * SC + k;
                    const int gr = gc[k] + (abs2[k] >= ge[k] ? 1 : 0);
                    lvm1[k] = lc[k] - 1 + (abs2[k] >= ce[k] ? 1 : 0);
                    lv_span = max(lv_span, (unsigned int)lvm1[k]);
                    trow[tl + e * T] = (unsigned char)gr;
                    const bool g0 = gr == 0, gm = gr == cmax;
                    cnt_g0 += (unsigned int)__popcll(__ballot(g0));
                    cnt_gmax += (unsigned int)__popcll(__ballot(gm));
                    atomicAdd((g0 || gm) ? trash : &my_chist[gr], 1u);
                    // one bin at a time: interleaving the eight bins keeps the compare masks alive and spills SGPRs
                    __builtin_amdgcn_sched_barrier(0);
                }
                // centi-bel histogram: levels 1..999 are plain LDS atomics; the ends (below -100 dBfs, above 0 dBfs, and the
                // -inf / +inf / NaN keys of worker.js:105) are rare and take a wave-uniform slow path for the whole half
                if (__builtin_expect(__ballot(lv_span >= (unsigned int)(SP_CB_HIST_SIZE - 1)) == 0ull, 1)) {
#pragma unroll
                    for (int k = 0; k < SC; k++) atomicAdd(&my_cbhist[1 + lvm1[k]], 1u);
                } else {
#pragma unroll 1
                    for (int k = 0; k < SC; k++) {
                        double a2 = abs2[0];
#pragma unroll
                        for (int j = 1; j < SC; j++) a2 = k == j ? abs2[j] : a2;
                        int lv = lvm1[0];
#pragma unroll
                        for (int j = 1; j < SC; j++) lv = k == j ? lvm1[j] : lv;
                        lv += 1;
                        // -inf / +inf / NaN dB: ToInt32 gives 0, i.e. bin 0        worker.js:105
                        const bool special = !(a2 > 0.0) || a2 == spjs::inf();
                        const bool l0 = lv == 0, lx = lv == SP_CB_HIST_SIZE;
                        cnt_cb0 += (unsigned int)__popcll(__ballot(special));
                        cnt_cb_last += (unsigned int)__popcll(__ballot(!special && l0));
                        atomicAdd((special || l0 || lx) ? trash : &my_cbhist[lv], 1u);
                    }
                }
            }
            }
            }
            // frame min / max over its T threads: |X|^2 >= +0 and never NaN here, so the order of the doubles is the order
            // of their bit patterns and two fire-and-forget LDS atomics replace a six-step cross-lane reduction
            if (live) {
                unsigned long long *slot = s_mm + 2 * (fr * MMS + (tl & (MMS - 1)));
                atomicMin(slot, (unsigned long long)__double_as_longlong(mn));
                atomicMax(slot + 1, (unsigned long long)__double_as_longlong(mx));
            }
        }
        drain_x0 = x0;   // drained ahead of the next epilogue (or after the loop): early waves start their next frame first
    }

    // ---- end of the workgroup's frames: histograms to the context accumulators, last write-out -------------------------
    if ((tid & 63) == 0) {                               // per-wave counters of the clipped / end bins
        if (cnt_g0) atomicAdd(&s_chist[0], cnt_g0);
        if (cnt_gmax) atomicAdd(&s_chist[cmax], cnt_gmax);
        if (cnt_cb_last) atomicAdd(&s_cbhist[0], cnt_cb_last);                 // level 0 = bin 999
        if (cnt_cb0) atomicAdd(&s_cbhist[SP_CB_HIST_SIZE - 1], cnt_cb0);       // specials = bin 0
    }
    lds_barrier();   // LDS histograms and the last tile are complete
    // the device atomics are issued ahead of the last write-out, so they complete under its HBM burst
    for (int i = tid; i < a.lut_len; i += kLdsThreads) {
        unsigned int v = 0;
        for (int c = 0; c < kHistCopies; c++) v += s_chist[i + c * a.lut_len];
        if (v) atomicAdd(&a.c_hist[i], (unsigned long long)v);
    }
    for (int i = tid; i < SP_CB_HIST_SIZE; i += kLdsThreads) {
        unsigned int v = 0;
        for (int c = 0; c < kHistCopies; c++) v += s_cbhist[i + c * SP_CB_HIST_SIZE];
        if (v) atomicAdd(&a.cb_hist[SP_CB_HIST_SIZE - 1 - i], (unsigned long long)v);
    }
    if (drain_x0 >= 0) {   // last group
        drain(drain_x0, 0, 1);
        lds_barrier();
    }
    if (tid < group_frames) {                            // extreme |X|^2 of this workgroup's frames: s_mm[0..1] are {+inf, 0} again
        if (blk_mn != 0x7ff0000000000000ull) atomicMin(&s_mm[0], blk_mn);
        if (blk_mx != 0ull) atomicMax(&s_mm[1], blk_mx);
    }
    lds_barrier();
    if (tid == 0) {                                      // one pair of device atomics per workgroup (dBfs range, k_finish_frames)
        if (s_mm[0] != 0x7ff0000000000000ull) atomicMin(&a.mm_acc[0], s_mm[0]);
        if (s_mm[1] != 0ull) atomicMax(&a.mm_acc[1], s_mm[1]);
    }
}

// Per-n launchers: each lives in its own translation unit (sp_inst_lds.hip compiled once per LOG2N) so that the 96 variants build
// in parallel.  `device` indexes the per-device record of the LDS opt-in (function attributes belong to a device's code object).
constexpr int kMaxDevices = 64;
template <int L>
int launch_lds_n(const FrameArgs &a, int format, const double2 *stage_tw, int grid, int lds_bytes, int gf, int groups, int prefetch, int device,
                 hipStream_t stream);
#define SP_DECL(L)                                                                                                              \
    template <>                                                                                                                 \
    int launch_lds_n<L>(const FrameArgs &, int, const double2 *, int, int, int, int, int, int, hipStream_t);
SP_DECL(6) SP_DECL(7) SP_DECL(8) SP_DECL(9) SP_DECL(10) SP_DECL(11) SP_DECL(12) SP_DECL(13)
#undef SP_DECL

#ifdef SP_INST_LDS_LOG2N
template <int L, bool C, int P>
inline int launch_lds_variant(const FrameArgs &a, int format, const double2 *stage_tw, int grid, int lds_bytes, int gf, int groups, int device,
                              hipStream_t stream)
{
    static bool attr_set[kMaxDevices] = {};
    if (device < 0 || device >= kMaxDevices || !attr_set[device]) {
        if (hipFuncSetAttribute((const void *)k_lds_r16<L, C, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return SP_ERR_HIP;
        if (device >= 0 && device < kMaxDevices) attr_set[device] = true;
    }
    hipLaunchKernelGGL((k_lds_r16<L, C, P>), dim3((unsigned)grid), dim3(kLdsThreads), (size_t)lds_bytes, stream, a, format, stage_tw, gf, groups);
    return SP_OK;
}

template <>
int launch_lds_n<SP_INST_LDS_LOG2N>(const FrameArgs &a, int format, const double2 *stage_tw, int grid, int lds_bytes, int gf, int groups, int prefetch,
                                int device, hipStream_t stream)
{
    constexpr int L = SP_INST_LDS_LOG2N;
#define SP_V(C, P) return launch_lds_variant<L, C, P>(a, format, stage_tw, grid, lds_bytes, gf, groups, device, stream);
#define SP_CH(C)                                                                                              \
    switch (prefetch) {                                                                                       \
    case 8: SP_V(C, 8) case 4: SP_V(C, 4) case 3: SP_V(C, 3) case 2: SP_V(C, 2) case 1: SP_V(C, 1) default: SP_V(C, 0) \
    }
    if (a.channel_mode) { SP_CH(true) } else { SP_CH(false) }
#undef SP_V
#undef SP_CH
}
#endif

// Host-side launch.  Returns SP_OK or SP_ERR_UNSUPPORTED.
inline int launch_lds(const FrameArgs &a, int format, const double2 *stage_tw, int cu_count, int device, hipStream_t stream)
{
    if (!lds_kernel_supports(a.n) || a.lut_len > kLdsMaxLut || a.lut_len < 2 || !(a.gray_b <= kLdsMaxGrayB)) return SP_ERR_UNSUPPORTED;
    const int n = a.n;
    // tile height: 32 frames give 128-byte row segments; small images use shorter groups so every CU gets work
    int want = 32;
#ifdef SP_EXPERIMENT_KNOBS
    static const int cu_env = getenv("SP_CU_LIMIT") ? atoi(getenv("SP_CU_LIMIT")) : 0;   // measurements only (tools/overhead.py)
    if (cu_env > 0 && cu_env < cu_count) cu_count = cu_env;
#endif
    while (want > 4 && (a.x_end - a.frame0 + want - 1) / want < 2 * cu_count) want >>= 1;
    const int gf = lds_group_frames(n, want);
    const int groups = (a.x_end - a.frame0 + gf - 1) / gf;
    const LdsLayout lay = lds_layout(n, a.lut_len, gf);
    int grid = groups < cu_count ? groups : cu_count;
    grid = (grid + 7) & ~7;
    if (lay.total > 160 * 1024) return SP_ERR_UNSUPPORTED;
    // next-frame register prefetch: frames inside the buffer, 1-, 2-, 3-, 4- or 8-byte samples
    int prefetch = (a.in_bounds && (a.sample_width <= 4 || a.sample_width == 8)) ? a.sample_width : 0;
    // 3-byte samples are fetched as unaligned dwords, one byte low for a frame that ends with the buffer: that frame must not
    // start at byte 0
    if (prefetch == 3 && !(a.width >= 2 && frame_start(a.stride, a.width - 1) >= 1)) prefetch = 0;
    switch (a.levels) {
#define SP_L(L) case L: return launch_lds_n<L>(a, format, stage_tw, grid, lay.total, gf, groups, prefetch, device, stream);
        SP_L(6) SP_L(7) SP_L(8) SP_L(9) SP_L(10) SP_L(11) SP_L(12) SP_L(13)
#undef SP_L
    default: return SP_ERR_UNSUPPORTED;
    }
}

}  // namespace spk
