// sp_frame_parts.h — building blocks of the frame-loop kernel k_frames (sp_kernel_frames.h) for 64 <= n <= 8192 (gfx950).
//
// Shape of the work (lib/worker.js:68-137 per frame): n samples in, n RGBA pixels out, an n-point DFT in between whose
// butterfly graph and f64 operation order are fixed by bit-exactness with lib/fft_nayuki.js:54-96.  There is no dense
// contraction here, so no MFMA: the kernel is f64-VALU work fed from HBM through registers, with LDS used only to
// re-distribute points between register passes and to transpose the colour indices for coalesced image stores.
//
// Decomposition (what this header implements; the kernel composes it)
//   * 16 points per thread, T = n/16 threads per frame (one wave = one frame at n = 1024).  A 512-thread workgroup
//     runs 8192/n frames at a time.
//   * The radix-2 DIT graph is executed as register passes of <= 4 consecutive stages (fft_pass).  During a pass a thread owns
//     the 16 positions that differ in a 4-bit "window" [ws, ws+4) of the position index; every butterfly of those
//     stages pairs two of its own registers.  Between passes the 16 values go through LDS (exchange: real parts, then
//     imaginary parts, through the same padded buffer) and come back under the next window; n = 512 / 1024 do their second
//     re-distribution in registers (exchange_permlane: v_permlane16_swap / v_permlane32_swap).  The arithmetic of each
//     butterfly is exactly the reference's: 4 multiplies, 2 adds for the twiddle product, 4 adds/subtracts, each
//     rounded on its own (compile with -ffp-contract=off).
//   * Twiddles: per-stage tables (PassTw, load_pass_tw), in LDS for the early stages and in L2 above; the reads of a pass go
//     out as one batch ahead of its re-distribution.
//   * Input: thread tl of a frame owns positions 16*tl + e, i.e. samples rev4(e)*T + rev(tl): every load instruction
//     of a frame covers one contiguous run of T samples (whole cache lines), permuted across lanes.  (Register 1 of thread 0 is
//     sample n/2, the frame's raw centre sample: the loaders hand it out for gauge_amps, worker.js:130-131.)  1-, 2-, 3-, 4-
//     and 8-byte samples are requested one frame ahead into registers (issue_raw, decode_frame); the other formats and frames
//     that leave the buffer take the checked loaders (load_frame).
#pragma once

#include "sp_kernels_common.h"

namespace spk {

constexpr int kLdsMinLog2 = 6, kLdsMaxLog2 = 13;
constexpr int kLdsMaxLut = 256;      // colour indices travel through a byte tile
constexpr float kLdsMaxGrayB = 2000.0f;   // slope bound of the f32 index scales (sp_api.hip: plan_frames_capable)
constexpr int kTilePad = 4;   // tile row pitch = n + 4 bytes: conflict-free dword reads across 8 frame quads

__host__ __device__ inline bool frame_parts_support(int n)
{
    return n >= (1 << kLdsMinLog2) && n <= (1 << kLdsMaxLog2) && (n & (n - 1)) == 0;
}

// Stages 1..kLdsTwMaxStage may keep their twiddle tables in LDS (table of stage s = entries [2^(s-1), 2^s) of stage_tw);
// stages above read HBM/L2.
constexpr int kLdsTwMaxStage = 10;

__host__ __device__ inline constexpr bool lds_win_in_lds(int n) { return n <= 1024; }

// The T threads of a frame fold their extreme |X|^2 into LDS with atomics; lanes are spread over several slots per frame so
// that few lanes of a wave meet on one word (64 on one word cost ~5 % of the kernel).
__host__ __device__ inline constexpr int lds_mm_slots(int n) { return n / 64 < 1 ? 1 : (n / 64 > 16 ? 16 : n / 64); }

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
// FIRST_WAIT = false: the positions this re-distribution writes were last read by the writing wave itself (the one before it was
// wave-private, below), so the real parts' writes need not wait for anybody.
template <int WS_FROM, int WS_TO, bool BLOCK_SYNC, bool SECOND, class SYNC = FrameSyncDefault<BLOCK_SYNC>, bool FIRST_WAIT = true>
__device__ inline void exchange(double (&v)[16], double *from, const double *to, SYNC &&sync = SYNC())   // from / to alias: no __restrict__
{
    if constexpr (SECOND) sync();   // previous readers are done with the buffer
    else if constexpr (FIRST_WAIT) sync.wait();
#pragma unroll
    for (int e = 0; e < 16; e++) from[win_off(e, WS_FROM)] = v[e];
    sync();
    // one ds_read_b64 per value: the compiler pairs them into ds_read2_b64, which the LDS serves at half the rate per byte
    const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) double *)to;
#pragma unroll
    for (int e = 0; e < 16; e++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[e]) : "v"(addr), "n"(win_off(e, WS_TO) * 8));
    // the caller waits for them with exchange_wait() once both components are on their way
    if constexpr (SECOND) sync.arrive();   // (a wave's LDS operations execute in order: the announcement follows the reads)
}

// The compiler does not see the reads of exchange() (they are written as asm): every value passes through
// this wait before it is used.
__device__ inline void exchange_wait(double (&re)[16], double (&im)[16])
{
#define SP_W8(v, o) "+v"(v[o]), "+v"(v[o + 1]), "+v"(v[o + 2]), "+v"(v[o + 3]), "+v"(v[o + 4]), "+v"(v[o + 5]), "+v"(v[o + 6]), "+v"(v[o + 7])
    asm volatile("s_waitcnt lgkmcnt(0)" : SP_W8(re, 0), SP_W8(re, 8)::"memory");
    asm volatile("" : SP_W8(im, 0), SP_W8(im, 8)::"memory");
#undef SP_W8
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
                                  const double (&win)[16], double (&re)[16], double (&im)[16], double2 *centre)
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
        if (centre) *centre = make_double2(vi[1], vq[1]);
    } else {
#pragma unroll 1
        for (int e = 0; e < 16; e++) {
            const int64_t pos = start + rev4(e) * T + sidx;
            // runtime register index: keep this rare path small (it is only taken when frames leave the buffer)
            const double vi = spfmt::sample_checked<FMT>(view, pos, 0);
            const double vq = spfmt::sample_checked<FMT>(view, pos, 1);
            if (e == 1 && centre) *centre = make_double2(vi, vq);
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
// UNIFORM: the frame is wave-uniform (T >= 64) and `start` has been made so (readfirstlane): the frame's base address is SALU work
// and every load is SGPR base + one loop-invariant 32-bit lane offset + an immediate - no address arithmetic in the VALU and no
// per-register base pointers (the compiler's own choice, sixteen uniform bases `base + rev4(e)*T*BYTES`, cost 14 spilled SGPRs and as
// many v_readlane per frame at n = 1024).  Below a wave per frame the addresses are left to the compiler as before.
template <int BYTES, bool UNIFORM = false>
__device__ inline void issue_raw(const uint8_t *__restrict__ base, int64_t start, int T, int sidx, uint32_t (&lo)[16],
                                 uint32_t (&hi)[BYTES == 8 ? 16 : 1], int back = 0)
{
    const uint8_t *fb;     // the frame's first byte (minus `back`), or the lane's first sample
    size_t lane_off = 0;
    if constexpr (UNIFORM) {
        fb = base + start * BYTES - back;
        lane_off = (size_t)(unsigned)(sidx * BYTES);
    } else {
        fb = base - back;
    }
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const uint8_t *p = UNIFORM ? fb + lane_off + (size_t)(rev4(e) * T * BYTES) : fb + (start + rev4(e) * T + sidx) * BYTES;
        if constexpr (BYTES == 3) {
            lo[e] = *(const uint32_t *)p;
        } else if constexpr (BYTES == 1) {
            lo[e] = *p;
        } else if constexpr (BYTES == 2) {
            lo[e] = *(const uint16_t *)p;
        } else if constexpr (BYTES == 4) {
            lo[e] = *(const uint32_t *)p;
        } else {
            // keep the two words of a sample in one 64-bit load result (an even-aligned register pair)
            const unsigned long long w = *(const unsigned long long *)p;
            lo[e] = (uint32_t)w;
            hi[e] = (uint32_t)(w >> 32);
        }
    }
}

// Returns true if the frame may hold infinities or NaNs (float formats; integer formats never do).
template <int FMT, int NHI>
__device__ inline bool decode_frame(const uint32_t (&lo)[16], const uint32_t (&hi)[NHI], const double (&win)[16], double (&re)[16],
                                    double (&im)[16], double2 *centre, int shift = 0)
{
#pragma unroll
    for (int e = 0; e < 16; e++) {
        double vi, vq;
        spfmt::decode_raw<FMT>((FMT == SP_FMT_CU12 || FMT == SP_FMT_CS12) ? lo[e] >> shift : lo[e], hi[NHI == 16 ? e : 0], vi, vq);
        re[e] = win[e] * vi;                                                   // worker.js:73-74
        im[e] = win[e] * vq;
        // thread 0 of the frame: sample n/2, the raw centre sample of gauge_amps (worker.js:130-131), straight to its LDS slot
        if (e == 1 && centre) *centre = make_double2(vi, vq);
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

constexpr int kMaxDevices = 64;

}  // namespace spk
