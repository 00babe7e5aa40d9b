// sp_kernel_frames.h — the frame-loop kernel for 64 <= n <= 8192 (gfx950).
//
// lib/worker.js:68-137 per frame (decode, taper, the butterfly graph of lib/fft_nayuki.js:54-96, |X|^2 -> dB -> indices -> RGBA, side
// outputs), built from the parts of sp_frame_parts.h around what bounds it on MI355X.  tools/op_cost.hip (profiles/r02_op_cost.txt): an
// f64 multiply or add costs 4.4-4.7 issue cycles per wave-instruction per SIMD whatever the occupancy, v_permlane32_swap 8, v_log_f32
// 8.5, conversions / floor / fract / compares 4.4, 32-bit integer and f32 multiply-add 2.5; one wave alone reaches half of that, two
// waves reach it.  The reference's unfused butterflies are 800 f64 wave-instructions per 1024-point frame, so the loop is bound by
// VALU issue, not by HBM, LDS or latency, and the design moves work off the VALU or removes it:
//   * first pass (stages 1-4): its eight twiddles cos / sin(2 pi k / 16) are the same doubles for every n >= 16 (the table
//     index k*n/16 is scaled by a power of two before the division by n), so they are literals: no LDS reads, no registers;
//     the butterflies whose twiddle is (1, 0) skip their products when the frame is finite (integer formats always; float
//     frames after one f32 multiply-add per raw word), the ones whose sine is exactly 1 skip two products always;
//   * the re-distribution between passes goes through a padded, wave-private LDS buffer; at n = 512 / 1024 the second one is a
//     register transpose (v_permlane16_swap / v_permlane32_swap: what it costs the VALU the LDS round trip costs the LDS pipe);
//   * input: the raw words of the NEXT frame are requested right after the current frame is decoded (1-, 2-, 3-, 4-, 8-byte samples).
//     One rule follows from the chip completing a wave's vector-memory operations IN ORDER: whatever waits for a younger operation - a
//     reload of a spilled register, a table load - waits for that prefetch too.  So no prefetching variant may spill inside the loop
//     (tests/test_isa_checks.py); the variants that would (8-byte samples at n >= 2048; the L/R split at n >= 2048 and with 8-byte samples
//     from n = 512) request a frame's samples when it starts; and the prologue requests the first frame BEHIND its table loads (n <= 1024);
//   * epilogue: colour index and centi-bel level are floor(a + b*log2(|X|^2)) in f32; a lane is sent to the exact edge tables
//     only if its f32 value lies within a proven error margin of an integer (a few lanes in ten thousand), so the common path has
//     no LDS read and no f64 compare; one histogram atomic per pixel on the merged cell (colour index + level), which the
//     workgroup turns back into the two histograms at its end; one colour byte per pixel into an LDS tile [frame][bin]; frame
//     extremes of |X|^2 by LDS integer atomics on the bit patterns;
//   * after a group of frames the workgroup writes the tile out through the RGBA LUT, in two slices around the passes of the next
//     group's first frame: 16-byte stores, 128-byte row segments in spectrogram layout; in waterfall layout one tile dword per item as
//     four dword stores, 256 contiguous bytes of an image row per wave and store instruction;
//   * n >= 2048 (a frame spans several waves): the waves of a frame meet through an LDS counter, announced early and waited for
//     late where the dataflow allows, instead of the workgroup barrier;
//   * the workgroup's last write-out is split between the first and the second waves of the SIMDs (n = 1024): the first ones
//     finish ~7 us earlier and write their half meanwhile;
//   * the request's side outputs come from this kernel too - one launch per sp_plan_execute: the gauges of a group of frames are
//     evaluated (the reference's software log10, three per frame) by two waves inside the next group, where the first waves of the
//     SIMDs have slack; at its end every workgroup adds its own share of the two histograms and of the dBfs range to the reply with
//     fire-and-forget atomics (workgroup 0 has cleared the reply and published the request's number), so no workgroup waits for
//     another and nothing makes a dependent trip to memory (a last-workgroup ticket costs three: profiles/r04_experiments.txt).
// Measured alternatives (three waves per SIMD, two workgroups per CU, LDS-DMA input, other batch / slice / chain counts) are recorded in
// DESIGN.md section 6.2; the cost-attribution switches and per-wave clock stamps that produced profiles/ live in
// tools/experiments/frames_instrumentation.patch (tools/build_variant.sh applies it), not here.
#pragma once

#include <atomic>
#include <cstdio>

#include "sp_frame_parts.h"

namespace spk2 {

using namespace spk;

// cos / sin(2 pi k / 16), k = 0..7, as sphost::twiddles produces them for every n >= 16 (sp_api.hip checks it per plan).
struct Tw16 {
    double c, s;
};
__device__ constexpr Tw16 kTw16[8] = {
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.d906bcf328d46p-1, 0x1.87de2a6aea963p-2},
    {0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bccp-1},
    {0x1.87de2a6aea964p-2, 0x1.d906bcf328d46p-1},
    {0x1.1a62633145c07p-54, 0x1.0000000000000p+0},
    {-0x1.87de2a6aea962p-2, 0x1.d906bcf328d46p-1},
    {-0x1.6a09e667f3bccp-1, 0x1.6a09e667f3bcdp-1},
    {-0x1.d906bcf328d46p-1, 0x1.87de2a6aea965p-2},
};
inline constexpr Tw16 kTw16Host[8] = {
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.d906bcf328d46p-1, 0x1.87de2a6aea963p-2},
    {0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bccp-1},
    {0x1.87de2a6aea964p-2, 0x1.d906bcf328d46p-1},
    {0x1.1a62633145c07p-54, 0x1.0000000000000p+0},
    {-0x1.87de2a6aea962p-2, 0x1.d906bcf328d46p-1},
    {-0x1.6a09e667f3bccp-1, 0x1.6a09e667f3bcdp-1},
    {-0x1.d906bcf328d46p-1, 0x1.87de2a6aea965p-2},
};

// a group's row pieces are written with non-temporal stores from this many frames per group on (4 bytes per frame and row)
#ifndef SP_NT_MIN_GROUP
#define SP_NT_MIN_GROUP 32
#endif

constexpr int kMmSlotsMax = 4;
constexpr int kMaxCells = kLdsMaxLut + SP_CB_HIST_SIZE + 2;   // merged histogram cells (sp_host.h Thresholds)

__host__ __device__ inline constexpr int mm_slots(int n)
{
    return lds_mm_slots(n) < kMmSlotsMax ? lds_mm_slots(n) : kMmSlotsMax;
}

constexpr int kFrameThreads = 512;   // one workgroup per CU: eight waves, two per SIMD

// 16 points per thread, whole frames per workgroup: 64 <= n <= 8192
__host__ __device__ inline bool frames_kernel_supports(int n) { return frame_parts_support(n); }

// n >= 2048: the twiddle tables of stages 1-9 only stay in LDS (stage 10 joins the later ones in L2: two more loads per thread and
// frame), which makes room for a 64 KiB tile: 32 / 16 / 8 frames per group instead of 16 / 8 / 4, i.e. 128 / 64 / 32-byte pieces of
// the image rows and half as many group barriers (config 5 wrote 2.1 x its image with 16-byte pieces; config 3: -0.7 %, cf32 at
// n = 2048: -5 %, config 5: -5 %)
__host__ __device__ inline constexpr int frames_tw_max_stage(int n) { return n >= 2048 ? 9 : kLdsTwMaxStage; }
__host__ __device__ inline constexpr int frames_tw_entries(int n) { return n < (1 << frames_tw_max_stage(n)) ? n : (1 << frames_tw_max_stage(n)); }

// frames per output group (tile height): a multiple of the frames per round and of 4 (the write-out handles frame quads)
__host__ __device__ inline int group_frames_for(int n, int want)
{
    const int fpb = kFrameThreads * 16 / n;
    int unit = fpb;
    while (unit % 4) unit *= 2;          // lcm(fpb, 4) for fpb in {1, 2, 3, 6, 12, ...}
    int cap = (n >= 2048 ? 65536 : 32768) / n;
    if (cap > want) cap = want;
    int f = cap / unit * unit;
    if (f < unit) f = unit;
    return f;
}

// n <= 1024: a group's side outputs are evaluated behind the SECOND barrier of the following group's first frame, which takes a second
// set of frame-extreme slots (n >= 2048 has no LDS left for one: they are evaluated behind the first barrier there)
__host__ __device__ inline constexpr bool late_side_outputs(int n) { return n <= 1024; }

// The RGBA LUT and the merged histogram cells sit at FIXED LDS addresses in front of everything whose size depends on the request, so
// that a pixel's LUT read and its histogram atomic are `ds_* vaddr offset:imm` with vaddr = index * 4 alone: one VALU instruction per
// pixel for either address (v_lshlrev_b32_sdwa of a tile byte; v_add_lshl_u32 of colour index + level) instead of two.
constexpr int kOffLut = 0;                                        // u32[kLdsMaxLut]
constexpr int kOffCells = kOffLut + kLdsMaxLut * 4;               // u32[kMaxCells]: word c counts the pixels with colour index + level == c
constexpr int kOffXch = (kOffCells + kMaxCells * 4 + 15) & ~15;   // exchange buffers, then the rest of Layout

struct Layout {
    int off_tw, off_gedge, off_cbedge, off_mm, off_tile, off_done, off_amp, off_win, total;
};

__host__ __device__ inline Layout layout(int n, int lut_len, int group_frames)
{
    Layout l;
    const int fpb = kFrameThreads * 16 / n;
    int o = kOffXch + fpb * (n + n / 16) * 8;                    // exchange buffers
    l.off_tw = o;     o += frames_tw_entries(n) * 16;
    l.off_gedge = o;  o += lut_len * 8;                          // exact edge tables (read by the few lanes the f32 test sends there)
    l.off_cbedge = o; o += (SP_CB_HIST_SIZE + 1) * 8;
    o = (o + 15) & ~15;
    l.off_mm = o;     o += (late_side_outputs(n) ? 2 : 1) * group_frames * mm_slots(n) * 2 * 8;   // frame extremes (n <= 1024: by group parity)
    l.off_tile = o;   o += (group_frames * (n + kTilePad) + 15) & ~15;
    l.off_done = o;   o += 32;                                   // arrival counters: the two wave sets (last write-out), the frames' waves; [7]: "last workgroup"
    o = (o + 15) & ~15;
    l.off_amp = o;    o += 16 + 2 * group_frames * 16;           // the workgroup's extreme |X|^2 so far; raw centre samples of two groups' frames
    l.off_win = o;    o += lds_win_in_lds(n) ? n * 8 : 0;
    l.total = (o + 15) & ~15;
    return l;
}

// v_min_f64 / v_max_f64 as single instructions: fmin() / fmax() first quiet a possible signalling NaN in each operand with a
// v_max_f64 x, x, x of its own, three instructions per call.  The hardware minimum / maximum already ignores a (quiet) NaN
// operand, which is all the frame extremes need (worker.js:102-103: comparisons with NaN are false).
__device__ inline double min_raw(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ inline double max_raw(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Fire-and-forget LDS minimum / maximum of doubles (no NaN operands here).  As instructions: the compiler's atomic optimiser turns
// an atomic with a wave-uniform address into a loop over the active lanes, ~10 instructions per lane; the LDS serialises the lanes itself.
__device__ inline void lds_min_f64(double *p, double v)
{
    asm volatile("ds_min_f64 %0, %1" ::"v"((unsigned)(size_t)(__attribute__((address_space(3))) double *)p), "v"(v) : "memory");
}
__device__ inline void lds_max_f64(double *p, double v)
{
    asm volatile("ds_max_f64 %0, %1" ::"v"((unsigned)(size_t)(__attribute__((address_space(3))) double *)p), "v"(v) : "memory");
}

// Inclusive prefix sum over the 64 lanes of a wave in the VALU: row shifts, then the two row broadcasts of the GFX9 family (a shuffle
// scan is six dependent trips through the LDS crossbar).
__device__ inline unsigned wave_scan_u32(unsigned x)
{
#define SP_DPP_ADD(ctrl, rows) x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rows, 0xf, false);
    SP_DPP_ADD(0x111, 0xf)   // row_shr:1
    SP_DPP_ADD(0x112, 0xf)   // row_shr:2
    SP_DPP_ADD(0x114, 0xf)   // row_shr:4
    SP_DPP_ADD(0x118, 0xf)   // row_shr:8
    SP_DPP_ADD(0x142, 0xa)   // row_bcast:15 -> rows 1, 3
    SP_DPP_ADD(0x143, 0xc)   // row_bcast:31 -> rows 2, 3
#undef SP_DPP_ADD
    return x;
}

// 16-byte non-temporal store (dst is 16-byte aligned)
__device__ inline void store_nt(uint8_t *dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {a, b, c, d};
    __builtin_nontemporal_store(v, (u32x4 *)dst);
}

// 16-byte store at a wave-uniform base + a 32-bit lane offset: `global_store_dwordx4 voff, data, s[base]`.  From `base + off` the
// compiler builds the 64-bit address in the VALU (a v_mov of the zero high half and a v_lshl_add_u64 per store).
__device__ inline void store16_at(uint8_t *base, unsigned off, uint32_t a, uint32_t b, uint32_t c, uint32_t d, bool nt)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {a, b, c, d};
    if (nt) asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(off), "v"(v), "s"(base) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(off), "v"(v), "s"(base) : "memory");
}

// LDS accesses at a compile-time offset plus a byte offset held in a VGPR, through a pointer made from the integer (the dynamic LDS
// block starts at address 0: k_frames has no static LDS and checks it once): `ds_* vaddr offset:imm`.  Going through `smem + ...`
// instead leaves a `v_add_u32 v, 0, v` per access behind - the block's address is only replaced by its value after the last folding pass.
typedef __attribute__((address_space(3))) uint32_t *LdsU32;
__device__ inline uint32_t lds_read_u32(int fixed_off, unsigned var_off) { return *(LdsU32)(uintptr_t)(unsigned)(fixed_off + var_off); }
__device__ inline void lds_count(int fixed_off, unsigned var_off)
{
    __hip_atomic_fetch_add((LdsU32)(uintptr_t)(unsigned)(fixed_off + var_off), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// 4 * byte J of a dword in ONE instruction (sub-dword addressing): the LDS byte offset of a tile byte's LUT entry.
template <int J>
__device__ inline unsigned byte_times4(unsigned w)
{
    unsigned r;
    if constexpr (J == 0) asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(w));
    else if constexpr (J == 1) asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(w));
    else if constexpr (J == 2) asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(w));
    else asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(w));
    return r;
}

// Does any of the raw f32 words (lo[e], hi[e]: the two halves of one 64-bit load) hold an infinity or a NaN?  x*0 is NaN exactly for
// those; v_pk_fma_f32 takes a sample's two words at once.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// The launch arguments as they lie in the kernel-argument segment (FrameArgs is k_frames' first parameter), behind an opaque copy of
// the segment pointer: fields read through it are fetched (s_load) where they are used - the side outputs once per group, the
// request's end - instead of sitting in SGPRs from the kernel's first instruction on (the compiler loads every field of a by-value
// argument it can see at the entry; the two dozen that only the side outputs need cost as many spilled SGPRs in the frame loop).
typedef const __attribute__((address_space(4))) FrameArgs *LateArgs;
__device__ inline LateArgs late_args()
{
    LateArgs p = (LateArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// First register pass: stages 1-4 inside window [0, 4) with literal twiddles.
template <bool TRIV>
__device__ inline void fft_pass1(double (&re)[16], double (&im)[16])
{
#pragma unroll
    for (int s = 1; s <= 4; s++) {
        const int u = s - 1;
#pragma unroll
        for (int e0 = 0; e0 < 16; e0++) {
            if (e0 & (1 << u)) continue;
            const int e1 = e0 | (1 << u);
            const int k = (e0 & ((1 << u) - 1)) << (4 - s);     // fft_nayuki.js:76-78: table index j * n / size, in units of n / 16
            const double c = kTw16[k].c, sn = kTw16[k].s;
            const double rl = re[e1], il = im[e1];
            double tpre, tpim;                                   // fft_nayuki.js:80-81
            if (TRIV && k == 0) {
                // (1, 0): x*1 + y*0 == x bit for bit for finite x, y (only the sign of a zero can differ; nothing depends on it)
                tpre = rl;
                tpim = il;
            } else if (k == 4) {
                // sine exactly 1: y*1 == y for every y, NaN and infinities included
                tpre = rl * c + il;
                tpim = il * c - rl;
            } else {
                tpre = rl * c + il * sn;
                tpim = il * c - rl * sn;
            }
            const double rj = re[e0], ij = im[e0];
            re[e1] = rj - tpre;
            im[e1] = ij - tpim;
            re[e0] = rj + tpre;
            im[e0] = ij + tpim;
        }
    }
}

template <int NHI>
__device__ inline bool raw_f32_nonfinite(const uint32_t (&lo)[16], const uint32_t (&hi)[NHI])
{
    static_assert(NHI == 16, "8-byte samples: two f32 words each");
    unsigned long long s = 0ull;
    const unsigned long long zero = 0ull;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const unsigned long long w = ((unsigned long long)hi[e] << 32) | lo[e];   // the register pair the 64-bit load filled
        asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(s) : "v"(w), "v"(zero));
    }
    const float s0 = __uint_as_float((uint32_t)s), s1 = __uint_as_float((uint32_t)(s >> 32));
    return __ballot(s0 != s0 || s1 != s1) != 0ull;
}


// The waves that share a frame (n = 2048: two, n = 4096: four) meet through an LDS counter instead of the workgroup barrier, which
// held every frame of a round to the pace of the slowest wave and kept all waves in the same phase (all in the VALU, then all in the
// LDS).  Every LDS operation a wave has issued is ahead of its increment in the LDS queue (a wave's LDS operations execute in
// order), so "my writes are visible" and "my reads are done" both hold once the partners see the count.
template <bool COUNTER, bool BLOCK_SYNC>
struct FrameMeet {
    unsigned addr;     // LDS byte address of the frame's counter (wave-uniform)
    unsigned target;   // the count once every wave of the frame has arrived the next time
    unsigned step;     // waves per frame
    // One asm block each (a C loop around an atomic splits the kernel's big basic blocks and costs the register allocator 60+ spilled
    // VGPRs).  arrive(): lane 0 adds one.  wait(): the wave polls until every wave of the frame has arrived as often as itself.
    // A wave alternates arrive and wait strictly, so no wave is ever two arrivals ahead and the count cannot be reached early.
    __device__ inline void arrive()
    {
        if constexpr (COUNTER) {
            unsigned long long save;
            unsigned a_v, one_v;
            asm volatile("v_mov_b32 %[a_v], %[addr]\n\t"
                         "v_mov_b32 %[one_v], 1\n\t"
                         "s_mov_b64 %[save], exec\n\t"
                         "s_mov_b64 exec, 1\n\t"
                         "ds_add_u32 %[a_v], %[one_v]\n\t"
                         "s_mov_b64 exec, %[save]"
                         : [save] "=&s"(save), [a_v] "=&v"(a_v), [one_v] "=&v"(one_v)
                         : [addr] "s"(addr)
                         : "memory");
        }
    }
    __device__ inline void wait()
    {
        if constexpr (COUNTER) {
            target = (unsigned)__builtin_amdgcn_readfirstlane((int)(target + step));   // wave-uniform, kept in an SGPR
            unsigned a_v, got_v, got_s;
            asm volatile("v_mov_b32 %[a_v], %[addr]\n"
                         "L_sp_meet_%=:\n\t"
                         "ds_read_b32 %[got_v], %[a_v]\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "v_readfirstlane_b32 %[got_s], %[got_v]\n\t"
                         "s_sub_i32 %[got_s], %[got_s], %[target]\n\t"
                         "s_cmp_lt_i32 %[got_s], 0\n\t"
                         "s_cbranch_scc1 L_sp_meet_%="
                         : [a_v] "=&v"(a_v), [got_v] "=&v"(got_v), [got_s] "=&s"(got_s)
                         : [addr] "s"(addr), [target] "s"(target)
                         : "memory", "scc");
        } else {
            spk::frame_sync<BLOCK_SYNC>();
        }
    }
    __device__ inline void operator()()
    {
        arrive();
        wait();
    }
};

template <int LOG2N, bool CH, int PFB>
__global__ __launch_bounds__(kFrameThreads, 1) void k_frames(const FrameArgs a, const int format, const double2 *__restrict__ stage_tw,
                                                       const int group_frames, const int groups)
{
    constexpr int kThreads = kFrameThreads;   // eight waves, two per SIMD, one workgroup per CU
    constexpr int N = 1 << LOG2N;
    constexpr int T = N / 16;                       // threads per frame
    constexpr int FPB = kThreads / T;               // frames per round
    constexpr bool BLOCK_SYNC = T > 64;
    constexpr int TWMAX = frames_tw_max_stage(N);
    constexpr int NPASS = (LOG2N + 3) / 4;
    constexpr bool PERMLANE_MID = LOG2N == 13;
    constexpr bool STAGED = PFB == 0;   // the generic loaders leave no registers for a whole pass's twiddles: read stage by stage
    // 8-byte samples at n >= 2048: a frame's samples are requested when it starts, not one frame ahead (the prefetch registers of
    // the next frame were what spilled there: cf32, n = 2048: 411 -> 358 us per 32 768 frames); its partner wave covers the latency
    // The L/R split at n >= 2048 likewise (its 16 partner values on top of a frame's 32 spill ~32 registers with the prefetch kept): a
    // spill reload waits for every vector-memory operation issued before it - in-order completion - i.e. for the prefetch itself.
    // Requested at frame start, 12 spilled registers are left and configs 3 / 5 in channel mode take 12 % less time (1.39 -> 1.22 ms,
    // 3.41 -> 2.96 ms).  (The taper from L2 per frame instead of registers: no spills at all, and slower than either.)
    // (8-byte samples with the split at n = 1024: 22 spilled registers -> 0, 90.1 -> 88.0 us at config 2's shape; n = 512: 14 -> 0,
    // 81.6 -> 77.8 us per 2^24 samples; n = 256, 6 spilled registers, is 2 % faster with the prefetch and keeps it)
    constexpr bool LATE_PF = ((PFB == 8 || (CH && PFB != 0)) && LOG2N >= 11) || (CH && PFB == 8 && LOG2N >= 9);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Layout lay = layout(N, a.lut_len, group_frames);
    double *s_xch = (double *)(smem + kOffXch);
    double2 *s_tw = (double2 *)(smem + lay.off_tw);
    const double *edge_g = (const double *)(smem + lay.off_gedge);
    const double *edge_cb = (const double *)(smem + lay.off_cbedge);
    unsigned long long *s_mm = (unsigned long long *)(smem + lay.off_mm);
    unsigned char *s_tile = smem + lay.off_tile;
    unsigned int *const s_lut = (unsigned int *)(smem + kOffLut);
    unsigned int *const s_cells = (unsigned int *)(smem + kOffCells);
    [[maybe_unused]] unsigned int *s_done = (unsigned int *)(smem + lay.off_done);
    double *s_red = (double *)(smem + lay.off_amp);                           // the workgroup's share of dBfs_min / dBfs_max so far
    double2 *s_amp = (double2 *)(smem + lay.off_amp + 16);                    // [2][group_frames] (I, Q) of sample n/2, by group parity

    // (lds_read_u32 / lds_count address the dynamic LDS block from 0)
    if ((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();
    const int tid = threadIdx.x;
    [[maybe_unused]] const int lane = tid & 63;
    const int fs = tid / T;                         // frame slot within a round
    const int tl = tid % T;                         // thread within the frame
    double *xbuf = s_xch + fs * (N + N / 16);
    constexpr bool COUNTER_SYNC = BLOCK_SYNC && T < kThreads;   // a frame's waves are not the whole workgroup (n = 2048, 4096)
    FrameMeet<COUNTER_SYNC, BLOCK_SYNC> meet{
        (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(const __attribute__((address_space(3))) unsigned int *)(s_done + 2 + fs)),
        0u, (unsigned)(T / 64)};
    const int tile_pitch = N + kTilePad;
    const int cmax = a.lut_len - 1;

    // groups are dealt so that workgroups sharing an XCD (blockIdx % 8) own neighbouring groups
    const int xcd = blockIdx.x & 7, lane_in_xcd = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int chunk = (groups + 7) >> 3;
    const int g_end = min(groups, (xcd + 1) * chunk);

    constexpr bool PF = PFB != 0;
    const int sidx_pf = (int)(__brev((unsigned)tl) >> (32 - (LOG2N - 4)));
    const int rounds = (group_frames + FPB - 1) / FPB;
    uint32_t raw_lo[PF ? 16 : 1], raw_hi[PFB == 8 ? 16 : 1];
    int raw_back = 0;
    auto request = [&](int xq) {
        if constexpr (PF) {
            // (the prefetching variants only run when every frame lies inside the buffer: launch_frames)
            const int xc = xq < a.x_end ? xq : a.x_end - 1;
            constexpr bool UNI = T >= 64;   // a frame per wave or more: its start is wave-uniform
            const int sv = frame_start_in_bounds(a.stride, xc);
            const int64_t st = UNI ? __builtin_amdgcn_readfirstlane(sv) : sv;
            if constexpr (PFB == 3) raw_back = (st + N) * 3 + 1 > a.nbytes ? 1 : 0;
            issue_raw<PFB, UNI>(a.bytes, st, T, sidx_pf, raw_lo, raw_hi, raw_back);
        }
    };
    // n = 1024, 32-frame groups: the first / second waves of the SIMDs each take one half of a group's frames
    const bool HALVES = T == 64 && group_frames == 32;
    const int fs0 = HALVES ? (fs / (FPB / 2)) * (group_frames / 2) + fs % (FPB / 2) : fs;     // the slot's frame in a group's first round
    // n <= 1024 requests the first frame's samples behind the table loads, below (config 2: -1.6 us per launch); above, where the tables
    // are a quarter of the size and the taper goes to registers after them, the old order measures the same (n = 2048) or 1.3 % better
    // (n = 8192: the other order shifts the loop's register allocation)
    constexpr bool REQ_AFTER_TABLES = PF && !LATE_PF && LOG2N <= 10;
    if (PF && !LATE_PF && !REQ_AFTER_TABLES && xcd * chunk + lane_in_xcd < g_end) request(a.frame0 + (xcd * chunk + lane_in_xcd) * group_frames + fs0);

    // workgroup 0's first wave owns the reply's initial state in the first launch of a request (below)
    const bool owner = blockIdx.x == 0 && __builtin_amdgcn_readfirstlane(tid >> 6) == 0 && a.first;   // (wave-uniform)
    constexpr bool WIN_LDS = lds_win_in_lds(N);   // taper in LDS for n <= 1024, in registers for the whole launch above
    double *s_win = (double *)(smem + lay.off_win);
    constexpr int MMS = mm_slots(N);
    constexpr bool LATE_SIDE = late_side_outputs(N);
    {
        // tables -> LDS: every global load is issued before the first LDS store (one memory latency for the prologue)
        constexpr int WINK = WIN_LDS ? (N + kThreads - 1) / kThreads : 1;
        double win_r[WINK];
        if constexpr (WIN_LDS) {
#pragma unroll
            for (int k = 0; k < WINK; k++) {
                const int i = tid + k * kThreads, e = i / T, t = i % T;
                win_r[k] = i < N ? a.window[rev4(e) * T + (int)(__brev((unsigned)t) >> (32 - (LOG2N - 4)))] : 0.0;
            }
        }
        constexpr int NTW = frames_tw_entries(N);
        constexpr int TWK = (NTW + kThreads - 1) / kThreads;
        double2 tw_r[TWK > 0 ? TWK : 1];
#pragma unroll
        for (int k = 0; k < TWK; k++) {
            const int i = tid + k * kThreads;
            tw_r[k] = i < NTW ? stage_tw[i] : make_double2(0.0, 0.0);
        }
        const unsigned int lut_r = tid < a.lut_len ? a.lut_rgba[tid] : 0u;       // lut_len <= 256 < kThreads
        double cb_r[(SP_CB_HIST_SIZE + kThreads) / kThreads];
        const double ge_r = tid < a.lut_len ? a.gray_edge[tid] : 0.0;
#pragma unroll
        for (int k = 0; k < (SP_CB_HIST_SIZE + kThreads) / kThreads; k++) {
            const int i = tid + k * kThreads;
            cb_r[k] = i <= SP_CB_HIST_SIZE ? a.cb_edge[i] : 0.0;
        }
        // Workgroup 0 of a request's first launch clears the reply's histograms and sets its dBfs range to (0, -200): its first wave
        // alone, so that the wave knows when the stores have landed (publish() below).  Fire-and-forget, behind the table loads.
        if (owner) {
            const LateArgs la = late_args();
            unsigned long long *const out_c = la->out_c, *const out_cb = la->out_cb;
            unsigned long long *const out_mm = (unsigned long long *)la->out_minmax;
            constexpr int kClr = (kLdsMaxLut + SP_CB_HIST_SIZE + 63) / 64;
#pragma unroll
            for (int k = 0; k < kClr; k++) {
                const int i = tid + 64 * k;
                unsigned long long *const dst = i < kLdsMaxLut ? (out_c && i < a.lut_len ? out_c + i : nullptr)
                                                               : (out_cb && i < kLdsMaxLut + SP_CB_HIST_SIZE ? out_cb + (i - kLdsMaxLut) : nullptr);
                if (dst) __hip_atomic_store(dst, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid < 2 && out_mm)
                __hip_atomic_store(out_mm + tid, tid ? 0xc069000000000000ull : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // -200.0, 0.0
        }
        // The first frame's samples are requested BEHIND the table loads (vector-memory operations complete in order: requested ahead of
        // them, the wait for the tables - L2 hits - was a wait for the samples from HBM), and unconditionally (a frame past the end is
        // clamped), so that the compiler can count the 16 younger loads in that wait: s_waitcnt vmcnt(16).
        if constexpr (REQ_AFTER_TABLES) request(a.frame0 + (xcd * chunk + lane_in_xcd) * group_frames + fs0);
        // what needs no table is set up while the loads are in flight (a table load takes ~2.3 us at the start of a launch)
        for (int i = tid; i < a.cells; i += kThreads) s_cells[i] = 0;
        if (tid < 8) s_done[tid] = 0;
        if (tid < 2) s_red[tid] = tid ? -200.0 : 0.0;                             // worker.js:35-36
        for (int i = tid; i < (LATE_SIDE ? 2 : 1) * group_frames * MMS; i += kThreads) {
            s_mm[2 * i] = 0x7ff0000000000000ull;
            s_mm[2 * i + 1] = 0ull;
        }
#pragma unroll
        for (int k = 0; k < TWK; k++) {
            const int i = tid + k * kThreads;
            if (i < NTW) s_tw[i] = tw_r[k];
        }
        if (tid < a.lut_len) s_lut[tid] = lut_r;
        if (tid < a.lut_len) ((double *)(smem + lay.off_gedge))[tid] = ge_r;
#pragma unroll
        for (int k = 0; k < (SP_CB_HIST_SIZE + kThreads) / kThreads; k++) {
            const int i = tid + k * kThreads;
            if (i <= SP_CB_HIST_SIZE) ((double *)(smem + lay.off_cbedge))[i] = cb_r[k];
        }
        if constexpr (WIN_LDS) {
#pragma unroll
            for (int k = 0; k < WINK; k++) {
                const int i = tid + k * kThreads;
                if (i < N) s_win[i] = win_r[k];
            }
        }
    }

    const double *const wbase = s_win + tl;   // stored as the threads read it: entry e*T + tl = taper[rev4(e)*T + rev(tl)]
    double win_reg[WIN_LDS ? 1 : 16];
    if constexpr (!WIN_LDS) {
        const int sidx = (int)(__brev((unsigned)tl) >> (32 - (LOG2N - 4)));
#pragma unroll
        for (int e = 0; e < 16; e++) win_reg[e] = a.window[rev4(e) * T + sidx];
    }
    lds_barrier();

    const spfmt::View view{a.bytes, a.nbytes, a.nelem};
    uint32_t pf_word = 0;
    // epilogue constants (sp_host.cpp build_thresholds): t = a + b*log2(|X|^2), already lowered by the margin
    const float g_a = a.g2_a, g_b = a.g2_b, g_m = a.g2_m;
    const float c_a = a.c2_a, c_b = a.c2_b, c_m = a.c2_m, c_lo = a.c2_lo, c_hi = a.c2_hi;
    const float thr = fminf(a.g2_thr, a.c2_thr);
    // A VALU instruction reads ONE scalar register: with both coefficients of a scale in SGPRs the compiler copies one of them into a
    // VGPR again for every batch of bins (24 v_mov per frame).  The addends and the upper clamp bound live in VGPRs instead.
    // (n <= 1024, where registers are left: above, the loop sits at the 256-VGPR limit and three more spill)
    float g_a_v = g_a, c_a_v = c_a, c_hi_v = c_hi;
    constexpr bool COEF_VGPR = LOG2N <= 10 && !CH && !(PFB == 8 && LOG2N < 9);
    if constexpr (COEF_VGPR) asm volatile("" : "+v"(g_a_v), "+v"(c_a_v), "+v"(c_hi_v));
    // ... and, at n = 1024, both scales of a batch's two bins as packed pairs (below that size the loop measures the same with and
    // without, and the launch-bound config 1 pays 1 % for the longer set-up: profiles/r05_experiments.txt)
    constexpr bool PK_SCALES = COEF_VGPR && LOG2N == 10;
    [[maybe_unused]] f32x2 g_b2 = {g_b, g_b}, g_a2 = {g_a, g_a}, c_b2 = {c_b, c_b}, c_a2 = {c_a, c_a};
    if constexpr (PK_SCALES) asm volatile("" : "+v"(g_b2), "+v"(g_a2), "+v"(c_b2), "+v"(c_a2));
    // clamp bounds of the colour value: clipped pixels sit in the middle of the first / last step, far from the risky zone
    const float g_lo = 0.5f, g_hi = (float)cmax + 0.5f;
    const int cell_sp0 = a.cells - 2;   // -inf / NaN dB (colour 0, bin 0), +inf dB is the next one (last colour, bin 0)

    // Side outputs of a finished group of frames (worker.js:124-136), by the workgroup's first 3 * group_frames threads: gauge_mins and
    // gauge_maxs from the frame's extreme |X|^2 (d is monotone in |X|^2, so the frame's extreme d belong to them), gauge_amps from its
    // raw centre sample: one software log10 per output.  The frame's clamped extremes are also its share of the request's dBfs range
    // (worker.js:124-125): they are folded into the workgroup's; the frame's slots are reset.
    auto side_outputs = [&](const int x0, const int par) {
        if (__builtin_amdgcn_readfirstlane(tid) >= 3 * group_frames) return;   // (wave-uniform: the waves that hold none of those threads)
        const LateArgs la = late_args();
        // three scalar loads, selected per lane below (the compiler turns a select between fields into ONE indexed vector load, whose
        // wait covers every outstanding vector-memory operation of the wave: the sample prefetch, ~2 us)
        uint8_t *out_min = la->gauge_mins, *out_max = la->gauge_maxs, *out_amp = la->gauge_amps;
        asm volatile("" : "+s"(out_min), "+s"(out_max), "+s"(out_amp));
        const double gain = la->gain, range = la->range, bn_db = la->block_norm_db;
        if (tid >= 3 * group_frames) return;
        const int role = (tid >= group_frames ? 1 : 0) + (tid >= 2 * group_frames ? 1 : 0), f = tid - role * group_frames;
        double arg;
        if (role < 2) {
            unsigned long long ext = role ? 0ull : 0x7ff0000000000000ull;
#pragma unroll
            for (int k = 0; k < MMS; k++) {
                unsigned long long *slot = s_mm + 2 * (((LATE_SIDE ? par : 0) * group_frames + f) * MMS + k) + role;
                const unsigned long long v = *slot;
                ext = role ? (v > ext ? v : ext) : (v < ext ? v : ext);
                *slot = role ? 0ull : 0x7ff0000000000000ull;
            }
            arg = __longlong_as_double((long long)ext);
        } else {
            const double2 c = s_amp[par * group_frames + f];
            arg = c.x * c.x + c.y * c.y;                                                       // worker.js:130-131
        }
        const double l5 = 5 * spjs::log10(arg);
        double v;
        if (role == 2) {
            v = l5 + gain;
        } else {
            const double d = (l5 + bn_db + gain) - gain;                                       // dBfs - gain, worker.js:100
            v = role ? (d > -200.0 ? d : -200.0) : (d < 0.0 ? d : 0.0);                        // worker.js:82-83, 102-103
            if (role) lds_max_f64(&s_red[1], v);
            else lds_min_f64(&s_red[0], v);
        }
        uint8_t *const out = role == 0 ? out_min : role == 1 ? out_max : out_amp;
        if (out && x0 + f < a.x_end) out[x0 + f] = clamp_u8(0.5 + (range + v) * 256 / range);   // worker.js:128-136
    };
    // write-out of tile rows [f0, f0 + fcount) by the threads [t0, t0 + dthreads), slice `part` of `nparts`
    auto drain_rows = [&](const int x0, const int part, const int nparts, const int f0, const int fcount, const int t0, const int dthreads,
                          const bool nt_rows) {
        const int dt = tid - t0;
        if (dt < 0) return;
        // (the image's address, width and layout are read from the argument segment here, once per write-out, instead of sitting in
        // SGPRs through every frame)
        const LateArgs la = late_args();
        uint8_t *const img = la->rgba;
        const int img_width = la->width, img_waterfall = la->waterfall, img_fast = la->rgba_fast;
        if (img) {
            if (!img_waterfall) {
                // spectrogram: image is n rows x width columns; row y holds bin (n/2 - y) mod n            worker.js:90,117
                // The tile keeps a frame as the epilogue leaves it: 16 bytes per thread, byte e = bin tl + e*T.  A write-out item is
                // one of a thread's four dwords (bins tl + (4*e4 + j)*T, j = 0..3) of 4 consecutive frames: four 16-byte stores in four
                // rows; the items of a row segment (8 frame quads) sit in lanes 4 apart, and a wave's dword reads are conflict-free
                // (tile pitch = 1 dword mod 8).
                const int quads = fcount / 4;                     // a power of two (launch_frames)
                const int lq = 31 - __builtin_clz((unsigned)quads);
                const int items = (N / 4) * quads;
                for (int it0 = dt + part * 2 * dthreads; it0 < items; it0 += nparts * 2 * dthreads) {
                    uint32_t gb[2][4];
                    int i0v[2], xav[2];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int it = it0 + u * dthreads;
                        const int itc = it < items ? it : it0;
                        const int e4 = itc & 3, fq = (itc >> 2) & (quads - 1), tq = (itc >> 2) >> lq;   // tq: thread of the frame
                        i0v[u] = tq + 4 * e4 * T;
                        xav[u] = it < items ? x0 + f0 + fq * 4 : a.x_end;
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            gb[u][k] = *(const uint32_t *)(s_tile + __umul24((unsigned)(f0 + fq * 4), (unsigned)tile_pitch) + k * tile_pitch + tq * 16 + e4 * 4);
                    }
                    uint32_t px[2][4][4];
                    const auto lut_at = [&](unsigned off4) { return lds_read_u32(kOffLut, off4); };
#pragma unroll
                    for (int u = 0; u < 2; u++)
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            px[u][0][k] = lut_at(byte_times4<0>(gb[u][k]));
                            px[u][1][k] = lut_at(byte_times4<1>(gb[u][k]));
                            px[u][2][k] = lut_at(byte_times4<2>(gb[u][k]));
                            px[u][3][k] = lut_at(byte_times4<3>(gb[u][k]));
                        }
                    if (img_fast) {
                        // rows are 16-byte aligned, the width is a multiple of 4 and the image is below 4 GiB: 32-bit offsets from the
                        // uniform base (24-bit multiplies), no per-store checks
#pragma unroll
                        for (int u = 0; u < 2; u++) {
                            const int xa = xav[u];
                            if (xa >= a.x_end) continue;
                            const unsigned y0 = (unsigned)(N / 2 - i0v[u]) & (N - 1);
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const unsigned y = (y0 - (unsigned)(j * T)) & (N - 1);
                                const unsigned off = (__umul24(y, (unsigned)img_width) + (unsigned)xa) * 4u;
                                // written once, never read by this kernel: non-temporal where a group's row segments are whole
                                // 128-byte lines, so that the image does not displace the capture's lines in L2 (measured: 2.5 % of the
                                // kernel at n = 1024); shorter segments (large n) are pieces of lines that L2 has to merge with the
                                // neighbouring groups' pieces (non-temporal there doubled the HBM traffic)
                                store16_at(img, off, px[u][j][0], px[u][j][1], px[u][j][2], px[u][j][3], nt_rows);
                            }
                        }
                        continue;
                    }
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int xa = xav[u];
                        if (xa >= a.x_end) continue;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int i = i0v[u] + j * T;
                            const int y = (N / 2 - i) & (N - 1);
                            uint8_t *dst = img + ((size_t)y * (size_t)img_width + (size_t)xa) * 4;
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
                // An item is one dword of the tile - the colour bytes of bins t + (4*e4 + j)*T, j = 0..3, of one frame - read once and
                // stored as four pixels T columns apart; consecutive lanes take consecutive t, so each of a wave's four store
                // instructions covers 64 consecutive pixels of an image row.  (Four consecutive COLUMNS per item - one 16-byte store,
                // but four byte reads from four tile columns - took 6 ... 14 % more of the kernel than the spectrogram layout.)
                const int items = fcount * (N / 4);
                for (int it = dt + part * dthreads; it < items; it += nparts * dthreads) {
                    const int tq = it % T, e4 = (it / T) & 3, f = f0 + it / (4 * T);
                    const int xa = x0 + f;
                    if (xa >= a.x_end) continue;
                    const uint32_t gb = *(const uint32_t *)(s_tile + f * tile_pitch + tq * 16 + e4 * 4);
                    // columns (i + n/2 - 1) mod n of bins i = tq + (4*e4 + j)*T: c0 + j*T without a wrap inside an item - except for
                    // the one item per frame whose first pixel is the row's last (bin n/2): its other three start the row
                    const int c0 = (tq + 4 * e4 * T + N / 2 - 1) & (N - 1);
                    uint32_t *const row = (uint32_t *)(img + (size_t)(img_width - 1 - xa) * N * 4);
                    uint32_t *const p = row + (c0 == N - 1 ? -1 : c0);
                    const auto lut_at = [&](unsigned off4) { return lds_read_u32(kOffLut, off4); };
                    __builtin_nontemporal_store(lut_at(byte_times4<0>(gb)), row + c0);
                    __builtin_nontemporal_store(lut_at(byte_times4<1>(gb)), p + 1 * T);
                    __builtin_nontemporal_store(lut_at(byte_times4<2>(gb)), p + 2 * T);
                    __builtin_nontemporal_store(lut_at(byte_times4<3>(gb)), p + 3 * T);
                }
            }
        }
    };
    // non-temporal stores where a group's row pieces are whole 128-byte lines (below)
    auto drain = [&](const int x0, const int part, const int nparts) { drain_rows(x0, part, nparts, 0, group_frames, 0, kThreads, group_frames >= SP_NT_MIN_GROUP); };
    int drain_x0 = -1;
    int gpar = 0;   // parity of the workgroup's current group (s_amp)
    meet.arrive();   // the first re-distribution only waits (exchange<.., SECOND = false>)
    for (int g = xcd * chunk + lane_in_xcd; g < g_end; g += per_xcd) {
        const int x0 = a.frame0 + g * group_frames;
        for (int r = 0; r < rounds; r++) {
            // HALVES: the first waves of the SIMDs (slots 0 .. FPB/2-1) own the group's first half of the frames, the second waves the
            // other half, so that each set can write its half out by itself after the workgroup's last group
            const int fr = HALVES ? (fs / (FPB / 2)) * (group_frames / 2) + r * (FPB / 2) + fs % (FPB / 2) : r * FPB + fs;
            const int xr = x0 + fr;
            if (fr >= group_frames) continue;   // a slot without a frame in the group's last round (its next frame is already requested)
            const bool live = xr < a.x_end;
            const int x = live ? xr : a.x_end - 1;
            const int64_t start = frame_start(a.stride, x);

            double re[16], im[16];
            double win[16];
            double2 *const centre = tl == 0 ? &s_amp[gpar * group_frames + fr] : nullptr;   // thread 0 of the frame: where its raw centre sample goes
            bool nonfinite = true;   // wave-uniform
#pragma unroll
            for (int e = 0; e < 16; e++) win[e] = WIN_LDS ? wbase[e * T] : win_reg[WIN_LDS ? 0 : e];
            // the frame this slot processes next: the same slot one round on, or its frame in the workgroup's next group
            const int xn = (r + 1 < rounds && (HALVES || fr + FPB < group_frames)) ? xr + (HALVES ? FPB / 2 : FPB)
                                                                         : (g + per_xcd < g_end ? a.frame0 + (g + per_xcd) * group_frames + fs0 : -1);
            if constexpr (PF && LATE_PF) request(xr);
            if constexpr (PF) {
                if constexpr (PFB == 1) {
                    if (format == SP_FMT_CU4) nonfinite = decode_frame<SP_FMT_CU4, 1>(raw_lo, raw_hi, win, re, im, centre);
                    else nonfinite = decode_frame<SP_FMT_CS4, 1>(raw_lo, raw_hi, win, re, im, centre);
                } else if constexpr (PFB == 3) {
                    if (format == SP_FMT_CU12) nonfinite = decode_frame<SP_FMT_CU12, 1>(raw_lo, raw_hi, win, re, im, centre, 8 * raw_back);
                    else nonfinite = decode_frame<SP_FMT_CS12, 1>(raw_lo, raw_hi, win, re, im, centre, 8 * raw_back);
                } else if constexpr (PFB == 2) {
                    if (format == SP_FMT_CU8) nonfinite = decode_frame<SP_FMT_CU8, 1>(raw_lo, raw_hi, win, re, im, centre);
                    else nonfinite = decode_frame<SP_FMT_CS8, 1>(raw_lo, raw_hi, win, re, im, centre);
                } else if constexpr (PFB == 4) {
                    if (format == SP_FMT_CU16) nonfinite = decode_frame<SP_FMT_CU16, 1>(raw_lo, raw_hi, win, re, im, centre);
                    else nonfinite = decode_frame<SP_FMT_CS16, 1>(raw_lo, raw_hi, win, re, im, centre);
                } else {
                    if (format == SP_FMT_CU32) nonfinite = decode_frame<SP_FMT_CU32, 16>(raw_lo, raw_hi, win, re, im, centre);
                    else if (format == SP_FMT_CS32) nonfinite = decode_frame<SP_FMT_CS32, 16>(raw_lo, raw_hi, win, re, im, centre);
                    else {
                        decode_frame<SP_FMT_CF32, 16>(raw_lo, raw_hi, win, re, im, centre);
                        nonfinite = raw_f32_nonfinite<16>(raw_lo, raw_hi);
                    }
                }
                if (!LATE_PF && xn >= 0) request(xn);           // in flight during this frame's butterflies
            } else {
                asm volatile("" ::"v"(pf_word));
                if (a.in_bounds && xn >= 0 && xn < a.x_end) {
                    const int lines = (N * a.sample_width + 127) >> 7;
                    const int64_t nb = (int64_t)frame_start(a.stride, xn) * a.sample_width;
                    for (int l = tl; l < lines; l += T) pf_word = *(const uint32_t *)(a.bytes + ((nb + (int64_t)l * 128) & ~(int64_t)3));
                }
                switch (format) {
#define SP_CASE(F) case F: load_frame<F>(a, view, start, tl, T, LOG2N, win, re, im, centre); break;
                    SP_CASE(SP_FMT_CU4) SP_CASE(SP_FMT_CS4) SP_CASE(SP_FMT_CU8) SP_CASE(SP_FMT_CS8) SP_CASE(SP_FMT_CU12)
                    SP_CASE(SP_FMT_CS12) SP_CASE(SP_FMT_CU16) SP_CASE(SP_FMT_CS16) SP_CASE(SP_FMT_CU32) SP_CASE(SP_FMT_CS32)
                    SP_CASE(SP_FMT_CU64) SP_CASE(SP_FMT_CS64) SP_CASE(SP_FMT_CF32)
#undef SP_CASE
                default: load_frame<SP_FMT_CF64>(a, view, start, tl, T, LOG2N, win, re, im, centre); break;
                }
            }

            unsigned tw_off = 0;
            asm volatile("" : "+s"(tw_off));
            const double2 *tw = stage_tw + tw_off;
            // the previous group's write-out goes in two slices around this frame's passes, so that its stores drain while the SIMDs compute
            if (drain_x0 >= 0) {
                lds_barrier();
                if constexpr (!LATE_SIDE) side_outputs(drain_x0, gpar ^ 1);
                drain(drain_x0, 0, 2);
            }
            // ---- first pass: literal twiddles ------------------------------------------------------------------------------
            if constexpr (PFB == 0) {
                fft_pass1<false>(re, im);            // frames may leave the buffer (NaN samples), 16-byte formats
            } else if constexpr (PFB == 8) {
                if (nonfinite) fft_pass1<false>(re, im);
                else fft_pass1<true>(re, im);
            } else {
                fft_pass1<true>(re, im);             // integer samples times a finite taper (sp_api.hip: plan_frames_capable)
            }
            if constexpr (NPASS >= 2) {
                constexpr int WS1 = LOG2N >= 8 ? 4 : LOG2N - 4;
                constexpr int E1 = LOG2N >= 8 ? 8 : LOG2N;
                double *const b0 = xbuf + pad_idx(win_pos(tl, 0, 0)), *const b1 = xbuf + pad_idx(win_pos(tl, 0, WS1));
                PassTw<WS1, 5, STAGED ? 4 : E1, TWMAX> tw1;
                if constexpr (!STAGED) load_pass_tw(tw1, tl, s_tw, tw);
                // The first re-distribution never leaves a wave, whatever n: under window 0 as under window [4,8) the 64 threads of a
                // wave hold exactly the positions [1024 w, 1024 w + 1023] of their frame.  So the waves of a frame (n >= 2048) only
                // wait, before its first writes, for the partners' last reads of the frame before; between its writes and reads the
                // LDS's in-order execution of a wave's operations is all that is needed, as at n <= 1024.
                meet.wait();
                exchange<0, WS1, false, false>(re, b0, b1);
                exchange<0, WS1, false, true>(im, b0, b1);
                exchange_wait(re, im);
                if constexpr (STAGED) fft_pass_staged<WS1, 5, E1, TWMAX>(re, im, tl, s_tw, tw);
                else fft_pass<WS1, 5, E1>(re, im, tw1);
                if constexpr (PERMLANE_MID) {
                    // n = 8192: the second re-distribution stays inside the wave too - the register transpose of the 1024-point
                    // layout (window [4,8) -> [6,10) of the wave's block), two stages there, and only then the one re-distribution
                    // that crosses waves, to window [9,13) for the last three stages.  Four workgroup barriers per frame instead of
                    // eight; the swaps cost the VALU, which has the time at this size.
                    PassTw<6, 9, STAGED ? 8 : 10, TWMAX> tw2;
                    if constexpr (!STAGED) load_pass_tw(tw2, tl, s_tw, tw);
                    exchange_permlane<10>(re);
                    exchange_permlane<10>(im);
                    if constexpr (STAGED) fft_pass_staged<6, 9, 10, TWMAX>(re, im, tl, s_tw, tw);
                    else fft_pass<6, 9, 10>(re, im, tw2);
                    constexpr int WS3 = LOG2N - 4;
                    double *const b2 = xbuf + pad_idx(win_pos(tl, 0, 6)), *const b3 = xbuf + pad_idx(win_pos(tl, 0, WS3));
                    PassTw<WS3, 11, STAGED ? 10 : LOG2N, TWMAX> tw3;
                    if constexpr (!STAGED) load_pass_tw(tw3, tl, s_tw, tw);
                    exchange<6, WS3, BLOCK_SYNC, false, decltype(meet) &, false>(re, b2, b3, meet);
                    exchange<6, WS3, BLOCK_SYNC, true>(im, b2, b3, meet);
                    exchange_wait(re, im);
                    if constexpr (STAGED) fft_pass_staged<WS3, 11, LOG2N, TWMAX>(re, im, tl, s_tw, tw);
                    else fft_pass<WS3, 11, LOG2N>(re, im, tw3);
                } else if constexpr (NPASS >= 3) {
                    constexpr int WS2 = LOG2N >= 12 ? 8 : LOG2N - 4;
                    constexpr int E2 = LOG2N >= 12 ? 12 : LOG2N;
                    double *const b2 = xbuf + pad_idx(win_pos(tl, 0, WS2));
                    PassTw<WS2, 9, STAGED ? 8 : E2, TWMAX> tw2;
                    if constexpr (!STAGED) load_pass_tw(tw2, tl, s_tw, tw);
                    if constexpr (LOG2N == 9 || LOG2N == 10) {
                        // two register bits against lane bits 4 / 5: v_permlane16_swap / v_permlane32_swap.  The swaps cost the VALU
                        // about what the LDS round trip costs the LDS pipe (measured: 1.2 % of the kernel in favour of the swaps)
                        exchange_permlane<LOG2N>(re);
                        exchange_permlane<LOG2N>(im);
                    } else {
                        // (writes the positions the wave itself read last: no wait before them)
                        exchange<WS1, WS2, BLOCK_SYNC, false, decltype(meet) &, false>(re, b1, b2, meet);
                        exchange<WS1, WS2, BLOCK_SYNC, true>(im, b1, b2, meet);
                        exchange_wait(re, im);
                    }
                    if constexpr (STAGED) fft_pass_staged<WS2, 9, E2, TWMAX>(re, im, tl, s_tw, tw);
                    else fft_pass<WS2, 9, E2>(re, im, tw2);
                    static_assert(NPASS <= 3, "four passes (n = 8192) take the register-transpose branch above");
                }
            }
            // now register e of thread tl holds bin i = tl + e*T

            if constexpr (CH) {   // fft_nayuki.js:103-119, partner bin n-i fetched through LDS
                // (the partner values eight at a time where registers are short, n >= 2048: two LDS waits per component instead of one,
                // and 16 registers fewer at the frame's register peak)
                constexpr int PH = LOG2N >= 11 ? 8 : 16;
                double pp[PH];
                meet.wait();   // (announced after the last re-distribution's reads)
#pragma unroll
                for (int e = 0; e < 16; e++) xbuf[pad_idx(tl + e * T)] = re[e];
                meet();
#pragma unroll
                for (int h = 0; h < 16; h += PH) {
#pragma unroll
                    for (int k = 0; k < PH; k++) pp[k] = xbuf[pad_idx((N - (tl + (h + k) * T)) & (N - 1))];
#pragma unroll
                    for (int k = 0; k < PH; k++) {
                        const int e = h + k, i = tl + e * T;
                        const double orr = re[e];
                        if (i == 0) {
                        } else if (i == N / 2) {
                            re[e] = 0.0;
                        } else if (i < N / 2) {
                            re[e] = 0.5 * (orr + pp[k]);
                        } else {
                            re[e] = 0.5 * (-pp[k] + orr);
                        }
                    }
                    if (PH < 16) asm volatile("" ::: "memory");   // the second half's reads stay behind the first half's arithmetic
                }
                meet();
#pragma unroll
                for (int e = 0; e < 16; e++) xbuf[pad_idx(tl + e * T)] = im[e];
                meet();
#pragma unroll
                for (int h = 0; h < 16; h += PH) {
#pragma unroll
                    for (int k = 0; k < PH; k++) pp[k] = xbuf[pad_idx((N - (tl + (h + k) * T)) & (N - 1))];
#pragma unroll
                    for (int k = 0; k < PH; k++) {
                        const int e = h + k, i = tl + e * T;
                        const double oi = im[e];
                        if (i == 0 || i == N / 2) {
                            im[e] = 0.0;
                        } else if (i < N / 2) {
                            im[e] = 0.5 * (oi - pp[k]);
                        } else {
                            im[e] = 0.5 * (pp[k] + oi);
                        }
                    }
                    if (PH < 16) asm volatile("" ::: "memory");
                }
                meet.arrive();   // for the next frame's first re-distribution
            }

            if (drain_x0 >= 0) {
                drain(drain_x0, 1, 2);
                lds_barrier();
                // The previous group's side outputs, here: the two waves that evaluate them (a software log10, ~1 us) next meet the
                // others at the start of the following group, where the first waves of the SIMDs arrive early anyway; in front of this
                // barrier they held everybody up.  (The frames' extremes and centre samples are kept per group parity for it.)
                if constexpr (LATE_SIDE) side_outputs(drain_x0, gpar ^ 1);
                drain_x0 = -1;
            }
            // ---- |X|^2 -> colour index, centi-bel level ---------------------------------------------------------------------
            // t = a + b*log2((float)|X|^2) in f32 is within the margin m of the real-valued position of |X|^2 on the index
            // scale (sp_host.cpp); a and the clamp bounds are lowered by m, so floor(t) is exact unless fract(t) >= 1 - 2m.
            // Lanes past that threshold (and centi-bel values at or beyond the ends of the scale, +-inf and NaN among them, whose
            // clamp bounds lie past it by construction) take the exact edge compare.
            // four independent min / max chains: a dependent f64 operation waits several issue slots
            double mn4[4] = {spjs::inf(), spjs::inf(), spjs::inf(), spjs::inf()}, mx4[4] = {0.0, 0.0, 0.0, 0.0};
            uint32_t *trow = (uint32_t *)(s_tile + fr * tile_pitch + tl * 16);
            if (live) {
                constexpr int EB = 2;   // bins per batch
                [[maybe_unused]] uint32_t tile_word = 0;          // four colour bytes per tile dword
                constexpr bool TILE_BYTES = LOG2N >= 10;         // (n <= 512: no gain measured; n = 2048: neutral; n = 8192: -0.6 %)
                [[maybe_unused]] const unsigned trow_addr = (unsigned)(size_t)(__attribute__((address_space(3))) uint32_t *)trow;
                // a batch of bins at a time: independent chains for the VALU, one branch per batch, four colour bytes per tile dword
#pragma unroll
                for (int q = 0; q < 16 / EB; q++) {
                    double abs2[EB];
                    float tg[EB], tc[EB];
                    int gi[EB], cell[EB];
                    unsigned cell4[EB];              // 4 * (colour index + level): the byte offset of the pixel's merged cell
                    float worst = 0.0f;   // largest fractional part of the batch, either scale
                    // written stage by stage: the four chains are independent, and every step of a chain waits on the one before
                    float l2[EB];
#pragma unroll
                    for (int k = 0; k < EB; k++) abs2[k] = re[EB * q + k] * re[EB * q + k] + im[EB * q + k] * im[EB * q + k];   // worker.js:92
#pragma unroll
                    for (int k = 0; k < EB; k++) l2[k] = (float)abs2[k];
#pragma unroll
                    for (int k = 0; k < EB; k++) l2[k] = __log2f(l2[k]);
#pragma unroll
                    for (int k = 0; k < EB; k++) {
                        mn4[k & 3] = min_raw(mn4[k & 3], abs2[k]);
                        mx4[k & 3] = max_raw(mx4[k & 3], abs2[k]);
                    }
                    if constexpr (PK_SCALES) {
                        // one v_pk_fma_f32 per scale for the batch's two bins (4.7 issue cycles instead of 2 x 3.5, one instruction
                        // fewer per bin); each half rounds like v_fma_f32
                        static_assert(EB == 2, "a packed fma takes the batch's two bins");
                        const f32x2 lp = {l2[0], l2[1]};
                        const f32x2 tgp = __builtin_elementwise_fma(g_b2, lp, g_a2), tcp = __builtin_elementwise_fma(c_b2, lp, c_a2);
                        tg[0] = tgp.x; tg[1] = tgp.y;
                        tc[0] = tcp.x; tc[1] = tcp.y;
                    } else {
#pragma unroll
                        for (int k = 0; k < EB; k++) {
                            tg[k] = fmaf(g_b, l2[k], g_a_v);
                            tc[k] = fmaf(c_b, l2[k], c_a_v);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < EB; k++) {
                        tg[k] = __builtin_amdgcn_fmed3f(tg[k], g_lo, g_hi);
                        tc[k] = __builtin_amdgcn_fmed3f(tc[k], c_lo, c_hi_v);
                    }
#pragma unroll
                    for (int k = 0; k < EB; k++) {
                        gi[k] = floor_to_int(tg[k]);                                   // colour index
                        cell[k] = floor_to_int(tc[k]);                                 // level (= 999 - centi-bel bin)
                    }
#pragma unroll
                    for (int k = 0; k < EB; k++) {
                        // (the clamps have turned a NaN into a bound, so the fractional parts are numbers; one threshold, the
                        // smaller of the two, serves both scales)
                        worst = fmaxf(fmaxf(worst, __builtin_amdgcn_fractf(tg[k])), __builtin_amdgcn_fractf(tc[k]));   // one v_max3_f32
                    }
                    if (__builtin_expect(__ballot(!(worst < thr)) != 0ull, 0)) {
                        // Rare (one batch in eleven), and nearly always for ONE lane on ONE bin and ONE scale: each (bin, scale) is
                        // decided on its own - the nearest edge of that scale for every lane (one LDS read, one exact comparison),
                        // only the risky lanes keep the result (edges: sp_host.h Thresholds) - so a typical visit costs a quarter
                        // of deciding everything for the whole batch.
#pragma unroll
                        for (int k = 0; k < EB; k++) {
                            const bool rgk = !(__builtin_amdgcn_fractf(tg[k]) < thr), rck = !(__builtin_amdgcn_fractf(tc[k]) < thr);
                            int lev = cell[k];
                            if (__ballot(rgk) != 0ull) {
                                const int r = min(max((int)rintf(tg[k] + g_m), 1), cmax);
                                const int g = abs2[k] >= edge_g[r] ? r : r - 1;
                                gi[k] = rgk ? g : gi[k];
                            }
                            if (__ballot(rck) != 0ull) {
                                const int r = min(max((int)rintf(tc[k] + c_m), 1), SP_CB_HIST_SIZE);
                                const int l = abs2[k] >= edge_cb[r] ? r : r - 1;
                                // -inf / NaN dB: colour 0; +inf dB: last colour; all three: ToInt32 gives key 0 = bin 0      worker.js:105,111
                                // (the clamp bounds of the level scale are risky by construction, so these lanes always come here)
                                // (their cells lie behind the regular ones: the level is set so that colour index + level names them)
                                const bool dark = !(abs2[k] > 0.0), bright = abs2[k] == spjs::inf();
                                gi[k] = rck && dark ? 0 : gi[k];
                                lev = rck ? (dark ? cell_sp0 : bright ? cell_sp0 + 1 - gi[k] : l) : lev;
                            }
                            cell[k] = lev;
                        }
                    }
                    // the merged cell's byte offset, 4 * (colour index + level), as ONE instruction (left to the compiler it becomes an
                    // add on one side of the branch above and a shift on the other)
#pragma unroll
                    for (int k = 0; k < EB; k++) asm("v_add_lshl_u32 %0, %1, %2, 2" : "=v"(cell4[k]) : "v"(gi[k]), "v"(cell[k]));
#pragma unroll
                    for (int k = 0; k < EB; k++) {
                        const int e = EB * q + k;                 // compile-time after unrolling
                        if constexpr (TILE_BYTES) {
                            // one ds_write_b8 per bin (base + immediate): packing four indices into a dword first costs three
                            // v_lshl_or_b32 per dword, and every VALU instruction costs what an f64 operation costs; the LDS pipe has room
                            asm volatile("ds_write_b8 %0, %1 offset:%2" ::"v"(trow_addr), "v"(gi[k]), "n"(e) : "memory");
                        } else {
                            tile_word = (e & 3) == 0 ? (uint32_t)gi[k] : tile_word | ((uint32_t)gi[k] << (8 * (e & 3)));
                            if ((e & 3) == 3) trow[e >> 2] = tile_word;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < EB; k++) {
                        lds_count(kOffCells, cell4[k]);
                    }
                }
                const double mn = min_raw(min_raw(mn4[0], mn4[1]), min_raw(mn4[2], mn4[3]));
                const double mx = max_raw(max_raw(mx4[0], mx4[1]), max_raw(mx4[2], mx4[3]));
                unsigned long long *slot = s_mm + 2 * (((LATE_SIDE ? gpar : 0) * group_frames + fr) * MMS + (tl & (MMS - 1)));
                atomicMin(slot, (unsigned long long)__double_as_longlong(mn));
                atomicMax(slot + 1, (unsigned long long)__double_as_longlong(mx));
            }
            // Workgroup 0's first wave publishes the request's number once its clearing stores have landed: after its first frame (group 0
            // is workgroup 0's, and every slot has a frame in a group's first round), when they long have.
            if (g == 0 && r == 0 && a.first && __builtin_amdgcn_readfirstlane(tid >> 6) == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const LateArgs la = late_args();
                if (tid == 0) __hip_atomic_store(la->flag, la->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        drain_x0 = x0;
        gpar ^= 1;
    }

    // ---- end of the workgroup's frames: last write-out and side outputs, the workgroup's share of histograms and dBfs range ----------
    const LateArgs la = late_args();
    // requested now, used behind the last barrier: the cell ranges of this thread's histogram outputs and the request's number as
    // workgroup 0 published it
    const uint16_t *const cell_g = la->cell_g, *const cell_l = la->cell_l;
    const int gi_c = tid < a.lut_len ? tid : 0;
    const int cg_lo = cell_g[gi_c], cg_hi = cell_g[gi_c + 1];
    int l_lo[2], l_hi[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int gi = tid + u * kThreads;
        const int l_cb = gi < SP_CB_HIST_SIZE ? SP_CB_HIST_SIZE - 1 - gi : 0;              // bin gi counts level 999 - gi
        l_lo[u] = cell_l[l_cb];
        l_hi[u] = cell_l[l_cb + 1];
    }
    const unsigned int seen = __hip_atomic_load(la->flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (HALVES && drain_x0 >= 0 && a.rgba) {
        // The workgroup's last write-out overlaps nothing.  The first waves of the SIMDs reach it ~7 us before the second ones (config 2;
        // issue arbitration favours the older wave, s_setprio does not change that - tools/stamps.py) and would wait at the barrier:
        // each set of four waves meets by itself and writes its own half of the group, 64-byte pieces of the image rows, so half of the
        // chip's last stores are under way while the second waves still compute.
        const int half = tid >> 8;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(&s_done[half], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(&s_done[half], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u) __builtin_amdgcn_s_sleep(2);
        // (non-temporal: left in L2, the 64-byte pieces are written back when the kernel ends, +1.2 us instead of -1.1 us; the first
        // set also taking half of the second set's rows once those are ready: no further gain)
        drain_rows(drain_x0, 0, 1, half * (group_frames / 2), group_frames / 2, half * (kThreads / 2), kThreads / 2, true);
    }
    lds_barrier();
    if (drain_x0 >= 0 && !(HALVES && a.rgba)) drain(drain_x0, 0, 1);
    // ---- the workgroup's share of the request's histograms and dBfs range (worker.js:105-113, 124-125, 140-155) ----------------------
    // No workgroup finishes for the others (that costs the last one three dependent trips to memory, 6 us): every workgroup turns its
    // own merged cells into histogram counts and adds them to the reply itself, with fire-and-forget atomics the launch's end waits
    // for anyway.  Workgroup 0 has zeroed the reply's histograms and set its dBfs range to (0, -200) at the start of the request's
    // first launch and published the request's number behind that (below); everybody checks the number before its first add.
    {
        // cells -> prefix sums: every count is a difference of two prefix sums over the cells (sp_host.h Thresholds).  A thread takes
        // kPer consecutive cells, the workgroup scans the 512 partial sums (in each wave with shuffles, the eight wave totals through
        // LDS).  The exchange buffers are idle by now and hold the prefix.  (Counts of one workgroup fit 32 bits, as s_cells does.)
        constexpr int kPer = 3;
        static_assert(kThreads * kPer >= kMaxCells, "every cell needs a thread");
        unsigned int *const s_pre = (unsigned int *)(smem + kOffXch);         // [kThreads * kPer + 1]: s_pre[c] = sum of the cells [0, c)
        unsigned int *const s_part = s_pre + kThreads * kPer + 4;             // [kThreads / 64] wave totals
        unsigned int v[kPer], run = 0;
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            const int c = tid * kPer + k;
            v[k] = c < a.cells ? s_cells[c] : 0u;
            run += v[k];
        }
        const unsigned int incl = wave_scan_u32(run);
        if (lane == 63) s_part[tid >> 6] = incl;
        lds_barrier();
        unsigned int base = incl - run;                                       // sum of the cells below this thread's first
        {
            const uint4 p0 = *(const uint4 *)s_part, p1 = *(const uint4 *)(s_part + 4);   // (one batch of reads, not one per wave below)
            const unsigned int part[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
            const int wave = tid >> 6;
#pragma unroll
            for (int w = 0; w < 7; w++) base += w < wave ? part[w] : 0u;
        }
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            s_pre[tid * kPer + k] = base;
            base += v[k];
        }
        if (tid == kThreads - 1) s_pre[kThreads * kPer] = base;
        lds_barrier();
        if (seen != la->seq) {
            // (never in practice: workgroup 0 - dispatched first: the lowest workgroup number - published the number tens of microseconds
            // ago.  The wait is bounded: ~2 s of polling end in a trap, i.e. a failed launch, instead of a hung device.)
            unsigned polls = 0;
            while (__hip_atomic_load(la->flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != la->seq) {
                __builtin_amdgcn_s_sleep(32);
                if (++polls > (1u << 22)) __builtin_trap();
            }
        }
        const int sp0 = a.cells - 2, sp1 = a.cells - 1;                       // -inf / NaN dB (colour 0, bin 0); +inf dB (last colour, bin 0)
        const unsigned int n0 = s_pre[sp0 + 1] - s_pre[sp0], n1 = s_pre[sp1 + 1] - s_pre[sp1];
        unsigned long long *const out_c = la->out_c, *const out_cb = la->out_cb;
        if (tid < a.lut_len && out_c) {
            const unsigned int cnt = s_pre[cg_hi] - s_pre[cg_lo] + (tid == 0 ? n0 : 0u) + (tid == a.lut_len - 1 ? n1 : 0u);
            if (cnt) atomicAdd(&out_c[tid], (unsigned long long)cnt);
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int gi = tid + u * kThreads;
            if (gi < SP_CB_HIST_SIZE && out_cb) {
                const unsigned int cnt = s_pre[l_hi[u]] - s_pre[l_lo[u]] + (gi == 0 ? n0 + n1 : 0u);
                if (cnt) atomicAdd(&out_cb[gi], (unsigned long long)cnt);
            }
        }
        // The last group's gauges come behind the adds (two waves, one software log10: ~1 us during which everybody's adds and stores
        // are on their way), and behind them the workgroup's share of the dBfs range.
        if (drain_x0 >= 0) side_outputs(drain_x0, gpar ^ 1);
        lds_barrier();
        double *const out_mm = la->out_minmax;
        if (tid < 2 && out_mm) {
            typedef __attribute__((address_space(1))) double *GlobalF64;
            if (tid == 0) __builtin_amdgcn_global_atomic_fmin_f64((GlobalF64)&out_mm[0], s_red[0]);
            else __builtin_amdgcn_global_atomic_fmax_f64((GlobalF64)&out_mm[1], s_red[1]);
        }
    }
}

// Per-n launchers, one translation unit each (sp_inst_frames.hip is compiled once per LOG2N).
template <int L>
int launch_frames_n(const FrameArgs &a, int format, const double2 *stage_tw, int grid, int lds_bytes, int gf, int groups, int prefetch,
                    int device, hipStream_t stream);
#define SP_DECL(L)                                                                                                              \
    template <>                                                                                                                 \
    int launch_frames_n<L>(const FrameArgs &, int, const double2 *, int, int, int, int, int, int, hipStream_t);
SP_DECL(6) SP_DECL(7) SP_DECL(8) SP_DECL(9) SP_DECL(10) SP_DECL(11) SP_DECL(12) SP_DECL(13)
#undef SP_DECL

#ifdef SP_INST_FRAMES_LOG2N
// per-device, per-variant opt-in to the full LDS (function attributes belong to the device's code object)
template <int L, bool C, int P>
inline int launch_variant(const FrameArgs &a, int format, const double2 *stage_tw, int grid, int lds_bytes, int gf, int groups, int device,
                          hipStream_t stream)
{
    // (contexts of several devices render on different threads: the flags are atomic, and setting the attribute twice is harmless)
    static std::atomic<bool> attr_set[kMaxDevices];
    if (device < 0 || device >= kMaxDevices || !attr_set[device].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void *)k_frames<L, C, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return SP_ERR_HIP;
        if (device >= 0 && device < kMaxDevices) attr_set[device].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_frames<L, C, P>), dim3((unsigned)grid), dim3(kFrameThreads), (size_t)lds_bytes, stream, a, format, stage_tw, gf, groups);
    return SP_OK;
}

template <>
int launch_frames_n<SP_INST_FRAMES_LOG2N>(const FrameArgs &a, int format, const double2 *stage_tw, int grid, int lds_bytes, int gf, int groups,
                                   int prefetch, int device, hipStream_t stream)
{
    constexpr int L = SP_INST_FRAMES_LOG2N;
#define SP_V(C, P) return launch_variant<L, C, P>(a, format, stage_tw, grid, lds_bytes, gf, groups, device, stream);
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
inline int launch_frames(const FrameArgs &a, int format, const double2 *stage_tw, int cu_count, int device, hipStream_t stream)
{
    int prefetch = (a.in_bounds && (a.sample_width <= 4 || a.sample_width == 8)) ? a.sample_width : 0;
    if (prefetch == 3 && !(a.width >= 2 && frame_start(a.stride, a.width - 1) >= 1)) prefetch = 0;
    if (!frames_kernel_supports(a.n) || a.lut_len > kLdsMaxLut || a.lut_len < 2) return SP_ERR_UNSUPPORTED;
    const int n = a.n;
    int want = 32;
    while (want > 4 && (a.x_end - a.frame0 + want - 1) / want < 2 * cu_count) want >>= 1;
    const int gf = group_frames_for(n, want);
    if (gf & (gf - 1)) return SP_ERR_UNSUPPORTED;   // (the write-out splits item numbers with shifts; every n in range gives a power of two)
    const int groups = (a.x_end - a.frame0 + gf - 1) / gf;
    const Layout lay = layout(n, a.lut_len, gf);
    if (lay.total > 160 * 1024) return SP_ERR_UNSUPPORTED;
    int grid = groups < cu_count ? groups : cu_count;
    grid = (grid + 7) & ~7;
    switch (a.levels) {
#define SP_L(L) case L: return launch_frames_n<L>(a, format, stage_tw, grid, lay.total, gf, groups, prefetch, device, stream);
        SP_L(6) SP_L(7) SP_L(8) SP_L(9) SP_L(10) SP_L(11) SP_L(12) SP_L(13)
#undef SP_L
    default: return SP_ERR_UNSUPPORTED;
    }
}

}  // namespace spk2
