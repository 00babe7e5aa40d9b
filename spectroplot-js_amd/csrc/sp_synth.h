// sp_synth.h — device-side synthetic I/Q ("trinoise": gated triangle-wave tone + hash noise) for benchmarks.
// Bit-identical to tests/siggen.py and oracle/js/siggen.js, so any frame of a device-generated capture can be
// regenerated on the CPU and checked against the oracle.  Definition: DESIGN.md "Synthetic input".
#pragma once

#include <hip/hip_runtime.h>

#include "sp_formats.h"

namespace spk {

struct SynthArgs {
    uint8_t *out;
    uint64_t t0, count;
    uint32_t seed, step, gshift;
    double amp, namp;
};

__device__ inline uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

__device__ inline double synth_value(const SynthArgs &a, uint64_t t, uint32_t c)
{
    uint32_t ph = (((uint32_t)t & 0xffffu) * a.step) & 0xffffu;
    if (c) ph = (ph - 16384u) & 0xffffu;
    const int32_t d = (int32_t)ph - 32768;
    const double tri = (double)((d < 0 ? -d : d) - 16384) / 16384.0;
    const double gate = (((uint32_t)t >> a.gshift) & 1) ? 1.0 : 0.25;   // JS `t >>> gshift` is a uint32 shift
    const double u = (double)fmix32(a.seed ^ (uint32_t)(2 * t + c)) / 4294967296.0 - 0.5;
    return (a.amp * gate) * tri + a.namp * u;
}

__device__ inline double quant(double v, double mul, double add, double lo, double hi)
{
    const double q = floor(v * mul + add);
    return q < lo ? lo : q > hi ? hi : q;
}

template <int FMT>
__global__ __launch_bounds__(256) void k_synth_trinoise(const SynthArgs a)
{
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= a.count) return;
    const uint64_t t = a.t0 + k;
    const double vi = synth_value(a, t, 0), vq = synth_value(a, t, 1);
    uint8_t *o = a.out + k * (uint64_t)spfmt::describe(FMT).width;
    if constexpr (FMT == SP_FMT_CF32) {
        *(float2 *)o = make_float2((float)vi, (float)vq);
    } else if constexpr (FMT == SP_FMT_CF64) {
        *(double2 *)o = make_double2(vi, vq);
    } else if constexpr (FMT == SP_FMT_CS16) {
        const int32_t i = (int32_t)quant(vi, 32767, 0.5, -32768, 32767), q = (int32_t)quant(vq, 32767, 0.5, -32768, 32767);
        *(uint32_t *)o = ((uint32_t)i & 0xffffu) | ((uint32_t)q << 16);
    } else if constexpr (FMT == SP_FMT_CU16) {
        const uint32_t i = (uint32_t)quant(vi, 32767.5, 32768, 0, 65535), q = (uint32_t)quant(vq, 32767.5, 32768, 0, 65535);
        *(uint32_t *)o = i | (q << 16);
    } else if constexpr (FMT == SP_FMT_CS8) {
        const int32_t i = (int32_t)quant(vi, 127, 0.5, -128, 127), q = (int32_t)quant(vq, 127, 0.5, -128, 127);
        *(uint16_t *)o = (uint16_t)(((uint32_t)i & 0xffu) | (((uint32_t)q & 0xffu) << 8));
    } else if constexpr (FMT == SP_FMT_CU8) {
        const uint32_t i = (uint32_t)quant(vi, 127.5, 128, 0, 255), q = (uint32_t)quant(vq, 127.5, 128, 0, 255);
        *(uint16_t *)o = (uint16_t)(i | (q << 8));
    } else if constexpr (FMT == SP_FMT_CS32) {
        const int32_t i = (int32_t)quant(vi, 2147483647, 0.5, -2147483648.0, 2147483647);
        const int32_t q = (int32_t)quant(vq, 2147483647, 0.5, -2147483648.0, 2147483647);
        *(uint2 *)o = make_uint2((uint32_t)i, (uint32_t)q);
    } else if constexpr (FMT == SP_FMT_CU32) {
        const uint32_t i = (uint32_t)quant(vi, 2147483647.5, 2147483648.0, 0, 4294967295.0);
        const uint32_t q = (uint32_t)quant(vq, 2147483647.5, 2147483648.0, 0, 4294967295.0);
        *(uint2 *)o = make_uint2(i, q);
    } else if constexpr (FMT == SP_FMT_CS64 || FMT == SP_FMT_CU64) {
        uint32_t hi_i, hi_q;
        if constexpr (FMT == SP_FMT_CS64) {
            hi_i = (uint32_t)(int32_t)quant(vi, 2147483647, 0.5, -2147483648.0, 2147483647);
            hi_q = (uint32_t)(int32_t)quant(vq, 2147483647, 0.5, -2147483648.0, 2147483647);
        } else {
            hi_i = (uint32_t)quant(vi, 2147483647.5, 2147483648.0, 0, 4294967295.0);
            hi_q = (uint32_t)quant(vq, 2147483647.5, 2147483648.0, 0, 4294967295.0);
        }
        const uint32_t s2 = a.seed ^ 0x10101010u;
        *(uint4 *)o = make_uint4(fmix32(s2 ^ (uint32_t)(2 * t)), hi_i, fmix32(s2 ^ (uint32_t)(2 * t + 1)), hi_q);
    } else if constexpr (FMT == SP_FMT_CS12 || FMT == SP_FMT_CU12) {
        uint32_t i12, q12;
        if constexpr (FMT == SP_FMT_CS12) {
            i12 = (uint32_t)(int32_t)quant(vi, 2047, 0.5, -2048, 2047) & 0xfffu;
            q12 = (uint32_t)(int32_t)quant(vq, 2047, 0.5, -2048, 2047) & 0xfffu;
        } else {
            i12 = (uint32_t)quant(vi, 2047.5, 2048, 0, 4095);
            q12 = (uint32_t)quant(vq, 2047.5, 2048, 0, 4095);
        }
        o[0] = (uint8_t)(i12 & 0xff);
        o[1] = (uint8_t)(((i12 >> 8) & 0x0f) | ((q12 & 0x0f) << 4));
        o[2] = (uint8_t)(q12 >> 4);
    } else {   // CS4 / CU4
        uint32_t i4, q4;
        if constexpr (FMT == SP_FMT_CS4) {
            i4 = (uint32_t)(int32_t)quant(vi, 7, 0.5, -8, 7) & 0xfu;
            q4 = (uint32_t)(int32_t)quant(vq, 7, 0.5, -8, 7) & 0xfu;
        } else {
            i4 = (uint32_t)quant(vi, 7.5, 8, 0, 15);
            q4 = (uint32_t)quant(vq, 7.5, 8, 0, 15);
        }
        o[0] = (uint8_t)((i4 << 4) | q4);
    }
}

}  // namespace spk
