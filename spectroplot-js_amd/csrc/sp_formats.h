// sp_formats.h — sample formats of the reference's SampleView (lib/samples.js:15-169) for host and device.
//
// A format is {typed-view element, bias, scale, bytes per complex sample}; six formats need unpacking
// (lib/samples.js:313-390).  Values are produced exactly as the reference does: element -> f64, subtract the bias,
// multiply by the scale (two roundings, never fused).  Reads outside the buffer follow JavaScript's `undefined`
// coercions: NaN through arithmetic, 0 through bitwise operators.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>

#include "../../include/spectroplot_hip.h"
#include "sp_jsmath.h"

namespace spfmt {

struct Format {
    int32_t id;
    int32_t width;   // bytes per complex sample
    int32_t elem;    // element size of the typed view the reference constructs
    double bias;
    double scale;
};

SP_HD inline Format describe(int32_t id)
{
    switch (id) {
    case SP_FMT_CU4: return {id, 1, 1, 7.5, 1.0 / 7.5};
    case SP_FMT_CS4: return {id, 1, 1, 0.0, 1.0 / 8.0};
    case SP_FMT_CS8: return {id, 2, 1, 0.0, 1.0 / 128.0};
    case SP_FMT_CU12: return {id, 3, 1, 2047.5, 1.0 / 2047.5};
    case SP_FMT_CS12: return {id, 3, 1, 0.0, 1.0 / 2048.0};
    case SP_FMT_CU16: return {id, 4, 2, 32767.5, 1.0 / 32768.0};
    case SP_FMT_CS16: return {id, 4, 2, 0.0, 1.0 / 32768.0};
    case SP_FMT_CU32: return {id, 8, 4, 2147483647.5, 1.0 / 2147483648.0};
    case SP_FMT_CS32: return {id, 8, 4, 0.0, 1.0 / 2147483648.0};
    case SP_FMT_CU64: return {id, 16, 4, 1.0, 1.0};
    case SP_FMT_CS64: return {id, 16, 4, 0.0, 1.0};
    case SP_FMT_CF32: return {id, 8, 4, 0.0, 1.0};
    case SP_FMT_CF64: return {id, 16, 8, 0.0, 1.0};
    default: return {SP_FMT_CU8, 2, 1, 127.5, 1.0 / 127.5};
    }
}

struct View {
    const uint8_t *p;
    int64_t nbytes;
    int64_t nelem;   // length of the typed view
};

template <typename T>
SP_HD inline T load_as(const uint8_t *q)
{
    T t;
    memcpy(&t, q, sizeof t);
    return t;
}

// Component c (0 = I, 1 = Q) of complex sample `pos`, with the reference's out-of-range behaviour.
template <int FMT>
SP_HD inline double sample_checked(const View &v, int64_t pos, int c)
{
    const Format f = describe(FMT);
    if constexpr (FMT == SP_FMT_CU4 || FMT == SP_FMT_CS4) {
        const int32_t b = (pos >= 0 && pos < v.nbytes) ? v.p[pos] : 0;
        if constexpr (FMT == SP_FMT_CU4) return ((double)(c ? (b & 15) : (b >> 4)) - f.bias) * f.scale;
        const int32_t s = c ? (int32_t)((uint32_t)b << 28) >> 28 : (int32_t)((uint32_t)(b & 0xf0) << 24) >> 28;
        return (double)s * f.scale;
    } else if constexpr (FMT == SP_FMT_CU12 || FMT == SP_FMT_CS12) {
        const int64_t o = 3 * pos;
        const int32_t b0 = (o >= 0 && o < v.nbytes) ? v.p[o] : 0;
        const int32_t b1 = (o + 1 >= 0 && o + 1 < v.nbytes) ? v.p[o + 1] : 0;
        const int32_t b2 = (o + 2 >= 0 && o + 2 < v.nbytes) ? v.p[o + 2] : 0;
        if constexpr (FMT == SP_FMT_CU12) {
            const int32_t s = c ? ((b2 << 4) | (b1 >> 4)) : (((b1 & 15) << 8) | b0);
            return ((double)s - f.bias) * f.scale;
        }
        const int32_t s = c ? (int32_t)(((uint32_t)b2 << 24) | ((uint32_t)(b1 & 0xf0) << 16)) >> 20
                            : (int32_t)(((uint32_t)(b1 & 15) << 28) | ((uint32_t)b0 << 20)) >> 20;
        return (double)s * f.scale;
    } else if constexpr (FMT == SP_FMT_CU64 || FMT == SP_FMT_CS64) {
        const int64_t ilo = 4 * pos + 2 * c, ihi = ilo + 1;
        double lo = spjs::qnan(), hi;
        if (ilo >= 0 && ilo < v.nelem) lo = (double)load_as<uint32_t>(v.p + 4 * ilo);
        if (ihi >= 0 && ihi < v.nelem) {
            const uint32_t w = load_as<uint32_t>(v.p + 4 * ihi);
            hi = FMT == SP_FMT_CS64 ? (double)(int32_t)w : (double)w;
        } else {
            hi = FMT == SP_FMT_CS64 ? 0.0 : spjs::qnan();
        }
        const double s = hi / 2147483648.0 + lo / 18446744073709551616.0;
        return FMT == SP_FMT_CS64 ? s : s - f.bias;
    } else {
        const int64_t i = 2 * pos + c;
        if (i < 0 || i >= v.nelem) return spjs::qnan();
        const uint8_t *q = v.p + i * f.elem;
        double e;
        if constexpr (FMT == SP_FMT_CU8) e = (double)q[0];
        else if constexpr (FMT == SP_FMT_CS8) e = (double)(int8_t)q[0];
        else if constexpr (FMT == SP_FMT_CU16) e = (double)load_as<uint16_t>(q);
        else if constexpr (FMT == SP_FMT_CS16) e = (double)load_as<int16_t>(q);
        else if constexpr (FMT == SP_FMT_CU32) e = (double)load_as<uint32_t>(q);
        else if constexpr (FMT == SP_FMT_CS32) e = (double)load_as<int32_t>(q);
        else if constexpr (FMT == SP_FMT_CF32) e = (double)load_as<float>(q);
        else e = load_as<double>(q);
        if constexpr (FMT == SP_FMT_CF32 || FMT == SP_FMT_CF64) return e;   // (e - 0) * 1 == e for every e
        return (e - f.bias) * f.scale;
    }
}

// Both components of an in-range sample (the caller guarantees 0 <= pos and (pos+1)*width <= nbytes).
template <int FMT>
SP_HD inline void sample_fast(const uint8_t *p, int64_t pos, double &vi, double &vq)
{
    const Format f = describe(FMT);
    if constexpr (FMT == SP_FMT_CU4) {
        const int32_t b = p[pos];
        vi = ((double)(b >> 4) - f.bias) * f.scale;
        vq = ((double)(b & 15) - f.bias) * f.scale;
    } else if constexpr (FMT == SP_FMT_CS4) {
        const int32_t b = p[pos];
        vi = (double)((int32_t)((uint32_t)(b & 0xf0) << 24) >> 28) * f.scale;
        vq = (double)((int32_t)((uint32_t)b << 28) >> 28) * f.scale;
    } else if constexpr (FMT == SP_FMT_CU12 || FMT == SP_FMT_CS12) {
        const uint8_t *q = p + 3 * pos;
        const int32_t b0 = q[0], b1 = q[1], b2 = q[2];
        if constexpr (FMT == SP_FMT_CU12) {
            vi = ((double)(((b1 & 15) << 8) | b0) - f.bias) * f.scale;
            vq = ((double)((b2 << 4) | (b1 >> 4)) - f.bias) * f.scale;
        } else {
            vi = (double)((int32_t)(((uint32_t)(b1 & 15) << 28) | ((uint32_t)b0 << 20)) >> 20) * f.scale;
            vq = (double)((int32_t)(((uint32_t)b2 << 24) | ((uint32_t)(b1 & 0xf0) << 16)) >> 20) * f.scale;
        }
    } else if constexpr (FMT == SP_FMT_CU64 || FMT == SP_FMT_CS64) {
        const uint8_t *q = p + 16 * pos;
        const uint32_t w0 = load_as<uint32_t>(q), w1 = load_as<uint32_t>(q + 4), w2 = load_as<uint32_t>(q + 8),
                       w3 = load_as<uint32_t>(q + 12);
        if constexpr (FMT == SP_FMT_CS64) {
            vi = (double)(int32_t)w1 / 2147483648.0 + (double)w0 / 18446744073709551616.0;
            vq = (double)(int32_t)w3 / 2147483648.0 + (double)w2 / 18446744073709551616.0;
        } else {
            vi = ((double)w1 / 2147483648.0 + (double)w0 / 18446744073709551616.0) - f.bias;
            vq = ((double)w3 / 2147483648.0 + (double)w2 / 18446744073709551616.0) - f.bias;
        }
    } else if constexpr (FMT == SP_FMT_CU8) {
        const uint32_t w = load_as<uint16_t>(p + 2 * pos);
        vi = ((double)(w & 0xff) - f.bias) * f.scale;
        vq = ((double)(w >> 8) - f.bias) * f.scale;
    } else if constexpr (FMT == SP_FMT_CS8) {
        const uint32_t w = load_as<uint16_t>(p + 2 * pos);
        vi = (double)(int8_t)(w & 0xff) * f.scale;
        vq = (double)(int8_t)(w >> 8) * f.scale;
    } else if constexpr (FMT == SP_FMT_CU16) {
        const uint32_t w = load_as<uint32_t>(p + 4 * pos);
        vi = ((double)(w & 0xffff) - f.bias) * f.scale;
        vq = ((double)(w >> 16) - f.bias) * f.scale;
    } else if constexpr (FMT == SP_FMT_CS16) {
        const uint32_t w = load_as<uint32_t>(p + 4 * pos);
        vi = (double)(int16_t)(w & 0xffff) * f.scale;
        vq = (double)(int16_t)(w >> 16) * f.scale;
    } else if constexpr (FMT == SP_FMT_CU32) {
        const uint64_t w = load_as<uint64_t>(p + 8 * pos);
        vi = ((double)(uint32_t)w - f.bias) * f.scale;
        vq = ((double)(uint32_t)(w >> 32) - f.bias) * f.scale;
    } else if constexpr (FMT == SP_FMT_CS32) {
        const uint64_t w = load_as<uint64_t>(p + 8 * pos);
        vi = (double)(int32_t)(uint32_t)w * f.scale;
        vq = (double)(int32_t)(uint32_t)(w >> 32) * f.scale;
    } else if constexpr (FMT == SP_FMT_CF32) {
        struct F2 { float x, y; };
        const F2 w = load_as<F2>(p + 8 * pos);
        vi = (double)w.x;
        vq = (double)w.y;
    } else {
        struct D2 { double x, y; };
        const D2 w = load_as<D2>(p + 16 * pos);
        vi = w.x;
        vq = w.y;
    }
}

// Both components from the raw little-endian words of one sample that is already in registers (1-, 2-, 3-, 4- and 8-byte samples:
// lo = the first four bytes, hi = the next four).  Same arithmetic as sample_fast.
template <int FMT>
SP_HD inline void decode_raw(uint32_t lo, uint32_t hi, double &vi, double &vq)
{
    const Format f = describe(FMT);
    if constexpr (FMT == SP_FMT_CU8) {
        vi = ((double)(lo & 0xff) - f.bias) * f.scale;
        vq = ((double)((lo >> 8) & 0xff) - f.bias) * f.scale;
    } else if constexpr (FMT == SP_FMT_CS8) {
        vi = (double)(int8_t)(lo & 0xff) * f.scale;
        vq = (double)(int8_t)((lo >> 8) & 0xff) * f.scale;
    } else if constexpr (FMT == SP_FMT_CU16) {
        vi = ((double)(lo & 0xffff) - f.bias) * f.scale;
        vq = ((double)(lo >> 16) - f.bias) * f.scale;
    } else if constexpr (FMT == SP_FMT_CS16) {
        vi = (double)(int16_t)(lo & 0xffff) * f.scale;
        vq = (double)(int16_t)(lo >> 16) * f.scale;
    } else if constexpr (FMT == SP_FMT_CU4) {
        const int32_t b = lo & 0xff;
        vi = ((double)(b >> 4) - f.bias) * f.scale;
        vq = ((double)(b & 15) - f.bias) * f.scale;
    } else if constexpr (FMT == SP_FMT_CS4) {
        const int32_t b = lo & 0xff;
        vi = (double)((int32_t)((uint32_t)(b & 0xf0) << 24) >> 28) * f.scale;
        vq = (double)((int32_t)((uint32_t)b << 28) >> 28) * f.scale;
    } else if constexpr (FMT == SP_FMT_CU12 || FMT == SP_FMT_CS12) {
        const int32_t b0 = lo & 0xff, b1 = (lo >> 8) & 0xff, b2 = (lo >> 16) & 0xff;   // the sample's three bytes
        if constexpr (FMT == SP_FMT_CU12) {
            vi = ((double)(((b1 & 15) << 8) | b0) - f.bias) * f.scale;
            vq = ((double)((b2 << 4) | (b1 >> 4)) - f.bias) * f.scale;
        } else {
            vi = (double)((int32_t)(((uint32_t)(b1 & 15) << 28) | ((uint32_t)b0 << 20)) >> 20) * f.scale;
            vq = (double)((int32_t)(((uint32_t)b2 << 24) | ((uint32_t)(b1 & 0xf0) << 16)) >> 20) * f.scale;
        }
    } else if constexpr (FMT == SP_FMT_CU32) {
        vi = ((double)lo - f.bias) * f.scale;
        vq = ((double)hi - f.bias) * f.scale;
    } else if constexpr (FMT == SP_FMT_CS32) {
        vi = (double)(int32_t)lo * f.scale;
        vq = (double)(int32_t)hi * f.scale;
    } else {
        static_assert(FMT == SP_FMT_CF32, "decode_raw covers the 1-, 2-, 3-, 4- and 8-byte sample formats");
        float a, b;
        memcpy(&a, &lo, 4);
        memcpy(&b, &hi, 4);
        vi = (double)a;
        vq = (double)b;
    }
}

}  // namespace spfmt
