// sp_jsmath.h — the three engine intrinsics the reference's results depend on, for host and device code.
//
// The reference evaluates Math.log10 per pixel (lib/worker.js:93), Math.cos/Math.sin for the twiddle tables
// (lib/fft_nayuki.js:45-46) and Math.cos for the tapers (lib/windows.js).  On V8 these are the fdlibm 5.3 routines
// (e_log.c, e_log10.c, k_cos.c, k_sin.c, e_rem_pio2.c); they are restated here so that tables and thresholds built
// by this library are bit-identical to what the reference computes under Node.  Everything must be compiled with
// -ffp-contract=off: each operation rounds on its own.
//
// The routines follow Sun Microsystems' fdlibm 5.3 (constants and evaluation order are the algorithm); its notice is kept here
// as it requires:
//
//   ====================================================
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//
//   Developed at SunSoft, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this
//   software is freely granted, provided that this notice
//   is preserved.
//   ====================================================
#pragma once

#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define SP_HD __host__ __device__
#else
#define SP_HD
#endif

namespace spjs {

SP_HD inline uint64_t bits(double x) { uint64_t u; memcpy(&u, &x, sizeof u); return u; }
SP_HD inline double from_bits(uint64_t u) { double x; memcpy(&x, &u, sizeof x); return x; }
SP_HD inline int32_t hi32(double x) { return (int32_t)(bits(x) >> 32); }
SP_HD inline uint32_t lo32(double x) { return (uint32_t)bits(x); }
SP_HD inline double make(int32_t hi, uint32_t lo) { return from_bits(((uint64_t)(uint32_t)hi << 32) | lo); }
SP_HD inline double inf() { return from_bits(0x7ff0000000000000ull); }
SP_HD inline double qnan() { return from_bits(0x7ff8000000000000ull); }

// natural log on a normalised, positive, finite argument path (fdlibm __ieee754_log)
SP_HD inline double log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double two54 = 1.80143985094819840000e+16;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    int32_t hx = hi32(x);
    const uint32_t lx = lo32(x);
    int32_t k = 0;
    if (hx < 0x00100000) {
        if (((hx & 0x7fffffff) | lx) == 0) return -inf();
        if (hx < 0) return qnan();
        k -= 54;
        x *= two54;
        hx = hi32(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    x = make(hx | (i ^ 0x3ff00000), lo32(x));
    k += (i >> 20);
    const double f = x - 1.0;
    if ((0x000fffff & (2 + hx)) < 3) {
        if (f == 0.0) {
            if (k == 0) return 0.0;
            const double dk = (double)k;
            return dk * ln2_hi + dk * ln2_lo;
        }
        const double R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        const double dk = (double)k;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    const double s = f / (2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    i = hx - 0x6147a;
    const double w = z * z;
    const int32_t j = 0x6b851 - hx;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    const double R = t2 + t1;
    if (i > 0) {
        const double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

// Math.log10
SP_HD inline double log10(double x)
{
    const double two54 = 1.80143985094819840000e+16, ivln10 = 4.34294481903251816668e-01;
    const double log10_2hi = 3.01029995663611771306e-01, log10_2lo = 3.69423907715893078616e-13;
    int32_t hx = hi32(x);
    uint32_t lx = lo32(x);
    int32_t k = 0;
    if (hx < 0x00100000) {
        if (((hx & 0x7fffffff) | lx) == 0) return -inf();
        if (hx < 0) return qnan();
        k -= 54;
        x *= two54;
        hx = hi32(x);
        lx = lo32(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    const int32_t i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
    hx = (hx & 0x000fffff) | ((0x3ff - i) << 20);
    const double y = (double)(k + i);
    const double z = y * log10_2lo + ivln10 * spjs::log(make(hx, lx));
    return z + y * log10_2hi;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// ---- trigonometry: host only (tables are built once per plan) -------------------------------------------

inline double kcos(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const int32_t ix = hi32(x) & 0x7fffffff;
    if (ix < 0x3e400000 && (int)x == 0) return 1.0;
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    if (ix < 0x3fd33333) return 1.0 - (0.5 * z - (z * r - x * y));
    const double qx = ix > 0x3fe90000 ? 0.28125 : make(ix - 0x00200000, 0);
    const double iz = 0.5 * z - qx;
    const double a = 1.0 - qx;
    return a - (iz - (z * r - x * y));
}

inline double ksin(double x, double y, bool have_tail)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const int32_t ix = hi32(x) & 0x7fffffff;
    if (ix < 0x3e400000 && (int)x == 0) return x;
    const double z = x * x;
    const double v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (!have_tail) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

// x mod pi/2 for |x| up to 2^19*pi/2 (the reference never exceeds 6*pi); returns the quadrant
inline int rem_pio2(double x, double y[2])
{
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                 pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
                 pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
                 pio2_3t = 8.47842766036889956997e-32;
    const int32_t hx = hi32(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix <= 0x3fe921fb) { y[0] = x; y[1] = 0; return 0; }
    if (ix < 0x4002d97c) {
        const double sgn = hx > 0 ? 1.0 : -1.0;
        double z = x - sgn * pio2_1;
        if (ix != 0x3ff921fb) {
            y[0] = z - sgn * pio2_1t;
            y[1] = (z - y[0]) - sgn * pio2_1t;
        } else {
            z -= sgn * pio2_2;
            y[0] = z - sgn * pio2_2t;
            y[1] = (z - y[0]) - sgn * pio2_2t;
        }
        return hx > 0 ? 1 : -1;
    }
    double t = x < 0 ? -x : x;
    const int32_t n = (int32_t)(t * invpio2 + 0.5);
    const double fn = (double)n;
    double r = t - fn * pio2_1;
    double w = fn * pio2_1t;
    // fdlibm consults a table of the high words of n*pi/2 (n <= 32); the product below has the same high word
    const bool near_multiple = !(n < 32 && ix != hi32((double)n * 1.57079632679489655800e+00));
    y[0] = r - w;
    if (near_multiple) {
        const int32_t j = ix >> 20;
        int32_t i = j - ((hi32(y[0]) >> 20) & 0x7ff);
        if (i > 16) {
            t = r;
            w = fn * pio2_2;
            r = t - w;
            w = fn * pio2_2t - ((t - r) - w);
            y[0] = r - w;
            i = j - ((hi32(y[0]) >> 20) & 0x7ff);
            if (i > 49) {
                t = r;
                w = fn * pio2_3;
                r = t - w;
                w = fn * pio2_3t - ((t - r) - w);
                y[0] = r - w;
            }
        }
    }
    y[1] = (r - y[0]) - w;
    if (hx < 0) { y[0] = -y[0]; y[1] = -y[1]; return -n; }
    return n;
}

inline double cos(double x)
{
    const int32_t ix = hi32(x) & 0x7fffffff;
    if (ix <= 0x3fe921fb) return kcos(x, 0.0);
    if (ix >= 0x7ff00000) return x - x;
    double y[2];
    switch (rem_pio2(x, y) & 3) {
    case 0: return kcos(y[0], y[1]);
    case 1: return -ksin(y[0], y[1], true);
    case 2: return -kcos(y[0], y[1]);
    default: return ksin(y[0], y[1], true);
    }
}

inline double sin(double x)
{
    const int32_t ix = hi32(x) & 0x7fffffff;
    if (ix <= 0x3fe921fb) return ksin(x, 0.0, false);
    if (ix >= 0x7ff00000) return x - x;
    double y[2];
    switch (rem_pio2(x, y) & 3) {
    case 0: return ksin(y[0], y[1], true);
    case 1: return kcos(y[0], y[1]);
    case 2: return -ksin(y[0], y[1], true);
    default: return -kcos(y[0], y[1]);
    }
}
#endif  // host

// ToInt32 (`~~x`): truncation with wrap-around modulo 2^32; NaN and infinities give 0.
SP_HD inline int32_t to_int32(double d)
{
    const uint64_t u = bits(d);
    const int e = (int)((u >> 52) & 0x7ff) - 1023;
    if (e < 0) return 0;                 // |d| < 1 (and zeros, subnormals)
    if (e >= 84 || e == 1024) return 0;  // multiples of 2^32, infinities, NaN
    const uint64_t m = (u & 0x000fffffffffffffull) | 0x0010000000000000ull;
    const uint32_t mag = e <= 52 ? (uint32_t)(m >> (52 - e)) : (uint32_t)(m << (e - 52));
    return (int32_t)((u >> 63) ? (0u - mag) : mag);
}

}  // namespace spjs
