/*
 * ORACLE — restatement of the three engine intrinsics the reference's arithmetic depends on:
 * Math.log10, Math.cos, Math.sin as computed by V8 7.8 (Node 12.22.9, the engine the golden vectors were made on).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Third-party dependency absent from /root/reference: V8 7.8.279.23 `src/base/ieee754.cc`, which is a port of
 * Sun's fdlibm 5.3 (e_log.c, e_log10.c, k_cos.c, k_sin.c, e_rem_pio2.c, s_cos.c, s_sin.c).  The published fdlibm
 * algorithms are restated here; parity is anchored on tests/golden/math_log10.bin and math_trig.bin (16384 + 8192
 * values produced by that engine), the twiddle tables and the window tables of the reference run.
 *
 * Compile with -ffp-contract=off: every operation must round separately, as it does in the engine.
 *
 * The routines below follow Sun Microsystems' fdlibm 5.3, whose files carry this notice, which is kept here as required:
 *
 *   ====================================================
 *   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
 *
 *   Developed at SunSoft, a Sun Microsystems, Inc. business.
 *   Permission to use, copy, modify, and distribute this
 *   software is freely granted, provided that this notice
 *   is preserved.
 *   ====================================================
 */
#ifndef SP_ORACLE_V8MATH_H
#define SP_ORACLE_V8MATH_H

#include <stdint.h>
#include <string.h>

static inline uint64_t v8m_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double v8m_from_bits(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
static inline int32_t v8m_hi(double x) { return (int32_t)(v8m_bits(x) >> 32); }
static inline uint32_t v8m_lo(double x) { return (uint32_t)v8m_bits(x); }
static inline double v8m_with_hi(double x, int32_t hi) { return v8m_from_bits(((uint64_t)(uint32_t)hi << 32) | v8m_lo(x)); }

/* natural logarithm, fdlibm e_log.c */
static double v8m_log(double x)
{
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        two54 = 1.80143985094819840000e+16,
        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
        Lg7 = 1.479819860511658591e-01;
    double hfsq, f, s, z, R, w, t1, t2, dk;
    int32_t k, hx, i, j;
    uint32_t lx;

    hx = v8m_hi(x);
    lx = v8m_lo(x);
    k = 0;
    if (hx < 0x00100000) {                   /* x < 2**-1022 */
        if (((hx & 0x7fffffff) | lx) == 0) return -two54 / 0.0;   /* log(+-0) = -inf */
        if (hx < 0) return (x - x) / 0.0;                            /* log(-#) = NaN */
        k -= 54;
        x *= two54;                          /* subnormal, scale up */
        hx = v8m_hi(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    i = (hx + 0x95f64) & 0x100000;
    x = v8m_with_hi(x, hx | (i ^ 0x3ff00000)); /* normalise x or x/2 */
    k += (i >> 20);
    f = x - 1.0;
    if ((0x000fffff & (2 + hx)) < 3) {       /* |f| < 2**-20 */
        if (f == 0.0) {
            if (k == 0) return 0.0;
            dk = (double)k;
            return dk * ln2_hi + dk * ln2_lo;
        }
        R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        dk = (double)k;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    s = f / (2.0 + f);
    dk = (double)k;
    z = s * s;
    i = hx - 0x6147a;
    w = z * z;
    j = 0x6b851 - hx;
    t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    R = t2 + t1;
    if (i > 0) {
        hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

/* Math.log10, fdlibm e_log10.c as carried by V8 */
static double v8m_log10(double x)
{
    static const double two54 = 1.80143985094819840000e+16, ivln10 = 4.34294481903251816668e-01,
        log10_2hi = 3.01029995663611771306e-01, log10_2lo = 3.69423907715893078616e-13;
    double y, z;
    int32_t i, k, hx;
    uint32_t lx;

    hx = v8m_hi(x);
    lx = v8m_lo(x);
    k = 0;
    if (hx < 0x00100000) {
        if (((hx & 0x7fffffff) | lx) == 0) return -two54 / 0.0;
        if (hx < 0) return (x - x) / 0.0;
        k -= 54;
        x *= two54;
        hx = v8m_hi(x);
        lx = v8m_lo(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
    hx = (hx & 0x000fffff) | ((0x3ff - i) << 20);
    y = (double)(k + i);
    x = v8m_from_bits(((uint64_t)(uint32_t)hx << 32) | lx);
    z = y * log10_2lo + ivln10 * v8m_log(x);
    return z + y * log10_2hi;
}

/* cosine kernel on [-pi/4, pi/4], fdlibm k_cos.c */
static double v8m_kcos(double x, double y)
{
    static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
        C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double a, iz, z, r, qx;
    int32_t ix = v8m_hi(x) & 0x7fffffff;
    if (ix < 0x3e400000) {                   /* |x| < 2**-27 */
        if ((int)x == 0) return 1.0;
    }
    z = x * x;
    r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    if (ix < 0x3fd33333) return 1.0 - (0.5 * z - (z * r - x * y));
    if (ix > 0x3fe90000) qx = 0.28125;
    else qx = v8m_from_bits((uint64_t)(uint32_t)(ix - 0x00200000) << 32);
    iz = 0.5 * z - qx;
    a = 1.0 - qx;
    return a - (iz - (z * r - x * y));
}

/* sine kernel on [-pi/4, pi/4], fdlibm k_sin.c */
static double v8m_ksin(double x, double y, int iy)
{
    static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
        S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z, r, v;
    int32_t ix = v8m_hi(x) & 0x7fffffff;
    if (ix < 0x3e400000) {
        if ((int)x == 0) return x;
    }
    z = x * x;
    v = z * x;
    r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (iy == 0) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

/* argument reduction, fdlibm e_rem_pio2.c — only the ranges the reference can reach (|x| <= 2^19 * pi/2) */
static int v8m_rem_pio2(double x, double *y)
{
    static const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
        pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21,
        pio2_3 = 2.02226624871116645580e-21, pio2_3t = 8.47842766036889956997e-32;
    double z, w, t, r, fn;
    int32_t i, j, n, ix, hx;

    hx = v8m_hi(x);
    ix = hx & 0x7fffffff;
    if (ix <= 0x3fe921fb) { y[0] = x; y[1] = 0; return 0; }
    if (ix < 0x4002d97c) {                   /* |x| < 3pi/4: n = +-1 */
        if (hx > 0) {
            z = x - pio2_1;
            if (ix != 0x3ff921fb) { y[0] = z - pio2_1t; y[1] = (z - y[0]) - pio2_1t; }
            else { z -= pio2_2; y[0] = z - pio2_2t; y[1] = (z - y[0]) - pio2_2t; }
            return 1;
        }
        z = x + pio2_1;
        if (ix != 0x3ff921fb) { y[0] = z + pio2_1t; y[1] = (z - y[0]) + pio2_1t; }
        else { z += pio2_2; y[0] = z + pio2_2t; y[1] = (z - y[0]) + pio2_2t; }
        return -1;
    }
    /* medium size; larger arguments (Payne-Hanek) never occur for 2*pi*i/n, i < n/2, nor for 6*pi*i/(n-1), i < n */
    t = x < 0 ? -x : x;
    n = (int32_t)(t * invpio2 + 0.5);
    fn = (double)n;
    r = t - fn * pio2_1;
    w = fn * pio2_1t;
    {
        /* high word of n*pi/2 for n = 1..32 decides whether the quick result is accurate enough */
        int quick = 0;
        if (n < 32) {
            int32_t hw = v8m_hi((double)n * 1.57079632679489655800e+00);
            quick = (ix != hw);
        }
        if (quick) {
            y[0] = r - w;
        } else {
            j = ix >> 20;
            y[0] = r - w;
            i = j - ((v8m_hi(y[0]) >> 20) & 0x7ff);
            if (i > 16) {                    /* 2nd iteration, good to 118 bits */
                t = r;
                w = fn * pio2_2;
                r = t - w;
                w = fn * pio2_2t - ((t - r) - w);
                y[0] = r - w;
                i = j - ((v8m_hi(y[0]) >> 20) & 0x7ff);
                if (i > 49) {                /* 3rd iteration, 151 bits */
                    t = r;
                    w = fn * pio2_3;
                    r = t - w;
                    w = fn * pio2_3t - ((t - r) - w);
                    y[0] = r - w;
                }
            }
        }
    }
    y[1] = (r - y[0]) - w;
    if (hx < 0) { y[0] = -y[0]; y[1] = -y[1]; return -n; }
    return n;
}

static double v8m_cos(double x)
{
    double y[2];
    int32_t n, ix = v8m_hi(x) & 0x7fffffff;
    if (ix <= 0x3fe921fb) return v8m_kcos(x, 0.0);
    if (ix >= 0x7ff00000) return x - x;
    n = v8m_rem_pio2(x, y);
    switch (n & 3) {
    case 0: return v8m_kcos(y[0], y[1]);
    case 1: return -v8m_ksin(y[0], y[1], 1);
    case 2: return -v8m_kcos(y[0], y[1]);
    default: return v8m_ksin(y[0], y[1], 1);
    }
}

static double v8m_sin(double x)
{
    double y[2];
    int32_t n, ix = v8m_hi(x) & 0x7fffffff;
    if (ix <= 0x3fe921fb) return v8m_ksin(x, 0.0, 0);
    if (ix >= 0x7ff00000) return x - x;
    n = v8m_rem_pio2(x, y);
    switch (n & 3) {
    case 0: return v8m_ksin(y[0], y[1], 1);
    case 1: return v8m_kcos(y[0], y[1]);
    case 2: return -v8m_ksin(y[0], y[1], 1);
    default: return -v8m_kcos(y[0], y[1]);
    }
}

#endif
