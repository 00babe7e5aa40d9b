/*
 * ORACLE (C) — scalar f64 restatement of the reference's compute worker, for parity tests and the CPU baseline.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load libsp_oracle.so.  Nothing in spectroplot-js_amd/ includes, links or calls anything in oracle/.
 *
 * Pinned: tests/test_oracle_golden.py checks this file bit-for-bit against every vector in tests/golden/, which
 * were produced by executing the real lib/worker.js under Node 12 (oracle/gen_golden.js).
 *
 * Restated reference behaviour (file:line relative to /root/reference):
 *   spo_decode     lib/samples.js:15-169 (format table), :313-400 (accessors)
 *   spo_window     lib/windows.js:14-88
 *   spo_twiddles   lib/fft_nayuki.js:29-48
 *   spo_fft        lib/fft_nayuki.js:54-96 (transform), :103-119 (splitreal)
 *   spo_render     lib/worker.js:23-156
 * Engine intrinsics (Math.log10/cos/sin of V8 7.8): oracle/v8math.h.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  JS has no fused multiply-add.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

#include "v8math.h"

#define SPO_OK 0
#define SPO_ERR_NOT_POW2 (-1)     /* fft_nayuki.js:38-39 throws 'Length is not a power of 2' */
#define SPO_ERR_BYTE_LENGTH (-2)  /* typed-array constructor RangeError: byte length not a multiple of the element size */
#define SPO_ERR_ARG (-3)

/* ---- JS number coercions ------------------------------------------------------------------------------- */

/* ToInt32, i.e. what `~~x` yields */
static int32_t js_toint32(double d)
{
    if (!(d == d) || d == INFINITY || d == -INFINITY) return 0;
    double t = trunc(d);
    double m = fmod(t, 4294967296.0);
    if (m < 0) m += 4294967296.0;
    return (int32_t)(uint32_t)m;
}

/* store into a Uint8ClampedArray: round half to even, clamp to 0..255, NaN -> 0 */
static uint8_t js_clamp_u8(double v)
{
    if (!(v > 0)) return 0;
    if (v >= 255) return 255;
    return (uint8_t)nearbyint(v);
}

/* ---- sample formats (samples.js:15-169) ------------------------------------------------------------------ */

enum { K_U8, K_S8, K_U16, K_S16, K_U32, K_S32, K_F32, K_F64, K_CU4, K_CS4, K_CU12, K_CS12, K_CU64, K_CS64 };

typedef struct {
    int kind;
    double bias, scale;
    int width;            /* bytes per complex sample */
    int elem;             /* element size of the typed view */
    const uint8_t *p;
    size_t nbytes;
    int64_t nelem;        /* view.length */
    double count;         /* sampleCount = byteLength / sampleWidth (may be fractional) */
} spo_view;

static void upcase(const char *s, char *out, size_t cap)
{
    size_t i = 0;
    for (; s[i] && i + 1 < cap; i++) out[i] = (char)toupper((unsigned char)s[i]);
    out[i] = 0;
}

static int view_open(spo_view *v, const char *format, const uint8_t *p, size_t nbytes)
{
    char f[32];
    upcase(format ? format : "", f, sizeof f);
    v->p = p; v->nbytes = nbytes;
#define FMT(K, B, S, W, E) do { v->kind = K; v->bias = B; v->scale = S; v->width = W; v->elem = E; } while (0)
    if (!strcmp(f, "CU4")) FMT(K_CU4, 7.5, 1.0 / 7.5, 1, 1);
    else if (!strcmp(f, "CS4")) FMT(K_CS4, 0, 1.0 / 8.0, 1, 1);
    else if (!strcmp(f, "CS8") || !strcmp(f, "COMPLEX16S")) FMT(K_S8, 0, 1.0 / 128.0, 2, 1);
    else if (!strcmp(f, "CU16")) FMT(K_U16, 32767.5, 1.0 / 32768.0, 4, 2);
    else if (!strcmp(f, "CS16")) FMT(K_S16, 0, 1.0 / 32768.0, 4, 2);
    else if (!strcmp(f, "CU12")) FMT(K_CU12, 2047.5, 1.0 / 2047.5, 3, 1);
    else if (!strcmp(f, "CS12")) FMT(K_CS12, 0, 1.0 / 2048.0, 3, 1);
    else if (!strcmp(f, "CU32")) FMT(K_U32, 2147483647.5, 1.0 / 2147483648.0, 8, 4);
    else if (!strcmp(f, "CS32")) FMT(K_S32, 0, 1.0 / 2147483648.0, 8, 4);
    else if (!strcmp(f, "CU64")) FMT(K_CU64, 1.0, 1.0, 16, 4);
    else if (!strcmp(f, "CS64")) FMT(K_CS64, 0, 1.0, 16, 4);
    else if (!strcmp(f, "CF32") || !strcmp(f, "CFILE") || !strcmp(f, "COMPLEX")) FMT(K_F32, 0, 1.0, 8, 4);
    else if (!strcmp(f, "CF64")) FMT(K_F64, 0, 1.0, 16, 8);
    else FMT(K_U8, 127.5, 1.0 / 127.5, 2, 1);   /* CU8, DATA, COMPLEX16U and every unknown name */
#undef FMT
    if (nbytes % (size_t)v->elem) return SPO_ERR_BYTE_LENGTH;
    v->nelem = (int64_t)(nbytes / (size_t)v->elem);
    v->count = (double)nbytes / (double)v->width;
    return SPO_OK;
}

/* element of the typed view as a double; out of range -> `undefined` -> NaN in float context */
static double view_elem(const spo_view *v, int64_t i)
{
    if (i < 0 || i >= v->nelem) return NAN;
    const uint8_t *q = v->p + i * v->elem;
    switch (v->kind) {
    case K_U8: return (double)q[0];
    case K_S8: return (double)(int8_t)q[0];
    case K_U16: { uint16_t t; memcpy(&t, q, 2); return (double)t; }
    case K_S16: { int16_t t; memcpy(&t, q, 2); return (double)t; }
    case K_U32: { uint32_t t; memcpy(&t, q, 4); return (double)t; }
    case K_S32: { int32_t t; memcpy(&t, q, 4); return (double)t; }
    case K_F32: { float t; memcpy(&t, q, 4); return (double)t; }
    default: { double t; memcpy(&t, q, 8); return t; }
    }
}

/* byte read under bitwise operators: `undefined` coerces to 0 */
static int32_t view_byte0(const spo_view *v, int64_t i)
{
    return (i < 0 || (uint64_t)i >= v->nbytes) ? 0 : (int32_t)v->p[i];
}

/* component c (0 = I, 1 = Q) of sample pos */
static double view_sample(const spo_view *v, int64_t pos, int c)
{
    switch (v->kind) {
    case K_CU4: {
        int32_t b = view_byte0(v, pos);
        int32_t s = c ? (b & 0x0f) : ((b & 0xf0) >> 4);
        return ((double)s - v->bias) * v->scale;
    }
    case K_CS4: {
        int32_t b = view_byte0(v, pos);
        int32_t s = c ? (int32_t)((uint32_t)(b & 0x0f) << 28) >> 28 : (int32_t)((uint32_t)(b & 0xf0) << 24) >> 28;
        return (double)s * v->scale;
    }
    case K_CU12: {
        int32_t b0 = view_byte0(v, 3 * pos), b1 = view_byte0(v, 3 * pos + 1), b2 = view_byte0(v, 3 * pos + 2);
        int32_t s = c ? ((b2 << 4) | ((b1 & 0xf0) >> 4)) : (((b1 & 0x0f) << 8) | b0);
        return ((double)s - v->bias) * v->scale;
    }
    case K_CS12: {
        int32_t b0 = view_byte0(v, 3 * pos), b1 = view_byte0(v, 3 * pos + 1), b2 = view_byte0(v, 3 * pos + 2);
        int32_t s = c ? (int32_t)(((uint32_t)b2 << 24) | ((uint32_t)(b1 & 0xf0) << 16)) >> 20
                      : (int32_t)(((uint32_t)(b1 & 0x0f) << 28) | ((uint32_t)b0 << 20)) >> 20;
        return (double)s * v->scale;
    }
    case K_CU64: case K_CS64: {
        /* view is a Uint32Array: words lo, hi per component */
        int64_t ilo = 4 * pos + 2 * c, ihi = ilo + 1;
        double lo = NAN, hi;
        uint32_t w;
        if (ilo >= 0 && ilo < v->nelem) { memcpy(&w, v->p + 4 * ilo, 4); lo = (double)w; }
        if (ihi >= 0 && ihi < v->nelem) {
            memcpy(&w, v->p + 4 * ihi, 4);
            hi = v->kind == K_CS64 ? (double)(int32_t)w : (double)w;
        } else {
            hi = v->kind == K_CS64 ? 0.0 : NAN;    /* `undefined >> 0` is 0, `undefined / x` is NaN */
        }
        double s = hi / 2147483648.0 + lo / 18446744073709551616.0;
        return v->kind == K_CS64 ? s : s - v->bias;
    }
    default:
        return (view_elem(v, 2 * pos + c) - v->bias) * v->scale;
    }
}

int spo_decode(const char *format, const uint8_t *buf, size_t nbytes, int64_t pos_lo, int64_t count, double *out_iq,
               double *sample_count, int *sample_width)
{
    spo_view v;
    int rc = view_open(&v, format, buf, nbytes);
    if (rc) return rc;
    if (sample_count) *sample_count = v.count;
    if (sample_width) *sample_width = v.width;
    for (int64_t k = 0; k < count; k++) {
        out_iq[2 * k] = view_sample(&v, pos_lo + k, 0);
        out_iq[2 * k + 1] = view_sample(&v, pos_lo + k, 1);
    }
    return SPO_OK;
}

/* ---- windows (windows.js:14-88) ------------------------------------------------------------------------ */

static const double JS_PI = 3.141592653589793;

int spo_window(const char *name, int n, double *out, double *weight)
{
    double sum = 0.0;
    double nm1 = (double)(n - 1);
    for (int i = 0; i < n; i++) {
        double di = (double)i, w;
        if (!strcmp(name, "rectangular")) w = 1.0;
        else if (!strcmp(name, "bartlett")) w = 1.0 - fabs((di - 0.5 * nm1) / (0.5 * nm1));
        else if (!strcmp(name, "hamming")) w = 0.54 - 0.46 * v8m_cos(2.0 * JS_PI * di / nm1);
        else if (!strcmp(name, "hann")) w = 0.5 * (1.0 - v8m_cos(2.0 * JS_PI * di / nm1));
        else if (!strcmp(name, "blackman"))
            w = 0.42 - (0.5 * v8m_cos((2.0 * JS_PI * di) / nm1)) + (0.08 * v8m_cos((4.0 * JS_PI * di) / nm1));
        else if (!strcmp(name, "blackmanHarris"))
            w = 0.35875 - (0.48829 * v8m_cos((2.0 * JS_PI * di) / nm1)) + (0.14128 * v8m_cos((4.0 * JS_PI * di) / nm1))
                - (0.01168 * v8m_cos((6.0 * JS_PI * di) / nm1));
        else return SPO_ERR_ARG;
        out[i] = w;
        sum += w;
    }
    *weight = sum;
    return SPO_OK;
}

/* ---- FFT (fft_nayuki.js) ------------------------------------------------------------------------------- */

static int ilog2_exact(int n)
{
    for (int i = 0; i < 31; i++) if ((1 << i) == n) return i;
    return -1;
}

int spo_twiddles(int n, double *cosv, double *sinv)
{
    if (ilog2_exact(n) < 0) return SPO_ERR_NOT_POW2;
    for (int i = 0; i < n / 2; i++) {
        double a = 2 * JS_PI * (double)i / (double)n;   /* ((2*pi)*i)/n, left to right */
        cosv[i] = v8m_cos(a);
        sinv[i] = v8m_sin(a);
    }
    return SPO_OK;
}

static void fft_core(int n, int levels, const double *ct, const double *st, double *re, double *im)
{
    for (int i = 0; i < n; i++) {
        unsigned x = (unsigned)i, y = 0;
        for (int b = 0; b < levels; b++) { y = (y << 1) | (x & 1); x >>= 1; }
        if ((int)y > i) {
            double t = re[i]; re[i] = re[y]; re[y] = t;
            t = im[i]; im[i] = im[y]; im[y] = t;
        }
    }
    for (int size = 2; size <= n && size > 0; size *= 2) {
        int half = size / 2, step = n / size;
        for (int i = 0; i < n; i += size) {
            for (int j = i, k = 0; j < i + half; j++, k += step) {
                int l = j + half;
                double tpre = re[l] * ct[k] + im[l] * st[k];
                double tpim = -re[l] * st[k] + im[l] * ct[k];
                re[l] = re[j] - tpre;
                im[l] = im[j] - tpim;
                re[j] += tpre;
                im[j] += tpim;
            }
        }
    }
}

static void split_real(int n, double *re, double *im)
{
    im[0] = 0;
    re[n / 2] = im[0];        /* the reference zeroes imag[0] first, so bin n/2 becomes (0, 0) */
    im[n / 2] = 0;
    for (int i = 1; i < n / 2; i++) {
        double lr = 0.5 * (re[i] + re[n - i]);
        double li = 0.5 * (im[i] - im[n - i]);
        double rr = 0.5 * (im[i] + im[n - i]);
        double ri = 0.5 * (-re[i] + re[n - i]);
        re[i] = lr; im[i] = li; re[n - i] = rr; im[n - i] = ri;
    }
}

int spo_fft(int n, double *re, double *im, int split)
{
    int levels = ilog2_exact(n);
    if (levels < 0) return SPO_ERR_NOT_POW2;
    double *ct = (double *)malloc(sizeof(double) * (size_t)(n / 2 + 1));
    double *st = (double *)malloc(sizeof(double) * (size_t)(n / 2 + 1));
    spo_twiddles(n, ct, st);
    fft_core(n, levels, ct, st, re, im);
    if (split) split_real(n, re, im);
    free(ct); free(st);
    return SPO_OK;
}

/* ---- render (worker.js:23-156) ----------------------------------------------------------------------------- */

/*
 * lut_rgb: lut_len x 3 bytes (the reference's cmap entries are integers 0..255; the caller has already forced the ends).
 * Outputs: rgba[4*width*n], gauge_*[width], c_hist[lut_len], cB_hist[1000], *dmin, *dmax;
 * optional db_plane[width*n] receives (dBfs - gain) per frame x, bin i at [x*n + i]; optional abs2_plane likewise.
 */
int spo_render(const char *format, const uint8_t *buf, size_t nbytes, int n, const double *windowc, double block_norm,
               double gain, double range, const uint8_t *lut_rgb, int lut_len, int width, int channel_mode, int waterfall,
               uint8_t *rgba, uint8_t *gauge_mins, uint8_t *gauge_maxs, uint8_t *gauge_amps,
               int64_t *c_hist, int64_t *cB_hist, double *dmin_out, double *dmax_out, double *db_plane, double *abs2_plane)
{
    spo_view v;
    int rc = view_open(&v, format, buf, nbytes);
    if (rc) return rc;
    int levels = ilog2_exact(n);
    if (levels < 0) return SPO_ERR_NOT_POW2;
    if (lut_len < 1 || width < 0) return SPO_ERR_ARG;

    const double block_norm_db = 10 * v8m_log10(block_norm);
    const double color_max = (double)(lut_len - 1);
    const double color_norm = (double)lut_len / -range;
    const double stride = (v.count - (double)n) / (double)(width - 1);
    const int W = width;
    double dmin = 0.0, dmax = -200.0;

    memset(cB_hist, 0, sizeof(int64_t) * 1000);
    memset(c_hist, 0, sizeof(int64_t) * (size_t)lut_len);

    double *ct = (double *)malloc(sizeof(double) * (size_t)(n / 2 + 1));
    double *st = (double *)malloc(sizeof(double) * (size_t)(n / 2 + 1));
    double *re = (double *)malloc(sizeof(double) * (size_t)n);
    double *im = (double *)malloc(sizeof(double) * (size_t)n);
    spo_twiddles(n, ct, st);

    for (int x = 0; x < W; x++) {
        const int32_t start = js_toint32(0.5 + stride * (double)x);
        for (int k = 0; k < n; k++) {
            int64_t pos = (int64_t)start + k;
            re[k] = windowc[k] * view_sample(&v, pos, 0);
            im[k] = windowc[k] * view_sample(&v, pos, 1);
        }
        fft_core(n, levels, ct, st, re, im);
        if (channel_mode) split_real(n, re, im);

        double fmin = 0.0, fmax = -200.0;
        for (int i = 0; i < n; i++) {
            /* n/2 is a float division in JS: for n = 1 it would be 0.5; n >= 2 here */
            int y = i <= n / 2 ? n / 2 - i : n / 2 + n - i;
            double abs2 = re[i] * re[i] + im[i] * im[i];
            double dBfs = 5 * v8m_log10(abs2) + block_norm_db + gain;
            double d = dBfs - gain;
            if (d < fmin) fmin = d;
            if (d > fmax) fmax = d;
            int32_t cB = js_toint32(0.5 + d * -10);
            int32_t bin = cB >= 1000 ? 999 : cB;
            if (bin >= 0) cB_hist[bin] += 1;     /* negative keys never land in the array */
            double grayU = color_max - dBfs * color_norm;
            int32_t gray = js_toint32(0.5 + (grayU < 0 ? 0 : grayU > color_max ? color_max : grayU));
            c_hist[gray] += 1;
            const uint8_t *c = lut_rgb + 3 * gray;
            size_t j = waterfall ? ((size_t)n * (size_t)(W - 1 - x) + (size_t)(n - 1 - y)) * 4
                                 : ((size_t)x + (size_t)W * (size_t)y) * 4;
            rgba[j] = c[0]; rgba[j + 1] = c[1]; rgba[j + 2] = c[2]; rgba[j + 3] = 255;
            if (db_plane) db_plane[(size_t)x * n + i] = d;
            if (abs2_plane) abs2_plane[(size_t)x * n + i] = abs2;
        }
        if (fmin < dmin) dmin = fmin;
        if (fmax > dmax) dmax = fmax;
        gauge_mins[x] = js_clamp_u8(0.5 + (range + fmin) * 256 / range);
        gauge_maxs[x] = js_clamp_u8(0.5 + (range + fmax) * 256 / range);
        int64_t mid = (int64_t)start + n / 2;
        double ci = view_sample(&v, mid, 0), cq = view_sample(&v, mid, 1);
        double amp = 5 * v8m_log10(ci * ci + cq * cq) + gain;
        gauge_amps[x] = js_clamp_u8(0.5 + (range + amp) * 256 / range);
    }
    *dmin_out = dmin;
    *dmax_out = dmax;
    free(ct); free(st); free(re); free(im);
    return SPO_OK;
}

double spo_log10(double x) { return v8m_log10(x); }
double spo_cos(double x) { return v8m_cos(x); }
double spo_sin(double x) { return v8m_sin(x); }
