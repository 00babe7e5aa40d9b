// sp_host.cpp — host-side tables (see sp_host.h).  Compiled with -ffp-contract=off.
#include "sp_host.h"

#include <cctype>
#include <cmath>
#include <cstring>

namespace sphost {

int32_t parse_format(const char *name)
{
    char f[24];
    size_t i = 0;
    for (; name && name[i] && i + 1 < sizeof f; i++) f[i] = (char)toupper((unsigned char)name[i]);
    f[i] = 0;
    static const struct { const char *name; int32_t id; } table[] = {
        {"CU4", SP_FMT_CU4}, {"CS4", SP_FMT_CS4}, {"CU8", SP_FMT_CU8}, {"DATA", SP_FMT_CU8}, {"COMPLEX16U", SP_FMT_CU8},
        {"CS8", SP_FMT_CS8}, {"COMPLEX16S", SP_FMT_CS8}, {"CU16", SP_FMT_CU16}, {"CS16", SP_FMT_CS16},
        {"CU12", SP_FMT_CU12}, {"CS12", SP_FMT_CS12}, {"CU32", SP_FMT_CU32}, {"CS32", SP_FMT_CS32},
        {"CU64", SP_FMT_CU64}, {"CS64", SP_FMT_CS64}, {"CF32", SP_FMT_CF32}, {"CFILE", SP_FMT_CF32},
        {"COMPLEX", SP_FMT_CF32}, {"CF64", SP_FMT_CF64},
    };
    for (const auto &e : table)
        if (!strcmp(f, e.name)) return e.id;
    return SP_FMT_CU8;   // lib/samples.js:149-155: default
}

int32_t log2_exact(int64_t n)
{
    for (int32_t i = 0; i < 31; i++)
        if (((int64_t)1 << i) == n) return i;
    return -1;
}

static const double kPi = 3.141592653589793;   // Math.PI

void twiddles(int32_t n, double *cos_table, double *sin_table)
{
    for (int32_t i = 0; i < n / 2; i++) {
        const double a = 2 * kPi * (double)i / (double)n;
        cos_table[i] = spjs::cos(a);
        sin_table[i] = spjs::sin(a);
    }
}

bool window(const char *name, int32_t n, double *out, double *weight)
{
    enum { RECT, BARTLETT, HAMMING, HANN, BLACKMAN, BH } kind;
    if (!strcmp(name, "rectangular")) kind = RECT;
    else if (!strcmp(name, "bartlett")) kind = BARTLETT;
    else if (!strcmp(name, "hamming")) kind = HAMMING;
    else if (!strcmp(name, "hann")) kind = HANN;
    else if (!strcmp(name, "blackman")) kind = BLACKMAN;
    else if (!strcmp(name, "blackmanHarris")) kind = BH;
    else return false;
    const double m = (double)(n - 1);
    double sum = 0.0;
    for (int32_t k = 0; k < n; k++) {
        const double i = (double)k;
        double w = 1.0;
        switch (kind) {
        case RECT: break;
        case BARTLETT: w = 1.0 - std::fabs((i - 0.5 * m) / (0.5 * m)); break;
        case HAMMING: w = 0.54 - 0.46 * spjs::cos(2.0 * kPi * i / m); break;
        case HANN: w = 0.5 * (1.0 - spjs::cos(2.0 * kPi * i / m)); break;
        case BLACKMAN:
            w = 0.42 - (0.5 * spjs::cos((2.0 * kPi * i) / m)) + (0.08 * spjs::cos((4.0 * kPi * i) / m));
            break;
        case BH:
            w = 0.35875 - (0.48829 * spjs::cos((2.0 * kPi * i) / m)) + (0.14128 * spjs::cos((4.0 * kPi * i) / m))
                - (0.01168 * spjs::cos((6.0 * kPi * i) / m));
            break;
        }
        out[k] = w;
        sum += w;
    }
    *weight = sum;
    return true;
}

int32_t lookup_key(const char *const *keys, int32_t count, const char *name)
{
    if (!name || !*name) return -1;
    for (int32_t i = 0; i < count; i++)
        if (!strcmp(keys[i], name)) return i;
    auto lower_eq = [](const char *key, const char *want, bool prefix) {
        size_t i = 0;
        for (; want[i]; i++)
            if (!key[i] || tolower((unsigned char)key[i]) != tolower((unsigned char)want[i])) return false;
        return prefix || key[i] == 0;
    };
    for (int32_t i = 0; i < count; i++)
        if (lower_eq(keys[i], name, false)) return i;
    for (int32_t i = 0; i < count; i++)
        if (lower_eq(keys[i], name, true)) return i;
    return -1;
}

const char *window_by_name(const char *name)
{
    // key order of the reference's `windows` module namespace (sorted; tests/golden/parse.json window_key_order)
    static const char *const keys[] = {"bartlettWindow", "blackmanHarrisWindow", "blackmanWindow", "hammingWindow", "hannWindow", "rectangularWindow"};
    static const char *const plain[] = {"bartlett", "blackmanHarris", "blackman", "hamming", "hann", "rectangular"};
    const int32_t i = lookup_key(keys, 6, name);
    return plain[i < 0 ? 1 : i];
}

// ---- computed colour maps -------------------------------------------------------------------------------------------------------------
// Each channel of a computed map is a piecewise function of the entry number; a Piece is one branch of it, in the branch's own
// operation order (the bytes depend on it).  Two families: lib/soxcmap.js works on x = i / (stops - 1) with sine ramps and rounds to
// nearest (Math.round), lib/naivecmap.js works on i itself with linear ramps and truncates (~~).
namespace {

constexpr double kPi = 3.141592653589793;   // Math.PI

// Math.round for the values that occur here (>= 0, far below 2^52): halves go up; decided on the fraction itself, not on v + 0.5
// (0.49999999999999994 + 0.5 rounds to 1.0 in f64; Math.round gives 0)
inline uint8_t js_round_u8(double v)
{
    const double f = std::floor(v);
    return (uint8_t)(int)(v - f >= 0.5 ? f + 1 : f);
}
// ~~v for 0 <= v < 2^31: truncation
inline uint8_t js_trunc_u8(double v) { return (uint8_t)(int)v; }

// lib/soxcmap.js:14-42: quarter-sine rises between two knots, a half-sine hump and a linear tail for blue
void sox_entry(double x, uint8_t *rgb)
{
    auto rise = [x](double from, double width) { return 1 * spjs::sin((x - from) / width * kPi / 2); };
    const double r = x < .13 ? 0.0 : x < .73 ? rise(.13, .60) : 1.0;
    const double g = x < .60 ? 0.0 : x < .91 ? rise(.60, .31) : 1.0;
    const double b = x < .60 ? .5 * spjs::sin((x - .00) / .60 * kPi) : x < .78 ? 0.0 : (x - .78) / .22;
    rgb[0] = js_round_u8(255 * r);
    rgb[1] = js_round_u8(255 * g);
    rgb[2] = js_round_u8(255 * b);
}

}  // namespace

bool cmap_generate(const char *key, int32_t stops, uint8_t *rgb)
{
    if (!key || stops < 1 || !rgb) return false;
    const std::string k = key;
    const double n = (double)stops;
    for (int32_t idx = 0; idx < stops; idx++) {
        const double i = (double)idx;
        uint8_t *o = rgb + 3 * (size_t)idx;
        if (k == "sox_cmap") {
            sox_entry(i / (n - 1.0), o);                                                  // lib/soxcmap.js:15
        } else if (k == "grayscale_cmap") {
            o[0] = o[1] = o[2] = js_trunc_u8(i * 255 / n);                                // lib/naivecmap.js:43-46
        } else if (k == "roentgen_cmap") {
            o[0] = o[1] = o[2] = js_trunc_u8(255 - (i * 255 / n));                        // :55-58
        } else if (k == "naive_cmap") {                                                   // :16-37: four quarters, blue - red - yellow - white
            double r, g, b;
            if (i < n / 4) { b = i * 128 / (n / 4); g = 0; r = 0; }
            else if (i < n / 2) { b = 256 - i / 2; g = 0; r = i - n / 4; }
            else if (i < n * 3 / 4) { b = 0; g = i - n / 2; r = 255; }
            else { b = i - n * 3 / 4; g = 255; r = 255; }
            o[0] = js_trunc_u8(r); o[1] = js_trunc_u8(g); o[2] = js_trunc_u8(b);
        } else if (k == "phosphor_cmap") {                                                // :67-78: green up to 191, then towards white
            const double h = n / 2;
            double r, g, b;
            if (i < h) { r = 0; g = i * 191 / h; b = 0; }
            else { r = (i - h) * 255 / h; g = 191 + (i - h) * 64 / h; b = (i - h) * 255 / h; }
            o[0] = js_trunc_u8(r); o[1] = js_trunc_u8(g); o[2] = js_trunc_u8(b);
        } else {
            return false;
        }
    }
    return true;
}

PixelMath::PixelMath(double block_norm, double gain_, double range_, int32_t lut_len)
    : block_norm_db(10 * spjs::log10(block_norm)), gain(gain_), range(range_), color_max((double)(lut_len - 1)),
      color_norm((double)lut_len / -range_)
{
}

double PixelMath::dbfs(double abs2) const { return 5 * spjs::log10(abs2) + block_norm_db + gain; }
double PixelMath::rel_db(double abs2) const { return dbfs(abs2) - gain; }

int32_t PixelMath::gray(double abs2) const
{
    const double u = color_max - dbfs(abs2) * color_norm;
    return spjs::to_int32(0.5 + (u < 0 ? 0 : u > color_max ? color_max : u));
}

int32_t PixelMath::centibel(double abs2) const { return spjs::to_int32(0.5 + rel_db(abs2) * -10); }

namespace {

const uint64_t kMinPos = 1;                        // smallest subnormal
const uint64_t kMaxFinite = 0x7fefffffffffffffull;

// smallest positive finite double (as bits) at which pred becomes true; pred must be monotone false -> true
template <typename Pred>
double first_true(uint64_t lo, Pred pred)
{
    if (pred(spjs::from_bits(lo))) return spjs::from_bits(lo);
    uint64_t hi = kMaxFinite;
    if (!pred(spjs::from_bits(hi))) return spjs::inf();
    // invariant: pred(lo) false, pred(hi) true
    while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (pred(spjs::from_bits(mid))) hi = mid;
        else lo = mid;
    }
    return spjs::from_bits(hi);
}

// The same answer, found from a guess: the step sits within a few ulps of where the real-valued formula puts it, so a bracket of
// 2^12 ulps around the guess (widened by 16 x while it does not hold the step) and ~13 halvings replace the 62 halvings over all
// positive doubles.  Any failure of the bracket search falls back to the full search; pred(lo) is known to be false.
template <typename Pred>
double first_true_near(uint64_t lo, double guess, Pred pred)
{
    if (pred(spjs::from_bits(lo))) return spjs::from_bits(lo);
    if (!(guess > 0.0) || guess == spjs::inf()) return first_true(lo, pred);
    const uint64_t gb = spjs::bits(guess);
    for (uint64_t delta = (uint64_t)1 << 12; delta < ((uint64_t)1 << 56); delta <<= 4) {
        uint64_t l = gb > delta ? gb - delta : kMinPos, h = gb + delta;
        if (l < lo) l = lo;
        if (h > kMaxFinite) h = kMaxFinite;
        if (h <= l) return first_true(lo, pred);
        const bool pl = l == lo ? false : pred(spjs::from_bits(l));
        if (pl) continue;                              // the step is below the bracket
        if (!pred(spjs::from_bits(h))) {
            if (h == kMaxFinite) return spjs::inf();
            continue;                                  // the step is above the bracket
        }
        while (h - l > 1) {
            const uint64_t mid = l + (h - l) / 2;
            if (pred(spjs::from_bits(mid))) h = mid;
            else l = mid;
        }
        return spjs::from_bits(h);
    }
    return first_true(lo, pred);
}

}  // namespace

Thresholds build_thresholds(const PixelMath &pm, int32_t lut_len)
{
    Thresholds t;
    t.gray_edge.assign((size_t)lut_len, 0.0);
    t.cb_edge.assign(SP_CB_HIST_SIZE + 1, 0.0);

    // first guesses: dbfs = 5*log10(2)*log2(abs2) + block_norm_db + gain
    const double c = 5.0 * 0.30102999566398120;
    const double per_db = (double)lut_len / pm.range;
    // real-valued position of abs2 on the two scales: colour 0.5 + color_max + per_db*dbfs, level 999.5 + 10*rel_db; step g / j is
    // where the position reaches g / j
    const double pos_g_a = 0.5 + pm.color_max + per_db * (pm.block_norm_db + pm.gain), pos_g_b = per_db * c;
    const double pos_c_a = (SP_CB_HIST_SIZE - 0.5) + 10.0 * pm.block_norm_db, pos_c_b = 10.0 * c;

    uint64_t lo = kMinPos;
    for (int32_t g = 1; g < lut_len; g++) {
        const double e = first_true_near(lo, std::exp2(((double)g - pos_g_a) / pos_g_b), [&](double a) { return pm.gray(a) >= g; });
        t.gray_edge[(size_t)g] = e;
        if (e != spjs::inf()) lo = spjs::bits(e);   // edges are non-decreasing
    }
    auto level = [&](double a) {
        const int32_t cb = pm.centibel(a);
        if (cb < 0) return SP_CB_HIST_SIZE;
        return SP_CB_HIST_SIZE - 1 - (cb > SP_CB_HIST_SIZE - 1 ? SP_CB_HIST_SIZE - 1 : cb);
    };
    lo = kMinPos;
    for (int32_t j = 1; j <= SP_CB_HIST_SIZE; j++) {
        const double e = first_true_near(lo, std::exp2(((double)j - pos_c_a) / pos_c_b), [&](double a) { return level(a) >= j; });
        t.cb_edge[(size_t)j] = e;
        if (e != spjs::inf()) lo = spjs::bits(e);
    }

    // gray  = floor(0.5 + color_max + per_db*dbfs)  -> guess floor(color_max + per_db*dbfs)   in {gray-1, gray}
    // level = floor(999.5 + 10*rel_db) (off the edges) -> guess floor(999 + 10*rel_db)       in {level-1, level}
    t.gray_a = (float)(pm.color_max + per_db * (pm.block_norm_db + pm.gain));
    t.gray_b = (float)(per_db * c);
    t.cb_a = (float)((SP_CB_HIST_SIZE - 1) + 10.0 * pm.block_norm_db);
    t.cb_b = (float)(10.0 * c);

    // ---- k_frames: floor(a + b*L) with L = v_log_f32((float)abs2), all in f32 -------------------------------------------------
    // real-valued position of abs2:  colour 0.5 + color_max + per_db*dbfs,  level 999.5 + 10*rel_db  (dbfs = c*log2(abs2) + ...)
    // Only lanes whose level value lies inside its clamp range [0, 1000) take the fast path (the others are decided against the
    // edge tables whatever the colour test says), i.e. L in [-c_a/c_b, (1000 - c_a)/c_b] =: [-l_max, l_max] (plus one for slack).
    // Error of the f32 evaluation there:  (float)abs2: 2^-24 relative = 2^-24 / ln 2 in L;  v_log_f32: at most 1.0 ulp of its result
    // over every positive normal f32 (tools/log_error.hip, profiles/r02_v_log_f32_error.txt);  b rounded to f32: |b| * 2^-24 * l_max;
    // a rounded to f32: (|a| + 1) * 2^-24;  the fma's own rounding: half an ulp of the largest value that is not clamped (colour:
    // lut_len, level: 1024).  The reference's own arithmetic places its steps within ~1e-12 of the real-valued positions (f64).
    // Safety factor 1.25.
    const double g_a = 0.5 + pm.color_max + per_db * (pm.block_norm_db + pm.gain), g_b = per_db * c;
    const double c_a = (SP_CB_HIST_SIZE - 0.5) + 10.0 * pm.block_norm_db, c_b = 10.0 * c;
    const double l_max = std::fmax(std::fabs(-c_a / c_b), std::fabs((SP_CB_HIST_SIZE - c_a) / c_b)) + 1.0;
    auto margin = [&](double a, double b, double tmax) {
        const double ulp_l = std::ldexp(1.0, std::ilogb(l_max) - 23);
        const double e_l = 0x1p-24 / 0.6931471805599453 + ulp_l;
        const double ulp_t = std::ldexp(1.0, std::ilogb(tmax) - 23);
        return 1.25 * (std::fabs(b) * e_l + std::fabs(b) * 0x1p-24 * l_max + (std::fabs(a) + 1.0) * 0x1p-24 + 0.5 * ulp_t) + 1e-7;
    };
    const double g_m = margin(g_a, g_b, (double)lut_len), c_m = margin(c_a, c_b, 1024.0);
    t.g2_a = (float)(g_a - g_m);
    t.g2_b = (float)g_b;
    t.g2_m = (float)g_m;
    t.g2_thr = std::nextafterf((float)(1.0 - 2.0 * g_m), 0.0f);
    t.c2_a = (float)(c_a - c_m);
    t.c2_b = (float)c_b;
    t.c2_m = (float)c_m;
    t.c2_thr = std::nextafterf((float)(1.0 - 2.0 * c_m), 0.0f);
    // level values below 0 or from 1000 - 2m up (the two-step bin 0 and the dropped keys, worker.js:105-106) always go to the
    // edge tables: their clamp bounds have a fractional part past the threshold
    t.c2_lo = -(float)(0.5 * c_m);
    t.c2_hi = (float)((double)SP_CB_HIST_SIZE - c_m);
    {
        // merged cells: an infinite (unreachable) edge is never counted, and the indices behind it own no cell
        const int32_t end = (lut_len - 1) + SP_CB_HIST_SIZE + 1;
        auto count_le = [](const std::vector<double> &edge, double v) {   // edges [1..] are non-decreasing
            int32_t lo = 0, hi = (int32_t)edge.size() - 1;               // number of j >= 1 with edge[j] <= v
            while (lo < hi) {
                const int32_t mid = (lo + hi + 1) >> 1;
                if (edge[(size_t)mid] <= v) lo = mid;
                else hi = mid - 1;
            }
            return lo;
        };
        t.cells = end + 2;
        t.cell_g.assign((size_t)lut_len + 1, (uint16_t)end);
        t.cell_l.assign(SP_CB_HIST_SIZE + 2, (uint16_t)end);
        t.cell_g[0] = 0;
        t.cell_l[0] = 0;
        for (int32_t g = 1; g < lut_len; g++)
            if (t.gray_edge[(size_t)g] != spjs::inf()) t.cell_g[(size_t)g] = (uint16_t)(g + count_le(t.cb_edge, t.gray_edge[(size_t)g]));
        for (int32_t l = 1; l <= SP_CB_HIST_SIZE; l++)
            if (t.cb_edge[(size_t)l] != spjs::inf()) t.cell_l[(size_t)l] = (uint16_t)(l + count_le(t.gray_edge, t.cb_edge[(size_t)l]));
    }
    auto fractf_ = [](float v) { return v - std::floor(v); };
    t.frames_ok = g_m < 0.125 && c_m < 0.125 && l_max < 120.0 && std::isfinite(t.g2_a) && std::isfinite(t.c2_a) && fractf_(t.c2_lo) >= t.c2_thr
                  && fractf_(t.c2_hi) >= t.c2_thr && t.c2_hi < (float)SP_CB_HIST_SIZE;
    return t;
}

}  // namespace sphost
