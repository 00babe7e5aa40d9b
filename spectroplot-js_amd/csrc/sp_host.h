// sp_host.h — host-side tables of a render plan: format names, tapers, twiddles and the exact threshold tables that
// replace the per-pixel Math.log10 of the reference.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "sp_formats.h"

namespace sphost {

// lib/samples.js:22-162: upper-cased name -> format; aliases; everything unknown is CU8.
int32_t parse_format(const char *name);

// lib/fft_nayuki.js:33-39
int32_t log2_exact(int64_t n);

// lib/fft_nayuki.js:42-47: cos(2*pi*i/n), sin(2*pi*i/n), i < n/2
void twiddles(int32_t n, double *cos_table, double *sin_table);

// lib/windows.js:14-88; returns false for an unknown name
bool window(const char *name, int32_t n, double *out, double *weight);

// lib/utils.js:25-40 over a list of keys: exact, then case-insensitive, then case-insensitive prefix; first hit in key order, -1: none
int32_t lookup_key(const char *const *keys, int32_t count, const char *name);
// lookup(windows, name) || windows.blackmanHarrisWindow (lib/spectroplot.js:241): the generator's plain name ("hann", ...)
const char *window_by_name(const char *name);

// The reference's COMPUTED colour maps (lib/soxcmap.js:12-49 `sox`, lib/naivecmap.js:13-81 `naive`, `grayscale`, `roentgen`,
// `phosphor`): `stops` entries of r, g, b as those functions evaluate them - the engine's Math.sin, Math.round (floor(x + 0.5)) and
// `~~` (truncation) included.  `key` is the map's key in the reference's table ("sox_cmap", ...).  false: not a computed map.
bool cmap_generate(const char *key, int32_t stops, uint8_t *rgb);

// The per-pixel arithmetic of lib/worker.js:92-113 as a function of abs2 = re^2 + im^2.
struct PixelMath {
    double block_norm_db;   // 10 * log10(block_norm)                      worker.js:31
    double gain, range;
    double color_max;       // lut_len - 1                                  worker.js:37
    double color_norm;      // lut_len / -range                             worker.js:38

    PixelMath(double block_norm, double gain, double range, int32_t lut_len);
    double dbfs(double abs2) const;        // 5*log10(abs2) + block_norm_db + gain        worker.js:93
    double rel_db(double abs2) const;      // dbfs - gain                                  worker.js:102-105
    int32_t gray(double abs2) const;       // colour index                                 worker.js:111-112
    int32_t centibel(double abs2) const;   // ~~(0.5 + rel_db * -10), before the 999 cap   worker.js:105
};

// Threshold tables: for finite abs2 > 0 both indices are monotone step functions of abs2, so
//   gray(abs2)  = #{ g in 1..lut_len-1 : abs2 >= gray_edge[g] }            (gray_edge[0] = 0)
//   level(abs2) = #{ j in 1..1000      : abs2 >= cb_edge[j] }              (cb_edge[0] = 0)
//   cB bin      = 999 - level, dropped when level == 1000 (negative key in the reference)
// abs2 == 0, +inf and NaN take bin 0 (ToInt32 of an infinity / NaN is 0) and are handled by the kernels.
// Unreachable edges are +inf.  Edges are the smallest doubles at which the restated arithmetic changes its result.
struct Thresholds {
    std::vector<double> gray_edge;   // [lut_len]
    std::vector<double> cb_edge;     // [1001]
    // first-guess coefficients for the LDS kernel: floor(a + b*log2(abs2)) is the exact index or one below it
    float gray_a, gray_b, cb_a, cb_b;
    // k_frames (sp_kernel_frames.h): t = a + b*log2((float)abs2) evaluated in f32 differs from the real-valued position of abs2
    // on the index scale by less than the margin m (error model: sp_host.cpp).  a is stored lowered by m, so floor(t) is the
    // exact index unless fract(t) >= thr = 1 - 2m; such lanes are decided against the edge tables.  frames_ok is false when a
    // margin is too wide to be useful (extreme gain / range slopes).
    float g2_a, g2_b, g2_thr, g2_m;
    float c2_a, c2_b, c2_thr, c2_m, c2_lo, c2_hi;   // c2_lo / c2_hi: clamp bounds of the level value, both past the threshold
    bool frames_ok;
    // k_frames counts one histogram for both scales: cell = colour index + level.  Both are monotone step functions of abs2, so
    // their sum steps by one at every edge of either scale and names the merged interval abs2 lies in; the counts of colour index g
    // are the cells [cell_g[g], cell_g[g+1]), those of level l the cells [cell_l[l], cell_l[l+1]).  Two more cells take the keys
    // that are not monotone (worker.js:105: -inf / NaN dB -> bin 0 with colour 0, +inf dB -> bin 0 with the last colour).
    std::vector<uint16_t> cell_g;    // [lut_len + 1]
    std::vector<uint16_t> cell_l;    // [SP_CB_HIST_SIZE + 2]
    int32_t cells;                   // lut_len + SP_CB_HIST_SIZE + 2 (the last two: special keys)
};
Thresholds build_thresholds(const PixelMath &pm, int32_t lut_len);

}  // namespace sphost
