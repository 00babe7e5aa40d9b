// Test harness (built and run by tests/test_thresholds_cpu.py): the edge tables of sphost::build_thresholds against an independent
// full bisection over all positive doubles with the same public pixel arithmetic (sphost::PixelMath::gray / centibel), for a
// spread of gains, ranges, norms and LUT lengths.  Prints the number of differing edges and the time per table set.
//
// A bisection finds THE step only if the index is monotone in abs2 - an assumption about the restated engine log10.
// The second part checks it where it matters: around every finite edge of every table set, every double within 64 ulps below the edge
// (4096 for the default request) must still give the lower index and every double within as many ulps above it the upper one, i.e. the
// index does not wobble back and forth in the neighbourhood a pixel would have to fall into to be mis-binned.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "sp_host.h"

static double from_bits(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static uint64_t to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

template <typename Pred>
static double full_search(uint64_t lo, Pred pred)
{
    if (pred(from_bits(lo))) return from_bits(lo);
    uint64_t hi = 0x7fefffffffffffffull;
    if (!pred(from_bits(hi))) return from_bits(0x7ff0000000000000ull);
    while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (pred(from_bits(mid))) hi = mid;
        else lo = mid;
    }
    return from_bits(hi);
}

int main()
{
    const double gains[] = {6.0, 0.0, -10.0, 59.0, 2500.0, -2500.0, 33.3};
    const double ranges[] = {30.0, 6.0, 45.5, 120.0, 3000.0};
    const double norms[] = {1.0 / 512.0, 1.0, 1e-6, 37.5};
    const int luts[] = {256, 2, 3, 17, 255, 300};
    long bad = 0, sets = 0, wobbles = 0, probed = 0;
    double ms = 0;
    const uint64_t inf_bits = 0x7ff0000000000000ull;
    // pred must be false on [e - w ulps, e) and true on [e, e + w ulps]
    auto neighbourhood = [&](double e, uint64_t w, auto pred) {
        const uint64_t b = to_bits(e);
        if (b >= inf_bits || b == 0) return;
        for (uint64_t k = 1; k <= w && k < b; k++) {
            probed++;
            if (pred(from_bits(b - k))) wobbles++;
        }
        for (uint64_t k = 0; k <= w && b + k < inf_bits; k++) {
            probed++;
            if (!pred(from_bits(b + k))) wobbles++;
        }
    };
    for (double gain : gains)
        for (double range : ranges)
            for (double bn : norms)
                for (int lut_len : luts) {
                    sphost::PixelMath pm(bn, gain, range, lut_len);
                    const auto t0 = std::chrono::steady_clock::now();
                    const sphost::Thresholds th = sphost::build_thresholds(pm, lut_len);
                    ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    sets++;
                    const uint64_t w = (gain == 6.0 && range == 30.0 && lut_len == 256) ? 4096 : 64;
                    uint64_t lo = 1;
                    for (int g = 1; g < lut_len; g++) {
                        const double e = full_search(lo, [&](double a) { return pm.gray(a) >= g; });
                        if (to_bits(e) != to_bits(th.gray_edge[(size_t)g])) bad++;
                        neighbourhood(e, w, [&](double a) { return pm.gray(a) >= g; });
                        if (e != from_bits(0x7ff0000000000000ull)) lo = to_bits(e);
                    }
                    auto level = [&](double a) {
                        const int32_t cb = pm.centibel(a);
                        if (cb < 0) return SP_CB_HIST_SIZE;
                        return SP_CB_HIST_SIZE - 1 - (cb > SP_CB_HIST_SIZE - 1 ? SP_CB_HIST_SIZE - 1 : cb);
                    };
                    lo = 1;
                    for (int j = 1; j <= SP_CB_HIST_SIZE; j++) {
                        const double e = full_search(lo, [&](double a) { return level(a) >= j; });
                        if (to_bits(e) != to_bits(th.cb_edge[(size_t)j])) bad++;
                        neighbourhood(e, w, [&](double a) { return level(a) >= j; });
                        if (e != from_bits(0x7ff0000000000000ull)) lo = to_bits(e);
                    }
                }
    printf("%ld table sets, %ld differing edges, %.3f ms per set\n", sets, bad, ms / sets);
    printf("%ld doubles around the edges probed, %ld on the wrong side of their edge\n", probed, wobbles);
    return bad || wobbles ? 1 : 0;
}
