// Exhaustive host check of csrc/od_sincos.h (the descriptor kernel's sin / cos): every float in [0, 2 pi] must round to the same
// float as the oracle's (float)sin((double)x), (float)cos((double)x).  Built and run by tests/test_sincos_exhaustive.py.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "../orb-slam3_amd/csrc/od_sincos.h"

int main() {
    const float top = 6.2831860f;                               // > 360 * (float)(pi / 180)
    uint32_t ub; memcpy(&ub, &top, 4);
    std::atomic<unsigned long long> bad{0};
    const int T = 8;
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
        unsigned long long b = 0;
        for (uint64_t u = t; u <= ub; u += T) {
            const uint32_t v = (uint32_t)u; float x; memcpy(&x, &v, 4);
            double s, c; od_sincos(x, &s, &c);
            const float fs = (float)s, fc = (float)c, rs = (float)std::sin((double)x), rc = (float)std::cos((double)x);
            if (memcmp(&fs, &rs, 4) || memcmp(&fc, &rc, 4)) { if (b < 3) printf("x=%a: %a %a, expected %a %a\n", x, fs, fc, rs, rc); ++b; }
        }
        bad += b; });
    for (auto& t : th) t.join();
    printf("floats %u mismatches %llu\n", ub + 1u, bad.load());
    return bad.load() ? 1 : 0;
}
