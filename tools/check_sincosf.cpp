// check_sincosf.cpp -- csrc/libm_port.hpp against the host's own libm, every float in [-lim, lim] (default 2.0: the
// fine-CFO step stays within |x| <= pi/2), both builds of the routine (with / without FMA contraction), and the
// constants against the table inside the host's libm.so.6.
//   g++ -O2 -std=c++17 -mfma -ffp-contract=off -pthread -o check_sincosf tools/check_sincosf.cpp -lm && ./check_sincosf [lim]
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <thread>
#include <vector>

#include "../iridium-sniffer_amd/csrc/libm_port.hpp"

extern "C" float _Complex cexpf(float _Complex);

static inline uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(int argc, char **argv)
{
    const float lim = argc > 1 ? (float)atof(argv[1]) : 2.0f;
    const uint32_t top = bits(lim);
    const int n_thr = (int)std::thread::hardware_concurrency() > 0 ? (int)std::thread::hardware_concurrency() : 4;
    std::atomic<unsigned long long> bad_fused{0}, bad_plain{0}, total{0};
    std::vector<std::thread> th;
    for (int t = 0; t < n_thr; t++)
        th.emplace_back([&, t] {
            unsigned long long bf = 0, bp = 0, n = 0;
            for (uint64_t u = t; u <= top; u += n_thr) {
                for (int sgn = 0; sgn < 2; sgn++) {
                    const uint32_t w = (uint32_t)u | (sgn ? 0x80000000u : 0u);
                    float y;
                    memcpy(&y, &w, 4);
                    float _Complex a;
                    __real__ a = 0.0f;
                    __imag__ a = y;
                    const float _Complex z = cexpf(a);
                    const float hr = __real__ z, hi = __imag__ z;
                    float r, i;
                    irdm::libm_cexpf_i<true>(y, &r, &i);
                    if (bits(r) != bits(hr) || bits(i) != bits(hi)) bf++;
                    irdm::libm_cexpf_i<false>(y, &r, &i);
                    if (bits(r) != bits(hr) || bits(i) != bits(hi)) bp++;
                    n++;
                }
            }
            bad_fused += bf;
            bad_plain += bp;
            total += n;
        });
    for (auto &x : th) x.join();
    printf("cexpf(i*x), every float with |x| <= %g: %llu values; mismatches vs the host libm: fused build %llu, plain build %llu\n",
           (double)lim, total.load(), bad_fused.load(), bad_plain.load());
    printf("host libm runs the %s build\n", bad_fused == 0 ? "fused (FMA)" : bad_plain == 0 ? "plain" : "UNKNOWN");
    return (bad_fused == 0 || bad_plain == 0) ? 0 : 1;
}
