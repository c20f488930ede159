// k1_bench.hip -- K1 alone on the chip: the 32-points-per-lane kernel (fft_mag_p32_kernel) against the radix-16 kernel it
// replaces, same random chunk, same window / twiddles / prefilter levels.  Prints per variant the mean launch time
// (HIP events around `reps` back-to-back launches) and the traffic rate, and compares magnitudes (bit for bit) and
// candidate lists (as sets).  Build: see the Makefile target `ubench` ... or
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I iridium-sniffer_amd/csrc -I include tools/ubench/k1_bench.hip \
//         -L iridium-sniffer_amd -lirdm_hip -Wl,-rpath,'$ORIGIN/../../iridium-sniffer_amd' -o tools/ubench/k1_bench
// Usage: k1_bench [log_n=13] [frames=8192] [reps=20] [fmt=2] [hits_per_frame=200]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "kernels.hpp"
#include "host_design.hpp"

using namespace irdm;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

int main(int argc, char **argv)
{
    const int log_n = argc > 1 ? atoi(argv[1]) : 13;
    const int frames = argc > 2 ? atoi(argv[2]) : 8192;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    const int fmt = argc > 4 ? atoi(argv[4]) : 2;
    const int hits = argc > 5 ? atoi(argv[5]) : 200;
    const int n = 1 << log_n, cap = 4096;
    const size_t ns = (size_t)n * frames;
    const int bps = fmt == 2 ? 8 : (fmt == 1 ? 4 : 2);

    std::vector<float> window = design_blackman(n);
    for (auto &w : window) w /= 0.42f;
    std::vector<cfloat> tw = design_twiddles(n);

    std::mt19937 rng(7);
    std::vector<unsigned char> h_iq(ns * bps);
    if (fmt == 2) {
        std::normal_distribution<float> g(0.0f, 0.002f);
        float *f = reinterpret_cast<float *>(h_iq.data());
        for (size_t i = 0; i < 2 * ns; i++) f[i] = g(rng);
    } else {
        for (auto &b : h_iq) b = (unsigned char)(rng() & 0xff);
    }

    void *d_iq;
    float *d_win, *d_mag[2], *d_pre;
    float2 *d_tw;
    unsigned *d_cnt[2];
    ListEntry *d_ent[2];
    CK(hipMalloc(&d_iq, ns * bps));
    CK(hipMalloc(&d_win, sizeof(float) * n));
    CK(hipMalloc(&d_tw, sizeof(float2) * n / 2));
    CK(hipMalloc(&d_pre, sizeof(float) * n));
    for (int v = 0; v < 2; v++) {
        CK(hipMalloc(&d_mag[v], sizeof(float) * ns));
        CK(hipMalloc(&d_cnt[v], sizeof(unsigned) * frames));
        CK(hipMalloc(&d_ent[v], sizeof(ListEntry) * (size_t)frames * cap));
        CK(hipMemset(d_mag[v], 0xff, sizeof(float) * ns));
    }
    CK(hipMemcpy(d_iq, h_iq.data(), ns * bps, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_win, window.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tw, tw.data(), sizeof(float2) * n / 2, hipMemcpyHostToDevice));

    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    // prefilter levels: a first pass without lists, then the level that about `hits` bins of a frame exceed
    std::vector<float> pre(n, 3.0e38f);
    CK(hipMemcpy(d_pre, pre.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    g_fft_kernel = 0;
    if (launch_fft_mag(log_n, fmt, d_iq, d_win, d_tw, d_mag[0], frames, st) != 0) return 3;
    CK(hipStreamSynchronize(st));
    {
        std::vector<float> row(n);
        CK(hipMemcpy(row.data(), d_mag[0] + (size_t)(frames / 2) * n, sizeof(float) * n, hipMemcpyDeviceToHost));
        std::vector<float> s = row;
        std::sort(s.begin(), s.end());
        const float level = hits > 0 ? s[std::max(0, n - 1 - hits)] : 3.0e38f;
        for (auto &p : pre) p = level;
        CK(hipMemcpy(d_pre, pre.data(), sizeof(float) * n, hipMemcpyHostToDevice));
        printf("# n %d frames %d fmt %d reps %d; prefilter level %.3e (about %d candidates per frame)\n", n, frames, fmt, reps, level, hits);
    }

    const double traffic = (double)ns * bps + (double)ns * 4;          // samples read + magnitudes written
    double ms_v[2] = { 0, 0 };
    for (int v = 0; v < 2; v++) {
        g_fft_kernel = v;
        for (int lists = 0; lists < 2; lists++) {
            auto go = [&]() {
                if (lists) {
                    CK(hipMemsetAsync(d_cnt[v], 0, sizeof(unsigned) * frames, st));
                    return launch_fft_mag_lists(log_n, fmt, d_iq, d_win, d_tw, d_mag[v], frames, d_pre, d_cnt[v], d_ent[v], cap, st);
                }
                return launch_fft_mag(log_n, fmt, d_iq, d_win, d_tw, d_mag[v], frames, st);
            };
            for (int i = 0; i < 3; i++)
                if (go() != 0) return 3;
            CK(hipStreamSynchronize(st));
            float best = 1e30f, total = 0;
            for (int i = 0; i < reps; i++) {
                if (lists) CK(hipMemsetAsync(d_cnt[v], 0, sizeof(unsigned) * frames, st));
                CK(hipEventRecord(e0, st));
                const int rc = lists ? launch_fft_mag_lists(log_n, fmt, d_iq, d_win, d_tw, d_mag[v], frames, d_pre, d_cnt[v], d_ent[v], cap, st)
                                     : launch_fft_mag(log_n, fmt, d_iq, d_win, d_tw, d_mag[v], frames, st);
                if (rc != 0) return 3;
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
                total += ms;
            }
            const double mean = total / reps;
            printf("%-4s lists %d: mean %.4f ms  best %.4f ms  traffic %.2f TB/s (best %.2f)\n", v ? "p32" : "r16", lists, mean, best,
                   traffic / (mean * 1e-3) / 1e12, traffic / (best * 1e-3) / 1e12);
            if (lists) ms_v[v] = mean;
        }
    }
    g_fft_kernel = 1;

    // compare
    std::vector<float> m0(ns), m1(ns);
    CK(hipMemcpy(m0.data(), d_mag[0], sizeof(float) * ns, hipMemcpyDeviceToHost));
    CK(hipMemcpy(m1.data(), d_mag[1], sizeof(float) * ns, hipMemcpyDeviceToHost));
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < ns; i++)
        if (memcmp(&m0[i], &m1[i], 4) != 0) {
            if (!bad) first = i;
            bad++;
        }
    printf("magnitudes: %zu of %zu differ%s\n", bad, ns, bad ? "" : " (bit-identical)");
    if (bad) printf("  first at frame %zu bin %zu: r16 %.9g p32 %.9g\n", first / n, first % n, m0[first], m1[first]);
    std::vector<unsigned> c0(frames), c1(frames);
    CK(hipMemcpy(c0.data(), d_cnt[0], sizeof(unsigned) * frames, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c1.data(), d_cnt[1], sizeof(unsigned) * frames, hipMemcpyDeviceToHost));
    std::vector<ListEntry> l0((size_t)frames * cap), l1((size_t)frames * cap);
    CK(hipMemcpy(l0.data(), d_ent[0], sizeof(ListEntry) * l0.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(l1.data(), d_ent[1], sizeof(ListEntry) * l1.size(), hipMemcpyDeviceToHost));
    size_t lbad = 0, total_c = 0;
    for (int f = 0; f < frames; f++) {
        total_c += c1[f];
        if (c0[f] != c1[f]) {
            lbad++;
            continue;
        }
        const unsigned k = std::min<unsigned>(c0[f], cap);
        auto key = [](const ListEntry &a, const ListEntry &b) { return a.bin < b.bin; };
        std::sort(l0.begin() + (size_t)f * cap, l0.begin() + (size_t)f * cap + k, key);
        std::sort(l1.begin() + (size_t)f * cap, l1.begin() + (size_t)f * cap + k, key);
        if (memcmp(&l0[(size_t)f * cap], &l1[(size_t)f * cap], sizeof(ListEntry) * k) != 0) lbad++;
    }
    printf("lists: %zu of %d frames differ; %.1f candidates per frame\n", lbad, frames, (double)total_c / frames);
    printf("speed-up with lists: %.3f\n", ms_v[1] > 0 ? ms_v[0] / ms_v[1] : 0.0);
    return (bad || lbad) ? 1 : 0;
}
