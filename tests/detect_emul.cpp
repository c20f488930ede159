// detect_emul.cpp -- TEST INFRASTRUCTURE: csrc/detect.hip as the GPU build compiles it -- K1 (window, N-point FFT in
// registers and LDS, fftshift, |.|^2, the candidate lists written in its store stage; the radix-2 LDS kernel of the small sizes), and
// the dense sequential detector scan -- on the CPU emulation of tests/hip_emul/hip/hip_runtime.h, with the product's own
// window and twiddle designs (csrc/host_design.cpp), against the oracle (tests/test_kernels_emul.py).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>

#include "detect_emul.inc"
// (csrc/host_design.cpp is built by hipcc = clang, which has __builtin_complex in C++; g++ spells it with __real__ / __imag__)
#define __builtin_complex(re, im) ({ float _Complex z_; __real__ z_ = (re); __imag__ z_ = (im); z_; })
#include "host_design.cpp"

using namespace irdm;

extern "C" {

// K1 over n_frames frames of `iq` (fmt 2 cf32, 1 ci16, 0 ci8): mag [n_frames][n] by the kernel the size takes (8192 / 16384
// points: 32 points per lane; 4096: radix 16; up to 2048: the radix-2 LDS kernel); variant 1: with the band scan's candidate
// lists (pre [n] given; counts [n_frames], entries [n_frames][cap]).  order: fftshift_mag's AVX2 (1) / generic (0) form.
// Returns 0, or 1 if this FFT size has no such kernel.
int detect_emul_k1(const void *iq, int fmt, int n, int n_frames, int variant, float *mag, const float *pre,
                   unsigned *counts, ListEntry *entries, int cap)
{
    const int log_n = 31 - __builtin_clz((unsigned)n);
    std::vector<float> window = design_blackman(n);
    for (int i = 0; i < n; i++) window[i] /= 0.42f;               // burst_detect.c:249-250, as csrc/create.cpp prepares it
    std::vector<cfloat> tw = design_twiddles(n);
    const float2 *tw2 = reinterpret_cast<const float2 *>(tw.data());
    const int order = (variant & 8) ? 0 : 1;
    if ((variant & 3) == 1) {
        memset(counts, 0, sizeof(unsigned) * n_frames);
        return launch_fft_mag_lists(log_n, fmt, iq, window.data(), tw2, mag, n_frames, pre, counts, entries, cap, nullptr, nullptr, order);
    }
    return launch_fft_mag(log_n, fmt, iq, window.data(), tw2, mag, n_frames, nullptr, nullptr, order);
}

// the dense sequential scan (detect_scan_kernel) over the whole magnitude plane from the first frame of a stream, in
// chunks of chunk_frames; returns the finished bursts in emission order
int detect_emul_scan(const float *mag, int n_frames, int n, int pre_len, int post_len, int width, int max_bursts, int max_len,
                     float threshold, int chunk_frames, GoneBurst *out, int out_cap, float *sum_out)
{
    DetParams D;
    D.n = n;
    D.log_n = 31 - __builtin_clz((unsigned)n);
    D.pre_len = pre_len;
    D.post_len = post_len;
    D.width = width;
    D.max_bursts = max_bursts;
    D.max_len = max_len;
    D.threshold = threshold;
    std::vector<DetState> st_store(1);
    DetState *st = st_store.data();
    memset(st, 0, sizeof(DetState));
    std::vector<float> sum(n, 0.0f), hist((size_t)kHistory * n, 0.0f);
    std::vector<PeakCand> ca(n), cb(n);
    const int gone_cap = 8192;
    std::vector<GoneBurst> gone(gone_cap);
    int total = 0;
    for (int f0 = 0; f0 < n_frames; f0 += chunk_frames) {
        const int F = std::min(chunk_frames, n_frames - f0);
        st->n_gone = 0;
        if (launch_detect_scan(D, st, sum.data(), hist.data(), mag + (size_t)f0 * n, F, gone.data(), gone_cap, ca.data(),
                               cb.data(), nullptr) != 0)
            return -3;
        if (st->overflow) return -4;
        for (uint32_t i = 0; i < st->n_gone; i++) {
            if (total >= out_cap) return -2;
            out[total++] = gone[i];
        }
    }
    memcpy(sum_out, sum.data(), sizeof(float) * n);
    return total;
}

}
