// host_design.hpp -- create-time constants computed on the host with the host libm
// (so they are bit-identical to what the reference computes on the same machine)
// and uploaded once: windows, filter taps, FFT twiddles, sync-word templates,
// per-bin rotator increments.
#pragma once
#include <complex>
#include <vector>

namespace irdm {

typedef std::complex<float> cfloat;

// window_func.c:19-24
std::vector<float> design_blackman(int n);
// fir_filter.c:143-182 / :74-111 / :115-139 / :186-193
std::vector<float> design_lpf(float gain, float sample_rate, float cutoff, float transition);
std::vector<float> design_rrc(float gain, float sample_rate, float symbol_rate, float alpha, int ntaps);
std::vector<float> design_rc(float sample_rate, float symbol_rate, float alpha, int ntaps);
std::vector<float> design_box(int length);

// pinned FFT twiddles tw[k], k < n/2 (DESIGN.md "Pinned FFT")
std::vector<cfloat> design_twiddles(int n);
// pinned radix-2 DIT FFT on the host (used only for the create-time sync templates)
void host_fft(std::vector<cfloat> &data, const std::vector<cfloat> &tw, int dir);

// burst_downmix.c:138-219: FFT of the reversed, conjugated, RC-shaped preamble+UW
std::vector<cfloat> design_sync_template(const std::vector<float> &rc, int corr_n, float sps,
                                         bool uplink, int *sync_len);

// cexpf(-2 pi ((bin - n/2)/n) i) per detector bin (burst_downmix.c:663-669)
std::vector<cfloat> design_rotator_incr(int n);
// cexpf(-2 pi offset i) (burst_downmix.c:716-717)
cfloat fine_rotator_incr(float center_offset);

}  // namespace irdm
