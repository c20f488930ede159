// common.hpp -- shared device/host helpers for the gfx950 Iridium hot path.
//
// Arithmetic contract (DESIGN.md "Arithmetic contract"): every float operation
// on the data path is a separately rounded IEEE-754 binary32 add/sub/mul/div,
// in the order the reference's scalar C code performs it.  This translation
// unit is compiled with -ffp-contract=off so hipcc never fuses a*b+c.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// The detector's kernels (prefilter, band scan) are chains of small, latency-bound launches that decide when the next
// chunk may be scanned; their wavefronts share SIMDs with the per-burst chains' long-running ones (decimator, demod),
// and the issue arbiter serves the oldest wavefront first.  s_setprio raises the wavefront's issue priority (0..3).
#define IRDM_DETECTOR_PRIO() __builtin_amdgcn_s_setprio(3)

#define IRDM_HIP_CHECK(expr)                                                        \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            fprintf(stderr, "irdm_hip: %s failed: %s (%s:%d)\n", #expr,             \
                    hipGetErrorString(_e), __FILE__, __LINE__);                     \
            return -1;                                                              \
        }                                                                           \
    } while (0)

#define IRDM_HIP_CHECK_NULL(expr)                                                   \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            fprintf(stderr, "irdm_hip: %s failed: %s (%s:%d)\n", #expr,             \
                    hipGetErrorString(_e), __FILE__, __LINE__);                     \
            return nullptr;                                                         \
        }                                                                           \
    } while (0)

namespace irdm {

// C99 Annex G complex product for finite operands, as GCC emits it for
// `float complex * float complex` (rotator.h:38-39, burst_downmix.c:554-555,
// qpsk_demod.c:153): (a+bi)(c+di) = (ac - bd) + (ad + bc)i, four rounded products.
__host__ __device__ __forceinline__ float2 cmul(float2 x, float2 y)
{
    float ac = x.x * y.x, bd = x.y * y.y, ad = x.x * y.y, bc = x.y * y.x;
    return make_float2(ac - bd, ad + bc);
}

__host__ __device__ __forceinline__ float mag2(float2 v)
{
    float a = v.x * v.x, b = v.y * v.y;
    return a + b;
}

// |v|^2 as the reference's AVX2 kernels round it (simd_avx2.c:196-197, :316: _mm_fmadd_ps(re, re, _mm_mul_ps(im, im))):
// im*im rounded, the sum fused.  mag2_simd: the form option "fir_order" selects (0 generic, 1 AVX2).
__host__ __device__ __forceinline__ float mag2_fma(float2 v)
{
    const float b = v.y * v.y;
    return __builtin_fmaf(v.x, v.x, b);
}
__host__ __device__ __forceinline__ float mag2_simd(float2 v, int order) { return order ? mag2_fma(v) : mag2(v); }

// glibc 2.35 cabsf/hypotf for finite inputs: sqrt in double of the exactly
// representable double sum of squares, rounded once to float
// (sysdeps/ieee754/flt-32/e_hypotf.c).  Exercised against the oracle (host libm cabsf) through every
// stage-B / stage-C parity test: start detection maxima, correlation peaks and the PLL would differ otherwise.
__host__ __device__ __forceinline__ float cabs_f(float2 v)
{
    double x = (double)v.x, y = (double)v.y;
    return (float)sqrt(x * x + y * y);
}

// ---- kernel clock (option "kernel_clock"): first wavefront in / last wavefront out of a launch ----
// A record is kKClkWords 64-bit words in device memory: [0..63] earliest entry per slot, [64..127] latest exit per slot
// (10 ns ticks of the 100 MHz wall clock, s_memrealtime; slot = workgroup index mod 64 so that the atomics of a
// chip-filling grid do not queue on one address), [128] sum of the launches' spans, [129] launches, [130] last span.
// kclk_fold_kernel (downmix.hip), enqueued behind the kernel on its stream, folds the slots into the sums and re-arms
// them.  What bench.py's roofline divides the algorithmic bytes by: the kernel's own span on the device, free of the
// dispatch wait a host-side event bracket includes.
constexpr int kKClkWords = 136;
__device__ __forceinline__ void kclk_enter(unsigned long long *k)
{
    if (k && (threadIdx.x & 63) == 0) atomicMin(&k[blockIdx.x & 63], (unsigned long long)wall_clock64());
}
__device__ __forceinline__ void kclk_leave(unsigned long long *k)
{
    if (k && (threadIdx.x & 63) == 0) atomicMax(&k[64 + (blockIdx.x & 63)], (unsigned long long)wall_clock64());
}

// ---- pinned FFT: radix-2 decimation in time (DESIGN.md "Pinned FFT") ----
// The data sits in LDS in bit-reversed order on entry and natural order on exit.
// tw[k] = (float)cos(2 pi k/N), (float)(-sin(2 pi k/N)), k < N/2, tw[0]=(1,0),
// tw[N/4]=(0,-1) exact; W.b is the four-product form; the two exact twiddles are
// applied as copies/swaps.  DIR=-1 forward, +1 backward (conjugated twiddles).
template <int LOGN, int NT, int DIR>
__device__ __forceinline__ void fft_lds_radix2(float2 *s, const float2 *__restrict__ tw)
{
    constexpr int N = 1 << LOGN;
    const int tid = threadIdx.x;
#pragma unroll 1
    for (int st = 1; st <= LOGN; st++) {
        const int half = 1 << (st - 1);
        const int sh = LOGN - st;
        for (int b = tid; b < N / 2; b += NT) {
            const int j = b & (half - 1);
            const int i0 = ((b >> (st - 1)) << st) + j;
            const int i1 = i0 + half;
            const int tix = j << sh;
            float2 a = s[i0], v = s[i1], t;
            if (tix == 0) {
                t = v;
            } else if (tix == N / 4) {
                t = DIR < 0 ? make_float2(v.y, -v.x) : make_float2(-v.y, v.x);
            } else {
                float2 w = tw[tix];
                if (DIR > 0) w.y = -w.y;
                t = cmul(w, v);
            }
            s[i0] = make_float2(a.x + t.x, a.y + t.y);
            s[i1] = make_float2(a.x - t.x, a.y - t.y);
        }
        __syncthreads();
    }
}

// One butterfly of the pinned FFT (the body of fft_lds_radix2's inner loop): stage twiddle index tix of N points.
template <int N, int DIR>
__device__ __forceinline__ void fft_bfly2(float2 &a, float2 &v, int tix, const float2 *__restrict__ tw)
{
    float2 t;
    if (tix == 0) {
        t = v;
    } else if (tix == N / 4) {
        t = DIR < 0 ? make_float2(v.y, -v.x) : make_float2(-v.y, v.x);
    } else {
        float2 w = tw[tix];
        if (DIR > 0) w.y = -w.y;
        t = cmul(w, v);
    }
    const float2 a0 = a;
    a = make_float2(a0.x + t.x, a0.y + t.y);
    v = make_float2(a0.x - t.x, a0.y - t.y);
}

// The same transform from stage FIRST on (the stages before it done by the caller), TWO stages per barrier: a thread
// takes the four points base + {0, h, 2h, 3h} (h = the first stage's butterfly span) through stage st -- (0, h) and
// (2h, 3h), one twiddle -- and stage st + 1 -- (0, 2h) and (h, 3h) -- in registers: the same butterflies on the same
// operands as fft_lds_radix2, half the LDS round trips and barriers.  NA arrays side by side (s + a * pitch): the two
// backward transforms of the sync correlation share their barriers.  An odd number of stages ends with a single one.
template <int LOGN, int NT, int DIR, int FIRST, int NA = 1>
__device__ __forceinline__ void fft_lds_radix2x2(float2 *s, const float2 *__restrict__ tw, int pitch = 0)
{
    constexpr int N = 1 << LOGN;
    const int tid = threadIdx.x;
    int st = FIRST;
#pragma unroll 1
    for (; st + 1 <= LOGN; st += 2) {
        const int h = 1 << (st - 1);
        const int sh = LOGN - st;
        for (int g = tid; g < NA * (N / 4); g += NT) {
            float2 *sa = s + (g / (N / 4)) * pitch;
            const int q = g % (N / 4);
            const int j = q & (h - 1);
            const int i0 = ((q >> (st - 1)) << (st + 1)) + j;
            float2 e0 = sa[i0], e1 = sa[i0 + h], e2 = sa[i0 + 2 * h], e3 = sa[i0 + 3 * h];
            fft_bfly2<N, DIR>(e0, e1, j << sh, tw);
            fft_bfly2<N, DIR>(e2, e3, j << sh, tw);
            fft_bfly2<N, DIR>(e0, e2, j << (sh - 1), tw);
            fft_bfly2<N, DIR>(e1, e3, (j + h) << (sh - 1), tw);
            sa[i0] = e0; sa[i0 + h] = e1; sa[i0 + 2 * h] = e2; sa[i0 + 3 * h] = e3;
        }
        __syncthreads();
    }
    if (st <= LOGN) {
        const int half = 1 << (st - 1);
        const int sh = LOGN - st;
        for (int g = tid; g < NA * (N / 2); g += NT) {
            float2 *sa = s + (g / (N / 2)) * pitch;
            const int b = g % (N / 2);
            const int j = b & (half - 1);
            const int i0 = ((b >> (st - 1)) << st) + j;
            float2 a = sa[i0], v = sa[i0 + half];
            fft_bfly2<N, DIR>(a, v, j << sh, tw);
            sa[i0] = a; sa[i0 + half] = v;
        }
        __syncthreads();
    }
}

__host__ __device__ __forceinline__ unsigned bitrev(unsigned v, int bits)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(v) >> (32 - bits);
#else
    unsigned r = 0;
    for (int b = 0; b < bits; b++)
        if (v & (1u << b)) r |= 1u << (bits - 1 - b);
    return r;
#endif
}


// One IQ sample of the device-resident input in the configured format -> cf32, exactly as the reference ingests it:
//   fmt 2: cf32 as is (burst_detector_feed_cf32, burst_detect.c:846)
//   fmt 0: ci8, int8 / 128.0f (simd_convert_i8_cf, simd_generic.c:147-153)
//   fmt 1: ci16, narrowed to int8 by (int8_t)(v >> 8) (spewer_thread, main.c:245-246), then as ci8
template <int FMT>
__device__ __forceinline__ float2 load_iq(const void *__restrict__ iq, size_t i)
{
    if (FMT == 2) {
        return reinterpret_cast<const float2 *>(iq)[i];
    } else if (FMT == 1) {
        const short2 v = reinterpret_cast<const short2 *>(iq)[i];
        return make_float2((float)(v.x >> 8) / 128.0f, (float)(v.y >> 8) / 128.0f);
    } else {
        const char2 v = reinterpret_cast<const char2 *>(iq)[i];
        return make_float2((float)v.x / 128.0f, (float)v.y / 128.0f);
    }
}
__device__ __forceinline__ float2 load_iq(int fmt, const void *__restrict__ iq, size_t i)
{
    return fmt == 2 ? load_iq<2>(iq, i) : (fmt == 1 ? load_iq<1>(iq, i) : load_iq<0>(iq, i));
}

}  // namespace irdm
