/*
 * irdm_oracle.c -- CPU ORACLE (test infrastructure, never shipped, never on the
 * product path).  Plain-C restatement of the reference's scalar hot path:
 *   stage A  burst_detect.c   (window -> FFT -> |.|^2 -> threshold state machine -> IQ cut)
 *   stage B  burst_downmix.c  (rotate -> 801-tap FIR /M -> LPF -> start -> CFO -> RRC -> sync)
 *   stage C  qpsk_demod.c     (Gardner -> PLL -> slicer -> UW -> DQPSK -> bits/LLR)
 *   surface  frame_output.c   (RAW line)
 *   bit layer frame_decode.c / ida_decode.c (access code, de-interleave, BCH + Chase, IRA / IBC / IDA fields)
 * Citations are file:line into the reference tree.  Build with
 *   gcc -O2 -std=gnu99 -ffp-contract=off   (no FMA contraction: the reference's
 *   scalar path is built -O3 -msse4.1 without -mfma, CMakeLists.txt:6-18)
 *
 * FFT: the reference uses FFTW3 (external, absent) -> PARITY UNPINNED at that
 * boundary.  The oracle uses the pinned radix-2 DIT float32 FFT defined below.
 */
#define _GNU_SOURCE
#include <complex.h>
#include <inttypes.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "irdm_oracle.h"

typedef float complex cf;

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* =====================================================================
 * Pinned FFT: radix-2 decimation-in-time, natural-order output.
 *   X = E + W.O ; X' = E - W.O, W = tw[k] = (float)cos(2 pi k/N), (float)(-sin(2 pi k/N))
 *   computed in long double, with tw[0] = (1,0) and tw[N/4] = (0,-1) exact.
 *   W.O is the four-product form (wr*or - wi*oi, wr*oi + wi*or), every product
 *   and sum rounded to float separately; multiplications by the two exact
 *   twiddles are skipped (exact anyway).  Backward = conjugated twiddles,
 *   no normalisation (FFTW_BACKWARD convention, burst_downmix.c:347-354).
 * ===================================================================== */

typedef struct {
    int n;
    float *wr, *wi;
    int *rev;
} fft_plan_t;

#define MAX_PLANS 16
static fft_plan_t g_plans[MAX_PLANS];
static int g_nplans;

static const fft_plan_t *fft_get_plan(int n)
{
    for (int i = 0; i < g_nplans; i++)
        if (g_plans[i].n == n)
            return &g_plans[i];
    if (g_nplans >= MAX_PLANS)
        abort();
    fft_plan_t *p = &g_plans[g_nplans];
    p->n = n;
    p->wr = malloc(sizeof(float) * (n / 2 + 1));
    p->wi = malloc(sizeof(float) * (n / 2 + 1));
    p->rev = malloc(sizeof(int) * n);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < n / 2; k++) {
        long double a = two_pi * (long double)k / (long double)n;
        p->wr[k] = (float)cosl(a);
        p->wi[k] = (float)(-sinl(a));
    }
    p->wr[0] = 1.0f;
    p->wi[0] = 0.0f;
    if (n >= 4) {
        p->wr[n / 4] = 0.0f;
        p->wi[n / 4] = -1.0f;
    }
    int bits = 0;
    while ((1 << bits) < n)
        bits++;
    for (int i = 0; i < n; i++) {
        int r = 0;
        for (int b = 0; b < bits; b++)
            if (i & (1 << b))
                r |= 1 << (bits - 1 - b);
        p->rev[i] = r;
    }
    g_nplans++;
    return p;
}

void orc_fft(float *data, int n, int dir)
{
    const fft_plan_t *p = fft_get_plan(n);
    for (int i = 0; i < n; i++) {
        int r = p->rev[i];
        if (r > i) {
            float tr = data[2 * i], ti = data[2 * i + 1];
            data[2 * i] = data[2 * r];
            data[2 * i + 1] = data[2 * r + 1];
            data[2 * r] = tr;
            data[2 * r + 1] = ti;
        }
    }
    for (int m = 2; m <= n; m <<= 1) {
        int half = m >> 1;
        int step = n / m;
        for (int k = 0; k < n; k += m) {
            for (int j = 0; j < half; j++) {
                float *a = &data[2 * (k + j)];
                float *b = &data[2 * (k + j + half)];
                float br = b[0], bi = b[1];
                float tr, ti;
                int tix = j * step;
                if (tix == 0) {
                    tr = br;
                    ti = bi;
                } else if (4 * tix == n) {
                    if (dir < 0) { tr = bi; ti = -br; }
                    else         { tr = -bi; ti = br; }
                } else {
                    float wr = p->wr[tix];
                    float wi = dir < 0 ? p->wi[tix] : -p->wi[tix];
                    float p0 = wr * br, p1 = wi * bi, p2 = wr * bi, p3 = wi * br;
                    tr = p0 - p1;
                    ti = p2 + p3;
                }
                float ar = a[0], ai = a[1];
                a[0] = ar + tr;
                a[1] = ai + ti;
                b[0] = ar - tr;
                b[1] = ai - ti;
            }
        }
    }
}

/* =====================================================================
 * DSP primitives (simd_generic.c:76-178 scalar variants)
 * ===================================================================== */

/* simd_generic.c:76-84: acc += taps[k]*in[i+k], k ascending, real x complex */
void orc_fir_ccf(const float *taps, int ntaps, const float *in, float *out, int n)
{
    for (int i = 0; i < n; i++) {
        float ar = 0.0f, ai = 0.0f;
        const float *p = in + 2 * (size_t)i;
        for (int k = 0; k < ntaps; k++) {
            float t = taps[k];
            ar += t * p[2 * k];
            ai += t * p[2 * k + 1];
        }
        out[2 * (size_t)i] = ar;
        out[2 * (size_t)i + 1] = ai;
    }
}

/* simd_generic.c:86-96 */
void orc_fir_ccf_dec(const float *taps, int ntaps, const float *in, float *out,
                     int n_out, int decimation)
{
    for (int i = 0; i < n_out; i++) {
        float ar = 0.0f, ai = 0.0f;
        const float *p = in + 2 * (size_t)i * (size_t)decimation;
        for (int k = 0; k < ntaps; k++) {
            float t = taps[k];
            ar += t * p[2 * k];
            ai += t * p[2 * k + 1];
        }
        out[2 * (size_t)i] = ar;
        out[2 * (size_t)i + 1] = ai;
    }
}

/* simd_avx2.c:62-108 (avx2_fir_ccf_dec) -- the decimating FIR the reference runs on an x86 host with AVX2 + FMA unless
 * --no-simd is given (simd_init, simd_generic.c:33-57; main.c:567): the taps four at a time into FOUR accumulators
 * with fused multiply-adds (_mm256_fmadd_ps: acc_j = fma(t[4m+j], x[4m+j], acc_j), m ascending), the horizontal sum
 * (a0 + a2) + (a1 + a3) (:89-96), then the remaining ntaps % 4 taps one by one with a separately rounded product and
 * sum (:102-105; the file is compiled -std=c99, CMakeLists.txt:6, so GCC does not contract the scalar tail --
 * oracle/Makefile compiles the reference file with those flags and tests/test_oracle_vs_ref.py pins this function to
 * it bit for bit).  This translation unit is compiled -ffp-contract=off -mfma: fmaf() is the fused instruction, nothing
 * else is fused. */
void orc_fir_ccf_dec_avx2(const float *taps, int ntaps, const float *in, float *out,
                          int n_out, int decimation)
{
    for (int i = 0; i < n_out; i++) {
        const float *p = in + 2 * (size_t)i * (size_t)decimation;
        float re[4] = { 0.0f, 0.0f, 0.0f, 0.0f }, im[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        int k = 0;
        for (; k + 3 < ntaps; k += 4)
            for (int j = 0; j < 4; j++) {
                re[j] = fmaf(taps[k + j], p[2 * (k + j)], re[j]);
                im[j] = fmaf(taps[k + j], p[2 * (k + j) + 1], im[j]);
            }
        float ar = (re[0] + re[2]) + (re[1] + re[3]);
        float ai = (im[0] + im[2]) + (im[1] + im[3]);
        for (; k < ntaps; k++) {
            ar += taps[k] * p[2 * k];
            ai += taps[k] * p[2 * k + 1];
        }
        out[2 * (size_t)i] = ar;
        out[2 * (size_t)i + 1] = ai;
    }
}

/* Which form of the dispatched kernels (simd_kernels.h) stages A and B use (orc_detector_*, orc_downmix_*,
 * orc_run_stream): 0 = simd_generic.c (what --no-simd selects, and every non-x86 host), 1 = simd_avx2.c (the reference's
 * default on x86: simd_init, simd_generic.c:33-57).  The two differ in arithmetic in five kernels -- fir_ccf_dec (above),
 * fir_ccf, fir_fff, fftshift_mag and mag_squared (below: fused multiply-adds in the vector body, the generic form in the
 * scalar tail) -- and agree bit for bit in the rest (window_cf, baseline_update, relative_mag, convert_i8_cf, max_float:
 * the same operations; csquare_window: a*b + b*a == 2*(a*b) exactly).  Both are the reference's arithmetic; the product
 * has the matching option "fir_order" (alias "simd_order").  Set before any stream is run (read-only afterwards). */
static int g_orc_fir_order = 0;
void orc_set_fir_order(int order) { g_orc_fir_order = order ? 1 : 0; }
int orc_get_fir_order(void) { return g_orc_fir_order; }

/* simd_avx2.c:28-55 (avx2_fir_ccf): outputs four at a time, acc = fma(taps[k], in[i + k], acc) for k ascending (:36-40);
 * the last n % 4 outputs in the generic form (:45-54, compiled -std=c99: not contracted) */
void orc_fir_ccf_avx2(const float *taps, int ntaps, const float *in, float *out, int n)
{
    int i = 0;
    for (; i + 3 < n; i += 4)
        for (int j = 0; j < 4; j++) {
            const float *p = in + 2 * (size_t)(i + j);
            float ar = 0.0f, ai = 0.0f;
            for (int k = 0; k < ntaps; k++) {
                ar = fmaf(taps[k], p[2 * k], ar);
                ai = fmaf(taps[k], p[2 * k + 1], ai);
            }
            out[2 * (size_t)(i + j)] = ar;
            out[2 * (size_t)(i + j) + 1] = ai;
        }
    if (i < n) orc_fir_ccf(taps, ntaps, in + 2 * (size_t)i, out + 2 * (size_t)i, n - i);
}

/* simd_generic.c:98-106 */
void orc_fir_fff(const float *taps, int ntaps, const float *in, float *out, int n)
{
    for (int i = 0; i < n; i++) {
        float acc = 0.0f;
        for (int k = 0; k < ntaps; k++)
            acc += taps[k] * in[i + k];
        out[i] = acc;
    }
}

/* simd_avx2.c:115-138 (avx2_fir_fff): outputs eight at a time with fused multiply-adds, the last n % 8 generic */
void orc_fir_fff_avx2(const float *taps, int ntaps, const float *in, float *out, int n)
{
    int i = 0;
    for (; i + 7 < n; i += 8)
        for (int j = 0; j < 8; j++) {
            float acc = 0.0f;
            for (int k = 0; k < ntaps; k++)
                acc = fmaf(taps[k], in[i + j + k], acc);
            out[i + j] = acc;
        }
    if (i < n) orc_fir_fff(taps, ntaps, in + i, out + i, n - i);
}

/* simd_generic.c:108-112: complex x real = two real products */
void orc_window_cf(const float *samples, const float *window, float *out, int n)
{
    for (int i = 0; i < n; i++) {
        out[2 * i] = samples[2 * i] * window[i];
        out[2 * i + 1] = samples[2 * i + 1] * window[i];
    }
}

/* simd_generic.c:114-127: mag[i] = |X[(i + N/2) % N]|^2, re*re + im*im (two roundings + add) */
void orc_fftshift_mag(const float *fft_out, float *mag_shifted, int fft_size)
{
    int half = fft_size / 2;
    for (int i = 0; i < fft_size; i++) {
        int src = (i + half) & (fft_size - 1);
        float re = fft_out[2 * src], im = fft_out[2 * src + 1];
        float a = re * re, b = im * im;
        mag_shifted[i] = a + b;
    }
}

/* simd_avx2.c:177-221 (avx2_fftshift_mag): fma(re, re, im*im) -- the product im*im rounded, the sum fused (:196-197,
 * :205-206); half is a multiple of four for every FFT size of the path, the generic tail (:210-220) for any other */
void orc_fftshift_mag_avx2(const float *fft_out, float *mag_shifted, int fft_size)
{
    int half = fft_size / 2, vec = half & ~3;
    for (int i = 0; i < half; i++) {
        for (int side = 0; side < 2; side++) {
            int src = side ? i : half + i, dst = side ? half + i : i;
            float re = fft_out[2 * src], im = fft_out[2 * src + 1];
            float b = im * im;
            if (i < vec)
                mag_shifted[dst] = fmaf(re, re, b);
            else {
                float a = re * re;
                mag_shifted[dst] = a + b;
            }
        }
    }
}

/* simd_generic.c:129-135: (sum - old) + new, two separate float ops */
void orc_baseline_update(float *sum, const float *old_hist, const float *new_mag, int n)
{
    for (int i = 0; i < n; i++) {
        float s = sum[i] - old_hist[i];
        sum[i] = s + new_mag[i];
    }
}

/* simd_generic.c:137-145 */
void orc_relative_mag(const float *mag, const float *baseline, float *out, int n)
{
    for (int i = 0; i < n; i++)
        out[i] = baseline[i] > 0 ? mag[i] / baseline[i] : 0.0f;
}

/* simd_generic.c:147-153 */
void orc_convert_i8_cf(const int8_t *iq, float *out, size_t n)
{
    for (size_t i = 0; i < 2 * n; i++)
        out[i] = iq[i] / 128.0f;
}

/* simd_generic.c:155-162 */
void orc_mag_squared(const float *in, float *out, int n)
{
    for (int i = 0; i < n; i++) {
        float a = in[2 * i] * in[2 * i], b = in[2 * i + 1] * in[2 * i + 1];
        out[i] = a + b;
    }
}

/* simd_avx2.c:304-323 (avx2_mag_squared): fma(re, re, im*im) four at a time, the last n % 4 generic */
void orc_mag_squared_avx2(const float *in, float *out, int n)
{
    int vec = n & ~3;
    for (int i = 0; i < vec; i++)
        out[i] = fmaf(in[2 * i], in[2 * i], in[2 * i + 1] * in[2 * i + 1]);
    if (vec < n) orc_mag_squared(in + 2 * (size_t)vec, out + vec, n - vec);
}

/* simd_generic.c:164-170 */
float orc_max_float(const float *in, int n)
{
    float m = -1e30f;
    for (int i = 0; i < n; i++)
        if (in[i] > m)
            m = in[i];
    return m;
}

/* C99 Annex G complex product as GCC emits it for finite operands:
 * (a+bi)(c+di) = (ac - bd) + (ad + bc)i, four rounded products. */
static inline cf cmul(cf x, cf y)
{
    float a = crealf(x), b = cimagf(x), c = crealf(y), d = cimagf(y);
    float ac = a * c, bd = b * d, ad = a * d, bc = b * c;
    return CMPLXF(ac - bd, ad + bc);
}

/* simd_generic.c:172-178: (s*s)*w */
void orc_csquare_window(const float *in, const float *window, float *out, int n)
{
    for (int i = 0; i < n; i++) {
        cf s = CMPLXF(in[2 * i], in[2 * i + 1]);
        cf q = cmul(s, s);
        out[2 * i] = crealf(q) * window[i];
        out[2 * i + 1] = cimagf(q) * window[i];
    }
}

/* window_func.c:19-24 */
void orc_blackman_window(float *w, int n)
{
    for (int i = 0; i < n; i++)
        w[i] = 0.42f - 0.5f * cosf(2.0f * (float)M_PI * i / (n - 1))
                     + 0.08f * cosf(4.0f * (float)M_PI * i / (n - 1));
}

/* fir_filter.c:67-70 */
static float sinc_f(float x)
{
    if (fabsf(x) < 1e-10f)
        return 1.0f;
    return sinf((float)M_PI * x) / ((float)M_PI * x);
}

/* fir_filter.c:143-182: windowed sinc, 4-term Blackman-Harris, unity DC gain */
int orc_lpf_taps(float *out, int cap, float gain, float sample_rate, float cutoff,
                 float transition)
{
    int ntaps = (int)(4.0f / (transition / sample_rate));
    ntaps |= 1;
    if (ntaps > cap)
        return -1;
    int center = ntaps / 2;
    float omega_c = 2.0f * (float)M_PI * cutoff / sample_rate;
    float total = 0;
    for (int i = 0; i < ntaps; i++) {
        float n = i - center;
        float h;
        if (fabsf(n) < 1e-10f)
            h = omega_c / (float)M_PI;
        else
            h = sinf(omega_c * n) / ((float)M_PI * n);
        float w = 0.35875f
                - 0.48829f * cosf(2.0f * (float)M_PI * i / (ntaps - 1))
                + 0.14128f * cosf(4.0f * (float)M_PI * i / (ntaps - 1))
                - 0.01168f * cosf(6.0f * (float)M_PI * i / (ntaps - 1));
        out[i] = h * w;
        total += out[i];
    }
    if (fabsf(total) > 0) {
        float scale = gain / total;
        for (int i = 0; i < ntaps; i++)
            out[i] *= scale;
    }
    return ntaps;
}

/* fir_filter.c:74-111 */
int orc_rrc_taps(float *out, int cap, float gain, float sample_rate, float symbol_rate,
                 float alpha, int ntaps)
{
    ntaps |= 1;
    if (ntaps > cap)
        return -1;
    float sps = sample_rate / symbol_rate;
    int center = ntaps / 2;
    float energy = 0;
    for (int i = 0; i < ntaps; i++) {
        float t = (i - center) / sps;
        if (fabsf(t) < 1e-10f) {
            out[i] = (1.0f - alpha + 4.0f * alpha / (float)M_PI);
        } else if (fabsf(fabsf(t) - 1.0f / (4.0f * alpha)) < 1e-6f) {
            out[i] = alpha / sqrtf(2.0f) *
                ((1.0f + 2.0f / (float)M_PI) * sinf((float)M_PI / (4.0f * alpha)) +
                 (1.0f - 2.0f / (float)M_PI) * cosf((float)M_PI / (4.0f * alpha)));
        } else {
            float num = sinf((float)M_PI * t * (1.0f - alpha)) +
                        4.0f * alpha * t * cosf((float)M_PI * t * (1.0f + alpha));
            float den = (float)M_PI * t * (1.0f - (4.0f * alpha * t) * (4.0f * alpha * t));
            out[i] = num / den;
        }
        energy += out[i] * out[i];
    }
    float scale = gain / sqrtf(energy);
    for (int i = 0; i < ntaps; i++)
        out[i] *= scale;
    return ntaps;
}

/* fir_filter.c:115-139 */
int orc_rc_taps(float *out, int cap, float sample_rate, float symbol_rate, float alpha,
                int ntaps)
{
    ntaps |= 1;
    if (ntaps > cap)
        return -1;
    float sps = sample_rate / symbol_rate;
    int center = ntaps / 2;
    for (int i = 0; i < ntaps; i++) {
        float t = (i - center) / sps;
        if (fabsf(t) < 1e-10f) {
            out[i] = 1.0f;
        } else if (alpha > 0 && fabsf(fabsf(t) - 1.0f / (2.0f * alpha)) < 1e-6f) {
            out[i] = (float)M_PI / (4.0f) * sinc_f(1.0f / (2.0f * alpha));
        } else {
            float cos_term = cosf((float)M_PI * alpha * t);
            float den = 1.0f - (2.0f * alpha * t) * (2.0f * alpha * t);
            out[i] = sinc_f(t) * cos_term / den;
        }
    }
    return ntaps;
}

/* fir_filter.c:186-193 */
int orc_box_taps(float *out, int cap, int length)
{
    if (length > cap)
        return -1;
    float v = 1.0f / length;
    for (int i = 0; i < length; i++)
        out[i] = v;
    return length;
}

/* rotator.h:36-46: out = in*phase; phase *= incr (sequential float recurrence),
 * one renormalisation after the loop */
void orc_rotator_rotate_n(float *phase, const float *incr, float *out, const float *in, int n)
{
    cf ph = CMPLXF(phase[0], phase[1]);
    cf inc = CMPLXF(incr[0], incr[1]);
    for (int i = 0; i < n; i++) {
        cf x = CMPLXF(in[2 * i], in[2 * i + 1]);
        cf y = cmul(x, ph);
        out[2 * i] = crealf(y);
        out[2 * i + 1] = cimagf(y);
        ph = cmul(ph, inc);
    }
    float mag = cabsf(ph);
    if (mag > 0)
        ph = CMPLXF(crealf(ph) / mag, cimagf(ph) / mag);
    phase[0] = crealf(ph);
    phase[1] = cimagf(ph);
}

/* =====================================================================
 * Stage A: burst detector (burst_detect.c)
 * ===================================================================== */

#define IR_DEFAULT_BURST_WIDTH 40000    /* iridium.h:40 */
#define IR_DEFAULT_THRESHOLD 16.0f      /* iridium.h:37 */
#define IR_DEFAULT_HISTORY 512          /* iridium.h:46 */

typedef struct {
    uint64_t id, start, stop, last_active;
    int center_bin;
    float magnitude, noise, peak_rel, base_sum;
} ab_t;

typedef struct { int bin; float rel; } pk_t;

struct orc_detector {
    double center_frequency;
    int sample_rate, n, pre_len, post_len, width, max_bursts, max_len, hist;
    float threshold;
    float *window;
    float *history;      /* [hist][n] */
    float *sum;          /* [n] */
    int hist_idx, primed;
    float *mag, *rel, *mask;
    cf *fft_buf;
    ab_t *act; int n_act, cap_act;
    ab_t *gone; int n_gone, cap_gone;
    pk_t *peaks; int n_peaks;
    pk_t *peaks_tmp;
    uint64_t burst_id, tagged, sample_count, index;
    int squelch;
    cf *ring; size_t ring_size, ring_w; uint64_t ring_start;
    cf *conv; size_t conv_cap;
    float *mag_sink; size_t mag_sink_cap; size_t frames_done;
};

static void ab_push(ab_t **arr, int *n, int *cap, const ab_t *b)
{
    if (*n >= *cap) {
        *cap = *cap ? *cap * 2 : 64;
        *arr = realloc(*arr, sizeof(ab_t) * (size_t)*cap);
    }
    (*arr)[(*n)++] = *b;
}

/* burst_detect.c:174-323 (derived parameters) */
orc_detector_t *orc_detector_create(double center_frequency, int sample_rate,
                                    float threshold_db, int fft_size)
{
    orc_detector_t *d = calloc(1, sizeof(*d));
    d->center_frequency = center_frequency;
    d->sample_rate = sample_rate;
    if (fft_size > 0) {
        d->n = fft_size;
    } else {
        int lg = (int)round(log2(sample_rate / 1000.0));   /* :184 */
        d->n = 1 << lg;
    }
    d->pre_len = 2 * d->n;                                  /* :191 */
    d->post_len = (int)(sample_rate * 16e-3);               /* :195 */
    d->width = IR_DEFAULT_BURST_WIDTH / (sample_rate / d->n);   /* :201 integer division */
    d->max_bursts = (int)((sample_rate / (float)IR_DEFAULT_BURST_WIDTH) * 0.8f);  /* :207 */
    d->max_len = (int)(sample_rate * 0.09);                 /* :213 */
    d->hist = IR_DEFAULT_HISTORY;
    float tdb = threshold_db > 0 ? threshold_db : IR_DEFAULT_THRESHOLD;
    float enbw = 1.72f;
    d->threshold = powf(10.0f, tdb / 10.0f) / d->hist / enbw;   /* :226 */

    d->window = malloc(sizeof(float) * (size_t)d->n);
    orc_blackman_window(d->window, d->n);
    for (int i = 0; i < d->n; i++)
        d->window[i] /= 0.42f;                              /* :249-250 */

    d->history = calloc((size_t)d->n * d->hist, sizeof(float));
    d->sum = calloc((size_t)d->n, sizeof(float));
    d->mag = calloc((size_t)d->n, sizeof(float));
    d->rel = calloc((size_t)d->n, sizeof(float));
    d->mask = malloc(sizeof(float) * (size_t)d->n);
    for (int i = 0; i < d->n; i++)
        d->mask[i] = 1.0f;
    d->fft_buf = malloc(sizeof(cf) * (size_t)d->n);
    d->peaks = malloc(sizeof(pk_t) * (size_t)d->n);
    d->peaks_tmp = malloc(sizeof(pk_t) * (size_t)d->n);

    d->ring_size = (size_t)d->max_len + d->pre_len + d->post_len + (size_t)d->n * 4;  /* :292-293 */
    if (d->ring_size < (size_t)(2 * sample_rate))
        d->ring_size = 2 * (size_t)sample_rate;             /* :295-296 */
    /* the reference mallocs the ring (:297); fresh pages read as zero */
    d->ring = calloc(d->ring_size, sizeof(cf));
    return d;
}

void orc_detector_destroy(orc_detector_t *d)
{
    if (!d)
        return;
    free(d->window); free(d->history); free(d->sum); free(d->mag); free(d->rel);
    free(d->mask); free(d->fft_buf); free(d->peaks); free(d->peaks_tmp);
    free(d->act); free(d->gone); free(d->ring); free(d->conv);
    free(d);
}

int orc_detector_fft_size(const orc_detector_t *d) { return d->n; }
uint64_t orc_detector_tagged(const orc_detector_t *d) { return d->tagged; }
const float *orc_detector_baseline_sum(const orc_detector_t *d) { return d->sum; }
const float *orc_detector_last_magnitude(const orc_detector_t *d) { return d->mag; }
size_t orc_detector_frames_done(const orc_detector_t *d) { return d->frames_done; }
void orc_detector_set_mag_sink(orc_detector_t *d, float *sink, size_t max_frames)
{
    d->mag_sink = sink;
    d->mag_sink_cap = max_frames;
}

/* burst_detect.c:679-687: window, FFT, fftshift+|.|^2 */
void orc_detector_magnitude_frame(orc_detector_t *d, const float *iq_frame, float *mag_out)
{
    float *buf = (float *)d->fft_buf;
    orc_window_cf(iq_frame, d->window, buf, d->n);
    orc_fft(buf, d->n, -1);
    (g_orc_fir_order ? orc_fftshift_mag_avx2 : orc_fftshift_mag)(buf, mag_out, d->n);
}

/* burst_detect.c:438-454 */
static void det_update_post(orc_detector_t *d, int force)
{
    if (d->n_act == 0 || force) {
        float *h = d->history + (size_t)d->hist_idx * d->n;
        orc_baseline_update(d->sum, h, d->mag, d->n);
        memcpy(h, d->mag, sizeof(float) * d->n);
        if (++d->hist_idx == d->hist) {
            d->primed = 1;
            d->hist_idx = 0;
        }
    }
}

/* burst_detect.c:473-486 */
static void det_mask_one(orc_detector_t *d, int center_bin)
{
    int lo = center_bin - d->width / 2;
    int hi = center_bin + d->width / 2;
    if (lo < 0) lo = 0;
    if (hi >= d->n) hi = d->n - 1;
    for (int i = lo; i <= hi; i++)
        d->mask[i] = 0.0f;
}

static void det_rebuild_mask(orc_detector_t *d)
{
    for (int i = 0; i < d->n; i++)
        d->mask[i] = 1.0f;
    for (int i = 0; i < d->n_act; i++)
        det_mask_one(d, d->act[i].center_bin);
}

/* stable merge sort, descending by rel: glibc qsort() is a stable merge sort
 * (msort.c) whenever its temporary buffer can be allocated, so ties keep
 * ascending-bin order (burst_detect.c:164-170, :551) */
static void peaks_sort_desc(pk_t *a, pk_t *tmp, int n)
{
    if (n < 2)
        return;
    int h = n / 2;
    peaks_sort_desc(a, tmp, h);
    peaks_sort_desc(a + h, tmp, n - h);
    int i = 0, j = h, k = 0;
    while (i < h && j < n) {
        if (a[j].rel > a[i].rel)
            tmp[k++] = a[j++];
        else
            tmp[k++] = a[i++];
    }
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, sizeof(pk_t) * (size_t)n);
}

/* burst_detect.c:637-651 / :689-698: state machine on one magnitude frame (in d->mag) */
static void det_state_machine(orc_detector_t *d)
{
    const int n = d->n;
    const float thr = d->threshold;
    if (d->primed) {
        orc_relative_mag(d->mag, d->sum, d->rel, n);               /* :426-434 */

        for (int i = 0; i < d->n_act; i++) {                         /* :458-469 */
            int cb = d->act[i].center_bin;
            if ((cb > 0 && d->rel[cb - 1] > thr) || d->rel[cb] > thr ||
                (cb < n - 1 && d->rel[cb + 1] > thr))
                d->act[i].last_active = d->index;
        }

        for (int i = 0; i < n; i++)                                  /* :522-525 */
            d->rel[i] *= d->mask[i];

        /* :529-552 */
        d->n_peaks = 0;
        int half_bw = d->width / 2;
        int dc = n / 2;
        for (int bin = half_bw; bin < n - half_bw; bin++) {
            if (bin >= dc - 3 && bin <= dc + 3)
                continue;
            if (d->rel[bin] > thr) {
                d->peaks[d->n_peaks].bin = bin;
                d->peaks[d->n_peaks].rel = d->rel[bin];
                d->n_peaks++;
            }
        }
        peaks_sort_desc(d->peaks, d->peaks_tmp, d->n_peaks);

        /* :490-518 */
        int force_noise = 0;
        for (int i = 0; i < d->n_act;) {
            ab_t *b = &d->act[i];
            int too_long = 0;
            if (d->max_len > 0 && b->last_active - b->start > (uint64_t)d->max_len) {
                force_noise = 1;
                too_long = 1;
            }
            if (b->last_active + (uint64_t)d->post_len <= d->index || too_long) {
                b->stop = d->index;
                ab_push(&d->gone, &d->n_gone, &d->cap_gone, b);
                memmove(b, b + 1, sizeof(ab_t) * (size_t)(d->n_act - i - 1));
                d->n_act--;
            } else {
                i++;
            }
        }
        if (force_noise)
            det_update_post(d, 1);

        det_rebuild_mask(d);                                         /* :482-486 */

        /* :556-632 */
        for (int i = 0; i < d->n_peaks; i++) {
            const pk_t *p = &d->peaks[i];
            if (d->mask[p->bin] == 0.0f)
                continue;
            ab_t b;
            memset(&b, 0, sizeof(b));
            b.id = d->burst_id;
            b.center_bin = p->bin;
            d->burst_id += 10;
            b.peak_rel = p->rel;
            b.magnitude = 10.0f * log10f(p->rel * d->hist * 1.72f);
            b.start = d->index - (uint64_t)d->pre_len;
            b.last_active = b.start;
            b.base_sum = d->sum[b.center_bin];
            b.noise = 10.0f * log10f(d->sum[b.center_bin] / d->hist
                                     / ((float)n * n)
                                     / 1.72f
                                     / ((float)d->sample_rate / n));
            ab_push(&d->act, &d->n_act, &d->cap_act, &b);
            det_mask_one(d, b.center_bin);
        }
        if (d->max_bursts > 0 && d->n_act > d->max_bursts) {         /* squelch :594-627 */
            for (int i = 0; i < d->n_act; i++) {
                if (d->act[i].start != d->index - (uint64_t)d->pre_len) {
                    d->act[i].stop = d->index;
                    ab_push(&d->gone, &d->n_gone, &d->cap_gone, &d->act[i]);
                }
            }
            d->n_act = 0;
            det_rebuild_mask(d);
            d->squelch += 3;
            if (d->squelch >= 10) {
                d->hist_idx = 0;
                d->primed = 0;
                memset(d->history, 0, sizeof(float) * (size_t)n * d->hist);
                memset(d->sum, 0, sizeof(float) * n);
                d->squelch = 0;
            }
        } else if (d->squelch > 0) {
            d->squelch--;
        }
    }
    det_update_post(d, 0);                                           /* :698 */
}

static void det_process_frame(orc_detector_t *d, const cf *samples)
{
    orc_detector_magnitude_frame(d, (const float *)samples, d->mag);
    if (d->mag_sink && d->frames_done < d->mag_sink_cap)
        memcpy(d->mag_sink + d->frames_done * d->n, d->mag, sizeof(float) * d->n);
    d->frames_done++;
    det_state_machine(d);
}

/* burst_detect.c:703-742 with :401-422 */
static void det_emit(orc_detector_t *d, orc_burst_cb cb, void *user)
{
    for (int i = 0; i < d->n_gone; i++) {
        const ab_t *b = &d->gone[i];
        uint64_t start = b->start;
        uint64_t stop = b->stop + (uint64_t)d->pre_len;
        if (start < d->ring_start)
            start = d->ring_start;
        if (stop <= start)
            continue;
        size_t len = (size_t)(stop - start);
        cf *buf = malloc(sizeof(cf) * len);
        size_t pos = (size_t)(start % d->ring_size);
        for (size_t k = 0; k < len; k++) {
            buf[k] = d->ring[pos];
            if (++pos == d->ring_size)
                pos = 0;
        }
        orc_burst_rec_t rec;
        memset(&rec, 0, sizeof(rec));
        rec.id = b->id;
        rec.start = b->start;
        rec.stop = b->stop;
        rec.last_active = b->last_active;
        rec.center_bin = b->center_bin;
        rec.magnitude = b->magnitude;
        rec.noise = b->noise;
        rec.peak_rel = b->peak_rel;
        rec.base_sum = b->base_sum;
        rec.num_samples = len;
        rec.avail_end = d->sample_count;
        if (cb)
            cb(&rec, (const float *)buf, user);
        free(buf);
        d->tagged++;
    }
    d->n_gone = 0;
}

/* burst_detect.c:773-774 / :866-867 then :821-836 / :905-920, then :840-841 */
static void det_feed_common(orc_detector_t *d, const cf *samples, size_t n,
                            orc_burst_cb cb, void *user)
{
    for (size_t i = 0; i < n; i++) {                     /* :390-394 */
        d->ring[d->ring_w] = samples[i];
        if (++d->ring_w == d->ring_size)
            d->ring_w = 0;
    }
    /* :396-398 -- note: evaluated with the sample_count *before* this block is added */
    if (d->sample_count > d->ring_size)
        d->ring_start = d->sample_count - d->ring_size;
    d->sample_count += n;

    cf *tmp = NULL;
    while (d->index + (uint64_t)d->n <= d->sample_count) {
        size_t pos = (size_t)(d->index % d->ring_size);
        if (pos + (size_t)d->n <= d->ring_size) {
            det_process_frame(d, &d->ring[pos]);
        } else {
            if (!tmp)
                tmp = malloc(sizeof(cf) * (size_t)d->n);
            size_t first = d->ring_size - pos;
            memcpy(tmp, &d->ring[pos], first * sizeof(cf));
            memcpy(tmp + first, d->ring, ((size_t)d->n - first) * sizeof(cf));
            det_process_frame(d, tmp);
        }
        d->index += (uint64_t)d->n;
    }
    free(tmp);
    if (d->n_gone > 0)
        det_emit(d, cb, user);
}

static void det_reserve_conv(orc_detector_t *d, size_t n)
{
    if (n > d->conv_cap) {
        free(d->conv);
        d->conv_cap = n;
        d->conv = malloc(sizeof(cf) * n);
    }
}

void orc_detector_feed_cf32(orc_detector_t *d, const float *iq, size_t n,
                            orc_burst_cb cb, void *user)
{
    det_reserve_conv(d, n);
    memcpy(d->conv, iq, sizeof(cf) * n);                 /* :862-863 */
    det_feed_common(d, d->conv, n, cb, user);
}

void orc_detector_feed_i8(orc_detector_t *d, const int8_t *iq, size_t n,
                          orc_burst_cb cb, void *user)
{
    det_reserve_conv(d, n);
    orc_convert_i8_cf(iq, (float *)d->conv, n);          /* :770 */
    det_feed_common(d, d->conv, n, cb, user);
}

/* =====================================================================
 * Stage B: burst downmix (burst_downmix.c)
 * ===================================================================== */

#define IR_SYMBOLS_PER_SECOND 25000
#define IR_UW_LENGTH 12
#define IR_PREAMBLE_SHORT 16
#define IR_PREAMBLE_LONG 64
#define IR_SIMPLEX_FREQ_MIN 1626000000
static const int UW_DL[IR_UW_LENGTH] = { 0, 2, 2, 2, 2, 0, 0, 0, 2, 0, 0, 2 };   /* iridium.h:30 */
static const int UW_UL[IR_UW_LENGTH] = { 2, 2, 0, 0, 0, 2, 0, 0, 2, 0, 2, 2 };   /* iridium.h:31 */

#define DM_WORK (2 * 1024 * 1024)     /* burst_downmix.c:366 */

struct orc_downmix {
    int out_rate, search_depth, pre_start;
    float sps;
    float *in_taps; int in_ntaps;
    float *noise_taps; int noise_ntaps;
    float *start_taps; int start_ntaps;
    float *rrc_taps; int rrc_ntaps;
    float *rc_taps; int rc_ntaps;
    int cfo_n, cfo_total;
    float *cfo_window;
    int corr_n, sync_search;
    cf *dl_fft, *ul_fft; int dl_len, ul_len;
    cf *wa, *wb, *fftbuf, *ifft_dl, *ifft_ul;
    float *magf, *magfilt;
};

/* burst_downmix.c:138-219 */
static void dm_make_sync(orc_downmix_t *dm, const int *uw, int preamble_len, int uplink,
                         cf **fft_out, int *len_out)
{
    cf s0 = CMPLXF(1.0f, 1.0f), s1 = CMPLXF(-1.0f, -1.0f);
    int total = preamble_len + IR_UW_LENGTH;
    cf *sym = calloc(total, sizeof(cf));
    for (int i = 0; i < preamble_len; i++)
        sym[i] = uplink ? ((i % 2 == 0) ? s1 : s0) : s0;
    for (int i = 0; i < IR_UW_LENGTH; i++)
        sym[preamble_len + i] = uw[i] == 0 ? s0 : s1;

    int isps = (int)roundf(dm->sps);
    int plen = total * isps - (isps - 1);
    int half = (dm->rc_ntaps - 1) / 2;
    cf *buf = calloc(plen + dm->rc_ntaps - 1, sizeof(cf));
    for (int i = 0; i < total; i++)
        buf[half + i * isps] = sym[i];
    free(sym);
    cf *shaped = malloc(sizeof(cf) * plen);
    orc_fir_ccf(dm->rc_taps, dm->rc_ntaps, (const float *)buf, (float *)shaped, plen);
    free(buf);

    /* reverse + conjugate (:189-195) */
    cf *tmpl = calloc(dm->corr_n, sizeof(cf));
    for (int i = 0; i < plen && i < dm->corr_n; i++)
        tmpl[i] = conjf(shaped[plen - 1 - i]);
    free(shaped);
    orc_fft((float *)tmpl, dm->corr_n, -1);
    *fft_out = tmpl;
    *len_out = plen;
}

/* burst_downmix.c:223-373 */
orc_downmix_t *orc_downmix_create(void)
{
    orc_downmix_t *dm = calloc(1, sizeof(*dm));
    dm->out_rate = 10 * IR_SYMBOLS_PER_SECOND;                       /* :233 */
    dm->sps = (float)dm->out_rate / IR_SYMBOLS_PER_SECOND;           /* :236 */
    dm->search_depth = dm->out_rate;                                 /* :237-238 */
    dm->pre_start = (int)(100 * 1e-6f * dm->out_rate);               /* :241 */

    dm->in_taps = malloc(sizeof(float) * 1024);
    dm->in_ntaps = orc_lpf_taps(dm->in_taps, 1024, 1.0f, 10000000.0f,
                                dm->out_rate * 0.4f, dm->out_rate * 0.2f);   /* :252-258 */
    dm->noise_taps = malloc(sizeof(float) * 256);
    dm->noise_ntaps = orc_lpf_taps(dm->noise_taps, 256, 1.0f, (float)dm->out_rate,
                                   40000.0f / 2.0f, 40000.0f);       /* :265-268 */
    int box = (int)(dm->sps * 2);                                    /* :281-282 */
    if (box < 3) box = 3;
    dm->start_taps = malloc(sizeof(float) * 64);
    dm->start_ntaps = orc_box_taps(dm->start_taps, 64, box);
    dm->rrc_taps = malloc(sizeof(float) * 64);
    dm->rrc_ntaps = orc_rrc_taps(dm->rrc_taps, 64, 1.0f, (float)dm->out_rate,
                                 (float)IR_SYMBOLS_PER_SECOND, 0.4f, 51);    /* :292-293 */
    dm->rc_taps = malloc(sizeof(float) * 64);
    dm->rc_ntaps = orc_rc_taps(dm->rc_taps, 64, (float)dm->out_rate,
                               (float)IR_SYMBOLS_PER_SECOND, 0.4f, 51);      /* :301-302 */

    int raw = (int)(dm->sps * 26);                                   /* :310-313 */
    dm->cfo_n = 1;
    while (dm->cfo_n * 2 <= raw)
        dm->cfo_n *= 2;
    dm->cfo_total = dm->cfo_n * 16;                                  /* :55, :315 */
    dm->cfo_window = malloc(sizeof(float) * dm->cfo_n);
    orc_blackman_window(dm->cfo_window, dm->cfo_n);                  /* :321 */

    dm->sync_search = (int)((IR_PREAMBLE_LONG + IR_UW_LENGTH + 8) * dm->sps);   /* :324-325 */
    int ul_samples = (int)((IR_PREAMBLE_SHORT + IR_UW_LENGTH) * dm->sps);       /* :328-329 */
    dm->corr_n = 1;
    while (dm->corr_n < dm->sync_search + ul_samples)
        dm->corr_n <<= 1;                                            /* :330 */

    dm_make_sync(dm, UW_DL, IR_PREAMBLE_SHORT, 0, &dm->dl_fft, &dm->dl_len);    /* :358-360 */
    dm_make_sync(dm, UW_UL, IR_PREAMBLE_SHORT, 1, &dm->ul_fft, &dm->ul_len);    /* :361-363 */

    dm->wa = malloc(sizeof(cf) * DM_WORK);
    dm->wb = malloc(sizeof(cf) * DM_WORK);
    dm->magf = malloc(sizeof(float) * DM_WORK);
    dm->magfilt = malloc(sizeof(float) * DM_WORK);
    int big = dm->cfo_total > dm->corr_n ? dm->cfo_total : dm->corr_n;
    dm->fftbuf = malloc(sizeof(cf) * big);
    dm->ifft_dl = malloc(sizeof(cf) * dm->corr_n);
    dm->ifft_ul = malloc(sizeof(cf) * dm->corr_n);
    return dm;
}

void orc_downmix_destroy(orc_downmix_t *dm)
{
    if (!dm)
        return;
    free(dm->in_taps); free(dm->noise_taps); free(dm->start_taps); free(dm->rrc_taps);
    free(dm->rc_taps); free(dm->cfo_window); free(dm->dl_fft); free(dm->ul_fft);
    free(dm->wa); free(dm->wb); free(dm->magf); free(dm->magfilt); free(dm->fftbuf);
    free(dm->ifft_dl); free(dm->ifft_ul);
    free(dm);
}

const float *orc_downmix_taps(const orc_downmix_t *dm, int which, int *ntaps)
{
    switch (which) {
    case 0: *ntaps = dm->in_ntaps; return dm->in_taps;
    case 1: *ntaps = dm->noise_ntaps; return dm->noise_taps;
    case 2: *ntaps = dm->start_ntaps; return dm->start_taps;
    case 3: *ntaps = dm->rrc_ntaps; return dm->rrc_taps;
    default: *ntaps = dm->rc_ntaps; return dm->rc_taps;
    }
}

const float *orc_downmix_sync_fft(const orc_downmix_t *dm, int uplink, int *sync_len)
{
    *sync_len = uplink ? dm->ul_len : dm->dl_len;
    return (const float *)(uplink ? dm->ul_fft : dm->dl_fft);
}

const float *orc_downmix_cfo_window(const orc_downmix_t *dm, int *n)
{
    *n = dm->cfo_n;
    return dm->cfo_window;
}

/* burst_downmix.c:441-478 */
static int dm_find_start(orc_downmix_t *dm, const cf *frame, int frame_len)
{
    int search = dm->search_depth;
    if (search > frame_len) search = frame_len;
    int mag_len = search + dm->start_ntaps - 1;
    if (mag_len > frame_len) mag_len = frame_len;
    (g_orc_fir_order ? orc_mag_squared_avx2 : orc_mag_squared)((const float *)frame, dm->magf, mag_len);
    int half = (dm->start_ntaps - 1) / 2;
    int flen = mag_len - dm->start_ntaps + 1;
    if (flen <= 0) return 0;
    if (flen > search) flen = search;
    (g_orc_fir_order ? orc_fir_fff_avx2 : orc_fir_fff)(dm->start_taps, dm->start_ntaps, dm->magf, dm->magfilt, flen);
    float mx = orc_max_float(dm->magfilt, flen);
    float thr = 0.45f * mx;
    int start = 0;
    for (start = 0; start < flen; start++)
        if (dm->magfilt[start] >= thr)
            break;
    if (start > 0) {
        start = start + half - dm->pre_start;
        if (start < 0) start = 0;
    }
    return start;
}

static float parabolic(float alpha, float beta, float gamma)
{
    float denom = alpha - 2.0f * beta + gamma;
    if (fabsf(denom) > 1e-10f)
        return 0.5f * (alpha - gamma) / denom;
    return 0;
}

static inline float mag2(cf v)
{
    float re = crealf(v), im = cimagf(v);
    float a = re * re, b = im * im;
    return a + b;
}

/* burst_downmix.c:482-535 */
static float dm_fine_cfo(orc_downmix_t *dm, const cf *frame, int frame_len)
{
    int n = dm->cfo_n;
    if (n > frame_len) n = frame_len;
    int tot = dm->cfo_total;
    memset(dm->fftbuf, 0, sizeof(cf) * tot);
    orc_csquare_window((const float *)frame, dm->cfo_window, (float *)dm->fftbuf, n);
    orc_fft((float *)dm->fftbuf, tot, -1);
    float best = 0;
    int bi = 0;
    for (int i = 0; i < tot; i++) {
        float m = mag2(dm->fftbuf[i]);
        if (m > best) { best = m; bi = i; }
    }
    int idx = bi >= tot / 2 ? bi - tot : bi;
    float corr = 0;
    if (bi > 0 && bi < tot - 1) {
        int im1 = idx - 1 < 0 ? idx - 1 + tot : idx - 1;
        int ip1 = idx + 1 < 0 ? idx + 1 + tot : idx + 1;
        corr = parabolic(mag2(dm->fftbuf[im1]), best, mag2(dm->fftbuf[ip1]));
    }
    return (idx + corr) / tot / 2.0f;
}

/* burst_downmix.c:539-639 */
static int dm_correlate(orc_downmix_t *dm, const cf *frame, int frame_len, int *direction,
                        float *correction, cf *peak)
{
    int sl = dm->sync_search;
    if (sl > frame_len) sl = frame_len;
    int nn = dm->corr_n;
    memset(dm->fftbuf, 0, sizeof(cf) * nn);
    memcpy(dm->fftbuf, frame, sizeof(cf) * sl);
    orc_fft((float *)dm->fftbuf, nn, -1);
    for (int i = 0; i < nn; i++) {
        dm->ifft_dl[i] = cmul(dm->fftbuf[i], dm->dl_fft[i]);
        dm->ifft_ul[i] = cmul(dm->fftbuf[i], dm->ul_fft[i]);
    }
    orc_fft((float *)dm->ifft_dl, nn, +1);
    orc_fft((float *)dm->ifft_ul, nn, +1);
    float mdl = 0, mul = 0;
    int odl = 0, oul = 0;
    for (int i = 0; i < sl; i++) {
        float m = mag2(dm->ifft_dl[i]);
        if (m > mdl) { mdl = m; odl = i; }
    }
    for (int i = 0; i < sl; i++) {
        float m = mag2(dm->ifft_ul[i]);
        if (m > mul) { mul = m; oul = i; }
    }
    int off, slen;
    const cf *res;
    if (mdl >= mul) { *direction = 1; off = odl; res = dm->ifft_dl; slen = dm->dl_len; }
    else            { *direction = 2; off = oul; res = dm->ifft_ul; slen = dm->ul_len; }
    *peak = res[off];
    float c = 0;
    if (off > 0 && off < sl - 1)
        c = parabolic(mag2(res[off - 1]), mag2(res[off]), mag2(res[off + 1]));
    *correction = c;
    int pre_off = off - slen + 1;
    int pre_syms = (*direction == 1) ? IR_PREAMBLE_SHORT : 32;          /* :633-634 */
    return pre_off + (int)(pre_syms * dm->sps);
}

/* burst_downmix.c:643-797 */
int orc_downmix_process(orc_downmix_t *dm, const orc_burst_rec_t *rec, const float *samples,
                        double center_frequency, int sample_rate, int fft_size,
                        uint64_t start_time_ns, orc_frame_t *out)
{
    memset(out, 0, offsetof(orc_frame_t, samples));
    out->id = rec->id;
    if (rec->num_samples < 100) { out->drop_reason = 1; return 0; }
    int n = (int)rec->num_samples;
    if (n > DM_WORK) n = DM_WORK;
    memcpy(dm->wa, samples, sizeof(cf) * (size_t)n);

    uint64_t timestamp = start_time_ns +
        (uint64_t)((double)rec->start / sample_rate * 1e9);             /* :659-660 */

    /* step 1 (:663-672) */
    float rel_freq = (rec->center_bin - fft_size / 2) / (float)fft_size;
    {
        float inc = -2.0f * (float)M_PI * rel_freq;
        cf e = cexpf(inc * I);
        float ph[2] = { 1.0f, 0.0f };
        float in2[2] = { crealf(e), cimagf(e) };
        orc_rotator_rotate_n(ph, in2, (float *)dm->wa, (const float *)dm->wa, n);
        center_frequency += rel_freq * sample_rate;
    }

    /* step 2 (:417-437) */
    int decim = (int)roundf((float)sample_rate / dm->out_rate);
    if (decim < 1) decim = 1;
    int dec_len = (n - dm->in_ntaps + 1) / decim;
    if (dec_len <= 0) dec_len = 0;
    if (dec_len > DM_WORK) dec_len = DM_WORK;
    if (dec_len > 0) {
(g_orc_fir_order ? orc_fir_ccf_dec_avx2 : orc_fir_ccf_dec)(dm->in_taps, dm->in_ntaps, (const float *)dm->wa, (float *)dm->wb,
                        dec_len, decim);
        timestamp += (uint64_t)((dm->in_ntaps / 2) * 1000000000ULL / sample_rate);
    }
    out->dec_len = dec_len;
    if (dec_len < 100) { out->drop_reason = 2; return 0; }

    /* step 2b (:683-698) */
    {
        int nl = dec_len - dm->noise_ntaps + 1;
        if (nl > 0) {
            int half = (dm->noise_ntaps - 1) / 2;
            int pad = dec_len + dm->noise_ntaps - 1;
            if (pad > DM_WORK) pad = DM_WORK;
            memset(dm->wa, 0, sizeof(cf) * (size_t)pad);
            memcpy(&dm->wa[half], dm->wb, sizeof(cf) * (size_t)dec_len);
            (g_orc_fir_order ? orc_fir_ccf_avx2 : orc_fir_ccf)(dm->noise_taps, dm->noise_ntaps, (const float *)dm->wa,
                        (float *)dm->wb, dec_len);
        }
        memcpy(dm->wa, dm->wb, sizeof(cf) * (size_t)dec_len);
    }

    /* step 3 */
    int start = dm_find_start(dm, dm->wa, dec_len);
    out->start = start;
    if (start >= dec_len - 100) { out->drop_reason = 3; return 0; }
    int frame_len = dec_len - start;

    /* step 4 */
    float coff = dm_fine_cfo(dm, &dm->wa[start], frame_len);
    out->center_offset = coff;

    /* step 5 (:713-720) */
    {
        float inc = -2.0f * (float)M_PI * coff;
        cf e = cexpf(inc * I);
        float ph[2] = { 1.0f, 0.0f };
        float in2[2] = { crealf(e), cimagf(e) };
        orc_rotator_rotate_n(ph, in2, (float *)dm->wb, (const float *)&dm->wa[start], frame_len);
        center_frequency += coff * dm->out_rate;
    }

    /* step 6 (:723-734) */
    {
        int half = (dm->rrc_ntaps - 1) / 2;
        int pad = frame_len + dm->rrc_ntaps - 1;
        if (pad > DM_WORK) pad = DM_WORK;
        memset(dm->wa, 0, sizeof(cf) * (size_t)pad);
        memcpy(&dm->wa[half], dm->wb, sizeof(cf) * (size_t)frame_len);
        (g_orc_fir_order ? orc_fir_ccf_avx2 : orc_fir_ccf)(dm->rrc_taps, dm->rrc_ntaps, (const float *)dm->wa, (float *)dm->wb,
                    frame_len);
    }

    /* step 7 */
    int direction;
    float uw_corr;
    cf peak;
    int uw_start = dm_correlate(dm, dm->wb, frame_len, &direction, &uw_corr, &peak);
    out->uw_start_idx = uw_start;
    out->corr_re = crealf(peak);
    out->corr_im = cimagf(peak);
    out->direction = direction;
    if (uw_start < 0 || uw_start >= frame_len) { out->drop_reason = 4; return 0; }

    /* step 8 (:750-760): constant phase, incr = 1 */
    {
        float mag = cabsf(peak);
        cf pc = mag > 0 ? conjf(CMPLXF(crealf(peak) / mag, cimagf(peak) / mag)) : 1.0f;
        float ph[2] = { crealf(pc), cimagf(pc) };
        float in2[2] = { 1.0f, 0.0f };
        orc_rotator_rotate_n(ph, in2, (float *)dm->wa, (const float *)dm->wb, frame_len);
    }

    /* step 9 (:763-793) */
    int max_len, min_len;
    if (center_frequency > IR_SIMPLEX_FREQ_MIN) {
        max_len = (int)(444 * dm->sps);
        min_len = (int)(80 * dm->sps);
    } else {
        max_len = (int)(191 * dm->sps);
        min_len = (int)(131 * dm->sps);
    }
    int avail = frame_len - uw_start;
    if (avail < min_len) { out->drop_reason = 5; return 0; }
    int ext = avail < max_len ? avail : max_len;

    out->timestamp = timestamp + (uint64_t)((double)start / dm->out_rate * 1e9);
    out->center_frequency = center_frequency;
    out->sample_rate = (float)dm->out_rate;
    out->samples_per_symbol = dm->sps;
    out->magnitude = rec->magnitude;
    out->noise = rec->noise;
    out->uw_start = uw_corr;
    out->num_samples = ext;
    memcpy(out->samples, &dm->wa[uw_start], sizeof(cf) * (size_t)ext);
    out->drop_reason = 0;
    return 1;
}

/* =====================================================================
 * Stage C: qpsk_demod.c
 * ===================================================================== */

#define SQRT1_2F 0.70710678118654752f

/* qpsk_demod.c:56-81: Catmull-Rom; complex x real products, left-to-right sums */
static cf cubic(const cf *in, int n, float pos)
{
    int idx = (int)pos;
    float mu = pos - idx;
    if (idx < 1) idx = 1;
    if (idx >= n - 2) idx = n - 3;
    cf s0 = in[idx - 1], s1 = in[idx], s2 = in[idx + 1], s3 = in[idx + 2];
    float mu2 = mu * mu;
    float mu3 = mu2 * mu;
    cf a = -0.5f * s0 + 1.5f * s1 - 1.5f * s2 + 0.5f * s3;
    cf b = s0 - 2.5f * s1 + 2.0f * s2 - 0.5f * s3;
    cf c = -0.5f * s0 + 0.5f * s2;
    cf d = s1;
    return a * mu3 + b * mu2 + c * mu + d;
}

/* qpsk_demod.c:85-130 */
static int gardner(const cf *in, int n_samples, float sps, cf *out)
{
    int n = 0;
    float pos = 0.0f, toff = 0.0f;
    cf prev = 0;
    while (pos < n_samples - 3) {
        cf on = cubic(in, n_samples, pos);
        out[n] = on;
        if (n > 0) {
            float mid_pos = pos - sps * 0.5f;
            if (mid_pos >= 1.0f) {
                cf mid = cubic(in, n_samples, mid_pos);
                cf diff = prev - on;
                float err = crealf(cmul(diff, conjf(mid)));
                if (err > 1.0f) err = 1.0f;
                if (err < -1.0f) err = -1.0f;
                toff += 0.0002f * err;
                float adj = 0.02f * err + toff;
                if (adj > 0.5f) adj = 0.5f;
                if (adj < -0.5f) adj = -0.5f;
                pos += adj;
            }
        }
        prev = on;
        n++;
        pos += sps;
    }
    return n;
}

/* qpsk_demod.c:134-141 */
static int decim_simple(const cf *in, int n_samples, float sps, cf *out)
{
    int n = 0;
    for (int i = 0; i < n_samples; i += (int)sps)
        out[n++] = in[i];
    return n;
}

/* qpsk_demod.c:145-195 */
static float pll(const cf *in, cf *out, int n, float alpha)
{
    cf phi = CMPLXF(1.0f, 0.0f);
    float total = 0.0f;
    for (int i = 0; i < n; i++) {
        out[i] = cmul(in[i], phi);
        float re = crealf(out[i]), im = cimagf(out[i]);
        cf xh;
        if (re >= 0 && im >= 0)      xh = CMPLXF(SQRT1_2F, SQRT1_2F);
        else if (re >= 0)            xh = CMPLXF(SQRT1_2F, -SQRT1_2F);
        else if (im < 0)             xh = CMPLXF(-SQRT1_2F, -SQRT1_2F);
        else                         xh = CMPLXF(-SQRT1_2F, SQRT1_2F);
        cf er = cmul(conjf(xh), out[i]);
        float em = cabsf(er);
        if (em < 1e-10f)
            continue;
        cf unit = CMPLXF(crealf(er) / em, cimagf(er) / em);
        float ang = cargf(unit);
        float sa = alpha * ang;
        cf corr = CMPLXF(cosf(sa), sinf(sa));
        total += sa;
        phi = cmul(conjf(corr), phi);
        float pm = cabsf(phi);
        if (pm > 0)
            phi = CMPLXF(crealf(phi) / pm, cimagf(phi) / pm);
    }
    return total;
}

/* qpsk_demod.c:199-260 */
static int slicer(const cf *burst, int n_symbols, int *symbols, float *level, int *confidence)
{
    float max_mag = 0;
    int low = 0, n = 0;
    float *offs = malloc(sizeof(float) * (n_symbols + 1));
    float *mags = malloc(sizeof(float) * (n_symbols + 1));
    for (int i = 0; i < n_symbols; i++) {
        float re = crealf(burst[i]), im = cimagf(burst[i]);
        float a = re * re, b = im * im;
        float mag = sqrtf(a + b);
        mags[i] = mag;
        if (mag > max_mag) max_mag = mag;
        if (re >= 0 && im >= 0) symbols[i] = 0;
        else if (re < 0 && im >= 0) symbols[i] = 1;
        else if (re < 0) symbols[i] = 2;
        else symbols[i] = 3;
        float phase = (atan2f(im, re) + (float)M_PI) * 180.0f / (float)M_PI;
        offs[i] = 45.0f - fmodf(phase, 90.0f);
        n++;
        if (mag < max_mag / 8.0f) {
            if (++low >= 3) { n -= 3; break; }
        } else {
            low = 0;
        }
    }
    int n_ok = 0;
    float sum = 0;
    for (int i = 0; i < n; i++) {
        sum += mags[i];
        if (fabsf(offs[i]) <= 22)
            n_ok++;
    }
    *level = n > 0 ? sum / n : 0;
    *confidence = n > 0 ? (100 * n_ok) / n : 0;
    free(offs);
    free(mags);
    return n;
}

/* qpsk_demod.c:277-293 */
static int uw_hard(const int *sym, int n, int direction)
{
    if (n < IR_UW_LENGTH) return 0;
    const int *uw = direction == 1 ? UW_DL : UW_UL;
    int diffs = 0;
    for (int i = 0; i < IR_UW_LENGTH; i++) {
        int df = abs(sym[i] - uw[i]);
        if (df == 3) df = 1;
        diffs += df;
    }
    return diffs <= 2;
}

/* qpsk_demod.c:297-325 */
static float uw_soft(const cf *pll_out, int n, int direction)
{
    if (n < IR_UW_LENGTH) return 999.0f;
    const int *uw = direction == 1 ? UW_DL : UW_UL;
    float total = 0.0f;
    for (int i = 0; i < IR_UW_LENGTH; i++) {
        float expect = (float)M_PI * 0.25f + uw[i] * (float)M_PI * 0.5f;
        float actual = cargf(pll_out[i]);
        if (actual < 0) actual += 2.0f * (float)M_PI;
        float df = actual - expect;
        if (df > (float)M_PI) df -= 2.0f * (float)M_PI;
        if (df < -(float)M_PI) df += 2.0f * (float)M_PI;
        total += fabsf(df) * (float)(2.0 / M_PI);
    }
    return total;
}

/* qpsk_demod.c:393-535 */
int orc_qpsk_demod(const orc_frame_t *in, int use_gardner, orc_demod_t *out)
{
    static const int dq[4] = { 0, 2, 3, 1 };                       /* :46 */
    memset(out, 0, sizeof(*out));
    int sps = (int)(in->samples_per_symbol + 0.5f);
    if (sps < 1) sps = 1;
    int max_sym = in->num_samples / sps + 1;
    cf *dec = malloc(sizeof(cf) * (size_t)(max_sym + 8));
    cf *po = malloc(sizeof(cf) * (size_t)(max_sym + 8));
    int *sym = malloc(sizeof(int) * (size_t)(max_sym + 8));
    const cf *x = (const cf *)in->samples;

    int nsym = use_gardner ? gardner(x, in->num_samples, in->samples_per_symbol, dec)
                           : decim_simple(x, in->num_samples, in->samples_per_symbol, dec);
    float total_phase = pll(dec, po, nsym, 0.2f);
    float level;
    int conf;
    int ns = slicer(po, nsym, sym, &level, &conf);

    int direction = in->direction;
    int dl = uw_hard(sym, ns, 1), ul = uw_hard(sym, ns, 2);
    if (!dl && !ul) {
        float de = uw_soft(po, ns, 1), ue = uw_soft(po, ns, 2);
        float mn = de < ue ? de : ue;
        if (mn > 3.0f) {
            free(dec); free(po); free(sym);
            out->id = in->id;
            out->ok = 0;
            return 0;
        }
        direction = ue < de ? 2 : 1;
    } else {
        if (ul && !dl) direction = 2;
        else if (dl && !ul) direction = 1;
    }

    int old = 0;                                                    /* :264-273 */
    for (int i = 0; i < ns; i++) {
        int s = sym[i];
        int df = (s - old + 4) % 4;
        old = s;
        sym[i] = dq[df];
    }
    for (int i = 0; i < ns; i++) {                                  /* :329-335 */
        out->bits[2 * i] = (sym[i] >> 1) & 1;
        out->bits[2 * i + 1] = sym[i] & 1;
    }
    float sm = 0;                                                   /* :489-503 */
    for (int i = 0; i < ns; i++)
        sm += cabsf(po[i]);
    float scale = (ns > 0 && sm > 0) ? (SQRT1_2F / (sm / ns)) : 1.0f;
    for (int i = 0; i < ns; i++) {
        out->llr[2 * i] = fabsf(crealf(po[i])) * scale;
        out->llr[2 * i + 1] = fabsf(cimagf(po[i])) * scale;
    }
    out->id = in->id;
    out->timestamp = in->timestamp;
    out->direction = direction;
    out->magnitude = in->magnitude;
    out->noise = in->noise;
    out->confidence = conf;
    out->level = level;
    out->n_symbols = ns;
    out->n_payload_symbols = ns - IR_UW_LENGTH;
    out->n_bits = 2 * ns;
    out->total_phase = total_phase;
    out->ok = 1;
    if (ns > 0) {                                                   /* :521-527 */
        double duration = (double)ns / IR_SYMBOLS_PER_SECOND;
        out->center_frequency = in->center_frequency + total_phase / duration / M_PI / 2.0;
    } else {
        out->center_frequency = in->center_frequency;
    }
    free(dec); free(po); free(sym);
    return 1;
}

/* =====================================================================
 * RAW line (frame_output.c:144-199)
 * ===================================================================== */
int orc_format_raw(const orc_demod_t *f, const char *file_info, uint64_t *t0_io,
                   char *buf, size_t cap)
{
    char auto_info[64];
    if (*t0_io == 0)
        *t0_io = (f->timestamp / 1000000000ULL) * 1000000000ULL;     /* :149 */
    uint64_t t0 = *t0_io;
    if (!file_info || !file_info[0]) {
        snprintf(auto_info, sizeof(auto_info), "i-%" PRIu64 "-t1", (uint64_t)(t0 / 1000000000ULL));
        file_info = auto_info;
    }
    double ts_ms = (double)(f->timestamp - t0) / 1000000.0;
    int freq_hz = (int)(f->center_frequency + 0.5);
    int payload = f->n_payload_symbols < 0 ? 0 : f->n_payload_symbols;
    int pos = snprintf(buf, cap, "RAW: %s %012.4f %010d N:%05.2f%+06.2f I:%011" PRIu64
                       " %3d%% %.5f %3d ",
                       file_info, ts_ms, freq_hz, f->magnitude, f->noise, f->id,
                       f->confidence, f->level, payload);
    if (pos < 0 || (size_t)pos + (size_t)f->n_bits + 2 > cap)
        return -1;
    for (int i = 0; i < f->n_bits; i++)
        buf[pos++] = (char)('0' + f->bits[i]);
    buf[pos++] = '\n';
    buf[pos] = 0;
    return pos;
}

/* =====================================================================
 * Post-demod bit layer: frame_decode() (frame_decode.c)
 * ===================================================================== */
#define BL_POLY_RA 1207u   /* BCH(31,21), frame_decode.c:36 */
#define BL_POLY_HDR 29u    /* BCH(7,3),   frame_decode.c:37 */
#define BL_CHASE 5         /* frame_decode.c:48 */

static const uint8_t bl_access_dl[24] = { 0,0,1,1,0,0,0,0,0,0,1,1,0,0,0,0,1,1,1,1,0,0,1,1 };   /* :51-53 */
static const uint8_t bl_access_ul[24] = { 1,1,0,0,1,1,0,0,0,0,1,1,1,1,0,0,1,1,1,1,1,1,0,0 };   /* :54-56 */

typedef struct { int errs; uint32_t locator; } bl_syn_t;
static bl_syn_t bl_syn_ra[1024], bl_syn_hdr[16];
static int bl_ready;

static uint32_t bl_bits_to_uint(const uint8_t *b, int n)               /* :66-72 */
{
    uint32_t v = 0;
    for (int i = 0; i < n; i++) v = (v << 1) | (b[i] & 1u);
    return v;
}

static void bl_uint_to_bits(uint32_t v, uint8_t *b, int n)             /* :74-80 */
{
    for (int i = n - 1; i >= 0; i--) { b[i] = (uint8_t)(v & 1u); v >>= 1; }
}

static uint32_t bl_rem(uint32_t poly, uint32_t val)                     /* gf2_remainder, :82-91 */
{
    if (!val) return 0;
    int pb = 32 - __builtin_clz(poly);
    for (int i = 31; i >= pb - 1; i--)
        if (val & (1u << i)) val ^= poly << (i - pb + 1);
    return val;
}

static void bl_build(uint32_t poly, int nbits, int max_err, bl_syn_t *syn, int size)   /* :95-129 */
{
    for (int i = 0; i < size; i++) { syn[i].errs = -1; syn[i].locator = 0; }
    for (int b1 = 0; b1 < nbits; b1++) {
        uint32_t v = 1u << b1, r = bl_rem(poly, v);
        if (r < (uint32_t)size) { syn[r].errs = 1; syn[r].locator = v; }
    }
    if (max_err >= 2)
        for (int b1 = 0; b1 < nbits; b1++)
            for (int b2 = b1 + 1; b2 < nbits; b2++) {
                uint32_t v = (1u << b1) | (1u << b2), r = bl_rem(poly, v);
                if (r < (uint32_t)size && syn[r].errs < 0) { syn[r].errs = 2; syn[r].locator = v; }
            }
}

static void bl_init(void)                                                /* frame_decode_init, :131-135 */
{
    if (bl_ready) return;
    bl_build(BL_POLY_RA, 31, 2, bl_syn_ra, 1024);
    bl_build(BL_POLY_HDR, 7, 1, bl_syn_hdr, 16);
    bl_ready = 1;
}

/* de_interleave (:156-176): odd symbols in reverse -> out1, even symbols in reverse -> out2 */
static void bl_deint2(const uint8_t *in, uint8_t *o1, uint8_t *o2)
{
    int p = 0;
    for (int s = 31; s >= 1; s -= 2) { o1[p++] = in[2 * s]; o1[p++] = in[2 * s + 1]; }
    p = 0;
    for (int s = 30; s >= 0; s -= 2) { o2[p++] = in[2 * s]; o2[p++] = in[2 * s + 1]; }
}

static void bl_deint2f(const float *in, float *o1, float *o2)           /* :203-215 */
{
    int p = 0;
    for (int s = 31; s >= 1; s -= 2) { o1[p++] = in[2 * s]; o1[p++] = in[2 * s + 1]; }
    p = 0;
    for (int s = 30; s >= 0; s -= 2) { o2[p++] = in[2 * s]; o2[p++] = in[2 * s + 1]; }
}

/* chase_bch_decode_p (:224-295) */
static int bl_chase(const uint8_t *blk, const float *llr, uint8_t *data, uint8_t *chk)
{
    uint32_t val = bl_bits_to_uint(blk, 31);
    uint32_t syn = bl_rem(BL_POLY_RA, val);
    if (syn == 0) {
        bl_uint_to_bits(val >> 10, data, 21);
        bl_uint_to_bits(val & 0x3FF, chk, 10);
        return 0;
    }
    if (syn < 1024 && bl_syn_ra[syn].errs >= 0) {
        val ^= bl_syn_ra[syn].locator;
        bl_uint_to_bits(val >> 10, data, 21);
        bl_uint_to_bits(val & 0x3FF, chk, 10);
        return bl_syn_ra[syn].errs;
    }
    if (!llr) return -1;
    int pos[31];
    for (int i = 0; i < 31; i++) pos[i] = i;
    for (int i = 0; i < BL_CHASE; i++) {           /* partial selection sort, first minimum wins */
        int mi = i;
        for (int j = i + 1; j < 31; j++)
            if (llr[pos[j]] < llr[pos[mi]]) mi = j;
        int t = pos[i]; pos[i] = pos[mi]; pos[mi] = t;
    }
    uint32_t fm[BL_CHASE];
    for (int i = 0; i < BL_CHASE; i++) fm[i] = 1u << (30 - pos[i]);
    uint32_t base = bl_bits_to_uint(blk, 31);
    for (int mask = 1; mask < (1 << BL_CHASE); mask++) {
        uint32_t f = base;
        for (int b = 0; b < BL_CHASE; b++)
            if (mask & (1 << b)) f ^= fm[b];
        syn = bl_rem(BL_POLY_RA, f);
        if (syn == 0) {
            bl_uint_to_bits(f >> 10, data, 21);
            bl_uint_to_bits(f & 0x3FF, chk, 10);
            return 0;
        }
        if (syn < 1024 && bl_syn_ra[syn].errs >= 0) {
            f ^= bl_syn_ra[syn].locator;
            bl_uint_to_bits(f >> 10, data, 21);
            bl_uint_to_bits(f & 0x3FF, chk, 10);
            return bl_syn_ra[syn].errs;
        }
    }
    return -1;
}

static int bl_parity(const uint8_t *blk, const uint8_t *d, const uint8_t *c)   /* check_parity32, :399-407 */
{
    int ones = 0;
    for (int i = 0; i < 21; i++) ones += d[i];
    for (int i = 0; i < 10; i++) ones += c[i];
    ones += blk[31];
    return (ones % 2) == 0;
}

static int bl_uint(const uint8_t *b, int n)                              /* extract_uint, :309-315 */
{
    int v = 0;
    for (int i = 0; i < n; i++) v = (v << 1) | b[i];
    return v;
}

static int bl_s12(const uint8_t *b)                                       /* extract_signed12, :299-307 */
{
    int mag = 0;
    for (int i = 1; i < 12; i++) mag = (mag << 1) | b[i];
    return b[0] ? (mag - (1 << 11)) : mag;
}

/* remaining 64-bit blocks (:478-497, :569-588): appends 2 x 21 data bits per good block pair */
static void bl_blocks(const uint8_t *data, const float *llr, int offset, int limit, uint8_t *stream, int cap,
                      int *len)
{
    uint8_t di1[32], di2[32], d1[21], d2[21], c1[10], c2[10];
    float l1[32], l2[32];
    while (offset + 64 <= limit && *len + 42 <= cap) {
        bl_deint2(data + offset, di1, di2);
        if (llr) bl_deint2f(llr + offset, l1, l2);
        int ea = bl_chase(di1, llr ? l1 : NULL, d1, c1);
        int eb = bl_chase(di2, llr ? l2 : NULL, d2, c2);
        if (ea < 0 || eb < 0) break;
        if (!bl_parity(di1, d1, c1)) break;
        if (!bl_parity(di2, d2, c2)) break;
        memcpy(stream + *len, d1, 21); *len += 21;
        memcpy(stream + *len, d2, 21); *len += 21;
        offset += 64;
    }
}

int orc_frame_decode(const uint8_t *bits, const float *llr_all, int n_bits, orc_decoded_t *out)
{
    bl_init();
    memset(out, 0, sizeof(*out));
    if (n_bits < 24) return 0;                                           /* :424-425 */
    int is_dl = memcmp(bits, bl_access_dl, 24) == 0;
    int is_ul = memcmp(bits, bl_access_ul, 24) == 0;
    if (!is_dl && !is_ul) return 0;
    const uint8_t *data = bits + 24;
    const float *llr = llr_all ? llr_all + 24 : NULL;
    int data_len = n_bits - 24;

    if (data_len >= 6 + 64) {                                            /* IBC, :441-505 */
        uint32_t hv = bl_bits_to_uint(data, 6);
        uint32_t hs = bl_rem(BL_POLY_HDR, hv);
        int hdr_ok = 0;
        if (hs == 0) hdr_ok = 1;
        else if (hs < 16 && bl_syn_hdr[hs].errs >= 0) { hv ^= bl_syn_hdr[hs].locator; hdr_ok = 1; }
        if (hdr_ok) {
            uint8_t hd[3], di1[32], di2[32], d1[21], d2[21], c1[10], c2[10];
            float l1[32], l2[32];
            bl_uint_to_bits(hv >> 4, hd, 3);
            bl_deint2(data + 6, di1, di2);
            if (llr) bl_deint2f(llr + 6, l1, l2);
            int e1 = bl_chase(di1, llr ? l1 : NULL, d1, c1);
            int e2 = bl_chase(di2, llr ? l2 : NULL, d2, c2);
            if (e1 >= 0 && e2 >= 0 && bl_parity(di1, d1, c1) && bl_parity(di2, d2, c2)) {
                int bc_type = bl_uint(hd, 3);
                int ibc_max = data_len < 262 ? data_len : 262;
                uint8_t st[256];
                int len = 0;
                memcpy(st, d1, 21); len += 21;
                memcpy(st + len, d2, 21); len += 21;
                bl_blocks(data, llr, 6 + 64, ibc_max, st, (int)sizeof(st), &len);
                out->type = 2;
                out->bch_len = len;
                out->bc_type = bc_type;                                  /* parse_ibc, :368-393 */
                if (len >= 42) {
                    out->sat_id = bl_uint(st, 7);
                    out->beam_id = bl_uint(st + 7, 6);
                    out->timeslot = st[14];
                    out->sv_blocking = st[15];
                    if (len >= 84 && bl_uint(st + 42, 6) == 1) {
                        uint32_t t = 0;
                        for (int i = 52; i < 84; i++) t = (t << 1) | st[i];
                        out->iri_time = t;
                    }
                }
                return 1;
            }
        }
    }

    if (data_len >= 96) {                                                /* IRA, :514-595 */
        uint8_t r1[32], r2[32], r3[32], d1[21], d2[21], d3[21], c1[10], c2[10], c3[10];
        float a1[32], a2[32], a3[32];
        int p1 = 0, p2 = 0, p3 = 0;                                      /* de_interleave3, :178-199 */
        for (int s = 47; s >= 2; s -= 3) { r1[p1++] = data[2 * s]; r1[p1++] = data[2 * s + 1]; }
        for (int s = 46; s >= 1; s -= 3) { r2[p2++] = data[2 * s]; r2[p2++] = data[2 * s + 1]; }
        for (int s = 45; s >= 0; s -= 3) { r3[p3++] = data[2 * s]; r3[p3++] = data[2 * s + 1]; }
        if (llr) {
            p1 = p2 = p3 = 0;
            for (int s = 47; s >= 2; s -= 3) { a1[p1++] = llr[2 * s]; a1[p1++] = llr[2 * s + 1]; }
            for (int s = 46; s >= 1; s -= 3) { a2[p2++] = llr[2 * s]; a2[p2++] = llr[2 * s + 1]; }
            for (int s = 45; s >= 0; s -= 3) { a3[p3++] = llr[2 * s]; a3[p3++] = llr[2 * s + 1]; }
        }
        int e1 = bl_chase(r1, llr ? a1 : NULL, d1, c1);
        int e2 = bl_chase(r2, llr ? a2 : NULL, d2, c2);
        int e3 = bl_chase(r3, llr ? a3 : NULL, d3, c3);
        if (e1 >= 0 && e2 >= 0 && e3 >= 0 && bl_parity(r1, d1, c1) && bl_parity(r2, d2, c2) && bl_parity(r3, d3, c3)) {
            uint8_t st[512];
            int len = 0;
            memcpy(st, d1, 21); len += 21;
            memcpy(st + len, d2, 21); len += 21;
            memcpy(st + len, d3, 21); len += 21;
            bl_blocks(data, llr, 96, data_len, st, (int)sizeof(st), &len);
            out->type = 1;
            out->bch_len = len;
            if (len >= 63) {                                             /* parse_ira, :317-366 */
                out->sat_id = bl_uint(st, 7);
                out->beam_id = bl_uint(st + 7, 6);
                int x = bl_s12(st + 13), y = bl_s12(st + 25), z = bl_s12(st + 37);
                out->pos_xyz[0] = x; out->pos_xyz[1] = y; out->pos_xyz[2] = z;
                double xy = sqrt((double)x * x + (double)y * y);
                out->lat = atan2((double)z, xy) * 180.0 / M_PI;
                out->lon = atan2((double)y, (double)x) * 180.0 / M_PI;
                out->alt = (int)(sqrt((double)x * x + (double)y * y + (double)z * z) * 4.0) - 6378 + 23;
                int off = 63;
                while (off + 42 <= len && out->n_pages < 12) {
                    const uint8_t *pg = st + off;
                    int all1 = 1;
                    for (int i = 0; i < 42; i++) if (!pg[i]) { all1 = 0; break; }
                    if (all1) break;
                    uint32_t tmsi = 0;
                    for (int i = 0; i < 32; i++) tmsi = (tmsi << 1) | pg[i];
                    out->page_tmsi[out->n_pages] = tmsi;
                    out->page_msc[out->n_pages] = bl_uint(pg + 34, 5);
                    out->n_pages++;
                    off += 42;
                }
            }
            return 1;
        }
    }
    return 0;
}

/* =====================================================================
 * Post-demod bit layer: ida_decode() (ida_decode.c)
 * ===================================================================== */
#define DA_POLY 3545u      /* BCH(31,20), ida_decode.c:33 */
static bl_syn_t da_syn[2048], da_syn1[16], da_syn2[256], da_syn3[32];
static int da_ready;

static const int da_perm[46] = {                                        /* ida_decode.c:54-60 */
    40, 39, 36, 35, 32, 31, 28, 27, 24, 23, 20, 19, 16, 15, 12, 11, 8, 7, 4, 3,
    41, 38, 37, 34, 33, 30, 29, 26, 25, 22, 21, 18, 17, 14, 13, 10, 9, 6, 5, 2,
    1, 46, 45, 44, 43, 42
};

static void da_init(void)                                                /* ida_decode_init, :96-102 */
{
    if (da_ready) return;
    bl_build(DA_POLY, 31, 2, da_syn, 2048);
    bl_build(29u, 7, 1, da_syn1, 16);
    bl_build(465u, 14, 1, da_syn2, 256);
    bl_build(41u, 26, 2, da_syn3, 32);
    da_ready = 1;
}

/* chase_bch_da (:107-172) */
static int da_chase(const uint8_t *blk, const float *llr, uint8_t *data, int *fixed)
{
    uint32_t val = bl_bits_to_uint(blk, 31);
    uint32_t syn = bl_rem(DA_POLY, val);
    if (syn == 0) { bl_uint_to_bits(val >> 11, data, 20); *fixed = 0; return 0; }
    if (syn < 2048 && da_syn[syn].errs >= 0) {
        val ^= da_syn[syn].locator;
        bl_uint_to_bits(val >> 11, data, 20);
        *fixed = 1;
        return da_syn[syn].errs;
    }
    if (!llr) return -1;
    int pos[31];
    for (int i = 0; i < 31; i++) pos[i] = i;
    for (int i = 0; i < BL_CHASE; i++) {
        int mi = i;
        for (int j = i + 1; j < 31; j++)
            if (llr[pos[j]] < llr[pos[mi]]) mi = j;
        int t = pos[i]; pos[i] = pos[mi]; pos[mi] = t;
    }
    uint32_t fm[BL_CHASE];
    for (int i = 0; i < BL_CHASE; i++) fm[i] = 1u << (30 - pos[i]);
    uint32_t base = bl_bits_to_uint(blk, 31);
    for (int mask = 1; mask < (1 << BL_CHASE); mask++) {
        uint32_t f = base;
        for (int b = 0; b < BL_CHASE; b++)
            if (mask & (1 << b)) f ^= fm[b];
        syn = bl_rem(DA_POLY, f);
        if (syn == 0) { bl_uint_to_bits(f >> 11, data, 20); *fixed = 1; return 0; }
        if (syn < 2048 && da_syn[syn].errs >= 0) {
            f ^= da_syn[syn].locator;
            bl_uint_to_bits(f >> 11, data, 20);
            *fixed = 1;
            return da_syn[syn].errs;
        }
    }
    return -1;
}

static void da_deint(const uint8_t *in, int n_sym, uint8_t *o1, uint8_t *o2)   /* de_interleave_n, :259-272 */
{
    int p = 0;
    for (int s = n_sym - 1; s >= 1; s -= 2) { o1[p++] = in[2 * s]; o1[p++] = in[2 * s + 1]; }
    p = 0;
    for (int s = n_sym - 2; s >= 0; s -= 2) { o2[p++] = in[2 * s]; o2[p++] = in[2 * s + 1]; }
}

static void da_deintf(const float *in, int n_sym, float *o1, float *o2)        /* :176-189 */
{
    int p = 0;
    for (int s = n_sym - 1; s >= 1; s -= 2) { o1[p++] = in[2 * s]; o1[p++] = in[2 * s + 1]; }
    p = 0;
    for (int s = n_sym - 2; s >= 0; s -= 2) { o2[p++] = in[2 * s]; o2[p++] = in[2 * s + 1]; }
}

/* decode_lcw (:193-252) */
static int da_lcw(const uint8_t *data, int data_len, orc_ida_t *o)
{
    if (data_len < 46) return 0;
    uint8_t sw[46], lb[46];
    for (int i = 0; i < 46; i += 2) { sw[i] = data[i + 1]; sw[i + 1] = data[i]; }
    for (int i = 0; i < 46; i++) lb[i] = sw[da_perm[i] - 1];
    uint32_t v1 = bl_bits_to_uint(lb, 7), s1 = bl_rem(29u, v1);
    if (s1 != 0) {
        if (s1 >= 16 || da_syn1[s1].errs < 0) return 0;
        v1 ^= da_syn1[s1].locator;
    }
    int ft = (int)(v1 >> 4) & 7;
    uint32_t v2 = bl_bits_to_uint(lb + 7, 13) << 1, s2 = bl_rem(465u, v2);
    if (s2 != 0) {
        if (s2 >= 256 || da_syn2[s2].errs < 0) return 0;
        v2 ^= da_syn2[s2].locator;
    }
    uint32_t v3 = bl_bits_to_uint(lb + 20, 26), s3 = bl_rem(41u, v3);
    if (s3 != 0) {
        if (s3 >= 32 || da_syn3[s3].errs < 0) return 0;
        v3 ^= da_syn3[s3].locator;
    }
    int d2 = (int)(v2 >> 8) & 0x3F;
    o->ft = ft;
    o->lcw_ft = (d2 >> 4) & 3;
    o->lcw_code = d2 & 0xF;
    o->lcw3_val = (uint32_t)((int)(v3 >> 5));
    o->ec_lcw = (s1 != 0) + (s2 != 0) + (s3 != 0);
    return 1;
}

/* descramble_payload (:276-377) */
static int da_descramble(const uint8_t *data, const float *llr, int data_len, uint8_t *st, int max_bch, int *fixederrs)
{
    int len = 0;
    *fixederrs = 0;
    int n_full = data_len / 124, remain = data_len % 124;
    for (int blk = 0; blk < n_full; blk++) {
        const uint8_t *b = data + blk * 124;
        const float *bl = llr ? llr + blk * 124 : NULL;
        uint8_t comb[124];
        float lcomb[124];
        da_deint(b, 62, comb, comb + 62);
        if (bl) da_deintf(bl, 62, lcomb, lcomb + 62);
        static const int order[4] = { 3, 1, 2, 0 };
        for (int c = 0; c < 4; c++) {
            if (len + 20 > max_bch) break;
            int off = order[c] * 31, fixed = 0;
            uint8_t od[20];
            int e = da_chase(comb + off, bl ? lcomb + off : NULL, od, &fixed);
            if (e < 0) return len;
            *fixederrs += fixed;
            memcpy(st + len, od, 20);
            len += 20;
        }
    }
    if (remain >= 4 && len + 2 * (remain / 2 - 1) <= max_bch) {
        int ns = remain / 2;
        uint8_t h1[64], h2[64];
        float l1[64], l2[64];
        const float *ll = llr ? llr + n_full * 124 : NULL;
        da_deint(data + n_full * 124, ns, h1, h2);
        if (ll) da_deintf(ll, ns, l1, l2);
        if (ns > 1 && len + 20 <= max_bch) {
            uint8_t comb[128];
            float lcomb[128];
            int cl = 0;
            for (int i = 1; i < ns && cl < 128; i++) { comb[cl] = h2[i]; if (ll) lcomb[cl] = l2[i]; cl++; }
            for (int i = 1; i < ns && cl < 128; i++) { comb[cl] = h1[i]; if (ll) lcomb[cl] = l1[i]; cl++; }
            int pos = 0;
            while (pos + 31 <= cl && len + 20 <= max_bch) {
                uint8_t od[20];
                int fixed = 0;
                int e = da_chase(comb + pos, ll ? lcomb + pos : NULL, od, &fixed);
                if (e < 0) break;
                *fixederrs += fixed;
                memcpy(st + len, od, 20);
                len += 20;
                pos += 31;
            }
        }
    }
    return len;
}

static uint16_t da_crc(const uint8_t *d, int n)                          /* crc_ccitt, :381-394 */
{
    uint16_t crc = 0xFFFF;
    for (int i = 0; i < n; i++) {
        crc ^= (uint16_t)((uint16_t)d[i] << 8);
        for (int j = 0; j < 8; j++)
            crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x1021) : (uint16_t)(crc << 1);
    }
    return crc;
}

static int da_bits(const char *b, int from, int to)                      /* MSB-first field of the lcw3 bit string */
{
    int v = 0;
    for (int i = from; i < to; i++) v = (v << 1) | (b[i] - '0');
    return v;
}

/* format_lcw_header (:405-539) */
void orc_format_lcw_header(int ft, int lcw_ft, int lcw_code, uint32_t lcw3_val, char *out, int outsz)
{
    char b[32], code[128], rem[64], raw[128];
    const char *ty;
    for (int i = 0; i < 21; i++) b[i] = (char)('0' + ((lcw3_val >> (20 - i)) & 1));
    b[21] = 0;
    switch (lcw_ft) {
    case 0:
        ty = "maint";
        switch (lcw_code) {
        case 0:
            snprintf(code, sizeof(code), "sync[status:%d,dtoa:%d,dfoa:%d]", b[1] - '0', da_bits(b, 3, 13), da_bits(b, 13, 21));
            snprintf(rem, sizeof(rem), "%c|%c", b[0], b[2]);
            break;
        case 1:
            snprintf(code, sizeof(code), "switch[dtoa:%d,dfoa:%d]", da_bits(b, 3, 13), da_bits(b, 13, 21));
            snprintf(rem, sizeof(rem), "%.3s", b);
            break;
        case 3:
            snprintf(code, sizeof(code), "maint[2][lqi:%d,power:%d,f_dtoa:%d,f_dfoa:%d]",
                     (b[1] - '0') * 2 + (b[2] - '0'), da_bits(b, 3, 6), da_bits(b, 6, 13), da_bits(b, 13, 20));
            snprintf(rem, sizeof(rem), "%c|%c", b[0], b[20]);
            break;
        case 6:
            snprintf(code, sizeof(code), "geoloc");
            snprintf(rem, sizeof(rem), "%s", b);
            break;
        case 12:
            snprintf(code, sizeof(code), "maint[1][lqi:%d,power:%d]", (b[19] - '0') * 2 + (b[20] - '0'), da_bits(b, 16, 19));
            b[16] = 0;
            snprintf(rem, sizeof(rem), "%s", b);
            break;
        case 15:
            snprintf(code, sizeof(code), "<silent>");
            snprintf(rem, sizeof(rem), "%s", b);
            break;
        default:
            snprintf(code, sizeof(code), "rsrvd(%d)", lcw_code);
            snprintf(rem, sizeof(rem), "%s", b);
            break;
        }
        break;
    case 1:
        ty = "acchl";
        if (lcw_code == 1) {
            char segm[16];
            memcpy(segm, b + 8, 8);
            segm[8] = 0;
            snprintf(code, sizeof(code), "acchl[msg_type:%01x,bloc_num:%01x,sapi_code:%01x,segm_list:%s]",
                     da_bits(b, 1, 4), b[4] - '0', da_bits(b, 5, 8), segm);
            snprintf(rem, sizeof(rem), "%c,%02x", b[0], da_bits(b, 16, 21));
        } else {
            snprintf(code, sizeof(code), "rsrvd(%d)", lcw_code);
            snprintf(rem, sizeof(rem), "%s", b);
        }
        break;
    case 2:
        ty = "hndof";
        switch (lcw_code) {
        case 3:
            snprintf(code, sizeof(code),
                     "handoff_resp[cand:%c,denied:%d,ref:%d,slot:%d,sband_up:%d,sband_dn:%d,access:%d]",
                     (b[2] - '0') == 0 ? 'P' : 'S', b[3] - '0', b[4] - '0', 1 + (b[6] - '0') * 2 + (b[7] - '0'),
                     da_bits(b, 8, 13), da_bits(b, 13, 18), da_bits(b, 18, 21) + 1);
            snprintf(rem, sizeof(rem), "%.2s,%c", b, b[5]);
            break;
        case 12: {
            char first[12], second[11];
            memcpy(first, b, 11); first[11] = 0;
            memcpy(second, b + 11, 10); second[10] = 0;
            snprintf(code, sizeof(code), "handoff_cand");
            snprintf(rem, sizeof(rem), "%s,%s", first, second);
            break;
        }
        case 15:
            snprintf(code, sizeof(code), "<silent>");
            snprintf(rem, sizeof(rem), "%s", b);
            break;
        default:
            snprintf(code, sizeof(code), "rsrvd(%d)", lcw_code);
            snprintf(rem, sizeof(rem), "%s", b);
            break;
        }
        break;
    default:
        ty = "rsrvd";
        snprintf(code, sizeof(code), "<%d>", lcw_code);
        snprintf(rem, sizeof(rem), "%s", b);
        break;
    }
    snprintf(raw, sizeof(raw), "LCW(%d,T:%s,C:%s,%s)", ft, ty, code, rem);
    snprintf(out, (size_t)outsz, "%-110s ", raw);
}

int orc_ida_decode(const uint8_t *bits, const float *llr_all, int n_bits, int direction, orc_ida_t *o)
{
    bl_init();
    da_init();
    memset(o, 0, sizeof(*o));
    if (n_bits < 24 + 46 + 124) return 0;                                /* :547-548 */
    if (direction != 1 && direction != 2) return 0;                      /* :551-552 */
    const uint8_t *data = bits + 24;
    const float *llr = llr_all ? llr_all + 24 : NULL;
    int data_len = n_bits - 24;
    orc_ida_t lcw;
    memset(&lcw, 0, sizeof(lcw));
    if (!da_lcw(data, data_len, &lcw)) return 0;
    if (lcw.ft != 2) return 0;
    int payload_len = data_len - 46;
    if (payload_len < 124) return 0;
    uint8_t st[512];
    int fixederrs = 0;
    int len = da_descramble(data + 46, llr ? llr + 46 : NULL, payload_len, st, (int)sizeof(st), &fixederrs);
    if (len < 196) return 0;                                             /* :577-578 */
    int cont = st[3];
    int da_ctr = (st[5] << 2) | (st[6] << 1) | st[7];
    int da_len = (st[11] << 4) | (st[12] << 3) | (st[13] << 2) | (st[14] << 1) | st[15];
    int zero1 = (st[17] << 2) | (st[18] << 1) | st[19];
    if (zero1 != 0) return 0;
    if (da_len > 20) return 0;
    uint8_t payload[20];
    for (int i = 0; i < 20; i++) {
        uint8_t by = 0;
        for (int b = 0; b < 8; b++) by = (uint8_t)((by << 1) | st[20 + i * 8 + b]);
        payload[i] = by;
    }
    int crc_ok = 0;
    uint16_t stored = 0, computed = 0;
    if (da_len > 0 && len >= 196) {                                      /* :606-637 */
        for (int i = 0; i < 16; i++) stored = (uint16_t)((stored << 1) | st[9 * 20 + i]);
        int crc_bits = 20 + 12 + (len - 20 - 4);
        int crc_bytes = (crc_bits + 7) / 8;
        uint8_t buf[64];
        if (crc_bytes <= (int)sizeof(buf)) {
            memset(buf, 0, sizeof(buf));
            int bp = 0;
            for (int i = 0; i < 20; i++) { buf[bp / 8] |= (uint8_t)(st[i] << (7 - (bp % 8))); bp++; }
            bp += 12;
            for (int i = 20; i < len - 4; i++) { buf[bp / 8] |= (uint8_t)(st[i] << (7 - (bp % 8))); bp++; }
            computed = da_crc(buf, (bp + 7) / 8);
            crc_ok = computed == 0;
        }
    }
    o->ok = 1;
    o->ft = lcw.ft; o->lcw_ft = lcw.lcw_ft; o->lcw_code = lcw.lcw_code; o->ec_lcw = lcw.ec_lcw; o->lcw3_val = lcw.lcw3_val;
    o->da_ctr = da_ctr; o->da_len = da_len; o->cont = cont; o->crc_ok = crc_ok;
    o->stored_crc = stored; o->computed_crc = computed;
    o->fixederrs = fixederrs;
    o->payload_len = da_len > 0 ? da_len : 20;
    memcpy(o->payload, payload, (size_t)o->payload_len);
    o->bch_len = len;
    memcpy(o->bch_stream, st, (size_t)(len < 256 ? len : 256));
    orc_format_lcw_header(lcw.ft, lcw.lcw_ft, lcw.lcw_code, lcw.lcw3_val, o->lcw_header, (int)sizeof(o->lcw_header));
    return 1;
}

/* =====================================================================
 * Whole stream, reference file-mode plumbing (main.c:223-284 spewer,
 * burst_detect.c:941-956, burst_downmix.c:801-824, main.c:307-373)
 * ===================================================================== */
typedef struct {
    const orc_stream_cfg_t *cfg;
    orc_stream_out_t *out;
    orc_downmix_t *dm;
    int fft_size;
    int overflow;
} stream_ctx_t;

static void stream_cb(const orc_burst_rec_t *rec, const float *samples, void *user)
{
    stream_ctx_t *c = user;
    orc_stream_out_t *o = c->out;
    if (o->n_bursts >= o->cap_bursts || o->n_frames >= o->cap_frames) {
        c->overflow = 1;
        return;
    }
    o->bursts[o->n_bursts++] = *rec;
    orc_frame_t *fr = &o->frames[o->n_frames++];
    int ok = orc_downmix_process(c->dm, rec, samples, c->cfg->center_frequency,
                                 c->cfg->sample_rate, c->fft_size,
                                 c->cfg->start_time_ns, fr);
    if (!ok)
        return;
    if (o->n_demods >= o->cap_demods) {
        c->overflow = 1;
        return;
    }
    orc_demod_t *dmo = &o->demods[o->n_demods];
    orc_frame_t tmp_in;
    memcpy(&tmp_in, fr, offsetof(orc_frame_t, samples) + sizeof(float) * 2 * (size_t)fr->num_samples);
    if (orc_qpsk_demod(&tmp_in, c->cfg->use_gardner, dmo))
        o->n_demods++;
}

int orc_run_stream(const void *iq, size_t n_samples, const orc_stream_cfg_t *cfg,
                   orc_stream_out_t *out)
{
    stream_ctx_t c;
    memset(&c, 0, sizeof(c));
    c.cfg = cfg;
    c.out = out;
    out->n_bursts = out->n_frames = out->n_demods = 0;
    orc_detector_t *det = orc_detector_create(cfg->center_frequency, cfg->sample_rate,
                                              cfg->threshold_db, 0);
    c.dm = orc_downmix_create();
    c.fft_size = det->n;
    size_t block = cfg->block > 0 ? (size_t)cfg->block : 32768;
    int8_t *tmp8 = NULL;
    if (cfg->format == 1)
        tmp8 = malloc(2 * block);
    for (size_t off = 0; off < n_samples; off += block) {
        size_t r = n_samples - off < block ? n_samples - off : block;
        if (cfg->format == 2) {
            orc_detector_feed_cf32(det, (const float *)iq + 2 * off, r, stream_cb, &c);
        } else if (cfg->format == 0) {
            orc_detector_feed_i8(det, (const int8_t *)iq + 2 * off, r, stream_cb, &c);
        } else {
            const int16_t *p = (const int16_t *)iq + 2 * off;
            for (size_t i = 0; i < 2 * r; i++)
                tmp8[i] = (int8_t)(p[i] >> 8);                       /* main.c:245-246 */
            orc_detector_feed_i8(det, tmp8, r, stream_cb, &c);
        }
    }
    free(tmp8);
    out->n_tagged = det->tagged;
    out->n_samples = det->sample_count;
    orc_detector_destroy(det);
    orc_downmix_destroy(c.dm);
    return c.overflow ? -1 : 0;
}
