// host_design.cpp -- see host_design.hpp.  Compiled with -ffp-contract=off.
#include "host_design.hpp"

#include <math.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace irdm {

std::vector<float> design_blackman(int n)
{
    std::vector<float> w(n);
    for (int i = 0; i < n; i++)
        w[i] = 0.42f - 0.5f * cosf(2.0f * (float)M_PI * i / (n - 1))
                     + 0.08f * cosf(4.0f * (float)M_PI * i / (n - 1));
    return w;
}

std::vector<float> design_lpf(float gain, float sample_rate, float cutoff, float transition)
{
    int ntaps = (int)(4.0f / (transition / sample_rate));
    ntaps |= 1;
    std::vector<float> taps(ntaps);
    const int center = ntaps / 2;
    const float omega_c = 2.0f * (float)M_PI * cutoff / sample_rate;
    float total = 0;
    for (int i = 0; i < ntaps; i++) {
        const float n = i - center;
        float h;
        if (fabsf(n) < 1e-10f) h = omega_c / (float)M_PI;
        else                   h = sinf(omega_c * n) / ((float)M_PI * n);
        const float w = 0.35875f
                      - 0.48829f * cosf(2.0f * (float)M_PI * i / (ntaps - 1))
                      + 0.14128f * cosf(4.0f * (float)M_PI * i / (ntaps - 1))
                      - 0.01168f * cosf(6.0f * (float)M_PI * i / (ntaps - 1));
        taps[i] = h * w;
        total += taps[i];
    }
    if (fabsf(total) > 0) {
        const float scale = gain / total;
        for (int i = 0; i < ntaps; i++) taps[i] *= scale;
    }
    return taps;
}

std::vector<float> design_rrc(float gain, float sample_rate, float symbol_rate, float alpha, int ntaps)
{
    ntaps |= 1;
    std::vector<float> taps(ntaps);
    const float sps = sample_rate / symbol_rate;
    const int center = ntaps / 2;
    float energy = 0;
    for (int i = 0; i < ntaps; i++) {
        const float t = (i - center) / sps;
        if (fabsf(t) < 1e-10f) {
            taps[i] = (1.0f - alpha + 4.0f * alpha / (float)M_PI);
        } else if (fabsf(fabsf(t) - 1.0f / (4.0f * alpha)) < 1e-6f) {
            taps[i] = alpha / sqrtf(2.0f) *
                ((1.0f + 2.0f / (float)M_PI) * sinf((float)M_PI / (4.0f * alpha)) +
                 (1.0f - 2.0f / (float)M_PI) * cosf((float)M_PI / (4.0f * alpha)));
        } else {
            const float num = sinf((float)M_PI * t * (1.0f - alpha)) +
                              4.0f * alpha * t * cosf((float)M_PI * t * (1.0f + alpha));
            const float den = (float)M_PI * t * (1.0f - (4.0f * alpha * t) * (4.0f * alpha * t));
            taps[i] = num / den;
        }
        energy += taps[i] * taps[i];
    }
    const float scale = gain / sqrtf(energy);
    for (int i = 0; i < ntaps; i++) taps[i] *= scale;
    return taps;
}

static float sinc_pi(float x)
{
    if (fabsf(x) < 1e-10f) return 1.0f;
    return sinf((float)M_PI * x) / ((float)M_PI * x);
}

std::vector<float> design_rc(float sample_rate, float symbol_rate, float alpha, int ntaps)
{
    ntaps |= 1;
    std::vector<float> taps(ntaps);
    const float sps = sample_rate / symbol_rate;
    const int center = ntaps / 2;
    for (int i = 0; i < ntaps; i++) {
        const float t = (i - center) / sps;
        if (fabsf(t) < 1e-10f) {
            taps[i] = 1.0f;
        } else if (alpha > 0 && fabsf(fabsf(t) - 1.0f / (2.0f * alpha)) < 1e-6f) {
            taps[i] = (float)M_PI / (4.0f) * sinc_pi(1.0f / (2.0f * alpha));
        } else {
            const float cos_term = cosf((float)M_PI * alpha * t);
            const float den = 1.0f - (2.0f * alpha * t) * (2.0f * alpha * t);
            taps[i] = sinc_pi(t) * cos_term / den;
        }
    }
    return taps;
}

std::vector<float> design_box(int length)
{
    return std::vector<float>(length, 1.0f / length);
}

std::vector<cfloat> design_twiddles(int n)
{
    std::vector<cfloat> tw(n / 2 > 0 ? n / 2 : 1);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < n / 2; k++) {
        const long double a = two_pi * (long double)k / (long double)n;
        tw[k] = cfloat((float)cosl(a), (float)(-sinl(a)));
    }
    tw[0] = cfloat(1.0f, 0.0f);
    if (n >= 4) tw[n / 4] = cfloat(0.0f, -1.0f);
    return tw;
}

static inline cfloat cmul4(cfloat x, cfloat y)
{
    const float ac = x.real() * y.real(), bd = x.imag() * y.imag();
    const float ad = x.real() * y.imag(), bc = x.imag() * y.real();
    return cfloat(ac - bd, ad + bc);
}

void host_fft(std::vector<cfloat> &d, const std::vector<cfloat> &tw, int dir)
{
    const int n = (int)d.size();
    int bits = 0;
    while ((1 << bits) < n) bits++;
    for (int i = 0; i < n; i++) {
        int r = 0;
        for (int b = 0; b < bits; b++)
            if (i & (1 << b)) r |= 1 << (bits - 1 - b);
        if (r > i) std::swap(d[i], d[r]);
    }
    for (int st = 1; st <= bits; st++) {
        const int half = 1 << (st - 1);
        for (int b = 0; b < n / 2; b++) {
            const int j = b & (half - 1);
            const int i0 = ((b >> (st - 1)) << st) + j, i1 = i0 + half;
            const int tix = j << (bits - st);
            const cfloat a = d[i0], v = d[i1];
            cfloat t;
            if (tix == 0) t = v;
            else if (4 * tix == n) t = dir < 0 ? cfloat(v.imag(), -v.real()) : cfloat(-v.imag(), v.real());
            else {
                cfloat w = tw[tix];
                if (dir > 0) w = cfloat(w.real(), -w.imag());
                t = cmul4(w, v);
            }
            d[i0] = cfloat(a.real() + t.real(), a.imag() + t.imag());
            d[i1] = cfloat(a.real() - t.real(), a.imag() - t.imag());
        }
    }
}

std::vector<cfloat> design_sync_template(const std::vector<float> &rc, int corr_n, float sps,
                                         bool uplink, int *sync_len)
{
    static const int uw_dl[12] = { 0, 2, 2, 2, 2, 0, 0, 0, 2, 0, 0, 2 };   // iridium.h:30
    static const int uw_ul[12] = { 2, 2, 0, 0, 0, 2, 0, 0, 2, 0, 2, 2 };   // iridium.h:31
    const int preamble = 16, uw_len = 12;
    const cfloat s0(1.0f, 1.0f), s1(-1.0f, -1.0f);
    const int total = preamble + uw_len;
    std::vector<cfloat> sym(total);
    for (int i = 0; i < preamble; i++) sym[i] = uplink ? ((i % 2 == 0) ? s1 : s0) : s0;
    const int *uw = uplink ? uw_ul : uw_dl;
    for (int i = 0; i < uw_len; i++) sym[preamble + i] = uw[i] == 0 ? s0 : s1;

    const int isps = (int)roundf(sps);
    const int plen = total * isps - (isps - 1);
    const int ntaps = (int)rc.size();
    const int half = (ntaps - 1) / 2;
    std::vector<cfloat> buf(plen + ntaps - 1, cfloat(0, 0));
    for (int i = 0; i < total; i++) buf[half + i * isps] = sym[i];
    std::vector<cfloat> shaped(plen);
    for (int i = 0; i < plen; i++) {                    // generic_fir_ccf order
        float ar = 0.0f, ai = 0.0f;
        for (int k = 0; k < ntaps; k++) {
            ar += rc[k] * buf[i + k].real();
            ai += rc[k] * buf[i + k].imag();
        }
        shaped[i] = cfloat(ar, ai);
    }
    std::vector<cfloat> tmpl(corr_n, cfloat(0, 0));
    for (int i = 0; i < plen && i < corr_n; i++) tmpl[i] = std::conj(shaped[plen - 1 - i]);
    host_fft(tmpl, design_twiddles(corr_n), -1);
    *sync_len = plen;
    return tmpl;
}

static cfloat cexp_i(float phase_inc)
{
    // cexpf(phase_inc * I) exactly as the reference spells it (burst_downmix.c:669, :717)
    // (phase_inc * I has real part +-0; exp(+-0) = 1 either way)
    float _Complex z = __builtin_cexpf(__builtin_complex(0.0f, phase_inc));
    return cfloat(__real__ z, __imag__ z);
}

std::vector<cfloat> design_rotator_incr(int n)
{
    std::vector<cfloat> inc(n);
    for (int bin = 0; bin < n; bin++) {
        const float rel = (bin - n / 2) / (float)n;
        const float phase_inc = -2.0f * (float)M_PI * rel;
        inc[bin] = cexp_i(phase_inc);
    }
    return inc;
}

cfloat fine_rotator_incr(float center_offset)
{
    const float phase_inc = -2.0f * (float)M_PI * center_offset;
    return cexp_i(phase_inc);
}

}  // namespace irdm
