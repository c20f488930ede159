// libm_port.hpp -- glibc's sincosf / cexpf(i*x) restated for the device, bit for bit.
//
// The fine-CFO rotator's increment is cexpf(-2*pi*offset * I) (burst_downmix.c:716-717).  glibc 2.35 evaluates it as
// __sincosf (sysdeps/ieee754/flt-32/s_sincosf.c, s_sincosf.h, s_sincosf_data.c: the ARM optimized-routines design):
// argument promoted to double, quadrant reduction by pi/2, two degree-7/8 polynomials in double, results rounded to
// float; cexpf then multiplies both by expf(+-0) = 1 (exact).  Every operation is an IEEE binary64 add / multiply /
// fused multiply-add, all of which gfx950 executes correctly rounded, so the same sequence gives the same bits.
//
// Which sequence: x86-64 glibc selects between two builds of the SAME source at load time (ifunc): one compiled with
// FMA contraction (-mfma: every a + b*c below is one rounding) and one without.  FUSED picks the build; the pipeline
// takes the one that reproduces the host's own cexpf on a probe set at create time (and keeps the host step if neither
// does), so "the libm the reference would run with on this host" stays the contract.  The constants are the table
// __sincosf_table of the image's libm.so.6, read at 0xb30c0 (tools/check_sincosf.cpp prints and compares them).
//
// Domain: |y| < 120 (the large-argument reduction is not restated; the CFO step stays within |y| <= pi/2 + eps).
// tools/check_sincosf.cpp compares this file with the host's sincosf over EVERY float in [-2, 2] (host-compiled, both
// builds), tests/test_gpu_libm.py does the same for the device through irdm_sincosf_probe.
#pragma once
#include <stdint.h>
#include <string.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define IRDM_LIBM_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define IRDM_LIBM_HD static inline
#endif

namespace irdm {

template <bool FUSED>
IRDM_LIBM_HD double libm_madd(double a, double b, double c)      // a * b + c
{
#if defined(__HIP_DEVICE_COMPILE__)
    return FUSED ? __fma_rn(a, b, c) : __dadd_rn(__dmul_rn(a, b), c);
#else
    if (FUSED) return __builtin_fma(a, b, c);
    volatile double p = a * b;       // (the product is rounded on its own whatever the host compiler's contraction rules)
    return p + c;
#endif
}

// sin / cos polynomials of one quadrant (s_sincosf.h: sincosf_poly), p = 0: quadrants 0 / 1, p = 1: quadrants 2 / 3
template <bool FUSED>
IRDM_LIBM_HD void libm_sincosf_poly(double x, double x2, int p, double *s_out, double *c_out)
{
    const double sg = p ? -1.0 : 1.0;
    const double c0 = sg * 0x1p0, c1 = sg * -0x1.ffffffd0c621cp-2, c2 = sg * 0x1.55553e1068f19p-5,
                 c3 = sg * -0x1.6c087e89a359dp-10, c4 = sg * 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double x4 = x2 * x2;
    const double x3 = x2 * x;
    const double c2v = libm_madd<FUSED>(x2, c4, c3);
    const double s1v = libm_madd<FUSED>(x2, s3, s2);
    const double c1v = libm_madd<FUSED>(x2, c1, c0);
    const double x5 = x3 * x2;
    const double x6 = x4 * x2;
    const double s = libm_madd<FUSED>(x3, s1, x);
    const double c = libm_madd<FUSED>(x4, c2, c1v);
    *s_out = libm_madd<FUSED>(x5, s1v, s);
    *c_out = libm_madd<FUSED>(x6, c2v, c);
}

// returns 0, or 1 if |y| is outside the restated domain (outputs untouched)
template <bool FUSED>
IRDM_LIBM_HD int libm_sincosf(float y, float *sinp, float *cosp)
{
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(y);
#else
    memcpy(&u, &y, 4);
#endif
    const uint32_t top = (u >> 20) & 0x7ff;          // abstop12
    double x = (double)y;
    if (top < 0x3f4) {                               // |y| < pi/4
        if (top < 0x398) {                           // |y| < 2^-12
            *sinp = y;
            *cosp = 1.0f;
            return 0;
        }
        double s, c;
        libm_sincosf_poly<FUSED>(x, x * x, 0, &s, &c);
        *sinp = (float)s;
        *cosp = (float)c;
        return 0;
    }
    if (top >= 0x42f) return 1;                      // |y| >= 120: reduce_large / inf / nan
    // reduce_fast without the rounding intrinsics: 2/pi prescaled by 2^24, the quadrant ends up in bits 24..31
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = libm_madd<FUSED>(-(double)n, 0x1.921FB54442D18p0, x);
    const double sign = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    double s, c;
    libm_sincosf_poly<FUSED>(x * sign, x * x, (n & 2) ? 1 : 0, &s, &c);
    if (n & 1) {
        *sinp = (float)c;
        *cosp = (float)s;
    } else {
        *sinp = (float)s;
        *cosp = (float)c;
    }
    return 0;
}

// cexpf(y * I) as glibc's __cexpf evaluates it for a zero real part: (cos y, sin y) times expf(+-0) = 1; arguments
// below FLT_MIN skip sincosf
template <bool FUSED>
IRDM_LIBM_HD int libm_cexpf_i(float y, float *re, float *im)
{
    const float ay = y < 0 ? -y : y;
    if (!(ay > 0x1p-126f)) {
        *re = 1.0f;
        *im = y;
        return 0;
    }
    float s = 0.0f, c = 0.0f;
    const int rc = libm_sincosf<FUSED>(y, &s, &c);
    *re = c;
    *im = s;
    return rc;
}

}  // namespace irdm
