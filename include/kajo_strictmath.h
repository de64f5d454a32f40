/*
 * kajo_strictmath.h -- elementary functions of the STRICT numerics mode (interface contract).
 *
 * The reference calls libm's sinf/cosf/asinf/acosf/powf (renderer/cpu/Random.cpp:77-102,
 * renderer/cpu/Light.cpp:26-49, renderer/cpu/BSDF.cpp:61-74, renderer/Image.cpp:14-17).
 * glibc's results cannot be reproduced bit for bit by a GPU's math library, and one
 * flipped branch changes a whole path, so "does the GPU integrator
 * take exactly the decisions the CPU integrator takes" is only testable when both sides
 * evaluate these five functions with the same arithmetic. This header is that arithmetic:
 * range reduction + truncated series in IEEE binary64 using only + - * / fma sqrt floor and
 * integer bit moves, rounded once to binary32 at the end. Compiled with FP contraction off
 * it yields identical bits from g++ on x86-64 and from hipcc on gfx950; its error before
 * the final rounding is < 1e-11 relative, i.e. the float result equals the correctly rounded
 * one except within ~1e-4 of a rounding boundary (tests/test_strictmath.py pins it against
 * libm to <= 1 ulp).
 *
 * sqrtf and the four basic operations need no counterpart: they are correctly rounded on
 * both targets (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt).
 *
 * Plain C subset; no state; every function is pure.
 */
#ifndef KAJO_STRICTMATH_H
#define KAJO_STRICTMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define KSM_FN __host__ __device__ static inline
#else
#define KSM_FN static inline
#endif

/* One rounding per Horner step: fma is exactly specified, so x86-64 (vfmadd / glibc fma) and gfx950 (v_fma_f64)
   still agree bit for bit, with half the operations of a separate multiply and add. */
#if defined(__HIP_DEVICE_COMPILE__)
/* v_fma_f64 with the coefficient as a scalar operand: left to itself the compiler picks the two-address v_fmac_f64 and
   first moves every 64-bit coefficient into a VGPR pair (more moves than the fma saves, and spills) */
KSM_FN double ksm_fma_coeff(double a, double b, double c)
{
    double r;
    __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
#define KSM_FMA(a, b, c) ksm_fma_coeff((a), (b), (c))
#else
#define KSM_FMA(a, b, c) __builtin_fma((a), (b), (c))
#endif

KSM_FN uint64_t ksm_bits(double d)
{
    uint64_t u;
    __builtin_memcpy(&u, &d, 8);
    return u;
}

KSM_FN double ksm_from_bits(uint64_t u)
{
    double d;
    __builtin_memcpy(&d, &u, 8);
    return d;
}

/* 2^e for e in [-1022, 1023] */
KSM_FN double ksm_pow2(int e)
{
    return ksm_from_bits((uint64_t)(e + 1023) << 52);
}

/* sin r, |r| <= pi/4 (Taylor to r^15, truncation < 6e-17 relative) */
KSM_FN double ksm_sin_kernel(double r)
{
    double z = r * r;
    double p = -0x1.ae7f3e733b81fp-41;
    p = KSM_FMA(p, z, 0x1.6124613a86d09p-33);
    p = KSM_FMA(p, z, -0x1.ae64567f544e4p-26);
    p = KSM_FMA(p, z, 0x1.71de3a556c734p-19);
    p = KSM_FMA(p, z, -0x1.a01a01a01a01ap-13);
    p = KSM_FMA(p, z, 0x1.1111111111111p-7);
    p = KSM_FMA(p, z, -0x1.5555555555555p-3);
    return r + r * (z * p);
}

/* cos r, |r| <= pi/4 (Taylor to r^14, truncation < 1.1e-15) */
KSM_FN double ksm_cos_kernel(double r)
{
    double z = r * r;
    double p = -0x1.93974a8c07c9dp-37;
    p = KSM_FMA(p, z, 0x1.1eed8eff8d898p-29);
    p = KSM_FMA(p, z, -0x1.27e4fb7789f5cp-22);
    p = KSM_FMA(p, z, 0x1.a01a01a01a01ap-16);
    p = KSM_FMA(p, z, -0x1.6c16c16c16c17p-10);
    p = KSM_FMA(p, z, 0x1.5555555555555p-5);
    p = KSM_FMA(p, z, -0x1.0000000000000p-1);
    return 1.0 + z * p;
}

/* Cody-Waite reduction by pi/2: x = k*pi/2 + r, |r| <= pi/4 (+eps); valid for |x| < 2^20 */
KSM_FN double ksm_reduce_pio2(double x, int* quadrant)
{
    double k = __builtin_floor(x * 0x1.45f306dc9c883p-1 + 0.5);
    double r = (x - k * 0x1.921fb54400000p+0) - k * 0x1.0b4611a626331p-34;
    *quadrant = (int)((long long)k & 3);
    return r;
}

KSM_FN float kajo_sinf(float xf)
{
    double x = (double)xf;
    if (!(__builtin_fabs(x) < 1048576.0))
        return (float)(x - x); /* NaN for NaN/inf/huge: never produced by the integrator */
    int q;
    double r = ksm_reduce_pio2(x, &q);
    double s = ksm_sin_kernel(r);
    double c = ksm_cos_kernel(r);
    double v = (q & 1) ? c : s;
    return (float)((q & 2) ? -v : v);
}

KSM_FN float kajo_cosf(float xf)
{
    double x = (double)xf;
    if (!(__builtin_fabs(x) < 1048576.0))
        return (float)(x - x);
    int q;
    double r = ksm_reduce_pio2(x, &q);
    double s = ksm_sin_kernel(r);
    double c = ksm_cos_kernel(r);
    double v = (q & 1) ? s : c;
    return (float)(((q + 1) & 2) ? -v : v);
}

/* asin x for |x| <= 0.5: x + x z P(z), z = x^2, P = degree-9 interpolant of (asin(x)/x - 1)/z at the Chebyshev nodes of
   [0, 1/4] (computed with mpmath at 60 digits, coefficients rounded to binary64); max relative error 1.4e-14 */
KSM_FN double ksm_asin_kernel(double x)
{
    double z = x * x;
    double p = 0x1.c93a92d53b4f1p-6;
    p = KSM_FMA(p, z, -0x1.815314c864b09p-9);
    p = KSM_FMA(p, z, 0x1.00d47e7966d94p-6);
    p = KSM_FMA(p, z, 0x1.b02442413f6bap-7);
    p = KSM_FMA(p, z, 0x1.1dc2ef640046fp-6);
    p = KSM_FMA(p, z, 0x1.6e72146fda29ep-6);
    p = KSM_FMA(p, z, 0x1.f1c81c59ea536p-6);
    p = KSM_FMA(p, z, 0x1.6db6d8e71341bp-5);
    p = KSM_FMA(p, z, 0x1.33333335a9cd6p-4);
    p = KSM_FMA(p, z, 0x1.5555555554f05p-3);
    return x + x * (z * p);
}

#define KSM_PIO2 0x1.921fb54442d18p+0
#define KSM_PI 0x1.921fb54442d18p+1

KSM_FN double ksm_asin(double x)
{
    double a = __builtin_fabs(x);
    if (a <= 0.5)
        return ksm_asin_kernel(x);
    /* |x| > 1 gives sqrt(negative) = NaN; NaN input lands here too and stays NaN */
    double s = __builtin_sqrt((1.0 - a) * 0.5);
    double r = KSM_PIO2 - 2.0 * ksm_asin_kernel(s);
    return x < 0.0 ? -r : r;
}

KSM_FN float kajo_asinf(float x)
{
    return (float)ksm_asin((double)x);
}

KSM_FN float kajo_acosf(float xf)
{
    double x = (double)xf;
    double a = __builtin_fabs(x);
    if (a <= 0.5)
        return (float)(KSM_PIO2 - ksm_asin_kernel(x));
    double s = __builtin_sqrt((1.0 - a) * 0.5);
    double t = 2.0 * ksm_asin_kernel(s);
    return (float)(x < 0.0 ? KSM_PI - t : t);
}

/* natural log of a positive, normal double */
KSM_FN double ksm_log(double x)
{
    uint64_t b = ksm_bits(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    uint64_t mb = (b & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m = ksm_from_bits(mb);
    if (m > 0x1.6a09e667f3bcdp+0) { /* sqrt 2 */
        m = m * 0.5;
        e = e + 1;
    }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 0x1.af286bca1af28p-4; /* 2/19: the series 2 s (1 + z/3 + z^2/5 + ...) to z^9, truncation < 3e-17 */
    p = KSM_FMA(p, z, 0x1.e1e1e1e1e1e1ep-4);
    p = KSM_FMA(p, z, 0x1.1111111111111p-3);
    p = KSM_FMA(p, z, 0x1.3b13b13b13b14p-3);
    p = KSM_FMA(p, z, 0x1.745d1745d1746p-3);
    p = KSM_FMA(p, z, 0x1.c71c71c71c71cp-3);
    p = KSM_FMA(p, z, 0x1.2492492492492p-2);
    p = KSM_FMA(p, z, 0x1.999999999999ap-2);
    p = KSM_FMA(p, z, 0x1.5555555555555p-1);
    p = KSM_FMA(p, z, 0x1.0000000000000p+1);
    double de = (double)e;
    return de * 0x1.62e42fef00000p-1 + (de * 0x1.473de6af278edp-34 + s * p);
}

/* e^t, finite t */
KSM_FN double ksm_exp(double t)
{
    if (t > 709.0)
        return ksm_from_bits(0x7ff0000000000000ull);
    if (t < -745.0)
        return 0.0;
    double k = __builtin_floor(t * 0x1.71547652b82fep+0 + 0.5);
    double r = (t - k * 0x1.62e42fef00000p-1) - k * 0x1.473de6af278edp-34;
    double p = 0x1.1eed8eff8d898p-29; /* 1/12!: Taylor to r^12, |r| <= ln2/2, truncation < 2e-16 */
    p = KSM_FMA(p, r, 0x1.ae64567f544e4p-26);
    p = KSM_FMA(p, r, 0x1.27e4fb7789f5cp-22);
    p = KSM_FMA(p, r, 0x1.71de3a556c734p-19);
    p = KSM_FMA(p, r, 0x1.a01a01a01a01ap-16);
    p = KSM_FMA(p, r, 0x1.a01a01a01a01ap-13);
    p = KSM_FMA(p, r, 0x1.6c16c16c16c17p-10);
    p = KSM_FMA(p, r, 0x1.1111111111111p-7);
    p = KSM_FMA(p, r, 0x1.5555555555555p-5);
    p = KSM_FMA(p, r, 0x1.5555555555555p-3);
    p = KSM_FMA(p, r, 0x1.0000000000000p-1);
    p = KSM_FMA(p, r, 1.0);
    p = KSM_FMA(p, r, 1.0);
    int ki = (int)k;
    int k1 = ki >> 1;
    int k2 = ki - k1;
    return (p * ksm_pow2(k1)) * ksm_pow2(k2);
}

/*
 * powf for the domain the integrator uses: x >= 0 (a clamped cosine, a uniform variate or
 * a clamped colour), any finite y. x < 0 or NaN operands give NaN; pow(0, 0) = 1,
 * pow(0, y > 0) = 0, pow(0, y < 0) = inf; pow(x, 0) = 1; pow(1, y) = 1; pow(inf, .) is not
 * special-cased beyond what exp(y log x) yields.
 */
KSM_FN float kajo_powf(float xf, float yf)
{
    double x = (double)xf;
    double y = (double)yf;
    if (!(x == x) || !(y == y))
        return (float)(x + y);
    if (y == 0.0)
        return 1.0f;
    if (x < 0.0)
        return (float)__builtin_sqrt(x); /* NaN */
    if (x == 0.0)
        return y > 0.0 ? 0.0f : (float)ksm_from_bits(0x7ff0000000000000ull);
    if (x == ksm_from_bits(0x7ff0000000000000ull))
        return y > 0.0 ? (float)x : 0.0f;
    return (float)ksm_exp(y * ksm_log(x));
}

#endif /* KAJO_STRICTMATH_H */
