/*
 * kajo_strictmath.h -- elementary functions of the STRICT numerics mode (interface contract).
 *
 * The reference calls libm's sinf/cosf/asinf/acosf/powf (renderer/cpu/Random.cpp:77-102,
 * renderer/cpu/Light.cpp:26-49, renderer/cpu/BSDF.cpp:61-74, renderer/Image.cpp:14-17).
 * glibc's results cannot be reproduced bit for bit by a GPU's math library, and one
 * flipped branch changes a whole path, so "does the GPU integrator
 * take exactly the decisions the CPU integrator takes" is only testable when both sides
 * evaluate these five functions with the same arithmetic. This header is that arithmetic.
 *
 * Version 2 (round 2). sin, cos, asin, acos are evaluated in IEEE binary32 with fused multiply-adds
 * only -- Cody-Waite reduction against a three-float pi/2, error-free sums (TwoSum / Fast2Sum / the
 * exact FMA residual of a square or a square root) wherever a rounding error would otherwise reach the
 * last place, short polynomials -- and are branch-free: a wave64 of lanes in different quadrants or on
 * both sides of |x| = 1/2 executes ONE instruction stream of full-rate v_fma_f32 (version 1 evaluated
 * binary64 series behind per-lane branches: half-rate arithmetic, 64-bit constants in scalar registers,
 * every branch executed by every wave; it was half of the STRICT kernel's time). powf keeps binary64 --
 * y * log2(x) needs ~36 bits -- but without the division and with series no longer than that needs.
 *
 * Every operation is +, -, *, fma, sqrt, floor or an integer bit move on IEEE values, all correctly
 * rounded on x86-64 (g++ -ffp-contract=off; fmaf/fma are single instructions with -mfma and exact
 * library functions without) and on gfx950 (hipcc -ffp-contract=off; v_fma_f32/v_fma_f64, and hipcc's
 * default correctly rounded sqrt), so both produce identical bits.
 *
 * Accuracy, checked exhaustively on the CPU against the correctly rounded value (binary64 libm, rounded):
 *   sin, cos   every binary32 in [-2, 6.5] (the integrator's arguments lie in [-pi/2, 2 pi]): within 1 ulp,
 *              equal to the correctly rounded value for 99.93 % / 99.95 % of the arguments
 *   asin, acos every binary32 in [-1, 1]: within 1 ulp, correctly rounded for 99.97 % / 99.98 %
 *   pow        x over every 7th binary32 in (0, 1] (incl. subnormals) for y in {100, 1000, 10, 3, 2.2, .5,
 *              1/2.2, 1/11, 1/101}: within 1 ulp, correctly rounded for >= 99.968 %
 * (tools/strictmath_exhaustive.c; tests/test_strictmath.py samples the same claims in the CPU suite and
 * checks the GPU's bits against the CPU's.)
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

#define KSM_FMAF(a, b, c) __builtin_fmaf((a), (b), (c))

/* One rounding per Horner step of the binary64 series of powf: fma is exactly specified, so x86-64 and gfx950 agree. */
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
/* The LEADING coefficient of a Horner chain is a vector operand (v_fma_f64 takes one scalar source). Left to itself the compiler
   keeps it in a VGPR pair across the whole render loop -- a loop invariant -- and, in a loop short of registers, spills it; made
   here from two literals where it is used (two v_mov_b32), it occupies nothing in between. */
KSM_FN double ksm_lead_coeff(uint32_t lo, uint32_t hi)
{
    __asm__ volatile("" : "+v"(lo), "+v"(hi));
    const uint64_t u = ((uint64_t)hi << 32) | lo;
    double d;
    __builtin_memcpy(&d, &u, 8);
    return d;
}
#define KSM_LEAD(lo, hi, value) ksm_lead_coeff((lo), (hi))
#else
#define KSM_FMA(a, b, c) __builtin_fma((a), (b), (c))
#define KSM_LEAD(lo, hi, value) (value)
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

KSM_FN uint32_t ksm_bits32(float f)
{
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
}

KSM_FN float ksm_from_bits32(uint32_t u)
{
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

/* ---- sin / cos ---------------------------------------------------------------------------------------------- */

/* x = k pi/2 + (r + lo), |r| <= pi/4 (+ a rounding), *q = k mod 4; valid for |x| < 16.
   pi/2 = P1 + P2 + P3 to 2^-75. k P1 is subtracted exactly (for k != 0 both x and k P1 are multiples of 2^-24 and
   the difference is below 1); k P2 is formed exactly as ph + pl and added with a TwoSum, because next to a multiple of
   pi/2 the first difference cancels to a few units of 2^-24 and is then SMALLER than k P2. */
KSM_FN void ksm_reduce_pio2f(float x, float* r_, float* lo_, uint32_t* q)
{
    const float magic = 12582912.0f; /* 1.5 * 2^23: the integer nearest x * 2/pi lands in the low mantissa bits */
    const float t = KSM_FMAF(x, 0x1.45f306p-1f, magic);
    *q = ksm_bits32(t) & 3u;
    const float k = t - magic;
    const float P1 = 0x1.921fb6p+0f, P2 = -0x1.777a5cp-25f, P3 = -0x1.ee59dap-50f;
    const float rh = KSM_FMAF(-k, P1, x);
    const float ph = -k * P2;
    const float pl = KSM_FMAF(-k, P2, -ph);
    const float r = rh + ph;
    const float bb = r - rh;
    const float e = (rh - (r - bb)) + (ph - bb);
    *lo_ = KSM_FMAF(-k, P3, e + pl);
    *r_ = r;
}

/* sin(r + lo) and cos(r + lo), |r| <= pi/4, |lo| <= ulp(r)/2 */
KSM_FN void ksm_sincos_kernelsf(float r, float lo, float* s, float* c)
{
    const float z = r * r;
    const float zl = KSM_FMAF(r, r, -z); /* r^2 = z + zl exactly */
    /* sin = r + r z S(z) + lo (1 - z/2); S = degree-3 interpolant of (sin r - r) / r^3 at the Chebyshev nodes of [0, (pi/4)^2] */
    float ps = 0x1.6dbbeep-19f;
    ps = KSM_FMAF(ps, z, -0x1.a013a2p-13f);
    ps = KSM_FMAF(ps, z, 0x1.11110ep-7f);
    ps = KSM_FMAF(ps, z, -0x1.555556p-3f);
    const float w = r * z;
    const float lo2 = KSM_FMAF(lo * z, -0.5f, lo);
    *s = r + KSM_FMAF(w, ps, lo2);
    /* cos = (1 - z/2) + [rounding error of that] + z^2 C(z) - r lo - zl/2 */
    float pc = -0x1.2522e6p-22f;
    pc = KSM_FMAF(pc, z, 0x1.a015c0p-16f);
    pc = KSM_FMAF(pc, z, -0x1.6c16c0p-10f);
    pc = KSM_FMAF(pc, z, 0x1.555556p-5f);
    const float hz = 0.5f * z;
    const float wc = 1.0f - hz;
    const float ec = (1.0f - wc) - hz; /* exact */
    const float tail = KSM_FMAF(z * z, pc, ec) - KSM_FMAF(r, lo, 0.5f * zl);
    *c = wc + tail;
}

/* sin x and cos x together (the integrator always wants both of 2 pi s, Random.cpp:84-86, Light.cpp:43-44) */
KSM_FN void kajo_sincosf(float x, float* sn, float* cs)
{
    float r, lo, s, c;
    uint32_t q;
    ksm_reduce_pio2f(x, &r, &lo, &q);
    ksm_sincos_kernelsf(r, lo, &s, &c);
    const float a = (q & 1u) ? c : s;
    const float b = (q & 1u) ? s : c;
    const float vs = ksm_from_bits32(ksm_bits32(a) ^ ((q & 2u) << 30));
    const float vc = ksm_from_bits32(ksm_bits32(b) ^ (((q + 1u) & 2u) << 30));
    const int ok = __builtin_fabsf(x) < 16.0f; /* NaN for NaN/inf/large: never produced by the integrator */
    const float bad = ksm_from_bits32(0x7fc00000u);
    *sn = ok ? vs : bad;
    *cs = ok ? vc : bad;
}

KSM_FN float kajo_sinf(float x)
{
    float s, c;
    kajo_sincosf(x, &s, &c);
    return s;
}

KSM_FN float kajo_cosf(float x)
{
    float s, c;
    kajo_sincosf(x, &s, &c);
    return c;
}

/* ---- asin / acos -------------------------------------------------------------------------------------------- */

#define KSM_PIO2_HI 0x1.921fb6p+0f
#define KSM_PIO2_LO -0x1.777a5cp-25f
#define KSM_PI_HI 0x1.921fb6p+1f
#define KSM_PI_LO -0x1.777a5cp-24f

/* a = |x| in [0, 1]. Returns b and t with asin(b*) = b + t, where b* = a for a <= 1/2 and otherwise
   b* = sqrt((1 - a) / 2) carried as the correctly rounded root b plus its exact residual (asin a = pi/2 - 2 asin b*).
   t = b z A(z) + c, z = b*^2; A = degree-6 interpolant of (asin(b)/b - 1)/b^2 at the Chebyshev nodes of [0, 1/4]. */
KSM_FN void ksm_asin_coref(float a, int* small_, float* b_, float* t_)
{
    const int small = a <= 0.5f;
    const float w = (1.0f - a) * 0.5f; /* exact for a >= 1/2 */
    const float s = __builtin_sqrtf(w);
    const float es = KSM_FMAF(-s, s, w); /* w - s^2, exact: the true root is s + es / (2 s) */
    /* 1/s to 1.5 %: integer seed and one Newton step -- the same bits everywhere, unlike a hardware reciprocal */
    float r0 = ksm_from_bits32(0x7EF311C7u - ksm_bits32(s));
    r0 = r0 * KSM_FMAF(-s, r0, 2.0f);
    const float c = (0.5f * es) * r0;
    const float b = small ? a : s;
    const float z = small ? a * a : w;
    const float cc = small ? 0.0f : c;
    float p = 0x1.fbaa70p-6f;
    p = KSM_FMAF(p, z, 0x1.5a41fcp-7f);
    p = KSM_FMAF(p, z, 0x1.82e318p-6f);
    p = KSM_FMAF(p, z, 0x1.efed0cp-6f);
    p = KSM_FMAF(p, z, 0x1.6dc0f6p-5f);
    p = KSM_FMAF(p, z, 0x1.33331ep-4f);
    p = KSM_FMAF(p, z, 0x1.555556p-3f);
    *t_ = KSM_FMAF(b * z, p, cc);
    *b_ = b;
    *small_ = small;
}

KSM_FN float kajo_asinf(float x)
{
    const float a = __builtin_fabsf(x);
    float b, t;
    int small;
    ksm_asin_coref(a, &small, &b, &t); /* |x| > 1: sqrt of a negative number, NaN */
    const float rs = b + t;
    /* pi/2 - 2 (b + t): the constant minus 2 b with its rounding error kept (Fast2Sum), then the small terms */
    const float b2 = 2.0f * b;
    const float u = KSM_PIO2_HI - b2;
    const float eu = (KSM_PIO2_HI - u) - b2;
    const float rl = u + (eu + KSM_FMAF(-2.0f, t, KSM_PIO2_LO));
    const float r = small ? rs : rl;
    return ksm_from_bits32(ksm_bits32(r) | (ksm_bits32(x) & 0x80000000u));
}

KSM_FN float kajo_acosf(float x)
{
    const float a = __builtin_fabsf(x);
    float b, t;
    int small;
    ksm_asin_coref(a, &small, &b, &t);
    const int neg = x < 0.0f;
    /* |x| <= 1/2: pi/2 -+ (a + t).   x > 1/2: 2 (b + t).   x < -1/2: pi - 2 (b + t). */
    const float m = small ? b : 2.0f * b;
    const float tt = small ? t : 2.0f * t;
    const float sg = (small && neg) ? 1.0f : -1.0f;
    const float kHi = small ? KSM_PIO2_HI : KSM_PI_HI;
    const float kLo = small ? KSM_PIO2_LO : KSM_PI_LO;
    const float sm = sg * m;
    const float u = kHi + sm;
    const float eu = (kHi - u) + sm;
    const float rc = u + (eu + KSM_FMAF(sg, tt, kLo));
    const float rpos = m + tt;
    return (!small && !neg) ? rpos : rc;
}

/* ---- pow ---------------------------------------------------------------------------------------------------- */

/*
 * powf for the domain the integrator uses: x >= 0 (a clamped cosine, a uniform variate or
 * a clamped colour), any finite y. x < 0 or NaN operands give NaN; pow(0, 0) = 1,
 * pow(0, y > 0) = 0, pow(0, y < 0) = inf; pow(x, 0) = 1; pow(1, y) = 1; pow(inf, y) = inf or 0 by the sign of y.
 * 2^(y log2 x) in binary64: x = 2^e m, m in [sqrt 1/2, sqrt 2); ln m = f + f^2 P(f), f = m - 1, P = degree-12 interpolant
 * of (ln(1 + f)/f - 1)/f (2e-11: 0.004 ulp of the result at y = 100); 2^r = 1 + r E(r) on |r| <= 1/2, E of degree 6.
 */
KSM_FN float kajo_powf(float xf, float yf)
{
    const double x = (double)xf, y = (double)yf;
    const uint64_t b = ksm_bits(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double m = ksm_from_bits((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    const int big = m > 0x1.6a09e667f3bcdp+0;
    m = big ? m * 0.5 : m;
    e += big;
    const double f = m - 1.0;
    double p = KSM_LEAD(0xc29da973u, 0xbfa8e19bu, -0x1.8e19bc29da973p-5);
    p = KSM_FMA(p, f, 0x1.736e0e73ee60bp-4);
    p = KSM_FMA(p, f, -0x1.79daf10d6f80dp-4);
    p = KSM_FMA(p, f, 0x1.724867257e45ep-4);
    p = KSM_FMA(p, f, -0x1.9595a848b180fp-4);
    p = KSM_FMA(p, f, 0x1.c6f0024436e8ap-4);
    p = KSM_FMA(p, f, -0x1.0018f0a2ade65p-3);
    p = KSM_FMA(p, f, 0x1.249408b37ef53p-3);
    p = KSM_FMA(p, f, -0x1.5554c938cf03dp-3);
    p = KSM_FMA(p, f, 0x1.9999907fd085ep-3);
    p = KSM_FMA(p, f, -0x1.00000092d41c6p-2);
    p = KSM_FMA(p, f, 0x1.5555555b18590p-2);
    p = KSM_FMA(p, f, -0x1.ffffffffcc906p-2);
    const double lg = __builtin_fma(f * f, p, f);
    const double L = __builtin_fma(lg, 0x1.71547652b82fep+0, (double)e);
    double t = y * L;
    t = t > 1100.0 ? 1100.0 : t;
    t = t < -1100.0 ? -1100.0 : t; /* a NaN passes through */
    const double n = __builtin_floor(t + 0.5);
    const double r = t - n;
    double q = KSM_LEAD(0x1594758eu, 0x3ef00a58u, 0x1.00a581594758ep-16);
    q = KSM_FMA(q, r, 0x1.443fffc90db59p-13);
    q = KSM_FMA(q, r, 0x1.5d879ead06a82p-10);
    q = KSM_FMA(q, r, 0x1.3b2a1b7152befp-7);
    q = KSM_FMA(q, r, 0x1.c6b08d883dca1p-5);
    q = KSM_FMA(q, r, 0x1.ebfbe045f4d3cp-3);
    q = KSM_FMA(q, r, 0x1.62e42fefa39efp-1);
    double v = __builtin_fma(q, r, 1.0);
    const int ni = (int)n, k1 = ni >> 1, k2 = ni - k1;
    v = (v * ksm_from_bits((uint64_t)(k1 + 1023) << 52)) * ksm_from_bits((uint64_t)(k2 + 1023) << 52);
    float res = (float)v;
    const float inf = __builtin_inff();
    res = xf == 0.0f ? (yf > 0.0f ? 0.0f : inf) : res;
    res = xf == inf ? (yf > 0.0f ? inf : 0.0f) : res;
    res = xf < 0.0f ? ksm_from_bits32(0x7fc00000u) : res;
    res = yf == 0.0f ? 1.0f : res;
    res = (xf != xf || yf != yf) ? xf + yf : res;
    return res;
}

#endif /* KAJO_STRICTMATH_H */
