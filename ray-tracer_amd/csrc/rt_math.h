/*
 * rt_math.h — deterministic single-precision transcendentals for the path tracer.
 *
 * Why this exists: the reference calls log/cos (device: src/utils.cu:236-238) and
 * tan/sin/cos (host: src/camera.cu:47, src/matrix.cu:139-140) from the platform libm.
 * A path tracer turns a 1-ulp difference in one of those into a different hit/miss
 * decision, so "CPU result == GPU result" needs transcendentals that are the SAME
 * function on both sides.  Everything below is built only from IEEE-754 binary32
 * + - * / and fma (plus integer bit operations, and binary64 + - * for the huge-argument
 * path of sin/cos and for asin), so it evaluates bit-identically under
 *   gcc   -O2 -ffp-contract=off -mfma     (oracle "det" mode, host code)
 *   hipcc -O3 -ffp-contract=off gfx950    (device code; f32 denormals are on, / and sqrt
 *                                          are correctly rounded by default in HIP).
 * The fused multiply-adds are EXPLICIT (RT_FMAF / RT_FMA: one rounding, the same on every
 * IEEE platform; without -mfma the host falls back to libm's correctly rounded fmaf) and only
 * inside these functions, which are this repository's own: the polynomials and the argument
 * reduction cost a third fewer instructions on the GPU (round 4).  The compilers stay at
 * -ffp-contract=off: the reference's own `a*b+c` expressions are two roundings everywhere.
 *
 * Accuracy (measured in tests/test_math.py against glibc): rt_logf <= 1 ulp on the
 * RNG's range, rt_sinf/rt_cosf <= 1 ulp absolute-in-[0,1] terms for |x| <= 3000.
 * They are NOT bit-identical to glibc or to CUDA's libdevice; DESIGN.md explains what
 * that means for parity.
 *
 * The header is plain C99 / C++ / HIP.
 */
#ifndef RT_MATH_H
#define RT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ static inline
#else
#define RT_HD static inline
#endif

#ifdef RT_MATH_NO_FMA        /* A/B builds only (tools/build_variants.py): the unfused forms, round 3's functions exactly */
#define RT_FMAF(a, b, c) ((a) * (b) + (c))
#else
#define RT_FMAF(a, b, c) __builtin_fmaf((a), (b), (c))
#endif

RT_HD uint32_t rt_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
RT_HD float rt_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

/* float -> int the way the reference's platform converts (CUDA cvt.rzi.s32.f32; gfx950's v_cvt_i32_f32 does the
 * same): truncation, NaN -> 0, out of range -> saturated.  C leaves NaN and out-of-range conversions undefined and
 * x86 returns INT_MIN for them, so a plain cast would make the CPU oracle and the GPU disagree exactly where the
 * reference's texture lookups (src/material.cu:90-99, :119-124) and display conversion (src/main.cu:343-371) can
 * meet such a value: a sphere's texture u is asin((P.y - c.y) / r), and at the pole that quotient exceeds 1 by an
 * ulp now and then -> NaN.  (Found by tests/soak/soak_parity.py: one pixel in 3,000 random scenes.) */
/* A NaN that reaches the frame buffer is stored as THE quiet NaN 0x7fc00000.  The reference can produce one (a
 * GRADIENT-textured sphere hit at its pole: the texture colour is the NaN u or v itself, src/material.cu:80-82) and
 * leaves its sign and payload to the platform - x86 and gfx950 already differ in the NaN that 0/0 gives - so
 * "bit-identical frames" needs one representation. */
RT_HD float rt_canon_nan(float x) { return x != x ? rt_u2f(0x7fc00000u) : x; }

RT_HD int rt_f2i(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return -2147483647 - 1;
    return (int)x;
}

/* ---- natural logarithm ------------------------------------------------------------
 * x = 2^k * m, m in [sqrt(1/2), sqrt(2)); f = m - 1; s = f / (2 + f); z = s^2.
 * log(1+f) = 2 atanh(s) = f - f^2/2 + s (f^2/2 + R),  R = z (2/3 + 2/5 z + 2/7 z^2 + 2/9 z^3)
 * (series truncation 2/11 z^5 < 2e-9 relative for |s| <= 0.1716).
 * log(0) = -inf and log(1) = 0 exactly: the reference's Box-Muller feeds both
 * (SURVEY.md App. A.13). */
RT_HD float rt_logf(float x)
{
    const float LN2_HI = 0.693145751953125f;      /* 0x3f317200, 15 significant bits: k*LN2_HI exact */
    const float LN2_LO = 1.428606765330187e-06f;  /* 0x35bfbe8e */
    uint32_t ix = rt_f2u(x);
    int k = 0;
    if (ix < 0x00800000u || (ix >> 31)) {
        if ((ix << 1) == 0u) return rt_u2f(0xff800000u);   /* +-0 -> -inf */
        if (ix >> 31) return rt_u2f(0x7fc00000u);          /* negative -> NaN */
        x = x * 33554432.0f;                               /* subnormal: scale by 2^25 */
        k = -25;
        ix = rt_f2u(x);
    } else if (ix >= 0x7f800000u) {
        return x + x;                                      /* inf, NaN */
    }
    k += (int)(ix >> 23) - 127;
    uint32_t m = ix & 0x007fffffu;
    uint32_t mb;
    if (m >= 0x003504f4u) { mb = m | 0x3f000000u; k += 1; }   /* m >= sqrt(2): use m/2 */
    else                  { mb = m | 0x3f800000u; }
    float f = rt_u2f(mb) - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float R = z * RT_FMAF(z, RT_FMAF(z, RT_FMAF(z, 0.2222222222222222f, 0.2857142857142857f), 0.4f), 0.6666666666666666f);
    float hfsq = 0.5f * f * f;
    float dk = (float)k;
    float t = RT_FMAF(s, hfsq + R, dk * LN2_LO);
    return RT_FMAF(dk, LN2_HI, f - (hfsq - t));
}

/* ---- sine / cosine -----------------------------------------------------------------
 * Argument reduction r = x - n*pi/2 with pi/2 split into three 13-bit chunks + a tail,
 * so n*chunk is exact for |n| <= 2048; beyond that (host-only use) the reduction runs in
 * binary64.  Kernels on [-pi/4, pi/4] are the Taylor polynomials through r^9 / r^10. */
RT_HD float rt__sin_k(float r)
{
    float z = r * r;
    float p = RT_FMAF(z, RT_FMAF(z, RT_FMAF(z, 2.7557319223985893e-06f, -1.9841269841269841e-04f), 8.3333333333333332e-03f), -1.6666666666666666e-01f);
    return RT_FMAF(r, z * p, r);
}

RT_HD float rt__cos_k(float r)
{
    float z = r * r;
    float p = RT_FMAF(z, RT_FMAF(z, RT_FMAF(z, -2.7557319223985888e-07f, 2.4801587301587302e-05f), -1.3888888888888889e-03f), 4.1666666666666664e-02f);
    float hz = 0.5f * z;
    return 1.0f - RT_FMAF(-(z * z), p, hz);
}

/* returns quadrant (n mod 4) and writes the reduced argument */
RT_HD int rt__rem_pio2(float x, float *r_out)
{
    const float TWO_OVER_PI = 0.6366197466850281f;        /* 0x3f22f983 */
    const float P1 = 1.570556640625f;                     /* 0x3fc90800, 13 bits */
    const float P2 = 0.0002396702766418457f;              /* 0x397b5000, 13 bits */
    const float P3 = 1.5890691429376602e-08f;             /* 0x32888000 */
    const float P4 = 2.5633440682570896e-12f;             /* 0x2c34611a */
    const float MAGIC = 12582912.0f;                      /* 1.5 * 2^23: (t + MAGIC) - MAGIC = rint(t) */
    uint32_t ax = rt_f2u(x) & 0x7fffffffu;
    if (ax <= 0x3f490fdau) { *r_out = x; return 0; }      /* |x| <= pi/4 */
    if (ax < 0x45490000u) {                               /* |x| < 3216: n <= 2048 */
        float fn = RT_FMAF(x, TWO_OVER_PI, MAGIC) - MAGIC;
        float r = RT_FMAF(-fn, P1, x);
        r = RT_FMAF(-fn, P2, r);
        r = RT_FMAF(-fn, P3, r);
        r = RT_FMAF(-fn, P4, r);
        *r_out = r;
        return (int)fn & 3;
    }
    if (ax >= 0x7f800000u) { *r_out = x - x; return 0; }  /* inf, NaN -> NaN */
    {
        const double D_TWO_OVER_PI = 0.6366197723675814;
        const double D_HI = 1.5707963267341256;           /* 33 bits of pi/2 */
        const double D_LO = 6.077100506506192e-11;
        const double D_MAGIC = 6755399441055744.0;        /* 1.5 * 2^52 */
        double xd = (double)x;
        double dn = (xd * D_TWO_OVER_PI + D_MAGIC) - D_MAGIC;
        double rd = (xd - dn * D_HI) - dn * D_LO;
        *r_out = (float)rd;
        /* dn can exceed int range for huge |x|: reduce mod 4 in binary64 first */
        double q = dn * 0.25;
        double qf = (q + D_MAGIC) - D_MAGIC;              /* rint(dn/4) */
        int n4 = (int)(dn - 4.0 * qf);                    /* in [-2, 2] */
        return n4 & 3;
    }
}

/* ---- the two calls of the Box-Muller transform, on the arguments it can produce (round 4) ------------------
 * normal_num (src/utils.cu:234-239) takes the logarithm of u = rt_u01(r) - a binary32 number in [0, 1], zero or normal, never negative,
 * NaN or infinite - and the cosine of theta = rt_theta(r) in [0, 6.28318].  rt_logf and rt_cosf spend a good part of their
 * instructions telling such arguments from the others (on the GPU: nested exec-mask branches, ~12 scalar instructions per call, and
 * the binary64 reduction for huge angles sits in the instruction stream).  These two are rt_logf and rt_cosf WITHOUT the cases that
 * cannot occur - every operation that remains is the same operation in the same order - and tests/test_rng_exhaustive.py checks
 * rt_logf_0_1(rt_u01(r)) == rt_logf(rt_u01(r)) and rt_cosf_0_2pi(rt_theta(r)) == rt_cosf(rt_theta(r)) for every one of the 2^32
 * values of r.  (Below pi / 4 rt_cosf skips the reduction; here it runs and finds n = 0, r = x: x * (2 / pi) rounds to 0 and
 * fma(-0, P, x) is x.)  The oracle keeps calling rt_logf and rt_cosf. */
/* a / b for operands far from the ends of the binary32 range, from the correctly rounded reciprocal r = RN(1 / b): q0 = RN(a * r) is
 * within an ulp of the quotient, its residual a - b * q0 is exact as an fma, and q0 + residual * r rounds to RN(a / b) (Markstein's
 * theorem; it needs every intermediate to be a normal number, which the caller's operand ranges must guarantee).  On the device the
 * reciprocal is v_rcp_f32 plus one correction step - equal to 1.0f / b for every b with 2^-126 <= |b| <= 2^126, checked for all
 * 2^32 patterns (tests/test_gpu_math.py) - on the host the division operator.  Six instructions on the GPU where the compiler's IEEE
 * expansion of a / b takes eleven (two v_div_scale, v_div_fmas and v_div_fixup deal with the operands this form excludes). */
RT_HD float rt__div_benign(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_rcpf(b);
    const float r = __builtin_fmaf(y, __builtin_fmaf(-b, y, 1.0f), y);
#else
    const float r = 1.0f / b;
#endif
    const float q0 = a * r;
    return RT_FMAF(RT_FMAF(-b, q0, a), r, q0);
}

/* short_divide (a compile-time constant at every call): s = f / (2 + f) through rt__div_benign instead of the division operator - the
 * same value; which one is faster depends on the kernel around it (profiles/r04/experiments/log_divide.txt) */
RT_HD float rt_logf_0_1(float x, int short_divide)
{
    const float LN2_HI = 0.693145751953125f;
    const float LN2_LO = 1.428606765330187e-06f;
    uint32_t ix = rt_f2u(x);
    if (ix == 0u) return rt_u2f(0xff800000u);                /* log(0) = -inf */
    int k = (int)(ix >> 23) - 127;
    uint32_t m = ix & 0x007fffffu;
    uint32_t mb;
    if (m >= 0x003504f4u) { mb = m | 0x3f000000u; k += 1; }
    else                  { mb = m | 0x3f800000u; }
    float f = rt_u2f(mb) - 1.0f;
    /* f is 0 or 2^-24 <= |f| <= 0.415, 2 + f in [1.7, 2.42]: quotient, product and residual are all normal numbers (or exact zeros) */
    float s = short_divide ? rt__div_benign(f, 2.0f + f) : f / (2.0f + f);
    float z = s * s;
    float R = z * RT_FMAF(z, RT_FMAF(z, RT_FMAF(z, 0.2222222222222222f, 0.2857142857142857f), 0.4f), 0.6666666666666666f);
    float hfsq = 0.5f * f * f;
    float dk = (float)k;
    float t = RT_FMAF(s, hfsq + R, dk * LN2_LO);
    return RT_FMAF(dk, LN2_HI, f - (hfsq - t));
}

RT_HD float rt_cosf_0_2pi(float x)
{
    const float TWO_OVER_PI = 0.6366197466850281f;
    const float P1 = 1.570556640625f, P2 = 0.0002396702766418457f, P3 = 1.5890691429376602e-08f, P4 = 2.5633440682570896e-12f;
    const float MAGIC = 12582912.0f;
    float fn = RT_FMAF(x, TWO_OVER_PI, MAGIC) - MAGIC;
    float r = RT_FMAF(-fn, P1, x);
    r = RT_FMAF(-fn, P2, r);
    r = RT_FMAF(-fn, P3, r);
    r = RT_FMAF(-fn, P4, r);
    int n = (int)fn & 3;
    float c = (n & 1) ? rt__sin_k(r) : rt__cos_k(r);
    return ((n + 1) & 2) ? -c : c;
}

RT_HD float rt_sinf(float x)
{
    float r;
    int n = rt__rem_pio2(x, &r);
    float s = (n & 1) ? rt__cos_k(r) : rt__sin_k(r);
    return (n & 2) ? -s : s;
}

RT_HD float rt_cosf(float x)
{
    float r;
    int n = rt__rem_pio2(x, &r);
    float c = (n & 1) ? rt__sin_k(r) : rt__cos_k(r);
    return ((n + 1) & 2) ? -c : c;
}

RT_HD float rt_tanf(float x) { return rt_sinf(x) / rt_cosf(x); }


/* ---- arcsine / arccosine / fifth power (binary64) -------------------------------------------
 * Needed by Ray::refract (src/ray.cu:101-102,123: acos/asin of double arguments, pow(x, 5))
 * and by the sphere texture coordinates (src/objects.cu:84-85: float asin/acos, evaluated here
 * through the binary64 routines and rounded once).  asin(x) on |x| <= 1/2 is its Maclaurin
 * series through x^55 (terms fall by > 4x each; the remainder is below 2e-18); for 1/2 < |x| <= 1
 * asin(x) = pi/2 - 2 asin(sqrt((1-|x|)/2)).  Only IEEE + - * / sqrt.  (The binary64 Horner chain stays multiply-then-add:
 * as 27 v_fma_f64 the compiler schedules it - and everything inlined around it - into 128 registers with spills.) */
RT_HD double rt__asin_series(double x)
{
    /* Horner, highest coefficient first; coefficients (2n)! / (4^n n!^2 (2n+1)) */
    const double z = x * x;
    double p = 0.0019650336162772837;
    p = p * z + 0.0020776610325181676;
    p = p * z + 0.0022014739737101384;
    p = p * z + 0.002338091892111975;
    p = p * z + 0.0024894486782468836;
    p = p * z + 0.0026578706382072901;
    p = p * z + 0.0028461784011089421;
    p = p * z + 0.0030578216492580306;
    p = p * z + 0.0032970595034734849;
    p = p * z + 0.0035692053938259347;
    p = p * z + 0.0038809645588376691;
    p = p * z + 0.0042409070936793632;
    p = p * z + 0.0046601434869150962;
    p = p * z + 0.0051533096823199046;
    p = p * z + 0.0057400376708419236;
    p = p * z + 0.0064472103118896487;
    p = p * z + 0.0073125258735988454;
    p = p * z + 0.0083903358096168151;
    p = p * z + 0.0097616095291940784;
    p = p * z + 0.011551800896139705;
    p = p * z + 0.013964843750000001;
    p = p * z + 0.017352764423076924;
    p = p * z + 0.022372159090909092;
    p = p * z + 0.030381944444444444;
    p = p * z + 0.044642857142857144;
    p = p * z + 0.074999999999999997;
    p = p * z + 0.16666666666666666;
    p = p * z + 1.0;
    return x * p;
}

#if defined(__HIPCC__)
#define RT_SQRT_F64(x) __builtin_sqrt(x)
#else
#define RT_SQRT_F64(x) __builtin_sqrt(x)
#endif

RT_HD double rt_asin(double x)
{
    const double PIO2_HI = 1.5707963267948966, PIO2_LO = 6.123233995736766e-17;
    const double ax = x < 0 ? -x : x;
    if (!(ax <= 1.0)) return (x - x) / (x - x);                 /* |x| > 1 or NaN -> NaN */
    if (ax <= 0.5) return rt__asin_series(x);
    const double s = RT_SQRT_F64((1.0 - ax) * 0.5);
    const double r = (PIO2_HI - 2.0 * rt__asin_series(s)) + PIO2_LO;
    return x < 0 ? -r : r;
}

RT_HD double rt_acos(double x)
{
    const double PIO2_HI = 1.5707963267948966, PIO2_LO = 6.123233995736766e-17;
    const double ax = x < 0 ? -x : x;
    if (!(ax <= 1.0)) return (x - x) / (x - x);
    if (ax <= 0.5) return (PIO2_HI - rt__asin_series(x)) + PIO2_LO;
    const double s = RT_SQRT_F64((1.0 - ax) * 0.5);
    const double t = 2.0 * rt__asin_series(s);
    return x > 0 ? t : (2.0 * PIO2_HI - t) + 2.0 * PIO2_LO;
}

RT_HD float rt_asinf(float x) { return (float)rt_asin((double)x); }
RT_HD float rt_acosf(float x) { return (float)rt_acos((double)x); }
/* pow(x, 5) as the reference's Schlick term uses it (src/ray.cu:195), in binary64 */
RT_HD double rt_pow5(double x) { const double x2 = x * x; return (x2 * x2) * x; }

#endif /* RT_MATH_H */
