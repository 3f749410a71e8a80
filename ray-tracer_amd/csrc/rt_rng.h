/*
 * rt_rng.h — the reference's per-pixel random stream (src/utils.cu:220-239) and the three
 * quantities the hot path derives from one draw, without the binary64 DIVISION the reference
 * performs per draw.
 *
 * Reference, per draw:   r = pcg_hash(state);  u = (float)(r / 4294967295.0)      (binary64 divide)
 *   jitter offset        (float)(((double)u - 0.5) * 2 * (double)0.001f)          (src/ray.cu:135-137)
 *   Box-Muller angle     (float)(2 * 3.14159 * (double)u)                         (src/utils.cu:236)
 *   Box-Muller radius    uses logf(u)                                             (src/utils.cu:237)
 *
 * Identity used: r / (2^32 - 1) = r * 2^-32 * (1 + 2^-32 + 2^-64 + ...).  The correction lifts an
 * integer r by less than 1, so rounding it to binary64 and then to binary32 is "round r to 24
 * significant bits, ties AWAY from zero" (a tie can only be broken upwards), and
 * r = 2^32 - 1 rounds to 2^32, i.e. u = 1.  Two values of r are the exception: for
 * r = 2^32 - 129 and r = 2^32 - 641 the binary64 quotient rounds onto a binary32 midpoint that
 * the exact quotient lies just below, and round-half-even then goes UP (double rounding); they
 * are bumped onto the midpoint first.  Hence u = R * 2^-32 with R an integer with at most 24
 * significant bits, R in [0, 2^32], obtained with integer operations; (double)u is then
 * R * 2^-32 exactly, so the two binary64 products keep their single binary64 rounding:
 *   jitter = (float)((R - 2^31) * 2^-32 * (2 * (double)0.001f))
 *   theta  = (float)(R * (6.28318 * 2^-32))
 *
 * tests/test_rng_exhaustive.py checks all three against the reference expressions for EVERY
 * one of the 2^32 values of r (host build; the device executes the same IEEE binary64 multiply,
 * add and conversions).  Plain C99 / C++ / HIP.
 */
#ifndef RT_RNG_H
#define RT_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RT_RNG_HD __host__ __device__ static inline
#else
#define RT_RNG_HD static inline
#endif

/* src/utils.cu:222-226: LCG step + PCG output hash */
RT_RNG_HD uint32_t rt_pcg_next(uint32_t *state)
{
    uint32_t ns = *state * 747796405u + 2891336453u;
    *state = ns;
    uint32_t r = ((ns >> ((ns >> 28) + 4u)) ^ ns) * 277803737u;
    return (r >> 22) ^ r;
}

/* r rounded to 24 significant bits, ties away; *is_one is set when the result is 2^32 (u == 1),
 * in which case the returned value is unspecified */
RT_RNG_HD uint32_t rt_round24(uint32_t r, int *is_one)
{
    r += ((r | 0x200u) == 0xffffff7fu);     /* the two double-rounding cases: 0xfffffd7f, 0xffffff7f */
    int lz = r ? __builtin_clz(r) : 32;
    int s = 8 - lz;                         /* bits to drop; <= 0 when r < 2^24 */
    s = s < 0 ? 0 : s;
    uint32_t half = (1u << s) >> 1;
    uint32_t r2 = r + half;
    *is_one = r2 < r;                       /* carried out of 32 bits */
    return (r2 >> s) << s;
}

/* (float)(r / 4294967295.0) */
RT_RNG_HD float rt_u01(uint32_t r)
{
    int one;
    uint32_t R = rt_round24(r, &one);
    return one ? 1.0f : (float)R * 2.3283064365386963e-10f;   /* exact conversion, exact scaling by 2^-32 */
}

/* (float)(((double)u - 0.5) * 2 * (double)0.001f) with u = rt_u01(r) */
RT_RNG_HD float rt_jitter(uint32_t r)
{
    const double K = (2.0 * (double)0.001f) * 2.3283064365386963e-10;   /* 2c * 2^-32, exact */
    int one;
    uint32_t R = rt_round24(r, &one);
    double t = (double)R - 2147483648.0;                                 /* exact */
    float v = (float)(t * K);
    return one ? 0.001f : v;                                             /* u = 1: 0.5 * 2 * c = c */
}

/* (float)(2 * 3.14159 * (double)u) with u = rt_u01(r) */
RT_RNG_HD float rt_theta(uint32_t r)
{
    const double K = 6.28318 * 2.3283064365386963e-10;                  /* exact scaling */
    int one;
    uint32_t R = rt_round24(r, &one);
    float v = (float)((double)R * K);
    return one ? (float)6.28318 : v;
}

#endif
