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
 * Here u = (float)((double)r * C) with C = 2^-32 + 2^-64 (0x1.00000001p-32, the binary64 nearest to
 * 1 / 4294967295): one binary64 multiply instead of the divide (a ~40-instruction sequence on the GPU).
 * r / (2^32 - 1) = r * 2^-32 * (1 + 2^-32 + 2^-64 + ...) and r * C drops the terms from 2^-96 on; the
 * two binary64 results can differ in the last place, but never across a binary32 rounding boundary:
 * tests/test_rng_exhaustive.py compares all three functions with the reference's expressions for
 * EVERY one of the 2^32 values of r (host build; the device executes the same IEEE binary64 multiply,
 * add and conversions, and tests/test_gpu_math.py compares device and host bits).  The neighbours of C
 * fail that test (2 and 14 inputs), C itself does not.  The jitter and the angle are then the
 * reference's own expressions on (double)u: (x - 0.5) * 2 is exact, so ((x - 0.5) * 2) * c equals
 * (x - 0.5) * (2c) with 2c exact; 2 * 3.14159 is exact, 6.28318.
 *
 * (Round 1 derived u with integer operations - "round r to 24 significant bits, ties away, two
 * exceptions" - which is the same function in ~16 instructions instead of 3.)
 * Plain C99 / C++ / HIP.
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

#define RT_RNG_INV_2P32M1 0x1.00000001p-32    /* 2^-32 + 2^-64 */

/* (float)(r / 4294967295.0) */
RT_RNG_HD float rt_u01(uint32_t r)
{
    return (float)((double)r * RT_RNG_INV_2P32M1);
}

/* (float)(((double)u - 0.5) * 2 * (double)0.001f) with u = rt_u01(r).
 * As ONE binary32 fused multiply-add (round 4): K = 2 * 0.001f is a binary32 number (a doubled one) and so is K / 2 = 0.001f, and
 * (u - 0.5) * K = u * K - K / 2 as real numbers; the reference's expression rounds that value to binary32 at the end (its binary64
 * steps in between are exact or err far below a binary32 half-ulp), and fmaf(u, K, -K / 2) rounds the exact value once.  The same
 * function: tests/test_rng_exhaustive.py compares it with the reference's expression for all 2^32 values of r.  One v_fma_f32
 * instead of v_cvt_f64_f32 + v_add_f64 + v_mul_f64 + v_cvt_f32_f64, three times per generated ray.  RT_RNG_JITTER_F64 keeps the
 * binary64 form (A/B builds). */
RT_RNG_HD float rt_jitter(uint32_t r)
{
#ifdef RT_RNG_JITTER_F64
    const double K = 2.0 * (double)0.001f;
    return (float)(((double)rt_u01(r) - 0.5) * K);
#else
    return __builtin_fmaf(rt_u01(r), 2.0f * 0.001f, -0.001f);
#endif
}

/* (float)(2 * 3.14159 * (double)u) with u = rt_u01(r).
 * (Round 4 also tried this one in binary32: fmaf(u, CH, u * CL) with CH + CL = 6.28318 split into two binary32 numbers is the same
 * function for all 2^32 values of r - checked exhaustively - and two instructions instead of three, but the kernels with a mesh got
 * 0.4 % slower with it and the others no faster than with the jitter alone: profiles/r04/experiments/rng_binary32_forms.txt.) */
RT_RNG_HD float rt_theta(uint32_t r)
{
    return (float)(6.28318 * (double)rt_u01(r));
}

#endif
