/* Exhaustive check of ray-tracer_amd/csrc/rt_rng.h against the reference's expressions
 * (src/utils.cu:228,236; src/ray.cu:135-137) for every 32-bit hash output r.
 * Built and run by tests/test_rng_exhaustive.py:  gcc -O2 -ffp-contract=off -pthread
 * argv[1] = number of threads, argv[2] = stride (1 = all 2^32 values).  Prints the number of
 * mismatches per function; exit status 0 iff all are zero. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../ray-tracer_amd/csrc/rt_math.h"
#include "../ray-tracer_amd/csrc/rt_rng.h"

static uint64_t g_stride = 1;
static int g_threads = 1;
static uint64_t bad_u[256], bad_j[256], bad_t[256], bad_l[256], bad_c[256];

static inline uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static void *work(void *arg)
{
    int id = (int)(intptr_t)arg;
    const float RANGE = 0.001f;
    uint64_t bu = 0, bj = 0, bt = 0, bl = 0, bc = 0;
    uint64_t lo = ((1ull << 32) / g_threads) * id, hi = id == g_threads - 1 ? (1ull << 32) : ((1ull << 32) / g_threads) * (id + 1);
    for (uint64_t x = lo; x < hi; x += g_stride) {
        uint32_t r = (uint32_t)x;
        volatile double q = (double)r / 4294967295.0;          /* the reference's divide */
        float u = (float)q;
        float jit = (float)(((double)u - 0.5) * 2 * (double)RANGE);
        float theta = (float)(2 * 3.14159 * (double)u);
        bu += fbits(rt_u01(r)) != fbits(u);
        bj += fbits(rt_jitter(r)) != fbits(jit);
        bt += fbits(rt_theta(r)) != fbits(theta);
        /* the Box-Muller calls: log and cos restricted to the arguments a draw can produce against the general functions */
        bl += fbits(rt_logf_0_1(rt_u01(r), 0)) != fbits(rt_logf(rt_u01(r)));
        bl += fbits(rt_logf_0_1(rt_u01(r), 1)) != fbits(rt_logf(rt_u01(r)));      /* ... with the division as reciprocal + residual step */
        bc += fbits(rt_cosf_0_2pi(rt_theta(r))) != fbits(rt_cosf(rt_theta(r)));
    }
    bad_u[id] = bu; bad_j[id] = bj; bad_t[id] = bt; bad_l[id] = bl; bad_c[id] = bc;
    return NULL;
}

int main(int argc, char **argv)
{
    g_threads = argc > 1 ? atoi(argv[1]) : 1;
    if (g_threads < 1) g_threads = 1;
    if (g_threads > 256) g_threads = 256;
    g_stride = argc > 2 ? strtoull(argv[2], NULL, 10) : 1;
    if (g_stride < 1) g_stride = 1;
    pthread_t th[256];
    for (int i = 0; i < g_threads; i++) pthread_create(&th[i], NULL, work, (void *)(intptr_t)i);
    uint64_t u = 0, j = 0, t = 0, l = 0, c = 0;
    for (int i = 0; i < g_threads; i++) { pthread_join(th[i], NULL); u += bad_u[i]; j += bad_j[i]; t += bad_t[i]; l += bad_l[i]; c += bad_c[i]; }
    /* the ends of the range, whatever the stride */
    const uint32_t edge[] = {0u, 1u, 0x00ffffffu, 0x01000000u, 0x01000001u, 0x01ffffffu, 0x7fffffffu, 0x80000000u, 0xffffff7fu, 0xffffff80u, 0xfffffffeu, 0xffffffffu};
    for (unsigned i = 0; i < sizeof edge / sizeof edge[0]; i++) {
        float uu = (float)((double)edge[i] / 4294967295.0);
        u += fbits(rt_u01(edge[i])) != fbits(uu);
        j += fbits(rt_jitter(edge[i])) != fbits((float)(((double)uu - 0.5) * 2 * (double)0.001f));
        t += fbits(rt_theta(edge[i])) != fbits((float)(2 * 3.14159 * (double)uu));
        l += fbits(rt_logf_0_1(rt_u01(edge[i]), 0)) != fbits(rt_logf(rt_u01(edge[i])));
        l += fbits(rt_logf_0_1(rt_u01(edge[i]), 1)) != fbits(rt_logf(rt_u01(edge[i])));
        c += fbits(rt_cosf_0_2pi(rt_theta(edge[i]))) != fbits(rt_cosf(rt_theta(edge[i])));
    }
    printf("mismatches u01=%llu jitter=%llu theta=%llu log=%llu cos=%llu\n", (unsigned long long)u, (unsigned long long)j, (unsigned long long)t,
           (unsigned long long)l, (unsigned long long)c);
    return (u | j | t | l | c) ? 1 : 0;
}
