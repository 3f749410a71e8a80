// Candidates for 1 / sqrt(m) and sqrt(m), correctly rounded as the compiler's IEEE expansions give them, from v_rsq_f32 (development tool, round 4).
// All binary32 inputs in [2^-64, 2^64]: which candidate equals sqrtf(m) / 1.0f / sqrtf(m) bit for bit; then the one-step square root on every
// positive normal input by exponent range, to see how wide its guard may be.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/rsq_forms.hip -o tools/ubench/rsq_forms && tools/ubench/rsq_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float sqrt_rsq(float m, float &y_out)
{
    const float y = __builtin_amdgcn_rsqf(m);
    const float s0 = m * y, h = 0.5f * y;
    const float e = __builtin_fmaf(-s0, s0, m);
    y_out = y;
    return __builtin_fmaf(e, h, s0);
}
__device__ __forceinline__ float sqrt_rsq2(float m, float &y_out)       // a second correction step
{
    float y; float s = sqrt_rsq(m, y);
    const float e = __builtin_fmaf(-s, s, m);
    y_out = y;
    return __builtin_fmaf(e, 0.5f * y, s);
}
__device__ __forceinline__ float cur_sqrt(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float rdn = __builtin_fmaf(-dn, s, x), rup = __builtin_fmaf(-up, s, x);
    s = rdn <= 0.0f ? dn : s;
    s = rup > 0.0f ? up : s;
    return s;
}
__device__ __forceinline__ float rcp1(float x) { float y = __builtin_amdgcn_rcpf(x); float e = __builtin_fmaf(-x, y, 1.0f); return __builtin_fmaf(y, e, y); }
__global__ void check(unsigned long long *bad)
{
    const uint32_t u = blockIdx.x * 1024u + threadIdx.x;          // 2^28 per pass
    for (uint32_t hi = 0; hi < 8; hi++) {                          // positive patterns only
        const uint32_t b = u | (hi << 28);
        if (b < 0x1f800000u || b > 0x5f800000u) continue;           // 2^-64 .. 2^64
        const float m = __uint_as_float(b);
        const float want_s = sqrtf(m), want_i = 1.0f / want_s;
        float y;
        const float s1 = sqrt_rsq(m, y);
        float e2 = __builtin_fmaf(-s1, y, 1.0f);
        const float i1 = __builtin_fmaf(e2, y, y);                                   // B1
        float e3 = __builtin_fmaf(-s1, i1, 1.0f);
        const float i2 = __builtin_fmaf(e3, i1, i1);                                 // B2
        const float i3 = rcp1(s1);                                                   // B3
        float y2; const float s2 = sqrt_rsq2(m, y2);
        const float i4 = rcp1(s2);                                                   // B4: two-step sqrt + short rcp
        const float i0 = rcp1(cur_sqrt(m));                                          // the tree's form
        auto ne = [](float a, float c) { return __float_as_uint(a) != __float_as_uint(c); };
        if (ne(s1, want_s)) atomicAdd(&bad[0], 1ull);
        if (ne(s2, want_s)) atomicAdd(&bad[1], 1ull);
        if (ne(i1, want_i)) atomicAdd(&bad[2], 1ull);
        if (ne(i2, want_i)) atomicAdd(&bad[3], 1ull);
        if (ne(i3, want_i)) atomicAdd(&bad[4], 1ull);
        if (ne(i4, want_i)) atomicAdd(&bad[5], 1ull);
        if (ne(i0, want_i)) atomicAdd(&bad[6], 1ull);
        atomicAdd(&bad[7], 1ull);
    }
}
__global__ void check_range(unsigned long long *bad)      // bad[256]: wrong one-step square roots per biased exponent
{
    const uint32_t u = blockIdx.x * 1024u + threadIdx.x;
    for (uint32_t hi = 0; hi < 8; hi++) {
        const uint32_t b = u | (hi << 28);
        if (b < 0x00800000u || b >= 0x7f800000u) continue;
        const float m = __uint_as_float(b);
        float y;
        if (__float_as_uint(sqrt_rsq(m, y)) != __float_as_uint(sqrtf(m))) atomicAdd(&bad[b >> 23], 1ull);
    }
}
int main()
{
    {
        unsigned long long *d2, h2[256];
        (void)hipMalloc(&d2, sizeof h2); (void)hipMemset(d2, 0, sizeof h2);
        hipLaunchKernelGGL(check_range, dim3(1u << 18), dim3(1024), 0, 0, d2);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h2, d2, sizeof h2, hipMemcpyDeviceToHost);
        int lo = -1, hi = -1;
        for (int e = 1; e < 255; e++) if (h2[e] == 0) { if (lo < 0) lo = e; hi = e; } else if (lo >= 0 && hi == e - 1) { printf("one-step sqrt from rsq: exact for biased exponents %d..%d (2^%d .. 2^%d), first wrong above: exponent %d (%llu wrong)\n", lo, hi, lo - 127, hi - 126, e, h2[e]); }
        unsigned long long below = 0; for (int e = 1; e < lo; e++) below += h2[e];
        printf("one-step sqrt from rsq: exact range of biased exponents %d..%d; wrong below it: %llu\n", lo, hi, below);
    }
    unsigned long long *d, h[8];
    (void)hipMalloc(&d, sizeof h); (void)hipMemset(d, 0, sizeof h);
    hipLaunchKernelGGL(check, dim3(1u << 18), dim3(1024), 0, 0, d);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("inputs 2^-64..2^64: %llu\n sqrt from rsq, one step: %llu wrong\n sqrt from rsq, two steps: %llu wrong\n 1/sqrt B1 (rsq as the reciprocal's start, 1 step): %llu wrong\n 1/sqrt B2 (2 steps): %llu wrong\n 1/sqrt B3 (one-step sqrt + rcp + 1 step): %llu wrong\n 1/sqrt B4 (two-step sqrt + rcp + 1 step): %llu wrong\n 1/sqrt as in the tree: %llu wrong\n",
           h[7], h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
    return 0;
}
