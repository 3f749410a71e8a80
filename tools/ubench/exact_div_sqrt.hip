// Which inputs does a short correctly-rounded reciprocal / square root get right?  (development tool, round 4)
// All 2^32 binary32 patterns: rt_fast_rcp(x) against 1.0f / x and rt_fast_sqrt(x) against sqrtf(x) as the compiler expands them
// (IEEE, denormals on), bit for bit (any NaN equals any NaN), mismatches counted by class of the input.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/exact_div_sqrt.hip -o tools/ubench/exact_div_sqrt && tools/ubench/exact_div_sqrt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ float fast_rcp(float x)
{
    float y = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, y, 1.0f);
    y = __builtin_fmaf(y, e, y);
    e = __builtin_fmaf(-x, y, 1.0f);
    return __builtin_fmaf(y, e, y);
}
__device__ __forceinline__ float fast_rcp1(float x)        // one refinement + one residual correction on the quotient
{
    float y = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, y, 1.0f);
    return __builtin_fmaf(y, e, y);
}
__device__ __forceinline__ float fast_sqrt(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float rdn = __builtin_fmaf(-dn, s, x), rup = __builtin_fmaf(-up, s, x);
    s = rdn <= 0.0f ? dn : s;
    s = rup > 0.0f ? up : s;
    return s;
}
// classes: 0 zero, 1 denormal, 2 normal with |x| < 2^-64, 3 normal 2^-64 <= |x| <= 2^64, 4 normal |x| in (2^64, 2^126], 5 normal |x| > 2^126, 6 inf, 7 nan; +8 for negative
__device__ __forceinline__ int cls(uint32_t u)
{
    const uint32_t a = u & 0x7fffffffu;
    int c;
    if (a == 0) c = 0; else if (a < 0x00800000u) c = 1; else if (a > 0x7f800000u) c = 7; else if (a == 0x7f800000u) c = 6;
    else if (a < 0x1f800000u) c = 2; else if (a <= 0x5f800000u) c = 3; else if (a <= 0x7e800000u) c = 4; else c = 5;
    return c + ((u >> 31) ? 8 : 0);
}
__global__ void check(unsigned long long *bad)      // bad[3][16]
{
    const uint32_t u = blockIdx.x * 1024u + threadIdx.x;
    for (uint32_t hi = 0; hi < 16; hi++) {
        const uint32_t b = u | (hi << 28);
        const float x = __uint_as_float(b);
        const float r0 = 1.0f / x, r1 = fast_rcp(x), r2 = fast_rcp1(x), q0 = sqrtf(x), q1 = fast_sqrt(x);
        auto same = [](float a, float c) { return (a != a && c != c) || __float_as_uint(a) == __float_as_uint(c); };
        const int c = cls(b);
        if (!same(r0, r1)) atomicAdd(&bad[c], 1ull);
        if (!same(r0, r2)) atomicAdd(&bad[16 + c], 1ull);
        if (!same(q0, q1)) atomicAdd(&bad[32 + c], 1ull);
    }
}
int main()
{
    unsigned long long *d, h[48];
    (void)hipMalloc(&d, sizeof h); (void)hipMemset(d, 0, sizeof h);
    hipLaunchKernelGGL(check, dim3(1u << 18), dim3(1024), 0, 0, d);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[8] = {"zero", "denormal", "normal < 2^-64", "normal 2^-64..2^64", "normal 2^64..2^126", "normal > 2^126", "inf", "nan"};
    const char *what[3] = {"rcp + 2 x (residual, correction)", "rcp + 1 x (residual, correction)", "sqrt + (-1 ulp, +1 ulp) residual test"};
    for (int k = 0; k < 3; k++) {
        printf("%s: mismatches against the compiler's IEEE expansion, by class of the input\n", what[k]);
        for (int s = 0; s < 2; s++) for (int c = 0; c < 8; c++) printf("   %s%-20s %llu\n", s ? "-" : "+", names[c], h[16 * k + 8 * s + c]);
    }
    return 0;
}
