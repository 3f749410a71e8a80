// VALU issue peak of one SIMD (development tool): W waves per SIMD, each a stream of independent
// (or dependent) FP32 multiplies-and-adds; reports wave-instructions per tick per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 8192
template <int DEP>
__global__ void k(float *out, unsigned long long *cyc, float x)
{
    float a = x + threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (DEP) {
#pragma unroll 64
        for (int i = 0; i < N; i++) a = a * 1.0001f + 0.5f;
    } else {
#pragma unroll 8
        for (int i = 0; i < N / 8; i++) { a = a * 1.0001f + 0.5f; b = b * 1.0001f + 0.5f; c = c * 1.0001f + 0.5f; d = d * 1.0001f + 0.5f;
                                          e = e * 1.0001f + 0.5f; f = f * 1.0001f + 0.5f; g = g * 1.0001f + 0.5f; h = h * 1.0001f + 0.5f; }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a + b + c + d + e + f + g + h;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 8192); hipMalloc(&cyc, 8);
    for (int dep = 0; dep < 2; dep++)
        for (int waves = 1; waves <= 16; waves *= 2) {      // waves per CU; 4 SIMDs -> waves/4 per SIMD (>= 1)
            for (int r = 0; r < 2; r++) {
                if (dep) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
                else hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
                hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            }
            const double per_simd = (waves < 4 ? 1.0 : waves / 4.0);
            printf("%s, %2d waves in the CU (%.0f per SIMD): %.3f ticks per instruction per wave, %.3f wave-instructions per tick per SIMD\n",
                   dep ? "dependent" : "independent", waves, per_simd, (double)h / (2.0 * N), per_simd * 2.0 * N / (double)h);
        }
    return 0;
}
