// Does the shader clock depend on how busy the GPU is?  (development tool; round 4)
// A frame's longest jobs rendered alone (tools/tail_probe.py: 16 tiles on an otherwise empty GPU) take LONGER than the whole frame.
// Each wave here runs the same dependent v_fma_f32 chain - a fixed number of shader cycles whatever else runs - and reports the
// time it took on the constant 100 MHz clock (wall_clock64).  Launches of 1, 16, 256 and 1024 single-wave workgroups (at most one
// wave per SIMD) and of 256 x 1024 threads (four per SIMD, for reference: there the chains share issue slots).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/clock_probe.hip -o tools/ubench/clock_probe && tools/ubench/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(float *out, unsigned long long *ticks, float x, int iters)
{
    float a = x + threadIdx.x;
    const unsigned long long w0 = wall_clock64();
#pragma unroll 16
    for (int it = 0; it < iters; it++) a = __builtin_fmaf(a, x, 1.0f);
    const unsigned long long w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = w1 - w0;
}
int main()
{
    const int iters = 1 << 24;
    float *out; unsigned long long *ticks;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&ticks, 4096 * 8);
    const int shapes[5][2] = {{1, 64}, {16, 64}, {256, 64}, {1024, 64}, {256, 1024}};
    for (int rep = 0; rep < 2; rep++)
        for (auto &s : shapes) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(s[0]), dim3(s[1]), 0, 0, out, ticks, 0.999f, iters);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            const int waves = s[0] * s[1] / 64;
            std::vector<unsigned long long> h(waves);
            (void)hipMemcpy(h.data(), ticks, waves * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            printf("%5d workgroups x %4d threads: kernel %8.2f ms; a wave's chain of %d dependent v_fma_f32: min %.2f median %.2f max %.2f ms => %.2f / %.2f ns per instruction\n",
                   s[0], s[1], ms, iters, h[0] / 1e5, h[waves / 2] / 1e5, h[waves - 1] / 1e5, h[0] * 10.0 / iters, h[waves / 2] * 10.0 / iters);
            fflush(stdout);
        }
    return 0;
}
