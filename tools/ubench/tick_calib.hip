// What is a "cycle" of __builtin_readcyclecounter() (s_memtime) on this GPU, and what does a SIMD really issue?  (development tool)
// Every CU runs one 1024-thread workgroup (four waves per SIMD) of 8 independent v_fma_f32 chains; the kernel is timed with HIP
// events (wall clock) and, inside, with s_memtime and wall_clock64() (the 100 MHz constant clock).  Prints ticks per second and
// wave-instructions per second per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/ubench/tick_calib.hip -o tools/ubench/tick_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAINS 8
template <int PACKED>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *info, float x, int iters)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    extern __shared__ float pad[];                       /* 100 KB of LDS per workgroup: exactly one workgroup per CU */
    if (x < 0) pad[threadIdx.x] = x;
    v2f a[CHAINS];
    for (int i = 0; i < CHAINS; i++) { a[i].x = x + threadIdx.x + i; a[i].y = x * 0.5f + i; }
    const v2f m = {1.0001f * x, 0.9999f * x};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#pragma unroll 8
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (PACKED) a[i] = __builtin_elementwise_fma(a[i], m, m);
            else a[i].x = __builtin_fmaf(a[i].x, m.x, m.y);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < CHAINS; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { info[0] = t1 - t0; info[1] = w1 - w0; }
}
int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 1 << 18;
    float *out; unsigned long long *info, h[2];
    (void)hipMalloc(&out, (size_t)cus * 1024 * 4); (void)hipMalloc(&info, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    (void)hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int packed = 0; packed < 2; packed++)
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0, 0);
            if (packed) hipLaunchKernelGGL(k<1>, dim3(cus), dim3(1024), 100 * 1024, 0, out, info, 1.0f, iters);
            else hipLaunchKernelGGL(k<0>, dim3(cus), dim3(1024), 100 * 1024, 0, out, info, 1.0f, iters);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(h, info, 16, hipMemcpyDeviceToHost);
            const double instr_per_wave = (double)iters * CHAINS, waves_per_simd = 4.0;
            if (rep)
                printf("%s: kernel %.3f ms by HIP events; in-kernel: %llu s_memtime ticks, %llu wall_clock64 ticks (100 MHz => %.3f ms) => s_memtime runs at %.1f MHz;\n"
                       "    %.3e wave-instructions per second per SIMD = %.3f per cycle at the reported %d MHz; %.1f TFLOP/s over %d CUs\n",
                       packed ? "v_pk_fma_f32" : "v_fma_f32   ", ms, h[0], h[1], h[1] / 1e5, h[0] / (h[1] / 1e8) / 1e6,
                       instr_per_wave * waves_per_simd / (ms / 1e3), instr_per_wave * waves_per_simd / (ms / 1e3) / (p.clockRate * 1e3), p.clockRate / 1000,
                       instr_per_wave * 16 * cus * 64 * (packed ? 4.0 : 2.0) / (ms / 1e3) / 1e12, cus);
        }
    return 0;
}
