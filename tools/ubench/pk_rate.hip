// Issue cost of packed-f32 VALU instructions against plain ones on one CU (development tool): W waves per SIMD, each a
// stream of 8 independent chains of ONE instruction kind; prints shader cycles per wave-instruction per wave and
// wave-instructions per cycle per SIMD.  A v_pk_* instruction does two f32 operations per lane.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/ubench/pk_rate.hip -o tools/ubench/pk_rate && tools/ubench/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define N 4096
#define OP_MUL 0
#define OP_PK_MUL 1
#define OP_PK_ADD 2
#define OP_PK_FMA 3
#define OP_MIN 4
#define OP_MAX3 5
#define OP_FMA 6
#define OP_MUL_F64 7
template <int OP, int CHAINS>
__global__ void k(float *out, unsigned long long *cyc, float x)
{
    v2f a[CHAINS];
    double d[CHAINS];
    for (int i = 0; i < CHAINS; i++) { a[i].x = x + threadIdx.x + i; a[i].y = x * 0.5f + i; d[i] = a[i].x; }
    const v2f m = {1.0001f * x, 0.9999f * x};          /* run-time values: nothing folds */
    const double md = 1.0001 * x;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    /* (unrolled far enough that the loop's own scalar instructions and branch are < 2 % of the stream: a first version
     * unrolled the one-chain loop by 4 and measured 12 "cycles per dependent instruction", 7 of them loop overhead) */
#pragma unroll 64
    for (int it = 0; it < N / CHAINS; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            /* plain C, not inline asm: the compiler pads every asm statement with an s_nop (hazard recognizer), which a
             * first version of this benchmark counted as instruction latency */
            if (OP == OP_MUL) a[i].x = a[i].x * m.x;
            if (OP == OP_FMA) a[i].x = __builtin_fmaf(a[i].x, m.x, m.x);
            if (OP == OP_MAX3) a[i].x = __builtin_fmaxf(__builtin_fmaxf(a[i].x, m.x), m.y);
            if (OP == OP_PK_MUL) a[i] = a[i] * m;
            if (OP == OP_PK_ADD) a[i] = a[i] + m;
            if (OP == OP_PK_FMA) a[i] = __builtin_elementwise_fma(a[i], m, m);
            if (OP == OP_MUL_F64) d[i] = d[i] * md;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < CHAINS; i++) s += a[i].x + a[i].y + (float)d[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP, int CHAINS>
static void run1(const char *name, float *out, unsigned long long *cyc)
{
    for (int waves = 4; waves <= 16; waves *= 2) {      // waves in the one workgroup = on the one CU; 4 SIMDs
        unsigned long long h = 0;
        for (int r = 0; r < 2; r++) {
            hipLaunchKernelGGL((k<OP, CHAINS>), dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        }
        const double per_simd = waves / 4.0;
        printf("%-12s %s, %d waves/SIMD: %.2f cycles per instruction per wave, %.3f wave-instructions per cycle per SIMD\n", name,
               CHAINS == 1 ? "ONE dependent chain " : "8 independent chains", (int)per_simd, (double)h / N, per_simd * N / (double)h);
    }
}
template <int OP>
static void run(const char *name, float *out, unsigned long long *cyc)
{
    run1<OP, 8>(name, out, cyc);
    run1<OP, 1>(name, out, cyc);
}
int main()
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 8192); (void)hipMalloc(&cyc, 8);
    run<OP_MUL>("v_mul_f32", out, cyc);
    run<OP_FMA>("v_fma_f32", out, cyc);
    run<OP_MAX3>("v_max3_f32", out, cyc);
    run<OP_PK_MUL>("v_pk_mul_f32", out, cyc);
    run<OP_PK_ADD>("v_pk_add_f32", out, cyc);
    run<OP_PK_FMA>("v_pk_fma_f32", out, cyc);
    run<OP_MUL_F64>("v_mul_f64", out, cyc);
    return 0;
}
