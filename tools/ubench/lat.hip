// Single-wave latency microbenchmarks for gfx950 (development tool): cycles per instruction of
// dependent / independent VALU chains, SALU chains, LDS b128 reads (conflict-free, node-like
// random, broadcast).  hipcc --offload-arch=gfx950 -O3 lat.hip -o lat && ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define N 4096
__global__ void k_dep(float *out, unsigned long long *cyc, float x)
{
    float a = x + threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 64
    for (int i = 0; i < N; i++) a = a * 1.0001f + 0.5f;       // mul+add dependent (contract may fuse -> one fma)
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_ind(float *out, unsigned long long *cyc, float x)
{
    float a = x + threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N / 4; i++) { a = a * 1.0001f + 0.5f; b = b * 1.0001f + 0.5f; c = c * 1.0001f + 0.5f; d = d * 1.0001f + 0.5f; }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a + b + c + d; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_minmax(float *out, unsigned long long *cyc, float x)
{
    float a = x + threadIdx.x, b = x * 2;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 64
    for (int i = 0; i < N; i++) { a = fminf(a, b) ; b = fmaxf(b, a + 1.0f); }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a + b; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// LDS pointer chase: each lane follows idx = lds[idx].w ; mode decides the address pattern
__global__ void k_lds(float *out, unsigned long long *cyc, int mode, int active)
{
    __shared__ v4f lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) {
        unsigned nxt = mode == 0 ? (unsigned)((i + 64) & 2047)                    // lanes stay consecutive: conflict-free
                     : mode == 1 ? (unsigned)(((i * 1103515245u + 12345u) >> 7) & 2047 & ~3)   // random 64-byte aligned "nodes"
                                 : 0u;                                            // broadcast
        v4f v; v.x = i; v.y = 1; v.z = 2; v.w = __uint_as_float(nxt); lds[i] = v;
    }
    __syncthreads();
    unsigned idx = threadIdx.x; float acc = 0;
    if ((int)threadIdx.x < active) {
        unsigned long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < 1024; i++) { v4f v = lds[idx]; acc += v.x; idx = __float_as_uint(v.w); }
        unsigned long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) cyc[0] = t1 - t0;
    }
    out[threadIdx.x] = acc + idx;
}
// 4 b128 reads per step like a BVH node (64 B), next node from the 4th quad
__global__ void k_node(float *out, unsigned long long *cyc, int active)
{
    __shared__ v4f lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) {
        unsigned nxt = (unsigned)(((i * 1103515245u + 12345u) >> 7) & 2047 & ~3);
        v4f v; v.x = i; v.y = 1; v.z = 2; v.w = __uint_as_float(nxt); lds[i] = v;
    }
    __syncthreads();
    unsigned idx = (threadIdx.x * 4) & 2047; float acc = 0;
    if ((int)threadIdx.x < active) {
        unsigned long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < 1024; i++) { v4f a = lds[idx], b = lds[idx + 1], c = lds[idx + 2], d = lds[idx + 3]; acc += a.x + b.y + c.z; idx = __float_as_uint(d.w) & ~3u; }
        unsigned long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) cyc[0] = t1 - t0;
    }
    out[threadIdx.x] = acc + idx;
}
// the same 64 bytes fetched as N reads of W bytes: is the cost per instruction or per byte?
template <int W>
__global__ void k_node_w(float *out, unsigned long long *cyc, int active)
{
    __shared__ v4f lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) {
        unsigned nxt = (unsigned)(((i * 1103515245u + 12345u) >> 7) & 2047 & ~3);
        v4f v; v.x = i; v.y = 1; v.z = 2; v.w = __uint_as_float(nxt); lds[i] = v;
    }
    __syncthreads();
    unsigned idx = (threadIdx.x * 4) & 2047; float acc = 0;
    if ((int)threadIdx.x < active) {
        unsigned long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < 1024; i++) {
            const volatile float *f = (const volatile float *)&lds[idx];
            float last = 0;
            if (W == 4) { for (int k = 0; k < 16; k++) { float v = f[k]; acc += v; last = v; } }
            else if (W == 8) { const v2f *g = (const v2f *)&lds[idx]; v2f v0 = g[0], v1 = g[1], v2 = g[2], v3 = g[3], v4 = g[4], v5 = g[5], v6 = g[6], v7 = g[7]; acc += v0.x + v1.x + v2.x + v3.x + v4.x + v5.x + v6.x; last = v7.y; }
            else if (W == 48) { v4f a = lds[idx], b = lds[idx + 1], c = lds[idx + 2]; v2f d = ((const v2f *)&lds[idx + 3])[1]; acc += a.x + b.y + c.z; last = d.y; }
            else if (W == 32) { v4f a = lds[idx], d = lds[idx + 3]; acc += a.x; last = d.w; }      // only 2 of the 4 quads
            idx = __float_as_uint(last) & ~3u;
        }
        unsigned long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) cyc[0] = t1 - t0;
    }
    out[threadIdx.x] = acc + idx;
}
__global__ void k_salu(float *out, unsigned long long *cyc, int x)
{
    int a = x;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 64
    for (int i = 0; i < N; i++) a = (a ^ 0x55) + 3;
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
#define RUN(name, ops, ...) for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(name, dim3(1), dim3(64), 0, 0, __VA_ARGS__); hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); if (r) printf("%-44s %8.2f cycles per op\n", #name " " #__VA_ARGS__, (double)h / (ops)); }
    RUN(k_dep, 2.0 * N, out, cyc, 1.0f)
    RUN(k_ind, 2.0 * N, out, cyc, 1.0f)
    RUN(k_minmax, 3.0 * N, out, cyc, 1.0f)
    RUN(k_salu, 2.0 * N, out, cyc, 3)
    RUN(k_lds, 1024.0, out, cyc, 0, 64)
    RUN(k_lds, 1024.0, out, cyc, 1, 64)
    RUN(k_lds, 1024.0, out, cyc, 1, 4)
    RUN(k_lds, 1024.0, out, cyc, 2, 64)
    RUN(k_node, 1024.0, out, cyc, 64)
    RUN(k_node, 1024.0, out, cyc, 16)
    RUN(k_node, 1024.0, out, cyc, 4)
    RUN(k_node_w<4>, 1024.0, out, cyc, 64)
    RUN(k_node_w<8>, 1024.0, out, cyc, 64)
    RUN(k_node_w<48>, 1024.0, out, cyc, 64)
    RUN(k_node_w<32>, 1024.0, out, cyc, 64)
    return 0;
}
