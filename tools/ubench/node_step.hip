// Where do the cycles of one BVH node step go when a wave has the SIMD to itself?  (development
// tool)  The loop below is the descend loop of rt_kernel.hip on a synthetic complete binary tree
// in LDS; variants drop one ingredient each.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
#define INF_F 1073741824.0f
__device__ __forceinline__ bool box_test(float bx0, float by0, float bz0, float bx1, float by1, float bz1, float ox, float oy, float oz, float ix, float iy, float iz, float &tmin_out)
{
    float tmin = 0.0f, tmax = INF_F;
    float t1 = (bx0 - ox) * ix, t2 = (bx1 - ox) * ix;
    tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
    t1 = (by0 - oy) * iy; t2 = (by1 - oy) * iy;
    tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
    t1 = (bz0 - oz) * iz; t2 = (bz1 - oz) * iz;
    tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
    tmin_out = tmin;
    return tmin < tmax && tmax > 0.0f;
}
// MODE bits: 1 = no stack write, 2 = no box arithmetic (next from data only), 4 = fixed trip (no ballot exit),
//            8 = simplified predicates (push = tmin < min(tmax, best)), 16 = phase-ordered box arithmetic
template <int MODE>
__global__ void k_step(float *out, unsigned long long *res, int iters, int active)
{
    __shared__ v4f nodes[4 * 511];
    __shared__ uint2 stack[12 * 64];
    for (int i = threadIdx.x; i < 511; i += 64) {
        // boxes around the unit cube, children 2i+1 / 2i+2 (wrapping), so every ray enters both
        v4f q0 = {-1.f - 0.001f * i, -1.f, -1.f, 1.f}, q1 = {1.f, 1.f, -1.5f, -1.f}, q2 = {-1.f, 1.5f + 0.001f * i, 1.f, 1.f};
        v4f q3; q3.x = __uint_as_float((unsigned)((2 * i + 1) % 511)); q3.y = __uint_as_float((unsigned)((2 * i + 2 + threadIdx.x) % 511)); q3.z = 0; q3.w = 0;
        nodes[4 * i] = q0; nodes[4 * i + 1] = q1; nodes[4 * i + 2] = q2; nodes[4 * i + 3] = q3;
    }
    __syncthreads();
    const int tid = threadIdx.x;
    float ox = 0.01f * tid, oy = 0.02f, oz = -0.03f, ix = 1.0f / (0.3f + 0.001f * tid), iy = 1.0f / 0.4f, iz = 1.0f / 0.86f;
    float best = INF_F;
    unsigned cur = tid % 511;
    int sp = 0;
    unsigned long long steps = 0, t0 = 0, t1 = 0;
    if (tid < active) {
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; it++) {
            if (MODE & 32) {
                for (;;) {
                    if (!(cur & 0x80000000u)) {
                    steps++;
                    const v4f *n = nodes + 4 * (int)(cur & 0x3fffffffu);
                    v4f q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
                    float ld = 0.f, rd = 1.f;
                    bool l_push = true, r_push = true;
                    if (!(MODE & 2)) {
                        float atmax, btmax;
                        if (MODE & 16) {
                            float a0 = q0.x - ox, a1 = q0.w - ox, a2 = q0.y - oy, a3 = q1.x - oy, a4 = q0.z - oz, a5 = q1.y - oz;
                            float b0 = q1.z - ox, b1 = q2.y - ox, b2 = q1.w - oy, b3 = q2.z - oy, b4 = q2.x - oz, b5 = q2.w - oz;
                            __builtin_amdgcn_sched_barrier(0);
                            a0 *= ix; a1 *= ix; a2 *= iy; a3 *= iy; a4 *= iz; a5 *= iz; b0 *= ix; b1 *= ix; b2 *= iy; b3 *= iy; b4 *= iz; b5 *= iz;
                            __builtin_amdgcn_sched_barrier(0);
                            const float an0 = fminf(a0, a1), af0 = fmaxf(a0, a1), an1 = fminf(a2, a3), af1 = fmaxf(a2, a3), an2 = fminf(a4, a5), af2 = fmaxf(a4, a5);
                            const float bn0 = fminf(b0, b1), bf0 = fmaxf(b0, b1), bn1 = fminf(b2, b3), bf1 = fmaxf(b2, b3), bn2 = fminf(b4, b5), bf2 = fmaxf(b4, b5);
                            __builtin_amdgcn_sched_barrier(0);
                            ld = fmaxf(fmaxf(fmaxf(0.0f, an0), an1), an2); atmax = fminf(fminf(fminf(INF_F, af0), af1), af2);
                            rd = fmaxf(fmaxf(fmaxf(0.0f, bn0), bn1), bn2); btmax = fminf(fminf(fminf(INF_F, bf0), bf1), bf2);
                        } else {
                            // same as box_test, keeping tmax
                            float tmin = 0.0f, tmax = INF_F;
                            float t1 = (q0.x - ox) * ix, t2 = (q0.w - ox) * ix;
                            tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
                            t1 = (q0.y - oy) * iy; t2 = (q1.x - oy) * iy;
                            tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
                            t1 = (q0.z - oz) * iz; t2 = (q1.y - oz) * iz;
                            tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
                            ld = tmin; atmax = tmax;
                            tmin = 0.0f; tmax = INF_F;
                            t1 = (q1.z - ox) * ix; t2 = (q2.y - ox) * ix;
                            tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
                            t1 = (q1.w - oy) * iy; t2 = (q2.z - oy) * iy;
                            tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
                            t1 = (q2.x - oz) * iz; t2 = (q2.w - oz) * iz;
                            tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
                            rd = tmin; btmax = tmax;
                        }
                        if (MODE & 8) {
                            l_push = ld < fminf(atmax, best);
                            r_push = rd < fminf(btmax, best);
                        } else {
                            const bool lh = ld < atmax && atmax > 0.0f, rh = rd < btmax && btmax > 0.0f;
                            l_push = lh && ld < best; r_push = rh && rd < best;
                        }
                    }
                    const unsigned lref = __float_as_uint(q3.x), rref = __float_as_uint(q3.y);
                    const bool l_first = ld < rd;
                    const bool both = l_push && r_push, entered = l_push || r_push;
                    const unsigned dref = l_first ? lref : rref;
                    const float dd = l_first ? ld : rd;
                    if (!(MODE & 1)) stack[sp * 64 + tid] = make_uint2(__float_as_uint(dd), dref);
                    sp = (sp + (both ? 1 : 0)) & 7;
                    const unsigned next = both ? (l_first ? rref : lref) : (l_push ? lref : rref);
                    cur = entered ? next : 0x80000000u;

                    }
                    const int n_still = __popcll(__ballot(!(cur & 0x80000000u)));
                    if (n_still < 8 || (steps & 3) == 0) break;
                }
            } else {
            for (;;) {
                steps++;
                const v4f *n = nodes + 4 * (int)(cur & 0x3fffffffu);
                v4f q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
                float ld = 0.f, rd = 1.f;
                bool l_push = true, r_push = true;
                if (!(MODE & 2)) {
                    float atmax, btmax;
                    if (MODE & 16) {
                        float a0 = q0.x - ox, a1 = q0.w - ox, a2 = q0.y - oy, a3 = q1.x - oy, a4 = q0.z - oz, a5 = q1.y - oz;
                        float b0 = q1.z - ox, b1 = q2.y - ox, b2 = q1.w - oy, b3 = q2.z - oy, b4 = q2.x - oz, b5 = q2.w - oz;
                        __builtin_amdgcn_sched_barrier(0);
                        a0 *= ix; a1 *= ix; a2 *= iy; a3 *= iy; a4 *= iz; a5 *= iz; b0 *= ix; b1 *= ix; b2 *= iy; b3 *= iy; b4 *= iz; b5 *= iz;
                        __builtin_amdgcn_sched_barrier(0);
                        const float an0 = fminf(a0, a1), af0 = fmaxf(a0, a1), an1 = fminf(a2, a3), af1 = fmaxf(a2, a3), an2 = fminf(a4, a5), af2 = fmaxf(a4, a5);
                        const float bn0 = fminf(b0, b1), bf0 = fmaxf(b0, b1), bn1 = fminf(b2, b3), bf1 = fmaxf(b2, b3), bn2 = fminf(b4, b5), bf2 = fmaxf(b4, b5);
                        __builtin_amdgcn_sched_barrier(0);
                        ld = fmaxf(fmaxf(fmaxf(0.0f, an0), an1), an2); atmax = fminf(fminf(fminf(INF_F, af0), af1), af2);
                        rd = fmaxf(fmaxf(fmaxf(0.0f, bn0), bn1), bn2); btmax = fminf(fminf(fminf(INF_F, bf0), bf1), bf2);
                    } else {
                        // same as box_test, keeping tmax
                        float tmin = 0.0f, tmax = INF_F;
                        float t1 = (q0.x - ox) * ix, t2 = (q0.w - ox) * ix;
                        tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
                        t1 = (q0.y - oy) * iy; t2 = (q1.x - oy) * iy;
                        tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
                        t1 = (q0.z - oz) * iz; t2 = (q1.y - oz) * iz;
                        tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
                        ld = tmin; atmax = tmax;
                        tmin = 0.0f; tmax = INF_F;
                        t1 = (q1.z - ox) * ix; t2 = (q2.y - ox) * ix;
                        tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
                        t1 = (q1.w - oy) * iy; t2 = (q2.z - oy) * iy;
                        tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
                        t1 = (q2.x - oz) * iz; t2 = (q2.w - oz) * iz;
                        tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
                        rd = tmin; btmax = tmax;
                    }
                    if (MODE & 8) {
                        l_push = ld < fminf(atmax, best);
                        r_push = rd < fminf(btmax, best);
                    } else {
                        const bool lh = ld < atmax && atmax > 0.0f, rh = rd < btmax && btmax > 0.0f;
                        l_push = lh && ld < best; r_push = rh && rd < best;
                    }
                }
                const unsigned lref = __float_as_uint(q3.x), rref = __float_as_uint(q3.y);
                const bool l_first = ld < rd;
                const bool both = l_push && r_push, entered = l_push || r_push;
                const unsigned dref = l_first ? lref : rref;
                const float dd = l_first ? ld : rd;
                if (!(MODE & 1)) stack[sp * 64 + tid] = make_uint2(__float_as_uint(dd), dref);
                sp = (sp + (both ? 1 : 0)) & 7;
                const unsigned next = both ? (l_first ? rref : lref) : (l_push ? lref : rref);
                cur = entered ? next : 0x80000000u;
                if (cur & 0x80000000u) break;
                if (MODE & 4) { if ((steps & 3) == 0) break; }
                else if (MODE & 64) { if ((steps & 1) == 0 && (__popcll(__ballot(1)) < 8 || (steps & 3) == 0)) break; }
                else if (__popcll(__ballot(1)) < 8 || (steps & 3) == 0) break;     // the ballot of the real loop; a bounded trip count
            }
            }
            cur &= 0x1ff;
        }
        t1 = __builtin_readcyclecounter();
    }
    out[tid] = best + cur + sp;
    unsigned long long total = steps;
    if (tid == 0) { res[0] = t1 - t0; res[1] = total; }
}
int main()
{
    float *out; unsigned long long *res, h[2];
    hipMalloc(&out, 4096); hipMalloc(&res, 16);
#define RUN(M, act) for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k_step<M>, dim3(1), dim3(64), 0, 0, out, res, 20000, act); hipDeviceSynchronize(); hipMemcpy(h, res, 16, hipMemcpyDeviceToHost); if (r) printf("mode %d active %2d: %7.1f cycles per node step of lane 0 (%llu steps)\n", M, act, (double)h[0] / (double)h[1], h[1]); }
    RUN(0, 64) RUN(8, 64) RUN(12, 64) RUN(40, 64) RUN(72, 64) RUN(32, 64)
    return 0;
}
