// One ray per WAVE against one ray per LANE: cycles per BVH node step when a wave has a single ray left (the tail of a frame:
// a pixel's samples are one sequential stream, so the frame is as long as its most expensive pixel).  Development tool,
// round 4.
//
//   per-lane  : rt_kernel.hip's node step (two six-med3 slab tests, child selection, branch-free stack write), ONE lane active
//   per-wave  : the same step with the twelve slab distances of the node's two boxes on twelve lanes - lane l owns plane
//               (box l / 6, side (l % 6) / 3, axis l % 3): one subtract and one multiply in all, the near / far pair of an axis by
//               DPP row shifts, the three axes folded by two more, the decisions read from two compare masks in scalar code,
//               cur / sp / best in SGPRs.  NaN-dropping max / min over a set are order-independent and select one of their
//               operands, so the entry distances and decisions are the per-lane step's bit for bit (checked here: both
//               kernels must visit the same nodes).
// Both run the same walk: a 511-node tree in LDS with pseudo-random child boxes, restart at the root with a nudged ray
// whenever a leaf level or a miss is reached.  Times: s_memtime over `steps` node steps, for a wave alone on its SIMD and
// with 15 other waves doing the same (16 waves per CU = 4 per SIMD, the render kernel's occupancy).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/coop_step.hip -o tools/ubench/coop_step && tools/ubench/coop_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define NNODES 511
#define LEAF 0x80000000u
#define INF_F 1073741824.0f

__device__ __forceinline__ bool box_enter_med3(float bx0, float by0, float bz0, float bx1, float by1, float bz1, float ox, float oy, float oz,
                                               float ix, float iy, float iz, float best, float &tmin_out)
{
    const float x0 = (bx0 - ox) * ix, x1 = (bx1 - ox) * ix;
    const float y0 = (by0 - oy) * iy, y1 = (by1 - oy) * iy;
    const float z0 = (bz0 - oz) * iz, z1 = (bz1 - oz) * iz;
    const float lo = __builtin_amdgcn_fmed3f(z0, z1, __builtin_amdgcn_fmed3f(y0, y1, __builtin_amdgcn_fmed3f(x0, x1, 0.0f)));
    const float hi = __builtin_amdgcn_fmed3f(z0, z1, __builtin_amdgcn_fmed3f(y0, y1, __builtin_amdgcn_fmed3f(x0, x1, best)));
    tmin_out = lo;
    return lo < hi;
}

__device__ void build_tree(v4f *nodes, int tid, int nt)
{
    // node i: children 2i+1 / 2i+2 (indices >= 255 are "leaves"); boxes: slabs around the z axis that the rays below mostly enter,
    // with pseudo-random offsets so that every outcome (none / left / right / both, either order) occurs
    for (int i = tid; i < NNODES; i += nt) {
        unsigned h = (unsigned)i * 2654435761u;
        auto rnd = [&]() { h = h * 1664525u + 1013904223u; return (float)(h >> 8) * (1.0f / 16777216.0f); };
        float lx = -1.5f + rnd(), ly = -1.5f + rnd(), lz = 1.0f + 3.0f * rnd();
        float rx = -1.5f + rnd(), ry = -1.5f + rnd(), rz = 1.0f + 3.0f * rnd();
        v4f q0 = {lx, ly, lz, lx + 1.0f + rnd()}, q1 = {ly + 1.0f + rnd(), lz + 0.5f + rnd(), rx, ry}, q2 = {rz, rx + 1.0f + rnd(), ry + 1.0f + rnd(), rz + 0.5f + rnd()};
        const unsigned l = 2 * i + 1, r = 2 * i + 2;
        v4f q3;
        q3.x = __uint_as_float(l >= 255 ? (LEAF | l) : l); q3.y = __uint_as_float(r >= 255 ? (LEAF | r) : r); q3.z = 0; q3.w = 0;
        nodes[4 * i] = q0; nodes[4 * i + 1] = q1; nodes[4 * i + 2] = q2; nodes[4 * i + 3] = q3;
    }
}

// ---------------------------------------------------------------- one ray per lane (lane 0 of every wave active)
__global__ __launch_bounds__(1024) void k_lane(unsigned long long *res, unsigned *sum, int steps)
{
    __shared__ v4f nodes[4 * NNODES];
    __shared__ uint2 stack[12 * 1024];
    build_tree(nodes, threadIdx.x, blockDim.x);
    __syncthreads();
    const int tid = threadIdx.x, NT = blockDim.x;
    if ((tid & 63) != 0) return;
    float ox = 0.05f + 0.001f * (tid >> 6), oy = -0.03f, oz = -4.0f;
    float dx = 0.02f, dy = 0.03f, dz = 0.999f;
    float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    float best = INF_F;
    unsigned cur = 0, check = 0;
    int sp = 0, n = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (n < steps) {
        for (;;) {
            n++;
            const v4f *nd = nodes + 4 * (int)(cur & 0x3fffffffu);
            const v4f q0 = nd[0], q1 = nd[1], q2 = nd[2];
            const uint2 refs = *(const uint2 *)(nd + 3);
            float ld, rd;
            const bool l_push = box_enter_med3(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, ox, oy, oz, ix, iy, iz, best, ld);
            const bool r_push = box_enter_med3(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, ox, oy, oz, ix, iy, iz, best, rd);
            const bool l_first = ld < rd, both = l_push && r_push, entered = l_push || r_push;
            stack[(sp & 7) * NT + tid] = make_uint2(__float_as_uint(l_first ? ld : rd), l_first ? refs.x : refs.y);
            sp += both ? 1 : 0;
            const unsigned next = both ? (l_first ? refs.y : refs.x) : (l_push ? refs.x : refs.y);
            cur = entered ? next : LEAF;
            check = check * 31u + cur;
            if ((int)cur < 0 || n >= steps) break;
        }
        // restart: nudge the ray (deterministically) and begin at the root again
        ox += 0.013f; if (ox > 0.6f) ox -= 1.1f;
        oy += 0.007f; if (oy > 0.5f) oy -= 0.9f;
        cur = 0; sp = 0;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    res[tid >> 6] = t1 - t0;
    sum[tid >> 6] = check;
}

// ---------------------------------------------------------------- one ray per wave
__device__ __forceinline__ float dpp_shl(float v, int n)      // lane i <- lane i + n of its row of 16 (own value where there is none)
{
    switch (n) {
        case 1: return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(v), __float_as_uint(v), 0x101, 0xf, 0xf, false));
        case 3: return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(v), __float_as_uint(v), 0x103, 0xf, 0xf, false));
        case 6: return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(v), __float_as_uint(v), 0x106, 0xf, 0xf, false));
        default: return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(v), __float_as_uint(v), 0x102, 0xf, 0xf, false));
    }
}

__global__ __launch_bounds__(1024) void k_wave(unsigned long long *res, unsigned *sum, int steps)
{
    __shared__ v4f nodes[4 * NNODES];
    __shared__ uint2 stack[12 * 16];
    build_tree(nodes, threadIdx.x, blockDim.x);
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // lane l < 12 owns plane (box = l / 6, side = (l % 6) / 3, axis = l % 3): node float index box * 6 + side * 3 + axis = l.
    // (lanes >= 12 compute on plane 0's data and are ignored)
    const int pl = lane < 12 ? lane : 0, axis = pl % 3;
    float ox = 0.05f + 0.001f * wave, oy = -0.03f, oz = -4.0f;
    const float dx = 0.02f, dy = 0.03f, dz = 0.999f;
    const float inv_a = axis == 0 ? 1.0f / dx : (axis == 1 ? 1.0f / dy : 1.0f / dz);
    float o_a = axis == 0 ? ox : (axis == 1 ? oy : oz);
    const float best = INF_F;
    const float *nf = (const float *)nodes;
    unsigned cur = 0, check = 0;
    int sp = 0, n = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (n < steps) {
        for (;;) {
            n++;
            const int base = 16 * (int)(cur & 0x3fffffffu);                    // wave-uniform (floats)
            const float b = nf[base + pl];                                      // ds_read_b32, one plane per lane
            const uint2 refs = *(const uint2 *)(nf + base + 12);               // broadcast
            const float t = (b - o_a) * inv_a;
            // the slab of this lane's axis: its own plane and the one three lanes up (side 1) - valid in lanes with side 0
            const float tp = dpp_shl(t, 3);
            float nr = fminf(t, tp), fr = fmaxf(t, tp);
            // fold the three axes (lanes +1, +2 of the box's first lane)
            nr = fmaxf(fmaxf(nr, dpp_shl(nr, 1)), dpp_shl(nr, 2));
            fr = fminf(fminf(fr, dpp_shl(fr, 1)), dpp_shl(fr, 2));
            const float tmin = fmaxf(nr, 0.0f), tmax = fminf(fr, best);        // lanes 0 (left box) and 6 (right box)
            const unsigned long long push = __ballot(tmin < tmax);
            const float rd_at0 = dpp_shl(tmin, 6);
            const unsigned long long first = __ballot(tmin < rd_at0);          // bit 0: ld < rd
            const bool l_push = push & 1ull, r_push = (push >> 6) & 1ull, l_first = first & 1ull;
            const bool both = l_push && r_push, entered = l_push || r_push;
            const unsigned lref = __builtin_amdgcn_readfirstlane(refs.x), rref = __builtin_amdgcn_readfirstlane(refs.y);
            // the deferred sibling's entry: written by the lane that holds its distance
            if (lane == (l_first ? 0 : 6)) stack[(sp & 7) * 16 + wave] = make_uint2(__float_as_uint(tmin), l_first ? lref : rref);
            sp += both ? 1 : 0;
            const unsigned next = both ? (l_first ? rref : lref) : (l_push ? lref : rref);
            cur = entered ? next : LEAF;
            check = check * 31u + cur;
            if ((int)cur < 0 || n >= steps) break;
        }
        ox += 0.013f; if (ox > 0.6f) ox -= 1.1f;
        oy += 0.007f; if (oy > 0.5f) oy -= 0.9f;
        o_a = axis == 0 ? ox : (axis == 1 ? oy : oz);
        cur = 0; sp = 0;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { res[wave] = t1 - t0; sum[wave] = check; }
}

int main()
{
    unsigned long long *res; unsigned *sum;
    (void)hipMalloc(&res, 16 * 8); (void)hipMalloc(&sum, 16 * 4);
    const int steps = 20000;
    for (int waves = 1; waves <= 16; waves *= 4) {
        unsigned long long hl[16], hw[16]; unsigned cl[16], cw[16];
        for (int r = 0; r < 2; r++) {
            hipLaunchKernelGGL(k_lane, dim3(1), dim3(64 * waves), 0, 0, res, sum, steps);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(hl, res, 8 * waves, hipMemcpyDeviceToHost); (void)hipMemcpy(cl, sum, 4 * waves, hipMemcpyDeviceToHost);
            hipLaunchKernelGGL(k_wave, dim3(1), dim3(64 * waves), 0, 0, res, sum, steps);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(hw, res, 8 * waves, hipMemcpyDeviceToHost); (void)hipMemcpy(cw, sum, 4 * waves, hipMemcpyDeviceToHost);
        }
        unsigned long long ml = 0, mw = 0; bool same = true;
        for (int w = 0; w < waves; w++) { ml = hl[w] > ml ? hl[w] : ml; mw = hw[w] > mw ? hw[w] : mw; same = same && cl[w] == cw[w]; }
        printf("%2d wave(s) on the CU: per-lane step %7.1f cycles, per-wave step %7.1f cycles  (%.2fx)   same node sequence: %s\n", waves,
               (double)ml / steps, (double)mw / steps, (double)ml / (double)mw, same ? "yes" : "NO");
    }
    return 0;
}
