/*
 * rt_kernel.hip — the per-pixel hot path as a hand-written HIP kernel for gfx950 (MI355X).
 *
 * What it computes is the reference's get_pixel_colour (src/raytracer.cu:116-136) and
 * everything below it: camera ray (src/camera.cu:24-29, src/ray.cu:147-155), per-bounce
 * direction jitter (src/ray.cu:130-142), closest hit over the object list
 * (src/raytracer.cu:24-46) with sphere / Moller-Trumbore triangle / quad / one-way quad /
 * cuboid / BVH mesh tests (src/objects.cu), Lambertian-metal-emissive scattering
 * (src/ray.cu:67-75,157-186), the PCG stream (src/utils.cu:220-239) and the progressive
 * average (src/raytracer.cu:97-113).  The arithmetic (types, order, the double-precision
 * fragments) is the reference's; the program structure is not:
 *
 *  - one wave = one 8x8 pixel tile; waves pull tiles from a global counter until none are
 *    left, so a workgroup never idles behind its slowest tile;
 *  - the spp loop and the bounce loop are ONE flat loop per lane: a lane whose sample ended
 *    starts its next sample in the next iteration instead of waiting for the longest path of
 *    the wave (the RNG stream stays per-pixel-sequential, SURVEY.md §7 hard part 3);
 *  - the whole scene (BVH nodes with child boxes inline, 48-byte triangles, per-object
 *    shading record) is staged into LDS once per workgroup; the object list itself is read
 *    with scalar loads because every lane walks it in the same order;
 *  - BVH traversal keeps the current node in a register and only the deferred sibling (with
 *    its entry distance) on a per-lane LDS stack laid out [entry][thread], which is
 *    bank-conflict-free; a box is tested once, not twice as in the reference, by carrying the
 *    entry distance instead of re-testing on pop (same predicate, same outcome);
 *  - RNG state, ray, throughput and accumulators live in registers.
 *
 * No MFMA: there is no dense contraction anywhere in this workload.
 * Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (the reference's a*b+c are two
 * roundings; contraction would change hit/miss decisions).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device_scene.h"
#include "rt_math.h"
#include "rt_rng.h"

#define RT_WAVE 64

/* 16-byte vector for LDS / global accesses: a single ds_read_b128 / global_load_dwordx4 each
 * (a struct of four floats gets split into narrower loads by the optimiser) */
typedef float v4f __attribute__((ext_vector_type(4)));

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
/* src/utils.cu:130-136: (x*x' + y*y') + z*z' */
__device__ __forceinline__ float dot(V3 a, V3 b) { float nx = a.x * b.x, ny = a.y * b.y, nz = a.z * b.z; return nx + ny + nz; }
/* src/utils.cu:146-153 */
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
/* src/utils.cu:118-128: one reciprocal of the magnitude, three multiplies */
__device__ __forceinline__ V3 normalised(V3 a)
{
    float m = a.x * a.x + a.y * a.y + a.z * a.z;
    float inv = 1.0f / sqrtf(m);
    return v3(a.x * inv, a.y * inv, a.z * inv);
}
__device__ __forceinline__ V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }

/* src/utils.cu:234-239 — Box-Muller cosine branch, theta drawn first.  rt_rng.h produces the
 * reference's (float)(r / 4294967295.0) and the binary64 products derived from it without the
 * binary64 divide, bit for bit (tests/test_rng_exhaustive.py covers all 2^32 inputs). */
__device__ __forceinline__ float normal_num(uint32_t &state)
{
    float theta = rt_theta(rt_pcg_next(&state));
    float rho = sqrtf(-2.0f * rt_logf(rt_u01(rt_pcg_next(&state))));
    return rho * rt_cosf(theta);
}

struct Lds {
    const v4f *nodes;
    const v4f *tris;
    const v4f *objs;
    float *stack_d;      /* [RT_STACK_ENTRIES][NT] entry distance */
    uint32_t *stack_r;   /* [RT_STACK_ENTRIES][NT] node reference */
};

/* BoundingBox::ray_hits src/objects.cu:404-434.  fminf/fmaxf drop a NaN operand like CUDA's
 * min/max; the result only ever feeds comparisons, so the sign of a zero is irrelevant. */
__device__ __forceinline__ bool box_test(float bx0, float by0, float bz0, float bx1, float by1, float bz1,
                                         V3 o, V3 inv, float &tmin_out)
{
    float tmin = 0.0f, tmax = RT_INF_F;
    float t1 = (bx0 - o.x) * inv.x, t2 = (bx1 - o.x) * inv.x;
    tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
    t1 = (by0 - o.y) * inv.y; t2 = (by1 - o.y) * inv.y;
    tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
    t1 = (bz0 - o.z) * inv.z; t2 = (bz1 - o.z) * inv.z;
    tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
    tmin_out = tmin;
    return tmin < tmax && tmax > 0.0f;
}

/* Triangle::hit src/objects.cu:135-163 (Moller-Trumbore, two-sided, no early out) */
__device__ __forceinline__ bool tri_test(const v4f *tris, int idx, V3 o, V3 d, float &t_out, float &u_out, float &v_out)
{
    v4f q0 = tris[3 * idx], q1 = tris[3 * idx + 1], q2 = tris[3 * idx + 2];
    V3 p0 = v3(q0.x, q0.y, q0.z), s1 = v3(q0.w, q1.x, q1.y), s2 = v3(q1.z, q1.w, q2.x);
    V3 p_vec = cross(d, s2);
    float det = dot(s1, p_vec);
    float inv_det = 1.0f / det;
    V3 t_vec = o - p0;
    float u = dot(t_vec, p_vec) * inv_det;
    V3 q_vec = cross(t_vec, s1);
    float v = dot(d, q_vec) * inv_det;
    float w = 1.0f - u - v;
    float dist = dot(s2, q_vec) * inv_det;
    t_out = dist; u_out = u; v_out = v;
    return dist > RT_EPS_F && u >= 0.0f && v >= 0.0f && w >= 0.0f;
}

/* Quad::hit src/objects.cu:223-236 — t1 if it hits, whatever t2's distance; else t2 */
__device__ __forceinline__ bool quad_test(const v4f *tris, int first, V3 o, V3 d, float &t_out, int &prim_out)
{
    float t1, t2, u, v;
    bool h1 = tri_test(tris, first, o, d, t1, u, v);
    bool h2 = tri_test(tris, first + 1, o, d, t2, u, v);
    t_out = h1 ? t1 : t2;
    prim_out = h1 ? first : first + 1;
    return h1 || h2;
}

/* BVH::traverse src/objects.cu:487-532 + check_leaf_node :586-600 on the compact tree.
 * Visit order, push order and every comparison are the reference's; only the bookkeeping
 * differs (see the file header). */
template <int NT>
__device__ __forceinline__ bool mesh_test(const Lds &L, const rt_object &ob, V3 o, V3 d, V3 inv, int tid,
                                          float &t_out, int &prim_out)
{
    float best = RT_INF_F;
    int best_prim = -1;
    float rd;
    /* the root is pushed unconditionally and tested when popped (:494-501) */
    bool rh = box_test(ob.v[0], ob.v[1], ob.v[2], ob.v[3], ob.v[4], ob.v[5], o, inv, rd);
    if (rh && !(rd > best)) {
        uint32_t cur = ob.root_ref;
        int sp = 0;
        for (;;) {
            if (cur & RT_REF_LEAF) {
                int start = (int)(cur & RT_REF_START_MASK);
                int count = (int)((cur >> RT_REF_COUNT_SHIFT) & RT_REF_COUNT_MAX);
                for (int k = 0; k < count; k++) {
                    float t, u, v;
                    bool h = tri_test(L.tris, start + k, o, d, t, u, v);
                    if (h && t < best) { best = t; best_prim = start + k; }
                }
            } else {
                const v4f *n = L.nodes + 4 * (int)cur;
                v4f q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
                float ld, rdist;
                bool lh = box_test(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, inv, ld);
                bool rh2 = box_test(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, inv, rdist);
                uint32_t lref = __float_as_uint(q3.x), rref = __float_as_uint(q3.y);
                bool l_push = lh && ld < best;
                bool r_push = rh2 && rdist < best;
                bool l_first = ld < rdist;
                /* "first" is pushed first and therefore visited second */
                uint32_t first_ref = l_first ? lref : rref, second_ref = l_first ? rref : lref;
                float first_d = l_first ? ld : rdist;
                bool first_push = l_first ? l_push : r_push, second_push = l_first ? r_push : l_push;
                if (second_push) {
                    if (first_push) {
                        L.stack_d[sp * NT + tid] = first_d;
                        L.stack_r[sp * NT + tid] = first_ref;
                        sp++;
                    }
                    cur = second_ref;       /* popped immediately: its distance is still < best */
                    continue;
                }
                if (first_push) { cur = first_ref; continue; }
            }
            /* pop: skip entries that the best hit has overtaken (:501) */
            bool found = false;
            while (sp > 0) {
                sp--;
                float dd = L.stack_d[sp * NT + tid];
                if (!(dd > best)) { cur = L.stack_r[sp * NT + tid]; found = true; break; }
            }
            if (!found) break;
        }
    }
    t_out = best;
    prim_out = best_prim;
    return best_prim >= 0;
}

template <int NT, bool HAS_MESH>
__global__ __launch_bounds__(NT) void rt_render_kernel(const rt_kernel_args a)
{
    extern __shared__ v4f lds_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & (RT_WAVE - 1);

    /* stage the scene into LDS: coalesced 16-byte loads, one pass per workgroup */
    for (int i = tid; i < a.blob_f4; i += NT) lds_raw[i] = ((const v4f *)a.blob)[i];
    Lds L;
    L.nodes = lds_raw + a.off_nodes;
    L.tris = lds_raw + a.off_tris;
    L.objs = lds_raw + a.off_objlds;
    L.stack_d = (float *)(lds_raw + a.blob_f4);
    L.stack_r = (uint32_t *)(L.stack_d + RT_STACK_ENTRIES * NT);
    __syncthreads();

    const V3 cam_pos = v3(a.cam[0], a.cam[1], a.cam[2]);
    const V3 tl = v3(a.cam[3], a.cam[4], a.cam[5]);
    const V3 du = v3(a.cam[6], a.cam[7], a.cam[8]);
    const V3 dv = v3(a.cam[9], a.cam[10], a.cam[11]);
    const V3 sky = v3(a.sky[0], a.sky[1], a.sky[2]);
    const int W = a.width, H = a.height;
    const int spp = a.rays_per_pixel, limit = a.reflection_limit;

    for (;;) {
        /* one 8x8 tile per wave, handed out by a global counter */
        uint32_t tile = 0;
        if (lane == 0) tile = atomicAdd(a.tile_counter, 1u);
        tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile);
        if (tile >= (uint32_t)a.num_tiles) break;

        const int tiles_per_band = a.tiles_x * (a.band_rows >> 3);
        const int band_local = (int)tile / tiles_per_band;            /* k-th band owned by this launch */
        const int in_band = (int)tile % tiles_per_band;
        const int band = a.band_first + band_local * a.band_stride;   /* global band index */
        const int ty = in_band / a.tiles_x, tx = in_band % a.tiles_x;
        const int px = tx * 8 + (lane & 7);
        const int py_in_band = ty * 8 + (lane >> 3);
        const int py = band * a.band_rows + py_in_band;
        const bool in_image = px < W && py < H;

        /* src/raytracer.cu:123-127 */
        const int array_index = (py * W + px) * 3;
        uint32_t rng = (uint32_t)array_index * 3145739u + a.seed_time;

        /* Ray::set_direction_origin src/ray.cu:147-155, cam_pixel_to_world src/camera.cu:24-29 */
        V3 plane_point = du * (float)px + dv * (float)py;
        V3 view_pos = tl + plane_point;
        const V3 primary = normalised(view_pos - cam_pos);

        V3 colour = v3(0.f, 0.f, 0.f);
        V3 fin = v3(0.f, 0.f, 0.f), thr = v3(1.f, 1.f, 1.f);
        V3 o = cam_pos, d = primary;
        int sample = (in_image && limit > 0) ? 0 : spp;
        int bounce = 0;

        while (sample < spp) {
            /* Ray::apply_antialias src/ray.cu:130-142 (binary64 offset arithmetic) */
            if (a.antialias) {
                V3 off;
                off.x = rt_jitter(rt_pcg_next(&rng));
                off.y = rt_jitter(rt_pcg_next(&rng));
                off.z = rt_jitter(rt_pcg_next(&rng));
                d = normalised(d + off);
            }

            /* get_ray_collision src/raytracer.cu:24-46 */
            float best_t = RT_INF_F;
            int best_obj = -1, best_prim = -1;
            V3 inv = v3(0.f, 0.f, 0.f);
            if (HAS_MESH) inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);   /* src/ray.cu:198-202 */

            for (int i = 0; i < a.num_objects; i++) {
                const rt_object ob = a.objects[i];
                bool hit = false;
                float t = RT_INF_F;
                int prim = -1;
                switch (ob.type) {
                    case RT_OBJ_SPHERE: {   /* Sphere::hit src/objects.cu:40-79: near root, > 1e-6 */
                        V3 cq = v3(ob.v[0], ob.v[1], ob.v[2]) - o;
                        float qa = dot(d, d);
                        float qb = dot(d, cq) * (-2.0f);
                        float qc = dot(cq, cq) - ob.v[3] * ob.v[3];
                        float disc = qb * qb - 4.0f * qa * qc;
                        if (disc >= 0.0f) {
                            float dist = (-qb - sqrtf(disc)) / (2.0f * qa);
                            if (dist > RT_EPS_F) { hit = true; t = dist; }
                        }
                        break;
                    }
                    case RT_OBJ_TRIANGLE: {
                        float u, v;
                        hit = tri_test(L.tris, ob.prim_start, o, d, t, u, v);
                        prim = ob.prim_start;
                        break;
                    }
                    case RT_OBJ_ONE_WAY_QUAD:   /* src/objects.cu:273-280 */
                        if (dot(d, v3(ob.v[0], ob.v[1], ob.v[2])) < 0.0f) break;
                        /* fall through */
                    case RT_OBJ_QUAD:
                        hit = quad_test(L.tris, ob.prim_start, o, d, t, prim);
                        break;
                    case RT_OBJ_CUBOID: {       /* src/objects.cu:305-322: strict <, first face wins ties */
                        float cb = RT_INF_F;
                        for (int f = 0; f < 6; f++) {
                            float ft; int fp;
                            bool fh = quad_test(L.tris, ob.prim_start + 2 * f, o, d, ft, fp);
                            if (fh && ft < cb) { cb = ft; prim = fp; hit = true; }
                        }
                        t = cb;
                        break;
                    }
                    case RT_OBJ_MESH:
                        if (HAS_MESH) hit = mesh_test<NT>(L, ob, o, d, inv, tid, t, prim);
                        break;
                }
                /* `<=`: the later object wins ties (:36); the precision_error term is a no-op
                 * for accepted hits */
                if (hit && t <= best_t) { best_t = t; best_obj = i; best_prim = prim; }
            }

            bool end_sample;
            if (best_obj < 0) {
                /* src/raytracer.cu:76-80 */
                fin = fin + sky * thr;
                end_sample = true;
            } else {
                const v4f ma = L.objs[3 * best_obj], mb = L.objs[3 * best_obj + 1];
                const uint32_t packed = __float_as_uint(mb.w);
                const int mtype = (int)(packed & 3u);
                /* hit point and normal: Ray::get_pos src/ray.cu:63-65; Sphere :66; Triangle :158 */
                V3 P = d * best_t + o;
                V3 N;
                float tex_u = 0.f, tex_v = 0.f;
                if (packed & 32u) {
                    const v4f sc = L.objs[3 * best_obj + 2];
                    N = normalised(P - v3(sc.x, sc.y, sc.z));
                } else {
                    const v4f q2 = L.tris[3 * best_prim + 2];
                    V3 n = v3(q2.y, q2.z, q2.w);
                    N = (dot(n, d) > 0.0f) ? neg(n) : n;
                    if (packed & 16u) {
                        /* Triangle::assign_texture_coords src/objects.cu:160,196-199, called as (w,u,v) */
                        float t, u, v;
                        tri_test(L.tris, best_prim, o, d, t, u, v);
                        float w = 1.0f - u - v;
                        const float *uv = a.tri_uv + 6 * best_prim;
                        tex_u = uv[0] * w + uv[2] * u + uv[4] * v;
                        tex_v = uv[1] * w + uv[3] * u + uv[5] * v;
                    }
                }
                /* Ray::reflect src/ray.cu:67-75 with diffuse_reflect :157-170,
                 * true_lambertian_reflect :172-178, perfect_reflect :180-186, lerp :32-34 */
                float gx = normal_num(rng);
                float gy = normal_num(rng);
                float gz = normal_num(rng);
                V3 rv = v3(gx, gy, gz);
                if (dot(rv, N) < 0.0f) rv = neg(rv);
                rv = normalised(rv);
                V3 diffuse_dir = normalised(N + rv);
                float dn = dot(d, N);
                V3 specular_dir = normalised(d - (N * 2.0f) * dn);
                o = P;
                d = normalised(diffuse_dir + (specular_dir - diffuse_dir) * ma.w);

                /* src/raytracer.cu:86-90 */
                if (mtype == RT_DEV_MAT_EMISSIVE) {
                    fin = fin + v3(mb.x, mb.y, mb.z) * thr;
                } else {
                    V3 tc;
                    const int tex = (int)((packed >> 2) & 3u);
                    if (tex == 0) {
                        tc = v3(ma.x, ma.y, ma.z);
                    } else if (tex == 1) {
                        tc = v3(tex_u, tex_v, 0.f);                              /* gradient src/material.cu:80-82 */
                    } else {
                        const int nsq = (int)(packed >> 8);                      /* checkerboard :90-99 */
                        int uc = (int)(tex_u * (float)nsq), vc = (int)(tex_v * (float)nsq);
                        tc = ((uc + vc) % 2 == 0) ? v3(ma.x, ma.y, ma.z) : v3(mb.x, mb.y, mb.z);
                    }
                    thr = thr * tc;
                }
                bounce++;
                end_sample = bounce >= limit;
            }

            if (end_sample) {
                /* src/raytracer.cu:102-105: next sample restarts from a copy of the primary ray */
                colour = colour + fin;
                sample++;
                fin = v3(0.f, 0.f, 0.f); thr = v3(1.f, 1.f, 1.f);
                o = cam_pos; d = primary; bounce = 0;
            }
        }

        if (in_image) {
            /* src/raytracer.cu:107-112 and :133-135 */
            if (limit <= 0) colour = v3(0.f, 0.f, 0.f);
            colour = colour / (float)spp;
            V3 previous = v3(0.f, 0.f, 0.f);
            if (a.prev) previous = v3(a.prev[array_index], a.prev[array_index + 1], a.prev[array_index + 2]);
            V3 previous_sum = previous * (float)a.frame_num;
            V3 res = (colour + previous_sum) / (float)(a.frame_num + 1);
            const int out_row = a.compact ? (band_local * a.band_rows + py_in_band) : py;
            float *dst = a.out + ((size_t)out_row * (size_t)W + (size_t)px) * 3;
            dst[0] = res.x; dst[1] = res.y; dst[2] = res.z;
        }
    }
}

/* float -> RGBA8 of src/main.cu:343-371 */
__global__ void rt_rgba8_kernel(const float *rgb, int n_pixels, uint8_t *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    uint32_t packed = 0xff000000u;
    for (int c = 0; c < 3; c++) {
        int colour = (int)(rgb[3 * i + c] * 255.0f);
        colour = colour > 255 ? 255 : (colour < 0 ? 0 : colour);
        packed |= (uint32_t)colour << (8 * c);
    }
    ((uint32_t *)out)[i] = packed;
}

/* ---- launchers (called from rt_capi.cpp) -------------------------------------------------- */
extern "C" hipError_t rt_launch_render(const rt_kernel_args *args, int has_mesh, int threads, int blocks, size_t lds_bytes, hipStream_t stream)
{
#define RT_CASE(NTV)                                                                                   \
    case NTV:                                                                                          \
        if (has_mesh) {                                                                                \
            (void)hipFuncSetAttribute((const void *)rt_render_kernel<NTV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
            hipLaunchKernelGGL((rt_render_kernel<NTV, true>), dim3(blocks), dim3(NTV), lds_bytes, stream, *args);  \
        } else {                                                                                       \
            (void)hipFuncSetAttribute((const void *)rt_render_kernel<NTV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
            hipLaunchKernelGGL((rt_render_kernel<NTV, false>), dim3(blocks), dim3(NTV), lds_bytes, stream, *args); \
        }                                                                                              \
        break;
    switch (threads) {
        RT_CASE(256)
        RT_CASE(512)
        RT_CASE(768)
        RT_CASE(1024)
        default: return hipErrorInvalidValue;
    }
#undef RT_CASE
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_rgba8(const float *rgb, int n_pixels, uint8_t *out, hipStream_t stream)
{
    int blocks = (n_pixels + 255) / 256;
    hipLaunchKernelGGL(rt_rgba8_kernel, dim3(blocks), dim3(256), 0, stream, rgb, n_pixels, out);
    return hipGetLastError();
}
