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
 *  - every lane is a small state machine (fetch pixel -> generate bounce -> mesh traversal ->
 *    shade -> ...).  Lanes take pixels one by one (tile-major ids handed out per wave from a
 *    global tile counter), so no lane waits for the slowest pixel of a tile; the spp loop and
 *    the bounce loop are one flat sequence per lane, so no lane waits for the longest path of
 *    the wave (the RNG stream stays per-pixel-sequential, SURVEY.md §7 hard part 3);
 *  - BVH traversal is decoupled from the bounce loop: lanes that need it park in a wait state
 *    and the wave runs traversal steps only while enough lanes (a ballot count) are
 *    traversing; lanes whose ray missed every mesh box, or finished early, go on shading and
 *    generating instead of idling behind the longest traversal;
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
    const v4f *meshes;
    const v4f *objtab;   /* the object list (rt_object, 3 x 16 B each), read with wave-uniform addresses */
    uint2 *stack;        /* [stack_entries + 1][NT] deferred sibling: (entry distance bits, reference) */
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

/* Development instrumentation (-DRT_STATS, tools/stats_run.py): per code section, how many
 * times a wave executed it and with how many active lanes.  Compiled out of the product. */
#ifdef RT_STATS
#define RT_STAT(slot) do { unsigned long long m_ = __ballot(1); if (lane == __builtin_ctzll(m_)) { st_exec[slot] += 1u; st_lanes[slot] += (unsigned)__popcll(m_); } } while (0)
#else
#define RT_STAT(slot) do { } while (0)
#endif
/* ... and (lap timer, s_memtime) where a wave's time goes: RT_LAP(slot) charges the time since the
 * previous lap to `slot` */
enum { TM_CTL = 0, TM_SHADE = 1, TM_FETCH = 2, TM_GEN = 3, TM_MESH = 4, TM_DESCEND = 5, TM_LEAF = 6, TM_POP = 7, TM_N = 8 };
#ifdef RT_STATS
#define RT_LAP(slot) do { unsigned long long now_ = __builtin_readcyclecounter(); st_time[slot] += now_ - st_last; st_last = now_; } while (0)
/* inside the divergent `if (w_active)` block: leave it, lap with every lane, enter it again (the
 * timers are per-lane registers; only laps that all lanes execute measure the wave) */
#define RT_LAP_SPLIT(slot) } RT_LAP(slot); if (w_active) {
#else
#define RT_LAP(slot) do { } while (0)
#define RT_LAP_SPLIT(slot)
#endif
enum { ST_ITER = 0, ST_SHADE = 1, ST_SHADE_HIT = 2, ST_FETCH = 3, ST_GEN = 4, ST_MESH = 5, ST_MESH_START = 6, ST_WORK_ITER = 7, ST_NODE = 8, ST_LEAF_TRI = 9, ST_POP = 10, ST_DONE_MESH = 11, ST_N = 12 };

/* lane states of the render loop */
enum { M_FETCH = 0, M_GEN = 1, M_MESH = 2, M_WAIT = 3, M_SHADE = 4, M_DONE = 5, M_IDLE = 6 };

/* SCENE_LDS = false is the fallback for scenes larger than a CU's LDS: the same code reads the
 * scene sections from global memory (they stay L2 / Infinity-Cache resident) and only the
 * traversal stack lives in LDS. */
template <int NT, bool HAS_MESH, bool SCENE_LDS>
__global__ __launch_bounds__(NT) void rt_render_kernel(const rt_kernel_args a)
{
    extern __shared__ v4f lds_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & (RT_WAVE - 1);

    Lds L;
    if (SCENE_LDS) {
        /* stage the scene into LDS: coalesced 16-byte loads, one pass per workgroup */
        for (int i = tid; i < a.blob_f4; i += NT) lds_raw[i] = ((const v4f *)a.blob)[i];
        L.nodes = lds_raw + a.off_nodes;
        L.tris = lds_raw + a.off_tris;
        L.objs = lds_raw + a.off_objlds;
        L.meshes = lds_raw + a.off_meshes;
        L.objtab = lds_raw + a.off_objtab;
        L.stack = (uint2 *)(lds_raw + a.blob_f4);
    } else {
        const v4f *g = (const v4f *)a.blob;
        L.nodes = g + a.off_nodes;
        L.tris = g + a.off_tris;
        L.objs = g + a.off_objlds;
        L.meshes = g + a.off_meshes;
        L.objtab = g + a.off_objtab;
        L.stack = (uint2 *)lds_raw;
    }
    __syncthreads();
    if ((tid >> 6) >= a.max_waves) return;

    const V3 cam_pos = v3(a.cam[0], a.cam[1], a.cam[2]);
    const V3 tl = v3(a.cam[3], a.cam[4], a.cam[5]);
    const V3 du = v3(a.cam[6], a.cam[7], a.cam[8]);
    const V3 dv = v3(a.cam[9], a.cam[10], a.cam[11]);
    const V3 sky = v3(a.sky[0], a.sky[1], a.sky[2]);
    const int W = a.width, H = a.height;
    const int spp = a.rays_per_pixel, limit = a.reflection_limit;
    const int tiles_per_band = a.tiles_x * (a.band_rows >> 3);

    /* ---- per-lane pixel state (registers) ---- */
    int mode = M_FETCH;
    uint32_t rng = 0;
    V3 colour = v3(0.f, 0.f, 0.f), fin = colour, thr = colour, o = colour, d = colour, inv = colour, primary = colour;
    int sample = 0, bounce = 0, px = 0, py = 0;
    float cur_n = 1.0f;                  /* Ray::current_refractive_index src/ray.cu:56,144 */
    float best_t = RT_INF_F;
    int best_obj = -1, best_prim = -1, next_mesh = 0;
    /* ---- per-lane traversal state (registers + LDS stack) ---- */
    bool w_active = false;
    uint32_t cur = 0;
    int sp = 0, w_prim = -1, w_obj = -1;
    float w_best = RT_INF_F;
    /* ---- wave-uniform pixel chunk: linear pixel ids [chunk_next, chunk_end) of one 8x8 tile ---- */
    uint32_t chunk_next = 0, chunk_end = 0;
    bool exhausted = false;
#ifdef RT_COSTMAP
    /* development build (tools/costmap.py): the frame holds, per pixel, (own traversal steps,
     * start tick, end tick) of the 100 MHz wall clock instead of the colour */
    unsigned c_steps = 0, c_t0 = 0, c_wsteps = 0;
#define RT_COST(x) do { x; } while (0)
#else
#define RT_COST(x) do { } while (0)
#endif
#ifdef RT_STATS
    unsigned st_exec[ST_N], st_lanes[ST_N];
    for (int i = 0; i < ST_N; i++) { st_exec[i] = 0; st_lanes[i] = 0; }
    unsigned long long st_time[TM_N], st_last = __builtin_readcyclecounter();
    const unsigned long long st_wall0 = wall_clock64();
    for (int i = 0; i < TM_N; i++) st_time[i] = 0;
#endif

    for (;;) {
        RT_STAT(ST_ITER);
        RT_LAP(TM_CTL);
        /* ================= SHADE: the closest hit of this bounce is known ================== */
        if (mode == M_SHADE) {
            RT_STAT(ST_SHADE);
            bool end_sample;
            if (best_obj < 0) {
                /* src/raytracer.cu:76-80 */
                fin = fin + sky * thr;
                end_sample = true;
            } else {
                RT_STAT(ST_SHADE_HIT);
                const v4f ma = L.objs[RT_OBJLDS_F4 * best_obj], mb = L.objs[RT_OBJLDS_F4 * best_obj + 1];
                const uint32_t packed = __float_as_uint(mb.w);
                const int mtype = (int)(packed & 3u);
                /* hit point and normal: Ray::get_pos src/ray.cu:63-65; Sphere :66; Triangle :158 */
                V3 P = d * best_t + o;
                V3 N;
                float tex_u = 0.f, tex_v = 0.f;
                if (packed & 32u) {
                    const v4f sc = L.objs[RT_OBJLDS_F4 * best_obj + 2];
                    N = normalised(P - v3(sc.x, sc.y, sc.z));
                    if (packed & 16u) {
                        /* Sphere::assign_texture_coords src/objects.cu:82-97 (latitude / longitude) */
                        const float PI = 3.141592653589793f;
                        const float theta = rt_asinf((P.y - sc.y) / sc.w);
                        const float phi = rt_acosf((P.x - sc.x) / sc.w);
                        tex_u = (theta + PI / 2) / PI;
                        const float v_ratio = (1 - phi / PI) / 2;
                        const int behind = P.z > sc.z ? 1 : 0;
                        const int mult = 1 - 2 * behind;
                        tex_v = (float)(1 * behind) + (float)mult * v_ratio;
                    }
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
                /* update_ray src/raytracer.cu:49-64: REFRACTIVE goes through Ray::refract
                 * (src/ray.cu:77-128, Snell + Schlick + total internal reflection), which falls
                 * back to reflect(); everything else reflects */
                bool do_reflect = true;
                V3 refr_dir = v3(0.f, 0.f, 0.f);
                if (mtype == RT_DEV_MAT_REFRACTIVE) {
                    const float mat_n = L.objs[RT_OBJLDS_F4 * best_obj + 3].x;
                    float n1, n2;
                    V3 rn;
                    if (dot(N, d) > 0.0f) { n1 = mat_n; n2 = cur_n; rn = N; }        /* leaving the object */
                    else                  { n1 = cur_n; n2 = mat_n; rn = neg(N); }   /* entering */
                    cur_n = n2;
                    /* min(float, double) is CUDA's double overload; acos / asin of doubles */
                    const float theta1 = (float)rt_acos(fmin((double)dot(d, rn), 1.0));
                    const float theta2 = (float)rt_asin(fmin((double)(n1 * rt_sinf(theta1) / n2), 1.0));
                    const float critical_angle = rt_asinf(n2 / n1);
                    /* get_reflection_coeff :188-196: pow(float, int) is the double pow */
                    const float sqrt_r0 = (n1 - n2) / (n1 + n2);
                    const float r0 = sqrt_r0 * sqrt_r0;
                    const float cos_theta = rt_cosf(theta1);
                    const float reflection_coeff = (float)((double)r0 + (double)(1.0f - r0) * rt_pow5((double)(1.0f - cos_theta)));
                    do_reflect = theta1 > critical_angle;
                    if (!do_reflect) do_reflect = reflection_coeff > rt_u01(rt_pcg_next(&rng));   /* `||` short-circuits */
                    if (!do_reflect) {
                        V3 perp = v3(0.f, 0.f, 0.f);
                        if (theta1 != 0.0f) perp = (d - rn * rt_cosf(theta1)) / rt_sinf(theta1);
                        refr_dir = normalised(rn * rt_cosf(theta2) + perp * rt_sinf(theta2));
                    }
                }
                if (do_reflect) {
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
                    d = normalised(diffuse_dir + (specular_dir - diffuse_dir) * ma.w);
                } else {
                    d = refr_dir;
                }
                o = P;

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
                    } else if (tex == 3) {
                        /* image src/material.cu:119-124: nearest texel; an out-of-range index is clamped */
                        const int iw = (int)__float_as_uint(ma.x), ih = (int)__float_as_uint(ma.y);
                        const int uc = (int)((float)(iw - 1) * tex_u), vc = (int)((float)(ih - 1) * tex_v);
                        int idx = vc * iw + uc;
                        idx = idx < 0 ? 0 : (idx > iw * ih - 1 ? iw * ih - 1 : idx);
                        const float *tx = a.tex_data + (size_t)__float_as_uint(ma.z) + 3 * (size_t)idx;
                        tc = v3(tx[0], tx[1], tx[2]);
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
            mode = M_GEN;
            if (end_sample) {
                /* src/raytracer.cu:102-105: the next sample restarts from a copy of the primary ray */
                colour = colour + fin;
                sample++;
                fin = v3(0.f, 0.f, 0.f); thr = v3(1.f, 1.f, 1.f);
                o = cam_pos; d = primary; bounce = 0; cur_n = 1.0f;
                if (sample >= spp) {
                    /* src/raytracer.cu:107-112 and :133-135 */
                    const int array_index = (py * W + px) * 3;
                    V3 c = colour / (float)spp;
                    V3 previous = v3(0.f, 0.f, 0.f);
                    if (a.prev) previous = v3(a.prev[array_index], a.prev[array_index + 1], a.prev[array_index + 2]);
                    V3 previous_sum = previous * (float)a.frame_num;
                    V3 res = (c + previous_sum) / (float)(a.frame_num + 1);
                    int out_row = py;
                    if (a.compact) {
                        const int band = py / a.band_rows;
                        out_row = ((band - a.band_first) / a.band_stride) * a.band_rows + (py - band * a.band_rows);
                    }
                    float *dst = a.out + ((size_t)out_row * (size_t)W + (size_t)px) * 3;
#ifdef RT_COSTMAP
                    res = v3(__uint_as_float(RT_COSTMAP == 2 ? c_wsteps : c_steps), __uint_as_float(c_t0), __uint_as_float((unsigned)wall_clock64()));
#endif
                    dst[0] = res.x; dst[1] = res.y; dst[2] = res.z;
                    mode = M_FETCH;
                }
            }
        }

        RT_LAP(TM_SHADE);
        /* ================= FETCH: lanes without a pixel take the next ones ==================
         * Linear pixel ids are tile-major (64 per 8x8 tile), tiles come from a global counter;
         * a wave asks for one tile at a time and hands its ids out to whichever lanes are free. */
        {
            if (a.chunk_log2 < 6) {
                /* experiment: lanes left over by a small ticket sit idle until the wave's pixels are done */
                const bool busy = __ballot(mode == M_GEN || mode == M_MESH || mode == M_WAIT || mode == M_SHADE) != 0ull;
                if (busy && mode == M_FETCH) mode = M_IDLE;
                if (!busy && mode == M_IDLE) mode = M_FETCH;
            }
            const bool want = mode == M_FETCH;
            const unsigned long long mask = __ballot(want);
            if (mask) {
                const int need = __popcll(mask);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                int taken = 0;
                int my_id = -1;
                bool got_ticket = false;
                for (;;) {
                    const int avail = (int)(chunk_end - chunk_next);
                    const int take = avail < need - taken ? avail : need - taken;
                    if (want && rank >= taken && rank < taken + take) my_id = (int)chunk_next + (rank - taken);
                    chunk_next += (uint32_t)take;
                    taken += take;
                    if (taken == need || exhausted || (a.chunk_log2 < 6 && got_ticket)) break;
                    uint32_t t = 0;
                    if (lane == 0) t = atomicAdd(a.tile_counter, 1u);
                    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
                    if (t >= (uint32_t)a.num_tiles) { exhausted = true; break; }
                    /* ticket -> tile through a permutation.  A pixel's samples are sequential, so
                     * the frame cannot finish before its most expensive tile does; the host
                     * therefore lists the tiles whose centre ray enters a mesh box first
                     * (longest-job-first), each class scattered by a stride coprime to the tile
                     * count so that neighbouring (equally expensive) tiles do not land on the
                     * waves of one CU.  Any order gives the same image. */
                    t = a.tile_order ? a.tile_order[t]
                                     : (uint32_t)(((unsigned long long)t * (unsigned long long)a.tile_stride) % (unsigned long long)a.num_tiles);
                    chunk_next = t << a.chunk_log2;
                    chunk_end = chunk_next + (1u << a.chunk_log2);
                    got_ticket = true;
                }
                if (want) {
                    if (my_id < 0) {
                        if (exhausted) mode = M_DONE;
                    } else {
                        const int tile = my_id >> 6, within = my_id & 63;
                        const int band_local = tile / tiles_per_band;
                        const int in_band = tile - band_local * tiles_per_band;
                        const int band = a.band_first + band_local * a.band_stride;
                        const int ty = in_band / a.tiles_x, tx = in_band - ty * a.tiles_x;
                        px = tx * 8 + (within & 7);
                        py = band * a.band_rows + ty * 8 + (within >> 3);
                        if (px < W && py < H) {
                            /* src/raytracer.cu:123-127; Ray::set_direction_origin src/ray.cu:147-155,
                             * cam_pixel_to_world src/camera.cu:24-29 */
                            const int array_index = (py * W + px) * 3;
                            rng = (uint32_t)array_index * 3145739u + a.seed_time;
                            RT_COST(c_steps = 0; c_wsteps = 0; c_t0 = (unsigned)wall_clock64());
                            V3 plane_point = du * (float)px + dv * (float)py;
                            primary = normalised((tl + plane_point) - cam_pos);
                            colour = v3(0.f, 0.f, 0.f);
                            fin = v3(0.f, 0.f, 0.f); thr = v3(1.f, 1.f, 1.f);
                            o = cam_pos; d = primary;
                            bounce = 0; cur_n = 1.0f;
                            /* a zero bounce limit traces nothing: every sample is (0,0,0) */
                            sample = limit > 0 ? 0 : spp;
                            if (sample >= spp) {
                                const float q = 0.0f / (float)spp;               /* NaN for spp == 0, like the reference */
                                V3 previous = v3(0.f, 0.f, 0.f);
                                if (a.prev) previous = v3(a.prev[array_index], a.prev[array_index + 1], a.prev[array_index + 2]);
                                V3 res = (v3(q, q, q) + previous * (float)a.frame_num) / (float)(a.frame_num + 1);
                                int out_row = py;
                                if (a.compact) out_row = band_local * a.band_rows + (py - band * a.band_rows);
                                float *dst = a.out + ((size_t)out_row * (size_t)W + (size_t)px) * 3;
                                dst[0] = res.x; dst[1] = res.y; dst[2] = res.z;
                                /* stays in M_FETCH: takes another pixel next time round */
                            } else {
                                mode = M_GEN;
                            }
                        }
                        /* a pixel outside the image (ragged edge tile): stay in M_FETCH */
                    }
                }
            }
        }

        RT_LAP(TM_FETCH);
        /* ================= GEN: jitter the direction, test the simple objects ============== */
        if (mode == M_GEN) {
            RT_STAT(ST_GEN);
            /* Ray::apply_antialias src/ray.cu:130-142 */
            if (a.antialias) {
                V3 off;
                off.x = rt_jitter(rt_pcg_next(&rng));
                off.y = rt_jitter(rt_pcg_next(&rng));
                off.z = rt_jitter(rt_pcg_next(&rng));
                d = normalised(d + off);
            }
            if (HAS_MESH) inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);   /* src/ray.cu:198-202 */

            /* get_ray_collision src/raytracer.cu:24-46 over the non-mesh objects, in list order
             * (`<=`: the later object wins ties, :36; the precision_error term is a no-op for
             * accepted hits).  Meshes are merged afterwards with the same
             * rule made explicit: smaller distance, or equal distance and larger list index. */
            best_t = RT_INF_F; best_obj = -1; best_prim = -1;
            for (int i = 0; i < a.num_objects; i++) {
                /* rt_object from LDS: every lane reads the same address (broadcast) */
                const v4f ob0 = L.objtab[3 * i], ob1 = L.objtab[3 * i + 1], ob2 = L.objtab[3 * i + 2];
                rt_object ob;
                ob.type = (int32_t)__float_as_uint(ob0.x); ob.prim_start = (int32_t)__float_as_uint(ob0.y);
                ob.need_uv = (int32_t)__float_as_uint(ob0.z); ob.root_ref = __float_as_uint(ob0.w);
                ob.v[0] = ob1.x; ob.v[1] = ob1.y; ob.v[2] = ob1.z; ob.v[3] = ob1.w;
                ob.v[4] = ob2.x; ob.v[5] = ob2.y; ob.v[6] = ob2.z; ob.v[7] = ob2.w;
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
                    default: break;             /* RT_OBJ_MESH: below */
                }
                if (hit && t <= best_t) { best_t = t; best_obj = i; best_prim = prim; }
            }
            next_mesh = 0;
            mode = (HAS_MESH && a.num_meshes > 0) ? M_MESH : M_SHADE;
        }

        RT_LAP(TM_GEN);
        if (HAS_MESH) {
            /* ================= MESH: find the next mesh whose root box the ray enters ======= */
            while (mode == M_MESH) {
                RT_STAT(ST_MESH);
                if (next_mesh >= a.num_meshes) { mode = M_SHADE; break; }
                const v4f m0 = L.meshes[2 * next_mesh], m1 = L.meshes[2 * next_mesh + 1];
                next_mesh++;
                /* a NaN direction (Box-Muller on a zero draw) fails every
                 * triangle test: the mesh cannot be hit, no need to walk it */
                if (d.x != d.x || d.y != d.y || d.z != d.z) continue;
                /* the root is pushed unconditionally and tested when popped (src/objects.cu:494-501) */
                const uint32_t root_ref = __float_as_uint(m1.z);
                float rd;
                const bool rh = box_test(m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, o, inv, rd);
                if (!rh || rd > RT_INF_F || ((root_ref & RT_REF_CHAIN) && !(rd < RT_INF_F))) continue;
                cur = root_ref; sp = 0; w_best = RT_INF_F; w_prim = -1; w_obj = (int)__float_as_uint(m1.w);
                w_active = true;
                mode = M_WAIT;
                RT_STAT(ST_MESH_START);
            }

            RT_LAP(TM_MESH);
            /* ================= WORK: BVH traversal steps (src/objects.cu:487-532, :586-600) ====
             * Runs while enough lanes are traversing; lanes whose ray is finished go back to
             * shading as soon as the traversing group is small.  Visit order, push order and
             * every comparison are the reference's. */
            for (;;) {
                const int n_active = __popcll(__ballot(w_active));
                if (n_active == 0) break;
                const int n_ready = __popcll(__ballot(mode != M_WAIT && mode != M_DONE && mode != M_IDLE));
                if (n_ready > 0 && (n_active < a.work_threshold || n_ready >= a.ready_break)) break;
#if defined(RT_COSTMAP) && RT_COSTMAP == 2
                c_wsteps += 1;      /* wave-level macro steps this lane lived through */
#endif
                bool at_leaf = false, need_pop = false;
                RT_LAP(TM_CTL);
                if (w_active) {
                    RT_STAT(ST_WORK_ITER);
                    /* one macro step: descend to a leaf (or run out of children), test the
                     * leaf's triangles, pop the next deferred sibling.  (A variant that
                     * schedules node steps and single-triangle steps by lane majority issued
                     * ~30 % fewer wave instructions but ran slower: the extra ballots and
                     * branches lengthen each wave's serial instruction stream, and at the 4
                     * waves/SIMD an LDS-resident scene allows that latency is not hidden.) */
                    at_leaf = (cur & RT_REF_LEAF) != 0u;
                    need_pop = at_leaf;
                    if (!at_leaf) {
                        /* The body is branch-free: the deferred sibling is ALWAYS written to the
                         * slot above the top of the stack (one 8-byte LDS store) and the stack
                         * pointer moves only when both children are entered, so the only
                         * divergent branch of the loop is its exit.  The loop also ends, for
                         * everybody, once fewer than `descend_keep`/64 of the lanes that entered
                         * it are still descending: those lanes just stay on their internal node
                         * and go on next step, instead of making the others wait out the deepest
                         * descent of the wave. */
                        const int n_enter = __popcll(__ballot(1));
                        const int n_keep = (n_enter * a.descend_keep) >> 6;
                        for (;;) {
                            RT_STAT(ST_NODE);
                            RT_COST(c_steps++);
                            const v4f *n = L.nodes + 4 * (int)(cur & RT_REF_NODE_MASK);
                            v4f q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
                            float ld, rdist;
                            const bool lh = box_test(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, inv, ld);
                            const bool rh2 = box_test(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, inv, rdist);
                            const uint32_t lref = __float_as_uint(q3.x), rref = __float_as_uint(q3.y);
                            const bool l_push = lh && ld < w_best;
                            const bool r_push = rh2 && rdist < w_best;
                            const bool l_first = ld < rdist;
                            /* Of two entered children the one pushed first (left when l_first) is
                             * visited second: it is the deferred sibling.  The other is popped
                             * immediately (its distance is still < best).  With one entered child
                             * that child is next and nothing is deferred. */
                            const bool both = l_push && r_push;
                            const bool entered = l_push || r_push;
                            const uint32_t deferred_ref = l_first ? lref : rref;
                            const float deferred_d = l_first ? ld : rdist;
                            L.stack[sp * NT + tid] = make_uint2(__float_as_uint(deferred_d), deferred_ref);
                            sp += both ? 1 : 0;
                            const uint32_t next = both ? (l_first ? rref : lref) : (l_push ? lref : rref);
                            cur = entered ? next : cur;
                            at_leaf = entered && (next & RT_REF_LEAF) != 0u;
                            need_pop = !entered || at_leaf;
                            if (need_pop) break;
                            if (__popcll(__ballot(1)) < n_keep) break;      /* wave-uniform */
                        }
                    }
                    RT_LAP_SPLIT(TM_DESCEND)
                    if (at_leaf) {
                        /* leaf: strict <, first triangle wins ties (:596) */
                        const int start = (int)(cur & RT_REF_START_MASK);
                        const int count = (int)((cur >> RT_REF_COUNT_SHIFT) & RT_REF_COUNT_MAX);
                        for (int k = 0; k < count; k++) {
                            RT_STAT(ST_LEAF_TRI);
                            RT_COST(c_steps++);
                            float t, u, v;
                            bool h = tri_test(L.tris, start + k, o, d, t, u, v);
                            if (h && t < w_best) { w_best = t; w_prim = start + k; }
                        }
                    }
                    /* pop: an entry is taken iff !(dist > best) (:501); through a collapsed chain
                     * iff dist < best (:517) */
                    RT_LAP_SPLIT(TM_LEAF)
                    bool found = !need_pop;                 /* still descending: nothing to pop */
                    while (need_pop && sp > 0) {
                        RT_STAT(ST_POP);
                        sp--;
                        const uint2 e = L.stack[sp * NT + tid];
                        const float dd = __uint_as_float(e.x);
                        const uint32_t rr = e.y;
                        const bool take = (rr & RT_REF_CHAIN) ? (dd < w_best) : !(dd > w_best);
                        if (take) { cur = rr; found = true; break; }
                    }
                    if (!found) {
                        RT_STAT(ST_DONE_MESH);
                        /* this mesh is done: merge (smaller distance, or equal and later in the list) */
                        if (w_prim >= 0 && (w_best < best_t || (w_best == best_t && w_obj > best_obj))) {
                            best_t = w_best; best_obj = w_obj; best_prim = w_prim;
                        }
                        w_active = false;
                        mode = next_mesh >= a.num_meshes ? M_SHADE : M_MESH;
                    }
                }
                RT_LAP(TM_POP);
            }
        }

        if (__ballot(mode != M_DONE) == 0ull) break;
    }
#ifdef RT_STATS
    for (int i = 0; i < ST_N; i++) {
        if (st_exec[i]) { atomicAdd(&a.stats[2 * i], (unsigned long long)st_exec[i]); atomicAdd(&a.stats[2 * i + 1], (unsigned long long)st_lanes[i]); }
    }
    RT_LAP(TM_CTL);
    if (lane == 0) {
        for (int i = 0; i < TM_N; i++) atomicAdd(&a.stats[24 + i], st_time[i]);
        atomicAdd(&a.stats[24 + TM_N], wall_clock64() - st_wall0);      /* summed wave lifetimes, 100 MHz ticks */
        atomicAdd(&a.stats[24 + TM_N + 1], 1ull);                        /* waves */
    }
#endif
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

/* Element-wise evaluation of the shared math / RNG headers on the device, for the test that
 * checks them bit for bit against the same headers compiled for the host (rt_debug_eval). */
__global__ void rt_eval_kernel(int op, const uint32_t *in, uint32_t *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t u = in[i];
    const float x = __uint_as_float(u);
    float r = 0.0f;
    switch (op) {
        case 0: r = rt_logf(x); break;
        case 1: r = rt_cosf(x); break;
        case 2: r = rt_sinf(x); break;
        case 3: r = rt_asinf(x); break;
        case 4: r = rt_acosf(x); break;
        case 5: r = rt_u01(u); break;
        case 6: r = rt_jitter(u); break;
        case 7: r = rt_theta(u); break;
        case 8: r = sqrtf(x); break;                 /* the IEEE operations parity relies on */
        case 9: r = 1.0f / x; break;
        case 10: r = (float)rt_pow5((double)x); break;
        default: break;
    }
    out[i] = __float_as_uint(r);
}

extern "C" hipError_t rt_launch_eval(int op, const uint32_t *in, uint32_t *out, int n, hipStream_t stream)
{
    hipLaunchKernelGGL(rt_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, op, in, out, n);
    return hipGetLastError();
}

/* ---- launchers (called from rt_capi.cpp) -------------------------------------------------- */
template <int NT, bool HAS_MESH, bool SCENE_LDS>
static void rt_launch_one(const rt_kernel_args *args, int blocks, size_t lds_bytes, hipStream_t stream)
{
    (void)hipFuncSetAttribute((const void *)rt_render_kernel<NT, HAS_MESH, SCENE_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL((rt_render_kernel<NT, HAS_MESH, SCENE_LDS>), dim3(blocks), dim3(NT), lds_bytes, stream, *args);
}

extern "C" hipError_t rt_launch_render(const rt_kernel_args *args, int has_mesh, int scene_in_lds, int threads, int blocks, size_t lds_bytes, hipStream_t stream)
{
    if (!scene_in_lds) {
        /* global-memory scene: one shape per mesh flag is enough */
        if (has_mesh && threads == 1024) rt_launch_one<1024, true, false>(args, blocks, lds_bytes, stream);
        else if (!has_mesh && threads == 256) rt_launch_one<256, false, false>(args, blocks, lds_bytes, stream);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
#define RT_CASE(NTV)                                                                                   \
    case NTV:                                                                                          \
        if (has_mesh) rt_launch_one<NTV, true, true>(args, blocks, lds_bytes, stream);                 \
        else rt_launch_one<NTV, false, true>(args, blocks, lds_bytes, stream);                         \
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
