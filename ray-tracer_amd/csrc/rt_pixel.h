/*
 * rt_pixel.h — device code shared by the render kernels of rt_kernel.hip: vector helpers, the
 * primitive tests, and the three per-pixel sections of the lane state machine (SHADE, FETCH, GEN).
 *
 * The arithmetic (types, order, the double-precision fragments) is the reference's; file:line
 * citations are on each piece.  -ffp-contract=off is assumed (see rt_kernel.hip).
 */
#ifndef RT_PIXEL_H
#define RT_PIXEL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device_scene.h"
#include "rt_math.h"
#include "rt_rng.h"

#define RT_WAVE 64

/* 16-byte vector for LDS / global accesses: a single ds_read_b128 / global_load_dwordx4 each
 * (a struct of four floats gets split into narrower loads by the optimiser) */
typedef float v4f __attribute__((ext_vector_type(4)));

/* ---- 1 / x and sqrt(x), correctly rounded, in a third of the instructions (round 4) ---------------------------------
 * The compiler expands `1.0f / x` into 11 instructions (v_div_scale x 2, v_rcp, five fma, v_div_fmas, v_div_fixup) and sqrtf
 * into 17 + 5 s_nop, most of it for inputs a renderer never sees: denormals, results that underflow, zero, infinity.  On gfx950
 *   v_rcp_f32 + one Newton step (two fma)                is 1.0f / x bit for bit for every x with 2^-126 <= |x| <= 2^126,
 *   v_sqrt_f32 + the -1 / +1 ulp residual test (2 fma)   is sqrtf(x) bit for bit for every x with 2^-64 <= x < inf,
 * checked EXHAUSTIVELY - all 2^32 inputs against the compiler's expansions on the device, tests/test_gpu_math.py
 * (tools/ubench/exact_div_sqrt.hip, profiles/r04/experiments/exact_div_sqrt.txt: outside those ranges every single input fails,
 * inside none).  rt_sqrt / rt_rcp_sqrt take the short forms when EVERY active lane's operand is inside the range (one subtract,
 * one compare, a wave-uniform branch) and the compiler's otherwise: the value is the IEEE one for every input, always.  Used
 * where it pays: the normalisations (1 / sqrt: 41 -> 24 instructions), the sphere test's and Box-Muller's roots; same-box A/B:
 * three-sphere -8 %, cube -3.6 %, reference scene 0 -0.9 %, monkey -0.3 % (profiles/r04/experiments/exact_div_sqrt_ab.txt). */
__device__ __forceinline__ bool rt_rcp_in_range(float x) { return ((__float_as_uint(x) & 0x7fffffffu) - 0x00800000u) <= 0x7e000000u; }
__device__ __forceinline__ bool rt_sqrt_in_range(float x) { return (__float_as_uint(x) - 0x1f800000u) < 0x60000000u; }
__device__ __forceinline__ float rt_rcp_short(float x)
{
    const float y = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(y, __builtin_fmaf(-x, y, 1.0f), y);
}
/* sqrtf(x) for x in the range above, from the reciprocal square root: s0 = x * rsq(x) is the root to ~2 ulp, and one step
 * s0 + (x - s0^2) * (rsq / 2) with the residual as an fma lands on the correctly rounded value for EVERY binary32 from 2^-102 up
 * (tools/ubench/rsq_forms.hip on the device; tests/test_gpu_math.py's exhaustive test runs this very function against sqrtf over
 * all 2^32 patterns).  Five instructions - v_rsq_f32, two multiplies, two fma - where round 4's first form (v_sqrt_f32, then a
 * residual test of the neighbours one ulp down and up: RT_SQRT_BY_NEIGHBOURS keeps it for A/B builds) took nine, four of them
 * compares and selects. */
__device__ __forceinline__ float rt_sqrt_short(float x)
{
#ifdef RT_SQRT_BY_NEIGHBOURS
    float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float rdn = __builtin_fmaf(-dn, s, x), rup = __builtin_fmaf(-up, s, x);
    s = rdn <= 0.0f ? dn : s;
    return rup > 0.0f ? up : s;
#else
    const float y = __builtin_amdgcn_rsqf(x);
    const float s0 = x * y, h = 0.5f * y;
    return __builtin_fmaf(__builtin_fmaf(-s0, s0, x), h, s0);
#endif
}
__device__ __forceinline__ float rt_sqrt(float x)       /* == sqrtf(x) */
{
    if (__ballot(!rt_sqrt_in_range(x)) == 0ull) return rt_sqrt_short(x);
    return sqrtf(x);
}
/* 1.0f / sqrtf(m): sqrt of an in-range m lies in [2^-32, 2^64], inside the reciprocal's range - one check for both */
__device__ __forceinline__ float rt_rcp_sqrt(float m)
{
    if (__ballot(!rt_sqrt_in_range(m)) == 0ull) return rt_rcp_short(rt_sqrt_short(m));
    return 1.0f / sqrtf(m);
}

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
    float inv = rt_rcp_sqrt(m);
    return v3(a.x * inv, a.y * inv, a.z * inv);
}
__device__ __forceinline__ V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }

/* src/utils.cu:234-239 — Box-Muller cosine branch, theta drawn first.  rt_rng.h produces the
 * reference's (float)(r / 4294967295.0) and the binary64 products derived from it without the
 * binary64 divide, bit for bit (tests/test_rng_exhaustive.py covers all 2^32 inputs). */
template <bool SHORT_DIVIDE, bool GENERAL_FUNCTIONS>
__device__ __forceinline__ float normal_num(uint32_t &state)
{
    float theta = rt_theta(rt_pcg_next(&state));
#ifdef RT_GENERIC_BOX_MULLER       /* (A/B builds: the general-purpose log and cos) */
    float rho = rt_sqrt(-2.0f * rt_logf(rt_u01(rt_pcg_next(&state))));
    return rho * rt_cosf(theta);
#else
    if (GENERAL_FUNCTIONS) {         /* (the hybrid kernels: see px_shade) */
        float rho_g = rt_sqrt(-2.0f * rt_logf(rt_u01(rt_pcg_next(&state))));
        return rho_g * rt_cosf(theta);
    }
    /* log on [0, 1] and cos on [0, 6.28318]: rt_logf / rt_cosf without the cases these arguments cannot be (rt_math.h) */
    float rho = rt_sqrt(-2.0f * rt_logf_0_1(rt_u01(rt_pcg_next(&state)), SHORT_DIVIDE ? 1 : 0));
    return rho * rt_cosf_0_2pi(theta);
#endif
}

/* the scene sections (LDS, or global memory for what of a large scene does not fit a CU's LDS) */
struct Lds {
    const v4f *nodes;
    const v4f *tris;
    const v4f *objs;
    const v4f *meshes;
    const v4f *objtab;   /* the object list (rt_object, 3 x 16 B each), read with wave-uniform addresses */
};

/* c ? a : b as a v_cndmask_b32 in its VOP3 form (mask from an SGPR pair).  The compiler prefers the VOP2 form, which reads
 * the mask from VCC - and two of THOSE back to back cost the issuing wave 16 cycles each instead of 4 on gfx950
 * (tools/ubench/valu_tput.hip K_CNDMASK / K_CC2 against K_CNDMASK_S / K_CC2S; profiles/r04/experiments/valu_tput.txt).
 * Used where the traversal loops select two values on one condition. */
__device__ __forceinline__ uint32_t rt_sel_u32(unsigned long long lanes, uint32_t a, uint32_t b)       /* lanes = __ballot(condition) */
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(lanes));
    return r;
}
__device__ __forceinline__ float rt_sel_f32(unsigned long long lanes, float a, float b) { return __uint_as_float(rt_sel_u32(lanes, __float_as_uint(a), __float_as_uint(b))); }

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

/* The same slab test combined with the traversal's `entry distance < best` (src/objects.cu:509,
 * :517): enter = hit && tmin < best.  tmin >= 0 always (it starts from 0 and fmaxf drops NaNs), so
 * `tmin < tmax` already implies `tmax > 0`, and with neither tmax nor best ever NaN the two
 * remaining comparisons fold into one: tmin < min(tmax, best).  Same decisions, 3 compares and 2
 * mask operations fewer per box - measurable where a wave's serial instruction stream is the
 * critical path (tools/ubench/node_step.hip: 693 -> 633 cycles per node step). */
__device__ __forceinline__ bool box_enter(float bx0, float by0, float bz0, float bx1, float by1, float bz1,
                                          V3 o, V3 inv, float best, float &tmin_out)
{
    float tmin = 0.0f, tmax = RT_INF_F;
    float t1 = (bx0 - o.x) * inv.x, t2 = (bx1 - o.x) * inv.x;
    tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
    t1 = (by0 - o.y) * inv.y; t2 = (by1 - o.y) * inv.y;
    tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
    t1 = (bz0 - o.z) * inv.z; t2 = (bz1 - o.z) * inv.z;
    tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
    tmin_out = tmin;
    return tmin < fminf(tmax, best);
}

/* The same decision and, where the box is entered, the same entry distance from six v_med3_f32 instead of ten
 * min / max (round 4; the kernel is bound by instruction issue and min / max / med3 / compares cost twice an add or a
 * multiply there).  clamp(t; a, b) = med3(a, b, t) puts t into the slab's interval [min(a,b), max(a,b)].
 * phi = clamp_z o clamp_y o clamp_x is non-decreasing; the kernel enters iff phi(0) < phi(best):
 *  - if the reference enters (tmin < min(tmax, best), tmin = max(0, near_k), tmax = min(INF, far_k)): every near_k <= tmin <
 *    far_k, so clamping 0 from below only ever raises it to the next near_k: phi(0) = tmin, the SAME float (a maximum
 *    selects one of its operands); likewise phi(best) = min(best, far_k) > tmin: entered, with the reference's distance;
 *  - if it does not: the slabs' intervals are either disjoint somewhere (then phi is constant) or have a common
 *    intersection [N, F] onto which phi clamps, and max(0, N) >= min(best, F) gives phi(0) >= phi(best); with phi
 *    monotone that is equality: not entered.
 * best <= RT_INF_F always (w_best starts there and only falls), so min(INF, ...) needs no instruction.  The argument
 * needs every product to be a number: (b - o) * inv is NaN only for 0 * inf, i.e. a direction component of exactly 0
 * (NaN directions never traverse); rays with one take box_enter (the caller checks, wave-uniformly). */
__device__ __forceinline__ bool box_enter_med3(float bx0, float by0, float bz0, float bx1, float by1, float bz1,
                                               V3 o, V3 inv, float best, float &tmin_out)
{
    const float x0 = (bx0 - o.x) * inv.x, x1 = (bx1 - o.x) * inv.x;
    const float y0 = (by0 - o.y) * inv.y, y1 = (by1 - o.y) * inv.y;
    const float z0 = (bz0 - o.z) * inv.z, z1 = (bz1 - o.z) * inv.z;
    const float lo = __builtin_amdgcn_fmed3f(z0, z1, __builtin_amdgcn_fmed3f(y0, y1, __builtin_amdgcn_fmed3f(x0, x1, 0.0f)));
    const float hi = __builtin_amdgcn_fmed3f(z0, z1, __builtin_amdgcn_fmed3f(y0, y1, __builtin_amdgcn_fmed3f(x0, x1, best)));
    tmin_out = lo;
    return lo < hi;
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

/* The same test for the traversal's leaf loop, with the outcome as a lane mask (compares written straight to SGPR pairs and
 * combined there): which lanes' rays hit AND are closer than `best`.  (__ballot of a bool built from several compares costs a
 * v_cndmask and a v_cmp to rebuild the mask.) */
#define RT_FCMP_OGT 2
#define RT_FCMP_OGE 3
#define RT_FCMP_OLT 4
__device__ __forceinline__ unsigned long long tri_closer_lanes(const v4f *tris, int idx, V3 o, V3 d, float best, float &t_out)
{
    float t, u, v;
    v4f q0 = tris[3 * idx], q1 = tris[3 * idx + 1], q2 = tris[3 * idx + 2];
    V3 p0 = v3(q0.x, q0.y, q0.z), s1 = v3(q0.w, q1.x, q1.y), s2 = v3(q1.z, q1.w, q2.x);
    V3 p_vec = cross(d, s2);
    float det = dot(s1, p_vec);
    float inv_det = 1.0f / det;        /* (the short reciprocal behind a range check is SLOWER here - monkey +3 % early in round 4, +1.4 % on its final code, cube -0.9 %: the check and its branch sit in the leaf loop) */
    V3 t_vec = o - p0;
    u = dot(t_vec, p_vec) * inv_det;
    V3 q_vec = cross(t_vec, s1);
    v = dot(d, q_vec) * inv_det;
    float w = 1.0f - u - v;
    t = dot(s2, q_vec) * inv_det;
    t_out = t;
    /* u >= 0 && v >= 0 && w >= 0 is one compare of v_minimum3_f32 (gfx950; IEEE-754-2019 minimum: a NaN operand gives NaN,
     * which fails the compare exactly as it fails its own; -0 >= 0 holds either way) */
    const float m = __builtin_elementwise_minimum(__builtin_elementwise_minimum(u, v), w);
    return __builtin_amdgcn_fcmpf(t, RT_EPS_F, RT_FCMP_OGT) & __builtin_amdgcn_fcmpf(m, 0.0f, RT_FCMP_OGE) & __builtin_amdgcn_fcmpf(t, best, RT_FCMP_OLT);
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

/* lane states of the render loop */
enum { M_FETCH = 0, M_GEN = 1, M_MESH = 2, M_WAIT = 3, M_SHADE = 4, M_DONE = 5 };

/* per-lane pixel state (registers) */
struct Px {
    int mode;
    uint32_t rng;
    V3 colour, fin, thr, o, d, inv, primary;
    int sample, bounce;
    unsigned id;                 /* the pixel: 64 * (tile of this launch) + (row in tile * 8 + column in tile) */
    float cur_n;                 /* Ray::current_refractive_index src/ray.cu:56,144 */
    float best_t;
    int best_obj, best_prim, next_mesh;
    /* bits RT_FRAME_BITS-1..0: which frame of a multi-frame launch this pixel belongs to; bits 30..RT_FRAME_BITS: what it has
     * cost so far (RT_COST_* units), reported per tile when the launch collects costs (tile_cost); bit 31: one of its rays
     * has entered a mesh */
    unsigned frame_steps;
#ifdef RT_COSTMAP
    /* development build (tools/costmap.py): the frame holds, per pixel, (own traversal steps,
     * start tick, end tick) of the 100 MHz wall clock instead of the colour */
    unsigned c_steps, c_t0, c_wsteps;
#endif
};
#ifdef RT_COSTMAP
#define RT_COST(x) do { x; } while (0)
#else
#define RT_COST(x) do { } while (0)
#endif

/* wave-uniform frame constants */
struct Frame {
    V3 cam_pos, tl, du, dv, sky;
    int W, H, spp, limit, tiles_per_band;
};

/* wave-uniform pixel chunk: linear pixel ids [next, end) of one 8x8 tile */
struct Chunk {
    uint32_t next, end;
    int frame;                   /* the frame (of a multi-frame launch) the ids belong to */
    bool exhausted;
};

__device__ __forceinline__ void px_init(Px &p)
{
    const V3 z = v3(0.f, 0.f, 0.f);
    p.mode = M_FETCH; p.rng = 0;
    p.colour = z; p.fin = z; p.thr = z; p.o = z; p.d = z; p.inv = z; p.primary = z;
    p.sample = 0; p.bounce = 0; p.id = 0;
    p.cur_n = 1.0f; p.best_t = RT_INF_F;
    p.best_obj = -1; p.best_prim = -1; p.next_mesh = 0; p.frame_steps = 0;
    RT_COST(p.c_steps = 0; p.c_t0 = 0; p.c_wsteps = 0);
}

__device__ __forceinline__ void frame_init(Frame &f, const rt_kernel_args &a)
{
    f.cam_pos = v3(a.cam[0], a.cam[1], a.cam[2]);
    f.tl = v3(a.cam[3], a.cam[4], a.cam[5]);
    f.du = v3(a.cam[6], a.cam[7], a.cam[8]);
    f.dv = v3(a.cam[9], a.cam[10], a.cam[11]);
    f.sky = v3(a.sky[0], a.sky[1], a.sky[2]);
    f.W = a.width; f.H = a.height;
    f.spp = a.rays_per_pixel; f.limit = a.reflection_limit;
    f.tiles_per_band = a.tiles_x * (a.band_rows >> 3);
}

/* tile t of this launch -> its place in the image (tx, ty in tiles) and, for the compact band layout, the row of
 * the output buffer its first pixel row goes to */
__device__ __forceinline__ void tile_place(const rt_kernel_args &a, const Frame &f, int t, int &tx, int &ty, int &compact_row)
{
    if (a.tile_list) {
        const int g = (int)a.tile_list[t];
        ty = g / a.tiles_x; tx = g - ty * a.tiles_x;
        compact_row = 0;
    } else {
        const int band_local = t / f.tiles_per_band;
        const int in_band = t - band_local * f.tiles_per_band;
        const int tyb = in_band / a.tiles_x;
        tx = in_band - tyb * a.tiles_x;
        ty = (a.band_first + band_local * a.band_stride) * (a.band_rows >> 3) + tyb;
        compact_row = band_local * a.band_rows + tyb * 8;
    }
}

/* A pixel's samples are done: blend with the previous frame and store (src/raytracer.cu:107-112,
 * :133-135).
 *
 * Multi-frame launches (rt_kernel_args.partial != NULL) render consecutive progressive frames of one
 * view in ONE launch: every frame has its own seed, so frame k+1 of a pixel can be traced while
 * frame k of the same pixel is still being traced by another wave - the only thing that is
 * sequential is the blend, (c + prev * n) / (n + 1) with prev = the pixel's value after frame k.
 * So each frame only stores c, the mean of its own samples, into its plane of a scratch buffer
 * (plain stores, no ordering between frames needed), and a small kernel launched behind this one
 * (rt_blend_kernel) folds the planes into the frame buffer in frame order; NaN pixels are made the one
 * canonical quiet NaN there (any NaN plane value makes the blended value a NaN). */
__device__ __forceinline__ void px_finish_pixel(Px &p, const rt_kernel_args &a, const Frame &f)
{
    const V3 c = p.colour / (float)f.spp;
    const int tile = (int)(p.id >> 6), within = (int)(p.id & 63u);
    int tx, ty, compact_row;
    tile_place(a, f, tile, tx, ty, compact_row);
    const int px = tx * 8 + (within & 7), py = ty * 8 + (within >> 3);
    size_t pixel = (size_t)py * (size_t)f.W + (size_t)px;
    if (a.compact) pixel = a.tile_list ? (size_t)p.id : (size_t)(compact_row + (within >> 3)) * (size_t)f.W + (size_t)px;
    /* (first launch of a view) what this pixel cost, charged to its tile */
    if (a.tile_cost && (p.frame_steps & (unsigned)(RT_MAX_BATCH_FRAMES - 1)) == 0u) {
        /* (frame 0 of the launch only: the figures describe the view, not the launch, and a 32-frame sum could wrap)
         * cost units in bits 31..1 of the tile's sum; bit 0: some pixel of the tile traversed a mesh (those tiles are
         * the long jobs the schedule puts first, rt_capi.cpp build_job_order) */
        unsigned units = (p.frame_steps & 0x7fffffffu) >> RT_FRAME_BITS;
        units = units > RT_COST_PIXEL_CAP ? RT_COST_PIXEL_CAP : units;
        atomicAdd(a.tile_cost + tile, units << 1);
        /* ... and the tile's most expensive pixel: a tile-frame is one job, as long as its longest pixel (a pixel's samples
         * are one sequential stream), and the schedule starts the longest jobs first */
        atomicMax(a.tile_peak + tile, units);
        if (p.frame_steps >> 31) atomicOr(a.tile_cost + tile, 1u);
    }
    p.mode = M_FETCH;
    if (a.partial) {
        float *dst = a.partial + ((size_t)(p.frame_steps & (unsigned)(RT_MAX_BATCH_FRAMES - 1)) * a.partial_plane + pixel) * 3;
        dst[0] = c.x; dst[1] = c.y; dst[2] = c.z;
        return;
    }
    float *dst = a.out + pixel * 3;
    const int array_index = (py * f.W + px) * 3;
    V3 previous = v3(0.f, 0.f, 0.f);
    if (a.prev) previous = v3(a.prev[array_index], a.prev[array_index + 1], a.prev[array_index + 2]);
    V3 previous_sum = previous * (float)a.frame_num;
    V3 res = (c + previous_sum) / (float)(a.frame_num + 1);
#ifdef RT_COSTMAP
    res = v3(__uint_as_float(RT_COSTMAP == 2 ? p.c_wsteps : p.c_steps), __uint_as_float(p.c_t0), __uint_as_float((unsigned)wall_clock64()));
#endif
    dst[0] = rt_canon_nan(res.x); dst[1] = rt_canon_nan(res.y); dst[2] = rt_canon_nan(res.z);
}

/* the end of a sample (src/raytracer.cu:102-105): add it to the pixel, restart from a copy of the
 * primary ray; after the last sample the pixel is finished */
__device__ __forceinline__ void px_end_sample(Px &p, const rt_kernel_args &a, const Frame &f)
{
    p.colour = p.colour + p.fin;
    p.sample++;
    p.fin = v3(0.f, 0.f, 0.f); p.thr = v3(1.f, 1.f, 1.f);
    p.o = f.cam_pos; p.d = p.primary; p.bounce = 0; p.cur_n = 1.0f;
    if (p.sample >= f.spp) px_finish_pixel(p, a, f);
}

/* ================= SHADE, a ray that hit nothing (src/raytracer.cu:76-80): sky, end of sample == */
__device__ __forceinline__ void px_shade_miss(Px &p, const rt_kernel_args &a, const Frame &f)
{
    p.fin = p.fin + f.sky * p.thr;
    p.mode = M_GEN;
    px_end_sample(p, a, f);
}

/* ================= SHADE: the closest hit of this bounce is known (p.best_obj >= 0) ========= */
/* SHORT_DIVIDE: the logarithm's division in its short form (rt_math.h rt__div_benign; the same values): faster in every kernel but the
 * 1024-thread mesh kernel (three-sphere -4.2 %, cube -1.4 %, monkey +0.4 %), which keeps the division operator */
/* GENERAL_FUNCTIONS: Box-Muller through rt_logf / rt_cosf instead of their forms for a draw's arguments (again the same values): the hybrid
 * kernels (nodes in LDS, triangles from L2) are 2.8 % FASTER that way on the 6,000-triangle scene and indifferent on the 50,880-triangle one
 * (profiles/r04/experiments/box_muller_on_its_domain.txt) */
template <bool SHORT_DIVIDE, bool GENERAL_FUNCTIONS>
__device__ __forceinline__ void px_shade(Px &p, const rt_kernel_args &a, const Frame &f, const Lds &L)
{
    V3 &o = p.o, &d = p.d;
    {
        const int best_obj = p.best_obj, best_prim = p.best_prim;
        const v4f ma = L.objs[RT_OBJLDS_F4 * best_obj], mb = L.objs[RT_OBJLDS_F4 * best_obj + 1];
        const uint32_t packed = __float_as_uint(mb.w);
        const int mtype = (int)(packed & 3u);
        /* hit point and normal: Ray::get_pos src/ray.cu:63-65; Sphere :66; Triangle :158 */
        V3 P = d * p.best_t + o;
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
            if (dot(N, d) > 0.0f) { n1 = mat_n; n2 = p.cur_n; rn = N; }        /* leaving the object */
            else                  { n1 = p.cur_n; n2 = mat_n; rn = neg(N); }   /* entering */
            p.cur_n = n2;
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
            if (!do_reflect) do_reflect = reflection_coeff > rt_u01(rt_pcg_next(&p.rng));   /* `||` short-circuits */
            if (!do_reflect) {
                V3 perp = v3(0.f, 0.f, 0.f);
                if (theta1 != 0.0f) perp = (d - rn * rt_cosf(theta1)) / rt_sinf(theta1);
                refr_dir = normalised(rn * rt_cosf(theta2) + perp * rt_sinf(theta2));
            }
        }
        if (do_reflect) {
            /* Ray::reflect src/ray.cu:67-75 with diffuse_reflect :157-170,
             * true_lambertian_reflect :172-178, perfect_reflect :180-186, lerp :32-34 */
            float gx = normal_num<SHORT_DIVIDE, GENERAL_FUNCTIONS>(p.rng);
            float gy = normal_num<SHORT_DIVIDE, GENERAL_FUNCTIONS>(p.rng);
            float gz = normal_num<SHORT_DIVIDE, GENERAL_FUNCTIONS>(p.rng);
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
            p.fin = p.fin + v3(mb.x, mb.y, mb.z) * p.thr;
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
                const int uc = rt_f2i((float)(iw - 1) * tex_u), vc = rt_f2i((float)(ih - 1) * tex_v);
                int idx = (int)((uint32_t)vc * (uint32_t)iw + (uint32_t)uc);       /* wraps like the 32-bit machine arithmetic */
                idx = idx < 0 ? 0 : (idx > iw * ih - 1 ? iw * ih - 1 : idx);
                const float *tx = a.tex_data + (size_t)__float_as_uint(ma.z) + 3 * (size_t)idx;
                tc = v3(tx[0], tx[1], tx[2]);
            } else {
                const int nsq = (int)(packed >> 8);                      /* checkerboard :90-99 */
                const int uc = rt_f2i(tex_u * (float)nsq), vc = rt_f2i(tex_v * (float)nsq);
                tc = ((int)((uint32_t)uc + (uint32_t)vc) % 2 == 0) ? v3(ma.x, ma.y, ma.z) : v3(mb.x, mb.y, mb.z);
            }
            p.thr = p.thr * tc;
        }
        p.bounce++;
        p.frame_steps += (unsigned)(RT_COST_HIT * RT_MAX_BATCH_FRAMES);
    }
    p.mode = M_GEN;
    if (p.bounce >= f.limit) px_end_sample(p, a, f);
}

/* ================= FETCH: lanes without a pixel take the next ones =========================
 * Linear pixel ids are tile-major (64 per 8x8 tile), tiles come from a global counter; a wave
 * asks for one tile at a time and hands its ids out to whichever lanes are free.  Must be called
 * by the whole wave. */
__device__ __forceinline__ void px_fetch(Px &p, Chunk &ch, const rt_kernel_args &a, const Frame &f, int lane)
{
    const bool want = p.mode == M_FETCH;
    const unsigned long long mask = __ballot(want);
    if (!mask) return;
    const int need = __popcll(mask);
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    int taken = 0;
    int my_id = -1, my_frame = 0;
    for (;;) {
        const int avail = (int)(ch.end - ch.next);
        const int take = avail < need - taken ? avail : need - taken;
        if (want && rank >= taken && rank < taken + take) { my_id = (int)ch.next + (rank - taken); my_frame = ch.frame; }
        ch.next += (uint32_t)take;
        taken += take;
        if (taken == need || ch.exhausted) break;
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(a.tile_counter, 1u);
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        /* Tickets of a multi-frame launch: first the `num_heavy_tiles` most expensive tiles (the
         * head of tile_order) of frame 0, of frame 1, ... of the last frame, then the other tiles
         * frame by frame.  Frames only meet in the blend of a pixel, so the longest jobs of EVERY
         * frame can start at once and everything else fills in behind them: the launch is then as
         * long as its work, not as its last frame's tail.  (num_heavy_tiles == 0, or one frame:
         * plainly frame by frame.) */
        if (t >= (uint32_t)a.num_tiles * (uint32_t)a.num_frames) { ch.exhausted = true; break; }
        if (a.job_order) {
            /* the host has laid out the whole schedule (rt_capi.cpp: longest job first over all frames) */
            const uint32_t job = a.job_order[t];
            ch.frame = (int)(job >> RT_JOB_FRAME_SHIFT);
            t = job & RT_JOB_TILE_MASK;
            ch.next = t * 64u;
            ch.end = t * 64u + 64u;
            continue;
        }
        uint32_t fr;
        const uint32_t nh = (uint32_t)a.num_heavy_tiles;
        if (t < nh * (uint32_t)a.num_frames) {
            fr = t / nh;
            t -= fr * nh;
        } else {
            const uint32_t nl = (uint32_t)a.num_tiles - nh;
            t -= nh * (uint32_t)a.num_frames;
            fr = t / nl;
            t = nh + (t - fr * nl);
        }
        ch.frame = (int)fr;
        /* ticket -> tile through a permutation.  A pixel's samples are sequential, so the frame
         * cannot finish before its most expensive tile does; the host therefore lists the tiles
         * whose centre ray enters a mesh box first (longest-job-first), each class scattered by a
         * stride coprime to the tile count so that neighbouring (equally expensive) tiles do not
         * land on the waves of one CU.  Any order gives the same image. */
        t = a.tile_order ? a.tile_order[t]
                         : (uint32_t)(((unsigned long long)t * (unsigned long long)a.tile_stride) % (unsigned long long)a.num_tiles);
        ch.next = t * 64u;
        ch.end = t * 64u + 64u;
    }
    if (!want) return;
    if (my_id < 0) { p.mode = M_DONE; return; }
    const int tile = my_id >> 6, within = my_id & 63;
    int tx, ty, compact_row;
    tile_place(a, f, tile, tx, ty, compact_row);
    const int px = tx * 8 + (within & 7);
    const int py = ty * 8 + (within >> 3);
    p.id = (unsigned)my_id;
    if (px < f.W && py < f.H) {
        /* src/raytracer.cu:123-127; Ray::set_direction_origin src/ray.cu:147-155,
         * cam_pixel_to_world src/camera.cu:24-29 */
        const int array_index = (py * f.W + px) * 3;
        p.frame_steps = (unsigned)my_frame;
        p.rng = (uint32_t)array_index * 3145739u + a.seeds[my_frame];
        RT_COST(p.c_steps = 0; p.c_wsteps = 0; p.c_t0 = (unsigned)wall_clock64());
        V3 plane_point = f.du * (float)px + f.dv * (float)py;
        p.primary = normalised((f.tl + plane_point) - f.cam_pos);
        p.colour = v3(0.f, 0.f, 0.f);
        p.fin = v3(0.f, 0.f, 0.f); p.thr = v3(1.f, 1.f, 1.f);
        p.o = f.cam_pos; p.d = p.primary;
        p.bounce = 0; p.cur_n = 1.0f;
        /* a zero bounce limit traces nothing: every sample is (0,0,0) */
        p.sample = f.limit > 0 ? 0 : f.spp;
        if (p.sample >= f.spp) {
            const float q = 0.0f / (float)f.spp;               /* NaN for spp == 0, like the reference */
            p.colour = v3(q, q, q) * (float)f.spp;             /* px_finish_pixel divides by spp again: q either way */
            px_finish_pixel(p, a, f);                          /* M_FETCH: takes another pixel next time round */
        } else {
            p.mode = M_GEN;
        }
    }
    /* a pixel outside the image (ragged edge tile): stay in M_FETCH */
}

/* ================= GEN: jitter the direction, test the simple objects ====================== */
template <bool HAS_MESH>
__device__ __forceinline__ void px_gen(Px &p, const rt_kernel_args &a, const Lds &L)
{
    V3 &o = p.o, &d = p.d;
    p.frame_steps += (unsigned)(RT_COST_GEN * RT_MAX_BATCH_FRAMES);
    /* Ray::apply_antialias src/ray.cu:130-142 */
    if (a.antialias) {
        V3 off;
        off.x = rt_jitter(rt_pcg_next(&p.rng));
        off.y = rt_jitter(rt_pcg_next(&p.rng));
        off.z = rt_jitter(rt_pcg_next(&p.rng));
        d = normalised(d + off);
    }
    if (HAS_MESH) p.inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);   /* src/ray.cu:198-202 (three short reciprocals behind one range check: -0.3 % cube, +0.5 % monkey - not taken) */

    /* get_ray_collision src/raytracer.cu:24-46 over the non-mesh objects, in list order
     * (`<=`: the later object wins ties, :36; the precision_error term is a no-op for
     * accepted hits).  Meshes are merged afterwards with the same
     * rule made explicit: smaller distance, or equal distance and larger list index. */
    float best_t = RT_INF_F;
    int best_obj = -1, best_prim = -1;
    for (int i = 0; i < a.num_objects; i++) {
        /* rt_object from LDS: every lane reads the same address (broadcast) */
        const v4f ob0 = L.objtab[3 * i], ob1 = L.objtab[3 * i + 1], ob2 = L.objtab[3 * i + 2];
        rt_object ob;
        /* (the record is the same for every lane, but sending its type through an SGPR - scalar branches instead of exec-mask
         * regions - was slower: +1.2 % on reference scene 0, profiles/r04/experiments/small_instruction_savings.txt) */
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
#if defined(RT_SPHERE_IEEE_DIVIDE)
                    float dist = (-qb - rt_sqrt(disc)) / (2.0f * qa);
#else
                    /* The near root's division in its short form (rt_math.h rt__div_benign).  d comes out of normalised(): a unit vector to a
                     * few ulp, so the divisor is 2 to a few ulp - or d has a NaN (divisor NaN: the quotient is NaN either way), or it is the
                     * zero vector (a vector whose squared length overflowed: then b and the dividend are zeros too and both forms give
                     * 0 / 0 = NaN).  For a dividend that form's precondition excludes - below 2^-100 in magnitude,
                     * or infinite - it may return another value than the operator, but never one that changes what follows: such a quotient is
                     * below RT_EPS_F (rejected), or - an infinite dividend: inf from the operator, NaN from the short form - fails
                     * `dist > RT_EPS_F` or `t <= best_t` (best_t <= 2^30) alike; every distance that IS accepted comes from a dividend between
                     * 2e-6 and 2^31, where the two agree bit for bit. */
                    float dist = rt__div_benign(-qb - rt_sqrt(disc), 2.0f * qa);
#endif
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
                for (int fc = 0; fc < 6; fc++) {
                    float ft; int fp;
                    bool fh = quad_test(L.tris, ob.prim_start + 2 * fc, o, d, ft, fp);
                    if (fh && ft < cb) { cb = ft; prim = fp; hit = true; }
                }
                t = cb;
                break;
            }
            default: break;             /* RT_OBJ_MESH: traversed separately */
        }
        if (hit && t <= best_t) { best_t = t; best_obj = i; best_prim = prim; }
    }
    p.best_t = best_t; p.best_obj = best_obj; p.best_prim = best_prim;
    p.next_mesh = 0;
    p.mode = (HAS_MESH && a.num_meshes > 0) ? M_MESH : M_SHADE;
}

#endif
