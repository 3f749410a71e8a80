/*
 * rt_oracle.c — CPU oracle for the path-tracer hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of Ben-Edwards44/Ray-Tracer's renderer, written from the behaviour
 * of its sources (cited as file:line relative to /root/reference; no reference source is
 * copied or compiled).  The reference is CUDA + SFML and cannot be built in this image, so
 * parity is pinned differently — see the "Pin" paragraph below and DESIGN.md §3.
 *
 * Pin: in ORC_MATH_LIBM mode, built with `gcc -O2 -ffp-contract=off` against glibc 2.35,
 * this restatement reproduces the outputs the reference itself produced when its sources
 * were compiled CPU-only in this container during the survey (SURVEY.md App. A.12 camera
 * floats; App. C.2 framebuffer sha256 prefixes, means and pixel values; §4 BVH leaf
 * histograms).  tests/test_oracle_pin.py checks exactly that.  ORC_MATH_DET mode is the same
 * code with log/cos/sin/tan rebound to ray-tracer_amd/csrc/rt_math.h.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 */
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../ray-tracer_amd/csrc/rt_math.h"

/* ------------------------------------------------------------------------------------------
 * Vec3 — src/utils.cu:13-163.  All by-value float arithmetic, left to right.
 * ---------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 v = {x, y, z}; return v; }
static inline v3 v3_from(const float *p) { v3 v = {p[0], p[1], p[2]}; return v; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_div(v3 a, float s) { return v3_make(a.x / s, a.y / s, a.z / s); }
/* :130-136 */
static inline float v3_dot(v3 a, v3 b) { float nx = a.x * b.x, ny = a.y * b.y, nz = a.z * b.z; return nx + ny + nz; }
/* :146-153 */
static inline v3 v3_cross(v3 a, v3 b) { return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
/* :118-121 */
static inline float v3_magnitude(v3 a) { float m = a.x * a.x + a.y * a.y + a.z * a.z; return sqrtf(m); }
/* :123-128 — one reciprocal, three multiplies */
static inline v3 v3_normalised(v3 a) { float inv = 1 / v3_magnitude(a); return v3_make(a.x * inv, a.y * inv, a.z * inv); }
/* :155-162 */
static inline v3 v3_set_mag(v3 a, float mag) { float scale = mag / v3_magnitude(a); return v3_make(a.x * scale, a.y * scale, a.z * scale); }

/* src/objects.cu:6-7 — `1 << 31 - 1` parses as 1 << 30 */
#define ORC_INF_I (1 << 30)
static const float ORC_INF = (float)ORC_INF_I;
static const float ORC_EPS = 0.000001f;

/* ------------------------------------------------------------------------------------------
 * RNG — src/utils.cu:220-231.  PCG hash on a 32-bit LCG; uint / 4294967295.0 is a double
 * divide narrowed to float.
 * ---------------------------------------------------------------------------------------- */
static inline float pcg_next(uint32_t *state)
{
    uint32_t ns = *state * 747796405u + 2891336453u;
    *state = ns;
    uint32_t r = ((ns >> ((ns >> 28) + 4)) ^ ns) * 277803737u;
    r = (r >> 22) ^ r;
    return (float)((double)r / 4294967295.0);
}

/* ------------------------------------------------------------------------------------------
 * Scene types
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    v3 p[3];                 /* src/objects.cu:103 */
    v3 side1, side2, normal; /* :166-167, :106 (precompute :175-186) */
    float tu[3], tv[3];      /* texture_points :171; zero when the reference leaves them unset */
} tri_t;

typedef struct { v3 bl, tr; float width, height, depth; int assigned; } box_t;  /* :353-437 */

typedef struct {            /* BoundingBoxTris + left/right pointers, :440-445, :479-480 */
    box_t box;
    int left, right;
    int ntris;
    int *tri_idx;
} node_t;

typedef struct {
    tri_t *tris; int ntris;
    node_t *nodes; int nnodes, cap;
    int root;
} mesh_t;

enum { OBJ_SPHERE = 0, OBJ_TRIANGLE = 1, OBJ_QUAD = 2, OBJ_ONE_WAY_QUAD = 3, OBJ_CUBOID = 4, OBJ_MESH = 5 };  /* :804-809 */

typedef struct {
    int type;
    orc_material mat;
    v3 center; float radius;       /* sphere */
    tri_t tri;                     /* triangle */
    tri_t quad[2];                 /* quad / one-way quad */
    v3 owq_normal;
    tri_t cub[12];                 /* cuboid: 6 quads */
    mesh_t mesh;
} object_t;

struct orc_scene {
    int math_mode;
    object_t *objs; int nobjs, cap;
};

typedef struct {
    v3 origin, direction, direction_inv;
    uint32_t *rng;
    int antialias;
    float current_refractive_index;    /* src/ray.cu:56,144: 1 (air) for every fresh copy of the primary ray */
} ray_t;

typedef struct { int hits; float dist; v3 point, normal; float u, v; } hit_t;   /* RayHitData src/ray.cu:22-29 */

typedef struct { int rays_per_pixel, reflection_limit, antialias; v3 sky; } rsettings_t;   /* RenderData src/raytracer.cu:4-12 */

/* src/ray.cu:198-202 */
static inline void ray_change_direction(ray_t *r, v3 d)
{
    r->direction = d;
    r->direction_inv = v3_make(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
}

/* ------------------------------------------------------------------------------------------
 * Materials — src/material.cu
 * ---------------------------------------------------------------------------------------- */
void orc_material_standard(orc_material *m, int tex_type, const float colour[3], float smoothness)
{   /* :157-165 */
    memset(m, 0, sizeof *m);
    m->type = ORC_MAT_STANDARD;
    m->tex_type = tex_type;
    if (colour) memcpy(m->colour, colour, 12);
    m->smoothness = smoothness;
    m->need_uv = tex_type != ORC_TEX_COLOUR;
}

void orc_material_checkerboard(orc_material *m, const float light[3], const float dark[3], int num_squares, float smoothness)
{   /* :32-40 + :157-165 */
    orc_material_standard(m, ORC_TEX_CHECKERBOARD, NULL, smoothness);
    memcpy(m->light, light, 12);
    memcpy(m->dark, dark, 12);
    m->num_squares = num_squares;
}

void orc_material_emissive(orc_material *m, const float colour[3], float strength)
{   /* :167-173; unset fields defined as zero (App. A.9) */
    memset(m, 0, sizeof *m);
    m->type = ORC_MAT_EMISSIVE;
    m->emitted[0] = colour[0] * strength;
    m->emitted[1] = colour[1] * strength;
    m->emitted[2] = colour[2] * strength;
}

void orc_material_image(orc_material *m, int w, int h, const float *rgb, float smoothness)
{   /* Texture::create_image :42-51 + create_standard */
    orc_material_standard(m, ORC_TEX_IMAGE, NULL, smoothness);
    m->img_w = w; m->img_h = h; m->img_rgb = rgb;
}

void orc_material_refractive(orc_material *m, const float colour[3], float n)
{   /* :175-185 */
    memset(m, 0, sizeof *m);
    m->type = ORC_MAT_REFRACTIVE;
    m->tex_type = ORC_TEX_COLOUR;
    memcpy(m->colour, colour, 12);
    m->refractive_index = n;
    m->smoothness = 1;
}

/* Texture::get_texture_colour :53-69 */
static inline v3 texture_colour(const orc_material *m, float u, float v)
{
    switch (m->tex_type) {
        case ORC_TEX_COLOUR: return v3_from(m->colour);                 /* :75-77 */
        case ORC_TEX_GRADIENT: return v3_make(u, v, 0);                 /* :80-82 */
        case ORC_TEX_CHECKERBOARD: {                                    /* :90-99 */
            /* float -> int as CUDA converts (NaN -> 0, saturating), rt_math.h rt_f2i: a plain C cast is undefined
             * for the NaN a sphere's u can be at the pole (asin of 1 + ulp), and x86 would give INT_MIN */
            int uc = rt_f2i(u * m->num_squares);
            int vc = rt_f2i(v * m->num_squares);
            return ((int)((uint32_t)uc + (uint32_t)vc) % 2 == 0) ? v3_from(m->light) : v3_from(m->dark);
        }
        case ORC_TEX_IMAGE: {                                           /* :119-124 */
            int uc = rt_f2i((m->img_w - 1) * u);
            int vc = rt_f2i((m->img_h - 1) * v);
            int idx = (int)((uint32_t)vc * (uint32_t)m->img_w + (uint32_t)uc);
            /* the reference indexes unchecked; an out-of-range texel is clamped here (and in the kernel) */
            if (idx < 0) idx = 0;
            if (idx > m->img_w * m->img_h - 1) idx = m->img_w * m->img_h - 1;
            return v3_from(m->img_rgb + 3 * idx);
        }
        default: return v3_make(0, 0, 0);
    }
}

/* ------------------------------------------------------------------------------------------
 * Shapes — src/objects.cu
 * ---------------------------------------------------------------------------------------- */
/* Triangle::precompute :175-186 (plane / area are dead) */
static tri_t tri_make(v3 a, v3 b, v3 c)
{
    tri_t t;
    memset(&t, 0, sizeof t);
    t.p[0] = a; t.p[1] = b; t.p[2] = c;
    t.side1 = v3_sub(b, a);
    t.side2 = v3_sub(c, a);
    t.normal = v3_normalised(v3_cross(t.side1, t.side2));
    return t;
}

/* Quad::create_triangles :244-253 — t1 = (v1,v2,v3), t2 = (v1,v4,v3); uv corners (0,0)(1,0)(1,1)(0,1) */
static void quad_make(tri_t out[2], v3 p1, v3 p2, v3 p3, v3 p4)
{
    out[0] = tri_make(p1, p2, p3);
    out[0].tu[0] = 0; out[0].tv[0] = 0; out[0].tu[1] = 1; out[0].tv[1] = 0; out[0].tu[2] = 1; out[0].tv[2] = 1;
    out[1] = tri_make(p1, p4, p3);
    out[1].tu[0] = 0; out[1].tv[0] = 0; out[1].tu[1] = 0; out[1].tv[1] = 1; out[1].tu[2] = 1; out[1].tv[2] = 1;
}

static inline hit_t hit_miss(void) { hit_t h; memset(&h, 0, sizeof h); h.hits = 0; h.dist = ORC_INF; return h; }

/* Sphere::hit :40-79 — near root only, accepted when > 1e-6 */
static inline hit_t sphere_hit(const object_t *o, const ray_t *ray, orc_stats *st)
{
    st->sphere_tests++;
    v3 c_min_q = v3_sub(o->center, ray->origin);
    float a = v3_dot(ray->direction, ray->direction);
    float b = v3_dot(ray->direction, c_min_q) * (-2);
    float c = v3_dot(c_min_q, c_min_q) - o->radius * o->radius;
    float disc = b * b - 4 * a * c;
    hit_t h = hit_miss();
    if (disc >= 0) {
        float dist = (-b - sqrtf(disc)) / (2 * a);
        if (dist > ORC_EPS) {
            st->sphere_hits++;
            v3 hp = v3_add(v3_scale(ray->direction, dist), ray->origin);   /* Ray::get_pos src/ray.cu:63-65 */
            h.hits = 1;
            h.dist = dist;
            h.point = hp;
            h.normal = v3_normalised(v3_sub(hp, o->center));
            /* assign_texture_coords :82-97 (asin / acos) is a pure function of the hit point; it
             * is evaluated for the winning sphere in rt_oracle_core.inc, where the math binding is */
        }
    }
    return h;
}

/* Triangle::hit :135-163 — Moller-Trumbore, two-sided, no early out */
static inline hit_t tri_hit(const tri_t *t, const ray_t *ray, int need_uv, orc_stats *st)
{
    st->tri_tests++;
    v3 p_vec = v3_cross(ray->direction, t->side2);
    float det = v3_dot(t->side1, p_vec);
    float inv_det = 1 / det;
    v3 t_vec = v3_sub(ray->origin, t->p[0]);
    float u = v3_dot(t_vec, p_vec) * inv_det;
    v3 q_vec = v3_cross(t_vec, t->side1);
    float v = v3_dot(ray->direction, q_vec) * inv_det;
    float w = 1 - u - v;
    float dist = v3_dot(t->side2, q_vec) * inv_det;
    int hits = dist > ORC_EPS && u >= 0 && v >= 0 && w >= 0;
    hit_t h;
    h.hits = hits;
    h.dist = dist * hits + ORC_INF_I * (1 - hits);                       /* :156 */
    h.point = v3_add(v3_scale(ray->direction, dist), ray->origin);
    h.normal = v3_scale(t->normal, (float)(1 - 2 * (v3_dot(t->normal, ray->direction) > 0)));   /* :158 */
    h.u = 0; h.v = 0;
    if (need_uv) {
        /* :160, :196-199 — called as (w, u, v): uv = tp0*w + tp1*u + tp2*v */
        h.u = t->tu[0] * w + t->tu[1] * u + t->tu[2] * v;
        h.v = t->tv[0] * w + t->tv[1] * u + t->tv[2] * v;
    }
    return h;
}

/* Quad::hit :223-236 — t1 if it hits (whatever t2's distance), else t2 */
static inline hit_t quad_hit(const tri_t q[2], const ray_t *ray, int need_uv, orc_stats *st)
{
    hit_t h1 = tri_hit(&q[0], ray, need_uv, st);
    hit_t h2 = tri_hit(&q[1], ray, need_uv, st);
    return h1.hits ? h1 : h2;
}

/* BoundingBox::grow :369-393 */
static void box_grow(box_t *b, const tri_t *t)
{
    for (int i = 0; i < 3; i++) {
        v3 p = t->p[i];
        if (!b->assigned) { b->bl = p; b->tr = p; b->assigned = 1; continue; }
        b->bl.x = fminf(b->bl.x, p.x); b->bl.y = fminf(b->bl.y, p.y); b->bl.z = fminf(b->bl.z, p.z);
        b->tr.x = fmaxf(b->tr.x, p.x); b->tr.y = fmaxf(b->tr.y, p.y); b->tr.z = fmaxf(b->tr.z, p.z);
    }
    b->width = b->tr.x - b->bl.x;
    b->height = b->tr.y - b->bl.y;
    b->depth = b->tr.z - b->bl.z;
}

/* CUDA's min/max on floats are fminf/fmaxf: a NaN operand is dropped */
static inline float nmin(float a, float b) { return (b != b) ? a : ((a != a) ? b : (a < b ? a : b)); }
static inline float nmax(float a, float b) { return (b != b) ? a : ((a != a) ? b : (a > b ? a : b)); }

/* BoundingBox::ray_hits :404-434 — slab test, strict tmin < tmax */
static inline int box_hit(const box_t *b, const ray_t *ray, float *dist, orc_stats *st)
{
    st->box_tests++;
    float tmin = 0, tmax = ORC_INF;
    float t1 = (b->bl.x - ray->origin.x) * ray->direction_inv.x;
    float t2 = (b->tr.x - ray->origin.x) * ray->direction_inv.x;
    tmin = nmax(tmin, nmin(t1, t2)); tmax = nmin(tmax, nmax(t1, t2));
    t1 = (b->bl.y - ray->origin.y) * ray->direction_inv.y;
    t2 = (b->tr.y - ray->origin.y) * ray->direction_inv.y;
    tmin = nmax(tmin, nmin(t1, t2)); tmax = nmin(tmax, nmax(t1, t2));
    t1 = (b->bl.z - ray->origin.z) * ray->direction_inv.z;
    t2 = (b->tr.z - ray->origin.z) * ray->direction_inv.z;
    tmin = nmax(tmin, nmin(t1, t2)); tmax = nmin(tmax, nmax(t1, t2));
    *dist = tmin;
    return tmin < tmax && tmax > 0;
}

/* BVH::check_leaf_node :586-600 */
static inline hit_t leaf_hit(const mesh_t *m, const node_t *n, const ray_t *ray, int need_uv, orc_stats *st)
{
    hit_t closest = hit_miss();
    for (int i = 0; i < n->ntris; i++) {
        hit_t h = tri_hit(&m->tris[n->tri_idx[i]], ray, need_uv, st);
        if (h.hits && h.dist < closest.dist) closest = h;
    }
    return closest;
}

/* BVH::traverse :487-532 — explicit 32-entry stack (src/utils.cu:188-217); each node's box is
 * tested when it is considered as a child AND again when it is popped; of two pushed
 * children the one pushed LAST (the farther one when l_first) is visited first. */
static inline hit_t mesh_hit(const mesh_t *m, const ray_t *ray, int need_uv, orc_stats *st)
{
    int stack[32];
    int top = -1;
    hit_t best = hit_miss();
    stack[++top] = m->root;
    while (top != -1) {
        int cur = stack[top--];
        const node_t *n = &m->nodes[cur];
        float d;
        int bh = box_hit(&n->box, ray, &d, st);
        if (!bh || d > best.dist) continue;
        int l = n->left, r = n->right;
        if (l == -1 && r == -1) {
            hit_t lh = leaf_hit(m, n, ray, need_uv, st);
            if (lh.hits && lh.dist < best.dist) best = lh;
            continue;
        }
        float ld, rd;
        int lhit = box_hit(&m->nodes[l].box, ray, &ld, st);
        int rhit = box_hit(&m->nodes[r].box, ray, &rd, st);
        int l_push = lhit && ld < best.dist;
        int r_push = rhit && rd < best.dist;
        int l_first = ld < rd;
        if (l_first) {
            if (l_push) stack[++top] = l;
            if (r_push) stack[++top] = r;
        } else {
            if (r_push) stack[++top] = r;
            if (l_push) stack[++top] = l;
        }
    }
    return best;
}

/* Object::hit :827-842 and the per-shape rules */
static inline hit_t object_hit(const object_t *o, const ray_t *ray, orc_stats *st)
{
    int need_uv = o->mat.need_uv;
    switch (o->type) {
        case OBJ_SPHERE: return sphere_hit(o, ray, st);
        case OBJ_TRIANGLE: return tri_hit(&o->tri, ray, need_uv, st);
        case OBJ_QUAD: return quad_hit(o->quad, ray, need_uv, st);
        case OBJ_ONE_WAY_QUAD:                                            /* :273-280 */
            if (v3_dot(ray->direction, o->owq_normal) < 0) return hit_miss();
            return quad_hit(o->quad, ray, need_uv, st);
        case OBJ_CUBOID: {                                                /* :305-322 */
            hit_t best = hit_miss();
            for (int i = 0; i < 6; i++) {
                hit_t f = quad_hit(&o->cub[2 * i], ray, need_uv, st);
                if (f.hits && f.dist < best.dist) best = f;
            }
            return best;
        }
        case OBJ_MESH: return mesh_hit(&o->mesh, ray, need_uv, st);
    }
    return hit_miss();
}

/* get_ray_collision src/raytracer.cu:24-46 — `<=`: the later object wins ties; the
 * precision_error test is a no-op for accepted hits (App. A.6) but is evaluated as written. */
static inline hit_t scene_collision(const orc_scene *s, const ray_t *ray, const object_t **obj, orc_stats *st)
{
    hit_t best = hit_miss();
    for (int i = 0; i < s->nobjs; i++) {
        hit_t h = object_hit(&s->objs[i], ray, st);
        if (!h.hits) continue;
        int closest = h.dist <= best.dist;
        int precision_error = (float)(-ORC_EPS < h.dist) < ORC_EPS;      /* (bool)(-eps<d) < eps */
        if (closest && !precision_error) { best = h; *obj = &s->objs[i]; }
    }
    return best;
}

/* ------------------------------------------------------------------------------------------
 * The hot path, instantiated for both math bindings
 * ---------------------------------------------------------------------------------------- */
#define ORC_SUFFIX _libm
#define ORC_LOGF(x) logf(x)
#define ORC_COSF(x) cosf(x)
#define ORC_SINF(x) sinf(x)
#define ORC_ASINF(x) asinf(x)
#define ORC_ACOSF(x) acosf(x)
#define ORC_ASIN(x) asin(x)
#define ORC_ACOS(x) acos(x)
#define ORC_POW5(x) pow((x), 5.0)
#include "rt_oracle_core.inc"
#undef ORC_SUFFIX
#undef ORC_LOGF
#undef ORC_COSF
#undef ORC_SINF
#undef ORC_ASINF
#undef ORC_ACOSF
#undef ORC_ASIN
#undef ORC_ACOS
#undef ORC_POW5

#define ORC_SUFFIX _det
#define ORC_LOGF(x) rt_logf(x)
#define ORC_COSF(x) rt_cosf(x)
#define ORC_SINF(x) rt_sinf(x)
#define ORC_ASINF(x) rt_asinf(x)
#define ORC_ACOSF(x) rt_acosf(x)
#define ORC_ASIN(x) rt_asin(x)
#define ORC_ACOS(x) rt_acos(x)
#define ORC_POW5(x) rt_pow5(x)
#include "rt_oracle_core.inc"
#undef ORC_SUFFIX
#undef ORC_LOGF
#undef ORC_COSF
#undef ORC_SINF
#undef ORC_ASINF
#undef ORC_ACOSF
#undef ORC_ASIN
#undef ORC_ACOS
#undef ORC_POW5

/* ------------------------------------------------------------------------------------------
 * BVH build — src/objects.cu:602-739 (host)
 * ---------------------------------------------------------------------------------------- */
typedef struct { int idx; float key; } keyed_t;

/* sort_triangles :655-706 — top-down merge sort; on equal keys the RIGHT run's element is
 * emitted first (strict `<` at :693) */
static void merge_sort(keyed_t *a, int len)
{
    if (len <= 1) return;
    int mid = len / 2;
    keyed_t *left = malloc(sizeof(keyed_t) * (size_t)mid);
    keyed_t *right = malloc(sizeof(keyed_t) * (size_t)(len - mid));
    memcpy(left, a, sizeof(keyed_t) * (size_t)mid);
    memcpy(right, a + mid, sizeof(keyed_t) * (size_t)(len - mid));
    merge_sort(left, mid);
    merge_sort(right, len - mid);
    int li = 0, ri = 0;
    for (int k = 0; k < len; k++) {
        int add_left;
        if (li >= mid) add_left = 0;
        else if (ri >= len - mid) add_left = 1;
        else add_left = left[li].key < right[ri].key;
        a[k] = add_left ? left[li++] : right[ri++];
    }
    free(left);
    free(right);
}

/* get_ref_point :708-719 */
static v3 box_ref_point(const box_t *b)
{
    if (b->width >= b->height && b->width >= b->depth)
        return v3_make(b->bl.x + b->width / 2, b->bl.y, b->bl.z + b->depth / 2);
    else if (b->height >= b->width && b->height >= b->depth)
        return v3_make(b->bl.x, b->bl.y + b->height / 2, b->bl.z + b->depth / 2);
    else
        return v3_make(b->bl.x + b->width / 2, b->bl.y + b->height / 2, b->bl.z);
}

static int mesh_add_node(mesh_t *m, const box_t *box, int left, int right, const int *idx, int n)
{   /* add_tree_node :721-739 */
    if (m->nnodes == m->cap) { m->cap = m->cap ? m->cap * 2 : 2048; m->nodes = realloc(m->nodes, sizeof(node_t) * (size_t)m->cap); }
    node_t *nd = &m->nodes[m->nnodes];
    nd->box = *box;
    nd->left = left; nd->right = right;
    nd->ntris = n;
    nd->tri_idx = malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    if (n > 0) memcpy(nd->tri_idx, idx, sizeof(int) * (size_t)n);
    return m->nnodes++;
}

/* build :602-624 — post-order numbering, fixed depth, split_triangles :626-653 */
static int mesh_build(mesh_t *m, const int *idx, int n, int depth)
{
    box_t box;
    memset(&box, 0, sizeof box);      /* BoundingBox() :364-367: corners (0,0,0); dims unset -> 0 */
    for (int i = 0; i < n; i++) box_grow(&box, &m->tris[idx[i]]);
    if (depth <= 0) return mesh_add_node(m, &box, -1, -1, idx, n);

    v3 ref = box_ref_point(&box);
    keyed_t *keyed = malloc(sizeof(keyed_t) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        keyed[i].idx = idx[i];
        keyed[i].key = v3_magnitude(v3_sub(m->tris[idx[i]].p[0], ref));
    }
    merge_sort(keyed, n);
    int mid = n / 2;
    int nl = 0, nr = 0;
    int *li = malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int *ri = malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        if (i <= mid) li[nl++] = keyed[i].idx; else ri[nr++] = keyed[i].idx;   /* :645 */
    }
    free(keyed);
    int l = mesh_build(m, li, nl, depth - 1);
    int r = mesh_build(m, ri, nr, depth - 1);
    free(li);
    free(ri);
    return mesh_add_node(m, &box, l, r, idx, n);
}

/* ------------------------------------------------------------------------------------------
 * Scene construction — Object::create_* src/objects.cu:845-906
 * ---------------------------------------------------------------------------------------- */
orc_scene *orc_scene_new(int math_mode)
{
    orc_scene *s = calloc(1, sizeof *s);
    s->math_mode = math_mode;
    return s;
}

void orc_scene_free(orc_scene *s)
{
    if (!s) return;
    for (int i = 0; i < s->nobjs; i++) {
        mesh_t *m = &s->objs[i].mesh;
        for (int k = 0; k < m->nnodes; k++) free(m->nodes[k].tri_idx);
        free(m->nodes);
        free(m->tris);
    }
    free(s->objs);
    free(s);
}

int orc_scene_num_objects(const orc_scene *s) { return s->nobjs; }

static object_t *scene_new_object(orc_scene *s, int type, const orc_material *m)
{
    if (m->tex_type == ORC_TEX_IMAGE && m->need_uv && !m->img_rgb) { fprintf(stderr, "rt_oracle: IMAGE texture without data\n"); abort(); }
    if (s->nobjs == s->cap) { s->cap = s->cap ? s->cap * 2 : 16; s->objs = realloc(s->objs, sizeof(object_t) * (size_t)s->cap); }
    object_t *o = &s->objs[s->nobjs++];
    memset(o, 0, sizeof *o);
    o->type = type;
    o->mat = *m;
    return o;
}

void orc_add_sphere(orc_scene *s, const float c[3], float r, const orc_material *m)
{
    object_t *o = scene_new_object(s, OBJ_SPHERE, m);
    o->center = v3_from(c);
    o->radius = r;
}

void orc_add_triangle(orc_scene *s, const float p1[3], const float p2[3], const float p3[3], const orc_material *m)
{
    object_t *o = scene_new_object(s, OBJ_TRIANGLE, m);
    o->tri = tri_make(v3_from(p1), v3_from(p2), v3_from(p3));
}

void orc_add_triangle_uv(orc_scene *s, const float p[9], const float uv[6], const orc_material *m)
{   /* Triangle(Vertex, Vertex, Vertex, Material) :120-133 */
    object_t *o = scene_new_object(s, OBJ_TRIANGLE, m);
    o->tri = tri_make(v3_from(p), v3_from(p + 3), v3_from(p + 6));
    for (int i = 0; i < 3; i++) { o->tri.tu[i] = uv[2 * i]; o->tri.tv[i] = uv[2 * i + 1]; }
}

void orc_add_quad(orc_scene *s, const float p1[3], const float p2[3], const float p3[3], const float p4[3], const orc_material *m)
{
    object_t *o = scene_new_object(s, OBJ_QUAD, m);
    quad_make(o->quad, v3_from(p1), v3_from(p2), v3_from(p3), v3_from(p4));
}

void orc_add_one_way_quad(orc_scene *s, const float p1[3], const float p2[3], const float p3[3], const float p4[3], int invert_normal, const orc_material *m)
{   /* :261-289 */
    object_t *o = scene_new_object(s, OBJ_ONE_WAY_QUAD, m);
    quad_make(o->quad, v3_from(p1), v3_from(p2), v3_from(p3), v3_from(p4));
    int multiplier = 1 - 2 * (invert_normal != 0);
    o->owq_normal = v3_scale(o->quad[0].normal, (float)multiplier);
}

void orc_add_cuboid(orc_scene *s, const float tl_near_p[3], float width, float height, float depth, const orc_material *m)
{   /* Cuboid::create_faces :327-349 */
    object_t *o = scene_new_object(s, OBJ_CUBOID, m);
    v3 tl_near = v3_from(tl_near_p);
    v3 w = v3_make(width, 0, 0), h = v3_make(0, height, 0), d = v3_make(0, 0, depth);
    v3 tr_near = v3_add(tl_near, w);
    v3 br_near = v3_sub(tr_near, h);
    v3 bl_near = v3_sub(tl_near, h);
    v3 tl_far = v3_add(tl_near, d);
    v3 tr_far = v3_add(tl_far, w);
    v3 br_far = v3_sub(tr_far, h);
    v3 bl_far = v3_sub(tl_far, h);
    quad_make(&o->cub[0], tl_near, tr_near, br_near, bl_near);   /* front */
    quad_make(&o->cub[2], tl_far, tr_far, br_far, bl_far);       /* back */
    quad_make(&o->cub[4], tl_near, bl_near, bl_far, tl_far);     /* left */
    quad_make(&o->cub[6], tr_near, br_near, br_far, tr_far);     /* right */
    quad_make(&o->cub[8], bl_near, br_near, br_far, bl_far);     /* bottom */
    quad_make(&o->cub[10], tl_near, tr_near, tr_far, tl_far);    /* top */
}

static void mesh_finish(object_t *o)
{   /* Mesh :780-787: BVH(host_triangles, device_triangles, 10) */
    mesh_t *m = &o->mesh;
    int *idx = malloc(sizeof(int) * (size_t)(m->ntris > 0 ? m->ntris : 1));
    for (int i = 0; i < m->ntris; i++) idx[i] = i;
    m->root = mesh_build(m, idx, m->ntris, 10);
    free(idx);
}

void orc_add_mesh(orc_scene *s, const float *tris, int n, const orc_material *m)
{
    object_t *o = scene_new_object(s, OBJ_MESH, m);
    o->mesh.ntris = n;
    o->mesh.tris = malloc(sizeof(tri_t) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) o->mesh.tris[i] = tri_make(v3_from(tris + 9 * i), v3_from(tris + 9 * i + 3), v3_from(tris + 9 * i + 6));
    mesh_finish(o);
}

int orc_mesh_bvh_info(const orc_scene *s, int object_index, int *num_nodes, int *hist, int hist_len)
{
    if (object_index < 0 || object_index >= s->nobjs || s->objs[object_index].type != OBJ_MESH) return -1;
    const mesh_t *m = &s->objs[object_index].mesh;
    *num_nodes = m->nnodes;
    for (int i = 0; i < hist_len; i++) hist[i] = 0;
    for (int i = 0; i < m->nnodes; i++) {
        if (m->nodes[i].left == -1 && m->nodes[i].right == -1) {
            int k = m->nodes[i].ntris < hist_len - 1 ? m->nodes[i].ntris : hist_len - 1;
            hist[k]++;
        }
    }
    return 0;
}

int orc_trace_one(const orc_scene *s, const float origin[3], const float dir[3], float out[8])
{
    orc_stats st; memset(&st, 0, sizeof st);
    ray_t ray; memset(&ray, 0, sizeof ray);
    ray.origin = v3_from(origin);
    ray_change_direction(&ray, v3_from(dir));
    const object_t *ob = NULL;
    hit_t h = scene_collision(s, &ray, &ob, &st);
    out[0] = h.dist;
    out[1] = h.point.x; out[2] = h.point.y; out[3] = h.point.z;
    out[4] = h.normal.x; out[5] = h.normal.y; out[6] = h.normal.z;
    out[7] = -1;
    if (h.hits) out[7] = (float)(ob - s->objs);
    return h.hits;
}

/* ------------------------------------------------------------------------------------------
 * Host matrices — src/matrix.cu (float, naive triple loop, sum starts at 0)
 * ---------------------------------------------------------------------------------------- */
static void mat3_mul(const float a[3][3], const float b[3][3], float out[3][3])
{   /* Matrix::operator* :29-51 */
    float tmp[3][3];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            float sum = 0;
            for (int i = 0; i < 3; i++) sum += a[r][i] * b[i][c];
            tmp[r][c] = sum;
        }
    memcpy(out, tmp, sizeof tmp);
}

static float m_sin(float x, int mode) { return mode == ORC_MATH_DET ? rt_sinf(x) : sinf(x); }
static float m_cos(float x, int mode) { return mode == ORC_MATH_DET ? rt_cosf(x) : cosf(x); }
static float m_tan(float x, int mode) { return mode == ORC_MATH_DET ? rt_tanf(x) : tanf(x); }

/* RotationMatrix :99-150 */
static void mat3_rotation(int axis, float angle, int mode, float m[3][3])
{
    float s = m_sin(angle, mode), c = m_cos(angle, mode);
    if (axis == 0)      { float t[3][3] = {{1, 0, 0}, {0, c, s}, {0, -s, c}}; memcpy(m, t, sizeof t); }
    else if (axis == 1) { float t[3][3] = {{c, 0, -s}, {0, 1, 0}, {s, 0, c}}; memcpy(m, t, sizeof t); }
    else                { float t[3][3] = {{c, -s, 0}, {s, c, 0}, {0, 0, 1}}; memcpy(m, t, sizeof t); }
}

/* (3x3) * (3xN) with the same loop order as Matrix::operator* */
static void mat3_apply(const float m[3][3], float *vx, float *vy, float *vz, int n)
{
    for (int c = 0; c < n; c++) {
        float in[3] = {vx[c], vy[c], vz[c]};
        float o[3];
        for (int r = 0; r < 3; r++) {
            float sum = 0;
            for (int i = 0; i < 3; i++) sum += m[r][i] * in[i];
            o[r] = sum;
        }
        vx[c] = o[0]; vy[c] = o[1]; vz[c] = o[2];
    }
}

/* ------------------------------------------------------------------------------------------
 * .obj loader — src/obj_read.cu
 * ---------------------------------------------------------------------------------------- */
struct orc_obj {
    int math_mode;
    int nverts; float *vx, *vy, *vz;      /* vertex_mat rows (3 x N) */
    int nfaces; int *face_arity; int *face_start; int *face_idx; int nidx;
};

orc_obj *orc_obj_load(const char *path, int math_mode)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;                  /* read_file :8-10 throws "Could not find file to open." */
    orc_obj *o = calloc(1, sizeof *o);
    o->math_mode = math_mode;
    int vcap = 0, fcap = 0, icap = 0;
    char *line = NULL; size_t cap = 0; ssize_t len;
    while ((len = getline(&line, &cap, f)) >= 0) {
        while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
        /* split_string(line, ' ') :27-44 — token 0 is the text before the first space */
        if (line[0] == 'v' && line[1] == ' ') {
            /* read_vertices :92-112: stof on tokens 1..3 (tokens carry a leading space, which stof skips) */
            if (o->nverts == vcap) {
                vcap = vcap ? vcap * 2 : 1024;
                o->vx = realloc(o->vx, 4 * (size_t)vcap); o->vy = realloc(o->vy, 4 * (size_t)vcap); o->vz = realloc(o->vz, 4 * (size_t)vcap);
            }
            char *p = line + 2, *e;
            o->vx[o->nverts] = strtof(p, &e); p = e;
            o->vy[o->nverts] = strtof(p, &e); p = e;
            o->vz[o->nverts] = strtof(p, &e);
            o->nverts++;
        } else if (line[0] == 'f' && line[1] == ' ') {
            /* extract_faces :121-147: every further token -> text before the first '/' -> stoi - 1 */
            if (o->nfaces == fcap) {
                fcap = fcap ? fcap * 2 : 1024;
                o->face_arity = realloc(o->face_arity, 4 * (size_t)fcap); o->face_start = realloc(o->face_start, 4 * (size_t)fcap);
            }
            o->face_start[o->nfaces] = o->nidx;
            int arity = 0;
            char *p = line + 1;
            while (*p) {
                while (*p == ' ') p++;
                if (!*p) break;
                int vi = (int)strtol(p, NULL, 10) - 1;
                if (o->nidx == icap) { icap = icap ? icap * 2 : 4096; o->face_idx = realloc(o->face_idx, 4 * (size_t)icap); }
                o->face_idx[o->nidx++] = vi;
                arity++;
                while (*p && *p != ' ') p++;
            }
            o->face_arity[o->nfaces++] = arity;
        }
    }
    free(line);
    fclose(f);
    return o;
}

void orc_obj_free(orc_obj *o)
{
    if (!o) return;
    free(o->vx); free(o->vy); free(o->vz); free(o->face_arity); free(o->face_start); free(o->face_idx); free(o);
}

void orc_obj_enlarge(orc_obj *o, float scale)
{   /* :59-64 — EnlargementMatrix(scale, 3) * vertex_mat */
    float m[3][3] = {{scale, 0, 0}, {0, scale, 0}, {0, 0, scale}};
    mat3_apply(m, o->vx, o->vy, o->vz, o->nverts);
}

void orc_obj_rotate(orc_obj *o, float ax, float ay, float az)
{   /* :66-76 — ((x_rot * y_rot) * z_rot) * vertex_mat */
    float rx[3][3], ry[3][3], rz[3][3], t[3][3];
    mat3_rotation(0, ax, o->math_mode, rx);
    mat3_rotation(1, ay, o->math_mode, ry);
    mat3_rotation(2, az, o->math_mode, rz);
    mat3_mul(rx, ry, t);
    mat3_mul(t, rz, t);
    mat3_apply(t, o->vx, o->vy, o->vz, o->nverts);
}

void orc_obj_translate(orc_obj *o, float dx, float dy, float dz)
{   /* :78-86 */
    for (int i = 0; i < o->nverts; i++) { o->vx[i] += dx; o->vy[i] += dy; o->vz[i] += dz; }
}

int orc_obj_num_vertices(const orc_obj *o) { return o->nverts; }
int orc_obj_num_faces(const orc_obj *o) { return o->nfaces; }
int orc_obj_face_arity(const orc_obj *o, int face) { return o->face_arity[face]; }

void orc_obj_vertices(const orc_obj *o, float *out)
{
    for (int i = 0; i < o->nverts; i++) { out[3 * i] = o->vx[i]; out[3 * i + 1] = o->vy[i]; out[3 * i + 2] = o->vz[i]; }
}

int orc_obj_num_triangles(const orc_obj *o)
{
    int n = 0;
    for (int i = 0; i < o->nfaces; i++) {
        if (o->face_arity[i] == 3) n += 1;
        else if (o->face_arity[i] == 4) n += 2;
        else return -1;                    /* src/main.cu:141 throws std::logic_error */
    }
    return n;
}

static v3 obj_vertex(const orc_obj *o, int vi) { return v3_make(o->vx[vi], o->vy[vi], o->vz[vi]); }

void orc_obj_triangles(const orc_obj *o, float *out)
{   /* SceneObjects::create_mesh src/main.cu:127-148: quad -> (v1,v2,v3) + (v1,v4,v3) */
    int k = 0;
    for (int i = 0; i < o->nfaces; i++) {
        const int *fi = o->face_idx + o->face_start[i];
        int order[2][3] = {{0, 1, 2}, {0, 3, 2}};
        int nt = o->face_arity[i] == 4 ? 2 : 1;
        for (int t = 0; t < nt; t++)
            for (int j = 0; j < 3; j++) {
                v3 p = obj_vertex(o, fi[order[t][j]]);
                out[k++] = p.x; out[k++] = p.y; out[k++] = p.z;
            }
    }
}

int orc_add_mesh_faces(orc_scene *s, const orc_obj *o, const orc_material *m)
{
    int n = orc_obj_num_triangles(o);
    if (n < 0) return -1;
    object_t *ob = scene_new_object(s, OBJ_MESH, m);
    ob->mesh.ntris = n;
    ob->mesh.tris = malloc(sizeof(tri_t) * (size_t)(n > 0 ? n : 1));
    int k = 0;
    for (int i = 0; i < o->nfaces; i++) {
        const int *fi = o->face_idx + o->face_start[i];
        if (o->face_arity[i] == 3) {
            ob->mesh.tris[k++] = tri_make(obj_vertex(o, fi[0]), obj_vertex(o, fi[1]), obj_vertex(o, fi[2]));
        } else {
            quad_make(&ob->mesh.tris[k], obj_vertex(o, fi[0]), obj_vertex(o, fi[1]), obj_vertex(o, fi[2]), obj_vertex(o, fi[3]));
            k += 2;
        }
    }
    mesh_finish(ob);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Camera — src/camera.cu:34-108 (pose constants of :34-41; W, H as parameters)
 * ---------------------------------------------------------------------------------------- */
static v3 cam_rotate_point(v3 p, float xr, float yr, float zr, int mode)
{   /* rotate_point :63-69: Rx * Ry * Rz * p, left to right */
    float rx[3][3], ry[3][3], rz[3][3], t[3][3];
    mat3_rotation(0, xr, mode, rx);
    mat3_rotation(1, yr, mode, ry);
    mat3_rotation(2, zr, mode, rz);
    mat3_mul(rx, ry, t);
    mat3_mul(t, rz, t);
    float x = p.x, y = p.y, z = p.z;
    mat3_apply(t, &x, &y, &z, 1);
    return v3_make(x, y, z);
}

void orc_camera_default(int W, int H, int math_mode, float out[12])
{
    const float PI = 3.141592653589793f;                 /* :9 (float constant) */
    const v3 CAM_POS = {0, 0, 0};
    const float FOV = 60 * (PI / 180);
    const float FOCAL_LEN = 0.1f;
    const float X_ROT = 0 * (PI / 180), Y_ROT = 0 * (PI / 180), Z_ROT = 0 * (PI / 180);
    const float ASPECT = (float)W / (float)H;            /* :7 */

    float viewport_width = 2 * FOCAL_LEN * m_tan(FOV / 2, math_mode);   /* :47 */
    float viewport_height = viewport_width / ASPECT;
    /* get_u :71-83, get_v :85-97 */
    v3 u = v3_sub(cam_rotate_point(v3_make(1, 0, 0), X_ROT, Y_ROT, Z_ROT, math_mode), v3_make(0, 0, 0));
    v3 delta_u = v3_set_mag(u, viewport_width / (float)W);
    v3 v = v3_sub(cam_rotate_point(v3_make(0, -1, 0), X_ROT, Y_ROT, Z_ROT, math_mode), v3_make(0, 0, 0));
    v3 delta_v = v3_set_mag(v, viewport_height / (float)H);
    v3 plane_normal = v3_normalised(v3_cross(delta_v, delta_u));        /* :53 */
    /* get_tl_pos :99-108 */
    v3 u_step = v3_div(v3_scale(delta_u, (float)(-W)), 2.0f);
    v3 v_step = v3_div(v3_scale(delta_v, (float)(-H)), 2.0f);
    v3 focal = v3_add(v3_scale(plane_normal, FOCAL_LEN), CAM_POS);
    v3 tl = v3_add(v3_add(u_step, v_step), focal);
    out[0] = CAM_POS.x; out[1] = CAM_POS.y; out[2] = CAM_POS.z;
    out[3] = tl.x; out[4] = tl.y; out[5] = tl.z;
    out[6] = delta_u.x; out[7] = delta_u.y; out[8] = delta_u.z;
    out[9] = delta_v.x; out[10] = delta_v.y; out[11] = delta_v.z;
}

/* ------------------------------------------------------------------------------------------
 * Render driver — rows handed to worker threads in bands of 8 (dynamic)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const orc_scene *s; const float *cam; int W, H; rsettings_t rs;
    int32_t time_ms, frame_num; int y0, y1;
    const float *prev; float *out;
    atomic_int next_row;
    orc_stats stats; pthread_mutex_t lock;
} job_t;

static void *render_worker(void *arg)
{
    job_t *j = arg;
    orc_stats st; memset(&st, 0, sizeof st);
    for (;;) {
        int y = atomic_fetch_add(&j->next_row, 8);
        if (y >= j->y1) break;
        int ye = y + 8 < j->y1 ? y + 8 : j->y1;
        for (int yy = y; yy < ye; yy++)
            for (int x = 0; x < j->W; x++) {
                if (j->s->math_mode == ORC_MATH_DET) pixel_det(j->s, j->cam, j->W, x, yy, &j->rs, j->time_ms, j->frame_num, j->prev, j->out, &st);
                else pixel_libm(j->s, j->cam, j->W, x, yy, &j->rs, j->time_ms, j->frame_num, j->prev, j->out, &st);
            }
    }
    pthread_mutex_lock(&j->lock);
    uint64_t *d = (uint64_t *)&j->stats; const uint64_t *a = (const uint64_t *)&st;
    for (size_t i = 0; i < sizeof(orc_stats) / 8; i++) d[i] += a[i];
    pthread_mutex_unlock(&j->lock);
    return NULL;
}

void orc_render(const orc_scene *s, const float cam[12], int W, int H,
                int rays_per_pixel, int reflection_limit, int antialias, const float sky[3],
                int32_t time_ms, int32_t frame_num, int y0, int y1,
                const float *prev, float *out, int nthreads, orc_stats *stats)
{
    job_t j;
    memset(&j, 0, sizeof j);
    j.s = s; j.cam = cam; j.W = W; j.H = H;
    j.rs.rays_per_pixel = rays_per_pixel; j.rs.reflection_limit = reflection_limit; j.rs.antialias = antialias;
    j.rs.sky = v3_from(sky);
    j.time_ms = time_ms; j.frame_num = frame_num;
    j.y0 = y0 < 0 ? 0 : y0; j.y1 = y1 > H ? H : y1;
    j.prev = prev; j.out = out;
    atomic_init(&j.next_row, j.y0);
    pthread_mutex_init(&j.lock, NULL);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    for (int i = 1; i < nthreads; i++) pthread_create(&th[i], NULL, render_worker, &j);
    render_worker(&j);
    for (int i = 1; i < nthreads; i++) pthread_join(th[i], NULL);
    pthread_mutex_destroy(&j.lock);
    if (stats) *stats = j.stats;
}

/* ------------------------------------------------------------------------------------------
 * Known-answer helpers
 * ---------------------------------------------------------------------------------------- */
float orc_pcg_next(uint32_t *state) { return pcg_next(state); }

float orc_normal_next(uint32_t *state, int math_mode)
{
    orc_stats st; memset(&st, 0, sizeof st);
    return math_mode == ORC_MATH_DET ? normal_num_det(state, &st) : normal_num_libm(state, &st);
}

float orc_math_logf(float x, int mode) { return mode == ORC_MATH_DET ? rt_logf(x) : logf(x); }
float orc_math_cosf(float x, int mode) { return m_cos(x, mode); }
float orc_math_sinf(float x, int mode) { return m_sin(x, mode); }
float orc_math_tanf(float x, int mode) { return m_tan(x, mode); }
double orc_math_asin(double x, int mode) { return mode == ORC_MATH_DET ? rt_asin(x) : asin(x); }
double orc_math_acos(double x, int mode) { return mode == ORC_MATH_DET ? rt_acos(x) : acos(x); }

void orc_to_rgba8(const float *rgb, int W, int H, uint8_t *out)
{   /* parse_pixel_colours src/main.cu:343-371: int(px*255), clamp 0..255, alpha 255 */
    for (int i = 0; i < W * H; i++) {
        for (int c = 0; c < 3; c++) {
            int colour = rt_f2i(rgb[3 * i + c] * 255);
            if (colour > 255) colour = 255; else if (colour < 0) colour = 0;
            out[4 * i + c] = (uint8_t)colour;
        }
        out[4 * i + 3] = 255;
    }
}
