/*
 * rt_host.cpp — host-side producers of the kernel's inputs: materials, object factories,
 * the .obj loader with its transforms, the camera, the BVH builder and the flattener that
 * emits the compact device layout (rt_device_scene.h).
 *
 * These mirror the reference's host code (file:line cited per function, relative to the
 * reference checkout) in float arithmetic with the same operation order, because their
 * outputs are kernel inputs and a 1-ulp difference there moves pixels (hard
 * part 1).  sin/cos/tan come from rt_math.h, not libm, so the same inputs give the same
 * bits on every host.  Built with -ffp-contract=off.
 */
#include "rt_host.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>

#include "rt_math.h"

namespace {

std::vector<std::string> split_string(const std::string &str, char split_char);

struct F3 {
    float x, y, z;
};

inline F3 f3(float x, float y, float z) { return F3{x, y, z}; }
inline F3 f3(const float *p) { return F3{p[0], p[1], p[2]}; }
inline F3 add(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline F3 sub(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline F3 scale(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
inline F3 divide(F3 a, float s) { return f3(a.x / s, a.y / s, a.z / s); }
inline F3 cross(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float magnitude(F3 a) { float m = a.x * a.x + a.y * a.y + a.z * a.z; return std::sqrt(m); }       /* src/utils.cu:118-121 */
inline F3 normalised(F3 a) { float inv = 1 / magnitude(a); return f3(a.x * inv, a.y * inv, a.z * inv); } /* :123-128 */
inline F3 set_mag(F3 a, float m) { float s = m / magnitude(a); return f3(a.x * s, a.y * s, a.z * s); }   /* :155-162 */
inline void store(float *dst, F3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

/* Triangle::precompute src/objects.cu:175-186 */
HostTri make_tri(F3 a, F3 b, F3 c)
{
    HostTri t{};
    F3 s1 = sub(b, a), s2 = sub(c, a);
    store(t.p0, a);
    store(t.s1, s1);
    store(t.s2, s2);
    store(t.n, normalised(cross(s1, s2)));
    return t;
}

/* Quad::create_triangles src/objects.cu:244-253 */
void make_quad(std::vector<HostTri> &out, F3 p1, F3 p2, F3 p3, F3 p4)
{
    HostTri t1 = make_tri(p1, p2, p3);
    const float uv1[6] = {0, 0, 1, 0, 1, 1};
    std::memcpy(t1.uv, uv1, sizeof uv1);
    HostTri t2 = make_tri(p1, p4, p3);
    const float uv2[6] = {0, 0, 0, 1, 1, 1};
    std::memcpy(t2.uv, uv2, sizeof uv2);
    out.push_back(t1);
    out.push_back(t2);
}

rt_status fail(rt_scene_builder *b, rt_status code, const char *msg)
{
    if (b) b->err = msg;
    return code;
}

void keep_texels(HostObject &o)
{   /* Texture::allocate_memory src/material.cu:107-117 copies the texels to the device at
     * creation; here the builder takes its own copy so the caller's array may go away */
    if (o.mat.tex_type == RT_TEX_IMAGE && o.mat.type != RT_MAT_EMISSIVE && o.mat.img_rgb)
        o.texels.assign(o.mat.img_rgb, o.mat.img_rgb + (size_t)o.mat.img_w * (size_t)o.mat.img_h * 3);
    o.mat.img_rgb = nullptr;
}

rt_status check_material(rt_scene_builder *b, const rt_material *m, bool is_sphere)
{
    if (!b || !m) return fail(b, RT_ERR_INVALID, "null argument");
    (void)is_sphere;
    if (m->type != RT_MAT_STANDARD && m->type != RT_MAT_EMISSIVE && m->type != RT_MAT_REFRACTIVE) return fail(b, RT_ERR_INVALID, "unknown material type");
    if (m->tex_type < 0 || m->tex_type > RT_TEX_IMAGE) return fail(b, RT_ERR_INVALID, "unknown texture type");
    if (m->tex_type == RT_TEX_IMAGE && m->type != RT_MAT_EMISSIVE &&
        (m->img_w <= 0 || m->img_h <= 0 || !m->img_rgb || (int64_t)m->img_w * m->img_h > (1 << 26))) return fail(b, RT_ERR_INVALID, "IMAGE texture without data");
    if (m->tex_type == RT_TEX_CHECKERBOARD && (m->num_squares < 0 || m->num_squares >= (1 << 24))) return fail(b, RT_ERR_INVALID, "num_squares out of range");
    return RT_OK;
}

/* ---------------- BVH build: reference src/objects.cu:602-719 ----------------------------- */
struct Box {
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};   /* BoundingBox() :364-367 */
};

struct ChildRef {
    uint32_t ref;
    Box box;
    int stack_need = 0;   /* deferred-sibling entries a traversal of this subtree can hold at once */
};

struct BvhBuilder {
    std::vector<rt_node> nodes;
    std::vector<int> order;      /* triangle indices in leaf order */
    std::string err;

    Box bounds(const std::vector<int> &idx, const std::vector<std::array<F3, 3>> &pts) const
    {
        Box b;
        bool assigned = false;
        for (int i : idx)
            for (int k = 0; k < 3; k++) {
                F3 p = pts[i][k];
                if (!assigned) {
                    b.lo[0] = b.hi[0] = p.x; b.lo[1] = b.hi[1] = p.y; b.lo[2] = b.hi[2] = p.z;
                    assigned = true;
                    continue;
                }
                b.lo[0] = std::min(b.lo[0], p.x); b.lo[1] = std::min(b.lo[1], p.y); b.lo[2] = std::min(b.lo[2], p.z);
                b.hi[0] = std::max(b.hi[0], p.x); b.hi[1] = std::max(b.hi[1], p.y); b.hi[2] = std::max(b.hi[2], p.z);
            }
        return b;
    }

    /* get_ref_point :708-719 */
    static F3 ref_point(const Box &b)
    {
        float width = b.hi[0] - b.lo[0], height = b.hi[1] - b.lo[1], depth = b.hi[2] - b.lo[2];
        if (width >= height && width >= depth) return f3(b.lo[0] + width / 2, b.lo[1], b.lo[2] + depth / 2);
        if (height >= width && height >= depth) return f3(b.lo[0], b.lo[1] + height / 2, b.lo[2] + depth / 2);
        return f3(b.lo[0] + width / 2, b.lo[1] + height / 2, b.lo[2]);
    }

    ChildRef build(const std::vector<int> &idx, int depth, const std::vector<std::array<F3, 3>> &pts)
    {
        ChildRef out;
        if (idx.empty()) {   /* a whole empty subtree behaves like one empty leaf with the (0,0,0) box */
            out.ref = RT_REF_EMPTY_LEAF;
            return out;
        }
        out.box = bounds(idx, pts);
        if (depth <= 0) {
            if (idx.size() > RT_REF_COUNT_MAX || order.size() + idx.size() > RT_REF_START_MASK) {
                err = "mesh too dense for the compact BVH encoding (leaf > 1023 triangles or > 1M triangles)";
                out.ref = RT_REF_EMPTY_LEAF;
                return out;
            }
            out.ref = RT_REF_LEAF | ((uint32_t)idx.size() << RT_REF_COUNT_SHIFT) | (uint32_t)order.size();
            for (int i : idx) order.push_back(i);
            return out;
        }
        /* split_triangles :626-653: key = |points[0] - ref|; sort_triangles :655-706 is a merge
         * sort that emits the RIGHT run first on equal keys, i.e. an ascending sort in which
         * equal keys come out in reverse input order */
        F3 ref = ref_point(out.box);
        struct Keyed { float key; int pos; int idx; };
        std::vector<Keyed> keyed(idx.size());
        for (size_t i = 0; i < idx.size(); i++) keyed[i] = Keyed{magnitude(sub(pts[idx[i]][0], ref)), (int)i, idx[i]};
        std::sort(keyed.begin(), keyed.end(), [](const Keyed &a, const Keyed &b) { return a.key < b.key || (a.key == b.key && a.pos > b.pos); });
        size_t mid = keyed.size() / 2;
        std::vector<int> left, right;
        for (size_t i = 0; i < keyed.size(); i++) (i <= mid ? left : right).push_back(keyed[i].idx);   /* :645 */
        ChildRef l = build(left, depth - 1, pts);
        ChildRef r = build(right, depth - 1, pts);
        if (right.empty()) {
            /* single-child node: same triangle set, hence the same box, as its left child.
             * Not stored; the edge is marked instead (see RT_REF_CHAIN). */
            l.ref |= RT_REF_CHAIN;
            return l;
        }
        rt_node n;
        n.q[0] = rt_f4{l.box.lo[0], l.box.lo[1], l.box.lo[2], l.box.hi[0]};
        n.q[1] = rt_f4{l.box.hi[1], l.box.hi[2], r.box.lo[0], r.box.lo[1]};
        n.q[2] = rt_f4{r.box.lo[2], r.box.hi[0], r.box.hi[1], r.box.hi[2]};
        n.q[3] = rt_f4{rt_u2f(l.ref), rt_u2f(r.ref), 0.0f, 0.0f};
        out.ref = (uint32_t)nodes.size();
        out.stack_need = 1 + std::max(l.stack_need, r.stack_need);
        nodes.push_back(n);
        return out;
    }
};

rt_status add_mesh_tris(rt_scene_builder *b, const std::vector<HostTri> &tris, const std::vector<std::array<F3, 3>> &pts, const rt_material *m)
{
    HostObject o;
    o.type = RT_OBJ_MESH;
    o.mat = *m;
    BvhBuilder bb;
    std::vector<int> idx(tris.size());
    for (size_t i = 0; i < tris.size(); i++) idx[i] = (int)i;
    ChildRef root = bb.build(idx, RT_BVH_DEPTH, pts);   /* Mesh :786: BVH(..., 10) */
    if (!bb.err.empty()) return fail(b, RT_ERR_UNSUPPORTED, bb.err.c_str());
    o.root_ref = root.ref;
    o.stack_need = root.stack_need;
    for (int k = 0; k < 3; k++) { o.v[k] = root.box.lo[k]; o.v[3 + k] = root.box.hi[k]; }
    o.nodes = std::move(bb.nodes);
    o.tris.reserve(tris.size());
    for (int i : bb.order) o.tris.push_back(tris[i]);
    keep_texels(o);
    b->objs.push_back(std::move(o));
    return RT_OK;
}

/* ---------------- host matrices: src/matrix.cu ------------------------------------------- */
typedef float M3[3][3];

void m3_mul(const M3 a, const M3 b, M3 out)
{   /* Matrix::operator* :29-51 */
    M3 t;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            float sum = 0;
            for (int i = 0; i < 3; i++) sum += a[r][i] * b[i][c];
            t[r][c] = sum;
        }
    std::memcpy(out, t, sizeof(M3));
}

void m3_rotation(int axis, float angle, M3 m)
{   /* RotationMatrix :99-150 */
    float s = rt_sinf(angle), c = rt_cosf(angle);
    const M3 rx = {{1, 0, 0}, {0, c, s}, {0, -s, c}};
    const M3 ry = {{c, 0, -s}, {0, 1, 0}, {s, 0, c}};
    const M3 rz = {{c, -s, 0}, {s, c, 0}, {0, 0, 1}};
    std::memcpy(m, axis == 0 ? rx : (axis == 1 ? ry : rz), sizeof(M3));
}

void m3_rotation_xyz(float ax, float ay, float az, M3 out)
{   /* x_rot * y_rot * z_rot, left to right (src/obj_read.cu:74, src/camera.cu:66) */
    M3 rx, ry, rz;
    m3_rotation(0, ax, rx);
    m3_rotation(1, ay, ry);
    m3_rotation(2, az, rz);
    m3_mul(rx, ry, out);
    m3_mul(out, rz, out);
}

F3 m3_apply(const M3 m, F3 p)
{
    float in[3] = {p.x, p.y, p.z}, o[3];
    for (int r = 0; r < 3; r++) {
        float sum = 0;
        for (int i = 0; i < 3; i++) sum += m[r][i] * in[i];
        o[r] = sum;
    }
    return f3(o[0], o[1], o[2]);
}

}  // namespace

/* ================================ C ABI: materials ========================================= */
extern "C" void rt_material_standard(rt_material *m, const float colour[3], float smoothness)
{
    std::memset(m, 0, sizeof *m);
    m->type = RT_MAT_STANDARD;
    m->tex_type = RT_TEX_COLOUR;
    std::memcpy(m->colour, colour, 12);
    m->smoothness = smoothness;
    m->need_uv = 0;
}

extern "C" void rt_material_checkerboard(rt_material *m, const float light[3], const float dark[3], int32_t num_squares, float smoothness)
{
    std::memset(m, 0, sizeof *m);
    m->type = RT_MAT_STANDARD;
    m->tex_type = RT_TEX_CHECKERBOARD;
    std::memcpy(m->light, light, 12);
    std::memcpy(m->dark, dark, 12);
    m->num_squares = num_squares;
    m->smoothness = smoothness;
    m->need_uv = 1;
}

extern "C" void rt_material_gradient(rt_material *m, float smoothness)
{
    std::memset(m, 0, sizeof *m);
    m->type = RT_MAT_STANDARD;
    m->tex_type = RT_TEX_GRADIENT;
    m->smoothness = smoothness;
    m->need_uv = 1;
}

extern "C" void rt_material_refractive(rt_material *m, const float colour[3], float n)
{
    std::memset(m, 0, sizeof *m);
    m->type = RT_MAT_REFRACTIVE;
    m->tex_type = RT_TEX_COLOUR;
    std::memcpy(m->colour, colour, 12);
    m->refractive_index = n;
    m->need_uv = 0;
    m->smoothness = 1;                 /* src/material.cu:182 */
}

extern "C" void rt_material_image(rt_material *m, int32_t width, int32_t height, const float *rgb, float smoothness)
{
    std::memset(m, 0, sizeof *m);
    m->type = RT_MAT_STANDARD;
    m->tex_type = RT_TEX_IMAGE;
    m->smoothness = smoothness;
    m->need_uv = 1;
    m->img_w = width;
    m->img_h = height;
    m->img_rgb = rgb;
}

extern "C" rt_status rt_image_texture_load(const char *path, const char *name, int32_t *width, int32_t *height, float **rgb)
{   /* ImageTexture::parse_file / parse_rgb_values src/main.cu:58-90 */
    if (!path || !name || !width || !height || !rgb) return RT_ERR_INVALID;
    *rgb = nullptr;
    std::ifstream file(path);
    if (!file) return RT_ERR_IO;
    std::vector<std::string> lines;
    std::string line;
    while (std::getline(file, line)) lines.push_back(line);
    for (size_t i = 0; i + 3 < lines.size(); i++) {
        if (lines[i] != name) continue;
        try {
            *width = std::stoi(lines[i + 1]);
            *height = std::stoi(lines[i + 2]);
            std::vector<std::string> tok = split_string(lines[i + 3], ' ');
            std::vector<float> v;
            /* "the last character will just be ''": loop to the last but one token, in threes */
            for (size_t k = 0; k + 1 < tok.size() && k + 2 < tok.size(); k += 3) {
                v.push_back(std::stof(tok[k]));
                v.push_back(std::stof(tok[k + 1]));
                v.push_back(std::stof(tok[k + 2]));
            }
            if (*width <= 0 || *height <= 0 || v.size() < (size_t)*width * (size_t)*height * 3) return RT_ERR_INVALID;
            float *out = (float *)std::malloc(v.size() * sizeof(float));
            if (!out) return RT_ERR_NOMEM;
            std::memcpy(out, v.data(), v.size() * sizeof(float));
            *rgb = out;
            return RT_OK;
        } catch (const std::exception &) {
            return RT_ERR_INVALID;
        }
    }
    return RT_ERR_INVALID;             /* "Image file not found." */
}

extern "C" void rt_image_texture_free(float *rgb) { std::free(rgb); }

extern "C" void rt_material_emissive(rt_material *m, const float colour[3], float strength)
{
    std::memset(m, 0, sizeof *m);
    m->type = RT_MAT_EMISSIVE;
    m->emitted_light[0] = colour[0] * strength;
    m->emitted_light[1] = colour[1] * strength;
    m->emitted_light[2] = colour[2] * strength;
}

/* ================================ C ABI: scene builder ===================================== */
extern "C" rt_status rt_scene_builder_create(rt_scene_builder **out)
{
    if (!out) return RT_ERR_INVALID;
    *out = new (std::nothrow) rt_scene_builder();
    return *out ? RT_OK : RT_ERR_NOMEM;
}

extern "C" void rt_scene_builder_destroy(rt_scene_builder *b) { delete b; }
extern "C" const char *rt_scene_builder_error(const rt_scene_builder *b) { return b ? b->err.c_str() : "null builder"; }
extern "C" int32_t rt_scene_builder_num_objects(const rt_scene_builder *b) { return b ? (int32_t)b->objs.size() : 0; }

extern "C" rt_status rt_scene_add_sphere(rt_scene_builder *b, const float center[3], float radius, const rt_material *m)
{
    if (rt_status s = check_material(b, m, true)) return s;
    HostObject o;
    o.type = RT_OBJ_SPHERE;
    o.mat = *m;
    o.v[0] = center[0]; o.v[1] = center[1]; o.v[2] = center[2]; o.v[3] = radius;
    keep_texels(o);
    b->objs.push_back(std::move(o));
    return RT_OK;
}

extern "C" rt_status rt_scene_add_triangle(rt_scene_builder *b, const float p1[3], const float p2[3], const float p3[3], const rt_material *m)
{
    if (rt_status s = check_material(b, m, false)) return s;
    HostObject o;
    o.type = RT_OBJ_TRIANGLE;
    o.mat = *m;
    o.tris.push_back(make_tri(f3(p1), f3(p2), f3(p3)));
    keep_texels(o);
    b->objs.push_back(std::move(o));
    return RT_OK;
}

extern "C" rt_status rt_scene_add_triangle_uv(rt_scene_builder *b, const float p[9], const float uv[6], const rt_material *m)
{
    if (rt_status s = check_material(b, m, false)) return s;
    HostObject o;
    o.type = RT_OBJ_TRIANGLE;
    o.mat = *m;
    HostTri t = make_tri(f3(p), f3(p + 3), f3(p + 6));
    std::memcpy(t.uv, uv, 24);
    o.tris.push_back(t);
    keep_texels(o);
    b->objs.push_back(std::move(o));
    return RT_OK;
}

extern "C" rt_status rt_scene_add_quad(rt_scene_builder *b, const float p1[3], const float p2[3], const float p3[3], const float p4[3], const rt_material *m)
{
    if (rt_status s = check_material(b, m, false)) return s;
    HostObject o;
    o.type = RT_OBJ_QUAD;
    o.mat = *m;
    make_quad(o.tris, f3(p1), f3(p2), f3(p3), f3(p4));
    keep_texels(o);
    b->objs.push_back(std::move(o));
    return RT_OK;
}

extern "C" rt_status rt_scene_add_one_way_quad(rt_scene_builder *b, const float p1[3], const float p2[3], const float p3[3], const float p4[3], int32_t invert_normal, const rt_material *m)
{
    if (rt_status s = check_material(b, m, false)) return s;
    HostObject o;
    o.type = RT_OBJ_ONE_WAY_QUAD;
    o.mat = *m;
    make_quad(o.tris, f3(p1), f3(p2), f3(p3), f3(p4));
    /* OneWayQuad::get_normal_vec src/objects.cu:285-289 */
    float multiplier = (float)(1 - 2 * (invert_normal != 0));
    o.v[0] = o.tris[0].n[0] * multiplier; o.v[1] = o.tris[0].n[1] * multiplier; o.v[2] = o.tris[0].n[2] * multiplier;
    keep_texels(o);
    b->objs.push_back(std::move(o));
    return RT_OK;
}

extern "C" rt_status rt_scene_add_cuboid(rt_scene_builder *b, const float tl_near_pos[3], float width, float height, float depth, const rt_material *m)
{
    if (rt_status s = check_material(b, m, false)) return s;
    HostObject o;
    o.type = RT_OBJ_CUBOID;
    o.mat = *m;
    /* Cuboid::create_faces src/objects.cu:327-349 */
    F3 tl_near = f3(tl_near_pos), w = f3(width, 0, 0), h = f3(0, height, 0), d = f3(0, 0, depth);
    F3 tr_near = add(tl_near, w), br_near = sub(tr_near, h), bl_near = sub(tl_near, h);
    F3 tl_far = add(tl_near, d), tr_far = add(tl_far, w), br_far = sub(tr_far, h), bl_far = sub(tl_far, h);
    make_quad(o.tris, tl_near, tr_near, br_near, bl_near);
    make_quad(o.tris, tl_far, tr_far, br_far, bl_far);
    make_quad(o.tris, tl_near, bl_near, bl_far, tl_far);
    make_quad(o.tris, tr_near, br_near, br_far, tr_far);
    make_quad(o.tris, bl_near, br_near, br_far, bl_far);
    make_quad(o.tris, tl_near, tr_near, tr_far, tl_far);
    keep_texels(o);
    b->objs.push_back(std::move(o));
    return RT_OK;
}

extern "C" rt_status rt_scene_add_mesh(rt_scene_builder *b, const float *triangles, int32_t n, const rt_material *m)
{
    if (rt_status s = check_material(b, m, false)) return s;
    if (n < 0 || (n > 0 && !triangles)) return fail(b, RT_ERR_INVALID, "bad triangle array");
    std::vector<HostTri> tris;
    std::vector<std::array<F3, 3>> pts;
    tris.reserve((size_t)n);
    pts.reserve((size_t)n);
    for (int i = 0; i < n; i++) {
        F3 a = f3(triangles + 9 * i), bb = f3(triangles + 9 * i + 3), c = f3(triangles + 9 * i + 6);
        tris.push_back(make_tri(a, bb, c));
        pts.push_back({a, bb, c});
    }
    return add_mesh_tris(b, tris, pts, m);
}

extern "C" rt_status rt_scene_add_obj_mesh(rt_scene_builder *b, const rt_obj *o, const rt_material *m)
{
    if (rt_status s = check_material(b, m, false)) return s;
    if (!o) return fail(b, RT_ERR_INVALID, "null obj");
    /* SceneObjects::create_mesh src/main.cu:127-148 */
    std::vector<HostTri> tris;
    std::vector<std::array<F3, 3>> pts;
    auto vert = [&](int vi) { return f3(o->vx[(size_t)vi], o->vy[(size_t)vi], o->vz[(size_t)vi]); };
    for (const auto &face : o->faces) {
        for (int vi : face)
            if (vi < 0 || (size_t)vi >= o->vx.size()) return fail(b, RT_ERR_INVALID, "face references a missing vertex");
        if (face.size() == 3) {
            F3 a = vert(face[0]), bb = vert(face[1]), c = vert(face[2]);
            tris.push_back(make_tri(a, bb, c));
            pts.push_back({a, bb, c});
        } else if (face.size() == 4) {
            F3 p1 = vert(face[0]), p2 = vert(face[1]), p3 = vert(face[2]), p4 = vert(face[3]);
            make_quad(tris, p1, p2, p3, p4);
            pts.push_back({p1, p2, p3});
            pts.push_back({p1, p4, p3});
        } else {
            return fail(b, RT_ERR_UNSUPPORTED, "Only triangle or quad meshes are supported.\n");
        }
    }
    return add_mesh_tris(b, tris, pts, m);
}

/* ================================ C ABI: .obj loader ======================================= */
namespace {
/* split_string src/obj_read.cu:27-44: tokens after the first start with a space */
std::vector<std::string> split_string(const std::string &str, char split_char)
{
    std::vector<std::string> out;
    std::string cur;
    for (char c : str) {
        if (c == split_char) { out.push_back(cur); cur = " "; }
        else cur += c;
    }
    out.push_back(cur);
    return out;
}
}  // namespace

extern "C" rt_status rt_obj_load(const char *filename, rt_obj **out)
{
    if (!filename || !out) return RT_ERR_INVALID;
    *out = nullptr;
    std::ifstream file(filename);
    if (!file) return RT_ERR_IO;                      /* read_file :8-10 */
    rt_obj *o = new (std::nothrow) rt_obj();
    if (!o) return RT_ERR_NOMEM;
    std::string line;
    try {
        while (std::getline(file, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            std::vector<std::string> tok = split_string(line, ' ');
            if (tok[0] == "v") {                      /* read_vertices :92-112 */
                if (tok.size() < 4) throw std::invalid_argument("short vertex line");
                o->vx.push_back(std::stof(tok[1]));
                o->vy.push_back(std::stof(tok[2]));
                o->vz.push_back(std::stof(tok[3]));
            } else if (tok[0] == "f") {               /* extract_faces :121-147 */
                std::vector<int> face;
                for (size_t i = 1; i < tok.size(); i++) {
                    std::vector<std::string> inxs = split_string(tok[i], '/');
                    face.push_back(std::stoi(inxs[0]) - 1);
                }
                o->faces.push_back(std::move(face));
            }
        }
    } catch (const std::exception &) {
        delete o;
        return RT_ERR_INVALID;
    }
    *out = o;
    return RT_OK;
}

extern "C" void rt_obj_destroy(rt_obj *o) { delete o; }

extern "C" void rt_obj_enlarge(rt_obj *o, float s)
{   /* :59-64 EnlargementMatrix(s,3) * vertex_mat through the generic product */
    const M3 m = {{s, 0, 0}, {0, s, 0}, {0, 0, s}};
    for (size_t i = 0; i < o->vx.size(); i++) {
        F3 p = m3_apply(m, f3(o->vx[i], o->vy[i], o->vz[i]));
        o->vx[i] = p.x; o->vy[i] = p.y; o->vz[i] = p.z;
    }
}

extern "C" void rt_obj_rotate(rt_obj *o, float ax, float ay, float az)
{   /* :66-76 */
    M3 m;
    m3_rotation_xyz(ax, ay, az, m);
    for (size_t i = 0; i < o->vx.size(); i++) {
        F3 p = m3_apply(m, f3(o->vx[i], o->vy[i], o->vz[i]));
        o->vx[i] = p.x; o->vy[i] = p.y; o->vz[i] = p.z;
    }
}

extern "C" void rt_obj_translate(rt_obj *o, float dx, float dy, float dz)
{   /* :78-86 */
    for (size_t i = 0; i < o->vx.size(); i++) { o->vx[i] += dx; o->vy[i] += dy; o->vz[i] += dz; }
}

extern "C" int32_t rt_obj_num_vertices(const rt_obj *o) { return (int32_t)o->vx.size(); }
extern "C" int32_t rt_obj_num_faces(const rt_obj *o) { return (int32_t)o->faces.size(); }
extern "C" int32_t rt_obj_face_arity(const rt_obj *o, int32_t face) { return (face < 0 || (size_t)face >= o->faces.size()) ? -1 : (int32_t)o->faces[(size_t)face].size(); }

extern "C" void rt_obj_get_face(const rt_obj *o, int32_t face, int32_t *out)
{
    if (face < 0 || (size_t)face >= o->faces.size()) return;
    const auto &f = o->faces[(size_t)face];
    for (size_t i = 0; i < f.size(); i++) out[i] = f[i];
}

extern "C" rt_status rt_obj_from_arrays(const float *vertices, int32_t num_vertices, const int32_t *face_indices,
                                        const int32_t *face_arity, int32_t num_faces, rt_obj **out)
{
    if (!out || num_vertices < 0 || num_faces < 0 || (num_vertices && !vertices) || (num_faces && (!face_indices || !face_arity))) return RT_ERR_INVALID;
    rt_obj *o = new (std::nothrow) rt_obj();
    if (!o) return RT_ERR_NOMEM;
    for (int32_t i = 0; i < num_vertices; i++) {
        o->vx.push_back(vertices[3 * i]);
        o->vy.push_back(vertices[3 * i + 1]);
        o->vz.push_back(vertices[3 * i + 2]);
    }
    size_t k = 0;
    for (int32_t f = 0; f < num_faces; f++) {
        if (face_arity[f] < 0) { delete o; return RT_ERR_INVALID; }
        std::vector<int> face;
        for (int32_t j = 0; j < face_arity[f]; j++) face.push_back(face_indices[k++]);
        o->faces.push_back(std::move(face));
    }
    *out = o;
    return RT_OK;
}

extern "C" void rt_obj_get_vertices(const rt_obj *o, float *out)
{
    for (size_t i = 0; i < o->vx.size(); i++) { out[3 * i] = o->vx[i]; out[3 * i + 1] = o->vy[i]; out[3 * i + 2] = o->vz[i]; }
}

extern "C" int32_t rt_obj_num_triangles(const rt_obj *o)
{
    int32_t n = 0;
    for (const auto &f : o->faces) {
        if (f.size() == 3) n += 1;
        else if (f.size() == 4) n += 2;
        else return -1;
    }
    return n;
}

extern "C" rt_status rt_obj_get_triangles(const rt_obj *o, float *out)
{
    if (!o || !out) return RT_ERR_INVALID;
    if (rt_obj_num_triangles(o) < 0) return RT_ERR_UNSUPPORTED;
    /* a face may name a vertex the file does not have (index 0, a negative one, one past the end): the loader keeps what the file
     * says, like the reference's (src/obj_read.cu:121-147), and the uses check (rt_scene_add_obj_mesh does too) */
    for (const auto &f : o->faces)
        for (int vi : f)
            if (vi < 0 || (size_t)vi >= o->vx.size()) return RT_ERR_INVALID;
    size_t k = 0;
    auto put = [&](int vi) { out[k++] = o->vx[(size_t)vi]; out[k++] = o->vy[(size_t)vi]; out[k++] = o->vz[(size_t)vi]; };
    for (const auto &f : o->faces) {
        put(f[0]); put(f[1]); put(f[2]);
        if (f.size() == 4) { put(f[0]); put(f[3]); put(f[2]); }
    }
    return RT_OK;
}

/* ================================ C ABI: camera ============================================ */
extern "C" void rt_camera_make(int32_t W, int32_t H, const float pos[3], float fov, float focal_len,
                               float x_rot, float y_rot, float z_rot, rt_camera *out)
{   /* Camera::assign_constant_mem src/camera.cu:46-60 */
    const float aspect = (float)W / (float)H;                         /* :7 */
    float viewport_width = 2 * focal_len * rt_tanf(fov / 2);          /* :47 */
    float viewport_height = viewport_width / aspect;
    M3 rot;
    m3_rotation_xyz(x_rot, y_rot, z_rot, rot);                        /* rotate_point :63-69 */
    F3 u = sub(m3_apply(rot, f3(1, 0, 0)), f3(0, 0, 0));              /* get_u :71-83 */
    F3 delta_u = set_mag(u, viewport_width / (float)W);
    F3 v = sub(m3_apply(rot, f3(0, -1, 0)), f3(0, 0, 0));             /* get_v :85-97 */
    F3 delta_v = set_mag(v, viewport_height / (float)H);
    F3 plane_normal = normalised(cross(delta_v, delta_u));            /* :53 */
    F3 cam_pos = f3(pos);
    F3 u_step = divide(scale(delta_u, (float)(-W)), 2.0f);            /* get_tl_pos :99-108 */
    F3 v_step = divide(scale(delta_v, (float)(-H)), 2.0f);
    F3 focal = add(scale(plane_normal, focal_len), cam_pos);
    F3 tl = add(add(u_step, v_step), focal);
    store(out->cam_pos, cam_pos);
    store(out->tl_pixel_pos, tl);
    store(out->delta_u, delta_u);
    store(out->delta_v, delta_v);
    out->width = W;
    out->height = H;
}

extern "C" void rt_camera_default(int32_t W, int32_t H, rt_camera *out)
{   /* the pose constants of src/camera.cu:34-41 */
    const float PI = 3.141592653589793f;
    const float pos[3] = {0, 0, 0};
    rt_camera_make(W, H, pos, 60 * (PI / 180), 0.1f, 0 * (PI / 180), 0 * (PI / 180), 0 * (PI / 180), out);
}

/* ================================ flattening =============================================== */
std::string rt_flatten(const rt_scene_builder &b, FlatScene &out)
{
    out = FlatScene();
    size_t n_nodes = 0, n_tris = 0;
    bool any_uv = false;
    for (const HostObject &o : b.objs) {
        n_nodes += o.nodes.size();
        n_tris += o.tris.size();
        if (o.type == RT_OBJ_MESH) out.has_mesh = true;
        if (o.mat.need_uv && o.type != RT_OBJ_SPHERE) any_uv = true;
    }
    if (n_tris > RT_REF_START_MASK) return "scene has too many triangles for the compact encoding";
    size_t n_meshes = 0;
    for (const HostObject &o : b.objs) n_meshes += o.type == RT_OBJ_MESH;
    /* the triangles come last: a scene whose triangles do not fit a CU's LDS can still stage everything before them */
    out.off_nodes = 0;
    out.off_objlds = (int)(n_nodes * 4);
    out.off_meshes = out.off_objlds + (int)(b.objs.size() * RT_OBJLDS_F4);
    out.num_meshes = (int)n_meshes;
    out.off_objtab = out.off_meshes + (int)(n_meshes * 2);
    out.off_tris = out.off_objtab + (int)(b.objs.size() * 3);
    static_assert(sizeof(rt_object) == 48, "rt_object is three 16-byte units");
    out.blob.resize((size_t)out.off_tris + n_tris * 3);
    out.stack_entries = 1;
    out.num_nodes = (int)n_nodes;
    out.num_tris = (int)n_tris;
    if (any_uv) out.tri_uv.resize(n_tris * 6, 0.0f);

    size_t node_base = 0, tri_base = 0, mesh_i = 0;
    for (size_t oi = 0; oi < b.objs.size(); oi++) {
        const HostObject &o = b.objs[oi];
        rt_object ro;
        std::memset(&ro, 0, sizeof ro);
        ro.type = o.type;
        ro.prim_start = (int32_t)tri_base;
        ro.need_uv = o.mat.need_uv;
        ro.root_ref = o.root_ref;
        std::memcpy(ro.v, o.v, sizeof ro.v);
        auto rebase = [&](uint32_t ref) -> uint32_t {
            const uint32_t chain = ref & RT_REF_CHAIN;
            if (ref & RT_REF_LEAF) {
                uint32_t count = (ref >> RT_REF_COUNT_SHIFT) & RT_REF_COUNT_MAX;
                if (count == 0) return RT_REF_EMPTY_LEAF;
                uint32_t start = (ref & RT_REF_START_MASK) + (uint32_t)tri_base;
                return RT_REF_LEAF | chain | (count << RT_REF_COUNT_SHIFT) | start;
            }
            return ((ref & RT_REF_NODE_MASK) + (uint32_t)node_base) | chain;
        };
        if (o.type == RT_OBJ_MESH) {
            ro.root_ref = rebase(o.root_ref);
            /* mesh table entry: (root box min, max.x) (max.yz, root_ref, object index) */
            rt_f4 *mt = &out.blob[(size_t)out.off_meshes + mesh_i * 2];
            mt[0] = rt_f4{o.v[0], o.v[1], o.v[2], o.v[3]};
            mt[1] = rt_f4{o.v[4], o.v[5], rt_u2f(ro.root_ref), rt_u2f((uint32_t)oi)};
            mesh_i++;
            out.stack_entries = std::max(out.stack_entries, o.stack_need);
        }
        for (size_t k = 0; k < o.nodes.size(); k++) {
            rt_node n = o.nodes[k];
            n.q[3].x = rt_u2f(rebase(rt_f2u(n.q[3].x)));
            n.q[3].y = rt_u2f(rebase(rt_f2u(n.q[3].y)));
            std::memcpy(&out.blob[(size_t)out.off_nodes + (node_base + k) * 4], &n, sizeof n);
        }
        for (size_t k = 0; k < o.tris.size(); k++) {
            const HostTri &t = o.tris[k];
            rt_f4 *q = &out.blob[(size_t)out.off_tris + (tri_base + k) * 3];
            q[0] = rt_f4{t.p0[0], t.p0[1], t.p0[2], t.s1[0]};
            q[1] = rt_f4{t.s1[1], t.s1[2], t.s2[0], t.s2[1]};
            q[2] = rt_f4{t.s2[2], t.n[0], t.n[1], t.n[2]};
            if (any_uv) std::memcpy(&out.tri_uv[(tri_base + k) * 6], t.uv, 24);
        }
        /* per-object shading record */
        const rt_material &m = o.mat;
        rt_f4 *rec = &out.blob[(size_t)out.off_objlds + oi * RT_OBJLDS_F4];
        const float *A = (m.tex_type == RT_TEX_CHECKERBOARD) ? m.light : m.colour;
        const float *B = (m.type == RT_MAT_EMISSIVE) ? m.emitted_light : m.dark;
        uint32_t packed = RT_PACK_MAT(m.type, m.tex_type, m.need_uv ? 1 : 0, o.type == RT_OBJ_SPHERE ? 1 : 0, m.tex_type == RT_TEX_CHECKERBOARD ? m.num_squares : 0);
        rec[0] = rt_f4{A[0], A[1], A[2], m.smoothness};
        if (m.tex_type == RT_TEX_IMAGE && m.type != RT_MAT_EMISSIVE) {
            rec[0] = rt_f4{rt_u2f((uint32_t)m.img_w), rt_u2f((uint32_t)m.img_h), rt_u2f((uint32_t)out.tex_data.size()), m.smoothness};
            out.tex_data.insert(out.tex_data.end(), o.texels.begin(), o.texels.end());
        }
        rec[1] = rt_f4{B[0], B[1], B[2], rt_u2f(packed)};
        rec[2] = rt_f4{o.v[0], o.v[1], o.v[2], o.v[3]};
        rec[3] = rt_f4{m.refractive_index, 0.0f, 0.0f, 0.0f};
        out.objects.push_back(ro);
        std::memcpy(&out.blob[(size_t)out.off_objtab + oi * 3], &ro, sizeof ro);
        node_base += o.nodes.size();
        tri_base += o.tris.size();
    }
    return "";
}

rt_scene_builder::~rt_scene_builder() { delete debug_flat; }

extern "C" rt_status rt_debug_flatten(rt_scene_builder *b, rt_flat_view *out)
{
    if (!b || !out) return RT_ERR_INVALID;
    if (!b->debug_flat) b->debug_flat = new (std::nothrow) FlatScene();
    if (!b->debug_flat) return RT_ERR_NOMEM;
    std::string err = rt_flatten(*b, *b->debug_flat);
    if (!err.empty()) return fail(b, RT_ERR_UNSUPPORTED, err.c_str());
    const FlatScene &f = *b->debug_flat;
    out->blob = reinterpret_cast<const float *>(f.blob.data());
    out->blob_f4 = (int32_t)f.blob.size();
    out->off_nodes = f.off_nodes;
    out->off_tris = f.off_tris;
    out->off_objlds = f.off_objlds;
    out->off_meshes = f.off_meshes;
    out->num_meshes = f.num_meshes;
    out->stack_entries = f.stack_entries;
    out->objects = f.objects.data();
    out->num_objects = (int32_t)f.objects.size();
    out->object_stride = (int32_t)sizeof(rt_object);
    out->tri_uv = f.tri_uv.empty() ? nullptr : f.tri_uv.data();
    out->num_triangles = f.num_tris;
    out->num_nodes = f.num_nodes;
    out->has_mesh = f.has_mesh ? 1 : 0;
    return RT_OK;
}
