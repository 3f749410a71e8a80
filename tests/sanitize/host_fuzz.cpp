// Host-side code of the library under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only; test infrastructure).
// ray-tracer_amd/csrc/rt_host.cpp - the .obj and baked-texture parsers, the scene builder, the BVH build and the flattening into the
// device layout - is compiled with -fsanitize=address,undefined together with this driver and fed random well-formed and
// malformed input through the C ABI of include/rt_amd.h: files cut short, indices out of range or negative, numbers that do not
// parse, empty meshes, degenerate triangles, huge and tiny coordinates, NaNs.  Every call must return (OK or an error status)
// without the sanitizers firing; what a well-formed input flattens to is checked elsewhere (tests/test_host.py against the oracle).
//   g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-sanitize-recover=all -I include tests/sanitize/host_fuzz.cpp ray-tracer_amd/csrc/rt_host.cpp
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "rt_amd.h"

static std::mt19937_64 rng;
static double u01() { return std::uniform_real_distribution<double>(0.0, 1.0)(rng); }
static int irand(int lo, int hi) { return (int)std::uniform_int_distribution<long long>(lo, hi)(rng); }

static float odd_float()
{
    switch (irand(0, 9)) {
    case 0: return 0.0f;
    case 1: return -0.0f;
    case 2: return 1e30f;
    case 3: return -1e30f;
    case 4: return 1e-30f;
    case 5: return NAN;
    case 6: return INFINITY;
    default: return (float)((u01() - 0.5) * 8.0);
    }
}

static std::string number_text()
{
    switch (irand(0, 11)) {
    case 0: return "";
    case 1: return "abc";
    case 2: return "1e999";
    case 3: return "-";
    case 4: return "nan";
    case 5: return "0x10";
    case 6: return "99999999999999999999";
    default: { char b[64]; std::snprintf(b, sizeof b, "%.6g", (u01() - 0.5) * 10.0); return b; }
    }
}

static std::string write_obj(const std::string &dir, int id, int *nv_out)
{
    const std::string path = dir + "/f" + std::to_string(id) + ".obj";
    FILE *fp = std::fopen(path.c_str(), "wb");
    const int nv = irand(0, 40), nf = irand(0, 60);
    const bool well_formed = u01() < 0.5;
    for (int i = 0; i < nv; i++) {
        if (well_formed) std::fprintf(fp, "v %.6g %.6g %.6g\n", (u01() - 0.5) * 4, (u01() - 0.5) * 4, (u01() - 0.5) * 4);
        else {
            const int k = irand(0, 5);                                  /* tokens on the line */
            std::fprintf(fp, "v");
            for (int j = 0; j < k; j++) std::fprintf(fp, " %s", number_text().c_str());
            std::fprintf(fp, u01() < 0.2 ? "\r\n" : "\n");
        }
        if (u01() < 0.1) std::fprintf(fp, "# comment\nvn 0 1 0\nvt 0.5 0.5\n\n");
    }
    for (int i = 0; i < nf; i++) {
        const int ar = well_formed ? irand(3, 4) : irand(0, 6);
        std::fprintf(fp, "f");
        for (int j = 0; j < ar; j++) {
            long long idx = well_formed && nv > 0 ? irand(1, nv) : (long long)irand(-5, nv + 5);
            if (!well_formed && u01() < 0.1) std::fprintf(fp, " %s", number_text().c_str());
            else if (u01() < 0.3) std::fprintf(fp, " %lld/%d/%d", idx, irand(0, 9), irand(0, 9));
            else if (u01() < 0.1) std::fprintf(fp, " %lld//", idx);
            else std::fprintf(fp, " %lld", idx);
        }
        std::fprintf(fp, "\n");
    }
    if (!well_formed && u01() < 0.3) std::fprintf(fp, "f 1 2");           /* no newline at the end, short face */
    std::fclose(fp);
    *nv_out = nv;
    return path;
}

static void fuzz_obj_and_scene(const std::string &dir, int id)
{
    int nv = 0;
    const std::string path = write_obj(dir, id, &nv);
    rt_obj *o = nullptr;
    rt_status st = rt_obj_load(path.c_str(), &o);
    std::remove(path.c_str());
    rt_scene_builder *b = nullptr;
    if (rt_scene_builder_create(&b) != RT_OK) std::abort();
    rt_material m;
    float c[3] = {0.5f, 0.4f, 0.3f}, d[3] = {0.1f, 0.2f, 0.3f};
    switch (irand(0, 5)) {
    case 0: rt_material_standard(&m, c, (float)u01()); break;
    case 1: rt_material_checkerboard(&m, c, d, irand(-2, 12), (float)u01()); break;
    case 2: rt_material_gradient(&m, (float)u01()); break;
    case 3: rt_material_emissive(&m, c, (float)(u01() * 5)); break;
    case 4: rt_material_refractive(&m, c, (float)(0.5 + u01())); break;
    default: rt_material_standard(&m, c, odd_float()); break;
    }
    if (st == RT_OK && o) {
        (void)rt_obj_num_vertices(o);
        const int nfaces = rt_obj_num_faces(o);
        for (int f = -1; f <= nfaces; f++) {
            const int ar = rt_obj_face_arity(o, f);
            if (ar > 0) { std::vector<int32_t> idx((size_t)ar); rt_obj_get_face(o, f, idx.data()); }
        }
        if (u01() < 0.5) rt_obj_enlarge(o, odd_float());
        if (u01() < 0.5) rt_obj_rotate(o, odd_float(), (float)u01(), (float)u01());
        if (u01() < 0.5) rt_obj_translate(o, odd_float(), 0.f, 1.f);
        std::vector<float> vs((size_t)rt_obj_num_vertices(o) * 3 + 3);
        rt_obj_get_vertices(o, vs.data());
        const int nt = rt_obj_num_triangles(o);
        if (nt >= 0) { std::vector<float> tri((size_t)nt * 9 + 9); (void)rt_obj_get_triangles(o, tri.data()); }
        (void)rt_scene_add_obj_mesh(b, o, &m);
    }
    /* primitives with odd coordinates next to the mesh */
    const int np = irand(0, 6);
    for (int i = 0; i < np; i++) {
        float p[4][3];
        for (auto &q : p) for (float &x : q) x = u01() < 0.8 ? (float)((u01() - 0.5) * 6) : odd_float();
        switch (irand(0, 5)) {
        case 0: (void)rt_scene_add_sphere(b, p[0], u01() < 0.8 ? (float)u01() : odd_float(), &m); break;
        case 1: (void)rt_scene_add_triangle(b, p[0], p[1], p[2], &m); break;
        case 2: (void)rt_scene_add_quad(b, p[0], p[1], p[2], p[3], &m); break;
        case 3: (void)rt_scene_add_one_way_quad(b, p[0], p[1], p[2], p[3], irand(0, 1), &m); break;
        case 4: (void)rt_scene_add_cuboid(b, p[0], (float)u01(), odd_float(), (float)u01(), &m); break;
        default: {
            const int n = irand(0, 30);
            std::vector<float> tris((size_t)n * 9 + 1);
            for (float &x : tris) x = u01() < 0.9 ? (float)((u01() - 0.5) * 3) : odd_float();
            (void)rt_scene_add_mesh(b, tris.data(), n, &m);
            float uvp[9], uv[6];
            for (float &x : uvp) x = (float)u01();
            for (float &x : uv) x = odd_float();
            (void)rt_scene_add_triangle_uv(b, uvp, uv, &m);
        } break;
        }
    }
    rt_flat_view fv;
    std::memset(&fv, 0, sizeof fv);
    if (rt_debug_flatten(b, &fv) == RT_OK) {
        /* read what the view points at: the sanitizer checks the extents the library reports */
        volatile float sink = 0;
        for (long long i = 0; i < (long long)fv.blob_f4 * 4; i += 7) sink = sink + fv.blob[i];
        const unsigned char *ob = (const unsigned char *)fv.objects;
        for (long long i = 0; i < (long long)fv.num_objects * fv.object_stride; i += 5) sink = sink + ob[i];
        if (fv.tri_uv) for (long long i = 0; i < (long long)fv.num_triangles * 6; i += 3) sink = sink + fv.tri_uv[i];
    }
    (void)rt_scene_builder_error(b);
    (void)rt_scene_builder_num_objects(b);
    rt_scene_builder_destroy(b);
    rt_obj_destroy(o);
}

static void fuzz_texture_file(const std::string &dir, int id)
{
    const std::string path = dir + "/t" + std::to_string(id) + ".txt";
    FILE *fp = std::fopen(path.c_str(), "wb");
    const int entries = irand(0, 3);
    for (int e = 0; e < entries; e++) {
        const int w = irand(-1, 6), h = irand(-1, 5);
        std::fprintf(fp, "img%d.png\n", e);
        if (u01() < 0.9) std::fprintf(fp, "%d\n", w); else std::fprintf(fp, "%s\n", number_text().c_str());
        if (u01() < 0.9) std::fprintf(fp, "%d\n", h);
        const int n = u01() < 0.7 ? w * h * 3 : irand(0, 40);
        for (int i = 0; i < n; i++) { if (u01() < 0.95) std::fprintf(fp, "%.4g ", u01()); else std::fprintf(fp, "%s ", number_text().c_str()); }
        if (u01() < 0.8) std::fprintf(fp, "\n");
    }
    std::fclose(fp);
    for (int e = -1; e <= entries; e++) {
        int32_t w = 0, h = 0;
        float *rgb = nullptr;
        const std::string name = "img" + std::to_string(e) + ".png";
        if (rt_image_texture_load(path.c_str(), name.c_str(), &w, &h, &rgb) == RT_OK) {
            volatile float sink = 0;
            for (long long i = 0; i < (long long)w * h * 3; i++) sink = sink + rgb[i];
            rt_material m;
            rt_material_image(&m, w, h, rgb, 0.2f);
            rt_scene_builder *b = nullptr;
            if (rt_scene_builder_create(&b) != RT_OK) std::abort();
            float c[3] = {0, 0, 0};
            (void)rt_scene_add_sphere(b, c, 1.0f, &m);
            rt_flat_view fv;
            (void)rt_debug_flatten(b, &fv);
            rt_scene_builder_destroy(b);
            rt_image_texture_free(rgb);
        }
    }
    std::remove(path.c_str());
}

/* rt_obj_from_arrays with indices that may miss the vertex array, and meshes whose triangles all coincide (a BVH that cannot split
 * them: the 1,023-triangles-per-leaf limit must come back as an error) */
static void fuzz_arrays_and_big_leaves()
{
    const int nv = irand(0, 12), nf = irand(0, 20);
    std::vector<float> v((size_t)nv * 3 + 3);
    for (float &x : v) x = u01() < 0.9 ? (float)((u01() - 0.5) * 4) : odd_float();
    std::vector<int32_t> arity((size_t)nf + 1), idx;
    for (int f = 0; f < nf; f++) {
        arity[(size_t)f] = irand(0, 10) ? irand(3, 4) : irand(0, 6);
        for (int j = 0; j < arity[(size_t)f]; j++) idx.push_back(irand(0, 12) ? irand(0, nv > 0 ? nv - 1 : 0) : irand(-3, nv + 3));
    }
    idx.push_back(0);
    rt_obj *o = nullptr;
    rt_material m;
    float c[3] = {0.2f, 0.3f, 0.4f};
    rt_material_standard(&m, c, 0.1f);
    rt_scene_builder *b = nullptr;
    if (rt_scene_builder_create(&b) != RT_OK) std::abort();
    if (rt_obj_from_arrays(v.data(), nv, idx.data(), arity.data(), nf, &o) == RT_OK && o) {
        const int nt = rt_obj_num_triangles(o);
        if (nt >= 0) { std::vector<float> tri((size_t)nt * 9 + 9); (void)rt_obj_get_triangles(o, tri.data()); }
        (void)rt_scene_add_obj_mesh(b, o, &m);
    }
    if (irand(0, 3) == 0) {
        const int n = irand(900, 2500);
        std::vector<float> tris((size_t)n * 9);
        float base[9];
        for (float &x : base) x = (float)(u01() - 0.5);
        const bool jitter = irand(0, 1);
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 9; k++) tris[(size_t)i * 9 + k] = base[k] + (jitter ? (float)(u01() * 1e-6) : 0.0f);
        (void)rt_scene_add_mesh(b, tris.data(), n, &m);
    }
    rt_flat_view fv;
    std::memset(&fv, 0, sizeof fv);
    if (rt_debug_flatten(b, &fv) == RT_OK) {
        volatile float sink = 0;
        for (long long i = 0; i < (long long)fv.blob_f4 * 4; i += 11) sink = sink + fv.blob[i];
    }
    rt_scene_builder_destroy(b);
    rt_obj_destroy(o);
}

int main(int argc, char **argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: %s <scratch dir> <seed> <cases>\n", argv[0]); return 2; }
    const std::string dir = argv[1];
    rng.seed((unsigned long long)std::atoll(argv[2]));
    const int cases = std::atoi(argv[3]);
    for (int i = 0; i < cases; i++) {
        fuzz_obj_and_scene(dir, i);
        if (i % 4 == 0) fuzz_texture_file(dir, i);
        if (i % 3 == 0) fuzz_arrays_and_big_leaves();
        rt_camera cam;
        rt_camera_default(irand(1, 4000), irand(1, 3000), &cam);
        float pos[3] = {odd_float(), (float)u01(), -1.0f};
        rt_camera_make(irand(1, 500), irand(1, 500), pos, u01() < 0.9 ? (float)(0.2 + u01()) : odd_float(), (float)(0.1 + u01()), (float)u01(), odd_float(), (float)u01(), &cam);
    }
    std::printf("host fuzz: %d cases, sanitizers silent\n", cases);
    return 0;
}
