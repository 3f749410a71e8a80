/*
 * rt_oracle.h — C interface of the CPU oracle (TEST INFRASTRUCTURE, not product code).
 *
 * The oracle is a plain-C restatement of the reference renderer's per-pixel hot path
 * (Ben-Edwards44/Ray-Tracer, src/raytracer.cu + everything it calls) and of the host code
 * that produces its inputs (camera, .obj loader, transforms, BVH build).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (ray-tracer_amd/) never links or calls it.
 *
 * Two math bindings ("modes") of the SAME restatement:
 *   ORC_MATH_LIBM (0): log/cos/sin/tan come from the platform libm, as in the reference when
 *                      its sources are compiled for the CPU.  Used to pin the restatement
 *                      against the reference outputs recorded in SURVEY.md App. A.12 / C.2.
 *   ORC_MATH_DET  (1): the same calls are bound to ray-tracer_amd/csrc/rt_math.h, the
 *                      deterministic functions the HIP kernel uses.  This is the mode the GPU
 *                      result is compared with bit for bit.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_MATH_LIBM = 0, ORC_MATH_DET = 1 };

/* Material / texture tags: reference src/material.cu:7-10, :131-133 */
enum { ORC_TEX_COLOUR = 0, ORC_TEX_GRADIENT = 1, ORC_TEX_CHECKERBOARD = 2, ORC_TEX_IMAGE = 3 };
enum { ORC_MAT_STANDARD = 0, ORC_MAT_EMISSIVE = 1, ORC_MAT_REFRACTIVE = 2 };

typedef struct {
    int32_t type;             /* ORC_MAT_* */
    int32_t tex_type;         /* ORC_TEX_* */
    float colour[3];          /* COLOUR */
    float light[3], dark[3];  /* CHECKERBOARD */
    int32_t num_squares;
    int32_t img_w, img_h;     /* IMAGE */
    const float *img_rgb;     /* img_w*img_h*3, owned by caller, must outlive the scene */
    float smoothness;
    int32_t need_uv;
    float emitted[3];
    float refractive_index;
} orc_material;

/* Material factories: reference src/material.cu:157-185.  create_emissive leaves
 * smoothness / need_uv / texture uninitialised in the reference; the
 * oracle defines them as 0 / false / COLOUR(0,0,0). */
void orc_material_standard(orc_material *m, int tex_type, const float colour[3], float smoothness);
void orc_material_checkerboard(orc_material *m, const float light[3], const float dark[3], int num_squares, float smoothness);
void orc_material_emissive(orc_material *m, const float colour[3], float strength);
void orc_material_refractive(orc_material *m, const float colour[3], float n);
/* Material::create_standard(Texture::create_image(w, h, rgb), smoothness): rgb stays owned by the caller */
void orc_material_image(orc_material *m, int w, int h, const float *rgb, float smoothness);

typedef struct {
    uint64_t samples, bounce_iters, hits, rng_draws;
    uint64_t sphere_tests, sphere_hits, tri_tests, box_tests;
    /* the reference's dead get_ray_collision per pixel (src/raytracer.cu:98: its result is discarded): its
     * intersection tests, kept apart so that the per-sample figures above stay the live work's */
    uint64_t dead_sphere_tests, dead_tri_tests, dead_box_tests;
} orc_stats;

typedef struct orc_scene orc_scene;
typedef struct orc_obj orc_obj;

orc_scene *orc_scene_new(int math_mode);
void orc_scene_free(orc_scene *s);
int orc_scene_num_objects(const orc_scene *s);

/* Object factories: reference src/objects.cu:845-906 (list order = call order). */
void orc_add_sphere(orc_scene *s, const float c[3], float r, const orc_material *m);
void orc_add_triangle(orc_scene *s, const float p1[3], const float p2[3], const float p3[3], const orc_material *m);
void orc_add_triangle_uv(orc_scene *s, const float p[9], const float uv[6], const orc_material *m);
void orc_add_quad(orc_scene *s, const float p1[3], const float p2[3], const float p3[3], const float p4[3], const orc_material *m);
void orc_add_one_way_quad(orc_scene *s, const float p1[3], const float p2[3], const float p3[3], const float p4[3], int invert_normal, const orc_material *m);
void orc_add_cuboid(orc_scene *s, const float tl_near[3], float w, float h, float d, const orc_material *m);
/* tris: n*9 floats (three vertices each); builds the depth-10 BVH of src/objects.cu:602-719 */
void orc_add_mesh(orc_scene *s, const float *tris, int n, const orc_material *m);
/* faces as produced by ObjFileMesh (3 or 4 vertices each): reference src/main.cu:127-148.
 * returns 0, or -1 for a face that is neither a triangle nor a quad. */
int orc_add_mesh_faces(orc_scene *s, const orc_obj *o, const orc_material *m);

/* BVH introspection for tests: node count, leaf-size histogram (hist[k] = leaves holding k
 * triangles, k clipped to hist_len-1) */
int orc_mesh_bvh_info(const orc_scene *s, int object_index, int *num_nodes, int *hist, int hist_len);
/* closest hit of one ray against the whole scene (for brute-force cross-checks):
 * returns 1 on hit; out = {dist, hit_point xyz, normal xyz, object index} */
int orc_trace_one(const orc_scene *s, const float origin[3], const float dir[3], float out[8]);

/* .obj loader + transforms: reference src/obj_read.cu:8-147, src/matrix.cu */
orc_obj *orc_obj_load(const char *path, int math_mode);   /* NULL if the file cannot be opened */
void orc_obj_free(orc_obj *o);
void orc_obj_enlarge(orc_obj *o, float scale);
void orc_obj_rotate(orc_obj *o, float ax, float ay, float az);
void orc_obj_translate(orc_obj *o, float dx, float dy, float dz);
int orc_obj_num_vertices(const orc_obj *o);
int orc_obj_num_faces(const orc_obj *o);
int orc_obj_face_arity(const orc_obj *o, int face);
void orc_obj_vertices(const orc_obj *o, float *out /* num_vertices*3 */);
int orc_obj_num_triangles(const orc_obj *o);               /* after the quad split; -1 on bad arity */
void orc_obj_triangles(const orc_obj *o, float *out /* num_triangles*9 */);

/* Camera: reference src/camera.cu:34-108 with the image size made a parameter.
 * out = cam_pos, tl_pixel_pos, delta_u, delta_v (12 floats, src/camera.cu:12-21). */
void orc_camera_default(int W, int H, int math_mode, float out[12]);

/* The hot path: reference src/raytracer.cu:116-136 for every pixel of rows [y0, y1).
 * prev and out are full W*H*3 frames (rows outside [y0,y1) are untouched).  prev may be NULL (zeros).
 * stats may be NULL. */
void orc_render(const orc_scene *s, const float cam[12], int W, int H,
                int rays_per_pixel, int reflection_limit, int antialias, const float sky[3],
                int32_t time_ms, int32_t frame_num, int y0, int y1,
                const float *prev, float *out, int nthreads, orc_stats *stats);

/* known-answer helpers */
float orc_pcg_next(uint32_t *state);                       /* src/utils.cu:220-231 */
float orc_normal_next(uint32_t *state, int math_mode);     /* src/utils.cu:234-239 */
float orc_math_logf(float x, int math_mode);
float orc_math_cosf(float x, int math_mode);
float orc_math_sinf(float x, int math_mode);
float orc_math_tanf(float x, int math_mode);
double orc_math_asin(double x, int math_mode);
double orc_math_acos(double x, int math_mode);
/* float -> RGBA8 display conversion, reference src/main.cu:343-371 */
void orc_to_rgba8(const float *rgb, int W, int H, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
