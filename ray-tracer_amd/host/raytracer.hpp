/*
 * raytracer.hpp — C++ host-side mirror of the reference's scene / camera / render interface,
 * header-only over the C ABI (include/rt_amd.h, libraytracer_amd.so).
 *
 * A reference scene transcribes 1:1: the class and factory names, argument order and error
 * behaviour are the reference's (file:line cited per item, relative to the reference
 * checkout); what happens behind them is not.  Differences a caller sees:
 *   - image size, camera pose, spp, bounce limit, scene and seed are run-time values, not
 *     compile-time constants (SURVEY.md §5 "config / flags");
 *   - no __constant__ globals: Camera, RenderData and the committed scene are passed to
 *     render();
 *   - errors from the device layer are std::runtime_error("Error from HIP (<what>): <detail>")
 *     (reference: "Error from CUDA (...)", src/utils.cu:5-10).
 */
#ifndef RAYTRACER_AMD_HPP
#define RAYTRACER_AMD_HPP

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rt_amd.h"

namespace rtamd {

/* src/utils.cu:13-163 (host subset) */
struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() {}
    Vec3(float vx, float vy, float vz) : x(vx), y(vy), z(vz) {}
    Vec3 operator+(Vec3 o) const { return Vec3(x + o.x, y + o.y, z + o.z); }
    Vec3 operator-(Vec3 o) const { return Vec3(x - o.x, y - o.y, z - o.z); }
    Vec3 operator*(float s) const { return Vec3(x * s, y * s, z * s); }
    Vec3 operator/(float s) const { return Vec3(x / s, y / s, z / s); }
    const float *data() const { return &x; }
};

/* src/utils.cu:166-185 */
struct Vec2 {
    float x = 0, y = 0;
    Vec2() {}
    Vec2(float vx, float vy) : x(vx), y(vy) {}
};

/* src/objects.cu:19-22 */
struct Vertex {
    Vec3 world_point;
    Vec2 texture_point;
};

/* src/material.cu:4-125 */
class Texture {
public:
    static const int COLOUR = 0, GRADIENT = 1, CHECKERBOARD = 2, IMAGE = 3;
    int type = COLOUR;
    Vec3 colour, light, dark;
    int num_squares = 0;

    static Texture create_const_colour(Vec3 texture_colour) { Texture t; t.type = COLOUR; t.colour = texture_colour; return t; }
    static Texture create_gradient() { Texture t; t.type = GRADIENT; return t; }
    static Texture create_checkerboard(Vec3 light_colour, Vec3 dark_colour, int num_sq)
    {
        Texture t; t.type = CHECKERBOARD; t.light = light_colour; t.dark = dark_colour; t.num_squares = num_sq; return t;
    }
    /* src/material.cu:41-51; the texels (row-major rgb floats) must stay alive until the material has
     * been added to a scene (the scene keeps its own copy) */
    static Texture create_image(int width, int height, const float *rgb)
    {
        Texture t; t.type = IMAGE; t.img_w = width; t.img_h = height; t.img_rgb = rgb; return t;
    }
    int img_w = 0, img_h = 0;
    const float *img_rgb = nullptr;
};

/* ImageTexture src/main.cu:40-91: one entry of the baked texture file textures/parse_textures.py writes */
class ImageTexture {
public:
    std::string PARSED_TEXTURE_FILENAME = "textures/parsed_textures.txt";

    explicit ImageTexture(const std::string &filename) { parse_file(filename); }
    ImageTexture(const std::string &filename, const std::string &parsed_texture_filename) : PARSED_TEXTURE_FILENAME(parsed_texture_filename) { parse_file(filename); }
    Texture get_device_texture() const { return Texture::create_image(width, height, rgb_values.data()); }

private:
    int width = 0, height = 0;
    std::vector<float> rgb_values;

    void parse_file(const std::string &filename)
    {
        int32_t w = 0, h = 0;
        float *rgb = nullptr;
        rt_status st = rt_image_texture_load(PARSED_TEXTURE_FILENAME.c_str(), filename.c_str(), &w, &h, &rgb);
        if (st == RT_ERR_IO) throw std::runtime_error("Could not find file to open.");       /* src/obj_read.cu:10 */
        if (st != RT_OK) throw std::runtime_error("Image file not found.\n");                /* src/main.cu:72 */
        width = w; height = h;
        rgb_values.assign(rgb, rgb + (size_t)w * (size_t)h * 3);
        rt_image_texture_free(rgb);
    }
};

/* src/material.cu:128-186 */
class Material {
public:
    static const int STANDARD = 0, EMISSIVE = 1, REFRACTIVE = 2;
    rt_material c{};

    static Material create_standard(Texture mat_tex, float smoothness_val)
    {
        Material m;
        switch (mat_tex.type) {
            case Texture::COLOUR: rt_material_standard(&m.c, mat_tex.colour.data(), smoothness_val); break;
            case Texture::GRADIENT: rt_material_gradient(&m.c, smoothness_val); break;
            case Texture::CHECKERBOARD: rt_material_checkerboard(&m.c, mat_tex.light.data(), mat_tex.dark.data(), mat_tex.num_squares, smoothness_val); break;
            case Texture::IMAGE: rt_material_image(&m.c, mat_tex.img_w, mat_tex.img_h, mat_tex.img_rgb, smoothness_val); break;
            default: throw std::logic_error("unknown texture type");
        }
        return m;
    }
    static Material create_emissive(Vec3 emit_colour, float emit_strength)
    {
        Material m;
        rt_material_emissive(&m.c, emit_colour.data(), emit_strength);
        return m;
    }
    static Material create_refractive(Texture mat_tex, float n)
    {
        Material m;
        rt_material_refractive(&m.c, mat_tex.colour.data(), n);
        return m;
    }
};

/* src/obj_read.cu:47-147 */
class ObjFileMesh {
public:
    explicit ObjFileMesh(const std::string &filename)
    {
        rt_status st = rt_obj_load(filename.c_str(), &h_);
        if (st == RT_ERR_IO) throw std::runtime_error("Could not find file to open.");   /* :10 */
        if (st != RT_OK) throw std::runtime_error("Could not parse " + filename);
    }
    ~ObjFileMesh() { rt_obj_destroy(h_); }
    ObjFileMesh(const ObjFileMesh &) = delete;
    ObjFileMesh &operator=(const ObjFileMesh &) = delete;
    ObjFileMesh(ObjFileMesh &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }

    void enlarge(float scale_fact) { rt_obj_enlarge(h_, scale_fact); }
    void rotate(float x_angle, float y_angle, float z_angle) { rt_obj_rotate(h_, x_angle, y_angle, z_angle); }
    void translate(float offset_x, float offset_y, float offset_z) { rt_obj_translate(h_, offset_x, offset_y, offset_z); }
    int num_vertices() const { return rt_obj_num_vertices(h_); }
    int num_faces() const { return rt_obj_num_faces(h_); }
    const rt_obj *handle() const { return h_; }

private:
    rt_obj *h_ = nullptr;
};

/* The scene's object list: std::vector<Object> + Object::create_* (src/objects.cu:845-906,
 * src/main.cu:94-296).  Objects are appended in call order. */
class SceneObjects {
public:
    bool use_sky = true;

    SceneObjects()
    {
        if (rt_scene_builder_create(&b_) != RT_OK) throw std::bad_alloc();
    }
    /* SceneObjects(int test_scene) src/main.cu:100-122: scenes 0, 1 and 3 */
    SceneObjects(int test_scene, const std::string &models_dir) : SceneObjects()
    {
        switch (test_scene) {
            case 0: monkey_test_scene(models_dir); break;
            case 1: reflection_test_scene(); break;
            case 3: refract_test_scene(); break;
            case 2: texture_test_scene(); break;                                   /* needs textures/parsed_textures.txt (not shipped upstream), like the reference */
            case 4: throw std::logic_error("scene 4 is unseeded in the reference; use ray-tracer_amd.scenes.reference_scene4");
            default: throw std::domain_error("Test scene must be number between 0 and 3 (inclusive).\n");   /* :118 */
        }
    }
    ~SceneObjects() { rt_scene_builder_destroy(b_); }
    SceneObjects(const SceneObjects &) = delete;
    SceneObjects &operator=(const SceneObjects &) = delete;

    void create_sphere(Vec3 center, float radius, const Material &mat) { check(rt_scene_add_sphere(b_, center.data(), radius, &mat.c)); }
    void create_triangle(Vec3 p1, Vec3 p2, Vec3 p3, const Material &mat) { check(rt_scene_add_triangle(b_, p1.data(), p2.data(), p3.data(), &mat.c)); }
    void create_triangle(Vertex p1, Vertex p2, Vertex p3, const Material &mat)
    {
        const float p[9] = {p1.world_point.x, p1.world_point.y, p1.world_point.z, p2.world_point.x, p2.world_point.y, p2.world_point.z,
                            p3.world_point.x, p3.world_point.y, p3.world_point.z};
        const float uv[6] = {p1.texture_point.x, p1.texture_point.y, p2.texture_point.x, p2.texture_point.y, p3.texture_point.x, p3.texture_point.y};
        check(rt_scene_add_triangle_uv(b_, p, uv, &mat.c));
    }
    void create_quad(Vec3 p1, Vec3 p2, Vec3 p3, Vec3 p4, const Material &mat) { check(rt_scene_add_quad(b_, p1.data(), p2.data(), p3.data(), p4.data(), &mat.c)); }
    void create_one_way_quad(Vec3 p1, Vec3 p2, Vec3 p3, Vec3 p4, bool invert_normal, const Material &mat)
    {
        check(rt_scene_add_one_way_quad(b_, p1.data(), p2.data(), p3.data(), p4.data(), invert_normal ? 1 : 0, &mat.c));
    }
    void create_cuboid(Vec3 tl_near_pos, float width, float height, float depth, const Material &mat)
    {
        check(rt_scene_add_cuboid(b_, tl_near_pos.data(), width, height, depth, &mat.c));
    }
    /* SceneObjects::create_mesh src/main.cu:127-148 */
    void create_mesh(const ObjFileMesh &obj, const Material &mat) { check(rt_scene_add_obj_mesh(b_, obj.handle(), &mat.c)); }

    /* create_cornell_box src/main.cu:252-288 */
    void create_cornell_box(Vec3 tl_near_pos, float width, float height, float depth, float light_width)
    {
        use_sky = false;
        Material floor = Material::create_standard(Texture::create_checkerboard(Vec3(0.1f, 0.8f, 0.1f), Vec3(0.1f, 0.5f, 0.1f), 8), 0);
        Material l_wall = Material::create_standard(Texture::create_const_colour(Vec3(1, 0.2f, 0.2f)), 0);
        Material r_wall = Material::create_standard(Texture::create_const_colour(Vec3(0.3f, 0.3f, 1)), 0);
        Material back = Material::create_standard(Texture::create_const_colour(Vec3(0.2f, 0.2f, 0.2f)), 0);
        Material roof = Material::create_standard(Texture::create_const_colour(Vec3(0.9f, 0.9f, 0.9f)), 0);
        Material front = Material::create_standard(Texture::create_const_colour(Vec3(1, 1, 1)), 0);
        Vec3 w(width, 0, 0), h(0, height, 0), d(0, 0, depth), p = tl_near_pos;
        create_quad(p - h, p - h + w, p - h + w + d, p - h + d, floor);
        create_quad(p, p - h, p - h + d, p + d, l_wall);
        create_quad(p + w, p + w - h, p + w - h + d, p + w + d, r_wall);
        create_quad(p + d, p + w + d, p + w - h + d, p - h + d, back);
        create_quad(p, p + d, p + w + d, p + w, roof);
        create_one_way_quad(p, p + w, p + w - h, p - h, false, front);
        Material light_mat = Material::create_emissive(Vec3(1, 1, 1), 6);
        Vec3 light_tl(p.x + width / 2 - light_width / 2, p.y, p.z + depth / 2 - light_width / 2);
        create_cuboid(light_tl, light_width, 0.04f, light_width, light_mat);
    }

    int num_objects() const { return rt_scene_builder_num_objects(b_); }
    const rt_scene_builder *handle() const { return b_; }

private:
    rt_scene_builder *b_ = nullptr;

    void check(rt_status st)
    {
        if (st == RT_OK) return;
        std::string msg = rt_scene_builder_error(b_);
        if (st == RT_ERR_UNSUPPORTED) throw std::logic_error(msg);      /* src/main.cu:141 */
        throw std::invalid_argument(msg);
    }
    void monkey_test_scene(const std::string &models_dir)
    {   /* src/main.cu:150-170 */
        create_cornell_box(Vec3(-0.5f, 0.5f, 1.2f), 1, 1, 1, 0.5f);
        ObjFileMesh m(models_dir + "/low_poly_monkey.obj");
        m.enlarge(0.3f);
        m.rotate(0, 2.3f, 0);
        m.translate(0.1f, -0.1f, 1.6f);
        create_mesh(m, Material::create_standard(Texture::create_const_colour(Vec3(1, 1, 1)), 0));
        create_sphere(Vec3(-0.25f, -0.25f, 1.95f), 0.25f, Material::create_standard(Texture::create_const_colour(Vec3(0.8f, 0.8f, 0.8f)), 1));
    }
    /* texture_test_scene src/main.cu:189-204 */
    void texture_test_scene(const std::string &parsed_texture_filename = "textures/parsed_textures.txt")
    {
        create_cornell_box(Vec3(-0.5f, 0.5f, 1.2f), 1, 1, 1, 0.5f);
        ImageTexture earth("earth.png", parsed_texture_filename);
        create_sphere(Vec3(0, 0, 1.7f), 0.25f, Material::create_standard(earth.get_device_texture(), 0));
        Material tri_mat = Material::create_standard(Texture::create_checkerboard(Vec3(1, 1, 1), Vec3(0, 0, 0), 4), 0);
        create_triangle(Vertex{Vec3(0.1f, 0, 1.7f), Vec2(0, 0)}, Vertex{Vec3(0.6f, 0.5f, 1.9f), Vec2(0, 1)}, Vertex{Vec3(0.8f, 0.4f, 2), Vec2(1, 1)}, tri_mat);
    }
    void refract_test_scene()
    {   /* src/main.cu:206-213 */
        create_cornell_box(Vec3(-0.5f, 0.5f, 1.2f), 1, 1, 1, 0.5f);
        create_sphere(Vec3(0, -0.1f, 1.7f), 0.3f, Material::create_refractive(Texture::create_const_colour(Vec3(1, 1, 1)), 1.5f));
    }
    void reflection_test_scene()
    {   /* src/main.cu:172-187 */
        create_cornell_box(Vec3(-0.5f, 0.5f, 1.2f), 1, 1, 1, 0.5f);
        Texture t = Texture::create_const_colour(Vec3(1, 1, 1));
        create_sphere(Vec3(-0.2f, 0.2f, 1.7f), 0.15f, Material::create_standard(t, 0));
        create_sphere(Vec3(0.2f, 0.2f, 1.7f), 0.15f, Material::create_standard(t, 0.33f));
        create_sphere(Vec3(-0.2f, -0.2f, 1.7f), 0.15f, Material::create_standard(t, 0.66f));
        create_sphere(Vec3(0.2f, -0.2f, 1.7f), 0.15f, Material::create_standard(t, 1));
    }
};

/* src/camera.cu:32-108; assign_constant_mem() becomes the POD this object carries */
class Camera {
public:
    rt_camera c{};
    Camera(int width, int height) { rt_camera_default(width, height, &c); }                       /* pose constants of :34-41 */
    Camera(int width, int height, Vec3 pos, float fov, float focal_len, float x_rot, float y_rot, float z_rot)
    {
        rt_camera_make(width, height, pos.data(), fov, focal_len, x_rot, y_rot, z_rot, &c);
    }
};

/* src/raytracer.cu:4-12 with the defaults of RenderSettings src/main.cu:318-330 */
struct RenderData {
    rt_render_settings c{};
    RenderData(int rays_per_pixel = 100, int reflection_limit = 5, bool antialias = true, Vec3 sky_colour = Vec3(0, 0, 0))
    {
        c.rays_per_pixel = rays_per_pixel;
        c.reflection_limit = reflection_limit;
        c.antialias = antialias ? 1 : 0;
        c.sky_colour[0] = sky_colour.x; c.sky_colour[1] = sky_colour.y; c.sky_colour[2] = sky_colour.z;
    }
};

/* src/dispatch.cu:111-115 */
struct VariableRenderData {
    int frame_num;
    std::vector<float> previous_render;
};

/* One GPU: owns the HIP context and the uploaded scene (replaces the __constant__ symbols and
 * allocate_constant_mem, src/dispatch.cu:104-108). */
class Renderer {
public:
    explicit Renderer(int device = 0)
    {
        rt_status st = rt_ctx_create(device, &ctx_);
        if (st == RT_ERR_NO_DEVICE) throw std::runtime_error("Error from HIP (creating context): no usable GPU; there is no CPU fallback");
        if (st != RT_OK) throw std::runtime_error("Error from HIP (creating context): status " + std::to_string(st));
    }
    ~Renderer()
    {
        rt_scene_destroy(scene_);
        rt_ctx_destroy(ctx_);
    }
    Renderer(const Renderer &) = delete;
    Renderer &operator=(const Renderer &) = delete;

    void set_scene(const SceneObjects &objs)
    {
        rt_scene_destroy(scene_);
        scene_ = nullptr;
        check(rt_scene_commit(ctx_, objs.handle(), &scene_));
    }
    /* render(VariableRenderData*, int) src/dispatch.cu:156-163 */
    void render(const Camera &cam, const RenderData &rd, VariableRenderData *data, int current_time_ms)
    {
        if (data->previous_render.size() != (size_t)cam.c.width * (size_t)cam.c.height * 3) throw std::invalid_argument("previous_render has the wrong size");
        int32_t fn = data->frame_num;
        check(rt_render(ctx_, scene_, &cam.c, &rd.c, current_time_ms, &fn, data->previous_render.data()));
        data->frame_num = fn;
    }
    /* several passes of the main loop body at once (src/main.cu:421-424, one seed per frame): the same
     * image as that many render() calls, but the frames overlap on the GPU (rt_render_frames) */
    void render_frames(const Camera &cam, const RenderData &rd, VariableRenderData *data, const std::vector<int> &times_ms)
    {
        if (data->previous_render.size() != (size_t)cam.c.width * (size_t)cam.c.height * 3) throw std::invalid_argument("previous_render has the wrong size");
        if (times_ms.empty()) return;
        std::vector<int32_t> t(times_ms.begin(), times_ms.end());
        int32_t fn = data->frame_num;
        check(rt_render_frames(ctx_, scene_, &cam.c, &rd.c, t.data(), (int32_t)t.size(), &fn, data->previous_render.data()));
        data->frame_num = fn;
    }
    float last_kernel_ms()
    {
        float ms = 0;
        check(rt_last_kernel_ms(ctx_, &ms));
        return ms;
    }
    /* The main loop (src/main.cu:415-431) with frames in flight: submit_frame(get_time()) queues a frame and returns at once,
     * collect_frame() waits for the OLDEST submitted frame and blends it into data like render() would have.  With `depth`
     * frames submitted ahead the GPU stays full although every frame is seeded when it is submitted (rt_frame_submit):
     *     r.set_frames_in_flight(4);
     *     for (;;) { while (r.frames_in_flight() < 4) r.submit_frame(cam, rd, get_time()); r.collect_frame(&data); draw(data); }
     * The image after n collected frames is the image of n render() calls with the same seeds. */
    void set_frames_in_flight(int depth) { check(rt_frame_depth(ctx_, depth)); }
    int frames_in_flight() const { return rt_frames_pending(ctx_); }
    void submit_frame(const Camera &cam, const RenderData &rd, int current_time_ms) { check(rt_frame_submit(ctx_, scene_, &cam.c, &rd.c, current_time_ms, nullptr)); }
    void collect_frame(VariableRenderData *data)
    {
        int32_t fn = data->frame_num;
        check(rt_frame_collect_host(ctx_, &fn, data->previous_render.data()));
        data->frame_num = fn;
    }
    /* the camera moved (src/main.cu:392-407 restarts at frame 0): drop what is in flight */
    void discard_frames()
    {
        int32_t fn = 0;
        while (rt_frames_pending(ctx_) > 0) check(rt_frame_collect_host(ctx_, &fn, nullptr));
    }

private:
    rt_ctx *ctx_ = nullptr;
    rt_scene *scene_ = nullptr;
    void check(rt_status st)
    {
        if (st == RT_OK) return;
        std::string msg = rt_last_error(ctx_);
        if (st == RT_ERR_UNSUPPORTED) throw std::logic_error(msg);
        if (st == RT_ERR_INVALID) throw std::invalid_argument(msg);
        throw std::runtime_error(msg);                                  /* check_cuda_error src/utils.cu:5-10 */
    }
};

/* Several GPUs of one node behind the same render() (rt_render_multi): every device of `devices` renders the 8x8
 * tiles it owns - dealt out interleaved on the first call for a view, which measures them, and by measured cost
 * from the second on - and the tiles meet on devices[0].  The image is the single-GPU image bit for
 * bit (a pixel depends only on its coordinates, its seed and its previous value, src/raytracer.cu:118-131).
 * The reference is single-GPU; this is what its main loop calls when the node has more than one. */
class MultiRenderer {
public:
    explicit MultiRenderer(const std::vector<int> &devices)
    {
        if (devices.empty()) throw std::invalid_argument("no devices");
        for (int d : devices) {
            rt_ctx *c = nullptr;
            rt_status st = rt_ctx_create(d, &c);
            if (st != RT_OK) {
                release();
                if (st == RT_ERR_NO_DEVICE) throw std::runtime_error("Error from HIP (creating context): no usable GPU; there is no CPU fallback");
                throw std::runtime_error("Error from HIP (creating context): status " + std::to_string(st));
            }
            ranks_.push_back(rt_rank{c, nullptr});
        }
    }
    ~MultiRenderer() { release(); }
    MultiRenderer(const MultiRenderer &) = delete;
    MultiRenderer &operator=(const MultiRenderer &) = delete;

    int num_gpus() const { return (int)ranks_.size(); }
    /* every GPU holds the whole scene (it is tens of KB) */
    void set_scene(const SceneObjects &objs)
    {
        for (rt_rank &r : ranks_) {
            rt_scene_destroy(const_cast<rt_scene *>(r.scene));
            r.scene = nullptr;
            rt_scene *s = nullptr;
            check(r.ctx, rt_scene_commit(r.ctx, objs.handle(), &s));
            r.scene = s;
        }
    }
    /* render(VariableRenderData*, int) src/dispatch.cu:156-163 */
    void render(const Camera &cam, const RenderData &rd, VariableRenderData *data, int current_time_ms)
    {
        render_frames(cam, rd, data, std::vector<int>{current_time_ms});
    }
    void render_frames(const Camera &cam, const RenderData &rd, VariableRenderData *data, const std::vector<int> &times_ms)
    {
        if (data->previous_render.size() != (size_t)cam.c.width * (size_t)cam.c.height * 3) throw std::invalid_argument("previous_render has the wrong size");
        if (times_ms.empty()) return;
        std::vector<int32_t> t(times_ms.begin(), times_ms.end());
        int32_t fn = data->frame_num;
        check(ranks_[0].ctx, rt_render_multi(ranks_.data(), (int32_t)ranks_.size(), &cam.c, &rd.c, t.data(), (int32_t)t.size(), &fn, data->previous_render.data()));
        data->frame_num = fn;
    }
    /* does GPU i exchange its tiles with the first GPU directly (xGMI peer access), or staged by the runtime? */
    bool direct_peer_copies(int i) { return rt_peer_access(ranks_.at(0).ctx, ranks_.at((size_t)i).ctx) == 1; }
    /* the slowest rank's kernel time of the last call */
    float last_kernel_ms()
    {
        float worst = 0;
        for (rt_rank &r : ranks_) {
            float ms = 0;
            if (rt_last_kernel_ms(r.ctx, &ms) == RT_OK && ms > worst) worst = ms;
        }
        return worst;
    }

private:
    std::vector<rt_rank> ranks_;
    void release()
    {
        for (rt_rank &r : ranks_) {
            rt_scene_destroy(const_cast<rt_scene *>(r.scene));
            rt_ctx_destroy(r.ctx);
        }
        ranks_.clear();
    }
    static void check(rt_ctx *ctx, rt_status st)
    {
        if (st == RT_OK) return;
        std::string msg = rt_last_error(ctx);
        if (st == RT_ERR_UNSUPPORTED) throw std::logic_error(msg);
        if (st == RT_ERR_INVALID) throw std::invalid_argument(msg);
        throw std::runtime_error(msg);
    }
};

/* parse_pixel_colours src/main.cu:343-371: int(px*255), clamp, alpha 255 */
inline std::vector<uint8_t> parse_pixel_colours(const std::vector<float> &pixel_colours, int width, int height)
{
    std::vector<uint8_t> out((size_t)width * (size_t)height * 4);
    for (size_t i = 0; i < (size_t)width * (size_t)height; i++) {
        for (int k = 0; k < 3; k++) {
            /* a defined conversion (NaN -> 0, saturating), the same as rt_to_rgba8_device: a plain cast of a NaN or an
             * infinite value is undefined in C++ */
            const float scaled = pixel_colours[3 * i + k] * 255;
            int colour = scaled != scaled ? 0 : (scaled >= 2147483648.0f ? 2147483647 : (scaled <= -2147483648.0f ? -2147483647 - 1 : (int)scaled));
            if (colour > 255) colour = 255; else if (colour < 0) colour = 0;
            out[4 * i + k] = (uint8_t)colour;
        }
        out[4 * i + 3] = 255;
    }
    return out;
}

/* What the reference shows in its SFML window (src/main.cu:374-386) written to a file instead:
 * an 8-bit RGB PNG (the format of the reference's images/ directory) without any library - zlib
 * "stored" blocks, so the file is W*H*3 bytes plus headers. */
inline void write_png(const std::string &path, const std::vector<uint8_t> &rgba, int width, int height)
{
    auto crc32 = [](const uint8_t *p, size_t n, uint32_t crc) {
        crc = ~crc;
        for (size_t i = 0; i < n; i++) {
            crc ^= p[i];
            for (int k = 0; k < 8; k++) crc = (crc >> 1) ^ (0xedb88320u & (0u - (crc & 1u)));
        }
        return ~crc;
    };
    auto be32 = [](std::vector<uint8_t> &v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s)); };
    /* raw scanlines: filter byte 0 + RGB */
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * ((size_t)width * 3 + 1));
    for (int y = 0; y < height; y++) {
        raw.push_back(0);
        for (int x = 0; x < width; x++) {
            const uint8_t *px = &rgba[4 * ((size_t)y * (size_t)width + (size_t)x)];
            raw.push_back(px[0]); raw.push_back(px[1]); raw.push_back(px[2]);
        }
    }
    /* zlib stream of stored deflate blocks (at most 65535 bytes each) + adler32 */
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        const size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 255)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 255)); z.push_back((uint8_t)((~n >> 8) & 255));
        z.insert(z.end(), raw.begin() + (std::ptrdiff_t)off, raw.begin() + (std::ptrdiff_t)(off + n));
        for (size_t i = off; i < off + n; i++) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        if (n == 0) break;
    }
    be32(z, (b << 16) | a);
    std::vector<uint8_t> file = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    auto chunk = [&](const char *type, const std::vector<uint8_t> &data) {
        be32(file, (uint32_t)data.size());
        const size_t start = file.size();
        file.insert(file.end(), type, type + 4);
        file.insert(file.end(), data.begin(), data.end());
        be32(file, crc32(&file[start], file.size() - start, 0));
    };
    std::vector<uint8_t> ihdr;
    be32(ihdr, (uint32_t)width); be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   /* 8-bit RGB */
    chunk("IHDR", ihdr);
    chunk("IDAT", z);
    chunk("IEND", {});
    FILE *fp = std::fopen(path.c_str(), "wb");
    if (!fp) throw std::runtime_error("cannot open output file");
    const bool ok = std::fwrite(file.data(), 1, file.size(), fp) == file.size();
    std::fclose(fp);
    if (!ok) throw std::runtime_error("cannot write output file");
}

}  // namespace rtamd

#endif
