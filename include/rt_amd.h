/*
 * rt_amd.h — C ABI of the MI355X path tracer (libraytracer_amd.so).
 *
 * The reference (Ben-Edwards44/Ray-Tracer) has no FFI; the seam its hot path sits behind is
 * the host<->device edge in src/dispatch.cu plus three __constant__ uploads.  Each entry
 * point below names the reference interface it replaces (file:line relative to the
 * reference checkout).  Plain pointers and sizes only; nothing throws across this boundary:
 * every call returns an rt_status and leaves a message for rt_last_error().
 *
 * Threading: one rt_ctx per GPU, used from one host thread at a time (the reference is a
 * single host thread with blocking launches, src/dispatch.cu:139-141).  There are no hidden
 * globals: scene, camera and settings are arguments (the reference's __constant__ symbols
 * make it non-reentrant).
 *
 * One launch in flight per context.  A context owns the scratch its launches use (tile ticket
 * counter, the per-frame planes of a multi-frame launch, tile order and costs, timing events), so
 * the device-buffer entry points keep a context's launches in order: a launch on a different
 * stream than the previous one is queued behind it (hipStreamWaitEvent), it does not overlap it.
 * To overlap renders on one GPU use two contexts; to use several GPUs use one context per GPU
 * (rt_render_multi below, or one process per GPU).  The exception is rt_frame_submit / rt_frame_collect:
 * up to RT_PIPELINE_DEPTH progressive frames of one context in flight, each with scratch of its own.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t rt_status;
enum {
    RT_OK = 0,
    RT_ERR_INVALID = 1,      /* bad argument */
    RT_ERR_IO = 2,           /* "Could not find file to open."  (src/obj_read.cu:10) */
    RT_ERR_UNSUPPORTED = 3,  /* e.g. "Only triangle or quad meshes are supported." (src/main.cu:141) */
    RT_ERR_HIP = 4,          /* "Error from HIP (<what>): <hipGetErrorString>"  (src/utils.cu:5-10) */
    RT_ERR_NOMEM = 5,
    RT_ERR_NO_DEVICE = 6,    /* the HIP runtime reports no usable GPU: there is NO CPU fallback */
    RT_ERR_BUSY = 7          /* rt_frame_submit: RT_PIPELINE_DEPTH frames are in flight */
};

typedef struct rt_ctx rt_ctx;
typedef struct rt_scene rt_scene;
typedef struct rt_scene_builder rt_scene_builder;
typedef struct rt_obj rt_obj;

/* ---- materials: Texture / Material, src/material.cu:4-186 ------------------------------- */
enum { RT_TEX_COLOUR = 0, RT_TEX_GRADIENT = 1, RT_TEX_CHECKERBOARD = 2, RT_TEX_IMAGE = 3 };   /* :7-10 */
enum { RT_MAT_STANDARD = 0, RT_MAT_EMISSIVE = 1, RT_MAT_REFRACTIVE = 2 };                     /* :131-133 */

typedef struct rt_material {
    int32_t type;              /* RT_MAT_* */
    int32_t tex_type;          /* RT_TEX_* */
    float colour[3];           /* Texture::create_const_colour :21-26 */
    float light[3], dark[3];   /* Texture::create_checkerboard :32-40 */
    int32_t num_squares;
    float smoothness;          /* [0,1]: 0 diffuse, 1 mirror ("metal" = STANDARD with smoothness > 0) */
    int32_t need_uv;
    float emitted_light[3];    /* colour * strength, :170 */
    float refractive_index;
    int32_t img_w, img_h;      /* Texture::create_image :42-51 */
    const float *img_rgb;      /* img_w*img_h*3 floats, row-major; copied when the object is added */
} rt_material;

/* Material::create_standard(Texture::create_const_colour(colour), smoothness)  :157-165 */
void rt_material_standard(rt_material *m, const float colour[3], float smoothness);
/* Material::create_standard(Texture::create_checkerboard(light, dark, n), smoothness) */
void rt_material_checkerboard(rt_material *m, const float light[3], const float dark[3], int32_t num_squares, float smoothness);
/* Material::create_standard(Texture::create_gradient(), smoothness) */
void rt_material_gradient(rt_material *m, float smoothness);
/* Material::create_emissive(colour, strength) :167-173.  The reference leaves smoothness,
 * need_uv and texture uninitialised there; this ABI defines them as 0 / false / COLOUR(0,0,0). */
void rt_material_emissive(rt_material *m, const float colour[3], float strength);
/* Material::create_refractive(Texture::create_const_colour(colour), n) :175-185 (smoothness = 1) */
void rt_material_refractive(rt_material *m, const float colour[3], float n);
/* Material::create_standard(Texture::create_image(width, height, rgb), smoothness) :42-51;
 * nearest-texel lookup :119-124 (an out-of-range texel index is clamped, the reference reads
 * out of bounds) */
void rt_material_image(rt_material *m, int32_t width, int32_t height, const float *rgb, float smoothness);
/* ImageTexture src/main.cu:40-91: the entry `name` of a baked texture file
 * (textures/parse_textures.py format: name \n W \n H \n "r g b r g b ... ").  *rgb is malloc'ed;
 * release it with rt_image_texture_free.  RT_ERR_IO: file missing; RT_ERR_INVALID: name not
 * found ("Image file not found.", src/main.cu:72) */
rt_status rt_image_texture_load(const char *parsed_textures_path, const char *name, int32_t *width, int32_t *height, float **rgb);
void rt_image_texture_free(float *rgb);

/* ---- scene: Object::create_* src/objects.cu:845-906, SceneObjects src/main.cu:94-296 ------ */
rt_status rt_scene_builder_create(rt_scene_builder **out);
void rt_scene_builder_destroy(rt_scene_builder *b);
const char *rt_scene_builder_error(const rt_scene_builder *b);
/* objects are kept in call order = the reference's std::vector<Object> order (ties go to
 * the later object, src/raytracer.cu:36) */
rt_status rt_scene_add_sphere(rt_scene_builder *b, const float center[3], float radius, const rt_material *m);          /* :845-852 */
rt_status rt_scene_add_triangle(rt_scene_builder *b, const float p1[3], const float p2[3], const float p3[3], const rt_material *m);  /* :854-861 */
rt_status rt_scene_add_triangle_uv(rt_scene_builder *b, const float p[9], const float uv[6], const rt_material *m);    /* :863-870 */
rt_status rt_scene_add_quad(rt_scene_builder *b, const float p1[3], const float p2[3], const float p3[3], const float p4[3], const rt_material *m);  /* :872-879 */
rt_status rt_scene_add_one_way_quad(rt_scene_builder *b, const float p1[3], const float p2[3], const float p3[3], const float p4[3], int32_t invert_normal, const rt_material *m);  /* :881-888 */
rt_status rt_scene_add_cuboid(rt_scene_builder *b, const float tl_near_pos[3], float width, float height, float depth, const rt_material *m);  /* :890-897 */
/* Object::create_mesh :899-906 — triangles: n*9 floats; the fixed-depth-10 BVH of
 * src/objects.cu:602-719 is rebuilt host-side and stored compactly */
rt_status rt_scene_add_mesh(rt_scene_builder *b, const float *triangles, int32_t n, const rt_material *m);
/* SceneObjects::create_mesh src/main.cu:127-148 — faces of a loaded .obj (3 or 4 vertices) */
rt_status rt_scene_add_obj_mesh(rt_scene_builder *b, const rt_obj *o, const rt_material *m);
int32_t rt_scene_builder_num_objects(const rt_scene_builder *b);

/* ---- .obj loader: ObjFileMesh src/obj_read.cu:47-147 ------------------------------------- */
rt_status rt_obj_load(const char *filename, rt_obj **out);          /* ObjFileMesh(filename) :52-57 */
void rt_obj_destroy(rt_obj *o);
void rt_obj_enlarge(rt_obj *o, float scale_fact);                   /* :59-64 */
void rt_obj_rotate(rt_obj *o, float x_angle, float y_angle, float z_angle);   /* :66-76; sin/cos from rt_math.h */
void rt_obj_translate(rt_obj *o, float dx, float dy, float dz);     /* :78-86 */
int32_t rt_obj_num_vertices(const rt_obj *o);
int32_t rt_obj_num_faces(const rt_obj *o);
int32_t rt_obj_face_arity(const rt_obj *o, int32_t face);
void rt_obj_get_face(const rt_obj *o, int32_t face, int32_t *out /* arity 0-based vertex indices */);
/* the same object from arrays already in memory: vertices n*3 floats, faces as a flat 0-based
 * index list with one arity per face */
rt_status rt_obj_from_arrays(const float *vertices, int32_t num_vertices, const int32_t *face_indices,
                             const int32_t *face_arity, int32_t num_faces, rt_obj **out);
void rt_obj_get_vertices(const rt_obj *o, float *out /* num_vertices*3 */);
int32_t rt_obj_num_triangles(const rt_obj *o);                       /* after the quad split; -1 if a face is not 3/4-sided */
rt_status rt_obj_get_triangles(const rt_obj *o, float *out /* num_triangles*9 */);     /* RT_ERR_INVALID: a face names a vertex the file does not have */

/* ---- camera: src/camera.cu:12-21 (DeviceCamData) and :34-108 (Camera) -------------------- */
typedef struct rt_camera {
    float cam_pos[3];
    float tl_pixel_pos[3];
    float delta_u[3];
    float delta_v[3];
    int32_t width, height;     /* SCREEN_WIDTH / SCREEN_HEIGHT src/camera.cu:4-5, a parameter here */
} rt_camera;

/* Camera::assign_constant_mem :46-60 with the reference's pose constants (:34-41: origin,
 * FOV 60 deg, focal length 0.1, no rotation); tan/sin/cos from rt_math.h */
void rt_camera_default(int32_t width, int32_t height, rt_camera *out);
/* same with pose parameters (angles in radians, rotation order Rx*Ry*Rz as :63-69) */
void rt_camera_make(int32_t width, int32_t height, const float pos[3], float fov, float focal_len,
                    float x_rot, float y_rot, float z_rot, rt_camera *out);

/* ---- render settings: RenderData src/raytracer.cu:4-12 ----------------------------------- */
typedef struct rt_render_settings {
    int32_t rays_per_pixel;
    int32_t reflection_limit;
    int32_t antialias;
    float sky_colour[3];
} rt_render_settings;

/* ---- context + scene upload -------------------------------------------------------------- */
/* fails with RT_ERR_NO_DEVICE when HIP has no device: the product has no CPU path */
rt_status rt_ctx_create(int32_t device, rt_ctx **out);
void rt_ctx_destroy(rt_ctx *ctx);
const char *rt_last_error(const rt_ctx *ctx);
/* replaces create_gpu_struct src/main.cu:290-295 + allocate_constant_mem src/dispatch.cu:104-108:
 * flattens the builder's objects into the compact device layout and uploads it */
rt_status rt_scene_commit(rt_ctx *ctx, const rt_scene_builder *b, rt_scene **out);
void rt_scene_destroy(rt_scene *s);
/* introspection for tests: bytes of LDS per workgroup, node / triangle counts, the launch shape chosen for the
 * scene (workgroup size and how many workgroups are resident per CU); scene_in_lds: 1 the whole scene is staged
 * into LDS, 2 everything but the triangles (a mesh of more than ~1,500 triangles: its BVH still fits), 0 nothing
 * (the kernel reads the scene from global memory / L2) */
typedef struct rt_scene_info {
    int32_t num_objects, num_triangles, num_nodes, lds_bytes, scene_in_lds, threads_per_block, stack_entries, blocks_per_cu;
} rt_scene_info;
rt_status rt_scene_get_info(const rt_scene *s, rt_scene_info *out);

/* ---- the per-frame call: render() src/dispatch.cu:156-163 --------------------------------- */
/* Host-buffer form, same contract as the reference: previous_render (W*H*3 floats, row-major
 * interleaved RGB) is read, blended as (colour + prev*frame_num)/(frame_num+1)
 * (src/raytracer.cu:109-112) and overwritten; *frame_num is incremented.  time_ms is the seed
 * term the reference takes from the wall clock (src/main.cu:18-25). */
rt_status rt_render(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_settings *rs,
                    int32_t time_ms, int32_t *frame_num, float *previous_render);
/* n_frames passes of that loop body in one call (frame i seeded with times_ms[i]); the frames are
 * rendered by multi-frame launches (rt_render_device_batch below), the result is the same image. */
rt_status rt_render_frames(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_settings *rs,
                           const int32_t *times_ms, int32_t n_frames, int32_t *frame_num, float *previous_render);

/* Device-buffer form for callers that own HBM (PyTorch tensors) and streams.
 * Which pixels a call renders is given in one of two forms ("tile rows or tile list"):
 *  - bands: rows are handed out in bands of `band_rows` rows; the call renders the bands whose index b
 *    satisfies b % band_stride == band_first (band_stride = number of GPUs, band_first = rank);
 *  - a tile list (tile_list != NULL; the band fields are then ignored): the call renders the 8x8 tiles
 *    tile_list[0..num_tiles), each named by its index ty * ceil(width / 8) + tx in the image and listed at
 *    most once.  This is the form cost-balanced ownership over GPUs uses (rt_partition_tiles below).
 * d_prev (nullable = zeros) is always a full W*H*3 frame.  If compact == 0, d_out is a full frame and only
 * the owned pixels are written; if compact != 0, d_out holds only what the call owns: the owned bands back
 * to back (band k of this rank at row k*band_rows: the shape an all-gather wants), or the listed tiles back
 * to back (tile k at floats [192 k, 192 k + 192): its 64 pixels row by row; slots of a ragged edge tile that
 * lie outside the image are never written).
 * tile_cost, tile_peak (nullable, with a tile list only): per listed tile, its cost and the cost of its most
 * expensive pixel as rt_tile_costs reported them for an earlier launch of the same view; the launch is then
 * scheduled longest job first at once instead of measuring the tiles itself first (tile_peak == NULL: ordered
 * by tile_cost).  The arrays are host memory, read during the call.
 * The launch is asynchronous on `hip_stream` (a hipStream_t, NULL = default stream). */
typedef struct rt_tile_spec {
    int32_t band_rows;       /* > 0, multiple of 8 */
    int32_t band_first;
    int32_t band_stride;
    int32_t compact;
    const uint32_t *tile_list;
    const uint32_t *tile_cost;
    const uint32_t *tile_peak;
    int32_t num_tiles;
} rt_tile_spec;

rt_status rt_render_device(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_settings *rs,
                           int32_t time_ms, int32_t frame_num, const rt_tile_spec *tiles,
                           const float *d_prev, float *d_out, void *hip_stream);

/* n_frames (1..32) consecutive progressive frames of one view in ONE launch: what the reference's main
 * loop (src/main.cu:415-431) does with one render() per frame - frames frame_num, frame_num + 1, ...,
 * seeded with times_ms[0..n_frames) - accumulated IN PLACE in d_frame (same layout as d_out above; when
 * frame_num > 0 its content is the image after frame_num - 1, otherwise it is ignored).  The result is
 * bit-identical to n_frames rt_render_device calls.  Every frame has its own random stream, so frame
 * k + 1 of a pixel is traced while the expensive pixels of frame k are still running; each frame stores
 * its per-pixel mean into a scratch plane and a small kernel behind the render kernel folds the planes
 * into d_frame in frame order ((c + prev * n) / (n + 1), src/raytracer.cu:109-112).  A launch per frame
 * leaves most of the GPU idle while the few most expensive tiles finish.  The context
 * keeps n_frames planes of the frame's size in HBM. */
rt_status rt_render_device_batch(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_settings *rs,
                                 const int32_t *times_ms, int32_t n_frames, int32_t frame_num, const rt_tile_spec *tiles,
                                 float *d_frame, void *hip_stream);
/* Pipelined frames: the reference's main loop (src/main.cu:415-431: seed from the wall clock, render(), draw, poll) with the next
 * frame's launch issued BEFORE the previous frame is waited for.  A 1920x1080x1024-spp frame is as long as its most expensive
 * pixel - half of it runs on a nearly empty GPU - and rt_render_device_batch only fills that time when the
 * seeds of the coming frames are known in advance, which a loop that draws each seed at call time cannot offer.  Here it can:
 *   rt_frame_submit   queues ONE frame seeded with time_ms on a stream of the context's own (up to rt_frame_depth submitted
 *                     and not yet collected; RT_ERR_BUSY beyond).  The frame's per-pixel means go to a plane the context keeps;
 *                     nothing the caller owns is touched.  Frames in flight run side by side on the GPU.
 *   rt_frame_collect  takes the OLDEST submitted frame and folds it into d_frame as progressive frame `frame_num`
 *                     ((c + prev * frame_num) / (frame_num + 1), src/raytracer.cu:109-112; d_frame has the layout of
 *                     rt_render_device's d_out for the tile spec the frame was submitted with, and is ignored as input when
 *                     frame_num == 0).  Asynchronous: the fold runs behind the frame's render kernel and behind whatever the caller
 *                     has queued on `hip_stream` so far, and work queued on `hip_stream` afterwards sees the folded frame.
 *                     d_frame == NULL discards the frame (the camera moved: the reference restarts at frame 0,
 *                     src/main.cu:392-407).  Frames submitted with a tile LIST must be collected (or discarded) before frames of
 *                     another list are submitted (RT_ERR_BUSY).
 * Frames are collected in submission order, and the image after collecting frames 0..k is bit-identical to k + 1 calls of
 * rt_render_device with the same seeds (tests/test_gpu_pipeline.py).  The loop becomes
 *     submit(t0); for (;;) { submit(now()); collect(n++, d_frame, s); draw(d_frame); }
 * i.e. the picture on screen lags the newest seed by the frames in flight.  Measured (monkey, 1920x1080x1024 spp, bench.py
 * "pipelined"): 8,950 Msamples/s with 4 frames in flight and 9,800 with 8, against 4,600 one launch at a time and 10,800 for
 * rt_render_device_batch.
 * A view's first frame or two (new scene / camera / size / tile spec) run alone: they measure the tiles and sort the schedule,
 * and the host waits for the frames in flight before it rewrites either.  rt_last_kernel_ms does not see pipelined frames;
 * rt_ctx_synchronize waits for them; launches of the other entry points are queued behind them. */
#define RT_PIPELINE_DEPTH 8            /* at most */
#define RT_PIPELINE_DEFAULT_DEPTH 4
/* How many frames the caller is going to keep in flight (1..RT_PIPELINE_DEPTH; a context starts with RT_PIPELINE_DEFAULT_DEPTH):
 * rt_frame_submit refuses more, and - for scenes whose workgroup has a CU to itself, i.e. a mesh that fills the LDS - every frame is
 * launched on 1 / depth of the GPU's CUs: `depth` frames side by side, each bound by its work instead of by its longest pixel (alone
 * on the GPU a frame leaves most CUs idle for half its duration).  Other scenes' launches are full size and share the CUs.
 * Throughput grows with the depth, and so does a frame's latency (depth x the time per frame); 2, 4 and 8 are the useful values
 * (the depths in between measure no better than the next lower one).  depth 1 is rt_render_device with a plane in between.
 * The context keeps one plane (12 bytes per pixel of the launch's layout) per frame in flight.  Only while no frame is in flight
 * (RT_ERR_BUSY otherwise). */
rt_status rt_frame_depth(rt_ctx *ctx, int32_t depth);
rt_status rt_frame_submit(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_settings *rs,
                          int32_t time_ms, const rt_tile_spec *tiles);
rt_status rt_frame_collect(rt_ctx *ctx, int32_t frame_num, float *d_frame, void *hip_stream);
/* Host-buffer form of rt_frame_collect for whole frames (submitted with tiles == NULL), with rt_render's contract: previous_render
 * (W*H*3 floats) is read (when *frame_num > 0), blended with the oldest submitted frame and overwritten; *frame_num is incremented.
 * Returns when the frame is in previous_render; the younger frames keep running.  previous_render == NULL discards the frame. */
rt_status rt_frame_collect_host(rt_ctx *ctx, int32_t *frame_num, float *previous_render);
/* frames submitted and not collected */
int32_t rt_frames_pending(const rt_ctx *ctx);
/* blocks until the frame collected last has been folded into its d_frame (what a host that draws the frame itself waits for:
 * the cudaDeviceSynchronize of src/dispatch.cu:141 for this loop) - the younger frames keep running */
rt_status rt_frame_wait(rt_ctx *ctx);

/* number of rows a rank owns under a band tile spec (host helper for sizing compact buffers) */
int32_t rt_tile_owned_rows(const rt_tile_spec *tiles, int32_t height);

/* What the tiles of this context's current view cost: the first launch of a view (scene, camera, image size,
 * tile spec) adds up, per tile, the work of its pixels (traversal steps, generated rays and shaded hits,
 * weighted); this call waits for that launch and copies the figures out: tile_ids[i] = the tile's index in
 * the image (ty * ceil(width / 8) + tx), costs[i] its cost (opaque units; bit 0 says whether a ray of the tile
 * entered a mesh), peaks[i] (nullable) the cost of its most expensive pixel, for i < *count (= the tiles of
 * that launch, at most `capacity`).  The sum is what balances GPUs (rt_partition_tiles), the peak what orders
 * a launch: a tile of one frame is a job as long as its longest pixel, and the longest jobs must start first.  RT_ERR_INVALID if no launch of the current view has collected costs.  The reference
 * has no counterpart (one GPU, one thread per pixel, src/dispatch.cu:136-139); this is what lets N GPUs
 * share a frame by cost instead of by area. */
rt_status rt_tile_costs(rt_ctx *ctx, uint32_t *tile_ids, uint32_t *costs, uint32_t *peaks, int32_t capacity, int32_t *count);

/* Ownership of the tiles_x x tiles_y tiles of an image over n_ranks GPUs: owner[ty * tiles_x + tx] = rank.
 * cost == NULL: interleaved, owner = (tx + ty) % n_ranks (what a view's first, cost-collecting launch uses).
 * Otherwise cost[tile] is its measured cost (rt_tile_costs, summed over the ranks that rendered the view)
 * and tiles are dealt longest-processing-time-first: by decreasing cost, each to the rank with the least
 * cost so far (ties: the lower tile index first, the lower rank).  Deterministic: every rank that calls it
 * with the same costs gets the same owners.  Any partition renders the same image (a pixel depends only on
 * its index, seed and previous value, src/raytracer.cu:118-131). */
rt_status rt_partition_tiles(const uint32_t *cost, int32_t tiles_x, int32_t tiles_y, int32_t n_ranks, int32_t *owner);

/* The exchange step for tile lists on one GPU: copies between a compact image (the listed tiles back to
 * back, as a compact tile-list render writes them) and a full W*H*3 frame, both on ctx's GPU, asynchronously
 * on hip_stream.  to_frame != 0: frame <- compact (what the root does with every rank's gathered tiles);
 * to_frame == 0: compact <- frame (handing the image so far to its owners).  tile_list is host memory. */
rt_status rt_tiles_copy_device(rt_ctx *ctx, float *d_compact, float *d_frame, int32_t width, int32_t height,
                               const uint32_t *tile_list, int32_t num_tiles, int32_t to_frame, void *hip_stream);

/* How many progressive frames the multi-frame entry points put into one launch for an image of this size on
 * this context's GPU: at most 32, fewer when that many per-frame planes (12 bytes per pixel each) would take
 * more than a quarter of the GPU's memory. */
int32_t rt_max_batch_frames(rt_ctx *ctx, int32_t width, int32_t height);

/* Kernel timing by HIP events recorded on the launch stream around the render kernel of the
 * most recent rt_render / rt_render_device call; blocks until that kernel has finished. */
rt_status rt_last_kernel_ms(rt_ctx *ctx, float *ms);
/* Blocks until the most recent launch of this context has finished and returns its status: the
 * cudaDeviceSynchronize + cudaPeekAtLastError pair of src/dispatch.cu:141,161 for callers of the
 * asynchronous device-buffer entry points. */
rt_status rt_ctx_synchronize(rt_ctx *ctx);

/* ---- several GPUs of one node from one host thread ---------------------------------------------
 * What run_ray_tracer (src/dispatch.cu:127-153) does on one device, n devices do for the bands they
 * own: rank i of n_ranks renders the bands b with b % n_ranks == i (a pixel depends
 * only on its own coordinates, seed and previous value, so any partition gives the single-GPU image bit
 * for bit) on its own context, asynchronously; the band buffers travel to ranks[0]'s GPU with one peer
 * copy per rank (xGMI) and are de-interleaved there.  There is no reduction, hence no collective.
 * Every rank needs the scene committed on ITS context; a context may appear once.
 * rt_render_multi[_device] own the partition: the first call for a view deals the tiles out interleaved and
 * measures what they cost; from the second call on every rank owns a cost-balanced tile list
 * (rt_partition_tiles), so the ranks finish together. */
typedef struct rt_rank {
    rt_ctx *ctx;
    const rt_scene *scene;
} rt_rank;

/* render() src/dispatch.cu:156-163 for a node, host-buffer form: n_frames passes of the main loop body
 * (frame i seeded with times_ms[i]) accumulated into previous_render; *frame_num advances by n_frames.
 * Same image as rt_render_frames on one GPU. */
rt_status rt_render_multi(const rt_rank *ranks, int32_t n_ranks, const rt_camera *cam, const rt_render_settings *rs,
                          const int32_t *times_ms, int32_t n_frames, int32_t *frame_num, float *previous_render);
/* Device-buffer form: d_frame is a full W*H*3 frame on ranks[0]'s GPU, updated in place (its content is
 * the image after frame_num - 1 when frame_num > 0, ignored otherwise).  band_rows > 0 (a multiple of 8):
 * the static partition of round 2 - rank i owns the bands b % n_ranks == i; band_rows == 0: cost-balanced
 * tile lists as described above.  Asynchronous: the frame is complete in the order of hip_stream (a stream of
 * ranks[0]'s GPU, NULL = default stream); rt_ctx_synchronize on each rank reports kernel errors. */
/* (Diagnosis: with RT_AMD_MULTI_CAREFUL=1 in the environment when the ROOT context is created, the call waits on the host for
 * every stream involved after each of its phases - scatter-out, the ranks' kernels and copies, de-interleave - and an error names
 * the phase: same frames, no overlap.) */
rt_status rt_render_multi_device(const rt_rank *ranks, int32_t n_ranks, const rt_camera *cam, const rt_render_settings *rs,
                                 const int32_t *times_ms, int32_t n_frames, int32_t frame_num, int32_t band_rows,
                                 float *d_frame, void *hip_stream);
/* The exchange step alone, for callers that launch the ranks themselves (rt_render_device[_batch] with a
 * compact tile spec, bands or list): the compact buffer d_bands of the rank `src_tiles` describes, on src's GPU,
 * lands in the full frame d_frame on root's GPU - one peer copy + a de-interleave on root_stream - ordered
 * behind src's most recent launch. */
rt_status rt_gather(rt_ctx *root, float *d_frame, int32_t width, int32_t height, rt_ctx *src, const float *d_bands,
                    const rt_tile_spec *src_tiles, void *root_stream);

/* Direct (xGMI) access from a's GPU to b's memory and back: enables it if need be and reports 1 (peer access is
 * on in both directions: copies between the two go GPU to GPU), 0 (refused or unavailable: the runtime stages
 * them through the host - still correct, slower) or 1 when both contexts share a GPU; < 0: -rt_status. */
int32_t rt_peer_access(rt_ctx *a, rt_ctx *b);

/* float -> RGBA8 display conversion of src/main.cu:343-371 (int(px*255), clamp, alpha 255),
 * on the device: d_rgb W*H*3 floats -> d_rgba W*H*4 bytes */
rt_status rt_to_rgba8_device(rt_ctx *ctx, const float *d_rgb, int32_t width, int32_t height, uint8_t *d_rgba, void *hip_stream);

/* ---- introspection (tests only): the flattened device layout of a builder ---------------- */
/* Pointers stay valid until the next rt_debug_flatten call on the same builder or its
 * destruction.  blob is the LDS-staged part in 16-byte units (see
 * ray-tracer_amd/csrc/rt_device_scene.h); objects is the scalar-loaded object table. */
typedef struct rt_flat_view {
    const float *blob; int32_t blob_f4;
    int32_t off_nodes, off_tris, off_objlds, off_meshes, num_meshes, stack_entries;
    const void *objects; int32_t num_objects, object_stride;
    const float *tri_uv; int32_t num_triangles, num_nodes, has_mesh;
} rt_flat_view;
rt_status rt_debug_flatten(rt_scene_builder *b, rt_flat_view *out);
/* evaluates one function of the shared math / RNG headers ON THE DEVICE, element-wise, on raw
 * 32-bit patterns (host pointers).  op: 0 rt_logf, 1 rt_cosf, 2 rt_sinf, 3 rt_asinf, 4 rt_acosf,
 * 5 rt_u01, 6 rt_jitter, 7 rt_theta (5-7 take the uint32 hash output), 8 sqrtf, 9 1.0f/x,
 * 10 (float)rt_pow5, 11 / 12 the guarded short 1 / x and sqrt, 13 / 15 rt_logf_0_1(rt_u01(hash)) with the short / the operator's division, 14 rt_cosf_0_2pi(rt_theta(hash)):
 * the Box-Muller calls on a hash output */
rt_status rt_debug_eval(rt_ctx *ctx, int32_t op, const uint32_t *in, uint32_t *out, int32_t n);
/* test hook: the device code's short 1 / x and sqrt(x) against the compiler's IEEE expansions for all 2^32 inputs; out4 = {differing
 * inputs inside the reciprocal's range, inputs inside it, the same two for the square root}.  (The reference divides and takes roots
 * with CUDA's IEEE operators, src/utils.cu:118-128, src/objects.cu:40-79,135-163; the short forms are the same function there.) */
rt_status rt_debug_exhaustive(rt_ctx *ctx, unsigned long long *out4);
/* 48 section counters / timers of a development build compiled with -DRT_STATS (all zero otherwise) */
rt_status rt_debug_read_stats(rt_ctx *ctx, unsigned long long *out48);

const char *rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
