/*
 * rt_device_scene.h — the compact scene layout the HIP kernel reads, and the kernel argument
 * block.  Written by the host flattener (rt_host.cpp), read by rt_kernel.hip.
 *
 * The reference keeps the scene as an array of 5,328-byte `Object` unions, 216-byte triangles
 * and 3,616-byte BVH nodes with one device allocation per node (SURVEY.md §8 sizes).  Here:
 *
 *   LDS-staged blob (one coalesced copy per workgroup, 16-byte units), in this order:
 *     nodes   64 B  both children's boxes + child references (one LDS round trip per step)
 *     objlds  64 B  per top-level object: what shading needs for a per-lane object index
 *     meshes  32 B  per mesh object: root box, root reference, object index
 *     objtab  48 B  the object list (rt_object) every lane walks in the same order: wave-uniform
 *                   LDS reads (broadcast) instead of vector global loads
 *     tris    48 B  p0, side1, side2, unit normal            (reference src/objects.cu:175-186)
 *   A scene that does not fit a CU's LDS keeps everything before the triangles there when that fits (a depth-10 tree is
 *   at most 1,023 nodes = 64 KB, whatever the triangle count) and reads the triangles from global memory (L2); otherwise
 *   the kernel reads every section from global memory (RT_SCENE_* below).
 *   global memory, read once per textured hit:
 *     tri_uv  24 B  per-triangle texture coordinates (only when a material needs UVs)
 *
 * The BVH is the reference's tree (fixed depth 10, same split order, same leaf order), with
 * every node's box stored in its parent.  Nodes with an empty child (the reference's split
 * sends everything left once a node holds <= 2 triangles) are not stored: the edge into such
 * a chain points at its end and carries RT_REF_CHAIN, which preserves the one observable
 * effect of the chain (a strict distance test).  Empty subtrees are never entered by a ray
 * whose direction is not NaN (their (0,0,0)-(0,0,0) box fails the strict slab test), and a NaN
 * direction hits nothing, so the kernel answers NaN rays without traversing.
 */
#ifndef RT_DEVICE_SCENE_H
#define RT_DEVICE_SCENE_H

#include <stdint.h>

#define RT_BVH_DEPTH 10              /* reference src/objects.cu:786 */
/* where the render kernel reads the scene from (rt_scene_info.scene_in_lds) */
#define RT_SCENE_GLOBAL 0            /* every section from global memory; LDS holds only the traversal stacks */
#define RT_SCENE_LDS 1               /* the whole blob staged into LDS */
#define RT_SCENE_HYBRID 2            /* everything but the triangles in LDS, the triangles from global memory */
#define RT_STACK_ENTRIES RT_BVH_DEPTH /* at most one pending sibling per level below the root */
#define RT_FRAME_BITS 5               /* a pixel keeps the index of its frame within the launch in this many bits */
#define RT_MAX_BATCH_FRAMES (1 << RT_FRAME_BITS)   /* frames one launch can render */
#ifndef RT_SMALL_WG_WAVES
#define RT_SMALL_WG_WAVES 5            /* workgroups of fewer than 1024 threads are compiled for this many waves per SIMD (<= 96 VGPRs) */
#endif
/* Defaults of the render kernel's scheduling thresholds (lanes of a wave; see rt_kernel.hip; RT_AMD_* overrides them).
 * None of them changes an image.  Values: same-box sweeps over four scenes in the multi-frame regime,
 * profiles/r02/experiments/.  (Compiling them in as immediates instead of launch arguments was measured: no difference.) */
#define RT_DEF_WORK_THRESHOLD 4      /* traversal steps run while at least this many lanes traverse (4 against 8, round 3: -0.6 % monkey, -0.9 % cube, +-0 reference scene 0, an eighth of the image -1 %) */
#define RT_DEF_READY_BREAK 44        /* ... unless this many lanes have cheap work (generate / fetch / next mesh / a miss); 44 against 40 on round 4's final code: -0.3 % monkey, -0.6 % cube, +0.1 % reference scene 0 */
#define RT_DEF_HIT_BREAK 32          /* ... or this many hold a hit to shade (24 until the end of round 4: on its final code, where a shade pass is ~15 % cheaper and the traversal
                                         rounds between two of them count for more, 32 is -7.4 % on reference scene 0, -1.1 % cube, -0.3 % monkey: profiles/r04/experiments/knobs_final_build.txt) */
#define RT_DEF_HIT_LOW 16            /* ... or at least this many hold a hit and, with the cheap-work lanes, they are */
#define RT_DEF_MIX_BREAK 36          /*     this many together.  Re-swept on round 4's final code (the shade pass got cheaper, smaller batches pay sooner;
                                         profiles/r04/experiments/knobs_final_build.txt): 36 with ready_break 44 is +-0 on the cube's 256-thread workgroups ... */
#define RT_DEF_MIX_BREAK_1024 32     /* ... and the 1024-thread kernels take 32: monkey -1.6 %, 50,880-triangle surface -1.7 %, 6,000-triangle soup -0.4 %, reference
                                         scenes 0 / 1 / 3 within 0.3 % (the cube would lose 0.5 % with it) */
#define RT_DEF_DESCEND_KEEP 20       /* the descend loop ends once fewer than this many 64ths of its lanes remain (24 until the end of round 4; with hit_break 32: monkey -0.5 %, reference scene 0 -8.0 %,
                                         50,880-triangle surface -1.3 %, cube -0.6 % against the defaults before) */
#define RT_DEF_SHADE_BATCH 40        /* scenes without a mesh: hits are shaded once this many lanes hold one */
/* What a pixel is charged for (the tile costs a view's first launch collects: longest-job-first schedule, cost-balanced
 * tile ownership over GPUs): per traversal macro step, per generated bounce ray, per shaded hit.  The ratios are what was
 * fitted to the kernel times of 1/2/4/8-rank shares of the monkey run (tools/fit_cost_weights.py,
 * profiles/r03/experiments/cost_weights_driver_shape.txt, yield_cadence_and_cost_weights.txt: any weighting of this
 * shape predicts a rank's time to 1.6-1.8 % rms); round 4 divided round 3's (4, 3, 8) by four: a pixel's cost
 * lives in bits 30..RT_FRAME_BITS of its frame word, 2^26 units, and at one unit per macro step that is more than any
 * pixel the host lets collect costs can reach (rt_capi.cpp: launches of rays_per_pixel * reflection_limit >= 2^16 run on
 * the previous or the guessed schedule).  Only frame 0 of a launch is charged to its tile, capped at RT_COST_PIXEL_CAP, so
 * that a tile's 32-bit sum of 64 pixels (<< 1, bit 0 = "a ray entered a mesh") cannot wrap either. */
#ifndef RT_COST_STEP
#define RT_COST_STEP 1
#endif
#ifndef RT_COST_GEN
#define RT_COST_GEN 1
#endif
#ifndef RT_COST_HIT
#define RT_COST_HIT 2
#endif
#define RT_COST_PIXEL_CAP 0x00ffffffu   /* 64 pixels x 2^24 x 2 = 2^31 */
#define RT_JOB_FRAME_SHIFT 22          /* a job = tile | frame << 22 (2^28 pixels are 2^22 tiles) */
#define RT_JOB_TILE_MASK 0x003fffffu
#define RT_INF_F 1073741824.0f       /* reference `1 << 31 - 1` == 1 << 30, src/objects.cu:6 */
#define RT_EPS_F 0.000001f           /* FLOAT_PRECISION_ERROR src/objects.cu:7 */

/* child / root references: bit 31 leaf, bit 30 chain, leaves: bits 29..20 count, 19..0 first
 * triangle; internal nodes: bits 29..0 node index.
 * CHAIN marks an edge that stands for one or more collapsed single-child nodes.  In the
 * reference such a node re-tests the same box for its only non-empty child and pushes it only
 * if dist < best (strict, src/objects.cu:517), so an entry reached through a chain is taken
 * iff dist < best, where a plain popped entry is taken iff !(dist > best) (:501). */
#define RT_REF_LEAF 0x80000000u
#define RT_REF_CHAIN 0x40000000u
#define RT_REF_COUNT_SHIFT 20
#define RT_REF_COUNT_MAX 1023u
#define RT_REF_START_MASK 0x000fffffu
#define RT_REF_NODE_MASK 0x3fffffffu
#define RT_REF_EMPTY_LEAF RT_REF_LEAF          /* leaf with zero triangles */

enum { RT_OBJ_SPHERE = 0, RT_OBJ_TRIANGLE = 1, RT_OBJ_QUAD = 2, RT_OBJ_ONE_WAY_QUAD = 3, RT_OBJ_CUBOID = 4, RT_OBJ_MESH = 5 };   /* src/objects.cu:804-809 */

/* material tags as stored in rt_objlds.b.w bits [1:0] (reference src/material.cu:131-133) */
#define RT_DEV_MAT_STANDARD 0
#define RT_DEV_MAT_EMISSIVE 1
#define RT_DEV_MAT_REFRACTIVE 2

/* 16-byte aligned so LDS / global accesses become single b128 instructions */
typedef struct __attribute__((aligned(16))) { float x, y, z, w; } rt_f4;

/* q0 = (lmin.x lmin.y lmin.z lmax.x) q1 = (lmax.y lmax.z rmin.x rmin.y) q2 = (rmin.z rmax.x rmax.y rmax.z)
 * q3 = (lref, rref, 0, 0) as raw bits */
typedef struct { rt_f4 q[4]; } rt_node;

/* q0 = (p0.x p0.y p0.z s1.x) q1 = (s1.y s1.z s2.x s2.y) q2 = (s2.z n.x n.y n.z) */
typedef struct { rt_f4 q[3]; } rt_tri;

/* a = (A.rgb, smoothness): A = colour (COLOUR), light (CHECKERBOARD) or, as raw ints,
 *     (width, height, first float in tex_data) (IMAGE)
 * b = (B.rgb, packed): B = emitted light (EMISSIVE) or dark (CHECKERBOARD)
 *     packed bits: [1:0] material type, [3:2] texture type, [4] need_uv, [5] is_sphere, [31:8] num_squares
 * c = sphere (center.xyz, radius)
 * d = (refractive index, 0, 0, 0) */
typedef struct { rt_f4 a, b, c, d; } rt_objlds;
#define RT_OBJLDS_F4 4

#define RT_PACK_MAT(type, tex, need_uv, is_sphere, nsq) \
    ((uint32_t)(type) | ((uint32_t)(tex) << 2) | ((uint32_t)(need_uv) << 4) | ((uint32_t)(is_sphere) << 5) | ((uint32_t)(nsq) << 8))

/* SPHERE: v = center.xyz, radius
 * TRIANGLE / QUAD / CUBOID: prim_start = first triangle (1 / 2 / 12 of them)
 * ONE_WAY_QUAD: v[0..2] = its normal (t1.normal * multiplier, src/objects.cu:285-289)
 * MESH: root_ref, v[0..5] = root box min/max */
typedef struct {
    int32_t type;
    int32_t prim_start;
    int32_t need_uv;
    uint32_t root_ref;
    float v[8];
} rt_object;

typedef struct {
    float cam[12];                 /* cam_pos, tl_pixel_pos, delta_u, delta_v  (src/camera.cu:12-21) */
    int32_t width, height;
    int32_t rays_per_pixel, reflection_limit, antialias;   /* RenderData src/raytracer.cu:4-12 */
    float sky[3];
    uint32_t seeds[RT_MAX_BATCH_FRAMES];   /* per frame of the launch: (uint32)time_ms * 6291469  (src/raytracer.cu:127) */
    int32_t num_frames;            /* progressive frames rendered by this launch (1: a plain frame) */
    int32_t frame_num;             /* frame_num of the first of them */
    float *partial;                /* multi-frame (in-place) launches: num_frames planes of `partial_plane` pixels, one per frame,
                                      each receiving that frame's per-pixel mean colour; rt_blend_kernel folds them into `out`.
                                      NULL for a plain frame */
    int64_t partial_plane;
    /* tile assignment */
    int32_t band_rows, band_first, band_stride, compact;
    int32_t tiles_x;               /* 8x8 tiles per row of tiles */
    int32_t num_tiles;             /* tiles this launch renders */
    const uint32_t *tile_list;     /* or NULL (bands): local tile t is the image's tile tile_list[t] = ty * tiles_x + tx; with `compact`
                                      the output holds the listed tiles back to back, 64 pixels each, row-major inside a tile */
    uint32_t tile_stride;          /* ticket t renders tile (t * tile_stride) % num_tiles; coprime to num_tiles */
    const uint32_t *tile_order;    /* or, if not NULL, tile tile_order[t] (expensive-looking tiles first) */
    int32_t num_heavy_tiles;       /* multi-frame launches: this many leading entries of tile_order go first for ALL frames */
    const uint32_t *job_order;     /* or, if not NULL, the whole schedule of a multi-frame launch: ticket t renders tile
                                      (job_order[t] & RT_JOB_TILE_MASK) of frame (job_order[t] >> RT_JOB_FRAME_SHIFT) */
    /* scene */
    const rt_object *objects;
    int32_t num_objects;
    const rt_f4 *blob;             /* LDS-staged part */
    int32_t blob_f4;               /* its size in 16-byte units */
    int32_t off_nodes, off_tris, off_objlds, off_meshes, off_objtab;   /* section offsets in 16-byte units */
    int32_t num_meshes;
    int32_t stack_entries;         /* per-lane traversal stack depth (LDS) */
    int32_t work_threshold;        /* run traversal steps while at least this many lanes traverse */
    int32_t descend_keep;          /* leave the descend loop when fewer than descend_keep/64 of its lanes remain */
    int32_t ready_break;           /* ... unless at least this many lanes are ready to shade / generate */
    int32_t hit_break;             /* ... or this many hold a hit to shade */
    int32_t hit_low, mix_break;    /* ... or at least hit_low hold a hit and hits + cheap-work lanes together reach mix_break */
    int32_t shade_batch;           /* scenes without a mesh: hits are shaded once this many lanes hold one */
    const float *tri_uv;           /* 6 floats per triangle, or NULL */
    const float *tex_data;         /* IMAGE texture texels (rgb floats), or NULL */
    /* frame buffers */
    const float *prev;             /* full frame or NULL */
    float *out;
    uint32_t *tile_counter;        /* zeroed before the launch */
    uint32_t *tile_cost;           /* or NULL: per tile, what its pixels cost (RT_COST_* units, all frames of the launch) << 1, bit 0: a
                                      ray of the tile entered a mesh */
    uint32_t *tile_peak;           /* with tile_cost: per tile, the cost of its most expensive pixel (of any one frame) */
    unsigned long long *stats;     /* development builds only (-DRT_STATS): section counters */
} rt_kernel_args;

#endif
