/*
 * rt_capi.cpp — the device half of the C ABI (include/rt_amd.h): context, scene upload,
 * render launches, tile ownership over GPUs, timing.  Replaces the reference's host<->device seam
 * (src/dispatch.cu:104-163): where the reference allocates and frees four device buffers per
 * frame and copies the frame element-wise out of managed memory, the context here keeps
 * persistent HBM frame buffers and the device-buffer entry point takes caller-owned HBM.
 *
 * There is no CPU fallback: rt_ctx_create fails when HIP has no device.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <set>
#include <queue>
#include <string>
#include <vector>

#include "rt_host.h"

extern "C" hipError_t rt_launch_render(const rt_kernel_args *args, int has_mesh, int scene_in_lds, int threads, int blocks, size_t lds_bytes, hipStream_t stream);
extern "C" int rt_kernel_blocks_per_cu(int has_mesh, int scene_in_lds, int threads, size_t lds_bytes);
extern "C" hipError_t rt_launch_blend(const float *partial, long long plane_floats, int num_frames, int frame_num, float *frame, long long n_floats, hipStream_t stream);
extern "C" hipError_t rt_launch_blend_tiles(const float *partial, long long plane_floats, int num_frames, int frame_num, float *frame,
                                            const uint32_t *tile_list, int n_tiles, int tiles_x, int W, int H, hipStream_t stream);
extern "C" hipError_t rt_launch_tiles_copy(float *compact, float *frame, const uint32_t *tile_list, int n_tiles, int tiles_x, int W, int H, int to_frame, hipStream_t stream);
extern "C" hipError_t rt_launch_eval(int op, const uint32_t *in, uint32_t *out, int n, hipStream_t stream);
extern "C" hipError_t rt_launch_rgba8(const float *rgb, int n_pixels, uint8_t *out, hipStream_t stream);

#define RT_LDS_LIMIT 163840   /* 160 KiB per CU / per workgroup on gfx950 */
#define RT_MAX_BLOCKS_PER_CU 6

namespace {

/* a tile list on the device, found again by content (lists are a few KB to ~130 KB and change only with the view).
 * `host` is a pinned copy: what a hash hit is compared with (a 64-bit hash alone would render the wrong
 * tiles on a collision) and what the asynchronous upload reads.  An entry is recycled least-recently-used first; the
 * upload of its new content is ordered behind every launch that read the old one by events, never by a host wait. */
struct DevList {
    uint64_t hash = 0;
    int n = 0;
    uint32_t *d = nullptr;
    uint32_t *host = nullptr;            /* pinned, `cap` entries */
    size_t cap = 0;
    uint64_t seq = 0;                    /* rt_ctx::list_seq when last handed out (LRU order; > pin_floor: not recyclable) */
    hipStream_t stream = nullptr;        /* the stream of the launches it was last handed out for */
    hipEvent_t ev = nullptr;             /* scratch: "everything queued on `stream` so far" (orders a reuse on another stream) */
    hipEvent_t ev_up = nullptr;          /* "the upload from `host` is done": the pinned copy may be rewritten */
};

/* the root's landing area for one source context's compact image (rt_gather, rt_render_multi_device) */
struct Stage {
    float *d = nullptr;
    size_t cap = 0;                      /* floats */
    hipEvent_t ev_free = nullptr;        /* recorded on the root's stream after the last de-interleave that read the area */
    bool used = false;
};

/* rt_render_multi[_device] with cost-balanced tile lists: what the root remembers between calls */
struct MultiState {
    std::vector<uint32_t> key;           /* camera, image size, number of ranks, scene ids */
    int stage = 0;                       /* 0 nothing; 1 the interleaved lists are in use and a call has collected costs; 2 balanced */
    std::vector<std::vector<uint32_t>> lists, costs, peaks;   /* per rank: its tiles (image indices), their measured costs and peak pixel costs */
};

}  // namespace

/* One frame in flight of the pipelined per-frame entry points (rt_frame_submit / rt_frame_collect): everything a launch
 * writes that the context otherwise holds once - its stream, its ticket counter, the plane its pixels' means go to. */
struct FrameSlot {
    hipStream_t stream = nullptr;
    uint32_t *counter = nullptr;        /* 1 KB like rt_ctx::tile_counter */
    float *plane = nullptr;
    size_t plane_cap = 0;               /* floats */
    hipEvent_t ev_done = nullptr;       /* the frame's render kernel has finished (recorded on `stream`) */
    hipEvent_t ev_free = nullptr;       /* the blend that read the plane has finished (recorded on `stream` too: the blend runs there) */
    hipEvent_t ev_call = nullptr;       /* where the collector's stream stood when it asked for the frame */
    bool used = false, folded = false;  /* ev_done / ev_free have been recorded at least once */
    /* what rt_frame_collect needs to fold the plane into the caller's frame (the launch's output layout) */
    size_t plane_floats = 0;
    int compact = 0, listed = 0, band_rows = 8, band_first = 0, band_stride = 1, width = 0, height = 0, tiles_x = 0, n_tiles = 0;
};

struct Pipeline {
    FrameSlot slots[RT_PIPELINE_DEPTH];
    int order[RT_PIPELINE_DEPTH];       /* submitted and not collected, oldest first */
    int pending = 0;
    int depth = RT_PIPELINE_DEFAULT_DEPTH;   /* rt_frame_depth: how many frames the caller keeps in flight */
    int last_fold = -1;                 /* the slot whose frame was folded last (folds into one frame happen in collection order) */
    uint32_t *d_list = nullptr;         /* the view's tile list (listed tile specs), shared by the frames in flight */
    size_t list_cap = 0;
    std::vector<uint32_t> list_host;
};

struct rt_ctx {
    int device = 0;
    int num_cus = 0;
    size_t total_mem = 0;
    std::string err;
    uint32_t *tile_counter = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool have_timing = false;
    /* The launch scratch below (ticket counter, per-frame planes, tile order / costs, the two events) is
     * per context, so launches of one context are kept in order: a launch on another stream than the
     * previous one first waits for that one's stop event (one launch in flight per context). */
    hipStream_t last_stream = nullptr;
    bool launched = false;
    /* multi-GPU entry points (rt_render_multi*): this rank's compact buffer, its stream and "my image is in the root's
     * staging area" event; on the root also the staging areas, one per source context */
    float *d_bands = nullptr;
    size_t bands_cap = 0;                /* floats */
    std::map<rt_ctx *, Stage> stages;
    hipStream_t multi_stream = nullptr;
    hipEvent_t ev_multi = nullptr;
    MultiState multi;
    std::map<int, int> peer_ok;          /* device -> 1 direct access both ways, 0 refused */
    /* persistent frame buffers for the host-buffer entry point */
    float *d_prev = nullptr, *d_out = nullptr;
    size_t frame_bytes = 0;
    /* ---- the current view: (scene, camera, image, tile spec).  Its tiles, their order and what they cost ---- */
    std::vector<uint32_t> order_key;     /* what the view state below was built for */
    std::vector<uint32_t> tiles_host;    /* local tile t of the launch -> the tile's index in the image */
    std::vector<uint32_t> order_host;    /* ticket -> local tile (host copy of d_tile_order), when the view uses an order */
    bool use_order = false;
    uint32_t *d_tile_order = nullptr;
    size_t tile_order_cap = 0;
    uint32_t *d_tile_cost = nullptr;     /* per tile: the weighted work of its pixels, collected by the first launch of a view */
    size_t tile_cost_cap = 0;
    uint32_t *d_tile_peak = nullptr;     /* ... and of its most expensive pixel */
    size_t tile_peak_cap = 0;
    int cost_spp = 0;                    /* rays_per_pixel of the launch the figures come from (a pilot: 1) */
    int cost_state = 0;                  /* 0 nothing, 1 a launch of this view collected costs, 2 cost_host holds them (and the order is refined) */
    std::vector<uint32_t> cost_host;     /* the measured (or caller-supplied) costs, per local tile */
    std::vector<uint32_t> peak_host;     /* ... and peak pixel costs: what the schedule sorts by */
    int order_num_heavy = 0;             /* refined order: how many leading tiles go first for ALL frames of a multi-frame launch */
    float *d_partial = nullptr;          /* multi-frame launches: one plane of per-pixel frame means per frame */
    size_t partial_cap = 0;              /* floats */
    /* multi-frame launches, once the tile costs of the view are known: the whole schedule (ticket -> tile, frame),
     * longest job first over all frames (see build_job_order) */
    uint32_t *d_job_order = nullptr;
    size_t job_cap = 0;
    int job_frames = 0;                  /* what the uploaded schedule was built for (with order_key) */
    std::vector<DevList> dev_lists;
    size_t dev_list_cap = 64;            /* cached tile lists; beyond that the least recently used entry is recycled (raised when one call pins more) */
    uint64_t list_seq = 0, pin_floor = ~0ull;   /* entries handed out after pin_floor are held by the running call (rt_render_multi_device) */
    /* knobs (RT_AMD_*), none changes an image */
    int heavy_top = 1024;                /* RT_AMD_HEAVY_TOP: the refined order moves that many tiles at most to the front (0 = never refine) */
    int lpt = 1;                         /* RT_AMD_LPT=0: the round-1 ticket order (heavy tiles of frame 0, 1, ... first) */
    int lpt_top = 1 << 30;               /* RT_AMD_LPT_TOP: at most this many tiles (most expensive first) are scheduled by cost */
    int pilot = 1, pilot_min_spp = 32;   /* RT_AMD_PILOT=0: no pilot launch for a new view; RT_AMD_PILOT_MIN_SPP: only for at least this many samples per pixel */
    int lpt_by_peak = 1, top_by_peak = 0; /* RT_AMD_LPT_BY_PEAK: the multi-frame schedule sorts by the tile's peak pixel cost (0: by its summed cost); RT_AMD_TOP_BY_PEAK: so does the
                                          * one-frame order (default 0: by the sum - same-box A/B, one 1080p frame: monkey 509 vs 526 ms, cube 171 vs 178, reference scene 0 3,104 vs 3,083) */
    int heavy_first = 1;                 /* RT_AMD_HEAVY_FIRST=0 disables */
    int work_threshold = RT_DEF_WORK_THRESHOLD;      /* lanes; RT_AMD_WORK_THRESHOLD */
    int descend_keep = RT_DEF_DESCEND_KEEP;       /* RT_AMD_DESCEND_KEEP (0..64): 0 = run every descent to its end */
    int tile_scatter = 1;        /* RT_AMD_TILE_SCATTER=0: hand tiles out in raster order */
    int ready_break = RT_DEF_READY_BREAK;        /* lanes; RT_AMD_READY_BREAK; 65 = never */
    int hit_break = RT_DEF_HIT_BREAK;          /* lanes; RT_AMD_HIT_BREAK */
    int hit_low = RT_DEF_HIT_LOW, mix_break = -1;                 /* RT_AMD_HIT_LOW, RT_AMD_MIX_BREAK (0 = that rule off; -1 = not set: the default of the scene's workgroup shape) */
    int shade_batch = RT_DEF_SHADE_BATCH;        /* lanes; RT_AMD_SHADE_BATCH (1..64) */
    Pipeline pipe;
    int pipe_share = 2;                          /* RT_AMD_PIPE_SHARE: pipelined frames run on 1 / depth of the CUs - 0 never, 1 always, 2 when a workgroup fills a CU */
    int multi_careful = 0;                       /* RT_AMD_MULTI_CAREFUL=1: rt_render_multi_device waits on the host after every phase (diagnosis) */
};

struct rt_scene {
    rt_ctx *ctx = nullptr;
    rt_f4 *d_blob = nullptr;
    rt_object *d_objects = nullptr;
    float *d_tri_uv = nullptr;
    float *d_tex = nullptr;
    FlatScene flat;          /* host copy (sizes, offsets) */
    int threads = 0;         /* workgroup size chosen for this scene */
    int blocks_per_cu = 1;   /* ... and how many of them are resident on a CU */
    int scene_in_lds = RT_SCENE_LDS;    /* RT_SCENE_*: all of the scene in LDS, all but the triangles, or nothing */
    uint32_t uid = 0;        /* distinguishes scenes in the tile-order cache (addresses get reused) */
    size_t lds_bytes = 0;
};

namespace {

/* the live contexts: a root keeps a landing area per SOURCE context (rt_ctx::stages), which has to go when the source does */
std::mutex g_ctx_mutex;
std::set<rt_ctx *> g_live_ctxs;

/* check_cuda_error src/utils.cu:5-10 */
rt_status hip_fail(rt_ctx *ctx, hipError_t e, const char *what)
{
    if (ctx) ctx->err = std::string("Error from HIP (") + what + "): " + hipGetErrorString(e);
    return RT_ERR_HIP;
}

#define RT_HIP(ctx, call, what)                         \
    do {                                                \
        hipError_t e_ = (call);                         \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, what); \
    } while (0)

rt_status set_err(rt_ctx *ctx, rt_status code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

uint64_t fnv1a(const void *data, size_t bytes, uint64_t h = 1469598103934665603ull)
{
    const unsigned char *p = (const unsigned char *)data;
    for (size_t i = 0; i < bytes; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

/* A call that holds several lists at once (rt_render_multi_device: every rank's list on the root) pins what it is handed
 * from here on: pinned entries are not recycled, the cache grows instead.  unpin when the call has queued its work. */
void pin_tile_lists(rt_ctx *ctx) { ctx->pin_floor = ctx->list_seq; }
void unpin_tile_lists(rt_ctx *ctx) { ctx->pin_floor = ~0ull; }

void free_dev_list(DevList &dl)
{
    if (dl.d) (void)hipFree(dl.d);
    if (dl.host) (void)hipHostFree(dl.host);
    if (dl.ev) (void)hipEventDestroy(dl.ev);
    if (dl.ev_up) (void)hipEventDestroy(dl.ev_up);
    dl = DevList();
}

/* `list` (host) on ctx's GPU, for launches the caller queues on `stream` after this call.  A list seen before costs a hash
 * and a compare.  A new one takes a free entry or recycles the least recently used one: its content goes into the entry's
 * pinned buffer and is uploaded asynchronously on `stream`, behind (hipStreamWaitEvent) whatever was queued on the stream
 * the entry last served - no host synchronisation on the steady-state path, and no device-wide one ever (round 3 emptied a
 * full cache after hipDeviceSynchronize and waited for every upload).  The pointer stays valid until the entry
 * is recycled: after dev_list_cap other lists, and never while pinned. */
rt_status device_tile_list(rt_ctx *ctx, const uint32_t *list, int n, hipStream_t stream, const uint32_t **out)
{
    *out = nullptr;
    if (n <= 0) return RT_OK;
    const uint64_t h = fnv1a(list, (size_t)n * 4);
    for (DevList &dl : ctx->dev_lists)
        if (dl.hash == h && dl.n == n && std::memcmp(dl.host, list, (size_t)n * 4) == 0) {
            if (dl.stream != stream) {
                /* handed to another stream than the one its upload (or last use) was queued on: order the two */
                RT_HIP(ctx, hipEventRecord(dl.ev, dl.stream), "recording a tile list's use");
                RT_HIP(ctx, hipStreamWaitEvent(stream, dl.ev, 0), "ordering a tile list's use");
                dl.stream = stream;
            }
            dl.seq = ++ctx->list_seq;
            *out = dl.d;
            return RT_OK;
        }
    DevList *slot = nullptr;
    if (ctx->dev_lists.size() >= ctx->dev_list_cap) {
        for (DevList &dl : ctx->dev_lists)
            if ((ctx->pin_floor == ~0ull || dl.seq <= ctx->pin_floor) && (!slot || dl.seq < slot->seq)) slot = &dl;
        if (!slot) ctx->dev_list_cap = ctx->dev_lists.size() + 1;      /* everything is pinned by the running call: grow */
    }
    if (!slot) {
        ctx->dev_lists.emplace_back();
        slot = &ctx->dev_lists.back();
        RT_HIP(ctx, hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming), "creating a tile list's event");
        RT_HIP(ctx, hipEventCreateWithFlags(&slot->ev_up, hipEventDisableTiming), "creating a tile list's event");
    } else {
        /* recycle: the new upload must not overtake a queued launch that still reads the old content (same stream: stream
         * order; another stream: an event), and the pinned copy is rewritten only once ITS last upload is done - which it
         * has been since dev_list_cap lists ago, so that wait returns at once */
        RT_HIP(ctx, hipEventRecord(slot->ev, slot->stream), "recording a tile list's last use");
        if (slot->stream != stream) RT_HIP(ctx, hipStreamWaitEvent(stream, slot->ev, 0), "ordering a tile list's reuse");
    }
    if (slot->cap < (size_t)n) {
        /* a larger buffer: the old one goes once the launches reading it are done (the event just recorded) */
        if (slot->d) { (void)hipEventSynchronize(slot->ev); (void)hipFree(slot->d); (void)hipHostFree(slot->host); slot->d = nullptr; slot->host = nullptr; }   /* (rare: lists of one image size share a capacity) */
        const size_t cap = (size_t)n + (size_t)n / 4 + 64;
        RT_HIP(ctx, hipMalloc((void **)&slot->d, cap * 4), "allocating a tile list");
        hipError_t e = hipHostMalloc((void **)&slot->host, cap * 4, hipHostMallocDefault);
        if (e != hipSuccess) { (void)hipFree(slot->d); slot->d = nullptr; slot->cap = 0; slot->n = 0; slot->hash = 0; return hip_fail(ctx, e, "allocating a tile list's staging copy"); }
        slot->cap = cap;
    } else if (slot->n > 0) {
        (void)hipEventSynchronize(slot->ev_up);
    }
    std::memcpy(slot->host, list, (size_t)n * 4);
    slot->hash = h; slot->n = n; slot->stream = stream; slot->seq = ++ctx->list_seq;
    hipError_t e = hipMemcpyAsync(slot->d, slot->host, (size_t)n * 4, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipEventRecord(slot->ev_up, stream);
    if (e != hipSuccess) { slot->n = 0; slot->hash = 0; return hip_fail(ctx, e, "uploading a tile list"); }
    *out = slot->d;
    return RT_OK;
}

rt_status grow(rt_ctx *ctx, float **buf, size_t *cap, size_t need, const char *what)
{
    if (*cap >= need && *buf) return RT_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr; *cap = 0;
    RT_HIP(ctx, hipMalloc((void **)buf, (need ? need : 4) * 4), what);
    *cap = need;
    return RT_OK;
}

rt_status grow_u32(rt_ctx *ctx, uint32_t **buf, size_t *cap, size_t need, const char *what)
{
    if (*cap >= need && *buf) return RT_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr; *cap = 0;
    RT_HIP(ctx, hipMalloc((void **)buf, (need ? need : 1) * 4), what);
    *cap = need;
    return RT_OK;
}

int batch_cap(const rt_ctx *ctx, size_t plane_floats)
{
    const size_t plane_bytes = plane_floats * 4;
    if (plane_bytes == 0 || ctx->total_mem == 0) return RT_MAX_BATCH_FRAMES;
    const size_t k = (ctx->total_mem / 4) / plane_bytes;
    return k >= RT_MAX_BATCH_FRAMES ? RT_MAX_BATCH_FRAMES : (k < 1 ? 1 : (int)k);
}

}  // namespace

extern "C" const char *rt_version(void) { return "ray-tracer_amd 0.3 (gfx950)"; }

extern "C" rt_status rt_ctx_create(int32_t device, rt_ctx **out)
{
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return RT_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return RT_ERR_INVALID;
    rt_ctx *ctx = new (std::nothrow) rt_ctx();
    if (!ctx) return RT_ERR_NOMEM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return RT_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete ctx; return RT_ERR_NO_DEVICE; }
    ctx->num_cus = prop.multiProcessorCount;
    ctx->total_mem = prop.totalGlobalMem;
    if (const char *e = getenv("RT_AMD_WORK_THRESHOLD")) { int v = atoi(e); if (v >= 1 && v <= 64) ctx->work_threshold = v; }
    if (const char *e = getenv("RT_AMD_DESCEND_KEEP")) { int v = atoi(e); if (v >= 0 && v <= 64) ctx->descend_keep = v; }
    if (const char *e = getenv("RT_AMD_TILE_SCATTER")) ctx->tile_scatter = atoi(e) != 0;
    if (const char *e = getenv("RT_AMD_HEAVY_FIRST")) ctx->heavy_first = atoi(e) != 0;
    if (const char *e = getenv("RT_AMD_HEAVY_TOP")) { int v = atoi(e); if (v >= 0) ctx->heavy_top = v; }
    if (const char *e = getenv("RT_AMD_LPT")) ctx->lpt = atoi(e) != 0;
    if (const char *e = getenv("RT_AMD_LPT_TOP")) { int v = atoi(e); if (v >= 0) ctx->lpt_top = v; }
    if (const char *e = getenv("RT_AMD_PILOT")) ctx->pilot = atoi(e) != 0;
    if (const char *e = getenv("RT_AMD_PILOT_MIN_SPP")) { int v = atoi(e); if (v >= 1) ctx->pilot_min_spp = v; }
    if (const char *e = getenv("RT_AMD_LPT_BY_PEAK")) ctx->lpt_by_peak = atoi(e) != 0;
    if (const char *e = getenv("RT_AMD_TOP_BY_PEAK")) ctx->top_by_peak = atoi(e) != 0;
    if (const char *e = getenv("RT_AMD_HIT_BREAK")) { int v = atoi(e); if (v >= 1 && v <= 65) ctx->hit_break = v; }
    if (const char *e = getenv("RT_AMD_HIT_LOW")) { int v = atoi(e); if (v >= 0 && v <= 65) ctx->hit_low = v; }
    if (const char *e = getenv("RT_AMD_MIX_BREAK")) { int v = atoi(e); if (v >= 0 && v <= 130) ctx->mix_break = v; }
    if (const char *e = getenv("RT_AMD_SHADE_BATCH")) { int v = atoi(e); if (v >= 1 && v <= 64) ctx->shade_batch = v; }
    if (const char *e = getenv("RT_AMD_PIPE_SHARE")) { int v = atoi(e); if (v >= 0 && v <= 2) ctx->pipe_share = v; }
    if (const char *e = getenv("RT_AMD_MULTI_CAREFUL")) ctx->multi_careful = atoi(e) != 0;
    if (const char *e = getenv("RT_AMD_READY_BREAK")) { int v = atoi(e); if (v >= 1 && v <= 65) ctx->ready_break = v; }
    if (hipMalloc((void **)&ctx->tile_counter, 1024) != hipSuccess ||
        hipEventCreate(&ctx->ev_start) != hipSuccess || hipEventCreate(&ctx->ev_stop) != hipSuccess) {
        rt_ctx_destroy(ctx);
        return RT_ERR_HIP;
    }
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        g_live_ctxs.insert(ctx);
    }
    *out = ctx;
    return RT_OK;
}

extern "C" void rt_ctx_destroy(rt_ctx *ctx)
{
    if (!ctx) return;
    {
        /* landing areas other contexts keep for this one (it was a rank of their multi-GPU calls) */
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        g_live_ctxs.erase(ctx);
        for (rt_ctx *r : g_live_ctxs) {
            auto it = r->stages.find(ctx);
            if (it == r->stages.end()) continue;
            (void)hipSetDevice(r->device);
            if (it->second.ev_free) { (void)hipEventSynchronize(it->second.ev_free); (void)hipEventDestroy(it->second.ev_free); }
            if (it->second.d) (void)hipFree(it->second.d);
            r->stages.erase(it);
        }
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->tile_counter) (void)hipFree(ctx->tile_counter);
    if (ctx->d_prev) (void)hipFree(ctx->d_prev);
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    if (ctx->d_tile_order) (void)hipFree(ctx->d_tile_order);
    if (ctx->d_tile_cost) (void)hipFree(ctx->d_tile_cost);
    if (ctx->d_tile_peak) (void)hipFree(ctx->d_tile_peak);
    if (ctx->d_job_order) (void)hipFree(ctx->d_job_order);
    if (ctx->d_partial) (void)hipFree(ctx->d_partial);
    if (ctx->d_bands) (void)hipFree(ctx->d_bands);
    for (FrameSlot &fs : ctx->pipe.slots) {
        if (fs.stream) { (void)hipStreamSynchronize(fs.stream); (void)hipStreamDestroy(fs.stream); }
        if (fs.counter) (void)hipFree(fs.counter);
        if (fs.plane) (void)hipFree(fs.plane);
        if (fs.ev_done) (void)hipEventDestroy(fs.ev_done);
        if (fs.ev_free) (void)hipEventDestroy(fs.ev_free);
        if (fs.ev_call) (void)hipEventDestroy(fs.ev_call);
    }
    if (ctx->pipe.d_list) (void)hipFree(ctx->pipe.d_list);
    for (DevList &dl : ctx->dev_lists) free_dev_list(dl);
    for (auto &kv : ctx->stages) {
        if (kv.second.d) (void)hipFree(kv.second.d);
        if (kv.second.ev_free) (void)hipEventDestroy(kv.second.ev_free);
    }
    if (ctx->multi_stream) (void)hipStreamDestroy(ctx->multi_stream);
    if (ctx->ev_multi) (void)hipEventDestroy(ctx->ev_multi);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    delete ctx;
}

extern "C" const char *rt_last_error(const rt_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" rt_status rt_scene_commit(rt_ctx *ctx, const rt_scene_builder *b, rt_scene **out)
{
    if (!ctx || !b || !out) return RT_ERR_INVALID;
    *out = nullptr;
    rt_scene *s = new (std::nothrow) rt_scene();
    if (!s) return RT_ERR_NOMEM;
    s->ctx = ctx;
    static std::atomic<uint32_t> next_uid{1};
    s->uid = next_uid.fetch_add(1);
    std::string err = rt_flatten(*b, s->flat);
    if (!err.empty()) { delete s; return set_err(ctx, RT_ERR_UNSUPPORTED, err); }

    /* workgroup size: the largest one whose LDS (scene + per-lane traversal stacks) fits */
    const size_t blob_bytes = s->flat.blob.size() * sizeof(rt_f4);
    /* one spare stack entry: the traversal loop always writes the slot above the top */
    const size_t per_thread = s->flat.has_mesh ? (size_t)(s->flat.stack_entries + 1) * 8 : 0;
    s->threads = 0;
    (void)hipSetDevice(ctx->device);
    if (!s->flat.has_mesh) {
        /* 256-thread workgroups (six per CU) unless the object list is so long that only one or two copies of it
         * fit a CU's LDS: then the workgroup that keeps the most waves resident */
        int best_waves = 0;
        const int flat_candidates[4] = {256, 512, 768, 1024};
        for (int nt : flat_candidates) {
            if (blob_bytes > RT_LDS_LIMIT) break;
            int nb = rt_kernel_blocks_per_cu(0, 1, nt, blob_bytes);
            if (nb > RT_MAX_BLOCKS_PER_CU) nb = RT_MAX_BLOCKS_PER_CU;
            if (nb < 1) nb = 1;
            if (nb * (nt / 64) > best_waves) { best_waves = nb * (nt / 64); s->threads = nt; s->lds_bytes = blob_bytes; s->blocks_per_cu = nb; }
        }
    } else {
        /* the shape with the most resident waves per CU (registers, LDS: every workgroup stages its own copy of the
         * scene); the larger workgroup on a tie (fewer copies to stage) */
        int best_waves = 0;
        const int mesh_candidates[4] = {1024, 768, 512, 256};
        const char *force_nt = getenv("RT_AMD_THREADS");            /* development: force the workgroup size */
        if (force_nt) {
            const int v = atoi(force_nt);
            if (v != 256 && v != 512 && v != 768 && v != 1024) { delete s; return set_err(ctx, RT_ERR_INVALID, "RT_AMD_THREADS must be 256, 512, 768 or 1024"); }
        }
        for (int nt : mesh_candidates) {
            if (force_nt && atoi(force_nt) != nt) continue;
            const size_t lds = blob_bytes + per_thread * (size_t)nt;
            if (lds > RT_LDS_LIMIT) continue;
            int nb = rt_kernel_blocks_per_cu(1, 1, nt, lds);
            if (nb > RT_MAX_BLOCKS_PER_CU) nb = RT_MAX_BLOCKS_PER_CU;
            if (nb < 1) nb = 1;
            if (nb * (nt / 64) > best_waves) { best_waves = nb * (nt / 64); s->threads = nt; s->lds_bytes = lds; s->blocks_per_cu = nb; }
        }
    }
    s->scene_in_lds = RT_SCENE_LDS;
    const char *force_mode = getenv("RT_AMD_SCENE_MODE");           /* development: 0 forces the all-global kernel for scenes that do not fit LDS */
    if (s->threads == 0 && s->flat.has_mesh && !(force_mode && atoi(force_mode) == RT_SCENE_GLOBAL)) {
        /* The triangles do not fit, but everything before them in the blob may: BVH nodes (a depth-10 tree has at most
         * 1,023, whatever the triangle count), object records, object list.  Then only the triangles are read from global
         * memory (L2).  Worth it while at least half a CU's wave slots stay filled. */
        const size_t prefix_bytes = (size_t)s->flat.off_tris * sizeof(rt_f4);
        int best_waves = 0;
        const int hybrid_candidates[3] = {1024, 768, 512};
        for (int nt : hybrid_candidates) {
            const size_t lds = prefix_bytes + per_thread * (size_t)nt;
            if (lds > RT_LDS_LIMIT) continue;
            int nb = rt_kernel_blocks_per_cu(1, RT_SCENE_HYBRID, nt, lds);
            if (nb > RT_MAX_BLOCKS_PER_CU) nb = RT_MAX_BLOCKS_PER_CU;
            if (nb < 1) nb = 1;
            if (nb * (nt / 64) > best_waves) { best_waves = nb * (nt / 64); s->threads = nt; s->lds_bytes = lds; s->blocks_per_cu = nb; }
        }
        if (s->threads) s->scene_in_lds = RT_SCENE_HYBRID;
    }
    if (s->threads == 0) {
        /* larger than a CU's LDS: the kernel reads the scene from global memory (L2-resident),
         * LDS holds only the traversal stacks */
        s->scene_in_lds = RT_SCENE_GLOBAL;
        s->threads = s->flat.has_mesh ? 1024 : 256;
        /* (five or six waves per SIMD as 5-6 x 256 threads were measured on the 50,880- and 6,000-triangle scenes: 80.3 / 80.2 against
         * 80.6 Msamples/s and 57.4 / 57.5 against 57.5 - the path is bound by the L1's address processing, not by latency:
         * profiles/r04/experiments/big_mesh_global_5_waves.txt, pmc_vmem_sphere50k.txt) */
        s->lds_bytes = per_thread * (size_t)s->threads;
        if (s->lds_bytes > RT_LDS_LIMIT) {
            delete s;
            return set_err(ctx, RT_ERR_UNSUPPORTED, "BVH too deep for the per-lane LDS traversal stack");
        }
        /* (a 256-thread workgroup is admitted at most 6 times at this kernel's SGPR count, whatever the API says:
         * MI355X_MICROARCH.md, residency; surplus workgroups would only queue behind the resident ones) */
        int nb = rt_kernel_blocks_per_cu(s->flat.has_mesh ? 1 : 0, RT_SCENE_GLOBAL, s->threads, s->lds_bytes);
        s->blocks_per_cu = nb < 1 ? 1 : (nb > RT_MAX_BLOCKS_PER_CU ? RT_MAX_BLOCKS_PER_CU : nb);
    }
    if (const char *e = getenv("RT_AMD_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 8) s->blocks_per_cu = v; }
    /* the occupancy probes above discard their errors: a failed probe must not surface later as a launch error */
    (void)hipGetLastError();

    hipError_t e = hipMalloc((void **)&s->d_blob, blob_bytes > 0 ? blob_bytes : 16);
    if (e == hipSuccess && blob_bytes) e = hipMemcpy(s->d_blob, s->flat.blob.data(), blob_bytes, hipMemcpyHostToDevice);
    const size_t obj_bytes = s->flat.objects.size() * sizeof(rt_object);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_objects, obj_bytes > 0 ? obj_bytes : 16);
    if (e == hipSuccess && obj_bytes) e = hipMemcpy(s->d_objects, s->flat.objects.data(), obj_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && !s->flat.tri_uv.empty()) {
        e = hipMalloc((void **)&s->d_tri_uv, s->flat.tri_uv.size() * 4);
        if (e == hipSuccess) e = hipMemcpy(s->d_tri_uv, s->flat.tri_uv.data(), s->flat.tri_uv.size() * 4, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && !s->flat.tex_data.empty()) {
        e = hipMalloc((void **)&s->d_tex, s->flat.tex_data.size() * 4);
        if (e == hipSuccess) e = hipMemcpy(s->d_tex, s->flat.tex_data.data(), s->flat.tex_data.size() * 4, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        rt_scene_destroy(s);
        return hip_fail(ctx, e, "uploading scene");
    }
    *out = s;
    return RT_OK;
}

extern "C" void rt_scene_destroy(rt_scene *s)
{
    if (!s) return;
    if (s->ctx) (void)hipSetDevice(s->ctx->device);
    if (s->d_blob) (void)hipFree(s->d_blob);
    if (s->d_objects) (void)hipFree(s->d_objects);
    if (s->d_tri_uv) (void)hipFree(s->d_tri_uv);
    if (s->d_tex) (void)hipFree(s->d_tex);
    delete s;
}

extern "C" rt_status rt_scene_get_info(const rt_scene *s, rt_scene_info *out)
{
    if (!s || !out) return RT_ERR_INVALID;
    out->num_objects = (int32_t)s->flat.objects.size();
    out->num_triangles = s->flat.num_tris;
    out->num_nodes = s->flat.num_nodes;
    out->lds_bytes = (int32_t)s->lds_bytes;
    out->scene_in_lds = s->scene_in_lds;
    out->threads_per_block = s->threads;
    out->stack_entries = s->flat.stack_entries;
    out->blocks_per_cu = s->blocks_per_cu;
    return RT_OK;
}

extern "C" int32_t rt_tile_owned_rows(const rt_tile_spec *t, int32_t height)
{
    if (!t || t->band_rows <= 0 || t->band_stride <= 0 || t->band_first < 0 || t->band_first >= t->band_stride) return -1;
    int32_t bands_total = (height + t->band_rows - 1) / t->band_rows;
    int32_t owned = bands_total > t->band_first ? (bands_total - t->band_first + t->band_stride - 1) / t->band_stride : 0;
    return owned * t->band_rows;
}

extern "C" int32_t rt_max_batch_frames(rt_ctx *ctx, int32_t width, int32_t height)
{
    if (!ctx || width <= 0 || height <= 0) return 1;
    return batch_cap(ctx, (size_t)width * (size_t)height * 3);
}

/* Longest-processing-time-first ownership (see include/rt_amd.h) */
extern "C" rt_status rt_partition_tiles(const uint32_t *cost, int32_t tiles_x, int32_t tiles_y, int32_t n_ranks, int32_t *owner)
{
    if (!owner || tiles_x <= 0 || tiles_y <= 0 || n_ranks <= 0 || (int64_t)tiles_x * tiles_y > (int64_t)RT_JOB_TILE_MASK + 1) return RT_ERR_INVALID;
    const int n = tiles_x * tiles_y;
    if (!cost) {
        for (int ty = 0; ty < tiles_y; ty++)
            for (int tx = 0; tx < tiles_x; tx++) owner[ty * tiles_x + tx] = (tx + ty) % n_ranks;
        return RT_OK;
    }
    std::vector<uint32_t> idx((size_t)n);
    for (int i = 0; i < n; i++) idx[(size_t)i] = (uint32_t)i;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return cost[x] > cost[y]; });
    /* (load, rank): the least loaded rank first, the lower rank on a tie */
    typedef std::pair<uint64_t, int> LR;
    std::priority_queue<LR, std::vector<LR>, std::greater<LR>> heap;
    for (int r = 0; r < n_ranks; r++) heap.push(LR(0, r));
    for (uint32_t t : idx) {
        LR top = heap.top();
        heap.pop();
        owner[t] = top.second;
        top.first += cost[t];
        heap.push(top);
    }
    return RT_OK;
}

/* persistent waves: enough workgroups to fill the chip, each wave pulls 8x8 tiles */
static int launch_blocks(const rt_ctx *ctx, const rt_scene *scene, int num_tiles)
{
    const int waves_per_block = scene->threads / 64;
    int blocks = ctx->num_cus * scene->blocks_per_cu;
    const int needed = (num_tiles + waves_per_block - 1) / waves_per_block;
    return blocks > needed ? needed : blocks;
}

/* the schedule of a multi-frame launch (see its use in render_frames): `order` is the launch's tile order
 * (any permutation of 0..n-1; ties in cost keep it), cost[t] the measured cost of tile t - bit 0 set if a ray of the
 * tile entered a mesh: those are the long jobs; the others (sky, ground) follow frame by frame */
static void build_job_order(const std::vector<uint32_t> &order, const std::vector<uint32_t> &cost, const std::vector<uint32_t> &peak, uint32_t top_max,
                            uint32_t frames, std::vector<uint32_t> &jobs)
{
    const uint32_t n = (uint32_t)order.size();
    std::vector<uint32_t> idx;
    idx.reserve(n);
    for (uint32_t t : order) if (cost[t] & 1u) idx.push_back(t);
    /* a job lasts as long as its longest pixel: by decreasing peak pixel cost (measured at N = 8, 1024 spp: ordering by
     * the tile's SUM left a rank in four with a long pixel started late - ranks 611-718 ms; by peak 612-631) */
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return peak[x] > peak[y]; });
    const uint32_t top = top_max < (uint32_t)idx.size() ? top_max : (uint32_t)idx.size();
    jobs.clear();
    jobs.reserve((size_t)n * frames);
    /* by decreasing cost, the frames of a tile together */
    for (uint32_t r = 0; r < top; r++)
        for (uint32_t f = 0; f < frames; f++) jobs.push_back(idx[r] | (f << RT_JOB_FRAME_SHIFT));
    /* everything else frame by frame, in the launch's tile order */
    std::vector<char> taken(n, 0);
    for (uint32_t r = 0; r < top; r++) taken[idx[r]] = 1;
    for (uint32_t f = 0; f < frames; f++)
        for (uint32_t t : order) if (!taken[t]) jobs.push_back(t | (f << RT_JOB_FRAME_SHIFT));
}

namespace {

uint32_t gcd_u32(uint32_t x, uint32_t y) { while (y) { uint32_t t = x % y; x = y; y = t; } return x; }

/* a stride near n / golden ratio, made coprime to n */
uint32_t coprime_stride(uint32_t n)
{
    if (n <= 2u) return 1u;
    uint32_t st = (uint32_t)(n * 0.6180339887) | 1u;
    while (gcd_u32(st, n) != 1u) st += 2u;
    return st % n ? st % n : 1u;
}

/* The view's measured costs are on the device (cost_state 1): bring them to the host on `stream` (one
 * synchronisation), and, when the view uses a tile order, move the `heavy_top` most expensive tiles to its front. */
/* the render kernels of every pipelined frame have finished (host wait); with_folds: so have the blends that read their planes
 * and the view's tile list */
rt_status drain_pipeline(rt_ctx *ctx, bool with_folds)
{
    for (FrameSlot &fs : ctx->pipe.slots) {
        if (fs.used) RT_HIP(ctx, hipEventSynchronize(fs.ev_done), "waiting for the frames in flight");
        if (with_folds && fs.folded) RT_HIP(ctx, hipEventSynchronize(fs.ev_free), "waiting for the frames in flight");
    }
    return RT_OK;
}

rt_status read_costs_and_refine(rt_ctx *ctx, hipStream_t stream)
{
    const uint32_t n = (uint32_t)ctx->tiles_host.size();
    if (ctx->cost_state == 1) {
        /* on the launch stream: the launch that collected the costs ran on it (or this stream has been ordered
         * behind it), and a blocking copy on the null stream would not wait for a non-blocking stream's kernel */
        std::vector<uint32_t> cost(n), peak(n);
        RT_HIP(ctx, hipMemcpyAsync(cost.data(), ctx->d_tile_cost, (size_t)n * 4, hipMemcpyDeviceToHost, stream), "reading tile costs");
        RT_HIP(ctx, hipMemcpyAsync(peak.data(), ctx->d_tile_peak, (size_t)n * 4, hipMemcpyDeviceToHost, stream), "reading tile costs");
        RT_HIP(ctx, hipStreamSynchronize(stream), "reading tile costs");
        ctx->cost_host.swap(cost);
        ctx->peak_host.swap(peak);
    }
    if (ctx->use_order && ctx->heavy_top > 0) {
        const std::vector<uint32_t> &cost = ctx->cost_host, &peak = ctx->peak_host;
        std::vector<uint32_t> idx;
        idx.reserve(n);
        for (uint32_t t : ctx->order_host) if (cost[t] & 1u) idx.push_back(t);          /* tiles with a ray in a mesh */
        const std::vector<uint32_t> &key = ctx->top_by_peak ? peak : cost;
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return key[x] > key[y]; });
        const uint32_t top = (uint32_t)ctx->heavy_top < (uint32_t)idx.size() ? (uint32_t)ctx->heavy_top : (uint32_t)idx.size();
        std::vector<char> taken(n, 0);
        std::vector<uint32_t> merged;
        merged.reserve(n);
        const uint32_t st = ctx->tile_scatter ? coprime_stride(top) : 1u;
        for (uint32_t i = 0; i < top; i++) { const uint32_t t = idx[(size_t)(((uint64_t)i * st) % top)]; merged.push_back(t); taken[t] = 1; }
        for (uint32_t t : ctx->order_host) if (!taken[t]) merged.push_back(t);
        /* nothing may still be indexing the old order: the stream was synchronised above, or (caller-supplied costs)
         * the order has not been used by a launch yet */
        RT_HIP(ctx, hipMemcpyAsync(ctx->d_tile_order, merged.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream), "uploading tile order");
        RT_HIP(ctx, hipStreamSynchronize(stream), "uploading tile order");     /* `merged` is a local */
        ctx->order_host.swap(merged);
        ctx->order_num_heavy = (int)top;
    }
    ctx->cost_state = 2;
    ctx->job_frames = 0;                    /* any uploaded schedule is for another view */
    return RT_OK;
}

}  // namespace

/* one launch rendering n_frames consecutive progressive frames (n_frames == 1: a plain frame with an
 * optional separate previous frame; > 1: d_out is updated in place) */
/* (c + prev * n) / (n + 1) over the pixels a launch owns, frame by frame (src/raytracer.cu:109-112): `planes` holds n_frames planes of
 * per-pixel means in the launch's output layout.  Compact layouts, and a full frame rendered whole: one pass over the buffer; a
 * full-layout frame of which the launch owns some bands / tiles: only those are folded */
static rt_status fold_planes(rt_ctx *ctx, const float *planes, size_t plane_floats, int n_frames, int frame_num, float *d_out, int compact, bool listed,
                             const uint32_t *d_tile_list, int n_tiles, int tiles_x, int band_rows, int band_first, int band_stride, int width, int height,
                             hipStream_t stream)
{
    if (compact || (!listed && band_stride == 1)) {
        RT_HIP(ctx, rt_launch_blend(planes, (long long)plane_floats, n_frames, frame_num, d_out, (long long)plane_floats, stream), "launching blend kernel");
    } else if (listed) {
        RT_HIP(ctx, rt_launch_blend_tiles(planes, (long long)plane_floats, n_frames, frame_num, d_out, d_tile_list, n_tiles, tiles_x, width, height, stream),
               "launching blend kernel");
    } else {
        const int bands_total = (height + band_rows - 1) / band_rows;
        for (int b = band_first; b < bands_total; b += band_stride) {
            const int row0 = b * band_rows, row1 = row0 + band_rows < height ? row0 + band_rows : height;
            const long long off = (long long)row0 * width * 3, cnt = (long long)(row1 - row0) * width * 3;
            RT_HIP(ctx, rt_launch_blend(planes + off, (long long)plane_floats, n_frames, frame_num, d_out + off, cnt, stream), "launching blend kernel");
        }
    }
    return RT_OK;
}

/* slot != NULL: a pipelined frame (rt_frame_submit) - the launch runs on the slot's stream with the slot's ticket counter, its
 * pixels' means go to the slot's plane and nothing is folded here; d_prev, d_out and hip_stream are not used. */
static rt_status render_frames(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_settings *rs,
                               const int32_t *times_ms, int32_t n_frames, int32_t frame_num, const rt_tile_spec *tiles,
                               const float *d_prev, float *d_out, void *hip_stream, bool in_place, FrameSlot *slot = nullptr)
{
    if (!ctx || !scene || !cam || !rs || (!d_out && !slot)) return set_err(ctx, RT_ERR_INVALID, "null argument");
    if (scene->ctx != ctx) return set_err(ctx, RT_ERR_INVALID, "scene belongs to another context");
    if (cam->width <= 0 || cam->height <= 0 || cam->width > 32768 || cam->height > 32768 || (int64_t)cam->width * cam->height > (1 << 28))
        return set_err(ctx, RT_ERR_INVALID, "bad image size (at most 32768 pixels on a side and 2^28 in all)");
    if (rs->rays_per_pixel < 0 || rs->reflection_limit < 0) return set_err(ctx, RT_ERR_INVALID, "bad render settings");
    rt_tile_spec full;
    std::memset(&full, 0, sizeof full);
    full.band_rows = 8; full.band_stride = 1;
    const rt_tile_spec *t = tiles ? tiles : &full;
    const bool listed = t->tile_list != nullptr;
    const int tiles_x = (cam->width + 7) / 8, tiles_y = (cam->height + 7) / 8;
    if (listed) {
        if (t->num_tiles < 0 || t->num_tiles > tiles_x * tiles_y) return set_err(ctx, RT_ERR_INVALID, "bad tile spec (num_tiles)");
    } else {
        if (t->tile_cost || t->tile_peak) return set_err(ctx, RT_ERR_INVALID, "bad tile spec (tile_cost / tile_peak need a tile_list)");
        if (t->band_rows <= 0 || (t->band_rows & 7) || t->band_stride <= 0 || t->band_first < 0 || t->band_first >= t->band_stride)
            return set_err(ctx, RT_ERR_INVALID, "bad tile spec (band_rows must be a positive multiple of 8, 0 <= band_first < band_stride)");
    }
    hipStream_t stream = slot ? slot->stream : (hipStream_t)hip_stream;
    RT_HIP(ctx, hipSetDevice(ctx->device), "selecting device");
    /* one launch in flight per context: the scratch is shared (see rt_ctx) ... */
    if (ctx->launched && (slot || ctx->last_stream != stream)) RT_HIP(ctx, hipStreamWaitEvent(stream, ctx->ev_stop, 0), "ordering the launch behind the previous one");
    /* ... except pipelined frames, which bring their own (FrameSlot) and only read the view's: they overlap each other, and an
     * ordinary launch is queued behind all of them */
    if (!slot)
        for (FrameSlot &fs : ctx->pipe.slots)
            if (fs.used) RT_HIP(ctx, hipStreamWaitEvent(stream, fs.ev_done, 0), "ordering the launch behind the frames in flight");

    rt_kernel_args a;
    std::memset(&a, 0, sizeof a);
    std::memcpy(a.cam + 0, cam->cam_pos, 12);
    std::memcpy(a.cam + 3, cam->tl_pixel_pos, 12);
    std::memcpy(a.cam + 6, cam->delta_u, 12);
    std::memcpy(a.cam + 9, cam->delta_v, 12);
    a.width = cam->width;
    a.height = cam->height;
    a.rays_per_pixel = rs->rays_per_pixel;
    a.reflection_limit = rs->reflection_limit;
    a.antialias = rs->antialias ? 1 : 0;
    std::memcpy(a.sky, rs->sky_colour, 12);
    for (int i = 0; i < n_frames; i++) a.seeds[i] = (uint32_t)times_ms[i] * 6291469u;       /* src/raytracer.cu:127 */
    a.num_frames = n_frames;
    a.frame_num = frame_num;
    a.band_rows = listed ? 8 : t->band_rows;
    a.band_first = listed ? 0 : t->band_first;
    a.band_stride = listed ? 1 : t->band_stride;
    a.compact = t->compact ? 1 : 0;
    a.tiles_x = tiles_x;
    const int owned_rows = listed ? 0 : rt_tile_owned_rows(t, cam->height);
    a.num_tiles = listed ? t->num_tiles : (owned_rows / 8) * a.tiles_x;
    const uint32_t n = a.num_tiles > 0 ? (uint32_t)a.num_tiles : 0u;
    a.tile_stride = ctx->tile_scatter ? coprime_stride(n ? n : 1u) : 1u;

    a.objects = scene->d_objects;
    a.num_objects = (int32_t)scene->flat.objects.size();
    a.blob = scene->d_blob;
    a.blob_f4 = (int32_t)scene->flat.blob.size();
    a.off_nodes = scene->flat.off_nodes;
    a.off_tris = scene->flat.off_tris;
    a.off_objlds = scene->flat.off_objlds;
    a.off_meshes = scene->flat.off_meshes;
    a.off_objtab = scene->flat.off_objtab;
    a.num_meshes = scene->flat.num_meshes;
    a.stack_entries = scene->flat.stack_entries;
    a.work_threshold = ctx->work_threshold;
    a.ready_break = ctx->ready_break;
    a.hit_break = ctx->hit_break;
    const int mix_break = ctx->mix_break >= 0 ? ctx->mix_break : (scene->threads == 1024 ? RT_DEF_MIX_BREAK_1024 : RT_DEF_MIX_BREAK);
    a.hit_low = ctx->hit_low > 0 && mix_break > 0 ? ctx->hit_low : ctx->hit_break;
    a.mix_break = ctx->hit_low > 0 && mix_break > 0 ? mix_break : 1000;
    a.shade_batch = ctx->shade_batch;
    a.descend_keep = ctx->descend_keep;
    a.tri_uv = scene->d_tri_uv;
    a.tex_data = scene->d_tex;
    a.prev = d_prev;
    a.out = d_out;
    a.tile_counter = slot ? slot->counter : ctx->tile_counter;
    a.stats = (unsigned long long *)(a.tile_counter + 16);   /* 48 x u64 after the counter; used by -DRT_STATS builds only */


    /* one plane of the launch's output layout (floats) */
    const size_t plane_floats = a.compact ? (listed ? (size_t)n * 192 : (size_t)owned_rows * (size_t)cam->width * 3)
                                          : (size_t)cam->height * (size_t)cam->width * 3;

    /* ---- the view: which tiles, in which order, at what cost ---------------------------------------- */
    bool collecting = false;
    if (n > 0) {
        std::vector<uint32_t> key;
        key.push_back(scene->uid); key.push_back((uint32_t)ctx->tile_scatter);
        for (int i = 0; i < 12; i++) { uint32_t u; std::memcpy(&u, &a.cam[i], 4); key.push_back(u); }
        key.push_back((uint32_t)a.width); key.push_back((uint32_t)a.height);
        key.push_back((uint32_t)a.reflection_limit);      /* (a launch at bounce limit 0 or 1 says nothing about one at 8) */
        if (listed) {
            uint64_t h = fnv1a(t->tile_list, (size_t)n * 4);
            if (t->tile_cost) h = fnv1a(t->tile_cost, (size_t)n * 4, h);
            if (t->tile_cost && t->tile_peak) h = fnv1a(t->tile_peak, (size_t)n * 4, h);
            key.push_back(0xffffffffu); key.push_back(n); key.push_back((uint32_t)h); key.push_back((uint32_t)(h >> 32)); key.push_back(t->tile_cost ? 1u : 0u);
        } else {
            key.push_back((uint32_t)a.band_rows); key.push_back((uint32_t)a.band_first); key.push_back((uint32_t)a.band_stride);
        }
        /* a pixel's cost counter is 26 bits wide (rt_device_scene.h RT_COST_*): launches that could overflow it (65,536 or
         * more bounces per pixel) do not collect costs and run on the schedule the view already has */
        const bool cost_fits = (long long)rs->rays_per_pixel * (long long)(rs->reflection_limit > 0 ? rs->reflection_limit : 1) < 65536ll;
        /* figures from a launch with far fewer samples (a 1-spp preview, a profiler's warm-up launch) are provisional
         * like a pilot's: this launch runs on them and measures again */
        const bool new_view = key != ctx->order_key;
        const bool provisional = !new_view && ctx->cost_state == 2 && ctx->cost_spp > 0 && (long long)ctx->cost_spp * 8 <= (long long)rs->rays_per_pixel &&
                                 !(listed && t->tile_cost) && cost_fits;
        if (slot) {
            /* Pipelined frames overlap as long as they only READ the view (tile order, tile list).  A launch that writes it - a
             * new view, a pilot, a launch that measures tile costs, the one that sorts by them - first waits (on the host)
             * for the frames in flight: once or twice per view. */
            const bool will_pilot = ctx->pilot && ctx->use_order && rs->rays_per_pixel >= ctx->pilot_min_spp && rs->reflection_limit > 0;
            const bool list_changed = listed && (ctx->pipe.list_host.size() != (size_t)n || std::memcmp(ctx->pipe.list_host.data(), t->tile_list, (size_t)n * 4) != 0);
            if (list_changed)
                for (int j = 0; j < ctx->pipe.pending; j++)
                    if (ctx->pipe.slots[ctx->pipe.order[j]].listed)
                        return set_err(ctx, RT_ERR_BUSY, "frames of another tile list are waiting to be collected (rt_frame_collect; d_frame == NULL discards one)");
            if (new_view || list_changed || (ctx->cost_state == 0 && (cost_fits || will_pilot)) || (ctx->cost_state == 1 && ctx->use_order) || provisional) {
                rt_status st = drain_pipeline(ctx, new_view || list_changed);
                if (st != RT_OK) return st;
            }
            if (listed && (list_changed || !ctx->pipe.d_list)) {
                rt_status st = grow_u32(ctx, &ctx->pipe.d_list, &ctx->pipe.list_cap, n, "allocating the pipeline's tile list");
                if (st != RT_OK) return st;
                ctx->pipe.list_host.clear();
                RT_HIP(ctx, hipMemcpy(ctx->pipe.d_list, t->tile_list, (size_t)n * 4, hipMemcpyHostToDevice), "uploading the pipeline's tile list");
                ctx->pipe.list_host.assign(t->tile_list, t->tile_list + n);
            }
        }
        if (new_view) {
            /* a new view.  local tile -> tile of the image */
            std::vector<uint32_t> th(n);
            if (listed) {
                std::vector<char> seen((size_t)tiles_x * tiles_y, 0);
                for (uint32_t i = 0; i < n; i++) {
                    const uint32_t g = t->tile_list[i];
                    if (g >= (uint32_t)(tiles_x * tiles_y) || seen[g]) return set_err(ctx, RT_ERR_INVALID, "bad tile spec (a tile index outside the image, or listed twice)");
                    seen[g] = 1;
                    th[i] = g;
                }
            } else {
                const int tiles_per_band = a.tiles_x * (a.band_rows >> 3);
                for (uint32_t i = 0; i < n; i++) {
                    const int band_local = (int)i / tiles_per_band, in_band = (int)i % tiles_per_band;
                    const int band = a.band_first + band_local * a.band_stride;
                    th[i] = (uint32_t)((band * (a.band_rows >> 3) + in_band / a.tiles_x) * a.tiles_x + in_band % a.tiles_x);
                }
            }
            ctx->order_key.clear();              /* (until everything below has succeeded) */
            ctx->tiles_host.swap(th);
            ctx->cost_state = 0;
            ctx->cost_host.clear();
            ctx->peak_host.clear();
            ctx->order_num_heavy = 0;
            ctx->job_frames = 0;
            rt_status st;
            if ((st = grow_u32(ctx, &ctx->d_tile_cost, &ctx->tile_cost_cap, n, "allocating tile costs")) != RT_OK) return st;
            if ((st = grow_u32(ctx, &ctx->d_tile_peak, &ctx->tile_peak_cap, n, "allocating tile costs")) != RT_OK) return st;
            ctx->use_order = ctx->heavy_first && scene->flat.num_meshes > 0 && n > 1;
            if (ctx->use_order) {
                /* longest-job-first, first guess: tiles whose centre ray enters a mesh root box are handed out
                 * first.  Host float math, a heuristic only: any order renders the same image. */
                std::vector<uint32_t> heavy, light;
                for (uint32_t i = 0; i < n; i++) {
                    const int ty = (int)(ctx->tiles_host[i] / (uint32_t)a.tiles_x), tx = (int)(ctx->tiles_host[i] % (uint32_t)a.tiles_x);
                    const float px = tx * 8 + 4.0f, py = ty * 8 + 4.0f;
                    float d[3], o[3];
                    for (int k = 0; k < 3; k++) { o[k] = a.cam[k]; d[k] = a.cam[3 + k] + a.cam[6 + k] * px + a.cam[9 + k] * py - o[k]; }
                    bool hit = false;
                    for (size_t m = 0; m < scene->flat.objects.size() && !hit; m++) {
                        const rt_object &ob = scene->flat.objects[m];
                        if (ob.type != RT_OBJ_MESH) continue;
                        float tmin = 0.0f, tmax = 3.0e38f;
                        for (int k = 0; k < 3; k++) {
                            /* grow the box by a margin: the tile is 8 pixels wide and paths leave it */
                            const float ext = 0.15f * (ob.v[3 + k] - ob.v[k]) + 1e-3f;
                            const float inv = 1.0f / d[k];
                            float t1 = (ob.v[k] - ext - o[k]) * inv, t2 = (ob.v[3 + k] + ext - o[k]) * inv;
                            if (t1 > t2) { float s = t1; t1 = t2; t2 = s; }
                            if (t1 > tmin) tmin = t1;
                            if (t2 < tmax) tmax = t2;
                        }
                        hit = tmin <= tmax;
                    }
                    (hit ? heavy : light).push_back(i);
                }
                std::vector<uint32_t> order;
                order.reserve(n);
                for (const std::vector<uint32_t> *cls : {&heavy, &light}) {
                    const uint32_t m = (uint32_t)cls->size();
                    const uint32_t st2 = ctx->tile_scatter ? coprime_stride(m) : 1u;
                    for (uint32_t i = 0; i < m; i++) order.push_back((*cls)[(size_t)(((uint64_t)i * st2) % m)]);
                }
                if ((st = grow_u32(ctx, &ctx->d_tile_order, &ctx->tile_order_cap, n, "allocating tile order")) != RT_OK) return st;
                RT_HIP(ctx, hipMemcpyAsync(ctx->d_tile_order, order.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream), "uploading tile order");
                RT_HIP(ctx, hipStreamSynchronize(stream), "uploading tile order");     /* `order` is a local */
                ctx->order_host.swap(order);
            }
            if (listed && t->tile_cost) {
                /* the caller knows what the tiles cost (an earlier launch of the view, possibly on other GPUs) */
                ctx->cost_host.assign(t->tile_cost, t->tile_cost + n);
                if (t->tile_peak) ctx->peak_host.assign(t->tile_peak, t->tile_peak + n);
                else ctx->peak_host = ctx->cost_host;
                if ((st = read_costs_and_refine(ctx, stream)) != RT_OK) return st;
            }
            ctx->order_key = key;
        }
        if (listed && slot) {
            a.tile_list = ctx->pipe.d_list;
        } else if (listed) {
            rt_status st = device_tile_list(ctx, t->tile_list, (int)n, stream, &a.tile_list);
            if (st != RT_OK) return st;
        }
        /* The first launch of a view adds up, per tile, the work of its pixels; the second reads the sums back (one
         * synchronisation) and moves the `heavy_top` most expensive tiles to the front of the order, most expensive
         * first within rounds of one ticket per wave.  In a multi-frame launch those tiles go first for ALL frames
         * (see px_fetch): with the true costs that is worth 10 % at three frames per launch (with the guess, nothing). */
        if (ctx->cost_state == 0) {
            /* Pilot: a view's first launch would run on the guessed order (measured: 315 instead of 228 ms per frame for
             * the monkey's first five frames).  One sample per pixel of the launch's first frame, rendered into a scratch
             * plane, measures the tiles well enough to schedule by (~1/spp of a frame + one synchronisation); the launch
             * itself then measures them properly for the launches after it.  Nothing of the pilot reaches the image. */
            if (ctx->pilot && ctx->use_order && rs->rays_per_pixel >= ctx->pilot_min_spp && rs->reflection_limit > 0) {
                rt_status st = grow(ctx, &ctx->d_partial, &ctx->partial_cap, plane_floats * (size_t)(in_place ? n_frames : 1), "allocating the pilot's scratch plane");
                if (st != RT_OK) return st;
                rt_kernel_args ap = a;
                ap.rays_per_pixel = 1;
                ap.num_frames = 1;
                ap.partial = ctx->d_partial;
                ap.partial_plane = (int64_t)(plane_floats / 3);
                ap.prev = nullptr;
                ap.tile_order = ctx->d_tile_order;
                ap.tile_cost = ctx->d_tile_cost;
                ap.tile_peak = ctx->d_tile_peak;
                RT_HIP(ctx, hipMemsetAsync(ctx->d_tile_cost, 0, (size_t)n * 4, stream), "clearing tile costs");
                RT_HIP(ctx, hipMemsetAsync(ctx->d_tile_peak, 0, (size_t)n * 4, stream), "clearing tile costs");
                RT_HIP(ctx, hipMemsetAsync(ap.tile_counter, 0, 512, stream), "clearing tile counter");
                RT_HIP(ctx, rt_launch_render(&ap, scene->flat.has_mesh ? 1 : 0, scene->scene_in_lds, scene->threads, launch_blocks(ctx, scene, (int)n), scene->lds_bytes, stream),
                       "launching the pilot");
                ctx->cost_state = 1; ctx->cost_spp = 1;
                if ((st = read_costs_and_refine(ctx, stream)) != RT_OK) return st;      /* -> 2: a provisional schedule */
            }
            if (cost_fits) {
                RT_HIP(ctx, hipMemsetAsync(ctx->d_tile_cost, 0, (size_t)n * 4, stream), "clearing tile costs");
                RT_HIP(ctx, hipMemsetAsync(ctx->d_tile_peak, 0, (size_t)n * 4, stream), "clearing tile costs");
                a.tile_cost = ctx->d_tile_cost;
                a.tile_peak = ctx->d_tile_peak;
                collecting = true;
            }
        } else {
            if (ctx->cost_state == 1 && ctx->use_order) {
                rt_status st = read_costs_and_refine(ctx, stream);
                if (st != RT_OK) return st;
            }
            if (provisional && ctx->cost_state == 2) {
                RT_HIP(ctx, hipMemsetAsync(ctx->d_tile_cost, 0, (size_t)n * 4, stream), "clearing tile costs");
                RT_HIP(ctx, hipMemsetAsync(ctx->d_tile_peak, 0, (size_t)n * 4, stream), "clearing tile costs");
                a.tile_cost = ctx->d_tile_cost;
                a.tile_peak = ctx->d_tile_peak;
                collecting = true;
            }
        }
        if (ctx->use_order) {
            a.num_heavy_tiles = (ctx->cost_state == 2 && n_frames > 1) ? ctx->order_num_heavy : 0;
            a.tile_order = ctx->d_tile_order;
            /* Multi-frame launch of a view whose tile costs are known: the host lays out the whole schedule.
             * A job is one tile of one frame; a pixel's samples are one sequential random stream, so a job
             * cannot be split and the launch is at least as long as its longest job - which therefore has to
             * start at once, for EVERY frame (frames only meet in the blend behind the kernel).  Longest
             * processing time first: jobs in order of decreasing measured tile cost, all frames of a tile
             * together; tiles that cost nothing follow frame by frame.  Consecutive tickets are also similar
             * in cost, so the lanes a wave refills as its cheap pixels finish collect pixels of one kind.  Any
             * schedule renders the same image.  (Measured, 20 frames of the monkey configuration: +6 % on one
             * GPU and +25 % on the share one of 8 GPUs renders, against "the 1,024 most expensive tiles of frame
             * 0, of frame 1, ... first".) */
            if (ctx->lpt && ctx->cost_state == 2 && n_frames > 1 && n <= (RT_JOB_TILE_MASK + 1u) && ctx->cost_host.size() == n && ctx->peak_host.size() == n) {
                if (ctx->job_frames != n_frames) {
                    std::vector<uint32_t> jobs;
                    build_job_order(ctx->order_host, ctx->cost_host, ctx->lpt_by_peak ? ctx->peak_host : ctx->cost_host, (uint32_t)ctx->lpt_top, (uint32_t)n_frames, jobs);
                    rt_status st = grow_u32(ctx, &ctx->d_job_order, &ctx->job_cap, jobs.size(), "allocating the launch schedule");
                    if (st != RT_OK) return st;
                    RT_HIP(ctx, hipMemcpyAsync(ctx->d_job_order, jobs.data(), jobs.size() * 4, hipMemcpyHostToDevice, stream), "uploading the launch schedule");
                    RT_HIP(ctx, hipStreamSynchronize(stream), "uploading the launch schedule");     /* `jobs` is a local */
                    ctx->job_frames = n_frames;
                }
                a.job_order = ctx->d_job_order;
            }
        }
    }
    if (slot) {
        if (slot->plane_cap < plane_floats && slot->folded) RT_HIP(ctx, hipEventSynchronize(slot->ev_free), "waiting for a frame's blend");   /* (the plane is about to be replaced) */
        rt_status st = grow(ctx, &slot->plane, &slot->plane_cap, plane_floats, "allocating a pipelined frame's plane");
        if (st != RT_OK) return st;
        a.partial = slot->plane;
        a.partial_plane = (int64_t)(plane_floats / 3);
        slot->plane_floats = plane_floats; slot->compact = a.compact; slot->listed = listed ? 1 : 0;
        slot->band_rows = a.band_rows; slot->band_first = a.band_first; slot->band_stride = a.band_stride;
        slot->width = cam->width; slot->height = cam->height; slot->tiles_x = tiles_x; slot->n_tiles = a.num_tiles;
    } else if (in_place) {
        const size_t need = plane_floats * (size_t)n_frames;
        rt_status st = grow(ctx, &ctx->d_partial, &ctx->partial_cap, need, "allocating the per-frame planes of a multi-frame launch");
        if (st != RT_OK) return st;
        a.partial = ctx->d_partial;
        a.partial_plane = (int64_t)(plane_floats / 3);
    }
    if (!slot) {
        RT_HIP(ctx, hipEventRecord(ctx->ev_start, stream), "recording start event");
        ctx->have_timing = false;
    }
    if (a.num_tiles > 0) {
        int blocks = launch_blocks(ctx, scene, a.num_tiles);
        if (slot) {
            /* A pipelined frame gets its share of the CUs, not all of them: a frame alone on the GPU is as long as its longest
             * pixel and leaves most CUs idle (or held by a workgroup with one busy wave) for half of that time; on 1 / depth of
             * the CUs the same frame is bound by its work instead, and `depth` of them fill the GPU (measured, monkey 1080p
             * 1024 spp, 4 in flight: 286 ms per frame with full-size launches, 242 with quarter-size ones).
             * Only where a workgroup has its CU to itself (a mesh that fills the LDS): smaller workgroups of several launches share
             * CUs anyway, and full-size launches are then the faster ones (three-sphere 61.7 against 65.1 ms, cube 112.8 against 118.0). */
            const int d = ctx->pipe.depth;
            if (ctx->pipe_share == 1 || (ctx->pipe_share == 2 && scene->blocks_per_cu == 1)) blocks = (blocks + d - 1) / d;
        }
        RT_HIP(ctx, hipMemsetAsync(a.tile_counter, 0, 512, stream), "clearing tile counter");
        RT_HIP(ctx, rt_launch_render(&a, scene->flat.has_mesh ? 1 : 0, scene->scene_in_lds, scene->threads, blocks, scene->lds_bytes, stream), "launching render kernel");
        if (collecting) { ctx->cost_state = 1; ctx->cost_spp = rs->rays_per_pixel; }   /* only now: a failed launch leaves no costs to sort on */
    }
    if (slot) {
        RT_HIP(ctx, hipEventRecord(slot->ev_done, stream), "recording a pipelined frame's end");
        slot->used = true;
        return RT_OK;
    }
    if (in_place && a.num_tiles > 0) {
        rt_status st = fold_planes(ctx, ctx->d_partial, plane_floats, n_frames, frame_num, d_out, a.compact, listed, a.tile_list, (int)n, tiles_x,
                                   a.band_rows, a.band_first, a.band_stride, cam->width, cam->height, stream);
        if (st != RT_OK) return st;
    }
    RT_HIP(ctx, hipEventRecord(ctx->ev_stop, stream), "recording stop event");
    ctx->have_timing = true;
    ctx->launched = true;
    ctx->last_stream = stream;
    return RT_OK;
}

extern "C" rt_status rt_render_device(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_settings *rs,
                                      int32_t time_ms, int32_t frame_num, const rt_tile_spec *tiles,
                                      const float *d_prev, float *d_out, void *hip_stream)
{
    return render_frames(ctx, scene, cam, rs, &time_ms, 1, frame_num, tiles, d_prev, d_out, hip_stream, false);
}

/* n_frames consecutive progressive frames (frame_num, frame_num + 1, ...; seeds times_ms[i]) in ONE
 * launch, accumulated in place in d_frame (layout of rt_render_device's d_out; its content is the
 * previous frame when frame_num > 0, ignored otherwise).  The result is bit-identical to n_frames
 * calls of rt_render_device; the point is that frame k + 1 is traced while the expensive pixels of
 * frame k are still running, which a launch per frame cannot do. */
extern "C" rt_status rt_render_device_batch(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_settings *rs,
                                            const int32_t *times_ms, int32_t n_frames, int32_t frame_num, const rt_tile_spec *tiles,
                                            float *d_frame, void *hip_stream)
{
    if (!times_ms || n_frames < 1 || n_frames > RT_MAX_BATCH_FRAMES) return set_err(ctx, RT_ERR_INVALID, "n_frames must be 1..32");
    if (frame_num < 0) return set_err(ctx, RT_ERR_INVALID, "bad frame number");
    return render_frames(ctx, scene, cam, rs, times_ms, n_frames, frame_num, tiles, nullptr, d_frame, hip_stream, true);
}

/* ---- pipelined frames --------------------------------------------------------------------------------------------------- */
static rt_status init_slot(rt_ctx *ctx, FrameSlot &fs)
{
    /* (each member on its own: a call that failed half-way is finished by the next one) */
    if (!fs.stream) {
        /* Frames only overlap if their streams sit on different hardware queues.  The runtime keeps a pool of them per stream
         * priority (four each by default, GPU_MAX_HW_QUEUES) and the caller's own streams - the null stream, PyTorch's - already
         * live in the normal-priority pool: a fourth frame's stream would share a queue there and run behind its neighbour
         * (measured: 4 in flight 333 ms per frame, worse than 3; with streams of the high-priority pool 286-290).  Slots 0-3 take
         * the high-priority pool, 4-7 the low-priority one; the priority itself is immaterial (the frames are each other's only
         * competitors: alternating the pools over the slots measures the same at every depth). */
        int lo = 0, hi = 0;
        const int k = (int)(&fs - ctx->pipe.slots);
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
            RT_HIP(ctx, hipStreamCreateWithPriority(&fs.stream, hipStreamNonBlocking, k < 4 ? hi : lo), "creating a pipelined frame's stream");
        else
            RT_HIP(ctx, hipStreamCreateWithFlags(&fs.stream, hipStreamNonBlocking), "creating a pipelined frame's stream");
    }
    if (!fs.counter) RT_HIP(ctx, hipMalloc((void **)&fs.counter, 1024), "allocating a pipelined frame's ticket counter");
    if (!fs.ev_done) RT_HIP(ctx, hipEventCreateWithFlags(&fs.ev_done, hipEventDisableTiming), "creating a pipelined frame's event");
    if (!fs.ev_free) RT_HIP(ctx, hipEventCreateWithFlags(&fs.ev_free, hipEventDisableTiming), "creating a pipelined frame's event");
    if (!fs.ev_call) RT_HIP(ctx, hipEventCreateWithFlags(&fs.ev_call, hipEventDisableTiming), "creating a pipelined frame's event");
    return RT_OK;
}

extern "C" rt_status rt_frame_submit(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_settings *rs,
                                     int32_t time_ms, const rt_tile_spec *tiles)
{
    if (!ctx) return RT_ERR_INVALID;
    Pipeline &pl = ctx->pipe;
    if (pl.pending >= pl.depth) return set_err(ctx, RT_ERR_BUSY, "as many frames as rt_frame_depth allows are in flight: collect one first");
    RT_HIP(ctx, hipSetDevice(ctx->device), "selecting device");
    /* a free slot: one that is not waiting to be collected */
    int k = -1;
    for (int i = 0; i < RT_PIPELINE_DEPTH && k < 0; i++) {
        bool busy = false;
        for (int j = 0; j < pl.pending; j++) busy = busy || pl.order[j] == i;
        if (!busy) k = i;
    }
    FrameSlot &fs = pl.slots[k];
    rt_status st = init_slot(ctx, fs);
    if (st != RT_OK) return st;
    /* (the slot's plane is rewritten behind the blend that read it last: that ran on the slot's stream) */
    st = render_frames(ctx, scene, cam, rs, &time_ms, 1, 0, tiles, nullptr, nullptr, nullptr, true, &fs);
    if (st != RT_OK) return st;
    pl.order[pl.pending++] = k;
    return RT_OK;
}

extern "C" rt_status rt_frame_collect(rt_ctx *ctx, int32_t frame_num, float *d_frame, void *hip_stream)
{
    if (!ctx) return RT_ERR_INVALID;
    Pipeline &pl = ctx->pipe;
    if (pl.pending <= 0) return set_err(ctx, RT_ERR_INVALID, "no frame has been submitted");
    if (frame_num < 0) return set_err(ctx, RT_ERR_INVALID, "bad frame number");
    RT_HIP(ctx, hipSetDevice(ctx->device), "selecting device");
    FrameSlot &fs = pl.slots[pl.order[0]];
    if (d_frame && fs.n_tiles > 0) {
        /* The blend runs on the frame's own stream, right behind its render kernel, and the caller's stream only waits for it
         * (a kernel queued on the caller's stream may sit behind whatever shares that stream's hardware queue).  Ordered behind
         * what the caller has queued so far (it may still be reading d_frame) and behind the previous frame's blend. */
        hipStream_t stream = (hipStream_t)hip_stream;
        RT_HIP(ctx, hipEventRecord(fs.ev_call, stream), "recording the collector's position");
        RT_HIP(ctx, hipStreamWaitEvent(fs.stream, fs.ev_call, 0), "ordering the blend behind the collector's stream");
        if (pl.last_fold >= 0 && pl.last_fold != pl.order[0])
            RT_HIP(ctx, hipStreamWaitEvent(fs.stream, pl.slots[pl.last_fold].ev_free, 0), "ordering the blend behind the previous frame's");
        rt_status st = fold_planes(ctx, fs.plane, fs.plane_floats, 1, frame_num, d_frame, fs.compact, fs.listed != 0, pl.d_list, fs.n_tiles, fs.tiles_x,
                                   fs.band_rows, fs.band_first, fs.band_stride, fs.width, fs.height, fs.stream);
        if (st != RT_OK) return st;
        RT_HIP(ctx, hipEventRecord(fs.ev_free, fs.stream), "recording a pipelined frame's blend");
        fs.folded = true;
        pl.last_fold = pl.order[0];
        RT_HIP(ctx, hipStreamWaitEvent(stream, fs.ev_free, 0), "ordering the collector's stream behind the blend");
    }
    for (int j = 1; j < pl.pending; j++) pl.order[j - 1] = pl.order[j];
    pl.pending--;
    return RT_OK;
}

extern "C" int32_t rt_frames_pending(const rt_ctx *ctx) { return ctx ? ctx->pipe.pending : 0; }

extern "C" rt_status rt_frame_depth(rt_ctx *ctx, int32_t depth)
{
    if (!ctx) return RT_ERR_INVALID;
    if (depth < 1 || depth > RT_PIPELINE_DEPTH) return set_err(ctx, RT_ERR_INVALID, "the depth of the frame pipeline is 1..RT_PIPELINE_DEPTH");
    if (ctx->pipe.pending > 0) return set_err(ctx, RT_ERR_BUSY, "frames are in flight: collect them before changing the depth");
    ctx->pipe.depth = depth;
    return RT_OK;
}

namespace { rt_status ensure_frame_buffers(rt_ctx *ctx, size_t bytes); }

/* host-buffer form of rt_frame_collect, with rt_render's contract for previous_render and *frame_num */
extern "C" rt_status rt_frame_collect_host(rt_ctx *ctx, int32_t *frame_num, float *previous_render)
{
    if (!ctx || !frame_num) return set_err(ctx, RT_ERR_INVALID, "null argument");
    Pipeline &pl = ctx->pipe;
    if (pl.pending <= 0) return set_err(ctx, RT_ERR_INVALID, "no frame has been submitted");
    if (!previous_render) return rt_frame_collect(ctx, 0, nullptr, nullptr);          /* discard */
    if (*frame_num < 0) return set_err(ctx, RT_ERR_INVALID, "bad frame number");
    FrameSlot &fs = pl.slots[pl.order[0]];
    if (fs.compact || fs.listed || fs.band_stride != 1) return set_err(ctx, RT_ERR_INVALID, "the host-buffer form collects whole frames (submitted without a tile spec)");
    RT_HIP(ctx, hipSetDevice(ctx->device), "selecting device");
    const size_t bytes = (size_t)fs.width * (size_t)fs.height * 3 * sizeof(float);
    rt_status st = ensure_frame_buffers(ctx, bytes);
    if (st != RT_OK) return st;
    /* everything on the frame's own stream, behind its render kernel; the host waits for that stream only - the younger frames
     * keep running (a hipDeviceSynchronize, as in rt_render, would wait for them too) */
    if (*frame_num > 0) RT_HIP(ctx, hipMemcpyAsync(ctx->d_out, previous_render, bytes, hipMemcpyHostToDevice, fs.stream), "copying previous frame");
    st = rt_frame_collect(ctx, *frame_num, ctx->d_out, fs.stream);
    if (st != RT_OK) return st;
    RT_HIP(ctx, hipMemcpyAsync(previous_render, ctx->d_out, bytes, hipMemcpyDeviceToHost, fs.stream), "copying frame to host");
    RT_HIP(ctx, hipStreamSynchronize(fs.stream), "render kernel");
    *frame_num += 1;                                   /* src/dispatch.cu:159 */
    return RT_OK;
}

extern "C" rt_status rt_frame_wait(rt_ctx *ctx)
{
    if (!ctx) return RT_ERR_INVALID;
    const Pipeline &pl = ctx->pipe;
    if (pl.last_fold < 0) return RT_OK;
    RT_HIP(ctx, hipSetDevice(ctx->device), "selecting device");
    RT_HIP(ctx, hipEventSynchronize(pl.slots[pl.last_fold].ev_free), "waiting for the collected frame");
    return RT_OK;
}

extern "C" rt_status rt_tile_costs(rt_ctx *ctx, uint32_t *tile_ids, uint32_t *costs, uint32_t *peaks, int32_t capacity, int32_t *count)
{
    if (!ctx || !count || capacity < 0 || (capacity > 0 && (!tile_ids || !costs))) return set_err(ctx, RT_ERR_INVALID, "null argument");
    *count = 0;
    RT_HIP(ctx, hipSetDevice(ctx->device), "selecting device");
    if (ctx->order_key.empty() || ctx->cost_state == 0) return set_err(ctx, RT_ERR_INVALID, "no launch of the current view has collected tile costs");
    if (ctx->cost_state == 1) {
        rt_status st = drain_pipeline(ctx, false);       /* (a pipelined frame may be the launch that measures, or be reading the order) */
        if (st != RT_OK) return st;
        st = read_costs_and_refine(ctx, ctx->last_stream);
        if (st != RT_OK) return st;
    }
    const size_t n = ctx->tiles_host.size();
    if (ctx->cost_host.size() != n || ctx->peak_host.size() != n) return set_err(ctx, RT_ERR_INVALID, "no launch of the current view has collected tile costs");
    const size_t m = n < (size_t)capacity ? n : (size_t)capacity;
    for (size_t i = 0; i < m; i++) { tile_ids[i] = ctx->tiles_host[i]; costs[i] = ctx->cost_host[i]; if (peaks) peaks[i] = ctx->peak_host[i]; }
    *count = (int32_t)n;
    return RT_OK;
}

extern "C" rt_status rt_tiles_copy_device(rt_ctx *ctx, float *d_compact, float *d_frame, int32_t width, int32_t height,
                                          const uint32_t *tile_list, int32_t num_tiles, int32_t to_frame, void *hip_stream)
{
    if (!ctx || !d_compact || !d_frame || width <= 0 || height <= 0 || num_tiles < 0 || (num_tiles > 0 && !tile_list)) return set_err(ctx, RT_ERR_INVALID, "bad argument");
    const int tiles_x = (width + 7) / 8, tiles_y = (height + 7) / 8;
    for (int32_t i = 0; i < num_tiles; i++)
        if (tile_list[i] >= (uint32_t)(tiles_x * tiles_y)) return set_err(ctx, RT_ERR_INVALID, "a tile index outside the image");
    RT_HIP(ctx, hipSetDevice(ctx->device), "selecting device");
    const uint32_t *d_list = nullptr;
    rt_status st = device_tile_list(ctx, tile_list, num_tiles, (hipStream_t)hip_stream, &d_list);
    if (st != RT_OK) return st;
    RT_HIP(ctx, rt_launch_tiles_copy(d_compact, d_frame, d_list, num_tiles, tiles_x, width, height, to_frame ? 1 : 0, (hipStream_t)hip_stream), "launching the tile copy");
    return RT_OK;
}

/* test hook: evaluates one function of rt_math.h / rt_rng.h on the DEVICE for n inputs given
 * and returned as raw 32-bit patterns in host memory */
extern "C" rt_status rt_debug_eval(rt_ctx *ctx, int32_t op, const uint32_t *in, uint32_t *out, int32_t n)
{
    if (!ctx || !in || !out || n <= 0 || op < 0 || op > 15) return set_err(ctx, RT_ERR_INVALID, "bad argument");
    RT_HIP(ctx, hipSetDevice(ctx->device), "selecting device");
    uint32_t *d_in = nullptr, *d_out = nullptr;
    RT_HIP(ctx, hipMalloc((void **)&d_in, (size_t)n * 4), "allocating eval input");
    hipError_t e = hipMalloc((void **)&d_out, (size_t)n * 4);
    if (e == hipSuccess) e = hipMemcpy(d_in, in, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = rt_launch_eval(op, d_in, d_out, n, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return hip_fail(ctx, e, "evaluating on the device");
    return RT_OK;
}

/* test hook: the short reciprocal and square root of the device code (rt_pixel.h rt_rcp_short / rt_sqrt_short) against the
 * compiler's IEEE expansions for every one of the 2^32 binary32 inputs, on the device.  out4 = {reciprocal: inputs inside its
 * range that differ, inputs inside its range; square root: likewise} */
extern "C" hipError_t rt_launch_exhaustive(unsigned long long *out4, hipStream_t stream);
extern "C" rt_status rt_debug_exhaustive(rt_ctx *ctx, unsigned long long *out4)
{
    if (!ctx || !out4) return set_err(ctx, RT_ERR_INVALID, "bad argument");
    RT_HIP(ctx, hipSetDevice(ctx->device), "selecting device");
    unsigned long long *d = nullptr;
    RT_HIP(ctx, hipMalloc((void **)&d, 32), "allocating counters");
    hipError_t e = hipMemset(d, 0, 32);
    if (e == hipSuccess) e = rt_launch_exhaustive(d, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out4, d, 32, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(ctx, e, "exhaustive check on the device");
    return RT_OK;
}

/* development hook: copies the 48 section counters / timers of a -DRT_STATS build (zeros otherwise) */
extern "C" rt_status rt_debug_read_stats(rt_ctx *ctx, unsigned long long *out48)
{
    if (!ctx || !out48) return RT_ERR_INVALID;
    RT_HIP(ctx, hipDeviceSynchronize(), "waiting for render kernel");
    RT_HIP(ctx, hipMemcpy(out48, ctx->tile_counter + 16, 48 * 8, hipMemcpyDeviceToHost), "reading stats");
    return RT_OK;
}

extern "C" rt_status rt_last_kernel_ms(rt_ctx *ctx, float *ms)
{
    if (!ctx || !ms) return RT_ERR_INVALID;
    if (!ctx->have_timing) return set_err(ctx, RT_ERR_INVALID, "no render has been launched yet");
    RT_HIP(ctx, hipEventSynchronize(ctx->ev_stop), "waiting for render kernel");
    RT_HIP(ctx, hipEventElapsedTime(ms, ctx->ev_start, ctx->ev_stop), "reading kernel time");
    return RT_OK;
}

namespace {

rt_status ensure_frame_buffers(rt_ctx *ctx, size_t bytes)
{
    if (bytes == ctx->frame_bytes) return RT_OK;
    if (ctx->d_prev) (void)hipFree(ctx->d_prev);
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    ctx->d_prev = ctx->d_out = nullptr;
    ctx->frame_bytes = 0;
    RT_HIP(ctx, hipMalloc((void **)&ctx->d_prev, bytes), "allocating previous-frame buffer");
    RT_HIP(ctx, hipMalloc((void **)&ctx->d_out, bytes), "allocating frame buffer");
    ctx->frame_bytes = bytes;
    return RT_OK;
}

}  // namespace

extern "C" rt_status rt_render(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_settings *rs,
                               int32_t time_ms, int32_t *frame_num, float *previous_render)
{
    if (!ctx || !cam || !frame_num || !previous_render) return set_err(ctx, RT_ERR_INVALID, "null argument");
    if (cam->width <= 0 || cam->height <= 0) return set_err(ctx, RT_ERR_INVALID, "bad image size");
    RT_HIP(ctx, hipSetDevice(ctx->device), "selecting device");
    const size_t bytes = (size_t)cam->width * (size_t)cam->height * 3 * sizeof(float);
    rt_status st = ensure_frame_buffers(ctx, bytes);
    if (st != RT_OK) return st;
    RT_HIP(ctx, hipMemcpy(ctx->d_prev, previous_render, bytes, hipMemcpyHostToDevice), "copying previous frame");
    st = rt_render_device(ctx, scene, cam, rs, time_ms, *frame_num, nullptr, ctx->d_prev, ctx->d_out, nullptr);
    if (st != RT_OK) return st;
    RT_HIP(ctx, hipDeviceSynchronize(), "render kernel");
    RT_HIP(ctx, hipMemcpy(previous_render, ctx->d_out, bytes, hipMemcpyDeviceToHost), "copying frame to host");
    *frame_num += 1;                                   /* src/dispatch.cu:159 */
    RT_HIP(ctx, hipPeekAtLastError(), "final check after render");   /* src/dispatch.cu:161-162 */
    return RT_OK;
}

/* the same for n_frames consecutive frames at once (seeds times_ms[i]): one multi-frame launch per
 * rt_max_batch_frames frames; *frame_num advances by n_frames */
extern "C" rt_status rt_render_frames(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_settings *rs,
                                      const int32_t *times_ms, int32_t n_frames, int32_t *frame_num, float *previous_render)
{
    if (!ctx || !cam || !frame_num || !previous_render || !times_ms || n_frames < 1) return set_err(ctx, RT_ERR_INVALID, "null argument");
    if (cam->width <= 0 || cam->height <= 0) return set_err(ctx, RT_ERR_INVALID, "bad image size");
    RT_HIP(ctx, hipSetDevice(ctx->device), "selecting device");
    const size_t bytes = (size_t)cam->width * (size_t)cam->height * 3 * sizeof(float);
    rt_status st = ensure_frame_buffers(ctx, bytes);
    if (st != RT_OK) return st;
    RT_HIP(ctx, hipMemcpy(ctx->d_out, previous_render, bytes, hipMemcpyHostToDevice), "copying previous frame");
    const int32_t cap = batch_cap(ctx, bytes / 4);
    for (int32_t done = 0; done < n_frames;) {
        const int32_t k = n_frames - done < cap ? n_frames - done : cap;
        st = rt_render_device_batch(ctx, scene, cam, rs, times_ms + done, k, *frame_num + done, nullptr, ctx->d_out, nullptr);
        if (st != RT_OK) return st;
        done += k;
    }
    RT_HIP(ctx, hipDeviceSynchronize(), "render kernel");
    RT_HIP(ctx, hipMemcpy(previous_render, ctx->d_out, bytes, hipMemcpyDeviceToHost), "copying frame to host");
    *frame_num += n_frames;
    RT_HIP(ctx, hipPeekAtLastError(), "final check after render");
    return RT_OK;
}

/* waits for the most recent launch of this context */
extern "C" rt_status rt_ctx_synchronize(rt_ctx *ctx)
{
    if (!ctx) return RT_ERR_INVALID;
    RT_HIP(ctx, hipSetDevice(ctx->device), "selecting device");
    {
        rt_status st = drain_pipeline(ctx, true);
        if (st != RT_OK) return st;
    }
    if (!ctx->launched) return RT_OK;
    RT_HIP(ctx, hipEventSynchronize(ctx->ev_stop), "waiting for render kernel");
    return RT_OK;
}

/* =============================================================================================
 * Multi-GPU from one host thread (SURVEY.md §8(b) "rt_gather", §8(e)): every rank renders the tiles it
 * owns on its own GPU into a compact buffer; the buffers travel to the root's GPU with one peer copy per
 * rank (xGMI) and are de-interleaved there.  No collective is needed: nothing is reduced, every pixel has
 * one owner.  Ownership is either static (bands: rank i owns the bands b % n == i) or cost-balanced (tile
 * lists dealt longest-processing-time-first from the costs the view's first launch measures).
 * ============================================================================================= */
namespace {

struct BandLayout {
    int W = 0, H = 0, band_rows = 8, n = 1;
    size_t chunk() const { return (size_t)band_rows * (size_t)W * 3; }            /* floats per band */
    int bands_total() const { return (H + band_rows - 1) / band_rows; }
    int bands_of(int i) const { const int t = bands_total(); return t > i ? (t - i + n - 1) / n : 0; }
    size_t floats_of(int i) const { return (size_t)bands_of(i) * chunk(); }
};

/* bands of rank i between a full frame and a compact buffer, both on the current device.  The last band
 * of the image may be ragged: only its valid rows exist in the full frame. */
hipError_t copy_bands(const BandLayout &L, int i, float *full, float *compact, bool to_full, hipStream_t stream)
{
    const int nb = L.bands_of(i);
    if (nb == 0) return hipSuccess;
    const size_t chunk_bytes = L.chunk() * 4, full_pitch = chunk_bytes * (size_t)L.n;
    const int last_band = i + (nb - 1) * L.n;
    const int last_rows = std::min(L.band_rows, L.H - last_band * L.band_rows);
    const int whole = last_rows == L.band_rows ? nb : nb - 1;
    float *full0 = full + (size_t)i * L.chunk();
    hipError_t e = hipSuccess;
    if (whole > 0) {
        e = to_full ? hipMemcpy2DAsync(full0, full_pitch, compact, chunk_bytes, chunk_bytes, (size_t)whole, hipMemcpyDeviceToDevice, stream)
                    : hipMemcpy2DAsync(compact, chunk_bytes, full0, full_pitch, chunk_bytes, (size_t)whole, hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
    }
    if (whole < nb) {
        const size_t bytes = (size_t)last_rows * (size_t)L.W * 12;
        float *f = full + (size_t)last_band * L.chunk(), *c = compact + (size_t)(nb - 1) * L.chunk();
        e = to_full ? hipMemcpyAsync(f, c, bytes, hipMemcpyDeviceToDevice, stream) : hipMemcpyAsync(c, f, bytes, hipMemcpyDeviceToDevice, stream);
    }
    return e;
}

rt_status multi_prepare(rt_ctx *ctx)
{
    RT_HIP(ctx, hipSetDevice(ctx->device), "selecting device");
    if (!ctx->multi_stream) RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->multi_stream, hipStreamNonBlocking), "creating the rank's stream");
    if (!ctx->ev_multi) RT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_multi, hipEventDisableTiming), "creating the rank's event");
    return RT_OK;
}

/* device-to-device bytes between two contexts' GPUs (a plain copy when they share one), on a stream of
 * the CURRENT device */
hipError_t copy_between(float *dst, int dst_dev, const float *src, int src_dev, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return hipSuccess;
    if (dst_dev == src_dev) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
    return hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, stream);
}

/* The root's landing area for src's compact image, at least `need` floats, ready to be written in the order of `s0`:
 * s0 has been made to wait for the last de-interleave that read it (which may have been queued on another stream
 * of the root by an earlier call).  Leaves the root's device current. */
rt_status stage_for(rt_ctx *root, rt_ctx *src, size_t need, hipStream_t s0, Stage **out)
{
    RT_HIP(root, hipSetDevice(root->device), "selecting device");
    /* rt_ctx_destroy of a source erases its entry from every live root's map under g_ctx_mutex: look the
     * entry up under the same lock.  (std::map nodes do not move: the reference stays valid while src is alive.) */
    Stage *sgp;
    { std::lock_guard<std::mutex> lock(g_ctx_mutex); sgp = &root->stages[src]; }
    Stage &sg = *sgp;
    if (!sg.ev_free) RT_HIP(root, hipEventCreateWithFlags(&sg.ev_free, hipEventDisableTiming), "creating the staging event");
    if (sg.cap < need || !sg.d) {
        /* nothing may still be reading or writing the old area */
        if (sg.used) RT_HIP(root, hipEventSynchronize(sg.ev_free), "waiting for the staging area");
        if (src->multi_stream) { (void)hipSetDevice(src->device); (void)hipStreamSynchronize(src->multi_stream); (void)hipSetDevice(root->device); }
        rt_status st = grow(root, &sg.d, &sg.cap, need, "allocating the gather staging area");
        if (st != RT_OK) return st;
        sg.used = false;
    }
    if (sg.used) RT_HIP(root, hipStreamWaitEvent(s0, sg.ev_free, 0), "ordering behind the previous de-interleave");
    *out = &sg;
    return RT_OK;
}

int peer_access(rt_ctx *a, rt_ctx *b)
{
    if (a->device == b->device) return 1;
    auto it = a->peer_ok.find(b->device);
    if (it != a->peer_ok.end()) return it->second;
    int ok = 1;
    for (int dir = 0; dir < 2; dir++) {
        const int from = dir ? b->device : a->device, to = dir ? a->device : b->device;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, from, to) != hipSuccess || !can) { ok = 0; continue; }
        if (hipSetDevice(from) != hipSuccess) { ok = 0; continue; }
        const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = 0;
    }
    (void)hipGetLastError();           /* "already enabled" must not surface as a later launch error */
    (void)hipSetDevice(a->device);
    a->peer_ok[b->device] = ok;
    b->peer_ok[a->device] = ok;
    return ok;
}

}  // namespace

extern "C" int32_t rt_peer_access(rt_ctx *a, rt_ctx *b)
{
    if (!a || !b) return -RT_ERR_INVALID;
    return peer_access(a, b);
}

/* The exchange step alone: the compact buffer `d_bands` of the rank described by `src_tiles`
 * (on src's GPU) lands in the full frame `d_frame` on root's GPU.  Ordered behind src's most recent
 * launch; the frame is complete in `root_stream` order. */
extern "C" rt_status rt_gather(rt_ctx *root, float *d_frame, int32_t width, int32_t height, rt_ctx *src, const float *d_bands,
                               const rt_tile_spec *src_tiles, void *root_stream)
{
    if (!root || !src || !d_frame || !d_bands || !src_tiles) return set_err(root, RT_ERR_INVALID, "null argument");
    if (width <= 0 || height <= 0) return set_err(root, RT_ERR_INVALID, "bad image size");
    const bool listed = src_tiles->tile_list != nullptr;
    const int tiles_x = (width + 7) / 8, tiles_y = (height + 7) / 8;
    BandLayout L;
    size_t floats;
    if (listed) {
        if (src_tiles->num_tiles < 0 || src_tiles->num_tiles > tiles_x * tiles_y) return set_err(root, RT_ERR_INVALID, "bad tile spec");
        for (int32_t i = 0; i < src_tiles->num_tiles; i++)
            if (src_tiles->tile_list[i] >= (uint32_t)(tiles_x * tiles_y)) return set_err(root, RT_ERR_INVALID, "a tile index outside the image");
        floats = (size_t)src_tiles->num_tiles * 192;
    } else {
        if (src_tiles->band_rows <= 0 || (src_tiles->band_rows & 7) || src_tiles->band_stride <= 0 ||
            src_tiles->band_first < 0 || src_tiles->band_first >= src_tiles->band_stride)
            return set_err(root, RT_ERR_INVALID, "bad tile spec");
        L.W = width; L.H = height; L.band_rows = src_tiles->band_rows; L.n = src_tiles->band_stride;
        floats = L.floats_of(src_tiles->band_first);
    }
    rt_status st = multi_prepare(root);
    if (st != RT_OK) return st;
    hipStream_t s0 = (hipStream_t)root_stream;
    float *from = const_cast<float *>(d_bands);
    Stage *sg = nullptr;
    if (src->device != root->device) {
        /* one peer copy of the whole compact buffer into the root's staging area for this source: behind src's last
         * launch, and behind the de-interleave of the PREVIOUS gather from this source, which reads the same area on the
         * root's stream (src may be a frame ahead of the root) */
        if ((st = multi_prepare(src)) != RT_OK) return set_err(root, st, rt_last_error(src));
        (void)peer_access(src, root);
        if ((st = stage_for(root, src, floats, s0, &sg)) != RT_OK) return st;
        RT_HIP(src, hipSetDevice(src->device), "selecting device");
        if (sg->used) RT_HIP(src, hipStreamWaitEvent(src->multi_stream, sg->ev_free, 0), "ordering the copy behind the previous de-interleave");
        if (src->launched) RT_HIP(src, hipStreamWaitEvent(src->multi_stream, src->ev_stop, 0), "ordering the gather behind the render");
        RT_HIP(src, copy_between(sg->d, root->device, d_bands, src->device, floats * 4, src->multi_stream), "copying the rank's image to the root GPU");
        RT_HIP(src, hipEventRecord(src->ev_multi, src->multi_stream), "recording the gather event");
        RT_HIP(root, hipSetDevice(root->device), "selecting device");
        RT_HIP(root, hipStreamWaitEvent(s0, src->ev_multi, 0), "ordering the de-interleave behind the copy");
        from = sg->d;
    } else {
        RT_HIP(root, hipSetDevice(root->device), "selecting device");
        if (src->launched) RT_HIP(root, hipStreamWaitEvent(s0, src->ev_stop, 0), "ordering the gather behind the render");
    }
    if (listed) {
        const uint32_t *d_list = nullptr;
        if ((st = device_tile_list(root, src_tiles->tile_list, src_tiles->num_tiles, s0, &d_list)) != RT_OK) return st;
        RT_HIP(root, rt_launch_tiles_copy(from, d_frame, d_list, src_tiles->num_tiles, tiles_x, width, height, 1, s0), "de-interleaving tiles");
    } else {
        RT_HIP(root, copy_bands(L, src_tiles->band_first, d_frame, from, true, s0), "de-interleaving bands");
    }
    if (sg) {
        RT_HIP(root, hipEventRecord(sg->ev_free, s0), "recording the staging event");
        sg->used = true;
    }
    return RT_OK;
}

namespace {

/* what rank i of a multi-GPU call renders and where its image lives */
struct RankPlan {
    rt_tile_spec spec;
    size_t floats = 0;                   /* size of its compact image */
    const uint32_t *root_list = nullptr; /* tile lists: the rank's list on the ROOT's GPU (for the copies to and from the frame) */
};

}  // namespace

/* n_frames consecutive progressive frames over n_ranks GPUs, accumulated in place in d_frame (a full
 * frame on ranks[0]'s GPU).  Replaces run_ray_tracer src/dispatch.cu:127-153 for a node: what one
 * device did there, n do here, each for the tiles it owns. */
extern "C" rt_status rt_render_multi_device(const rt_rank *ranks, int32_t n_ranks, const rt_camera *cam, const rt_render_settings *rs,
                                            const int32_t *times_ms, int32_t n_frames, int32_t frame_num, int32_t band_rows,
                                            float *d_frame, void *hip_stream)
{
    if (!ranks || n_ranks < 1 || !ranks[0].ctx) return RT_ERR_INVALID;
    rt_ctx *root = ranks[0].ctx;
    if (!cam || !rs || !times_ms || !d_frame || n_frames < 1 || frame_num < 0) return set_err(root, RT_ERR_INVALID, "null argument");
    if (cam->width <= 0 || cam->height <= 0 || band_rows < 0 || (band_rows & 7)) return set_err(root, RT_ERR_INVALID, "bad image size or band_rows (0, or a positive multiple of 8)");
    for (int i = 0; i < n_ranks; i++) {
        if (!ranks[i].ctx || !ranks[i].scene || ranks[i].scene->ctx != ranks[i].ctx) return set_err(root, RT_ERR_INVALID, "rank without a context, or a scene committed on another context");
        for (int j = 0; j < i; j++) if (ranks[j].ctx == ranks[i].ctx) return set_err(root, RT_ERR_INVALID, "a context may appear once (its launch scratch is not shared between ranks)");
    }
    const int W = cam->width, H = cam->height;
    const int tiles_x = (W + 7) / 8, tiles_y = (H + 7) / 8;
    const bool listed = band_rows == 0;
    hipStream_t s0 = (hipStream_t)hip_stream;
    rt_status st;
    if ((st = multi_prepare(root)) != RT_OK) return st;
    for (int i = 0; i < n_ranks; i++) {
        rt_ctx *c = ranks[i].ctx;
        if ((st = multi_prepare(c)) != RT_OK) return set_err(root, st, rt_last_error(c));
        (void)peer_access(c, root);      /* direct xGMI copies instead of staging through the host when the platform allows */
    }
    RT_HIP(root, hipSetDevice(root->device), "selecting device");
    /* this call holds every rank's list on the root's GPU at once (the copies to and from the frame), and rank 0's launch
     * looks its own up as well: what the cache hands out from here on is not recycled before the call has queued its work */
    struct Pin { rt_ctx *c; explicit Pin(rt_ctx *c_) : c(c_) { pin_tile_lists(c); } ~Pin() { unpin_tile_lists(c); } } pin(root);

    /* ---- who renders what -------------------------------------------------------------------------- */
    BandLayout L;
    L.W = W; L.H = H; L.band_rows = listed ? 8 : band_rows; L.n = n_ranks;
    std::vector<RankPlan> plan((size_t)n_ranks);
    MultiState &ms = root->multi;
    bool collect_after = false;
    if (listed) {
        std::vector<uint32_t> key;
        for (int i = 0; i < 3; i++) { uint32_t u; std::memcpy(&u, &cam->cam_pos[i], 4); key.push_back(u); }
        for (int i = 0; i < 3; i++) { uint32_t u; std::memcpy(&u, &cam->tl_pixel_pos[i], 4); key.push_back(u); }
        for (int i = 0; i < 3; i++) { uint32_t u; std::memcpy(&u, &cam->delta_u[i], 4); key.push_back(u); }
        for (int i = 0; i < 3; i++) { uint32_t u; std::memcpy(&u, &cam->delta_v[i], 4); key.push_back(u); }
        key.push_back((uint32_t)W); key.push_back((uint32_t)H); key.push_back((uint32_t)n_ranks);
        /* the bounce limit, and whether this is a preview-quality launch (< 8 samples) or a real one: ownership measured on
         * the one says little about the other */
        key.push_back((uint32_t)rs->reflection_limit); key.push_back(rs->rays_per_pixel < 8 ? 0u : 1u);
        for (int i = 0; i < n_ranks; i++) { key.push_back(ranks[i].scene->uid); key.push_back((uint32_t)ranks[i].ctx->device); }
        std::vector<int32_t> owner((size_t)tiles_x * tiles_y);
        if (key != ms.key || ms.lists.size() != (size_t)n_ranks) {
            /* a new view: interleaved ownership; this call's first launches measure what the tiles cost */
            if ((st = rt_partition_tiles(nullptr, tiles_x, tiles_y, n_ranks, owner.data())) != RT_OK) return set_err(root, st, "cannot partition the image");
            ms.lists.assign((size_t)n_ranks, std::vector<uint32_t>());
            ms.costs.assign((size_t)n_ranks, std::vector<uint32_t>());
            ms.peaks.assign((size_t)n_ranks, std::vector<uint32_t>());
            for (size_t g = 0; g < owner.size(); g++) ms.lists[(size_t)owner[g]].push_back((uint32_t)g);
            ms.key = key;
            ms.stage = 0;
        } else if (ms.stage == 1) {
            /* the previous call measured the tiles: collect every rank's figures (this waits for those launches, which
             * the caller has normally consumed already), then deal the tiles out again by cost.  All read-backs happen
             * here, before any rank is launched, so no rank's launch waits on another rank's kernel. */
            std::vector<uint32_t> cost((size_t)tiles_x * tiles_y, 0u), peak((size_t)tiles_x * tiles_y, 0u);
            bool have_all = true;
            for (int i = 0; i < n_ranks && have_all; i++) {
                const size_t cnt = ms.lists[(size_t)i].size();
                if (cnt == 0) continue;
                std::vector<uint32_t> ids(cnt), cs(cnt), ps(cnt);
                int32_t got = 0;
                if (rt_tile_costs(ranks[i].ctx, ids.data(), cs.data(), ps.data(), (int32_t)cnt, &got) != RT_OK || (size_t)got != cnt) { have_all = false; break; }
                for (size_t k = 0; k < cnt && have_all; k++) {
                    if (ids[k] != ms.lists[(size_t)i][k]) have_all = false;     /* the rank's context has rendered another view since */
                    else { cost[ids[k]] = cs[k]; peak[ids[k]] = ps[k]; }
                }
            }
            RT_HIP(root, hipSetDevice(root->device), "selecting device");
            if (have_all) {
                if ((st = rt_partition_tiles(cost.data(), tiles_x, tiles_y, n_ranks, owner.data())) != RT_OK) return set_err(root, st, "cannot partition the image");
                ms.lists.assign((size_t)n_ranks, std::vector<uint32_t>());
                ms.costs.assign((size_t)n_ranks, std::vector<uint32_t>());
                ms.peaks.assign((size_t)n_ranks, std::vector<uint32_t>());
                for (size_t g = 0; g < owner.size(); g++) {
                    ms.lists[(size_t)owner[g]].push_back((uint32_t)g); ms.costs[(size_t)owner[g]].push_back(cost[g]); ms.peaks[(size_t)owner[g]].push_back(peak[g]);
                }
                ms.stage = 2;
            } else {
                ms.stage = 0;            /* (a rank's view was replaced in between: measure again) */
            }
        }
        collect_after = ms.stage == 0;
        for (int i = 0; i < n_ranks; i++) {
            RankPlan &p = plan[(size_t)i];
            std::memset(&p.spec, 0, sizeof p.spec);
            p.spec.compact = 1;
            p.spec.tile_list = ms.lists[(size_t)i].data();
            p.spec.num_tiles = (int32_t)ms.lists[(size_t)i].size();
            p.spec.tile_cost = (ms.stage == 2 && p.spec.num_tiles > 0) ? ms.costs[(size_t)i].data() : nullptr;
            p.spec.tile_peak = (ms.stage == 2 && p.spec.num_tiles > 0) ? ms.peaks[(size_t)i].data() : nullptr;
            /* (an empty vector's data() may be null, which would read as "no list") */
            static const uint32_t none = 0;
            if (!p.spec.tile_list) p.spec.tile_list = &none;
            p.floats = (size_t)p.spec.num_tiles * 192;
            if ((st = device_tile_list(root, p.spec.tile_list, p.spec.num_tiles, s0, &p.root_list)) != RT_OK) return st;
        }
    } else {
        for (int i = 0; i < n_ranks; i++) {
            RankPlan &p = plan[(size_t)i];
            std::memset(&p.spec, 0, sizeof p.spec);
            p.spec.band_rows = band_rows; p.spec.band_first = i; p.spec.band_stride = n_ranks; p.spec.compact = 1;
            p.floats = L.floats_of(i);
        }
    }

    /* ---- buffers: every rank's compact image on its GPU, a landing area per rank on the root's ---------- */
    std::vector<Stage *> stage((size_t)n_ranks, nullptr);
    for (int i = 0; i < n_ranks; i++) {
        rt_ctx *c = ranks[i].ctx;
        if (c->bands_cap < plan[(size_t)i].floats || !c->d_bands) {
            RT_HIP(c, hipSetDevice(c->device), "selecting device");
            RT_HIP(c, hipStreamSynchronize(c->multi_stream), "waiting for the rank's image buffer");
            if ((st = grow(c, &c->d_bands, &c->bands_cap, plan[(size_t)i].floats, "allocating the rank's image buffer")) != RT_OK) return set_err(root, st, rt_last_error(c));
        }
        if ((st = stage_for(root, c, plan[(size_t)i].floats, s0, &stage[(size_t)i])) != RT_OK) return st;
    }

    /* ---- the image so far goes out to its owners.  (root->ev_multi is recorded twice in this function: here, on the
     * caller's stream, as "the ranks may start", and further down on rank 0's stream as "rank 0's image is in the
     * staging area"; a hipStreamWaitEvent captures the record that precedes it, and the calls are in that order.) */
    RT_HIP(root, hipSetDevice(root->device), "selecting device");
    if (frame_num > 0) {
        for (int i = 0; i < n_ranks; i++) {
            if (plan[(size_t)i].floats == 0) continue;
            if (listed) RT_HIP(root, rt_launch_tiles_copy(stage[(size_t)i]->d, d_frame, plan[(size_t)i].root_list, plan[(size_t)i].spec.num_tiles, tiles_x, W, H, 0, s0), "interleaving tiles");
            else RT_HIP(root, copy_bands(L, i, d_frame, stage[(size_t)i]->d, false, s0), "interleaving bands");
        }
    }
    /* RT_AMD_MULTI_CAREFUL=1 (diagnosis): the host waits for every stream involved after each phase, so that no ordering between
     * devices rests on an event - if a frame is wrong or a call hangs in the asynchronous form but not in this one, the
     * fault is in the event choreography, otherwise in the copies themselves; an error names the phase it was found in.
     * bench.py's capi_multi leg falls back to it by itself (this path has never run on two physical GPUs). */
    auto careful = [&](const char *phase) -> rt_status {
        if (!root->multi_careful) return RT_OK;
        for (int i = 0; i < n_ranks; i++) {
            rt_ctx *c = ranks[i].ctx;
            hipError_t e = hipSetDevice(c->device);
            if (e == hipSuccess) e = hipStreamSynchronize(c->multi_stream);
            if (e != hipSuccess) { (void)hipSetDevice(root->device); return hip_fail(root, e, phase); }
        }
        hipError_t e = hipSetDevice(root->device);
        if (e == hipSuccess) e = hipStreamSynchronize(s0);
        if (e != hipSuccess) return hip_fail(root, e, phase);
        return RT_OK;
    };
    /* (frame 0 ignores the buffers' content, but the staging areas may still be read by an earlier call's
     * de-interleave: stage_for has ordered s0 behind that, and the ranks start behind s0) */
    RT_HIP(root, hipEventRecord(root->ev_multi, s0), "recording the start event");
    if ((st = careful("careful mode: waiting for the caller's stream before the ranks start")) != RT_OK) return st;
    for (int i = 0; i < n_ranks; i++) {
        rt_ctx *c = ranks[i].ctx;
        RT_HIP(c, hipSetDevice(c->device), "selecting device");
        RT_HIP(c, hipStreamWaitEvent(c->multi_stream, root->ev_multi, 0), "ordering the ranks behind the caller's stream");
        if (frame_num > 0)
            RT_HIP(c, copy_between(c->d_bands, c->device, stage[(size_t)i]->d, root->device, plan[(size_t)i].floats * 4, c->multi_stream), "copying the image so far to its owner");
    }
    if ((st = careful("careful mode: waiting for the image so far to reach its owners (scatter-out)")) != RT_OK) return st;
    /* ---- every rank renders its tiles (asynchronous launches from this one thread) and sends them back ---- */
    for (int i = 0; i < n_ranks; i++) {
        rt_ctx *c = ranks[i].ctx;
        if (plan[(size_t)i].floats == 0) continue;
        RT_HIP(c, hipSetDevice(c->device), "selecting device");
        const int32_t cap = batch_cap(c, plan[(size_t)i].floats);
        for (int32_t done = 0; done < n_frames;) {
            const int32_t k = std::min<int32_t>(n_frames - done, cap);
            st = rt_render_device_batch(c, ranks[i].scene, cam, rs, times_ms + done, k, frame_num + done, &plan[(size_t)i].spec, c->d_bands, c->multi_stream);
            if (st != RT_OK) return set_err(root, st, rt_last_error(c));
            done += k;
        }
        RT_HIP(c, copy_between(stage[(size_t)i]->d, root->device, c->d_bands, c->device, plan[(size_t)i].floats * 4, c->multi_stream), "copying the rank's image to the root GPU");
        RT_HIP(c, hipEventRecord(c->ev_multi, c->multi_stream), "recording the gather event");
    }
    if ((st = careful("careful mode: waiting for the ranks' kernels and their copies to the root GPU")) != RT_OK) return st;
    RT_HIP(root, hipSetDevice(root->device), "selecting device");
    for (int i = 0; i < n_ranks; i++) {
        if (plan[(size_t)i].floats == 0) continue;
        RT_HIP(root, hipStreamWaitEvent(s0, ranks[i].ctx->ev_multi, 0), "ordering the de-interleave behind the copies");
        if (listed) RT_HIP(root, rt_launch_tiles_copy(stage[(size_t)i]->d, d_frame, plan[(size_t)i].root_list, plan[(size_t)i].spec.num_tiles, tiles_x, W, H, 1, s0), "de-interleaving tiles");
        else RT_HIP(root, copy_bands(L, i, d_frame, stage[(size_t)i]->d, true, s0), "de-interleaving bands");
        RT_HIP(root, hipEventRecord(stage[(size_t)i]->ev_free, s0), "recording the staging event");
        stage[(size_t)i]->used = true;
    }
    if ((st = careful("careful mode: waiting for the de-interleave into the frame")) != RT_OK) return st;
    if (listed && collect_after) ms.stage = 1;
    return RT_OK;
}

/* render() src/dispatch.cu:156-163 for a node: host-buffer form of the above */
extern "C" rt_status rt_render_multi(const rt_rank *ranks, int32_t n_ranks, const rt_camera *cam, const rt_render_settings *rs,
                                     const int32_t *times_ms, int32_t n_frames, int32_t *frame_num, float *previous_render)
{
    if (!ranks || n_ranks < 1 || !ranks[0].ctx) return RT_ERR_INVALID;
    rt_ctx *ctx = ranks[0].ctx;
    if (!cam || !frame_num || !previous_render || !times_ms || n_frames < 1) return set_err(ctx, RT_ERR_INVALID, "null argument");
    if (cam->width <= 0 || cam->height <= 0) return set_err(ctx, RT_ERR_INVALID, "bad image size");
    RT_HIP(ctx, hipSetDevice(ctx->device), "selecting device");
    const size_t bytes = (size_t)cam->width * (size_t)cam->height * 3 * sizeof(float);
    rt_status st = ensure_frame_buffers(ctx, bytes);
    if (st != RT_OK) return st;
    RT_HIP(ctx, hipMemcpy(ctx->d_out, previous_render, bytes, hipMemcpyHostToDevice), "copying previous frame");
    st = rt_render_multi_device(ranks, n_ranks, cam, rs, times_ms, n_frames, *frame_num, 0, ctx->d_out, nullptr);
    if (st != RT_OK) return st;
    RT_HIP(ctx, hipSetDevice(ctx->device), "selecting device");
    RT_HIP(ctx, hipStreamSynchronize(nullptr), "render kernels");
    for (int i = 0; i < n_ranks; i++)
        if ((st = rt_ctx_synchronize(ranks[i].ctx)) != RT_OK) return set_err(ctx, st, rt_last_error(ranks[i].ctx));
    RT_HIP(ctx, hipSetDevice(ctx->device), "selecting device");
    RT_HIP(ctx, hipMemcpy(previous_render, ctx->d_out, bytes, hipMemcpyDeviceToHost), "copying frame to host");
    *frame_num += n_frames;
    RT_HIP(ctx, hipPeekAtLastError(), "final check after render");
    return RT_OK;
}

extern "C" rt_status rt_to_rgba8_device(rt_ctx *ctx, const float *d_rgb, int32_t width, int32_t height, uint8_t *d_rgba, void *hip_stream)
{
    if (!ctx || !d_rgb || !d_rgba || width <= 0 || height <= 0) return set_err(ctx, RT_ERR_INVALID, "bad argument");
    RT_HIP(ctx, hipSetDevice(ctx->device), "selecting device");
    RT_HIP(ctx, rt_launch_rgba8(d_rgb, width * height, d_rgba, (hipStream_t)hip_stream), "launching rgba8 kernel");
    return RT_OK;
}
