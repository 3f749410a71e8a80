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
 *    shading record, the object list) is staged into LDS once per workgroup; the object list is
 *    read with wave-uniform addresses (one broadcast per quad) because every lane walks it in
 *    the same order;
 *  - BVH traversal keeps the current node in a register and only the deferred sibling (with
 *    its entry distance) on a per-lane LDS stack laid out [entry][thread], which is
 *    bank-conflict-free; a box is tested once, not twice as in the reference, by carrying the
 *    entry distance instead of re-testing on pop (same predicate, same outcome);
 *  - RNG state, ray, throughput and accumulators live in registers;
 *  - one launch can render several consecutive progressive frames (rt_render_device_batch): a
 *    ticket is one tile of one frame, the host lays the tickets out longest job first over ALL
 *    frames (rt_capi.cpp build_job_order), so every frame's expensive tiles start at once and the
 *    cheap ones fill in behind them; every frame stores its per-pixel mean in a plane of its own and
 *    rt_blend_kernel (below) folds the planes into the frame buffer in frame order.
 *
 * The per-pixel sections (shade / fetch / generate, the primitive tests) are in rt_pixel.h; this
 * file has the two kernels built from them - rt_render_kernel (the default) and the opt-in
 * rt_render_pool_kernel - and the launchers.
 *
 * No MFMA: there is no dense contraction anywhere in this workload.
 * Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize (the reference's a*b+c
 * are two roundings: contraction would change hit/miss decisions; the SLP vectorizer's packed f32
 * pairs cost 40 % more registers and 10 % of the time).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_pixel.h"

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
#define RT_LAP_SPLIT(slot) } RT_LAP(slot); if (p.mode == M_WAIT) {
#define RT_LAP_SPLIT_LEAF(slot) } } RT_LAP(slot); if (p.mode == M_WAIT) { if (cur & RT_REF_LEAF) {
#else
#define RT_LAP(slot) do { } while (0)
#define RT_LAP_SPLIT(slot)
#define RT_LAP_SPLIT_LEAF(slot)
#endif
enum { ST_ITER = 0, ST_SHADE = 1, ST_SHADE_HIT = 2, ST_FETCH = 3, ST_GEN = 4, ST_MESH = 5, ST_MESH_START = 6, ST_WORK_ITER = 7, ST_NODE = 8, ST_LEAF_TRI = 9, ST_POP = 10, ST_DONE_MESH = 11, ST_N = 12 };

#ifdef RT_STATS
#define RT_STATS_FLUSH() do {                                                                              \
    for (int i = 0; i < ST_N; i++) {                                                                          \
        if (st_exec[i]) { atomicAdd(&a.stats[2 * i], (unsigned long long)st_exec[i]); atomicAdd(&a.stats[2 * i + 1], (unsigned long long)st_lanes[i]); } \
    }                                                                                                         \
    RT_LAP(TM_CTL);                                                                                           \
    if (lane == 0) {                                                                                          \
        for (int i = 0; i < TM_N; i++) atomicAdd(&a.stats[24 + i], st_time[i]);                               \
        atomicAdd(&a.stats[24 + TM_N], wall_clock64() - st_wall0);      /* summed wave lifetimes, 100 MHz ticks */ \
        atomicAdd(&a.stats[24 + TM_N + 1], 1ull);                        /* waves */                          \
    }                                                                                                         \
} while (0)
#else
#define RT_STATS_FLUSH() do { } while (0)
#endif

/* SCENE_LDS = false is the fallback for scenes larger than a CU's LDS: the same code reads the
 * scene sections from global memory (they stay L2 / Infinity-Cache resident) and only the
 * traversal stack lives in LDS. */
/* Occupancy.  Built with -fno-slp-vectorize the kernel needs ~90 VGPRs (the SLP vectorizer's packed-f32 pairs cost
 * 128 and ~10 % of the time).  The 1024-thread workgroup of a large mesh scene is one per CU = four waves per SIMD,
 * whatever the registers (its LDS holds the scene and 1024 traversal stacks); the smaller workgroups - scenes without a
 * mesh, and mesh scenes small enough for several workgroups per CU - are register-bound, so they are compiled for
 * five waves per SIMD (<= 96 VGPRs; the allocator then lands on 79-80, which lets six be resident).  The launcher
 * asks the runtime how many workgroups of the chosen shape fit a CU (rt_kernel_blocks_per_cu). */
template <int NT, bool HAS_MESH, bool SCENE_LDS>
__global__ __launch_bounds__(NT, NT == 1024 ? 4 : RT_SMALL_WG_WAVES) void rt_render_kernel(const rt_kernel_args a)
{
    extern __shared__ v4f lds_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & (RT_WAVE - 1);

    Lds L;
    uint2 *stack;        /* [stack_entries + 1][NT] deferred sibling: (entry distance bits, reference) */
    if (SCENE_LDS) {
        /* stage the scene into LDS: coalesced 16-byte loads, one pass per workgroup */
        for (int i = tid; i < a.blob_f4; i += NT) lds_raw[i] = ((const v4f *)a.blob)[i];
        L.nodes = lds_raw + a.off_nodes;
        L.tris = lds_raw + a.off_tris;
        L.objs = lds_raw + a.off_objlds;
        L.meshes = lds_raw + a.off_meshes;
        L.objtab = lds_raw + a.off_objtab;
        stack = (uint2 *)(lds_raw + a.blob_f4);
    } else {
        const v4f *g = (const v4f *)a.blob;
        L.nodes = g + a.off_nodes;
        L.tris = g + a.off_tris;
        L.objs = g + a.off_objlds;
        L.meshes = g + a.off_meshes;
        L.objtab = g + a.off_objtab;
        stack = (uint2 *)lds_raw;
    }
    __syncthreads();

    Frame f;
    frame_init(f, a);
    Px p;
    px_init(p);
    /* ---- per-lane traversal state (registers + LDS stack); a lane is traversing iff M_WAIT ---- */
    uint32_t cur = 0;
    int sp = 0, w_prim = -1;
    float w_best = RT_INF_F;
    Chunk ch;
    ch.next = 0; ch.end = 0; ch.frame = 0; ch.exhausted = false;
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
        /* A ray that hit nothing costs a handful of instructions (sky, end of sample): it is
         * finished on the spot and the lane generates its next ray in this same round.  Hits
         * (several hundred instructions: three Box-Muller draws, four normalisations) are shaded
         * in batches: without a mesh the lanes holding one wait until `shade_batch` of them
         * do, or nobody else can move, while the others go on generating; with a mesh the
         * traversal loop below already yields in batches (`ready_break`). */
        if (p.mode == M_SHADE && p.best_obj < 0) px_shade_miss(p, a, f);
        {
            const int n_hit = __popcll(__ballot(p.mode == M_SHADE));
            const bool others = __ballot(p.mode == M_GEN || (p.mode == M_FETCH && !ch.exhausted)) != 0ull;
            const int n_trav = __popcll(__ballot(p.mode == M_WAIT));
            if (n_hit > 0 && (HAS_MESH ? (n_hit >= a.hit_low || n_trav < a.work_threshold)
                                        : (n_hit >= a.shade_batch || !others))) {
                if (p.mode == M_SHADE) {
                    RT_STAT(ST_SHADE);
                    px_shade(p, a, f, L);
                }
            }
        }
        RT_LAP(TM_SHADE);
        px_fetch(p, ch, a, f, lane);
        RT_LAP(TM_FETCH);
        if (p.mode == M_GEN) {
            RT_STAT(ST_GEN);
            px_gen<HAS_MESH>(p, a, L);
        }
        RT_LAP(TM_GEN);

        if (HAS_MESH) {
            /* ================= MESH: find the next mesh whose root box the ray enters ======= */
            while (p.mode == M_MESH) {
                RT_STAT(ST_MESH);
                if (p.next_mesh >= a.num_meshes) { p.mode = M_SHADE; break; }
                const v4f m0 = L.meshes[2 * p.next_mesh], m1 = L.meshes[2 * p.next_mesh + 1];
                p.next_mesh++;
                /* a NaN direction (Box-Muller on a zero draw) fails every
                 * triangle test: the mesh cannot be hit, no need to walk it */
                if (p.d.x != p.d.x || p.d.y != p.d.y || p.d.z != p.d.z) continue;
                /* the root is pushed unconditionally and tested when popped (src/objects.cu:494-501) */
                const uint32_t root_ref = __float_as_uint(m1.z);
                float rd;
                const bool rh = box_test(m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, p.o, p.inv, rd);
                if (!rh || rd > RT_INF_F || ((root_ref & RT_REF_CHAIN) && !(rd < RT_INF_F))) continue;
                cur = root_ref; sp = 0; w_best = RT_INF_F; w_prim = -1;
                p.mode = M_WAIT;
                RT_STAT(ST_MESH_START);
            }

            RT_LAP(TM_MESH);
            /* ================= WORK: BVH traversal steps (src/objects.cu:487-532, :586-600) ====
             * Runs while enough lanes are traversing; lanes whose ray is finished go back to
             * shading as soon as the traversing group is small.  Visit order, push order and
             * every comparison are the reference's. */
            const V3 o = p.o, d = p.d, inv = p.inv;
            for (;;) {
                const int n_active = __popcll(__ballot(p.mode == M_WAIT));
                if (n_active == 0) break;
                /* lanes holding a hit wait for a batch of `hit_break`; the cheap kinds of ready
                 * lane (generate, fetch, next mesh, a miss) for one of `ready_break` */
                const bool is_hit = p.mode == M_SHADE && p.best_obj >= 0;
                const int n_hit = __popcll(__ballot(is_hit));
                const int n_light = __popcll(__ballot(p.mode != M_WAIT && p.mode != M_DONE && !is_hit));
                /* ... or a smaller batch of hits that, together with the cheap-work lanes, is worth the round: where
                 * every traversal ends in a hit (a closed scene) hits fill a big batch fast and big batches are what
                 * the 700-instruction shade wants; where most rays escape (an open scene) hits are rare, the lanes
                 * holding one would idle for long, and the round is paid for by the escaped lanes anyway */
                if (n_hit + n_light > 0 &&
                    (n_active < a.work_threshold || n_hit >= a.hit_break ||
                     n_light >= a.ready_break ||
                     (n_hit >= a.hit_low && n_hit + n_light >= a.mix_break))) break;
#if defined(RT_COSTMAP) && RT_COSTMAP == 2
                p.c_wsteps += 1;      /* wave-level macro steps this lane lived through */
#endif
                RT_LAP(TM_CTL);
                if (p.mode == M_WAIT) {
                    RT_STAT(ST_WORK_ITER);
                    p.frame_steps += (unsigned)RT_MAX_BATCH_FRAMES;   /* one more traversal macro step (the bits above the frame index) */
                    /* one macro step: descend to a leaf (or run out of children), test the
                     * leaf's triangles, pop the next deferred sibling.  The lane's whole
                     * traversal state is `cur` (+ the stack): an internal node to descend from,
                     * or a leaf whose triangles are tested and after which the stack is popped;
                     * "no child entered" is the empty leaf. */
                    if (!(cur & RT_REF_LEAF)) {
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
                            RT_COST(p.c_steps++);
                            const v4f *n = L.nodes + 4 * (int)(cur & RT_REF_NODE_MASK);
                            v4f q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
                            float ld, rdist;
                            const bool l_push = box_enter(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, inv, w_best, ld);
                            const bool r_push = box_enter(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, inv, w_best, rdist);
                            const uint32_t lref = __float_as_uint(q3.x), rref = __float_as_uint(q3.y);
                            const bool l_first = ld < rdist;
                            /* Of two entered children the one pushed first (left when l_first) is
                             * visited second: it is the deferred sibling.  The other is popped
                             * immediately (its distance is still < best).  With one entered child
                             * that child is next and nothing is deferred. */
                            const bool both = l_push && r_push;
                            const bool entered = l_push || r_push;
                            const uint32_t deferred_ref = l_first ? lref : rref;
                            const float deferred_d = l_first ? ld : rdist;
                            stack[sp * NT + tid] = make_uint2(__float_as_uint(deferred_d), deferred_ref);
                            sp += both ? 1 : 0;
                            const uint32_t next = both ? (l_first ? rref : lref) : (l_push ? lref : rref);
                            cur = entered ? next : RT_REF_EMPTY_LEAF;
                            if (cur & RT_REF_LEAF) break;
                            if (__popcll(__ballot(1)) < n_keep) break;      /* wave-uniform */
                        }
                    }
                    RT_LAP_SPLIT(TM_DESCEND)
                    if (cur & RT_REF_LEAF) {
                        /* leaf: strict <, first triangle wins ties (:596) */
                        const int start = (int)(cur & RT_REF_START_MASK);
                        const int count = (int)((cur >> RT_REF_COUNT_SHIFT) & RT_REF_COUNT_MAX);
                        for (int k = 0; k < count; k++) {
                            RT_STAT(ST_LEAF_TRI);
                            RT_COST(p.c_steps++);
                            float t, u, v;
                            bool h = tri_test(L.tris, start + k, o, d, t, u, v);
                            if (h && t < w_best) { w_best = t; w_prim = start + k; }
                        }
                        RT_LAP_SPLIT_LEAF(TM_LEAF)
                        /* pop one entry: it is taken iff !(dist > best) (:501); through a
                         * collapsed chain iff dist < best (:517) - the distance is never NaN, so
                         * that is dist < best, or dist == best on a plain edge.  A lane whose entry
                         * is refused stays on the empty leaf and pops again next step (rare). */
                        if (sp > 0) {
                            RT_STAT(ST_POP);
                            sp--;
                            const uint2 e = stack[sp * NT + tid];
                            const float dd = __uint_as_float(e.x);
                            const bool take = dd < w_best || (dd == w_best && !(e.y & RT_REF_CHAIN));
                            cur = take ? e.y : RT_REF_EMPTY_LEAF;
                        } else {
                            RT_STAT(ST_DONE_MESH);
                            /* this mesh is done: merge (smaller distance, or equal and later in the list);
                             * its place in the object list is read again here rather than kept in a register */
                            const int w_obj = (int)__float_as_uint(L.meshes[2 * (p.next_mesh - 1) + 1].w);
                            if (w_prim >= 0 && (w_best < p.best_t || (w_best == p.best_t && w_obj > p.best_obj))) {
                                p.best_t = w_best; p.best_obj = w_obj; p.best_prim = w_prim;
                            }
                            p.mode = p.next_mesh >= a.num_meshes ? M_SHADE : M_MESH;
                        }
                    }
                }
                RT_LAP(TM_POP);
            }
        }

        if (__ballot(p.mode != M_DONE) == 0ull) break;
    }
    RT_STATS_FLUSH();
}

/* =============================================================================================
 * The pooled kernel (mesh scenes): pixels stay with their lanes, RAYS do not.
 *
 * Measured on the kernel above (tools/costmap.py, tools/wave_probe.py): a wave that owns 64
 * expensive pixels executes ~4x the traversal steps any one of its lanes needs (at every step
 * only the lanes in the same phase - box tests or triangle tests - take part: ~19 and ~10 of 64),
 * and a frame is as long as its slowest wave.  Here a lane that needs a mesh traversal writes
 * its ray to a record in LDS (slot = thread id) and posts the slot on a queue; ANY wave of the
 * workgroup that has nothing better to do takes up to 64 posted rays of ONE phase and steps them:
 *
 *   node queue -> box-test executor: descends / pops until a ray reaches a leaf (posted on the
 *                 leaf queue) or runs out of stack (finished: the owner lane is told);
 *   leaf queue -> triangle-test executor: tests the leaf's triangles, posts the ray back.
 *
 * So box tests run with (nearly) full waves of rays that all need a box test, likewise triangle
 * tests, the 16 waves of a workgroup share the rays of its most expensive pixels, and waves whose
 * own pixels are finished keep executing for the others until the workgroup is done.
 *
 * The traversal is still the reference's (src/objects.cu:487-532): same visit order, same push
 * order, same comparisons.  Two representation changes make a ray small enough to park
 * (48 bytes + 2 bytes per stack level): the stack holds only the index of the node whose second
 * child was deferred, and a pop re-derives that child's entry distance by running the parent's
 * two box tests again - the same operations on the same operands, hence the same bits (and the
 * same test the reference itself repeats when it pops, :499-501).
 *
 * Queue protocol (LDS, workgroup scope): head / tail counters are monotonic; a producer reserves
 * positions with one atomic add on tail, waits for each position to be EMPTY, then stores the
 * slot id with release semantics; a consumer reserves [head, head + n) with a compare-and-swap
 * bounded by tail, waits for each position to be filled, takes the id and stores EMPTY.  Nobody
 * waits on anything but another wave's few-instruction critical section.
 * ============================================================================================= */
#define RT_POOL_DONE 0xffffffffu        /* record.cur once the traversal is finished */
#define RT_POOL_EMPTY 0xffffffffu       /* a free queue position */
#define RT_POOL_QCAP 1024u              /* queue capacity (power of two, >= threads per workgroup) */
#define RT_META_POP 0x80000000u         /* record.meta: the ray must pop before it goes on */
#define RT_META_SP_SHIFT 24             /* ... bits 28..24 stack pointer, bits 23..0 best triangle + 1 */
#define RT_META_PRIM_MASK 0x00ffffffu
#define RT_POOL_SPIN_LIMIT (1 << 24)
enum { E_FREE = 0, E_NODE = 1, E_POP = 2, E_PARK_LEAF = 3, E_PARK_DONE = 4 };
/* control words */
enum { C_NODE_HEAD = 0, C_NODE_TAIL = 1, C_LEAF_HEAD = 2, C_LEAF_TAIL = 3, C_LIVE = 4, C_ABORT = 5, C_WORDS = 16 };

#define RT_LD(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define RT_ST(ptr, v) __hip_atomic_store((ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define RT_ST_REL(ptr, v) __hip_atomic_store((ptr), (v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP)
#define RT_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup")

__device__ __forceinline__ int rt_rank(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

/* posts `slot` for every lane with doit set (one reservation per wave) */
__device__ __forceinline__ void pool_push(uint32_t *tail, uint32_t *items, bool doit, uint32_t slot, uint32_t *abort_flag)
{
    if (doit) {
        const unsigned long long m = __ballot(1);
        const int rank = rt_rank(m);
        uint32_t base = 0;
        if (rank == 0) base = __hip_atomic_fetch_add(tail, (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        uint32_t *it = items + ((base + (uint32_t)rank) & (RT_POOL_QCAP - 1u));
        int spin = 0;
        while (RT_LD(it) != RT_POOL_EMPTY) {
            if (++spin > RT_POOL_SPIN_LIMIT) { RT_ST(abort_flag, 1u); break; }
        }
        RT_ST_REL(it, slot);
    }
}

/* the whole wave calls; lanes with `want` set receive a posted slot id, or -1 */
__device__ __forceinline__ int pool_grab(uint32_t *head_tail, uint32_t *items, bool want, int lane, uint32_t *abort_flag)
{
    const unsigned long long m = __ballot(want);
    const int n_want = __popcll(m);
    if (n_want == 0) return -1;
    uint32_t h = 0;
    int n = 0;
    if (lane == 0) {
        for (;;) {
            h = RT_LD(head_tail);
            const uint32_t t = RT_LD(head_tail + 1);
            const int avail = (int)(t - h);
            n = avail < n_want ? avail : n_want;
            if (n <= 0) { n = 0; break; }
            uint32_t expected = h;
            if (__hip_atomic_compare_exchange_strong(head_tail, &expected, h + (uint32_t)n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        }
    }
    h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
    n = __builtin_amdgcn_readfirstlane(n);
    int slot = -1;
    const int rank = rt_rank(m);
    if (want && rank < n) {
        uint32_t *it = items + ((h + (uint32_t)rank) & (RT_POOL_QCAP - 1u));
        uint32_t v;
        int spin = 0;
        while ((v = RT_LD(it)) == RT_POOL_EMPTY) {
            if (++spin > RT_POOL_SPIN_LIMIT) { RT_ST(abort_flag, 1u); v = 0u; break; }
        }
        RT_ST(it, RT_POOL_EMPTY);
        slot = (int)v;
    }
    RT_ACQUIRE();
    return slot;
}

template <int NT, bool SCENE_LDS>
__global__ __launch_bounds__(NT) void rt_render_pool_kernel(const rt_kernel_args a)
{
    extern __shared__ v4f lds_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & (RT_WAVE - 1);

    Lds L;
    v4f *dyn;
    if (SCENE_LDS) {
        for (int i = tid; i < a.blob_f4; i += NT) lds_raw[i] = ((const v4f *)a.blob)[i];
        L.nodes = lds_raw + a.off_nodes;
        L.tris = lds_raw + a.off_tris;
        L.objs = lds_raw + a.off_objlds;
        L.meshes = lds_raw + a.off_meshes;
        L.objtab = lds_raw + a.off_objtab;
        dyn = lds_raw + a.blob_f4;
    } else {
        const v4f *g = (const v4f *)a.blob;
        L.nodes = g + a.off_nodes;
        L.tris = g + a.off_tris;
        L.objs = g + a.off_objlds;
        L.meshes = g + a.off_meshes;
        L.objtab = g + a.off_objtab;
        dyn = lds_raw;
    }
    /* ray records, one per thread: q0 = (origin, best distance) q1 = (direction, current reference)
     * q2 = (1/direction, meta) */
    v4f *rq0 = dyn, *rq1 = dyn + NT, *rq2 = dyn + 2 * NT;
    uint32_t *node_q = (uint32_t *)(dyn + 3 * NT);
    uint32_t *leaf_q = node_q + RT_POOL_QCAP;
    uint32_t *ctl = leaf_q + RT_POOL_QCAP;
    uint16_t *stack = (uint16_t *)(ctl + C_WORDS);       /* [stack_entries + 1][NT] parent node of the deferred child */
    for (int i = tid; i < (int)RT_POOL_QCAP; i += NT) { node_q[i] = RT_POOL_EMPTY; leaf_q[i] = RT_POOL_EMPTY; }
    if (tid < C_WORDS) ctl[tid] = tid == C_LIVE ? (uint32_t)(NT / RT_WAVE) : 0u;
    __syncthreads();
#define REC_BEST(slot) (((float *)(rq0 + (slot))) + 3)
#define REC_CUR(slot) (((uint32_t *)(rq1 + (slot))) + 3)
#define REC_META(slot) (((uint32_t *)(rq2 + (slot))) + 3)

    Frame f;
    frame_init(f, a);
    Px p;
    px_init(p);
    int w_obj = -1;             /* object index of the mesh this lane's ray is in */
    bool retired = false;       /* wave-uniform: every pixel of this wave is finished */
    Chunk ch;
    ch.next = 0; ch.end = 0; ch.frame = 0; ch.exhausted = false;
#ifdef RT_STATS
    unsigned st_exec[ST_N], st_lanes[ST_N];
    for (int i = 0; i < ST_N; i++) { st_exec[i] = 0; st_lanes[i] = 0; }
    unsigned long long st_time[TM_N], st_last = __builtin_readcyclecounter();
    const unsigned long long st_wall0 = wall_clock64();
    for (int i = 0; i < TM_N; i++) st_time[i] = 0;
#endif
    unsigned long long idle_since = 0;

    for (;;) {
        RT_STAT(ST_ITER);
        /* ---- traversals of my lanes that have finished: merge (smaller distance, or equal and
         * later in the object list, src/raytracer.cu:36) */
        if (p.mode == M_WAIT && RT_LD(REC_CUR(tid)) == RT_POOL_DONE) {
            RT_ACQUIRE();
            const float w_best = *REC_BEST(tid);
            const int w_prim = (int)(*REC_META(tid) & RT_META_PRIM_MASK) - 1;
            if (w_prim >= 0 && (w_best < p.best_t || (w_best == p.best_t && w_obj > p.best_obj))) {
                p.best_t = w_best; p.best_obj = w_obj; p.best_prim = w_prim;
            }
            p.mode = p.next_mesh >= a.num_meshes ? M_SHADE : M_MESH;
        }
        const int n_live = __popcll(__ballot(p.mode != M_DONE));
        int n_ready = __popcll(__ballot(p.mode == M_SHADE || p.mode == M_MESH || p.mode == M_GEN || p.mode == M_FETCH));
        const int half_live = (n_live + 1) >> 1;
        const int thr = a.ready_break < half_live ? a.ready_break : (half_live > 0 ? half_live : 1);
        int n_node = __builtin_amdgcn_readfirstlane((int)(RT_LD(ctl + C_NODE_TAIL) - RT_LD(ctl + C_NODE_HEAD)));
        int n_leaf = __builtin_amdgcn_readfirstlane((int)(RT_LD(ctl + C_LEAF_TAIL) - RT_LD(ctl + C_LEAF_HEAD)));
        RT_LAP(TM_CTL);

        /* ---- the pixels' own work, once enough lanes want it (or there is nothing else to do) */
        if (n_ready >= thr || (n_ready > 0 && n_node + n_leaf <= 0)) {
            idle_since = 0;
            if (p.mode == M_SHADE && p.best_obj < 0) px_shade_miss(p, a, f);
            if (p.mode == M_SHADE) {
                RT_STAT(ST_SHADE);
                px_shade(p, a, f, L);
            }
            RT_LAP(TM_SHADE);
            px_fetch(p, ch, a, f, lane);
            RT_LAP(TM_FETCH);
            if (p.mode == M_GEN) {
                RT_STAT(ST_GEN);
                px_gen<true>(p, a, L);
            }
            RT_LAP(TM_GEN);
            /* MESH: the next mesh whose root box the ray enters; its traversal is posted */
            bool submit = false;
            uint32_t sub_root = 0;
            while (p.mode == M_MESH) {
                RT_STAT(ST_MESH);
                if (p.next_mesh >= a.num_meshes) { p.mode = M_SHADE; break; }
                const v4f m0 = L.meshes[2 * p.next_mesh], m1 = L.meshes[2 * p.next_mesh + 1];
                p.next_mesh++;
                /* a NaN direction (Box-Muller on a zero draw) fails every
                 * triangle test: the mesh cannot be hit, no need to walk it */
                if (p.d.x != p.d.x || p.d.y != p.d.y || p.d.z != p.d.z) continue;
                /* the root is pushed unconditionally and tested when popped (src/objects.cu:494-501) */
                const uint32_t root_ref = __float_as_uint(m1.z);
                float rd;
                const bool rh = box_test(m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, p.o, p.inv, rd);
                if (!rh || rd > RT_INF_F || ((root_ref & RT_REF_CHAIN) && !(rd < RT_INF_F))) continue;
                w_obj = (int)__float_as_uint(m1.w);
                submit = true; sub_root = root_ref;
                p.mode = M_WAIT;
                RT_STAT(ST_MESH_START);
            }
            if (submit) {
                v4f r0, r1, r2;
                r0.x = p.o.x; r0.y = p.o.y; r0.z = p.o.z; r0.w = RT_INF_F;
                r1.x = p.d.x; r1.y = p.d.y; r1.z = p.d.z; r1.w = __uint_as_float(sub_root);
                r2.x = p.inv.x; r2.y = p.inv.y; r2.z = p.inv.z; r2.w = __uint_as_float(0u);
                rq0[tid] = r0; rq1[tid] = r1; rq2[tid] = r2;
            }
            pool_push(ctl + C_NODE_TAIL, node_q, submit && !(sub_root & RT_REF_LEAF), (uint32_t)tid, ctl + C_ABORT);
            pool_push(ctl + C_LEAF_TAIL, leaf_q, submit && (sub_root & RT_REF_LEAF), (uint32_t)tid, ctl + C_ABORT);
            RT_LAP(TM_MESH);
            n_node = __builtin_amdgcn_readfirstlane((int)(RT_LD(ctl + C_NODE_TAIL) - RT_LD(ctl + C_NODE_HEAD)));
            n_leaf = __builtin_amdgcn_readfirstlane((int)(RT_LD(ctl + C_LEAF_TAIL) - RT_LD(ctl + C_LEAF_HEAD)));
        }

        if (n_leaf > 0 && (n_leaf >= n_node || n_leaf >= a.pool_leaf_batch)) {
            /* ================= triangle-test executor: one batch ============================ */
            idle_since = 0;
            const int e_slot = pool_grab(ctl + C_LEAF_HEAD, leaf_q, true, lane, ctl + C_ABORT);
            if (e_slot >= 0) {
                const v4f r0 = rq0[e_slot], r1 = rq1[e_slot];
                uint32_t e_meta = *REC_META(e_slot);
                const V3 e_o = v3(r0.x, r0.y, r0.z), e_d = v3(r1.x, r1.y, r1.z);
                float e_best = r0.w;
                const uint32_t e_cur = __float_as_uint(r1.w);
                /* leaf: strict <, first triangle wins ties (src/objects.cu:596) */
                const int start = (int)(e_cur & RT_REF_START_MASK);
                const int count = (int)((e_cur >> RT_REF_COUNT_SHIFT) & RT_REF_COUNT_MAX);
                uint32_t e_prim1 = e_meta & RT_META_PRIM_MASK;
                for (int k = 0; k < count; k++) {
                    RT_STAT(ST_LEAF_TRI);
                    float t, u, v;
                    const bool h = tri_test(L.tris, start + k, e_o, e_d, t, u, v);
                    if (h && t < e_best) { e_best = t; e_prim1 = (uint32_t)(start + k + 1); }
                }
                *REC_BEST(e_slot) = e_best;
                const uint32_t e_sp = (e_meta >> RT_META_SP_SHIFT) & 31u;
                const bool finished = e_sp == 0u;          /* nothing deferred: the traversal is over */
                *REC_META(e_slot) = e_prim1 | (e_sp << RT_META_SP_SHIFT) | (finished ? 0u : RT_META_POP);
                if (finished) RT_ST_REL(REC_CUR(e_slot), RT_POOL_DONE);
                pool_push(ctl + C_NODE_TAIL, node_q, !finished, (uint32_t)e_slot, ctl + C_ABORT);
            }
            RT_LAP(TM_LEAF);
        } else if (n_node > 0) {
            /* ================= box-test executor =========================================== */
            idle_since = 0;
            int e_state = E_FREE, e_slot = 0, e_sp = 0;
            V3 e_o = v3(0.f, 0.f, 0.f), e_inv = e_o;
            float e_best = RT_INF_F;
            uint32_t e_cur = 0, e_prim1 = 0;
            for (;;) {
                const int n_act = __popcll(__ballot(e_state == E_NODE || e_state == E_POP));
                const int n_park = __popcll(__ballot(e_state >= E_PARK_LEAF));
                if (n_act == 0 || RT_WAVE - n_act >= a.pool_fill || (n_park > 0 && n_act < a.pool_low)) {
                    /* ---- hand on the rays that left the box-test phase ... */
                    if (e_state == E_PARK_LEAF) {
                        *REC_CUR(e_slot) = e_cur;
                        *REC_META(e_slot) = e_prim1 | ((uint32_t)e_sp << RT_META_SP_SHIFT);
                    }
                    pool_push(ctl + C_LEAF_TAIL, leaf_q, e_state == E_PARK_LEAF, (uint32_t)e_slot, ctl + C_ABORT);
                    if (e_state == E_PARK_DONE) RT_ST_REL(REC_CUR(e_slot), RT_POOL_DONE);
                    if (e_state >= E_PARK_LEAF) e_state = E_FREE;
                    /* ---- ... see whether my own pixels want me back ... */
                    const bool mine = p.mode == M_WAIT && RT_LD(REC_CUR(tid)) == RT_POOL_DONE;
                    const int n_mine = __popcll(__ballot(mine)) + n_ready;
                    const bool preempt = n_mine >= thr;
                    /* ---- ... and take on waiting rays */
                    if (!preempt) {
                        const int s = pool_grab(ctl + C_NODE_HEAD, node_q, e_state == E_FREE, lane, ctl + C_ABORT);
                        if (s >= 0) {
                            const v4f r0 = rq0[s], r2 = rq2[s];
                            e_slot = s;
                            e_o = v3(r0.x, r0.y, r0.z); e_best = r0.w;
                            e_inv = v3(r2.x, r2.y, r2.z);
                            const uint32_t meta = __float_as_uint(r2.w);
                            e_cur = *REC_CUR(s);
                            e_prim1 = meta & RT_META_PRIM_MASK;
                            e_sp = (int)((meta >> RT_META_SP_SHIFT) & 31u);
                            e_state = (meta & RT_META_POP) ? E_POP : E_NODE;
                        }
                    }
                    const bool held = e_state == E_NODE || e_state == E_POP;
                    if (preempt || __ballot(held) == 0ull) {
                        /* leave: whatever is still in flight goes back on the queue */
                        if (held) {
                            *REC_CUR(e_slot) = e_cur;
                            *REC_META(e_slot) = e_prim1 | ((uint32_t)e_sp << RT_META_SP_SHIFT) | (e_state == E_POP ? RT_META_POP : 0u);
                        }
                        pool_push(ctl + C_NODE_TAIL, node_q, held, (uint32_t)e_slot, ctl + C_ABORT);
                        break;
                    }
                }
                if (e_state == E_NODE || e_state == E_POP) {
                    const bool popping = e_state == E_POP;
                    if (popping && e_sp == 0) {
                        e_state = E_PARK_DONE;          /* nothing deferred: the traversal is over */
                    } else {
                        RT_STAT(ST_NODE);
                        /* a pop re-runs the two box tests of the node whose second child was deferred */
                        uint32_t idx = e_cur & RT_REF_NODE_MASK;
                        if (popping) { e_sp--; idx = (uint32_t)stack[e_sp * NT + e_slot]; }
                        const v4f *n = L.nodes + 4 * (int)idx;
                        const v4f q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
                        float ld, rdist;
                        const bool lh = box_test(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, e_o, e_inv, ld);
                        const bool rh2 = box_test(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, e_o, e_inv, rdist);
                        const uint32_t lref = __float_as_uint(q3.x), rref = __float_as_uint(q3.y);
                        const bool l_first = ld < rdist;
                        /* Of two entered children the one pushed first (left when l_first) is
                         * visited second: it is the deferred sibling (src/objects.cu:503-531). */
                        const uint32_t deferred_ref = l_first ? lref : rref;
                        const float deferred_d = l_first ? ld : rdist;
                        if (popping) {
                            /* an entry is taken iff !(dist > best) (:501); through a collapsed
                             * chain iff dist < best (:517) */
                            const bool take = (deferred_ref & RT_REF_CHAIN) ? (deferred_d < e_best) : !(deferred_d > e_best);
                            if (take) {
                                e_cur = deferred_ref;
                                e_state = (deferred_ref & RT_REF_LEAF) ? E_PARK_LEAF : E_NODE;
                            }
                        } else {
                            const bool l_push = lh && ld < e_best;
                            const bool r_push = rh2 && rdist < e_best;
                            const bool both = l_push && r_push;
                            const bool entered = l_push || r_push;
                            stack[e_sp * NT + e_slot] = (uint16_t)idx;
                            e_sp += both ? 1 : 0;
                            const uint32_t next = both ? (l_first ? rref : lref) : (l_push ? lref : rref);
                            if (entered) {
                                e_cur = next;
                                if (next & RT_REF_LEAF) e_state = E_PARK_LEAF;
                            } else {
                                e_state = E_POP;
                            }
                        }
                    }
                }
            }
            RT_LAP(TM_DESCEND);
        } else {
            /* ---- nothing to execute */
            if (!retired && __ballot(p.mode != M_DONE) == 0ull) {
                retired = true;
                if (lane == 0) __hip_atomic_fetch_sub(ctl + C_LIVE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            const uint32_t live = (uint32_t)__builtin_amdgcn_readfirstlane((int)RT_LD(ctl + C_LIVE));
            const uint32_t aborted = (uint32_t)__builtin_amdgcn_readfirstlane((int)RT_LD(ctl + C_ABORT));
            if ((retired && live == 0u) || aborted) break;
            if (n_ready == 0) {
                /* watchdog: no wave of a healthy workgroup idles this long (100 MHz ticks) */
                const unsigned long long now = wall_clock64();
                if (idle_since == 0) idle_since = now;
                if (now - idle_since > 3000000000ull) { RT_ST(ctl + C_ABORT, 1u); }
                __builtin_amdgcn_s_sleep(2);
            }
            RT_LAP(TM_POP);
        }
    }
    if (lane == 0 && RT_LD(ctl + C_ABORT)) atomicAdd(a.tile_counter + 1, 1u);
    RT_STATS_FLUSH();
#undef REC_BEST
#undef REC_CUR
#undef REC_META
}

/* The sequential part of a multi-frame launch (src/raytracer.cu:109-112, once per frame): the image
 * after frame n is (c_n + image * n) / (n + 1), c_n = that frame's per-pixel mean (plane n - frame_num
 * of `partial`).  In place on `frame`; its content is used only when frame_num > 0. */
__global__ void rt_blend_kernel(const float *partial, long long plane_floats, int num_frames, int frame_num, float *frame, long long n_floats)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_floats) return;
    float r = frame_num > 0 ? frame[i] : 0.0f;
    for (int k = 0; k < num_frames; k++) {
        const int n = frame_num + k;
        const float previous_sum = r * (float)n;
        r = (partial[(long long)k * plane_floats + i] + previous_sum) / (float)(n + 1);
    }
    frame[i] = rt_canon_nan(r);
}

extern "C" hipError_t rt_launch_blend(const float *partial, long long plane_floats, int num_frames, int frame_num, float *frame, long long n_floats, hipStream_t stream)
{
    const long long blocks = (n_floats + 255) / 256;
    hipLaunchKernelGGL(rt_blend_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, partial, plane_floats, num_frames, frame_num, frame, n_floats);
    return hipGetLastError();
}

/* float -> RGBA8 of src/main.cu:343-371 */
__global__ void rt_rgba8_kernel(const float *rgb, int n_pixels, uint8_t *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    uint32_t packed = 0xff000000u;
    for (int c = 0; c < 3; c++) {
        int colour = rt_f2i(rgb[3 * i + c] * 255.0f);
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
static int rt_blocks_one(size_t lds_bytes)
{
    int n = 0;
    (void)hipFuncSetAttribute((const void *)rt_render_kernel<NT, HAS_MESH, SCENE_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)rt_render_kernel<NT, HAS_MESH, SCENE_LDS>, NT, lds_bytes) != hipSuccess) n = 0;
    return n;
}

/* workgroups of this shape of rt_render_kernel that are resident on one CU (registers, LDS, wave slots), as the
 * runtime reports it; 0 if the shape is not built */
extern "C" int rt_kernel_blocks_per_cu(int has_mesh, int scene_in_lds, int threads, size_t lds_bytes)
{
    if (!scene_in_lds) {
        if (has_mesh && threads == 1024) return rt_blocks_one<1024, true, false>(lds_bytes);
        if (!has_mesh && threads == 256) return rt_blocks_one<256, false, false>(lds_bytes);
        return 0;
    }
    switch (threads) {
        case 256: return has_mesh ? rt_blocks_one<256, true, true>(lds_bytes) : rt_blocks_one<256, false, true>(lds_bytes);
        case 512: return has_mesh ? rt_blocks_one<512, true, true>(lds_bytes) : rt_blocks_one<512, false, true>(lds_bytes);
        case 768: return has_mesh ? rt_blocks_one<768, true, true>(lds_bytes) : rt_blocks_one<768, false, true>(lds_bytes);
        case 1024: return has_mesh ? rt_blocks_one<1024, true, true>(lds_bytes) : rt_blocks_one<1024, false, true>(lds_bytes);
        default: return 0;
    }
}

template <int NT, bool HAS_MESH, bool SCENE_LDS>
static void rt_launch_one(const rt_kernel_args *args, int blocks, size_t lds_bytes, hipStream_t stream)
{
    (void)hipFuncSetAttribute((const void *)rt_render_kernel<NT, HAS_MESH, SCENE_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL((rt_render_kernel<NT, HAS_MESH, SCENE_LDS>), dim3(blocks), dim3(NT), lds_bytes, stream, *args);
}

template <int NT, bool SCENE_LDS>
static void rt_launch_pool(const rt_kernel_args *args, int blocks, size_t lds_bytes, hipStream_t stream)
{
    (void)hipFuncSetAttribute((const void *)rt_render_pool_kernel<NT, SCENE_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL((rt_render_pool_kernel<NT, SCENE_LDS>), dim3(blocks), dim3(NT), lds_bytes, stream, *args);
}

/* mesh scenes, pooled traversal */
extern "C" hipError_t rt_launch_render_pool(const rt_kernel_args *args, int scene_in_lds, int threads, int blocks, size_t lds_bytes, hipStream_t stream)
{
    if (!scene_in_lds) {
        if (threads != 1024) return hipErrorInvalidValue;
        rt_launch_pool<1024, false>(args, blocks, lds_bytes, stream);
        return hipGetLastError();
    }
    switch (threads) {
        case 256: rt_launch_pool<256, true>(args, blocks, lds_bytes, stream); break;
        case 512: rt_launch_pool<512, true>(args, blocks, lds_bytes, stream); break;
        case 768: rt_launch_pool<768, true>(args, blocks, lds_bytes, stream); break;
        case 1024: rt_launch_pool<1024, true>(args, blocks, lds_bytes, stream); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
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
