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
 * file has the render kernel built from them, the small streaming kernels and the launchers.
 *
 * No MFMA: there is no dense contraction anywhere in this workload.
 * Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize (the reference's a*b+c
 * are two roundings: contraction would change hit/miss decisions; the SLP vectorizer's packed f32
 * pairs cost 40 % more registers and 10 % of the time).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_pixel.h"

/* LLVM integer-compare predicates for __builtin_amdgcn_uicmp / sicmp (lane mask of a compare, straight into an SGPR pair) */
#define RT_ICMP_EQ 32
#define RT_ICMP_NE 33
#define RT_ICMP_SGE 39

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
#elif defined(RT_MARK)
/* (tools/isa_sections.py: section boundaries as comments in the assembly) */
#define RT_LAP(slot) asm volatile("; LAP " #slot)
#define RT_LAP_SPLIT(slot) asm volatile("; LAP " #slot);
#define RT_LAP_SPLIT_LEAF(slot) asm volatile("; LAP " #slot);
#else
#define RT_LAP(slot) do { } while (0)
#define RT_LAP_SPLIT(slot)
#define RT_LAP_SPLIT_LEAF(slot)
#endif
enum { ST_ITER = 0, ST_SHADE = 1, ST_SHADE_HIT = 2, ST_FETCH = 3, ST_GEN = 4, ST_MESH = 5, ST_MESH_START = 6, ST_WORK_ITER = 7, ST_NODE = 8, ST_LEAF_TRI = 9, ST_POP = 10, ST_DONE_MESH = 11, ST_N = 12 };

#if defined(RT_STATS)
#define RT_STAT_PARAMS , unsigned *st_exec, unsigned *st_lanes, int lane
#define RT_STAT_ARGS , st_exec, st_lanes, lane
#elif defined(RT_COSTMAP)
#define RT_STAT_PARAMS , Px &p
#define RT_STAT_ARGS , p
#else
#define RT_STAT_PARAMS
#define RT_STAT_ARGS
#endif
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

/* The descend loop of a traversal macro step: from an internal node down to a leaf (or to "no child entered").
 * The body is branch-free: the deferred sibling is ALWAYS written to the slot above the top of the stack (one 8-byte
 * LDS store) and the stack pointer moves only when both children are entered, so the only divergent branch of the
 * loop is its exit.  The loop also ends, for everybody, once fewer than `descend_keep`/64 of the lanes that entered
 * it are still descending: those lanes just stay on their internal node and go on next step, instead of making the
 * others wait out the deepest descent of the wave.  MED3: box_enter_med3 (rays without a zero direction component). */
template <int NT, bool MED3>
__device__ __forceinline__ void rt_descend(uint32_t &cur, int &sp, uint2 *my_stack, const Lds &L, V3 o, V3 inv, float w_best, int descend_keep RT_STAT_PARAMS)
{
    const int n_enter = __popcll(__ballot(1));
    const int n_keep = (n_enter * descend_keep) >> 6;
    for (;;) {
        RT_STAT(ST_NODE);
        RT_COST(p.c_steps++);
        const v4f *n = L.nodes + 4 * (int)(cur & RT_REF_NODE_MASK);
        const v4f q0 = n[0], q1 = n[1], q2 = n[2];
        const uint2 refs = *(const uint2 *)(n + 3);          /* the two child references: 8 of the last 16 bytes */
        float ld, rdist;
        const bool l_push = MED3 ? box_enter_med3(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, inv, w_best, ld)
                                 : box_enter(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, inv, w_best, ld);
        const bool r_push = MED3 ? box_enter_med3(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, inv, w_best, rdist)
                                 : box_enter(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, inv, w_best, rdist);
        const uint32_t lref = refs.x, rref = refs.y;
        const bool l_first = ld < rdist;
        /* Of two entered children the one pushed first (left when l_first) is visited second: it is the deferred
         * sibling.  The other is popped immediately (its distance is still < best).  With one entered child that
         * child is next and nothing is deferred. */
        const bool both = l_push && r_push;
        const bool entered = l_push || r_push;
        const unsigned long long l_first_lanes = __builtin_amdgcn_fcmpf(ld, rdist, RT_FCMP_OLT);
        const uint32_t deferred_ref = rt_sel_u32(l_first_lanes, lref, rref);
        const float deferred_d = rt_sel_f32(l_first_lanes, ld, rdist);
        my_stack[sp * NT] = make_uint2(__float_as_uint(deferred_d), deferred_ref);
        sp += both ? 1 : 0;
        const uint32_t next = both ? (l_first ? rref : lref) : (l_push ? lref : rref);
        cur = entered ? next : RT_REF_EMPTY_LEAF;
        {
            /* One divergent exit: the lanes that leave - those that reached a leaf, or everybody once fewer than n_keep are
             * still on an internal node - are computed as a lane mask in five instructions (one compare, four scalar) and handed
             * to the compiler as the loop's exit condition (inverse ballot: no instruction).  The compiler's own rendering of
             * "leaf || count < n_keep" took 16 scalar instructions and 3 branches per node step (round 3), then, with the count
             * passed through a VGPR, 9 + 3 vector ones (-2 %, profiles/r04/experiments/keep_rule_single_exit.txt). */
            unsigned long long stop, internal;
            int cnt;
            asm volatile("v_cmp_gt_i32_e64 %0, 0, %3\n\t"
                         "s_andn2_b64 %1, exec, %0\n\t"
                         "s_bcnt1_i32_b64 %2, %1\n\t"
                         "s_cmp_lt_u32 %2, %4\n\t"
                         "s_cselect_b64 %0, exec, %0"
                         : "=&s"(stop), "=&s"(internal), "=&s"(cnt) : "v"(cur), "s"(n_keep) : "scc");
            if (__builtin_amdgcn_inverse_ballot_w64(stop)) break;
        }
    }
}

/* MODE (RT_SCENE_*): where the scene is read from.  RT_SCENE_LDS: the whole blob is staged into LDS.  For scenes larger
 * than a CU's LDS the same code reads the triangles (RT_SCENE_HYBRID: the BVH nodes and the object records still fit) or
 * every section (RT_SCENE_GLOBAL) from global memory - they stay L2 / Infinity-Cache resident.  (Requesting a leaf's next
 * triangle before testing the current one, and testing two at a time, were measured on the 6,000- and 50,880-triangle
 * scenes: no difference - the compiler keeps the tests sequential and four waves per SIMD already cover the L2 latency.) */
/* Occupancy.  Built with -fno-slp-vectorize the kernel needs ~90 VGPRs (the SLP vectorizer's packed-f32 pairs cost
 * 128 and ~10 % of the time).  The 1024-thread workgroup of a large mesh scene is one per CU = four waves per SIMD,
 * whatever the registers (its LDS holds the scene and 1024 traversal stacks); the smaller workgroups - scenes without a
 * mesh, and mesh scenes small enough for several workgroups per CU - are register-bound, so they are compiled for
 * five waves per SIMD (<= 96 VGPRs; the allocator then lands on 79-80, which lets six be resident).  The launcher
 * asks the runtime how many workgroups of the chosen shape fit a CU (rt_kernel_blocks_per_cu). */
template <int NT, bool HAS_MESH, int MODE>
__global__ __launch_bounds__(NT, NT == 1024 ? 4 : (HAS_MESH ? RT_SMALL_WG_WAVES : RT_SMALL_WG_WAVES + 1)) void rt_render_kernel(const rt_kernel_args a)
{
    extern __shared__ v4f lds_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & (RT_WAVE - 1);

    Lds L;
    uint2 *stack;        /* [stack_entries + 1][NT] deferred sibling: (entry distance bits, reference) */
    if (MODE != RT_SCENE_GLOBAL) {
        /* stage the scene (or its part before the triangles) into LDS: coalesced 16-byte loads, one pass per workgroup */
        const int staged = MODE == RT_SCENE_LDS ? a.blob_f4 : a.off_tris;
        for (int i = tid; i < staged; i += NT) lds_raw[i] = ((const v4f *)a.blob)[i];
        L.nodes = lds_raw + a.off_nodes;
        L.objs = lds_raw + a.off_objlds;
        L.meshes = lds_raw + a.off_meshes;
        L.objtab = lds_raw + a.off_objtab;
        L.tris = MODE == RT_SCENE_LDS ? lds_raw + a.off_tris : (const v4f *)a.blob + a.off_tris;
        stack = (uint2 *)(lds_raw + staged);
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
    uint2 *const my_stack = stack + tid;        /* this lane's column of the [entry][thread] stack */

    Frame f;
    frame_init(f, a);
    Px p;
    px_init(p);
    /* ---- per-lane traversal state (registers + LDS stack); a lane is traversing iff M_WAIT ---- */
    uint32_t cur = 0;
    int sp = 0, w_prim = -1;
    float w_best = RT_INF_F;
    uint32_t w_zero_dir = 0u;    /* this traversal's ray has a direction component of exactly zero (box_enter_med3); an integer: its lane mask is then one compare */
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
            const int n_hit = __popcll(__builtin_amdgcn_uicmp((unsigned)p.mode, (unsigned)M_SHADE, RT_ICMP_EQ));
            const bool others = (__builtin_amdgcn_uicmp((unsigned)p.mode, (unsigned)M_GEN, RT_ICMP_EQ) |
                                 (ch.exhausted ? 0ull : __builtin_amdgcn_uicmp((unsigned)p.mode, (unsigned)M_FETCH, RT_ICMP_EQ))) != 0ull;
            const int n_trav = __popcll(__builtin_amdgcn_uicmp((unsigned)p.mode, (unsigned)M_WAIT, RT_ICMP_EQ));
            if (n_hit > 0 && (HAS_MESH ? (n_hit >= a.hit_low || n_trav < a.work_threshold)
                                        : (n_hit >= a.shade_batch || !others))) {
                if (p.mode == M_SHADE) {
                    RT_STAT(ST_SHADE);
                    px_shade<!(NT == 1024 && HAS_MESH), MODE == RT_SCENE_HYBRID>(p, a, f, L);
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
                w_zero_dir = (p.d.x == 0.0f || p.d.y == 0.0f || p.d.z == 0.0f) ? 1u : 0u;
                p.mode = M_WAIT;
                p.frame_steps |= 0x80000000u;           /* (cost bookkeeping: this pixel traverses) */
                RT_STAT(ST_MESH_START);
            }

            RT_LAP(TM_MESH);
            /* ================= WORK: BVH traversal steps (src/objects.cu:487-532, :586-600) ====
             * Runs while enough lanes are traversing; lanes whose ray is finished go back to
             * shading as soon as the traversing group is small.  Visit order, push order and
             * every comparison are the reference's. */
            const V3 o = p.o, d = p.d, inv = p.inv;
            for (;;) {
                /* the three lane counts from four compare masks combined in scalar registers (a __ballot of a bool built from
                 * several compares is rebuilt through a v_cndmask and a v_cmp) */
                const unsigned long long m_wait = __builtin_amdgcn_uicmp((unsigned)p.mode, (unsigned)M_WAIT, RT_ICMP_EQ);
                const int n_active = __popcll(m_wait);
                if (n_active == 0) break;
                /* lanes holding a hit wait for a batch of `hit_break`; the cheap kinds of ready
                 * lane (generate, fetch, next mesh, a miss) for one of `ready_break` */
                const unsigned long long m_hit = __builtin_amdgcn_uicmp((unsigned)p.mode, (unsigned)M_SHADE, RT_ICMP_EQ) & __builtin_amdgcn_sicmp(p.best_obj, 0, RT_ICMP_SGE);
                const unsigned long long m_done = __builtin_amdgcn_uicmp((unsigned)p.mode, (unsigned)M_DONE, RT_ICMP_EQ);
                const int n_hit = __popcll(m_hit);
                const int n_light = __popcll(__ballot(1) & ~(m_wait | m_done | m_hit));
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
                    p.frame_steps += (unsigned)(RT_COST_STEP * RT_MAX_BATCH_FRAMES);   /* one more traversal macro step (the bits above the frame index) */
                    /* one macro step: descend to a leaf (or run out of children), test the
                     * leaf's triangles, pop the next deferred sibling.  The lane's whole
                     * traversal state is `cur` (+ the stack): an internal node to descend from,
                     * or a leaf whose triangles are tested and after which the stack is popped;
                     * "no child entered" is the empty leaf. */
                    if (!(cur & RT_REF_LEAF)) {
                        /* two copies of the loop: the six-med3 slab test where no traversing ray of the wave has a direction
                         * component of exactly zero (always, in practice), the reference's min / max form otherwise */
                        if (__builtin_amdgcn_uicmp(w_zero_dir, 0u, RT_ICMP_NE) == 0ull) rt_descend<NT, true>(cur, sp, my_stack, L, o, inv, w_best, a.descend_keep RT_STAT_ARGS);
                        else rt_descend<NT, false>(cur, sp, my_stack, L, o, inv, w_best, a.descend_keep RT_STAT_ARGS);
                    }
                    RT_LAP_SPLIT(TM_DESCEND)
                    if (cur & RT_REF_LEAF) {
                        /* leaf: strict <, first triangle wins ties (:596) */
                        const int start = (int)(cur & RT_REF_START_MASK);
                        const int count = (int)((cur >> RT_REF_COUNT_SHIFT) & RT_REF_COUNT_MAX);
                        for (int k = 0; k < count; k++) {
                            RT_STAT(ST_LEAF_TRI);
                            RT_COST(p.c_steps++);
                            float t;
                            const unsigned long long closer = tri_closer_lanes(L.tris, start + k, o, d, w_best, t);
                            w_best = rt_sel_f32(closer, t, w_best);
                            w_prim = (int)rt_sel_u32(closer, (uint32_t)(start + k), (uint32_t)w_prim);
                        }
                        RT_LAP_SPLIT_LEAF(TM_LEAF)
                        /* pop one entry: it is taken iff !(dist > best) (:501); through a
                         * collapsed chain iff dist < best (:517) - the distance is never NaN, so
                         * that is dist < best, or dist == best on a plain edge.  A lane whose entry
                         * is refused stays on the empty leaf and pops again next step (rare). */
                        if (sp > 0) {
                            RT_STAT(ST_POP);
                            sp--;
                            const uint2 e = my_stack[sp * NT];
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

/* float i of a tile list's compact image (tile k = i / 192, 64 pixels of 3 floats, row-major inside the tile) -> its
 * index in the full W x H frame, or -1 for the part of a ragged edge tile that lies outside the image */
__device__ __forceinline__ long long rt_tile_float_index(long long i, const uint32_t *tile_list, int tiles_x, int W, int H)
{
    const long long k = i / 192;
    const int r = (int)(i - k * 192), within = r / 3, c = r - within * 3;
    const int g = (int)tile_list[k];
    const int ty = g / tiles_x, tx = g - ty * tiles_x;
    const int x = tx * 8 + (within & 7), y = ty * 8 + (within >> 3);
    if (x >= W || y >= H) return -1;
    return ((long long)y * W + x) * 3 + c;
}

/* the same fold for a launch that rendered a LIST of tiles into a full-layout frame: only the listed tiles' pixels are
 * touched (planes and frame are both full W x H frames) */
__global__ void rt_blend_tiles_kernel(const float *partial, long long plane_floats, int num_frames, int frame_num, float *frame,
                                      const uint32_t *tile_list, long long n_floats, int tiles_x, int W, int H)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_floats) return;
    const long long j = rt_tile_float_index(i, tile_list, tiles_x, W, H);
    if (j < 0) return;
    float r = frame_num > 0 ? frame[j] : 0.0f;
    for (int k = 0; k < num_frames; k++) {
        const int n = frame_num + k;
        const float previous_sum = r * (float)n;
        r = (partial[(long long)k * plane_floats + j] + previous_sum) / (float)(n + 1);
    }
    frame[j] = rt_canon_nan(r);
}

extern "C" hipError_t rt_launch_blend_tiles(const float *partial, long long plane_floats, int num_frames, int frame_num, float *frame,
                                            const uint32_t *tile_list, int n_tiles, int tiles_x, int W, int H, hipStream_t stream)
{
    const long long n_floats = (long long)n_tiles * 192;
    hipLaunchKernelGGL(rt_blend_tiles_kernel, dim3((unsigned)((n_floats + 255) / 256)), dim3(256), 0, stream, partial, plane_floats, num_frames, frame_num,
                       frame, tile_list, n_floats, tiles_x, W, H);
    return hipGetLastError();
}

/* The exchange step of the tile-list partition: a rank's compact image (its tiles back to back)
 * <-> the full frame.  to_frame: frame[tile pixels] = compact; otherwise compact = frame[tile pixels].  Streaming:
 * 12 B read + 12 B written per pixel; the compact side is contiguous, the frame side comes in 96-byte runs. */
__global__ void rt_tiles_copy_kernel(float *compact, float *frame, const uint32_t *tile_list, long long n_floats, int tiles_x, int W, int H, int to_frame)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_floats) return;
    const long long j = rt_tile_float_index(i, tile_list, tiles_x, W, H);
    if (j < 0) return;
    if (to_frame) frame[j] = compact[i];
    else compact[i] = frame[j];
}

extern "C" hipError_t rt_launch_tiles_copy(float *compact, float *frame, const uint32_t *tile_list, int n_tiles, int tiles_x, int W, int H, int to_frame, hipStream_t stream)
{
    const long long n_floats = (long long)n_tiles * 192;
    if (n_floats == 0) return hipSuccess;
    hipLaunchKernelGGL(rt_tiles_copy_kernel, dim3((unsigned)((n_floats + 255) / 256)), dim3(256), 0, stream, compact, frame, tile_list, n_floats, tiles_x, W, H, to_frame);
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
        case 11: r = rt_rcp_in_range(x) ? rt_rcp_short(x) : 1.0f / x; break;       /* per lane what rt_rcp / rt_sqrt do per wave */
        case 12: r = rt_sqrt_in_range(x) ? rt_sqrt_short(x) : sqrtf(x); break;
        case 13: r = rt_logf_0_1(rt_u01(u), 1); break;                               /* the Box-Muller calls on a hash output (13: the short divide, 15: the operator) */
        case 14: r = rt_cosf_0_2pi(rt_theta(u)); break;
        case 15: r = rt_logf_0_1(rt_u01(u), 0); break;
        default: break;
    }
    out[i] = __float_as_uint(r);
}

/* rt_rcp_short / rt_sqrt_short against the compiler's IEEE expansions for ALL 2^32 inputs (tests/test_gpu_math.py): out[0] = inputs in
 * rt_rcp's range whose short form differs from 1.0f / x, out[1] = inputs in the range, out[2], out[3] the same for the square root */
__global__ void rt_exhaustive_kernel(unsigned long long *out)
{
    const uint32_t lo = blockIdx.x * 1024u + threadIdx.x;
    unsigned bad_r = 0, in_r = 0, bad_s = 0, in_s = 0;
    for (uint32_t hi = 0; hi < 16; hi++) {
        const float x = __uint_as_float(lo | (hi << 28));
        if (rt_rcp_in_range(x)) { in_r++; bad_r += __float_as_uint(rt_rcp_short(x)) != __float_as_uint(1.0f / x); }
        if (rt_sqrt_in_range(x)) { in_s++; bad_s += __float_as_uint(rt_sqrt_short(x)) != __float_as_uint(sqrtf(x)); }
    }
    if (bad_r) atomicAdd(&out[0], (unsigned long long)bad_r);
    atomicAdd(&out[1], (unsigned long long)in_r);
    if (bad_s) atomicAdd(&out[2], (unsigned long long)bad_s);
    atomicAdd(&out[3], (unsigned long long)in_s);
}

extern "C" hipError_t rt_launch_exhaustive(unsigned long long *out4, hipStream_t stream)
{
    hipLaunchKernelGGL(rt_exhaustive_kernel, dim3(1u << 18), dim3(1024), 0, stream, out4);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_eval(int op, const uint32_t *in, uint32_t *out, int n, hipStream_t stream)
{
    hipLaunchKernelGGL(rt_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, op, in, out, n);
    return hipGetLastError();
}

/* ---- launchers (called from rt_capi.cpp) -------------------------------------------------- */
template <int NT, bool HAS_MESH, int MODE>
static int rt_blocks_one(size_t lds_bytes)
{
    int n = 0;
    (void)hipFuncSetAttribute((const void *)rt_render_kernel<NT, HAS_MESH, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)rt_render_kernel<NT, HAS_MESH, MODE>, NT, lds_bytes) != hipSuccess) n = 0;
    return n;
}

/* workgroups of this shape of rt_render_kernel that are resident on one CU (registers, LDS, wave slots), as the
 * runtime reports it; 0 if the shape is not built */
extern "C" int rt_kernel_blocks_per_cu(int has_mesh, int scene_in_lds, int threads, size_t lds_bytes)
{
    if (scene_in_lds == RT_SCENE_GLOBAL) {
        if (has_mesh && threads == 1024) return rt_blocks_one<1024, true, RT_SCENE_GLOBAL>(lds_bytes);
        if (!has_mesh && threads == 256) return rt_blocks_one<256, false, RT_SCENE_GLOBAL>(lds_bytes);
        return 0;
    }
    if (scene_in_lds == RT_SCENE_HYBRID) {
        if (!has_mesh) return 0;
        switch (threads) {
            case 512: return rt_blocks_one<512, true, RT_SCENE_HYBRID>(lds_bytes);
            case 768: return rt_blocks_one<768, true, RT_SCENE_HYBRID>(lds_bytes);
            case 1024: return rt_blocks_one<1024, true, RT_SCENE_HYBRID>(lds_bytes);
            default: return 0;
        }
    }
    switch (threads) {
        case 256: return has_mesh ? rt_blocks_one<256, true, RT_SCENE_LDS>(lds_bytes) : rt_blocks_one<256, false, RT_SCENE_LDS>(lds_bytes);
        case 512: return has_mesh ? rt_blocks_one<512, true, RT_SCENE_LDS>(lds_bytes) : rt_blocks_one<512, false, RT_SCENE_LDS>(lds_bytes);
        case 768: return has_mesh ? rt_blocks_one<768, true, RT_SCENE_LDS>(lds_bytes) : rt_blocks_one<768, false, RT_SCENE_LDS>(lds_bytes);
        case 1024: return has_mesh ? rt_blocks_one<1024, true, RT_SCENE_LDS>(lds_bytes) : rt_blocks_one<1024, false, RT_SCENE_LDS>(lds_bytes);
        default: return 0;
    }
}

template <int NT, bool HAS_MESH, int MODE>
static void rt_launch_one(const rt_kernel_args *args, int blocks, size_t lds_bytes, hipStream_t stream)
{
    (void)hipFuncSetAttribute((const void *)rt_render_kernel<NT, HAS_MESH, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL((rt_render_kernel<NT, HAS_MESH, MODE>), dim3(blocks), dim3(NT), lds_bytes, stream, *args);
}

extern "C" hipError_t rt_launch_render(const rt_kernel_args *args, int has_mesh, int scene_in_lds, int threads, int blocks, size_t lds_bytes, hipStream_t stream)
{
    if (scene_in_lds == RT_SCENE_GLOBAL) {
        /* global-memory scene: one shape per mesh flag is enough */
        if (has_mesh && threads == 1024) rt_launch_one<1024, true, RT_SCENE_GLOBAL>(args, blocks, lds_bytes, stream);
        else if (!has_mesh && threads == 256) rt_launch_one<256, false, RT_SCENE_GLOBAL>(args, blocks, lds_bytes, stream);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (scene_in_lds == RT_SCENE_HYBRID) {
        if (!has_mesh) return hipErrorInvalidValue;
        switch (threads) {
            case 512: rt_launch_one<512, true, RT_SCENE_HYBRID>(args, blocks, lds_bytes, stream); break;
            case 768: rt_launch_one<768, true, RT_SCENE_HYBRID>(args, blocks, lds_bytes, stream); break;
            case 1024: rt_launch_one<1024, true, RT_SCENE_HYBRID>(args, blocks, lds_bytes, stream); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
#define RT_CASE(NTV)                                                                                   \
    case NTV:                                                                                          \
        if (has_mesh) rt_launch_one<NTV, true, RT_SCENE_LDS>(args, blocks, lds_bytes, stream);         \
        else rt_launch_one<NTV, false, RT_SCENE_LDS>(args, blocks, lds_bytes, stream);                 \
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
