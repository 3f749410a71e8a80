// The GPU-free entry points of rt_capi.cpp under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only; test infrastructure):
// rt_partition_tiles (longest-processing-time-first ownership), rt_tile_owned_rows, and every entry point's refusal of null / bad
// arguments before it touches HIP.  rt_capi.cpp is compiled as host C++ against the HIP runtime's API header and linked with the
// runtime library; the kernel launchers (rt_kernel.hip) are replaced by stubs that fail - nothing here reaches a launch.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "rt_amd.h"
#include "rt_device_scene.h"

extern "C" hipError_t rt_launch_render(const rt_kernel_args *, int, int, int, int, size_t, hipStream_t) { return hipErrorUnknown; }
extern "C" int rt_kernel_blocks_per_cu(int, int, int, size_t) { return 1; }
extern "C" hipError_t rt_launch_blend(const float *, long long, int, int, float *, long long, hipStream_t) { return hipErrorUnknown; }
extern "C" hipError_t rt_launch_blend_tiles(const float *, long long, int, int, float *, const uint32_t *, int, int, int, int, hipStream_t) { return hipErrorUnknown; }
extern "C" hipError_t rt_launch_tiles_copy(float *, float *, const uint32_t *, int, int, int, int, int, hipStream_t) { return hipErrorUnknown; }
extern "C" hipError_t rt_launch_eval(int, const uint32_t *, uint32_t *, int, hipStream_t) { return hipErrorUnknown; }
extern "C" hipError_t rt_launch_rgba8(const float *, int, uint8_t *, hipStream_t) { return hipErrorUnknown; }
extern "C" hipError_t rt_launch_exhaustive(unsigned long long *, hipStream_t) { return hipErrorUnknown; }

int main(int argc, char **argv)
{
    std::mt19937_64 rng((unsigned long long)(argc > 1 ? std::atoll(argv[1]) : 1));
    const int cases = argc > 2 ? std::atoi(argv[2]) : 1000;
    auto irand = [&](long long lo, long long hi) { return (long long)std::uniform_int_distribution<long long>(lo, hi)(rng); };
    for (int c = 0; c < cases; c++) {
        const int tx = (int)irand(-1, 40), ty = (int)irand(-1, 30), n = (int)irand(-1, 9);
        const long long tiles = (long long)(tx > 0 ? tx : 0) * (ty > 0 ? ty : 0);
        std::vector<uint32_t> cost((size_t)tiles + 1);
        const int kind = (int)irand(0, 3);
        for (auto &x : cost) x = kind == 0 ? 0u : kind == 1 ? (uint32_t)irand(0, 5) : kind == 2 ? (uint32_t)irand(0, 0xffffffffll) : 0xffffffffu;
        std::vector<int32_t> owner((size_t)tiles + 1, -7);
        const rt_status st = rt_partition_tiles(irand(0, 3) ? cost.data() : nullptr, tx, ty, n, owner.data());
        if (st == RT_OK) {
            std::vector<unsigned long long> load((size_t)n, 0ull);
            for (long long i = 0; i < tiles; i++) {
                if (owner[(size_t)i] < 0 || owner[(size_t)i] >= n) { std::fprintf(stderr, "owner out of range\n"); return 1; }
                load[(size_t)owner[(size_t)i]] += cost[(size_t)i];
            }
            if (owner[(size_t)tiles] != -7) { std::fprintf(stderr, "wrote past the end\n"); return 1; }
        }
        rt_tile_spec ts;
        std::memset(&ts, 0, sizeof ts);
        ts.band_rows = (int32_t)irand(-8, 64); ts.band_first = (int32_t)irand(-1, 5); ts.band_stride = (int32_t)irand(-1, 5);
        (void)rt_tile_owned_rows(&ts, (int32_t)irand(-5, 5000));
        (void)rt_tile_owned_rows(nullptr, 10);
    }
    /* null / bad arguments: refused before HIP is touched (there is no GPU here; rt_ctx_create must say so, not crash) */
    rt_ctx *ctx = nullptr;
    const rt_status cs = rt_ctx_create(0, &ctx);
    if (cs == RT_OK) { rt_ctx_destroy(ctx); std::printf("(a GPU is present: context created and destroyed)\n"); }
    (void)rt_ctx_create(0, nullptr);
    (void)rt_ctx_create(-3, &ctx);
    rt_ctx_destroy(nullptr);
    rt_scene_destroy(nullptr);
    (void)rt_last_error(nullptr);
    int32_t fn = 0;
    (void)rt_render(nullptr, nullptr, nullptr, nullptr, 0, &fn, nullptr);
    (void)rt_render_frames(nullptr, nullptr, nullptr, nullptr, nullptr, 0, &fn, nullptr);
    (void)rt_render_device(nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr);
    (void)rt_render_device_batch(nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr);
    (void)rt_frame_submit(nullptr, nullptr, nullptr, nullptr, 0, nullptr);
    (void)rt_frame_collect(nullptr, 0, nullptr, nullptr);
    (void)rt_frame_collect_host(nullptr, &fn, nullptr);
    (void)rt_frame_wait(nullptr);
    (void)rt_frame_depth(nullptr, 3);
    (void)rt_frames_pending(nullptr);
    (void)rt_tile_costs(nullptr, nullptr, nullptr, nullptr, 0, &fn);
    (void)rt_tiles_copy_device(nullptr, nullptr, nullptr, 8, 8, nullptr, 0, 1, nullptr);
    (void)rt_max_batch_frames(nullptr, 8, 8);
    (void)rt_last_kernel_ms(nullptr, nullptr);
    (void)rt_ctx_synchronize(nullptr);
    (void)rt_render_multi(nullptr, 0, nullptr, nullptr, nullptr, 0, &fn, nullptr);
    (void)rt_render_multi_device(nullptr, 0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, nullptr);
    (void)rt_gather(nullptr, nullptr, 8, 8, nullptr, nullptr, nullptr, nullptr);
    (void)rt_peer_access(nullptr, nullptr);
    (void)rt_scene_commit(nullptr, nullptr, nullptr);
    (void)rt_scene_get_info(nullptr, nullptr);
    std::printf("capi host fuzz: %d partitions, sanitizers silent\n", cases);
    return 0;
}
