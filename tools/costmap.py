"""Per-pixel cost / schedule map of one frame (development tool).  Needs the -DRT_COSTMAP build:
   python -c "import importlib; importlib.import_module('ray-tracer_amd.build').build_variant('costmap', ['-DRT_COSTMAP=1'])"
   RT_AMD_LIB=ray-tracer_amd/libraytracer_amd_costmap.so python tools/costmap.py monkey 256 gpurun_out/costmap.npz
The frame then holds per pixel (own traversal steps, start tick, end tick), ticks of the 100 MHz
wall clock."""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "monkey"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dest = sys.argv[3] if len(sys.argv) > 3 else None
W, H = 1920, 1080
objs, sky = rt.scenes.CONFIG_SCENES[name]()
ctx = rt.Context(0)
scene = ctx.commit(rt.SceneObjects(objs))
out = torch.empty((H, W, 3), device="cuda:0")
for _ in range(2):
    rt.render_device(ctx, scene, rt.Camera(W, H), rt.RenderData(spp, 8, True, sky), 12345, 0, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
ms = ctx.last_kernel_ms()
m = out.cpu().numpy().view(np.uint32)
steps, t0, t1 = m[..., 0].astype(np.int64), m[..., 1].astype(np.int64), m[..., 2].astype(np.int64)
base = t0.min()
t0 = (t0 - base) & 0xffffffff; t1 = (t1 - base) & 0xffffffff
tick_ms = 1e-5
print("%s %dx%d spp=%d: kernel %.1f ms; last pixel ends at %.1f ms" % (name, W, H, spp, ms, t1.max() * tick_ms))
dur = (t1 - t0) * tick_ms
print("per-pixel steps: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %d" % (steps.mean(), *np.percentile(steps, [50, 90, 99]), steps.max()))
print("per-pixel duration ms: mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
# tiles
th, tw = H // 8, W // 8
ts = steps[:th * 8, :tw * 8].reshape(th, 8, tw, 8)
tile_max = ts.max(axis=(1, 3)); tile_sum = ts.sum(axis=(1, 3))
tt0 = t0[:th * 8, :tw * 8].reshape(th, 8, tw, 8).min(axis=(1, 3)) * tick_ms
tt1 = t1[:th * 8, :tw * 8].reshape(th, 8, tw, 8).max(axis=(1, 3)) * tick_ms
order = np.argsort(-tt1.ravel())[:12]
print("latest-finishing tiles: (end ms, start ms, max lane steps, mean lane steps, ns per max-lane step)")
for i in order:
    y, x = divmod(int(i), tw)
    print("  tile(%3d,%3d) end %.1f start %.1f  max %d mean %.0f  -> %.0f ns/step" % (x, y, tt1[y, x], tt0[y, x], tile_max[y, x], tile_sum[y, x] / 64,
          1e6 * (tt1[y, x] - tt0[y, x]) / max(tile_max[y, x], 1)))
print("total steps %.3e; steps/ms at full frame %.3e" % (steps.sum(), steps.sum() / ms))
late = tt0 > 1.0
print("tiles started after 1 ms: %d of %d; heaviest late tile mean-steps %.0f vs heaviest overall %.0f" % (late.sum(), late.size, (tile_sum[late] / 64).max() if late.any() else 0, (tile_sum / 64).max()))
hist, edges = np.histogram(t1 * tick_ms, bins=20)
print("pixels finishing per time bin:", list(zip(np.round(edges[:-1]).astype(int).tolist(), hist.tolist())))
if dest:
    np.savez_compressed(dest, steps=steps.astype(np.uint32), t0=t0.astype(np.uint32), t1=t1.astype(np.uint32), ms=ms)
