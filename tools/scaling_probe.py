"""What the ranks of an N-GPU strong-scaling run do, measured on ONE GPU (development tool): for N = 1, 2, 4, 8
renders, one rank after another, the bands each rank would own of `frames` progressive frames (one multi-frame
launch per rank, as bench.py does) and prints the slowest rank's kernel time; T(1)/T(N) projects the strong-scaling
speed-up (the gather adds ~0.1 ms per rank).  usage: scaling_probe.py [spp] [frames] [W] [H] [scene]; RT_PROBE_N=1,8 picks the Ns"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
dm = importlib.import_module("ray-tracer_amd.distributed")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
name = sys.argv[5] if len(sys.argv) > 5 else "monkey"
objs, sky = rt.scenes.CONFIG_SCENES[name]()
cam, rd = rt.Camera(W, H), rt.RenderData(spp, 8, True, sky)
st = torch.cuda.current_stream().cuda_stream
t1 = None
for n in [int(x) for x in os.environ.get("RT_PROBE_N", "1,2,4,8").split(",")]:
    per_rank = []
    for r in range(n):
        ctx = rt.Context(0)                      # a rank is a process with its own context: its own tile-order cache
        scene = ctx.commit(rt.SceneObjects(objs))
        buf = torch.zeros((dm.max_owned_rows(H, 8, n), W, 3), device="cuda:0")
        # warm-up launch of 5 frames (collects tile costs, like bench.py --warmup 5), then the timed launch
        rt.render_device_batch(ctx, scene, cam, rd, [12345 + i for i in range(min(5, frames))], 0, buf.data_ptr(), band_first=r, band_stride=n, compact=True, stream=st)
        rt.render_device_batch(ctx, scene, cam, rd, [12345 + i for i in range(frames)], 0, buf.data_ptr(), band_first=r, band_stride=n, compact=True, stream=st)
        per_rank.append(ctx.last_kernel_ms())
        del scene, ctx
    worst = max(per_rank)
    t1 = t1 or worst
    print("N=%d: slowest rank %.1f ms (ranks: %s) -> %.0f Msamples/s, speed-up %.2fx (efficiency %.0f%%)" % (
        n, worst, " ".join("%.0f" % v for v in per_rank), W * H * spp * frames / worst / 1e3, t1 / worst, 100 * t1 / worst / n), flush=True)
