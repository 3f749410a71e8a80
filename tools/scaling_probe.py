"""What one rank of an N-GPU run does, measured on ONE GPU (development tool): renders the bands
rank 0 would own for N = 1, 2, 4, 8 and prints the kernel time; T(1)/T(N) is the best-case
strong-scaling speed-up (the gather adds ~0.1 ms)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
dm = importlib.import_module("ray-tracer_amd.distributed")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
objs, sky = rt.scenes.monkey()
ctx = rt.Context(0)
scene = ctx.commit(rt.SceneObjects(objs))
cam, rd = rt.Camera(W, H), rt.RenderData(spp, 8, True, sky)
t1 = None
for n in (1, 2, 4, 8):
    worst = 0.0
    for r in sorted(set((0, n // 2, n - 1))):
        buf = torch.empty((dm.max_owned_rows(H, 8, n), W, 3), device="cuda:0")
        ts = []
        for _ in range(3):
            rt.render_device(ctx, scene, cam, rd, 12345, 0, buf.data_ptr(), band_first=r, band_stride=n, compact=True, stream=torch.cuda.current_stream().cuda_stream)
            ts.append(ctx.last_kernel_ms())
        worst = max(worst, sorted(ts)[1])
    t1 = t1 or worst
    print("N=%d: slowest rank %.1f ms  -> speed-up %.2fx (efficiency %.0f%%)" % (n, worst, t1 / worst, 100 * t1 / worst / n), flush=True)
