"""Frame-by-frame launches against one multi-frame launch (development tool): the same F
progressive frames of the 1080p monkey config.   python tools/batch_probe.py spp F"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
spp, F = int(sys.argv[1]), int(sys.argv[2])
name = sys.argv[3] if len(sys.argv) > 3 else "monkey"
W, H = 1920, 1080
objs, sky = rt.scenes.CONFIG_SCENES[name]()
ctx = rt.Context(0)
scene = ctx.commit(rt.SceneObjects(objs))
cam, rd = rt.Camera(W, H), rt.RenderData(spp, 8, True, sky)
times = [12345 + i for i in range(F)]
st = torch.cuda.current_stream().cuda_stream
a = torch.zeros((H, W, 3), device="cuda:0"); b = torch.empty_like(a); fr = torch.empty_like(a)
for rnd in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x, y = a, b
    for i, t in enumerate(times):
        rt.render_device(ctx, scene, cam, rd, t, i, y.data_ptr(), d_prev=x.data_ptr() if i else None, stream=st)
        x, y = y, x
    torch.cuda.synchronize(); t1 = time.perf_counter()
    rt.render_device_batch(ctx, scene, cam, rd, times, 0, fr.data_ptr(), stream=st)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    same = torch.equal(x.view(torch.int32), fr.view(torch.int32))
    print("%s %d spp x %d frames: frame by frame %.1f ms (%.0f Msamples/s), one launch %.1f ms (%.0f Msamples/s), identical=%s" % (
        name, spp, F, (t1 - t0) * 1e3, W * H * spp * F / (t1 - t0) / 1e6, (t2 - t1) * 1e3, W * H * spp * F / (t2 - t1) / 1e6, same), flush=True)
