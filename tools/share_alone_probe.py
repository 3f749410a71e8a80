import importlib, sys, time, os
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/ray-tracer_amd") else os.getcwd())
import torch
rt = importlib.import_module("ray-tracer_amd")
objs, sky = rt.scenes.CONFIG_SCENES["monkey"]()
W, H = 1920, 1080
ctx = rt.Context(0); scene = ctx.commit(rt.SceneObjects(objs))
cam, rd = rt.Camera(W, H), rt.RenderData(1024, 8, True, sky)
st = torch.cuda.current_stream().cuda_stream
fr = torch.zeros((H, W, 3), device="cuda:0")
for depth in (1, 2, 4, 8):
    rt.frame_depth(ctx, depth)
    for i in range(2):
        rt.frame_submit(ctx, scene, cam, rd, 777 + i); rt.frame_collect(ctx, i, fr.data_ptr(), stream=st); rt.frame_wait(ctx)
    t0 = time.perf_counter()
    for i in range(4):
        rt.frame_submit(ctx, scene, cam, rd, 12345 + i); rt.frame_collect(ctx, i, fr.data_ptr(), stream=st); rt.frame_wait(ctx)
    el = (time.perf_counter() - t0) / 4
    print("a frame ALONE on 1/%d of the CUs: %.1f ms  (x CUs share: %.1f ms of whole-GPU time; the multi-frame launch needs 206.7)" % (depth, el * 1e3, el * 1e3 / depth), flush=True)
