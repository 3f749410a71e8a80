"""Quick GPU parity + timing probe (development tool; the real checks live in tests/)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
from oracle import binding as B

ctx = rt.Context(0)
md = rt.scenes.models_dir()
ok = True
for name, limit in (("three_sphere", 4), ("cube", 8), ("monkey", 8), ("reference_scene0", 5), ("reference_scene1", 5)):
    objs, sky = rt.scenes.CONFIG_SCENES[name]()
    W = H = 256
    spp = 16
    so = rt.SceneObjects(objs)
    scene = ctx.commit(so)
    cam = rt.Camera(W, H)
    data = rt.VariableRenderData(W, H)
    rt.render(ctx, scene, cam, rt.RenderData(spp, limit, True, sky), data, 12345)
    gpu = data.previous_render.copy()
    ref = B.Scene(objs, B.MATH_DET, md).render(cam.floats(), W, H, spp, limit, sky)
    diff = np.abs(gpu - ref)
    nbad = int((gpu.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
    print(f"{name}: info={scene.info()} kernel_ms={ctx.last_kernel_ms():.3f} max|diff|={diff.max():.3g} pixels_differing={nbad}/{W*H} mean={gpu.mean():.6f}/{ref.mean():.6f}", flush=True)
    ok &= nbad == 0
print("PARITY", "OK" if ok else "FAIL", flush=True)

import torch
for name, spp in (("three_sphere", 64), ("cube", 64), ("monkey", 64), ("monkey", 256)):
    objs, sky = rt.scenes.CONFIG_SCENES[name]()
    W, H = 1920, 1080
    scene = ctx.commit(rt.SceneObjects(objs))
    cam = rt.Camera(W, H)
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
    rd = rt.RenderData(spp, 8, True, sky)
    for it in range(2):
        rt.render_device(ctx, scene, cam, rd, 12345, 0, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        ms = ctx.last_kernel_ms()
    print(f"{name} 1920x1080 spp={spp}: {ms:.2f} ms -> {W*H*spp/ms/1e3:.1f} Msamples/s mean={out.mean().item():.6f}", flush=True)
