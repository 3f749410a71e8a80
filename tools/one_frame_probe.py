import importlib, sys, os, time
sys.path.insert(0, os.getcwd())
import torch
rt = importlib.import_module("ray-tracer_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "monkey"
W, H, spp, limit = (1000, 800, 100, 5) if name == "reference_scene0" else (1920, 1080, 1024, 8)
objs, sky = rt.scenes.CONFIG_SCENES[name]()
ctx = rt.Context(0); scene = ctx.commit(rt.SceneObjects(objs))
cam, rd = rt.Camera(W, H), rt.RenderData(spp, limit, True, sky)
st = torch.cuda.current_stream().cuda_stream
fr = torch.zeros((H, W, 3), device="cuda:0")
for i in range(2):
    rt.render_device(ctx, scene, cam, rd, 777 + i, 0, fr.data_ptr(), stream=st)
ms = []
for i in range(8):
    rt.render_device(ctx, scene, cam, rd, 12345 + i, 0, fr.data_ptr(), stream=st)
    ms.append(ctx.last_kernel_ms())
ms.sort()
print("%s first_pack=%s: one frame %.1f ms median of 8 (min %.1f max %.1f)" % (name, os.environ.get("RT_AMD_FIRST_PACK", "0"), ms[4], ms[0], ms[-1]), flush=True)
