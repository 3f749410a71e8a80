"""One render of a config scene for rocprofv3 (development tool).  usage: profile_run.py [scene] [spp] [W] [H]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "monkey"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
objs, sky = rt.scenes.CONFIG_SCENES[name]()
ctx = rt.Context(0)
scene = ctx.commit(rt.SceneObjects(objs))
out = torch.empty((H, W, 3), device="cuda:0")
rt.render_device(ctx, scene, rt.Camera(W, H), rt.RenderData(spp, 8, True, sky), 12345, 0, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
ms = ctx.last_kernel_ms()
print("%s %dx%d spp=%d: %.2f ms, %.1f Msamples/s" % (name, W, H, spp, ms, W * H * spp / ms / 1e3))
