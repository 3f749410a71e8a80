"""sha256 of a handful of small frames rendered by the loaded library (development tool; bit-exactness check between library builds):
config scenes and reference scenes at 96x64, 8 spp, 3 frames in one launch + one plain frame."""
import hashlib, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
h = hashlib.sha256()
W, H = 96, 64
ctx = rt.Context(0)
for name in sorted(rt.scenes.CONFIG_SCENES):
    objs, sky = rt.scenes.CONFIG_SCENES[name]()
    scene = ctx.commit(rt.SceneObjects(objs))
    out = torch.zeros((H, W, 3), device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    rt.render_device(ctx, scene, rt.Camera(W, H), rt.RenderData(8, 8, True, sky), 777, 0, out.data_ptr(), stream=st)
    torch.cuda.synchronize()
    h.update(out.cpu().numpy().tobytes())
    rt.render_device_batch(ctx, scene, rt.Camera(W, H), rt.RenderData(8, 8, True, sky), [100, 101, 102], 0, out.data_ptr(), stream=st)
    torch.cuda.synchronize()
    h.update(out.cpu().numpy().tobytes())
print(h.hexdigest()[:16])
