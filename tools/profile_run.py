"""One launch on a config scene for rocprofv3 (development tool): a plain frame, or `frames` progressive
frames in one multi-frame launch.  usage: profile_run.py [scene] [spp] [W] [H] [frames]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "monkey"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
frames = int(sys.argv[5]) if len(sys.argv) > 5 else 1
objs, sky = rt.scenes.CONFIG_SCENES[name]()
ctx = rt.Context(0)
scene = ctx.commit(rt.SceneObjects(objs))
out = torch.empty((H, W, 3), device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
# the first launch of a view also measures tile costs (atomics per pixel): get that out of the way with one frame at the
# SAME sample count - figures from a launch with an eighth of the samples or fewer are provisional and the next launch
# would measure again (rt_capi.cpp), which is not the kind of launch bench.py times.  The summaries (tools/summarize_profile.py,
# tools/fit_traffic.py, tools/pmc_bound.sh) read the LAST dispatch of the kernel only.
rt.render_device(ctx, scene, rt.Camera(W, H), rt.RenderData(spp, 8, True, sky), 12345, 0, out.data_ptr(), stream=st)
torch.cuda.synchronize()
if frames > 1:
    rt.render_device_batch(ctx, scene, rt.Camera(W, H), rt.RenderData(spp, 8, True, sky), [12345 + i for i in range(frames)], 0, out.data_ptr(), stream=st)
else:
    rt.render_device(ctx, scene, rt.Camera(W, H), rt.RenderData(spp, 8, True, sky), 12345, 0, out.data_ptr(), stream=st)
ms = ctx.last_kernel_ms()
print("%s %dx%d spp=%d x %d frame(s) in one launch: %.2f ms, %.1f Msamples/s" % (name, W, H, spp, frames, ms, W * H * spp * frames / ms / 1e3))
