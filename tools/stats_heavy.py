"""Section statistics on a view where the mesh fills the frame (every tile is 'heavy').
Needs the -DRT_STATS build: RT_AMD_LIB=ray-tracer_amd/libraytracer_amd_stats.so python tools/stats_heavy.py"""
import ctypes as C, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W = H = 1024
objs, sky = rt.scenes.monkey()
ctx = rt.Context(0)
scene = ctx.commit(rt.SceneObjects(objs))
cam = rt.Camera(W, H, pos=(0.1, -0.1, 0.0), fov=0.42, focal_len=0.1)
out = torch.empty((H, W, 3), device="cuda:0")
for _ in range(2):
    rt.render_device(ctx, scene, cam, rt.RenderData(spp, 8, True, sky), 12345, 0, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    ms = ctx.last_kernel_ms()
buf = (C.c_uint64 * 48)()
rt.lib().rt_debug_read_stats(ctx._h, buf)
names = ["ITER", "SHADE", "SHADE_HIT", "FETCH", "GEN", "MESH", "MESH_START", "WORK_ITER", "NODE", "LEAF_TRI", "POP", "DONE_MESH"]
cost = {"ITER": 20, "SHADE": 30, "SHADE_HIT": 700, "GEN": 300, "MESH": 40, "WORK_ITER": 15, "NODE": 84, "LEAF_TRI": 95, "POP": 12}
samples = W * H * spp
print("monkey close-up %dx%d spp=%d: %.2f ms, %.1f Msamples/s, frame mean %.4f" % (W, H, spp, ms, samples / ms / 1e3, out.mean().item()))
tot = sum(buf[2 * i] * 64 * cost.get(n, 0) for i, n in enumerate(names))
for i, n in enumerate(names):
    ex, ln = buf[2 * i], buf[2 * i + 1]
    if ex:
        print("  %-10s wave-execs %11d lanes/exec %5.1f  lane-execs/sample %8.3f  share of issued slots %5.1f%%" % (n, ex, ln / ex, ln / samples, 100.0 * ex * 64 * cost.get(n, 0) / tot))
