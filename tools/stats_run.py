"""Section statistics of the render kernel (development tool).  Needs the -DRT_STATS build:
   python -c "import importlib; importlib.import_module('ray-tracer_amd.build').build_variant('stats', ['-DRT_STATS'])"
   RT_AMD_LIB=ray-tracer_amd/libraytracer_amd_stats.so python tools/stats_run.py monkey 16"""
import ctypes as C, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "monkey"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 1
W, H = 1920, 1080
objs, sky = rt.scenes.CONFIG_SCENES[name]()
ctx = rt.Context(0)
scene = ctx.commit(rt.SceneObjects(objs))
out = torch.empty((H, W, 3), device="cuda:0")
if frames > 1:
    rt.render_device_batch(ctx, scene, rt.Camera(W, H), rt.RenderData(spp, 8, True, sky), [12345 + i for i in range(frames)], 0, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
else:
    rt.render_device(ctx, scene, rt.Camera(W, H), rt.RenderData(spp, 8, True, sky), 12345, 0, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
ms = ctx.last_kernel_ms()
buf = (C.c_uint64 * 48)()
rt.lib().rt_debug_read_stats(ctx._h, buf)
names = ["ITER", "SHADE", "SHADE_HIT", "FETCH", "GEN", "MESH", "MESH_START", "WORK_ITER", "NODE", "LEAF_TRI", "POP", "DONE_MESH"]
cost = {"ITER": 20, "SHADE": 30, "SHADE_HIT": 700, "GEN": 300, "MESH": 40, "WORK_ITER": 15, "NODE": 75, "LEAF_TRI": 95, "POP": 12}
print("%s %dx%d spp=%d x %d frame(s): %.2f ms, %.1f Msamples/s  (info %s)" % (name, W, H, spp, frames, ms, W * H * spp * frames / ms / 1e3, scene.info()))
samples = W * H * spp * frames
tot_slots = tot_useful = 0
for i, n in enumerate(names):
    ex, ln = buf[2 * i], buf[2 * i + 1]
    c = cost.get(n, 0)
    tot_slots += ex * 64 * c
    tot_useful += ln * c
    if ex:
        print("  %-10s wave-execs %12d  lanes/exec %5.1f  lane-execs/sample %7.3f  est. slots %5.1f%%" % (n, ex, ln / ex, ln / samples, 0))
print("  estimated lane utilisation (cost-weighted): %.3f" % (tot_useful / max(tot_slots, 1)))
for i, n in enumerate(names):
    ex = buf[2 * i]; c = cost.get(n, 0)
    if ex and c:
        print("    %-10s share of issued slots %5.1f%%  (util %.2f)" % (n, 100.0 * ex * 64 * c / tot_slots, buf[2 * i + 1] / ex / 64))

tn = ["CTL", "SHADE", "FETCH", "GEN", "MESH", "DESCEND", "LEAF", "POP"]
t = [buf[24 + i] for i in range(8)]
if sum(t):
    print("  wave time by section: " + "  ".join("%s %.1f%%" % (n, 100.0 * v / sum(t)) for n, v in zip(tn, t)))
    print("  waves %d, mean wave lifetime %.1f ms" % (buf[33], buf[32] / max(buf[33], 1) * 1e-5))
