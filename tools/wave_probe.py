"""Where one wave's time goes (development tool, -DRT_STATS build): renders a tw x th-tile window
of the 1920x1080 monkey frame (same rays as those tiles of the full frame; seeds differ) and
prints the lap-timer split.  `python tools/wave_probe.py 118 64 1 1 256` = one wave on an idle GPU.
   RT_AMD_LIB=ray-tracer_amd/libraytracer_amd_stats.so python tools/wave_probe.py tx ty tw th spp"""
import ctypes as C, importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
tx, ty, tw, th, spp = (int(v) for v in sys.argv[1:6])
objs, sky = rt.scenes.monkey()
f = rt.Camera(1920, 1080).floats().astype(np.float32).copy()
f[3:6] = f[3:6] + f[6:9] * np.float32(8 * tx) + f[9:12] * np.float32(8 * ty)
W, H = 8 * tw, 8 * th
cam = rt.Camera(W, H, floats=f)
ctx = rt.Context(0)
scene = ctx.commit(rt.SceneObjects(objs))
out = torch.empty((H, W, 3), device="cuda:0")
for _ in range(2):
    rt.render_device(ctx, scene, cam, rt.RenderData(spp, 8, True, sky), 12345, 0, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
ms = ctx.last_kernel_ms()
buf = (C.c_uint64 * 48)()
rt.lib().rt_debug_read_stats(ctx._h, buf)
names = ["ITER", "SHADE", "SHADE_HIT", "FETCH", "GEN", "MESH", "MESH_START", "WORK_ITER", "NODE", "LEAF_TRI", "POP", "DONE_MESH"]
print("window tiles (%d,%d)+%dx%d spp=%d: kernel %.2f ms" % (tx, ty, tw, th, spp, ms))
for i, n in enumerate(names):
    if buf[2 * i]:
        print("  %-10s wave-execs %10d  lanes/exec %5.1f" % (n, buf[2 * i], buf[2 * i + 1] / buf[2 * i]))
tn = ["CTL", "SHADE", "FETCH", "GEN", "MESH", "DESCEND", "LEAF", "POP"]
t = [buf[24 + i] for i in range(8)]
waves, life = buf[33], buf[32]
tot = float(sum(t))
if tot == 0:
    sys.exit(0)          # not a -DRT_STATS build: only the kernel time is meaningful
print("  waves %d, mean wave lifetime %.2f ms; timer total %.3e ticks (%.1f ticks per us of lifetime)" % (waves, life / max(waves, 1) * 1e-5, tot, tot / max(life * 1e-2, 1e-9)))
for n, v in zip(tn, t):
    print("    %-8s %5.1f%%" % (n, 100.0 * v / tot))
ex = {n: buf[2 * i] for i, n in enumerate(names)}
tick_us = (life * 1e-2) / tot      # us per timer tick
for n, sect in (("NODE", "DESCEND"), ("LEAF_TRI", "LEAF"), ("POP", "POP"), ("SHADE", "SHADE"), ("GEN", "GEN")):
    if ex[n]:
        print("    %-8s %.0f ns per wave-exec" % (sect, 1e3 * t[tn.index(sect)] * tick_us / ex[n]))
