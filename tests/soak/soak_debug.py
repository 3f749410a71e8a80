"""Details of one soak seed (test infrastructure, run by hand on the GPU box): python tests/soak/soak_debug.py <seed>"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("ray-tracer_amd")
from oracle import binding as orc
from test_gpu_parity import _random_scene
seed = int(sys.argv[1])
ctx = rt.Context(0)
md = rt.scenes.models_dir()
objs, sky = _random_scene(seed)
W, H, spp, limit = 96 + 8 * (seed % 5), 64 + 8 * (seed % 3), 3 + seed % 3, 2 + seed % 7
print("seed", seed, "WxH", W, H, "spp", spp, "limit", limit, "sky", sky)
for i, o in enumerate(objs):
    print(i, o[0], (o[-1][0] if isinstance(o[-1], tuple) else o[-1]), ("ntris=%d" % len(o[1]) if o[0] == "mesh" else ""))
cam = rt.Camera(W, H)
scene = ctx.commit(rt.SceneObjects(objs))
print(scene.info())
o = orc.Scene(objs, orc.MATH_DET, md)
rd = rt.RenderData(spp, limit, True, sky)
for t in (seed, seed + 1, seed + 2):
    want = o.render(cam.floats(), W, H, spp, limit, sky, time_ms=t)
    d = rt.VariableRenderData(W, H)
    rt.render(ctx, scene, cam, rd, d, t)
    diff = (d.previous_render != want).any(axis=2)
    print("time", t, "differing pixels", int(diff.sum()))
    for y, x in zip(*np.nonzero(diff)):
        print("   px", x, y, "hip", d.previous_render[y, x], "oracle", want[y, x])
    # per-sample: render with 1 spp repeatedly is not the same stream; instead find the first differing spp count
    if diff.any():
        for s in range(1, spp + 1):
            w2 = o.render(cam.floats(), W, H, s, limit, sky, time_ms=t)
            d2 = rt.VariableRenderData(W, H)
            rt.render(ctx, scene, cam, rt.RenderData(s, limit, True, sky), d2, t)
            n = int((d2.previous_render != w2).any(axis=2).sum())
            print("   spp", s, "differing", n)
            if n:
                for l in range(1, limit + 1):
                    w3 = o.render(cam.floats(), W, H, s, l, sky, time_ms=t)
                    d3 = rt.VariableRenderData(W, H)
                    rt.render(ctx, scene, cam, rt.RenderData(s, l, True, sky), d3, t)
                    print("      limit", l, "differing", int((d3.previous_render != w3).any(axis=2).sum()))
                break
