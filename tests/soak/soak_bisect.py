"""Shrinks a mismatching soak scene (test infrastructure, run by hand on the GPU box): drops objects one at a time while the HIP frame still differs
from the oracle.   python tests/soak/soak_bisect.py <seed> <time_ms> <spp> <limit>"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("ray-tracer_amd")
from oracle import binding as orc
from test_gpu_parity import _random_scene
seed, t, spp, limit = [int(x) for x in sys.argv[1:5]]
ctx = rt.Context(0)
md = rt.scenes.models_dir()
objs, sky = _random_scene(seed)
W, H = 96 + 8 * (seed % 5), 64 + 8 * (seed % 3)
cam = rt.Camera(W, H)

def differs(description):
    scene = ctx.commit(rt.SceneObjects(description))
    want = orc.Scene(description, orc.MATH_DET, md).render(cam.floats(), W, H, spp, limit, sky, time_ms=t)
    d = rt.VariableRenderData(W, H)
    rt.render(ctx, scene, cam, rt.RenderData(spp, limit, True, sky), d, t)
    diff = (d.previous_render != want).any(axis=2)
    return [(int(x), int(y)) for y, x in zip(*np.nonzero(diff))], d.previous_render, want

cur = list(objs)
changed = True
while changed:
    changed = False
    for i in range(len(cur)):
        trial = cur[:i] + cur[i + 1:]
        px, _, _ = differs(trial)
        if px:
            cur = trial
            changed = True
            break
px, got, want = differs(cur)
print("minimal scene: %d objects, differing pixels %s" % (len(cur), px))
for o in cur:
    if o[0] == "mesh":
        print(("mesh", np.asarray(o[1]).tolist(), o[2]))
    else:
        print(o)
for x, y in px:
    print("px", x, y, "hip", got[y, x], "oracle", want[y, x])
    cp = cam.floats()
    d = cp[3:6] + cp[6:9] * np.float32(x) + cp[9:12] * np.float32(y) - cp[0:3]
    sc = orc.Scene(cur, orc.MATH_DET, md)
    print("primary (un-jittered) hit:", sc.trace_one(cp[0:3], (d / np.linalg.norm(d)).astype(np.float32)))
