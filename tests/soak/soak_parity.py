"""One-off parity soak (test infrastructure, run by hand on the GPU box): many seeded random mixed scenes (tests/test_gpu_parity.py::_random_scene),
each rendered by a one-frame launch and by a three-frame launch (twice: the second runs on the measured-cost
schedule), against the CPU oracle, bit for bit.   python tests/soak/soak_parity.py <first seed> <count>"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("ray-tracer_amd")
from oracle import binding as orc
from test_gpu_parity import _random_scene
first, count = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "random"      # "random": random mixed scenes; "config": the config / reference scenes from random cameras; "bigmesh": triangle soups
ctx = rt.Context(0)
md = rt.scenes.models_dir()
bad = 0
names = ["monkey", "cube", "reference_scene0", "reference_scene1", "reference_scene2", "reference_scene3", "reference_scene4", "three_sphere"]
committed = {}
for seed in range(first, first + count):
    W, H, spp, limit = 96 + 8 * (seed % 5), 64 + 8 * (seed % 3), 3 + seed % 3, 2 + seed % 7
    if os.environ.get("RT_SOAK_SIZE"):                      # e.g. 640x360x32: bigger frames, more samples per pixel
        W, H, spp = [int(x) for x in os.environ["RT_SOAK_SIZE"].split("x")]
    if mode == "config":
        name = names[seed % len(names)]
        objs, sky = rt.scenes.CONFIG_SCENES[name]()
        rng = np.random.default_rng(seed)
        # a camera somewhere in front of / inside the scene, looking roughly at it
        pos = tuple(float(x) for x in rng.uniform([-0.6, -0.3, -0.5], [0.6, 0.6, 0.8]))
        cam = rt.Camera(W, H, pos=pos, fov=float(rng.uniform(0.6, 1.4)), focal_len=0.1, rot=tuple(float(x) for x in rng.uniform(-0.35, 0.35, 3)))
        if name not in committed:
            committed[name] = ctx.commit(rt.SceneObjects(objs))
        scene = committed[name]
    elif mode == "bigmesh":
        # one or two random triangle soups of 100..3,000 triangles (leaves of the fixed-depth-10 tree then hold several
        # triangles; the larger ones do not fit LDS and run the global-memory variant) + spheres
        rng = np.random.default_rng(seed)
        objs = []
        for _ in range(int(rng.integers(1, 3))):
            n = int(rng.integers(100, 3000))
            c = rng.uniform([-0.8, -0.5, 1.4], [0.8, 0.6, 3.0])
            tris = (c + rng.normal(0, 0.35, (n, 1, 3)) + rng.normal(0, 0.06, (n, 3, 3))).astype(np.float32).reshape(n, 9)
            mat = ("standard", tuple(float(x) for x in rng.uniform(0.2, 1, 3)), float(rng.uniform(0, 0.6))) if rng.integers(0, 3) else ("refractive", (1.0, 1.0, 1.0), 1.5)
            objs.append(("mesh", tris, mat))
        objs.append(("sphere", (0, 1.3, 1.5), 0.5, ("emissive", (1, 1, 1), 5.0)))
        if rng.integers(0, 2):
            objs.append(("sphere", (0, -100.5, 1.5), 100, ("standard", (0.5, 0.5, 0.5), 0.2)))
        sky = (0.8, 1.0, 1.0) if rng.integers(0, 2) else (0.0, 0.0, 0.0)
        cam = rt.Camera(W, H)
        scene = ctx.commit(rt.SceneObjects(objs))
    else:
        objs, sky = _random_scene(seed)
        cam = rt.Camera(W, H)
        scene = ctx.commit(rt.SceneObjects(objs))
    o = orc.Scene(objs, orc.MATH_DET, md)
    want1 = o.render(cam.floats(), W, H, spp, limit, sky, time_ms=seed)
    want = want1
    for k in (1, 2):
        want = o.render(cam.floats(), W, H, spp, limit, sky, time_ms=seed + k, frame_num=k, prev=want)
    rd = rt.RenderData(spp, limit, True, sky)
    d = rt.VariableRenderData(W, H)
    rt.render(ctx, scene, cam, rd, d, seed)
    ok = np.array_equal(d.previous_render.view(np.uint32), want1.view(np.uint32))
    for rep in range(2):
        d3 = rt.VariableRenderData(W, H)
        rt.render_frames(ctx, scene, cam, rd, d3, [seed, seed + 1, seed + 2])
        ok = ok and np.array_equal(d3.previous_render.view(np.uint32), want.view(np.uint32))
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, flush=True)
    if (seed - first + 1) % 500 == 0:
        print("... %d scenes, %d mismatches so far" % (seed - first + 1, bad), flush=True)     # a long run must not look hung
print("soak: %d scenes, %d mismatches" % (count, bad), flush=True)
sys.exit(1 if bad else 0)
