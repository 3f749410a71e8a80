"""Per-tile cost COMPONENTS of a view (development tool for fitting RT_COST_STEP / _GEN / _HIT): builds three variants
of the library that charge a pixel for one thing each (traversal macro steps / generated rays / shaded hits), renders
the whole image once with each (a 5-frame launch, like a warm-up) and saves the three tile maps.

    cost_maps.py out.npz [spp] [W] [H] [scene]            (run on the GPU box; the variants are built there)
"""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHILD = r'''
import importlib, sys, numpy as np, torch
sys.path.insert(0, %r)
rt = importlib.import_module("ray-tracer_amd")
out, spp, W, H, name = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
objs, sky = rt.scenes.CONFIG_SCENES[name]()
ctx = rt.Context(0)
scene = ctx.commit(rt.SceneObjects(objs))
buf = torch.zeros((H, W, 3), device="cuda:0")
rt.render_device_batch(ctx, scene, rt.Camera(W, H), rt.RenderData(spp, 8, True, sky), [12345 + i for i in range(5)], 0, buf.data_ptr(),
                       stream=torch.cuda.current_stream().cuda_stream)
ids, c = ctx.tile_costs()
full = np.zeros(((W + 7) // 8) * ((H + 7) // 8), np.uint32)
full[ids] = c
np.save(out, full)
''' % ROOT
if __name__ == "__main__":
    import numpy as np
    out = sys.argv[1]
    spp = sys.argv[2] if len(sys.argv) > 2 else "1024"
    W, H = (sys.argv[3], sys.argv[4]) if len(sys.argv) > 4 else ("1920", "1080")
    name = sys.argv[5] if len(sys.argv) > 5 else "monkey"
    bmod = importlib.import_module("ray-tracer_amd.build")
    maps = {}
    for key, flags in (("step", ["-DRT_COST_STEP=1", "-DRT_COST_GEN=0", "-DRT_COST_HIT=0"]), ("gen", ["-DRT_COST_STEP=0", "-DRT_COST_GEN=1", "-DRT_COST_HIT=0"]),
                       ("hit", ["-DRT_COST_STEP=0", "-DRT_COST_GEN=0", "-DRT_COST_HIT=1"])):
        lib = os.path.join(os.path.dirname(bmod.LIB), "libraytracer_amd_cost_%s.so" % key)
        if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(bmod.build()):
            lib = bmod.build_variant("cost_" + key, flags)      # (build these before `gpurun`: the .so files travel with the snapshot)
        tmp = "/tmp/cost_%s.npy" % key
        subprocess.check_call([sys.executable, "-c", CHILD, tmp, spp, W, H, name], env=dict(os.environ, RT_AMD_LIB=lib))
        maps[key] = np.load(tmp)
        print(key, "total", int(maps[key].astype(np.int64).sum()), flush=True)
    np.savez_compressed(out, **maps)
