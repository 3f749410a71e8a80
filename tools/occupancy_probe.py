"""What waves per SIMD are worth to a monkey-like traversal (development tool): the first `ntri` triangles of the
monkey mesh (the reference's fixed-depth-10 tree, so traversal looks like the full mesh's) + the config's two spheres
fit LDS several times over, so the same scene can run as 1 x 1024, 2 x 512 or 4-6 x 256 threads per CU.
   RT_AMD_THREADS=256 RT_AMD_BLOCKS_PER_CU=5 python tools/occupancy_probe.py 150 256 8
Round 2: 167 ms for every 256- and 512-thread shape (4, 5 or 6 waves per SIMD), 171.5 ms for 1 x 1024; a 640-thread
workgroup (10 waves: an uneven 3,3,2,2 over the SIMDs) ran at 214-222 ms whether one or two were asked for per CU - the
second one is evidently not co-resident - so "two 640-thread workgroups" is not a way to 5 waves per SIMD."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
ntri = int(sys.argv[1]) if len(sys.argv) > 1 else 150
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 8
W, H = 1920, 1080
objs, sky = rt.scenes.monkey()
m = rt.ObjFileMesh(os.path.join(rt.scenes.models_dir(), "low_poly_monkey.obj"))
for t in objs[0][2]:
    getattr(m, t[0])(*t[1:])
tris = m.triangles()
# the triangles nearest the camera-facing centre of the head, so the part is one connected blob
c = tris.reshape(-1, 3, 3).mean(axis=1)
order = ((c - c.mean(axis=0)) ** 2).sum(axis=1).argsort()
part = tris[order[:ntri]]
so = rt.SceneObjects()
so.create_mesh(part, rt.Material.from_desc(objs[0][3]))
so.add_description(objs[1:])
ctx = rt.Context(0)
scene = ctx.commit(so)
out = torch.empty((H, W, 3), device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
cam = rt.Camera(W, H)
rt.render_device(ctx, scene, cam, rt.RenderData(1, 8, True, sky), 12345, 0, out.data_ptr(), stream=st)
torch.cuda.synchronize()
rt.render_device_batch(ctx, scene, cam, rt.RenderData(spp, 8, True, sky), [12345 + i for i in range(frames)], 0, out.data_ptr(), stream=st)
ms = ctx.last_kernel_ms()
print("%d triangles, %s: launch: %.2f ms, %.1f Msamples/s" % (ntri, scene.info(), ms, W * H * spp * frames / ms / 1e3))
