"""Interleaved A/B of RT_AMD_WORK_THRESHOLD values in ONE process (development tool)."""
import importlib, os, sys, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "monkey"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ths = [x for x in (sys.argv[3] if len(sys.argv) > 3 else "1,4,8,12,16,24,32").split(",")]   # "th" or "th:ready_break"
W, H = 1920, 1080
objs, sky = rt.scenes.CONFIG_SCENES[name]()
ctxs = {}
for th in ths:
    os.environ["RT_AMD_WORK_THRESHOLD"] = th.split(":")[0]
    os.environ["RT_AMD_READY_BREAK"] = th.split(":")[1] if ":" in th else "65"
    c = rt.Context(0)
    ctxs[th] = (c, c.commit(rt.SceneObjects(objs)))
out = torch.empty((H, W, 3), device="cuda:0")
res = {th: [] for th in ths}
for rnd in range(int(os.environ.get("AB_ROUNDS", "5"))):
    for th in ths:
        c, sc = ctxs[th]
        rt.render_device(c, sc, rt.Camera(W, H), rt.RenderData(spp, 8, True, sky), 12345, 0, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        res[th].append(c.last_kernel_ms())
for th in ths:
    v = sorted(res[th][1:])
    print("TH=%s  median %.2f ms  min %.2f ms  -> %.0f Msamples/s" % (th, v[len(v)//2], v[0], W*H*spp/v[len(v)//2]/1e3))
