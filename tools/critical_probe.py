"""Critical-path probe (development tool): times rank 0's share of an 8-GPU run (one tile per
wave, so the time is the most expensive tile's) and the full frame, for threshold settings
"work_threshold:ready_break[:descend_keep[:hit_break]]" given on the command line."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
dm = importlib.import_module("ray-tracer_amd.distributed")
spp = int(sys.argv[1]); sets = sys.argv[2].split(",")
W, H = 1920, 1080
objs, sky = rt.scenes.monkey()
cam, rd = rt.Camera(W, H), rt.RenderData(spp, 8, True, sky)
ctxs = []
for s in sets:
    parts = s.split(":")
    os.environ["RT_AMD_WORK_THRESHOLD"], os.environ["RT_AMD_READY_BREAK"] = parts[0], parts[1]
    os.environ["RT_AMD_DESCEND_KEEP"] = parts[2] if len(parts) > 2 else "24"
    os.environ["RT_AMD_HIT_BREAK"] = parts[3] if len(parts) > 3 else "24"
    c = rt.Context(0)
    ctxs.append((c, c.commit(rt.SceneObjects(objs))))
res = {s: ([], []) for s in sets}
buf8 = torch.empty((dm.max_owned_rows(H, 8, 8), W, 3), device="cuda:0")
buf1 = torch.empty((H, W, 3), device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for rnd in range(4):
    for s, (c, sc) in zip(sets, ctxs):
        rt.render_device(c, sc, cam, rd, 12345, 0, buf8.data_ptr(), band_first=0, band_stride=8, compact=True, stream=st)
        res[s][0].append(c.last_kernel_ms())
        rt.render_device(c, sc, cam, rd, 12345, 0, buf1.data_ptr(), stream=st)
        res[s][1].append(c.last_kernel_ms())
for s in sets:
    a, b = sorted(res[s][0][1:]), sorted(res[s][1][1:])
    print("th:rb=%-6s  1/8 of frame %.1f ms   full frame %.1f ms (%.0f Msamples/s)" % (s, a[1], b[1], W * H * spp / b[1] / 1e3), flush=True)
