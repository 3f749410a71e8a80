"""Partition soak (development tool, GPU only - no oracle): random scenes, image sizes, rank counts, band heights and
frame counts; the multi-GPU entry points (several contexts on this GPU standing in for several GPUs:
rt_render_multi_device over two or three calls - static bands, or band_rows 0 = interleaved tile lists on the first
call and cost-balanced ones from the second - and rt_render_device_batch + rt_gather by hand, bands or a random
partition into tile lists with cost hints) must reproduce the single-context frame bit for bit.
   python tests/soak/soak_partition.py <first seed> <count>"""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("ray-tracer_amd")
from test_gpu_parity import _random_scene
first, count = int(sys.argv[1]), int(sys.argv[2])
ctxs = [rt.Context(0) for _ in range(6)]
names = ["monkey", "cube", "reference_scene0", "three_sphere"]
bad = 0
side = torch.cuda.Stream()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    if seed % 3 == 0:
        objs, sky = rt.scenes.CONFIG_SCENES[names[(seed // 3) % len(names)]]()
    else:
        objs, sky = _random_scene(seed)
    W, H = int(rng.integers(1, 200)), int(rng.integers(1, 150))
    n = int(rng.integers(1, 7))
    band_rows = int(rng.choice([0, 0, 8, 8, 16, 24]))           # 0: cost-balanced tile lists
    f1, f2 = int(rng.integers(1, 5)), int(rng.integers(0, 4))
    f3 = int(rng.integers(0, 3)) if band_rows == 0 else 0
    spp, limit = int(rng.integers(1, 4)), int(rng.integers(1, 7))
    cam, rd = rt.Camera(W, H), rt.RenderData(spp, limit, True, sky)
    so = rt.SceneObjects(objs)
    scenes = [c.commit(so) for c in ctxs[:n]]
    times = [int(x) for x in rng.integers(-2**31, 2**31 - 1, f1 + f2 + f3)]
    ref = rt.VariableRenderData(W, H)
    rt.render_frames(ctxs[0], scenes[0], cam, rd, ref, times)
    frame = torch.full((H, W, 3), -1.0, device="cuda:0")
    torch.cuda.synchronize()
    stream = side.cuda_stream if seed % 2 else None
    rt.render_multi_device(ctxs[:n], scenes, cam, rd, times[:f1], 0, frame.data_ptr(), band_rows=band_rows, stream=stream)
    if f2:
        rt.render_multi_device(ctxs[:n], scenes, cam, rd, times[f1:f1 + f2], f1, frame.data_ptr(), band_rows=band_rows, stream=stream)
    if f3:
        rt.render_multi_device(ctxs[:n], scenes, cam, rd, times[f1 + f2:], f1 + f2, frame.data_ptr(), band_rows=band_rows, stream=stream)
    torch.cuda.synchronize()
    ok = np.array_equal(frame.cpu().numpy().view(np.uint32), ref.previous_render.view(np.uint32))
    # by hand: compact launches + rt_gather
    out = torch.full((H, W, 3), -1.0, device="cuda:0")
    bufs = []
    if band_rows == 0:
        # a random partition into tile lists (any partition must do), random costs as hints for half the ranks
        ntiles = ((W + 7) // 8) * ((H + 7) // 8)
        owner = rng.integers(0, n, ntiles)
        for r in range(n):
            ids = rng.permutation(np.flatnonzero(owner == r)).astype(np.uint32)
            buf = torch.zeros((max(len(ids), 1) * 192,), device="cuda:0")
            bufs.append(buf)
            hints = rng.integers(0, 1000, len(ids)).astype(np.uint32) if r % 2 else None
            peaks = rng.integers(0, 100, len(ids)).astype(np.uint32) if r % 4 == 1 else None
            rt.render_device_batch(ctxs[r], scenes[r], cam, rd, times, 0, buf.data_ptr(), compact=True, stream=stream, tile_list=ids, tile_cost=hints, tile_peak=peaks)
            rt.gather(ctxs[0], out.data_ptr(), W, H, ctxs[r], buf.data_ptr(), stream=stream, tile_list=ids)
    else:
        for r in range(n):
            rows = rt.tile_owned_rows(H, band_rows, r, n)
            buf = torch.zeros((max(rows, 1), W, 3), device="cuda:0")
            bufs.append(buf)
            if rows:
                rt.render_device_batch(ctxs[r], scenes[r], cam, rd, times, 0, buf.data_ptr(), band_rows=band_rows, band_first=r, band_stride=n, compact=True, stream=stream)
                rt.gather(ctxs[0], out.data_ptr(), W, H, ctxs[r], buf.data_ptr(), band_rows, r, n, stream=stream)
    torch.cuda.synchronize()
    ok = ok and np.array_equal(out.cpu().numpy().view(np.uint32), ref.previous_render.view(np.uint32))
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, "WxH", W, H, "ranks", n, "band_rows", band_rows, "frames", f1, f2, f3, flush=True)
    if (seed - first + 1) % 1000 == 0:
        print("... %d cases, %d mismatches so far" % (seed - first + 1, bad), flush=True)
print("partition soak: %d cases, %d mismatches" % (count, bad), flush=True)
sys.exit(1 if bad else 0)
