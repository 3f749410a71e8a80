"""What the ranks of an N-GPU strong-scaling run do, measured on ONE GPU (development tool): for N = 1, 2, 4, 8
renders, one rank after another, what each rank would own of `frames` progressive frames (one multi-frame launch
per rank, as bench.py does) and prints every rank's kernel time; the slowest is the projected N-GPU time
(the gather adds ~0.1 ms per rank), T(1)/T(N) the projected strong-scaling speed-up.

    scaling_probe.py [spp] [frames] [W] [H] [scene]
    RT_PROBE_N=1,8        which Ns
    RT_PROBE_PART=lists   cost-balanced tile lists (bench.py's default; the warm-up runs on the interleaved ownership and
                          measures the tiles) | bands (round 2: band b of 8 rows -> rank b % N) | both
    RT_PROBE_DUMP=path    also writes {N: {part: {"ms": [...], "lists": [[tile ids]...]}}} as JSON (tools/fit_cost_weights.py)
    RT_PROBE_LISTS=path   lists only: take the ownership from an earlier dump instead of computing it (the tiles are still
                          measured first, for the hints): same partition, different builds' schedules
"""
import importlib, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
dm = importlib.import_module("ray-tracer_amd.distributed")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
name = sys.argv[5] if len(sys.argv) > 5 else "monkey"
parts = os.environ.get("RT_PROBE_PART", "lists")
parts = ["lists", "bands"] if parts == "both" else [parts]
objs, sky = rt.scenes.CONFIG_SCENES[name]()
so = rt.SceneObjects(objs)
cam, rd = rt.Camera(W, H), rt.RenderData(spp, 8, True, sky)
st = torch.cuda.current_stream().cuda_stream
warm = [12345 + i for i in range(min(5, frames))]
timed = [12345 + i for i in range(frames)]
tiles_x, tiles_y = dm.tiles_xy(W, H)
dump, t1 = {}, None
for n in [int(x) for x in os.environ.get("RT_PROBE_N", "1,2,4,8").split(",")]:
    for part in parts:
        per_rank, lists = [], None
        if part == "lists":
            # phase 1: every rank renders its interleaved share once and measures its tiles; the figures are summed
            lists0 = dm.tile_lists(dm.initial_ownership(W, H, n), n)
            cost, peak = np.zeros(tiles_x * tiles_y, np.uint32), np.zeros(tiles_x * tiles_y, np.uint32)
            for r in range(n):
                ctx = rt.Context(0)                      # a rank is a process with its own context: its own view state
                scene = ctx.commit(so)
                buf = torch.zeros(dm.compact_floats(lists0), device="cuda:0")
                rt.render_device_batch(ctx, scene, cam, rd, warm, 0, buf.data_ptr(), compact=True, stream=st, tile_list=lists0[r])
                ids, c, pk = ctx.tile_costs(with_peaks=True)
                cost[ids], peak[ids] = c, pk
                del scene, ctx
            lists = dm.tile_lists(rt.partition_tiles(W, H, n, cost), n)
            if os.environ.get("RT_PROBE_LISTS"):
                lists = [np.asarray(l, np.uint32) for l in json.load(open(os.environ["RT_PROBE_LISTS"]))["runs"][str(n)]["lists"]["lists"]]
        for r in range(n):
            ctx = rt.Context(0)
            scene = ctx.commit(so)
            if part == "lists":
                buf = torch.zeros(dm.compact_floats(lists), device="cuda:0")
                rt.render_device_batch(ctx, scene, cam, rd, timed, 0, buf.data_ptr(), compact=True, stream=st, tile_list=lists[r], tile_cost=cost[lists[r]], tile_peak=peak[lists[r]])
            else:
                buf = torch.zeros((dm.max_owned_rows(H, 8, n), W, 3), device="cuda:0")
                # warm-up launch of 5 frames (collects tile costs, like bench.py --warmup 5), then the timed launch
                rt.render_device_batch(ctx, scene, cam, rd, warm, 0, buf.data_ptr(), band_first=r, band_stride=n, compact=True, stream=st)
                rt.render_device_batch(ctx, scene, cam, rd, timed, 0, buf.data_ptr(), band_first=r, band_stride=n, compact=True, stream=st)
            per_rank.append(ctx.last_kernel_ms())
            del scene, ctx
        worst, mean = max(per_rank), sum(per_rank) / len(per_rank)
        t1 = t1 or worst
        print("N=%d %-5s: slowest rank %.1f ms, mean %.1f (ranks: %s; spread %+.1f%% / %+.1f%% of the mean) -> %.0f Msamples/s, speed-up %.2fx (efficiency %.0f%%)" % (
            n, part, worst, mean, " ".join("%.0f" % v for v in per_rank), 100 * (min(per_rank) / mean - 1), 100 * (worst / mean - 1),
            W * H * spp * frames / worst / 1e3, t1 / worst, 100 * t1 / worst / n), flush=True)
        if part == "bands":
            lists = [np.array([(b * tiles_x) + x for b in dm.owned_bands(H, 8, r, n) for x in range(tiles_x)], np.uint32) for r in range(n)]
        dump.setdefault(str(n), {})[part] = {"ms": per_rank, "lists": [l.tolist() for l in lists]}
if os.environ.get("RT_PROBE_DUMP"):
    with open(os.environ["RT_PROBE_DUMP"], "w") as f:
        json.dump({"scene": name, "W": W, "H": H, "spp": spp, "frames": frames, "runs": dump}, f)
