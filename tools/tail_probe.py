#!/usr/bin/env python3
"""How long is a frame's longest job when it has a SIMD (nearly) to itself from the start?

One 1920x1080x1024-spp frame takes ~456 ms, the multi-frame launch 207 ms per frame: the frame is as long as its most expensive
pixel.  For its first ~200 ms that pixel's wave shares its SIMD with three busy waves.  This probe renders the K most expensive
tiles of the view TOGETHER WITH enough tiles that cost nothing (sky: cost bit 0 clear) to make 4,096 list entries, with cost hints
that put the K expensive ones first: the launch then has its full grid (a listed launch is sized ceil(tiles / waves per workgroup)
- K tiles alone would be packed into K / 16 workgroups, four heavy waves per SIMD, the opposite of what is asked here; the first
version of this probe did exactly that and its table was discarded), all 4,096 waves ask for a ticket at once, the K long jobs go
to the first K that ask - spread over the CUs - and everything else is over in a moment.  If that launch is much shorter than
the whole frame, CUs of their own for the longest jobs would shorten a frame; if it is as long, nothing would.

    python tools/tail_probe.py [--scene monkey] [--spp 1024] [--tops 16,64,256,1024]
"""
import argparse, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="monkey")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--limit", type=int, default=8)
    ap.add_argument("--tops", default="16,64,256,1024")
    args = ap.parse_args()
    import numpy as np
    import torch
    rt = importlib.import_module("ray-tracer_amd")
    objs, sky = rt.scenes.CONFIG_SCENES[args.scene]()
    W, H = args.width, args.height
    ctx = rt.Context(0)
    scene = ctx.commit(rt.SceneObjects(objs))
    cam, rd = rt.Camera(W, H), rt.RenderData(args.spp, args.limit, True, sky)
    st = torch.cuda.current_stream().cuda_stream
    fr = torch.zeros((H, W, 3), device="cuda:0")
    for i in range(2):
        rt.render_device(ctx, scene, cam, rd, 12345, 0, fr.data_ptr(), stream=st)
    print("whole frame: %.1f ms" % ctx.last_kernel_ms(), flush=True)
    ids, cost, peak = ctx.tile_costs(with_peaks=True)
    order = np.argsort(-(peak.astype(np.int64)), kind="stable")
    free = ids[(cost & 1) == 0]                                    # tiles no ray of which entered a mesh
    print("%d of %d tiles cost nothing worth the name" % (len(free), len(ids)), flush=True)
    for k in [int(x) for x in args.tops.split(",")]:
        heavy = ids[order[:k]].astype(np.uint32)
        pad = free[:max(0, 4096 - k)].astype(np.uint32)
        sel = np.concatenate([heavy, pad])
        hint = np.concatenate([cost[order[:k]], np.zeros(len(pad), np.uint32)]).astype(np.uint32)
        hpeak = np.concatenate([peak[order[:k]], np.zeros(len(pad), np.uint32)]).astype(np.uint32)
        buf = torch.zeros((len(sel) * 192,), device="cuda:0")
        c2 = rt.Context(0)                      # (a context per list: its view state is the list's)
        s2 = c2.commit(rt.SceneObjects(objs))
        for i in range(2):
            rt.render_device(c2, s2, cam, rd, 12345, 0, buf.data_ptr(), stream=st, tile_list=sel, tile_cost=hint, tile_peak=hpeak, compact=True)
        ms = c2.last_kernel_ms()
        print("the %5d most expensive tiles + %4d free ones, full grid: %.1f ms  (the expensive ones' share of the frame's cost %.3f)"
              % (k, len(pad), ms, cost[order[:k]].astype(np.float64).sum() / cost.astype(np.float64).sum()), flush=True)


if __name__ == "__main__":
    main()
