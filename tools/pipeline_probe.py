#!/usr/bin/env python3
"""How much of the one-launch-per-frame tail does overlapping the launches of consecutive frames recover?

A frame ends with a few expensive tiles running alone for half its duration ("One frame").  The multi-frame
launch fills that time with the next frames but needs their seeds in advance; the reference's loop draws each seed from the
wall clock when it calls render() (src/main.cu:18-25, 415-431).  A caller that submits frame k + 1 (seed drawn at call time)
before waiting for frame k gets the same overlap from the hardware if the launches can run side by side.  This probe measures
that through rt_frame_submit / rt_frame_collect with rt_frame_depth = 1 .. RT_PIPELINE_DEPTH frames in flight (1 = one launch at a time; each frame runs on 1 / depth of the CUs); the host
waits for every collected frame as a caller that draws it would.  (The first version of this probe used D contexts on D
streams: profiles/r04/experiments/pipelined_frames.txt.)

    python tools/pipeline_probe.py [--scene monkey] [--spp 1024] [--frames 20] [--depths 1,2,3,4,6,8]
"""
import argparse
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="monkey")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--limit", type=int, default=8)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--depths", default="1,2,3,4,6,8")
    ap.add_argument("--host", action="store_true", help="host-buffer form (rt_frame_collect_host: upload of the previous image, fold, download per frame)")
    args = ap.parse_args()
    import torch
    rt = importlib.import_module("ray-tracer_amd")
    objs, sky = rt.scenes.CONFIG_SCENES[args.scene]()
    dev = torch.device("cuda", 0)
    cam = rt.Camera(args.width, args.height)
    rd = rt.RenderData(args.spp, args.limit, True, sky)
    samples = args.width * args.height * args.spp * args.frames
    ctx = rt.Context(0)
    scene = ctx.commit(rt.SceneObjects(objs))
    st = torch.cuda.current_stream().cuda_stream
    fr = torch.zeros((args.height, args.width, 3), dtype=torch.float32, device=dev)
    for i in range(2):                                                 # the view's first frames measure the tiles and sort the schedule
        rt.frame_submit(ctx, scene, cam, rd, 777 + i)
        rt.frame_collect(ctx, i, fr.data_ptr(), stream=st)
    torch.cuda.synchronize()
    data = rt.VariableRenderData(args.width, args.height)
    for depth in [int(x) for x in args.depths.split(",")]:
        rt.frame_depth(ctx, depth)
        if args.host:
            data.frame_num = 0
            t0 = time.perf_counter()
            sent = 0
            while data.frame_num < args.frames:
                while sent < args.frames and rt.frames_pending(ctx) < depth:
                    rt.frame_submit(ctx, scene, cam, rd, 12345 + sent); sent += 1
                rt.frame_collect_host(ctx, data)
            el = time.perf_counter() - t0
            print("in flight %d  %8.1f ms per frame  %8.0f Msamples/s   (host buffers: 2 x %.1f MB over PCIe per frame)" % (depth, el / args.frames * 1e3, samples / el / 1e6, data.previous_render.nbytes / 1e6), flush=True)
            continue
        t0 = time.perf_counter()
        n = 0
        for i in range(args.frames):
            if rt.frames_pending(ctx) == depth:
                rt.frame_collect(ctx, n, fr.data_ptr(), stream=st); n += 1
                rt.frame_wait(ctx)              # (the caller draws the frame)
            rt.frame_submit(ctx, scene, cam, rd, 12345 + i)
        while rt.frames_pending(ctx):
            rt.frame_collect(ctx, n, fr.data_ptr(), stream=st); n += 1
            rt.frame_wait(ctx)
        el = time.perf_counter() - t0
        print("in flight %d  %8.1f ms per frame  %8.0f Msamples/s" % (depth, el / args.frames * 1e3, samples / el / 1e6), flush=True)


if __name__ == "__main__":
    main()
