#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (W x H x spp) of the path-tracer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one frame: every rank renders the 8-row bands it
owns (band b belongs to rank b % N) of the frame into a compact HBM buffer; one gather (RCCL over
xGMI) brings the bands to rank 0, which de-interleaves them.  Inputs (scene, camera, previous
frame) are resident in HBM before the timed region; the frame stays in HBM.

The K timed steps are K consecutive PROGRESSIVE frames of the view (frame_num 0..K-1, seeds
12345 + i) - the reference's own main loop, src/main.cu:415-431 - rendered by one multi-frame
launch per rank (rt_render_device_batch, at most 16 frames per launch) and gathered once: every
sample of every frame is traced, each pixel's frames are blended in order, and the image is
bit-identical to K launches (--check verifies it; tests/test_gpu_parity.py).  What the single
launch buys: a frame ends with a few expensive tiles running alone for half its duration, and
the next frame, which has its own random stream, fills the idle GPU meanwhile.  The line also
carries "frame_by_frame" (the same K steps with one launch per step, N = 1 only), and
--frame-by-frame makes that the measured mode.

Scaling.  A pixel's samples are sequential (one RNG stream per pixel, reference
src/raytracer.cu:127-131), so a frame cannot finish before its most expensive pixels have run
their 1024 samples one after another: on the monkey config that critical path is ~80 % of the
single-GPU frame time, and cutting the SAME 1920x1080 frame into N parts cannot
go below it.  The default for N > 1 is therefore WEAK scaling, the shape of BASELINE.json's own
8-GPU configuration (a larger image tiled over the GPUs): the image area grows with N at fixed
aspect ratio and field of view (1920x1080, 2720x1530, 3840x2160, 5440x3060 for N = 1, 2, 4, 8),
spp and bounce limit unchanged, so per-GPU work is constant.  `--scaling strong` keeps
1920x1080 for every N; a weak-scaling line also reports, under "strong_scaling", the time of the
1920x1080 frame cut over the same N GPUs, for the record.

Default workload = BASELINE.json configs[3], the configuration the north-star target is quoted
on and the largest single-GPU one: low_poly_monkey + emissive sphere light + ground sphere,
1920x1080, 1024 spp, 8 bounces (configs[1]/[2] are selectable with --scene).

Besides the contract's keys the JSON line carries
  roofline      the render kernel against the HBM roof (algorithmic bytes: 24 B per pixel per
                frame + the scene once) — honest reading: this workload is
                nowhere near HBM-bound, the fraction is tiny by construction;
  valu          the same kernel against the FP32 vector peak, with algorithmic FLOPs per sample
                from the counting rule of SURVEY.md §8(d) applied to the oracle's counters;
  cpu_baseline  the CPU oracle (a port of the reference's algorithm, oracle/) timed on this
                box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP32_VECTOR_PEAK_TFLOPS = 157.3


def flops_per_sample(st):
    """28*iters + 26*box + 68*tri + 26*sphere_misses + 51*sphere_hits + 105*hits"""
    n = float(st["samples"])
    return (28 * st["bounce_iters"] + 26 * st["box_tests"] + 68 * st["tri_tests"] +
            26 * (st["sphere_tests"] - st["sphere_hits"]) + 51 * st["sphere_hits"] + 105 * st["hits"]) / n


def cpu_baseline(rt, objs, sky, W, H, limit, spp_full, budget_s=12.0):
    """The oracle (det mode, all host threads available to this process) on a bounded sample:
    the full frame at a reduced spp (per-sample work does not depend on spp)."""
    from oracle import binding as B
    B.build()
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # a one-GPU box's CPU share is 16 cores; RT_BENCH_CPU_THREADS overrides
    threads = int(os.environ.get("RT_BENCH_CPU_THREADS", min(threads, 16)))
    sc = B.Scene(objs, B.MATH_DET, rt.scenes.models_dir())
    cam = rt.Camera(W, H).floats()
    t = time.perf_counter()
    sc.render(cam, W, H, 1, limit, sky, nthreads=threads)
    probe = time.perf_counter() - t
    spp = int(max(1, min(spp_full, budget_s / max(probe, 1e-3))))
    t = time.perf_counter()
    _, st = sc.render(cam, W, H, spp, limit, sky, nthreads=threads, with_stats=True)
    dt = time.perf_counter() - t
    return {"value": W * H * spp / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": "%dx%d full frame at %d spp of %d, %d bounces, %.1f s" % (W, H, spp, spp_full, limit, dt)}, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="monkey", choices=["three_sphere", "cube", "monkey"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--limit", type=int, default=8)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = image area grows with N (default), strong = the same WxH frame for every N")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frame-by-frame", action="store_true", help="one launch + one gather per step instead of one multi-frame launch for all steps")
    ap.add_argument("--no-frame-by-frame-leg", action="store_true", help="skip the extra one-launch-per-step measurement (keeps a profile's launches all of one kind)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo + --share-gpu rehearses N ranks on ONE GPU (RCCL refuses duplicate devices)")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses cuda:0 (rehearsal on a one-GPU box)")
    ap.add_argument("--check", action="store_true", help="rank 0 also renders the frame alone and checks the gathered frame equals it bit for bit")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU; the HIP path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    rt = importlib.import_module("ray-tracer_amd")
    dm = importlib.import_module("ray-tracer_amd.distributed")
    W, H, spp, limit = args.width, args.height, args.spp, args.limit
    if world > 1 and args.scaling == "weak":
        # same aspect ratio and camera field of view, world x the pixels (widths kept multiples of 16)
        W = int(round(args.width * world ** 0.5 / 16.0)) * 16
        H = int(round(W * args.height / args.width))
    objs, sky = rt.scenes.CONFIG_SCENES[args.scene]()
    ctx = rt.Context(local_rank)
    so = rt.SceneObjects(objs)
    scene = ctx.commit(so)
    info = scene.info()
    rd = rt.RenderData(spp, limit, True, sky)
    band_rows = 8
    stream = torch.cuda.current_stream().cuda_stream

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    MAX_BATCH = 16

    def measure(W, H, warmup, steps, batched):
        """`warmup` untimed + `steps` timed steps on a W x H image over all ranks.  batched: the steps
        are consecutive progressive frames (frame_num 0, 1, ...; seeds 12345 + i) rendered by ONE
        launch per rank (rt_render_device_batch, at most 16 frames per launch) and gathered once at
        the end; otherwise one launch + one gather per step (each an independent frame 0).  Returns
        the wall time (max over ranks), rank 0's last gathered frame, this rank's per-launch kernel
        times and how many frames each of those launches rendered."""
        cam = rt.Camera(W, H)
        local = torch.zeros((dm.max_owned_rows(H, band_rows, world), W, 3), dtype=torch.float32, device=dev)
        gathered = torch.empty((world,) + tuple(local.shape), dtype=torch.float32, device=dev) if (world > 1 and rank == 0) else None
        kernel_ms, frames_per_launch = [], []

        def run(n, record):
            if batched:
                done = 0
                while done < n:
                    k = min(MAX_BATCH, n - done)
                    rt.render_device_batch(ctx, scene, cam, rd, [12345 + done + i for i in range(k)], done, local.data_ptr(),
                                           band_first=rank, band_stride=world, compact=True, stream=stream)
                    if record:
                        kernel_ms.append(ctx.last_kernel_ms())   # HIP events on the launch stream; waits for the kernel only
                        frames_per_launch.append(k)
                    done += k
                return dm.gather_frame(local, W, H, band_rows, rank, world, dst=0, out=gathered)
            frame = None
            for _ in range(n):
                rt.render_device(ctx, scene, cam, rd, 12345, 0, local.data_ptr(), band_first=rank, band_stride=world, compact=True, stream=stream)
                frame = dm.gather_frame(local, W, H, band_rows, rank, world, dst=0, out=gathered)
                if record:
                    kernel_ms.append(ctx.last_kernel_ms())
                    frames_per_launch.append(1)
            return frame

        if warmup > 0:
            run(warmup, False)
        fence()
        t0 = time.perf_counter()
        frame = run(steps, True)
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, frame, kernel_ms, frames_per_launch, cam

    batched = not args.frame_by_frame
    elapsed, frame, kernel_ms, frames_per_launch, cam = measure(W, H, args.warmup, args.steps, batched)
    per_frame = None
    if batched and rank == 0 and world == 1 and not args.no_frame_by_frame_leg:
        # for the record: the same number of steps with one launch per frame (what `value` was before
        # multi-frame launches existed); not the headline value
        f_elapsed, _, _, _, _ = measure(W, H, 0, args.steps, False)
        per_frame = {"value": W * H * spp * args.steps / f_elapsed / 1e6, "unit": "Msamples/s", "ms_per_step": f_elapsed / args.steps * 1e3,
                     "note": "one launch per step"}
    strong = None
    if world > 1 and args.scaling == "weak":
        # for the record: the SAME args.width x args.height frame cut over the N GPUs; not the headline value
        s_elapsed, _, _, _, _ = measure(args.width, args.height, 1, args.steps, batched)
        strong = {"image": "%dx%d" % (args.width, args.height), "value": args.width * args.height * spp * args.steps / s_elapsed / 1e6,
                  "unit": "Msamples/s", "ms_per_step": s_elapsed / args.steps * 1e3}

    samples_per_step = W * H * spp
    value = samples_per_step * args.steps / elapsed / 1e6

    # per-launch figures for THIS rank's kernel
    my_rows = min(rt.tile_owned_rows(H, band_rows, rank, world), H)
    my_pixels = 0
    for b in dm.owned_bands(H, band_rows, rank, world):
        my_pixels += (min((b + 1) * band_rows, H) - b * band_rows) * W
    avg_kernel_s = sum(kernel_ms) / len(kernel_ms) / 1e3
    avg_frames = sum(frames_per_launch) / float(len(frames_per_launch))
    flat = so.debug_flatten()
    scene_bytes = flat["blob"].nbytes + flat["objects"].nbytes
    algo_bytes = 24.0 * my_pixels * avg_frames + scene_bytes
    achieved_gbs = algo_bytes / avg_kernel_s / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        key = "%s_%dx%d_s%d_l%d_n%d_f%d" % (args.scene, W, H, spp, limit, world, int(round(avg_frames)))
        traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
    roofline = {"bound": "hbm", "kernel": "rt_render_kernel", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms_avg": avg_kernel_s * 1e3, "frames_per_launch": avg_frames,
                "note": "24 B/pixel/frame + scene once; the path is VALU/latency-bound, see 'valu'"
                        + ("; a multi-frame launch writes one plane of per-pixel means per frame, folded into the frame by a small kernel behind it" if batched else "")}

    out = {"metric": "Msamples/sec (WxHxspp) at %dx%d, %d bounces" % (args.width, args.height, limit), "value": value, "unit": "Msamples/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
           "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%s scene, %dx%d, %d spp, %d bounces, antialias on, time_ms 12345 (BASELINE configs[%d])"
                      % (args.scene, W, H, spp, limit, {"three_sphere": 1, "cube": 2, "monkey": 3}[args.scene]),
                      "parallelism": "image bands of 8 rows interleaved over %d GPU(s) + gather to rank 0" % world,
                      "steps": ("consecutive progressive frames (seeds 12345 + i) in one launch per rank of up to %d frames, gathered once" % MAX_BATCH) if batched
                               else "one launch + one gather per step",
                      "image": "%dx%d" % (W, H),
                      "threads_per_block": info["threads_per_block"], "lds_bytes": info["lds_bytes"]},
           "roofline": roofline}
    if strong is not None:
        out["strong_scaling"] = strong
    if per_frame is not None:
        out["frame_by_frame"] = per_frame

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            cb, st = cpu_baseline(rt, objs, sky, W, H, limit, spp)
            out["cpu_baseline"] = cb
            fps = flops_per_sample(st)
            tflops = fps * (my_pixels * spp * avg_frames) / avg_kernel_s / 1e12
            out["valu"] = {"achieved": tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP32_VECTOR_PEAK_TFLOPS,
                           "algorithmic_flops_per_sample": fps,
                           "per_sample": {k: v / float(st["samples"]) for k, v in st.items() if k != "samples"}}
            out["gpu_over_cpu"] = value / cb["value"]
        if frame is not None:
            out["frame_mean"] = float(frame.mean().item())
            if args.check:
                # rank 0 alone, one launch per frame, must give the gathered image bit for bit
                x = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
                y = torch.empty_like(x)
                if batched:
                    for i in range(args.steps):
                        rt.render_device(ctx, scene, cam, rd, 12345 + i, i, y.data_ptr(), d_prev=x.data_ptr() if i else None, stream=stream)
                        x, y = y, x
                else:
                    rt.render_device(ctx, scene, cam, rd, 12345, 0, x.data_ptr(), stream=stream)
                torch.cuda.synchronize()
                out["gathered_equals_single_launch"] = bool(torch.equal(frame.contiguous().view(torch.int32), x.view(torch.int32)))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
