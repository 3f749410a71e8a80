#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (W x H x spp) of the path-tracer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one frame of the workload.  The K timed steps are K
consecutive PROGRESSIVE frames of the view (frame_num 0..K-1, seeds 12345 + i) - the reference's own
main loop, src/main.cu:415-431 - rendered by multi-frame launches (rt_render_device_batch, at most 32
frames per launch): every sample of every frame is traced, each pixel's frames are blended in order,
and the image is bit-identical to K launches (--check verifies it; tests/test_gpu_parity.py).  What
the single launch buys: a frame ends with a few expensive tiles running alone for half its duration,
and the next frame, which has its own random stream, fills the idle GPU meanwhile.  The line also
carries "frame_by_frame" (the same K steps with one launch per step, N = 1 only), and
--frame-by-frame makes that the measured mode.

Workloads (--config, BASELINE.json configs[i]; --scene/--width/--height/--spp/--limit override):
    1  three-sphere Lambertian + sky, 1920x1080, 1024 spp, 8 bounces
    2  models/cube.obj + ground sphere, 1920x1080, 1024 spp, 8 bounces
    3  models/low_poly_monkey.obj + emissive sphere light, 1920x1080, 1024 spp, 8 bounces   (default: the
       configuration the metric and the north-star target are quoted on)
    4  the monkey scene, 3840x2160, 4096 spp, 8 bounces (the 8-GPU configuration; runs on any N)

N GPUs.  One process per GPU; every rank renders the 8-row bands it owns (band b belongs to rank b % N)
of every frame into a compact HBM buffer, one gather (RCCL over xGMI) brings the bands to rank 0, which
de-interleaves them.  The headline for every N is STRONG scaling of the configured image (1920x1080 for
the metric): K frames cut over N GPUs, each rank rendering its share of all K frames in one launch.  A
single 1080p frame cannot scale (a pixel's 1024 samples are one sequential random stream and the most
expensive tile alone takes ~95 % of a one-GPU frame), a sequence of frames can: each GPU
overlaps its K x tiles/N tile-frames.  For N > 1 the line also carries, as extras, the weak-scaling
measurement (image area grows with N at fixed aspect and field of view) and BASELINE configs[4]
(3840x2160, 4096 spp).  When WORLD_SIZE is not set and N > 1 this script starts its own N ranks
(torch.distributed.run) before touching the GPU and relays rank 0's line.

Besides the contract's keys the JSON line carries
  roofline      the render kernel against the HBM roof (algorithmic bytes: 24 B per pixel per frame + the
                scene once), with the rocprofv3-measured HBM bytes of the same launch
                shape (profiles/traffic.json) and the GB/s they amount to - honest reading: this
                workload is nowhere near HBM-bound, the fraction is tiny by construction;
  valu          the same kernel against the FP32 vector peak, with algorithmic FLOPs per sample from
                the counting rule of SURVEY.md §8(d) applied to the oracle's counters;
  cpu_baseline  the CPU oracle (a port of the reference's algorithm, oracle/) timed on this box's host
                cores on a bounded sample of the same workload (rank 0, N=1 only), all cores and one.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP32_VECTOR_PEAK_TFLOPS = 157.3
MAX_BATCH = 32                 # RT_MAX_BATCH_FRAMES

CONFIGS = {1: ("three_sphere", 1920, 1080, 1024, 8), 2: ("cube", 1920, 1080, 1024, 8),
           3: ("monkey", 1920, 1080, 1024, 8), 4: ("monkey", 3840, 2160, 4096, 8)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json configs[i] (default 3: the metric's configuration)")
    ap.add_argument("--scene", default=None, choices=["three_sphere", "cube", "monkey"])
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--limit", type=int, default=None)
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the same WxH frames for every N (default, the metric's 1920x1080), weak = image area grows with N")
    ap.add_argument("--no-extras", action="store_true", help="N > 1: skip the weak-scaling and configs[4] side measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frame-by-frame", action="store_true", help="one launch + one gather per step instead of multi-frame launches")
    ap.add_argument("--no-frame-by-frame-leg", action="store_true", help="skip the extra one-launch-per-step measurement (keeps a profile's launches all of one kind)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo + --share-gpu rehearses N ranks on ONE GPU (RCCL refuses duplicate devices)")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses cuda:0 (rehearsal on a one-GPU box)")
    ap.add_argument("--check", action="store_true", help="rank 0 also renders the frames alone and checks the gathered frame equals it bit for bit")
    args = ap.parse_args()
    scene, W, H, spp, limit = CONFIGS[args.config]
    args.scene = args.scene or scene
    args.width = args.width or W
    args.height = args.height or H
    args.spp = args.spp or spp
    args.limit = limit if args.limit is None else args.limit
    return args


def spawn_ranks(n):
    """N > 1 without a launcher: start the N ranks as children of this process - which has not touched the GPU -
    and relay rank 0's line.  (Never re-exec a process that has initialised HIP.)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stderr.write(r.stdout[-4000:])
    sys.exit(r.returncode if r.returncode else (0 if lines else 1))


def flops_per_sample(st):
    """28*iters + 26*box + 68*tri + 26*sphere_misses + 51*sphere_hits + 105*hits"""
    n = float(st["samples"])
    return (28 * st["bounce_iters"] + 26 * st["box_tests"] + 68 * st["tri_tests"] +
            26 * (st["sphere_tests"] - st["sphere_hits"]) + 51 * st["sphere_hits"] + 105 * st["hits"]) / n


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(rt, objs, sky, W, H, limit, spp_full, budget_s=12.0):
    """The oracle (det mode) on a bounded sample of the workload: the full frame at a reduced spp (per-sample
    work does not depend on spp), on all host threads available to this process and on one."""
    from oracle import binding as B
    B.build()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # a one-GPU box's CPU share is 16 cores; RT_BENCH_CPU_THREADS overrides
    threads = int(os.environ.get("RT_BENCH_CPU_THREADS", min(avail, 16)))
    sc = B.Scene(objs, B.MATH_DET, rt.scenes.models_dir())
    cam = rt.Camera(W, H).floats()
    t = time.perf_counter()
    sc.render(cam, W, H, 1, limit, sky, nthreads=threads)
    probe = time.perf_counter() - t
    spp = int(max(1, min(spp_full, budget_s / max(probe, 1e-3))))
    t = time.perf_counter()
    _, st = sc.render(cam, W, H, spp, limit, sky, nthreads=threads, with_stats=True)
    dt = time.perf_counter() - t
    value = W * H * spp / dt / 1e6
    # one thread: a band of rows through the middle of the image (where the geometry is), ~4 s
    rows = max(8, min(H, int(4.0 * value * 1e6 / threads / W) // 8 * 8))
    y0 = max(0, (H - rows) // 2) // 8 * 8
    t = time.perf_counter()
    sc.render(cam, W, H, 1, limit, sky, nthreads=1, y0=y0, y1=y0 + rows)
    dt1 = time.perf_counter() - t
    # the band is not the whole frame's mix of work: scale by what the same band costs with all threads
    t = time.perf_counter()
    sc.render(cam, W, H, 1, limit, sky, nthreads=threads, y0=y0, y1=y0 + rows)
    dtn = time.perf_counter() - t
    single = value * dtn / dt1 if dt1 > 0 else None
    return {"value": value, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": "%dx%d full frame at %d spp of %d, %d bounces, %.1f s" % (W, H, spp, spp_full, limit, dt),
            "cpu_model": cpu_model(), "host_threads_available": avail,
            "single_thread": {"value": single, "unit": "Msamples/s", "cores": 1,
                              "sample": "rows %d..%d at 1 spp on one thread (%.1f s) against the same rows on %d threads (%.2f s), "
                                        "applied to the all-thread figure" % (y0, y0 + rows, dt1, threads, dtn)}}, st


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)

    import importlib
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU; the HIP path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (%d visible); --share-gpu --backend gloo rehearses on one" % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend_note = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=dev)
                probe = torch.zeros(1, device=dev)
                dist.all_reduce(probe)                      # RCCL builds its communicator here, not at init
                torch.cuda.synchronize()
            except Exception as e:                          # noqa: BLE001  (a measurement beats no measurement: say so in the line)
                backend_note = "nccl (RCCL) failed to initialise (%s: %s); the gather went through gloo and host memory" % (type(e).__name__, str(e)[:200])
                try:
                    dist.destroy_process_group()
                except Exception:                           # noqa: BLE001
                    pass
                args.backend = "gloo"
                dist.init_process_group("gloo")
        else:
            dist.init_process_group("gloo")

    rt = importlib.import_module("ray-tracer_amd")
    dm = importlib.import_module("ray-tracer_amd.distributed")
    objs, sky = rt.scenes.CONFIG_SCENES[args.scene]()
    ctx = rt.Context(local_rank)
    so = rt.SceneObjects(objs)
    scene = ctx.commit(so)
    info = scene.info()
    band_rows = 8
    stream = torch.cuda.current_stream().cuda_stream

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(W, H, spp, limit, warmup, steps, batched):
        """`warmup` untimed + `steps` timed steps on a W x H image over all ranks.  batched: the steps are
        consecutive progressive frames (frame_num 0, 1, ...; seeds 12345 + i) rendered by multi-frame
        launches (rt_render_device_batch, at most MAX_BATCH frames per launch, launches of equal size) and
        gathered once at the end; otherwise one launch + one gather per step (each an independent frame 0).
        Returns the wall time (max over ranks), rank 0's gathered frame, this rank's per-launch kernel times
        and how many frames each of those launches rendered."""
        cam = rt.Camera(W, H)
        rd = rt.RenderData(spp, limit, True, sky)
        local = torch.zeros((dm.max_owned_rows(H, band_rows, world), W, 3), dtype=torch.float32, device=dev)
        gathered = torch.empty((world,) + tuple(local.shape), dtype=torch.float32, device=dev) if (world > 1 and rank == 0) else None
        kernel_ms, frames_per_launch = [], []

        def run(n, record):
            if batched:
                launches = (n + MAX_BATCH - 1) // MAX_BATCH
                done = 0
                for i in range(launches):
                    k = (n - done + (launches - i) - 1) // (launches - i)
                    rt.render_device_batch(ctx, scene, cam, rd, [12345 + done + j for j in range(k)], done, local.data_ptr(),
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
            if args.backend == "gloo":
                t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, frame, kernel_ms, frames_per_launch, cam, rd

    W, H, spp, limit = args.width, args.height, args.spp, args.limit
    if world > 1 and args.scaling == "weak":
        W, H = weak_image(args.width, args.height, world)
    batched = not args.frame_by_frame
    elapsed, frame, kernel_ms, frames_per_launch, cam, rd = measure(W, H, spp, limit, args.warmup, args.steps, batched)
    per_frame = None
    if batched and rank == 0 and world == 1 and not args.no_frame_by_frame_leg:
        # for the record: the same number of steps with one launch per frame (a single 1920x1080x1024spp render is
        # this figure); not the headline value
        f_elapsed = measure(W, H, spp, limit, 0, args.steps, False)[0]
        per_frame = {"value": W * H * spp * args.steps / f_elapsed / 1e6, "unit": "Msamples/s", "ms_per_step": f_elapsed / args.steps * 1e3,
                     "note": "one launch per step"}
    extras = {}
    if world > 1 and not args.no_extras:
        if args.scaling != "weak":
            # side figure: weak scaling (image area grows with N at the same aspect ratio and field of view)
            w_W, w_H = weak_image(args.width, args.height, world)
            w_el = measure(w_W, w_H, spp, limit, 1, args.steps, batched)[0]
            extras["weak_scaling"] = {"image": "%dx%d" % (w_W, w_H), "spp": spp, "steps": args.steps, "value": w_W * w_H * spp * args.steps / w_el / 1e6,
                                      "unit": "Msamples/s", "ms_per_step": w_el / args.steps * 1e3}
        if args.config != 4:
            # side figure: BASELINE configs[4] (3840x2160, 4096 spp), two progressive frames in one launch per rank
            _, c_W, c_H, c_spp, c_limit = CONFIGS[4]
            if args.scene == "monkey":
                c_el = measure(c_W, c_H, c_spp if args.spp == CONFIGS[args.config][3] else args.spp, c_limit, 0, 2, True)[0]
                c_spp_used = c_spp if args.spp == CONFIGS[args.config][3] else args.spp
                extras["config4"] = {"workload": "monkey scene, %dx%d, %d spp, %d bounces (BASELINE configs[4])" % (c_W, c_H, c_spp_used, c_limit),
                                     "steps": 2, "warmup": 0, "value": c_W * c_H * c_spp_used * 2 / c_el / 1e6, "unit": "Msamples/s", "ms_per_step": c_el / 2 * 1e3}

    samples_per_step = W * H * spp
    value = samples_per_step * args.steps / elapsed / 1e6

    # per-launch figures for THIS rank's kernel
    my_pixels = 0
    for b in dm.owned_bands(H, band_rows, rank, world):
        my_pixels += (min((b + 1) * band_rows, H) - b * band_rows) * W
    avg_kernel_s = sum(kernel_ms) / len(kernel_ms) / 1e3
    avg_frames = sum(frames_per_launch) / float(len(frames_per_launch))
    flat = so.debug_flatten()
    scene_bytes = flat["blob"].nbytes + flat["objects"].nbytes
    algo_bytes = 24.0 * my_pixels * avg_frames + scene_bytes
    achieved_gbs = algo_bytes / avg_kernel_s / 1e9
    traffic, traffic_src = measured_traffic(args.scene, W, H, spp, limit, my_pixels, avg_frames, batched)
    roofline = {"bound": "hbm", "kernel": "rt_render_kernel", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                "hbm_measured_gbs": (traffic / avg_kernel_s / 1e9) if traffic else None, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms_avg": avg_kernel_s * 1e3, "frames_per_launch": avg_frames,
                "note": "24 B/pixel/frame + scene once; the path is VALU/latency-bound by construction (no dense contraction, 0.023 B per sample), see 'valu'"
                        + ("; a multi-frame launch writes one plane of per-pixel means per frame, folded into the frame by a small kernel behind it" if batched else "")}

    cfg_idx = [i for i, c in CONFIGS.items() if c == (args.scene, W, H, spp, limit)]
    ranks_info = None
    if world > 1:
        mine = {"rank": rank, "device": torch.cuda.get_device_name(local_rank), "local_rank": local_rank,
                "kernel_ms": [round(x, 3) for x in kernel_ms]}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        ranks_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": allr}
        if backend_note:
            ranks_info["note"] = backend_note
    out = {"metric": "Msamples/sec (WxHxspp) at %dx%d, %d bounces" % (W, H, limit), "value": value, "unit": "Msamples/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
           "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%s scene, %dx%d, %d spp, %d bounces, antialias on, time_ms 12345%s"
                      % (args.scene, W, H, spp, limit, (" (BASELINE configs[%d])" % cfg_idx[0]) if cfg_idx else ""),
                      "parallelism": "image bands of 8 rows interleaved over %d GPU(s) + gather to rank 0" % world,
                      "steps": ("consecutive progressive frames (seeds 12345 + i) in launches of up to %d frames per rank, gathered once" % MAX_BATCH) if batched
                               else "one launch + one gather per step",
                      "image": "%dx%d" % (W, H),
                      "threads_per_block": info["threads_per_block"], "blocks_per_cu": info["blocks_per_cu"], "lds_bytes": info["lds_bytes"]},
           "roofline": roofline}
    if ranks_info is not None:
        out["ranks"] = ranks_info
    if extras:
        out["extras"] = extras
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


def weak_image(width, height, world):
    """same aspect ratio and camera field of view, world x the pixels (widths kept multiples of 16)"""
    W = int(round(width * world ** 0.5 / 16.0)) * 16
    return W, int(round(W * height / width))


def measured_traffic(scene, W, H, spp, limit, my_pixels, frames, batched):
    """HBM bytes of one launch of this shape from profiles/traffic.json: rocprofv3 FETCH_SIZE / WRITE_SIZE passes
    (separate --pmc runs, gfx950 FETCH_SIZE x2 correction) of the same scene / image / spp, stored per frame of a
    launch (a launch of f frames moves fixed + f x per_frame bytes: each frame writes one plane) and scaled here to
    this rank's share of the pixels."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        tj = json.load(f)
    e = tj.get("%s_%dx%d_s%d_l%d" % (scene, W, H, spp, limit))
    if not e:
        return None, None
    share = my_pixels / float(W * H)
    if not batched:
        frames = 1
    exact = e.get("points", {}).get(str(int(frames))) if float(frames).is_integer() else None
    if exact is not None:          # this launch shape was profiled itself
        return share * exact, e.get("profile")
    return share * (e["fixed_bytes"] + frames * e["per_frame_bytes"]), e.get("profile")


if __name__ == "__main__":
    main()
