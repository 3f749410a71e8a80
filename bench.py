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
    ref0  the reference's own default workload (src/main.cu:11, 318-330, src/camera.cu:4-5): its scene 0
       (Cornell box + monkey + mirror sphere), 1000x800, 100 spp x 5 bounces per frame - the only workload the
       reference prints a number for (FPS, src/main.cu:423-428); the line carries "frames_per_s" beside Msamples/s
    --scene soup6k | sphere50k: meshes beyond a CU's LDS (6,000 random triangles; a 50,880-triangle surface)

N GPUs.  One process per GPU.  Every rank owns a list of 8x8 tiles of the image: the warm-up launch runs on an
interleaved ownership and measures what every tile costs, the costs are summed over the ranks (one small
all-reduce), every rank computes the same longest-processing-time-first ownership (rt_partition_tiles), and
the timed launches run on it; one gather (RCCL over xGMI) brings every rank's tiles to rank 0, which puts
them into the frame (rt_tiles_copy_device).  (--partition bands: round 2's static partition, band b of 8 rows
belongs to rank b % N.)  The headline for every N is STRONG scaling of the configured image (1920x1080 for
the metric): K frames cut over N GPUs, each rank rendering its share of all K frames in one launch.  A
single 1080p frame cannot scale (a pixel's 1024 samples are one sequential random stream and the most
expensive tile alone takes ~95 % of a one-GPU frame), a sequence of frames can: each GPU
overlaps its K x tiles/N tile-frames.  For N > 1 the line also carries, as extras, the weak-scaling
measurement (image area grows with N at fixed aspect and field of view), BASELINE configs[4]
(3840x2160, 4096 spp) and "capi_multi": the same K frames through rt_render_multi_device - the C ABI's
one-host-thread path a C++ maintainer of the reference would call - over all visible GPUs while the ranks wait,
timed, with the peer-access status of every GPU pair and a bit-equality check against the gathered frame (it
runs in a child process with a time limit: that path has never seen two GPUs, and its first contact must not
cost the run its headline).
When WORLD_SIZE is not set and N > 1 this script starts its own N ranks (torch.distributed.run) before
touching the GPU and relays rank 0's line.  If RCCL cannot initialise the run FAILS (exit code 3) unless
--allow-gloo-fallback is given.

Besides the contract's keys the JSON line carries
  backend, world_size   the process group that moved the tiles ("none" for N = 1)
  roofline      the render kernel against the HBM roof (algorithmic bytes: 24 B per pixel per frame + the
                scene once), with the rocprofv3-measured HBM bytes of the same launch
                shape (profiles/traffic.json) and the GB/s they amount to - honest reading: this
                workload is nowhere near HBM-bound, the fraction is tiny by construction;
  valu          the same kernel against the FP32 vector peak, with algorithmic FLOPs per sample from
                the counting rule of SURVEY.md §8(d) applied to the oracle's counters;
  cpu_baseline  the CPU oracle (a port of the reference's algorithm, oracle/) timed on this box's host
                cores on a bounded sample of the same workload (rank 0, N=1 only): on the 16-thread share
                of a one-GPU box, on all host threads ("all_threads") and on one.
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
MAX_BATCH = 32                 # RT_MAX_BATCH_FRAMES (rt_max_batch_frames lowers it for images whose planes would not fit)

CONFIGS = {1: ("three_sphere", 1920, 1080, 1024, 8), 2: ("cube", 1920, 1080, 1024, 8),
           3: ("monkey", 1920, 1080, 1024, 8), 4: ("monkey", 3840, 2160, 4096, 8),
           "ref0": ("reference_scene0", 1000, 800, 100, 5)}
SCENES = ["three_sphere", "cube", "monkey", "reference_scene0", "reference_scene1", "reference_scene3", "soup6k", "sphere50k"]
EXIT_NO_RCCL = 3


def config_key(s):
    return s if s == "ref0" else int(s)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=config_key, default=3, choices=[1, 2, 3, 4, "ref0"],
                    help="BASELINE.json configs[i] (default 3: the metric's configuration), or ref0: the reference's own default workload")
    ap.add_argument("--scene", default=None, choices=SCENES)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--limit", type=int, default=None)
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the same WxH frames for every N (default, the metric's 1920x1080), weak = image area grows with N")
    ap.add_argument("--partition", default="lists", choices=["lists", "bands"],
                    help="N > 1: lists = cost-balanced tile lists (default), bands = round 2's static bands (band b -> rank b %% N)")
    ap.add_argument("--no-extras", action="store_true", help="N > 1: skip the weak-scaling, configs[4] and capi_multi side measurements")
    ap.add_argument("--capi-multi", default=None, metavar="DEVICES",
                    help="also time rt_render_multi_device from ONE process over these devices (e.g. 0,0 rehearses two ranks on one GPU); "
                         "N > 1 does it over all visible GPUs unless --no-extras")
    ap.add_argument("--capi-multi-child", action="store_true", help=argparse.SUPPRESS)     # internal: this process IS the capi_multi measurement
    ap.add_argument("--capi-self-test", action="store_true", help=argparse.SUPPRESS)        # internal: the child stops after its 4-spp first contact and prints that frame's hash
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frame-by-frame", action="store_true", help="one launch + one gather per step instead of multi-frame launches")
    ap.add_argument("--no-frame-by-frame-leg", action="store_true", help="skip the extra one-launch-per-step measurement (keeps a profile's launches all of one kind)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo + --share-gpu rehearses N ranks on ONE GPU (RCCL refuses duplicate devices)")
    ap.add_argument("--allow-gloo-fallback", action="store_true", help="if RCCL cannot initialise, measure through gloo + host memory instead of failing")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses cuda:0 (rehearsal on a one-GPU box)")
    ap.add_argument("--check", action="store_true", help="rank 0 also renders the frames alone and checks the gathered frame equals it bit for bit")
    ap.add_argument("--self-test", action="store_true",
                    help="N > 1 (on by default there unless --no-extras): before anything is timed, 2 frames at 4 spp through BOTH multi-GPU drivers "
                         "(process group + gather, and rt_render_multi_device in a child) must equal rank 0 rendering alone, bit for bit")
    ap.add_argument("--no-self-test", action="store_true")
    ap.add_argument("--no-host-group", action="store_true", help=argparse.SUPPRESS)      # rehearsal: behave as if the gloo side group could not be created
    args = ap.parse_args(argv)
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


def cgroup_cpu_quota():
    """cores the container's cgroup grants (cpu.max quota / period), or None: a box can show 256 threads in its affinity
    mask and still be throttled to a 16-core share"""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                return None if txt[0] == "max" else float(txt[0]) / float(txt[1])
            q = float(txt[0])
            return None if q <= 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError, IndexError):
            continue
    return None


def cpu_baseline(rt, objs, sky, W, H, limit, spp_full, budget_s=12.0):
    """The oracle (det mode) on a bounded sample of the workload: the full frame at a reduced spp (per-sample
    work does not depend on spp).  Three figures: on the 16-thread CPU share of a one-GPU box (the headline
    `value`; RT_BENCH_CPU_THREADS overrides), on ALL host threads available to this process ("all_threads",
    what SURVEY.md §8(d) asks for), and on one thread."""
    from oracle import binding as B
    B.build()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = int(os.environ.get("RT_BENCH_CPU_THREADS", min(avail, 16)))
    sc = B.Scene(objs, B.MATH_DET, rt.scenes.models_dir())
    cam = rt.Camera(W, H).floats()

    def timed(nthreads, budget):
        t = time.perf_counter()
        sc.render(cam, W, H, 1, limit, sky, nthreads=nthreads)
        probe = time.perf_counter() - t
        spp = int(max(1, min(spp_full, budget / max(probe, 1e-3))))
        t = time.perf_counter()
        _, st = sc.render(cam, W, H, spp, limit, sky, nthreads=nthreads, with_stats=True)
        dt = time.perf_counter() - t
        return W * H * spp / dt / 1e6, spp, dt, st

    value, spp, dt, st = timed(threads, budget_s)
    out = {"value": value, "unit": "Msamples/s", "cores": threads, "kind": "port",
           "sample": "%dx%d full frame at %d spp of %d, %d bounces, %.1f s" % (W, H, spp, spp_full, limit, dt),
           "cpu_model": cpu_model(), "host_threads_available": avail, "cgroup_cpu_quota_cores": cgroup_cpu_quota()}
    if avail > threads:
        v_all, spp_all, dt_all, _ = timed(avail, budget_s)
        out["all_threads"] = {"value": v_all, "unit": "Msamples/s", "cores": avail,
                              "sample": "%dx%d full frame at %d spp of %d, %d bounces, %.1f s" % (W, H, spp_all, spp_full, limit, dt_all)}
    else:
        out["all_threads"] = {"value": value, "unit": "Msamples/s", "cores": threads, "sample": "the figure above: this process has no more threads"}
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
    out["single_thread"] = {"value": single, "unit": "Msamples/s", "cores": 1,
                            "sample": "rows %d..%d at 1 spp on one thread (%.1f s) against the same rows on %d threads (%.2f s), "
                                      "applied to the %d-thread figure" % (y0, y0 + rows, dt1, threads, dtn, threads)}
    return out, st


def init_process_group(args, dev, world):
    """-> (backend actually used, note or None).  A failed RCCL initialisation ends the run with EXIT_NO_RCCL unless
    --allow-gloo-fallback: a gloo line must never be mistaken for an xGMI measurement."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.backend == "gloo":
        dist.init_process_group("gloo")
        return "gloo", None
    try:
        dist.init_process_group("nccl", device_id=dev)
        probe = torch.zeros(1, device=dev)
        dist.all_reduce(probe)                      # RCCL builds its communicator here, not at init
        torch.cuda.synchronize()
        return "nccl", None
    except Exception as e:                          # noqa: BLE001
        msg = "nccl (RCCL) failed to initialise (%s: %s)" % (type(e).__name__, str(e)[:300])
        if not args.allow_gloo_fallback:
            sys.stderr.write("bench.py: %s; --allow-gloo-fallback measures through gloo and host memory instead\n" % msg)
            sys.stderr.flush()
            os._exit(EXIT_NO_RCCL)                  # every rank fails the same way; no half-initialised group to tear down
        try:
            dist.destroy_process_group()
        except Exception:                           # noqa: BLE001
            pass
        dist.init_process_group("gloo")
        return "gloo", msg + "; the gather went through gloo and host memory"


def main():
    args = parse_args()
    if args.capi_multi_child:
        capi_multi_child(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)

    import importlib
    import numpy as np
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
    backend, backend_note = "none", None
    host_group, host_group_kind = None, "none"
    if world > 1:
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")       # one node by contract: gloo over loopback, whatever the hostname resolves to
        backend, backend_note = init_process_group(args, dev, world)
        # a host-side group for waiting without occupying the GPU (an RCCL barrier is a kernel that spins on every rank's GPU)
        host_group_kind = "gloo"                        # what the ranks wait in while rank 0's child uses their GPUs
        if backend != "gloo":
            try:
                if args.no_host_group:
                    raise RuntimeError("--no-host-group")
                host_group = dist.new_group(backend="gloo")
            except Exception:                           # noqa: BLE001
                # without a host-side group the only way to wait is an RCCL barrier - a kernel spinning on the very GPUs the
                # capi_multi child renders on.  Then that measurement runs AFTER the process group is gone (see below).
                host_group, host_group_kind = None, "none"

    rt = importlib.import_module("ray-tracer_amd")
    dm = importlib.import_module("ray-tracer_amd.distributed")
    objs, sky = rt.scenes.CONFIG_SCENES[args.scene]()
    ctx = rt.Context(local_rank)
    so = rt.SceneObjects(objs)
    scene = ctx.commit(so)
    info = scene.info()
    band_rows = 8
    stream = torch.cuda.current_stream().cuda_stream
    use_lists = world > 1 and args.partition == "lists"

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(W, H, spp, limit, warmup, steps, batched):
        """`warmup` untimed + `steps` timed steps on a W x H image over all ranks.  batched: the steps are
        consecutive progressive frames (frame_num 0, 1, ...; seeds 12345 + i) rendered by multi-frame
        launches (rt_render_device_batch, launches of equal size) and gathered once at the end; otherwise one
        launch + one gather per step (each an independent frame 0).  Tile lists: the warm-up runs on the
        interleaved ownership (at least one frame: it is what measures the tiles), then the ranks agree on the
        cost-balanced ownership.  Returns a dict: the wall time (max over ranks), rank 0's gathered frame, this
        rank's per-launch kernel times and frames per launch, its pixels, the partition."""
        cam = rt.Camera(W, H)
        rd = rt.RenderData(spp, limit, True, sky)
        max_batch = min(MAX_BATCH, ctx.max_batch_frames(W, H))
        kernel_ms, frames_per_launch, gather_ms = [], [], []
        part = {"kind": "tile lists (cost-balanced)" if use_lists else "bands of 8 rows"}

        def run(n, record, lists, hints, local, gathered):
            spec = dict(tile_list=lists[rank], tile_cost=hints[0] if hints else None, tile_peak=hints[1] if hints else None, compact=True) if lists is not None \
                else dict(band_first=rank, band_stride=world, compact=True)

            def collect():
                # the exchange on its own clock (events on the current stream, where dist.gather and the tile scatter are
                # queued): a lost scaling point can then be put down to the gather or to the kernels from the one line
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if lists is not None:
                    f = dm.gather_tiles(local, lists, W, H, rank, world, ctx=ctx, dst=0, out=gathered, stream=stream)
                else:
                    f = dm.gather_frame(local, W, H, band_rows, rank, world, dst=0, out=gathered)
                e1.record()
                if record and world > 1:
                    gather_ms.append((e0, e1))
                return f
            if batched:
                launches = (n + max_batch - 1) // max_batch
                done = 0
                for i in range(launches):
                    k = (n - done + (launches - i) - 1) // (launches - i)
                    rt.render_device_batch(ctx, scene, cam, rd, [12345 + done + j for j in range(k)], done, local.data_ptr(), stream=stream, **spec)
                    if record:
                        kernel_ms.append(ctx.last_kernel_ms())   # HIP events on the launch stream; waits for the kernel only
                        frames_per_launch.append(k)
                    done += k
                return collect()
            frame = None
            for _ in range(n):
                rt.render_device(ctx, scene, cam, rd, 12345, 0, local.data_ptr(), stream=stream, **spec)
                frame = collect()
                if record:
                    kernel_ms.append(ctx.last_kernel_ms())
                    frames_per_launch.append(1)
            return frame

        def buffers(lists):
            if lists is not None:
                local = torch.zeros(dm.compact_floats(lists), dtype=torch.float32, device=dev)
            else:
                local = torch.zeros((dm.max_owned_rows(H, band_rows, world), W, 3), dtype=torch.float32, device=dev)
            gathered = torch.empty((world,) + tuple(local.shape), dtype=torch.float32, device=dev) if (world > 1 and rank == 0) else None
            return local, gathered

        lists = hints = None
        if use_lists:
            lists0 = dm.tile_lists(dm.initial_ownership(W, H, world), world)
            local, gathered = buffers(lists0)
            run(max(warmup, 1), False, lists0, None, local, gathered)
            owner, cost, peak = dm.balanced_ownership(ctx, W, H, lists0, rank, world, device=dev)
            lists = dm.tile_lists(owner, world)
            hints = (cost[lists[rank]], peak[lists[rank]])
            loads = [int(cost[l].astype(np.int64).sum()) for l in lists]
            part.update({"tiles_per_rank": [int(len(l)) for l in lists], "cost_share_per_rank": [round(x / float(max(1, sum(loads))), 4) for x in loads]})
            local, gathered = buffers(lists)
            my_pixels = sum(min(8, W - (int(g) % ((W + 7) // 8)) * 8) * min(8, H - (int(g) // ((W + 7) // 8)) * 8) for g in lists[rank])
        else:
            local, gathered = buffers(None)
            if warmup > 0:
                run(warmup, False, None, None, local, gathered)
            my_pixels = sum((min((b + 1) * band_rows, H) - b * band_rows) * W for b in dm.owned_bands(H, band_rows, rank, world))
        fence()
        t0 = time.perf_counter()
        frame = run(steps, True, lists, hints, local, gathered)
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            if backend == "gloo":
                t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        torch.cuda.synchronize()
        return {"elapsed": elapsed, "frame": frame, "kernel_ms": kernel_ms, "frames_per_launch": frames_per_launch, "cam": cam, "rd": rd,
                "my_pixels": my_pixels, "partition": part, "gather_ms": [a.elapsed_time(b) for a, b in gather_ms]}

    W, H, spp, limit = args.width, args.height, args.spp, args.limit
    self_test = None
    if world > 1 and not args.no_self_test and (args.self_test or not args.no_extras):
        self_test = run_self_test(args, rt, dm, ctx, scene, sky, dev, rank, world, stream, measure, host_group, host_group_kind)
    if world > 1 and args.scaling == "weak":
        W, H = weak_image(args.width, args.height, world)
    batched = not args.frame_by_frame
    m = measure(W, H, spp, limit, args.warmup, args.steps, batched)
    elapsed, frame, kernel_ms, frames_per_launch, cam, rd = m["elapsed"], m["frame"], m["kernel_ms"], m["frames_per_launch"], m["cam"], m["rd"]
    per_frame = None
    if batched and rank == 0 and world == 1 and not args.no_frame_by_frame_leg:
        # for the record: the same number of steps with one launch per frame (a single 1920x1080x1024spp render is
        # this figure); not the headline value
        f_elapsed = measure(W, H, spp, limit, 0, args.steps, False)["elapsed"]
        per_frame = {"value": W * H * spp * args.steps / f_elapsed / 1e6, "unit": "Msamples/s", "ms_per_step": f_elapsed / args.steps * 1e3,
                     "frames_per_s": args.steps / f_elapsed, "note": "one launch per step (the reference's main-loop cadence, src/main.cu:415-431)"}
    pipelined = None
    if batched and rank == 0 and world == 1 and not args.no_frame_by_frame_leg and hasattr(rt, "frame_submit"):
        # one launch per frame through rt_frame_submit / rt_frame_collect: seeds handed over one call at a time (the reference's loop
        # draws them from the wall clock), the next frames submitted before the oldest is waited for, each on 1 / depth of the CUs
        def run_pipelined(depth, n_frames):
            pf = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
            rt.frame_depth(ctx, depth)
            fence()
            t0 = time.perf_counter()
            n_done = 0
            for i in range(n_frames):
                if rt.frames_pending(ctx) == depth:
                    rt.frame_collect(ctx, n_done, pf.data_ptr(), stream=stream)
                    n_done += 1
                    rt.frame_wait(ctx)        # the caller draws frame n_done - 1 here; the younger frames keep the GPU busy
                rt.frame_submit(ctx, scene, cam, rd, 12345 + i)
            while rt.frames_pending(ctx):
                rt.frame_collect(ctx, n_done, pf.data_ptr(), stream=stream)
                n_done += 1
                rt.frame_wait(ctx)
            fence()
            el = time.perf_counter() - t0
            return {"frames_in_flight": depth, "steps": n_frames, "value": W * H * spp * n_frames / el / 1e6, "unit": "Msamples/s",
                    "ms_per_step": el / n_frames * 1e3, "frames_per_s": n_frames / el}, pf
        # (never a deeper pipeline than there are steps: every frame runs on 1 / depth of the CUs, so a depth the run cannot fill idles the rest)
        p4, pf = run_pipelined(max(1, min(rt.PIPELINE_DEFAULT_DEPTH, args.steps)), args.steps)
        pipelined = dict(p4)
        pipelined["equals_batched"] = bool(frame is not None and torch.equal(pf.view(torch.int32), frame.contiguous().view(torch.int32)))
        del pf
        # the deepest pipeline over a run long enough to show its steady state (filling and draining it costs a frame's latency)
        pipelined["deepest"] = run_pipelined(rt.PIPELINE_DEPTH, max(args.steps, 3 * rt.PIPELINE_DEPTH))[0]
        pipelined["note"] = ("one launch per step through rt_frame_submit / rt_frame_collect: each seed handed over at call time, frames_in_flight submitted "
                             "before the oldest is waited for, each on 1 / frames_in_flight of the CUs; the time includes filling and draining the pipeline")
    extras = {}
    if world > 1 and not args.no_extras:
        if args.scaling != "weak":
            # side figure: weak scaling (image area grows with N at the same aspect ratio and field of view)
            w_W, w_H = weak_image(args.width, args.height, world)
            w_el = measure(w_W, w_H, spp, limit, 1, args.steps, batched)["elapsed"]
            extras["weak_scaling"] = {"image": "%dx%d" % (w_W, w_H), "spp": spp, "steps": args.steps, "value": w_W * w_H * spp * args.steps / w_el / 1e6,
                                      "unit": "Msamples/s", "ms_per_step": w_el / args.steps * 1e3}
        if args.config != 4 and args.scene == "monkey":
            # side figure: BASELINE configs[4] (3840x2160, 4096 spp), two progressive frames in one launch per rank
            _, c_W, c_H, c_spp, c_limit = CONFIGS[4]
            c_spp_used = c_spp if args.spp == CONFIGS[3][3] else args.spp
            c_el = measure(c_W, c_H, c_spp_used, c_limit, 0, 2, True)["elapsed"]
            extras["config4"] = {"workload": "monkey scene, %dx%d, %d spp, %d bounces (BASELINE configs[4])" % (c_W, c_H, c_spp_used, c_limit),
                                 "steps": 2, "warmup": 1, "value": c_W * c_H * c_spp_used * 2 / c_el / 1e6, "unit": "Msamples/s", "ms_per_step": c_el / 2 * 1e3}
    # the C ABI's one-thread multi-GPU path, from rank 0 while the other ranks wait (their GPUs are idle)
    capi_devices = None
    if args.capi_multi:
        capi_devices = [int(x) for x in args.capi_multi.split(",")]
    elif world > 1 and not args.no_extras and batched:
        capi_devices = [0] * world if args.share_gpu else list(range(min(world, torch.cuda.device_count())))
    capi_after_group = capi_runs_after_group(bool(capi_devices), world, backend, host_group is not None)
    if capi_devices and not capi_after_group:
        if rank == 0:
            extras["capi_multi"] = capi_multi_with_fallback(args, capi_devices, W, H, spp, limit, frame if batched else None)
        if world > 1:
            dist.barrier(group=host_group)          # gloo: the waiting ranks sleep in a socket, their GPUs are the child's

    samples_per_step = W * H * spp
    value = samples_per_step * args.steps / elapsed / 1e6

    # per-launch figures for THIS rank's kernel
    my_pixels = m["my_pixels"]
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
    cfg_note = ""
    if cfg_idx:
        cfg_note = " (the reference's default workload, src/main.cu:318-330)" if cfg_idx[0] == "ref0" else " (BASELINE configs[%d])" % cfg_idx[0]
    ranks_info = None
    if world > 1:
        mine = {"rank": rank, "device": torch.cuda.get_device_name(local_rank), "local_rank": local_rank,
                "kernel_ms": [round(x, 3) for x in kernel_ms], "gather_ms": [round(x, 3) for x in m["gather_ms"]]}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        ranks_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": allr, "partition": m["partition"]}
        if backend_note:
            ranks_info["note"] = backend_note
    out = {"metric": "Msamples/sec (WxHxspp) at %dx%d, %d bounces" % (W, H, limit), "value": value, "unit": "Msamples/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
           # every N renders the SAME image (strong scaling) unless --scaling weak; an N = 1 line is the curve's first point
           "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "backend": backend, "world_size": world,
           "config": {"workload": "%s scene, %dx%d, %d spp, %d bounces, antialias on, time_ms 12345%s" % (args.scene, W, H, spp, limit, cfg_note),
                      "parallelism": ("%s over %d GPUs + gather to rank 0" % (m["partition"]["kind"], world)) if world > 1 else "one GPU, whole image",
                      "steps": ("consecutive progressive frames (seeds 12345 + i) in launches of up to %d frames per rank, gathered once" % MAX_BATCH) if batched
                               else "one launch + one gather per step",
                      "image": "%dx%d" % (W, H),
                      "threads_per_block": info["threads_per_block"], "blocks_per_cu": info["blocks_per_cu"], "lds_bytes": info["lds_bytes"],
                      "scene_in_lds": info["scene_in_lds"], "triangles": info["num_triangles"]},
           "frames_per_s": args.steps / elapsed,
           "roofline": roofline}
    if ranks_info is not None:
        out["ranks"] = ranks_info
        out["host_group"] = host_group_kind
        g = m["gather_ms"]
        out["gather_ms"] = (sum(g) / len(g)) if g else None        # rank 0: dist.gather + the scatter of every rank's tiles into the frame, per measurement
        out["gather_ms_note"] = "events on rank 0's stream around dist.gather and the tile scatter; includes waiting for the slowest rank's kernel"
    if self_test is not None:
        out["self_test"] = self_test
    if capi_after_group:
        # no host-side group: rank 0 runs the one-thread driver only once the process group (and with it every RCCL kernel) is gone
        frame_ref = frame if batched else None
        dist.destroy_process_group()
        if rank == 0:
            extras["capi_multi"] = capi_multi_with_fallback(args, capi_devices, W, H, spp, limit, frame_ref)
            extras["capi_multi"]["ran"] = "after destroy_process_group (no gloo side group to wait in)"
    if extras:
        out["extras"] = extras
    if per_frame is not None:
        out["frame_by_frame"] = per_frame
    if pipelined is not None:
        out["pipelined"] = pipelined

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
            out["gpu_over_cpu_all_threads"] = value / cb["all_threads"]["value"]
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
    if world > 1 and not capi_after_group:
        dist.destroy_process_group()


def capi_runs_after_group(have_devices, world, backend, have_host_group):
    """When does rank 0 run the one-host-thread driver (rt_render_multi_device in a child, on ALL GPUs)?  While the other ranks
    wait for it - which must not occupy their GPUs: a gloo barrier sleeps in a socket, an RCCL barrier is a kernel spinning
    on every GPU.  So: under the barrier if the waiting is host-side (a gloo process group, or the gloo side group beside RCCL);
    otherwise only after destroy_process_group, when no collective can be running any more.  -> True for the latter."""
    return bool(have_devices) and world > 1 and backend != "gloo" and not have_host_group


def run_self_test(args, rt, dm, ctx, scene, sky, dev, rank, world, stream, measure, host_group, host_group_kind):
    """Before anything is timed: 2 progressive frames at 4 samples per pixel of the run's image through BOTH multi-GPU drivers -
    this process group (tile lists or bands + dist.gather) and the C ABI's rt_render_multi_device in a child of rank 0 - against
    rank 0 rendering the two frames alone.  Seconds of work; a wrong or hanging exchange is then known before the long legs,
    and the line says which driver it was."""
    import hashlib
    import torch
    import torch.distributed as dist
    W, H, limit = args.width, args.height, args.limit
    res = {"workload": "2 frames at 4 spp of the run's image", "process_group": None, "capi_multi": None}
    m = measure(W, H, 4, limit, 1, 2, True)
    sha = None
    if rank == 0:
        cam, rd = rt.Camera(W, H), rt.RenderData(4, limit, True, sky)
        alone = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        rt.render_device_batch(ctx, scene, cam, rd, [12345, 12346], 0, alone.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        res["process_group"] = bool(torch.equal(m["frame"].contiguous().view(torch.int32), alone.view(torch.int32)))
        sha = hashlib.sha256(alone.cpu().numpy().tobytes()).hexdigest()
    if host_group_kind == "gloo" or dist.get_backend() == "gloo":
        # (without a host-side group the child would share its GPUs with a spinning RCCL barrier: the C ABI's leg of the
        # self-test is then left to the timed capi_multi run, which checks its frame as well)
        if rank == 0:
            devices = [0] * world if args.share_gpu else list(range(min(world, torch.cuda.device_count())))
            r = capi_multi(args, devices, W, H, 4, limit, None, timeout_s=120, self_test=True)
            res["capi_multi"] = (r.get("frame_sha256") == sha) if "error" not in r else r
        dist.barrier(group=host_group)
    else:
        res["capi_multi"] = "skipped: no host-side group (checked by the timed capi_multi run instead)"
    return res


def capi_multi(args, devices, W, H, spp, limit, reference_frame, timeout_s=300, self_test=False):
    """`steps` progressive frames through rt_render_multi_device (include/rt_amd.h; replaces run_ray_tracer
    src/dispatch.cu:127-163 for a node) over `devices` from ONE host thread - in a CHILD process with a time limit: the
    peer-copy path has never run on more than one GPU, and whatever it does on its first real node (an exception, a
    hang) must not cost the run its headline line.  The child prints a JSON object with the sha256 of its frame, which
    must be the gathered frame's (the torch.distributed run's)."""
    import hashlib
    cmd = [sys.executable, os.path.abspath(__file__), "--capi-multi-child", "--capi-multi", ",".join(str(d) for d in devices), "--scene", args.scene,
           "--width", str(W), "--height", str(H), "--spp", str(spp), "--limit", str(limit), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    if self_test:
        cmd.append("--capi-self-test")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK")}
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": "no answer within %d s (child killed)" % timeout_s, "devices": devices}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        err = {"error": "child exit code %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:]), "devices": devices}
        if lines:                                   # the child's own account: phase, every rank's rt_last_error
            err.update(json.loads(lines[-1]))
        return err
    out = json.loads(lines[-1])
    if reference_frame is not None:
        out["equals_gathered_frame"] = out.pop("frame_sha256") == hashlib.sha256(reference_frame.contiguous().cpu().numpy().tobytes()).hexdigest()
    return out


def capi_multi_with_fallback(args, devices, W, H, spp, limit, reference_frame):
    """capi_multi; if the asynchronous form fails, hangs or renders another frame than the gathered one, once more with
    RT_AMD_MULTI_CAREFUL=1 (the host waits after every phase of rt_render_multi_device, no ordering between devices rests on an
    event): the pair of outcomes says whether a fault is in the event choreography or in the copies."""
    first = capi_multi(args, devices, W, H, spp, limit, reference_frame)
    if "error" not in first and first.get("equals_gathered_frame", True):
        return first
    os.environ["RT_AMD_MULTI_CAREFUL"] = "1"
    try:
        second = capi_multi(args, devices, W, H, spp, limit, reference_frame, timeout_s=240)
    finally:
        os.environ.pop("RT_AMD_MULTI_CAREFUL", None)
    first["careful_mode"] = second
    first["careful_mode"]["note"] = "RT_AMD_MULTI_CAREFUL=1: the host waits for every stream after each phase"
    return first


def capi_multi_child(args):
    """the measurement itself (see capi_multi): one context per entry of --capi-multi, cost-balanced tile lists, peer copies to
    the first device.  Two untimed calls first (the view's first call measures the tiles on an interleaved ownership, the
    second deals them out by cost), then the timed call renders frames 0..steps-1 into a fresh frame."""
    import hashlib
    import importlib
    import torch
    rt = importlib.import_module("ray-tracer_amd")
    devices = [int(x) for x in args.capi_multi.split(",")]
    W, H, spp, limit, steps = args.width, args.height, args.spp, args.limit, args.steps
    objs, sky = rt.scenes.CONFIG_SCENES[args.scene]()
    so = rt.SceneObjects(objs)
    ctxs = [rt.Context(d) for d in devices]
    scenes = [c.commit(so) for c in ctxs]
    cam, rd = rt.Camera(W, H), rt.RenderData(spp, limit, True, sky)
    root = torch.device("cuda", devices[0])
    phase = "start"

    def fail(e):
        # which call was being made, and what every rank's context last complained about (rt_last_error names the step that was
        # being queued: scatter-out, launch, peer copy, de-interleave)
        print(json.dumps({"error": "%s: %s" % (type(e).__name__, str(e)[:300]), "phase": phase, "devices": devices,
                          "last_errors": [c.last_error() for c in ctxs]}), flush=True)
        sys.exit(4)

    with torch.cuda.device(root):
        s0 = torch.cuda.current_stream(root).cuda_stream
        scratch = torch.zeros((H, W, 3), dtype=torch.float32, device=root)
        frame = torch.zeros((H, W, 3), dtype=torch.float32, device=root)

        def sync():
            torch.cuda.synchronize(root)
            for c in ctxs:
                c.synchronize()
        try:
            # first contact at 4 samples per pixel: a hang or a fault on the peer-copy path shows in seconds, not at the time limit
            phase = "first contact: 2 frames at 4 spp (measuring call, interleaved ownership)"
            rd4 = rt.RenderData(4, limit, True, sky)
            rt.render_multi_device(ctxs, scenes, cam, rd4, [12345, 12346], 0, scratch.data_ptr(), stream=s0)
            sync()
            phase = "first contact: 2 frames at 4 spp (balanced ownership)"
            rt.render_multi_device(ctxs, scenes, cam, rd4, [12345, 12346], 0, scratch.data_ptr(), stream=s0)
            sync()
            small_sha = hashlib.sha256(scratch.cpu().numpy().tobytes()).hexdigest()
            if args.capi_self_test:
                print(json.dumps({"devices": devices, "frame_sha256": small_sha, "phase": "self-test"}), flush=True)
                return
            phase = "warm-up at full sample count (measuring call)"
            rt.render_multi_device(ctxs, scenes, cam, rd, [12345 + i for i in range(max(1, args.warmup))], 0, scratch.data_ptr(), stream=s0)
            phase = "warm-up at full sample count (balanced ownership)"
            rt.render_multi_device(ctxs, scenes, cam, rd, [12345], 0, scratch.data_ptr(), stream=s0)
            sync()
            phase = "timed call"
            t0 = time.perf_counter()
            rt.render_multi_device(ctxs, scenes, cam, rd, [12345 + i for i in range(steps)], 0, frame.data_ptr(), stream=s0)
            sync()
            elapsed = time.perf_counter() - t0
        except Exception as e:                      # noqa: BLE001
            fail(e)
        out = {"entry": "rt_render_multi_device (one host thread in a child process, cost-balanced tile lists, peer copies to the first device)",
               "devices": devices, "steps": steps, "value": W * H * spp * steps / elapsed / 1e6, "unit": "Msamples/s", "ms_per_step": elapsed / steps * 1e3,
               "kernel_ms_per_rank": [round(c.last_kernel_ms(), 3) for c in ctxs],
               "p2p": [{"pair": [devices[0], d], "direct": bool(ctxs[0].peer_access(c) == 1)} for d, c in list(zip(devices, ctxs))[1:]],
               "frame_sha256": hashlib.sha256(frame.cpu().numpy().tobytes()).hexdigest()}
    print(json.dumps(out), flush=True)


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
