"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle's det mode
on the same seeded inputs, bit-exact (float32 framebuffers compared as uint32), plus
size-independent properties at BASELINE.json's full image size.

Tolerance: none.  The north star allows L-inf < 1e-4; a path tracer cannot meet a tolerance
like that by being "close" (one flipped hit/miss decision moves a pixel by ~albedo/spp), so the
bar here is bit-identical results."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def eq(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32))


def hip_render(rt, ctx, objs, W, H, spp, limit, sky, time_ms=12345, antialias=True, frame_num=0, prev=None):
    scene = ctx.commit(rt.SceneObjects(objs))
    data = rt.VariableRenderData(W, H)
    data.frame_num = frame_num
    if prev is not None:
        data.previous_render[...] = prev
    rt.render(ctx, scene, rt.Camera(W, H), rt.RenderData(spp, limit, antialias, sky), data, time_ms)
    return data.previous_render.copy()


CASES = [
    # scene, W, H, spp, limit, antialias
    ("three_sphere", 256, 256, 16, 4, True),       # BASELINE configs[0]
    ("three_sphere", 250, 203, 5, 8, True),        # ragged: neither dimension a multiple of 8
    ("three_sphere", 64, 40, 8, 8, False),         # antialias off (src/ray.cu:131)
    ("cube", 256, 256, 16, 8, True),
    ("monkey", 256, 256, 16, 8, True),
    ("monkey", 320, 180, 6, 8, True),              # 16:9 like the headline config
    ("monkey", 57, 33, 3, 2, False),
    ("reference_scene0", 250, 200, 8, 5, True),    # the reference's own default scene, its aspect ratio
    ("reference_scene1", 250, 200, 8, 5, True),
    ("reference_scene2", 250, 200, 8, 5, True),    # image-textured sphere (UVs via asin/acos) + checkerboard triangle
    ("reference_scene3", 250, 200, 8, 5, True),    # refractive sphere: Snell, Schlick, total internal reflection
    ("reference_scene3", 128, 96, 4, 12, False),
    ("reference_scene4", 320, 180, 6, 8, True),    # 100 spheres (standard / refractive / default) on a checkerboard quad
]


@pytest.mark.parametrize("name,W,H,spp,limit,aa", CASES, ids=["%s-%dx%d-s%d-l%d-%s" % (c[0], c[1], c[2], c[3], c[4], "aa" if c[5] else "noaa") for c in CASES])
def test_hip_equals_oracle(rt, orc, ctx, models_dir, name, W, H, spp, limit, aa):
    objs, sky = rt.scenes.CONFIG_SCENES[name]()
    got = hip_render(rt, ctx, objs, W, H, spp, limit, sky, time_ms=987654321, antialias=aa)
    want = orc.Scene(objs, orc.MATH_DET, models_dir).render(rt.Camera(W, H).floats(), W, H, spp, limit, sky, time_ms=987654321, antialias=aa)
    assert eq(got, want)
    assert np.isfinite(got).all()


@pytest.mark.parametrize("spp,limit", [(1, 1), (3, 0), (2, 1), (1, 8)])
def test_degenerate_settings(rt, orc, ctx, models_dir, spp, limit):
    objs, sky = rt.scenes.monkey()
    got = hip_render(rt, ctx, objs, 72, 48, spp, limit, sky)
    want = orc.Scene(objs, orc.MATH_DET, models_dir).render(rt.Camera(72, 48).floats(), 72, 48, spp, limit, sky)
    assert eq(got, want)


def test_negative_time_seed_wraps(rt, orc, ctx, models_dir):
    """get_time() truncates ms-since-epoch into an int that is usually negative
    (src/main.cu:18-25); the seed arithmetic wraps (src/raytracer.cu:127)"""
    objs, sky = rt.scenes.three_sphere()
    t = -1234567890
    got = hip_render(rt, ctx, objs, 64, 64, 4, 4, sky, time_ms=t)
    want = orc.Scene(objs, orc.MATH_DET, models_dir).render(rt.Camera(64, 64).floats(), 64, 64, 4, 4, sky, time_ms=t)
    assert eq(got, want)


def test_progressive_accumulation(rt, orc, ctx, models_dir):
    objs, sky = rt.scenes.cube()
    W, H = 80, 56
    o = orc.Scene(objs, orc.MATH_DET, models_dir)
    cam = rt.Camera(W, H).floats()
    prev_g = prev_o = None
    for frame, t in enumerate((5, 6, 7)):
        prev_g = hip_render(rt, ctx, objs, W, H, 3, 8, sky, time_ms=t, frame_num=frame, prev=prev_g)
        prev_o = o.render(cam, W, H, 3, 8, sky, time_ms=t, frame_num=frame, prev=prev_o)
        assert eq(prev_g, prev_o)


def test_top_level_primitives_and_tie_rules(rt, orc, ctx, models_dir):
    """quads (t1-first rule), one-way quad culling, cuboid (strict <), top-level triangle with
    UVs + gradient / checkerboard textures, coincident objects (later object wins, `<=`)"""
    objs = [
        ("quad", (-1, -0.5, 1), (1, -0.5, 1), (1, -0.5, 3), (-1, -0.5, 3), ("checkerboard", (0.9, 0.9, 0.9), (0.2, 0.2, 0.2), 6, 0.1)),
        ("one_way_quad", (-1, 1, 0.5), (1, 1, 0.5), (1, -1, 0.5), (-1, -1, 0.5), False, ("standard", (1, 1, 1), 0)),
        ("one_way_quad", (-1, 1, 3.2), (1, 1, 3.2), (1, -1, 3.2), (-1, -1, 3.2), True, ("standard", (0.4, 0.8, 0.4), 0)),
        ("cuboid", (-0.3, 0.3, 1.6), 0.6, 0.5, 0.4, ("standard", (0.8, 0.3, 0.3), 0.5)),
        ("triangle_uv", [(-0.9, 0.9, 2.5), (0.9, 0.9, 2.5), (0.0, -0.2, 2.0)], [(0, 0), (1, 0), (0.5, 1)], ("gradient", 0)),
        ("triangle", (-0.9, 0.9, 2.5), (0.9, 0.9, 2.5), (0.0, -0.2, 2.0), ("standard", (0.2, 0.2, 0.9), 0)),    # coincident: wins the tie
        ("sphere", (0.5, -0.2, 1.4), 0.2, ("emissive", (1, 0.9, 0.8), 4)),
        ("sphere", (0.5, -0.2, 1.4), 0.2, ("standard", (0.5, 0.5, 0.5), 1)),                                      # coincident sphere: wins
    ]
    sky = (0.8, 1.0, 1.0)
    got = hip_render(rt, ctx, objs, 160, 120, 8, 6, sky)
    want = orc.Scene(objs, orc.MATH_DET, models_dir).render(rt.Camera(160, 120).floats(), 160, 120, 8, 6, sky)
    assert eq(got, want)
    assert got.std() > 0.05


def test_many_meshes_and_spheres(rt, orc, ctx, models_dir):
    """several BVH meshes in one scene (the stack is reused per mesh) + a 40-sphere field"""
    rng = np.random.default_rng(3)
    objs = [("obj", "cube.obj", [("enlarge", 0.2), ("rotate", 0.3 * k, 0.5, 0.1 * k), ("translate", -0.8 + 0.8 * k, 0.1, 2.0)], ("standard", (0.8, 0.5 + 0.1 * k, 0.2), 0.2 * k)) for k in range(3)]
    for _ in range(40):
        c = (float(rng.uniform(-2, 2)), float(rng.uniform(-0.6, 0.8)), float(rng.uniform(1.2, 5)))
        objs.append(("sphere", c, float(rng.uniform(0.05, 0.2)), ("standard", tuple(float(x) for x in rng.uniform(0.2, 1, 3)), float(rng.uniform(0, 1)))))
    objs.append(("sphere", (0, -100.5, 1.5), 100, ("standard", (0.5, 0.5, 0.5), 0)))
    got = hip_render(rt, ctx, objs, 128, 96, 6, 8, (0.8, 1.0, 1.0))
    want = orc.Scene(objs, orc.MATH_DET, models_dir).render(rt.Camera(128, 96).floats(), 128, 96, 6, 8, (0.8, 1.0, 1.0))
    assert eq(got, want)


def test_empty_scene_is_sky(rt, ctx):
    got = hip_render(rt, ctx, [], 40, 24, 2, 3, (0.8, 1.0, 1.0))
    assert eq(got, np.broadcast_to(np.array([0.8, 1.0, 1.0], np.float32), (24, 40, 3)))


def test_repeatability(rt, ctx):
    objs, sky = rt.scenes.monkey()
    a = hip_render(rt, ctx, objs, 200, 120, 8, 8, sky)
    b = hip_render(rt, ctx, objs, 200, 120, 8, 8, sky)
    assert eq(a, b)


def test_device_api_tiles_reassemble(rt, ctx):
    """rt_render_device with band_first/band_stride/compact (the multi-GPU partition) writes
    exactly the rows it owns, and the parts reassemble to the single-launch frame"""
    import torch
    dist_mod = __import__("importlib").import_module("ray-tracer_amd.distributed")
    objs, sky = rt.scenes.monkey()
    W, H, spp = 200, 132, 4           # 17 bands of 8 rows, the last one ragged (132 = 16*8 + 4)
    scene = ctx.commit(rt.SceneObjects(objs))
    cam, rd = rt.Camera(W, H), rt.RenderData(spp, 8, True, sky)
    stream = torch.cuda.current_stream().cuda_stream
    full = torch.full((H, W, 3), -1.0, device="cuda:0")
    rt.render_device(ctx, scene, cam, rd, 12345, 0, full.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    assert (full >= 0).all()
    for world in (2, 3):
        # full-frame output: each "rank" writes only its own bands
        acc = torch.full((H, W, 3), -1.0, device="cuda:0")
        for r in range(world):
            rt.render_device(ctx, scene, cam, rd, 12345, 0, acc.data_ptr(), band_first=r, band_stride=world, stream=stream)
            torch.cuda.synchronize()
            rows = torch.zeros(H, dtype=torch.bool)
            for b in dist_mod.owned_bands(H, 8, r, world):
                rows[b * 8:(b + 1) * 8] = True
            written = (acc >= 0).all(dim=2).all(dim=1).cpu()
            expect = torch.zeros(H, dtype=torch.bool)
            for rr in range(r + 1):
                for b in dist_mod.owned_bands(H, 8, rr, world):
                    expect[b * 8:(b + 1) * 8] = True
            assert torch.equal(written, expect)
        assert torch.equal(acc.view(torch.int32), full.view(torch.int32))
        # compact output + assemble (what the gather does)
        maxrows = dist_mod.max_owned_rows(H, 8, world)
        stacked = torch.zeros((world, maxrows, W, 3), device="cuda:0")
        for r in range(world):
            rt.render_device(ctx, scene, cam, rd, 12345, 0, stacked[r].data_ptr(), band_first=r, band_stride=world, compact=True, stream=stream)
        torch.cuda.synchronize()
        frame = dist_mod.assemble(stacked, W, H, 8, world)
        assert torch.equal(frame.contiguous().view(torch.int32), full.view(torch.int32))


@pytest.mark.parametrize("name", ["monkey", "three_sphere", "cube"])
def test_full_size_properties(rt, orc, ctx, models_dir, name):
    """BASELINE.json configs[1], [2], [3] at their real image size (1920x1080, 8 bounces; small spp) - the only
    place the 256-thread kernels run at their real 5-6 workgroups per CU with 32,400 tiles:
    (1) partition invariance - bands rendered as 8 'ranks' reassemble bit-exactly to the single-launch frame
        (a checksum of checksums over the parts equals the whole);
    (2) the same for the round-3 partition: cost-balanced TILE LISTS (costs measured by the first launch, 8 ranks
        dealt longest-processing-time-first), each rank a 5-frame multi-frame launch with the costs as hints, tiles put
        back with rt_tiles_copy_device - against 5 frames rendered frame by frame;
    (3) bands chosen at random equal the oracle's rows for those bands;
    (4) every value is finite."""
    import torch
    dist_mod = __import__("importlib").import_module("ray-tracer_amd.distributed")
    objs, sky = rt.scenes.CONFIG_SCENES[name]()
    W, H, spp = 1920, 1080, 4
    ctx = rt.Context(0)                      # a fresh context: views and their costs start empty
    scene = ctx.commit(rt.SceneObjects(objs))
    cam, rd = rt.Camera(W, H), rt.RenderData(spp, 8, True, sky)
    stream = torch.cuda.current_stream().cuda_stream
    full = torch.empty((H, W, 3), device="cuda:0")
    rt.render_device(ctx, scene, cam, rd, 12345, 0, full.data_ptr(), stream=stream)
    ids, cost, peak = ctx.tile_costs(with_peaks=True)        # the launch above was the first of its view: it measured the tiles
    assert np.array_equal(ids, np.arange(240 * 135, dtype=np.uint32)) and cost.min() > 0 and peak.min() > 0
    world = 8
    stacked = torch.zeros((world, dist_mod.max_owned_rows(H, 8, world), W, 3), device="cuda:0")
    for r in range(world):
        rt.render_device(ctx, scene, cam, rd, 12345, 0, stacked[r].data_ptr(), band_first=r, band_stride=world, compact=True, stream=stream)
    torch.cuda.synchronize()
    frame = dist_mod.assemble(stacked, W, H, 8, world).contiguous()
    assert torch.equal(frame.view(torch.int32), full.view(torch.int32))
    assert torch.isfinite(full).all()
    del stacked, frame
    # (2) five progressive frames: frame by frame on the whole image ...
    times = [12345 + i for i in range(5)]
    x, y = full.clone(), torch.empty_like(full)
    for i in range(1, 5):
        rt.render_device(ctx, scene, cam, rd, times[i], i, y.data_ptr(), d_prev=x.data_ptr(), stream=stream)
        x, y = y, x
    # ... and as 8 cost-balanced tile lists, one multi-frame launch each
    full_cost, full_peak = np.zeros(240 * 135, np.uint32), np.zeros(240 * 135, np.uint32)
    full_cost[ids], full_peak[ids] = cost, peak
    owner = rt.partition_tiles(W, H, world, full_cost)
    lists = dist_mod.tile_lists(owner, world)
    loads = np.array([full_cost[l].astype(np.int64).sum() for l in lists])
    assert loads.max() - loads.min() <= int(full_cost.max())
    out = torch.full((H, W, 3), -1.0, device="cuda:0")
    buf = torch.zeros(dist_mod.compact_floats(lists), device="cuda:0")
    for r in range(world):
        rt.render_device_batch(ctx, scene, cam, rd, times, 0, buf.data_ptr(), compact=True, stream=stream, tile_list=lists[r], tile_cost=full_cost[lists[r]], tile_peak=full_peak[lists[r]])
        rt.tiles_copy_device(ctx, buf.data_ptr(), out.data_ptr(), W, H, lists[r], True, stream)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int32), x.view(torch.int32))
    host = full.cpu().numpy()
    o = orc.Scene(objs, orc.MATH_DET, models_dir)
    rng = np.random.default_rng(11)
    for band in sorted(rng.choice(H // 8, 6, replace=False)):
        y0, y1 = int(band) * 8, int(band) * 8 + 8
        want = o.render(cam.floats(), W, H, spp, 8, sky, y0=y0, y1=y1)
        assert eq(host[y0:y1], want[y0:y1]), band


def test_rgba8_conversion(rt, orc, ctx):
    """src/main.cu:343-371: int(px*255), clamp to 0..255, alpha 255"""
    import ctypes as C
    import torch
    rng = np.random.default_rng(2)
    rgb = rng.uniform(-0.5, 2.0, (37, 53, 3)).astype(np.float32)
    rgb[0, 0] = [0.0, 1.0, 0.99999994]
    rgb[0, 1] = [7.8, 1.0 / 255, 254.9999 / 255]
    d = torch.from_numpy(rgb).to("cuda:0")
    out = torch.zeros((37, 53, 4), dtype=torch.uint8, device="cuda:0")
    rt.to_rgba8_device(ctx, d.data_ptr(), 53, 37, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = np.zeros((37, 53, 4), np.uint8)
    orc.lib().orc_to_rgba8(rgb.ctypes.data_as(C.POINTER(C.c_float)), 53, 37, want.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert np.array_equal(out.cpu().numpy(), want)


def test_argument_errors(rt, ctx):
    objs, sky = rt.scenes.three_sphere()
    scene = ctx.commit(rt.SceneObjects(objs))
    import torch
    buf = torch.zeros((16, 16, 3), device="cuda:0")
    with pytest.raises(ValueError):
        rt.render_device(ctx, scene, rt.Camera(16, 16), rt.RenderData(1, 1, True, sky), 0, 0, buf.data_ptr(), band_rows=12)
    with pytest.raises(ValueError):
        rt.render_device(ctx, scene, rt.Camera(16, 16), rt.RenderData(1, 1, True, sky), 0, 0, buf.data_ptr(), band_first=2, band_stride=2)
    with pytest.raises(ValueError):
        rt.render_device(ctx, scene, rt.Camera(16, 16), rt.RenderData(-1, 1, True, sky), 0, 0, buf.data_ptr())


def test_cpp_host_mirror_end_to_end(rt, orc, models_dir, tmp_path):
    """host/raytracer.hpp + example_main.cpp: the reference's main() flow (SceneObjects(scene),
    RenderSettings defaults 100 spp / 5 bounces, progressive frames, float->RGBA8) in C++ over
    the C ABI, compared with the oracle's progressive frames converted the same way."""
    import ctypes as C
    import subprocess
    bmod = __import__("importlib").import_module("ray-tracer_amd.build")
    exe = bmod.build_example()
    W, H, frames = 80, 64, 2
    # scene 2 reads textures/parsed_textures.txt (relative, like the reference): bake a procedural image there
    img = rt.scenes.procedural_image(24, 12, seed=3)
    (tmp_path / "textures").mkdir()
    with open(tmp_path / "textures" / "parsed_textures.txt", "w") as fh:
        fh.write("earth.png\n24\n12\n" + "".join("%s %s %s " % tuple(repr(float(c)) for c in px) for row in img for px in row) + "\n")
    for scene_num, name in ((1, "reference_scene1"), (0, "reference_scene0"), (2, "reference_scene2"), (3, "reference_scene3")):
        out = tmp_path / ("s%d.%s" % (scene_num, "png" if scene_num == 1 else "ppm"))
        subprocess.check_call([exe, models_dir, str(scene_num), str(W), str(H), str(frames), str(out)], timeout=300, cwd=str(tmp_path))
        if scene_num == 1:
            from test_png import decode_png
            got = decode_png(out)
        else:
            raw = out.read_bytes()
            header = ("P6\n%d %d\n255\n" % (W, H)).encode()
            assert raw.startswith(header)
            got = np.frombuffer(raw[len(header):], np.uint8).reshape(H, W, 3)
        objs, sky = rt.scenes.reference_scene2(img) if scene_num == 2 else rt.scenes.CONFIG_SCENES[name]()
        o = orc.Scene(objs, orc.MATH_DET, models_dir)
        prev = None
        for f in range(frames):
            prev = o.render(rt.Camera(W, H).floats(), W, H, 100, 5, sky, time_ms=12345 + f, frame_num=f, prev=prev)
        want = np.zeros((H, W, 4), np.uint8)
        orc.lib().orc_to_rgba8(prev.ctypes.data_as(C.POINTER(C.c_float)), W, H, want.ctypes.data_as(C.POINTER(C.c_uint8)))
        assert np.array_equal(got, want[:, :, :3]), name


def test_bench_two_ranks_on_one_gpu():
    """bench.py's N>1 path end to end, started the way the driver starts N = 1: plain `python bench.py --gpus 2`
    (no launcher: the script spawns its own ranks before touching the GPU).  Two ranks share cuda:0 over gloo -
    RCCL refuses duplicate devices, so this is the closest a one-GPU box gets to the multi-GPU run.  The
    headline is strong scaling of the metric's 1920x1080 image; the gathered frame must equal the single-launch
    frame bit for bit; weak scaling and configs[4] ride along as extras."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--spp", "4", "--backend", "gloo", "--share-gpu", "--check", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line, rank 0's
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["unit"] == "Msamples/s" and j["value"] > 0
    assert j["config"]["image"] == "1920x1080" and "1920x1080" in j["metric"]
    assert j["gathered_equals_single_launch"] is True
    assert j["ranks"]["world_size"] == 2 and j["ranks"]["backend"] == "gloo" and len(j["ranks"]["ranks"]) == 2
    assert "roofline" in j and j["roofline"]["bound"] == "hbm"
    assert j["extras"]["weak_scaling"]["image"] == "2720x1530" and j["extras"]["weak_scaling"]["value"] > 0
    assert "3840x2160" in j["extras"]["config4"]["workload"] and j["extras"]["config4"]["value"] > 0
    # under a launcher, weak scaling as the measured mode: the image area grows with N at the same aspect ratio
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + os.getpid() % 300), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--spp", "8", "--backend", "gloo", "--share-gpu", "--check", "--no-cpu-baseline", "--scaling", "weak"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["scaling"] == "weak" and j["config"]["image"] == "2720x1530" and j["gathered_equals_single_launch"] is True
    assert "2720x1530" in j["metric"]


def test_textured_and_refractive_primitives(rt, orc, ctx, models_dir):
    """every texture kind on spheres and on triangle-based shapes, refractive quads / cuboid /
    mesh (normals flipped against the ray), nested refractive spheres"""
    img = rt.scenes.procedural_image(40, 20, seed=5)
    objs = [
        ("sphere", (-0.7, 0.1, 1.8), 0.3, ("checkerboard", (1, 1, 1), (0.1, 0.1, 0.4), 6, 0.2)),
        ("sphere", (0.0, 0.1, 1.8), 0.3, ("gradient", 0)),
        ("sphere", (0.7, 0.1, 1.8), 0.3, ("image", img, 0.1)),
        ("quad", (-1.5, -0.4, 1), (1.5, -0.4, 1), (1.5, -0.4, 4), (-1.5, -0.4, 4), ("image", img, 0)),
        ("sphere", (0.0, 0.1, 1.2), 0.2, ("refractive", (0.9, 1.0, 0.9), 1.5)),
        ("sphere", (0.0, 0.1, 1.2), 0.1, ("refractive", (1.0, 0.8, 0.8), 1.2)),          # nested
        ("cuboid", (0.35, 0.0, 1.1), 0.25, 0.3, 0.25, ("refractive", (1, 1, 1), 1.3)),
        ("obj", "cube.obj", [("enlarge", 0.12), ("rotate", 0.5, 0.3, 0.1), ("translate", -0.5, -0.1, 1.2)], ("refractive", (0.8, 0.9, 1.0), 2.0)),
        ("quad", (-1.5, 1.0, 3.9), (1.5, 1.0, 3.9), (1.5, -0.4, 3.9), (-1.5, -0.4, 3.9), ("emissive", (1, 1, 0.9), 2)),
    ]
    sky = (0.8, 1.0, 1.0)
    got = hip_render(rt, ctx, objs, 200, 120, 8, 10, sky)
    want = orc.Scene(objs, orc.MATH_DET, models_dir).render(rt.Camera(200, 120).floats(), 200, 120, 8, 10, sky)
    assert eq(got, want)
    assert np.isfinite(got).all()


def _big_mesh_scene(rt, n=6000):
    """a mesh of n random small triangles over a checkerboard ground: too large for a CU's LDS"""
    return rt.scenes.soup6k(n)


def test_scene_larger_than_lds_uses_global_memory(rt, orc, ctx, models_dir, monkeypatch):
    """maximum sizes: a 6,000-triangle mesh (BVH + triangles ~ 350 KB) cannot be staged into a CU's 160 KB LDS.  Its BVH
    and records can (64 KB): the kernel then reads only the triangles from global memory (hybrid).  Three meshes' BVHs
    cannot either: everything is read from global memory.  Both must match the oracle; and so must the all-global kernel on
    the one-mesh scene (RT_AMD_SCENE_MODE=0, the development override)."""
    n = 6000
    objs, sky = _big_mesh_scene(rt, n)
    scene = ctx.commit(rt.SceneObjects(objs))
    info = scene.info()
    assert info["scene_in_lds"] == 2 and info["num_triangles"] == n      # BVH + records in LDS, triangles from L2
    got = hip_render(rt, ctx, objs, 160, 96, 4, 6, sky)
    want = orc.Scene(objs, orc.MATH_DET, models_dir).render(rt.Camera(160, 96).floats(), 160, 96, 4, 6, sky)
    assert eq(got, want)
    monkeypatch.setenv("RT_AMD_SCENE_MODE", "0")
    assert ctx.commit(rt.SceneObjects(objs)).info()["scene_in_lds"] == 0
    assert eq(hip_render(rt, ctx, objs, 160, 96, 4, 6, sky), want)
    monkeypatch.delenv("RT_AMD_SCENE_MODE")
    # three meshes of 2,000 triangles each: 3 x 1,023 nodes do not fit next to the traversal stacks
    tris = np.asarray(objs[0][1]).reshape(-1, 9)
    three = [("mesh", tris[i::3], objs[0][2]) for i in range(3)] + list(objs[1:])
    assert ctx.commit(rt.SceneObjects(three)).info()["scene_in_lds"] == 0
    got = hip_render(rt, ctx, three, 128, 80, 3, 5, sky)
    want = orc.Scene(three, orc.MATH_DET, models_dir).render(rt.Camera(128, 80).floats(), 128, 80, 3, 5, sky)
    assert eq(got, want)


def _random_scene(seed):
    """a seeded mix of every primitive and material kind, in random list order"""
    rng = np.random.default_rng(seed)

    def vec(lo, hi):
        return tuple(float(x) for x in rng.uniform(lo, hi, 3))

    def material():
        k = rng.integers(0, 7)
        col = vec(0.1, 1.0)
        if k == 0:
            return ("standard", col, float(rng.uniform(0, 1)))
        if k == 1:
            return ("standard", col, 0.0)
        if k == 2:
            return ("emissive", col, float(rng.uniform(0.5, 5)))
        if k == 3:
            return ("checkerboard", col, vec(0, 0.5), int(rng.integers(1, 12)), float(rng.uniform(0, 0.5)))
        if k == 4:
            return ("gradient", float(rng.uniform(0, 0.3)))
        if k == 5:
            return ("refractive", col, float(rng.uniform(1.05, 2.2)))
        return ("image", rng.uniform(0, 1, (int(rng.integers(2, 9)), int(rng.integers(2, 9)), 3)).astype(np.float32), 0.0)

    objs = []
    for _ in range(int(rng.integers(6, 14))):
        kind = rng.integers(0, 7)
        c = np.array(vec([-1.2, -0.7, 1.0], [1.2, 0.9, 3.5]))
        if kind == 0:
            objs.append(("sphere", tuple(c), float(rng.uniform(0.08, 0.45)), material()))
        elif kind == 1:
            p = [tuple(c + rng.normal(0, 0.35, 3)) for _ in range(3)]
            objs.append(("triangle_uv", p, [tuple(rng.uniform(0, 1, 2)) for _ in range(3)], material()))
        elif kind == 2:
            a, b = rng.normal(0, 0.4, 3), rng.normal(0, 0.4, 3)
            objs.append(("quad", tuple(c), tuple(c + a), tuple(c + a + b), tuple(c + b), material()))
        elif kind == 3:
            a, b = rng.normal(0, 0.5, 3), rng.normal(0, 0.5, 3)
            objs.append(("one_way_quad", tuple(c), tuple(c + a), tuple(c + a + b), tuple(c + b), bool(rng.integers(0, 2)), material()))
        elif kind == 4:
            objs.append(("cuboid", tuple(c), float(rng.uniform(0.1, 0.5)), float(rng.uniform(0.1, 0.5)), float(rng.uniform(0.1, 0.5)), material()))
        elif kind == 5:
            objs.append(("obj", "cube.obj", [("enlarge", float(rng.uniform(0.08, 0.25))), ("rotate", *[float(x) for x in rng.uniform(-3, 3, 3)]),
                                             ("translate", *[float(x) for x in c])], material()))
        else:
            n = int(rng.integers(1, 60))
            tris = (c + rng.normal(0, 0.25, (n, 1, 3)) + rng.normal(0, 0.08, (n, 3, 3))).astype(np.float32).reshape(n, 9)
            objs.append(("mesh", tris, material()))
    if rng.integers(0, 2):
        objs.append(("sphere", (0, -100.5, 1.5), 100, ("standard", (0.5, 0.5, 0.5), float(rng.uniform(0, 0.4)))))
    sky = (0.8, 1.0, 1.0) if rng.integers(0, 2) else (0.0, 0.0, 0.0)
    return objs, sky


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_random_mixed_scenes(rt, orc, ctx, models_dir, seed):
    objs, sky = _random_scene(seed)
    W, H, spp, limit = 112 + 8 * (seed % 3), 72, 5, 3 + seed % 6
    got = hip_render(rt, ctx, objs, W, H, spp, limit, sky, time_ms=1000 + seed)
    want = orc.Scene(objs, orc.MATH_DET, models_dir).render(rt.Camera(W, H).floats(), W, H, spp, limit, sky, time_ms=1000 + seed)
    assert eq(got, want)


@pytest.mark.parametrize("aa", [False, True])
def test_rays_with_a_zero_direction_component(rt, orc, ctx, models_dir, aa):
    """A direction component of exactly 0 makes 1/d infinite and (plane - origin) * (1/d) a NaN whenever the origin
    lies ON a box plane.  The kernel's six-med3 slab test (rt_pixel.h box_enter_med3) is only proved equal to the
    reference's min/max form (src/objects.cu:404-434, fminf/fmaxf drop a NaN operand) for products that are numbers, so
    traversals of such rays take the min/max copy of the loop.  Here: a camera at the origin whose pixel grid puts a whole
    column at x == 0 and a whole row at y == 0 (antialias off: the primary rays keep those zeros), meshes whose vertices -
    hence leaf and inner boxes - have coordinates of exactly 0 in x and y, and a mirror-like ground so that bounced rays
    start ON box planes too."""
    rng = np.random.default_rng(5)
    n = 160
    tris = (rng.normal(0, 0.9, (n, 1, 3)) + rng.normal(0, 0.35, (n, 3, 3))).astype(np.float32)
    tris[..., 2] -= 4.0
    tris[rng.random((n, 3)) < 0.35, 0] = 0.0          # many vertices in the plane x == 0 ...
    tris[rng.random((n, 3)) < 0.35, 1] = 0.0          # ... and y == 0
    objs = [("mesh", tris.reshape(n, 9), ("standard", (0.8, 0.6, 0.4), 0.9)),
            ("obj", "cube.obj", [("enlarge", 0.5), ("translate", 0.5, 0.5, -3.0)], ("standard", (0.3, 0.9, 0.5), 0.2)),   # faces in x == 0 and y == 0
            ("sphere", (3.0, 4.0, -4.0), 1.5, ("emissive", (1.0, 1.0, 1.0), 6)),
            ("quad", (-6, -2, 0), (6, -2, 0), (6, -2, -9), (-6, -2, -9), ("standard", (0.6, 0.6, 0.6), 1.0))]
    W, H = 64, 48
    cam = np.array([0, 0, 0, -4.0, 3.0, -8.0, 0.125, 0, 0, 0, -0.125, 0], np.float32)    # pixel (32, y): x == 0; (x, 24): y == 0
    scene = ctx.commit(rt.SceneObjects(objs))
    data = rt.VariableRenderData(W, H)
    rt.render(ctx, scene, rt.Camera(W, H, floats=cam), rt.RenderData(6, 5, aa, (0.5, 0.7, 1.0)), data, 4242)
    want = orc.Scene(objs, orc.MATH_DET, models_dir).render(cam, W, H, 6, 5, (0.5, 0.7, 1.0), time_ms=4242, antialias=aa)
    assert eq(data.previous_render, want)


@pytest.mark.parametrize("aa", [False, True])
def test_extreme_operands_of_the_sphere_test(rt, orc, ctx, models_dir, aa):
    """The device computes the sphere test's near root with a short division (rt_math.h rt__div_benign) whose precondition the code
    argues from `d` being a unit vector (rt_pixel.h).  Here the operands the argument has to cover: cameras so far away that a ray's
    squared length overflows (d = 0 or NaN), spheres at astronomic distances (infinite discriminants and dividends), a ray origin
    exactly ON a sphere (dividends that cancel to ~0), radius 0 and a sphere around the camera, mixed with ordinary geometry so that
    accepted hits exist next to the rejected ones.  The oracle divides with the operator: the frames must be the same bits."""
    std = ("standard", (0.8, 0.7, 0.6), 0.3)
    objs = [("sphere", (0.0, 0.0, -4.0), 1.0, std),
            ("sphere", (0.0, -1001.0, -4.0), 1000.0, ("standard", (0.5, 0.5, 0.5), 0.0)),
            ("sphere", (3.0e25, 1.0e25, -2.0e25), 1.0e24, ("emissive", (1.0, 1.0, 1.0), 3)),       # squared distances overflow
            ("sphere", (-2.0, 0.5, -3.0), 0.0, std),                                               # radius 0
            ("sphere", (0.0, 0.0, 0.0), 0.5, ("standard", (0.9, 0.2, 0.2), 0.8)),                   # the camera of the first view sits inside it
            ("sphere", (1.5, 0.25, -3.5), 1.0e-20, std)]                                           # vanishing radius
    W, H = 48, 32
    sky = (0.5, 0.7, 1.0)
    cams = [np.array([0, 0, 0, -1.5, 1.0, -2.0, 0.0625, 0, 0, 0, -0.0625, 0], np.float32),                      # ordinary (inside the red sphere)
            np.array([1e20, 0, 0, -1.5, 1.0, -2.0, 0.0625, 0, 0, 0, -0.0625, 0], np.float32),                   # |direction|^2 overflows: d = 0
            np.array([0, 0, 0, -3e38, 3e38, -3e38, 1e37, 0, 0, 0, -1e37, 0], np.float32),                       # infinite intermediate values
            np.array([0, 0, -3.0, -1.5, 1.0, -5.0, 0.0625, 0, 0, 0, -0.0625, 0], np.float32),                   # origin exactly ON the unit sphere at (0, 0, -4)
            np.array([0, 0, 0, -1.5e-30, 1.0e-30, -2.0e-30, 6.25e-32, 0, 0, 0, -6.25e-32, 0], np.float32)]      # a pixel grid 1e-30 wide
    scene = ctx.commit(rt.SceneObjects(objs))
    o = orc.Scene(objs, orc.MATH_DET, models_dir)
    for i, cam in enumerate(cams):
        data = rt.VariableRenderData(W, H)
        rt.render(ctx, scene, rt.Camera(W, H, floats=cam), rt.RenderData(5, 6, aa, sky), data, 31 + i)
        want = o.render(cam, W, H, 5, 6, sky, time_ms=31 + i, antialias=aa)
        got = data.previous_render
        nan_g, nan_w = np.isnan(got), np.isnan(want)
        assert np.array_equal(nan_g, nan_w), i
        assert eq(np.where(nan_g, 0, got), np.where(nan_w, 0, want)), i


@pytest.mark.parametrize("name", ["monkey", "three_sphere"])
def test_tiny_and_thin_images(rt, orc, ctx, models_dir, name):
    """images smaller than one tile, one pixel wide or high, one pixel in total"""
    objs, sky = rt.scenes.CONFIG_SCENES[name]()
    o = orc.Scene(objs, orc.MATH_DET, models_dir)
    for W, H, spp, limit in ((1, 1, 3, 8), (7, 3, 2, 4), (8, 8, 1, 1), (9, 17, 2, 8), (64, 1, 2, 3), (1, 40, 2, 3), (130, 5, 1, 8)):
        got = hip_render(rt, ctx, objs, W, H, spp, limit, sky, time_ms=777)
        want = o.render(rt.Camera(W, H).floats(), W, H, spp, limit, sky, time_ms=777)
        assert eq(got, want), (W, H)


@pytest.mark.parametrize("name,W,H,spp,limit,frames", [("monkey", 200, 120, 6, 8, 4), ("three_sphere", 160, 96, 4, 8, 5),
                                                       ("reference_scene0", 125, 100, 3, 5, 3), ("cube", 96, 64, 5, 8, 16),
                                                       ("reference_scene2", 100, 80, 3, 5, 4),      # image texture
                                                       ("reference_scene3", 96, 72, 3, 8, 4),       # refraction (a draw inside shading)
                                                       ("reference_scene4", 120, 68, 2, 6, 4)])     # 100 spheres, no mesh
def test_multi_frame_launch_equals_frame_by_frame(rt, orc, ctx, models_dir, name, W, H, spp, limit, frames):
    """rt_render_device_batch: F progressive frames in ONE launch (frame k+1 is traced while frame k's
    expensive pixels still run; the blend of a pixel waits for its previous frame) must give the image
    of F launches - and of the oracle's progressive loop - bit for bit."""
    import torch
    objs, sky = rt.scenes.CONFIG_SCENES[name]()
    scene = ctx.commit(rt.SceneObjects(objs))
    cam, rd = rt.Camera(W, H), rt.RenderData(spp, limit, True, sky)
    times = [1000 + 37 * i for i in range(frames)]
    st = torch.cuda.current_stream().cuda_stream
    # frame by frame, ping-pong buffers
    a = torch.zeros((H, W, 3), device="cuda:0"); b = torch.empty_like(a)
    for i, t in enumerate(times):
        rt.render_device(ctx, scene, cam, rd, t, i, b.data_ptr(), d_prev=a.data_ptr() if i else None, stream=st)
        a, b = b, a
    want = a.cpu().numpy()
    # all at once, in place
    fr = torch.full((H, W, 3), 7.0, device="cuda:0")          # garbage: frame_num 0 ignores it
    rt.render_device_batch(ctx, scene, cam, rd, times, 0, fr.data_ptr(), stream=st)
    assert eq(fr.cpu().numpy(), want)
    # the oracle's progressive loop
    o = orc.Scene(objs, orc.MATH_DET, models_dir)
    prev = None
    for i, t in enumerate(times):
        prev = o.render(cam.floats(), W, H, spp, limit, sky, time_ms=t, frame_num=i, prev=prev)
    assert eq(want, prev)
    # continuing an accumulation: frames 2.. on top of the image after frames 0-1, in two launches
    if frames >= 4:
        fr2 = torch.empty((H, W, 3), device="cuda:0")
        rt.render_device_batch(ctx, scene, cam, rd, times[:2], 0, fr2.data_ptr(), stream=st)
        rt.render_device_batch(ctx, scene, cam, rd, times[2:], 2, fr2.data_ptr(), stream=st)
        assert eq(fr2.cpu().numpy(), want)


def test_multi_frame_launch_on_bands(rt, ctx):
    """compact band buffers (the multi-GPU partition) through the multi-frame launch"""
    import importlib
    import torch
    dm = importlib.import_module("ray-tracer_amd.distributed")
    objs, sky = rt.scenes.monkey()
    scene = ctx.commit(rt.SceneObjects(objs))
    W, H, world = 176, 100, 3
    cam, rd = rt.Camera(W, H), rt.RenderData(5, 8, True, sky)
    times = [5, 6, 7]
    st = torch.cuda.current_stream().cuda_stream
    full = torch.empty((H, W, 3), device="cuda:0")
    rt.render_device_batch(ctx, scene, cam, rd, times, 0, full.data_ptr(), stream=st)
    parts = []
    for r in range(world):
        buf = torch.zeros((dm.max_owned_rows(H, 8, world), W, 3), device="cuda:0")
        rt.render_device_batch(ctx, scene, cam, rd, times, 0, buf.data_ptr(), band_first=r, band_stride=world, compact=True, stream=st)
        parts.append(buf)
    got = dm.assemble(torch.stack(parts), W, H, 8, world)
    assert torch.equal(got.contiguous().view(torch.int32), full.view(torch.int32))


def test_render_frames_host_buffers(rt, ctx):
    """rt_render_frames: 19 frames in one call (a 16-frame launch + a 3-frame launch) on top of two
    frames rendered one by one == 21 render() calls"""
    objs, sky = rt.scenes.cube()
    scene = ctx.commit(rt.SceneObjects(objs))
    W, H = 72, 40
    cam, rd = rt.Camera(W, H), rt.RenderData(2, 4, True, sky)
    a = rt.VariableRenderData(W, H)
    for i in range(21):
        rt.render(ctx, scene, cam, rd, a, 500 + i)
    b = rt.VariableRenderData(W, H)
    rt.render(ctx, scene, cam, rd, b, 500)
    rt.render(ctx, scene, cam, rd, b, 501)
    rt.render_frames(ctx, scene, cam, rd, b, [502 + i for i in range(19)])
    assert a.frame_num == b.frame_num == 21
    assert eq(a.previous_render, b.previous_render)


def test_multi_frame_launch_full_size_and_global_scene(rt, ctx):
    """the multi-frame launch where every CU takes part (1920x1080: hand-overs between XCDs), and on a
    scene that is read from global memory (LDS holds only the stacks): frames in one launch == frame
    by frame"""
    import torch
    st = torch.cuda.current_stream().cuda_stream
    for objs, sky, W, H, spp, frames in ((*rt.scenes.monkey(), 1920, 1080, 2, 5), (*_big_mesh_scene(rt), 320, 200, 2, 3)):
        scene = ctx.commit(rt.SceneObjects(objs))
        cam, rd = rt.Camera(W, H), rt.RenderData(spp, 8, True, sky)
        times = [900 + i for i in range(frames)]
        a = torch.zeros((H, W, 3), device="cuda:0"); b = torch.empty_like(a)
        for i, t in enumerate(times):
            rt.render_device(ctx, scene, cam, rd, t, i, b.data_ptr(), d_prev=a.data_ptr() if i else None, stream=st)
            a, b = b, a
        fr = torch.empty((H, W, 3), device="cuda:0")
        for _ in range(2):          # twice: the second launch of the view runs with the refined tile order
            rt.render_device_batch(ctx, scene, cam, rd, times, 0, fr.data_ptr(), stream=st)
            assert torch.equal(fr.view(torch.int32), a.view(torch.int32))


def test_batch_entry_one_frame_at_a_time(rt, ctx):
    """rt_render_device_batch with ONE frame per call, accumulating in place over several calls, is the
    frame-by-frame loop with a single buffer"""
    import torch
    objs, sky = rt.scenes.three_sphere()
    scene = ctx.commit(rt.SceneObjects(objs))
    W, H = 90, 50
    cam, rd = rt.Camera(W, H), rt.RenderData(3, 6, True, sky)
    st = torch.cuda.current_stream().cuda_stream
    a = torch.zeros((H, W, 3), device="cuda:0"); b = torch.empty_like(a)
    fr = torch.empty((H, W, 3), device="cuda:0")
    for i in range(4):
        rt.render_device(ctx, scene, cam, rd, 70 + i, i, b.data_ptr(), d_prev=a.data_ptr() if i else None, stream=st)
        a, b = b, a
        rt.render_device_batch(ctx, scene, cam, rd, [70 + i], i, fr.data_ptr(), stream=st)
        assert torch.equal(fr.view(torch.int32), a.view(torch.int32)), i
    with pytest.raises(ValueError, match="1..32"):
        rt.render_device_batch(ctx, scene, cam, rd, list(range(33)), 0, fr.data_ptr(), stream=st)     # more than 32 frames per launch
    # 32 frames in one launch (the frame index of a pixel uses all its bits) against 32 launches
    times = list(range(900, 932))
    x = torch.zeros((H, W, 3), device="cuda:0"); y = torch.empty_like(x)
    for i, t in enumerate(times):
        rt.render_device(ctx, scene, cam, rd, t, i, y.data_ptr(), d_prev=x.data_ptr() if i else None, stream=st)
        x, y = y, x
    rt.render_device_batch(ctx, scene, cam, rd, times, 0, fr.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert torch.equal(fr.view(torch.int32), x.view(torch.int32))
