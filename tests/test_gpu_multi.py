"""GPU tests of what sits around the kernel: launches on caller streams, the multi-GPU C ABI
(rt_render_multi / rt_render_multi_device / rt_gather — several contexts on the one GPU of the
test box stand in for several GPUs; the band partition, the peer-copy bookkeeping and the
de-interleave are the same code), BASELINE configs[4]'s image size, and the reference's own
camera floats passed verbatim.  Everything is compared bit for bit."""
import hashlib
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def eq(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32))


def oracle_progressive(orc, objs, models_dir, cam_floats, W, H, spp, limit, sky, times, first_frame=0, prev=None):
    o = orc.Scene(objs, orc.MATH_DET, models_dir)
    for k, t in enumerate(times):
        prev = o.render(cam_floats, W, H, spp, limit, sky, time_ms=t, frame_num=first_frame + k, prev=prev)
    return prev


def test_launches_on_a_non_default_stream(rt, orc, models_dir):
    """Three launches of a mesh scene on a torch side stream (hipStreamNonBlocking: the null stream does not
    wait for it): the first collects per-tile costs, the second reads them back and rewrites the tile order,
    the third runs on the refined order.  Every frame must equal the oracle."""
    import torch
    objs, sky = rt.scenes.monkey()
    W, H, spp = 320, 184, 6
    ctx = rt.Context(0)                      # a fresh context: its tile-order cache starts empty
    scene = ctx.commit(rt.SceneObjects(objs))
    cam, rd = rt.Camera(W, H), rt.RenderData(spp, 8, True, sky)
    side = torch.cuda.Stream()
    frames = [torch.full((H, W, 3), -1.0, device="cuda:0") for _ in range(3)]
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for i, f in enumerate(frames):
            rt.render_device(ctx, scene, cam, rd, 500 + i, 0, f.data_ptr(), stream=side.cuda_stream)
    side.synchronize()
    ctx.synchronize()
    o = orc.Scene(objs, orc.MATH_DET, models_dir)
    for i, f in enumerate(frames):
        assert eq(f.cpu().numpy(), o.render(cam.floats(), W, H, spp, 8, sky, time_ms=500 + i)), i
    # a multi-frame launch on the side stream, then one on the default stream: the second is queued behind the first
    a = torch.zeros((H, W, 3), device="cuda:0")
    b = torch.zeros((H, W, 3), device="cuda:0")
    rt.render_device_batch(ctx, scene, cam, rd, [1, 2, 3], 0, a.data_ptr(), stream=side.cuda_stream)
    rt.render_device_batch(ctx, scene, cam, rd, [1, 2, 3], 0, b.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = oracle_progressive(orc, objs, models_dir, cam.floats(), W, H, spp, 8, sky, [1, 2, 3])
    assert eq(a.cpu().numpy(), want) and eq(b.cpu().numpy(), want)


@pytest.mark.parametrize("name,W,H,n_ranks", [("monkey", 200, 132, 2), ("reference_scene0", 125, 100, 3), ("three_sphere", 96, 40, 8)])
def test_render_multi_host_buffers(rt, orc, models_dir, name, W, H, n_ranks):
    """rt_render_multi = render() for a node: n contexts, each with the scene, one call; progressive over two
    calls (the second scatters the image so far to its owners first).  Heights with a ragged last band, and
    more ranks than some sizes have bands for."""
    objs, sky = rt.scenes.CONFIG_SCENES[name]()
    limit = 5 if name.startswith("reference") else 8
    ctxs = [rt.Context(0) for _ in range(n_ranks)]
    so = rt.SceneObjects(objs)
    scenes = [c.commit(so) for c in ctxs]
    cam, rd = rt.Camera(W, H), rt.RenderData(4, limit, True, sky)
    data = rt.VariableRenderData(W, H)
    rt.render_multi(ctxs, scenes, cam, rd, data, [10, 11, 12])
    assert data.frame_num == 3
    want = oracle_progressive(orc, objs, models_dir, cam.floats(), W, H, 4, limit, sky, [10, 11, 12])
    assert eq(data.previous_render, want)
    # (the first call dealt the tiles out interleaved and measured them; this one owns them by cost)
    rt.render_multi(ctxs, scenes, cam, rd, data, [13])
    want = oracle_progressive(orc, objs, models_dir, cam.floats(), W, H, 4, limit, sky, [13], first_frame=3, prev=want)
    assert data.frame_num == 4 and eq(data.previous_render, want)
    # ... and a third call in the steady state, two frames
    rt.render_multi(ctxs, scenes, cam, rd, data, [14, 15])
    want = oracle_progressive(orc, objs, models_dir, cam.floats(), W, H, 4, limit, sky, [14, 15], first_frame=4, prev=want)
    assert data.frame_num == 6 and eq(data.previous_render, want)
    # and the single-context entry point gives the same image
    one = rt.VariableRenderData(W, H)
    rt.render_frames(ctxs[0], scenes[0], cam, rd, one, [10, 11, 12, 13, 14, 15])
    assert eq(one.previous_render, want)
    assert ctxs[0].peer_access(ctxs[-1]) == 1          # same GPU here; on a node: 1 = xGMI peer access is on


def test_render_multi_device_and_gather(rt, orc, models_dir):
    """device-buffer form on a caller stream, bands of 16 rows; then the same frame put together by hand from
    rt_render_device_batch launches + rt_gather (the exchange step alone)"""
    import torch
    objs, sky = rt.scenes.cube()
    W, H, spp, n = 176, 120, 5, 3
    ctxs = [rt.Context(0) for _ in range(n)]
    so = rt.SceneObjects(objs)
    scenes = [c.commit(so) for c in ctxs]
    cam, rd = rt.Camera(W, H), rt.RenderData(spp, 8, True, sky)
    side = torch.cuda.Stream()
    frame = torch.full((H, W, 3), -1.0, device="cuda:0")
    torch.cuda.synchronize()
    rt.render_multi_device(ctxs, scenes, cam, rd, [7, 8], 0, frame.data_ptr(), band_rows=16, stream=side.cuda_stream)
    rt.render_multi_device(ctxs, scenes, cam, rd, [9], 2, frame.data_ptr(), band_rows=16, stream=side.cuda_stream)
    side.synchronize()
    for c in ctxs:
        c.synchronize()
    want = oracle_progressive(orc, objs, models_dir, cam.floats(), W, H, spp, 8, sky, [7, 8, 9])
    assert eq(frame.cpu().numpy(), want)
    # by hand: every rank renders into its own compact buffer, rt_gather lands it in the frame
    dm = importlib.import_module("ray-tracer_amd.distributed")
    out = torch.full((H, W, 3), -1.0, device="cuda:0")
    bufs = []
    for r in range(n):
        rows = rt.tile_owned_rows(H, 8, r, n)
        buf = torch.zeros((max(rows, 1), W, 3), device="cuda:0")
        bufs.append(buf)
        if rows:
            rt.render_device_batch(ctxs[r], scenes[r], cam, rd, [7, 8, 9], 0, buf.data_ptr(), band_first=r, band_stride=n, compact=True, stream=side.cuda_stream)
            rt.gather(ctxs[0], out.data_ptr(), W, H, ctxs[r], buf.data_ptr(), 8, r, n, stream=side.cuda_stream)
    side.synchronize()
    assert eq(out.cpu().numpy(), want)
    assert dm.num_bands(H, 8) == 15


@pytest.mark.parametrize("W,H", [(203, 117), (64, 64)])
def test_tile_list_forms(rt, orc, models_dir, W, H):
    """rt_tile_spec's tile-list form against the whole-frame render, bit for bit: a random subset of tiles (ragged
    right and bottom edge tiles included) rendered (a) into a full-layout frame (other pixels untouched),
    (b) compact, put back with rt_tiles_copy_device and with rt_gather's list form, (c) as a multi-frame launch in
    place, compact and full layout, with and without cost hints; then the round trip frame -> compact -> frame."""
    import torch
    objs, sky = rt.scenes.monkey()
    ctx = rt.Context(0)
    scene = ctx.commit(rt.SceneObjects(objs))
    cam, rd = rt.Camera(W, H), rt.RenderData(5, 8, True, sky)
    st = torch.cuda.current_stream().cuda_stream
    times = [77, 78, 79]
    whole = torch.zeros((H, W, 3), device="cuda:0")
    rt.render_device(ctx, scene, cam, rd, times[0], 0, whole.data_ptr(), stream=st)
    whole3 = torch.zeros((H, W, 3), device="cuda:0")
    rt.render_device_batch(ctx, scene, cam, rd, times, 0, whole3.data_ptr(), stream=st)
    tx, ty = (W + 7) // 8, (H + 7) // 8
    rng = np.random.default_rng(W)
    ids = rng.permutation(tx * ty)[: (tx * ty) // 2].astype(np.uint32)
    ids[0], ids[1] = tx * ty - 1, tx - 1                           # the two ragged corners are in
    ids = np.unique(ids)
    rng.shuffle(ids)
    mask = np.zeros((ty * 8, tx * 8), bool)
    for g in ids:
        mask[(g // tx) * 8:(g // tx) * 8 + 8, (g % tx) * 8:(g % tx) * 8 + 8] = True
    mask = torch.from_numpy(mask[:H, :W]).to("cuda:0")
    # (a) full layout
    a = torch.full((H, W, 3), -1.0, device="cuda:0")
    rt.render_device(ctx, scene, cam, rd, times[0], 0, a.data_ptr(), stream=st, tile_list=ids)
    torch.cuda.synchronize()
    assert torch.equal(a[mask].view(torch.int32), whole[mask].view(torch.int32)) and bool((a[~mask] == -1.0).all())
    tile_ids, cost, peak = ctx.tile_costs(with_peaks=True)
    assert np.array_equal(tile_ids, ids) and cost.min() > 0 and peak.min() > 0 and (2 * peak.astype(np.int64) <= cost.astype(np.int64) + 1).all()
    # (b) compact + the two ways back into a frame
    c = torch.full((len(ids) * 192,), -2.0, device="cuda:0")
    rt.render_device(ctx, scene, cam, rd, times[0], 0, c.data_ptr(), compact=True, stream=st, tile_list=ids)
    for how in ("copy", "gather"):
        b = torch.full((H, W, 3), -1.0, device="cuda:0")
        if how == "copy":
            rt.tiles_copy_device(ctx, c.data_ptr(), b.data_ptr(), W, H, ids, True, st)
        else:
            rt.gather(ctx, b.data_ptr(), W, H, ctx, c.data_ptr(), stream=st, tile_list=ids)
        torch.cuda.synchronize()
        assert torch.equal(b.view(torch.int32), a.view(torch.int32)), how
    # (c) three progressive frames in one launch, in place
    for hints, peaks in ((None, None), (cost, None), (cost, peak)):
        c3 = torch.zeros((len(ids) * 192,), device="cuda:0")
        rt.render_device_batch(ctx, scene, cam, rd, times, 0, c3.data_ptr(), compact=True, stream=st, tile_list=ids, tile_cost=hints, tile_peak=peaks)
        b = torch.full((H, W, 3), -1.0, device="cuda:0")
        rt.tiles_copy_device(ctx, c3.data_ptr(), b.data_ptr(), W, H, ids, True, st)
        f3 = torch.full((H, W, 3), -1.0, device="cuda:0")
        rt.render_device_batch(ctx, scene, cam, rd, times, 0, f3.data_ptr(), stream=st, tile_list=ids, tile_cost=hints, tile_peak=peaks)
        torch.cuda.synchronize()
        for got in (b, f3):
            assert torch.equal(got[mask].view(torch.int32), whole3[mask].view(torch.int32)) and bool((got[~mask] == -1.0).all())
    # frame -> compact -> frame
    back = torch.zeros((len(ids) * 192,), device="cuda:0")
    rt.tiles_copy_device(ctx, back.data_ptr(), whole3.data_ptr(), W, H, ids, False, st)
    again = torch.full((H, W, 3), -1.0, device="cuda:0")
    rt.tiles_copy_device(ctx, back.data_ptr(), again.data_ptr(), W, H, ids, True, st)
    torch.cuda.synchronize()
    assert torch.equal(again[mask].view(torch.int32), whole3[mask].view(torch.int32))
    # errors: a tile outside the image, a tile listed twice, costs without a list
    with pytest.raises(ValueError):
        rt.render_device(ctx, scene, cam, rd, 1, 0, a.data_ptr(), stream=st, tile_list=[tx * ty])
    with pytest.raises(ValueError):
        rt.render_device(ctx, scene, cam, rd, 1, 0, a.data_ptr(), stream=st, tile_list=[3, 3])
    empty = torch.full((H, W, 3), -1.0, device="cuda:0")
    rt.render_device(ctx, scene, cam, rd, 1, 0, empty.data_ptr(), stream=st, tile_list=[])        # renders nothing
    torch.cuda.synchronize()
    assert bool((empty == -1.0).all())


def test_render_multi_device_balanced_lists(rt, orc, models_dir):
    """rt_render_multi_device with band_rows = 0 (cost-balanced tile lists) on a caller stream, three calls (interleaved
    + measuring, balanced, steady state), each against the oracle; then a call on ANOTHER stream of the root
    (the staging areas are ordered behind the previous call's de-interleave by events, not by stream order)"""
    import torch
    objs, sky = rt.scenes.monkey()
    W, H, spp, n = 232, 136, 3, 4
    ctxs = [rt.Context(0) for _ in range(n)]
    so = rt.SceneObjects(objs)
    scenes = [c.commit(so) for c in ctxs]
    cam, rd = rt.Camera(W, H), rt.RenderData(spp, 8, True, sky)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    frame = torch.full((H, W, 3), -1.0, device="cuda:0")
    torch.cuda.synchronize()
    calls = [([1, 2], s1), ([3], s1), ([4, 5, 6], s1), ([7], s2)]
    done = 0
    for times, s in calls:
        if s is s2:
            s1.synchronize()
        rt.render_multi_device(ctxs, scenes, cam, rd, times, done, frame.data_ptr(), stream=s.cuda_stream)
        done += len(times)
    s2.synchronize()
    for c in ctxs:
        c.synchronize()
    want = oracle_progressive(orc, objs, models_dir, cam.floats(), W, H, spp, 8, sky, [1, 2, 3, 4, 5, 6, 7])
    assert eq(frame.cpu().numpy(), want)


def test_tile_list_cache_turnover_during_a_multi_call(rt, orc, models_dir):
    """Found by tests/soak/soak_partition.py (seed 100026): a context caches the tile lists it has uploaded (64 of them) and
    empties the cache when it is full; rt_render_multi_device holds every rank's list on the root at once, and a turnover in
    the middle of the call left it with pointers to freed lists.  Here the root's cache is brought to the brim with 61 other
    lists first, so the very next multi-GPU call crosses the limit."""
    import torch
    objs, sky = rt.scenes.cube()
    W, H, spp, n = 88, 64, 2, 5
    ctxs = [rt.Context(0) for _ in range(n)]
    so = rt.SceneObjects(objs)
    scenes = [c.commit(so) for c in ctxs]
    cam, rd = rt.Camera(W, H), rt.RenderData(spp, 8, True, sky)
    st = torch.cuda.current_stream().cuda_stream
    frame = torch.zeros((H, W, 3), device="cuda:0")
    scratch = torch.zeros((H, W, 3), device="cuda:0")
    buf = torch.zeros(((W // 8) * (H // 8) * 192,), device="cuda:0")
    want = None
    done = 0
    for rnd, times in enumerate(([5], [6, 7], [8])):
        for k in range(61):                              # 61 distinct lists on the root between the multi-GPU calls
            rt.tiles_copy_device(ctxs[0], buf.data_ptr(), scratch.data_ptr(), W, H, [(k + 61 * rnd) % 88, (k * 7 + rnd) % 88 if (k * 7 + rnd) % 88 != (k + 61 * rnd) % 88 else 87 - (k % 80)], False, st)
        rt.render_multi_device(ctxs, scenes, cam, rd, times, done, frame.data_ptr(), stream=st)
        torch.cuda.synchronize()
        want = oracle_progressive(orc, objs, models_dir, cam.floats(), W, H, spp, 8, sky, times, first_frame=done, prev=want)
        done += len(times)
        assert eq(frame.cpu().numpy(), want), rnd


def test_tile_list_cache_recycles_entries(rt):
    """The tile-list cache holds 64 lists per context; the 65th recycles the least recently used entry - its device buffer
    is overwritten by an upload that is ordered behind the launches that read the old content (stream order, or an event
    across streams), with no device-wide synchronisation (round 3 emptied the whole cache after hipDeviceSynchronize,
    VERDICT r03).  300 distinct lists on two alternating streams, every one checked: a compact image scattered into a frame
    lands exactly on the listed tiles.  The caller's list memory is released right after each call (the context keeps its
    own pinned copy), and equal-hash-different-content cannot alias: a hit is compared with that copy."""
    import torch
    W, H = 96, 64
    tx = W // 8
    ctx = rt.Context(0)
    s0 = torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    rng = np.random.default_rng(3)
    frames, wants = [], []
    for k in range(300):
        n = int(rng.integers(1, 40))
        ids = rng.choice(tx * (H // 8), n, replace=False).astype(np.uint32)
        compact = torch.arange(n * 192, device="cuda:0", dtype=torch.float32) + 1000.0 * k
        frame = torch.zeros((H, W, 3), device="cuda:0")
        st = side.cuda_stream if k % 2 else s0
        if k % 2:
            side.wait_stream(torch.cuda.current_stream())          # (the tensors above were filled on the current stream)
        rt.tiles_copy_device(ctx, compact.data_ptr(), frame.data_ptr(), W, H, ids, True, st)
        ids_copy = ids.copy()
        ids[:] = 0                                                   # the caller's memory is dead after the call
        want = np.zeros((H, W, 3), np.float32)
        c = (np.arange(n * 192, dtype=np.float32) + np.float32(1000.0 * k)).reshape(n, 8, 8, 3)
        for j, g in enumerate(ids_copy):
            ty, txx = divmod(int(g), tx)
            want[ty * 8:ty * 8 + 8, txx * 8:txx * 8 + 8] = c[j]
        frames.append((frame, compact)); wants.append(want)
    torch.cuda.synchronize()
    for k, ((frame, _), want) in enumerate(zip(frames, wants)):
        assert np.array_equal(frame.cpu().numpy(), want), k


def test_multi_device_careful_mode_renders_the_same_frames(rt, monkeypatch):
    """RT_AMD_MULTI_CAREFUL=1 (read when the root context is created): rt_render_multi_device waits on the host after every
    phase.  Same frames as the asynchronous form, over three calls (interleaved ownership, balanced ownership, progressive
    frames on top)."""
    import torch
    objs, sky = rt.scenes.cube()
    W, H, spp, n = 104, 72, 3, 4
    cam, rd = rt.Camera(W, H), rt.RenderData(spp, 6, True, sky)
    so = rt.SceneObjects(objs)
    frames = {}
    for careful in ("0", "1"):
        monkeypatch.setenv("RT_AMD_MULTI_CAREFUL", careful)
        ctxs = [rt.Context(0) for _ in range(n)]
        scenes = [c.commit(so) for c in ctxs]
        st = torch.cuda.current_stream().cuda_stream
        frame = torch.zeros((H, W, 3), device="cuda:0")
        done = 0
        for times in ([11, 12], [13], [14, 15, 16]):
            rt.render_multi_device(ctxs, scenes, cam, rd, times, done, frame.data_ptr(), stream=st)
            done += len(times)
        torch.cuda.synchronize()
        for c in ctxs:
            c.synchronize()
        frames[careful] = frame.cpu().numpy()
    assert eq(frames["0"], frames["1"])
    single = rt.VariableRenderData(W, H)
    ctx = rt.Context(0)
    rt.render_frames(ctx, ctx.commit(so), cam, rd, single, [11, 12, 13, 14, 15, 16])
    assert eq(frames["1"], single.previous_render)


def test_preview_quality_costs_are_measured_again(rt):
    """a view's tile costs used to be measured once - a first launch at 1 sample per pixel (a preview, a profiler's warm-up)
    then fixed the schedule and the GPU ownership of every later launch.  Now figures from a launch with an eighth of the samples or
    fewer are provisional: the next full launch collects again, and the bounce limit is part of the view's key."""
    import torch
    objs, sky = rt.scenes.monkey()
    W, H = 160, 96
    ctx = rt.Context(0)
    scene = ctx.commit(rt.SceneObjects(objs))
    cam = rt.Camera(W, H)
    out = torch.zeros((H, W, 3), device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    rt.render_device(ctx, scene, cam, rt.RenderData(1, 8, True, sky), 5, 0, out.data_ptr(), stream=st)
    ids1, c1 = ctx.tile_costs()
    rt.render_device(ctx, scene, cam, rt.RenderData(64, 8, True, sky), 5, 0, out.data_ptr(), stream=st)    # runs on the 1-spp figures, measures again
    ids2, c2 = ctx.tile_costs()
    assert np.array_equal(ids1, ids2)
    heavy = (c1 >> 1) > 0
    assert ((c2 >> 1)[heavy] > 8 * (c1 >> 1)[heavy]).mean() > 0.9          # 64 samples' worth of work, not 1
    rt.render_device(ctx, scene, cam, rt.RenderData(64, 8, True, sky), 6, 0, out.data_ptr(), stream=st)    # same quality: the figures stand
    _, c3 = ctx.tile_costs()
    assert np.array_equal(c2, c3)
    rt.render_device(ctx, scene, cam, rt.RenderData(64, 2, True, sky), 6, 0, out.data_ptr(), stream=st)    # another bounce limit: another view
    _, c4 = ctx.tile_costs()
    assert not np.array_equal(c3, c4) and (c4 >> 1).sum() < (c3 >> 1).sum()


def test_render_multi_argument_errors(rt):
    objs, sky = rt.scenes.three_sphere()
    a, b = rt.Context(0), rt.Context(0)
    so = rt.SceneObjects(objs)
    sa, sb = a.commit(so), b.commit(so)
    cam, rd = rt.Camera(32, 32), rt.RenderData(1, 1, True, sky)
    data = rt.VariableRenderData(32, 32)
    with pytest.raises(ValueError):
        rt.render_multi([a, a], [sa, sa], cam, rd, data, [1])          # a context may appear once
    with pytest.raises(ValueError):
        rt.render_multi([a, b], [sa, sa], cam, rd, data, [1])          # scene committed on another context
    assert data.frame_num == 0


def test_cpp_multi_renderer(rt, models_dir, tmp_path):
    """host/raytracer.hpp MultiRenderer through example_main: device list "0,0,0" = three ranks on this box's
    GPU; the written image must be byte-identical to the single-Renderer run"""
    import subprocess
    bmod = importlib.import_module("ray-tracer_amd.build")
    exe = bmod.build_example()
    one, many = tmp_path / "one.ppm", tmp_path / "many.ppm"
    subprocess.check_call([exe, models_dir, "0", "96", "72", "3", str(one)], timeout=300, cwd=str(tmp_path))
    subprocess.check_call([exe, models_dir, "0", "96", "72", "3", str(many), "0,0,0"], timeout=300, cwd=str(tmp_path))
    assert one.read_bytes() == many.read_bytes() and len(one.read_bytes()) > 96 * 72 * 3


def test_config4_image_size(rt, orc, models_dir):
    """BASELINE.json configs[4]: the monkey scene at 3840x2160 tiled over 8 ranks (small spp here; the bench
    runs the 4096 spp).  (1) the 8-way band partition rendered with multi-frame launches into compact buffers
    reassembles bit-exactly to the single multi-frame launch; (2) the same through rt_render_multi_device with 8
    contexts; (3) six bands chosen at random equal the oracle's rows, frame by frame blended; (4) finite."""
    import torch
    dm = importlib.import_module("ray-tracer_amd.distributed")
    objs, sky = rt.scenes.monkey()
    W, H, spp, world, times = 3840, 2160, 2, 8, [12345, 12346]
    ctx = rt.Context(0)
    so = rt.SceneObjects(objs)
    scene = ctx.commit(so)
    cam, rd = rt.Camera(W, H), rt.RenderData(spp, 8, True, sky)
    stream = torch.cuda.current_stream().cuda_stream
    full = torch.zeros((H, W, 3), device="cuda:0")
    rt.render_device_batch(ctx, scene, cam, rd, times, 0, full.data_ptr(), stream=stream)
    stacked = torch.zeros((world, dm.max_owned_rows(H, 8, world), W, 3), device="cuda:0")
    for r in range(world):
        rt.render_device_batch(ctx, scene, cam, rd, times, 0, stacked[r].data_ptr(), band_first=r, band_stride=world, compact=True, stream=stream)
    torch.cuda.synchronize()
    frame = dm.assemble(stacked, W, H, 8, world).contiguous()
    assert torch.equal(frame.view(torch.int32), full.view(torch.int32))
    assert torch.isfinite(full).all()
    del stacked, frame
    ctxs = [ctx] + [rt.Context(0) for _ in range(world - 1)]
    scenes = [scene] + [c.commit(so) for c in ctxs[1:]]
    multi = torch.zeros((H, W, 3), device="cuda:0")
    rt.render_multi_device(ctxs, scenes, cam, rd, times, 0, multi.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    assert torch.equal(multi.view(torch.int32), full.view(torch.int32))
    host = full.cpu().numpy()
    o = orc.Scene(objs, orc.MATH_DET, models_dir)
    rng = np.random.default_rng(4)
    for band in sorted(rng.choice(H // 8, 6, replace=False)):
        y0, y1 = int(band) * 8, int(band) * 8 + 8
        f0 = o.render(cam.floats(), W, H, spp, 8, sky, time_ms=times[0], y0=y0, y1=y1)
        f1 = o.render(cam.floats(), W, H, spp, 8, sky, time_ms=times[1], frame_num=1, prev=f0, y0=y0, y1=y1)
        assert eq(host[y0:y1], f1[y0:y1]), band


@pytest.mark.parametrize("name", ["three_sphere", "cube", "monkey"])
def test_reference_camera_floats_reproduce_the_recorded_reference_frames(rt, orc, ctx, models_dir, golden_meta, name):
    """The 12 camera floats of the reference (src/camera.cu:46-60 evaluated with the glibc tanf its CPU build
    used; SURVEY.md App. A.12, committed as data in tests/golden/meta.json) passed VERBATIM through rt_camera:
    the HIP frame equals the oracle rendered with the same floats, and its sha256 is the one the survey
    recorded from the reference itself (App. C.2) - for all three config scenes, monkey included: at this
    size the only thing that separates det mode from the reference's libm build is the camera's tanf."""
    cam_floats = np.asarray(golden_meta["camera_libm"]["256x256"], np.float32)
    rec = golden_meta["reference_recorded_256x256_s16"][name]
    objs, sky = rt.scenes.CONFIG_SCENES[name]()
    scene = ctx.commit(rt.SceneObjects(objs))
    cam = rt.Camera(256, 256, floats=cam_floats)
    assert np.array_equal(cam.floats().view(np.uint32), cam_floats.view(np.uint32))
    data = rt.VariableRenderData(256, 256)
    rt.render(ctx, scene, cam, rt.RenderData(16, rec["limit"], True, sky), data, golden_meta["time_ms"])
    want = orc.Scene(objs, orc.MATH_DET, models_dir).render(cam_floats, 256, 256, 16, rec["limit"], sky, time_ms=golden_meta["time_ms"])
    assert eq(data.previous_render, want)
    assert hashlib.sha256(data.previous_render.tobytes()).hexdigest()[:16] == rec["sha256_prefix"]
    assert float("%.9g" % data.previous_render.mean(dtype=np.float64)) == rec["mean"]


def test_bvh_mesh_equals_brute_force_triangles(rt, ctx, models_dir, golden_meta):
    """SURVEY.md App. A.10 recorded of the reference: the monkey through its BVH and as 723 top-level triangles
    render bit-identically at 256x256, 16 spp, limit 8.  Same here on the GPU: the mesh kernel (compact reference
    tree in LDS) and the no-mesh kernel walking a 725-entry object list give the same frame, which is also the
    committed det-mode hash of that configuration."""
    import os
    objs, sky = rt.scenes.monkey()
    m = rt.ObjFileMesh(os.path.join(models_dir, "low_poly_monkey.obj"))
    for t in objs[0][2]:
        getattr(m, t[0])(*t[1:])
    tris = m.triangles().reshape(-1, 3, 3)
    brute = [("triangle", tuple(t[0]), tuple(t[1]), tuple(t[2]), objs[0][3]) for t in tris] + list(objs[1:])
    frames = []
    for description in (objs, brute):
        scene = ctx.commit(rt.SceneObjects(description))
        data = rt.VariableRenderData(256, 256)
        rt.render(ctx, scene, rt.Camera(256, 256), rt.RenderData(16, 8, True, sky), data, golden_meta["time_ms"])
        frames.append(data.previous_render.copy())
    assert eq(frames[0], frames[1])
    assert hashlib.sha256(frames[0].tobytes()).hexdigest() == golden_meta["sha256_256x256_s16"]["monkey"]["sha256"]


def test_flat_box_drop_quirk_on_the_device(rt, orc, ctx, models_dir):
    """a BVH leaf box of zero thickness is never entered (strict `tmin < tmax`), so the
    UNROTATED cube.obj loses most of its faces in the reference.  The HIP traversal must lose exactly the same
    ones: unrotated cube + ground + sky against the oracle, bit for bit, and it must differ from the same triangles
    rendered as top-level objects (which lose nothing)."""
    mat = ("standard", (0.8, 0.4, 0.2), 0)
    ground = ("sphere", (0, -100.5, 1.5), 100, ("standard", (0.5, 0.5, 0.5), 0))
    sky = (0.8, 1.0, 1.0)
    mesh = [("obj", "cube.obj", [("enlarge", 0.3), ("rotate", 0.0, 0.0, 0.0), ("translate", 0, 0, 1.8)], mat), ground]
    m = rt.ObjFileMesh(__import__("os").path.join(models_dir, "cube.obj"))
    m.enlarge(0.3); m.translate(0, 0, 1.8)
    brute = [("triangle", tuple(t[0]), tuple(t[1]), tuple(t[2]), mat) for t in m.triangles().reshape(-1, 3, 3)] + [ground]
    W, H, spp = 160, 120, 8
    frames = {}
    for key, description in (("mesh", mesh), ("brute", brute)):
        scene = ctx.commit(rt.SceneObjects(description))
        data = rt.VariableRenderData(W, H)
        rt.render(ctx, scene, rt.Camera(W, H), rt.RenderData(spp, 8, True, sky), data, 4242)
        want = orc.Scene(description, orc.MATH_DET, models_dir).render(rt.Camera(W, H).floats(), W, H, spp, 8, sky, time_ms=4242)
        assert eq(data.previous_render, want), key
        frames[key] = data.previous_render.copy()
    assert (frames["mesh"] != frames["brute"]).any(axis=2).sum() > 200        # the quirk is visible


def test_frames_per_launch_follow_the_memory_budget(rt, ctx):
    """rt_max_batch_frames: 32 planes of a 1080p or 2160p frame are nothing on a 288 GB GPU; 32 planes of the largest image the
    ABI accepts (2^28 pixels, 3.2 GB a plane) would be 103 GB, more than the quarter of the memory the planes may take"""
    assert ctx.max_batch_frames(1920, 1080) == 32 and ctx.max_batch_frames(3840, 2160) == 32
    big = ctx.max_batch_frames(32768, 8192)
    assert 1 <= big < 32


def test_more_frames_than_one_launch_holds(rt, orc, models_dir):
    """40 progressive frames in one call: rt_render_frames and rt_render_multi cut them into launches of at most 32
    frames (RT_MAX_BATCH_FRAMES), the second launch continuing from the first one's image; both must equal the
    oracle's 40-frame loop"""
    objs, sky = rt.scenes.cube()
    W, H, spp = 64, 48, 2
    times = list(range(1000, 1040))
    want = oracle_progressive(orc, objs, models_dir, rt.Camera(W, H).floats(), W, H, spp, 8, sky, times)
    ctxs = [rt.Context(0) for _ in range(2)]
    so = rt.SceneObjects(objs)
    scenes = [c.commit(so) for c in ctxs]
    cam, rd = rt.Camera(W, H), rt.RenderData(spp, 8, True, sky)
    one = rt.VariableRenderData(W, H)
    rt.render_frames(ctxs[0], scenes[0], cam, rd, one, times)
    assert one.frame_num == 40 and eq(one.previous_render, want)
    many = rt.VariableRenderData(W, H)
    rt.render_multi(ctxs, scenes, cam, rd, many, times)
    assert many.frame_num == 40 and eq(many.previous_render, want)


def test_nan_texture_coordinate_at_a_sphere_pole(rt, orc, ctx, models_dir):
    """Found by tests/soak/soak_parity.py (seed 7895, one pixel in 3,000 random scenes): at a sphere's pole
    (P.y - c.y) / r can exceed 1 by an ulp, asin gives NaN, and the image / checkerboard lookups convert a NaN u to
    int - undefined in C (x86: INT_MIN), 0 on CUDA and on gfx950.  rt_math.h's rt_f2i defines it (NaN -> 0,
    saturating) for the oracle and the kernel alike; this is the scene and the pixel that differed."""
    rng = np.random.default_rng(5)
    img = rng.uniform(0, 1, (7, 4, 3)).astype(np.float32)
    sky = (0.8, 1.0, 1.0)
    centre, radius = (-0.5656471126252296, -0.4661420880950526, 2.246162948711449), 0.11503499970308974
    W, H = 96, 80
    for mat in (("image", img, 0.0), ("checkerboard", (0.9, 0.8, 0.1), (0.1, 0.2, 0.3), 7, 0.0), ("gradient", 0.0)):
        objs = [("sphere", centre, radius, mat)]
        scene = ctx.commit(rt.SceneObjects(objs))
        for t in (7895, 7896, 7897, 1, 2):
            data = rt.VariableRenderData(W, H)
            rt.render(ctx, scene, rt.Camera(W, H), rt.RenderData(2, 2, True, sky), data, t)
            want = orc.Scene(objs, orc.MATH_DET, models_dir).render(rt.Camera(W, H).floats(), W, H, 2, 2, sky, time_ms=t)
            assert eq(data.previous_render, want), (mat[0], t)       # bit for bit, NaN pixels included (one canonical NaN)
    # seed 24964 of the soak: the gradient colour IS the NaN coordinate, so the pixel itself is NaN - on both sides
    objs = [("sphere", (-0.6021626149655434, 0.5782768403342793, 1.9494934676077007), 0.28375946780666705, ("gradient", 0.11130362971519248))]
    scene = ctx.commit(rt.SceneObjects(objs))
    data = rt.VariableRenderData(128, 72)
    rt.render(ctx, scene, rt.Camera(128, 72), rt.RenderData(4, 4, True, (0.0, 0.0, 0.0)), data, 24965)
    want = orc.Scene(objs, orc.MATH_DET, models_dir).render(rt.Camera(128, 72).floats(), 128, 72, 4, 4, (0.0, 0.0, 0.0), time_ms=24965)
    assert np.isnan(want).sum() >= 1 and eq(data.previous_render, want)
    assert (data.previous_render.view(np.uint32)[np.isnan(data.previous_render)] == 0x7fc00000).all()
