"""The N>1 path on CPU: two processes, gloo backend.  Each rank produces the bands it owns
(with the oracle standing in for the GPU kernel, which is exactly what the partition math must
be indifferent to), the compact buffers are gathered to rank 0 with
ray-tracer_amd.distributed.gather_frame, and the assembled frame must equal the oracle's
single-process frame bit for bit."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, W, H, spp, band_rows, result_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rt = importlib.import_module("ray-tracer_amd")
        dm = importlib.import_module("ray-tracer_amd.distributed")
        from oracle import binding as B
        objs, sky = rt.scenes.cube()
        sc = B.Scene(objs, B.MATH_DET, rt.scenes.models_dir())
        cam = B.camera_default(W, H, B.MATH_DET)
        local = torch.zeros((dm.max_owned_rows(H, band_rows, world), W, 3), dtype=torch.float32)
        for k, b in enumerate(dm.owned_bands(H, band_rows, rank, world)):
            y0, y1 = b * band_rows, min((b + 1) * band_rows, H)
            rows = sc.render(cam, W, H, spp, 8, sky, y0=y0, y1=y1, nthreads=2)[y0:y1]
            local[k * band_rows:k * band_rows + (y1 - y0)] = torch.from_numpy(rows.copy())
        frame = dm.gather_frame(local, W, H, band_rows, rank, world, dst=0)
        if rank == 0:
            np.save(result_path, frame.contiguous().numpy())
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("W,H,band_rows", [(96, 60, 8), (64, 40, 16)])
def test_two_rank_gather_reassembles_the_frame(tmp_path, W, H, band_rows):
    from oracle import binding as B
    B.build()
    rt = importlib.import_module("ray-tracer_amd")
    world, spp = 2, 3
    path = str(tmp_path / "frame.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, W, H, spp, band_rows, path), nprocs=world, join=True)
    got = np.load(path)
    objs, sky = rt.scenes.cube()
    want = B.Scene(objs, B.MATH_DET, rt.scenes.models_dir()).render(B.camera_default(W, H, B.MATH_DET), W, H, spp, 8, sky)
    assert got.shape == (H, W, 3)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


class _CostsFromPixels:
    """stands in for a GPU context in the CPU test: 'measured' tile costs derived from the tiles' pixels"""

    def __init__(self, ids, compact):
        self.ids, self.compact = ids, compact

    def tile_costs(self, with_peaks=False):
        c = self.compact[:len(self.ids) * 192].reshape(len(self.ids), 192)
        cost = (np.abs(np.nan_to_num(c)).sum(axis=1) * 100).astype(np.uint32) + 1
        return (self.ids, cost, (np.abs(np.nan_to_num(c)).max(axis=1) * 100).astype(np.uint32)) if with_peaks else (self.ids, cost)


def _tile_worker(rank, world, port, W, H, spp, result_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rt = importlib.import_module("ray-tracer_amd")
        dm = importlib.import_module("ray-tracer_amd.distributed")
        from oracle import binding as B
        objs, sky = rt.scenes.cube()
        sc = B.Scene(objs, B.MATH_DET, rt.scenes.models_dir())
        cam = B.camera_default(W, H, B.MATH_DET)
        tx, ty = dm.tiles_xy(W, H)
        # the oracle stands in for the kernel: render the frame, padded to whole tiles, and cut this rank's tiles out
        full = np.zeros((ty * 8, tx * 8, 3), np.float32)
        full[:H, :W] = sc.render(cam, W, H, spp, 8, sky, nthreads=2)
        tiles = full.reshape(ty, 8, tx, 8, 3).transpose(0, 2, 1, 3, 4).reshape(tx * ty, 192)

        def my_image(lists):
            buf = np.zeros(dm.compact_floats(lists), np.float32)
            buf[:len(lists[rank]) * 192] = tiles[lists[rank]].reshape(-1)
            return buf

        # pass 1: interleaved ownership; "measure"; pass 2: cost-balanced ownership, the same on both ranks
        lists = dm.tile_lists(dm.initial_ownership(W, H, world), world)
        owner, cost, peak = dm.balanced_ownership(_CostsFromPixels(lists[rank], my_image(lists)), W, H, lists, rank, world)
        assert peak.shape == cost.shape and (peak.astype(np.int64) <= cost.astype(np.int64)).all()
        lists2 = dm.tile_lists(owner, world)
        loads = [int(cost[l].sum()) for l in lists2]
        assert max(loads) - min(loads) <= int(cost.max()), loads          # LPT: within one job of each other
        frame = dm.gather_tiles(torch.from_numpy(my_image(lists2)), lists2, W, H, rank, world, dst=0)
        if rank == 0:
            np.save(result_path, frame.contiguous().numpy())
            np.save(result_path + ".owner.npy", owner)
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("W,H", [(96, 60), (70, 41)])
def test_two_rank_cost_balanced_tile_lists(tmp_path, W, H):
    """the round-3 N > 1 path on CPU: interleaved tile ownership, all-reduced costs, LPT ownership computed identically
    on both ranks, padded gather, tiles scattered into the frame (ragged edge tiles included)"""
    from oracle import binding as B
    B.build()
    rt = importlib.import_module("ray-tracer_amd")
    world, spp = 2, 2
    path = str(tmp_path / "frame.npy")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_tile_worker, args=(world, port, W, H, spp, path), nprocs=world, join=True)
    got = np.load(path)
    objs, sky = rt.scenes.cube()
    want = B.Scene(objs, B.MATH_DET, rt.scenes.models_dir()).render(B.camera_default(W, H, B.MATH_DET), W, H, spp, 8, sky)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    owner = np.load(path + ".owner.npy")
    assert sorted(set(owner.tolist())) == [0, 1]


def test_partition_tiles_is_lpt_and_deterministic():
    rt = importlib.import_module("ray-tracer_amd")
    rng = np.random.default_rng(3)
    W, H = 200, 120
    n = ((W + 7) // 8) * ((H + 7) // 8)
    inter = rt.partition_tiles(W, H, 3)
    tx = (W + 7) // 8
    assert all(inter[i] == ((i % tx) + (i // tx)) % 3 for i in range(n))
    cost = (rng.pareto(1.5, n) * 1000).astype(np.uint32) + 1
    for world in (1, 2, 5, 8):
        a, b = rt.partition_tiles(W, H, world, cost), rt.partition_tiles(W, H, world, cost.copy())
        assert np.array_equal(a, b) and a.min() >= 0 and a.max() < world
        loads = np.array([cost[a == r].astype(np.int64).sum() for r in range(world)])
        # longest-processing-time-first: no rank exceeds the mean by more than the largest job
        assert loads.max() <= cost.astype(np.int64).sum() / world + cost.max()
    # the reference implementation of the rule, in python
    order = sorted(range(n), key=lambda i: (-int(cost[i]), i))
    load, own = [0] * 4, [0] * n
    for i in order:
        r = min(range(4), key=lambda k: (load[k], k))
        own[i] = r
        load[r] += int(cost[i])
    assert rt.partition_tiles(W, H, 4, cost).tolist() == own
    with pytest.raises(ValueError):
        rt.partition_tiles(W, H, 0)


def test_band_ownership_math():
    dm = importlib.import_module("ray-tracer_amd.distributed")
    rt = importlib.import_module("ray-tracer_amd")
    for H, rows, world in ((1080, 8, 8), (2160, 8, 8), (1080, 8, 3), (203, 8, 2), (64, 16, 4), (8, 8, 8)):
        seen = []
        for r in range(world):
            own = dm.owned_bands(H, rows, r, world)
            assert rt.tile_owned_rows(H, rows, r, world) == len(own) * rows
            seen += own
        assert sorted(seen) == list(range(dm.num_bands(H, rows)))
        assert dm.max_owned_rows(H, rows, world) == max(len(dm.owned_bands(H, rows, r, world)) for r in range(world)) * rows


def test_assemble_is_the_inverse_of_the_partition():
    dm = importlib.import_module("ray-tracer_amd.distributed")
    W, H, rows, world = 5, 52, 8, 3
    full = torch.arange(H * W * 3, dtype=torch.float32).reshape(H, W, 3)
    stacked = torch.zeros((world, dm.max_owned_rows(H, rows, world), W, 3))
    for r in range(world):
        for k, b in enumerate(dm.owned_bands(H, rows, r, world)):
            y0, y1 = b * rows, min((b + 1) * rows, H)
            stacked[r, k * rows:k * rows + (y1 - y0)] = full[y0:y1]
    assert torch.equal(dm.assemble(stacked, W, H, rows, world), full)
