"""Image-tile data parallelism across the GPUs of one node.

A pixel depends only on (x, y, W, time_ms, frame_num, prev[pixel]) — reference
src/raytracer.cu:118-131 — so any partition of the image reproduces the single-GPU frame bit
for bit, and the only exchange is one gather of what the ranks rendered to rank 0 (RCCL over xGMI
when the process group's backend is "nccl").  Two ways of cutting the image:

* bands (round 1-2): rows are cut into bands of ``band_rows`` rows, rank r of ``world`` renders the
  bands b with b % world == r into a compact buffer; de-interleaving is one permute + reshape;
* tile lists (round 3, what bench.py uses for N > 1): every rank owns a list of 8x8 tiles.  The
  view's first launch runs on an interleaved ownership and measures what every tile costs; the
  costs and peak pixel costs are merged over the ranks (one small all-reduce, the only other collective), every rank
  computes the same longest-processing-time-first ownership from them (``rt_partition_tiles``) and
  from then on the ranks finish together.  A rank's compact image is its tiles back to back; rank 0
  puts each gathered image into the frame with ``rt_tiles_copy_device``.

One process per GPU; the process group is whatever ``torch.distributed`` was initialised
with (``gloo`` in the CPU tests).
"""
import importlib

import numpy as np
import torch
import torch.distributed as dist


def _rt():
    return importlib.import_module(__package__)


# ---- bands ----------------------------------------------------------------------------------------
def num_bands(height, band_rows):
    return (height + band_rows - 1) // band_rows


def owned_bands(height, band_rows, rank, world):
    return list(range(rank, num_bands(height, band_rows), world))


def max_owned_rows(height, band_rows, world):
    """rows in the largest per-rank compact buffer (rank 0 owns the most bands)"""
    return len(owned_bands(height, band_rows, 0, world)) * band_rows


def assemble(stacked, width, height, band_rows, world):
    """stacked: [world, max_owned_rows, W, 3] (rank-major compact buffers, padded to equal
    size) -> full frame [H, W, 3].  Band b = k*world + r sits at stacked[r, k]."""
    kmax = stacked.shape[1] // band_rows
    x = stacked.reshape(world, kmax, band_rows, width, 3).permute(1, 0, 2, 3, 4)
    return x.reshape(kmax * world * band_rows, width, 3)[:height]


def gather_frame(local, width, height, band_rows, rank, world, dst=0, group=None, out=None):
    """local: this rank's compact buffer [max_owned_rows, W, 3] (rows past the rank's own
    bands are padding).  Returns the assembled frame on ``dst`` and None elsewhere."""
    if world == 1:
        return local[:height] if band_rows * num_bands(height, band_rows) != height else local
    stacked = _gather(local, rank, world, dst, group, out)
    if rank != dst:
        return None
    return assemble(stacked, width, height, band_rows, world)


def _gather(local, rank, world, dst, group, out):
    """dist.gather of equally sized buffers -> [world, ...] on dst, None elsewhere"""
    if dist.get_backend(group) == "gloo" and local.is_cuda:
        # rehearsal mode (several ranks sharing one GPU, which RCCL refuses): gloo moves host memory
        host = local.cpu()
        host_out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype) if rank == dst else None
        dist.gather(host, gather_list=list(host_out.unbind(0)) if rank == dst else None, dst=dst, group=group)
        return host_out.to(local.device) if rank == dst else None
    gather_list = None
    if rank == dst:
        if out is None:
            out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        gather_list = list(out.unbind(0))
    dist.gather(local, gather_list=gather_list, dst=dst, group=group)
    return out if rank == dst else None


# ---- tile lists -----------------------------------------------------------------------------------
def tiles_xy(width, height):
    return (width + 7) // 8, (height + 7) // 8


def tile_lists(owner, world):
    """owner[tile] = rank -> per rank, the image indices of its tiles in ascending order (uint32)"""
    owner = np.asarray(owner)
    return [np.flatnonzero(owner == r).astype(np.uint32) for r in range(world)]


def compact_floats(lists):
    """floats in the largest rank's compact image (every rank pads to it for the gather)"""
    return max(1, max(len(l) for l in lists)) * 192


def initial_ownership(width, height, world):
    """interleaved ownership of a view's first, cost-collecting launch"""
    return _rt().partition_tiles(width, height, world)


def balanced_ownership(ctx, width, height, lists, rank, world, device=None, group=None):
    """After this rank's context has rendered its tiles of the interleaved ownership once: combines every rank's
    measured tile figures (one all-reduce of a [2, tiles] int64 tensor: 520 KB at 1920x1080; every tile was measured by
    exactly one rank, so the sum is a merge) and returns (owner, cost, peak) - the same on every rank.  cost[tile] is what
    the tile's pixels cost, owner the longest-processing-time-first ownership computed from it; peak[tile] is the cost of
    the tile's most expensive pixel, which is what a launch orders its jobs by."""
    rt = _rt()
    tx, ty = tiles_xy(width, height)
    both = np.zeros((2, tx * ty), np.int64)
    ids, c, pk = ctx.tile_costs(with_peaks=True)
    assert np.array_equal(ids, lists[rank]), "the context's current view is not this rank's tile list"
    both[0, ids] = c
    both[1, ids] = pk
    if world > 1:
        t = torch.from_numpy(both)
        if dist.get_backend(group) != "gloo":
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        both = t.cpu().numpy()
    both = np.minimum(both, 0xffffffff).astype(np.uint32)
    return rt.partition_tiles(width, height, world, both[0]), both[0], both[1]


def scatter_tiles(frame, compact, tile_ids, width, height, ctx=None, stream=None):
    """frame[H, W, 3] <- a compact tile-list image (tile k of tile_ids at floats [192 k, 192 k + 192)).  On the GPU
    this is the library's kernel (rt_tiles_copy_device); host tensors (the CPU tests) go through an index."""
    tile_ids = np.asarray(tile_ids, np.uint32)
    if tile_ids.size == 0:
        return
    if frame.is_cuda:
        _rt().tiles_copy_device(ctx, compact.data_ptr(), frame.data_ptr(), width, height, tile_ids, True, stream)
        return
    tx, _ = tiles_xy(width, height)
    within = np.arange(64)
    x = (tile_ids[:, None].astype(np.int64) % tx) * 8 + (within & 7)[None, :]
    y = (tile_ids[:, None].astype(np.int64) // tx) * 8 + (within >> 3)[None, :]
    ok = (x < width) & (y < height)
    src = compact.reshape(-1)[:tile_ids.size * 192].reshape(tile_ids.size, 64, 3)
    frame.reshape(-1, 3)[torch.from_numpy((y * width + x)[ok])] = src[torch.from_numpy(ok)]


def gather_tiles(local, lists, width, height, rank, world, ctx=None, dst=0, group=None, out=None, frame=None, stream=None):
    """local: this rank's compact image, a flat float32 tensor padded to compact_floats(lists).  On ``dst``: the
    frame [H, W, 3] with every rank's tiles in place (``frame`` is reused when given); None elsewhere."""
    if world == 1:
        stacked = local.unsqueeze(0)
    else:
        stacked = _gather(local, rank, world, dst, group, out)
    if rank != dst:
        return None
    if frame is None:
        frame = torch.empty((height, width, 3), dtype=local.dtype, device=local.device)
    for r in range(world):
        scatter_tiles(frame, stacked[r], lists[r], width, height, ctx=ctx, stream=stream)
    return frame
