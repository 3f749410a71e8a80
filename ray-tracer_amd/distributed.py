"""Image-tile data parallelism across the GPUs of one node.

A pixel depends only on (x, y, W, time_ms, frame_num, prev[pixel]) — reference
src/raytracer.cu:118-131 — so any partition of the image reproduces the single-GPU frame bit
for bit.  Rows are cut into bands of ``band_rows`` rows; rank r of ``world`` renders the bands
b with b % world == r (interleaved, because work is concentrated where the geometry is) into a
compact buffer, and the only exchange is one gather of those buffers to rank 0 (RCCL over
xGMI when the process group's backend is "nccl"; 12.4 MB per GPU at 3840x2160), followed by a
de-interleave that is a single permute + reshape.

One process per GPU; the process group is whatever ``torch.distributed`` was initialised
with (``gloo`` in the CPU tests).
"""
import torch
import torch.distributed as dist


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
    if dist.get_backend(group) == "gloo" and local.is_cuda:
        # rehearsal mode (several ranks sharing one GPU, which RCCL refuses): gloo moves host memory
        host = local.cpu()
        host_out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype) if rank == dst else None
        dist.gather(host, gather_list=list(host_out.unbind(0)) if rank == dst else None, dst=dst, group=group)
        if rank != dst:
            return None
        return assemble(host_out.to(local.device), width, height, band_rows, world)
    gather_list = None
    if rank == dst:
        if out is None:
            out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        gather_list = list(out.unbind(0))
    dist.gather(local, gather_list=gather_list, dst=dst, group=group)
    if rank != dst:
        return None
    return assemble(out, width, height, band_rows, world)
