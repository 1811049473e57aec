"""Image-space sharding of one frame over the ranks of a node.

Paths are independent and seeded by absolute pixel index (camera/camera.cpp:73), the scene is
read-only and with the default box film every sample lands in its own pixel (camera/film.cpp:13-17),
so rows are dealt round-robin in groups of `SHARD_ROWS` (interleaving balances cheap and expensive
image regions) and the only collective of the data path is ONE gather of the packed rows to rank 0.
Works on any torch.distributed backend: "nccl" (= RCCL over xGMI) on the GPUs, "gloo" in CPU tests.

Frames with a reconstruction filter (camera/film.cpp:61-79) are the one case with a real exchange step: a sample splats
into the pixels within the filter radius, i.e. into neighbouring shards' rows too, so every rank accumulates a full-frame
{rgb_sum, weight_sum} buffer (mcrt_render_film_device), the buffers are summed onto rank 0 with ONE reduce
(`reduce_film`) and rank 0 resolves the sum (mcrt_film_resolve_device).
"""
import numpy as np

SHARD_ROWS = 8


def shard_camera(cam, rank, world, shard_rows=SHARD_ROWS):
    c = cam.copy()
    c.shard_index, c.shard_count, c.shard_rows = int(rank), int(world), int(shard_rows)
    return c


def rows_of(cam, rank, world, shard_rows=SHARD_ROWS):
    """Row indices owned by `rank` (same rule as mcrt_shard_rows in the C ABI)."""
    y = np.arange(cam.height, dtype=np.int64)
    if world <= 1:
        return y
    return y[(y // shard_rows) % world == rank]


def max_rows(cam, world, shard_rows=SHARD_ROWS):
    return max(len(rows_of(cam, r, world, shard_rows)) for r in range(world))


def gather_frame(tile, cam, rank, world, dist=None, gather_list=None, shard_rows=SHARD_ROWS):
    """tile: torch tensor [max_rows, W, 3] float64 holding this rank's packed rows (padding rows
    ignored). Returns the assembled [H, W, 3] frame on rank 0 (None elsewhere)."""
    import torch

    if world > 1:
        if rank == 0 and gather_list is None:
            gather_list = [torch.empty_like(tile) for _ in range(world)]
        dist.gather(tile, gather_list if rank == 0 else None, dst=0)
        if rank != 0:
            return None
        parts = gather_list
    else:
        parts = [tile]
    frame = torch.zeros((cam.height, cam.width, 3), dtype=tile.dtype, device=tile.device)
    for r in range(world):
        rows = torch.from_numpy(rows_of(cam, r, world, shard_rows)).to(tile.device)
        frame[rows] = parts[r][: len(rows)]
    return frame


def reduce_film(rgbw, rank, world, dist=None):
    """rgbw: torch tensor [H, W, 4] float64 — this rank's splat sums over the FULL frame. After the call rank 0's tensor
    holds the sum over all ranks (other ranks' tensors are unspecified). Returns rgbw on rank 0, None elsewhere."""
    if world > 1:
        dist.reduce(rgbw, dst=0, op=dist.ReduceOp.SUM)
    return rgbw if rank == 0 else None
