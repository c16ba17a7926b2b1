"""Multi-GPU rendering: one process per GPU, rows sharded in interleaved blocks, one gather.

Pixels are independent and every sample's seed depends only on (x, y, spp, w) (src/Lib.hs:85-86),
so any partition of the rows reproduces the single-GPU image bit for bit; the only exchange is one
framebuffer all_gather at the end of the frame (RCCL over xGMI; `gloo` on CPU for tests).
This plays the role of massiv's `Par` scheduler (src/Lib.hs:73) one level up.
"""
import torch
import torch.distributed as dist

from . import _native as N

ROW_BLOCK = 8   # rows per block: small enough to balance the very non-uniform image, large enough for whole tiles


def shard_rows(w, row_block, rank, world):
    """Global row indices owned by `rank`, in local order (mirrors sq_shard_global_row)."""
    rows = []
    for b in range(rank, (w + row_block - 1) // row_block, world):
        rows.extend(range(b * row_block, min((b + 1) * row_block, w)))
    return rows


def gather_frame(local, w, row_block=ROW_BLOCK, group=None):
    """all_gather the per-rank compact row buffers [rows_r, h, C] and de-interleave into [w, h, C].

    Ranks may own different row counts (last block ragged): buffers are padded to the largest.
    Every rank returns the full frame.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    h, c = local.shape[1], local.shape[2]
    counts = [len(shard_rows(w, row_block, r, world)) for r in range(world)]
    assert local.shape[0] == counts[rank], (local.shape, counts, rank)
    if world == 1 and not dist.is_initialized():
        return local
    mx = max(counts)
    padded = local
    if local.shape[0] < mx:
        padded = torch.zeros((mx, h, c), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded.contiguous(), group=group)
    frame = torch.empty((w, h, c), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = torch.tensor(shard_rows(w, row_block, r, world), dtype=torch.long, device=local.device)
        if len(idx):
            frame[idx] = parts[r][: counts[r]]
    return frame


def render_frame(device_scene, cam, samples, w, h, cast=False, row_block=ROW_BLOCK, want="rgb", group=None):
    """Render this rank's rows on its GPU and gather the frame ([w, h, 3], on every rank)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    avg, rgb = device_scene.render_rows(cam, samples, w, h, cast=cast, shard=(row_block, rank, world),
                                        want_avg=(want == "avg"), want_rgb=(want == "rgb"))
    return gather_frame(avg if want == "avg" else rgb, w, row_block, group)
