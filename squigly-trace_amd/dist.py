"""Multi-GPU rendering: one process per GPU, rows sharded in interleaved blocks, one gather.

Pixels are independent and every sample's seed depends only on (x, y, spp, w) (src/Lib.hs:85-86),
so any partition of the rows reproduces the single-GPU image bit for bit; the only exchange is one
framebuffer all_gather at the end of the frame (RCCL over xGMI; `gloo` on CPU for tests).
This plays the role of massiv's `Par` scheduler (src/Lib.hs:73) one level up.
"""
import torch
import torch.distributed as dist

# Rows per block of the interleaved partition.  Measured on the headline frame at 8 ranks (tools/gpu_share_balance.py, every shard timed
# alone on one MI355X, profiles/r03i_share_balance.txt): slowest shard 8.62 / 8.56 / 8.67 / 8.83 / 9.00 / 10.39 ms with blocks of
# 1 / 2 / 4 / 8 / 16 / 32 rows (mean 8.5 ms throughout) -- the frame time of a strong-scaling run is the slowest rank's.
ROW_BLOCK = 2


def shard_rows(w, row_block, rank, world):
    """Global row indices owned by `rank`, in local order (mirrors sq_shard_global_row)."""
    rows = []
    for b in range(rank, (w + row_block - 1) // row_block, world):
        rows.extend(range(b * row_block, min((b + 1) * row_block, w)))
    return rows


class FramePlan:
    """Everything about one (w, row_block, world) partition that does not change from frame to frame.

    A frame is gathered as ONE all_gather_into_tensor of equal buffers [mx, h, C] (mx = the largest shard; shorter
    shards are padded) into [world * mx, h, C], followed by ONE index_select that de-interleaves the rows:
    frame[y] = gathered[src[y]] with src precomputed on the device.  Nothing here runs per frame on the host.
    """

    def __init__(self, w, row_block, world, device):
        self.w, self.row_block, self.world = w, row_block, world
        shards = [shard_rows(w, row_block, r, world) for r in range(world)]
        self.counts = [len(s) for s in shards]
        self.mx = max(self.counts) if world else 0
        src = torch.empty(w, dtype=torch.long)
        for r, rows in enumerate(shards):
            if rows:
                src[torch.tensor(rows, dtype=torch.long)] = r * self.mx + torch.arange(len(rows), dtype=torch.long)
        self.src = src.to(device)
        self._buffers = {}

    def buffers(self, h, c, dtype, device):
        """(send [mx, h, c], gathered [world * mx, h, c]): allocated once per frame shape."""
        key = (h, c, dtype)
        if key not in self._buffers:
            self._buffers[key] = (torch.zeros((self.mx, h, c), dtype=dtype, device=device),
                                  torch.empty((self.world * self.mx, h, c), dtype=dtype, device=device))
        return self._buffers[key]


_plans = {}


def plan_for(w, row_block, world, device):
    key = (w, row_block, world, str(device))
    if key not in _plans:
        _plans[key] = FramePlan(w, row_block, world, device)
    return _plans[key]


def gather_frame(local, w, row_block=ROW_BLOCK, group=None):
    """all_gather the per-rank compact row buffers [rows_r, h, C] and de-interleave into [w, h, C].

    Ranks may own different row counts (last block ragged): buffers are padded to the largest.
    Every rank returns the full frame.
    """
    initialised = dist.is_initialized()
    world = dist.get_world_size(group) if initialised else 1
    rank = dist.get_rank(group) if initialised else 0
    if not initialised:
        assert local.shape[0] == w, (local.shape, w)
        return local
    plan = plan_for(w, row_block, world, local.device)
    assert local.shape[0] == plan.counts[rank], (tuple(local.shape), plan.counts, rank)
    h, c = local.shape[1], local.shape[2]
    send, gathered = plan.buffers(h, c, local.dtype, local.device)
    if local.shape[0] == plan.mx and local.is_contiguous():
        # The common case: no staging copy, the collective reads the renderer's output tensor itself.  That is safe because
        # the collective is enqueued on (or ordered after) the stream the render was enqueued on, and because `local` is a
        # tensor render_rows allocated for THIS frame: nothing rewrites it before the collective has read it.  A caller that
        # passes its own reused buffer (render_rows(out_rgb=...)) must not enqueue the next render into it on another stream
        # before this call's result has been consumed.
        send = local
    else:
        send[: local.shape[0]].copy_(local)
    dist.all_gather_into_tensor(gathered.view(-1), send.view(-1), group=group)
    return gathered.index_select(0, plan.src)


def render_frame(device_scene, cam, samples, w, h, cast=False, row_block=ROW_BLOCK, want="rgb", group=None, events=None):
    """Render this rank's rows on its GPU and gather the frame ([w, h, 3], on every rank).

    events: a list that receives (start, rendered, gathered) torch.cuda.Event triples, one per call, recorded on the
    current stream (bench.py turns them into this rank's render time and gather time per frame)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    ev = None
    if events is not None:
        ev = tuple(torch.cuda.Event(enable_timing=True) for _ in range(3))
        ev[0].record()
    avg, rgb = device_scene.render_rows(cam, samples, w, h, cast=cast, shard=(row_block, rank, world),
                                        want_avg=(want == "avg"), want_rgb=(want == "rgb"))
    if ev is not None:
        ev[1].record()
    frame = gather_frame(avg if want == "avg" else rgb, w, row_block, group)
    if ev is not None:
        ev[2].record()
        events.append(ev)
    return frame
