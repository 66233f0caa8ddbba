"""
Multi-GPU rendering: one process per GPU, rays sharded, one RCCL all-gather of rendered tiles.

The reference's only parallelism is a single-process ``torch.nn.DataParallel(dim=1)`` over rays
that re-broadcasts ~113 MB of parameters plus the latent on every call (src/render/nerf.py:373-377,
SURVEY.md 2.3).  Here every rank keeps persistent weights and per-scene state, renders a
contiguous slice of the rays of a frame, and the frame is assembled with ONE all-gather of
(rays/G, 4) [rgb, depth] tiles -- bandwidth-trivial (320 KB/rank at 400x400), so it is issued
as a single collective on xGMI rather than bucketed.  With ``backend="nccl"`` (= RCCL on ROCm)
the tensors stay on the GPU; the ``gloo`` backend is used by the CPU tests of the sharding logic.
Scene-per-GPU mode (BASELINE config 5) needs no exchange at all.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world_size, rank, multiple=64):
    """Contiguous [lo, hi) slice of n rays for `rank`; every shard except the last is a multiple
    of the MLP tile (64 samples) and all shards are padded to the same length for all_gather."""
    per = -(-n // world_size)
    per = -(-per // multiple) * multiple
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi, per


def gather_tiles(tile, n, per, group=None):
    """all_gather equal-sized (per, C) tiles and trim to n rows."""
    world = dist.get_world_size(group)
    if tile.shape[0] < per:
        pad = torch.zeros(per - tile.shape[0], *tile.shape[1:], device=tile.device, dtype=tile.dtype)
        tile = torch.cat([tile, pad], 0)
    out = torch.empty(world * per, *tile.shape[1:], device=tile.device, dtype=tile.dtype)
    dist.all_gather_into_tensor(out, tile.contiguous(), group=group) if tile.is_cuda else \
        dist.all_gather(list(out.chunk(world, 0)), tile.contiguous(), group=group)
    return out[:n]


def render_sharded(render_fn, rays, group=None):
    """rays (N, 8) identical on every rank -> (rgb (N,3), depth (N)) on every rank.
    render_fn(rays_slice (n_i, 8)) -> (rgb (n_i,3), depth (n_i))."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = rays.shape[0]
    lo, hi, per = shard_bounds(n, world, rank)
    if hi > lo:
        rgb, depth = render_fn(rays[lo:hi])
        tile = torch.cat([rgb, depth[:, None]], dim=1)
    else:
        tile = torch.zeros(0, 4, device=rays.device, dtype=torch.float32)
    full = gather_tiles(tile, n, per, group)
    return full[:, :3], full[:, 3]
