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


def _world(group=None):
    """(world_size, rank); (1, 0) when torch.distributed is not initialised (single-process use)."""
    if not dist.is_available() or not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


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
    world, _ = _world(group)
    if world == 1:
        return tile[:n]
    if tile.shape[0] < per:
        pad = torch.zeros(per - tile.shape[0], *tile.shape[1:], device=tile.device, dtype=tile.dtype)
        tile = torch.cat([tile, pad], 0)
    if tile.is_cuda and dist.get_backend(group) != "gloo":
        out = torch.empty(world * per, *tile.shape[1:], device=tile.device, dtype=tile.dtype)
        dist.all_gather_into_tensor(out, tile.contiguous(), group=group)      # RCCL, tensors stay on the GPU
        return out[:n]
    # gloo (CPU tests, and multi-process tests that share one GPU, where RCCL refuses duplicate devices)
    host = tile.detach().cpu().contiguous()
    out = torch.empty(world * per, *tile.shape[1:], dtype=tile.dtype)
    dist.all_gather(list(out.chunk(world, 0)), host, group=group)
    return out[:n].to(tile.device)


def render_sharded(render_fn, rays, group=None):
    """rays (N, 8) identical on every rank -> (rgb (N,3), depth (N)) on every rank.
    render_fn(rays_slice (n_i, 8)) -> (rgb (n_i,3), depth (n_i))."""
    world, rank = _world(group)
    n = rays.shape[0]
    lo, hi, per = shard_bounds(n, world, rank)
    if hi > lo:
        rgb, depth = render_fn(rays[lo:hi])
        tile = torch.cat([rgb, depth[:, None]], dim=1)
    else:
        tile = torch.zeros(0, 4, device=rays.device, dtype=torch.float32)
    full = gather_tiles(tile, n, per, group)
    return full[:, :3], full[:, 3]


def render_frame_sharded(render_fn, pose, width, height, focal, z_near, z_far, c=None, group=None, yolo=False,
                         device=None):
    """One target view rendered by all ranks: rank r GENERATES rays [lo_r, hi_r) of the (H, W) pixel grid on its own
    device (no scatter), renders them with its persistent scene state and the frame is assembled with
    one all-gather of (rays / G, 4) tiles.  pose (4, 4) as util.gen_rays takes it (util.gen_rays_yolo with yolo=True).
    Returns (rgb (H, W, 3), depth (H, W)) on every rank."""
    from .util import gen_rays_range
    world, rank = _world(group)
    n = int(width) * int(height)
    lo, hi, per = shard_bounds(n, world, rank)
    pose = torch.as_tensor(pose, dtype=torch.float32).reshape(1, 4, 4)
    if hi > lo:
        rays = gen_rays_range(pose, width, height, focal, z_near, z_far, lo, hi - lo, c=c, yolo=yolo, device=device)
        rgb, depth = render_fn(rays)
        tile = torch.cat([rgb, depth[:, None]], dim=1)
    else:
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        tile = torch.zeros(0, 4, device=dev, dtype=torch.float32)
    full = gather_tiles(tile, n, per, group)
    return full[:, :3].reshape(height, width, 3), full[:, 3].reshape(height, width)


def _common_flat_span(params):
    """One fp32 tensor over the storage range that holds every ``p.grad`` of ``params`` when all of them are contiguous fp32
    views of ONE storage (and every parameter has a gradient); else None."""
    grads = [p.grad for p in params]
    if any(g is None or g.dtype != torch.float32 or not g.is_contiguous() for g in grads):
        return None
    st = grads[0].untyped_storage()
    if any(g.untyped_storage().data_ptr() != st.data_ptr() for g in grads[1:]):
        return None
    ivs = sorted((g.storage_offset(), g.storage_offset() + g.numel()) for g in grads)
    if any(a[1] > b[0] for a, b in zip(ivs, ivs[1:])):   # overlapping views: not a partition of the range
        return None
    lo, hi = ivs[0][0], ivs[-1][1]
    # The range is reduced (and divided) in place, so it must hold nothing but these gradients: neighbouring views may be
    # separated only by the allocation's alignment padding (bind_mlp_grads rounds every view up to 64 floats; those gaps
    # are zeros on every rank).  A wider gap is memory of somebody else -- e.g. the gradient of a parameter that was not
    # passed in -- and would be summed over the ranks as a side effect: take the bucket path then.
    if any(b[0] - a[1] >= 64 for a, b in zip(ivs, ivs[1:])):
        return None
    return torch.empty(0, dtype=torch.float32, device=grads[0].device).set_(st, lo, (hi - lo,))


def allreduce_gradients(params, group=None, bucket_bytes=64 << 20, average=True):
    """Data-parallel training over ranks that each ran ``loss.backward()`` on their own super-batch (the reference trains on
    one GPU; its DataParallel splits rays, not objects -- SURVEY.md 2.3): sum the ``.grad`` of ``params`` over the ranks and
    divide by the world size, in flat buckets of at most ``bucket_bytes`` (the two MLPs hold 27 MB of fp32 gradients: ONE
    collective, ring all-reduce on xGMI is per-link bound, so fewer and larger is better).  Parameters without a gradient on
    this rank contribute zeros, so every rank issues the same collectives.  No-op for a single process.  Returns the number of
    collectives issued."""
    world, _ = _world(group)
    params = [p for p in params if p.requires_grad]
    if world == 1 or not params:
        return 0
    # The renderer's backward hands out the MLP gradients as views of ONE flat fp32 allocation (PixelNeRFNet.bind_mlp_grads): when
    # every gradient of `params` lives in one storage, reduce that storage's covering range in place -- one collective, no
    # flatten / scatter copies (the alignment gaps between the views are zeros on every rank).
    span = _common_flat_span(params)
    if span is not None and span.numel() * 4 > bucket_bytes:
        span = None
    # every rank must take the same path: agree on "all of us have a flat span of the same length" (one 2-word collective)
    n_span = span.numel() if span is not None else 0
    on_host = dist.get_backend(group) == "gloo"
    ref_dev = "cpu" if on_host else next(p.device for p in params)
    vote = torch.tensor([n_span, -n_span], dtype=torch.int64, device=ref_dev)
    dist.all_reduce(vote, op=dist.ReduceOp.MIN, group=group)
    if not (n_span > 0 and int(vote[0]) == n_span and int(vote[1]) == -n_span):
        span = None
    if span is not None:
        if span.is_cuda and dist.get_backend(group) == "gloo":   # rehearsals on one GPU: gloo moves host memory
            host = span.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            span.copy_(host)
        else:
            dist.all_reduce(span, op=dist.ReduceOp.SUM, group=group)
        if average:
            span /= world
        return 1
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        nbytes = p.numel() * 4
        if cur and cur_bytes + nbytes > bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(p)
        cur_bytes += nbytes
    if cur:
        buckets.append(cur)
    for bucket in buckets:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32) for p in bucket])
        if flat.is_cuda and dist.get_backend(group) == "gloo":   # rehearsals on one GPU: gloo moves host memory
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= world
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off:off + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
    return len(buckets)
