"""Worker for the world_size-2 gloo test of the ray-sharding path (spawned processes do not run
conftest.py, so this module sets up its own imports)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(rank, world, port, n, q):
    import torch
    import torch.distributed as dist
    import pnyolo_pkg
    pnyolo_pkg.load()
    from pixel_nerf_yolo_amd import dist as pdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rays = torch.arange(n * 8, dtype=torch.float32).reshape(n, 8)

    def fake_render(r):  # deterministic per-ray function: what a rank's HIP render would return
        return torch.stack([r[:, 0] * 2, r[:, 1] + 1, r[:, 2] - 3], 1), r[:, 7] * 0.5

    rgb, depth = pdist.render_sharded(fake_render, rays)
    ok = torch.equal(rgb, torch.stack([rays[:, 0] * 2, rays[:, 1] + 1, rays[:, 2] - 3], 1)) and \
        torch.equal(depth, rays[:, 7] * 0.5)
    q.put((rank, bool(ok), tuple(rgb.shape)))
    dist.destroy_process_group()


def run_grads(rank, world, port, q):
    """allreduce_gradients over gloo: every rank holds different gradients (one parameter has none on rank 1, one is frozen);
    afterwards every rank must hold the mean, in one collective for the small bucket limit's worth of tensors."""
    import torch
    import torch.distributed as dist
    import pnyolo_pkg
    pnyolo_pkg.load()
    from pixel_nerf_yolo_amd import dist as pdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = [(512, 42), (512,), (512, 512), (4, 512), (4,)]
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    frozen = torch.nn.Parameter(torch.zeros(3), requires_grad=False)
    for i, p in enumerate(params):
        g = torch.Generator().manual_seed(100 * rank + i)
        p.grad = torch.randn(p.shape, generator=g)
    if rank == 1:
        params[1].grad = None                       # e.g. a parameter the rank's batch did not touch
    n_one = pdist.allreduce_gradients(params + [frozen])                      # everything fits one bucket
    expect = []
    for i, s in enumerate(shapes):
        gs = [torch.randn(s, generator=torch.Generator().manual_seed(100 * r + i)) for r in range(world)]
        if i == 1:
            gs[1] = torch.zeros(s)
        expect.append(sum(gs) / world)
    ok = all(torch.allclose(p.grad, e, rtol=0, atol=1e-7) for p, e in zip(params, expect)) and frozen.grad is None
    n_many = pdist.allreduce_gradients(params, bucket_bytes=512 * 512 * 4)   # second pass: several buckets, values = mean of means
    ok = ok and all(torch.allclose(p.grad, e, rtol=0, atol=1e-7) for p, e in zip(params, expect))
    # gradients as views of one flat buffer (what PixelNeRFNet.bind_mlp_grads hands out): reduced in place, one collective
    flat = torch.zeros(sum((torch.Size(s).numel() + 63) // 64 * 64 for s in shapes))
    off = 0
    for i, (p, s_) in enumerate(zip(params, shapes)):
        n = torch.Size(s_).numel()
        v = flat[off:off + n].view(s_)
        v.copy_(torch.randn(s_, generator=torch.Generator().manual_seed(100 * rank + i)))
        p.grad = v
        off += (n + 63) // 64 * 64
    ptr0 = flat.data_ptr()
    n_flat = pdist.allreduce_gradients(params)
    expect2 = [sum(torch.randn(s_, generator=torch.Generator().manual_seed(100 * r + i)) for r in range(world)) / world
               for i, s_ in enumerate(shapes)]
    ok = ok and n_flat == 1 and all(p.grad.data_ptr() >= ptr0 for p in params)
    ok = ok and all(torch.allclose(p.grad, e, rtol=0, atol=1e-7) for p, e in zip(params, expect2))
    # a SUBSET of the flat allocation's views (params[2] is left out): its storage lies between the views passed in, so the
    # in-place span must not be taken -- the un-passed gradient keeps its rank-local value
    before = params[2].grad.clone()
    pdist.allreduce_gradients([params[0], params[1], params[3], params[4]])
    ok = ok and torch.equal(params[2].grad, before)
    # ... and when one rank lacks a gradient the ranks agree on the bucket path (same result, no mismatch of collectives)
    if rank == 1:
        params[3].grad = None
    pdist.allreduce_gradients(params)
    ok = ok and params[3].grad is not None and all(torch.isfinite(p.grad).all() for p in params)
    q.put((rank, bool(ok), n_one, n_many))
    dist.destroy_process_group()


def run_render(rank, world, port, q):
    """world_size ranks sharing cuda:0 (gloo: RCCL refuses duplicate devices) render ONE frame with
    dist.render_frame_sharded through the HIP path: each rank generates and renders its own ray range with explicit
    draws sliced from a common seeded set; the assembled frame must equal rank 0's single-call render bit for bit."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import pnyolo_pkg
    pnyolo_pkg.load()
    from pixel_nerf_yolo_amd import conf as pconf, dist as pdist, synth
    from pixel_nerf_yolo_amd.model import make_model
    from pixel_nerf_yolo_amd.render import NeRFRenderer
    from pixel_nerf_yolo_amd.util import gen_rays

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    NS, H, W, kc, kf, kfd = 2, 40, 48, 16, 8, 4
    net = make_model(pconf.default_mv()["model"]).eval()
    for mlp, seed in ((net.mlp_coarse, 61), (net.mlp_fine, 62)):
        mlp.load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(seed).items()})
    net = net.to(dev)
    net.set_latent_projection("on")
    src, tgt = synth.scene_cameras(NS)
    focal, c = torch.tensor(50.0), torch.tensor([[W * 0.5, H * 0.5]])
    net.encode(torch.zeros(1, NS, 3, H, W), torch.from_numpy(src)[None], focal, c=c,
               latent=torch.from_numpy(synth.latent(63, NS, 512, H // 2, W // 2)))
    n = H * W
    rs = np.random.RandomState(64)
    draws = dict(u_coarse=rs.rand(n, kc).astype(np.float32), u_fine=rs.rand(n, kf - kfd).astype(np.float32),
                 u_fine2=rs.rand(n, kf - kfd).astype(np.float32), g_depth=rs.randn(n, kfd).astype(np.float32))
    ren = NeRFRenderer(n_coarse=kc, n_fine=kf, n_fine_depth=kfd, white_bkgd=True).eval()
    par = ren.bind_parallel(net, None, simple_output=True).eval()
    lo, hi, per = pdist.shard_bounds(n, world, rank)

    def render(rays):
        ren.draws = {k: v[lo:hi] for k, v in draws.items()}
        with torch.no_grad():
            rgb, depth = par(rays[None])
        return rgb[0], depth[0]

    rgb, depth = pdist.render_frame_sharded(render, tgt, W, H, focal, 0.8, 1.8, c=c[0], device=dev)
    ok, shape = True, tuple(rgb.shape)
    if rank == 0:
        rays = gen_rays(torch.from_numpy(tgt)[None], W, H, focal, 0.8, 1.8, c=c[0], device=dev).reshape(1, -1, 8)
        ren.draws = draws
        with torch.no_grad():
            r1, d1 = par(rays)
        ok = bool(torch.equal(r1[0].reshape(H, W, 3), rgb)) and bool(torch.equal(d1[0].reshape(H, W), depth))
    q.put((rank, ok, shape, (lo, hi)))
    dist.destroy_process_group()
