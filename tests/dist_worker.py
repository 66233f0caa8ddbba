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
