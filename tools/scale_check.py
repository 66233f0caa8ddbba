#!/usr/bin/env python3
"""Runs the BASELINE configs that are not the bench line at full size on one GPU (C3: YOLO
encoder-sized latent L=1792; C4: 400x400, 128+64 samples) and reports rays/s plus sanity
properties.  Usage (GPU box): python tools/scale_check.py [c3] [c4]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import pnyolo_pkg

pnyolo_pkg.load()
from pixel_nerf_yolo_amd import conf as pconf, synth
from pixel_nerf_yolo_amd.model import make_model
from pixel_nerf_yolo_amd.render import NeRFRenderer, make_renderer
from pixel_nerf_yolo_amd.util import gen_rays, gen_rays_yolo

dev = torch.device("cuda:0")


def load(mlp, seed, L, d_out):
    mlp.load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(seed, d_latent=L, d_out=d_out).items()})


def timed(fn, n=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


def c3():
    """128x128 render, 3 source views, L=1792 latent (YOLO-sized), 64+32 samples, NeRF renderer."""
    c = pconf.default_mv()
    c.d["model"]["encoder"]["backbone"] = "custom"
    net = make_model(c["model"]).eval()
    load(net.mlp_coarse, 1, 1792, 4)
    load(net.mlp_fine, 2, 1792, 4)
    net = net.to(dev)
    src, tgt = synth.scene_cameras(3)
    lat = torch.from_numpy(synth.latent(3, 3, 1792, 16, 16))
    focal, cc = torch.tensor(131.25), torch.tensor([[64.0, 64.0]])
    net.encode(torch.zeros(1, 3, 3, 128, 128), torch.from_numpy(src)[None], focal, c=cc, latent=lat)
    rays = gen_rays(torch.from_numpy(tgt)[None].to(dev), 128, 128, focal, 0.8, 1.8, c=cc[0]).reshape(1, -1, 8)
    ren = NeRFRenderer(n_coarse=64, n_fine=32, n_fine_depth=16, white_bkgd=True).eval()
    par = ren.bind_parallel(net, None, simple_output=True)
    with torch.no_grad():
        dt, (rgb, depth) = timed(lambda: par(rays))
    assert bool(torch.isfinite(rgb).all())
    fl = 2 * (3 * (42 * 512 + 3 * 1792 * 512 + 6 * 512 * 512) + 4 * 512 * 512 + 2048) * 160
    print("C3 (L=1792): %.0f rays/s, %.1f ms/frame, %.1f TFLOP/s algorithmic" % (16384 / dt, dt * 1e3, 16384 * fl / dt / 1e12))
    # YOLO renderer at the real data geometry (30x16 cells, 128 samples) -- reference yolo.conf
    cy = pconf.yolo()
    ny = make_model(cy["model"]).eval()
    load(ny.mlp_coarse, 4, 1792, 21)
    ny = ny.to(dev)
    flip = np.diag([1.0, -1.0, -1.0, 1.0]).astype(np.float32)
    s2, t2 = synth.scene_cameras(3, radius=6.0, phi=-25.0)
    w2c = np.stack([np.linalg.inv(p @ flip) for p in s2]).astype(np.float32)
    ny.encode(torch.zeros(1, 3, 3, 512, 960), torch.from_numpy(w2c)[None], torch.tensor([[700.0, 700.0]]),
              c=torch.tensor([[480.0, 256.0]]), latent=torch.from_numpy(synth.latent(5, 3, 1792, 16, 30)))
    ry = gen_rays_yolo(torch.from_numpy(np.linalg.inv(t2 @ flip).astype(np.float32))[None].to(dev), 30, 16,
                       [700.0 / 32, 700.0 / 32], [480.0 / 32, 256.0 / 32], 1.0, 13.0)
    yr = make_renderer(cy).bind_parallel(ny)
    with torch.no_grad():
        dt, out = timed(lambda: yr(ry), n=5)
    assert out.shape == (480, 3, 7) and bool(torch.isfinite(out).all())
    print("YOLO renderer (480 rays x 128 samples, one call): %.2f ms/view" % (dt * 1e3))


def small():
    """Training / visualisation-size launches (C2 scene): 128 rays x 64 coarse + 96 fine samples."""
    c = pconf.default_mv()
    net = make_model(c["model"]).eval()
    load(net.mlp_coarse, 1, 512, 4)
    load(net.mlp_fine, 2, 512, 4)
    net = net.to(dev)
    src, tgt = synth.scene_cameras(3)
    focal, cc = torch.tensor(131.25), torch.tensor([[64.0, 64.0]])
    net.encode(torch.zeros(1, 3, 3, 128, 128), torch.from_numpy(src)[None], focal, c=cc,
               latent=torch.from_numpy(synth.latent(3, 3, 512, 64, 64)))
    rays = gen_rays(torch.from_numpy(tgt)[None].to(dev), 128, 128, focal, 0.8, 1.8, c=cc[0]).reshape(1, -1, 8)
    ren = NeRFRenderer(n_coarse=64, n_fine=32, n_fine_depth=16, white_bkgd=True).eval()
    par = ren.bind_parallel(net, None, simple_output=True)
    for n in (128, 256, 1024):
        sub = rays[:, torch.arange(0, 16384, 16384 // n)[:n]].contiguous()
        with torch.no_grad():
            dt, _ = timed(lambda: par(sub), n=20)
        print("render of %5d rays (64 + 96 samples): %.3f ms  (%.0f rays/s), MLP shape %s" % (
            n, dt * 1e3, n / dt, os.environ.get("PNYOLO_MLP_VARIANT", "auto")))


def c4():
    """400x400 render, 3 views, 128 coarse + 64 fine (32 depth), L=1792: one GPU's share and the full frame."""
    c = pconf.default_mv()
    c.d["model"]["encoder"]["backbone"] = "custom"
    net = make_model(c["model"]).eval()
    load(net.mlp_coarse, 1, 1792, 4)
    load(net.mlp_fine, 2, 1792, 4)
    net = net.to(dev)
    src, tgt = synth.scene_cameras(3)
    focal, cc = torch.tensor(410.16), torch.tensor([[200.0, 200.0]])
    net.encode(torch.zeros(1, 3, 3, 400, 400), torch.from_numpy(src)[None], focal, c=cc,
               latent=torch.from_numpy(synth.latent(3, 3, 1792, 50, 50)))
    rays = gen_rays(torch.from_numpy(tgt)[None].to(dev), 400, 400, focal, 0.8, 1.8, c=cc[0]).reshape(1, -1, 8)
    ren = NeRFRenderer(n_coarse=128, n_fine=64, n_fine_depth=32, white_bkgd=True).eval()
    par = ren.bind_parallel(net, None, simple_output=True)
    share = rays[:, :20000].contiguous()   # 1/8 of the frame = what one of 8 GPUs renders
    with torch.no_grad():
        dt, (rgb, depth) = timed(lambda: par(share), n=1)
    assert bool(torch.isfinite(rgb).all()) and float(rgb.min()) >= -1e-4 and float(rgb.max()) <= 1 + 1e-4
    fl = 2 * (3 * (42 * 512 + 3 * 1792 * 512 + 6 * 512 * 512) + 4 * 512 * 512 + 2048) * 320
    print("C4 share (20000 of 160000 rays, 128+64): %.0f rays/s, %.2f s, %.1f TFLOP/s algorithmic"
          % (20000 / dt, dt, 20000 * fl / dt / 1e12))
    print("peak GPU memory: %.2f GB (torch) ; full 160000-ray frame on one GPU would take ~%.0f s"
          % (torch.cuda.max_memory_allocated() / 1e9, 8 * dt))


if __name__ == "__main__":
    which = sys.argv[1:] or ["c3", "c4"]
    if "c3" in which:
        c3()
    if "c4" in which:
        c4()
    if "small" in which:
        small()
