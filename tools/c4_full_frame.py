#!/usr/bin/env python3
"""GPU box: one full BASELINE config-4 frame (400x400, 3 views, L=1792, 128 coarse + 64 fine (32 depth) samples)
in a single render call on one GPU: size check of workspaces / tile counts, rays/s, output sanity."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import pnyolo_pkg

pnyolo_pkg.load()
from pixel_nerf_yolo_amd import conf as pconf, synth
from pixel_nerf_yolo_amd.model import make_model
from pixel_nerf_yolo_amd.render import NeRFRenderer
from pixel_nerf_yolo_amd.util import gen_rays

dev = torch.device("cuda:0")
c = pconf.default_mv()
c.d["model"]["encoder"]["backbone"] = "custom"
net = make_model(c["model"]).eval()
for mlp, seed in ((net.mlp_coarse, 1), (net.mlp_fine, 2)):
    mlp.load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(seed, d_latent=1792).items()})
net = net.to(dev)
src, tgt = synth.scene_cameras(3)
focal, cc = torch.tensor(410.16), torch.tensor([[200.0, 200.0]])
net.encode(torch.zeros(1, 3, 3, 400, 400), torch.from_numpy(src)[None], focal, c=cc,
           latent=torch.from_numpy(synth.latent(3, 3, 1792, 50, 50)))
rays = gen_rays(torch.from_numpy(tgt)[None], 400, 400, focal, 0.8, 1.8, c=cc[0]).reshape(1, -1, 8)
ren = NeRFRenderer(n_coarse=128, n_fine=64, n_fine_depth=32, white_bkgd=True).eval()
par = ren.bind_parallel(net, None, simple_output=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.no_grad():
    rgb, depth = par(rays)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert rgb.shape == (1, 160000, 3) and bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(depth).all())
assert float(rgb.min()) >= -1e-5 and float(rgb.max()) <= 1 + 1e-4
print("C4 full frame, 1 GPU, one call: %.2f s, %.0f rays/s, peak torch memory %.2f GB" % (
    dt, 160000 / dt, torch.cuda.max_memory_allocated() / 1e9))
