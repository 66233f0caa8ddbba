import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import pnyolo_pkg; pnyolo_pkg.load()
from pixel_nerf_yolo_amd import conf as pconf, synth
from pixel_nerf_yolo_amd.model import make_model
from pixel_nerf_yolo_amd.render import NeRFRenderer
from pixel_nerf_yolo_amd.util import gen_rays
dev = torch.device("cuda:0")
net = make_model(pconf.default_mv()["model"]).eval()
for mlp, seed in ((net.mlp_coarse, 1), (net.mlp_fine, 2)):
    mlp.load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(seed).items()})
net = net.to(dev)
src, tgt = synth.scene_cameras(3)
focal, cc = torch.tensor(131.25), torch.tensor([[64.0, 64.0]])
net.encode(torch.zeros(1, 3, 3, 128, 128), torch.from_numpy(src)[None], focal, c=cc, latent=torch.from_numpy(synth.latent(3, 3, 512, 64, 64)))
rays = gen_rays(torch.from_numpy(tgt)[None], 128, 128, focal, 0.8, 1.8, c=cc[0]).reshape(1, -1, 8)
ren = NeRFRenderer(n_coarse=64, n_fine=32, n_fine_depth=16, white_bkgd=True).eval()
par = ren.bind_parallel(net, None, simple_output=True)
sub = rays[:, :128].contiguous()
with torch.no_grad():
    par(sub); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): net._sync()
    t_sync = (time.perf_counter() - t0) / 200
    net.enable_kernel_timing(True)
    tot = kern = 0.0
    for _ in range(50):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        par(sub); torch.cuda.synchronize()
        tot += time.perf_counter() - t0
        kern += net.last_mlp_stats(full=True)["kernel_ms"] * 1e-3
    # host-only cost: launch without waiting
    t0 = time.perf_counter()
    for _ in range(50): par(sub)
    t_issue = (time.perf_counter() - t0) / 50
    torch.cuda.synchronize()
print("_sync: %.1f us; 128-ray render wall %.3f ms, MLP kernels %.3f ms; host issue time per call (async) %.3f ms" % (
    t_sync * 1e6, tot / 50 * 1e3, kern / 50 * 1e3, t_issue * 1e3))
