import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import pnyolo_pkg; pnyolo_pkg.load()
from pixel_nerf_yolo_amd import conf as pconf, synth
from pixel_nerf_yolo_amd.model import make_model
dev = torch.device("cuda:0")
net = make_model(pconf.default_mv()["model"]).eval()
for mlp, seed in ((net.mlp_coarse, 1), (net.mlp_fine, 2)):
    mlp.load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(seed).items()})
net = net.to(dev)
src, tgt = synth.scene_cameras(3)
net.encode(torch.zeros(1, 3, 3, 128, 128), torch.from_numpy(src)[None], torch.tensor(131.25), c=torch.tensor([[64.0, 64.0]]), latent=torch.from_numpy(synth.latent(3, 3, 512, 64, 64)))
net.set_latent_projection(os.environ.get("SWEEP_PROJECTION", "on"))
net.enable_kernel_timing(True)
rs = np.random.RandomState(0)
out = []
for tiles in (16, 64, 128, 192, 256, 300, 384, 520, 700, 1024):
    n = tiles * 64
    xyz = torch.from_numpy(rs.uniform(-0.5, 0.5, size=(1, n, 3)).astype(np.float32)).to(dev)
    vd = torch.from_numpy(rs.standard_normal((1, n, 3)).astype(np.float32)).to(dev)
    with torch.no_grad():
        net(xyz, coarse=True, viewdirs=vd); ms = 0.0
        for _ in range(5):
            net(xyz, coarse=True, viewdirs=vd); ms += net.last_mlp_stats(full=True)["kernel_ms"]
    out.append("%d:%.3f" % (tiles, ms / 5))
print("variant", os.environ.get("PNYOLO_MLP_VARIANT", "auto"), "precision", os.environ.get("PNYOLO_MLP_PRECISION", "auto"), "projection",
      os.environ.get("SWEEP_PROJECTION", "on"), "| 64-sample tiles:ms", " ".join(out))
