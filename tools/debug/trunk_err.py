# GPU box: per-tensor gradient error of the native training trunk against autograd through the oracle (batch statistics)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pnyolo_pkg; pnyolo_pkg.load()
import numpy as np, torch
import pnyolo_oracle as orc
from pixel_nerf_yolo_amd import conf as pconf, synth
from pixel_nerf_yolo_amd.model import make_model
DEV = "cuda:0"
pool = sys.argv[1] != "0" if len(sys.argv) > 1 else True
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1803
SB, ns, H, W = 2, 2, 64, 64
c = pconf.default_mv(); c.d["model"]["encoder"]["use_first_pool"] = pool
net = make_model(c["model"], stop_encoder_grad=False)
enc = synth.resnet34_state(SEED, residual_gain=0.25)
net.load_state_dict({k: torch.from_numpy(v) for k, v in enc.items()}, strict=False)
net = net.to(DEV).train()
images = torch.from_numpy(np.stack([synth.images(SEED + 1 + i, ns, H, W) for i in range(SB)]))
poses = np.stack([synth.scene_cameras(ns, radius=1.3 + 0.1 * i)[0] for i in range(SB)])
G = torch.from_numpy(np.random.RandomState(5).standard_normal((SB * ns, 512, H // 2, W // 2)).astype(np.float32))
net.encode(images, torch.from_numpy(poses), torch.tensor(0.9 * W))
lat = net.differentiable_latent()
(lat * G.to(DEV)).sum().backward()
enc_t = {k: torch.from_numpy(v.copy()) for k, v in enc.items() if "num_batches" not in k}
for k, t in enc_t.items():
    if t.is_floating_point() and "running" not in k: t.requires_grad_()
# fp64 oracle as the arbiter
enc64 = {k: t.detach().double().requires_grad_(t.requires_grad) for k, t in enc_t.items()}
lat_ref = orc.spatial_encoder(enc_t, images.reshape(-1, 3, H, W), use_first_pool=pool, training=True)[0]
(lat_ref * G).sum().backward()
import torch.nn.functional as F
def enc64_run():
    T0 = orc.T
    orc.T = lambda v: v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v))
    try:
        out = orc.spatial_encoder(enc64, images.reshape(-1, 3, H, W).double(), use_first_pool=pool, training=True)[0]
    finally:
        orc.T = T0
    return out
l64 = enc64_run(); (l64 * G.double()).sum().backward()
print("latent: hip vs fp32 oracle %.2e, fp32 oracle vs fp64 %.2e, hip vs fp64 %.2e (max |lat| %.2f)" % (
    float((lat.detach().cpu() - lat_ref.detach()).abs().max()), float((lat_ref.detach().double() - l64.detach()).abs().max()),
    float((lat.detach().cpu().double() - l64.detach()).abs().max()), float(l64.detach().abs().max())))
rows = []
for k, p in net.encoder.model.named_parameters():
    if k.startswith(("layer4", "fc")): continue
    r32 = enc_t["encoder.model." + k].grad; r64 = enc64["encoder.model." + k].grad
    sc = float(r64.abs().max())
    rows.append((float((p.grad.cpu().double() - r64).abs().max()) / sc, float((r32.double() - r64).abs().max()) / sc, k))
w32 = max(float((p.grad.cpu() - enc_t["encoder.model." + k].grad).abs().max()) / float(enc_t["encoder.model." + k].grad.abs().max()) for k, p in net.encoder.model.named_parameters() if not k.startswith(("layer4", "fc")))
print("seed %d pool %s: worst hip vs torch-fp32 %.2e" % (SEED, pool, w32))
rows.sort(reverse=True)
print("worst tensors: (hip vs fp64) (torch fp32 vs fp64) name")
for r in rows[:12]: print("  %.2e  %.2e  %s" % r)
