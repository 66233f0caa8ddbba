"""Shared builders for the parity tests: the same seeded scenes tools/make_golden.py used."""
import numpy as np
import torch

from pixel_nerf_yolo_amd import conf as pconf
from pixel_nerf_yolo_amd import synth
from pixel_nerf_yolo_amd.model import make_model


def load_mlp(mlp, seed, d_latent, d_out):
    sd = synth.mlp_state(seed, d_latent=d_latent, d_out=d_out)
    mlp.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)


def nerf_net(g, seed, device="cuda:0"):
    """PixelNeRFNet (HIP path) configured and seeded exactly like tools/make_golden.fixture_nerf."""
    c = pconf.default_mv()
    has_fine = int(g["Kf"]) > 0
    if not has_fine:
        c.d["model"]["mlp_fine"] = {"type": "empty"}
    net = make_model(c["model"]).eval()
    load_mlp(net.mlp_coarse, seed * 10 + 1, 512, 4)
    if has_fine:
        load_mlp(net.mlp_fine, seed * 10 + 2, 512, 4)
    net = net.to(device)
    ns, H, W = int(g["NS"]), int(g["H"]), int(g["W"])
    lat = torch.from_numpy(synth.latent(seed * 10 + 3, ns, 512, H // 2, W // 2))
    images = torch.zeros(1, ns, 3, H, W)
    net.encode(images, torch.from_numpy(g["src_poses"])[None], torch.tensor(float(g["focal"])),
               c=torch.from_numpy(g["c"])[None], latent=lat)
    return net


def oracle_scene(g, seed):
    import pnyolo_oracle as orc
    ns, H, W = int(g["NS"]), int(g["H"]), int(g["W"])
    mc = synth.mlp_state(seed * 10 + 1)
    mf = synth.mlp_state(seed * 10 + 2) if int(g["Kf"]) > 0 else None
    lat = synth.latent(seed * 10 + 3, ns, 512, H // 2, W // 2)
    return orc.Scene(mc, mf, lat, g["src_poses"], g["focal"], g["c"][None], W, H)


def maxabs(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a), dtype=torch.float32)
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b), dtype=torch.float32)
    return float((a - b).abs().max())
