"""Diagnostic: per-column difference of lin_in.weight.grad / lin_z.0.weight.grad between the fp32 and the f16x2 training forward."""
import os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import conftest  # noqa
import pnyolo_oracle as orc
from pixel_nerf_yolo_amd import conf as pconf, synth
from pixel_nerf_yolo_amd.model import make_model
from pixel_nerf_yolo_amd.render import NeRFRenderer
DEV = "cuda:0"
SB, ns, H, W, kc, kf, kfd, n = 1, 3, 64, 64, 32, 16, 8, 256
grads = {}
for prec in ("f32", "f16x2"):
    os.environ["PNYOLO_MLP_PRECISION"] = prec
    net = make_model(pconf.default_mv()["model"], stop_encoder_grad=True)
    net.mlp_coarse.load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(901).items()})
    net.mlp_fine.load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(902).items()})
    net = net.to(DEV).train()
    poses = np.stack([synth.scene_cameras(ns)[0]])
    lat = torch.from_numpy(synth.latent(903, ns, 512, H // 2, W // 2))
    net.encode(torch.zeros(SB, ns, 3, H, W), torch.from_numpy(poses), torch.tensor(0.9 * W), latent=lat)
    _, tgt = synth.scene_cameras(ns)
    rays_all = orc.gen_rays(tgt[None], W, H, 0.9 * W, 0.8, 1.8)[0].reshape(-1, 8)
    r0 = np.random.RandomState(32)
    rays = rays_all[torch.from_numpy(r0.choice(H * W, n, replace=False))][None].to(DEV)
    gt = torch.from_numpy(r0.uniform(0, 1, size=(SB, n, 3)).astype(np.float32)).to(DEV)
    ren = NeRFRenderer(n_coarse=kc, n_fine=0, n_fine_depth=0, white_bkgd=True).train()
    ren.draws = dict(u_coarse=r0.rand(n, kc).astype(np.float32))
    out = ren(net, rays, want_weights=True)
    torch.nn.functional.mse_loss(out["coarse"]["rgb"], gt).backward()
    grads[prec] = {k: p.grad.detach().cpu().clone() for k, p in net.mlp_coarse.named_parameters()}
for name in ("lin_in.weight", "lin_z.0.weight"):
    a, b = grads["f32"][name], grads["f16x2"][name]
    col = (a - b).norm(dim=0) / a.norm(dim=0).clamp_min(1e-30)
    print(name, "overall", float((a - b).norm() / a.norm()), "per input column (first 64):", [round(float(x), 4) for x in col[:64]])
    row = (a - b).norm(dim=1) / a.norm(dim=1).clamp_min(1e-30)
    print("   rows: max %.3e median %.3e" % (float(row.max()), float(row.median())))
