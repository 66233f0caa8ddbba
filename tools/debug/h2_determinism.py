"""Diagnostic: run-to-run determinism of the f16x2 kernel on identical inputs (no reference needed)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import conftest  # noqa
from helpers import nerf_net
os.environ["PNYOLO_PROJECTION"] = "on"
os.environ["PNYOLO_MLP_PRECISION"] = "f16x2"
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "nerf_c2.npz")))
DEV = "cuda:0"
dt = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32, device=DEV).contiguous()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6400
rng = np.random.default_rng(0)
idx = rng.integers(0, g["probe_xyz"].shape[0], n)
xyz, vd = dt(g["probe_xyz"][idx])[None], dt(g["probe_viewdirs"][idx])[None]
net = nerf_net(g, 7)
with torch.no_grad():
    outs = [net(xyz, coarse=True, viewdirs=vd)[0].cpu() for _ in range(8)]
# majority value per element = median over runs
med = torch.stack(outs).median(0).values
for i, o in enumerate(outs):
    e = (o - med).abs().max(-1).values
    bad = torch.nonzero(e > 1e-5).flatten()
    print("run", i, "max dev", float(e.max()), "n bad", int(bad.numel()), "local idx hist (m//8):", np.bincount((bad.numpy() % 64) // 8, minlength=8).tolist())
