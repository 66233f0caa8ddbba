"""Diagnostic: which samples of a single tile differ between the f16x2 kernel and the fp32 kernel, over repeated launches."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import conftest  # noqa
from helpers import nerf_net
os.environ["PNYOLO_PROJECTION"] = "on"
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "nerf_c2.npz")))
DEV = "cuda:0"
dt = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32, device=DEV).contiguous()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(0)
idx = rng.integers(0, g["probe_xyz"].shape[0], n)
xyz, vd = dt(g["probe_xyz"][idx])[None], dt(g["probe_viewdirs"][idx])[None]
outs = {}
for prec in ("f32", "f16x2"):
    os.environ["PNYOLO_MLP_PRECISION"] = prec
    net = nerf_net(g, 7)
    with torch.no_grad():
        outs[prec] = [net(xyz, coarse=True, viewdirs=vd)[0].cpu() for _ in range(6)]
ref = outs["f32"][0]
for i, o in enumerate(outs["f16x2"]):
    e = (o - ref).abs().max(-1).values
    bad = torch.nonzero(e > 1e-4).flatten().tolist()
    print("run", i, "max", float(e.max()), "bad samples", bad)
