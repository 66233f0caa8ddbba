"""Diagnostic: the test_query_golden sequence under both matrix precisions, with per-element error locations."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import conftest  # noqa
from helpers import nerf_net, maxabs
os.environ["PNYOLO_PROJECTION"] = "on"
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "nerf_c2.npz")))
DEV = "cuda:0"
dt = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32, device=DEV).contiguous()
for prec in ("f32", "f16x2"):
    os.environ["PNYOLO_MLP_PRECISION"] = prec
    net = nerf_net(g, 7)
    xyz, vd = dt(g["probe_xyz"])[None], dt(g["probe_viewdirs"])[None]
    with torch.no_grad():
        oc = net(xyz, coarse=True, viewdirs=vd)[0]
        of = net(xyz, coarse=False, viewdirs=vd)[0]
        of_again = net(xyz, coarse=False, viewdirs=vd)[0]
        oc_again = net(xyz, coarse=True, viewdirs=vd)[0]
    print(prec, "coarse", maxabs(oc, g["probe_out_coarse"]), "fine", maxabs(of, g["probe_out_fine"]),
          "repeat", float((of - of_again).abs().max()), float((oc - oc_again).abs().max()), "f16x2:", net.last_launch_f16x2())
    net.mlp_fine = None
    with torch.no_grad():
        o2 = net(xyz, coarse=False, viewdirs=vd)[0]
        o3 = net(xyz, coarse=True, viewdirs=vd)[0]
    e = (o2.cpu() - torch.from_numpy(g["probe_out_coarse"])).abs()
    print(prec, "fine=None:", float(e.max()), "coarse flag:", maxabs(o3, g["probe_out_coarse"]), "n bad", int((e > 1e-4).sum()), "of", e.numel(),
          "bad rows", torch.nonzero((e > 1e-4).any(-1)).flatten().tolist()[:20])
