import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import pnyolo_pkg; pnyolo_pkg.load()
import torch
from test_gpu_range import make_net, points
net = make_net(500); net.f16_range_policy = "lazy"
xyz, vd = points(150, 4)
with torch.no_grad():
    base = net(xyz, coarse=True, viewdirs=vd)
    print("f16x2", net.last_launch_f16x2(), "status", net.range_status())
    k0 = net._synced_key
    net.mlp_coarse.blocks[1].fc_0.weight[5, 9] = 1.0e5
    k1 = net._weights_key()
    print("changed", [a[0] for a, b in zip(k0, k1) if a != b], "dev_bound", net._dev_bound)
    try:
        out = net(xyz, coarse=True, viewdirs=vd)
        torch.cuda.synchronize()
        print("f16x2", net.last_launch_f16x2(), "status", net.range_status(), "finite", bool(torch.isfinite(out).all()))
    except Exception as e:
        print("EXC", type(e), e)
