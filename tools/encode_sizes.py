"""Scene-encode time of the HIP ResNet-34 trunk at the input sizes of BASELINE configs and of the reference's datasets
(VERDICT r1 item 8): 3x128x128 (C2), 3x300x400 (DTU), 3x400x400 (C4 geometry).  Prints ms per encode, TFLOP/s and the share
of a frame's render time.  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pnyolo_pkg; pnyolo_pkg.load()
from pixel_nerf_yolo_amd import conf as pconf, synth
from pixel_nerf_yolo_amd.model import make_model

dev = torch.device("cuda:0")
net = make_model(pconf.default_mv()["model"]).eval()
sd = {}
sd.update({"mlp_coarse." + k: v for k, v in synth.mlp_state(71).items()})
sd.update({"mlp_fine." + k: v for k, v in synth.mlp_state(72).items()})
sd.update(synth.resnet34_state(74, residual_gain=0.25))
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
net = net.to(dev)
src, _ = synth.scene_cameras(3)
poses = torch.from_numpy(src)[None]
for (H, W, frame_ms, what) in ((128, 128, 220.5, "C2 frame 220.5 ms"), (300, 400, None, "DTU input"), (400, 400, 4228.0, "C4 frame 4228 ms")):
    img = torch.from_numpy(synth.images(75, 3, H, W)).to(dev)
    for _ in range(3):
        net.encode(img[None], poses, torch.tensor(131.25 * W / 128))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        net.encode(img[None], poses, torch.tensor(131.25 * W / 128))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    # ResNet-34 trunk to layer3: 0.99 GMAC per 128x128 view, scales with the pixel count
    gflop = 2 * 0.99 * 3 * (H * W) / (128 * 128)
    share = (" = %.2f %% of the %s" % (100 * ms / frame_ms, what)) if frame_ms else " (%s)" % what
    print("encode 3x%dx%d: %.3f ms, %.1f TFLOP/s%s" % (H, W, ms, gflop / ms, share), flush=True)
