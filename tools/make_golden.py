#!/usr/bin/env python3
"""
Golden-vector generator: runs the REFERENCE (kofinandi/pixel-nerf-yolo, read-only at
/root/reference) on CPU on seeded inputs and writes small input/output fixtures to
tests/golden/*.npz.  It runs ONLY in the build container; nothing under tests/, bench.py or
smoke() imports the reference (it does not exist on the GPU box).

The reference has no golden vectors or known-answer tests of its own, so the
oracle (oracle/pnyolo_oracle.py) is pinned by what this script captures.

Import shims: the reference imports five absent third-party modules at module
top level -- cv2, torchvision(+.transforms, .models), dotmap, pyhocon, models.yolo (external
NeRF-YOLO checkout).  They are replaced in sys.modules by inert stubs; none of their
arithmetic is emulated:
  * dotmap.DotMap     -> dict with attribute access + toDict() (container only)
  * pyhocon           -> only needed for `import`; configs are the Conf class below, which
                         implements the ConfigTree accessors the reference calls
  * torchvision.models.resnet34 -> a local nn.Module with the public ResNet-34 layout
                         (BasicBlock [3,4,6,3]) built from torch.nn conv/bn; torchvision itself
                         stays absent, so results at that boundary are PARITY UNPINNED vs
                         torchvision and pinned only vs torch's own conv2d/batch_norm.
  * YOLOEncoder       -> not constructible here (needs ../NeRF-YOLO + yolov7.pt); the YOLO-mode
                         fixture patches a dummy module with dims=[1792] and feeds the latent
                         directly (encoder output is an input of the path; PARITY UNPINNED).
Usage:  python tools/make_golden.py            (writes tests/golden/)
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
import pnyolo_pkg  # noqa: E402

pnyolo_pkg.load()
from pixel_nerf_yolo_amd import synth  # noqa: E402


# --------------------------------------------------------------------------- shims
class DotMap(dict):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def toDict(self):
        return {k: (v.toDict() if isinstance(v, DotMap) else v) for k, v in self.items()}


class Conf:
    """Minimal stand-in for pyhocon.ConfigTree (accessors used at reference models.py:21-83,
    resnetfc.py:189-205, encoder.py:176-186, nerf.py:347-358, code.py:45-52)."""
    _MISSING = object()

    def __init__(self, d):
        self.d = d

    def _get(self, key, default=_MISSING):
        cur = self.d
        for part in key.split("."):
            if isinstance(cur, dict) and part in cur:
                cur = cur[part]
            else:
                if default is Conf._MISSING:
                    raise KeyError(key)
                return default
        return cur

    def __getitem__(self, key):
        v = self._get(key)
        return Conf(v) if isinstance(v, dict) else v

    def get_bool(self, k, default=_MISSING):
        return bool(self._get(k, default))

    def get_int(self, k, default=_MISSING):
        v = self._get(k, default)
        return v if v is None else int(v)

    def get_float(self, k, default=_MISSING):
        v = self._get(k, default)
        return v if v is None else float(v)

    def get_string(self, k, default=_MISSING):
        return self._get(k, default)

    def get_list(self, k, default=_MISSING):
        return self._get(k, default)


def _basic_block(cin, cout, stride, norm_layer):
    class BasicBlock(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
            self.bn1 = norm_layer(cout)
            self.relu = nn.ReLU(inplace=True)
            self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
            self.bn2 = norm_layer(cout)
            self.downsample = None
            if stride != 1 or cin != cout:
                self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), norm_layer(cout))

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out))
            return self.relu(out + idt)

    return BasicBlock()


class ResNet34Skeleton(nn.Module):
    """Public ResNet-34 layout (module structure only; see header)."""

    def __init__(self, pretrained=False, norm_layer=None):
        super().__init__()
        assert not pretrained, "no pretrained weights offline"
        norm_layer = norm_layer or nn.BatchNorm2d
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for li, (cout, n) in enumerate([(64, 3), (128, 4), (256, 6), (512, 3)], start=1):
            blocks = []
            for b in range(n):
                blocks.append(_basic_block(cin if b == 0 else cout, cout, (2 if (b == 0 and li > 1) else 1), norm_layer))
            setattr(self, "layer%d" % li, nn.Sequential(*blocks))
            cin = cout
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512, 1000)


def install_shims():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("cv2", COLORMAP_HOT=0)
    tv_models = mod("torchvision.models", resnet34=ResNet34Skeleton)
    tv_tf = mod("torchvision.transforms")
    mod("torchvision", models=tv_models, transforms=tv_tf)
    mod("dotmap", DotMap=DotMap)
    mod("pyhocon", ConfigFactory=object)

    class _NoYolo(nn.Module):
        def __init__(self, *a, **k):
            raise RuntimeError("NeRF-YOLO models.yolo.Model is not available in this container")

    mod("models")
    mod("models.yolo", Model=_NoYolo)
    sys.path.insert(0, REF_SRC)


# --------------------------------------------------------------------------- configs
def model_conf(backbone="resnet34", yolo=False, n_blocks=5, combine_layer=3, has_fine=True):
    mlp = {"type": "resnet", "n_blocks": n_blocks, "d_hidden": 512, "combine_layer": combine_layer,
           "combine_type": "average", "d_out": 4}
    c = {
        "use_encoder": True, "use_global_encoder": False, "use_xyz": True, "canon_xyz": False,
        "use_code": True, "code": {"num_freqs": 6, "freq_factor": 1.5, "include_input": True},
        "use_viewdirs": True, "use_code_viewdirs": False,
        "mlp_coarse": dict(mlp), "mlp_fine": dict(mlp) if has_fine else {"type": "empty"},
        "encoder": {"backbone": backbone, "pretrained": False, "num_layers": 4, "index_padding": "zeros"},
    }
    if yolo:
        c["mlp_coarse"].update({"d_out": 7, "num_scales": 1, "num_anchors_per_scale": 3, "yolo": True})
    return Conf(c)


def load_mlp(mlp, seed, d_latent, d_out):
    sd = synth.mlp_state(seed, d_latent=d_latent, d_out=d_out)
    mlp.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)


class Recorder:
    """Records every random draw the renderer makes (they are inputs of the path: SURVEY.md 7
    hard part 4) and every model call's output."""

    def __init__(self):
        self.draws = []
        self._orig = {}

    def __enter__(self):
        for name in ("rand", "rand_like", "randn_like"):
            self._orig[name] = getattr(torch, name)

            def wrapped(*a, _n=name, **k):
                out = self._orig[_n](*a, **k)
                self.draws.append((_n, out.clone()))
                return out

            setattr(torch, name, wrapped)
        return self

    def __exit__(self, *exc):
        for name, fn in self._orig.items():
            setattr(torch, name, fn)


def record_model_calls(net):
    calls = []
    orig = net.forward

    def fwd(xyz, coarse=True, viewdirs=None, far=False):
        out = orig(xyz, coarse=coarse, viewdirs=viewdirs, far=far)
        calls.append((bool(coarse), out.detach().clone()))
        return out

    net.forward = fwd
    return calls


def pick_rays(all_rays, n, seed):
    """A deterministic subset: centre crop + random others."""
    H, W, _ = all_rays.shape
    rs = np.random.RandomState(seed)
    ys, xs = np.meshgrid(np.arange(H // 2 - 3, H // 2 + 3), np.arange(W // 2 - 3, W // 2 + 3), indexing="ij")
    idx = (ys * W + xs).reshape(-1)
    rest = rs.choice(H * W, size=n - idx.size, replace=False)
    idx = np.concatenate([idx, rest])
    return all_rays.reshape(-1, 8)[idx], idx.astype(np.int64)


def np_(t):
    return t.detach().cpu().numpy()


# --------------------------------------------------------------------------- fixtures
def fixture_nerf(name, H, NS, Kc, Kf, Kfd, n_rays, seed, ebs, d_latent=512, lat_hw=None):
    """d_latent = 1792 builds the reference model with backbone = "custom" (BASELINE configs 3-5: the YOLOv7 encoder's
    channel count, custom_encoder.py:22) and the NeRF renderer / d_out = 4 MLPs; the backbone itself is outside the
    tree, so a dummy module with dims = [1792] is patched in and the latent (lat_hw, default H/2 x W/2) is an input."""
    import util
    import model.encoder as enc_mod
    from model import make_model
    from render import NeRFRenderer

    torch.manual_seed(seed)
    W = H
    focal = torch.tensor(131.25 * H / 128.0)
    c_img = torch.tensor([W * 0.5, H * 0.5])
    z_near, z_far = 0.8, 1.8
    if d_latent == 512:
        conf = model_conf(has_fine=Kf > 0)
    else:
        class DummyYolo(nn.Module):
            def __init__(self):
                super().__init__()
                self.dims = [d_latent]

        enc_mod.YOLOEncoder = DummyYolo
        conf = model_conf(backbone="custom", has_fine=Kf > 0)
    net = make_model(conf).eval()
    load_mlp(net.mlp_coarse, seed * 10 + 1, d_latent, 4)
    if Kf > 0:
        load_mlp(net.mlp_fine, seed * 10 + 2, d_latent, 4)

    src_poses, tgt_pose = synth.scene_cameras(NS)
    Hl, Wl = lat_hw if lat_hw is not None else (H // 2, W // 2)
    lat = synth.latent(seed * 10 + 3, NS, d_latent, Hl, Wl)
    images = torch.zeros(1, NS, 3, H, W)

    # encode(): run the reference's pose / intrinsics handling, but bypass the conv trunk by
    # replacing the encoder's forward with one that installs the seeded latent exactly as
    # SpatialEncoder.forward does at reference encoder.py:169-172.
    enc = net.encoder

    def fake_forward(x):
        enc.latent = torch.from_numpy(lat)
        enc.latent_scaling[0] = enc.latent.shape[-1]
        enc.latent_scaling[1] = enc.latent.shape[-2]
        enc.latent_scaling = enc.latent_scaling / (enc.latent_scaling - 1) * 2.0
        return enc.latent

    enc.forward = fake_forward
    net.encode(images, torch.from_numpy(src_poses)[None], focal, c=c_img[None])

    all_rays = util.gen_rays(torch.from_numpy(tgt_pose)[None], W, H, focal, z_near, z_far, c=c_img)[0]
    rays, ray_idx = pick_rays(all_rays, n_rays, seed)

    renderer = NeRFRenderer(n_coarse=Kc, n_fine=Kf, n_fine_depth=Kfd, depth_std=0.01,
                            white_bkgd=True, eval_batch_size=ebs).eval()
    calls = record_model_calls(net)
    with torch.no_grad(), Recorder() as rec:
        out = renderer(net, rays[None], want_weights=True)

    d = {
        "H": H, "W": W, "NS": NS, "Kc": Kc, "Kf": Kf, "Kfd": Kfd, "seed": seed, "d_latent": d_latent, "Hl": Hl, "Wl": Wl,
        "focal": np_(focal), "c": np_(c_img), "z_near": z_near, "z_far": z_far,
        "src_poses": src_poses, "tgt_pose": tgt_pose, "rays": np_(rays), "ray_idx": ray_idx,
        "all_rays_corner": np_(all_rays[:2, :3]),
        "enc_poses": np_(net.poses), "enc_focal": np_(net.focal), "enc_c": np_(net.c),
        "latent_scaling": np_(enc.latent_scaling),
        "coarse_rgb": np_(out.coarse.rgb[0]), "coarse_depth": np_(out.coarse.depth[0]),
        "coarse_weights": np_(out.coarse.weights[0]),
    }
    draws = rec.draws
    d["u_coarse"] = np_(draws[0][1])
    assert draws[0][0] == "rand_like" and draws[0][1].shape == (n_rays, Kc)
    coarse_calls = torch.cat([o for (cflag, o) in calls if cflag], dim=1)[0]
    d["coarse_out"] = np_(coarse_calls)  # (n_rays*Kc, 4) sigmoid(rgb), relu(sigma)
    if Kf > 0:
        i = 1
        if Kf - Kfd > 0:
            assert draws[1][0] == "rand" and draws[2][0] == "rand_like"
            d["u_fine"] = np_(draws[1][1])
            d["u_fine2"] = np_(draws[2][1])
            i = 3
        if Kfd > 0:
            assert draws[i][0] == "randn_like"
            d["g_depth"] = np_(draws[i][1])
        fine_calls = torch.cat([o for (cflag, o) in calls if not cflag], dim=1)[0]
        d["fine_out"] = np_(fine_calls)
        d["fine_rgb"] = np_(out.fine.rgb[0])
        d["fine_depth"] = np_(out.fine.depth[0])
        d["fine_weights"] = np_(out.fine.weights[0])
    # z_coarse re-derived by the reference's own sample_coarse on the recorded draw
    with torch.no_grad():
        orig = torch.rand_like
        torch.rand_like = lambda x: draws[0][1]
        d["z_coarse"] = np_(renderer.sample_coarse(rays))
        torch.rand_like = orig

    # direct model probe at the point the reference's own smoke test uses
    # (test/model_encode.py:79-81) plus random points, NS views
    rs = np.random.RandomState(seed + 77)
    pts = rs.uniform(-0.6, 0.6, size=(1, 41, 3)).astype(np.float32)
    pts[0, 0] = [5.26, -0.83, -0.18]
    vd = rs.standard_normal((1, 41, 3)).astype(np.float32)
    vd /= np.linalg.norm(vd, axis=-1, keepdims=True)
    vd[0, 0] = 0.0
    with torch.no_grad():
        d["probe_xyz"] = pts[0]
        d["probe_viewdirs"] = vd[0]
        d["probe_out_coarse"] = np_(net(torch.from_numpy(pts), coarse=True, viewdirs=torch.from_numpy(vd)))[0]
        if Kf > 0:
            d["probe_out_fine"] = np_(net(torch.from_numpy(pts), coarse=False, viewdirs=torch.from_numpy(vd)))[0]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("wrote", name, {k: getattr(v, "shape", v) for k, v in d.items() if k in ("rays", "coarse_out", "fine_rgb")})


def fixture_yolo(name, seed, n_rays=40, K=128, ebs=128, out_gain=1.0):
    """YOLO mode: gen_rays_yolo, world->cam poses used as given, raw 21-vector MLP output,
    YoloRenderer aggregation (reference yolo.py:37-114, models.py:119-120,222-224,254-264)."""
    import util
    import model.encoder as enc_mod
    from model import make_model
    from render.yolo import YoloRenderer

    torch.manual_seed(seed)

    class DummyYolo(nn.Module):
        def __init__(self):
            super().__init__()
            self.dims = [1792]

    enc_mod.YOLOEncoder = DummyYolo
    conf = model_conf(backbone="custom", yolo=True, has_fine=False)
    net = make_model(conf).eval()
    sd = synth.mlp_state(seed * 10 + 1, d_latent=1792, d_out=21, out_gain=out_gain)
    net.mlp_coarse.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    NS, H, W = 3, 128, 128
    Hl, Wl = 16, 16
    lat = synth.latent(seed * 10 + 3, NS, 1792, Hl, Wl)
    enc = net.encoder

    def fake_forward(x):
        enc.latent = torch.from_numpy(lat)
        enc.latent_scaling[0] = enc.latent.shape[-1]
        enc.latent_scaling[1] = enc.latent.shape[-2]
        enc.latent_scaling = enc.latent_scaling / (enc.latent_scaling - 1) * 2.0
        return enc.latent

    enc.forward = fake_forward
    # world->cam extrinsics: cameras on a circle looking at the origin, +z forward (OpenCV-like)
    src_c2w, tgt_c2w = synth.scene_cameras(NS, radius=6.0, phi=-25.0)
    flipyz = np.diag([1.0, -1.0, -1.0, 1.0]).astype(np.float32)
    src_w2c = np.stack([np.linalg.inv(p @ flipyz) for p in src_c2w]).astype(np.float32)
    tgt_w2c = np.linalg.inv(tgt_c2w @ flipyz).astype(np.float32)
    focal = torch.tensor([140.0, 150.0])
    c_img = torch.tensor([64.0, 60.0])
    net.encode(torch.zeros(1, NS, 3, H, W), torch.from_numpy(src_w2c)[None], focal[None], c=c_img[None])

    Wc, Hc = 16, 12
    rays_all = util.gen_rays_yolo(torch.from_numpy(tgt_w2c)[None], Wc, Hc, focal / 8, c_img / 8, 1.0, 13.0)
    rays = rays_all.reshape(-1, 8)[:n_rays]
    # include the reference test's hard-coded probe rays (test/yolo_renderer.py:12-15)
    rays = torch.cat([rays, torch.tensor([[1, 2, 3, 4, 5, 6, 0.1, 10], [0, 0, 0, 0, 1, 2, 0.1, 10]], dtype=torch.float32)])
    renderer = YoloRenderer(K, ebs, 1, 3)
    renderer.bind_parallel(net)
    calls = record_model_calls(net)
    with torch.no_grad(), Recorder() as rec:
        out = renderer(rays[None])
    d = {
        "seed": seed, "NS": NS, "H": H, "W": W, "K": K, "Hl": Hl, "Wl": Wl, "out_gain": out_gain,
        "focal": np_(focal), "c": np_(c_img), "src_w2c": src_w2c, "tgt_w2c": tgt_w2c,
        "Wc": Wc, "Hc": Hc, "rays_all": np_(rays_all[0]), "rays": np_(rays),
        "u_coarse": np_(rec.draws[0][1]),
        "raw_out": np_(torch.cat([o for (_, o) in calls], dim=1)[0]),
        "yolo_out": np_(out),
        "enc_poses": np_(net.poses), "enc_focal": np_(net.focal), "enc_c": np_(net.c),
    }
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("wrote", name, d["raw_out"].shape, d["yolo_out"].shape)


def fixture_encoder(name, seed, NS=2, H=64, W=48, use_first_pool=True, residual_gain=1.0):
    """SpatialEncoder.forward (reference encoder.py:110-173) over the ResNet-34 skeleton with
    seeded random weights, eval-mode batch norm."""
    from model.encoder import SpatialEncoder

    torch.manual_seed(seed)
    enc = SpatialEncoder(backbone="resnet34", pretrained=False, num_layers=4, index_padding="zeros",
                         use_first_pool=use_first_pool).eval()
    sd = synth.resnet34_state(seed * 10 + 5, prefix="model.", residual_gain=residual_gain)
    missing = enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert all(k.startswith(("model.layer4", "model.fc")) or "num_batches_tracked" in k for k in missing.missing_keys), missing
    img = synth.images(seed * 10 + 6, NS, H, W)
    with torch.no_grad():
        lat = enc(torch.from_numpy(img))
        levels = [np_(t) for t in enc.latents]
    # index(): bilinear lookup incl. out-of-image points (zeros padding), reference encoder.py:79-108
    rs = np.random.RandomState(seed)
    uv = rs.uniform(-8.0, max(H, W) + 8.0, size=(NS, 50, 2)).astype(np.float32)
    with torch.no_grad():
        samp = enc.index(torch.from_numpy(uv), None, torch.tensor([float(W), float(H)]))
    d = {"seed": seed, "NS": NS, "H": H, "W": W, "residual_gain": residual_gain, "latent": np_(lat),
         "latent_scaling": np_(enc.latent_scaling),
         "level1": levels[1][:, :8], "level3": levels[3][:, :8], "uv": uv, "index_out": np_(samp)}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("wrote", name, d["latent"].shape)


def fixture_rays(name):
    """util.gen_rays / gen_rays_yolo (reference util.py:240-278, 808-876) incl. the probe call of
    the reference's own smoke test (test/gen_rays_render.py:82)."""
    import util

    poses = torch.from_numpy(np.stack([synth.pose_spherical(30, -20, 1.3), synth.pose_spherical(200, -35, 2.0)]))
    r1 = util.gen_rays(poses, 20, 12, torch.tensor(35.5), 0.8, 1.8, c=None)
    r2 = util.gen_rays(poses[:1], 16, 16, torch.tensor([30.0, 28.0]), 0.5, 2.5, c=torch.tensor([7.5, 9.25]))
    flipyz = np.diag([1.0, -1.0, -1.0, 1.0]).astype(np.float32)
    w2c = torch.from_numpy(np.stack([np.linalg.inv(p.numpy() @ flipyz) for p in poses]).astype(np.float32))
    focal = torch.tensor([1450.0, 1460.0])
    cc = torch.tensor([960.0, 540.0])
    r3 = util.gen_rays_yolo(w2c, 48, 27, focal / 10, cc / 10, 5.0, 10.0)
    d = {"poses": np_(poses), "r1": np_(r1), "r2": np_(r2), "w2c": np_(w2c), "yolo_focal": np_(focal / 10),
         "yolo_c": np_(cc / 10), "r3": np_(r3)}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("wrote", name, r1.shape, r2.shape, r3.shape)


def fixture_yolo_tail(name, seed=9):
    """Detection tail of the YOLO eval path (reference util.py:633-689 convert_cells_to_bboxes,
    :691-722 nms -- including its remove-while-iterating behaviour --, :765-802 calculate_tp_fp_fn,
    :575-630 iou) on synthetic cell grids at the real data geometry (16 x 30 cells, 3 anchors)."""
    import util

    rs = np.random.RandomState(seed)
    h, w, A = 16, 30, 3
    anchors = torch.tensor([[0.28, 0.22], [0.38, 0.48], [0.9, 0.78]]) * torch.tensor([float(w), float(h)])
    cases = {}
    for ci, (n_obj, conf_lo) in enumerate([(6, 0.2), (25, 0.05), (0, 0.0)]):
        pred = np.zeros((1, h, w, A, 7), dtype=np.float32)
        pred[..., 0] = rs.uniform(0.0, conf_lo, size=(1, h, w, A))           # background confidences
        pred[..., 1:5] = rs.standard_normal((1, h, w, A, 4)) * 0.5
        pred[..., 3:5] -= 1.5                                                  # exp(.)*anchor stays moderate
        pred[..., 5:7] = rs.standard_normal((1, h, w, A, 2))
        tgt = np.zeros((1, h, w, A, 6), dtype=np.float32)
        for _ in range(n_obj):
            y, x, a = rs.randint(h), rs.randint(w), rs.randint(A)
            tgt[0, y, x, a] = [1.0, rs.uniform(0.2, 0.8), rs.uniform(0.2, 0.8), rs.uniform(1.0, 4.0), rs.uniform(1.0, 4.0), rs.randint(2)]
            # a cluster of confident, heavily overlapping predictions around the object (exercises NMS)
            for dy, dx in ((0, 0), (0, 1), (1, 0), (0, -1)):
                yy, xx = min(max(y + dy, 0), h - 1), min(max(x + dx, 0), w - 1)
                for aa in range(A):
                    if rs.rand() < 0.7:
                        pred[0, yy, xx, aa, 0] = rs.uniform(0.5, 1.0)
                        pred[0, yy, xx, aa, 3:5] = np.log(tgt[0, y, x, a, 3:5] / anchors[aa].numpy()) + rs.standard_normal(2) * 0.1
        p_boxes = util.convert_cells_to_bboxes(torch.from_numpy(pred), anchors, h, w, is_predictions=True)[0]
        t_boxes = util.convert_cells_to_bboxes(torch.from_numpy(tgt), anchors, h, w, is_predictions=False)[0]
        d = {"pred": pred, "tgt": tgt, "p_boxes": np.array(p_boxes, dtype=np.float64), "t_boxes": np.array(t_boxes, dtype=np.float64)}
        for thr_i, (iou_t, conf_t) in enumerate([(0.75, 0.45), (0.3, 0.1)]):
            kept, hc, above = util.nms([list(b) for b in p_boxes], iou_t, conf_t)
            d["nms%d_kept" % thr_i] = np.array(kept, dtype=np.float64).reshape(-1, 6)
            d["nms%d_meta" % thr_i] = np.array([iou_t, conf_t, hc, above], dtype=np.float64)
            tp, fp, fn = util.calculate_tp_fp_fn([list(b) for b in t_boxes], [list(b) for b in p_boxes], iou_t, conf_t, 0.2)
            d["tpfpfn%d" % thr_i] = np.array([tp, fp, fn], dtype=np.int64)
        for k, v in d.items():
            cases["c%d_%s" % (ci, k)] = v
    # duplicate rows: the reference's `bboxes_filtered.remove(box)` (util.py:719) deletes the first EQUAL row, which differs
    # from deleting the row at hand when an identical row further up was skipped by the remove-while-iterating loop.
    # F suppresses A and X' (= X); A's removal makes the iterator skip X; B (same confidence) sits between X and X'.
    F_ = [0.0, 0.9, 0.50, 0.50, 0.20, 0.20]
    A_ = [1.0, 0.8, 0.51, 0.50, 0.20, 0.20]
    X_ = [0.0, 0.8, 0.50, 0.52, 0.20, 0.20]
    B_ = [1.0, 0.8, 0.62, 0.62, 0.20, 0.20]
    dup = [F_, A_, X_, B_, list(X_), [0.0, 0.7, 0.10, 0.10, 0.05, 0.05]]
    for thr_i, iou_t in enumerate((0.5, 0.15)):
        kept, hc, above = util.nms([list(b) for b in dup], iou_t, 0.3)
        cases["dup_nms%d_kept" % thr_i] = np.array(kept, dtype=np.float64).reshape(-1, 6)
        cases["dup_nms%d_meta" % thr_i] = np.array([iou_t, 0.3, hc, above], dtype=np.float64)
    cases["dup_boxes"] = np.array(dup, dtype=np.float64)
    cases["anchors"] = anchors.numpy()
    cases["hw"] = np.array([h, w, A])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **cases)
    print("wrote", name, {k: v.shape for k, v in cases.items() if "kept" in k or "tpfpfn" in k})


def fixture_nerf_variants(name, seed=11):
    """Renderer options and batch shapes the standard fixtures do not exercise, from the reference itself:
      a_: lindisp=True, black background, importance samples only (n_fine_depth = 0)
      b_: depth samples only (n_fine == n_fine_depth: no importance draws)
      c_: super-batch of 2 scenes with per-scene focal (SB,2) and principal point (SB,2)."""
    import util
    from model import make_model
    from render import NeRFRenderer

    torch.manual_seed(seed)
    H = W = 32
    NS = 2
    z_near, z_far = 0.8, 1.8
    d = {"H": H, "W": W, "NS": NS, "seed": seed, "z_near": z_near, "z_far": z_far}

    def build(SB, focal, c_img, lat_seed):
        net = make_model(model_conf(has_fine=True)).eval()
        load_mlp(net.mlp_coarse, seed * 10 + 1, 512, 4)
        load_mlp(net.mlp_fine, seed * 10 + 2, 512, 4)
        lat = np.concatenate([synth.latent(lat_seed + i, NS, 512, H // 2, W // 2) for i in range(SB)])
        enc = net.encoder

        def fake_forward(x):
            enc.latent = torch.from_numpy(lat)
            enc.latent_scaling[0] = enc.latent.shape[-1]
            enc.latent_scaling[1] = enc.latent.shape[-2]
            enc.latent_scaling = enc.latent_scaling / (enc.latent_scaling - 1) * 2.0
            return enc.latent

        enc.forward = fake_forward
        poses = np.stack([synth.scene_cameras(NS, radius=1.3 + 0.2 * i)[0] for i in range(SB)])
        net.encode(torch.zeros(SB, NS, 3, H, W), torch.from_numpy(poses), focal, c=c_img)
        return net, lat, poses

    def run(prefix, net, rays, **opts):
        renderer = NeRFRenderer(depth_std=0.01, eval_batch_size=700, **opts).eval()
        with torch.no_grad(), Recorder() as rec:
            out = renderer(net, rays, want_weights=True)
        d[prefix + "rays"] = np_(rays)
        for i, (kind, t) in enumerate(rec.draws):
            d["%sdraw%d_%s" % (prefix, i, kind)] = np_(t)
        for part in ("coarse", "fine"):
            for k in ("rgb", "depth", "weights"):
                d["%s%s_%s" % (prefix, part, k)] = np_(out[part][k])

    def some_rays(pose, focal, c, n, rs):
        allr = util.gen_rays(torch.from_numpy(pose)[None], W, H, focal, z_near, z_far, c=c)[0].reshape(-1, 8)
        return allr[torch.from_numpy(rs.choice(H * W, n, replace=False))]

    rs = np.random.RandomState(seed)
    focal1, c1 = torch.tensor(33.0), torch.tensor([[15.5, 16.5]])
    net, lat, poses = build(1, focal1, c1, seed * 10 + 3)
    d["ab_latent_seed"], d["ab_poses"], d["ab_focal"], d["ab_c"] = seed * 10 + 3, poses, np_(focal1), np_(c1)
    tgt = synth.pose_spherical(110.0, -20.0, 1.3)
    rays = some_rays(tgt, focal1, c1[0], 24, rs)[None]
    run("a_", net, rays, n_coarse=16, n_fine=8, n_fine_depth=0, white_bkgd=False, lindisp=True)
    run("b_", net, rays, n_coarse=16, n_fine=8, n_fine_depth=8, white_bkgd=True, lindisp=False)

    focal2 = torch.tensor([[30.0, 30.0], [34.0, 33.0]])
    c2 = torch.tensor([[16.0, 16.0], [15.0, 17.5]])
    net2, lat2, poses2 = build(2, focal2, c2, seed * 10 + 5)
    d["c_latent_seed"], d["c_poses"], d["c_focal"], d["c_c"] = seed * 10 + 5, poses2, np_(focal2), np_(c2)
    rays2 = torch.stack([some_rays(synth.pose_spherical(100.0 + 30 * i, -20.0, 1.3), torch.tensor(31.0),
                                   torch.tensor([16.0, 16.0]), 20, rs) for i in range(2)])
    run("c_", net2, rays2, n_coarse=16, n_fine=8, n_fine_depth=4, white_bkgd=True, lindisp=False)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("wrote", name, sorted(k for k in d if "draw" in k))


def fixture_yolo_cull(name, seed=13):
    """YOLO mode's latent culling (reference models.py:222-224,254-264): latent := 0 where z_cam >= 0 and where it
    is NaN.  View 0 has the identity extrinsic so that camera-space coordinates are exact: the points include
    z == 0 with x == y == 0 (0/0), z == 0 with x != 0 (+-inf uv), z > 0, z < 0 and far out-of-image projections."""
    import model.encoder as enc_mod
    from model import make_model

    torch.manual_seed(seed)

    class DummyYolo(nn.Module):
        def __init__(self):
            super().__init__()
            self.dims = [1792]

    enc_mod.YOLOEncoder = DummyYolo
    net = make_model(model_conf(backbone="custom", yolo=True, has_fine=False)).eval()
    load_mlp(net.mlp_coarse, seed * 10 + 1, 1792, 21)
    NS, H, W, Hl, Wl = 2, 64, 64, 8, 8
    lat = synth.latent(seed * 10 + 3, NS, 1792, Hl, Wl)
    enc = net.encoder

    def fake_forward(x):
        enc.latent = torch.from_numpy(lat)
        enc.latent_scaling[0] = enc.latent.shape[-1]
        enc.latent_scaling[1] = enc.latent.shape[-2]
        enc.latent_scaling = enc.latent_scaling / (enc.latent_scaling - 1) * 2.0
        return enc.latent

    enc.forward = fake_forward
    src_c2w, _ = synth.scene_cameras(NS, radius=4.0, phi=-25.0)
    flipyz = np.diag([1.0, -1.0, -1.0, 1.0]).astype(np.float32)
    w2c = np.stack([np.eye(4, dtype=np.float32), np.linalg.inv(src_c2w[1] @ flipyz).astype(np.float32)])
    focal, c_img = torch.tensor([40.0, 44.0]), torch.tensor([32.0, 30.0])
    net.encode(torch.zeros(1, NS, 3, H, W), torch.from_numpy(w2c)[None], focal[None], c=c_img[None])
    rs = np.random.RandomState(seed)
    pts = rs.uniform(-3.0, 3.0, size=(96, 3)).astype(np.float32)
    pts[0] = [0.0, 0.0, 0.0]        # view 0: 0/0
    pts[1] = [0.5, 0.0, 0.0]        # view 0: +inf u, 0/0 v
    pts[2] = [-0.5, 0.25, 0.0]      # view 0: -inf, +inf
    pts[3] = [0.1, 0.1, 2.0]        # view 0: z > 0 -> culled
    pts[4] = [0.1, 0.1, -2.0]       # view 0: z < 0 -> kept, projects inside the image
    pts[5] = [50.0, -60.0, -0.5]    # view 0: far outside the image
    pts[6] = [0.0, 0.0, -1e-30]     # view 0: tiny negative z
    vd = rs.standard_normal((96, 3)).astype(np.float32)
    with torch.no_grad():
        out = net(torch.from_numpy(pts)[None], coarse=True, viewdirs=torch.from_numpy(vd)[None])[0]
    d = {"seed": seed, "NS": NS, "H": H, "W": W, "Hl": Hl, "Wl": Wl, "w2c": w2c, "focal": np_(focal), "c": np_(c_img),
         "xyz": pts, "viewdirs": vd, "out": np_(out)}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("wrote", name, d["out"].shape, "finite:", bool(np.isfinite(d["out"]).all()))


def fixture_sched(name):
    """NeRFRenderer.sched_step (reference nerf.py:324-344): sample-count schedule as the trainer drives it."""
    from render import NeRFRenderer
    sched = [[2, 4, 9], [32, 64, 96], [8, 16, 24]]
    steps = [1, 1, 1, 2, 5, 1, 3]
    ren = NeRFRenderer(n_coarse=16, n_fine=4, sched=sched)
    rows = []
    for st in steps:
        ren.sched_step(st)
        rows.append([ren.n_coarse, ren.n_fine, int(ren.iter_idx), int(ren.last_sched)])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), sched=np.array(sched), steps=np.array(steps), rows=np.array(rows))
    print("wrote", name, rows)


def fixture_mlp_shapes(name, seed=17):
    """ResnetFC shapes the standard fixtures do not cover (resnetfc.py:134-186 with other conf values): combine_layer = 0
    (the cross-view mean directly after lin_in, no lin_z at all), combine_layer = 1 with 4 blocks, and a single block
    with combine_layer = 1000 on one view.  Model probes only."""
    from model import make_model

    torch.manual_seed(seed)
    H = W = 32
    d = {"H": H, "W": W, "seed": seed}
    rs = np.random.RandomState(seed)
    pts = rs.uniform(-0.5, 0.5, size=(1, 70, 3)).astype(np.float32)
    vd = rs.standard_normal((1, 70, 3)).astype(np.float32)
    d["xyz"], d["viewdirs"] = pts[0], vd[0]
    for tag, (nb, cl, ns) in {"a": (2, 0, 2), "b": (4, 1, 3), "c": (1, 1000, 1)}.items():
        net = make_model(model_conf(n_blocks=nb, combine_layer=cl, has_fine=False)).eval()
        sd = synth.mlp_state(seed * 10 + ord(tag), n_blocks=nb, combine_layer=cl)
        net.mlp_coarse.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        lat = synth.latent(seed * 10 + 3, ns, 512, H // 2, W // 2)
        enc = net.encoder

        def fake_forward(x, enc=enc, lat=lat):
            enc.latent = torch.from_numpy(lat)
            enc.latent_scaling[0] = enc.latent.shape[-1]
            enc.latent_scaling[1] = enc.latent.shape[-2]
            enc.latent_scaling = enc.latent_scaling / (enc.latent_scaling - 1) * 2.0
            return enc.latent

        enc.forward = fake_forward
        poses = synth.scene_cameras(ns)[0]
        net.encode(torch.zeros(1, ns, 3, H, W), torch.from_numpy(poses)[None], torch.tensor(33.0))
        with torch.no_grad():
            out = net(torch.from_numpy(pts), coarse=True, viewdirs=torch.from_numpy(vd))[0]
        d[tag + "_cfg"] = np.array([nb, cl, ns])
        d[tag + "_poses"] = poses
        d[tag + "_out"] = np_(out)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("wrote", name, {k: v.shape for k, v in d.items() if k.endswith("_out")})


GRAD_SAMPLES = 192  # entries kept per gradient tensor (plus its sum, abs-sum and max-abs)


def grad_digest(t, rs):
    """A compact stand-in for a gradient tensor: sum, sum of |.|, max |.| and GRAD_SAMPLES seeded entries."""
    f = t.detach().reshape(-1).double()
    idx = rs.choice(f.numel(), size=min(GRAD_SAMPLES, f.numel()), replace=False)
    return np.array([float(f.sum()), float(f.abs().sum()), float(f.abs().max())]), idx.astype(np.int64), f[torch.from_numpy(idx)].numpy()


def fixture_grads(name, seed=19):
    """Gradients of the reference's training loss (PixelNerfTrainer.calc_losses: MSE on coarse.rgb + MSE on fine.rgb,
    trainlib/PixelNerfTrainer.py:133-156) w.r.t. every MLP parameter and the latent, on a small 2-view render --
    the pinned target of the backward pass (docs/backward_plan.md).  Per tensor a digest is stored, not the tensor."""
    import util
    from model import make_model
    from render import NeRFRenderer

    torch.manual_seed(seed)
    H = W = 32
    NS, Kc, Kf, Kfd, n_rays = 2, 16, 8, 4, 24
    net = make_model(model_conf(has_fine=True)).train()
    load_mlp(net.mlp_coarse, seed * 10 + 1, 512, 4)
    load_mlp(net.mlp_fine, seed * 10 + 2, 512, 4)
    lat = torch.from_numpy(synth.latent(seed * 10 + 3, NS, 512, H // 2, W // 2)).requires_grad_()
    enc = net.encoder

    def fake_forward(x):
        enc.latent = lat
        enc.latent_scaling[0] = enc.latent.shape[-1]
        enc.latent_scaling[1] = enc.latent.shape[-2]
        enc.latent_scaling = enc.latent_scaling / (enc.latent_scaling - 1) * 2.0
        return enc.latent

    enc.forward = fake_forward
    poses, tgt = synth.scene_cameras(NS)
    focal, c_img = torch.tensor(33.0), torch.tensor([[16.0, 16.0]])
    net.encode(torch.zeros(1, NS, 3, H, W), torch.from_numpy(poses)[None], focal, c=c_img)
    rs = np.random.RandomState(seed)
    allr = util.gen_rays(torch.from_numpy(tgt)[None], W, H, focal, 0.8, 1.8, c=c_img[0])[0].reshape(-1, 8)
    rays = allr[torch.from_numpy(rs.choice(H * W, n_rays, replace=False))][None]
    gt = torch.from_numpy(rs.uniform(0, 1, size=(1, n_rays, 3)).astype(np.float32))
    renderer = NeRFRenderer(n_coarse=Kc, n_fine=Kf, n_fine_depth=Kfd, depth_std=0.01, white_bkgd=True,
                            eval_batch_size=500).train()
    with Recorder() as rec:
        out = renderer(net, rays, want_weights=True)
    loss = torch.nn.functional.mse_loss(out.coarse.rgb, gt) + torch.nn.functional.mse_loss(out.fine.rgb, gt)
    loss.backward()
    d = {"H": H, "W": W, "NS": NS, "Kc": Kc, "Kf": Kf, "Kfd": Kfd, "seed": seed, "poses": poses, "focal": np_(focal),
         "c": np_(c_img), "rays": np_(rays[0]), "gt": np_(gt[0]), "loss": float(loss),
         "coarse_rgb": np_(out.coarse.rgb[0]), "fine_rgb": np_(out.fine.rgb[0])}
    for i, (kind, t) in enumerate(rec.draws):
        d["draw%d_%s" % (i, kind)] = np_(t)
    names = []
    for pre, mlp in (("mlp_coarse.", net.mlp_coarse), ("mlp_fine.", net.mlp_fine)):
        for k, p in mlp.named_parameters():
            assert p.grad is not None, pre + k
            names.append(pre + k)
            d["g:" + pre + k + ":stat"], d["g:" + pre + k + ":idx"], d["g:" + pre + k + ":val"] = grad_digest(p.grad, rs)
    d["g:latent:stat"], d["g:latent:idx"], d["g:latent:val"] = grad_digest(lat.grad, rs)
    d["grad_names"] = np.array(names + ["latent"])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("wrote", name, "loss %.6f" % float(loss), len(names), "parameter gradients + latent")


def fixture_enc_render(name, seed=23, H=128, NS=3, Kc=64, Kf=32, Kfd=16, n_rays=100, ebs=3000):
    """End to end, nothing bypassed: the reference's own SpatialEncoder.forward (encoder.py:110-173, ResNet-34 skeleton,
    eval-mode batch norm) on seeded images -> PixelNeRFNet.encode -> NeRFRenderer.forward (nerf.py:257-309) at the C2
    shape.  The trunk is the well-conditioned one bench.py uses (synth.resnet34_state(residual_gain=0.25): latent O(1)
    like a trained trunk's), so RGB / sigma are comparable at an ABSOLUTE 1e-4.  Images and weights are regenerated
    from their seeds by the test; the latent is stored as a seeded sample of entries + its max |.|."""
    import util
    from model import make_model
    from render import NeRFRenderer

    torch.manual_seed(seed)
    W = H
    focal = torch.tensor(131.25 * H / 128.0)
    c_img = torch.tensor([W * 0.5, H * 0.5])
    z_near, z_far = 0.8, 1.8
    net = make_model(model_conf(has_fine=True)).eval()
    load_mlp(net.mlp_coarse, seed * 10 + 1, 512, 4)
    load_mlp(net.mlp_fine, seed * 10 + 2, 512, 4)
    esd = synth.resnet34_state(seed * 10 + 4, prefix="encoder.model.", residual_gain=0.25)
    res = net.load_state_dict({k: torch.from_numpy(v) for k, v in esd.items()}, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(k.startswith(("encoder.model.layer4", "encoder.model.fc", "mlp_", "code.")) or "num_batches_tracked" in k
               for k in res.missing_keys), res.missing_keys
    src_poses, tgt_pose = synth.scene_cameras(NS)
    images = torch.from_numpy(synth.images(seed * 10 + 5, NS, H, W))
    with torch.no_grad():
        net.encode(images[None], torch.from_numpy(src_poses)[None], focal, c=c_img[None])
    lat = net.encoder.latent.detach()
    all_rays = util.gen_rays(torch.from_numpy(tgt_pose)[None], W, H, focal, z_near, z_far, c=c_img)[0]
    rays, ray_idx = pick_rays(all_rays, n_rays, seed)
    renderer = NeRFRenderer(n_coarse=Kc, n_fine=Kf, n_fine_depth=Kfd, depth_std=0.01, white_bkgd=True,
                            eval_batch_size=ebs).eval()
    calls = record_model_calls(net)
    with torch.no_grad(), Recorder() as rec:
        out = renderer(net, rays[None], want_weights=True)
    draws = rec.draws
    assert [k for k, _ in draws] == ["rand_like", "rand", "rand_like", "randn_like"]
    rs = np.random.RandomState(seed)
    lat_idx = rs.choice(lat.numel(), size=20000, replace=False).astype(np.int64)
    d = {
        "H": H, "W": W, "NS": NS, "Kc": Kc, "Kf": Kf, "Kfd": Kfd, "seed": seed, "residual_gain": 0.25,
        "focal": np_(focal), "c": np_(c_img), "z_near": z_near, "z_far": z_far, "src_poses": src_poses,
        "tgt_pose": tgt_pose, "rays": np_(rays), "ray_idx": ray_idx,
        "latent_shape": np.array(lat.shape), "latent_idx": lat_idx, "latent_val": np_(lat.reshape(-1)[torch.from_numpy(lat_idx)]),
        "latent_absmax": float(lat.abs().max()), "latent_std": float(lat.std()),
        "u_coarse": np_(draws[0][1]), "u_fine": np_(draws[1][1]), "u_fine2": np_(draws[2][1]), "g_depth": np_(draws[3][1]),
        "coarse_out": np_(torch.cat([o for (cf, o) in calls if cf], dim=1)[0]),
        "fine_out": np_(torch.cat([o for (cf, o) in calls if not cf], dim=1)[0]),
    }
    for part in ("coarse", "fine"):
        for k in ("rgb", "depth", "weights"):
            d["%s_%s" % (part, k)] = np_(out[part][k][0])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("wrote", name, "latent absmax %.3f std %.3f" % (d["latent_absmax"], d["latent_std"]),
          "sigma max %.2f" % float(d["fine_out"][:, 3].max()))


def fixture_ckpt(name, seed=29):
    """PixelNeRFNet.load_weights / save_weights (reference models.py:320-370) driven through the (resume, opt_init)
    combinations and the trainer's epochNum calls (train/trainlib/trainer.py:246-256) with real files in a scratch
    directory.  Checkpoints are told apart by a marker (every entry of mlp_coarse.lin_out.bias).  Stored as JSON."""
    import json
    import shutil
    import tempfile
    import warnings
    from types import SimpleNamespace
    from model import make_model

    torch.manual_seed(seed)

    def fresh(marker):
        net = make_model(model_conf(has_fine=True))
        with torch.no_grad():
            net.mlp_coarse.lin_out.bias.fill_(marker)
        return net

    def marker_of(obj):
        sd = obj if isinstance(obj, dict) else obj.state_dict()
        return float(sd["mlp_coarse.lin_out.bias"][0])

    def listing(root):
        out = {}
        for fn in sorted(os.listdir(root)):
            out[fn] = marker_of(torch.load(os.path.join(root, fn), map_location="cpu"))
        return out

    tmp = tempfile.mkdtemp(prefix="pny_ckpt_")
    rec = {"load": [], "save": []}
    try:
        for files in (("pixel_nerf_init", "pixel_nerf_latest"), ("pixel_nerf_latest",), ("pixel_nerf_init",), ()):
            root = os.path.join(tmp, "exp")
            shutil.rmtree(root, ignore_errors=True)
            os.makedirs(root)
            for fn, mk in (("pixel_nerf_init", 1.0), ("pixel_nerf_latest", 2.0)):
                if fn in files:
                    torch.save(fresh(mk).state_dict(), os.path.join(root, fn))
            for resume in (False, True):
                for opt_init in (False, True):
                    args = SimpleNamespace(checkpoints_path=tmp, name="exp", resume=resume)
                    net = fresh(0.0)
                    with warnings.catch_warnings(record=True) as w:
                        warnings.simplefilter("always")
                        ret = net.load_weights(args, opt_init=opt_init)
                    rec["load"].append({"files": list(files), "resume": resume, "opt_init": opt_init,
                                        "marker_after": marker_of(net), "returns_self": ret is net,
                                        "returns_none": ret is None, "warned": len(w) > 0})
        # save_weights: the trainer's sequence (best snapshot, per-epoch snapshot, plain save), then opt_init
        root = os.path.join(tmp, "exp")
        shutil.rmtree(root, ignore_errors=True)
        os.makedirs(root)
        args = SimpleNamespace(checkpoints_path=tmp, name="exp", resume=True)
        steps = [(3.0, dict(epochNum="_best")), (4.0, dict()), (5.0, dict(epochNum="_best")), (6.0, dict(epochNum="7")),
                 (7.0, dict()), (8.0, dict(opt_init=True)), (9.0, dict(opt_init=True)), (10.0, dict(opt_init=True, epochNum="x"))]
        for mk, kw in steps:
            ret = fresh(mk).save_weights(args, **kw)
            rec["save"].append({"marker": mk, "kwargs": kw, "returns_self": ret is not None, "files": listing(root)})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    with open(os.path.join(OUT, name + ".json"), "w") as fh:
        json.dump(rec, fh, indent=1, sort_keys=True)
    print("wrote", name, len(rec["load"]), "load cases,", len(rec["save"]), "save steps")


def main():
    os.makedirs(OUT, exist_ok=True)
    install_shims()
    only = set(sys.argv[1:])   # optional: fixture names to (re)generate; default all
    if only:
        g = globals()
        for fn_name in [k for k in g if k.startswith("fixture_")]:
            f = g[fn_name]

            def gated(name, *a, _f=f, **k):
                if name in only:
                    return _f(name, *a, **k)

            g[fn_name] = gated
    fixture_rays("rays")
    # C1-like: single view, coarse only (BASELINE config 1 at reduced ray count)
    fixture_nerf("nerf_c1", H=64, NS=1, Kc=32, Kf=0, Kfd=0, n_rays=80, seed=1, ebs=1000)
    # C2-like: 3 views, 64 coarse + 32 fine (16 depth), chunked model calls
    fixture_nerf("nerf_c2", H=128, NS=3, Kc=64, Kf=32, Kfd=16, n_rays=100, seed=7, ebs=3000)
    fixture_yolo("yolo_c3", seed=3)
    fixture_encoder("encoder", seed=4)
    fixture_yolo_tail("yolo_tail")
    fixture_encoder("encoder_nopool", seed=5, NS=1, H=48, W=32, use_first_pool=False)  # conf/exp/sn64.conf
    fixture_nerf_variants("nerf_variants")
    fixture_yolo_cull("yolo_cull")
    fixture_sched("sched")
    fixture_mlp_shapes("mlp_shapes")
    fixture_grads("nerf_grads")
    # BASELINE configs 3 and 4 at reduced ray counts: L = 1792 conditioning, NeRF renderer, d_out = 4
    fixture_nerf("nerf_c3", H=128, NS=3, Kc=64, Kf=32, Kfd=16, n_rays=100, seed=31, ebs=3000, d_latent=1792, lat_hw=(16, 16))
    fixture_nerf("nerf_c4", H=400, NS=3, Kc=128, Kf=64, Kfd=32, n_rays=64, seed=37, ebs=5000, d_latent=1792, lat_hw=(50, 50))
    # companions with O(1) magnitudes (absolute 1e-4 is meaningful): YOLO raw outputs, encoder latent
    fixture_yolo("yolo_c3_unit", seed=41, out_gain=0.05)
    fixture_encoder("encoder_unit", seed=43, NS=1, H=64, W=48, residual_gain=0.25)
    fixture_enc_render("enc_render")
    fixture_ckpt("ckpt")


if __name__ == "__main__":
    main()
