"""
BASELINE.json configs 3, 4 and 5 and the encoder -> render path under -m gpu (real MI355X, through the C ABI).

  C3  128x128, 3 views, L = 1792 conditioning ("custom" backbone), NeRF renderer 64 + 32 (16)
  C4  400x400, 3 views, L = 1792, 128 + 64 (32)
  C5  8 scenes x C3, one scene per GPU (here: 8 scene handles on one GPU, the per-rank work of that mode)
Each against (1) golden vectors captured from the reference itself at reduced ray counts
(tests/golden/nerf_c3.npz, nerf_c4.npz; tools/make_golden.py), (2) the oracle, and (3) size-independent properties
at the full frame size.  `enc_render` is the end-to-end fixture in which the REFERENCE ran its own
SpatialEncoder.forward: here images -> pny_scene_encode -> pny_render, RGB / sigma at an absolute 1e-4.

All tolerances in this file are ABSOLUTE 1e-4 (north_star), on fixtures whose magnitudes are O(1).
"""
import os

import numpy as np
import pytest
import torch

import pnyolo_oracle as orc
from helpers import (DEV, check_against_nerf_golden, dt, load_mlp, maxabs, nerf_net, oracle_scene, render_debug)
from pixel_nerf_yolo_amd import conf as pconf
from pixel_nerf_yolo_amd import synth
from pixel_nerf_yolo_amd.model import make_model
from pixel_nerf_yolo_amd.render import NeRFRenderer, make_renderer
from pixel_nerf_yolo_amd.util import gen_rays

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(params=["off", "on", "on+f16x2"])
def projection(request, monkeypatch):
    """Scene defaults read at pny_scene_create: reference operation order / projected latent on the fp32 matrix path /
    projected latent on the split-f16 matrix path (every projected launch).  Returns "off" or "on"."""
    mode, _, prec = request.param.partition("+")
    monkeypatch.setenv("PNYOLO_PROJECTION", mode)
    monkeypatch.setenv("PNYOLO_MLP_PRECISION", prec or "f32")
    # the f16x2 leg runs the 64-sample shape of the kernel (the one a full frame uses) whatever the launch size; the 32-sample
    # shape that small launches pick by themselves is held to it bit for bit (test_f16x2_split_shape_is_bit_identical)
    monkeypatch.setenv("PNYOLO_H2_SPLIT", "0")
    return mode


def c3_conf():
    c = pconf.default_mv()
    c.d["model"]["encoder"]["backbone"] = "custom"
    return c


# --------------------------------------------------------------------------- goldens from the reference
@pytest.mark.parametrize("name,seed", [("nerf_c3", 31), ("nerf_c4", 37)])
def test_render_c3_c4_golden(golden, name, seed, projection):
    g = golden(name)
    kc, kf, kfd = int(g["Kc"]), int(g["Kf"]), int(g["Kfd"])
    net = nerf_net(g, seed)
    assert net.d_latent == 1792 and net.d_out == 4
    ren = NeRFRenderer(n_coarse=kc, n_fine=kf, n_fine_depth=kfd, depth_std=0.01, white_bkgd=True).eval()
    draws = {k: g[k] for k in ("u_coarse", "u_fine", "u_fine2", "g_depth")}
    out, dbg = render_debug(ren, net, g["rays"], draws, kc, kc + kf)
    check_against_nerf_golden(g, out, dbg, TOL)
    st = net.last_mlp_stats(full=True)
    assert st["launches"] == 2 and st["projected"] == (projection == "on")
    xyz, vd = dt(g["probe_xyz"])[None], dt(g["probe_viewdirs"])[None]
    with torch.no_grad():
        assert maxabs(net(xyz, coarse=True, viewdirs=vd)[0], g["probe_out_coarse"]) < TOL
        assert maxabs(net(xyz, coarse=False, viewdirs=vd)[0], g["probe_out_fine"]) < TOL


def test_encoder_to_render_golden(golden, projection):
    """The reference ran SpatialEncoder.forward -> encode -> NeRFRenderer.forward; here the same seeded images go
    through pny_scene_encode (HIP ResNet-34 trunk) and pny_render.  Encoder error reaches the RGB / sigma comparison."""
    g = golden("enc_render")
    seed, ns, H, W = int(g["seed"]), int(g["NS"]), int(g["H"]), int(g["W"])
    net = make_model(pconf.default_mv()["model"]).eval()
    load_mlp(net.mlp_coarse, seed * 10 + 1, 512, 4)
    load_mlp(net.mlp_fine, seed * 10 + 2, 512, 4)
    esd = synth.resnet34_state(seed * 10 + 4, residual_gain=float(g["residual_gain"]))
    res = net.load_state_dict({k: torch.from_numpy(v) for k, v in esd.items()}, strict=False)
    assert not res.unexpected_keys
    net = net.to(DEV)
    images = torch.from_numpy(synth.images(seed * 10 + 5, ns, H, W))
    net.encode(images[None], torch.from_numpy(g["src_poses"])[None], torch.tensor(float(g["focal"])),
               c=torch.from_numpy(g["c"])[None])
    lat = net.latent(0)
    assert tuple(lat.shape) == tuple(int(v) for v in g["latent_shape"])
    e_lat = maxabs(lat.reshape(-1)[torch.from_numpy(g["latent_idx"]).to(DEV)], g["latent_val"])
    assert e_lat < TOL, e_lat                       # latent O(1) (std 1.8, max 22.6): absolute
    ren = NeRFRenderer(n_coarse=64, n_fine=32, n_fine_depth=16, depth_std=0.01, white_bkgd=True).eval()
    draws = {k: g[k] for k in ("u_coarse", "u_fine", "u_fine2", "g_depth")}
    out, dbg = render_debug(ren, net, g["rays"], draws, 64, 96)
    check_against_nerf_golden(g, out, dbg, TOL)


def test_yolo_render_unit_golden(golden, projection):
    """YOLO mode with O(1) raw outputs (lin_out x 0.05): absolute 1e-4 on the raw 21-vectors and the aggregate."""
    from pixel_nerf_yolo_amd.render import YoloRenderer
    g = golden("yolo_c3_unit")
    seed = int(g["seed"])
    net = make_model(pconf.yolo()["model"]).eval()
    sd = synth.mlp_state(seed * 10 + 1, d_latent=1792, d_out=21, out_gain=float(g["out_gain"]))
    net.mlp_coarse.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to(DEV)
    net.encode(torch.zeros(1, 3, 3, 128, 128), torch.from_numpy(g["src_w2c"])[None], torch.from_numpy(g["focal"])[None],
               c=torch.from_numpy(g["c"])[None], latent=torch.from_numpy(synth.latent(seed * 10 + 3, 3, 1792, 16, 16)))
    ren = make_renderer(pconf.yolo())
    assert isinstance(ren, YoloRenderer)
    par = ren.bind_parallel(net)
    n = g["rays"].shape[0]
    ren._debug_raw = torch.empty(n, 128, 21, device=DEV)
    ren.draws = dict(u_coarse=g["u_coarse"])
    with torch.no_grad():
        out = par(dt(g["rays"])[None])
    torch.cuda.synchronize()
    assert float(np.abs(g["raw_out"]).max()) < 5.0
    assert maxabs(ren._debug_raw.reshape(-1, 21), g["raw_out"]) < TOL
    assert maxabs(out, g["yolo_out"]) < TOL


def test_encoder_unit_golden(golden):
    """ResNet-34 trunk with a latent of O(1) (std 1.8, max 19): absolute 1e-4 on every latent entry."""
    g = golden("encoder_unit")
    seed, ns, H, W = int(g["seed"]), int(g["NS"]), int(g["H"]), int(g["W"])
    net = make_model(pconf.default_mv()["model"]).eval()
    sd = synth.resnet34_state(seed * 10 + 5, prefix="encoder.model.", residual_gain=float(g["residual_gain"]))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    net = net.to(DEV)
    src, _ = synth.scene_cameras(ns)
    net.encode(torch.from_numpy(synth.images(seed * 10 + 6, ns, H, W))[None], torch.from_numpy(src)[None], torch.tensor(60.0))
    lat = net.latent(0)
    assert lat.shape == g["latent"].shape and maxabs(lat, g["latent"]) < TOL


# --------------------------------------------------------------------------- C5: 8 scenes x 3 views
def test_c5_eight_scenes_vs_oracle(projection):
    """BASELINE config 5 (eval over a batch of 8 objects, 3 views each, L = 1792, 64 + 32): one encode of the
    super-batch, one render call over (SB = 8, B) rays; every scene against the oracle on its own ray subset."""
    SB, NS, H, W, B = 8, 3, 128, 128, 12
    kc, kf, kfd = 64, 32, 16
    net = make_model(c3_conf()["model"]).eval()
    load_mlp(net.mlp_coarse, 91, 1792, 4)
    load_mlp(net.mlp_fine, 92, 1792, 4)
    net = net.to(DEV)
    lat = np.concatenate([synth.latent(93 + i, NS, 1792, 16, 16) for i in range(SB)])
    poses = np.stack([synth.scene_cameras(NS, radius=1.2 + 0.05 * i)[0] for i in range(SB)])
    focal = torch.tensor([[125.0 + 2 * i, 126.0 + 2 * i] for i in range(SB)])       # per scene (SB, 2)
    c = torch.tensor([[64.0 + 0.5 * i, 64.0 - 0.5 * i] for i in range(SB)])
    net.encode(torch.zeros(SB, NS, 3, H, W), torch.from_numpy(poses), focal, c=c, latent=torch.from_numpy(lat))
    assert net.num_objs == SB and net.num_views_per_obj == NS
    rs = np.random.RandomState(17)
    rays = torch.stack([orc.gen_rays(synth.pose_spherical(100.0 + 20 * i, -20.0, 1.3)[None], W, H, 131.25, 0.8, 1.8)[0]
                        .reshape(-1, 8)[torch.from_numpy(rs.choice(H * W, B, replace=False))] for i in range(SB)])
    n = SB * B
    dr = dict(u_coarse=rs.rand(n, kc).astype(np.float32), u_fine=rs.rand(n, kf - kfd).astype(np.float32),
              u_fine2=rs.rand(n, kf - kfd).astype(np.float32), g_depth=rs.randn(n, kfd).astype(np.float32))
    ren = NeRFRenderer(n_coarse=kc, n_fine=kf, n_fine_depth=kfd, white_bkgd=True).eval()
    ren.draws = dr
    with torch.no_grad():
        out = ren(net, rays.to(DEV), want_weights=True)
    torch.cuda.synchronize()
    assert out["fine"]["rgb"].shape == (SB, B, 3) and out["fine"]["weights"].shape == (SB, B, kc + kf)
    flips = 0
    for i in range(SB):
        sc = orc.Scene(synth.mlp_state(91, d_latent=1792), synth.mlp_state(92, d_latent=1792), lat[i * NS:(i + 1) * NS],
                       poses[i], focal[i:i + 1], c[i:i + 1], W, H)
        d_i = {k: v.reshape(SB, B, -1)[i] for k, v in dr.items()}
        ref = orc.render(sc, rays[i], kc, kf, kfd, d_i["u_coarse"], d_i["u_fine"], d_i["u_fine2"], d_i["g_depth"])
        for k in ("rgb", "depth", "weights"):
            assert maxabs(out["coarse"][k][i], ref["coarse"][k]) < TOL, (i, k)
        diff = (out["fine"]["rgb"][i].cpu() - ref["fine"]["rgb"]).abs().max(dim=1)[0]
        flips += int((diff > TOL).sum())
    assert flips <= 2, flips


# --------------------------------------------------------------------------- full-size properties
def _full_frame(H, kc, kf, kfd, lat_hw, seed):
    NS, W = 3, H
    net = make_model(c3_conf()["model"]).eval()
    load_mlp(net.mlp_coarse, seed + 1, 1792, 4)
    load_mlp(net.mlp_fine, seed + 2, 1792, 4)
    net = net.to(DEV)
    src, tgt = synth.scene_cameras(NS)
    lat = torch.from_numpy(synth.latent(seed + 3, NS, 1792, *lat_hw))
    focal, c = torch.tensor(131.25 * H / 128.0), torch.tensor([[W * 0.5, H * 0.5]])
    net.encode(torch.zeros(1, NS, 3, H, W), torch.from_numpy(src)[None], focal, c=c, latent=lat)
    rays = gen_rays(dt(tgt)[None], W, H, focal, 0.8, 1.8, c=c[0]).reshape(1, -1, 8)
    ren = NeRFRenderer(n_coarse=kc, n_fine=kf, n_fine_depth=kfd, white_bkgd=True).eval()
    return net, ren, rays


def _frame_properties(net, ren, rays, kc, kf, kfd):
    n, kt = rays.shape[1], kc + kf
    dbg = {"z_fine": torch.empty(1, n, kt, device=DEV), "z_coarse": torch.empty(1, n, kc, device=DEV)}
    ren._debug_out = dbg
    ren.base_seed, ren._calls = 123, 0
    with torch.no_grad():
        a = ren(net, rays, want_weights=True)
    ren._calls = 0
    with torch.no_grad():
        b = ren(net, rays, want_weights=True)
    torch.cuda.synchronize()
    ren._debug_out = None
    # determinism in the seed: bit for bit
    assert torch.equal(a["fine"]["rgb"], b["fine"]["rgb"]) and torch.equal(a["fine"]["weights"], b["fine"]["weights"])
    assert torch.equal(a["coarse"]["depth"], b["coarse"]["depth"])
    zf, zc = dbg["z_fine"][0], dbg["z_coarse"][0]
    assert bool((zf[:, 1:] >= zf[:, :-1]).all())                                     # sortedness
    step = (1.8 - 0.8) / kc                                                          # one coarse sample per stratum
    k = torch.arange(kc, device=DEV)
    assert bool((zc >= 0.8 + k * step - 1e-5).all() and (zc <= 0.8 + (k + 1) * step + 1e-5).all())
    for part in ("coarse", "fine"):
        w = a[part]["weights"][0]
        assert bool(torch.isfinite(w).all()) and bool((w >= 0).all())
        assert float(w.sum(-1).max()) <= 1.0 + 1e-4                                  # sub-partition of unity
        rgb = a[part]["rgb"][0]
        assert float(rgb.min()) >= -1e-5 and float(rgb.max()) <= 1.0 + 1e-4          # white background
        d = a[part]["depth"][0]
        assert float(d.min()) >= 0.0 and float(d.max()) <= 1.8 + 1e-4
    return a


def _slice_invariance(net, ren, rays, lo, m, kc, kf, kfd):
    """A contiguous slice of the frame renders to the same bits whatever batch it is in (-> ray sharding is exact)."""
    ren._debug_out = None
    sub = rays[:, lo:lo + m].contiguous()
    g = torch.Generator().manual_seed(lo)
    draws = dict(u_coarse=torch.rand(m, kc, generator=g), u_fine=torch.rand(m, kf - kfd, generator=g),
                 u_fine2=torch.rand(m, kf - kfd, generator=g), g_depth=torch.randn(m, kfd, generator=g))
    ren.draws = draws
    with torch.no_grad():
        s1 = ren(net, sub)
    pad = torch.cat([rays[:, :37], sub, rays[:, -100:]], 1).contiguous()

    def padded(v, fn):
        return torch.cat([fn(37, v.shape[1]), v, fn(100, v.shape[1])], 0)

    ren.draws = {k: padded(v, torch.randn if k == "g_depth" else torch.rand) for k, v in draws.items()}
    with torch.no_grad():
        s2 = ren(net, pad)
    assert torch.equal(s1["fine"]["rgb"][0], s2["fine"]["rgb"][0, 37:37 + m])
    assert torch.equal(s1["fine"]["depth"][0], s2["fine"]["depth"][0, 37:37 + m])


def test_c3_full_frame_properties():
    """BASELINE config 3 at full size: 128 x 128 rays, L = 1792, 64 + 32 (16)."""
    net, ren, rays = _full_frame(128, 64, 32, 16, (16, 16), 200)
    assert rays.shape[1] == 16384
    _frame_properties(net, ren, rays, 64, 32, 16)
    net.set_latent_projection("on")
    _slice_invariance(net, ren, rays, 5000, 200, 64, 32, 16)


def test_c4_full_frame_properties():
    """BASELINE config 4 at full size on ONE GPU: a 400 x 400 frame (160 000 rays), L = 1792, 128 + 64 (32) --
    the whole frame the 8-rank run shards.  Also: the row-sharded render (what each rank of `bench.py --workload c4`
    does) assembles to exactly the single-call frame."""
    net, ren, rays = _full_frame(400, 128, 64, 32, (50, 50), 300)
    assert rays.shape[1] == 160000
    net.set_latent_projection("on")
    full = _frame_properties(net, ren, rays, 128, 64, 32)
    _slice_invariance(net, ren, rays, 70000, 192, 128, 64, 32)
    # sharding by contiguous ray ranges with the per-ray Philox streams keyed by the global ray index is not what the
    # renderer does (draws are per call); with explicit draws a shard reproduces its rows of the full frame bit for bit
    lo, m = 120000, 640
    g = torch.Generator().manual_seed(9)
    draws = dict(u_coarse=torch.rand(m, 128, generator=g), u_fine=torch.rand(m, 32, generator=g),
                 u_fine2=torch.rand(m, 32, generator=g), g_depth=torch.randn(m, 32, generator=g))
    ren.draws = draws
    with torch.no_grad():
        whole = ren(net, rays[:, lo:lo + m].contiguous())
    parts = []
    for r in range(4):
        sl = slice(r * m // 4, (r + 1) * m // 4)
        ren.draws = {k: v[sl] for k, v in draws.items()}
        with torch.no_grad():
            parts.append(ren(net, rays[:, lo + sl.start:lo + sl.stop].contiguous())["fine"]["rgb"][0])
    assert torch.equal(torch.cat(parts, 0), whole["fine"]["rgb"][0])
    assert bool(torch.isfinite(full["fine"]["rgb"]).all())


# --------------------------------------------------------------------------- multi-process, real render
def test_render_frame_sharded_two_ranks_one_gpu():
    """SURVEY.md 8e with the real HIP render in two processes (both on cuda:0, gloo transport): per-rank ray
    generation (pny_gen_rays_range), per-rank render, one all-gather; equals the single-call frame bit for bit."""
    import socket
    import torch.multiprocessing as mp
    import dist_worker
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=dist_worker.run_render, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1] and all(r[1] for r in res), res
    assert res[0][2] == (40, 48, 3) and res[0][3][1] == res[1][3][0]          # contiguous shards


def test_gen_rays_range_matches_full_call(golden):
    from pixel_nerf_yolo_amd.util import gen_rays_range, gen_rays_yolo
    g = golden("rays")
    poses = dt(g["poses"])
    full = gen_rays(poses, 20, 12, torch.tensor(35.5), 0.8, 1.8, c=None).reshape(-1, 8)
    for first, cnt in ((0, 480), (7, 100), (230, 20), (239, 2), (100, 380), (479, 1), (5, 0)):
        part = gen_rays_range(poses, 20, 12, torch.tensor(35.5), 0.8, 1.8, first, cnt)
        assert part.shape == (cnt, 8) and torch.equal(part, full[first:first + cnt])
    fy = gen_rays_yolo(dt(g["w2c"]), 48, 27, g["yolo_focal"], g["yolo_c"], 5.0, 10.0).reshape(-1, 8)
    part = gen_rays_range(dt(g["w2c"]), 48, 27, g["yolo_focal"], 5.0, 10.0, 1000, 777, c=g["yolo_c"], yolo=True)
    assert torch.equal(part, fy[1000:1777])
    # more images than one launch carries as kernel arguments (8): chunked launches, same bits
    many = torch.from_numpy(np.stack([synth.pose_spherical(10.0 * i, -20.0, 1.3) for i in range(19)]))
    ref = orc.gen_rays(many, 8, 6, 11.0, 0.5, 2.0)
    out = gen_rays(many.to(DEV), 8, 6, torch.tensor(11.0), 0.5, 2.0)
    assert out.shape == (19, 6, 8, 8) and maxabs(out, ref) < 1e-6
    part = gen_rays_range(many, 8, 6, torch.tensor(11.0), 0.5, 2.0, 40 * 8 + 5, 9 * 48 + 1)
    assert torch.equal(part, out.reshape(-1, 8)[325:325 + 433])
    with pytest.raises(Exception):
        gen_rays_range(poses, 20, 12, torch.tensor(35.5), 0.8, 1.8, 400, 100)   # beyond the grid


def test_eval_from_a_dataset_directory(tmp_path):
    """eval/eval.py:219-360 on files: SRNDataset item (pixel-nerf-yolo_amd/data.py) -> encode(source views) -> gen_rays(target
    poses) -> render -> PNG + PSNR / SSIM; a ray subset against the oracle fed with the same item."""
    from pixel_nerf_yolo_amd import data as pdata
    rs = np.random.RandomState(4)
    S, NV = 64, 4
    d = tmp_path / "cars_test" / "obj0"
    (d / "rgb").mkdir(parents=True)
    (d / "pose").mkdir()
    (d / "intrinsics.txt").write_text("%f %f %f 0.\n0. 0. 0.\n1.\n%d %d\n" % (65.6, S / 2, S / 2, S, S))
    for v in range(NV):
        img = np.full((S, S, 3), 255, np.uint8)
        img[12:52, 10:54] = rs.randint(0, 250, size=(40, 44, 3))
        pdata.imwrite(str(d / "rgb" / ("%06d.png" % v)), img)
        np.savetxt(str(d / "pose" / ("%06d.txt" % v)), (synth.pose_spherical(40.0 * v, -20.0, 1.3) @ np.diag([1.0, -1.0, -1.0, 1.0])).reshape(1, 16))
    item = pdata.get_split_dataset("srn", str(tmp_path / "cars"), want_split="test", training=False, image_size=(S, S))[0]
    net = make_model(pconf.default_mv()["model"]).eval()
    load_mlp(net.mlp_coarse, 301, 512, 4)
    load_mlp(net.mlp_fine, 302, 512, 4)
    esd = synth.resnet34_state(303, residual_gain=0.25)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in esd.items()}, strict=False)
    net = net.to(DEV)
    src, tgt = [0, 2], [1, 3]
    images, poses, focal, c = item["images"], item["poses"], item["focal"], item["c"]
    net.encode(images[src].unsqueeze(0), poses[src].unsqueeze(0).to(DEV), focal, c=c)
    rays = gen_rays(poses[tgt], S, S, focal, 0.8, 1.8, c=c).reshape(1, -1, 8)
    ren = NeRFRenderer(n_coarse=16, n_fine=8, n_fine_depth=4, white_bkgd=True).eval()
    par = ren.bind_parallel(net, None, simple_output=True).eval()
    n = rays.shape[1]
    draws = dict(u_coarse=rs.rand(n, 16).astype(np.float32), u_fine=rs.rand(n, 4).astype(np.float32),
                 u_fine2=rs.rand(n, 4).astype(np.float32), g_depth=rs.randn(n, 4).astype(np.float32))
    ren.draws = draws
    with torch.no_grad():
        rgb, depth = par(rays)
    frames = rgb[0].reshape(2, S, S, 3).cpu().numpy()
    gt = (images[tgt] * 0.5 + 0.5).permute(0, 2, 3, 1).numpy()
    p, s_ = pdata.write_views(str(tmp_path / "out"), frames, tgt, gt=gt)
    assert np.isfinite(p) and 0.0 <= s_ <= 1.0 and sorted(os.listdir(tmp_path / "out")) == ["000001.png", "000003.png"]
    # the same item through the oracle (its own trunk) on a ray subset
    lat, _ = orc.spatial_encoder(esd, images[src].numpy())
    sc = orc.Scene(synth.mlp_state(301), synth.mlp_state(302), lat, poses[src].numpy(), focal, c[None], S, S)
    sub = torch.from_numpy(rs.choice(n, 96, replace=False))
    ref = orc.render(sc, rays[0].cpu()[sub], 16, 8, 4, draws["u_coarse"][sub], draws["u_fine"][sub], draws["u_fine2"][sub],
                     draws["g_depth"][sub])
    diff = (rgb[0].cpu()[sub] - ref["fine"]["rgb"]).abs().max(dim=1)[0]
    assert int((diff > TOL).sum()) <= 2 and float(diff.median()) < 1e-5


@pytest.mark.parametrize("d_lat", [512, 1792])
def test_plain_c_host_through_the_abi(d_lat, tmp_path):
    """The boundary from a host that is not Python: examples/abi_host.c (C99, HIP's C API for its own device memory, default
    stream, no torch in the process) is compiled with gcc against include/pnyolo.h + libpnyolo.so, fed one file with the
    weights by state_dict name, a latent, cameras, rays and the renderer's draws, and its coarse / fine rgb and depth are
    compared with the Python mirror's on the same inputs: bit for bit."""
    import shutil
    import struct
    import subprocess
    from pixel_nerf_yolo_amd import lib as plib
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("gcc or the HIP headers are not installed")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe, libdir = str(tmp_path / "abi_host"), os.path.dirname(plib.LIB_PATH)
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(root, "include"),
                         "-I", "/opt/rocm/include", os.path.join(root, "examples", "abi_host.c"), "-o", exe, "-L", libdir, "-lpnyolo",
                         "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    ns, H, W, hw, n, kc, kf, kfd = 3, 64, 64, 16, 200, 32, 16, 8
    rs = np.random.RandomState(d_lat)
    sd = {}
    sd.update({"mlp_coarse." + k: v for k, v in synth.mlp_state(2101, d_latent=d_lat).items()})
    sd.update({"mlp_fine." + k: v for k, v in synth.mlp_state(2102, d_latent=d_lat).items()})
    lat = synth.latent(2103, ns, d_lat, hw, hw)
    poses, tgt = synth.scene_cameras(ns)
    focal, cc_ = np.array([0.9 * W, 0.9 * W], np.float32), np.array([W * 0.5, H * 0.5], np.float32)
    rays = gen_rays(torch.from_numpy(tgt)[None], W, H, torch.tensor(0.9 * W), 0.8, 1.8).reshape(-1, 8)[torch.from_numpy(rs.choice(H * W, n, replace=False))]
    dr = dict(u_coarse=rs.rand(n, kc).astype(np.float32), u_fine=rs.rand(n, kf - kfd).astype(np.float32),
              u_fine2=rs.rand(n, kf - kfd).astype(np.float32), g_depth=rs.randn(n, kfd).astype(np.float32))
    # ---- the Python mirror
    conf = pconf.default_mv()
    if d_lat != 512:
        conf.d["model"]["encoder"]["backbone"] = "custom"
    net = make_model(conf["model"]).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    net = net.to(DEV)
    net.encode(torch.zeros(1, ns, 3, H, W), torch.from_numpy(poses)[None], torch.tensor(0.9 * W), c=torch.from_numpy(cc_)[None],
               latent=torch.from_numpy(lat))
    ren = NeRFRenderer(n_coarse=kc, n_fine=kf, n_fine_depth=kfd, depth_std=0.01, white_bkgd=True).eval()
    ren.draws = dr
    with torch.no_grad():
        out = ren(net, rays[None].to(DEV))
    torch.cuda.synchronize()
    # ---- the same through the C program
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("<i", 0x504E5931))
        f.write(struct.pack("<6if4i", d_lat, 512, 4, 5, 3, 6, 1.5, 0, 1, 0, 1))
        f.write(struct.pack("<i", len(sd)))
        for k, v in sd.items():
            kb = k.encode()
            f.write(struct.pack("<i", len(kb)) + kb + struct.pack("<i", v.ndim) + struct.pack("<%dq" % v.ndim, *v.shape))
            f.write(np.ascontiguousarray(v, np.float32).tobytes())
        f.write(struct.pack("<6i", ns, d_lat, hw, hw, W, H))
        f.write(np.ascontiguousarray(lat, np.float32).tobytes())
        f.write(np.ascontiguousarray(poses, np.float32).tobytes())
        f.write(focal.tobytes() + cc_.tobytes())
        f.write(struct.pack("<5if", n, kc, kf, kfd, 1, 0.01))
        f.write(np.ascontiguousarray(rays.cpu().numpy(), np.float32).tobytes())
        for k in ("u_coarse", "u_fine", "u_fine2", "g_depth"):
            f.write(dr[k].tobytes())
    env = dict(os.environ)
    run = subprocess.run([exe, inp, outp], capture_output=True, text=True, env=env)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    got = np.fromfile(outp, np.float32)
    assert got.size == 8 * n
    parts = dict(rgb_c=got[:3 * n].reshape(n, 3), depth_c=got[3 * n:4 * n], rgb_f=got[4 * n:7 * n].reshape(n, 3), depth_f=got[7 * n:])
    ref = dict(rgb_c=out["coarse"]["rgb"][0], depth_c=out["coarse"]["depth"][0], rgb_f=out["fine"]["rgb"][0], depth_f=out["fine"]["depth"][0])
    for k in parts:
        assert np.array_equal(parts[k], ref[k].cpu().numpy()), (k, float(np.abs(parts[k] - ref[k].cpu().numpy()).max()))
