"""
Pins the oracle (oracle/pnyolo_oracle.py) to outputs of the reference itself, captured by
tools/make_golden.py (the reference holds no golden vectors of its own).
CPU only; no GPU, no reference import.
"""
import numpy as np
import torch

import pnyolo_oracle as orc
from pixel_nerf_yolo_amd import synth


def t(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32)


def maxabs(a, b):
    return float((t(a) - t(b)).abs().max())


def nerf_scene(g, seed):
    ns, H, W = int(g["NS"]), int(g["H"]), int(g["W"])
    L = int(g["d_latent"]) if "d_latent" in g else 512
    hl, wl = (int(g["Hl"]), int(g["Wl"])) if "Hl" in g else (H // 2, W // 2)
    mc = synth.mlp_state(seed * 10 + 1, d_latent=L)
    mf = synth.mlp_state(seed * 10 + 2, d_latent=L) if int(g["Kf"]) > 0 else None
    lat = synth.latent(seed * 10 + 3, ns, L, hl, wl)
    return orc.Scene(mc, mf, lat, g["src_poses"], g["focal"], g["c"][None], W, H)


def test_gen_rays(golden):
    g = golden("rays")
    r1 = orc.gen_rays(g["poses"], 20, 12, 35.5, 0.8, 1.8, c=None)
    r2 = orc.gen_rays(g["poses"][:1], 16, 16, [30.0, 28.0], 0.5, 2.5, c=[7.5, 9.25])
    assert r1.shape == g["r1"].shape and maxabs(r1, g["r1"]) < 2e-7
    assert maxabs(r2, g["r2"]) < 2e-7


def test_gen_rays_yolo(golden):
    g = golden("rays")
    r3 = orc.gen_rays_yolo(g["w2c"], 48, 27, g["yolo_focal"], g["yolo_c"], 5.0, 10.0)
    assert r3.shape == g["r3"].shape and maxabs(r3, g["r3"]) < 1e-6


def test_encode_cameras(golden):
    for name in ("nerf_c1", "nerf_c2"):
        g = golden(name)
        w2c, focal, c = orc.encode_cameras(g["src_poses"], g["focal"], g["c"][None], int(g["W"]), int(g["H"]))
        assert maxabs(w2c, g["enc_poses"]) == 0.0
        assert maxabs(focal, g["enc_focal"]) == 0.0 and maxabs(c, g["enc_c"]) == 0.0
    g = golden("yolo_c3")
    w2c, focal, c = orc.encode_cameras(g["src_w2c"], g["focal"][None], g["c"][None], 128, 128, yolo=True)
    assert maxabs(w2c, g["enc_poses"]) == 0.0 and maxabs(focal, g["enc_focal"]) == 0.0


def test_sample_coarse_bitwise(golden):
    for name in ("nerf_c1", "nerf_c2"):
        g = golden(name)
        z = orc.sample_coarse(t(g["rays"]), int(g["Kc"]), g["u_coarse"])
        assert maxabs(z, g["z_coarse"]) == 0.0


def test_query_probe(golden):
    for name, seed in (("nerf_c1", 1), ("nerf_c2", 7)):
        g = golden(name)
        sc = nerf_scene(g, seed)
        out = orc.query(sc, g["probe_xyz"], g["probe_viewdirs"], coarse=True)
        assert maxabs(out[:, :3], g["probe_out_coarse"][:, :3]) < 2e-6
        assert maxabs(out[:, 3], g["probe_out_coarse"][:, 3]) < 2e-5
        if "probe_out_fine" in g:
            out = orc.query(sc, g["probe_xyz"], g["probe_viewdirs"], coarse=False)
            assert maxabs(out, g["probe_out_fine"]) < 2e-5


def test_render_c1(golden):
    g = golden("nerf_c1")
    sc = nerf_scene(g, 1)
    r = orc.render(sc, g["rays"], 32, 0, 0, g["u_coarse"], chunk=1000)
    assert maxabs(r["coarse"]["out"].reshape(-1, 4), g["coarse_out"]) < 2e-5
    assert maxabs(r["coarse"]["weights"], g["coarse_weights"]) < 2e-6
    assert maxabs(r["coarse"]["rgb"], g["coarse_rgb"]) < 2e-6
    assert maxabs(r["coarse"]["depth"], g["coarse_depth"]) < 2e-6
    assert "fine" not in r


def test_render_c2(golden):
    g = golden("nerf_c2")
    sc = nerf_scene(g, 7)
    r = orc.render(sc, g["rays"], 64, 32, 16, g["u_coarse"], g["u_fine"], g["u_fine2"], g["g_depth"], chunk=3000)
    assert maxabs(r["coarse"]["out"].reshape(-1, 4), g["coarse_out"]) < 2e-5
    assert maxabs(r["coarse"]["rgb"], g["coarse_rgb"]) < 2e-6
    assert maxabs(r["fine"]["out"].reshape(-1, 4), g["fine_out"]) < 5e-5
    assert maxabs(r["fine"]["weights"], g["fine_weights"]) < 5e-6
    assert maxabs(r["fine"]["rgb"], g["fine_rgb"]) < 5e-6
    assert maxabs(r["fine"]["depth"], g["fine_depth"]) < 5e-6


def test_render_c3_c4(golden):
    """BASELINE configs 3 and 4 at reduced ray counts: L = 1792 conditioning (backbone = custom in the reference's
    config, latent supplied), NeRF renderer with 64 + 32 (16) and 128 + 64 (32) samples, 400 x 400 geometry for C4."""
    for name, seed, (kc, kf, kfd) in (("nerf_c3", 31, (64, 32, 16)), ("nerf_c4", 37, (128, 64, 32))):
        g = golden(name)
        assert int(g["d_latent"]) == 1792 and (int(g["Kc"]), int(g["Kf"]), int(g["Kfd"])) == (kc, kf, kfd)
        sc = nerf_scene(g, seed)
        r = orc.render(sc, g["rays"], kc, kf, kfd, g["u_coarse"], g["u_fine"], g["u_fine2"], g["g_depth"], chunk=5000)
        assert maxabs(r["coarse"]["z"], g["z_coarse"]) == 0.0
        assert maxabs(r["coarse"]["out"].reshape(-1, 4), g["coarse_out"]) < 2e-5, name
        assert maxabs(r["coarse"]["rgb"], g["coarse_rgb"]) < 2e-6, name
        assert maxabs(r["coarse"]["weights"], g["coarse_weights"]) < 2e-6, name
        assert maxabs(r["fine"]["out"].reshape(-1, 4), g["fine_out"]) < 5e-5, name
        assert maxabs(r["fine"]["weights"], g["fine_weights"]) < 5e-6, name
        assert maxabs(r["fine"]["rgb"], g["fine_rgb"]) < 5e-6, name
        assert maxabs(r["fine"]["depth"], g["fine_depth"]) < 5e-6, name
        out = orc.query(sc, g["probe_xyz"], g["probe_viewdirs"], coarse=False)
        assert maxabs(out, g["probe_out_fine"]) < 2e-5, name


def test_encoder_to_render(golden):
    """tests/golden/enc_render.npz: the reference's SpatialEncoder.forward -> encode -> NeRFRenderer.forward with
    nothing bypassed.  The oracle's trunk + render against it: latent, per-sample rgb / sigma and pixels."""
    g = golden("enc_render")
    seed, ns, H, W = int(g["seed"]), int(g["NS"]), int(g["H"]), int(g["W"])
    esd = synth.resnet34_state(seed * 10 + 4, residual_gain=float(g["residual_gain"]))
    lat, _ = orc.spatial_encoder(esd, synth.images(seed * 10 + 5, ns, H, W))
    assert tuple(lat.shape) == tuple(int(v) for v in g["latent_shape"])
    assert maxabs(lat.reshape(-1)[torch.from_numpy(g["latent_idx"])], g["latent_val"]) < 2e-5
    sc = orc.Scene(synth.mlp_state(seed * 10 + 1), synth.mlp_state(seed * 10 + 2), lat, g["src_poses"], g["focal"],
                   g["c"][None], W, H)
    r = orc.render(sc, g["rays"], 64, 32, 16, g["u_coarse"], g["u_fine"], g["u_fine2"], g["g_depth"], chunk=3000)
    assert maxabs(r["coarse"]["out"].reshape(-1, 4), g["coarse_out"]) < 5e-5
    assert maxabs(r["coarse"]["rgb"], g["coarse_rgb"]) < 5e-6
    assert maxabs(r["fine"]["out"].reshape(-1, 4), g["fine_out"]) < 1e-4
    assert maxabs(r["fine"]["rgb"], g["fine_rgb"]) < 1e-5
    assert maxabs(r["fine"]["depth"], g["fine_depth"]) < 1e-5


def test_stages_from_golden_inputs(golden):
    """Each stage fed the reference's own intermediate values -> bitwise / 1-ulp agreement."""
    g = golden("nerf_c2")
    rays = t(g["rays"])
    w, rgb, depth = orc.composite(rays, t(g["z_coarse"]), t(g["coarse_out"]).reshape(100, 64, 4), True)
    assert maxabs(w, g["coarse_weights"]) == 0.0 and maxabs(rgb, g["coarse_rgb"]) == 0.0
    assert maxabs(depth, g["coarse_depth"]) == 0.0
    zf = orc.sample_fine(rays, t(g["coarse_weights"]), g["u_fine"], g["u_fine2"], 64)
    zd = orc.sample_fine_depth(rays, t(g["coarse_depth"]), g["g_depth"], 0.01)
    zs, _ = torch.sort(torch.cat([t(g["z_coarse"]), zf, zd], -1), -1)
    w2, rgb2, d2 = orc.composite(rays, zs, t(g["fine_out"]).reshape(100, 96, 4), True)
    assert maxabs(w2, g["fine_weights"]) == 0.0 and maxabs(rgb2, g["fine_rgb"]) == 0.0


def test_yolo_render(golden):
    g = golden("yolo_c3")
    mc = synth.mlp_state(31, d_latent=1792, d_out=21)
    lat = synth.latent(33, 3, 1792, 16, 16)
    sc = orc.Scene(mc, None, lat, g["src_w2c"], g["focal"][None], g["c"][None], 128, 128, yolo=True)
    rays_all = orc.gen_rays_yolo(g["tgt_w2c"][None], int(g["Wc"]), int(g["Hc"]), g["focal"] / 8, g["c"] / 8, 1.0, 13.0)
    assert maxabs(rays_all[0], g["rays_all"]) < 1e-5
    r = orc.yolo_render(sc, g["rays"], 128, g["u_coarse"], chunk=128)
    scale = float(np.abs(g["raw_out"]).max())
    assert maxabs(r["raw"].reshape(-1, 21), g["raw_out"]) < 2e-5 * max(1.0, scale)
    assert maxabs(r["out"], g["yolo_out"]) < 5e-5
    assert maxabs(orc.yolo_aggregate(t(g["raw_out"]).reshape(42, 128, 21)), g["yolo_out"]) < 1e-6


def test_yolo_render_unit_magnitude(golden):
    """Companion of yolo_c3 with lin_out scaled by 0.05: raw outputs are O(1), so the comparison is absolute."""
    g = golden("yolo_c3_unit")
    seed = int(g["seed"])
    mc = synth.mlp_state(seed * 10 + 1, d_latent=1792, d_out=21, out_gain=float(g["out_gain"]))
    lat = synth.latent(seed * 10 + 3, 3, 1792, 16, 16)
    sc = orc.Scene(mc, None, lat, g["src_w2c"], g["focal"][None], g["c"][None], 128, 128, yolo=True)
    r = orc.yolo_render(sc, g["rays"], 128, g["u_coarse"], chunk=128)
    assert float(np.abs(g["raw_out"]).max()) < 5.0
    assert maxabs(r["raw"].reshape(-1, 21), g["raw_out"]) < 1e-5
    assert maxabs(r["out"], g["yolo_out"]) < 1e-5


def test_encoder_unit_magnitude(golden):
    """Companion of `encoder` on the well-conditioned trunk (residual_gain = 0.25): absolute comparison."""
    g = golden("encoder_unit")
    seed = int(g["seed"])
    sd = synth.resnet34_state(seed * 10 + 5, prefix="encoder.model.", residual_gain=float(g["residual_gain"]))
    lat, _ = orc.spatial_encoder(sd, synth.images(seed * 10 + 6, int(g["NS"]), int(g["H"]), int(g["W"])))
    assert maxabs(lat, g["latent"]) < 2e-5


def test_encoder(golden):
    g = golden("encoder")
    sd = synth.resnet34_state(45, prefix="encoder.model.")
    img = synth.images(46, int(g["NS"]), int(g["H"]), int(g["W"]))
    lat, levels = orc.spatial_encoder(sd, img)
    assert lat.shape == g["latent"].shape
    # the reference upsamples its level list in place (encoder.py:162-168), so the captured
    # levels are the upsampled ones = channel slices of the latent
    assert maxabs(lat[:, 64:72], g["level1"]) < 1e-4
    assert maxabs(lat[:, 256:264], g["level3"]) < 1e-4
    assert levels[3].shape[-2:] == (4, 3)
    assert maxabs(lat, g["latent"]) < 1e-4
    samp = orc.index_latent(t(g["latent"]), t(g["uv"]), int(g["W"]), int(g["H"]))
    scale = float(np.abs(g["index_out"]).max())  # random-weight trunk: activations reach O(100)
    assert maxabs(samp.permute(0, 2, 1), g["index_out"]) < 1e-6 * scale


def test_yolo_detection_tail(golden):
    """convert_cells_to_bboxes / nms / calculate_tp_fp_fn of the reference (util.py:633-802)."""
    g = golden("yolo_tail")
    h, w, A = (int(v) for v in g["hw"])
    for c in range(3):
        pb = orc.cells_to_bboxes(g["c%d_pred" % c][0], g["anchors"], h, w, True)
        tb = orc.cells_to_bboxes(g["c%d_tgt" % c][0], g["anchors"], h, w, False)
        assert maxabs(pb, g["c%d_p_boxes" % c]) == 0.0 and maxabs(tb, g["c%d_t_boxes" % c]) == 0.0
        for k in range(2):
            iou_t, conf_t, hc, above = g["c%d_nms%d_meta" % (c, k)]
            kept, hi, ab = orc.nms(t(g["c%d_p_boxes" % c]), iou_t, conf_t)
            ref = g["c%d_nms%d_kept" % (c, k)]
            assert len(kept) == ref.shape[0] and ab == int(above) and abs(hi - hc) < 1e-12
            if len(kept):
                assert np.array_equal(np.array(kept, dtype=np.float32), ref.astype(np.float32))
            assert orc.tp_fp_fn(t(g["c%d_t_boxes" % c]), t(g["c%d_p_boxes" % c]), iou_t, conf_t, 0.2) == \
                tuple(int(v) for v in g["c%d_tpfpfn%d" % (c, k)])


def test_nms_duplicate_rows(golden):
    """list.remove() semantics (util.py:719): the first EQUAL row is deleted, not the row at hand."""
    g = golden("yolo_tail")
    for k in range(2):
        iou_t, conf_t, hc, above = g["dup_nms%d_meta" % k]
        kept, hi, ab = orc.nms(t(g["dup_boxes"]), iou_t, conf_t)
        assert ab == int(above) and np.array_equal(np.array(kept, dtype=np.float32), g["dup_nms%d_kept" % k].astype(np.float32))
    assert g["dup_nms0_kept"][1, 2] > 0.6      # B precedes the surviving twin: positional deletion would swap them


def test_encoder_without_first_pool(golden):
    """conf/exp/sn64.conf of the reference: encoder.use_first_pool = False (encoder.py:145-146)."""
    g = golden("encoder_nopool")
    sd = synth.resnet34_state(55, prefix="encoder.model.")
    img = synth.images(56, int(g["NS"]), int(g["H"]), int(g["W"]))
    lat, levels = orc.spatial_encoder(sd, img, use_first_pool=False)
    assert lat.shape == g["latent"].shape and levels[1].shape[-2:] == levels[0].shape[-2:]
    assert maxabs(lat, g["latent"]) < 1e-6 * float(np.abs(g["latent"]).max())


def _variant_scene(g, which, scene=0):
    """Scenes of tests/golden/nerf_variants.npz (tools/make_golden.py fixture_nerf_variants)."""
    from pixel_nerf_yolo_amd import synth
    seed, ns, H, W = int(g["seed"]), int(g["NS"]), int(g["H"]), int(g["W"])
    mc, mf = synth.mlp_state(seed * 10 + 1), synth.mlp_state(seed * 10 + 2)
    if which == "ab":
        lat = synth.latent(int(g["ab_latent_seed"]), ns, 512, H // 2, W // 2)
        return orc.Scene(mc, mf, lat, g["ab_poses"][0], g["ab_focal"], g["ab_c"], W, H)
    lat = synth.latent(int(g["c_latent_seed"]) + scene, ns, 512, H // 2, W // 2)
    return orc.Scene(mc, mf, lat, g["c_poses"][scene], g["c_focal"][scene:scene + 1], g["c_c"][scene:scene + 1], W, H)


def test_render_variants(golden):
    """lindisp + black background + importance-only, depth-only fine pass, and a 2-scene super-batch with
    per-scene intrinsics: the oracle against the reference's own outputs on the recorded draws."""
    g = golden("nerf_variants")
    sc = _variant_scene(g, "ab")
    r = orc.render(sc, g["a_rays"][0], 16, 8, 0, g["a_draw0_rand_like"], g["a_draw1_rand"], g["a_draw2_rand_like"], None,
                   white_bkgd=False, lindisp=True)
    for part in ("coarse", "fine"):
        for k, tol in (("rgb", 5e-6), ("depth", 5e-6), ("weights", 5e-6)):
            assert maxabs(r[part][k], g["a_%s_%s" % (part, k)][0]) < tol, (part, k)
    r = orc.render(sc, g["b_rays"][0], 16, 8, 8, g["b_draw0_rand_like"], None, None, g["b_draw1_randn_like"],
                   white_bkgd=True, lindisp=False)
    for part in ("coarse", "fine"):
        for k in ("rgb", "depth", "weights"):
            assert maxabs(r[part][k], g["b_%s_%s" % (part, k)][0]) < 5e-6, (part, k)
    n = g["c_rays"].shape[1]
    for i in range(2):                                   # the reference flattens (SB, B) scene-major
        sl = slice(i * n, (i + 1) * n)
        r = orc.render(_variant_scene(g, "c", i), g["c_rays"][i], 16, 8, 4, g["c_draw0_rand_like"][sl],
                       g["c_draw1_rand"][sl], g["c_draw2_rand_like"][sl], g["c_draw3_randn_like"][sl])
        for part in ("coarse", "fine"):
            for k in ("rgb", "depth", "weights"):
                assert maxabs(r[part][k], g["c_%s_%s" % (part, k)][i]) < 5e-6, (i, part, k)


def test_yolo_latent_culling(golden):
    """models.py:222-224,254-264: latent zeroed where z_cam >= 0 and where it is NaN (0/0, inf projections)."""
    from pixel_nerf_yolo_amd import synth
    g = golden("yolo_cull")
    seed, ns = int(g["seed"]), int(g["NS"])
    lat = synth.latent(seed * 10 + 3, ns, 1792, int(g["Hl"]), int(g["Wl"]))
    sc = orc.Scene(synth.mlp_state(seed * 10 + 1, d_latent=1792, d_out=21), None, lat, g["w2c"], g["focal"][None],
                   g["c"][None], int(g["W"]), int(g["H"]), yolo=True)
    out = orc.query(sc, g["xyz"], g["viewdirs"], coarse=True)
    assert bool(torch.isfinite(out).all())
    scale = float(np.abs(g["out"]).max())
    assert maxabs(out, g["out"]) < 2e-6 * max(1.0, scale)


def test_mlp_shapes(golden):
    """combine_layer = 0 / 1 / none with 2 / 4 / 1 blocks: the oracle against the reference's model probes."""
    from pixel_nerf_yolo_amd import synth
    g = golden("mlp_shapes")
    seed, H, W = int(g["seed"]), int(g["H"]), int(g["W"])
    for tag in "abc":
        nb, cl, ns = (int(v) for v in g[tag + "_cfg"])
        lat = synth.latent(seed * 10 + 3, ns, 512, H // 2, W // 2)
        sc = orc.Scene(synth.mlp_state(seed * 10 + ord(tag), n_blocks=nb, combine_layer=cl), None, lat, g[tag + "_poses"],
                       np.float32(33.0), None, W, H, n_blocks=nb, combine_layer=cl)
        out = orc.query(sc, g["xyz"], g["viewdirs"], coarse=True)
        assert maxabs(out[:, :3], g[tag + "_out"][:, :3]) < 2e-6, tag
        assert maxabs(out[:, 3], g[tag + "_out"][:, 3]) < 2e-5, tag


def test_training_gradients(golden):
    """tests/golden/nerf_grads.npz: gradients of the reference's training loss (MSE on coarse.rgb + MSE on fine.rgb,
    PixelNerfTrainer.py:133-156) w.r.t. every MLP parameter and the latent.  The oracle is written in torch ops, so
    autograd through it must reproduce them: this pins the checker the backward pass (docs/backward_plan.md) will be
    held to.  Goldens store a digest per tensor (sum, sum |.|, max |.|, 192 seeded entries)."""
    from pixel_nerf_yolo_amd import synth
    g = golden("nerf_grads")
    seed, ns, H, W = int(g["seed"]), int(g["NS"]), int(g["H"]), int(g["W"])
    kc, kf, kfd = int(g["Kc"]), int(g["Kf"]), int(g["Kfd"])
    mc = {k: torch.from_numpy(v).requires_grad_() for k, v in synth.mlp_state(seed * 10 + 1).items()}
    mf = {k: torch.from_numpy(v).requires_grad_() for k, v in synth.mlp_state(seed * 10 + 2).items()}
    lat = torch.from_numpy(synth.latent(seed * 10 + 3, ns, 512, H // 2, W // 2)).requires_grad_()
    sc = orc.Scene(mc, mf, lat, g["poses"], g["focal"], g["c"], W, H)
    r = orc.render(sc, g["rays"], kc, kf, kfd, g["draw0_rand_like"], g["draw1_rand"], g["draw2_rand_like"],
                   g["draw3_randn_like"], white_bkgd=True)
    assert maxabs(r["coarse"]["rgb"].detach(), g["coarse_rgb"]) < 2e-6 and maxabs(r["fine"]["rgb"].detach(), g["fine_rgb"]) < 5e-6
    gt = torch.from_numpy(g["gt"])
    loss = torch.nn.functional.mse_loss(r["coarse"]["rgb"], gt) + torch.nn.functional.mse_loss(r["fine"]["rgb"], gt)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-6
    loss.backward()
    grads = {"mlp_coarse." + k: v.grad for k, v in mc.items()}
    grads.update({"mlp_fine." + k: v.grad for k, v in mf.items()})
    grads["latent"] = lat.grad
    assert sorted(grads) == sorted(g["grad_names"].tolist())
    for name, gr in grads.items():
        assert gr is not None, name
        stat, idx, val = g["g:%s:stat" % name], g["g:%s:idx" % name], g["g:%s:val" % name]
        f = gr.reshape(-1).double()
        scale = max(float(stat[2]), 1e-12)                               # the tensor's max |gradient|
        assert float((f[torch.from_numpy(idx)] - torch.from_numpy(val)).abs().max()) < 2e-4 * scale, name
        assert abs(float(f.abs().sum()) - float(stat[1])) < 2e-4 * float(stat[1]) + 1e-12, name
        assert abs(float(f.abs().max()) - float(stat[2])) < 2e-4 * scale, name


def test_package_torch_trunk_matches_oracle_trunk():
    """SpatialEncoder.forward_torch (the differentiable trunk encode() uses when the encoder trains) against the oracle's
    restatement on the same weights, eval-mode batch norm, CPU; and with use_first_pool = False (sn64.conf)."""
    import pnyolo_pkg
    pnyolo_pkg.load()
    from pixel_nerf_yolo_amd import conf as pconf, synth
    from pixel_nerf_yolo_amd.model import make_model
    enc = synth.resnet34_state(31, residual_gain=0.25)
    images = torch.from_numpy(synth.images(32, 2, 48, 64))
    for first_pool in (True, False):
        c = pconf.default_mv()
        c.d["model"]["encoder"]["use_first_pool"] = first_pool
        net = make_model(c["model"]).eval()
        net.load_state_dict({k: torch.from_numpy(v) for k, v in enc.items()}, strict=False)
        with torch.no_grad():
            got = net.encoder.forward_torch(images)
        ref = orc.spatial_encoder(enc, images, use_first_pool=first_pool)[0]
        assert got.shape == ref.shape == (2, 512, 24, 32)
        assert float((got - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
