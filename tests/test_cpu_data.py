"""
Data / image adapters (pixel-nerf-yolo_amd/data.py) on synthetic directory trees in the layout the
reference's readers expect (src/data/SRNDataset.py, src/data/YOLODataset.py).  CPU only.  PARITY UNPINNED against
imageio / cv2 / skimage (absent here): what is pinned is the tensor contract -- shapes, value ranges, coordinate
conventions, target-grid assignment -- and the metrics against their published definitions evaluated independently.
"""
import os

import numpy as np
import torch

from pixel_nerf_yolo_amd import data as pdata
from pixel_nerf_yolo_amd import synth


def _srn_tree(root, stage, n_obj=2, n_views=3, size=16):
    rs = np.random.RandomState(0)
    base = os.path.join(root, "cars_" + stage)
    truth = []
    for o in range(n_obj):
        d = os.path.join(base, "obj%02d" % o)
        os.makedirs(os.path.join(d, "rgb"))
        os.makedirs(os.path.join(d, "pose"))
        with open(os.path.join(d, "intrinsics.txt"), "w") as fh:
            fh.write("%f %f %f 0.\n0. 0. 0.\n1.\n%d %d\n" % (20.0 + o, size / 2, size / 2, size, size))
        imgs, poses = [], []
        for v in range(n_views):
            img = np.full((size, size, 4), 255, np.uint8)                  # RGBA on white; the object is a rectangle
            r0, r1, c0, c1 = 3 + v, 10 + v, 2 + o, 12
            img[r0:r1 + 1, c0:c1 + 1, :3] = rs.randint(0, 255, size=(r1 - r0 + 1, c1 - c0 + 1, 3))
            pdata.imwrite(os.path.join(d, "rgb", "%06d.png" % v), img)
            pose = synth.pose_spherical(40.0 * v, -20.0, 1.3)
            np.savetxt(os.path.join(d, "pose", "%06d.txt" % v), pose.reshape(1, 16))
            imgs.append(img[..., :3])
            poses.append(pose)
        truth.append((d, np.stack(imgs), np.stack(poses)))
    return os.path.join(root, "cars"), truth


def test_srn_dataset_contract(tmp_path):
    path, truth = _srn_tree(str(tmp_path), "val")
    ds = pdata.get_split_dataset("srn", path, want_split="val", training=False, image_size=(16, 16))
    assert len(ds) == 2 and ds.z_near == 0.8 and ds.z_far == 1.8 and ds.lindisp is False
    it = ds[1]
    d, imgs, poses = truth[1]
    assert it["path"] == d and it["images"].shape == (3, 3, 16, 16) and it["masks"].shape == (3, 1, 16, 16)
    assert torch.equal(it["images"], torch.from_numpy(imgs).permute(0, 3, 1, 2).float() / 255 * 2 - 1)   # [-1, 1]
    assert float(it["focal"]) == 21.0 and it["c"].tolist() == [8.0, 8.0]
    # poses: cam->world with the y / z axes flipped (SRNDataset.py:47-49,87)
    flip = np.diag([1.0, -1.0, -1.0, 1.0]).astype(np.float32)
    assert np.allclose(it["poses"].numpy(), poses @ flip, atol=1e-6)
    # bbox = [cmin, rmin, cmax, rmax] of the non-white pixels; mask = 1 inside
    assert it["bbox"][2].tolist() == [3.0, 5.0, 12.0, 12.0]
    assert float(it["masks"][2, 0, 5:13, 3:13].min()) == 1.0 and float(it["masks"][2].sum()) <= 8 * 10
    # resize path: intrinsics and boxes scale with the image (area interpolation)
    ds8 = pdata.SRNDataset(path, stage="val", image_size=(8, 8))
    it8 = ds8[1]
    assert it8["images"].shape == (3, 3, 8, 8) and float(it8["focal"]) == 10.5 and it8["c"].tolist() == [4.0, 4.0]
    assert torch.allclose(it8["bbox"], it["bbox"] * 0.5)
    assert torch.allclose(it8["images"], torch.nn.functional.avg_pool2d(it["images"], 2), atol=1e-6)


class _Conf(dict):
    pass


def test_yolo_dataset_contract(tmp_path):
    root = str(tmp_path)
    rs = np.random.RandomState(1)
    d = os.path.join(root, "scene0")
    os.makedirs(d)
    H, W = 64, 120
    K = np.array([[100.0, 0, 60.0], [0, 100.0, 32.0], [0, 0, 1]])
    for v in range(2):
        pdata.imwrite(os.path.join(d, "image_%04d.png" % v), rs.randint(0, 255, size=(H, W, 3)).astype(np.uint8))
        E = np.eye(4)
        E[:3, 3] = [0.1 * v, 0.2, 3.0]
        np.save(os.path.join(d, "extrinsic_%04d.npy" % v), E)
        with open(os.path.join(d, "projected_bboxes_%04d.txt" % v), "w") as fh:
            fh.write("1 0.30 0.40 0.20 0.30\n0 0.80 0.75 0.10 0.12\n")
    np.save(os.path.join(d, "intrinsic_0000.npy"), K)
    open(os.path.join(root, "test.lst"), "w").write("scene0\n")
    conf = _Conf({"yolo.image_scale": [0.5, 0.5], "model.mlp_coarse.num_scales": 1, "model.mlp_coarse.num_anchors_per_scale": 3,
                  "yolo.cell_sizes": [4], "yolo.anchors": [[(0.28, 0.22), (0.38, 0.48), (0.9, 0.78)]], "yolo.ignore_iou_thresh": 0.5})
    ds = pdata.get_split_dataset("yolo", root, want_split="test", training=False, conf=conf)
    assert len(ds) == 1 and ds.z_near == 1 and ds.z_far == 13.0
    it = ds[0]
    assert it["images"].shape == (2, 3, 32, 60) and float(it["images"].min()) >= -1 and float(it["images"].max()) <= 1
    assert it["focal"].tolist() == [50.0, 50.0] and it["c"].tolist() == [30.0, 16.0]
    # extrinsics: first row negated (YOLODataset.py:112), otherwise as stored
    assert it["poses"].shape == (2, 4, 4)
    assert torch.allclose(it["poses"][1], torch.tensor([[-1.0, 0.0, 0.0, -0.1], [0.0, 1.0, 0.0, 0.2], [0.0, 0.0, 1.0, 3.0],
                                                        [0.0, 0.0, 0.0, 1.0]]))
    tgt = it["bboxes"][0][0]                                     # view 0, scale 0: (8, 15, 3, 6)
    assert tgt.shape == (32 // 4, 60 // 4, 3, 6)
    i, j = int(8 * 0.40), int(15 * 0.30)
    cell = tgt[i, j]
    a = int((cell[:, 0] == 1).nonzero()[0])                      # the best-IoU anchor got the box
    ious = pdata.iou_wh(torch.tensor([0.20, 0.30]), ds.anchors)
    assert a == int(ious.argmax())
    assert torch.allclose(cell[a], torch.tensor([1.0, 15 * 0.30 - j, 8 * 0.40 - i, 0.20 * 15, 0.30 * 8, 1.0]))
    assert float((tgt[..., 0] == 1).sum()) == 2.0               # two boxes, one anchor each


def test_metrics_and_writer(tmp_path):
    rs = np.random.RandomState(2)
    gt = rs.uniform(0, 1, size=(2, 24, 20, 3)).astype(np.float32)
    img = np.clip(gt + rs.normal(0, 0.05, size=gt.shape), 0, 1).astype(np.float32)
    # psnr by its definition
    mse = np.mean((img[0].astype(np.float64) - gt[0]) ** 2)
    assert abs(pdata.psnr(img[0], gt[0]) - 10 * np.log10(1.0 / mse)) < 1e-9
    # ssim against a direct evaluation of the definition (explicit loops over windows, float64)
    a, b = img[0].astype(np.float64), gt[0].astype(np.float64)
    vals = []
    for c in range(3):
        for y in range(24 - 6):
            for x in range(20 - 6):
                wa, wb = a[y:y + 7, x:x + 7, c].ravel(), b[y:y + 7, x:x + 7, c].ravel()
                ua, ub = wa.mean(), wb.mean()
                va, vb = wa.var(ddof=1), wb.var(ddof=1)
                cab = np.sum((wa - ua) * (wb - ub)) / 48.0
                vals.append(((2 * ua * ub + 1e-4) * (2 * cab + 9e-4)) / ((ua * ua + ub * ub + 1e-4) * (va + vb + 9e-4)))
    assert abs(pdata.ssim(img[0], gt[0]) - float(np.mean(vals))) < 1e-9
    assert abs(pdata.ssim(gt[0], gt[0]) - 1.0) < 1e-12
    p, s = pdata.write_views(str(tmp_path / "out"), img, [3, 7], gt=gt, write_compare=True)
    assert sorted(os.listdir(tmp_path / "out")) == ["000003.png", "000003_compare.png", "000007.png", "000007_compare.png"]
    back = pdata.imread(str(tmp_path / "out" / "000007.png"))
    assert back.shape == (24, 20, 3) and np.array_equal(back, (img[1] * 255).astype(np.uint8))
    assert pdata.imread(str(tmp_path / "out" / "000003_compare.png")).shape == (24, 40, 3)
    assert abs(p - np.mean([pdata.psnr(img[i], gt[i]) for i in range(2)])) < 1e-9 and 0 < s < 1


def _rot(rs):
    q, _ = np.linalg.qr(rs.randn(3, 3))
    return q * np.sign(np.linalg.det(q))


def _dvr_tree(root, sub_format, n_views=3, size=(12, 16)):
    """<root>/<cat>/<obj>/{image, mask, cameras.npz} + split list, with known cameras."""
    rs = np.random.RandomState(3)
    cat = os.path.join(root, "02958343")
    obj = os.path.join(cat, "obj0")
    os.makedirs(os.path.join(obj, "image"))
    os.makedirs(os.path.join(obj, "mask"))
    prefix = "new_" if sub_format == "dtu" else "softras_"
    with open(os.path.join(cat, prefix + "val.lst"), "w") as fh:
        fh.write("obj0\n")
    H, W = size
    cams, truth = {}, []
    for v in range(n_views):
        img = rs.randint(0, 255, size=(H, W, 3)).astype(np.uint8)
        pdata.imwrite(os.path.join(obj, "image", "%04d.png" % v), img)
        mask = np.zeros((H, W), np.uint8)
        mask[2 + v:8, 3:10 + v] = 255
        pdata.imwrite(os.path.join(obj, "mask", "%04d.png" % v), mask)
        R, C = _rot(rs), rs.uniform(-1, 1, 3)
        if sub_format == "dtu":
            K = np.array([[30.0 + v, 0.0, W / 2 + 0.5], [0.0, 31.0 + v, H / 2 - 0.5], [0.0, 0.0, 1.0]])
            P = np.eye(4)
            P[:3] = K @ np.hstack([R, (-R @ C)[:, None]])
            cams["world_mat_%d" % v] = P * (1.0 + 0.1 * v)              # a projection matrix is defined up to scale
            S = np.diag([2.0, 2.0, 2.0, 1.0])
            S[:3, 3] = [0.1, -0.2, 0.3]
            cams["scale_mat_%d" % v] = S
            c2w = np.eye(4)
            c2w[:3, :3], c2w[:3, 3] = R.T, (C - S[:3, 3]) / 2.0
            truth.append((img, c2w, K))
        else:
            w2c = np.eye(4)
            w2c[:3, :3], w2c[:3, 3] = R, -R @ C
            cams["world_mat_%d" % v] = w2c[:3] if v == 0 else w2c       # both stored shapes occur in the release
            cams["camera_mat_%d" % v] = np.diag([2.5, 2.5, 1.0, 1.0])
            truth.append((img, np.linalg.inv(w2c), None))
    np.savez(os.path.join(obj, "cameras.npz"), **cams)
    return root, obj, truth


def test_dvr_shapenet_contract(tmp_path):
    root, obj, truth = _dvr_tree(str(tmp_path), "shapenet")
    ds = pdata.get_split_dataset("dvr", root, want_split="val", training=False)
    assert isinstance(ds, pdata.DVRDataset) and len(ds) == 1 and (ds.z_near, ds.z_far, ds.lindisp) == (1.2, 4.0, False)
    it = ds[0]
    assert it["path"] == obj and it["images"].shape == (3, 3, 12, 16) and it["masks"].shape == (3, 1, 12, 16) and "c" not in it
    assert torch.equal(it["images"][1], torch.from_numpy(truth[1][0]).permute(2, 0, 1).float() / 255 * 2 - 1)
    assert abs(float(it["focal"]) - 2.5 * 16 / 2) < 1e-6            # scale_focal: camera_mat is for an image spanning [-1, 1]
    world = np.array([[1.0, 0, 0, 0], [0, 0, -1.0, 0], [0, 1.0, 0, 0], [0, 0, 0, 1.0]])     # z-up object -> y-up world
    cam = np.diag([1.0, -1.0, -1.0, 1.0])
    for v in range(3):
        assert np.allclose(it["poses"][v].numpy(), world @ truth[v][1] @ cam, atol=1e-5)
    assert it["bbox"][1].tolist() == [3.0, 3.0, 10.0, 7.0]          # [cmin, rmin, cmax, rmax] of the mask
    it2 = pdata.DVRDataset(root, stage="val", image_size=(6, 8))[0]
    assert it2["images"].shape == (3, 3, 6, 8) and abs(float(it2["focal"]) - 10.0) < 1e-6
    assert torch.allclose(it2["bbox"], it["bbox"] * 0.5) and torch.allclose(it2["images"], torch.nn.functional.avg_pool2d(it["images"], 2), atol=1e-6)
    assert len(pdata.DVRDataset(root, stage="train")) == 0          # no list for that stage


def test_dvr_dtu_contract_and_colour_jitter(tmp_path):
    root, obj, truth = _dvr_tree(str(tmp_path), "dtu")
    ds = pdata.get_split_dataset("dvr_dtu", root, want_split="val", training=False)
    assert (ds.sub_format, ds.z_near, ds.z_far, ds.max_imgs) == ("dtu", 0.1, 5.0, 100000)
    it = ds[0]
    assert "bbox" not in it and it["images"].shape == (3, 3, 12, 16)
    # intrinsics: recovered from the projection matrices (up to scale), averaged over the views, in pixels (scale_focal False)
    Ks = np.stack([t[2] for t in truth])
    assert np.allclose(it["focal"].numpy(), [Ks[:, 0, 0].mean(), Ks[:, 1, 1].mean()], atol=1e-3)
    assert np.allclose(it["c"].numpy(), [Ks[:, 0, 2].mean(), Ks[:, 1, 2].mean()], atol=1e-3)
    flip = np.diag([1.0, -1.0, -1.0, 1.0])
    for v in range(3):                                               # camera centre normalised by scale_mat, y / z flipped
        assert np.allclose(it["poses"][v].numpy(), flip @ truth[v][1] @ flip, atol=1e-4)
    # training split: at most 49 views, wrapped in the colour jitter (one draw per item, all views alike)
    os.rename(os.path.join(root, "02958343", "new_val.lst"), os.path.join(root, "02958343", "new_train.lst"))
    tr = pdata.get_split_dataset("dvr_dtu", root, want_split="train", training=True)
    assert isinstance(tr, pdata.ColorJitterDataset) and tr.base_dset.max_imgs == 49 and tr.sub_format == "dtu" and tr.z_far == 5.0
    np.random.seed(0)
    jit = tr[0]["images"]
    assert jit.shape == (3, 3, 12, 16) and float(jit.min()) >= -1.0 and float(jit.max()) <= 1.0
    assert 1e-3 < float((jit - it["images"]).abs().max()) < 0.6     # changed, by at most a 10 % jitter of each kind
    still = pdata.ColorJitterDataset(ds, 0.0, 0.0, 0.0, 0.0)[0]["images"]
    assert torch.allclose(still, it["images"], atol=2e-6)            # zero ranges: identity (hue 0, factors 1)


def test_colour_adjustments_follow_their_definitions():
    g = torch.Generator().manual_seed(1)
    img = torch.rand(3, 5, 7, generator=g)
    gray = 0.2989 * img[0] + 0.587 * img[1] + 0.114 * img[2]
    assert torch.allclose(pdata.adjust_saturation(img, 0.0), gray.expand(3, -1, -1), atol=1e-6)
    assert torch.allclose(pdata.adjust_contrast(img, 0.0), gray.mean().expand(3, 5, 7), atol=1e-6)
    assert torch.allclose(pdata.adjust_brightness(img, 0.5), img * 0.5)
    assert float(pdata.adjust_brightness(img, 3.0).max()) <= 1.0
    # hue: a third of a turn permutes pure primaries; a full turn there and back is the identity
    prim = torch.tensor([1.0, 0.0, 0.0]).view(3, 1, 1)
    assert torch.allclose(pdata.adjust_hue(prim, 1.0 / 3.0), torch.tensor([0.0, 1.0, 0.0]).view(3, 1, 1), atol=1e-6)
    assert torch.allclose(pdata.adjust_hue(pdata.adjust_hue(img, 0.3), -0.3), img, atol=2e-6)
    import colorsys
    for _ in range(20):
        r, gg, b = (float(x) for x in torch.rand(3, generator=g))
        h, s, v = colorsys.rgb_to_hsv(r, gg, b)
        want = colorsys.hsv_to_rgb((h + 0.07) % 1.0, s, v)
        got = pdata.adjust_hue(torch.tensor([r, gg, b]).view(3, 1, 1), 0.07).flatten().tolist()
        assert np.allclose(got, want, atol=1e-6)


def test_multi_object_dataset_contract(tmp_path):
    import json
    root = str(tmp_path)
    d = os.path.join(root, "val", "scene0")
    os.makedirs(d)
    rs = np.random.RandomState(4)
    frames, truth = [], []
    for v in range(3):
        img = np.zeros((10, 12, 4), np.uint8)                         # transparent background
        img[2 + v:7, 3:9 + v, :3] = rs.randint(1, 255, size=(5 - v, 6 + v, 3))
        img[2 + v:7, 3:9 + v, 3] = 255
        pdata.imwrite(os.path.join(d, "r_%d_obj.png" % v), img)
        pose = synth.pose_spherical(30.0 * v, -25.0, 6.0)
        frames.append({"file_path": "./r_%d" % v, "transform_matrix": pose.tolist()})
        truth.append((img, pose))
    with open(os.path.join(d, "transforms.json"), "w") as fh:
        json.dump({"camera_angle_x": 0.8, "frames": frames}, fh)
    ds = pdata.get_split_dataset("multi_obj", root, want_split="val", training=False)
    assert isinstance(ds, pdata.MultiObjectDataset) and len(ds) == 1 and (ds.z_near, ds.z_far) == (4, 9)
    it = ds[0]
    assert it["images"].shape == (3, 3, 10, 12) and it["masks"].shape == (3, 1, 10, 12) and it["poses"].shape == (3, 4, 4)
    assert abs(it["focal"] - 0.5 * 12 / np.tan(0.4)) < 1e-9
    assert it["bbox"][1].tolist() == [3.0, 3.0, 9.0, 6.0]
    img1 = torch.from_numpy(truth[1][0])
    want = (img1[..., :3].permute(2, 0, 1).float() / 255 * 2 - 1) * (img1[..., 3] / 255.0) + (1 - img1[..., 3] / 255.0)
    assert torch.allclose(it["images"][1], want)                      # object over white
    assert float(it["images"][1][:, 0, 0].min()) == 1.0
    assert np.allclose(it["poses"][2].numpy(), truth[2][1], atol=1e-6)
    assert pdata.MultiObjectDataset(root, stage="val", n_views=2)[0] == {}     # view-count filter
