"""
Data and image adapters either side of the rendering path: the dataset readers that feed
``PixelNeRFNet.encode`` / ``gen_rays`` and the writers / metrics that consume rendered frames.

Host-side Python mirroring the reference's interface (same class names, constructor arguments, item dictionaries):
  SRNDataset   reference src/data/SRNDataset.py:10-136
  YOLODataset  reference src/data/YOLODataset.py:10-225 (incl. the anchor / cell target assignment)
  DVRDataset   reference src/data/DVRDataset.py:11-275 (ShapeNet / NMR renderings and the DTU sub-format)
  MultiObjectDataset  reference src/data/MultiObjectDataset.py:14-117 (scenes of several ShapeNet objects, NeRF-style transforms.json)
  ColorJitterDataset  reference src/data/data_util.py:13-55 (training-time augmentation of ``dvr_dtu`` and ``yolo``)
  get_split_dataset  reference src/data/__init__.py:12-76 (types ``srn``, ``multi_obj``, ``dvr``, ``dvr_gen``, ``dvr_dtu`` and ``yolo``)
  psnr / ssim / write_views  what eval/eval.py:291-359 does with skimage / imageio

PARITY UNPINNED against the third-party libraries the reference uses here -- imageio (decoding), cv2 (resize, projection-
matrix decomposition), torchvision (colour jitter), skimage (metrics) are not importable in this environment, so the reference's classes cannot be imported to capture
goldens.  Decoding / encoding goes through Pillow; ``cv2.resize`` (bilinear) is restated with
``F.interpolate(mode="bilinear", align_corners=False)``, which agrees with OpenCV's INTER_LINEAR up to its fixed-point
rounding on uint8; the metrics follow the published definitions (skimage ``structural_similarity`` defaults: 7x7 uniform
window, sample covariance, K1 = 0.01, K2 = 0.03).  Everything is pinned at the tensor-contract level by
tests/test_cpu_data.py on synthetic directory trees.
"""
import glob
import os

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ image io
def imread(path):
    """(H, W, C) uint8 array, as ``imageio.imread`` returns for 8-bit images."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode not in ("RGB", "RGBA", "L"):
            im = im.convert("RGBA" if "A" in im.getbands() else "RGB")
        return np.array(im)


def imwrite(path, arr):
    """``imageio.imwrite`` for (H, W[, C]) uint8 arrays (format from the extension)."""
    from PIL import Image
    a = np.asarray(arr)
    assert a.dtype == np.uint8, "imwrite takes uint8 images"
    Image.fromarray(a).save(path)


def image_to_tensor_balanced(img):
    """util.get_image_to_tensor_balanced (reference util.py:70-77): ToTensor + Normalize(0.5, 0.5) -> (C, H, W) in [-1, 1]."""
    t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).to(torch.float32).div(255.0)
    return (t - 0.5) / 0.5


def mask_to_tensor(mask):
    """util.get_mask_to_tensor (reference util.py:80-83): ToTensor + Normalize(0, 1) -> (1, H, W) in [0, 1]."""
    return torch.from_numpy(np.ascontiguousarray(mask)).permute(2, 0, 1).to(torch.float32).div(255.0)


def resize_bilinear_u8(img, fx, fy):
    """cv2.resize(img, (0, 0), fx=fx, fy=fy) (INTER_LINEAR) for uint8 (H, W, C): output size round(W fx) x round(H fy),
    half-pixel centres; rounded back to uint8."""
    h, w = img.shape[:2]
    oh, ow = int(round(h * fy)), int(round(w * fx))
    t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1)[None].to(torch.float32)
    out = F.interpolate(t, size=(oh, ow), mode="bilinear", align_corners=False)
    return out[0].permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).numpy()


# ------------------------------------------------------------------ SRN
class SRNDataset(torch.utils.data.Dataset):
    """reference src/data/SRNDataset.py: <path>_<stage>/<object>/{intrinsics.txt, rgb/*, pose/*}."""

    def __init__(self, path, stage="train", image_size=(128, 128), world_scale=1.0):
        super().__init__()
        self.base_path = path + "_" + stage
        self.dataset_name = os.path.basename(path)
        print("Loading SRN dataset", self.base_path, "name:", self.dataset_name)
        self.stage = stage
        assert os.path.exists(self.base_path)
        is_chair = "chair" in self.dataset_name
        if is_chair and stage == "train":
            tmp = os.path.join(self.base_path, "chairs_2.0_train")
            if os.path.exists(tmp):
                self.base_path = tmp
        self.intrins = sorted(glob.glob(os.path.join(self.base_path, "*", "intrinsics.txt")))
        self.image_to_tensor = image_to_tensor_balanced
        self.mask_to_tensor = mask_to_tensor
        self.image_size = image_size
        self.world_scale = world_scale
        self._coord_trans = torch.diag(torch.tensor([1, -1, -1, 1], dtype=torch.float32))
        self.z_near, self.z_far = (1.25, 2.75) if is_chair else (0.8, 1.8)
        self.lindisp = False

    def __len__(self):
        return len(self.intrins)

    def __getitem__(self, index):
        intrin_path = self.intrins[index]
        dir_path = os.path.dirname(intrin_path)
        rgb_paths = sorted(glob.glob(os.path.join(dir_path, "rgb", "*")))
        pose_paths = sorted(glob.glob(os.path.join(dir_path, "pose", "*")))
        assert len(rgb_paths) == len(pose_paths)
        with open(intrin_path, "r") as fh:
            lines = fh.readlines()
            focal, cx, cy, _ = map(float, lines[0].split())
            height, width = map(int, lines[-1].split())
        all_imgs, all_poses, all_masks, all_bboxes = [], [], [], []
        for rgb_path, pose_path in zip(rgb_paths, pose_paths):
            img = imread(rgb_path)[..., :3]
            img_tensor = self.image_to_tensor(img)
            mask = (img != 255).all(axis=-1)[..., None].astype(np.uint8) * 255      # white background = outside
            mask_tensor = self.mask_to_tensor(mask)
            pose = torch.from_numpy(np.loadtxt(pose_path, dtype=np.float32).reshape(4, 4))
            pose = pose @ self._coord_trans
            rows, cols = np.any(mask, axis=1), np.any(mask, axis=0)
            rnz, cnz = np.where(rows)[0], np.where(cols)[0]
            if len(rnz) == 0:
                raise RuntimeError("ERROR: Bad image at", rgb_path, "please investigate!")
            rmin, rmax = rnz[[0, -1]]
            cmin, cmax = cnz[[0, -1]]
            all_bboxes.append(torch.tensor([cmin, rmin, cmax, rmax], dtype=torch.float32))
            all_imgs.append(img_tensor)
            all_masks.append(mask_tensor)
            all_poses.append(pose)
        all_imgs, all_poses = torch.stack(all_imgs), torch.stack(all_poses)
        all_masks, all_bboxes = torch.stack(all_masks), torch.stack(all_bboxes)
        if tuple(all_imgs.shape[-2:]) != tuple(self.image_size):
            scale = self.image_size[0] / all_imgs.shape[-2]
            focal *= scale
            cx *= scale
            cy *= scale
            all_bboxes *= scale
            all_imgs = F.interpolate(all_imgs, size=self.image_size, mode="area")
            all_masks = F.interpolate(all_masks, size=self.image_size, mode="area")
        if self.world_scale != 1.0:
            focal *= self.world_scale
            all_poses[:, :3, 3] *= self.world_scale
        return {
            "path": dir_path, "img_id": index, "focal": torch.tensor(focal, dtype=torch.float32),
            "c": torch.tensor([cx, cy], dtype=torch.float32), "images": all_imgs, "masks": all_masks, "bbox": all_bboxes,
            "poses": all_poses,
        }


# ------------------------------------------------------------------ YOLO
def iou_wh(box_wh, anchors_wh):
    """util.iou(..., is_pred=False) (reference util.py:612-630): IoU of (w, h) pairs anchored at a common corner."""
    inter = torch.min(box_wh[..., 0], anchors_wh[..., 0]) * torch.min(box_wh[..., 1], anchors_wh[..., 1])
    union = box_wh[..., 0] * box_wh[..., 1] + anchors_wh[..., 0] * anchors_wh[..., 1] - inter
    return inter / union


class YOLODataset(torch.utils.data.Dataset):
    """reference src/data/YOLODataset.py: <path>/{train,val,test}.lst of scene directories holding image_XXXX.png,
    extrinsic_XXXX.npy, intrinsic_0000.npy and projected_bboxes_XXXX.txt (cls cx cy w h, normalised)."""

    def __init__(self, path, stage="train", z_near=1.2, z_far=4.0, conf=None):
        super().__init__()
        self.base_path = path
        assert os.path.exists(self.base_path)
        with open(os.path.join(self.base_path, {"train": "train.lst", "val": "val.lst", "test": "test.lst"}[stage]), "r") as fh:
            self.all_objs = [x.strip() for x in fh.readlines()]
        self.stage = stage
        self.image_to_tensor = image_to_tensor_balanced
        self.mask_to_tensor = mask_to_tensor
        print("Loading YOLO dataset", self.base_path, "stage", stage, len(self.all_objs), "objs")
        self.image_scale = conf["yolo.image_scale"]
        self.z_near, self.z_far = z_near, z_far
        self.num_scales = conf["model.mlp_coarse.num_scales"]
        self.num_anchors_per_scale = conf["model.mlp_coarse.num_anchors_per_scale"]
        self.cell_sizes = conf["yolo.cell_sizes"][:self.num_scales]
        anchors = conf["yolo.anchors"][:self.num_scales]
        self.anchors = torch.tensor([item for sub in anchors for item in sub], dtype=torch.float32)
        self.ignore_iou_thresh = conf["yolo.ignore_iou_thresh"]

    def __len__(self):
        return len(self.all_objs)

    def __getitem__(self, index):
        root_dir = os.path.join(self.base_path, self.all_objs[index])
        all_imgs, all_poses, all_bboxes = [], [], []
        n = 0
        while os.path.exists(os.path.join(root_dir, "image_{:04d}.png".format(n))):   # the reference stops at the first failing read
            img = imread(os.path.join(root_dir, "image_{:04d}.png".format(n)))[..., :3]
            img = resize_bilinear_u8(img, self.image_scale[0], self.image_scale[1])
            all_imgs.append(self.image_to_tensor(img))
            n += 1
        for i in range(n):
            pose = np.load(os.path.join(root_dir, "extrinsic_{:04d}.npy".format(i))).copy()
            pose[0] = pose[0] * -1                                                     # YOLODataset.py:112
            all_poses.append(torch.tensor(pose, dtype=torch.float32))
        for i in range(n):
            bb = np.roll(np.loadtxt(os.path.join(root_dir, "projected_bboxes_{:04d}.txt".format(i)), delimiter=" ", ndmin=2),
                         4, axis=1).tolist()                                           # -> cx, cy, w, h, cls
            all_bboxes.append(self._get_all_bboxes(bb, all_imgs[i].shape[1], all_imgs[i].shape[2]))
        intrinsic = np.load(os.path.join(root_dir, "intrinsic_0000.npy"))
        focal = torch.tensor(intrinsic[0, 0] * np.array(self.image_scale), dtype=torch.float32)
        c = torch.tensor(intrinsic[:2, 2] * self.image_scale, dtype=torch.float32)
        return {"path": root_dir, "img_id": index, "focal": focal, "images": torch.stack(all_imgs), "bboxes": all_bboxes,
                "poses": torch.stack(all_poses), "c": c}

    def _get_all_bboxes(self, bboxes, height, width):
        """Target grids per scale, (s_h, s_w, A, 6) = [objectness, x_cell, y_cell, w_cells, h_cells, class]: each box goes
        to the best free anchor of every scale; other anchors above the IoU threshold are marked -1 (ignored)."""
        grid_sizes = [(height // cs, width // cs) for cs in self.cell_sizes]
        targets = [torch.zeros((s_h, s_w, self.num_anchors_per_scale, 6)) for (s_h, s_w) in grid_sizes]
        for box in bboxes:
            iou_anchors = iou_wh(torch.tensor(box[2:4]), self.anchors)
            anchor_indices = iou_anchors.argsort(descending=True, dim=0)
            x, y, box_width, box_height, class_label = box
            has_anchor = [False] * self.num_scales
            for anchor_idx in anchor_indices:
                scale_idx = int(anchor_idx // self.num_anchors_per_scale)
                anchor_on_scale = int(anchor_idx % self.num_anchors_per_scale)
                s_h, s_w = grid_sizes[scale_idx]
                i, j = int(s_h * y), int(s_w * x)
                anchor_taken = targets[scale_idx][i, j, anchor_on_scale, 0]
                if not anchor_taken and not has_anchor[scale_idx]:
                    targets[scale_idx][i, j, anchor_on_scale, 0] = 1
                    targets[scale_idx][i, j, anchor_on_scale, 1:5] = torch.tensor(
                        [s_w * x - j, s_h * y - i, box_width * s_w, box_height * s_h])
                    targets[scale_idx][i, j, anchor_on_scale, 5] = int(class_label)
                    has_anchor[scale_idx] = True
                elif not anchor_taken and iou_anchors[anchor_idx] > self.ignore_iou_thresh:
                    targets[scale_idx][i, j, anchor_on_scale, 0] = -1
        return tuple(targets)


def get_split_dataset(dataset_type, datadir, want_split="all", training=True, **kwargs):
    """reference src/data/__init__.py:12-76: dataset class + flags per type name; the training split of ``dvr_dtu`` and
    ``yolo`` is wrapped in the colour-jitter augmentation."""
    flags, aug, aug_flags = {}, None, {}
    if dataset_type == "srn":
        dset_class = SRNDataset
    elif dataset_type == "multi_obj":
        dset_class = MultiObjectDataset
    elif dataset_type.startswith("dvr"):
        dset_class = DVRDataset
        if dataset_type == "dvr_gen":
            flags["list_prefix"] = "gen_"
        elif dataset_type == "dvr_dtu":
            flags.update(list_prefix="new_", sub_format="dtu", scale_focal=False, z_near=0.1, z_far=5.0)
            if training:
                flags["max_imgs"] = 49
            aug, aug_flags = ColorJitterDataset, {"extra_inherit_attrs": ["sub_format"]}
    elif dataset_type == "yolo":
        dset_class = YOLODataset
        flags["z_near"], flags["z_far"] = 1, 13.0
        aug = ColorJitterDataset
    else:
        raise NotImplementedError("dataset type %r is not implemented here (srn, multi_obj, dvr, dvr_gen, dvr_dtu, yolo)" % dataset_type)
    want_train = want_split not in ("val", "test")
    want_val = want_split not in ("train", "test")
    want_test = want_split not in ("train", "val")
    sets = [dset_class(datadir, stage=st, **flags, **kwargs) if want else None
            for st, want in (("train", want_train), ("val", want_val), ("test", want_test))]
    if sets[0] is not None and aug is not None:
        sets[0] = aug(sets[0], **aug_flags)
    if want_split in ("train", "val", "test"):
        return sets[("train", "val", "test").index(want_split)]
    return tuple(sets)


# ------------------------------------------------------------------ DVR (ShapeNet 64x64 / NMR, DTU)
def decompose_projection(P):
    """cv2.decomposeProjectionMatrix(P)[:3] for a 3x4 projection P = K [R | -R C]: the upper-triangular calibration K with a
    positive diagonal and the rotation R from an RQ factorisation of P[:, :3], and the camera centre C as a homogeneous
    4-vector (the null vector of P, scaled like OpenCV's: unit norm)."""
    import scipy.linalg
    P = np.asarray(P, dtype=np.float64)
    K, R = scipy.linalg.rq(P[:, :3])
    sign = np.diag(np.sign(np.diag(K)))          # RQ is unique up to the signs of K's diagonal
    K, R = K @ sign, sign @ R
    _, _, vt = np.linalg.svd(P)
    c = vt[-1]
    return K, R, c[:, None]


class DVRDataset(torch.utils.data.Dataset):
    """Objects of the DVR release (Niemeyer et al. 2020): <root>/<category>/<object>/{image/*.png|jpg, mask/*.png,
    cameras.npz}, split lists <root>/<category>/<list_prefix><stage>.lst.  Item = {path, img_id, focal, images (NV, 3, H, W)
    in [-1, 1], poses (NV, 4, 4) camera-to-world in the renderer's convention, masks?, and c (DTU) or bbox (ShapeNet)}."""

    def __init__(self, path, stage="train", list_prefix="softras_", image_size=None, sub_format="shapenet", scale_focal=True,
                 max_imgs=100000, z_near=1.2, z_far=4.0, skip_step=None, conf=None):
        super().__init__()
        assert os.path.exists(path), path
        assert stage in ("train", "val", "test")
        self.base_path, self.stage = path, stage
        self.all_objs = []
        for cat_dir in (d for d in glob.glob(os.path.join(path, "*")) if os.path.isdir(d)):
            lst = os.path.join(cat_dir, list_prefix + stage + ".lst")
            if os.path.exists(lst):
                with open(lst) as fh:
                    self.all_objs += [(os.path.basename(cat_dir), os.path.join(cat_dir, ln.strip())) for ln in fh.readlines()]
        self.image_to_tensor, self.mask_to_tensor = image_to_tensor_balanced, mask_to_tensor
        self.image_size = image_size
        flip_yz = torch.diag(torch.tensor([1.0, -1.0, -1.0, 1.0]))
        # world frame: DTU keeps its axes up to the y/z flip, ShapeNet's z-up objects are turned to y-up
        self._coord_trans_world = flip_yz if sub_format == "dtu" else torch.tensor(
            [[1.0, 0, 0, 0], [0, 0, -1.0, 0], [0, 1.0, 0, 0], [0, 0, 0, 1.0]])
        self._coord_trans_cam = flip_yz            # OpenCV camera (y down, z forward) -> OpenGL camera
        self.sub_format, self.scale_focal, self.max_imgs = sub_format, scale_focal, max_imgs
        self.z_near, self.z_far, self.lindisp = z_near, z_far, False

    def __len__(self):
        return len(self.all_objs)

    def __getitem__(self, index):
        _, root = self.all_objs[index]
        rgb_paths = sorted(f for f in glob.glob(os.path.join(root, "image", "*")) if f.endswith((".jpg", ".png")))
        mask_paths = sorted(glob.glob(os.path.join(root, "mask", "*.png"))) or [None] * len(rgb_paths)
        sel = np.arange(len(rgb_paths))
        if len(rgb_paths) > self.max_imgs:
            sel = np.random.choice(len(rgb_paths), self.max_imgs, replace=False)
            rgb_paths, mask_paths = [rgb_paths[i] for i in sel], [mask_paths[i] for i in sel]
        cams = np.load(os.path.join(root, "cameras.npz"))
        dtu = self.sub_format != "shapenet"
        imgs, poses, masks, bboxes = [], [], [], []
        focal, intr_sum = None, torch.zeros(4, dtype=torch.float64)     # DTU: fx, fy, cx, cy averaged over the views
        for i, rgb_path, mask_path in zip(sel, rgb_paths, mask_paths):
            img = imread(rgb_path)[..., :3]
            # scale_focal: intrinsics are given for an image spanning [-1, 1]
            xs, ys, delta = (img.shape[1] / 2.0, img.shape[0] / 2.0, 1.0) if self.scale_focal else (1.0, 1.0, 0.0)
            if dtu:
                K, R, t = decompose_projection(cams["world_mat_%d" % i][:3])
                K = K / K[2, 2]
                pose = np.eye(4, dtype=np.float32)
                pose[:3, :3] = R.T
                pose[:3, 3] = (t[:3] / t[3])[:, 0]
                scale = cams.get("scale_mat_%d" % i)
                if scale is not None:                                # normalisation of the scene into the unit sphere
                    pose[:3, 3:] -= scale[:3, 3:]
                    pose[:3, 3:] /= np.diagonal(scale[:3, :3])[..., None]
                intr_sum += torch.tensor([K[0, 0] * xs, K[1, 1] * ys, (K[0, 2] + delta) * xs, (K[1, 2] + delta) * ys])
            else:
                if "world_mat_inv_%d" % i in cams:
                    pose = cams["world_mat_inv_%d" % i]
                else:
                    w = cams["world_mat_%d" % i]
                    if w.shape[0] == 3:
                        w = np.vstack((w, np.array([0, 0, 0, 1])))
                    pose = np.linalg.inv(w)
                intr = cams["camera_mat_%d" % i]
                assert abs(intr[0, 0] - intr[1, 1]) < 1e-9
                f = intr[0, 0] * xs
                assert focal is None or abs(f - focal) < 1e-5, "views of one object must share the focal length"
                focal = f
            poses.append(self._coord_trans_world @ torch.tensor(pose, dtype=torch.float32) @ self._coord_trans_cam)
            imgs.append(self.image_to_tensor(img))
            if mask_path is not None:
                mask = imread(mask_path)
                mask = (mask[..., None] if mask.ndim == 2 else mask)[..., :1]
                rows, cols = np.where(np.any(mask, axis=1))[0], np.where(np.any(mask, axis=0))[0]
                if len(rows) == 0:
                    raise RuntimeError("empty mask for " + rgb_path)
                masks.append(self.mask_to_tensor(mask))
                bboxes.append(torch.tensor([cols[0], rows[0], cols[-1], rows[-1]], dtype=torch.float32))
        images, poses = torch.stack(imgs), torch.stack(poses)
        masks = torch.stack(masks) if masks else None
        if dtu:
            intr = (intr_sum / len(rgb_paths)).to(torch.float32)
            focal, c, bbox = intr[:2].clone(), intr[2:].clone(), None
        else:
            focal, c, bbox = focal, None, (torch.stack(bboxes) if bboxes else [])
        if self.image_size is not None and tuple(images.shape[-2:]) != tuple(self.image_size):
            scale = self.image_size[0] / images.shape[-2]
            focal = focal * scale
            if dtu:
                c = c * scale
            elif len(bbox):
                bbox = bbox * scale
            images = F.interpolate(images, size=tuple(self.image_size), mode="area")
            if masks is not None:
                masks = F.interpolate(masks, size=tuple(self.image_size), mode="area")
        item = {"path": root, "img_id": index, "focal": focal, "images": images, "poses": poses}
        if masks is not None:
            item["masks"] = masks
        if dtu:
            item["c"] = c
        else:
            item["bbox"] = bbox
        return item


# ------------------------------------------------------------------ multi-object scenes
class MultiObjectDataset(torch.utils.data.Dataset):
    """<root>/<stage>/**/transforms.json (camera_angle_x, frames[file_path, transform_matrix]) with <basename>_obj.png RGBA
    renderings beside it.  Item = {path, img_id, focal, images (NV, 3, H, W) composited on white, masks (NV, 1, H, W) = alpha,
    bbox (NV, 4) of the non-empty pixels, poses (NV, 4, 4)}; an instance whose view count differs from ``n_views`` yields {}."""

    def __init__(self, path, stage="train", z_near=4, z_far=9, n_views=None):
        super().__init__()
        self.base_path = os.path.join(path, stage)
        self.trans_files = sorted(os.path.join(root, "transforms.json") for root, _, files in os.walk(self.base_path)
                                  if "transforms.json" in files)
        self.image_to_tensor, self.mask_to_tensor = image_to_tensor_balanced, mask_to_tensor
        self.z_near, self.z_far, self.lindisp, self.n_views = z_near, z_far, False, n_views

    def __len__(self):
        return len(self.trans_files)

    def _check_valid(self, index):
        if self.n_views is None:
            return True
        import json
        trans_file = self.trans_files[index]
        try:
            with open(trans_file) as fh:
                transform = json.load(fh)
        except Exception:
            return False
        return (len(transform["frames"]) == self.n_views and
                len(glob.glob(os.path.join(os.path.dirname(trans_file), "*.png"))) == self.n_views)

    def __getitem__(self, index):
        import json
        if not self._check_valid(index):
            return {}
        trans_file = self.trans_files[index]
        dir_path = os.path.dirname(trans_file)
        with open(trans_file) as fh:
            transform = json.load(fh)
        imgs, masks, bboxes, poses = [], [], [], []
        for frame in transform["frames"]:
            base = os.path.splitext(os.path.basename(frame["file_path"]))[0]
            img = imread(os.path.join(dir_path, base + "_obj.png"))
            mask = self.mask_to_tensor(img[..., 3:4])
            rows, cols = np.where(np.any(img, axis=(1, 2)))[0], np.where(np.any(img, axis=(0, 2)))[0]
            if len(rows) == 0:
                box = [0, 0, mask.shape[-1], mask.shape[-2]]
            else:
                box = [cols[0], rows[0], cols[-1], rows[-1]]
            bboxes.append(torch.tensor(box, dtype=torch.float32))
            imgs.append(self.image_to_tensor(img[..., :3]) * mask + (1.0 - mask))     # white where transparent
            masks.append(mask)
            poses.append(torch.tensor(frame["transform_matrix"]))
        images = torch.stack(imgs)
        focal = 0.5 * images.shape[-1] / np.tan(0.5 * transform.get("camera_angle_x"))
        return {"path": dir_path, "img_id": index, "focal": focal, "images": images, "masks": torch.stack(masks),
                "bbox": torch.stack(bboxes), "poses": torch.stack(poses)}


# ------------------------------------------------------------------ colour jitter (training-time augmentation)
def _gray(img):
    return (0.2989 * img[..., 0, :, :] + 0.587 * img[..., 1, :, :] + 0.114 * img[..., 2, :, :]).unsqueeze(-3)


def _blend(a, b, ratio):
    return (ratio * a + (1.0 - ratio) * b).clamp(0.0, 1.0)


def adjust_brightness(img, factor):
    """torchvision ``adjust_brightness`` on a float (3, H, W) image in [0, 1]: blend with black."""
    return _blend(img, torch.zeros_like(img), factor)


def adjust_contrast(img, factor):
    """blend with the mean grey level of the image."""
    return _blend(img, _gray(img).mean(dim=(-3, -2, -1), keepdim=True), factor)


def adjust_saturation(img, factor):
    """blend with the grey image."""
    return _blend(img, _gray(img), factor)


def adjust_hue(img, factor):
    """rotate the hue by ``factor`` (in turns, |factor| <= 0.5): RGB -> HSV, h <- (h + factor) mod 1, HSV -> RGB."""
    assert -0.5 <= factor <= 0.5
    r, g, b = img.unbind(-3)
    maxc, minc = img.max(-3).values, img.min(-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    crd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    h = (h + factor) % 1.0
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int32) % 6
    v = maxc
    p, q, t = (v * (1.0 - s)).clamp(0, 1), (v * (1.0 - s * f)).clamp(0, 1), (v * (1.0 - s * (1.0 - f))).clamp(0, 1)
    sel = i.unsqueeze(-3) == torch.arange(6, device=img.device).view(-1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v), dim=-3)
    a2 = torch.stack((t, v, v, q, p, p), dim=-3)
    a3 = torch.stack((p, p, t, v, v, q), dim=-3)
    a4 = torch.stack((a1, a2, a3), dim=-4)
    return torch.einsum("...ijk,...xijk->...xjk", sel.to(img.dtype), a4)


class ColorJitterDataset(torch.utils.data.Dataset):
    """Wraps a dataset: ONE random (hue, saturation, brightness, contrast) draw per item, applied to all of its views
    (images are mapped from [-1, 1] to [0, 1] and back; order saturation, hue, contrast, brightness)."""

    def __init__(self, base_dset, hue_range=0.1, saturation_range=0.1, brightness_range=0.1, contrast_range=0.1,
                 extra_inherit_attrs=()):
        self.hue_range = [-hue_range, hue_range]
        self.saturation_range = [1 - saturation_range, 1 + saturation_range]
        self.brightness_range = [1 - brightness_range, 1 + brightness_range]
        self.contrast_range = [1 - contrast_range, 1 + contrast_range]
        self.base_dset = base_dset
        for attr in ["z_near", "z_far", "base_path", "image_to_tensor", *extra_inherit_attrs]:
            setattr(self, attr, getattr(base_dset, attr))

    def apply_color_jitter(self, images):
        hue, sat = np.random.uniform(*self.hue_range), np.random.uniform(*self.saturation_range)
        bri, con = np.random.uniform(*self.brightness_range), np.random.uniform(*self.contrast_range)
        for i in range(len(images)):
            t = (images[i] + 1.0) * 0.5
            t = adjust_brightness(adjust_contrast(adjust_hue(adjust_saturation(t, sat), hue), con), bri)
            images[i] = t * 2.0 - 1.0
        return images

    def __len__(self):
        return len(self.base_dset)

    def __getitem__(self, idx):
        data = self.base_dset[idx]
        data["images"] = self.apply_color_jitter(data["images"])
        return data


# ------------------------------------------------------------------ metrics / writers (eval/eval.py:291-359)
def psnr(img, gt, data_range=1.0):
    """skimage compare_psnr: 10 log10(data_range^2 / mse), float64."""
    a, b = np.asarray(img, dtype=np.float64), np.asarray(gt, dtype=np.float64)
    return float(10.0 * np.log10(data_range ** 2 / np.mean((a - b) ** 2)))


def ssim(img, gt, data_range=1.0, win_size=7):
    """skimage compare_ssim(multichannel=True, data_range=1) with its defaults: uniform win_size x win_size window, sample
    covariance (n / (n - 1)), K1 = 0.01, K2 = 0.03, mean over the valid (un-padded) region, then over channels.
    img, gt (H, W, C) in [0, data_range]."""
    a = torch.from_numpy(np.asarray(img, dtype=np.float64)).permute(2, 0, 1)[None]
    b = torch.from_numpy(np.asarray(gt, dtype=np.float64)).permute(2, 0, 1)[None]
    n = win_size * win_size
    cov_norm = n / (n - 1.0)
    filt = lambda t: F.avg_pool2d(t, win_size, stride=1)
    ux, uy = filt(a), filt(b)
    uxx, uyy, uxy = filt(a * a), filt(b * b), filt(a * b)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    return float(s.mean())


def write_views(out_dir, rgb, view_ids, gt=None, write_compare=False):
    """eval/eval.py:293-340: rgb (NV, H, W, 3) in [0, 1] -> <out_dir>/<id:06>.png (+ _compare.png beside the ground truth);
    returns (mean psnr, mean ssim) when gt is given."""
    os.makedirs(out_dir, exist_ok=True)
    rgb = np.clip(np.asarray(rgb, dtype=np.float32), 0.0, 1.0)
    tot_p = tot_s = 0.0
    for i, vid in enumerate(view_ids):
        imwrite(os.path.join(out_dir, "{:06}.png".format(int(vid))), (rgb[i] * 255).astype(np.uint8))
        if gt is not None:
            tot_s += ssim(rgb[i], gt[i])
            tot_p += psnr(rgb[i], gt[i])
            if write_compare:
                imwrite(os.path.join(out_dir, "{:06}_compare.png".format(int(vid))),
                        (np.hstack((rgb[i], np.asarray(gt[i], dtype=np.float32))) * 255).astype(np.uint8))
    if gt is None:
        return None
    return tot_p / len(view_ids), tot_s / len(view_ids)
