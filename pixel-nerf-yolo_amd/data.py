"""
Data and image adapters either side of the rendering path: the dataset readers that feed
``PixelNeRFNet.encode`` / ``gen_rays`` and the writers / metrics that consume rendered frames.

Host-side Python mirroring the reference's interface (same class names, constructor arguments, item dictionaries):
  SRNDataset   reference src/data/SRNDataset.py:10-136
  YOLODataset  reference src/data/YOLODataset.py:10-225 (incl. the anchor / cell target assignment)
  get_split_dataset  reference src/data/__init__.py:12-76 (types ``srn`` and ``yolo``)
  psnr / ssim / write_views  what eval/eval.py:291-359 does with skimage / imageio

PARITY UNPINNED against the third-party libraries the reference uses here -- imageio (decoding), cv2 (resize),
skimage (metrics) are not importable in this environment, so the reference's classes cannot be imported to capture
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
    """reference src/data/__init__.py:12-76 for the dataset types implemented here (no colour-jitter wrapper)."""
    flags = {}
    if dataset_type == "srn":
        dset_class = SRNDataset
    elif dataset_type == "yolo":
        dset_class = YOLODataset
        flags["z_near"], flags["z_far"] = 1, 13.0
    else:
        raise NotImplementedError("dataset type %r is not implemented here (srn, yolo)" % dataset_type)
    want_train = want_split not in ("val", "test")
    want_val = want_split not in ("train", "test")
    want_test = want_split not in ("train", "val")
    sets = [dset_class(datadir, stage=st, **flags, **kwargs) if want else None
            for st, want in (("train", want_train), ("val", want_val), ("test", want_test))]
    if want_split in ("train", "val", "test"):
        return sets[("train", "val", "test").index(want_split)]
    return tuple(sets)


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
