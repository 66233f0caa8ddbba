"""
Ray generation and the YOLO detection tail with the reference's signatures (src/util/util.py:240-278 ``gen_rays``,
:808-876 ``gen_rays_yolo``), executed by libpnyolo's gen_rays kernel.  Output lives on a CUDA
device (that of ``poses``, or the current one when ``poses`` is a CPU tensor as at the reference's call sites).
"""
import ctypes as C

import torch

from . import lib as _lib
from .lib import check, ptr, stream_of


def _pair(v, name):
    t = torch.as_tensor(v, dtype=torch.float32).detach().cpu().reshape(-1)
    if t.numel() == 1:
        t = t.repeat(2)
    assert t.numel() == 2, "%s must be a scalar or (x, y)" % name
    return (C.c_float * 2)(float(t[0]), float(t[1]))


def _device_of(poses, device):
    """Device the rays are generated on: `device` if given, the device of `poses` if that is a GPU, else the
    current CUDA device -- the reference's call sites pass CPU poses and move the result afterwards
    (eval/eval.py:258 `.to(device=device)`, a no-op then).  No GPU at all is an error: there is no CPU path."""
    if device is not None:
        dev = torch.device(device)
    elif poses.device.type == "cuda":
        dev = poses.device
    elif torch.cuda.is_available():
        dev = torch.device("cuda", torch.cuda.current_device())
    else:
        dev = poses.device
    if dev.type != "cuda":
        raise RuntimeError("libpnyolo needs a cuda device: there is no CPU path")
    return dev


def gen_rays(poses, width, height, focal, z_near, z_far, c=None, ndc=False, device=None):
    """
    :param poses (B, 4, 4) camera-to-world
    :return (B, H, W, 8) [origin(3), unit direction(3), near, far]
    """
    if ndc:
        raise NotImplementedError("ndc=True calls an undefined ndc_rays in the reference (util.py:262): dead branch")
    dev = _device_of(poses, device)
    B = poses.shape[0]
    out = torch.empty(B, height, width, 8, device=dev, dtype=torch.float32)
    f = _pair(focal, "focal")
    cc = _pair([width * 0.5, height * 0.5] if c is None else c, "c")
    p = poses.detach().to("cpu", torch.float32).contiguous()
    check(_lib.load().pny_gen_rays(ptr(p), B, int(width), int(height), f, cc, float(z_near), float(z_far), 0,
                                   ptr(out), stream_of(dev)))
    return out


def gen_rays_range(poses, width, height, focal, z_near, z_far, first_ray, n_rays, c=None, yolo=False, device=None):
    """Rays [first_ray, first_ray + n_rays) of the flattened (B, H, W) pixel grid of ``gen_rays`` (or
    ``gen_rays_yolo`` with yolo=True) -> (n_rays, 8), bit-identical to the corresponding rows of the full call.
    Not part of the reference's interface: it is how a rank of a ray-sharded render (dist.render_frame_sharded)
    produces its own slice of a frame on its own device."""
    dev = _device_of(poses, device)
    B = poses.shape[0]
    out = torch.empty(int(n_rays), 8, device=dev, dtype=torch.float32)
    f = _pair(focal, "focal")
    cc = _pair([width * 0.5, height * 0.5] if c is None else c, "c")
    p = poses.detach().to("cpu", torch.float32).contiguous()
    check(_lib.load().pny_gen_rays_range(ptr(p), B, int(width), int(height), f, cc, float(z_near), float(z_far),
                                         int(bool(yolo)), int(first_ray), int(n_rays), ptr(out), stream_of(dev)))
    return out


def gen_rays_yolo(poses, width, height, focal, c, z_near, z_far, device=None):
    """
    :param poses (B, 4, 4) world-to-camera extrinsics;  focal (2), c (2)
    :return (B, H, W, 8) [origin(3), direction(3) (not normalised), near, far]
    """
    dev = _device_of(poses, device)
    B = poses.shape[0]
    out = torch.empty(B, height, width, 8, device=dev, dtype=torch.float32)
    p = poses.detach().to("cpu", torch.float32).contiguous()
    check(_lib.load().pny_gen_rays(ptr(p), B, int(width), int(height), _pair(focal, "focal"), _pair(c, "c"),
                                   float(z_near), float(z_far), 1, ptr(out), stream_of(dev)))
    return out


# ------------------------------------------------------------------ YOLO detection tail
def _boxes_to_dev(bboxes, dev):
    t = torch.as_tensor(bboxes, dtype=torch.float32).reshape(-1, 6)
    return t.to(dev).contiguous()


def convert_cells_to_bboxes(predictions, anchors, h, w, is_predictions=True, as_tensor=False):
    """reference src/util/util.py:633-689.  predictions (B, h, w, A, 7 | 6) on a cuda device.
    Returns the reference's nested list (B x (A*h*w) x 6: [class, score, x, y, w, h]) or, with
    as_tensor=True, a (B, A*h*w, 6) device tensor (no host round trip)."""
    dev = _device_of(predictions, None)   # the reference's call site hands over CPU tensors (YoloTrainer.py:279-289)
    L = _lib.load()
    p = predictions.detach().to(dev, torch.float32).contiguous()
    B, A = p.shape[0], p.shape[3]
    assert p.shape[1] == h and p.shape[2] == w and p.shape[4] == (7 if is_predictions else 6)
    anc = torch.as_tensor(anchors, dtype=torch.float32).detach().cpu().reshape(-1, 2).contiguous()
    assert anc.shape[0] == A
    out = torch.empty(B, h * w * A, 6, device=dev, dtype=torch.float32)
    for b in range(B):
        check(L.pny_cells_to_bboxes(ptr(p[b]), ptr(anc), h, w, A, int(bool(is_predictions)), ptr(out[b]),
                                    stream_of(dev)))
    return out if as_tensor else out.cpu().tolist()


def nms(bboxes, iou_threshold, threshold, device=None, as_tensor=False):
    """reference src/util/util.py:691-722 -> (kept boxes, highest confidence, boxes above threshold).
    bboxes: list of [class, score, x, y, w, h] or an (n, 6) tensor."""
    dev = torch.device(device) if device is not None else (bboxes.device if torch.is_tensor(bboxes) else torch.device("cuda"))
    b = _boxes_to_dev(bboxes, dev)
    n = b.shape[0]
    if n == 0:
        raise ValueError("max() arg is an empty sequence")  # what the reference raises on an empty list
    L = _lib.load()
    kept = torch.empty(n, 6, device=b.device, dtype=torch.float32)
    meta = torch.zeros(2, device=b.device, dtype=torch.int32)
    hc = torch.empty(1, device=b.device, dtype=torch.float32)
    check(L.pny_nms(ptr(b), n, float(iou_threshold), float(threshold), ptr(kept), C.c_void_p(meta.data_ptr()), ptr(hc),
                    stream_of(b.device)))
    m = meta.cpu()
    kept = kept[: int(m[0])]
    return (kept if as_tensor else kept.cpu().tolist()), float(hc.item()), int(m[1])


def calculate_tp_fp_fn(target_bboxes, prediction_bboxes, nms_iou, nms_t, match_iou, print_hc=False, device=None):
    """reference src/util/util.py:765-802 -> (tp, fp, fn)."""
    dev = torch.device(device) if device is not None else (
        prediction_bboxes.device if torch.is_tensor(prediction_bboxes) else torch.device("cuda"))
    t, p = _boxes_to_dev(target_bboxes, dev), _boxes_to_dev(prediction_bboxes, dev)
    if t.shape[0] == 0 or p.shape[0] == 0:
        raise ValueError("max() arg is an empty sequence")  # nms() of the reference on an empty list
    out = torch.zeros(3, device=t.device, dtype=torch.int32)
    check(_lib.load().pny_tp_fp_fn(ptr(t), t.shape[0], ptr(p), p.shape[0], float(nms_iou), float(nms_t),
                                   float(match_iou), C.c_void_p(out.data_ptr()), stream_of(t.device)))
    tp, fp, fn = (int(v) for v in out.cpu())
    return tp, fp, fn


def calculate_precision_recall_f1(tp, fp, fn):
    """reference src/util/util.py:798-803 (host arithmetic on three integers)."""
    precision = tp / (tp + fp) if tp + fp > 0 else 0
    recall = tp / (tp + fn) if tp + fn > 0 else 0
    f1 = 2 * (precision * recall) / (precision + recall) if precision + recall > 0 else 0
    return precision, recall, f1
