"""
Ray generation with the reference's signatures (src/util/util.py:240-278 ``gen_rays``,
:808-876 ``gen_rays_yolo``), executed by libpnyolo's gen_rays kernel.  Output lives on the
device of ``poses`` (which must be a CUDA device: there is no CPU path).
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
    dev = torch.device(device) if device is not None else poses.device
    if dev.type != "cuda":
        raise RuntimeError("gen_rays (libpnyolo) needs a cuda device: pass poses on the GPU or device=")
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
