"""
Host-side mirror of the reference's model layer (src/model/): ``make_model`` and
``PixelNeRFNet`` with the same constructor, ``encode`` / ``forward`` / ``load_weights``
signatures and the same state_dict key names, executing on libpnyolo.so.

torch is used for what it is good at here: holding parameters / checkpoints (state_dict
compatibility with the reference's ``pixel_nerf_latest`` files), device memory and streams.
No torch operator runs on the hot path: ``encode`` and ``forward`` hand raw device pointers to
the C ABI (include/pnyolo.h).  Forward only -- a call under autograd raises (backward is the
"next" row of SURVEY.md 8f).
"""
import ctypes as C
import os
import os.path as osp
import warnings

import numpy as np
import torch
from torch import nn

from . import lib as _lib
from .lib import ModelDesc, check, ptr, stream_of


# ------------------------------------------------------------------ parameter containers
class _ResnetBlockFC(nn.Module):
    """Parameters of one block; init as reference src/model/resnetfc.py:19-51."""

    def __init__(self, size):
        super().__init__()
        self.fc_0 = nn.Linear(size, size)
        self.fc_1 = nn.Linear(size, size)
        nn.init.constant_(self.fc_0.bias, 0.0)
        nn.init.kaiming_normal_(self.fc_0.weight, a=0, mode="fan_in")
        nn.init.constant_(self.fc_1.bias, 0.0)
        nn.init.zeros_(self.fc_1.weight)


# Bumped by every parameter / buffer / submodule registration in the process (see PixelNeRFNet._weights_key).
_STRUCT_EPOCH = [0]


def _bump_struct_epoch(*_args, **_kwargs):
    _STRUCT_EPOCH[0] += 1


for _reg in ("register_module_parameter_registration_hook", "register_module_buffer_registration_hook",
             "register_module_module_registration_hook"):
    getattr(torch.nn.modules.module, _reg)(_bump_struct_epoch)


class ResnetFC(nn.Module):
    """Parameter container with the reference's names / shapes / init
    (src/model/resnetfc.py:66-132, from_conf :188-205).  The arithmetic lives in csrc/mlp.hip."""

    def __init__(self, d_in, d_out=4, n_blocks=5, d_latent=0, d_hidden=128, beta=0.0, combine_layer=1000,
                 combine_type="average", use_spade=False):
        super().__init__()
        if beta > 0 or use_spade or combine_type != "average":
            raise NotImplementedError("libpnyolo supports ReLU, no SPADE, average combine (the shipped configs)")
        self.lin_in = nn.Linear(d_in, d_hidden)
        nn.init.constant_(self.lin_in.bias, 0.0)
        nn.init.kaiming_normal_(self.lin_in.weight, a=0, mode="fan_in")
        self.lin_out = nn.Linear(d_hidden, d_out)
        nn.init.constant_(self.lin_out.bias, 0.0)
        nn.init.kaiming_normal_(self.lin_out.weight, a=0, mode="fan_in")
        self.n_blocks, self.d_latent, self.d_in, self.d_out, self.d_hidden = n_blocks, d_latent, d_in, d_out, d_hidden
        self.combine_layer, self.combine_type, self.use_spade = combine_layer, combine_type, use_spade
        self.blocks = nn.ModuleList([_ResnetBlockFC(d_hidden) for _ in range(n_blocks)])
        if d_latent != 0:
            n_lin_z = min(combine_layer, n_blocks)
            self.lin_z = nn.ModuleList([nn.Linear(d_latent, d_hidden) for _ in range(n_lin_z)])
            for i in range(n_lin_z):
                nn.init.constant_(self.lin_z[i].bias, 0.0)
                nn.init.kaiming_normal_(self.lin_z[i].weight, a=0, mode="fan_in")

    @classmethod
    def from_conf(cls, conf, d_in, **kwargs):
        if not conf.get_bool("yolo", False):
            d_out = conf.get_int("d_out", 4)
        else:
            d_out = conf.get_int("d_out", 7) * conf.get_int("num_anchors_per_scale", 3)
        return cls(d_in, d_out=d_out, n_blocks=conf.get_int("n_blocks", 5), d_hidden=conf.get_int("d_hidden", 128),
                   beta=conf.get_float("beta", 0.0), combine_layer=conf.get_int("combine_layer", 1000),
                   combine_type=conf.get_string("combine_type", "average"),
                   use_spade=conf.get_bool("use_spade", False), **kwargs)

    def forward(self, *a, **k):
        raise RuntimeError("ResnetFC is evaluated inside libpnyolo's fused kernel; call PixelNeRFNet.forward")


class PositionalEncoding(nn.Module):
    """Buffers `_freqs`, `_phases` as in reference src/model/code.py:11-28 (checkpoint keys)."""

    def __init__(self, num_freqs=6, d_in=3, freq_factor=np.pi, include_input=True):
        super().__init__()
        self.num_freqs, self.d_in, self.freq_factor, self.include_input = num_freqs, d_in, freq_factor, include_input
        freqs = freq_factor * 2.0 ** torch.arange(0, num_freqs)
        self.d_out = num_freqs * 2 * d_in + (d_in if include_input else 0)
        self.register_buffer("_freqs", torch.repeat_interleave(freqs, 2).view(1, -1, 1))
        ph = torch.zeros(2 * num_freqs)
        ph[1::2] = np.pi * 0.5
        self.register_buffer("_phases", ph.view(1, -1, 1))

    @classmethod
    def from_conf(cls, conf, d_in=3):
        return cls(conf.get_int("num_freqs", 6), d_in, conf.get_float("freq_factor", np.pi),
                   conf.get_bool("include_input", True))


def _basic_block(cin, cout, stride):
    blk = nn.Module()
    blk.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
    blk.bn1 = nn.BatchNorm2d(cout)
    blk.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
    blk.bn2 = nn.BatchNorm2d(cout)
    if stride != 1 or cin != cout:
        blk.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
    return blk


def _resnet34_params():
    """Parameter container with the ResNet-34 key names the reference's checkpoints hold
    (`encoder.model.*`).  layer4 is carried for strict state_dict loading only."""
    m = nn.Module()
    m.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
    m.bn1 = nn.BatchNorm2d(64)
    cin = 64
    for li, (cout, n) in enumerate([(64, 3), (128, 4), (256, 6), (512, 3)], start=1):
        setattr(m, "layer%d" % li, nn.Sequential(
            *[_basic_block(cin if b == 0 else cout, cout, 2 if (b == 0 and li > 1) else 1) for b in range(n)]))
        cin = cout
    return m


class SpatialEncoder(nn.Module):
    """Configuration + parameters of the spatial encoder (reference src/model/encoder.py:13-77).
    backbone 'resnet34' runs in csrc/encoder.hip; backbone 'custom' (YOLOv7, whose source and
    weights are outside the reference tree) has no kernels: its output must be supplied through
    ``PixelNeRFNet.encode(..., latent=...)``."""

    def __init__(self, backbone="resnet34", pretrained=True, num_layers=4, index_interp="bilinear",
                 index_padding="border", upsample_interp="bilinear", feature_scale=1.0, use_first_pool=True,
                 norm_type="batch"):
        super().__init__()
        self.use_custom_resnet = backbone == "custom"
        if self.use_custom_resnet:
            self.latent_size = 1792  # reference custom_encoder.py:22
            self.model = nn.Module()
        else:
            if backbone != "resnet34" or num_layers != 4 or norm_type != "batch" or feature_scale != 1.0:
                raise NotImplementedError("libpnyolo's encoder kernels cover backbone=resnet34, num_layers=4, batch "
                                          "norm, feature_scale=1 (every shipped conf/exp/*.conf)")
            self.latent_size = [0, 64, 128, 256, 512, 1024][num_layers]
            self.model = _resnet34_params()
            # `pretrained` ImageNet weights cannot be downloaded here; they arrive via load_weights
        if index_interp != "bilinear" or index_padding != "zeros" or upsample_interp != "bilinear":
            raise NotImplementedError("libpnyolo supports bilinear indexing with zeros padding "
                                      "(conf/default.conf:49 of the reference)")
        self.num_layers = num_layers
        self.use_first_pool = use_first_pool
        self.index_interp, self.index_padding, self.upsample_interp = index_interp, index_padding, upsample_interp

    def forward_torch(self, x):
        """The trunk as a differentiable torch graph (reference encoder.py:110-173, resnet34 branch): the ATen alternative to the
        library's training trunk (csrc/encoder_train.hip, model._TrunkFunction), taken by ``PixelNeRFNet.encode`` with
        PNYOLO_TRUNK=torch or when the library cannot read the trunk's parameters in place.  Convolutions and batch norms
        (train-mode statistics when ``self.training``) run through ATen and autograd, and the renderer's backward hands
        d loss / d latent back to this graph (pny_scene_bind_latent_grad)."""
        import torch.nn.functional as F
        if self.use_custom_resnet:
            raise RuntimeError("backbone=custom has no trunk here: supply the latent")
        m = self.model

        def block(blk, t):
            out = torch.relu(blk.bn1(blk.conv1(t)))
            out = blk.bn2(blk.conv2(out))
            return torch.relu(out + (blk.downsample(t) if hasattr(blk, "downsample") else t))

        x = torch.relu(m.bn1(m.conv1(x)))
        latents = [x]
        if self.use_first_pool:
            x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        for layer in (m.layer1, m.layer2, m.layer3):
            for blk in layer:
                x = block(blk, x)
            latents.append(x)
        size = latents[0].shape[-2:]
        return torch.cat([F.interpolate(t, size, mode="bilinear", align_corners=True) for t in latents], dim=1)

    @classmethod
    def from_conf(cls, conf):
        return cls(conf.get_string("backbone"), pretrained=conf.get_bool("pretrained", True),
                   num_layers=conf.get_int("num_layers", 4), index_interp=conf.get_string("index_interp", "bilinear"),
                   index_padding=conf.get_string("index_padding", "border"),
                   upsample_interp=conf.get_string("upsample_interp", "bilinear"),
                   feature_scale=conf.get_float("feature_scale", 1.0),
                   use_first_pool=conf.get_bool("use_first_pool", True))


def make_mlp(conf, d_in, d_latent=0, allow_empty=False, **kwargs):
    """reference src/model/model_util.py:5-15"""
    mlp_type = conf.get_string("type", "mlp")
    if mlp_type == "resnet":
        return ResnetFC.from_conf(conf, d_in, d_latent=d_latent, **kwargs)
    if mlp_type == "empty" and allow_empty:
        return None
    raise NotImplementedError("Unsupported MLP type (libpnyolo implements type = resnet)")


def make_encoder(conf, **kwargs):
    """reference src/model/model_util.py:18-26"""
    if conf.get_string("type", "spatial") != "spatial":
        raise NotImplementedError("Unsupported encoder type")
    return SpatialEncoder.from_conf(conf, **kwargs)


# ------------------------------------------------------------------ the model
class PixelNeRFNet(nn.Module):
    """Drop-in for reference src/model/models.py:15 (constructor :16-90, encode :92-151,
    forward :153-318, load/save_weights :320-370)."""

    def __init__(self, conf, stop_encoder_grad=False):
        super().__init__()
        self._conf = conf                  # kept for replicas on other devices (render._MultiDeviceRenderWrapper)
        self._encode_epoch = 0             # bumped by every encode(): replicas re-fetch the scene state when it moved
        self._last_encode = None
        self.encoder = make_encoder(conf["encoder"])
        self.use_encoder = conf.get_bool("use_encoder", True)
        self.use_xyz = conf.get_bool("use_xyz", False)
        self.normalize_z = conf.get_bool("normalize_z", True)
        self.stop_encoder_grad = stop_encoder_grad
        self.use_code = conf.get_bool("use_code", False)
        self.use_code_viewdirs = conf.get_bool("use_code_viewdirs", True)
        self.use_viewdirs = conf.get_bool("use_viewdirs", False)
        self.use_global_encoder = conf.get_bool("use_global_encoder", False)
        if not (self.use_encoder and self.use_xyz and self.normalize_z and self.use_code and self.use_viewdirs
                and not self.use_code_viewdirs and not self.use_global_encoder):
            raise NotImplementedError(
                "libpnyolo implements the shipped flag set: use_encoder, use_xyz, normalize_z, use_code, "
                "use_viewdirs, not use_code_viewdirs, no global encoder (conf/default.conf of the reference)")
        d_latent = self.encoder.latent_size
        self.code = PositionalEncoding.from_conf(conf["code"], d_in=3)
        if not self.code.include_input:
            raise NotImplementedError("code.include_input = False is not supported")
        d_in = self.code.d_out + 3
        self.latent_size = self.encoder.latent_size
        self.mlp_coarse = make_mlp(conf["mlp_coarse"], d_in, d_latent)
        # assignable: eval.py:140 of the reference sets `net.mlp_fine = None` for coarse-only runs
        self.mlp_fine = make_mlp(conf["mlp_fine"], d_in, d_latent, allow_empty=True)
        self.yolo = conf.get_bool("mlp_coarse.yolo", False)
        self.d_in = d_in
        if not self.yolo:
            self.d_out = conf.get_int("mlp_coarse.d_out", 4)
        else:
            self.d_out = conf.get_int("mlp_coarse.d_out", 7) * conf.get_int("mlp_coarse.num_anchors_per_scale", 3)
        self.d_latent = d_latent
        self.num_objs = 0
        self.num_views_per_obj = 1
        # native handles (created lazily on the module's CUDA device)
        self._h_model = None
        self._h_device = None
        self._h_has_fine = False
        self._h_scenes = []
        # Super-batch in ONE scene (include/pnyolo.h pny_scene_set_groups): the training render of SB > 1 objects runs on this
        # handle -- one MLP launch per pass over every object's tiles.  _group = dict(SB, NS) while it holds the last encode();
        # the per-object handles are then filled on first use (_scenes_pending).
        self._h_group = None
        self._group = None
        self._scenes_pending = None
        self._last_call_group = False
        self._synced_key = None
        self._dev_bound = False
        self._timing = False
        # What a no-grad call does when one of its F16X2 launches met a value outside the f16 range (include/pnyolo.h
        # pny_model_range_status): 'relaunch' (default) = wait for the call, and if the guard fired repeat it on the fp32
        # kernels with a warning, keeping the scenes pinned to f32 from then on; 'raise' = wait and raise PnyRangeError;
        # 'lazy' = do not wait: the next library call on the model fails with PNY_ERR_RANGE (what training calls always do --
        # their launches overlap on side streams).  Env PNYOLO_F16_RANGE.
        self.f16_range_policy = os.environ.get("PNYOLO_F16_RANGE", "relaunch")
        self._projection = None  # None = library default (auto, or env PNYOLO_PROJECTION)
        self._precision = None   # None = library default (auto, or env PNYOLO_MLP_PRECISION)

    # ---------------------------------------------------------------- native plumbing
    def _device(self):
        dev = self.mlp_coarse.lin_in.weight.device
        if dev.type != "cuda":
            raise RuntimeError("PixelNeRFNet (libpnyolo) runs on an MI355X only: move the module to a cuda device "
                               "first (there is no CPU path)")
        return dev

    def _free_native(self):
        L = _lib.load()
        for s in self._h_scenes:
            L.pny_scene_destroy(s)
        self._h_scenes = []
        if getattr(self, "_h_group", None) is not None:
            L.pny_scene_destroy(self._h_group)
        self._h_group, self._group, self._scenes_pending = None, None, None
        if self._h_model is not None:
            L.pny_model_destroy(self._h_model)
            self._h_model = None
        self._dev_bound = False

    def __del__(self):
        try:
            self._free_native()
        except Exception:
            pass

    def _weights_key(self):
        """Identity of every parameter / buffer value: (storage pointer, in-place version counter).  Building the
        state_dict costs ~170 us per call, so the tensor list is cached and rebuilt only after a structural change
        anywhere (a parameter, buffer or submodule registered or replaced: PyTorch's global registration hooks bump
        _STRUCT_EPOCH) or when the fine MLP is attached / detached."""
        sig = (_STRUCT_EPOCH[0], id(self.mlp_fine))
        if sig != getattr(self, "_tracked_sig", None):
            self._tracked = list(self.state_dict(keep_vars=True).items())
            self._tracked_sig = sig
        return tuple((k, v.data_ptr(), v._version) for k, v in self._tracked)

    def _sync(self):
        """Create the native model on first use; re-upload weights when parameters changed
        (load_weights, load_state_dict, optimizer step).  Scenes survive a weight reload; a move to
        another device rebuilds everything."""
        L = _lib.load()
        dev = self._device()
        if self._h_model is not None and self._h_device != str(dev):
            self._free_native()
        key = self._weights_key()
        if self._h_model is None:
            mc = self.mlp_coarse
            has_fine = any(k[0].startswith("mlp_fine.") for k in key)
            desc = ModelDesc(d_latent=self.d_latent, d_hidden=mc.d_hidden, d_out=self.d_out, n_blocks=mc.n_blocks,
                             combine_layer=min(mc.combine_layer, 1 << 20), num_freqs=self.code.num_freqs,
                             freq_factor=float(self.code.freq_factor), yolo=int(self.yolo),
                             has_fine=int(has_fine), device=dev.index or 0,
                             enc_use_first_pool=int(getattr(self.encoder, "use_first_pool", True)))
            h = C.c_void_p()
            check(L.pny_model_create(C.byref(h), C.byref(desc)))
            self._h_model, self._h_device, self._h_has_fine, self._synced_key = h, str(dev), has_fine, None
        if self.mlp_fine is not None and not self._h_has_fine:
            raise RuntimeError("a fine MLP was attached after the native model was created without one")
        if key != self._synced_key:
            h = self._h_model
            old = self._synced_key
            # Only MLP tensors changed IN PLACE (same storage, new version: an optimizer step) -> device-side refresh:
            # one kernel launch re-creates the packed operands from the live parameters (pny_model_refresh).
            # Encoder tensors stepped in place (encoder training: the training trunk reads the live parameters itself) only mark
            # the INFERENCE trunk's folded copy stale; it is re-uploaded when that trunk is next needed (encode() outside training).
            in_place = (old is not None and self._dev_bound and len(old) == len(key) and
                        all(a[:2] == b[:2] and (a[2] == b[2] or a[0].startswith(("mlp_", "encoder."))) for a, b in zip(old, key)))
            if in_place:
                if any(a[2] != b[2] and a[0].startswith("encoder.") for a, b in zip(old, key)):
                    self._enc_stale = True
                check(L.pny_model_refresh(h, stream_of(dev)))
            else:
                for name, t in self.state_dict().items():
                    if name.endswith("num_batches_tracked"):
                        continue
                    a = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
                    shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
                    check(L.pny_model_load_weights(h, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))
                check(L.pny_model_finalize(h))
                self._enc_stale = False
                self._dev_bound = True
                self._trunk_bound = True
                for name, t in self.state_dict(keep_vars=True).items():
                    if name.startswith("mlp_"):
                        ok = t.dtype == torch.float32 and t.is_contiguous() and t.device == dev
                        self._dev_bound = self._dev_bound and ok
                        if ok:
                            check(L.pny_model_bind_param(h, name.encode(), C.c_void_p(t.data_ptr())))
                    elif name.startswith("encoder.model.") and not name.endswith("num_batches_tracked") \
                            and not name.startswith(("encoder.model.layer4", "encoder.model.fc")):
                        # the training-mode trunk (pny_trunk_train_forward / _backward) reads its parameters and steps the
                        # running statistics where PyTorch keeps them
                        ok = t.dtype == torch.float32 and t.is_contiguous() and t.device == dev
                        self._trunk_bound = self._trunk_bound and ok
                        if ok:
                            check(L.pny_model_bind_param(h, name.encode(), C.c_void_p(t.data_ptr())))
            self._synced_key = key
        # `net.mlp_fine = None` (reference eval.py:140): fine pass falls back to the coarse MLP
        check(L.pny_model_use_fine(self._h_model, int(self.mlp_fine is not None)))

    def _new_scene(self):
        L = _lib.load()
        s = C.c_void_p()
        check(L.pny_scene_create(C.byref(s), self._h_model))
        check(L.pny_scene_enable_timing(s, int(self._timing)))
        if self._projection is not None:
            check(L.pny_scene_set_projection(s, _lib.PROJECTION[self._projection]))
        if self._precision is not None:
            check(L.pny_scene_set_precision(s, _lib.PRECISION[self._precision]))
        return s

    def _scene(self, i):
        if self._scenes_pending is not None:
            self._materialise_scenes()
        while len(self._h_scenes) <= i:
            self._h_scenes.append(self._new_scene())
        return self._h_scenes[i]

    def _group_scene(self):
        """The handle that holds a whole super-batch (pny_scene_set_groups); None unless the last encode() filled it."""
        return self._h_group if self._group is not None else None

    def _handles(self):
        return list(self._h_scenes) + ([self._h_group] if self._h_group is not None else [])

    def _stat_handles(self):
        if self._last_call_group and self._group is not None:
            return [self._h_group]
        return self._h_scenes[: max(self.num_objs, 1)]

    @staticmethod
    def _cam_rows(t, sb, SB, NS):
        """focal / c rows of object sb: batch 1 broadcasts, batch SB is per object, batch SB*NS per view."""
        if t.shape[0] == 1:
            return t[:1]
        if t.shape[0] == SB * NS:
            return t[sb * NS:(sb + 1) * NS]
        if t.shape[0] == SB:
            return t[sb:sb + 1]
        raise ValueError("focal / c batch must be 1, SB or SB*NS")

    def _materialise_scenes(self):
        """Per-object scene handles from the grouped one (cameras from the encode() arguments, latent copied out of it): only
        when something other than the training render asks for them after a grouped encode()."""
        pend, self._scenes_pending = self._scenes_pending, None
        L = _lib.load()
        dev = self._device()
        SB, NS = self._group["SB"], self._group["NS"]
        dims = [C.c_int() for _ in range(4)]
        check(L.pny_scene_latent_shape(self._h_group, *[C.byref(d) for d in dims]))
        lat = torch.empty([d.value for d in dims], device=dev, dtype=torch.float32)
        check(L.pny_scene_get_latent(self._h_group, ptr(lat), stream_of(dev)))
        for sb in range(SB):
            s = self._scene(sb)
            f_s = self._cam_rows(pend["focal"], sb, SB, NS).contiguous()
            c_s = self._cam_rows(pend["c"], sb, SB, NS).contiguous()
            p_s = pend["poses"][sb * NS:(sb + 1) * NS].contiguous()
            check(L.pny_scene_set_cameras(s, ptr(p_s), NS, ptr(f_s), f_s.shape[0], ptr(c_s), c_s.shape[0], pend["W"], pend["H"]))
            part = lat[sb * NS:(sb + 1) * NS]
            check(L.pny_scene_set_latent(s, ptr(part), NS, part.shape[1], part.shape[2], part.shape[3], stream_of(dev)))

    def enable_kernel_timing(self, on=True):
        """HIP-event timing of the MLP launches (bench.py's roofline leg)."""
        self._timing = bool(on)
        for s in self._handles():
            check(_lib.load().pny_scene_enable_timing(s, int(on)))

    def set_latent_projection(self, mode):
        """'auto' | 'on' | 'off' (include/pnyolo.h pny_scene_set_projection): whether the fused kernel
        interpolates per-scene projected latent maps (lin_z applied once per latent pixel) or runs the
        lin_z GEMMs per sample in the reference's operation order."""
        if mode not in _lib.PROJECTION:
            raise ValueError("latent projection mode must be one of %s" % sorted(_lib.PROJECTION))
        self._projection = mode
        for s in self._handles():
            check(_lib.load().pny_scene_set_projection(s, _lib.PROJECTION[mode]))
        return self

    def set_matrix_precision(self, mode):
        """'auto' | 'f32' | 'f16x2' (include/pnyolo.h pny_scene_set_precision): the matrix arithmetic of projected
        launches -- fp32 MFMA, or fp32 operands split into two f16 planes on the f16 matrix cores (same measured
        error, 5.3x the matrix rate).  Launches without projection always run fp32."""
        if mode not in _lib.PRECISION:
            raise ValueError("matrix precision must be one of %s" % sorted(_lib.PRECISION))
        self._precision = mode
        for s in self._handles():
            check(_lib.load().pny_scene_set_precision(s, _lib.PRECISION[mode]))
        return self

    def range_status(self, clear=False):
        """Bits (include/pnyolo.h PNY_RANGE_*) the F16X2 kernels of this model have reported so far; does not synchronise."""
        if self._h_model is None:
            return 0
        v = C.c_uint(0)
        check(_lib.load().pny_model_range_status(self._h_model, C.byref(v), int(bool(clear))))
        return int(v.value)

    def check_f16_range(self):
        """Wait for the device and raise PnyRangeError if an F16X2 launch left the f16 range since the last clear."""
        if self._h_model is None:
            return
        torch.cuda.synchronize(self._device())
        bits = self.range_status(clear=True)
        if bits:
            raise _lib.PnyRangeError(self._range_message(bits))

    @staticmethod
    def _range_message(bits):
        what = ", ".join(n for b, n in sorted(_lib.RANGE_BITS.items()) if bits & b)
        return ("an F16X2 launch met a value outside the f16 range (%s: |x| >= 65520, infinite or NaN); its results are "
                "invalid.  Pin the fp32 kernels with net.set_matrix_precision('f32') (or PNYOLO_MLP_PRECISION=f32)" % what)

    def guard_f16_range(self, call):
        """Runs a no-grad render / query (`call`, repeatable) under `f16_range_policy` and returns its result.  The guard can
        fire in two places: inside the call, when the flag was already up when it entered the library (a weight repacked out
        of range by the refresh of this very call, a lazily reported earlier launch: PnyRangeError from the library), or
        after it, when one of its own F16X2 launches met the value (found by waiting for the stream)."""
        pol = self.f16_range_policy
        try:
            out = call()
            if pol == "lazy" or not any(self.last_launch_f16x2(i) for i in range(len(self._h_scenes))):
                return out
            torch.cuda.current_stream(self._device()).synchronize()
            bits = self.range_status(clear=False)
            if not bits:
                return out
        except _lib.PnyRangeError:
            if pol != "relaunch":
                raise
            bits = self.range_status(clear=False)
        self.range_status(clear=True)
        if pol == "raise":
            raise _lib.PnyRangeError(self._range_message(bits))
        warnings.warn("libpnyolo: " + self._range_message(bits) + " -- repeating the call on the fp32 kernels; this model "
                      "stays on them (f16_range_policy = 'relaunch')")
        self.set_matrix_precision("f32")
        return call()

    def last_launch_f16x2(self, scene=0):
        """True when the last MLP launch of scene `scene` ran the f16x2 kernel."""
        v = C.c_int(0)
        h = self._h_group if (self._last_call_group and self._group is not None) else self._scene(scene)
        check(_lib.load().pny_scene_last_precision(h, C.byref(v)))
        return bool(v.value)

    def project_latent(self):
        """Compute the projected maps of the encoded scenes now (otherwise done lazily by the first
        large render call after encode)."""
        self._sync()
        for i in range(self.num_objs):
            check(_lib.load().pny_scene_project(self._scene(i), stream_of(self._device())))
        return self

    def last_mlp_stats(self, full=False):
        """(executed GEMM FLOPs, kernel ms, launches) of the last call, summed over scenes; with
        full=True a dict that also carries the reference-order FLOP count and the projection flag."""
        L = _lib.load()
        fl = ref = ms = 0.0
        n = 0
        proj = False
        for s in self._stat_handles():
            a, r, b, c, p = C.c_double(), C.c_double(), C.c_double(), C.c_int(), C.c_int()
            check(L.pny_scene_last_mlp_stats(s, C.byref(a), C.byref(r), C.byref(b), C.byref(c), C.byref(p)))
            fl += a.value
            ref += r.value
            ms += b.value
            n += c.value
            proj = proj or bool(p.value)
        if full:
            return dict(flops=fl, flops_reference=ref, kernel_ms=ms, launches=n, projected=proj)
        return fl, ms, n

    def last_backward_stats(self):
        """dict(flops=[3], kernel_ms=[3]) of the last backward call, summed over scenes: stash forward, dX chain,
        weight-gradient GEMMs (include/pnyolo.h pny_scene_last_backward_stats)."""
        L = _lib.load()
        fl, ms = [0.0] * 3, [0.0] * 3
        for s in self._stat_handles():
            a, b = (C.c_double * 3)(), (C.c_double * 3)()
            check(L.pny_scene_last_backward_stats(s, a, b))
            for i in range(3):
                fl[i] += a[i]
                ms[i] += b[i]
        return dict(flops=fl, kernel_ms=ms)

    # ---------------------------------------------------------------- training plumbing
    def trainable_mlp_parameters(self):
        """[(state_dict name, Parameter)] of the MLP parameters that require grad, in a fixed order."""
        out = []
        for pre, mlp in (("mlp_coarse.", self.mlp_coarse), ("mlp_fine.", self.mlp_fine)):
            if mlp is not None:
                out += [(pre + k, p) for k, p in mlp.named_parameters() if p.requires_grad]
        return out

    def check_differentiable(self):
        """Called by a training render / query when the latent of the last encode() carries no graph.  Fine with a frozen
        encoder (reference: --freeze_enc / stop_encoder_grad, train/train.py:70-73, models.py:33-35); with a trainable one the
        gradient would silently stop at the latent, so this is an error: encode() must run in train() mode with grad enabled
        (the library's training trunk, model._TrunkFunction) or be given a latent that requires grad."""
        if not self.stop_encoder_grad and any(p.requires_grad for p in self.encoder.parameters()):
            raise NotImplementedError(
                "the encoder has trainable parameters but the latent of the last encode() carries no graph (encode() ran in "
                "eval() mode or under no_grad): call net.encode(...) in train() mode with grad enabled, or freeze the encoder "
                "(make_model(conf, stop_encoder_grad=True) / requires_grad_(False) on net.encoder, the reference's --freeze_enc)")

    def differentiable_latent(self):
        """The latent tensor of the last encode(latent=...) if it requires grad and autograd is recording in train() mode
        (the render / query backward then returns d loss / d latent for it), else None."""
        if torch.is_grad_enabled() and self.training:
            return getattr(self, "_latent_src", None)
        return None

    def begin_latent_grad(self, meta, n_scenes, group=False):
        """Zeroed (SB * NS, Hl, Wl, L) accumulator bound slice by slice to the scenes (pny_scene_bind_latent_grad), or whole to
        the grouped scene."""
        n_lat, l_ch, hl, wl = meta[0]
        buf = torch.zeros(n_lat, hl, wl, l_ch, device=self._device(), dtype=torch.float32)
        if group:
            check(_lib.load().pny_scene_bind_latent_grad(self._h_group, ptr(buf)))
            return buf
        nsv = n_lat // n_scenes
        for sb in range(n_scenes):
            check(_lib.load().pny_scene_bind_latent_grad(self._scene(sb), ptr(buf[sb * nsv:(sb + 1) * nsv])))
        return buf

    def end_latent_grad(self, buf, meta, n_scenes, group=False):
        """Unbind and return the gradient in the latent's own layout (SB * NS, L, Hl, Wl), device and dtype."""
        if group:
            check(_lib.load().pny_scene_bind_latent_grad(self._h_group, None))
        for sb in range(0 if group else n_scenes):
            check(_lib.load().pny_scene_bind_latent_grad(self._scene(sb), None))
        return buf.permute(0, 3, 1, 2).contiguous().to(meta[1], meta[2])   # packed NCHW, like the latent it belongs to

    def bind_mlp_grads(self):
        """Fresh zeroed gradient buffers for trainable_mlp_parameters(), bound to the native model by name
        (pny_model_bind_grad); returns them in the same order.  The buffers are views of ONE flat allocation (one fill
        instead of one per tensor); only the flat tensor is kept alive here, so that autograd's AccumulateGrad can adopt the
        returned views as `.grad` instead of cloning each of them."""
        self._sync()
        L = _lib.load()
        named = list(self.trainable_mlp_parameters())
        offs, total = [], 0
        for _, p in named:
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64          # 256-byte aligned starts (the reduction writes float4 rows)
        flat = torch.zeros(max(total, 1), device=self._device(), dtype=torch.float32)
        grads = []
        for (name, p), o in zip(named, offs):
            g = flat[o:o + p.numel()].view(p.shape)
            check(L.pny_model_bind_grad(self._h_model, name.encode(), ptr(g)))
            grads.append(g)
        self._bound_grads = flat   # keeps the memory alive while it is bound
        return grads

    def fork_streams(self, n):
        """n stream contexts for n independent scenes: the current stream for the first scene (None) and n - 1 side streams
        ordered behind it for the others (all None when there is a single scene or PNYOLO_SCENE_STREAMS=0).  The current
        stream takes part because the runtime maps streams onto FOUR hardware queues, one of which is the current stream's:
        with four side streams two scenes of the reference's default super-batch (SB = 4) shared a queue and ran one after
        the other while two queues idled (kernel trace of a training step)."""
        import os
        if n <= 1 or os.environ.get("PNYOLO_SCENE_STREAMS", "1") == "0":
            return [None] * n
        dev = self._device()
        pool = getattr(self, "_side_streams", None)
        if pool is None or len(pool) < n - 1 or pool[0].device != dev:
            pool = self._side_streams = [torch.cuda.Stream(dev) for _ in range(n - 1)]
        main = torch.cuda.current_stream(dev)
        for st in pool[:n - 1]:
            st.wait_stream(main)
        return [None] + pool[:n - 1]

    def join_streams(self, streams):
        main = torch.cuda.current_stream(self._device())
        for st in streams:
            if st is not None:
                main.wait_stream(st)

    def last_flush_stats(self):
        """(GEMM FLOPs, kernel ms) of the last deferred weight-gradient flush (pny_model_last_flush_stats)."""
        a, b = C.c_double(), C.c_double()
        check(_lib.load().pny_model_last_flush_stats(self._h_model, C.byref(a), C.byref(b)))
        return a.value, b.value

    def invalidate_weights(self):
        """Force a re-upload of the parameters at the next call.  Needed after writes that PyTorch does not version:
        ``p.data.copy_(w)`` / ``p.data.zero_()`` leave ``p._version`` and ``data_ptr()`` unchanged."""
        self._synced_key = None
        return self

    # ---------------------------------------------------------------- reference API
    def encode(self, images, poses, focal, z_bounds=None, c=None, latent=None):
        """
        :param images (NS, 3, H, W) or (SB, NS, 3, H, W), in [-1, 1]
        :param poses (NS, 4, 4) or (SB, NS, 4, 4) cam->world (YOLO mode: world->cam extrinsics)
        :param focal () or (N) or (N, 2);  :param c None or () or (N) or (N, 2)
        :param latent optional (SB*NS, L, Hl, Wl): encoder bypass (required for backbone=custom)
        """
        self._sync()
        L = _lib.load()
        dev = self._device()
        self.num_objs = images.size(0)
        if images.dim() == 5:
            assert poses.dim() == 4 and poses.size(1) == images.size(1)
            self.num_views_per_obj = images.size(1)
            images = images.reshape(-1, *images.shape[2:])
            poses = poses.reshape(-1, 4, 4)
        else:
            self.num_views_per_obj = 1
        NS, SB = self.num_views_per_obj, self.num_objs
        H, W = int(images.shape[-2]), int(images.shape[-1])
        # focal / principal point formats: reference models.py:125-148
        focal = torch.as_tensor(focal, dtype=torch.float32).cpu()
        if focal.dim() == 0:
            focal = focal[None, None].repeat(1, 2)
        elif focal.dim() == 1:
            focal = focal.unsqueeze(-1).repeat(1, 2)
        focal = focal.contiguous()
        if c is None:
            c = torch.tensor([[W * 0.5, H * 0.5]], dtype=torch.float32)
        else:
            c = torch.as_tensor(c, dtype=torch.float32).cpu()
            if c.dim() == 0:
                c = c[None, None].repeat(1, 2)
            elif c.dim() == 1:
                c = c.unsqueeze(-1).repeat(1, 2)
        c = c.contiguous()
        poses_h = poses.detach().to("cpu", torch.float32).contiguous()
        self._encode_epoch += 1
        self._last_encode = dict(poses=poses_h.reshape(SB, NS, 4, 4) if images is not None else poses_h, focal=focal, c=c, H=H, W=W,
                                 NS=NS, SB=SB, five_d=self.num_views_per_obj == NS)
        st = stream_of(dev)
        if latent is None and self.encoder.use_custom_resnet:
            raise RuntimeError("backbone=custom (YOLOv7) has no kernels in this build (its source and weights are "
                               "outside the reference tree): pass the backbone output via encode(..., latent=...)")
        # Encoder training (the reference's default: train/train.py without --freeze_enc): the trunk runs on the library's training
        # kernels under autograd (_TrunkFunction; the ATen graph forward_torch as the alternative) and its output enters like a
        # supplied latent; the render backward returns d loss / d latent to it
        if (latent is None and not self.encoder.use_custom_resnet and torch.is_grad_enabled() and self.training
                and not self.stop_encoder_grad and any(p.requires_grad for p in self.encoder.parameters())):
            if self._native_trunk_training():
                tp = self._trunk_trainable()
                latent = _TrunkFunction.apply(self, images.detach().to(dev, torch.float32).contiguous(), *[p for _, p in tp])
            else:   # (batch norm modules in eval() mode under autograd, or parameters the library cannot read in place)
                latent = self.encoder.forward_torch(images.to(dev, torch.float32))
        # a latent that requires grad (this trunk's, or a trainable encoder outside the library): the render backward returns
        # d loss / d latent for it (render._RenderFunction, pny_scene_bind_latent_grad)
        self._latent_src = latent if (torch.is_tensor(latent) and latent.requires_grad) else None
        if latent is not None:
            latent = latent.detach().to(dev, torch.float32).contiguous()
            assert latent.dim() == 4 and latent.shape[0] == SB * NS, "latent must be (SB*NS, L, Hl, Wl)"
        else:
            if getattr(self, "_enc_stale", False):   # the trunk's weights were trained since their last upload
                self._synced_key = None
                self._sync()
            images = images.detach().to(dev, torch.float32).contiguous()
        self._group, self._scenes_pending = None, None
        if (SB > 1 and SB * NS <= 16 and os.environ.get("PNYOLO_GROUP", "1") != "0" and torch.is_grad_enabled() and self.training
                and (self.trainable_mlp_parameters() or self._latent_src is not None)):
            # Training on a super-batch: ONE grouped scene (pny_scene_set_groups) holds every object's views, so that each MLP
            # pass of the render and of its backward is one launch over all objects' tiles (the reference flattens the
            # super-batch the same way, nerf.py:283-288).  The per-object handles are filled only if something asks for them.
            if self._h_group is None:
                self._h_group = self._new_scene()
            g = self._h_group

            def all_rows(t):
                return torch.cat([self._cam_rows(t, sb, SB, NS).expand(NS, 2) for sb in range(SB)], dim=0).contiguous()
            f_all, c_all = all_rows(focal), all_rows(c)
            check(L.pny_scene_set_cameras(g, ptr(poses_h), SB * NS, ptr(f_all), SB * NS, ptr(c_all), SB * NS, W, H))
            check(L.pny_scene_set_groups(g, SB))
            if latent is not None:
                check(L.pny_scene_set_latent(g, ptr(latent), SB * NS, latent.shape[1], latent.shape[2], latent.shape[3], st))
            else:
                check(L.pny_scene_encode(g, ptr(images), SB * NS, H, W, st))
            self._group = dict(SB=SB, NS=NS)
            self._scenes_pending = dict(poses=poses_h, focal=focal, c=c, W=W, H=H)
            return
        scenes = [self._scene(sb) for sb in range(SB)]
        # the library's trunk over a super-batch: ONE pass over all SB * NS images (as the reference's encode flattens them),
        # cameras per scene below
        batch_encode = latent is None and SB > 1
        if batch_encode:
            arr = (C.c_void_p * SB)(*[s_ for s_ in scenes])
            check(L.pny_scenes_encode(arr, SB, ptr(images), NS, H, W, stream_of(dev)))
        streams = [None] * SB if batch_encode else self.fork_streams(SB)   # independent scenes: one stream each
        for sb in range(SB):
          with torch.cuda.stream(streams[sb]):
            s = scenes[sb]
            st = stream_of(dev)
            f_s, c_s = self._cam_rows(focal, sb, SB, NS).contiguous(), self._cam_rows(c, sb, SB, NS).contiguous()
            p_s = poses_h[sb * NS:(sb + 1) * NS].contiguous()
            check(L.pny_scene_set_cameras(s, ptr(p_s), NS, ptr(f_s), f_s.shape[0], ptr(c_s), c_s.shape[0], W, H))
            if latent is not None:
                lat = latent[sb * NS:(sb + 1) * NS]
                check(L.pny_scene_set_latent(s, ptr(lat), NS, lat.shape[1], lat.shape[2], lat.shape[3], st))
            elif not batch_encode:
                img = images[sb * NS:(sb + 1) * NS]
                check(L.pny_scene_encode(s, ptr(img), NS, H, W, st))
        self.join_streams(streams)

    # ---------------------------------------------------------------- encoder training on the library's trunk
    def _trunk_bn_modules(self):
        m = self.encoder.model
        mods = [m.bn1]
        for layer in (m.layer1, m.layer2, m.layer3):
            for blk in layer:
                mods += [blk.bn1, blk.bn2] + ([blk.downsample[1]] if hasattr(blk, "downsample") else [])
        return mods

    def _native_trunk_training(self):
        """The training-mode trunk runs on the library's kernels (csrc/encoder_train.hip) when batch norm works on batch
        statistics (every BatchNorm2d of the trunk in train() mode, momentum set, affine) and the parameters are readable in
        place; PNYOLO_TRUNK=torch forces the ATen graph (SpatialEncoder.forward_torch)."""
        if os.environ.get("PNYOLO_TRUNK", "native") == "torch" or not getattr(self, "_trunk_bound", False):
            return False
        bns = self._trunk_bn_modules()
        return (all(b.momentum is not None and b.affine and b.track_running_stats for b in bns)
                and len({bool(b.training) for b in bns}) == 1)     # all on batch statistics, or all on the running ones

    def _trunk_trainable(self):
        """(state_dict name, parameter) of the trunk parameters that take a gradient (layer4 / fc are not part of the trunk)."""
        return [("encoder." + k, p) for k, p in self.encoder.named_parameters()
                if p.requires_grad and k.startswith("model.") and not k.startswith(("model.layer4", "model.fc"))]

    def latent(self, sb=0):
        """(NS, L, Hl, Wl) latent of scene `sb` as the reference keeps it in encoder.latent."""
        L = _lib.load()
        s = self._scene(sb)
        dims = [C.c_int() for _ in range(4)]
        check(L.pny_scene_latent_shape(s, *[C.byref(d) for d in dims]))
        out = torch.empty([d.value for d in dims], device=self._device(), dtype=torch.float32)
        check(L.pny_scene_get_latent(s, ptr(out), stream_of(out.device)))
        return out

    def forward(self, xyz, coarse=True, viewdirs=None, far=False):
        """
        Predict (r, g, b, sigma) at world space points xyz (after encode()).
        :param xyz (SB, B, 3);  :param viewdirs (SB, B, 3)  ->  (SB, B, d_out)
        """
        if torch.is_grad_enabled() and xyz.requires_grad:
            raise RuntimeError("libpnyolo does not differentiate w.r.t. the query points: detach xyz")
        # autograd path in train() mode only: eval-mode calls outside no_grad (the reference's eval scripts are not
        # consistent about it) return plain tensors, as before
        if torch.is_grad_enabled() and self.training:
            params, lat = self.trainable_mlp_parameters(), self.differentiable_latent()
            if params or lat is not None:
                if lat is None:
                    self.check_differentiable()
                return _QueryFunction.apply(self, xyz, bool(coarse), viewdirs, len(params), *[p for _, p in params],
                                            *([lat] if lat is not None else []))
        return self.guard_f16_range(lambda: self._query(xyz, coarse, viewdirs))

    def _query(self, xyz, coarse, viewdirs):
        self._sync()
        L = _lib.load()
        dev = self._device()
        SB, B, _ = xyz.shape
        assert SB == self.num_objs, "super-batch of xyz must match the encoded scenes"
        assert viewdirs is not None, "use_viewdirs is set: viewdirs are required"
        self._last_call_group = False
        xyz = xyz.detach().to(dev, torch.float32).contiguous()
        viewdirs = viewdirs.detach().to(dev, torch.float32).reshape(SB, B, 3).contiguous()
        out = torch.empty(SB, B, self.d_out, device=dev, dtype=torch.float32)
        use_coarse = bool(coarse) or self.mlp_fine is None
        st = stream_of(dev)
        for sb in range(SB):
            check(L.pny_query(self._scene(sb), ptr(xyz[sb]), ptr(viewdirs[sb]), B, int(use_coarse), ptr(out[sb]), st))
        return out

    # ---------------------------------------------------------------- checkpoints
    def load_weights(self, args, opt_init=False, strict=True, device=None):
        """Same contract as reference models.py:320-349.  The file loaded is
        <checkpoints_path>/<name>/pixel_nerf_init when ``opt_init or not args.resume`` and pixel_nerf_latest
        otherwise; ``opt_init and not args.resume`` is a no-op that returns None (the reference's bare ``return``);
        a missing file warns -- it does not fail -- unless opt_init.  The file is a plain state_dict and is read
        with weights_only=True."""
        if opt_init and not args.resume:
            return
        ckpt_name = "pixel_nerf_init" if opt_init or not args.resume else "pixel_nerf_latest"
        model_path = "%s/%s/%s" % (args.checkpoints_path, args.name, ckpt_name)
        if device is None:
            device = self.mlp_coarse.lin_in.weight.device   # the reference asks self.poses.device: same module device
        if osp.exists(model_path):
            print("Load", model_path)
            self.load_state_dict(torch.load(model_path, map_location=device, weights_only=True), strict=strict)
        elif not opt_init:
            warnings.warn(
                ("WARNING: {} does not exist, not loaded!! Model will be re-initialized.\n"
                 + "If you are trying to load a pretrained model, STOP since it's not in the right place. "
                 + "If training, unless you are startin a new experiment, please remember to pass --resume."
                 ).format(model_path))
        return self

    def save_weights(self, args, opt_init=False, epochNum=""):
        """Same contract as reference models.py:351-370: the current pixel_nerf_latest (pixel_nerf_init with
        opt_init) is first copied to pixel_nerf_backup<epochNum> (pixel_nerf_init_backup); the state_dict is written
        only when ``epochNum == ""`` -- the trainer's ``epochNum="_best"`` / ``str(epoch - 1)`` calls
        (train/trainlib/trainer.py:246,251) only snapshot the file already on disk."""
        from shutil import copyfile
        ckpt_name = "pixel_nerf_init" if opt_init else "pixel_nerf_latest"
        backup_name = "pixel_nerf_init_backup" if opt_init else "pixel_nerf_backup" + epochNum
        ckpt_path = osp.join(args.checkpoints_path, args.name, ckpt_name)
        ckpt_backup_path = osp.join(args.checkpoints_path, args.name, backup_name)
        if osp.exists(ckpt_path):
            copyfile(ckpt_path, ckpt_backup_path)
        if epochNum == "":
            torch.save(self.state_dict(), ckpt_path)
        return self


class _TrunkFunction(torch.autograd.Function):
    """The ResNet-34 trunk in training mode on the library's kernels (pny_trunk_train_forward / _backward, reference
    src/model/encoder.py:139-173 under autograd): images (n, 3, H, W) -> latent (n, 512, H/2, W/2); backward fills the
    gradients of the trainable trunk parameters.  Batch statistics, running statistics stepped as nn.BatchNorm2d does."""

    @staticmethod
    def forward(ctx, net, images, *params):
        net._sync()
        L = _lib.load()
        dev = net._device()
        n, _, H, W = images.shape
        hl, wl = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        lat = torch.empty(n, 512, hl, wl, device=dev, dtype=torch.float32)
        bns = net._trunk_bn_modules()
        bn_eval = not bns[0].training
        check(L.pny_trunk_train_forward(net._h_model, ptr(images), n, H, W, float(bns[0].momentum), int(bn_eval), ptr(lat),
                                        stream_of(dev)))
        if not bn_eval:
            with torch.no_grad():
                for b in bns:
                    b.num_batches_tracked += 1
            net._enc_stale = True     # the inference trunk's folded batch norm is stale now (running statistics moved)
        ctx.net = net
        ctx.names = [k for k, _ in net._trunk_trainable()]
        ctx.shapes = [tuple(p.shape) for p in params]
        return lat

    @staticmethod
    def backward(ctx, d_lat):
        net = ctx.net
        L = _lib.load()
        dev = net._device()
        h = net._h_model
        sizes = [int(np.prod(s)) for s in ctx.shapes]
        offs = np.concatenate([[0], np.cumsum([(s + 63) // 64 * 64 for s in sizes])]).astype(np.int64)
        d_lat = d_lat.detach().to(dev, torch.float32).contiguous()
        # The trunk's backward needs d loss / d latent and nothing else of the renderer's backward: it runs on a stream of its
        # own that starts behind the point where the latent gradient was complete (render._RenderFunction.backward records it
        # in front of the MLPs' weight-gradient flush), beside that flush; the caller's stream takes it back at the end.
        main = torch.cuda.current_stream(dev)
        side = getattr(net, "_trunk_stream", None)
        if os.environ.get("PNYOLO_TRUNK_STREAM", "1") == "0":
            side = main
        elif side is None or side.device != dev:
            side = net._trunk_stream = torch.cuda.Stream(dev)
        ev, net._lat_grad_event = getattr(net, "_lat_grad_event", None), None
        if side is not main:
            if ev is not None:
                side.wait_event(ev)
            else:
                side.wait_stream(main)
        with torch.cuda.stream(side):
            flat = torch.zeros(int(offs[-1]), device=dev, dtype=torch.float32)
            grads = [flat[int(o):int(o) + s].view(shp) for o, s, shp in zip(offs[:-1], sizes, ctx.shapes)]
            for k, g in zip(ctx.names, grads):
                check(L.pny_model_bind_grad(h, k.encode(), C.c_void_p(g.data_ptr())))
            try:
                check(L.pny_trunk_train_backward(h, ptr(d_lat), stream_of(dev)))
            finally:
                for k in ctx.names:
                    check(L.pny_model_bind_grad(h, k.encode(), None))
        if side is not main:
            d_lat.record_stream(side)
            flat.record_stream(main)
            main.wait_stream(side)
        return (None, None) + tuple(grads)


class _QueryFunction(torch.autograd.Function):
    """PixelNeRFNet.forward under autograd: pny_query forward, pny_query_backward into bound gradient buffers."""

    @staticmethod
    def forward(ctx, net, xyz, coarse, viewdirs, n_params, *params):
        lat = params[n_params] if len(params) > n_params else None
        ctx.lat_meta = None if lat is None else (tuple(lat.shape), lat.device, lat.dtype)
        out = net._query(xyz, coarse, viewdirs)
        dev = net._device()
        ctx.net, ctx.coarse = net, coarse
        ctx.xyz = xyz.detach().to(dev, torch.float32).contiguous()
        ctx.dirs = viewdirs.detach().to(dev, torch.float32).reshape(ctx.xyz.shape).contiguous()
        return out

    @staticmethod
    def backward(ctx, g_out):
        net = ctx.net
        L = _lib.load()
        dev = net._device()
        grads = net.bind_mlp_grads()
        g_out = g_out.detach().to(dev, torch.float32).contiguous()
        use_coarse = bool(ctx.coarse) or net.mlp_fine is None
        st = stream_of(dev)
        SB = ctx.xyz.shape[0]
        lat_grad = net.begin_latent_grad(ctx.lat_meta, SB) if ctx.lat_meta is not None else None
        for sb in range(SB):
            check(L.pny_query_backward(net._scene(sb), ptr(ctx.xyz[sb]), ptr(ctx.dirs[sb]), ctx.xyz.shape[1], int(use_coarse),
                                       ptr(g_out[sb]), 1, st))
        extra = () if lat_grad is None else (net.end_latent_grad(lat_grad, ctx.lat_meta, SB),)
        return (None, None, None, None, None) + tuple(grads) + extra


def make_model(conf, *args, **kwargs):
    """reference src/model/__init__.py:4-11"""
    model_type = conf.get_string("type", "pixelnerf")
    if model_type == "pixelnerf":
        return PixelNeRFNet(conf, *args, **kwargs)
    raise NotImplementedError("Unsupported model type", model_type)
