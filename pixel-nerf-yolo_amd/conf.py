"""
Configuration values the hot path reads (SURVEY.md 5 "Config / flags").

The reference parses HOCON files with pyhocon (src/util/args.py:9-112); pyhocon is not a
dependency here.  ``Conf`` exposes the accessors the reference calls on a ConfigTree
(get_bool / get_int / get_float / get_string / get_list / []), so objects built from it have the
same ``from_conf`` constructors; a real pyhocon ConfigTree works in its place too.
``default_mv()`` / ``yolo()`` restate conf/default.conf + conf/default_mv.conf and
conf/exp/yolo.conf of the reference as Python dicts.
"""
import copy

_MISSING = object()


class Conf:
    def __init__(self, d):
        self.d = d

    def _get(self, key, default=_MISSING):
        cur = self.d
        for part in key.split("."):
            if isinstance(cur, dict) and part in cur:
                cur = cur[part]
            else:
                if default is _MISSING:
                    raise KeyError(key)
                return default
        return cur

    def __getitem__(self, key):
        v = self._get(key)
        return Conf(v) if isinstance(v, dict) else v

    def __contains__(self, key):
        return self._get(key, None) is not None

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


_MLP = {"type": "resnet", "n_blocks": 5, "d_hidden": 512, "d_out": 4, "combine_layer": 3, "combine_type": "average"}

_DEFAULT_MV = {
    "model": {
        "use_encoder": True, "use_global_encoder": False, "use_xyz": True, "canon_xyz": False,
        "use_code": True, "code": {"num_freqs": 6, "freq_factor": 1.5, "include_input": True},
        "use_viewdirs": True, "use_code_viewdirs": False,
        "mlp_coarse": dict(_MLP), "mlp_fine": dict(_MLP),
        "encoder": {"backbone": "resnet34", "pretrained": True, "num_layers": 4, "index_padding": "zeros"},
    },
    "renderer": {"type": "nerf", "n_coarse": 64, "n_fine": 32, "n_fine_depth": 16, "depth_std": 0.01, "sched": [],
                 "white_bkgd": True},
}


def default_mv():
    """conf/default.conf + conf/default_mv.conf of the reference (multi-view pixelNeRF)."""
    return Conf(copy.deepcopy(_DEFAULT_MV))


def yolo():
    """conf/exp/yolo.conf of the reference (YOLO renderer, custom backbone, no fine MLP)."""
    d = copy.deepcopy(_DEFAULT_MV)
    d["renderer"].update({"type": "yolo", "n_coarse": 128, "n_fine": 0, "white_bkgd": False, "eval_batch_size": 128})
    d["model"]["mlp_coarse"].update({"d_out": 7, "num_scales": 1, "num_anchors_per_scale": 3, "yolo": True})
    d["model"]["mlp_fine"] = {"type": "empty"}
    d["model"]["encoder"]["backbone"] = "custom"
    return Conf(d)


def sn64():
    """conf/exp/sn64.conf (and sn64_unseen.conf): default_mv with encoder.use_first_pool = False."""
    d = copy.deepcopy(_DEFAULT_MV)
    d["model"]["encoder"]["use_first_pool"] = False
    return Conf(d)


def dtu():
    """conf/exp/dtu.conf: default_mv with a black background."""
    d = copy.deepcopy(_DEFAULT_MV)
    d["renderer"]["white_bkgd"] = False
    return Conf(d)
