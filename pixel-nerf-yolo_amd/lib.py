"""
ctypes binding of libpnyolo.so (C ABI: include/pnyolo.h).

The library is built in-tree (``make -C pixel-nerf-yolo_amd/csrc``, driven by
``__graft_entry__.build()``) and loaded from this directory.  There is no CPU fallback: if the
shared object is missing or no MI355X is visible, calls raise.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# PNYOLO_LIB selects a diagnostic build (tools/stamp_build.sh); the product is libpnyolo.so
LIB_PATH = os.environ.get("PNYOLO_LIB") or os.path.join(_HERE, "libpnyolo.so")
CSRC = os.path.join(_HERE, "csrc")

c_float_p = C.POINTER(C.c_float)
c_i64_p = C.POINTER(C.c_int64)


class ModelDesc(C.Structure):
    _fields_ = [
        ("d_latent", C.c_int32), ("d_hidden", C.c_int32), ("d_out", C.c_int32), ("n_blocks", C.c_int32),
        ("combine_layer", C.c_int32), ("num_freqs", C.c_int32), ("freq_factor", C.c_float),
        ("yolo", C.c_int32), ("has_fine", C.c_int32), ("device", C.c_int32), ("enc_use_first_pool", C.c_int32),
    ]


class RenderOpts(C.Structure):
    _fields_ = [
        ("n_coarse", C.c_int32), ("n_fine", C.c_int32), ("n_fine_depth", C.c_int32), ("depth_std", C.c_float),
        ("white_bkgd", C.c_int32), ("lindisp", C.c_int32),
        ("u_coarse_dev", C.c_void_p), ("u_fine_dev", C.c_void_p), ("u_fine2_dev", C.c_void_p),
        ("g_depth_dev", C.c_void_p), ("seed", C.c_uint64),
        ("sigma_noise_coarse_dev", C.c_void_p), ("sigma_noise_fine_dev", C.c_void_p),
    ]


class RenderOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "rgb_coarse", "depth_coarse", "weights_coarse", "rgb_fine", "depth_fine", "weights_fine",
        "z_coarse", "z_fine", "sample_coarse", "sample_fine")]


class RenderSaved(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("z_coarse", "sample_coarse", "z_fine", "sample_fine", "depth_coarse")]


class RenderGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("rgb_coarse", "depth_coarse", "weights_coarse", "rgb_fine", "depth_fine",
                                          "weights_fine")]


# name -> (restype, argtypes); every symbol include/pnyolo.h declares
SIGNATURES = {
    "pny_version": (C.c_int, []),
    "pny_last_error": (C.c_char_p, []),
    "pny_model_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(ModelDesc)]),
    "pny_model_destroy": (None, [C.c_void_p]),
    "pny_model_load_weights": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, c_i64_p, C.c_int]),
    "pny_model_finalize": (C.c_int, [C.c_void_p]),
    "pny_model_use_fine": (C.c_int, [C.c_void_p, C.c_int]),
    "pny_model_bind_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    "pny_model_refresh": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pny_scene_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p]),
    "pny_scene_destroy": (None, [C.c_void_p]),
    "pny_scene_set_groups": (C.c_int, [C.c_void_p, C.c_int]),
    "pny_scene_set_cameras": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                        C.c_int, C.c_int]),
    "pny_scene_set_latent": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pny_scene_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pny_scenes_encode": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pny_scene_get_latent": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pny_scene_latent_shape": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_int)] * 4),
    "pny_gen_rays": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, C.c_float, C.c_float,
                               C.c_int, C.c_void_p, C.c_void_p]),
    "pny_gen_rays_range": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, C.c_float, C.c_float,
                                     C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "pny_query": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "pny_render": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(RenderOpts), C.POINTER(RenderOut),
                             C.c_void_p]),
    "pny_yolo_render": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "pny_sample_coarse": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p,
                                    C.c_void_p]),
    "pny_composite": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "pny_sample_fine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                  C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                  C.c_void_p, C.c_void_p]),
    "pny_yolo_aggregate": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pny_cells_to_bboxes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p]),
    "pny_nms": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                          C.c_void_p]),
    "pny_tp_fp_fn": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double,
                               C.c_void_p, C.c_void_p]),
    "pny_scene_set_projection": (C.c_int, [C.c_void_p, C.c_int]),
    "pny_scene_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "pny_scene_bind_latent_grad": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pny_scene_last_precision": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "pny_model_range_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint), C.c_int]),
    "pny_trunk_train_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "pny_trunk_train_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pny_scene_project": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pny_scene_last_mlp_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                           C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pny_scene_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "pny_scene_last_backward_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pny_yolo_render_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_void_p]),
    "pny_model_defer_weight_grads": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64]),
    "pny_scene_stash_next_render": (C.c_int, [C.c_void_p, C.c_int]),
    "pny_model_flush_weight_grads": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "pny_model_last_flush_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pny_model_bind_grad": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    "pny_query_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "pny_composite_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pny_render_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(RenderOpts), C.POINTER(RenderSaved),
                                      C.POINTER(RenderGrads), C.c_int, C.c_void_p]),
}

_lib = None
ABI_VERSION = 11
PROJECTION = {"off": 0, "on": 1, "auto": 2}
PRECISION = {"f32": 0, "f16x2": 1, "auto": 2}


RANGE_BITS = {1: "activation", 2: "gradient", 4: "weight"}   # include/pnyolo.h PNY_RANGE_*


class PnyError(RuntimeError):
    pass


class PnyRangeError(PnyError):
    """An F16X2 launch met a value outside the f16 range (include/pnyolo.h pny_model_range_status)."""


def build(verbose=False):
    """Compile libpnyolo.so for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise PnyError("building libpnyolo.so failed (see output above)")
    return LIB_PATH


def load():
    """dlopen libpnyolo.so and type every entry point.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PnyError(
            "libpnyolo.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the HIP path)" % LIB_PATH)
    # torch ships its own libamdhip64; it must be loaded FIRST so that libpnyolo.so binds to the
    # same HIP runtime instance -- streams and device pointers are shared across the boundary.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.pny_version() != ABI_VERSION:
        raise PnyError("libpnyolo.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc):
    if rc == -5:   # PNY_ERR_RANGE
        raise PnyRangeError("libpnyolo: %s (status %d)" % (load().pny_last_error().decode("utf-8", "replace"), rc))
    if rc != 0:
        raise PnyError("libpnyolo: %s (status %d)" % (load().pny_last_error().decode("utf-8", "replace"), rc))


def ptr(t):
    """Raw device / host pointer of a contiguous fp32 torch tensor (or None)."""
    if t is None:
        return None
    import torch
    assert t.dtype == torch.float32 and t.is_contiguous(), "libpnyolo takes contiguous fp32 tensors"
    return C.c_void_p(t.data_ptr())


def stream_of(device):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
