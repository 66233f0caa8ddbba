"""
Import helper: the package directory is named ``pixel-nerf-yolo_amd`` (a hyphen is not a
legal module name), so it is registered under the importable alias ``pixel_nerf_yolo_amd``.

    import pnyolo_pkg; pny = pnyolo_pkg.load()
    from pixel_nerf_yolo_amd import render      # works after load()
"""
import importlib.util
import os
import sys

ALIAS = "pixel_nerf_yolo_amd"
ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "pixel-nerf-yolo_amd")


def load():
    mod = sys.modules.get(ALIAS)
    if mod is not None:
        return mod
    spec = importlib.util.spec_from_file_location(
        ALIAS, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR]
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules[ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod
