import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pnyolo_pkg  # noqa: E402

pnyolo_pkg.load()

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "f16x2_forward: backward test that keeps the default (f16x2) training forward")
    config.addinivalue_line("markers", "one_backward_leg: the test does not depend on the backward's matrix arithmetic (default leg only)")


def pytest_collection_modifyitems(config, items):
    """The tests of the shipped DEFAULT training arithmetic exist only under the split-f16 backward: their combination with the
    module fixture's fp32-backward leg is not a test (it used to be collected and skipped)."""
    keep, drop = [], []
    for it in items:
        f32_leg = "[dw_f32" in it.nodeid
        na = f32_leg and "default_arithmetic" in it.nodeid
        # tests whose subject does not depend on the backward's matrix arithmetic (no MLP backward in them, or HIP against HIP
        # on the same arithmetic) run under the default leg only: marker one_backward_leg
        na = na or (f32_leg and it.get_closest_marker("one_backward_leg") is not None)
        # the per-object launch forms of the super-batch test: both legs for the default form ("group"), one for the others
        na = na or (f32_leg and "test_render_backward_super_batch" in it.nodeid and not it.nodeid.endswith("-group]"))
        (drop if na else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
                cache[name] = {k: f[k] for k in f.files}
        return cache[name]

    return load
