"""
Run-time guard of the F16X2 arithmetic's range (-m gpu; include/pnyolo.h pny_model_range_status).

The default matrix path splits every fp32 operand into two f16 planes: a relu output, a lin_in input or a weight of
magnitude >= 65520 has no f16 representation, and the kernel would return garbage where the reference (fp32 throughout,
src/model/resnetfc.py:134-186; it only prints when the output holds a NaN, src/model/models.py:174-270) still returns numbers.
Every test below feeds such a value and must get either the fp32 kernels' finite result (policy 'relaunch', the default
of no-grad calls) or a loud error ('raise', 'lazy', and every training call).
"""
import warnings

import numpy as np
import pytest
import torch

from helpers import DEV, dt, maxabs
from pixel_nerf_yolo_amd import conf as pconf
from pixel_nerf_yolo_amd import lib as plib
from pixel_nerf_yolo_amd import synth
from pixel_nerf_yolo_amd.model import make_model
from pixel_nerf_yolo_amd.render import NeRFRenderer

pytestmark = pytest.mark.gpu


def make_net(seed, ns=2, H=32, W=32, train=False, policy=None, scale_lin_in=1.0):
    c = pconf.default_mv()
    net = make_model(c["model"], stop_encoder_grad=True)
    for mlp, sd in ((net.mlp_coarse, synth.mlp_state(seed + 1)), (net.mlp_fine, synth.mlp_state(seed + 2))):
        sd = {k: torch.from_numpy(v) for k, v in sd.items()}
        sd["lin_in.weight"] = sd["lin_in.weight"] * scale_lin_in
        mlp.load_state_dict(sd)
    net = net.to(DEV)
    net = net.train() if train else net.eval()
    if policy is not None:
        net.f16_range_policy = policy
    poses, _ = synth.scene_cameras(ns)
    lat = synth.latent(seed + 3, ns, 512, H // 2, W // 2)
    net.encode(torch.zeros(1, ns, 3, H, W), torch.from_numpy(poses)[None], torch.tensor(0.9 * W), latent=torch.from_numpy(lat))
    return net


def points(n, seed, scale=0.5):
    rs = np.random.RandomState(seed)
    return (dt(rs.uniform(-scale, scale, size=(1, n, 3)).astype(np.float32)),
            dt(rs.standard_normal((1, n, 3)).astype(np.float32)))


def fp32_result(seed, xyz, vd, **kw):
    ref = make_net(seed, **kw).set_matrix_precision("f32")
    with torch.no_grad():
        out = ref(xyz, coarse=True, viewdirs=vd)
    assert not ref.last_launch_f16x2() and ref.range_status() == 0
    return out


def test_in_range_scene_reports_nothing():
    net = make_net(100)
    xyz, vd = points(300, 1)
    with torch.no_grad():
        out = net(xyz, coarse=True, viewdirs=vd)
    torch.cuda.synchronize()
    assert net.last_launch_f16x2() and net.range_status() == 0 and bool(torch.isfinite(out).all())


@pytest.mark.parametrize("case", ["input", "activation"])
def test_overflow_relaunches_on_fp32(case):
    """'input': a query point at 1e5 (lin_in's B operand leaves the f16 range in the prologue); 'activation': in-range inputs
    and weights, but relu(lin_in(x)) of a few 1e5 (the epilogue's guard).  Default policy: the call is repeated on the fp32
    kernels -- same numbers as a model pinned to f32 from the start, bit for bit -- with a warning, and stays there."""
    kw = dict(scale_lin_in=1.0) if case == "input" else dict(scale_lin_in=2000.0)
    xyz, vd = points(200, 2, scale=0.5)
    if case == "input":
        xyz[0, 7, 0] = 1.0e5
    else:
        xyz = xyz * 40.0          # |x| <= 20 (in range), lin_in.weight ~ N(0, 0.15) x 2000 -> |h| ~ 1e5
    net = make_net(200, **kw)
    with torch.no_grad(), pytest.warns(UserWarning, match="outside the f16 range"):
        out = net(xyz, coarse=True, viewdirs=vd)
    ref = fp32_result(200, xyz, vd, **kw)
    assert bool(torch.isfinite(out).all())
    assert torch.equal(out, ref)
    assert net.range_status() == 0 and not net.last_launch_f16x2()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")                       # pinned to f32 now: no second warning, no f16x2 launch
        out2 = net(xyz, coarse=True, viewdirs=vd)
    assert torch.equal(out2, ref)


def test_overflow_in_a_render_call_relaunches():
    """The same through NeRFRenderer (both passes, explicit draws replayed by the repeated call)."""
    ns, H, W, kc, kf, kfd, n = 2, 32, 32, 16, 8, 4, 64
    rs = np.random.RandomState(5)
    _, tgt = synth.scene_cameras(ns)
    import pnyolo_oracle as orc
    rays = orc.gen_rays(tgt[None], W, H, 0.9 * W, 0.8, 1.8)[0].reshape(-1, 8)[:n].clone()
    rays[3, 0] = 7.0e4                                        # one ray starts outside the f16 range
    draws = dict(u_coarse=rs.rand(n, kc).astype(np.float32), u_fine=rs.rand(n, kf - kfd).astype(np.float32),
                 u_fine2=rs.rand(n, kf - kfd).astype(np.float32), g_depth=rs.randn(n, kfd).astype(np.float32))
    outs = {}
    for prec in ("auto", "f32"):
        net = make_net(300).set_matrix_precision(prec)
        ren = NeRFRenderer(n_coarse=kc, n_fine=kf, n_fine_depth=kfd, white_bkgd=True).eval()
        ren.draws = draws
        with torch.no_grad(), warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            outs[prec] = ren(net, rays[None].to(DEV), want_weights=True)
        assert (len(w) == 1) == (prec == "auto"), [str(x.message) for x in w]
    for p in ("coarse", "fine"):
        for k in ("rgb", "depth", "weights"):
            assert bool(torch.isfinite(outs["auto"][p][k]).all())
            assert torch.equal(outs["auto"][p][k], outs["f32"][p][k]), (p, k)


def test_policy_raise_and_lazy():
    xyz, vd = points(100, 3)
    xyz[0, 0, 1] = -3.0e5
    net = make_net(400, policy="raise")
    with torch.no_grad(), pytest.raises(plib.PnyRangeError, match="activation"):
        net(xyz, coarse=True, viewdirs=vd)
    assert net.range_status() == 0                            # reported and cleared
    # lazy: the call itself returns; the NEXT library call fails until the flag is cleared
    net = make_net(400, policy="lazy")
    with torch.no_grad():
        net(xyz, coarse=True, viewdirs=vd)
        torch.cuda.synchronize()
        assert net.range_status() & 1
        with pytest.raises(plib.PnyRangeError, match="PNY_PRECISION_F32"):
            net(xyz, coarse=True, viewdirs=vd)
        with pytest.raises(plib.PnyRangeError):
            net.check_f16_range()                             # (also clears)
        net.set_matrix_precision("f32")
        out = net(xyz, coarse=True, viewdirs=vd)
    assert bool(torch.isfinite(out).all()) and torch.equal(out, fp32_result(400, xyz, vd))


def test_weight_leaves_the_range_in_an_optimizer_step():
    """pny_model_finalize checks the weights on the host; an in-place update (optimizer.step) is repacked on the device by
    pny_model_refresh, after the launch decision was made.  The repack kernel reports the weight; the no-grad call that
    follows is repeated on fp32, and AUTO stays on fp32 until the weights are finalized again."""
    net = make_net(500)
    xyz, vd = points(150, 4)
    with torch.no_grad():
        base = net(xyz, coarse=True, viewdirs=vd)
        assert net.last_launch_f16x2()
        net.mlp_coarse.blocks[1].fc_0.weight[5, 9] = 1.0e5          # in place: version bump -> device-side refresh
        with pytest.warns(UserWarning, match="weight"):
            out = net(xyz, coarse=True, viewdirs=vd)
    assert bool(torch.isfinite(out).all()) and not net.last_launch_f16x2()
    ref = make_net(500).set_matrix_precision("f32")
    with torch.no_grad():
        ref.mlp_coarse.blocks[1].fc_0.weight[5, 9] = 1.0e5
        assert torch.equal(out, ref(xyz, coarse=True, viewdirs=vd))
    assert not torch.equal(out, base)


def test_training_call_fails_loudly_on_the_next_call():
    """Training calls never wait for the device: a gradient that leaves the range (here an infinite upstream gradient) is
    reported by the chain kernel, and the next call on the model raises."""
    net = make_net(600, train=True)
    xyz, vd = points(128, 6)
    out = net(xyz, coarse=True, viewdirs=vd)
    g = torch.ones_like(out)
    g[0, 5, 1] = float("inf")
    out.backward(g)
    torch.cuda.synchronize()
    assert net.range_status() & 2
    with pytest.raises(plib.PnyRangeError, match="gradient"):
        net(xyz, coarse=True, viewdirs=vd)
    net.range_status(clear=True)
    net.zero_grad()
    out = net(xyz, coarse=True, viewdirs=vd)
    out.backward(torch.ones_like(out))
    torch.cuda.synchronize()
    assert net.range_status() == 0 and all(bool(torch.isfinite(p.grad).all()) for p in net.mlp_coarse.parameters())
