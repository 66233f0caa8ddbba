"""
Seeded synthetic inputs (weights, latents, cameras, rays) shared by the golden-fixture
generator (tools/make_golden.py), the parity tests and bench.py.

Everything is produced by numpy's legacy ``RandomState`` (bit-stable across numpy
versions), so a fixture only has to store a seed, not 27 MB of MLP weights.

Shapes / key names follow the reference state_dict (reference
src/model/resnetfc.py:66-132): ``lin_in.weight (H,d_in)``, ``lin_z.{i}.weight (H,L)``,
``blocks.{i}.fc_{0,1}.weight (H,H)``, ``lin_out.weight (d_out,H)``.
The reference's own init zeroes ``fc_1.weight`` and every bias (resnetfc.py:36-39), which
makes half of the network a no-op, so parity inputs perturb those as SURVEY.md 8c describes.
"""
import math

import numpy as np

D_IN = 42  # 3 + 3*2*6 positional code + 3 view dirs (reference src/model/models.py:49-60)


def mlp_state(seed, d_latent=512, d_hidden=512, d_out=4, n_blocks=5, combine_layer=3,
              d_in=D_IN, prefix="", out_gain=1.0):
    """Dict name -> float32 array for one ResnetFC (reference src/model/resnetfc.py:66-132).
    out_gain scales lin_out.weight (same random numbers): YOLO-mode raw outputs are unbounded, and with gain 1 they
    reach ~65, where an absolute 1e-4 is below fp32 resolution of the sum; 0.05 keeps them O(1)."""
    rs = np.random.RandomState(seed)
    sd = {}

    def lin(name, fan_out, fan_in, gain=1.0):
        std = gain * math.sqrt(2.0 / fan_in)
        sd[prefix + name + ".weight"] = (rs.standard_normal((fan_out, fan_in)) * std).astype(np.float32)
        sd[prefix + name + ".bias"] = (rs.standard_normal((fan_out,)) * 0.1).astype(np.float32)

    lin("lin_in", d_hidden, d_in)
    lin("lin_out", d_out, d_hidden, gain=out_gain)
    for i in range(n_blocks):
        lin("blocks.%d.fc_0" % i, d_hidden, d_hidden)
        lin("blocks.%d.fc_1" % i, d_hidden, d_hidden, gain=0.5)
    for i in range(min(combine_layer, n_blocks)):
        lin("lin_z.%d" % i, d_hidden, d_latent)
    return sd


def latent(seed, ns, channels, hl, wl):
    """Stand-in encoder output (NS, L, Hl, Wl) NCHW, N(0,1)*0.5 -- the shape SpatialEncoder stores
    (reference src/model/encoder.py:169)."""
    rs = np.random.RandomState(seed)
    return (rs.standard_normal((ns, channels, hl, wl)) * 0.5).astype(np.float32)


def images(seed, ns, h, w):
    """Source views in [-1, 1] (range of util.get_image_to_tensor_balanced, reference util.py:70-77)."""
    rs = np.random.RandomState(seed)
    return (rs.uniform(-1.0, 1.0, size=(ns, 3, h, w))).astype(np.float32)


def _rot_phi(phi):
    c, s = math.cos(phi), math.sin(phi)
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=np.float64)


def _rot_theta(th):
    c, s = math.cos(th), math.sin(th)
    return np.array([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1]], dtype=np.float64)


def pose_spherical(theta_deg, phi_deg, radius):
    """Camera-to-world pose on a sphere looking at the origin; the convention of the reference's
    util.pose_spherical (src/util/util.py:323-337) restated with numpy."""
    c2w = np.eye(4)
    c2w[2, 3] = radius
    c2w = _rot_phi(phi_deg / 180.0 * math.pi) @ c2w
    c2w = _rot_theta(theta_deg / 180.0 * math.pi) @ c2w
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)
    return (flip @ c2w).astype(np.float32)


def scene_cameras(ns, radius=1.3, phi=-20.0):
    """Source poses at theta = 0,40,80.. and one target pose at 120."""
    src = np.stack([pose_spherical(40.0 * i, phi, radius) for i in range(ns)])
    tgt = pose_spherical(120.0, phi, radius)
    return src, tgt


def resnet34_state(seed, prefix="encoder.model.", residual_gain=1.0):
    """Random ResNet-34 trunk parameters with the torchvision key names the reference
    checkpoint uses: conv1, bn1, layer1..layer3 (layer4/fc are unused by
    SpatialEncoder with num_layers=4, reference src/model/encoder.py:139-157).
    residual_gain scales every block's last batch-norm weight: eval-mode batch norm with random
    running statistics does not normalise, so with gain 1 the residual stream doubles in variance per
    block and the latent comes out O(100); 0.25 keeps it O(1) like a trained trunk's."""
    rs = np.random.RandomState(seed)
    sd = {}

    def conv(name, cout, cin, k):
        fan = cin * k * k
        sd[prefix + name + ".weight"] = (rs.standard_normal((cout, cin, k, k)) * math.sqrt(2.0 / fan)).astype(np.float32)

    def bn(name, c):
        sd[prefix + name + ".weight"] = rs.uniform(0.5, 1.5, size=(c,)).astype(np.float32)
        sd[prefix + name + ".bias"] = (rs.standard_normal((c,)) * 0.1).astype(np.float32)
        sd[prefix + name + ".running_mean"] = (rs.standard_normal((c,)) * 0.1).astype(np.float32)
        sd[prefix + name + ".running_var"] = rs.uniform(0.5, 1.5, size=(c,)).astype(np.float32)

    conv("conv1", 64, 3, 7)
    bn("bn1", 64)
    cin = 64
    for li, (cout, nblk) in enumerate([(64, 3), (128, 4), (256, 6)], start=1):
        for b in range(nblk):
            p = "layer%d.%d." % (li, b)
            conv(p + "conv1", cout, cin if b == 0 else cout, 3)
            bn(p + "bn1", cout)
            conv(p + "conv2", cout, cout, 3)
            bn(p + "bn2", cout)
            if residual_gain != 1.0:
                sd[prefix + p + "bn2.weight"] = (sd[prefix + p + "bn2.weight"] * residual_gain).astype(np.float32)
            if b == 0 and (cin != cout):
                conv(p + "downsample.0", cout, cin, 1)
                bn(p + "downsample.1", cout)
        cin = cout
    return sd
