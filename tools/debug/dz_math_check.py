"""CPU check of the formulas of mlp_dz_kernel (mlp_bwd.hip) against torch.autograd through the oracle's ops."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import pnyolo_pkg; pnyolo_pkg.load()
import pnyolo_oracle as orc
from pixel_nerf_yolo_amd import synth

torch.manual_seed(0)
ns, H, W, L = 2, 32, 32, 512
sd = {k: torch.from_numpy(v) for k, v in synth.mlp_state(191).items()}
lat = torch.from_numpy(synth.latent(193, ns, L, 16, 16))
poses, tgt = synth.scene_cameras(ns)
sc = orc.Scene(sd, None, lat, poses, torch.tensor(33.0), torch.tensor([[16.0, 16.0]]), W, H)
rays = orc.gen_rays(tgt[None], W, H, 33.0, 0.8, 1.8)[0].reshape(-1, 8)[[100, 500, 700]]
z = torch.tensor([1.0, 1.2, 0.9], requires_grad=True)
G = torch.randn(3, 4)

# --- forward with intermediates (copied from orc.query / resnetfc)
o, d = rays[:, :3], rays[:, 3:6]
p = o + z[:, None] * d
R, t = sc.w2c[:, :, :3], sc.w2c[:, :, 3]
xr = torch.matmul(R[:, None], p[None, :, :, None])[..., 0]; xr.retain_grad()
xc = xr + t[:, None]
code = orc.positional_encoding(xr.reshape(-1, 3))
vd = torch.matmul(R[:, None], d[None, :, :, None])[..., 0].reshape(-1, 3)
x_in = torch.cat((code, vd), dim=1); x_in.retain_grad()
uv = -xc[:, :, :2] / xc[:, :, 2:]
uv = uv * sc.focal.expand(ns, 2)[:, None] + sc.c.expand(ns, 2)[:, None]; uv.retain_grad()
zl = orc.index_latent(sc.latent, uv, W, H).reshape(ns * 3, -1)
h = orc._lin(sd, "lin_in", x_in)
dh_in = []
for blk in range(5):
    if blk == 3:
        h = h.reshape(ns, -1, 512).mean(dim=0)
    if blk < 3:
        h = h + orc._lin(sd, "lin_z.%d" % blk, zl)
        h.retain_grad(); dh_in.append(h)
    net = orc._lin(sd, "blocks.%d.fc_0" % blk, torch.relu(h))
    h = h + orc._lin(sd, "blocks.%d.fc_1" % blk, torch.relu(net))
out = orc._lin(sd, "lin_out", torch.relu(h))
out = torch.cat([torch.sigmoid(out[:, :3]), torch.relu(out[:, 3:4])], -1)
(out * G).sum().backward()
print("autograd dL/dz", z.grad.numpy())

# --- the kernel's formulas
Wl = Hl = 16
sx = (Wl / (Wl - 1.0) * 2.0) / W; sy = (Hl / (Hl - 1.0) * 2.0) / H
dz = np.zeros(3)
for v in range(ns):
    for s in range(3):
        row = v * 3 + s
        dh0 = dh_in[0].grad[row].numpy()
        gin = sd["lin_in.weight"].numpy().T @ dh0                    # (42,)
        print("  g_in vs autograd", np.abs(gin - x_in.grad[row].numpy()).max())
        xr_ = xr[v, s].detach().numpy(); xc_ = xc[v, s].detach().numpy()
        gx = gin[:3].copy()
        for e in range(3, 39):
            i = e - 3; fi, ph, dim = i // 6, (i // 3) & 1, i % 3
            freq = 1.5 * 2 ** fi
            gx[dim] += gin[e] * math.cos((math.pi / 2 if ph else 0.0) + xr_[dim] * freq) * freq
        fx, fy, cx, cy = float(sc.focal[0, 0]), float(sc.focal[0, 1]), 16.0, 16.0
        ux = -xc_[0] / xc_[2] * fx + cx; uy = -xc_[1] / xc_[2] * fy + cy
        ix = ((ux * sx - 1 + 1) / 2) * (Wl - 1); iy = ((uy * sy - 1 + 1) / 2) * (Hl - 1)
        x0, y0 = math.floor(ix), math.floor(iy); x1, y1 = x0 + 1, y0 + 1
        six = siy = 0.0
        latv = sc.latent[v].numpy()                                   # (L, Hl, Wl)
        for b in range(3):
            dhb = dh_in[b].grad[row].numpy()
            Wz = sd["lin_z.%d.weight" % b].numpy()
            def zp(x, y):
                if 0 <= x <= Wl - 1 and 0 <= y <= Hl - 1:
                    return Wz @ latv[:, y, x]
                return np.zeros(512)
            nw, ne, sw, se = zp(x0, y0), zp(x1, y0), zp(x0, y1), zp(x1, y1)
            six += dhb @ (-(y1 - iy) * nw + (y1 - iy) * ne - (iy - y0) * sw + (iy - y0) * se)
            siy += dhb @ (-(x1 - ix) * nw - (ix - x0) * ne + (x1 - ix) * sw + (ix - x0) * se)
        dux = six * sx * (Wl - 1) * 0.5; duy = siy * sy * (Hl - 1) * 0.5
        print("  duv vs autograd", dux - float(uv.grad[v, s, 0]), duy - float(uv.grad[v, s, 1]), float(uv.grad[v, s, 0]))
        sgn = -1.0
        inv = 1.0 / xc_[2]
        gx[0] += dux * sgn * fx * inv; gx[1] += duy * sgn * fy * inv
        gx[2] += -(dux * sgn * fx * xc_[0] + duy * sgn * fy * xc_[1]) * inv * inv
        print("  dxr vs autograd", np.abs(gx - xr.grad[v, s].numpy()).max())
        Rv = R[v].numpy()
        dp = Rv.T @ gx
        dz[s] += float(d[s].numpy() @ dp)
print("kernel formulas dL/dz", dz)
