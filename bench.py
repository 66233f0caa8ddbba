#!/usr/bin/env python3
"""
bench.py -- rays/sec of the pixelNeRF-YOLO rendering hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): 128x128 render, 3 source views, ResNet-34 encoder,
64 coarse + 32 fine samples (16 of them depth samples), white background; synthetic images,
seeded random weights of the reference architecture, cameras on a sphere.
A "step" = one full frame per rank from resident inputs (source images + 16384 rays in HBM ->
rgb/depth in HBM): scene encode (ResNet-34 trunk), latent projection (lin_z applied per latent pixel,
see include/pnyolo.h), coarse sampling, fused MLP, composite, importance sampling + sort, fused MLP,
composite, then (N > 1) one RCCL all-gather of the rendered (rays, 4) tiles.  Nothing is carried
over between steps: the per-scene state (latent, projected maps) is rebuilt inside every timed step
(the reference's eval loop encodes once per object and renders many views; this is the conservative
reading).  Weak scaling: every rank renders its own frame; value = all rays of all ranks /
max-over-ranks time.  `encode_ms` / `projection_ms` report the per-scene parts on their own.

One JSON line on rank 0 with the contract fields plus
  roofline:     dominant kernel = pny_mlp_kernel; achieved = the GEMM FLOPs the kernel executes
                (2/MAC, unpadded; DESIGN.md: 1.8668 GFLOP/ray with the projected latent, 2.6218
                GFLOP/ray -- SURVEY.md 8d -- in the reference's operation order, --projection off)
                / its HIP-event time measured inside libpnyolo on the launch stream; peak = 157.3
                TFLOP/s exact-fp32 MFMA (the dtype issued).  `reference_order_tflops` prices the
                same launches at the reference's FLOP count (not a utilisation figure).
  cpu_baseline: the oracle (oracle/pnyolo_oracle.py, a port) timed on the host cores on a bounded
                ray subset of the same frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "rays/sec (whole node), 64 samples/ray, 3-view 128×128 render"
H = W = 128
NS, KC, KF, KFD = 3, 64, 32, 16
FOCAL, Z_NEAR, Z_FAR = 131.25, 0.8, 1.8
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


D_LATENT = 512  # 1792 with --workload c3


def flop_per_ray(projected=False):
    """MLP GEMM FLOPs per ray: reference operation order, or with lin_z moved to the per-scene projection."""
    per_vs = 42 * 512 + (0 if projected else 3 * D_LATENT * 512) + 6 * 512 * 512
    post = 4 * 512 * 512 + 512 * 4
    per_sample = 2 * (NS * per_vs + post)
    return per_sample * (KC + (KC + KF))


WORKLOADS = {
    "c2": "C2: 128x128 render, 3 source views, ResNet34 encoder, 64 coarse + 32 fine (16 depth) samples, white bkgd",
    "c3": "C3: 128x128 render, 3 source views, YOLO-sized conditioning (L=1792 latent at 16x16 supplied through "
          "set_latent: the YOLOv7 backbone is outside the reference tree), 64 coarse + 32 fine (16 depth) samples",
}


def describe(workload="c2"):
    return {
        "metric": METRIC, "unit": "rays/s", "flop_per_ray": flop_per_ray(),
        "flop_per_ray_projected": flop_per_ray(True),
        "config": {"workload": WORKLOADS[workload], "rays_per_step_per_gpu": H * W},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-rays", type=int, default=1536, help="ray subset for the CPU baseline (0 = skip)")
    ap.add_argument("--projection", choices=["auto", "on", "off"], default="auto",
                    help="latent projection mode of the fused MLP (off = the reference's operation order)")
    ap.add_argument("--workload", choices=["c2", "c3"], default="c2",
                    help="c2 = BASELINE.json configs[1] (default, the bench line); c3 = configs[2], L=1792 conditioning")
    ap.add_argument("--describe", action="store_true", help="print the workload description and exit (no GPU)")
    args = ap.parse_args()
    global D_LATENT
    D_LATENT = 1792 if args.workload == "c3" else 512
    if args.describe:
        print(json.dumps(describe(args.workload)))
        return

    import numpy as np
    import torch
    import torch.distributed as dist

    import pnyolo_pkg
    pnyolo_pkg.load()
    from pixel_nerf_yolo_amd import conf as pconf, synth
    from pixel_nerf_yolo_amd.model import make_model
    from pixel_nerf_yolo_amd.render import NeRFRenderer
    from pixel_nerf_yolo_amd.util import gen_rays

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path for the product)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    # ---- scene: weights, encoder, cameras
    mconf = pconf.default_mv()
    if args.workload == "c3":
        mconf.d["model"]["encoder"]["backbone"] = "custom"   # d_latent = 1792 (reference custom_encoder.py:22)
    net = make_model(mconf["model"]).eval()
    sd = {}
    sd.update({"mlp_coarse." + k: v for k, v in synth.mlp_state(71, d_latent=D_LATENT).items()})
    sd.update({"mlp_fine." + k: v for k, v in synth.mlp_state(72, d_latent=D_LATENT).items()})
    if args.workload == "c2":
        sd.update(synth.resnet34_state(74, residual_gain=0.25))   # latent O(1), like a trained trunk
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    net = net.to(dev)
    net.set_latent_projection(args.projection)
    src, _ = synth.scene_cameras(NS)
    tgt = synth.pose_spherical(120.0 + 10.0 * rank, -20.0, 1.3)
    images = torch.from_numpy(synth.images(75, NS, H, W)).to(dev)
    focal, c = torch.tensor(FOCAL), torch.tensor([[W * 0.5, H * 0.5]])
    poses = torch.from_numpy(src)[None]

    lat_in = torch.from_numpy(synth.latent(76, NS, 1792, 16, 16)).to(dev) if args.workload == "c3" else None

    def encode():
        net.encode(images[None], poses, focal, c=c, latent=lat_in)   # c3: the backbone's output is an input

    encode()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        encode()
    torch.cuda.synchronize()
    encode_ms = (time.perf_counter() - t0) / 3 * 1e3
    projection_ms = None
    if args.projection != "off":
        net.project_latent()
        projection_ms = 0.0
        for _ in range(3):
            encode()                      # a new latent invalidates the projected maps
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            net.project_latent()          # coarse + fine maps
            torch.cuda.synchronize()
            projection_ms += (time.perf_counter() - t0) / 3 * 1e3

    rays = gen_rays(torch.from_numpy(tgt)[None].to(dev), W, H, focal, Z_NEAR, Z_FAR, c=c[0]).reshape(1, -1, 8)
    n_rays = rays.shape[1]
    ren = NeRFRenderer(n_coarse=KC, n_fine=KF, n_fine_depth=KFD, depth_std=0.01, white_bkgd=True).eval()
    par = ren.bind_parallel(net, None, simple_output=True).eval()
    gathered = torch.empty(world * n_rays, 4, device=dev) if world > 1 else None

    def step():
        encode()                          # per-scene state is rebuilt inside the step (module docstring)
        with torch.no_grad():
            rgb, depth = par(rays)
        if world > 1:
            tile = torch.cat([rgb[0], depth[0][:, None]], dim=1)
            dist.all_gather_into_tensor(gathered, tile)
        return rgb, depth

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # one-time setup that is not a step (so that --warmup 0 does not time it): RCCL creates its communicator on
    # the first collective, and HIP loads code objects / sets function attributes on a kernel's first launch
    with torch.no_grad():
        par(rays[:, :256].contiguous())   # 8x32 and (fine pass) 8x64 shapes, projected variant when enabled
        par(rays[:, :4096].contiguous())
    if world > 1:
        dist.all_gather_into_tensor(gathered, torch.zeros(n_rays, 4, device=dev))
    fence()

    for _ in range(args.warmup):
        step()
    net.enable_kernel_timing(True)
    fence()
    t0 = time.perf_counter()
    kern_ms = 0.0
    kern_flops = 0.0
    ref_flops = 0.0
    launches = 0
    projected = False
    for _ in range(args.steps):
        rgb, depth = step()
        # HIP-event times of this step's MLP launches (recorded on the launch stream inside
        # libpnyolo; reading them waits for the step, which the next step depends on anyway)
        st = net.last_mlp_stats(full=True)
        kern_flops += st["flops"]
        ref_flops += st["flops_reference"]
        kern_ms += st["kernel_ms"]
        launches += st["launches"]
        projected = projected or st["projected"]
    fence()
    elapsed = time.perf_counter() - t0
    net.enable_kernel_timing(False)
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(depth).all())

    # HBM-side traffic of the dominant kernel cannot be counted from inside this process: it is the
    # committed rocprofv3 PMC measurement of the same command (tools/profile_gpu.sh -> profiles/mlp_traffic.json)
    traffic = traffic_note = None
    try:
        with open(os.path.join(ROOT, "profiles", "mlp_traffic.json")) as fh:
            tj = json.load(fh)
        if bool(tj.get("projected_latent")) == bool(projected) and args.workload == "c2":
            traffic, traffic_note = tj["bytes_per_launch"], "%s: %s" % (tj["tag"], tj["method"])
    except (OSError, ValueError, KeyError):
        pass

    total_rays = world * n_rays * args.steps
    value = total_rays / elapsed
    achieved = kern_flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else None
    out = {
        "metric": METRIC, "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": dict(describe(args.workload)["config"], n_views=NS, n_coarse=KC, n_fine=KF, n_fine_depth=KFD,
                       global_rays_per_step=world * n_rays,
                       parallelism="rays sharded, 1 process/GPU, dp%d, 1 all-gather/frame" % world),
        "encode_ms": encode_ms, "projection_ms": projection_ms,
        "flop_per_ray": flop_per_ray(projected), "flop_per_ray_reference_order": flop_per_ray(False),
        "roofline": {
            "bound": "mfma", "kernel": "pny_mlp_kernel", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": (achieved / PEAK_F32_MFMA_TFLOPS) if achieved else None,
            "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_note, "launches": launches,
            "avg_launch_ms": (kern_ms / launches) if launches else None,
            "flops_per_launch": (kern_flops / launches) if launches else None,
            "projected_latent": projected,
            "reference_order_tflops": (ref_flops / (kern_ms * 1e-3) / 1e12) if kern_ms > 0 else None,
        },
    }

    if rank == 0 and args.cpu_rays > 0 and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pnyolo_oracle as orc  # CPU baseline leg: the oracle as the thing timed beside the GPU
        lat = net.latent(0).cpu().numpy()
        sc = orc.Scene(synth.mlp_state(71, d_latent=D_LATENT), synth.mlp_state(72, d_latent=D_LATENT), lat, src, focal, c,
                       W, H)
        nb = args.cpu_rays
        rs = np.random.RandomState(0)
        sub = rays[0, torch.from_numpy(rs.choice(n_rays, nb, replace=False)).to(dev)].cpu()
        draws = (rs.rand(nb, KC).astype(np.float32), rs.rand(nb, KF - KFD).astype(np.float32),
                 rs.rand(nb, KF - KFD).astype(np.float32), rs.randn(nb, KFD).astype(np.float32))
        # the box's CPU share for one GPU is 16 cores (more threads only oversubscribe the small GEMMs)
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        t0 = time.perf_counter()
        cpu_out = orc.render(sc, sub, KC, KF, KFD, *draws, chunk=50000)
        cpu_s = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": nb / cpu_s, "unit": "rays/s", "cores": torch.get_num_threads(),
                               "kind": "port", "sample": "%d random rays of the same %s frame, %.1f s" % (nb, args.workload.upper(), cpu_s)}
        # the same rays and draws through the HIP path: the checker's output beside the product's (not timed)
        ren.draws = dict(u_coarse=draws[0], u_fine=draws[1], u_fine2=draws[2], g_depth=draws[3])
        with torch.no_grad():
            gpu_out = ren(net, sub[None].to(dev))
        ec = (gpu_out["coarse"]["rgb"][0].cpu() - cpu_out["coarse"]["rgb"]).abs().max(dim=1)[0]
        ef = (gpu_out["fine"]["rgb"][0].cpu() - cpu_out["fine"]["rgb"]).abs().max(dim=1)[0]
        out["parity"] = {"rays": nb, "tolerance": 1e-4, "coarse_rgb_max_abs_err": float(ec.max()),
                         "fine_rgb_median_abs_err": float(ef.median()),
                         "fine_rays_over_tolerance": int((ef > 1e-4).sum()),
                         "note": "fine rays over tolerance are importance-sampling bin flips (discontinuous in the "
                                 "coarse weights, DESIGN.md section 2)"}
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
