#!/usr/bin/env python3
"""
bench.py -- rays/sec of the pixelNeRF-YOLO rendering hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 run plainly (no launcher) starts the N ranks itself: before anything touches the GPU this
process spawns `python -m torch.distributed.run ...` as a CHILD (never an exec) and relays rank 0's JSON line.

Workloads (BASELINE.json configs; synthetic images, seeded random weights of the reference architecture, cameras on
a sphere):
  c2 (default, the bench line)  128x128, 3 source views, ResNet-34 encoder (HIP trunk inside every step), 64 + 32 (16)
  c3                            the same with L = 1792 "custom"-backbone conditioning (latent supplied: the YOLOv7
                                backbone is outside the reference tree)
  c4                            ONE 400x400 frame, L = 1792, 128 + 64 (32): the frame's rays are sharded over the
                                ranks (each rank generates its own ray range on its device), one RCCL all-gather
                                of (rays/G, 4) tiles -> STRONG scaling (north_star's multi-GPU case)
  c5                            8-scene eval batch, one scene per GPU (each rank encodes and renders its own C3
                                scene, no exchange) -> weak scaling
A "step" = one full frame from resident inputs (source images / latent + camera parameters in HBM -> rgb/depth in
HBM): scene encode, latent projection, coarse sampling, fused MLP, composite, importance sampling + sort, fused MLP,
composite, then (N > 1, c2/c3/c4) one all-gather of the rendered tiles.  Nothing is carried over between steps: the
per-scene state is rebuilt inside every timed step (the reference encodes once per object and renders many views;
this is the conservative reading).  c2/c3/c5: every rank renders its own frame; value = all rays of all ranks /
max-over-ranks time.  c4: value = the frame's rays / max-over-ranks time.

One JSON line on rank 0 with the contract fields plus
  roofline      dominant kernel = the fused MLP kernel.  `achieved` = the fp32 GEMM FLOPs the launched kernel variant
                EXECUTES (2/MAC, unpadded; with the projected latent lin_z has left the kernel: 1.8668 GFLOP/ray at C2
                instead of SURVEY.md 8d's 2.6218) / its HIP-event time measured inside libpnyolo on the launch stream.
                Default matrix path (pny_mlp_h2_kernel): every fp32 operand is split into two f16 planes and a product is
                three v_mfma_f32_32x32x16_f16 with fp32 accumulation (same measured error as the fp32 MFMA; same goldens,
                same 1e-4 bar): the bound is the f16 matrix pipe at three issued MFMA FLOPs per fp32 GEMM FLOP,
                peak = 2516.8 / 3 = 838.9 TFLOP/s of fp32 GEMM work; `frac` = achieved / peak = the fraction of the
                f16 matrix pipe's time the kernel keeps it busy.  With --precision f32 (pny_mlp_kernel,
                v_mfma_f32_32x32x2_f32) peak = 157.3 TFLOP/s.  `frac_survey_formula` = rays/s x SURVEY.md 8d's FLOP/ray
                / (peak x GPUs), the formula of the scope table (it counts the lin_z work the projection removed).
  fp32_matrix_path           (N = 1) the same frame on the exact-fp32 MFMA kernel: rays/s, ms per launch, frac of 157.3.
  roofline_reference_order   (N = 1) the same frame timed with `--projection off`, i.e. the reference's operation
                order where executed FLOPs = SURVEY.md 8d's count: achieved / frac / ms per launch / rays per s.
  cpu_baseline  the oracle (oracle/pnyolo_oracle.py, a port) timed on the host cores on a bounded ray subset.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "rays/sec (whole node), 64 samples/ray, 3-view 128×128 render"
NS = 3
FOCAL128, Z_NEAR, Z_FAR = 131.25, 0.8, 1.8
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 16 * 157.3  # same table: the f32 matrix rate is 1/16 of the dense BF16/F16 rate (~2.5 PF)
PEAK_F16X2_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3  # split operands: x1 w1 + x2 w1 + x1 w2 = 3 issued MFMA FLOPs per fp32 FLOP

# name -> (image side, d_latent, latent side, n_coarse, n_fine, n_fine_depth, scaling, description)
WORKLOADS = {
    "c2": (128, 512, 64, 64, 32, 16, "weak",
           "C2: 128x128 render, 3 source views, ResNet34 encoder, 64 coarse + 32 fine (16 depth) samples, white bkgd"),
    "c3": (128, 1792, 16, 64, 32, 16, "weak",
           "C3: 128x128 render, 3 source views, YOLO-sized conditioning (L=1792 latent at 16x16 supplied through "
           "set_latent: the YOLOv7 backbone is outside the reference tree), 64 coarse + 32 fine (16 depth) samples"),
    "c4": (400, 1792, 50, 128, 64, 32, "strong",
           "C4: ONE 400x400 frame, 3 source views, L=1792 conditioning (latent at 50x50 supplied), 128 coarse + 64 fine "
           "(32 depth) samples, the frame's rays sharded across the ranks (per-rank ray generation), one all-gather of "
           "rendered tiles"),
    "c5": (128, 1792, 16, 64, 32, 16, "weak",
           "C5: 8-scene eval batch, one scene per GPU (scene r on rank r: own latent, own cameras, own 128x128 frame), "
           "L=1792 conditioning, 64 coarse + 32 fine (16 depth) samples, no exchange"),
}


def kernel_source_fingerprint():
    """sha256 over the sources of the dominant kernel (csrc/mlp_h2*.hip/.h, mlp_core.h): stored with the committed PMC traffic
    figure (profiles/mlp_traffic.json, tools/summarize_prof.py) and recomputed here, so that a traffic figure profiled on
    other kernel sources than the ones benched is marked stale in the line instead of passing silently."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pixel-nerf-yolo_amd", "csrc")
    for f in ("mlp_h2.hip", "mlp_h2_core.h", "mlp_core.h", "pny_common.h"):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def flop_per_ray(workload, projected=False):
    """MLP GEMM FLOPs per ray (2/MAC, unpadded): reference operation order, or with lin_z moved to the
    per-scene projection."""
    _, L, _, kc, kf, _, _, _ = WORKLOADS[workload]
    per_vs = 42 * 512 + (0 if projected else 3 * L * 512) + 6 * 512 * 512
    post = 4 * 512 * 512 + 512 * 4
    return 2 * (NS * per_vs + post) * (kc + (kc + kf))


def describe(workload="c2"):
    side = WORKLOADS[workload][0]
    return {
        "metric": METRIC, "unit": "rays/s", "flop_per_ray": flop_per_ray(workload),
        "flop_per_ray_projected": flop_per_ray(workload, True),
        "config": {"workload": WORKLOADS[workload][7], "rays_per_frame": side * side},
    }


def c3_leg(dev, args, steps=3):
    """The C3 frame (128 x 128, 3 source views, L = 1792 latent at 16 x 16 supplied -- the YOLOv7 backbone is outside the
    reference tree --, 64 + 32 (16) samples) timed like the main line: `steps` full frames between synchronisations."""
    import torch
    from pixel_nerf_yolo_amd import conf as pconf, synth
    from pixel_nerf_yolo_amd.model import make_model
    from pixel_nerf_yolo_amd.render import NeRFRenderer
    from pixel_nerf_yolo_amd.util import gen_rays_range
    side, d_latent, lat_side, kc, kf, kfd, _, desc = WORKLOADS["c3"]
    mconf = pconf.default_mv()
    mconf.d["model"]["encoder"]["backbone"] = "custom"
    net = make_model(mconf["model"]).eval()
    sd = {}
    sd.update({"mlp_coarse." + k: v for k, v in synth.mlp_state(71, d_latent=d_latent).items()})
    sd.update({"mlp_fine." + k: v for k, v in synth.mlp_state(72, d_latent=d_latent).items()})
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    net = net.to(dev)
    net.set_latent_projection(args.projection)
    net.set_matrix_precision(args.precision)
    src, _ = synth.scene_cameras(NS)
    focal, c = torch.tensor(FOCAL128 * side / 128.0), torch.tensor([[side * 0.5, side * 0.5]])
    lat = torch.from_numpy(synth.latent(76, NS, d_latent, lat_side, lat_side)).to(dev)
    images = torch.zeros(NS, 3, side, side, device=dev)
    poses = torch.from_numpy(src)[None]
    rays = gen_rays_range(torch.from_numpy(synth.pose_spherical(120.0, -20.0, 1.3))[None], side, side, focal, Z_NEAR, Z_FAR, 0,
                          side * side, c=c[0], device=dev).reshape(1, -1, 8)
    par = NeRFRenderer(n_coarse=kc, n_fine=kf, n_fine_depth=kfd, depth_std=0.01, white_bkgd=True).eval().bind_parallel(
        net, None, simple_output=True).eval()

    def step():
        net.encode(images[None], poses, focal, c=c, latent=lat)
        with torch.no_grad():
            return par(rays)

    step()
    net.enable_kernel_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms = flops = 0.0
    launches = 0
    for _ in range(steps):
        rgb, _ = step()
        st = net.last_mlp_stats(full=True)
        ms, flops, launches = ms + st["kernel_ms"], flops + st["flops"], launches + st["launches"]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    net.enable_kernel_timing(False)
    f16x2 = net.last_launch_f16x2()
    peak = PEAK_F16X2_TFLOPS if f16x2 else PEAK_F32_MFMA_TFLOPS
    ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else None
    return {"workload": desc, "rays_per_s": side * side * steps / el, "steps": steps, "ms_per_step": el / steps * 1e3,
            "ms_per_launch": ms / launches if launches else None, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
            "frac": ach / peak if ach else None, "flop_per_ray_reference_order": flop_per_ray("c3", False),
            "finite": bool(torch.isfinite(rgb).all()),
            "note": "north_star target config timed beside the C2 bench value (BASELINE.json's metric is quoted on C2, whose "
                    "encoder is pinned); `bench.py --workload c3` gives the full line with cpu_baseline and parity"}


def self_launch(args):
    """`bench.py --gpus N` outside a launcher: start the N ranks as a child process tree (this process has not touched
    the GPU and never execs) and relay their output; rank 0 prints the JSON line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run(cmd, env=env).returncode


def train_main(args):
    """--mode train: the reference's training step (PixelNerfTrainer.calc_losses + optimizer step, trainer.py) replayed on
    the C2 model with its default batch: SB = 4 objects x 3 source views of 128x128 (encoded inside every step with the
    frozen ResNet-34 trunk: --freeze_enc), 128 rays per object (train.py:55), 64 coarse + 32 fine (16 depth) samples,
    loss = MSE(coarse.rgb) + MSE(fine.rgb), loss.backward(), Adam(lr 1e-4).step().  With --gpus N every rank trains on its
    own super-batch (different objects and pixels) and the MLP gradients are averaged with ONE all-reduce per step
    (pixel_nerf_yolo_amd.dist.allreduce_gradients) before the optimizer step: weak scaling, value = all ranks' rays / max time.
    value = training rays/s.  roofline: GEMM FLOPs executed by the four MLP kernels of a step (forward, stash forward,
    dX chain, weight-gradient GEMMs) / their HIP-event time, against the fp32 MFMA peak."""
    import numpy as np
    import torch

    import pnyolo_pkg
    pnyolo_pkg.load()
    from pixel_nerf_yolo_amd import conf as pconf, synth
    from pixel_nerf_yolo_amd.model import make_model
    from pixel_nerf_yolo_amd.render import NeRFRenderer
    from pixel_nerf_yolo_amd.util import gen_rays

    import torch.distributed as dist
    from pixel_nerf_yolo_amd import dist as pdist

    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path for the product)"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    SB, NSV, H, W, RB, KC, KF, KFD = 4, NS, 128, 128, 128, 64, 32, 16
    steps = args.steps if args.steps is not None else 20   # (a 10-step window showed 12.8 ... 18.4 ms between runs on one box: one stall weighs 10 %)
    net = make_model(pconf.default_mv()["model"], stop_encoder_grad=not args.train_encoder)
    sd = {}
    sd.update({"mlp_coarse." + k: v for k, v in synth.mlp_state(71).items()})
    sd.update({"mlp_fine." + k: v for k, v in synth.mlp_state(72).items()})
    sd.update(synth.resnet34_state(74, residual_gain=0.25))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    net = net.to(dev).train()
    if args.precision != "auto":
        net.set_matrix_precision(args.precision)   # f32: the exact-fp32 kernels for forward and backward
    if not args.train_encoder:
        net.encoder.eval()                                         # train.py:70-73 (--freeze_enc)
        for p_ in net.encoder.parameters():
            p_.requires_grad_(False)
    ren = NeRFRenderer(n_coarse=KC, n_fine=KF, n_fine_depth=KFD, depth_std=0.01, white_bkgd=True).train()
    if os.environ.get("PNYOLO_BENCH_SEED"):   # diagnostic: another stream of sampling jitter
        ren.base_seed = int(os.environ["PNYOLO_BENCH_SEED"])
    par = ren.bind_parallel(net, None).train()
    opt = torch.optim.Adam([p_ for p_ in net.parameters() if p_.requires_grad], lr=1e-4)
    params = [p_ for p_ in net.parameters() if p_.requires_grad]
    rs = np.random.RandomState(5 + rank)
    images = torch.from_numpy(np.stack([synth.images(80 + SB * rank + i, NSV, H, W) for i in range(SB)])).to(dev)     # (SB, NS, 3, H, W)
    poses = torch.from_numpy(np.stack([synth.scene_cameras(NSV, radius=1.3 + 0.02 * i)[0] for i in range(SB)]))
    focal = torch.full((SB,), FOCAL128)
    tgt = torch.from_numpy(np.stack([synth.pose_spherical(120.0 + 10 * i, -20.0, 1.3) for i in range(SB)]))
    all_rays = gen_rays(tgt, W, H, torch.tensor(FOCAL128), Z_NEAR, Z_FAR, device=dev).reshape(SB, -1, 8)
    gt_all = torch.from_numpy(rs.uniform(0, 1, size=(SB, H * W, 3)).astype(np.float32)).to(dev)

    def step(i):
        pix = torch.from_numpy(np.random.RandomState(1000 + i + 7919 * rank).randint(0, H * W, size=(SB, RB))).to(dev)
        rays = torch.gather(all_rays, 1, pix[..., None].expand(-1, -1, 8))
        gt = torch.gather(gt_all, 1, pix[..., None].expand(-1, -1, 3))
        net.encode(images, poses, focal)
        out = par(rays, want_weights=True)
        loss = torch.nn.functional.mse_loss(out["coarse"]["rgb"], gt) + torch.nn.functional.mse_loss(out["fine"]["rgb"], gt)
        opt.zero_grad()
        loss.backward()
        if world > 1:
            pdist.allreduce_gradients(params)   # one 27 MB collective (staged through the host under the gloo rehearsal)
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if os.environ.get("PNYOLO_BENCH_NO_GC"):   # diagnostic: is the periodic slow step the collector?
        import gc
        gc.disable()
    net.enable_kernel_timing(True)   # before the warm-up: the library creates its HIP events on the first timed call (~1.5 ms)
    for i in range(max(args.warmup, 4)):
        l0 = step(i)
    fence()
    # Everything allocated so far (modules, synthetic arrays, compiled signatures) lives for the whole run: move it out of the
    # collector's generations, or a full collection walks it in the middle of the timed steps (measured: one step in ~15 took
    # 80-110 ms instead of 12.7 -- PNYOLO_BENCH_NO_GC=1 removes the step, this keeps the collector on)
    import gc
    gc.collect()
    gc.freeze()
    k_ms, k_fl = [0.0] * 4, [0.0] * 4
    prof = None
    if os.environ.get("PNYOLO_BENCH_HOST_PROFILE"):   # diagnostic: cProfile of the timed loop on stderr
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    step_ms, t_prev = [], t0
    curve = []
    for i in range(steps):
        loss = step(100 + i)
        if os.environ.get("PNYOLO_BENCH_LOSSES"):   # diagnostic: the loss of every step (device scalars, read after the loop)
            curve.append(loss.detach())
        f = net.last_mlp_stats(full=True)     # NOTE: reading event times waits for the step
        b = net.last_backward_stats()
        k_ms[0] += f["kernel_ms"]
        k_fl[0] += f["flops"]
        for j in range(3):
            k_ms[1 + j] += b["kernel_ms"][j]
            k_fl[1 + j] += b["flops"][j]
        ff, fm = net.last_flush_stats()       # deferred mode: ONE weight-gradient GEMM per MLP over all scenes' tiles
        k_ms[3] += fm
        t_now = time.perf_counter()
        step_ms.append((t_now - t_prev) * 1e3)
        t_prev = t_now
    fence()
    elapsed = time.perf_counter() - t0
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(28)
    if world > 1:
        tmax = torch.tensor([elapsed], device="cpu" if args.rehearse_one_gpu else dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    net.enable_kernel_timing(False)
    assert bool(torch.isfinite(loss))
    names = ["forward (projected latent)", "stash forward (reference order)", "dX chain", "weight-gradient GEMMs + reduce"]
    # the scenes of the super-batch run on side streams, so per-kernel event times overlap: utilisation is priced
    # against the step's WALL time (a lower bound on the kernels' own efficiency)
    tot_ms, tot_fl = elapsed * 1e3, sum(k_fl)
    # scenes not pinned to F32 run forward, dX chain and weight gradients as split-f16 products: priced against that ceiling
    bwd_env = os.environ.get("PNYOLO_BWD_PRECISION", "")
    h2_train = os.environ.get("PNYOLO_MLP_PRECISION", "") != "f32" and bwd_env != "f32" and args.precision != "f32"
    train_peak = PEAK_F16X2_TFLOPS if h2_train else PEAK_F32_MFMA_TFLOPS
    out = {
        "metric": "training rays/sec, 64+32 samples/ray, 3-view 128x128 conditioning", "value": world * SB * RB * steps / elapsed,
        "unit": "rays/s", "n_gpus": world, "n_ranks_seen": dist.get_world_size() if dist.is_initialized() else 1, "steps": steps, "warmup": max(args.warmup, 4), "ms_per_step": elapsed / steps * 1e3,
        "ms_per_step_min_median_max": [min(step_ms), sorted(step_ms)[len(step_ms) // 2], max(step_ms)],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (matrix products of forward, dX chain and weight gradients on f16x2 split operands, fp32 accumulate)" if h2_train else "f32",
        "data": "synthetic",
        "config": {"workload": "train step of the C2 model: SB=4 objects x 3 views 128x128 (%s), 128 rays/object, 64 coarse + "
                               "32 fine (16 depth), MSE coarse+fine, Adam" % (
                                   ("ResNet34 trunk TRAINED, batch-statistics batch norm: %s + latent-gradient kernel"
                                    % ("ATen graph forward / backward (PNYOLO_TRUNK=torch)" if os.environ.get("PNYOLO_TRUNK") == "torch"
                                       else "this library's training kernels forward / backward (csrc/encoder_train.hip)"))
                                   if args.train_encoder else "frozen ResNet34 trunk encoded every step"),
                   "super_batch": ("one grouped scene: one launch per MLP pass over all objects' tiles (pny_scene_set_groups)"
                                   if net._group is not None else "one scene per object, launches on side streams"),
                   "rays_per_step": world * SB * RB, "rays_per_step_this_rank": SB * RB,
                   "parallelism": "dp%d: one super-batch per rank, 1 gradient all-reduce (27 MB fp32) per step" % world},
        "loss_first": float(l0.detach()), "loss_last": float(loss.detach()),
        "roofline": {"bound": "mfma", "kernel": "MLP kernels of a training step", "achieved": tot_fl / (tot_ms * 1e-3) / 1e12,
                     "peak": train_peak, "unit": "TFLOP/s", "frac": tot_fl / (tot_ms * 1e-3) / 1e12 / train_peak,
                     "x_fp32_mfma_peak": tot_fl / (tot_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                     "frac_is": "GEMM FLOPs of the step's MLP kernels / step wall time (encode, sampling, composite, "
                                "optimizer and host time included)", "traffic": None,
                     "kernels": [{"name": n_, "event_ms_per_step_summed_over_concurrent_scenes": m_ / steps,
                                  "gflop_per_step": f_ / steps / 1e9}
                                 for n_, m_, f_ in zip(names, k_ms, k_fl)]},
        "cpu_baseline": None,
    }
    if os.environ.get("PNYOLO_BENCH_STEP_TIMES"):
        out["step_ms"] = step_ms
    if curve:
        out["loss_curve"] = [round(float(x), 5) for x in curve]
    if args.rehearse_one_gpu:
        out["rehearsal"] = "ranks share ONE GPU, gloo transport: mechanics check, not a measurement"
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before anything initialises HIP/HSA (dmabuf IPC for RCCL)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 5; 3 for c4)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-rays", type=int, default=None, help="ray subset for the CPU baseline (0 = skip)")
    ap.add_argument("--projection", choices=["auto", "on", "off"], default="auto",
                    help="latent projection mode of the fused MLP (off = the reference's operation order)")
    ap.add_argument("--precision", choices=["auto", "f32", "f16x2"], default="auto",
                    help="matrix arithmetic of projected launches (include/pnyolo.h pny_scene_set_precision)")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the --precision f32 leg of the N=1 line")
    ap.add_argument("--train-encoder", action="store_true",
                    help="--mode train: do not freeze the ResNet-34 trunk (the reference's default training graph): the trunk "
                         "runs as a torch graph, the renderer returns d loss / d latent to it")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2",
                    help="c2 = BASELINE.json configs[1] (default, the bench line); c3/c4/c5 = configs[2..4]")
    ap.add_argument("--no-reference-order", action="store_true", help="skip the --projection off leg of the N=1 line")
    ap.add_argument("--no-c3-leg", action="store_true", help="skip the C3 (north_star target config) leg of the default N=1 line")
    ap.add_argument("--mode", choices=["render", "train"], default="render",
                    help="render (default, the bench line) or train: one optimisation step of the reference's trainer "
                         "(train/trainlib/PixelNerfTrainer.py:58-156) on the C2 model")
    ap.add_argument("--describe", action="store_true", help="print the workload description and exit (no GPU)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="multi-rank REHEARSAL on a one-GPU box: every rank uses device 0 and the exchange goes over gloo "
                         "(RCCL refuses duplicate devices); checks the launch / sharding / gather mechanics, the line is "
                         "marked rehearsal and is not a measurement")
    args = ap.parse_args()
    wl = args.workload
    side, d_latent, lat_side, KC, KF, KFD, scaling, wl_text = WORKLOADS[wl]
    if args.steps is None:
        args.steps = 3 if wl == "c4" else 5
    if args.cpu_rays is None:
        args.cpu_rays = 256 if wl == "c4" else 1536
    if args.describe:
        print(json.dumps(describe(wl)))
        return 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    if args.mode == "train":
        return train_main(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    import pnyolo_pkg
    pnyolo_pkg.load()
    from pixel_nerf_yolo_amd import conf as pconf, dist as pdist, synth
    from pixel_nerf_yolo_amd.model import make_model
    from pixel_nerf_yolo_amd.render import NeRFRenderer
    from pixel_nerf_yolo_amd.util import gen_rays_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path for the product)"
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    n_ranks_seen = dist.get_world_size() if dist.is_initialized() else 1

    H = W = side
    focal_v = FOCAL128 * side / 128.0
    # ---- scene: weights, encoder / latent, cameras.  c5: rank r holds scene r (own weights are shared, own latent)
    scene_id = rank if wl == "c5" else 0
    mconf = pconf.default_mv()
    if d_latent != 512:
        mconf.d["model"]["encoder"]["backbone"] = "custom"   # d_latent = 1792 (reference custom_encoder.py:22)
    net = make_model(mconf["model"]).eval()
    sd = {}
    sd.update({"mlp_coarse." + k: v for k, v in synth.mlp_state(71, d_latent=d_latent).items()})
    sd.update({"mlp_fine." + k: v for k, v in synth.mlp_state(72, d_latent=d_latent).items()})
    if wl == "c2":
        sd.update(synth.resnet34_state(74, residual_gain=0.25))   # latent O(1), like a trained trunk
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    net = net.to(dev)
    net.set_latent_projection(args.projection)
    net.set_matrix_precision(args.precision)
    src, _ = synth.scene_cameras(NS, radius=1.3 + 0.02 * scene_id)
    # weak-scaling workloads: every rank renders its own target view; c4: ONE view for all ranks
    tgt = synth.pose_spherical(120.0 + (0.0 if wl == "c4" else 10.0 * rank), -20.0, 1.3)
    images = torch.from_numpy(synth.images(75, NS, H, W)).to(dev) if wl == "c2" else torch.zeros(NS, 3, H, W, device=dev)
    focal, c = torch.tensor(focal_v), torch.tensor([[W * 0.5, H * 0.5]])
    poses = torch.from_numpy(src)[None]
    lat_in = None
    if wl != "c2":   # the backbone's output is an input
        lat_in = torch.from_numpy(synth.latent(76 + scene_id, NS, d_latent, lat_side, lat_side)).to(dev)

    def encode():
        net.encode(images[None], poses, focal, c=c, latent=lat_in)

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

    n_frame = H * W
    if wl == "c4":
        lo, hi, per = pdist.shard_bounds(n_frame, world, rank)       # this rank's ray range of the ONE frame
    else:
        lo, hi, per = 0, n_frame, n_frame                            # own whole frame
    n_rays = hi - lo
    pose_t = torch.from_numpy(tgt)[None]

    def make_rays():
        return gen_rays_range(pose_t, W, H, focal, Z_NEAR, Z_FAR, lo, n_rays, c=c[0], device=dev).reshape(1, -1, 8)

    rays = make_rays()
    ren = NeRFRenderer(n_coarse=KC, n_fine=KF, n_fine_depth=KFD, depth_std=0.01, white_bkgd=True).eval()
    par = ren.bind_parallel(net, None, simple_output=True).eval()
    exchange = world > 1 and wl in ("c2", "c3")   # c4 exchanges inside render_frame_sharded, c5 not at all
    gathered = torch.empty(world * per, 4, device=dev) if exchange else None
    tile = torch.zeros(per, 4, device=dev) if exchange else None

    def all_gather_tiles():
        if args.rehearse_one_gpu:   # gloo has no device all-gather: stage through the host (rehearsal only)
            host = torch.empty(world * per, 4)
            dist.all_gather(list(host.chunk(world, 0)), tile.cpu())
            gathered.copy_(host)
        else:
            dist.all_gather_into_tensor(gathered, tile)

    def render_range(r):
        with torch.no_grad():
            rgb, depth = par(r[None])
        return rgb[0], depth[0]

    def step():
        encode()                          # per-scene state is rebuilt inside the step (module docstring)
        if wl == "c4":
            # the product's sharded-frame path: per-rank ray generation for [lo, hi), render, one all-gather
            return pdist.render_frame_sharded(render_range, tgt, W, H, focal, Z_NEAR, Z_FAR, c=c[0], device=dev)
        with torch.no_grad():
            rgb, depth = par(rays)
        if exchange:
            tile[:n_rays, :3] = rgb[0]
            tile[:n_rays, 3] = depth[0]
            all_gather_tiles()
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
    if exchange:
        all_gather_tiles()
    fence()

    def timed(steps):
        """K steps between fences; returns (elapsed s max over ranks, summed MLP kernel stats, last outputs)."""
        net.enable_kernel_timing(True)
        fence()
        t0 = time.perf_counter()
        acc = dict(ms=0.0, flops=0.0, ref=0.0, launches=0, projected=False, f16x2=False)
        for _ in range(steps):
            rgb, depth = step()
            # HIP-event times of this step's MLP launches (recorded on the launch stream inside
            # libpnyolo; reading them waits for the step, which the next step depends on anyway)
            st = net.last_mlp_stats(full=True)
            acc["flops"] += st["flops"]
            acc["ref"] += st["flops_reference"]
            acc["ms"] += st["kernel_ms"]
            acc["launches"] += st["launches"]
            acc["projected"] = acc["projected"] or st["projected"]
            acc["f16x2"] = acc["f16x2"] or net.last_launch_f16x2()
        fence()
        elapsed = time.perf_counter() - t0
        net.enable_kernel_timing(False)
        if world > 1:
            tmax = torch.tensor([elapsed], device="cpu" if args.rehearse_one_gpu else dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        return elapsed, acc, rgb, depth

    for _ in range(args.warmup):
        step()
    elapsed, acc, rgb, depth = timed(args.steps)
    assert bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(depth).all())
    projected = acc["projected"]

    # HBM-side traffic of the dominant kernel cannot be counted from inside this process: it is the
    # committed rocprofv3 PMC measurement of the same command (tools/profile_gpu.sh -> profiles/mlp_traffic.json)
    traffic = traffic_note = traffic_stale = None
    try:
        with open(os.path.join(ROOT, "profiles", "mlp_traffic.json")) as fh:
            tj = json.load(fh)
        if bool(tj.get("projected_latent")) == bool(projected) and bool(tj.get("f16x2")) == bool(acc["f16x2"]) and wl == "c2":
            traffic, traffic_note = tj["bytes_per_launch"], "%s: %s" % (tj["tag"], tj["method"])
            traffic_stale = tj.get("kernel_sources_sha256") != kernel_source_fingerprint()
    except (OSError, ValueError, KeyError):
        pass

    rays_per_step = n_frame if wl == "c4" else world * n_frame
    value = rays_per_step * args.steps / elapsed

    def roof(a, rays_per_s, n_gpus):
        ach = a["flops"] / (a["ms"] * 1e-3) / 1e12 if a["ms"] > 0 else None
        peak = PEAK_F16X2_TFLOPS if a["f16x2"] else PEAK_F32_MFMA_TFLOPS
        extra = {}
        if a["f16x2"]:
            extra = {"matrix_path": "f16x2: fp32 operands as two f16 planes, 3 x v_mfma_f32_32x32x16_f16 per product, fp32 "
                                    "accumulate (error = the fp32 MFMA's, tools/ubench/split_f16_check.hip)",
                     "peak_is": "dense f16 MFMA peak %.1f / 3 issued MFMA FLOPs per fp32 GEMM FLOP" % PEAK_F16_MFMA_TFLOPS,
                     "issued_f16_tflops": 3 * ach if ach else None,
                     "x_fp32_mfma_peak": (ach / PEAK_F32_MFMA_TFLOPS) if ach else None}
        return {
            "bound": "mfma", "kernel": "pny_mlp_h2_kernel" if a["f16x2"] else "pny_mlp_kernel", "achieved": ach, "peak": peak,
            "unit": "TFLOP/s", "frac": (ach / peak) if ach else None,
            "frac_is": "executed-FLOP utilisation of this rank's MLP launches (HIP events on the launch stream)",
            "frac_survey_formula": rays_per_s * flop_per_ray(wl, False) / (peak * 1e12 * n_gpus), **extra,
            "launches": a["launches"], "avg_launch_ms": (a["ms"] / a["launches"]) if a["launches"] else None,
            "flops_per_launch": (a["flops"] / a["launches"]) if a["launches"] else None,
            "projected_latent": a["projected"],
            "reference_order_tflops": (a["ref"] / (a["ms"] * 1e-3) / 1e12) if a["ms"] > 0 else None,
        }

    rl = roof(acc, value, world)
    rl.update({"traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_note,
               "traffic_stale": traffic_stale,
               "traffic_stale_is": "true when the kernel sources benched differ from the ones the PMC figure was profiled on "
                                   "(sha256 over csrc/mlp_h2.hip, mlp_h2_core.h, mlp_core.h, pny_common.h)"})
    out = {
        "metric": METRIC, "value": value, "unit": "rays/s", "n_gpus": world, "n_ranks_seen": n_ranks_seen,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32 (matrix products on f16x2 split operands, fp32 accumulate)" if acc["f16x2"] else "f32", "data": "synthetic",
        "config": dict(describe(wl)["config"], n_views=NS, n_coarse=KC, n_fine=KF, n_fine_depth=KFD,
                       global_rays_per_step=rays_per_step, rays_per_step_this_rank=n_rays,
                       parallelism={"c4": "one frame's rays sharded over %d ranks (per-rank ray generation), 1 process/GPU, "
                                          "1 all-gather/frame" % world,
                                    "c5": "scene per GPU, %d ranks, no exchange" % world}.get(
                                        wl, "frame per GPU, 1 process/GPU, dp%d, 1 all-gather/frame" % world)),
        "encode_ms": encode_ms, "projection_ms": projection_ms,
        "flop_per_ray": flop_per_ray(wl, projected), "flop_per_ray_reference_order": flop_per_ray(wl, False),
        "roofline": rl,
    }
    if args.rehearse_one_gpu:
        out["rehearsal"] = "ranks share ONE GPU, gloo transport: mechanics check, not a measurement"

    if world == 1 and acc["f16x2"] and not args.no_fp32_leg:
        # the same frame on the exact-fp32 MFMA kernel (v_mfma_f32_32x32x2_f32), projection unchanged
        net.set_matrix_precision("f32")
        step()
        k_f32 = max(2, min(3, args.steps))
        e3, a3, _, _ = timed(k_f32)
        r3 = roof(a3, rays_per_step * k_f32 / e3, 1)
        out["fp32_matrix_path"] = {
            "rays_per_s": rays_per_step * k_f32 / e3, "steps": k_f32, "kernel": r3["kernel"], "achieved": r3["achieved"],
            "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": r3["frac"], "ms_per_launch": r3["avg_launch_ms"],
            "note": "--precision f32: not the bench value"}
        net.set_matrix_precision(args.precision)

    if world == 1 and args.projection != "off" and not args.no_reference_order:
        # the same frame in the reference's operation order (lin_z per sample): executed FLOPs = SURVEY.md 8d's count
        net.set_latent_projection("off")
        step()
        k_ref = max(2, min(3, args.steps))
        e2, a2, _, _ = timed(k_ref)
        r2 = roof(a2, rays_per_step * k_ref / e2, 1)
        out["roofline_reference_order"] = {
            "achieved": r2["achieved"], "frac": r2["frac"], "unit": "TFLOP/s", "peak": PEAK_F32_MFMA_TFLOPS,
            "ms_per_launch": r2["avg_launch_ms"], "rays_per_s": rays_per_step * k_ref / e2, "steps": k_ref,
            "flop_per_ray": flop_per_ray(wl, False), "projected_latent": a2["projected"],
            "note": "--projection off: lin_z per (sample, view) as the reference orders it; not the bench value"}
        net.set_latent_projection(args.projection)

    if world == 1 and wl == "c2" and not args.no_c3_leg:
        # north_star's target config (BASELINE.json configs[2]: 3-view YOLO-sized conditioning, L = 1792) timed in the same
        # run: 3 steps of the C3 frame (encode with the supplied latent + projection + render) on the default kernel
        out["c3"] = c3_leg(dev, args)

    if rank == 0 and args.cpu_rays > 0 and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pnyolo_oracle as orc  # CPU baseline leg: the oracle as the thing timed beside the GPU
        encode()
        if wl == "c2":   # the oracle runs its OWN trunk on the same images: the parity leg below is end to end
            t0 = time.perf_counter()
            lat = orc.spatial_encoder(synth.resnet34_state(74, residual_gain=0.25), images.cpu().numpy())[0].numpy()
            cpu_encode_s = time.perf_counter() - t0
        else:
            lat, cpu_encode_s = net.latent(0).cpu().numpy(), None   # the backbone's output is an input on both sides
        sc = orc.Scene(synth.mlp_state(71, d_latent=d_latent), synth.mlp_state(72, d_latent=d_latent), lat, src, focal, c,
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
                               "kind": "port", "sample": "%d random rays of the same %s frame, %.1f s" % (nb, wl.upper(), cpu_s),
                               "encode_s": cpu_encode_s}
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
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
