"""
Host-side mirror of the reference's renderer layer (src/render/): ``NeRFRenderer``,
``_RenderWrapper`` / ``bind_parallel``, ``YoloRenderer``, ``make_renderer`` with the reference's
signatures and return conventions, executing on libpnyolo.so.

A render call is ONE C-ABI call per scene (pny_render / pny_yolo_render): sampling, point
generation, the fused MLP, compositing, importance sampling and the sort all stay on the device.
The reference's ``eval_batch_size`` point-chunking exists only to bound its (rows x 554)
intermediates; fusion removes them, so the attribute is kept for interface compatibility and
ignored.

Random draws: the reference jitters unconditionally (nerf.py:117) and draws three more tensors
in the fine pass -- they are inputs of the path.  By default a per-call Philox seed is used
(in-kernel generation); ``renderer.draws = dict(u_coarse=, u_fine=, u_fine2=, g_depth=)`` replays
explicit tensors instead (parity mode, consumed by the next call).
"""
import ctypes as C
import os

import torch

from . import lib as _lib
from .lib import RenderOpts, RenderOut, check, ptr, stream_of


class _RenderWrapper(torch.nn.Module):
    """reference src/render/nerf.py:21-48"""

    def __init__(self, net, renderer, simple_output):
        super().__init__()
        self.net = net
        self.renderer = renderer
        self.simple_output = simple_output

    def forward(self, rays, want_weights=False):
        if rays.shape[0] == 0:
            return (torch.zeros(0, 3, device=rays.device), torch.zeros(0, device=rays.device))
        outputs = self.renderer(self.net, rays, want_weights=want_weights and not self.simple_output)
        if self.simple_output:
            part = outputs["fine"] if self.renderer.using_fine else outputs["coarse"]
            return part["rgb"], part["depth"]
        return outputs  # plain nested dict, as DotMap.toDict() in the reference


class _MultiDeviceRenderWrapper(torch.nn.Module):
    """``bind_parallel(net, gpus)`` with more than one device in ONE process: the drop-in for the reference's
    ``torch.nn.DataParallel(_RenderWrapper(...), gpus, dim=1)`` (src/render/nerf.py:360-377; called by train/train.py:78 and
    eval/eval.py:150 with ``--gpu_id "0 1 ..."``) without its per-call replication.  The reference re-broadcasts every
    parameter and the latent to every device on EVERY call; here each listed device holds a PERSISTENT
    replica of the native model and of the encoded scenes:
      * weights go up once and are re-sent only when the master's parameters changed (version counters),
      * latent and cameras are re-sent only after the master's ``encode()`` (the encoder runs once, on the master),
      * a call splits the rays on dim 1 into one contiguous range per device, enqueues every range on its device's stream
        without waiting, and gathers the tiles on ``gpus[0]`` with peer copies.
    Results: a ray renders to the same bits whatever range it is in (slice invariance), so with explicit draws
    the assembled output equals the single-device call bit for bit; with Philox draws each range has its own seed.
    Training calls (grad mode) split the same way (_train_split): every device runs the training forward and backward of its ray
    range on its replica, the replica's parameters enter the graph as device copies of the master's (``p.to(device)``: autograd
    carries each device's gradients back and sums them into the master's ``.grad``, as DataParallel's replicate / gather do), and
    the replicas pick the stepped weights up at the next call.  A latent that takes a gradient (trainable encoder: the trunk runs
    once, on the master) is a leaf of every device's graph in the same way: each device returns its share of d loss / d latent.
    dist.py (one process per GPU, one gradient all-reduce) stays the recommended multi-GPU path: no Python serialisation of
    the per-device launches."""

    def __init__(self, net, renderer, gpus, simple_output):
        super().__init__()
        self.net, self.renderer, self.simple_output = net, renderer, simple_output
        self.gpus = [int(g) for g in gpus]
        self._replicas = [None] * len(self.gpus)   # [0] is `net` itself when it lives on gpus[0]
        self._seen = [None] * len(self.gpus)       # (weights key, encode epoch) each replica was last synchronised to
        self._warned = False

    def _replica(self, i):
        from .model import PixelNeRFNet
        net = self.net
        dev = torch.device("cuda", self.gpus[i])
        if i == 0 and net._device() == dev:
            return net
        r = self._replicas[i]
        if r is None:
            r = PixelNeRFNet(net._conf, stop_encoder_grad=True)
            if net.mlp_fine is None:
                r.mlp_fine = None
            r.load_state_dict(net.state_dict(), strict=False)
            r = r.to(dev).eval()
            for (_, pr), (_, pm) in zip(r.named_parameters(), net.named_parameters()):
                pr.requires_grad_(pm.requires_grad)
            r._projection, r._precision = net._projection, net._precision
            self._replicas[i] = r
        key = (net._weights_key(), net._encode_epoch)
        if self._seen[i] != key:
            if self._seen[i] is not None and self._seen[i][0] != key[0]:
                with torch.no_grad():      # in place: the replica's refresh path (one repack launch), not a re-upload
                    for (k, dst), (_, src) in zip(r.state_dict().items(), net.state_dict().items()):
                        dst.copy_(src)
                if (net.mlp_fine is None) != (r.mlp_fine is None):
                    r.mlp_fine = None
            le = net._last_encode
            if le is None:
                raise RuntimeError("bind_parallel(net, gpus): call net.encode(...) before rendering")
            r.train(net.training)          # (a training master: the replica's encode() fills its grouped scene too)
            with torch.cuda.device(dev):
                lat = torch.cat([net.latent(sb) for sb in range(le["SB"])]).to(dev)
                images = torch.zeros(le["SB"], le["NS"], 3, le["H"], le["W"])
                r.encode(images, le["poses"], le["focal"], c=le["c"], latent=lat)
            self._seen[i] = key
        return r

    def forward(self, rays, want_weights=False):
        if rays.shape[0] == 0:
            return (torch.zeros(0, 3, device=rays.device), torch.zeros(0, device=rays.device))
        net, ren = self.net, self.renderer
        want_weights = want_weights and not self.simple_output
        training = torch.is_grad_enabled() and net.training and (net.trainable_mlp_parameters() or net.differentiable_latent() is not None)
        B = rays.shape[1]
        n_dev = len(self.gpus)
        if training and B >= 64 * n_dev:
            outputs = self._train_split(rays, want_weights)
        elif training or B < 64 * n_dev:
            outputs = ren(net, rays, want_weights=want_weights)
        else:
            outputs = self._render_split(rays, want_weights)
        if self.simple_output:
            part = outputs["fine"] if "fine" in outputs else outputs["coarse"]
            return part["rgb"], part["depth"]
        return outputs

    def _train_split(self, rays, want_weights):
        """Training call over several devices (see the class docstring): _RenderFunction per device on its replica, with
        differentiable device copies of the master's parameters as the graph's leaves."""
        net, ren = self.net, self.renderer
        SB, B = rays.shape[0], rays.shape[1]
        n_dev = len(self.gpus)
        per = -(-B // n_dev)
        per = -(-per // 64) * 64
        bounds = [(min(B, i * per), min(B, (i + 1) * per)) for i in range(n_dev)]
        named = net.trainable_mlp_parameters()
        lat_src = net.differentiable_latent()
        if lat_src is None:
            net.check_differentiable()
        kf = int(ren.n_fine) if (ren.using_fine and ren.n_fine > 0) else 0
        draws, calls = ren.draws, ren._calls
        ren.draws = None
        dev0 = torch.device("cuda", self.gpus[0])
        parts = []
        try:
            for i, (lo, hi) in enumerate(bounds):
                if hi <= lo:
                    continue
                r = self._replica(i)
                r.train()
                dev = torch.device("cuda", self.gpus[i])
                with torch.cuda.device(dev):
                    ren._calls = calls + i
                    if draws is not None:
                        ren.draws = {k: torch.as_tensor(v).reshape(SB, B, -1)[:, lo:hi].reshape(SB * (hi - lo), -1) for k, v in draws.items()}
                    leaves = [p if p.device == dev else p.to(dev) for _, p in named]
                    n_params = len(leaves)
                    if lat_src is not None:   # a latent that takes a gradient (trainable encoder): each device returns its share
                        leaves.append(lat_src if lat_src.device == dev else lat_src.to(dev))
                    outs = _RenderFunction.apply(ren, r, rays[:, lo:hi].to(dev), kf > 0, n_params, *leaves)
                parts.append([o.to(dev0) for o in outs])
        finally:
            ren._calls, ren.draws = calls + n_dev, None
        cat = [torch.cat([p[j] for p in parts], dim=1) if parts[0][j].numel() else parts[0][j] for j in range(6)]
        res = {"coarse": {"rgb": cat[0], "depth": cat[1]}}
        if want_weights:
            res["coarse"]["weights"] = cat[2]
        if kf > 0:
            res["fine"] = {"rgb": cat[3], "depth": cat[4]}
            if want_weights:
                res["fine"]["weights"] = cat[5]
        return res

    def _render_split(self, rays, want_weights):
        net, ren = self.net, self.renderer
        SB, B = rays.shape[0], rays.shape[1]
        n_dev = len(self.gpus)
        per = -(-B // n_dev)
        per = -(-per // 64) * 64                   # whole 64-ray groups per device
        bounds = [(min(B, i * per), min(B, (i + 1) * per)) for i in range(n_dev)]
        draws, calls = ren.draws, ren._calls
        ren.draws = None
        reps = [self._replica(i) if hi > lo else None for i, (lo, hi) in enumerate(bounds)]
        master_policy = net.f16_range_policy

        def part(i, only_f32=False):
            lo, hi = bounds[i]
            r = reps[i]
            dev = torch.device("cuda", self.gpus[i])
            if only_f32:
                r.set_matrix_precision("f32")
            with torch.cuda.device(dev):
                ren._calls = calls + i
                if draws is not None:
                    ren.draws = {k: torch.as_tensor(v).reshape(SB, B, -1)[:, lo:hi].reshape(SB * (hi - lo), -1) for k, v in draws.items()}
                return ren._render(r, rays[:, lo:hi].to(dev), want_weights, save=False)[0]

        try:
            outs = [part(i) if reps[i] is not None else None for i in range(n_dev)]
            # f16-range guard (model.guard_f16_range), after every device has its work: wait per device, repeat a range on fp32
            for i, r in enumerate(reps):
                if r is None or master_policy == "lazy" or not any(r.last_launch_f16x2(s) for s in range(len(r._h_scenes))):
                    continue
                torch.cuda.current_stream(torch.device("cuda", self.gpus[i])).synchronize()
                bits = r.range_status(clear=False)
                if bits:
                    r.range_status(clear=True)
                    if master_policy == "raise":
                        raise _lib.PnyRangeError(r._range_message(bits))
                    import warnings
                    warnings.warn("libpnyolo: " + r._range_message(bits) + " -- repeating the rays of cuda:%d on the fp32 kernels" % self.gpus[i])
                    outs[i] = part(i, only_f32=True)
        finally:
            ren._calls = calls + n_dev
        dev0 = torch.device("cuda", self.gpus[0])
        res = {}
        for p in outs[[o is not None for o in outs].index(True)]:
            res[p] = {k: torch.cat([o[p][k].to(dev0) for o in outs if o is not None], dim=1) for k in outs[0][p]}
        return res


class _MultiDeviceYoloWrapper(_MultiDeviceRenderWrapper):
    """``YoloRenderer.bind_parallel(net, gpus)`` with several devices (reference src/render/yolo.py:116-121 wraps the renderer in
    DataParallel(dim=1)): the same persistent replicas, the rays (flattened to (N, 8) as YoloRenderer.forward does) split into
    one contiguous range per device, the (n, anchors, 7) results concatenated on gpus[0]."""

    def __init__(self, net, renderer, gpus):
        super().__init__(net, renderer, gpus, simple_output=False)

    def forward(self, rays):
        net, ren = self.net, self.renderer
        training = torch.is_grad_enabled() and net.training and (net.trainable_mlp_parameters() or net.differentiable_latent() is not None)
        flat = rays.reshape(-1, 8)
        N, n_dev = flat.shape[0], len(self.gpus)
        if N < 64 * n_dev:
            ren.net = net
            return ren(rays)
        per = -(-(-(-N // n_dev)) // 64) * 64
        bounds = [(min(N, i * per), min(N, (i + 1) * per)) for i in range(n_dev)]
        draws, calls = ren.draws, ren._calls
        if training:   # as _MultiDeviceRenderWrapper._train_split: per-device graphs whose leaves are device copies of the master's
            named, lat_src = net.trainable_mlp_parameters(), net.differentiable_latent()
            if lat_src is None:
                net.check_differentiable()
            dev0 = torch.device("cuda", self.gpus[0])
            parts = []
            try:
                for i, (lo, hi) in enumerate(bounds):
                    if hi <= lo:
                        continue
                    r = self._replica(i)
                    r.train()
                    dev = torch.device("cuda", self.gpus[i])
                    with torch.cuda.device(dev):
                        ren.net, ren._calls = r, calls + i
                        ren.draws = None if draws is None else {"u_coarse": torch.as_tensor(draws["u_coarse"]).reshape(N, -1)[lo:hi]}
                        leaves = [p if p.device == dev else p.to(dev) for _, p in named]
                        n_params = len(leaves)
                        if lat_src is not None:
                            leaves.append(lat_src if lat_src.device == dev else lat_src.to(dev))
                        parts.append(_YoloRenderFunction.apply(ren, flat[lo:hi].to(dev), n_params, *leaves).to(dev0))
            finally:
                ren.net, ren._calls, ren.draws = net, calls + n_dev, None
            return torch.cat(parts, dim=0)
        outs = []
        try:
            for i, (lo, hi) in enumerate(bounds):
                if hi <= lo:
                    continue
                r = self._replica(i)
                dev = torch.device("cuda", self.gpus[i])
                with torch.cuda.device(dev):
                    ren.net, ren._calls = r, calls + i
                    ren.draws = None if draws is None else {"u_coarse": torch.as_tensor(draws["u_coarse"]).reshape(N, -1)[lo:hi]}
                    outs.append((i, r, ren._render(flat[lo:hi].to(dev))[0]))
            for i, r, _ in outs:   # f16-range guard, after every device has its work
                if net.f16_range_policy == "lazy" or not r.last_launch_f16x2():
                    continue
                torch.cuda.current_stream(torch.device("cuda", self.gpus[i])).synchronize()
                bits = r.range_status(clear=True)
                if bits:
                    raise _lib.PnyRangeError(r._range_message(bits))
        finally:
            ren.net, ren._calls, ren.draws = net, calls + n_dev, None
        dev0 = torch.device("cuda", self.gpus[0])
        return torch.cat([o.to(dev0) for _, _, o in outs], dim=0)


class _RenderFunction(torch.autograd.Function):
    """NeRFRenderer.forward under autograd: forward = the ordinary pny_render (with its z / per-sample outputs kept),
    backward = pny_render_backward per scene, which accumulates into gradient buffers bound to the MLP parameters."""

    @staticmethod
    def forward(ctx, renderer, model, rays, has_fine, n_params, *params):
        # params = the trainable MLP parameters, optionally followed by the differentiable latent of encode(latent=...)
        lat = params[n_params] if len(params) > n_params else None
        ctx.lat_meta = None if lat is None else (tuple(lat.shape), lat.device, lat.dtype)
        # Reserve the model-level stash for every scene's tiles (pny_model_defer_weight_grads): the forward then writes the
        # GEMM operands the backward needs while it evaluates the MLPs (no recompute), the scenes' backward calls run on
        # side streams and ONE weight-gradient GEMM per MLP follows.  If the reservation exceeds the stash budget: plain
        # forward, scene-after-scene backward with recompute in chunks.
        model._sync()
        L = _lib.load()
        SB, B = rays.shape[0], rays.shape[1]
        kc = int(renderer.n_coarse)
        kt = kc + int(renderer.n_fine) if has_fine else 0
        tiles = lambda pts: -(-pts // 64)
        fine_mlp = model.mlp_fine is not None
        ct = tiles(B * kc) + (tiles(B * kt) if (has_fine and not fine_mlp) else 0)
        ft = tiles(B * kt) if (has_fine and fine_mlp) else 0
        ctx.deferred = L.pny_model_defer_weight_grads(model._h_model, 1, model.num_views_per_obj, SB * ct, SB * ft) == 0
        model._defer_token = getattr(model, "_defer_token", 0) + 1
        ctx.token = model._defer_token
        res, saved = renderer._render(model, rays, want_weights=True, save=True, stash=ctx.deferred)
        saved["depth_coarse"] = res["coarse"]["depth"]
        ctx.renderer, ctx.model, ctx.saved, ctx.has_fine = renderer, model, saved, has_fine
        ctx.white_bkgd = bool(renderer.white_bkgd)
        ctx.set_materialize_grads(False)
        outs = [res["coarse"]["rgb"], res["coarse"]["depth"], res["coarse"]["weights"]]
        if has_fine:
            outs += [res["fine"]["rgb"], res["fine"]["depth"], res["fine"]["weights"]]
        else:
            outs += [rays.new_zeros(0), rays.new_zeros(0), rays.new_zeros(0)]
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_rgb_c, g_depth_c, g_w_c, g_rgb_f, g_depth_f, g_w_f):
        model, sv, ren = ctx.model, ctx.saved, ctx.renderer
        L = _lib.load()
        dev = model._device()
        grads = model.bind_mlp_grads()            # zeroed fp32 buffers, one per trainable MLP parameter, bound by name
        SB, B = sv["rays"].shape[0], sv["rays"].shape[1]
        kc = sv["z_coarse"].shape[-1]
        kt = sv["z_fine"].shape[-1] if ctx.has_fine else 0

        def prep(t):
            if t is None or t.numel() == 0:
                return None
            return t.detach().to(dev, torch.float32).contiguous()

        ups = [prep(x) for x in (g_rgb_c, g_depth_c, g_w_c, g_rgb_f, g_depth_f, g_w_f)]
        # The reservation made by the forward is still ours unless another training forward ran in between: then this
        # call computes its weight gradients immediately (accumulate bit 1), scene after scene, with recompute.
        deferred = ctx.deferred and getattr(model, "_defer_token", None) == ctx.token
        acc = 1 if deferred else 3
        # d loss / d latent, accumulated by every scene's call into its own slice (channel-last).  Its zero fill runs on the
        # current stream, so it is enqueued BEFORE the side streams fork from that stream: scenes 1.. add into the buffer
        # with atomics from their own streams, and nothing else would order those behind the fill.
        group = bool(sv.get("group"))
        if group and not deferred:
            raise RuntimeError("the grouped training forward lost its stash reservation (another training forward ran before "
                               "this backward): call backward() before the next forward, or set PNYOLO_GROUP=0")
        lat_grad = model.begin_latent_grad(ctx.lat_meta, SB, group=group) if ctx.lat_meta is not None else None
        if group:   # one call over the SB * B rays on the grouped scene (see NeRFRenderer._render)
            s_ = _lib.RenderSaved(z_coarse=sv["z_coarse"].data_ptr(), sample_coarse=sv["sample_coarse"].data_ptr())
            if not getattr(ren, "_detach_fine_depth", False):
                s_.depth_coarse = sv["depth_coarse"].data_ptr()
            if ctx.has_fine:
                s_.z_fine = sv["z_fine"].data_ptr()
                s_.sample_fine = sv["sample_fine"].data_ptr()
            g_ = _lib.RenderGrads(*[None if p is None else p.data_ptr() for p in ups])
            check(L.pny_render_backward(model._h_group, ptr(sv["rays"]), SB * B, C.byref(sv["opts"][0]), C.byref(s_), C.byref(g_),
                                        acc, stream_of(dev)))
            extra = () if lat_grad is None else (model.end_latent_grad(lat_grad, ctx.lat_meta, SB, group=True),)
            if lat_grad is not None:
                model._lat_grad_event = torch.cuda.Event()
                model._lat_grad_event.record(torch.cuda.current_stream(dev))
            check(L.pny_model_flush_weight_grads(model._h_model, 1, stream_of(dev)))
            check(L.pny_model_defer_weight_grads(model._h_model, 0, 0, 0, 0))
            return (None, None, None, None, None) + tuple(grads) + extra
        streams = model.fork_streams(SB) if deferred else [None] * SB
        calls = []
        for sb in range(SB):
            s_ = _lib.RenderSaved(z_coarse=sv["z_coarse"][sb].data_ptr(), sample_coarse=sv["sample_coarse"][sb].data_ptr())
            if not getattr(ren, "_detach_fine_depth", False):   # test aid: treat the depth samples as constants
                s_.depth_coarse = sv["depth_coarse"][sb].data_ptr()
            if ctx.has_fine:
                s_.z_fine = sv["z_fine"][sb].data_ptr()
                s_.sample_fine = sv["sample_fine"][sb].data_ptr()
            calls.append((s_, _lib.RenderGrads(*[None if p is None else p[sb].data_ptr() for p in ups])))

        def run(sb, bits):
            with torch.cuda.stream(streams[sb]):
                check(L.pny_render_backward(model._scene(sb), ptr(sv["rays"][sb]), B, C.byref(sv["opts"][sb]), C.byref(calls[sb][0]),
                                            C.byref(calls[sb][1]), acc | bits, stream_of(dev)))
        # Optional (PNYOLO_SPLIT_FLUSH=1): every scene's FINE pass first, then mlp_fine's weight-gradient flush on a stream of its own
        # BESIDE the coarse passes instead of behind them.  Measured (round 3): no gain -- 12.5-13.5 ms per step against 12.3-12.5:
        # the chain kernels hold 152 KiB of a CU's LDS and the weight-gradient GEMM 128 KiB, so the two never share a CU and the
        # "quarter-busy" coarse chains cannot be filled in.  Off by default; kept because it is tested (bits 4 / 8 of accumulate).
        split = deferred and ctx.has_fine and model.mlp_fine is not None and os.environ.get("PNYOLO_SPLIT_FLUSH", "0") == "1"
        fstream = None
        if split:
            for sb in range(SB):
                run(sb, 4)     # (bit 4: the fine pass only; bit 8: the coarse pass only -- include/pnyolo.h pny_render_backward)
            main = torch.cuda.current_stream(dev)
            fstream = getattr(model, "_flush_stream", None)
            if fstream is None or fstream.device != dev:
                fstream = model._flush_stream = torch.cuda.Stream(dev)
            fstream.wait_stream(main)
            for st_ in streams:
                if st_ is not None:
                    fstream.wait_stream(st_)
            with torch.cuda.stream(fstream):
                check(L.pny_model_flush_weight_grads(model._h_model, 1 | 32, stream_of(dev)))
            for sb in range(SB):
                run(sb, 8)
        else:
            for sb in range(SB):
                run(sb, 0)
        if deferred:
            model.join_streams(streams)
        # d loss / d latent is complete here, BEFORE the weight-gradient flush is enqueued: the encoder's backward (the trunk's
        # kernels, model._TrunkFunction) starts behind this point on its own stream and runs beside the flush
        extra = () if lat_grad is None else (model.end_latent_grad(lat_grad, ctx.lat_meta, SB),)
        if lat_grad is not None:
            model._lat_grad_event = torch.cuda.Event()
            model._lat_grad_event.record(torch.cuda.current_stream(dev))
        if deferred:
            check(L.pny_model_flush_weight_grads(model._h_model, 1 | (16 if split else 0), stream_of(dev)))
            if fstream is not None:
                torch.cuda.current_stream(dev).wait_stream(fstream)
            check(L.pny_model_defer_weight_grads(model._h_model, 0, 0, 0, 0))
        return (None, None, None, None, None) + tuple(grads) + extra


class NeRFRenderer(torch.nn.Module):
    """Drop-in for reference src/render/nerf.py:51 (ctor :68-102, forward :257-309,
    sched_step :324-344, from_conf :346-358, bind_parallel :360-377)."""

    def __init__(self, n_coarse=128, n_fine=0, n_fine_depth=0, noise_std=0.0, depth_std=0.01,
                 eval_batch_size=100000, white_bkgd=False, lindisp=False, sched=None):
        super().__init__()
        self.n_coarse, self.n_fine, self.n_fine_depth = n_coarse, n_fine, n_fine_depth
        self.noise_std, self.depth_std = noise_std, depth_std
        self.eval_batch_size = eval_batch_size  # unused: no materialised per-point intermediates
        self.white_bkgd = white_bkgd
        self.lindisp = lindisp
        if lindisp:
            print("Using linear displacement rays")
        self.using_fine = n_fine > 0
        self.sched = sched
        if sched is not None and len(sched) == 0:
            self.sched = None
        self.register_buffer("iter_idx", torch.tensor(0, dtype=torch.long), persistent=True)
        self.register_buffer("last_sched", torch.tensor(0, dtype=torch.long), persistent=True)
        self.draws = None          # explicit random draws for the next call (parity mode)
        self.base_seed = 1234
        self._calls = 0

    def forward(self, model, rays, want_weights=False):
        """
        :param rays [origins (3), directions (3), near (1), far (1)] (SB, B, 8)
        :return dict(coarse=dict(rgb (SB,B,3), depth (SB,B)[, weights (SB,B,Kc)]), fine=dict(...))
        ``fine`` is absent when n_fine == 0 (callers test ``len(fine) > 0``, PixelNerfTrainer.py:140-143).

        Under autograd (grad mode on, ``model.train()`` and MLP parameters that require grad) the call goes through ``_RenderFunction``:
        same forward kernels, and ``loss.backward()`` fills ``.grad`` of the MLP parameters through
        pny_render_backward (include/pnyolo.h).  With an unfrozen encoder (the reference's default, train/train.py:66-73) the
        backward also returns d loss / d latent to the latent of the last ``encode()`` -- the library's training trunk
        (model._TrunkFunction) or a latent supplied by the caller."""
        if self.sched is not None and self.last_sched.item() > 0:
            self.n_coarse = self.sched[1][self.last_sched.item() - 1]
            self.n_fine = self.sched[2][self.last_sched.item() - 1]
        assert len(rays.shape) == 3
        grad_mode = torch.is_grad_enabled() and model.training
        params = model.trainable_mlp_parameters() if grad_mode else []
        lat_src = model.differentiable_latent()
        if not params and lat_src is None:
            draws, calls = self.draws, self._calls

            def call():   # repeatable: the same draws / seeds when the f16-range guard asks for the call again
                self.draws, self._calls = draws, calls
                return self._render(model, rays, want_weights, save=False)[0]
            return model.guard_f16_range(call)
        if lat_src is None:
            model.check_differentiable()     # (with a differentiable latent the caller's own encoder takes the gradient)
        kf = int(self.n_fine) if (self.using_fine and self.n_fine > 0) else 0
        outs = _RenderFunction.apply(self, model, rays, kf > 0, len(params), *[p for _, p in params],
                                     *([lat_src] if lat_src is not None else []))
        res = {"coarse": {"rgb": outs[0], "depth": outs[1]}}
        if want_weights:
            res["coarse"]["weights"] = outs[2]
        if kf > 0:
            res["fine"] = {"rgb": outs[3], "depth": outs[4]}
            if want_weights:
                res["fine"]["weights"] = outs[5]
        return res

    def _render(self, model, rays, want_weights, save, stash=False):
        """One pny_render call per scene.  save=True also returns what the backward needs: the detached device rays,
        the sample depths and the per-sample MLP outputs of both passes, and the options of every scene's call."""
        model._sync()
        L = _lib.load()
        dev = model._device()
        SB, B = rays.shape[0], rays.shape[1]
        assert SB == model.num_objs, "super-batch of rays must match the encoded scenes"
        rays = rays.detach().to(dev, torch.float32).contiguous()
        if rays.data_ptr() % 16:
            rays = rays.clone()  # the kernel reads a ray row as two 16-byte words
        kc, kf, kfd = int(self.n_coarse), int(self.n_fine), int(self.n_fine_depth)
        use_fine = self.using_fine and kf > 0
        if not use_fine:
            kf = kfd = 0
        f32 = dict(device=dev, dtype=torch.float32)
        res = {"coarse": {"rgb": torch.empty(SB, B, 3, **f32), "depth": torch.empty(SB, B, **f32)}}
        if want_weights:
            res["coarse"]["weights"] = torch.empty(SB, B, kc, **f32)
        if use_fine:
            res["fine"] = {"rgb": torch.empty(SB, B, 3, **f32), "depth": torch.empty(SB, B, **f32)}
            if want_weights:
                res["fine"]["weights"] = torch.empty(SB, B, kc + kf, **f32)
        saved = None
        extra = dict(getattr(self, "_debug_out", None) or {})
        if save:
            saved = {"rays": rays, "z_coarse": torch.empty(SB, B, kc, **f32), "sample_coarse": torch.empty(SB, B, kc, 4, **f32),
                     "opts": []}
            if use_fine:
                saved["z_fine"] = torch.empty(SB, B, kc + kf, **f32)
                saved["sample_fine"] = torch.empty(SB, B, kc + kf, 4, **f32)
            for k in ("z_coarse", "sample_coarse", "z_fine", "sample_fine"):
                if k in saved:
                    saved[k] = extra.pop(k, saved[k])   # a debug capture of the same buffer: share it
        draws, self.draws = self.draws, None
        self._calls += 1
        keep = []
        # Training on a super-batch held by ONE grouped scene (model.encode, pny_scene_set_groups): a single pny_render over
        # the SB * B rays -- one MLP launch per pass over every object's tiles instead of SB launches on side streams.  Needs
        # whole 64-sample tiles per object in both passes; otherwise the per-object path below (model._scene fills its handles).
        model._last_call_group = False
        g = model._group_scene() if stash else None
        if (g is not None and SB == model._group["SB"] and (B * kc) % 64 == 0 and (not use_fine or (B * (kc + kf)) % 64 == 0)
                and not extra):
            o = RenderOpts(n_coarse=kc, n_fine=kf, n_fine_depth=kfd, depth_std=float(self.depth_std),
                           white_bkgd=int(bool(self.white_bkgd)), lindisp=int(bool(self.lindisp)),
                           seed=(self.base_seed + 7919 * self._calls) & 0xFFFFFFFFFFFFFFFF)

            def flat(name, cols, scale=None):
                if cols == 0:
                    return None
                if draws is not None and name in draws:
                    t = torch.as_tensor(draws[name], dtype=torch.float32).reshape(SB * B, cols).to(dev)
                elif scale is not None:
                    t = torch.randn(SB * B, cols, device=dev, dtype=torch.float32)
                else:
                    return None
                t = (t * scale if scale is not None else t).contiguous()
                keep.append(t)
                return t.data_ptr()
            if draws is not None:
                o.u_coarse_dev = flat("u_coarse", kc)
                o.u_fine_dev = flat("u_fine", kf - kfd)
                o.u_fine2_dev = flat("u_fine2", kf - kfd)
                o.g_depth_dev = flat("g_depth", kfd)
            if self.training and self.noise_std > 0.0:
                o.sigma_noise_coarse_dev = flat("noise_coarse", kc, float(self.noise_std))
                if use_fine:
                    o.sigma_noise_fine_dev = flat("noise_fine", kc + kf, float(self.noise_std))
            out = RenderOut()
            out.rgb_coarse, out.depth_coarse = res["coarse"]["rgb"].data_ptr(), res["coarse"]["depth"].data_ptr()
            if want_weights:
                out.weights_coarse = res["coarse"]["weights"].data_ptr()
            if use_fine:
                out.rgb_fine, out.depth_fine = res["fine"]["rgb"].data_ptr(), res["fine"]["depth"].data_ptr()
                if want_weights:
                    out.weights_fine = res["fine"]["weights"].data_ptr()
            for name in ("z_coarse", "sample_coarse", "z_fine", "sample_fine"):
                if name in saved:
                    setattr(out, name, saved[name].data_ptr())
            saved["opts"], saved["keep"], saved["group"] = [o], keep, True
            check(L.pny_scene_stash_next_render(g, 1))
            check(L.pny_render(g, ptr(rays), SB * B, C.byref(o), C.byref(out), stream_of(dev)))
            model._last_call_group = True
            return res, saved
        streams = model.fork_streams(SB)      # scenes are independent: one side stream each (None = current stream)
        for sb in range(SB):
            with torch.cuda.stream(streams[sb]):
                o = RenderOpts(n_coarse=kc, n_fine=kf, n_fine_depth=kfd, depth_std=float(self.depth_std),
                               white_bkgd=int(bool(self.white_bkgd)), lindisp=int(bool(self.lindisp)),
                               seed=(self.base_seed + 7919 * self._calls + sb) & 0xFFFFFFFFFFFFFFFF)
                if draws is not None:
                    def dev_draw(name, cols):
                        if cols == 0:
                            return None
                        t = torch.as_tensor(draws[name], dtype=torch.float32).reshape(SB, B, cols)[sb].to(dev).contiguous()
                        keep.append(t)
                        return t.data_ptr()
                    o.u_coarse_dev = dev_draw("u_coarse", kc)
                    o.u_fine_dev = dev_draw("u_fine", kf - kfd)
                    o.u_fine2_dev = dev_draw("u_fine2", kf - kfd)
                    o.g_depth_dev = dev_draw("g_depth", kfd)
                if self.training and self.noise_std > 0.0:
                    # sigma noise (nerf.py:231-232), training only: drawn here (or replayed from draws["noise_coarse" /
                    # "noise_fine"], unit normals) and handed to the composite kernels already scaled
                    def noise(name, cols):
                        if draws is not None and name in draws:
                            t = torch.as_tensor(draws[name], dtype=torch.float32).reshape(SB, B, cols)[sb].to(dev)
                        else:
                            t = torch.randn(B, cols, device=dev, dtype=torch.float32)
                        t = (t * float(self.noise_std)).contiguous()
                        keep.append(t)
                        return t.data_ptr()
                    o.sigma_noise_coarse_dev = noise("noise_coarse", kc)
                    if use_fine:
                        o.sigma_noise_fine_dev = noise("noise_fine", kc + kf)
                out = RenderOut()
                out.rgb_coarse = res["coarse"]["rgb"][sb].data_ptr()
                out.depth_coarse = res["coarse"]["depth"][sb].data_ptr()
                if want_weights:
                    out.weights_coarse = res["coarse"]["weights"][sb].data_ptr()
                if use_fine:
                    out.rgb_fine = res["fine"]["rgb"][sb].data_ptr()
                    out.depth_fine = res["fine"]["depth"][sb].data_ptr()
                    if want_weights:
                        out.weights_fine = res["fine"]["weights"][sb].data_ptr()
                for name, buf in extra.items():
                    setattr(out, name, buf[sb].data_ptr())
                if save:
                    for name in ("z_coarse", "sample_coarse", "z_fine", "sample_fine"):
                        if name in saved:
                            setattr(out, name, saved[name][sb].data_ptr())
                    saved["opts"].append(o)
                    saved["keep"] = keep   # the explicit draws are read again by the backward (depth samples)
                if stash:   # training forward into the reserved stash (pny_scene_stash_next_render)
                    check(L.pny_scene_stash_next_render(model._scene(sb), 1))
                check(L.pny_render(model._scene(sb), ptr(rays[sb]), B, C.byref(o), C.byref(out), stream_of(dev)))
        model.join_streams(streams)
        return res, saved

    def sched_step(self, steps=1):
        """Sample-count schedule of the reference (nerf.py:324-344): sched = [iterations, n_coarse, n_fine];
        once iter_idx passes sched[0][i] the renderer switches to (sched[1][i], sched[2][i])."""
        if self.sched is None:
            return
        self.iter_idx += steps
        milestones, coarse_counts, fine_counts = self.sched
        stage = int(self.last_sched.item())
        while stage < len(milestones) and int(self.iter_idx.item()) >= milestones[stage]:
            self.n_coarse, self.n_fine = coarse_counts[stage], fine_counts[stage]
            print("INFO: NeRF sampling resolution changed on schedule ==> c", self.n_coarse, "f", self.n_fine)
            stage += 1
        self.last_sched.fill_(stage)

    @classmethod
    def from_conf(cls, conf, white_bkgd=False, lindisp=False, eval_batch_size=100000):
        return cls(conf.get_int("n_coarse", 128), conf.get_int("n_fine", 0),
                   n_fine_depth=conf.get_int("n_fine_depth", 0), noise_std=conf.get_float("noise_std", 0.0),
                   depth_std=conf.get_float("depth_std", 0.01), white_bkgd=conf.get_float("white_bkgd", white_bkgd),
                   lindisp=lindisp, eval_batch_size=conf.get_int("eval_batch_size", eval_batch_size),
                   sched=conf.get_list("sched", None))

    def bind_parallel(self, net, gpus=None, simple_output=False):
        """reference nerf.py:360-377.  The reference wraps the module in a single-process ``DataParallel(dim=1)`` that
        re-broadcasts all parameters and the latent on every call; a `gpus` list longer than one gives the same call
        signature on persistent per-device replicas instead (_MultiDeviceRenderWrapper).  The recommended multi-GPU path
        stays one process per GPU (dist.py: per-rank ray generation + one RCCL all-gather)."""
        if gpus is not None and len(gpus) > 1:
            return _MultiDeviceRenderWrapper(net, self, gpus, simple_output=simple_output)
        return _RenderWrapper(net, self, simple_output=simple_output)


class YoloRenderer(torch.nn.Module):
    """Drop-in for reference src/render/yolo.py:3 (forward :37-114, from_conf :28-35,
    bind_parallel :116-121)."""

    def __init__(self, n_coarse, eval_batch_size, num_scales, num_anchors_per_scale):
        super().__init__()
        self.net = None
        self.n_coarse = n_coarse
        self.eval_batch_size = eval_batch_size  # unused (see module docstring)
        self.num_scales = num_scales
        self.num_anchors_per_scale = num_anchors_per_scale
        self.draws = None
        self.base_seed = 4321
        self._calls = 0

    def bind_net(self, net):
        self.net = net

    @classmethod
    def from_conf(cls, conf):
        return cls(conf.get_int("renderer.n_coarse", 128), conf.get_int("renderer.eval_batch_size", 1024),
                   conf.get_int("model.mlp_coarse.num_scales", 1),
                   conf.get_int("model.mlp_coarse.num_anchors_per_scale", 3))

    def forward(self, rays):
        """rays (..., 8) flattened to (N, 8) as the reference does (SB is folded away: only SB=1
        is meaningful, yolo.py:38,81) -> (N, num_anchors_per_scale, 7).  Under autograd (trainable MLP parameters,
        grad mode on) the result carries a graph: backward = pny_yolo_render_backward (YoloTrainer.py:160-186)."""
        net = self.net
        params = net.trainable_mlp_parameters() if (torch.is_grad_enabled() and net.training) else []
        lat = net.differentiable_latent()
        if params or lat is not None:
            if lat is None:
                net.check_differentiable()
            return _YoloRenderFunction.apply(self, rays, len(params), *[p for _, p in params], *([lat] if lat is not None else []))
        draws, calls = self.draws, self._calls

        def call():
            self.draws, self._calls = draws, calls
            return self._render(rays)[0]
        return net.guard_f16_range(call)

    def _render(self, rays, keep_raw=False):
        net = self.net
        net._sync()
        L = _lib.load()
        dev = net._device()
        rays = rays.detach().to(dev, torch.float32).reshape(-1, 8).contiguous()
        if rays.data_ptr() % 16:
            rays = rays.clone()
        n = rays.shape[0]
        out = torch.empty(n, self.num_anchors_per_scale, 7, device=dev, dtype=torch.float32)
        draws, self.draws = self.draws, None
        self._calls += 1
        u = None
        if draws is not None:
            u = torch.as_tensor(draws["u_coarse"], dtype=torch.float32).reshape(n, self.n_coarse).to(dev).contiguous()
        raw = getattr(self, "_debug_raw", None)
        if raw is None and keep_raw:
            raw = torch.empty(n, int(self.n_coarse), net.d_out, device=dev, dtype=torch.float32)
        seed = (self.base_seed + 7919 * self._calls) & 0xFFFFFFFFFFFFFFFF
        check(L.pny_yolo_render(net._scene(0), ptr(rays), n, int(self.n_coarse), ptr(u), seed, ptr(out), ptr(raw),
                                stream_of(dev)))
        return out, dict(rays=rays, u=u, seed=seed, raw=raw, n_coarse=int(self.n_coarse))

    def bind_parallel(self, net, gpus=None):
        """reference yolo.py:116-121 (DataParallel(self, gpus, dim=1) for several devices: here persistent per-device
        replicas, _MultiDeviceYoloWrapper)."""
        self.net = net
        if gpus is not None and len(gpus) > 1:
            return _MultiDeviceYoloWrapper(net, self, gpus)
        return self


class _YoloRenderFunction(torch.autograd.Function):
    """YoloRenderer.forward under autograd: pny_yolo_render forward (raw per-sample vectors kept),
    pny_yolo_render_backward into gradient buffers bound to the parameters of mlp_coarse."""

    @staticmethod
    def forward(ctx, renderer, rays, n_params, *params):
        lat = params[n_params] if len(params) > n_params else None
        ctx.lat_meta = None if lat is None else (tuple(lat.shape), lat.device, lat.dtype)
        out, saved = renderer._render(rays, keep_raw=True)
        ctx.renderer, ctx.saved, ctx.net = renderer, saved, renderer.net   # (the net of THIS call: a per-device replica under bind_parallel)
        return out

    @staticmethod
    def backward(ctx, g_out):
        net, sv = ctx.net, ctx.saved
        L = _lib.load()
        dev = net._device()
        grads = net.bind_mlp_grads()
        g_out = g_out.detach().to(dev, torch.float32).contiguous()
        lat_grad = net.begin_latent_grad(ctx.lat_meta, 1) if ctx.lat_meta is not None else None
        check(L.pny_yolo_render_backward(net._scene(0), ptr(sv["rays"]), sv["rays"].shape[0], sv["n_coarse"], ptr(sv["u"]),
                                         sv["seed"], ptr(sv["raw"]), ptr(g_out), 1, stream_of(dev)))
        extra = () if lat_grad is None else (net.end_latent_grad(lat_grad, ctx.lat_meta, 1),)
        return (None, None, None) + tuple(grads) + extra


def make_renderer(conf, lindisp=False):
    """reference src/render/render_util.py:5-12"""
    renderer_type = conf.get_string("renderer.type", "nerf")
    if renderer_type == "nerf":
        return NeRFRenderer.from_conf(conf["renderer"], lindisp=lindisp)
    if renderer_type == "yolo":
        return YoloRenderer.from_conf(conf)
    raise NotImplementedError("Unsupported renderer type")
