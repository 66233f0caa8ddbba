/*
 * pnyolo.h -- C ABI of libpnyolo.so: the MI355X (gfx950) rendering hot path of
 * pixelNeRF-YOLO.  Plain pointers and sizes only; no torch types.
 *
 * The reference (kofinandi/pixel-nerf-yolo) is pure Python/PyTorch and has no FFI of its own:
 * the boundary it offers is the Python object protocol between its trainers / eval scripts
 * and src/render + src/model.  Each entry point below names the reference
 * interface it stands in for (file:line relative to the reference tree); the Python classes in
 * pixel-nerf-yolo_amd/ re-create those interfaces on top of this ABI.
 *
 * Conventions
 *   - every function returns 0 on success or a negative pny_status; it never throws.
 *     pny_last_error() returns a thread-local message for the last failure.
 *   - *_dev pointers are device (HIP) pointers owned by the caller and borrowed for the call;
 *     *_host pointers are host memory.  All floating point data is fp32, row-major.
 *   - work is enqueued on the caller's stream (hipStream_t passed as void*); the library does
 *     not synchronise except where a function says so.
 *   - a pny_model owns packed weights; a pny_scene owns the per-scene state the reference keeps
 *     in module buffers after encode() (latent, world->cam poses, intrinsics) plus a grow-only
 *     workspace.  One model / scene per device and per caller thread.
 *   - streams: the calls on one scene are ordered by the stream they are enqueued on.  A call that
 *     arrives on a different stream than the previous call on the same scene is ordered behind it by
 *     the library (one event record + stream wait, paid on the switch only), so a scene may migrate
 *     between streams; two streams must not drive the same scene concurrently from two threads.
 *     The previous stream must still exist at the switch or have been destroyed after draining.
 */
#ifndef PNYOLO_H
#define PNYOLO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNY_ABI_VERSION 11

typedef enum pny_status {
    PNY_OK = 0,
    PNY_ERR_ARG = -1,      /* bad argument / unsupported configuration */
    PNY_ERR_STATE = -2,    /* call order (e.g. render before weights / latent / cameras) */
    PNY_ERR_HIP = -3,      /* HIP runtime error (message carries hipGetErrorString) */
    PNY_ERR_NOGPU = -4,    /* no gfx950 device visible */
    PNY_ERR_RANGE = -5     /* an earlier F16X2 launch met a value outside the f16 range (pny_model_range_status) */
} pny_status;

typedef struct pny_model pny_model;
typedef struct pny_scene pny_scene;
typedef void* pny_stream; /* hipStream_t */

/* Values the reference reads from conf["model"] (src/model/models.py:21-83,
 * src/model/resnetfc.py:189-205, src/model/code.py:45-52). */
typedef struct pny_model_desc {
    int32_t d_latent;      /* encoder.latent_size: 512 (ResNet34, encoder.py:67) or 1792 (YOLO) */
    int32_t d_hidden;      /* mlp.d_hidden; this build supports 512 */
    int32_t d_out;         /* 4, or 7*num_anchors_per_scale = 21 in YOLO mode (models.py:80-83) */
    int32_t n_blocks;      /* mlp.n_blocks (5) */
    int32_t combine_layer; /* mlp.combine_layer (3): cross-view mean before this block */
    int32_t num_freqs;     /* code.num_freqs (6) */
    float freq_factor;     /* code.freq_factor (1.5) */
    int32_t yolo;          /* mlp_coarse.yolo: raw outputs, extrinsics used as given, z>=0 culling */
    int32_t has_fine;      /* mlp_fine.type != empty */
    int32_t device;        /* HIP device ordinal */
    int32_t enc_use_first_pool; /* encoder.use_first_pool (encoder.py:145-146); 0 in conf/exp/sn64.conf */
} pny_model_desc;

int pny_version(void);
const char* pny_last_error(void);

/* make_model(conf["model"]) -- src/model/__init__.py:4-11, src/model/models.py:16-90 */
int pny_model_create(pny_model** out, const pny_model_desc* desc);
void pny_model_destroy(pny_model* m);

/* PixelNeRFNet.load_weights -> load_state_dict, one call per state_dict tensor
 * (src/model/models.py:320-349).  name is the state_dict key ("mlp_coarse.lin_in.weight",
 * "mlp_fine.blocks.3.fc_1.bias", "encoder.model.layer2.0.downsample.1.running_var", ...);
 * unknown keys (code._freqs, layer4.*, fc.*, num_batches_tracked) are accepted and ignored.
 * data_host: fp32, shape as in the state_dict. */
int pny_model_load_weights(pny_model* m, const char* name, const float* data_host,
                           const int64_t* shape, int ndim);
/* Packs the loaded tensors into the MFMA operand order and uploads them.  Fails with
 * PNY_ERR_STATE and names the first missing MLP tensor.  Synchronous.  May be called again after
 * further pny_model_load_weights calls (weights changed); scenes of the model stay valid. */
int pny_model_finalize(pny_model* m);
/* Training: device-side weight refresh.  pny_model_bind_param tells the library where a state_dict tensor of the MLPs
 * lives on the device (fp32, contiguous, its state_dict shape; borrowed until rebound); after the parameters changed
 * in place (optimizer.step()), pny_model_refresh re-creates every packed operand of both MLPs from those tensors with
 * one kernel launch on `stream` (no host round trip, asynchronous; ordered behind the scenes' earlier calls).
 * pny_model_finalize must have run once (it fixes the layout).  The inference trunk's folded weights are not refreshed (they
 * follow pny_model_load_weights + finalize); the training trunk (pny_trunk_train_forward) reads its bound parameters directly. */
int pny_model_bind_param(pny_model* m, const char* name, const float* param_dev);
int pny_model_refresh(pny_model* m, pny_stream stream);
/* `net.mlp_fine = None` (reference eval/eval.py:140): with enable=0 the fine pass of pny_render and
 * pny_query(coarse=0) evaluate mlp_coarse.  Default 1 (ignored when the model has no fine MLP). */
int pny_model_use_fine(pny_model* m, int enable);

int pny_scene_create(pny_scene** out, pny_model* m);
void pny_scene_destroy(pny_scene* s);

/* Camera half of PixelNeRFNet.encode (src/model/models.py:115-148).
 * poses_host (ns,4,4): cam->world, inverted here to world->cam [R^T | -R^T t]; in YOLO mode
 * they are world->cam extrinsics and used as given.  focal_host (nf,2), c_host (nc,2) with
 * nf, nc in {1, ns}; fy is negated here in non-YOLO mode as the reference does.
 * Host-only: the cameras are kept in the handle and travel to the device as kernel arguments of each
 * later launch (no device copy, no synchronisation; launches already enqueued keep the cameras they
 * were launched with). */
int pny_scene_set_cameras(pny_scene* s, const float* poses_host, int ns, const float* focal_host, int nf,
                          const float* c_host, int nc, int width, int height);
/* Super-batch in ONE scene (ABI 11).  The reference flattens a super-batch everywhere: NeRFRenderer.forward takes rays
 * (SB, B, 8) and composites (SB * B) rays in one pass (src/render/nerf.py:283-288), PixelNeRFNet.forward conditions object o's
 * points on ITS views out of the (SB * NS) encoded ones (src/model/models.py:160-246, util.repeat_interleave).  With
 * n_objs > 1 the scene's ns views (cameras and latent, object-major) are n_objs objects' view lists of ns / n_objs views each,
 * and every later render / query / backward call splits its rays (points) into n_objs equal consecutive shares, share o seeing
 * views [o * ns / n_objs, (o + 1) * ns / n_objs): one MLP launch per pass covers all objects' tiles.  Every share's sample
 * count (rays x samples per ray, in both passes) must be a multiple of 64 -- PNY_ERR_ARG otherwise, and the caller falls
 * back to one scene per object; the backward of a grouped scene needs the deferred stash (pny_model_defer_weight_grads).
 * ns <= 16 views in total.  n_objs = 1 restores the default. */
int pny_scene_set_groups(pny_scene* s, int n_objs);
/* Encoder bypass: installs a latent (ns, L, Hl, Wl) NCHW as SpatialEncoder.forward would leave
 * it in self.latent (src/model/encoder.py:169-172); repacked to NHWC on device. */
int pny_scene_set_latent(pny_scene* s, const float* latent_dev, int ns, int channels, int hl, int wl,
                         pny_stream stream);
/* SpatialEncoder.forward, ResNet-34 trunk, eval-mode batch norm (src/model/encoder.py:139-173).
 * images_dev (ns,3,H,W) in [-1,1].  Leaves the 512-channel latent (H/2 x W/2) in the scene. */
int pny_scene_encode(pny_scene* s, const float* images_dev, int ns, int height, int width, pny_stream stream);
/* The same for the n_scenes objects of a super-batch (PixelNeRFNet.encode with images (SB, NS, 3, H, W), src/model/models.py:
 * 92-151, flattens them into ONE encoder call): images_dev (n_scenes * ns, 3, H, W), scene i owns images [i * ns, (i + 1) * ns).
 * One pass of the trunk over all of them (41 launches instead of 41 per scene); every scene must belong to the same model,
 * and all of them are entered on `stream`. */
int pny_scenes_encode(pny_scene** scenes, int n_scenes, const float* images_dev, int ns, int height, int width, pny_stream stream);
/* Copies the scene latent out as (ns, L, Hl, Wl) NCHW (test / debugging aid). */
int pny_scene_get_latent(pny_scene* s, float* latent_dev, pny_stream stream);
int pny_scene_latent_shape(pny_scene* s, int* ns, int* channels, int* hl, int* wl);

/* util.gen_rays (src/util/util.py:240-278, yolo_mode=0: poses cam->world, unit dirs, integer
 * pixel centres) and util.gen_rays_yolo (src/util/util.py:808-876, yolo_mode=1: poses are
 * world->cam extrinsics, K^-1 [x+.49,y+.49,1], not normalised).  out_dev (b,H,W,8). */
int pny_gen_rays(const float* poses_host, int b, int width, int height, const float focal[2],
                 const float c[2], float z_near, float z_far, int yolo_mode, float* out_dev,
                 pny_stream stream);
/* The same for rays [first_ray, first_ray + n_rays) of the flattened (b, H, W) pixel grid only:
 * out_dev (n_rays, 8), 16-byte aligned.  This is how a rank of a ray-sharded render produces its own
 * slice of a frame on its own device (no scatter of rays).  Asynchronous; the camera
 * blocks travel as kernel arguments (no device allocation, no copy, no synchronisation). */
int pny_gen_rays_range(const float* poses_host, int b, int width, int height, const float focal[2],
                       const float c[2], float z_near, float z_far, int yolo_mode, int64_t first_ray,
                       int64_t n_rays, float* out_dev, pny_stream stream);

/* PixelNeRFNet.forward for one scene (src/model/models.py:153-318):
 * xyz_dev, viewdirs_dev (n,3) world space -> out_dev (n,d_out) = [sigmoid rgb, relu sigma]
 * (YOLO mode: raw).  coarse=0 selects mlp_fine when the model has one. */
int pny_query(pny_scene* s, const float* xyz_dev, const float* viewdirs_dev, int64_t n, int coarse,
              float* out_dev, pny_stream stream);

/* NeRFRenderer options (src/render/nerf.py:68-102, from_conf :346-358). */
typedef struct pny_render_opts {
    int32_t n_coarse;
    int32_t n_fine;        /* total fine samples incl. depth samples; 0 = coarse only */
    int32_t n_fine_depth;
    float depth_std;
    int32_t white_bkgd;
    int32_t lindisp;
    /* The renderer's random draws (nerf.py:117,141,147,164) are inputs of the path.  Either all
     * needed pointers are given (parity mode), or they are NULL and an in-kernel Philox stream
     * keyed by `seed` generates them (perf mode). */
    const float* u_coarse_dev; /* (n, n_coarse) U[0,1) */
    const float* u_fine_dev;   /* (n, n_fine-n_fine_depth) U[0,1): inverse-cdf draw */
    const float* u_fine2_dev;  /* (n, n_fine-n_fine_depth) U[0,1): in-bin jitter */
    const float* g_depth_dev;  /* (n, n_fine_depth) N(0,1) */
    uint64_t seed;
    /* Training only (src/render/nerf.py:231-232: sigmas + randn_like(sigmas) * noise_std): the noise ADDED to sigma before
     * the composite's relu, already scaled by noise_std; (n, n_coarse) and (n, n_coarse+n_fine), or NULL (no noise). */
    const float* sigma_noise_coarse_dev;
    const float* sigma_noise_fine_dev;
} pny_render_opts;

/* Any output pointer may be NULL. */
typedef struct pny_render_out {
    float* rgb_coarse;     /* (n,3) */
    float* depth_coarse;   /* (n) */
    float* weights_coarse; /* (n,n_coarse) */
    float* rgb_fine;       /* (n,3) */
    float* depth_fine;     /* (n) */
    float* weights_fine;   /* (n,n_coarse+n_fine) */
    float* z_coarse;       /* (n,n_coarse) */
    float* z_fine;         /* (n,n_coarse+n_fine) sorted */
    float* sample_coarse;  /* (n,n_coarse,4) per-sample [rgb, sigma] from the coarse MLP */
    float* sample_fine;    /* (n,n_coarse+n_fine,4) */
} pny_render_out;

/* NeRFRenderer.forward for one scene (src/render/nerf.py:257-309): rays_dev (n,8) =
 * [origin, dir, near, far], 16-byte aligned (rows are read as two 16-byte words; PNY_ERR_ARG otherwise). */
int pny_render(pny_scene* s, const float* rays_dev, int64_t n, const pny_render_opts* opts,
               const pny_render_out* out, pny_stream stream);

/* YoloRenderer.forward (src/render/yolo.py:37-114): coarse sampling only, raw (A*7)-vectors, rays_dev 16-byte aligned,
 * out_dev (n, A, 7) = [max_k p, sum_k p v / (sum_k p + 1e-5)].  raw_dev (n,K,A*7) optional. */
int pny_yolo_render(pny_scene* s, const float* rays_dev, int64_t n, int n_coarse, const float* u_coarse_dev,
                    uint64_t seed, float* out_dev, float* raw_dev, pny_stream stream);

/* Stage entry points (used by the renderer above; exported for stage-wise parity tests). */
/* NeRFRenderer.sample_coarse, src/render/nerf.py:104-121 */
int pny_sample_coarse(const float* rays_dev, int64_t n, int n_coarse, int lindisp, const float* u_dev,
                      uint64_t seed, float* z_dev, pny_stream stream);
/* NeRFRenderer.composite arithmetic, src/render/nerf.py:184-188,229-250.  sample_dev (n,K,4). */
int pny_composite(const float* rays_dev, const float* z_dev, const float* sample_dev, int64_t n, int k,
                  int white_bkgd, float* weights_dev, float* rgb_dev, float* depth_dev, pny_stream stream);
/* sample_fine + sample_fine_depth + cat + sort, src/render/nerf.py:126-167,291-301.
 * z_out_dev (n, n_coarse+n_fine) ascending. */
int pny_sample_fine(const float* rays_dev, const float* z_coarse_dev, const float* weights_dev,
                    const float* depth_dev, int64_t n, int n_coarse, int n_fine, int n_fine_depth,
                    float depth_std, int lindisp, const float* u_dev, const float* u2_dev,
                    const float* g_dev, uint64_t seed, float* z_out_dev, pny_stream stream);
/* YoloRenderer aggregation, src/render/yolo.py:96-114.  raw_dev (n,K,A*7) -> out_dev (n,A,7). */
int pny_yolo_aggregate(const float* raw_dev, int64_t n, int k, int n_anchors, float* out_dev,
                       pny_stream stream);

/* ---- YOLO detection tail (callers: train/trainlib/YoloTrainer.py:283-286,347) ----
 * Boxes are rows of 6 floats [class, score, x, y, w, h] as the reference's lists hold them. */
/* util.convert_cells_to_bboxes for one image (src/util/util.py:633-689): cells_dev (h,w,a,7) raw
 * predictions (is_predictions=1: sigmoid xy, exp(wh)*anchor, argmax class) or (h,w,a,6) targets;
 * anchors_host (a,2), a <= 4; boxes_dev (h*w*a, 6) in (y, x, anchor) order. */
int pny_cells_to_bboxes(const float* cells_dev, const float* anchors_host, int h, int w, int n_anchors,
                        int is_predictions, float* boxes_dev, pny_stream stream);
/* util.nms (src/util/util.py:691-722), n <= 8192, including the reference's remove-while-iterating
 * behaviour.  kept_dev (n,6): survivors in output order; meta_dev[0] = survivors, meta_dev[1] = boxes
 * above `threshold`; highest_conf_dev[0] = max score of all inputs (-inf when n == 0). */
int pny_nms(const float* boxes_dev, int n, double iou_threshold, double threshold, float* kept_dev, int* meta_dev,
            float* highest_conf_dev, pny_stream stream);
/* util.calculate_tp_fp_fn (src/util/util.py:765-802): nms on both lists, then IoU matching.
 * out_dev: int[3] = tp, fp, fn. */
int pny_tp_fp_fn(const float* target_boxes_dev, int nt, const float* pred_boxes_dev, int np, double nms_iou,
                 double nms_threshold, double match_iou, int* out_dev, pny_stream stream);

/* Latent projection.  `x = x + lin_z[b](z)` (src/model/resnetfc.py:176-182) with z the bilinear
 * interpolation of the latent (src/model/encoder.py:101) is linear in the latent, so
 * lin_z[b](interp(latent)) == interp(lin_z[b](latent)) up to fp32 rounding: with projection ON the
 * library applies lin_z[b] to every latent PIXEL once per scene (cached until the latent or the
 * weights change) and the fused kernel interpolates the projected maps instead of running the lin_z
 * GEMMs per (sample, view).  Results stay inside the 1e-4 parity tolerance (tests/test_gpu_parity.py);
 * OFF executes the reference's operation order.  AUTO (default; env PNYOLO_PROJECTION=off|on
 * overrides at scene creation) projects every launch when the scene can use the F16X2 kernel (below), so that a ray's
 * result does not depend on the size of its batch; otherwise when a launch has >= 2x as many points as the latent has
 * pixels per view. */
#define PNY_PROJECTION_OFF 0
#define PNY_PROJECTION_ON 1
#define PNY_PROJECTION_AUTO 2
int pny_scene_set_projection(pny_scene* s, int mode);
/* Compute the projected maps now (coarse, and fine when the model has one) instead of lazily inside
 * the first large launch; a no-op when they are current.  Error when the mode is OFF. */
int pny_scene_project(pny_scene* s, pny_stream stream);

/* Matrix arithmetic of PROJECTED launches.  F32: v_mfma_f32_32x32x2_f32 on fp32 operands.  F16X2: every fp32 operand
 * is split into two f16 planes, x = f16(x) + f16(x - f16(x)) (22 significant bits; the second plane may be denormal,
 * which the matrix cores honour), and a product is x1 w1 + x2 w1 + x1 w2 on v_mfma_f32_32x32x16_f16 with fp32
 * accumulation: the same measured error against fp64 as the fp32 matrix path (tools/ubench/split_f16_check.hip,
 * 1.7e-6 vs 1.8e-6 at K = 512 on the network's magnitudes) at 5.3x its matrix rate; held to the same 1e-4 bar by the
 * same golden vectors.  Values beyond the f16 range (|x| > 65504 in an activation or weight) are NOT representable: use
 * F32 for such models (AUTO does so by itself when a WEIGHT loaded at pny_model_finalize is outside the range; weights changed
 * through pny_model_refresh and activations are not checked).  AUTO (default; env PNYOLO_MLP_PRECISION=f32|f16x2 overrides at scene creation) otherwise
 * equals F16X2: every projected launch, whatever its size, so that a ray's result does not depend on the batch it is rendered
 * in.  Launches without projection (training forward, the reference operation order, batches below the projection
 * threshold) always run F32; models with more than 6 residual blocks or combine_layer = 0 always run F32.
 * The setting also selects the arithmetic of the latent projection and of the BACKWARD pass (pny_render_backward,
 * pny_query_backward, pny_yolo_render_backward, pny_model_flush_weight_grads): scenes not pinned to F32 run the dX chain and
 * the weight-gradient GEMMs as split-f16 products with power-of-two gradient scaling (csrc/mlp_bwd_h2.hip,
 * pny_dw_gemm_h2_kernel); a deferred flush runs fp32 when any contributing scene was pinned to F32.  Env
 * PNYOLO_BWD_PRECISION=f32|f16x2 overrides the backward's choice (read at every call). */
#define PNY_PRECISION_F32 0
#define PNY_PRECISION_F16X2 1
#define PNY_PRECISION_AUTO 2
int pny_scene_set_precision(pny_scene* s, int mode);
/* 1 when the last MLP launch of the scene ran the F16X2 kernel. */
int pny_scene_last_precision(pny_scene* s, int* f16x2);

/* Run-time guard of the F16X2 arithmetic's range.  The split operands are f16 planes: a value of magnitude >= 65520 (or an
 * infinity) has no f16 representation, and a launch that meets one returns garbage where the reference -- fp32 throughout,
 * src/model/resnetfc.py:134-186; it only prints when its output holds a NaN, src/model/models.py:174-270 -- still returns
 * numbers.  Every F16X2 kernel therefore reports what it meets into one word per model that the host can read at any time
 * (pinned host memory, written with a system-scope atomic OR by the lanes that saw the value):
 *   PNY_RANGE_ACTIVATION  forward (render / query / training forward): a relu output or a lin_in input left the range;
 *   PNY_RANGE_GRADIENT    backward: the running max |dY| of a chain / weight-gradient launch is not finite (the chain works
 *                         in a per-tile scaled domain with 2^11 of headroom, csrc/mlp_bwd_h2.hip);
 *   PNY_RANGE_WEIGHT      pny_model_refresh repacked a weight of magnitude > 65504 or a NaN (pny_model_finalize checks on the
 *                         host and keeps AUTO on F32 by itself; a refresh runs on the device, after the launch decision).
 * pny_model_range_status returns the bits seen so far (`bits`, may be NULL) and, with clear != 0, resets them.  It does not
 * synchronise: the bits of a launch are complete once that launch has finished (synchronise its stream first).
 * While bits are set, every scene-level entry point (render, query, encode, backward ...) of the model fails with
 * PNY_ERR_RANGE -- results computed since the overflow are not to be trusted -- until they are cleared; PNY_RANGE_WEIGHT also
 * drops AUTO scenes to F32 until the next pny_model_finalize.  Recovery: clear, pny_scene_set_precision(F32), repeat the call
 * (pixel-nerf-yolo_amd/model.py does this transparently for no-grad calls: `f16_range_policy`). */
#define PNY_RANGE_ACTIVATION 1u
#define PNY_RANGE_GRADIENT 2u
#define PNY_RANGE_WEIGHT 4u
int pny_model_range_status(pny_model* m, unsigned* bits, int clear);

/* The ResNet-34 trunk in TRAINING mode (reference src/model/encoder.py:139-173 under autograd with the encoder unfrozen, the
 * default of train/train.py:66-73; batch norm on batch statistics as nn.BatchNorm2d in train()): forward and backward as this
 * library's kernels (csrc/encoder_train.hip).  Parameters are read where PyTorch keeps them: bind every `encoder.model.*`
 * tensor the trunk uses (conv `.weight`; bn `.weight`, `.bias`, `.running_mean`, `.running_var`) with pny_model_bind_param
 * first, and the gradient buffers of the trainable ones with pny_model_bind_grad.
 * pny_trunk_train_forward: images (n_images, 3, H, W) NCHW in [-1, 1] -> latent (n_images, 512, H/2, W/2) NCHW (the layout
 * PixelNeRFNet.encode(latent=) takes); running_mean / running_var are stepped in place with `momentum` (0.1 in the reference's
 * modules; 0 leaves them alone); bn_eval != 0 normalises with the running statistics instead (modules in eval() mode while
 * their parameters still train) and leaves them alone.  The activations stay in the model handle for ONE backward.
 * pny_trunk_train_backward: d loss / d latent (same shape) -> every bound gradient buffer is WRITTEN (conv weights in
 * (cout, cin, k, k), bn weight / bias).  Deterministic (no atomics). */
int pny_trunk_train_forward(pny_model* m, const float* images_dev, int n_images, int height, int width, float momentum, int bn_eval,
                            float* latent_nchw_dev, pny_stream stream);
int pny_trunk_train_backward(pny_model* m, const float* d_latent_nchw_dev, pny_stream stream);

/* Introspection for bench.py: GEMM FLOPs (2/MAC, unpadded, MLP only) of the last pny_render /
 * pny_query on this scene -- `flops` as executed by the fused kernel, `flops_reference` as the
 * reference's operation order would execute them (equal when the projection is off) --, the HIP-event
 * time of its MLP kernel launches (enable_timing), the launch count, and whether the last launch
 * used the projected latent.  Any out pointer may be NULL. */
int pny_scene_last_mlp_stats(pny_scene* s, double* flops, double* flops_reference, double* kernel_ms, int* launches,
                             int* projected);
int pny_scene_enable_timing(pny_scene* s, int enable);

/* ---- Backward pass (callers: train/trainlib/PixelNerfTrainer.py:133-156 `loss.backward()`) ----
 * What autograd computes in the reference for the parameters of mlp_coarse / mlp_fine through
 * NeRFRenderer.composite (src/render/nerf.py:229-250), the output head (src/model/models.py:312-317), the cross-view
 * mean (src/util/util.py:489-499) and ResnetFC.forward (src/model/resnetfc.py:134-186).  Exact-fp32 MFMA throughout.
 * The forward call is the ordinary pny_render / pny_query (its optional z / per-sample outputs are what the backward
 * needs); inside the backward call the MLP chain is evaluated once more in the reference's operation order with every
 * GEMM operand stashed in HBM (bounded by PNYOLO_STASH_GB, default 16: larger batches are processed in chunks), then
 * the dX chain and the weight-gradient GEMMs run over the stash.
 * The fine pass's depth samples depend on the coarse depth in the reference (nerf.py:156-167: no detach): that path
 * (gradient w.r.t. sample positions through the positional code, the projection and the bilinear latent lookup) is
 * included.  The latent is differentiated through pny_scene_bind_latent_grad (below), and the ResNet-34 trunk from there by
 * pny_trunk_train_backward; not differentiated: the rays, the cameras. */

/* Gradient target of the state_dict entry `name` ("mlp_coarse.blocks.2.fc_1.weight", ...): a device buffer of the
 * parameter's shape (fp32, contiguous) that the backward calls write / add into.  NULL unbinds.  Borrowed until
 * rebound; parameters without a bound target get no gradient. */
int pny_model_bind_grad(pny_model* m, const char* name, float* grad_dev);

/* Gradient w.r.t. the scene's latent -- the backward of `F.grid_sample` (src/model/encoder.py:101) composed with lin_z
 * (src/model/resnetfc.py:176-182), i.e. what reaches `encoder.latent` in the reference when the encoder trains:
 * grad_dev is a caller-owned, caller-zeroed fp32 buffer of the latent's shape in the library's layout (ns, Hl, Wl, L)
 * (channel-last); every backward call on the scene ADDS into it (float atomics: reproducible to fp32 rounding, not bit for
 * bit).  NULL unbinds.  d_latent must be a multiple of 256.  The gradient is handed to whatever produced the latent
 * (pny_scene_set_latent): pny_trunk_train_backward for the library's trunk, the caller's own backbone otherwise. */
int pny_scene_bind_latent_grad(pny_scene* s, float* grad_dev);

/* Backward of pny_query: d_out_dev (n, d_out) = dL/d(out).  accumulate = 0 overwrites the bound gradients of the
 * selected MLP, 1 adds to them. */
int pny_query_backward(pny_scene* s, const float* xyz_dev, const float* viewdirs_dev, int64_t n, int coarse,
                       const float* d_out_dev, int accumulate, pny_stream stream);

/* Backward of pny_composite (src/render/nerf.py:229-250): g_* = dL/d(rgb (n,3)), dL/d(depth (n)), dL/d(weights (n,k)),
 * any may be NULL.  d_sample_dev (n,k,4) = dL/d(per-sample [rgb, sigma]); d_z_dev (n,k), optional = dL/d(z) through
 * the deltas and the depth sum. */
int pny_composite_backward(const float* rays_dev, const float* z_dev, const float* sample_dev, int64_t n, int k,
                           int white_bkgd, const float* g_rgb_dev, const float* g_depth_dev, const float* g_weights_dev,
                           float* d_sample_dev, float* d_z_dev, pny_stream stream);

/* What the forward pny_render call left in its optional outputs (pny_render_out z_* / sample_*). */
typedef struct pny_render_saved {
    const float* z_coarse;      /* (n, n_coarse) */
    const float* sample_coarse; /* (n, n_coarse, 4) */
    const float* z_fine;        /* (n, n_coarse+n_fine) */
    const float* sample_fine;   /* (n, n_coarse+n_fine, 4) */
    const float* depth_coarse;  /* (n) the forward's coarse depth: the centre of the fine pass's depth samples.  Given, the
                                   fine loss is also propagated into mlp_coarse through those samples' positions, as the
                                   reference does (src/render/nerf.py:156-167, 296-298); NULL treats them as constants. */
} pny_render_saved;
/* Upstream gradients w.r.t. the outputs of pny_render; any may be NULL (= zero). */
typedef struct pny_render_grads {
    const float* rgb_coarse;     /* (n,3) */
    const float* depth_coarse;   /* (n) */
    const float* weights_coarse; /* (n,n_coarse) */
    const float* rgb_fine;
    const float* depth_fine;
    const float* weights_fine;
} pny_render_grads;
/* Backward of pny_render for one scene: composite backward + MLP backward of the fine pass (mlp_fine) and of the coarse
 * pass (mlp_coarse), into the bound gradients (accumulate as above; with one shared MLP the two passes add up).
 * accumulate bit 4: only the fine pass of this backward; bit 8: only the coarse pass (call with bit 4 first: the coarse
 * pass takes the depth-sample path's gradient the fine pass left in the scene) -- lets a caller put mlp_fine's
 * weight-gradient flush (pny_model_flush_weight_grads with bit 32) between the two. */
int pny_render_backward(pny_scene* s, const float* rays_dev, int64_t n, const pny_render_opts* opts,
                        const pny_render_saved* saved, const pny_render_grads* grads, int accumulate, pny_stream stream);

/* Backward of pny_yolo_render (src/render/yolo.py:96-114 aggregation + the MLP; caller: train/trainlib/YoloTrainer.py:160-186):
 * raw_dev (n, K, A*7) = the forward's raw_dev output, g_out_dev (n, A, 7) = dL/d(out); the sample depths are re-created
 * from u_coarse_dev / seed as the forward made them.  Gradients go to the bound targets of mlp_coarse. */
int pny_yolo_render_backward(pny_scene* s, const float* rays_dev, int64_t n, int n_coarse, const float* u_coarse_dev,
                             uint64_t seed, const float* raw_dev, const float* g_out_dev, int accumulate, pny_stream stream);

/* Deferred weight gradients.  A training batch holds several scenes (the reference's super-batch, SB objects x B rays,
 * train/train.py:23) whose backward calls are independent until the weight gradients are summed.  With deferral enabled
 * each pny_render_backward / pny_query_backward appends its tiles to a model-level stash (the calls of different scenes
 * may then run concurrently on different streams: nothing shared is written) and pny_model_flush_weight_grads runs ONE
 * weight-gradient GEMM per MLP over all of them (larger K per workgroup, one reduction).  coarse_tiles / fine_tiles =
 * 64-sample tiles to reserve for evaluations of mlp_coarse / mlp_fine: sum over the calls of ceil(points / 64).
 * PNY_ERR_ARG when the reservation exceeds the stash budget (the caller then stays in immediate mode).  The flush must be
 * ordered after the scenes' calls by the caller (same stream, or events). */
int pny_model_defer_weight_grads(pny_model* m, int enable, int ns, int64_t coarse_tiles, int64_t fine_tiles);
/* Stash in the forward.  With a reservation in place, enable = 1 makes the NEXT pny_render on this scene evaluate both
 * MLP passes with the stashing instantiation (reference operation order; rgb / sigma within fp32 rounding of the
 * projected evaluation) directly into the reservation; the pny_render_backward that follows (same reservation, z_* /
 * sample_* of that forward given in pny_render_saved) then skips its forward recompute and starts at the dX chain.
 * One-shot: the flag clears itself; a pass that does not fit the reservation runs the plain forward.  * accumulate bit 16: flush mlp_coarse's stash only; bit 32: mlp_fine's only (0: both, mlp_coarse first). */
int pny_scene_stash_next_render(pny_scene* s, int enable);
int pny_model_flush_weight_grads(pny_model* m, int accumulate, pny_stream stream);
/* GEMM FLOPs and HIP-event time of the last flush. */
int pny_model_last_flush_stats(pny_model* m, double* flops, double* kernel_ms);

/* Introspection for bench.py --mode train: GEMM FLOPs (2/MAC, unpadded) and HIP-event times (pny_scene_enable_timing)
 * of the three MLP kernels of the last pny_render_backward / pny_query_backward on this scene:
 * [0] stash forward (reference operation order), [1] dX chain, [2] weight-gradient GEMMs (+ reduction). */
int pny_scene_last_backward_stats(pny_scene* s, double flops[3], double kernel_ms[3]);

#ifdef __cplusplus
}
#endif
#endif /* PNYOLO_H */
