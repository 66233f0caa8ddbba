// Private declarations shared by the host-side translation units of libpnyolo (api.hip, train_api.hip).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "pny_common.h"
#include "encoder.h"

namespace pny {

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int reserve(size_t need) {
        if (need <= bytes) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) return hip_fail(e, "hipMalloc(workspace)");
        bytes = need;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    float* f() const { return reinterpret_cast<float*>(p); }
};


int fail(int code, const std::string& msg);

// Small host -> device parameter tables that change from call to call (work lists of the weight-gradient GEMMs): staged
// through a pinned block; the block is rewritten only after the previous asynchronous copy out of it has executed.
struct PinnedStage {
    void* host = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
    int prepare(size_t bytes) {   // returns 0 and a writable block of >= bytes
        if (pending) {
            hipError_t e = hipEventSynchronize(ev);
            if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize(table stage)");
            pending = false;
        }
        if (bytes > cap) {
            if (host) (void)hipHostFree(host);
            host = nullptr;
            cap = 0;
            hipError_t e = hipHostMalloc(&host, bytes * 2, hipHostMallocDefault);
            if (e != hipSuccess) return hip_fail(e, "hipHostMalloc(table stage)");
            cap = bytes * 2;
        }
        if (!ev) {
            hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e != hipSuccess) return hip_fail(e, "hipEventCreate(table stage)");
        }
        return 0;
    }
    int upload(void* dst_dev, size_t bytes, hipStream_t st) {
        hipError_t e = hipMemcpyAsync(dst_dev, host, bytes, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(table stage)");
        e = hipEventRecord(ev, st);
        if (e != hipSuccess) return hip_fail(e, "hipEventRecord(table stage)");
        pending = true;
        return 0;
    }
    void release() {
        if (host) (void)hipHostFree(host);
        if (ev) (void)hipEventDestroy(ev);
        host = nullptr;
        ev = nullptr;
        cap = 0;
        pending = false;
    }
};

// One packed sub-buffer of the model and how to rebuild it from the state_dict tensor(s) it came from: recorded by
// pny_model_finalize, replayed on the device by pny_model_refresh (pack.hip).
struct RepackEntry {
    int kind;
    std::string name, name2;
    size_t dst_off;      // float offset in the packed blob ...
    float* dst_abs;      // ... or an absolute device pointer (projection weights)
    int n_out, k_in, k_pad, count;
};

// transposed packed weights of one MLP (A operands of the backward chain, mlp_bwd.hip)
struct MlpWeightsT {
    const float* w_in_plain;  // lin_in.weight as stored (512, d_in): input gradients (mlp_bwd.hip mlp_dz_kernel)
    const float* wT_out;
    const float* wT_fc0[MAX_BLOCKS];
    const float* wT_fc1[MAX_BLOCKS];
    const float* wzT_cat;     // [lin_z[0]^T | lin_z[1]^T | ...] (d_latent x n_view_blocks * 512), n-tile-major (latent_grad.hip)
    const float* h2_in;       // split-f16 images (mlp_h2.hip)
    const float* h2_fc0[MAX_BLOCKS];
    const float* h2_fc1[MAX_BLOCKS];
    const float* h3_in;       // split-f16 images in the 16 x 16 x 32 MFMA's operand order (mlp_h2w.hip)
    const float* h3_fc0[MAX_BLOCKS];
    const float* h3_fc1[MAX_BLOCKS];
    const float* h2T_out;     // split-f16 images of the transposed matrices (mlp_bwd_h2.hip)
    const float* h2T_fc0[MAX_BLOCKS];
    const float* h2T_fc1[MAX_BLOCKS];
};

}  // namespace pny

namespace pny {
struct TrunkTrain;                       // training-mode trunk: saved activations and scratch (encoder_train.hip)
void trunk_release(TrunkTrain* t);
}  // namespace pny

using namespace pny;  // private header of host-side translation units only

struct pny_model {
    pny_model_desc desc;
    std::map<std::string, HostTensor> host;  // state_dict tensors as loaded
    bool finalized = false;
    bool use_fine = true;
    DevBuf packed;                            // all MLP weights, one allocation
    MlpWeights coarse{}, fine{};
    MlpWeightsT coarse_t{}, fine_t{};        // transposed packs for the backward chain
    std::map<std::string, float*> grads;     // gradient targets by state_dict name (pny_model_bind_grad)
    std::map<std::string, const float*> params_dev;  // live parameter tensors on the device (pny_model_bind_param)
    std::vector<RepackEntry> repack;
    DevBuf repack_jobs;
    int n_repack_jobs = 0;
    long long repack_max_elems = 0;
    bool repack_ready = false;
    std::vector<struct pny_scene*> scenes;   // scenes created on this model (stream ordering of a refresh)
    // Deferred weight gradients (pny_model_defer_weight_grads): the scenes' backward calls append their tiles to these
    // model-level stashes ([0] mlp_coarse, [1] mlp_fine) and ONE weight-gradient GEMM per MLP runs at the flush.
    bool defer = false;
    int defer_ns = 0;
    DevBuf dx_stash[2], ddy_stash[2], d_partial[2], d_bias[2], d_tables[2];
    PinnedStage d_stage[2];
    long long defer_cap[2] = {0, 0}, defer_used[2] = {0, 0};
    DevBuf d_absmax;                         // two words: running max |dY| of the deferred tiles per MLP (train_api.hip)
    bool defer_dw_f32 = false;               // a scene pinned to F32 contributed: the flush runs the fp32 weight-gradient GEMM
    hipStream_t aux_stream = nullptr;        // side stream of the weight-gradient GEMMs' clipped tiles (mlp_bwd.hip launch_dw_gemm)
    hipEvent_t aux_fork = nullptr, aux_join = nullptr;
    uint64_t defer_epoch = 0;                // bumped by every pny_model_defer_weight_grads(enable)
    hipEvent_t flush_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    double flush_flops = 0.0;
    int flush_launches = 0;
    EncoderWeights enc;                       // folded conv+bn (encoder.h)
    DevBuf enc_batch_work, enc_batch_lat;     // pny_scenes_encode: workspace and result of one trunk pass over several scenes
    bool has_encoder = false;
    TrunkTrain* trunk = nullptr;              // created by the first pny_trunk_train_forward
    // lin_z[0..nvb) of the coarse / fine MLP stacked into one (nvb*512 x d_latent) pixel-wise map
    ConvLayer zproj[2];
    std::vector<float*> zproj_allocs;
    bool has_zproj = false;
    bool f16_weights_ok = true;   // every MLP weight is representable in the f16 range (checked at finalize; AUTO precision needs it)
    unsigned* range_flag = nullptr;           // pinned host word the f16x2 kernels report PNY_RANGE_* bits into (pny_model_range_status)
    uint64_t generation = 0;                  // bumped by every finalize (scenes re-project)
};

struct pny_scene {
    pny_model* m = nullptr;
    int ns = 0, L = 0, hl = 0, wl = 0;  // latent
    int width = 0, height = 0;
    bool have_cams = false, have_latent = false;
    int cam_ns = 0;
    int n_objs = 1;       // pny_scene_set_groups: the views are n_objs objects' view lists, ns / n_objs each (MlpArgs::obj_pts)
    Cam cams[MAX_VIEWS];  // host copy; handed to every MLP launch as kernel arguments
    DevBuf latent, work, scratch, enc_work;
    // projected latent of the coarse [0] / fine [1] MLP (see ensure_projection)
    DevBuf zp[2];
    bool zp_valid[2] = {false, false};
    uint64_t zp_generation = 0;
    int zp_mode = PNY_PROJECTION_AUTO;
    float* latent_grad = nullptr;   // (ns, hl, wl, L) caller-owned accumulator of d loss / d latent (pny_scene_bind_latent_grad)
    int precision = PNY_PRECISION_AUTO;   // matrix arithmetic of projected launches (pny_scene_set_precision)
    bool last_f16x2 = false;
    bool last_projected = false;
    double last_flops_ref = 0.0;
    // timing of the MLP launches of the last call
    bool timing = false;
    std::vector<hipEvent_t> ev;
    int ev_used = 0;
    double last_flops = 0.0;
    int last_launches = 0;
    // stream the last call on this scene was enqueued on (see enter_stream)
    // training workspace (train_api.hip)
    DevBuf dy_absmax;                        // one word: running max |dY| of the chunk in flight (non-deferred backward)
    DevBuf x_stash, dy_stash, dw_partial, dw_bias, dw_tables, d_samp, out_tmp, dz_tmp, sel_tmp, gdepth_tmp;
    PinnedStage table_stage;
    // "stash in the forward": the next pny_render evaluates the MLPs with the STASH instantiation straight into the
    // model-level stash (deferred mode); what each pass wrote is remembered for the backward of the same epoch
    bool stash_next = false;
    struct StashedPass {
        bool valid = false;
        uint64_t epoch = 0;
        int which = 0;
        long long tile0 = 0, tiles = 0, n_points = 0;
    } stashed[2];   // [0] coarse pass, [1] fine pass
    // HIP-event timing of the backward kernels of the last backward call (enable_timing): per MLP pass 4 events
    std::vector<hipEvent_t> bev;
    int bev_used = 0;
    double bwd_flops[3] = {0.0, 0.0, 0.0};   // stash forward, dX chain, weight-gradient GEMMs
    hipStream_t last_stream = nullptr;
    bool has_last_stream = false;
    hipEvent_t order_ev = nullptr;
    bool order_ev_valid = false;             // order_ev was recorded behind the last call's work (api.hip mark_stream_point)
};


namespace pny {
int enter_stream(pny_scene* s, hipStream_t st);
int check_ready(pny_scene* s, const char* who);
int view_blocks(const pny_model_desc& d);
inline int obj_views(const pny_scene* s) { return s->ns / (s->n_objs > 0 ? s->n_objs : 1); }   // views per object
StashLayout stash_layout(const pny_model_desc& d, int ns, int L);
// projected latent maps of the coarse (0) / fine (1) MLP, computed if stale; force = regardless of the scene's mode
int ensure_projection(pny_scene* s, int which, long long n_points, hipStream_t st, const float** zp, bool force = false);
// MlpArgs of a launch on this scene in the reference's operation order (no projected latent); tiles of 64 samples
int fill_mlp_args(pny_scene* s, int mode, const float* xyz, const float* dirs, const float* rays, const float* z, int K,
                  long long n_points, int coarse, float* out, MlpArgs* a);
}  // namespace pny
