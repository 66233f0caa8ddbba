// Shared declarations of libpnyolo's translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/pnyolo.h"

namespace pny {

constexpr int HID = 512;        // d_hidden this build is specialised for
constexpr int MLP_THREADS = 512;
constexpr int MAX_BLOCKS = 8;
constexpr int MAX_VIEWS = 16;
constexpr int D_IN_PAD = 64;    // 42 inputs padded to 8 k-iterations (a multiple of the weight-ring depth)
constexpr int ACT_KG = 128;     // k-groups (4 features each) held by the LDS activation buffer

// World->camera pose and intrinsics of one source view (reference models.py:74-87 buffers).
struct Cam {
    float w2c[12];  // 3x4 row-major
    float fx, fy, cx, cy;
};

// Device pointers of one packed ResnetFC (reference resnetfc.py:66-132).  512-wide layers are
// stored in MFMA A-operand order (see pack_layer in api.hip); lin_out stays row-major.
struct MlpWeights {
    const float* w_in;
    const float* b_in;
    const float* w_z[MAX_BLOCKS];
    const float* b_z[MAX_BLOCKS];
    const float* w_fc0[MAX_BLOCKS];
    const float* b_fc0[MAX_BLOCKS];
    const float* w_fc1[MAX_BLOCKS];
    const float* b_fc1[MAX_BLOCKS];
    const float* w_out;
    const float* b_out;
};

// Training stash (64-sample tiles; every tensor slot is [feature/4][64 samples] float4, i.e. the LDS B-operand layout):
//   X record of a tile  = NS views x { x_in (16 rows), z (L/4 rows), per view block b: relu(h_in(b)), relu(net(b)) }
//                         + post part { per post block: relu(h_in(b)), relu(net(b)) ; relu(h_top) }
//   dY record of a tile = NS views x { per view block b: dnet(b), dh_in(b) }
//                         + post part { d_raw (16 rows), dh_top, per post block b: dnet(b), dh_in(b) }
//     (dh_in of the first post-combine block holds dhm = dh / NS, the gradient every view's last block receives; when
//      there is no post-combine block dhm IS dh_top)
// Offsets are in floats.  SLOT = 64 * 512 floats.
struct StashLayout {
    long long x_tile, dy_tile;       // record strides
    int x_view, x_in, x_z, x_act;    // view stride; offsets inside a view
    int x_post;                      // offset of the post part inside a record
    int dy_view;                     // view stride (offset of a view's first slot is v * dy_view)
    int dy_post;                     // offset of the post part: d_raw, dh_top, then the post blocks
};
constexpr int STASH_SLOT = 64 * HID;
constexpr int STASH_SMALL = 64 * D_IN_PAD;  // 16-row slots: x_in, d_raw

struct MlpArgs {
    MlpWeights w;
    const float* w_base;  // the allocation all packed layers of `w` live in (raw-buffer addressing, mlp.hip)
    unsigned w_bytes;
    const float* latent;  // (NS, Hl, Wl, L) channel-last
    // Projected latent (NS, Hl, Wl, zp_stride = n_view_blocks*512): lin_z[b] applied to every latent
    // pixel once per scene (api.hip ensure_projection).  Non-null selects the kernel variant that
    // interpolates these maps instead of running the lin_z GEMMs per sample (both maps are linear).
    const float* zp;
    int zp_stride;
    int tap_stride;       // floats per pixel of the map the taps address (L, or zp_stride)
    // point source: mode 0 = explicit points, mode 1 = rays + depths (point = o + z d)
    const float* xyz;
    const float* dirs;
    const float* rays;
    const float* z;
    float* out;       // (n_points, d_out)
    float* scratch;   // gridDim.x * tile * HID floats: cross-view running sum
    long long n_points;
    int K;            // samples per ray (mode 1)
    int mode;
    int idx32;        // n_points (+ one tile) fits 32 bits: sample -> ray index with a 32-bit division
    int NS, L, Hl, Wl;
    int n_blocks, combine_layer, d_out, yolo, num_freqs;
    float freq_factor;
    float sx, sy;     // latent_scaling / image_size (reference encoder.py:97)
    int n_tiles;
    // Grouped scene (pny_scene_set_groups: the reference's super-batch, NeRFRenderer.forward flattening (SB, B, 8) rays and
    // PixelNeRFNet.forward conditioning object o on its own NS views, nerf.py:283-288, models.py:160-218): the launch's samples
    // belong to n_objs objects in equal consecutive shares of obj_pts samples, object o sees cams / latent views
    // [o NS, (o + 1) NS) of the scene's view list.  obj_pts is a multiple of 64, so a tile never straddles objects.  0: one object.
    long long obj_pts;
    // training: activation stash written by the STASH instantiation (launch_mlp_stash); null otherwise
    float* stash_x;
    StashLayout lay;
    // split-f16 operand images of lin_in / fc_0 / fc_1 (mlp_h2.hip; api.hip pack_layer_h2), inside the same blob as `w`
    const float* h2_in;
    const float* h2_fc0[MAX_BLOCKS];
    const float* h2_fc1[MAX_BLOCKS];
    // the same layers in the 16 x 16 x 32 MFMA's operand order (mlp_h2w.hip; api.hip pack_layer_h3)
    const float* h3_in;
    const float* h3_fc0[MAX_BLOCKS];
    const float* h3_fc1[MAX_BLOCKS];
    // f16-range guard (include/pnyolo.h pny_model_range_status): host-visible word the f16x2 kernels OR PNY_RANGE_* bits into
    unsigned* range_flag;
    // Source-view cameras travel in the kernel-argument segment (NS entries used; 64 B each): a launch carries its own
    // copy, so pny_scene_set_cameras touches no device memory (no copy, no synchronisation, no stream to order against)
    Cam cams[MAX_VIEWS];
};

// ---- device-side weight repack (pack.hip): the packed operand layouts rebuilt from the live parameter tensors
enum { PACK_A = 0, PACK_AT = 1, PACK_NT = 2, PACK_COPY = 3, PACK_ADD2 = 4, PACK_H2 = 5, PACK_NTT = 6, PACK_H2T = 7, PACK_H3 = 8 };
struct PackJob {
    const float* src;
    const float* src2;
    float* dst;
    int kind, n_out, k_in, k_pad, count;
};
void launch_repack(const PackJob* jobs_dev, int n_jobs, long long max_elems, hipStream_t st, unsigned* range_flag = nullptr);

// ---- backward pass (mlp_bwd.hip)
struct BwdArgs {
    // TRANSPOSED packed weights (A operands of dX^T = W^T dY^T), inside the model's packed blob
    const float* wT_out;              // lin_out^T: 512 x d_out, K padded to D_IN_PAD
    const float* wT_fc0[MAX_BLOCKS];
    const float* wT_fc1[MAX_BLOCKS];
    // the same matrices as split-f16 images (mlp_bwd_h2.hip; api.hip pack_mlp PACK_H2T); null when the fp32 chain runs
    const float* h2T_out;
    const float* h2T_fc0[MAX_BLOCKS];
    const float* h2T_fc1[MAX_BLOCKS];
    const float* w_base;
    unsigned w_bytes;
    const float* x_stash;             // written by the STASH forward
    float* dy_stash;
    StashLayout lay;
    const float* out;                 // (n_points, d_out) forward outputs (after the head)
    const float* d_out_grad;          // (n_points, d_out) gradient w.r.t. those outputs
    long long n_points;
    int n_tiles;
    int NS, n_blocks, combine_layer, d_out, yolo;
    unsigned* dy_absmax;              // optional: atomic max of the bit pattern of |v| over everything written to the dY stash
    unsigned* range_flag;             // f16-range guard word of the model (see MlpArgs)
};
// One weight-gradient GEMM: C[a_rows][x_cols] = sum over (tile, view) dY_slot^T X_slot.
struct DwJob {
    long long a_off, x_off;   // slot offsets inside a dY / X record (floats), view 0
    int a_view, x_view;       // per-view strides in floats (0: the slot exists once per tile)
    int n_views;              // views reduced over
    int a_rows, x_cols;       // rows of C (512, or 64 for lin_out) and columns (512, 64 for lin_in, L for lin_z)
};
// One workgroup's share: output tile (mt, nt) of 256 x 256 over the (tile, view) range [tv_lo, tv_hi).
struct DwItem {
    int job, mt, nt, tv_lo, tv_hi;
    long long part_off;       // float offset of the split's [a_rows][x_cols] block in the partial buffer
    long long bias_off;       // float offset of the split's [a_rows] block in the bias partial buffer
};
// Where a job's reduced result goes: weight (rows x cols valid, row-major, the state_dict layout) and up to two bias
// vectors that share the column sum (lin_z bias folded next to the preceding layer's).
struct DwTarget {
    float* w;
    float* b0;
    float* b1;
    int rows, cols;           // valid extent written to w
    int prows, pcols;         // extent of one split's partial block
    int splits;
    long long part_off, bias_off;
};
// Gradient w.r.t. the depths of selected samples through the MLP inputs (mlp_bwd.hip mlp_dz_kernel).
struct DzArgs {
    const float* dy_stash;
    StashLayout lay;
    const int* sel;       // n_sel global sample indices (ray * K + position) inside this chunk, or -1
    int n_sel;
    long long p0;         // global index of the chunk's first sample
    const float* rays;    // full arrays of the pass
    const float* z;
    int K;
    const float* w_in;    // lin_in.weight (512, d_in) row-major
    int d_in;
    const float* zp;      // projected maps of this MLP (api.hip ensure_projection) or null
    int zp_stride;
    int NS, Hl, Wl, nvb, npost, yolo, num_freqs;
    long long obj_pts;    // grouped scene (MlpArgs::obj_pts): samples per object, 0 = one object
    float freq_factor, sx, sy;
    float* dz;            // (n_points of the pass) accumulated
    Cam cams[MAX_VIEWS];
};
void launch_locate_depth_samples(const float* rays, const float* depth_c, const float* g, uint64_t seed, const float* z_fine,
                                 long long n, int kt, int kfd, float depth_std, int* sel, hipStream_t st);
void launch_mlp_dz(const DzArgs& a, hipStream_t st);
void launch_depth_grad_gather(const int* sel, const float* dz, const float* g_in, long long n, int kfd, float* g_out, hipStream_t st);
void launch_yolo_aggregate_bwd(const float* raw, const float* g, long long n, int k, int na, float* d_raw, hipStream_t st);
void launch_mlp_bwd(const BwdArgs& a, int grid, hipStream_t st);
void launch_mlp_bwd_h2(const BwdArgs& a, int grid, hipStream_t st);   // mlp_bwd_h2.hip: needs a.h2T_*
void launch_dw_gemm(const DwJob* jobs_dev, const DwItem* items_dev, int n_part, int n_full, const float* x_stash,
                    const float* dy_stash, long long x_tile, long long dy_tile, float* partial, float* bias_partial, hipStream_t st,
                    hipStream_t aux, hipEvent_t ev_fork, hipEvent_t ev_join, const unsigned* dy_absmax = nullptr);
void launch_dw_reduce(const DwTarget* targets_dev, int n_targets, long long max_elems, const float* partial,
                      const float* bias_partial, int accumulate, hipStream_t st);
void launch_composite_bwd(const float* rays, const float* z, const float* samp, const float* noise, long long n, int k,
                          int white, const float* g_rgb, const float* g_depth, const float* g_w, float* d_samp, float* d_z,
                          hipStream_t st);

enum { MLP_8x64 = 0, MLP_16x64 = 1, MLP_8x32 = 2 };  // kernel shapes (mlp.hip Cfg)
int mlp_pick_variant(long long n_points);           // shape for a launch of n_points samples
void launch_mlp(const MlpArgs& a, int variant, int grid, hipStream_t st);
void launch_mlp_stash(const MlpArgs& a, int grid, hipStream_t st);  // 8x64 shape, reference op order, writes a.stash_x
// latent gradient (latent_grad.hip): a.tap_stride must be a.L; grad is (NS, Hl, Wl, L) channel-last, added into
// dy_absmax (device word, the chain kernel's running max |dY|) selects the split-f16 kernel; null: fp32 MFMA
void launch_latent_grad(const MlpArgs& a, const float* dy_stash, const StashLayout& lay, const float* w_cat, float* grad, int nvb,
                        hipStream_t st, const unsigned* dy_absmax = nullptr);
bool mlp_h2_supports(int n_blocks, int combine_layer);
void launch_mlp_h2(const MlpArgs& a, int grid, hipStream_t st);
void launch_mlp_h2s(const MlpArgs& a, int grid, hipStream_t st);   // mlp_h2s.hip: a.n_tiles in 32-sample tiles, grid <= 2 x CUs
void launch_mlp_h2_stash(const MlpArgs& a, int grid, hipStream_t st);   // + the backward's operand stash (a.stash_x, a.lay)     // 8x64 shape, projected latent, split-f16 operands (mlp_h2.hip)
bool mlp_h2w_supports(int n_blocks, int combine_layer);
void launch_mlp_h2n(const MlpArgs& a, int grid, hipStream_t st);   // mlp_h2n.hip: the same kernel as 8 waves x 256 registers
void launch_mlp_h2w(const MlpArgs& a, int grid, hipStream_t st);   // mlp_h2w.hip: 4 waves x 512 registers, needs a.h3_*; 64-sample tiles
int mlp_max_grid(int variant);      // resident workgroups = persistent grid size
int mlp_tile_samples(int variant);  // samples per workgroup tile (32 or 64)
size_t mlp_scratch_floats();

// render_kernels.hip
void launch_sample_coarse(const float* rays, long long n, int kc, int lindisp, const float* u, uint64_t seed,
                          float* z, hipStream_t st);
void launch_composite(const float* rays, const float* z, const float* samp, long long n, int k, int white,
                      float* w, float* rgb, float* depth, hipStream_t st, const float* noise = nullptr);
void launch_sample_fine(const float* rays, const float* zc, const float* w, const float* depth, long long n,
                        int kc, int kf, int kfd, float depth_std, int lindisp, const float* u, const float* u2,
                        const float* g, uint64_t seed, float* zout, hipStream_t st);
void launch_yolo_aggregate(const float* raw, long long n, int k, int na, float* out, hipStream_t st);
void launch_gen_rays(const float* cam16_host, int b, int w, int h, float znear, float zfar, int yolo, float* out,
                     hipStream_t st, long long first, long long count);
void launch_nchw_to_nhwc(const float* in, float* out, int n, int c, int hw, hipStream_t st);
void launch_nhwc_to_nchw(const float* in, float* out, int n, int c, int hw, hipStream_t st);

// f16-range guard: bits of include/pnyolo.h PNY_RANGE_*; the word lives in pinned host memory (system-scope atomic)
__device__ __forceinline__ void range_report(unsigned* flag, unsigned bit) {
    if (flag) __hip_atomic_fetch_or(flag, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// error plumbing
void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what);

}  // namespace pny

#define PNY_HIP(call)                                          \
    do {                                                       \
        hipError_t e_ = (call);                                \
        if (e_ != hipSuccess) return pny::hip_fail(e_, #call); \
    } while (0)
