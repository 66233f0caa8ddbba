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
    // Source-view cameras travel in the kernel-argument segment (NS entries used; 64 B each): a launch carries its own
    // copy, so pny_scene_set_cameras touches no device memory (no copy, no synchronisation, no stream to order against)
    Cam cams[MAX_VIEWS];
};

enum { MLP_8x64 = 0, MLP_16x64 = 1, MLP_8x32 = 2 };  // kernel shapes (mlp.hip Cfg)
int mlp_pick_variant(long long n_points);           // shape for a launch of n_points samples
void launch_mlp(const MlpArgs& a, int variant, int grid, hipStream_t st);
int mlp_max_grid(int variant);      // resident workgroups = persistent grid size
int mlp_tile_samples(int variant);  // samples per workgroup tile (32 or 64)
size_t mlp_scratch_floats();

// render_kernels.hip
void launch_sample_coarse(const float* rays, long long n, int kc, int lindisp, const float* u, uint64_t seed,
                          float* z, hipStream_t st);
void launch_composite(const float* rays, const float* z, const float* samp, long long n, int k, int white,
                      float* w, float* rgb, float* depth, hipStream_t st);
void launch_sample_fine(const float* rays, const float* zc, const float* w, const float* depth, long long n,
                        int kc, int kf, int kfd, float depth_std, int lindisp, const float* u, const float* u2,
                        const float* g, uint64_t seed, float* zout, hipStream_t st);
void launch_yolo_aggregate(const float* raw, long long n, int k, int na, float* out, hipStream_t st);
void launch_gen_rays(const float* cam16_host, int b, int w, int h, float znear, float zfar, int yolo, float* out,
                     hipStream_t st, long long first, long long count);
void launch_nchw_to_nhwc(const float* in, float* out, int n, int c, int hw, hipStream_t st);
void launch_nhwc_to_nchw(const float* in, float* out, int n, int c, int hw, hipStream_t st);

// error plumbing
void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what);

}  // namespace pny

#define PNY_HIP(call)                                          \
    do {                                                       \
        hipError_t e_ = (call);                                \
        if (e_ != hipSuccess) return pny::hip_fail(e_, #call); \
    } while (0)
