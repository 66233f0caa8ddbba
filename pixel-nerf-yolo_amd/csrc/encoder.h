// ResNet-34 trunk of SpatialEncoder (reference src/model/encoder.py:139-173) on gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

namespace pny {

struct ConvLayer {
    float* w = nullptr;      // packed MFMA A-operand order [Cout/32][J][64][4], K = (ky,kx,ci)
    float* scale = nullptr;  // eval-mode batch norm: gamma / sqrt(var + eps)
    float* shift = nullptr;  // beta - mean * scale
    int cin = 0, cin_p = 0, cout = 0, k = 0, stride = 1, pad = 0, J = 0;
};

struct EncoderWeights {
    ConvLayer conv1;
    // layer1..3: per block conv1, conv2 and optional downsample
    struct Block {
        ConvLayer c1, c2, ds;
        bool has_ds = false;
    };
    std::vector<Block> layers[3];
    std::vector<float*> allocs;
    using Getter = std::function<bool(const std::string&, const float**, std::vector<int64_t>*)>;
    bool build(const Getter& get, const std::string& prefix, std::string* err);
    void release();
};

// Pixel-wise linear map over a channel-last tensor, out[p][n] = sum_k W[n][k] in[p][k], run by the
// same implicit-GEMM kernel as a 1x1 convolution (scale 1, shift 0, no relu).  `mats` are nmat
// row-major (rows x k) matrices stacked along n; rows*nmat must be a multiple of 64, k of 8.
bool build_pixel_linear(const float* const* mats, int nmat, int rows, int k, ConvLayer* out, std::vector<float*>* allocs,
                        std::string* err);
// f16x2: split-f16 matrix products (pixel_linear_h2_kernel) instead of the fp32 MFMA
bool run_pixel_linear(const ConvLayer& L, const float* in, long long npix, float* out, hipStream_t st, bool f16x2 = false);

// One convolution through conv_mfma_kernel (encoder.hip): out = relu?(conv(in) * L.scale + L.shift + resid), channel-last.
// hout / wout explicit; dil_shift > 0 reads the input as if 2^dil_shift - 1 zeros stood between its samples.
bool run_conv_ex(const ConvLayer& L, const float* in, int n, int hin, int win, int hout, int wout, int dil_shift, const float* resid,
                 int relu, float* out, hipStream_t st);
int conv_out(int in, int k, int s, int p);
void encoder_latent_size(int height, int width, int* hl, int* wl);
size_t encoder_workspace_bytes(int ns, int height, int width, bool use_first_pool);
// images (ns,3,H,W) NCHW -> latent (ns, H0, W0, 512) channel-last
bool encoder_forward(const EncoderWeights& w, const float* images, int ns, int height, int width, bool use_first_pool,
                     float* work, float* latent_nhwc, hipStream_t st, std::string* err);

}  // namespace pny
