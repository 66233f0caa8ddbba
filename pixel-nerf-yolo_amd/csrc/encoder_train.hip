// ResNet-34 trunk of SpatialEncoder in TRAINING mode, forward and backward, as HIP kernels (gfx950).
//
// The reference trains the encoder by default (train/train.py:66-73 freezes it only with --freeze_enc;
// loss.backward() at train/trainlib/PixelNerfTrainer.py:156 differentiates src/model/encoder.py:139-173 through torchvision's
// BasicBlocks): convolutions, batch norm on BATCH statistics (the trunk is in train() mode), relu, residual adds, the 3x3/2
// max-pool and the bilinear pyramid.  Until round 3 that graph ran through ATen / MIOpen (11 ms of a
// 23.7 ms step, naive_conv_* kernels); here every piece is this library's:
//   forward  per convolution: conv_mfma_kernel (encoder.hip, the inference kernel with scale 1 / shift 0) on weights repacked
//            from the live parameters by ONE launch per step (trunk_pack_kernel), bn_stats_kernel (per-channel shifted sums over
//            fixed pixel ranges, deterministic), bn_apply_kernel (finalises the statistics in every workgroup in the same order,
//            normalises, adds the residual, relu; workgroup 0 also updates running_mean / running_var as nn.BatchNorm2d does);
//   backward per convolution: bn_bwd_stats_kernel (sum g, sum g x^ over the same pixel ranges, g = upstream gradient through the
//            relu mask), bn_bwd_apply_kernel (dy of the convolution output, d gamma, d beta), conv_dw_kernel (weight gradient:
//            a split-K GEMM over PIXELS on the fp32 MFMA, dW[co][(tap, ci)] = sum_p dy[p][co] patch[p][(tap, ci)], partials
//            reduced in a fixed order straight into the (cout, cin, k, k) gradient tensor), and the input gradient as a
//            transposed convolution through conv_mfma_kernel again (flipped, transposed filters from the same repack launch;
//            a stride-2 convolution's gradient is read as a dilated input);
//   maxpool_bwd_kernel and upsample_bwd_kernel are gathers (every input position collects from the windows / output pixels it
//            fed), so nothing in this file uses atomics: the trunk's gradients are bit-reproducible.
// All GEMMs are exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).  Activations channel-last, as in encoder.hip.
// Parity: tests/test_gpu_backward.py::test_encoder_training_gradients_vs_oracle_autograd (87 tensors against torch.autograd
// through the oracle's trunk, <= 1e-4 of each tensor's max).
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "api_internal.h"

namespace pny {

typedef float f32x16t __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------ weight repack
// kind 0: forward A-operand image [cout/32][J][64][4], K = (ky, kx, ci) padded to cin_p (encoder.hip build_conv);
// kind 1: the transposed convolution's image: rows = ci (cin of the convolution), K = (ky', kx', co) with FLIPPED taps,
//         W_t[ci][(ky', kx', co)] = W[co][ci][k-1-ky'][k-1-kx'];  J = k k cout / 8
struct TrunkPackJob {
    const float* src;   // (cout, cin, k, k) as PyTorch keeps it
    float* dst;
    int kind, cout, cin, cin_p, k, J, count;   // count = 16-byte elements
};

__global__ __launch_bounds__(256) void trunk_pack_kernel(const TrunkPackJob* __restrict__ jobs) {
    const TrunkPackJob jb = jobs[blockIdx.y];
    const int kk2 = jb.k * jb.k;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < jb.count; i += gridDim.x * blockDim.x) {
        const int l = i & 63, j = (i >> 6) % jb.J, nt = (i >> 6) / jb.J;
        const int n = 32 * nt + (l & 31), k0 = 8 * j + 4 * (l >> 5);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = k0 + r;
            float val = 0.f;
            if (jb.kind == 0) {
                const int tap = kk / jb.cin_p, ci = kk - tap * jb.cin_p;
                if (tap < kk2 && ci < jb.cin) val = jb.src[((size_t)n * jb.cin + ci) * kk2 + tap];
            } else {
                const int tap = kk / jb.cout, co = kk - tap * jb.cout;
                if (tap < kk2) val = jb.src[((size_t)co * jb.cin + n) * kk2 + (kk2 - 1 - tap)];   // flipped in y and x
            }
            v[r] = val;
        }
        reinterpret_cast<float4*>(jb.dst)[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ------------------------------------------------------------------------------------------------ batch norm, forward
// Per-channel statistics over P = N H W pixels of a channel-last (P, C) tensor.  A workgroup reduces the pixel range
// [b chunk, (b + 1) chunk); thread t owns channel quad t % (C / 4) and pixel lane t / (C / 4).  Sums are SHIFTED by the
// channel's value at pixel 0 (s1 = sum (y - K), s2 = sum (y - K)^2: var = s2 / P - (s1 / P)^2 without the cancellation of the
// plain sum of squares).  part[b][0 | 1][C].
constexpr int BN_THREADS = 256, BN_MAXB = 64;

__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(const float* __restrict__ y, long long P, int C, long long chunk,
                                                              float* __restrict__ part) {
    __shared__ float4 red[2][BN_THREADS];
    const int cq = C / 4, rows = BN_THREADS / cq;
    const int q = threadIdx.x % cq, r = threadIdx.x / cq;
    const long long p0 = (long long)blockIdx.x * chunk, p1 = p0 + chunk < P ? p0 + chunk : P;
    const float4 K = *reinterpret_cast<const float4*>(y + 4 * q);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    // four pixels per trip: the loads are independent, the sums stay in pixel order (the result does not depend on the unroll)
    for (long long p = p0 + r; p < p1; p += 4 * rows) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long pp = p + (long long)u * rows;
            v[u] = pp < p1 ? *reinterpret_cast<const float4*>(y + (size_t)pp * C + 4 * q) : K;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float dx = v[u].x - K.x, dy = v[u].y - K.y, dz = v[u].z - K.z, dw = v[u].w - K.w;
            s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
            s2.x += dx * dx; s2.y += dy * dy; s2.z += dz * dz; s2.w += dw * dw;
        }
    }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if (r == 0) {
        for (int i = 1; i < rows; ++i) {   // fixed order
            const float4 a = red[0][i * cq + q], b = red[1][i * cq + q];
            s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
            s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
        }
        *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 2 + 0) * C + 4 * q) = s1;
        *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 2 + 1) * C + 4 * q) = s2;
    }
}

// out = relu?((y - mean) invstd gamma + beta + resid).  Every workgroup finalises the statistics itself from the B partials
// (same order everywhere); workgroup 0 stores mean / invstd for the backward and steps the running statistics
// (nn.BatchNorm2d, momentum 0.1, unbiased variance).
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const float* __restrict__ y, const float* __restrict__ part, int B,
                                                              long long P, int C, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ resid,
                                                              int relu, float* __restrict__ out, float* __restrict__ mean_out,
                                                              float* __restrict__ invstd_out, float* run_mean, float* run_var,
                                                              float momentum, float eps, int use_running) {
    __shared__ float s_mean[256], s_scale[256], s_shift[256];
    for (int c = threadIdx.x; c < C; c += BN_THREADS) {
        float s1 = 0.f, s2 = 0.f;
        for (int b = 0; b < B; ++b) {
            s1 += part[((size_t)b * 2 + 0) * C + c];
            s2 += part[((size_t)b * 2 + 1) * C + c];
        }
        const float m = s1 / (float)P;
        float mean = y[c] + m;
        float var = s2 / (float)P - m * m;
        var = var > 0.f ? var : 0.f;
        if (use_running) {   // eval()-mode batch norm under autograd: the running statistics, which stay as they are
            mean = run_mean[c];
            var = run_var[c];
        }
        const float invstd = 1.0f / sqrtf(var + eps);
        s_mean[c] = mean;
        s_scale[c] = invstd * gamma[c];
        s_shift[c] = beta[c];
        if (blockIdx.x == 0) {
            mean_out[c] = mean;
            invstd_out[c] = invstd;
            if (run_mean && !use_running && momentum > 0.f) {
                run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean;
                const float unbiased = P > 1 ? var * ((float)P / (float)(P - 1)) : var;
                run_var[c] = (1.0f - momentum) * run_var[c] + momentum * unbiased;
            }
        }
    }
    __syncthreads();
    const int cq = C / 4;
    const long long total = P * cq;
    for (long long i = (long long)blockIdx.x * BN_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * BN_THREADS) {
        const int q = (int)(i % cq);
        const float4 v = *reinterpret_cast<const float4*>(y + i * 4);
        float4 o;
        o.x = (v.x - s_mean[4 * q + 0]) * s_scale[4 * q + 0] + s_shift[4 * q + 0];
        o.y = (v.y - s_mean[4 * q + 1]) * s_scale[4 * q + 1] + s_shift[4 * q + 1];
        o.z = (v.z - s_mean[4 * q + 2]) * s_scale[4 * q + 2] + s_shift[4 * q + 2];
        o.w = (v.w - s_mean[4 * q + 3]) * s_scale[4 * q + 3] + s_shift[4 * q + 3];
        if (resid) {
            const float4 r = *reinterpret_cast<const float4*>(resid + i * 4);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (relu) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        *reinterpret_cast<float4*>(out + i * 4) = o;
    }
}

// ------------------------------------------------------------------------------------------------ batch norm, backward
// g = d_out * (out > 0) (out == nullptr: no relu behind this batch norm);  x^ = (y - mean) invstd;
// part[b][0][c] = sum g, part[b][1][c] = sum g x^ over the workgroup's pixel range.
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_stats_kernel(const float* __restrict__ d_out, const float* __restrict__ out,
                                                                  const float* __restrict__ y, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, long long P, int C,
                                                                  long long chunk, float* __restrict__ part) {
    __shared__ float4 red[2][BN_THREADS];
    const int cq = C / 4, rows = BN_THREADS / cq;
    const int q = threadIdx.x % cq, r = threadIdx.x / cq;
    const long long p0 = (long long)blockIdx.x * chunk, p1 = p0 + chunk < P ? p0 + chunk : P;
    const float4 mu = *reinterpret_cast<const float4*>(mean + 4 * q), is = *reinterpret_cast<const float4*>(invstd + 4 * q);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    for (long long p = p0 + r; p < p1; p += 4 * rows) {
        float4 gg[4], aa[4], vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long pp = p + (long long)u * rows;
            const bool ok = pp < p1;
            const size_t o = (size_t)(ok ? pp : p) * C + 4 * q;
            gg[u] = *reinterpret_cast<const float4*>(d_out + o);
            aa[u] = out ? *reinterpret_cast<const float4*>(out + o) : make_float4(1.f, 1.f, 1.f, 1.f);
            vv[u] = *reinterpret_cast<const float4*>(y + o);
            if (!ok) gg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float4 g = gg[u];
            const float4 a = aa[u], v = vv[u];
            g.x = a.x > 0.f ? g.x : 0.f; g.y = a.y > 0.f ? g.y : 0.f; g.z = a.z > 0.f ? g.z : 0.f; g.w = a.w > 0.f ? g.w : 0.f;
            s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
            s2.x += g.x * ((v.x - mu.x) * is.x); s2.y += g.y * ((v.y - mu.y) * is.y);
            s2.z += g.z * ((v.z - mu.z) * is.z); s2.w += g.w * ((v.w - mu.w) * is.w);
        }
    }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if (r == 0) {
        for (int i = 1; i < rows; ++i) {
            const float4 a = red[0][i * cq + q], b = red[1][i * cq + q];
            s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
            s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
        }
        *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 2 + 0) * C + 4 * q) = s1;
        *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 2 + 1) * C + 4 * q) = s2;
    }
}

// dy = gamma invstd (g - mean(g) - x^ mean(g x^));  d gamma = sum g x^, d beta = sum g (workgroup 0; written, not added: the
// gradient buffers are fresh);  g_out (optional) = the masked upstream gradient, which is also the gradient of the block's
// identity branch.
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(const float* __restrict__ d_out, const float* __restrict__ out,
                                                                  const float* __restrict__ y, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                  const float* __restrict__ part, int B, long long P, int C,
                                                                  float* __restrict__ dy, float* __restrict__ g_out,
                                                                  float* __restrict__ d_gamma, float* __restrict__ d_beta,
                                                                  int use_running) {
    __shared__ float s_mean[256], s_is[256], s_a[256], s_mg[256], s_mgx[256];
    for (int c = threadIdx.x; c < C; c += BN_THREADS) {
        float s1 = 0.f, s2 = 0.f;
        for (int b = 0; b < B; ++b) {
            s1 += part[((size_t)b * 2 + 0) * C + c];
            s2 += part[((size_t)b * 2 + 1) * C + c];
        }
        s_mean[c] = mean[c];
        s_is[c] = invstd[c];
        s_a[c] = gamma[c] * invstd[c];
        // eval()-mode statistics do not depend on the batch: dy = gamma invstd g, no mean terms
        s_mg[c] = use_running ? 0.f : s1 / (float)P;
        s_mgx[c] = use_running ? 0.f : s2 / (float)P;
        if (blockIdx.x == 0) {
            if (d_gamma) d_gamma[c] = s2;
            if (d_beta) d_beta[c] = s1;
        }
    }
    __syncthreads();
    const int cq = C / 4;
    const long long total = P * cq;
    for (long long i = (long long)blockIdx.x * BN_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * BN_THREADS) {
        const int c0 = 4 * (int)(i % cq);
        float4 g = *reinterpret_cast<const float4*>(d_out + i * 4);
        if (out) {
            const float4 a = *reinterpret_cast<const float4*>(out + i * 4);
            g.x = a.x > 0.f ? g.x : 0.f; g.y = a.y > 0.f ? g.y : 0.f; g.z = a.z > 0.f ? g.z : 0.f; g.w = a.w > 0.f ? g.w : 0.f;
        }
        const float4 v = *reinterpret_cast<const float4*>(y + i * 4);
        float4 o;
        o.x = s_a[c0 + 0] * (g.x - s_mg[c0 + 0] - (v.x - s_mean[c0 + 0]) * s_is[c0 + 0] * s_mgx[c0 + 0]);
        o.y = s_a[c0 + 1] * (g.y - s_mg[c0 + 1] - (v.y - s_mean[c0 + 1]) * s_is[c0 + 1] * s_mgx[c0 + 1]);
        o.z = s_a[c0 + 2] * (g.z - s_mg[c0 + 2] - (v.z - s_mean[c0 + 2]) * s_is[c0 + 2] * s_mgx[c0 + 2]);
        o.w = s_a[c0 + 3] * (g.w - s_mg[c0 + 3] - (v.w - s_mean[c0 + 3]) * s_is[c0 + 3] * s_mgx[c0 + 3]);
        *reinterpret_cast<float4*>(dy + i * 4) = o;
        if (g_out) *reinterpret_cast<float4*>(g_out + i * 4) = g;
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[co][(tap, ci)] = sum over pixels p of dy[p][co] x[pixel of tap (ky, kx) at p][ci]: a GEMM whose contraction runs over the
// N H W output pixels.  A wave owns a 64 (co) x 128 (tap, ci) tile of the result for one slice of the pixels: 2 x 4 accumulator
// tiles of v_mfma_f32_32x32x2_f32; a k-step is TWO pixels (lane half hh takes pixel p + hh).  The row / column a lane feeds is
// free as long as the epilogue agrees, so a lane takes CONSECUTIVE channels -- 2 of co (one 8-byte load of dy), 4 of (tap, ci)
// (one 16-byte load of x: cin_p is a multiple of 4, so the four share a tap) -- and MFMA (i, j) multiplies component i of the
// one by component j of the other: rows 2 m + i, columns 4 m + j.  Partials [split][co][Kp] are summed by
// conv_dw_reduce_kernel in a fixed order.
struct DwcArgs {
    const float* dy;   // (P, cout)
    const float* x;    // (n, hin, win, cin_p)
    float* partial;    // [splits][cout][Kp]
    int n, hin, win, cin_p, hout, wout, cout, k, stride, pad, Kp, ntiles, splits;
    long long npix, chunk;
};

__global__ __launch_bounds__(256) void conv_dw_kernel(const DwcArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, hh = lane >> 5;
    const long long item = (long long)blockIdx.x * 4 + wave;
    const int mtiles = a.cout / 64;
    const long long n_items = (long long)mtiles * a.ntiles * a.splits;
    if (item >= n_items) return;
    const int split = (int)(item / (mtiles * a.ntiles));
    const int rem = (int)(item - (long long)split * mtiles * a.ntiles);
    const int mt = rem / a.ntiles, nt = rem - mt * a.ntiles;
    const int co = 64 * mt + 2 * m;
    const int nn = 128 * nt + 4 * m;
    const bool n_ok = nn < a.Kp;
    const int tap = n_ok ? nn / a.cin_p : 0, ci = n_ok ? nn - tap * a.cin_p : 0;
    const int ky = tap / a.k, kx = tap - ky * a.k;
    const long long p_lo = (long long)split * a.chunk, p_hi = p_lo + a.chunk < a.npix ? p_lo + a.chunk : a.npix;
    const int per = a.hout * a.wout;
    f32x16t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto fetch = [&](long long p, float2& A, float4& Bv) {
        A = make_float2(0.f, 0.f);
        Bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < p_hi) {
            A = *reinterpret_cast<const float2*>(a.dy + (size_t)p * a.cout + co);
            const int img = (int)(p / per), r = (int)(p - (long long)img * per);
            const int oy = r / a.wout, ox = r - oy * a.wout;
            const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
            if (n_ok && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win)
                Bv = *reinterpret_cast<const float4*>(a.x + (((size_t)img * a.hin + iy) * a.win + ix) * a.cin_p + ci);
        }
    };
    // eight pixels (four k-steps) per trip, the next trip's operands in flight underneath: the slices are short (tens of
    // pixels), so what a launch costs is the latency of its dependent loads
    float2 A0[4], A1[4];
    float4 B0[4], B1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) fetch(p_lo + 2 * u + hh, A0[u], B0[u]);
#define PNY_DW_STEP(A, Bv)                                                                                   \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, Bv.x, acc[0][0], 0, 0, 0);                           \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, Bv.y, acc[0][1], 0, 0, 0);                           \
    acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, Bv.z, acc[0][2], 0, 0, 0);                           \
    acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, Bv.w, acc[0][3], 0, 0, 0);                           \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, Bv.x, acc[1][0], 0, 0, 0);                           \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, Bv.y, acc[1][1], 0, 0, 0);                           \
    acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, Bv.z, acc[1][2], 0, 0, 0);                           \
    acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, Bv.w, acc[1][3], 0, 0, 0);
    for (long long p = p_lo; p < p_hi; p += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) fetch(p + 8 + 2 * u + hh, A1[u], B1[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) { PNY_DW_STEP(A0[u], B0[u]) }
        if (p + 8 < p_hi) {
#pragma unroll
            for (int u = 0; u < 4; ++u) fetch(p + 16 + 2 * u + hh, A0[u], B0[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) { PNY_DW_STEP(A1[u], B1[u]) }
        }
    }
#undef PNY_DW_STEP
    // accumulator (i, j), register 4 q + r of lane (m, hh): row (A lane) 8 q + 4 hh + r, column (B lane) m
    //   -> co = 64 mt + 2 (8 q + 4 hh + r) + i,  (tap, ci) = 128 nt + 4 m + j
    if (!n_ok) return;
    float* dst = a.partial + ((size_t)split * a.cout) * a.Kp;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 64 * mt + 2 * (8 * q + 4 * hh + r) + i;
                const float4 v = make_float4(acc[i][0][4 * q + r], acc[i][1][4 * q + r], acc[i][2][4 * q + r], acc[i][3][4 * q + r]);
                *reinterpret_cast<float4*>(dst + (size_t)row * a.Kp + nn) = v;
            }
}

// dW (cout, cin, k, k) = sum over the splits, fixed order
__global__ void conv_dw_reduce_kernel(const float* __restrict__ partial, int splits, int cout, int cin, int cin_p, int k, int Kp,
                                      float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int kk2 = k * k;
    if (i >= cout * cin * kk2) return;
    const int tap = i % kk2, ci = (i / kk2) % cin, co = i / (kk2 * cin);
    float s = 0.f;
    for (int sp = 0; sp < splits; ++sp) s += partial[((size_t)sp * cout + co) * Kp + tap * cin_p + ci];
    dw[i] = s;
}

// ------------------------------------------------------------------------------------------------ pool / pyramid, backward
// max_pool2d(3, stride 2, pad 1) backward as a gather: input position (iy, ix) collects the gradient of every window whose
// FIRST maximum in scan order it is (ATen keeps the first maximum: `val > maxval`).  d_in = result (+ add, optional).
__global__ void maxpool_bwd_kernel(const float* __restrict__ in, const float* __restrict__ g, const float* __restrict__ add,
                                   float* __restrict__ d_in, int n, int hin, int win, int c, int hout, int wout) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cq = c / 4;
    if (i >= (long long)n * hin * win * cq) return;
    const int q = (int)(i % cq);
    long long p = i / cq;
    const int ix = (int)(p % win);
    p /= win;
    const int iy = (int)(p % hin), img = (int)(p / hin);
    const float* base = in + (size_t)img * hin * win * c + 4 * q;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int oy_lo = iy > 0 ? (iy) / 2 : 0, oy_hi = (iy + 1) / 2;   // windows oy with 2 oy - 1 <= iy <= 2 oy + 1
    const int ox_lo = ix > 0 ? (ix) / 2 : 0, ox_hi = (ix + 1) / 2;
    for (int oy = oy_lo; oy <= oy_hi && oy < hout; ++oy)
        for (int ox = ox_lo; ox <= ox_hi && ox < wout; ++ox) {
            if (iy < 2 * oy - 1 || iy > 2 * oy + 1 || ix < 2 * ox - 1 || ix > 2 * ox + 1) continue;
            float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            int arg[4] = {-1, -1, -1, -1};
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx) {
                    const int yy = oy * 2 - 1 + dy, xx = ox * 2 - 1 + dx;
                    if (yy < 0 || yy >= hin || xx < 0 || xx >= win) continue;
                    const float4 v = *reinterpret_cast<const float4*>(base + ((size_t)yy * win + xx) * c);
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (vv[e] > best[e]) {
                            best[e] = vv[e];
                            arg[e] = yy * win + xx;
                        }
                }
            const float4 gg = *reinterpret_cast<const float4*>(g + (((size_t)img * hout + oy) * wout + ox) * c + 4 * q);
            const float gv[4] = {gg.x, gg.y, gg.z, gg.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (arg[e] == iy * win + ix) acc[e] += gv[e];
        }
    float4 o = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (add) {
        const float4 t = *reinterpret_cast<const float4*>(add + i * 4);
        o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
    }
    *reinterpret_cast<float4*>(d_in + i * 4) = o;
}

// Backward of upsample_concat_kernel (bilinear, align_corners = True) for one pyramid level, as a gather: low-resolution
// position (y, x) collects w_y w_x d_lat[oy][ox] from the output pixels whose footprint holds it, with the weights the forward
// computed.  d_lat (n, h0, w0, ctot), channels [coff, coff + c); result (n, hin, win, c) (+ add, optional).
__global__ void upsample_bwd_kernel(const float* __restrict__ d_lat, const float* __restrict__ add, float* __restrict__ d_in, int n,
                                    int hin, int win, int c, int h0, int w0, int ctot, int coff) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cq = c / 4;
    if (i >= (long long)n * hin * win * cq) return;
    const int q = (int)(i % cq);
    long long p = i / cq;
    const int x = (int)(p % win);
    p /= win;
    const int y = (int)(p % hin), img = (int)(p / hin);
    const float sy = h0 > 1 ? (float)(hin - 1) / (float)(h0 - 1) : 0.f;
    const float sx = w0 > 1 ? (float)(win - 1) / (float)(w0 - 1) : 0.f;
    // candidate output rows: sy * oy in (y - 1, y + 1)
    int oy_lo = sy > 0.f ? (int)floorf((float)(y - 1) / sy) : 0, oy_hi = sy > 0.f ? (int)ceilf((float)(y + 1) / sy) : h0 - 1;
    int ox_lo = sx > 0.f ? (int)floorf((float)(x - 1) / sx) : 0, ox_hi = sx > 0.f ? (int)ceilf((float)(x + 1) / sx) : w0 - 1;
    oy_lo = oy_lo < 0 ? 0 : oy_lo;
    ox_lo = ox_lo < 0 ? 0 : ox_lo;
    oy_hi = oy_hi > h0 - 1 ? h0 - 1 : oy_hi;
    ox_hi = ox_hi > w0 - 1 ? w0 - 1 : ox_hi;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        const float fy = sy * (float)oy;
        const int y0 = (int)fy, y1 = y0 + (y0 < hin - 1 ? 1 : 0);
        const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
        const float wy = (y0 == y ? ly0 : 0.f) + (y1 == y ? ly1 : 0.f);
        if (wy == 0.f) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const float fx = sx * (float)ox;
            const int x0 = (int)fx, x1 = x0 + (x0 < win - 1 ? 1 : 0);
            const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
            const float wx = (x0 == x ? lx0 : 0.f) + (x1 == x ? lx1 : 0.f);
            if (wx == 0.f) continue;
            const float4 g = *reinterpret_cast<const float4*>(d_lat + (((size_t)img * h0 + oy) * w0 + ox) * ctot + coff + 4 * q);
            const float w = wy * wx;
            acc.x += w * g.x; acc.y += w * g.y; acc.z += w * g.z; acc.w += w * g.w;
        }
    }
    if (add) {
        const float4 t = *reinterpret_cast<const float4*>(add + i * 4);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    *reinterpret_cast<float4*>(d_in + i * 4) = acc;
}

// dst = a + b (float4 elements); b may be null
__global__ void add2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ dst, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 v = reinterpret_cast<const float4*>(a)[i];
    if (b) {
        const float4 t = reinterpret_cast<const float4*>(b)[i];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    reinterpret_cast<float4*>(dst)[i] = v;
}

// (n, h, w, 4) image -> same kernel as encoder.hip's (declared there as a __global__; re-stated to keep the units separate)
__global__ void trunk_image_to_nhwc4_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int hw) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * hw) return;
    const int img = (int)(i / hw), p = (int)(i - (long long)img * hw);
    const float* s = in + (size_t)img * 3 * hw + p;
    *reinterpret_cast<float4*>(out + i * 4) = make_float4(s[0], s[hw], s[2 * (size_t)hw], 0.f);
}
__global__ void trunk_maxpool_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int hin, int win, int c, int hout,
                                     int wout) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cq = c / 4;
    if (i >= (long long)n * hout * wout * cq) return;
    const int q = (int)(i % cq);
    long long p = i / cq;
    const int ox = (int)(p % wout);
    p /= wout;
    const int oy = (int)(p % hout), img = (int)(p / hout);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int dy = 0; dy < 3; ++dy)
        for (int dx = 0; dx < 3; ++dx) {
            const int iy = oy * 2 - 1 + dy, ix = ox * 2 - 1 + dx;
            if (iy < 0 || iy >= hin || ix < 0 || ix >= win) continue;
            const float4 v = *reinterpret_cast<const float4*>(in + (((size_t)img * hin + iy) * win + ix) * c + 4 * q);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    *reinterpret_cast<float4*>(out + i * 4) = m;
}
__global__ void trunk_upsample_concat_kernel(const float* __restrict__ in, float* __restrict__ lat, int n, int hin, int win, int c,
                                             int h0, int w0, int ctot, int coff) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cq = c / 4;
    if (i >= (long long)n * h0 * w0 * cq) return;
    const int q = (int)(i % cq);
    long long p = i / cq;
    const int ox = (int)(p % w0);
    p /= w0;
    const int oy = (int)(p % h0), img = (int)(p / h0);
    const float sy = h0 > 1 ? (float)(hin - 1) / (float)(h0 - 1) : 0.f;
    const float sx = w0 > 1 ? (float)(win - 1) / (float)(w0 - 1) : 0.f;
    const float fy = sy * (float)oy, fx = sx * (float)ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < hin - 1 ? 1 : 0), x1 = x0 + (x0 < win - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
    const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const float* b = in + (size_t)img * hin * win * c + 4 * q;
    const float4 v00 = *reinterpret_cast<const float4*>(b + ((size_t)y0 * win + x0) * c);
    const float4 v01 = *reinterpret_cast<const float4*>(b + ((size_t)y0 * win + x1) * c);
    const float4 v10 = *reinterpret_cast<const float4*>(b + ((size_t)y1 * win + x0) * c);
    const float4 v11 = *reinterpret_cast<const float4*>(b + ((size_t)y1 * win + x1) * c);
    float4 o;
    o.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
    o.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
    o.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
    o.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
    *reinterpret_cast<float4*>(lat + (((size_t)img * h0 + oy) * w0 + ox) * ctot + coff + 4 * q) = o;
}

// ------------------------------------------------------------------------------------------------ host side
// One convolution + batch norm of the trunk, with everything the backward needs
struct TrunkUnit {
    std::string conv, bn;         // state_dict prefixes ("encoder.model.layer1.0.conv1", "...bn1")
    int cin, cin_p, cout, k, stride, pad;
    int hin, win, hout, wout;
    const float* x = nullptr;     // input activation (n, hin, win, cin_p)
    const float* resid = nullptr; // residual added before the relu (or null)
    bool relu = true;
    float *y = nullptr, *out = nullptr, *mean = nullptr, *invstd = nullptr;
    float *w_fwd = nullptr, *w_t = nullptr;   // packed operands (forward; transposed-flipped for the input gradient)
    int J = 0, Jt = 0;
};

struct TrunkTrain {
    DevBuf work, packs, jobs, part, dwpart, ones;
    std::vector<TrunkUnit> units;
    int n = 0, height = 0, width = 0;
    bool pool = true;
    int h[4] = {0, 0, 0, 0}, w[4] = {0, 0, 0, 0};
    float* img4 = nullptr;
    float* pooled = nullptr;
    const float* level_out[4] = {nullptr, nullptr, nullptr, nullptr};
    // gradient scratch (carved from `work`)
    static constexpr int NSCR = 8;
    float* scr[NSCR] = {};
    float* d_level[4] = {nullptr, nullptr, nullptr, nullptr};
    float* lat_nhwc = nullptr;
    // weight gradients run on a side stream beside the chain of input gradients (which is the critical path); an event per
    // scratch buffer says when the side stream's last reader of it is done
    hipStream_t side = nullptr;
    hipEvent_t ev_ready = nullptr, ev_join = nullptr, ev_buf[NSCR] = {};
    bool pend[NSCR] = {};
    bool have_forward = false;
    bool bn_eval = false;    // batch norm on the running statistics (modules in eval() mode under autograd)
    int n_jobs = 0;
};

void trunk_release(TrunkTrain* t) {
    if (!t) return;
    t->work.release();
    t->packs.release();
    t->jobs.release();
    t->part.release();
    t->dwpart.release();
    t->ones.release();
    if (t->side) (void)hipStreamDestroy(t->side);
    if (t->ev_ready) (void)hipEventDestroy(t->ev_ready);
    if (t->ev_join) (void)hipEventDestroy(t->ev_join);
    for (auto& e : t->ev_buf)
        if (e) (void)hipEventDestroy(e);
    delete t;
}

static size_t al64(size_t x) { return (x + 63) & ~(size_t)63; }
static void dw_split(const TrunkUnit& u, int n, int* splits, long long* chunk);

static int find_param(pny_model* m, const std::string& name, const float** out) {
    auto it = m->params_dev.find(name);
    if (it == m->params_dev.end()) return fail(PNY_ERR_STATE, "trunk training: no device pointer bound for '" + name + "' (pny_model_bind_param)");
    *out = it->second;
    return 0;
}
static float* find_grad(pny_model* m, const std::string& name) {
    auto it = m->grads.find(name);
    return it == m->grads.end() ? nullptr : it->second;
}

// lays out units, activations and packs for n images of height x width; (re)allocates when the shape changed
static int trunk_plan(pny_model* m, TrunkTrain& T, int n, int height, int width) {
    const bool pool = m->desc.enc_use_first_pool != 0;
    if (T.n == n && T.height == height && T.width == width && T.pool == pool && !T.units.empty()) return 0;
    T.units.clear();
    T.n = n;
    T.height = height;
    T.width = width;
    T.pool = pool;
    T.h[0] = conv_out(height, 7, 2, 3);
    T.w[0] = conv_out(width, 7, 2, 3);
    T.h[1] = pool ? conv_out(T.h[0], 3, 2, 1) : T.h[0];
    T.w[1] = pool ? conv_out(T.w[0], 3, 2, 1) : T.w[0];
    for (int i = 2; i < 4; ++i) {
        T.h[i] = conv_out(T.h[i - 1], 3, 2, 1);
        T.w[i] = conv_out(T.w[i - 1], 3, 2, 1);
    }
    const std::string pre = "encoder.model.";
    auto unit = [&](const std::string& conv, const std::string& bn, int cin, int cout, int k, int stride, int pad, int hin, int win) {
        TrunkUnit u;
        u.conv = pre + conv;
        u.bn = pre + bn;
        u.cin = cin;
        u.cin_p = (cin + 3) / 4 * 4;
        u.cout = cout;
        u.k = k;
        u.stride = stride;
        u.pad = pad;
        u.hin = hin;
        u.win = win;
        u.hout = conv_out(hin, k, stride, pad);
        u.wout = conv_out(win, k, stride, pad);
        u.J = (k * k * u.cin_p + 7) / 8;
        u.Jt = k * k * cout / 8;
        T.units.push_back(u);
    };
    unit("conv1", "bn1", 3, 64, 7, 2, 3, height, width);
    const int couts[3] = {64, 128, 256}, nblk[3] = {3, 4, 6};
    int cin = 64, hin = T.h[1], win = T.w[1];
    for (int li = 0; li < 3; ++li)
        for (int b = 0; b < nblk[li]; ++b) {
            const std::string p = "layer" + std::to_string(li + 1) + "." + std::to_string(b) + ".";
            const int stride = (b == 0 && li > 0) ? 2 : 1;
            const int bc = b == 0 ? cin : couts[li];
            if (b == 0 && (stride != 1 || bc != couts[li])) unit(p + "downsample.0", p + "downsample.1", bc, couts[li], 1, stride, 0, hin, win);
            unit(p + "conv1", p + "bn1", bc, couts[li], 3, stride, 1, hin, win);
            hin = conv_out(hin, 3, stride, 1);
            win = conv_out(win, 3, stride, 1);
            unit(p + "conv2", p + "bn2", couts[li], couts[li], 3, 1, 1, hin, win);
            if (b + 1 == nblk[li]) cin = couts[li];
        }
    // ---- workspace: image, per unit y / out / mean / invstd, pooled level, gradient scratch, latent
    size_t fl = al64((size_t)n * height * width * 4);
    size_t max_act = 0;
    for (const TrunkUnit& u : T.units) {
        const size_t act = (size_t)n * u.hout * u.wout * u.cout;
        fl += 2 * al64(act) + 2 * al64((size_t)u.cout);
        max_act = std::max(max_act, act);
    }
    max_act = std::max(max_act, (size_t)n * T.h[0] * T.w[0] * 64);
    fl += al64((size_t)n * T.h[1] * T.w[1] * 64);                   // pooled level 0
    fl += TrunkTrain::NSCR * al64(max_act);                           // gradient scratch pool
    const int ch[4] = {64, 64, 128, 256};
    for (int lv = 0; lv < 4; ++lv) fl += al64((size_t)n * T.h[lv] * T.w[lv] * ch[lv]);
    fl += al64((size_t)n * T.h[0] * T.w[0] * 512);
    int rc;
    if ((rc = T.work.reserve(fl * sizeof(float)))) return rc;
    size_t off = 0;
    auto carve = [&](size_t cnt) {
        float* p = T.work.f() + off;
        off += al64(cnt);
        return p;
    };
    T.img4 = carve((size_t)n * height * width * 4);
    for (TrunkUnit& u : T.units) {
        const size_t act = (size_t)n * u.hout * u.wout * u.cout;
        u.y = carve(act);
        u.out = carve(act);
        u.mean = carve(u.cout);
        u.invstd = carve(u.cout);
    }
    T.pooled = carve((size_t)n * T.h[1] * T.w[1] * 64);
    for (int i = 0; i < TrunkTrain::NSCR; ++i) T.scr[i] = carve(max_act);
    for (int lv = 0; lv < 4; ++lv) T.d_level[lv] = carve((size_t)n * T.h[lv] * T.w[lv] * ch[lv]);
    T.lat_nhwc = carve((size_t)n * T.h[0] * T.w[0] * 512);
    // ---- packed operands
    size_t pk = 0;
    for (TrunkUnit& u : T.units) pk += al64((size_t)(u.cout / 32) * u.J * 256) + (u.cin >= 32 ? al64((size_t)(u.cin / 32) * u.Jt * 256) : 0);
    if ((rc = T.packs.reserve(pk * sizeof(float)))) return rc;
    size_t po = 0;
    for (TrunkUnit& u : T.units) {
        u.w_fwd = T.packs.f() + po;
        po += al64((size_t)(u.cout / 32) * u.J * 256);
        u.w_t = nullptr;
        if (u.cin >= 32) {   // conv1's input gradient (the images) is never needed
            u.w_t = T.packs.f() + po;
            po += al64((size_t)(u.cin / 32) * u.Jt * 256);
        }
    }
    // ones / zeros for the raw convolution (scale 1, shift 0)
    if ((rc = T.ones.reserve(512 * sizeof(float)))) return rc;
    {
        std::vector<float> v(512, 0.f);
        for (int i = 0; i < 256; ++i) v[i] = 1.0f;
        PNY_HIP(hipMemcpy(T.ones.p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if ((rc = T.part.reserve((size_t)BN_MAXB * 2 * 256 * sizeof(float)))) return rc;
    {   // the largest weight-gradient partial buffer of any unit (never reallocated while a side-stream launch may read it)
        size_t mx = 0;
        for (const TrunkUnit& u : T.units) {
            int sp;
            long long ch_;
            dw_split(u, n, &sp, &ch_);
            mx = std::max(mx, (size_t)sp * u.cout * u.k * u.k * u.cin_p);
        }
        if ((rc = T.dwpart.reserve(mx * sizeof(float)))) return rc;
    }
    T.have_forward = false;
    T.n_jobs = 0;   // the job table depends on the parameter pointers: rebuilt at the next forward
    return 0;
}

static int trunk_upload_jobs(pny_model* m, TrunkTrain& T) {
    std::vector<TrunkPackJob> jobs;
    int rc;
    for (TrunkUnit& u : T.units) {
        const float* w = nullptr;
        if ((rc = find_param(m, u.conv + ".weight", &w))) return rc;
        TrunkPackJob j;
        j.src = w;
        j.dst = u.w_fwd;
        j.kind = 0;
        j.cout = u.cout;
        j.cin = u.cin;
        j.cin_p = u.cin_p;
        j.k = u.k;
        j.J = u.J;
        j.count = (u.cout / 32) * u.J * 64;
        jobs.push_back(j);
        if (u.w_t) {
            j.dst = u.w_t;
            j.kind = 1;
            j.J = u.Jt;
            j.count = (u.cin / 32) * u.Jt * 64;
            jobs.push_back(j);
        }
    }
    if ((rc = T.jobs.reserve(jobs.size() * sizeof(TrunkPackJob)))) return rc;
    PNY_HIP(hipMemcpy(T.jobs.p, jobs.data(), jobs.size() * sizeof(TrunkPackJob), hipMemcpyHostToDevice));
    T.n_jobs = (int)jobs.size();
    return 0;
}

static void bn_grid(long long P, int* B, long long* chunk) {
    long long b = (P + 255) / 256;
    if (b > BN_MAXB) b = BN_MAXB;
    if (b < 1) b = 1;
    *chunk = (P + b - 1) / b;
    *B = (int)((P + *chunk - 1) / *chunk);
}

static int unit_forward(pny_model* m, TrunkTrain& T, TrunkUnit& u, float momentum, hipStream_t st) {
    ConvLayer L;
    L.w = u.w_fwd;
    L.scale = T.ones.f();
    L.shift = T.ones.f() + 256;
    L.cin = u.cin;
    L.cin_p = u.cin_p;
    L.cout = u.cout;
    L.k = u.k;
    L.stride = u.stride;
    L.pad = u.pad;
    L.J = u.J;
    if (!run_conv_ex(L, u.x, T.n, u.hin, u.win, u.hout, u.wout, 0, nullptr, 0, u.y, st)) return fail(PNY_ERR_HIP, "trunk training: convolution launch failed");
    const long long P = (long long)T.n * u.hout * u.wout;
    int B;
    long long chunk;
    bn_grid(P, &B, &chunk);
    const float *gamma, *beta, *rm, *rv;
    int rc;
    if ((rc = find_param(m, u.bn + ".weight", &gamma)) || (rc = find_param(m, u.bn + ".bias", &beta)) ||
        (rc = find_param(m, u.bn + ".running_mean", &rm)) || (rc = find_param(m, u.bn + ".running_var", &rv)))
        return rc;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(B), dim3(BN_THREADS), 0, st, u.y, P, u.cout, chunk, T.part.f());
    const long long total = P * (u.cout / 4);
    long long grid = (total + BN_THREADS - 1) / BN_THREADS;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)grid), dim3(BN_THREADS), 0, st, u.y, T.part.f(), B, P, u.cout, gamma, beta, u.resid,
                       u.relu ? 1 : 0, u.out, u.mean, u.invstd, const_cast<float*>(rm), const_cast<float*>(rv), momentum, 1e-5f,
                       T.bn_eval ? 1 : 0);
    PNY_HIP(hipGetLastError());
    return 0;
}

// forward over n images (n, 3, H, W) NCHW -> latent (n, 512, hl, wl) NCHW; keeps the activations for trunk_train_backward
int trunk_train_forward(pny_model* m, const float* images, int n, int height, int width, float momentum, int bn_eval,
                        float* latent_nchw, hipStream_t st) {
    if (!m->trunk) m->trunk = new TrunkTrain();
    TrunkTrain& T = *m->trunk;
    T.bn_eval = bn_eval != 0;
    int rc;
    if ((rc = trunk_plan(m, T, n, height, width))) return rc;
    if (T.n_jobs == 0 && (rc = trunk_upload_jobs(m, T))) return rc;
    T.have_forward = false;
    // weights of this step: one launch rebuilds every packed operand from the live parameters
    {
        int mx = 0;
        for (const TrunkUnit& u : T.units) mx = std::max(mx, std::max((u.cout / 32) * u.J * 64, u.w_t ? (u.cin / 32) * u.Jt * 64 : 0));
        int bx = (mx + 255) / 256;
        if (bx > 128) bx = 128;
        hipLaunchKernelGGL(trunk_pack_kernel, dim3(bx, T.n_jobs), dim3(256), 0, st, reinterpret_cast<const TrunkPackJob*>(T.jobs.p));
    }
    const long long npx = (long long)n * height * width;
    hipLaunchKernelGGL(trunk_image_to_nhwc4_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st, images, T.img4, n, height * width);
    size_t ui = 0;
    TrunkUnit& u0 = T.units[ui++];
    u0.x = T.img4;
    u0.resid = nullptr;
    u0.relu = true;
    if ((rc = unit_forward(m, T, u0, momentum, st))) return rc;
    T.level_out[0] = u0.out;
    const float* x = u0.out;
    if (T.pool) {
        const long long np = (long long)n * T.h[1] * T.w[1] * 16;
        hipLaunchKernelGGL(trunk_maxpool_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, u0.out, T.pooled, n, T.h[0], T.w[0], 64,
                           T.h[1], T.w[1]);
        x = T.pooled;
    }
    const int nblk[3] = {3, 4, 6};
    for (int li = 0; li < 3; ++li) {
        for (int b = 0; b < nblk[li]; ++b) {
            const float* idt = x;
            if (T.units[ui].k == 1) {   // downsample branch: bn(conv1x1(x)), no relu
                TrunkUnit& ud = T.units[ui++];
                ud.x = x;
                ud.resid = nullptr;
                ud.relu = false;
                if ((rc = unit_forward(m, T, ud, momentum, st))) return rc;
                idt = ud.out;
            }
            TrunkUnit& u1 = T.units[ui++];
            u1.x = x;
            u1.resid = nullptr;
            u1.relu = true;
            if ((rc = unit_forward(m, T, u1, momentum, st))) return rc;
            TrunkUnit& u2 = T.units[ui++];
            u2.x = u1.out;
            u2.resid = idt;
            u2.relu = true;
            if ((rc = unit_forward(m, T, u2, momentum, st))) return rc;
            x = u2.out;
        }
        T.level_out[li + 1] = x;
    }
    const int ch[4] = {64, 64, 128, 256}, coff[4] = {0, 64, 128, 256};
    for (int lv = 0; lv < 4; ++lv) {
        const long long np = (long long)n * T.h[0] * T.w[0] * (ch[lv] / 4);
        hipLaunchKernelGGL(trunk_upsample_concat_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, T.level_out[lv], T.lat_nhwc, n,
                           T.h[lv], T.w[lv], ch[lv], T.h[0], T.w[0], 512, coff[lv]);
    }
    launch_nhwc_to_nchw(T.lat_nhwc, latent_nchw, n, 512, T.h[0] * T.w[0], st);
    PNY_HIP(hipGetLastError());
    T.have_forward = true;
    return 0;
}

// pixel slices of a unit's weight-gradient GEMM: ~4 waves per CU in all, at least 128 pixels per slice
static void dw_split(const TrunkUnit& u, int n, int* splits, long long* chunk) {
    const long long npix = (long long)n * u.hout * u.wout;
    const int Kp = u.k * u.k * u.cin_p;
    const long long base = (long long)(u.cout / 64) * ((Kp + 127) / 128);
    long long sp = (1024 + base - 1) / base;
    const long long max_splits = std::max(1ll, npix / 128);
    if (sp > max_splits) sp = max_splits;
    if (sp < 1) sp = 1;
    *chunk = ((npix + sp - 1) / sp + 3) / 4 * 4;   // the pixel loop walks 4 pixels per iteration
    *splits = (int)((npix + *chunk - 1) / *chunk);
}

static int unit_weight_grad(pny_model* m, TrunkTrain& T, const TrunkUnit& u, const float* dy, hipStream_t main_st) {
    float* dw = find_grad(m, u.conv + ".weight");
    if (!dw) return 0;
    // on the side stream, behind the kernel that wrote dy; the buffer dy lives in is marked busy until this is done
    hipStream_t st = main_st;
    if (T.side) {
        PNY_HIP(hipEventRecord(T.ev_ready, main_st));
        PNY_HIP(hipStreamWaitEvent(T.side, T.ev_ready, 0));
        st = T.side;
    }
    DwcArgs a;
    a.dy = dy;
    a.x = u.x;
    a.n = T.n;
    a.hin = u.hin;
    a.win = u.win;
    a.cin_p = u.cin_p;
    a.hout = u.hout;
    a.wout = u.wout;
    a.cout = u.cout;
    a.k = u.k;
    a.stride = u.stride;
    a.pad = u.pad;
    a.Kp = u.k * u.k * u.cin_p;
    a.ntiles = (a.Kp + 127) / 128;
    a.npix = (long long)T.n * u.hout * u.wout;
    const long long base = (long long)(u.cout / 64) * a.ntiles;
    dw_split(u, T.n, &a.splits, &a.chunk);
    if ((size_t)a.splits * u.cout * a.Kp * sizeof(float) > T.dwpart.bytes) return fail(PNY_ERR_STATE, "trunk backward: weight-gradient partials larger than planned");
    a.partial = T.dwpart.f();
    const long long items = base * a.splits;
    hipLaunchKernelGGL(conv_dw_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, a);
    const int tot = u.cout * u.cin * u.k * u.k;
    hipLaunchKernelGGL(conv_dw_reduce_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, T.dwpart.f(), a.splits, u.cout, u.cin, u.cin_p, u.k,
                       a.Kp, dw);
    PNY_HIP(hipGetLastError());
    if (T.side)
        for (int i = 0; i < TrunkTrain::NSCR; ++i)
            if (T.scr[i] == dy) {
                PNY_HIP(hipEventRecord(T.ev_buf[i], T.side));
                T.pend[i] = true;
            }
    return 0;
}

// d_out -> [dy of the unit's convolution output in dy_buf] (+ masked gradient in g_buf), d gamma / d beta, weight gradient;
// then, if dx_buf, the input gradient dx = conv_T(dy) + dx_add
static int unit_backward(pny_model* m, TrunkTrain& T, TrunkUnit& u, const float* d_out, float* dy_buf, float* g_buf, float* dx_buf,
                         const float* dx_add, hipStream_t st) {
    const long long P = (long long)T.n * u.hout * u.wout;
    int B;
    long long chunk;
    bn_grid(P, &B, &chunk);
    const float* gamma;
    int rc;
    if ((rc = find_param(m, u.bn + ".weight", &gamma))) return rc;
    const float* mask = u.relu ? u.out : nullptr;
    hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(B), dim3(BN_THREADS), 0, st, d_out, mask, u.y, u.mean, u.invstd, P, u.cout, chunk, T.part.f());
    const long long total = P * (u.cout / 4);
    long long grid = (total + BN_THREADS - 1) / BN_THREADS;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)grid), dim3(BN_THREADS), 0, st, d_out, mask, u.y, u.mean, u.invstd, gamma, T.part.f(), B,
                       P, u.cout, dy_buf, g_buf, find_grad(m, u.bn + ".weight"), find_grad(m, u.bn + ".bias"), T.bn_eval ? 1 : 0);
    PNY_HIP(hipGetLastError());
    if ((rc = unit_weight_grad(m, T, u, dy_buf, st))) return rc;
    if (dx_buf) {
        ConvLayer L;
        L.w = u.w_t;
        L.scale = T.ones.f();
        L.shift = T.ones.f() + 256;
        L.cin = L.cin_p = u.cout;
        L.cout = u.cin;
        L.k = u.k;
        L.stride = 1;
        L.pad = u.k - 1 - u.pad;
        L.J = u.Jt;
        const int sh = u.stride == 2 ? 1 : 0;
        if (!run_conv_ex(L, dy_buf, T.n, u.hout, u.wout, u.hin, u.win, sh, dx_add, 0, dx_buf, st))
            return fail(PNY_ERR_HIP, "trunk training: transposed convolution launch failed");
    }
    return 0;
}

// d loss / d latent (n, 512, hl, wl) NCHW -> gradients of every bound encoder parameter
int trunk_train_backward(pny_model* m, const float* d_latent_nchw, hipStream_t st) {
    if (!m->trunk || !m->trunk->have_forward) return fail(PNY_ERR_STATE, "pny_trunk_train_backward: no training forward to differentiate");
    TrunkTrain& T = *m->trunk;
    const int n = T.n;
    int rc;
    // pyramid: level gradients from the latent's channel groups
    float* d_lat = T.lat_nhwc;   // (the forward's channel-last latent is no longer needed)
    launch_nchw_to_nhwc(d_latent_nchw, d_lat, n, 512, T.h[0] * T.w[0], st);
    const int ch[4] = {64, 64, 128, 256}, coff[4] = {0, 64, 128, 256};
    for (int lv = 0; lv < 4; ++lv) {
        const long long np = (long long)n * T.h[lv] * T.w[lv] * (ch[lv] / 4);
        hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, d_lat, (const float*)nullptr, T.d_level[lv], n,
                           T.h[lv], T.w[lv], ch[lv], T.h[0], T.w[0], 512, coff[lv]);
    }
    // residual layers, last to first.  g_x = gradient w.r.t. the current block's output.  Scratch: a pool of NSCR buffers of
    // the largest activation; at most four are live at any point.
    const int nblk[3] = {3, 4, 6};
    if (!T.side && !getenv("PNYOLO_TRUNK_NO_SIDE_STREAM")) {
        PNY_HIP(hipStreamCreateWithFlags(&T.side, hipStreamNonBlocking));
        PNY_HIP(hipEventCreateWithFlags(&T.ev_ready, hipEventDisableTiming));
        PNY_HIP(hipEventCreateWithFlags(&T.ev_join, hipEventDisableTiming));
        for (auto& e : T.ev_buf) PNY_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    for (bool& b : T.pend) b = false;
    bool used[TrunkTrain::NSCR] = {};
    auto grab = [&]() -> float* {
        // prefer a buffer no weight-gradient launch on the side stream still reads
        for (int pass = 0; pass < 2; ++pass)
            for (int i = 0; i < TrunkTrain::NSCR; ++i)
                if (!used[i] && (pass == 1 || !T.pend[i])) {
                    used[i] = true;
                    if (T.pend[i]) {
                        (void)hipStreamWaitEvent(st, T.ev_buf[i], 0);
                        T.pend[i] = false;
                    }
                    return T.scr[i];
                }
        return nullptr;
    };
    auto drop = [&](const float* ptr) {
        for (int i = 0; i < TrunkTrain::NSCR; ++i)
            if (T.scr[i] == ptr) used[i] = false;
    };
    const float* g_x = T.d_level[3];
    size_t ui = T.units.size();
    for (int li = 2; li >= 0; --li) {
        for (int b = nblk[li] - 1; b >= 0; --b) {
            TrunkUnit& u2 = T.units[--ui];
            TrunkUnit& u1 = T.units[--ui];
            TrunkUnit* ud = (ui > 0 && T.units[ui - 1].k == 1) ? &T.units[--ui] : nullptr;
            // conv2 / bn2: upstream g_x through relu(out); g = the masked gradient = gradient of the identity branch
            float *dy2 = grab(), *g = grab(), *d_a1 = grab();
            if (!dy2 || !g || !d_a1) return fail(PNY_ERR_STATE, "trunk backward: scratch pool exhausted");
            if ((rc = unit_backward(m, T, u2, g_x, dy2, g, d_a1, nullptr, st))) return rc;
            drop(g_x);
            drop(dy2);
            // conv1 / bn1: upstream d_a1 through relu(a1); without a downsample branch the identity gradient g joins here
            float *dy1 = grab(), *dx = grab();
            if (!dy1 || !dx) return fail(PNY_ERR_STATE, "trunk backward: scratch pool exhausted");
            if ((rc = unit_backward(m, T, u1, d_a1, dy1, nullptr, dx, ud ? nullptr : g, st))) return rc;
            drop(d_a1);
            drop(dy1);
            if (ud) {   // downsample branch: upstream g (no relu behind its batch norm); its input gradient is added to dx
                float *dyd = grab(), *dx2 = grab();
                if (!dyd || !dx2) return fail(PNY_ERR_STATE, "trunk backward: scratch pool exhausted");
                if ((rc = unit_backward(m, T, *ud, g, dyd, nullptr, dx2, dx, st))) return rc;
                drop(dyd);
                drop(dx);
                dx = dx2;
            }
            drop(g);
            g_x = dx;
        }
        // the layer's input is the previous level's output, which also fed the pyramid (levels 2 and 1; level 0 goes through the pool)
        if (li > 0) {
            float* sum = grab();
            if (!sum) return fail(PNY_ERR_STATE, "trunk backward: scratch pool exhausted");
            const long long n4 = (long long)n * T.h[li] * T.w[li] * ch[li] / 4;
            hipLaunchKernelGGL(add2_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, g_x, (const float*)T.d_level[li], sum, n4);
            drop(g_x);
            g_x = sum;
        }
    }
    // level 0: through the max-pool (or directly), plus its pyramid share
    float* d_l0 = grab();
    TrunkUnit& u0 = T.units[0];
    if (T.pool) {
        const long long np = (long long)n * T.h[0] * T.w[0] * 16;
        hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, u0.out, g_x, (const float*)T.d_level[0], d_l0, n,
                           T.h[0], T.w[0], 64, T.h[1], T.w[1]);
    } else {
        const long long n4 = (long long)n * T.h[0] * T.w[0] * 16;
        hipLaunchKernelGGL(add2_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, g_x, (const float*)T.d_level[0], d_l0, n4);
    }
    drop(g_x);
    float* dy0 = grab();
    if (!d_l0 || !dy0) return fail(PNY_ERR_STATE, "trunk backward: scratch pool exhausted");
    if ((rc = unit_backward(m, T, u0, d_l0, dy0, nullptr, nullptr, nullptr, st))) return rc;
    if (T.side) {   // the caller's stream continues only behind the last weight gradient
        PNY_HIP(hipEventRecord(T.ev_join, T.side));
        PNY_HIP(hipStreamWaitEvent(st, T.ev_join, 0));
    }
    PNY_HIP(hipGetLastError());
    T.have_forward = false;
    return 0;
}

}  // namespace pny

extern "C" {

int pny_trunk_train_forward(pny_model* m, const float* images_dev, int n_images, int height, int width, float momentum, int bn_eval,
                            float* latent_nchw_dev, pny_stream stream) {
    if (!m || !images_dev || !latent_nchw_dev) return fail(PNY_ERR_ARG, "pny_trunk_train_forward: null argument");
    if (m->desc.d_latent != 512) return fail(PNY_ERR_ARG, "pny_trunk_train_forward: the ResNet-34 trunk yields 512 channels; model d_latent differs");
    if (n_images < 1 || height < 32 || width < 32) return fail(PNY_ERR_ARG, "pny_trunk_train_forward: bad shape");
    PNY_HIP(hipSetDevice(m->desc.device));
    return trunk_train_forward(m, images_dev, n_images, height, width, momentum, bn_eval, latent_nchw_dev, (hipStream_t)stream);
}

int pny_trunk_train_backward(pny_model* m, const float* d_latent_nchw_dev, pny_stream stream) {
    if (!m || !d_latent_nchw_dev) return fail(PNY_ERR_ARG, "pny_trunk_train_backward: null argument");
    PNY_HIP(hipSetDevice(m->desc.device));
    return trunk_train_backward(m, d_latent_nchw_dev, (hipStream_t)stream);
}

}  // extern "C"
