// ResNet-34 trunk of SpatialEncoder for gfx950 (reference src/model/encoder.py:139-173,
// num_layers = 4, use_first_pool, eval-mode batch norm; block layout = the public ResNet-34
// BasicBlock [3,4,6] architecture -- torchvision itself is not in the tree).
//
// Every convolution is an implicit GEMM on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32),
// out^T[cout][pixel] = W[cout][(ky,kx,ci)] * patch^T, with channel-last activations so that a
// B-operand fragment (4 consecutive input channels of one tap) is one 16-byte load and an
// accumulator quad (4 consecutive output channels of one pixel) is one 16-byte store; weights
// are pre-packed in lane order like the MLP's (api.hip pack_layer).  Batch norm is applied in
// the epilogue as x*scale + shift, then the residual add and relu -- one kernel per conv.
// The pyramid is upsampled (bilinear, align_corners=True) and concatenated directly into the
// channel-last latent the MLP kernel gathers from.
#include <cmath>
#include <cstring>

#include "encoder.h"

namespace pny {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float* in;
    const float* w;
    const float* scale;
    const float* shift;
    const float* resid;
    float* out;
    int n, hin, win, cin_p, hout, wout, cout, k, stride, pad, J, relu;
    int dil_shift;   // input dilation 1 << dil_shift: the transposed convolution of the backward pass reads the gradient of a
                     // stride-2 convolution at every second position of the output grid (encoder_train.hip); 0 otherwise
    long long npix;
};

// Output tile of a wave: (32*NT output channels) x (32*MT pixels), NT, MT in {1, 2}.
// SPLIT = 1: 4 waves per workgroup, one output tile each.
// SPLIT = 8: 8 waves per workgroup share ONE output tile, wave w takes k-iterations j = w (mod 8); the
//            partial accumulators are reduced through LDS and wave w finishes 1/8 of the (position, register)
//            pairs.  Used for the deep, spatially small layers, where a tile per wave leaves a handful of
//            workgroups on the chip running up to 288 dependent iterations.
// The host picks 32x32 tiles (NT = MT = 1) whenever 64x64 tiles would not give every CU a workgroup: the
// ResNet-34 trunk at 128x128 input has 192 .. 3072 output pixels per layer, i.e. 12 .. 48 tiles of 64x64.
template <int SPLIT, int NT, int MT>
__global__ __launch_bounds__(SPLIT == 1 ? 256 : 512) void conv_mfma_kernel(const ConvArgs a) {
    constexpr int TN = 32 * NT, TMp = 32 * MT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = lane & 31, hh = lane >> 5;
    const int tiles_n = a.cout / TN;
    const long long tile = SPLIT == 1 ? (long long)blockIdx.x * 4 + wave : (long long)blockIdx.x;
    const long long tile_m = tile / tiles_n;
    const int tile_n = (int)(tile - tile_m * tiles_n);
    if (tile_m * TMp >= a.npix) return;

    long long pix[MT];
    int iy0[MT], ix0[MT], img[MT];
    bool valid[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        long long p = tile_m * TMp + 32 * mt + m0;
        valid[mt] = p < a.npix;
        if (!valid[mt]) p = a.npix - 1;
        pix[mt] = p;
        const int per = a.hout * a.wout;
        img[mt] = (int)(p / per);
        const int r = (int)(p - (long long)img[mt] * per);
        const int oy = r / a.wout, ox = r - oy * a.wout;
        iy0[mt] = oy * a.stride - a.pad;
        ix0[mt] = ox * a.stride - a.pad;
    }
    f32x16 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.f;

    const float4* w0 = reinterpret_cast<const float4*>(a.w) + (size_t)(NT * tile_n) * a.J * 64 + lane;
    const int ntap = a.k * a.k;
    // One k-iteration's operands: NT weight fragments (packed, coalesced) and MT patch fragments
    // (4 consecutive input channels of one tap per lane; zero outside the image / beyond K).
    struct Frag {
        float4 a[NT], b[MT];
    };
    auto fetch = [&](int j, Frag& f) {
        const int jj = j < a.J ? j : a.J - 1;  // clamped prefetch past the end (discarded)
        const int k0 = 8 * jj + 4 * hh;
        const int tap = k0 / a.cin_p, ci = k0 - tap * a.cin_p;
        const int ky = tap / a.k, kx = tap - ky * a.k;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) f.a[nt] = w0[((size_t)nt * a.J + jj) * 64];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int iy = iy0[mt] + ky, ix = ix0[mt] + kx;
            bool ok = valid[mt] && tap < ntap && iy >= 0 && ix >= 0;
            if (a.dil_shift) {   // positions between the samples of a dilated input are zeros
                const int msk = (1 << a.dil_shift) - 1;
                ok = ok && !(iy & msk) && !(ix & msk);
                iy >>= a.dil_shift;
                ix >>= a.dil_shift;
            }
            ok = ok && iy < a.hin && ix < a.win;
            f.b[mt] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) f.b[mt] = *reinterpret_cast<const float4*>(a.in + (((size_t)img[mt] * a.hin + iy) * a.win + ix) * a.cin_p + ci);
        }
    };
    auto mac = [&](const Frag& f) {
#define PNY_STEP(c)                                                                         \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                       \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                   \
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[nt].c, f.b[mt].c, acc[nt][mt], 0, 0, 0);
        PNY_STEP(x)
        PNY_STEP(y)
        PNY_STEP(z)
        PNY_STEP(w)
#undef PNY_STEP
    };
    // static 3-slot software pipeline (slots never copied: see mlp.hip gemm_run): operands run two
    // k-iterations ahead of the MFMAs; these small convolutions are otherwise one L2 round trip per
    // iteration (up to 288 dependent iterations).
    Frag f0, f1, f2;
    const int jb = SPLIT == 1 ? 0 : wave;  // first k-iteration of this wave, stride SPLIT
    fetch(jb, f0);
    fetch(jb + SPLIT, f1);
    for (int j = jb; j < a.J; j += 3 * SPLIT) {
        fetch(j + 2 * SPLIT, f2);
        __builtin_amdgcn_sched_barrier(0);
        mac(f0);
        __builtin_amdgcn_sched_barrier(0);
        if (j + SPLIT < a.J) {
            fetch(j + 3 * SPLIT, f0);
            __builtin_amdgcn_sched_barrier(0);
            mac(f1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (j + 2 * SPLIT < a.J) {
            fetch(j + 4 * SPLIT, f1);
            __builtin_amdgcn_sched_barrier(0);
            mac(f2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (SPLIT > 1) {
        // reduce the SPLIT partial tiles: part[wave][tile position (nt,mt)][reg][lane]
        constexpr int NPOS = NT * MT, PAIRS = NPOS * 16, PER = PAIRS / SPLIT;
        static_assert(PAIRS % SPLIT == 0, "pairs per wave");
        extern __shared__ float part[];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) part[((wave * NPOS + nt * MT + mt) * 16 + r) * 64 + lane] = acc[nt][mt][r];
        __syncthreads();
        // each wave finishes PER of the PAIRS (position, reg) pairs: pairs p = PER*wave .. PER*wave + PER - 1
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int pr = PER * wave + i;  // = (nt*MT+mt)*16 + r
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < SPLIT; ++w) sum += part[((w * NPOS) * 16 + pr) * 64 + lane];
            const int pos = pr >> 4, r = pr & 15, nt = pos / MT, mt = pos - nt * MT;
            if (!valid[mt]) continue;
            const int c = TN * tile_n + 32 * nt + 8 * (r >> 2) + 4 * hh + (r & 3);
            float v = sum * a.scale[c] + a.shift[c];
            const size_t o = (size_t)pix[mt] * a.cout + c;
            if (a.resid) v += a.resid[o];
            if (a.relu) v = fmaxf(v, 0.f);
            a.out[o] = v;
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (!valid[mt]) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = TN * tile_n + 32 * nt + 8 * q + 4 * hh;
                const float4 sc = *reinterpret_cast<const float4*>(a.scale + c);
                const float4 sh = *reinterpret_cast<const float4*>(a.shift + c);
                float4 v;
                v.x = acc[nt][mt][4 * q + 0] * sc.x + sh.x;
                v.y = acc[nt][mt][4 * q + 1] * sc.y + sh.y;
                v.z = acc[nt][mt][4 * q + 2] * sc.z + sh.z;
                v.w = acc[nt][mt][4 * q + 3] * sc.w + sh.w;
                const size_t o = (size_t)pix[mt] * a.cout + c;
                if (a.resid) {
                    const float4 r = *reinterpret_cast<const float4*>(a.resid + o);
                    v.x += r.x;
                    v.y += r.y;
                    v.z += r.z;
                    v.w += r.w;
                }
                if (a.relu) {
                    v.x = fmaxf(v.x, 0.f);
                    v.y = fmaxf(v.y, 0.f);
                    v.z = fmaxf(v.z, 0.f);
                    v.w = fmaxf(v.w, 0.f);
                }
                *reinterpret_cast<float4*>(a.out + o) = v;
            }
    }
}

// Pixel-wise linear map as a tiled GEMM: out[p][n] = sum_k in[p][k] * W[n][k] for channel-last rows (the latent
// projection; K % 32 == 0, cout % 256 == 0).  Same transposed MFMA formulation as everywhere (weights = packed A
// operand read straight from L2, pixels = B operand), but the B operand is STAGED THROUGH LDS: the 1x1-convolution path
// reads it with one scattered 16-byte load per lane and k-iteration (64 cache lines per wave instruction, 16 useful
// bytes each), here a 128-pixel x 32-channel chunk is loaded as whole 128-byte lines (4 lanes per pixel row), written
// to LDS as float4 bt[k/4][pixel] and read back conflict-free as MFMA fragments; chunks are double-buffered, the
// global loads of chunk c+1 are in flight during the MFMAs of chunk c.  A workgroup (4 waves) owns 128 pixels x 256
// outputs, a wave 128 pixels x 64 outputs (2 x 4 tiles of 32x32), so a weight fragment feeds 4 MFMAs per k-step.
constexpr int PL_PIX = 128, PL_KC = 32, PL_NW = 4;

__global__ __launch_bounds__(64 * PL_NW) void pixel_linear_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                                  float* __restrict__ out, long long npix, int K, int cout) {
    __shared__ float4 bt[2][PL_KC / 4][PL_PIX + 1];  // +1: spreads the staging writes over the banks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = lane & 31, hh = lane >> 5;
    const int nblocks = cout / 256;
    const long long pb = blockIdx.x / nblocks;
    const int nb = (int)(blockIdx.x - pb * nblocks);
    const long long p0 = pb * PL_PIX;
    const int J = K / 8;
    // staging assignment: 4 lanes per pixel row, 32 bytes each (2 float4), 2 rows per thread
    const int srow = tid >> 2, sq = (tid & 3) * 2;
    auto stage_load = [&](int c, float4 (&v)[2][2]) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            long long p = p0 + srow + 64 * r;
            if (p >= npix) p = npix - 1;
            const float4* src = reinterpret_cast<const float4*>(in + (size_t)p * K + (size_t)c * PL_KC) + sq;
            v[r][0] = src[0];
            v[r][1] = src[1];
        }
    };
    auto stage_store = [&](int buf, const float4 (&v)[2][2]) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            bt[buf][sq][srow + 64 * r] = v[r][0];
            bt[buf][sq + 1][srow + 64 * r] = v[r][1];
        }
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.f;
    const int nt0 = nb * 8 + wave * 2;  // this wave's first 32-output tile
    const float4* wp = reinterpret_cast<const float4*>(w) + (size_t)nt0 * J * 64 + lane;
    const int nchunks = K / PL_KC;
    float4 sv[2][2];
    stage_load(0, sv);
    stage_store(0, sv);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) stage_load(c + 1, sv);
        float4 a[PL_KC / 8][2];
#pragma unroll
        for (int j = 0; j < PL_KC / 8; ++j)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) a[j][nt] = wp[((size_t)nt * J + (size_t)c * (PL_KC / 8) + j) * 64];
#pragma unroll
        for (int j = 0; j < PL_KC / 8; ++j) {
            float4 b[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) b[mt] = bt[buf][2 * j + hh][32 * mt + m0];
#define PNY_STEP(cc)                                                                          \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                          \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                      \
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][nt].cc, b[mt].cc, acc[nt][mt], 0, 0, 0);
            PNY_STEP(x)
            PNY_STEP(y)
            PNY_STEP(z)
            PNY_STEP(w)
#undef PNY_STEP
        }
        if (c + 1 < nchunks) {
            stage_store(buf ^ 1, sv);  // the other buffer: its readers finished before the barrier of the previous chunk
            __syncthreads();
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const long long p = p0 + 32 * mt + m0;
        if (p >= npix) continue;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = 32 * (nt0 + nt) + 8 * q + 4 * hh;
                float4 v;
                v.x = acc[nt][mt][4 * q + 0];
                v.y = acc[nt][mt][4 * q + 1];
                v.z = acc[nt][mt][4 * q + 2];
                v.w = acc[nt][mt][4 * q + 3];
                *reinterpret_cast<float4*>(out + (size_t)p * cout + n) = v;
            }
    }
}

// The same GEMM on the f16 matrix cores with split fp32 operands (the arithmetic of mlp_h2.hip: x = x1 + x2 in two f16 planes,
// x1 w1 + x2 w1 + x1 w2 on v_mfma_f32_32x32x16_f16, fp32 accumulation), for scenes whose MLP launches run the f16x2 kernels:
// in a training step the projection is redone for every scene and MLP after each optimizer step.  Same tiling and the same
// packed fp32 weights: a 16-k step takes the lane's OWN two float4 of the fp32 image (k = 16 J + 4 hh + 0..3 and
// 16 J + 8 + 4 hh + 0..3 -- which 8 k a fragment holds is free as long as both operands agree) and splits them in
// registers; the pixel operand is split once, on its way into LDS ([plane][k / 4][pixel] x 8 bytes).
typedef _Float16 plh8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void pl_split2(float a, float b, unsigned& p0, unsigned& p1) {
    float ra, rb;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p0) : "v"(a), "v"(b));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(p0), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(p0), "v"(b));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p1) : "v"(ra), "v"(rb));
}

__global__ __launch_bounds__(64 * PL_NW) void pixel_linear_h2_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                                     float* __restrict__ out, long long npix, int K, int cout) {
    __shared__ uint2 bp[2][2][PL_KC / 4][PL_PIX + 1];  // [buffer][plane][k / 4][pixel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = lane & 31, hh = lane >> 5;
    const int nblocks = cout / 256;
    const long long pb = blockIdx.x / nblocks;
    const int nb = (int)(blockIdx.x - pb * nblocks);
    const long long p0 = pb * PL_PIX;
    const int J = K / 8;
    const int srow = tid >> 2, sq = (tid & 3) * 2;
    auto stage_load = [&](int c, float4 (&v)[2][2]) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            long long p = p0 + srow + 64 * r;
            if (p >= npix) p = npix - 1;
            const float4* src = reinterpret_cast<const float4*>(in + (size_t)p * K + (size_t)c * PL_KC) + sq;
            v[r][0] = src[0];
            v[r][1] = src[1];
        }
    };
    auto stage_store = [&](int buf, const float4 (&v)[2][2]) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint2 q0, q1;
                pl_split2(v[r][h].x, v[r][h].y, q0.x, q1.x);
                pl_split2(v[r][h].z, v[r][h].w, q0.y, q1.y);
                bp[buf][0][sq + h][srow + 64 * r] = q0;
                bp[buf][1][sq + h][srow + 64 * r] = q1;
            }
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.f;
    const int nt0 = nb * 8 + wave * 2;
    const float4* wp = reinterpret_cast<const float4*>(w) + (size_t)nt0 * J * 64 + lane;
    const int nchunks = K / PL_KC;
    float4 sv[2][2];
    stage_load(0, sv);
    stage_store(0, sv);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) stage_load(c + 1, sv);
        float4 a[PL_KC / 8][2];
#pragma unroll
        for (int j = 0; j < PL_KC / 8; ++j)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) a[j][nt] = wp[((size_t)nt * J + (size_t)c * (PL_KC / 8) + j) * 64];
#pragma unroll
        for (int s = 0; s < PL_KC / 16; ++s) {
            plh8 a1[2], a2[2], b1[4], b2[4];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                uint4 u1, u2;
                pl_split2(a[2 * s][nt].x, a[2 * s][nt].y, u1.x, u2.x);
                pl_split2(a[2 * s][nt].z, a[2 * s][nt].w, u1.y, u2.y);
                pl_split2(a[2 * s + 1][nt].x, a[2 * s + 1][nt].y, u1.z, u2.z);
                pl_split2(a[2 * s + 1][nt].z, a[2 * s + 1][nt].w, u1.w, u2.w);
                a1[nt] = __builtin_bit_cast(plh8, u1);
                a2[nt] = __builtin_bit_cast(plh8, u2);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const uint2 l1 = bp[buf][0][4 * s + hh][32 * mt + m0], h1 = bp[buf][0][4 * s + 2 + hh][32 * mt + m0];
                const uint2 l2 = bp[buf][1][4 * s + hh][32 * mt + m0], h2 = bp[buf][1][4 * s + 2 + hh][32 * mt + m0];
                b1[mt] = __builtin_bit_cast(plh8, make_uint4(l1.x, l1.y, h1.x, h1.y));
                b2[mt] = __builtin_bit_cast(plh8, make_uint4(l2.x, l2.y, h2.x, h2.y));
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[nt], b1[mt], acc[nt][mt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[nt], b1[mt], acc[nt][mt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[nt], b2[mt], acc[nt][mt], 0, 0, 0);
        }
        if (c + 1 < nchunks) {
            stage_store(buf ^ 1, sv);
            __syncthreads();
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const long long p = p0 + 32 * mt + m0;
        if (p >= npix) continue;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = 32 * (nt0 + nt) + 8 * q + 4 * hh;
                float4 v;
                v.x = acc[nt][mt][4 * q + 0];
                v.y = acc[nt][mt][4 * q + 1];
                v.z = acc[nt][mt][4 * q + 2];
                v.w = acc[nt][mt][4 * q + 3];
                *reinterpret_cast<float4*>(out + (size_t)p * cout + n) = v;
            }
    }
}

// (n,3,H,W) -> (n,H,W,4) with a zero 4th channel
__global__ void image_to_nhwc4_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int hw) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * hw) return;
    const int img = (int)(i / hw), p = (int)(i - (long long)img * hw);
    const float* s = in + (size_t)img * 3 * hw + p;
    *reinterpret_cast<float4*>(out + i * 4) = make_float4(s[0], s[hw], s[2 * (size_t)hw], 0.f);
}

// max_pool2d(3, stride 2, pad 1), channel-last
__global__ void maxpool_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int hin, int win, int c,
                               int hout, int wout) {
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
            m.x = fmaxf(m.x, v.x);
            m.y = fmaxf(m.y, v.y);
            m.z = fmaxf(m.z, v.z);
            m.w = fmaxf(m.w, v.w);
        }
    *reinterpret_cast<float4*>(out + i * 4) = m;
}

// F.interpolate(bilinear, align_corners=True) of one pyramid level to (H0,W0), written at channel
// offset coff of the 512-channel latent (reference encoder.py:160-169).
__global__ void upsample_concat_kernel(const float* __restrict__ in, float* __restrict__ lat, int n, int hin, int win,
                                       int c, int h0, int w0, int ctot, int coff) {
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

// ----------------------------------------------------------------------------------- host side
static bool upload(const std::vector<float>& v, float** out, std::vector<float*>& allocs, std::string* err) {
    float* p = nullptr;
    if (hipMalloc((void**)&p, v.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        *err = "device allocation / copy of encoder weights failed";
        return false;
    }
    allocs.push_back(p);
    *out = p;
    return true;
}

static bool build_conv(const EncoderWeights::Getter& get, const std::string& conv, const std::string& bn, int cin,
                       int cout, int k, int stride, int pad, ConvLayer& L, std::vector<float*>& allocs,
                       std::string* err) {
    const float* w = nullptr;
    std::vector<int64_t> shp;
    if (!get(conv + ".weight", &w, &shp) || shp != std::vector<int64_t>{cout, cin, k, k}) {
        *err = "missing or mis-shaped '" + conv + ".weight'";
        return false;
    }
    const float *g = nullptr, *b = nullptr, *mu = nullptr, *var = nullptr;
    std::vector<int64_t> s1;
    if (!get(bn + ".weight", &g, &s1) || !get(bn + ".bias", &b, &s1) || !get(bn + ".running_mean", &mu, &s1) ||
        !get(bn + ".running_var", &var, &s1) || s1 != std::vector<int64_t>{cout}) {
        *err = "missing or mis-shaped batch-norm tensors '" + bn + ".*'";
        return false;
    }
    L.cin = cin;
    L.cin_p = (cin + 3) / 4 * 4;
    L.cout = cout;
    L.k = k;
    L.stride = stride;
    L.pad = pad;
    const int K = k * k * L.cin_p;
    L.J = (K + 7) / 8;
    // K order (ky, kx, ci); A-operand lane order as in api.hip pack_layer
    std::vector<float> pk((size_t)(cout / 32) * L.J * 64 * 4);
    size_t o = 0;
    for (int nt = 0; nt < cout / 32; ++nt)
        for (int j = 0; j < L.J; ++j)
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r) {
                    const int n = 32 * nt + (l & 31), kk = 8 * j + 4 * (l >> 5) + r;
                    const int tap = kk / L.cin_p, ci = kk % L.cin_p;
                    float val = 0.f;
                    if (tap < k * k && ci < cin) val = w[(((size_t)n * cin + ci) * k + tap / k) * k + tap % k];
                    pk[o++] = val;
                }
    std::vector<float> sc(cout), sh(cout);
    for (int c = 0; c < cout; ++c) {
        // ATen's inference batch norm: alpha = weight / sqrt(var + eps); beta = bias - mean * alpha
        const float invstd = 1.0f / std::sqrt(var[c] + 1e-5f);
        sc[c] = g[c] * invstd;
        sh[c] = b[c] - mu[c] * sc[c];
    }
    return upload(pk, &L.w, allocs, err) && upload(sc, &L.scale, allocs, err) && upload(sh, &L.shift, allocs, err);
}

bool build_pixel_linear(const float* const* mats, int nmat, int rows, int k, ConvLayer* out, std::vector<float*>* allocs,
                        std::string* err) {
    const int cout = nmat * rows;
    if (cout % 64 || k % 8) {
        *err = "pixel-linear map needs rows*nmat % 64 == 0 and k % 8 == 0";
        return false;
    }
    ConvLayer& L = *out;
    L.cin = L.cin_p = k;
    L.cout = cout;
    L.k = 1;
    L.stride = 1;
    L.pad = 0;
    L.J = k / 8;
    std::vector<float> pk((size_t)(cout / 32) * L.J * 64 * 4);
    size_t o = 0;
    for (int nt = 0; nt < cout / 32; ++nt)
        for (int j = 0; j < L.J; ++j)
            for (int l = 0; l < 64; ++l) {
                const int n = 32 * nt + (l & 31), kk = 8 * j + 4 * (l >> 5);
                const float* row = mats[n / rows] + (size_t)(n % rows) * k + kk;
                for (int r = 0; r < 4; ++r) pk[o++] = row[r];
            }
    const std::vector<float> one((size_t)cout, 1.0f), zero((size_t)cout, 0.0f);
    return upload(pk, &L.w, *allocs, err) && upload(one, &L.scale, *allocs, err) && upload(zero, &L.shift, *allocs, err);
}

bool EncoderWeights::build(const Getter& get, const std::string& pre, std::string* err) {
    release();
    if (!build_conv(get, pre + "conv1", pre + "bn1", 3, 64, 7, 2, 3, conv1, allocs, err)) return false;
    const int couts[3] = {64, 128, 256}, nblk[3] = {3, 4, 6};
    int cin = 64;
    for (int li = 0; li < 3; ++li) {
        layers[li].clear();
        for (int b = 0; b < nblk[li]; ++b) {
            Block blk;
            const std::string p = pre + "layer" + std::to_string(li + 1) + "." + std::to_string(b) + ".";
            const int stride = (b == 0 && li > 0) ? 2 : 1;
            const int bc = (b == 0) ? cin : couts[li];
            if (!build_conv(get, p + "conv1", p + "bn1", bc, couts[li], 3, stride, 1, blk.c1, allocs, err)) return false;
            if (!build_conv(get, p + "conv2", p + "bn2", couts[li], couts[li], 3, 1, 1, blk.c2, allocs, err)) return false;
            blk.has_ds = (b == 0 && (stride != 1 || bc != couts[li]));
            if (blk.has_ds &&
                !build_conv(get, p + "downsample.0", p + "downsample.1", bc, couts[li], 1, stride, 0, blk.ds, allocs, err))
                return false;
            layers[li].push_back(blk);
        }
        cin = couts[li];
    }
    return true;
}

void EncoderWeights::release() {
    for (float* p : allocs) (void)hipFree(p);
    allocs.clear();
    for (auto& l : layers) l.clear();
}

int conv_out(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }

void encoder_latent_size(int height, int width, int* hl, int* wl) {
    *hl = conv_out(height, 7, 2, 3);
    *wl = conv_out(width, 7, 2, 3);
}

struct Dims {
    int h[4], w[4];
};
static Dims pyramid(int height, int width, bool use_first_pool) {
    Dims d;
    d.h[0] = conv_out(height, 7, 2, 3);
    d.w[0] = conv_out(width, 7, 2, 3);
    // reference encoder.py:145-146: the max-pool in front of layer1 is optional (sn64.conf skips it)
    d.h[1] = use_first_pool ? conv_out(d.h[0], 3, 2, 1) : d.h[0];
    d.w[1] = use_first_pool ? conv_out(d.w[0], 3, 2, 1) : d.w[0];
    for (int i = 2; i < 4; ++i) {
        d.h[i] = conv_out(d.h[i - 1], 3, 2, 1);
        d.w[i] = conv_out(d.w[i - 1], 3, 2, 1);
    }
    return d;
}

static size_t align64(size_t x) { return (x + 63) & ~(size_t)63; }

size_t encoder_workspace_bytes(int ns, int height, int width, bool use_first_pool) {
    const Dims d = pyramid(height, width, use_first_pool);
    size_t fl = align64((size_t)ns * height * width * 4);          // nhwc4 image
    fl += align64((size_t)ns * d.h[0] * d.w[0] * 64);              // level 0
    const int ch[4] = {64, 64, 128, 256};
    for (int i = 1; i < 4; ++i) fl += 4 * align64((size_t)ns * d.h[i] * d.w[i] * ch[i]);  // x, tmp, out, ds
    return fl * sizeof(float);
}

static bool run_conv(const ConvLayer& L, const float* in, int n, int hin, int win, const float* resid, int relu,
                     float* out, hipStream_t st) {
    return run_conv_ex(L, in, n, hin, win, conv_out(hin, L.k, L.stride, L.pad), conv_out(win, L.k, L.stride, L.pad), 0, resid, relu,
                       out, st);
}

// The general form: explicit output size and an input dilation (transposed convolutions of the trunk's backward).
bool run_conv_ex(const ConvLayer& L, const float* in, int n, int hin, int win, int hout, int wout, int dil_shift, const float* resid,
                 int relu, float* out, hipStream_t st) {
    ConvArgs a;
    a.in = in;
    a.w = L.w;
    a.scale = L.scale;
    a.shift = L.shift;
    a.resid = resid;
    a.out = out;
    a.n = n;
    a.hin = hin;
    a.win = win;
    a.cin_p = L.cin_p;
    a.hout = hout;
    a.wout = wout;
    a.dil_shift = dil_shift;
    a.cout = L.cout;
    a.k = L.k;
    a.stride = L.stride;
    a.pad = L.pad;
    a.J = L.J;
    a.relu = relu;
    a.npix = (long long)n * a.hout * a.wout;
    // tile shape: 64x64 per wave where that still gives every CU work, else 32x32 (4x the tiles, 1/4 of the
    // dependent MFMA chain per wave); the K split inside the workgroup for long reductions over few tiles
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    const long long tiles64 = ((a.npix + 63) / 64) * (L.cout / 64);
    const long long tiles32 = ((a.npix + 31) / 32) * (L.cout / 32);
    const bool small = tiles64 < 2 * (long long)cus;
    const long long tiles = small ? tiles32 : tiles64;
    const bool split = L.J >= 32 && tiles <= 4 * (long long)cus;
    static bool attr_set[64] = {};  // per device: function attributes are per device
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    dev_ &= 63;
    if (!attr_set[dev_]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<8, 2, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 4 * 16 * 64 * (int)sizeof(float));
        attr_set[dev_] = true;
    }
    if (split && small) {
        hipLaunchKernelGGL((conv_mfma_kernel<8, 1, 1>), dim3((unsigned)tiles), dim3(512), 8 * 1 * 16 * 64 * sizeof(float), st, a);
    } else if (split) {
        hipLaunchKernelGGL((conv_mfma_kernel<8, 2, 2>), dim3((unsigned)tiles), dim3(512), 8 * 4 * 16 * 64 * sizeof(float), st, a);
    } else if (small) {
        hipLaunchKernelGGL((conv_mfma_kernel<1, 1, 1>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((conv_mfma_kernel<1, 2, 2>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, a);
    }
    return hipGetLastError() == hipSuccess;
}

bool run_pixel_linear(const ConvLayer& L, const float* in, long long npix, float* out, hipStream_t st, bool f16x2) {
    if (npix <= 0 || npix > 0x7fffffffll) return false;
    // the tiled GEMM wants K in 32-channel chunks and 256-output blocks (the latent projection: K = 512 | 1792,
    // cout = 512 * view blocks when that is a multiple of 256) and enough pixels to fill tiles; otherwise the generic
    // 1x1-convolution path
    if (L.cin_p % PL_KC == 0 && L.cout % 256 == 0 && npix >= 2 * PL_PIX) {
        const long long blocks = ((npix + PL_PIX - 1) / PL_PIX) * (L.cout / 256);
        if (blocks <= 0x7fffffffll) {
            if (f16x2)
                hipLaunchKernelGGL(pixel_linear_h2_kernel, dim3((unsigned)blocks), dim3(64 * PL_NW), 0, st, in, L.w, out, npix,
                                   L.cin_p, L.cout);
            else
                hipLaunchKernelGGL(pixel_linear_kernel, dim3((unsigned)blocks), dim3(64 * PL_NW), 0, st, in, L.w, out, npix,
                                   L.cin_p, L.cout);
            return hipGetLastError() == hipSuccess;
        }
    }
    return run_conv(L, in, 1, 1, (int)npix, nullptr, 0, out, st);
}

bool encoder_forward(const EncoderWeights& W, const float* images, int ns, int height, int width, bool use_first_pool,
                     float* work, float* lat, hipStream_t st, std::string* err) {
    const Dims d = pyramid(height, width, use_first_pool);
    size_t off = 0;
    auto carve = [&](size_t n) {
        float* p = work + off;
        off += align64(n);
        return p;
    };
    float* img4 = carve((size_t)ns * height * width * 4);
    float* l0 = carve((size_t)ns * d.h[0] * d.w[0] * 64);
    const int ch[4] = {64, 64, 128, 256};
    float* buf[4][4];
    for (int i = 1; i < 4; ++i)
        for (int b = 0; b < 4; ++b) buf[i][b] = carve((size_t)ns * d.h[i] * d.w[i] * ch[i]);

    auto bad = [&]() {
        *err = std::string("encoder kernel launch failed: ") + hipGetErrorString(hipGetLastError());
        return false;
    };
    const long long npx = (long long)ns * height * width;
    hipLaunchKernelGGL(image_to_nhwc4_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st, images, img4, ns,
                       height * width);
    if (!run_conv(W.conv1, img4, ns, height, width, nullptr, 1, l0, st)) return bad();
    if (use_first_pool) {
        const long long npool = (long long)ns * d.h[1] * d.w[1] * 16;
        hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)((npool + 255) / 256)), dim3(256), 0, st, l0, buf[1][0], ns,
                           d.h[0], d.w[0], 64, d.h[1], d.w[1]);
    }

    const float* level_out[4] = {l0, nullptr, nullptr, nullptr};
    const float* x = use_first_pool ? buf[1][0] : l0;
    int hin = d.h[1], win = d.w[1];
    for (int li = 0; li < 3; ++li) {
        const int lv = li + 1;
        // buffers of a level: [0],[2] alternate as block outputs ([1][0] first holds the pooled
        // input), [1] = conv1 output, [3] = downsampled identity
        for (size_t b = 0; b < W.layers[li].size(); ++b) {
            const EncoderWeights::Block& B = W.layers[li][b];
            const float* idt = x;
            if (B.has_ds) {
                if (!run_conv(B.ds, x, ns, hin, win, nullptr, 0, buf[lv][3], st)) return bad();
                idt = buf[lv][3];
            }
            if (!run_conv(B.c1, x, ns, hin, win, nullptr, 1, buf[lv][1], st)) return bad();
            const int ho = conv_out(hin, 3, B.c1.stride, 1), wo = conv_out(win, 3, B.c1.stride, 1);
            float* dst = (x == buf[lv][0]) ? buf[lv][2] : buf[lv][0];
            if (!run_conv(B.c2, buf[lv][1], ns, ho, wo, idt, 1, dst, st)) return bad();
            x = dst;
            hin = ho;
            win = wo;
        }
        level_out[lv] = x;
    }
    // pyramid -> latent: every level resampled to level 0's size (identity for level 0)
    const int coff[4] = {0, 64, 128, 256};
    for (int lv = 0; lv < 4; ++lv) {
        const long long np = (long long)ns * d.h[0] * d.w[0] * (ch[lv] / 4);
        hipLaunchKernelGGL(upsample_concat_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st,
                           level_out[lv], lat, ns, d.h[lv], d.w[lv], ch[lv], d.h[0], d.w[0], 512, coff[lv]);
    }
    if (hipGetLastError() != hipSuccess) return bad();
    return true;
}

}  // namespace pny
