// Gradient w.r.t. the latent (the backward of encoder.py:101 F.grid_sample composed with lin_z; SURVEY.md 8f rank 1,
// "(+ grid_sample)"): what autograd adds to `latent.grad` in the reference when the encoder trains.
//
//   forward:   h_in(b)[s, v] += lin_z[b]( sum_k w_k(s, v) . latent[v][pix_k(s, v)] )            (resnetfc.py:176-182)
//   backward:  dlatent[v][pix_k(s, v)][c] += w_k(s, v) . dz[s, v][c],   dz[s, v] = sum_b lin_z[b]^T . dh_in(b)[s, v]
//
// One kernel per (64-sample tile, view): dz as a tiled fp32-MFMA GEMM straight off the dY stash -- the dX chain
// (mlp_bwd.hip) has left dh_in(b) there in [feature/4][sample] float4 tiles, which IS the LDS B-operand layout of
// pixel_linear_kernel (encoder.hip), so staging is a plain copy -- with K = n_view_blocks x 512 and the stacked transposed
// weights [lin_z[0]^T | lin_z[1]^T | ...] as the packed A operand (api.hip pack_mlp, kept current by pny_model_refresh);
// the epilogue turns the accumulators through LDS and scatters 64 consecutive latent channels of one sample per atomic
// instruction into the sample's four taps (float atomics).  The sum order over samples is therefore not fixed: latent gradients are reproducible to fp32
// rounding, not bit for bit (the MLP parameter gradients stay deterministic).
// A workgroup (4 waves) owns 64 samples x 256 latent channels, a wave 64 x 64 (2 x 2 tiles of 32 x 32).
#include "mlp_core.h"

namespace pny {

constexpr int LG_KC = 32, LG_NW = 4;

__global__ __launch_bounds__(64 * LG_NW) void latent_grad_kernel(const MlpArgs a, const float* __restrict__ dy_stash, const StashLayout lay,
                                                                 const float* __restrict__ w_cat, float* __restrict__ grad, int nvb) {
    __shared__ float4 bt[2][LG_KC / 4][64 + 1];
    __shared__ __attribute__((aligned(16))) float tr[LG_NW][32][68];   // epilogue: [wave][sample of the half][channel], 16-byte aligned rows
    __shared__ int tap_off[64][4];
    __shared__ float tap_w[64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = lane & 31, hh = lane >> 5;
    const int nblocks = a.L / 256;
    const int nb = blockIdx.x % nblocks;
    const long long tv = blockIdx.x / nblocks;
    const int v = (int)(tv % a.NS);
    const long long tile = tv / a.NS;
    const int vabs = tile_view_base(a, tile * 64) + v;   // index into the scene's view list (grouped scenes)
    const int K = nvb * HID, J = K / 8;
    if (tid < 64) {
        long long s = tile * 64 + tid;
        int offs[4];
        float wgt[4];
        const bool live = s < a.n_points;
        if (!live) s = a.n_points - 1;
        sample_taps(a, vabs, s, offs, wgt);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tap_off[tid][k] = offs[k];
            tap_w[tid][k] = live ? wgt[k] : 0.0f;
        }
    }
    // B operand: chunk c = 32 features of block c / 16; the stash slot of dh_in(b) is (2 b + 1) slots into the view's part
    const float* dyv = dy_stash + tile * lay.dy_tile + (size_t)v * lay.dy_view;
    auto stage_load = [&](int c, float4 (&sv)[2]) {
        const int b = c / (HID / LG_KC), kg0 = (c % (HID / LG_KC)) * (LG_KC / 4);
        const float4* src = reinterpret_cast<const float4*>(dyv + (size_t)(2 * b + 1) * STASH_SLOT) + (size_t)kg0 * 64;
        sv[0] = src[tid];
        sv[1] = src[tid + 256];
    };
    auto stage_store = [&](int buf, const float4 (&sv)[2]) {
        bt[buf][tid >> 6][tid & 63] = sv[0];
        bt[buf][4 + (tid >> 6)][tid & 63] = sv[1];
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.f;
    const int nt0 = nb * 8 + wave * 2;
    const float4* wp = reinterpret_cast<const float4*>(w_cat) + (size_t)nt0 * J * 64 + lane;
    const int nchunks = K / LG_KC;
    float4 sv[2];
    stage_load(0, sv);
    stage_store(0, sv);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) stage_load(c + 1, sv);
        float4 wa[LG_KC / 8][2];
#pragma unroll
        for (int j = 0; j < LG_KC / 8; ++j)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) wa[j][nt] = wp[((size_t)nt * J + (size_t)c * (LG_KC / 8) + j) * 64];
#pragma unroll
        for (int j = 0; j < LG_KC / 8; ++j) {
            float4 b[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) b[mt] = bt[buf][2 * j + hh][32 * mt + m0];
#define PNY_STEP(cc)                                                                          \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                          \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                      \
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j][nt].cc, b[mt].cc, acc[nt][mt], 0, 0, 0);
            PNY_STEP(x)
            PNY_STEP(y)
            PNY_STEP(z)
            PNY_STEP(w)
#undef PNY_STEP
        }
        if (c + 1 < nchunks) {
            stage_store(buf ^ 1, sv);
            __syncthreads();
        }
    }
    // scatter.  In accumulator layout a lane holds 4 channels of ONE sample, i.e. a wave instruction would touch 32 different
    // pixels' lines; the wave's 64 channels x 32 samples are turned through LDS instead, so that one atomic instruction
    // adds 64 CONSECUTIVE channels of one sample's tap (two 128-byte lines): 16x fewer line operations at the L2
    float* gv = grad + (size_t)vabs * a.Hl * a.Wl * a.L + 32 * nt0 + lane;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        __syncthreads();   // (the staging buffers / the previous half's rows are no longer read)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 t;
                t.x = acc[nt][mt][4 * q + 0];
                t.y = acc[nt][mt][4 * q + 1];
                t.z = acc[nt][mt][4 * q + 2];
                t.w = acc[nt][mt][4 * q + 3];
                *reinterpret_cast<float4*>(&tr[wave][m0][32 * nt + 8 * q + 4 * hh]) = t;
            }
        __syncthreads();
        for (int m = 0; m < 32; ++m) {
            const float val = tr[wave][m][lane];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float wk = tap_w[32 * mt + m][k];     // wave-uniform
                if (wk != 0.0f) unsafeAtomicAdd(gv + tap_off[32 * mt + m][k], wk * val);
            }
        }
    }
}

// The same kernel on split-f16 matrix products (x1 w1 + x2 w1 + x1 w2 on v_mfma_f32_32x32x16_f16, fp32 accumulation: the
// arithmetic of mlp_h2.hip / pixel_linear_h2_kernel), for scenes whose backward runs the f16x2 kernels: the GEMM is 8 launches
// and ~0.4 TFLOP of a training step with the encoder unfrozen.  Gradients have no fixed magnitude, so the B operand is
// multiplied by the power of two that puts the launch's max |dY| (tracked by the chain kernel, BwdArgs::dy_absmax) at
// 2^13 .. 2^14 before it is split, and the accumulators by its inverse on the way out (exact), as pny_dw_gemm_h2_kernel does.
typedef _Float16 lgh8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void lg_split2(float a, float b, unsigned& p0, unsigned& p1) {
    float ra, rb;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p0) : "v"(a), "v"(b));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(p0), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(p0), "v"(b));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p1) : "v"(ra), "v"(rb));
}

// TPW = 64-sample tiles per workgroup: the weights (1.5 MB of fp32 per 256 output channels at three view blocks) are what
// the kernel moves -- a workgroup with ONE tile reads them for 64 samples, 4 608 workgroups of a fine pass pull 6.9 GB through
// the L1s -- so a workgroup takes TWO consecutive tiles of its view through every weight chunk (128 accumulator registers per
// wave, 33 KB of staging): half the weight bytes per sample.  TPW = 1 is kept for launches with a single tile.
template <int TPW>
__global__ __launch_bounds__(64 * LG_NW) void latent_grad_h2_kernel(const MlpArgs a, const float* __restrict__ dy_stash, const StashLayout lay,
                                                                    const float* __restrict__ w_cat, float* __restrict__ grad, int nvb,
                                                                    const unsigned* __restrict__ dy_absmax) {
    constexpr int TS = 64 * TPW, MT = 2 * TPW;      // samples and 32-sample m-tiles of the workgroup
    __shared__ uint2 bp[2][2][LG_KC / 4][TS + 1];   // [buffer][plane][k / 4][sample]
    __shared__ __attribute__((aligned(16))) float tr[LG_NW][32][68];
    __shared__ int tap_off[TS][4];
    __shared__ float tap_w[TS][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = lane & 31, hh = lane >> 5;
    const int nblocks = a.L / 256;
    const int nb = blockIdx.x % nblocks;
    const long long tv = blockIdx.x / nblocks;
    const int v = (int)(tv % a.NS);
    const long long tile0 = (tv / a.NS) * TPW;
    const int K = nvb * HID, J = K / 8;
    float scale = 1.0f, inv_scale = 1.0f;
    {
        const unsigned mb = *dy_absmax;
        const int e = (int)((mb >> 23) & 0xffu) - 127;
        if (mb != 0u && e > -100 && e < 100) {
            scale = __uint_as_float((unsigned)(127 + 13 - e) << 23);
            inv_scale = __uint_as_float((unsigned)(127 - 13 + e) << 23);
        }
    }
    // tile t of the workgroup (the last workgroup of a view may hold a single live tile: the dead one re-reads the live
    // tile's stash and scatters with weight 0)
    long long tile_of[TPW];
    int vabs_of[TPW];   // index into the scene's view list (grouped scenes: two tiles may belong to two objects)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        tile_of[t] = tile0 + t < a.n_tiles ? tile0 + t : tile0;
        vabs_of[t] = tile_view_base(a, tile_of[t] * 64) + v;
    }
    if (tid < TS) {
        const int t = tid >> 6;
        long long s = tile_of[t] * 64 + (tid & 63);
        int offs[4];
        float wgt[4];
        const bool live = (tile0 + t < a.n_tiles) && s < a.n_points;
        if (s >= a.n_points) s = a.n_points - 1;
        sample_taps(a, vabs_of[t], s, offs, wgt);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tap_off[tid][k] = offs[k];
            tap_w[tid][k] = live ? wgt[k] * inv_scale : 0.0f;   // the inverse scale rides on the tap weight
        }
    }
    const float* dyv[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) dyv[t] = dy_stash + tile_of[t] * lay.dy_tile + (size_t)v * lay.dy_view;
    auto stage_load = [&](int c, float4 (&sv)[TPW][2]) {
        const int b = c / (HID / LG_KC), kg0 = (c % (HID / LG_KC)) * (LG_KC / 4);
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const float4* src = reinterpret_cast<const float4*>(dyv[t] + (size_t)(2 * b + 1) * STASH_SLOT) + (size_t)kg0 * 64;
            sv[t][0] = src[tid];
            sv[t][1] = src[tid + 256];
        }
    };
    auto stage_store = [&](int buf, const float4 (&sv)[TPW][2]) {
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint2 q0, q1;
                lg_split2(sv[t][h].x * scale, sv[t][h].y * scale, q0.x, q1.x);
                lg_split2(sv[t][h].z * scale, sv[t][h].w * scale, q0.y, q1.y);
                bp[buf][0][4 * h + (tid >> 6)][64 * t + (tid & 63)] = q0;
                bp[buf][1][4 * h + (tid >> 6)][64 * t + (tid & 63)] = q1;
            }
    };
    f32x16 acc[2][MT];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.f;
    const int nt0 = nb * 8 + wave * 2;
    const float4* wp = reinterpret_cast<const float4*>(w_cat) + (size_t)nt0 * J * 64 + lane;
    const int nchunks = K / LG_KC;
    float4 sv[TPW][2];
    stage_load(0, sv);
    stage_store(0, sv);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) stage_load(c + 1, sv);
        // (fetching the fragments of chunk c + 1 while chunk c multiplies was measured: 32 more registers, 1.17 against 1.13 ms
        // per launch -- the exposed L2 latency at the head of a chunk is covered by the other workgroup of the CU)
        float4 wa[LG_KC / 8][2];
#pragma unroll
        for (int j = 0; j < LG_KC / 8; ++j)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) wa[j][nt] = wp[((size_t)nt * J + (size_t)c * (LG_KC / 8) + j) * 64];
#pragma unroll
        for (int st_ = 0; st_ < LG_KC / 16; ++st_) {
            // a 16-k step: the lane's own two float4 of the fp32 weight image (k = 16 s + 4 hh + 0..3 and 16 s + 8 + 4 hh + 0..3:
            // which 8 k a fragment holds is free as long as both operands agree), split in registers
            lgh8 a1[2], a2[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                uint4 u1, u2;
                lg_split2(wa[2 * st_][nt].x, wa[2 * st_][nt].y, u1.x, u2.x);
                lg_split2(wa[2 * st_][nt].z, wa[2 * st_][nt].w, u1.y, u2.y);
                lg_split2(wa[2 * st_ + 1][nt].x, wa[2 * st_ + 1][nt].y, u1.z, u2.z);
                lg_split2(wa[2 * st_ + 1][nt].z, wa[2 * st_ + 1][nt].w, u1.w, u2.w);
                a1[nt] = __builtin_bit_cast(lgh8, u1);
                a2[nt] = __builtin_bit_cast(lgh8, u2);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const uint2 l1 = bp[buf][0][4 * st_ + hh][32 * mt + m0], h1 = bp[buf][0][4 * st_ + 2 + hh][32 * mt + m0];
                const uint2 l2 = bp[buf][1][4 * st_ + hh][32 * mt + m0], h2 = bp[buf][1][4 * st_ + 2 + hh][32 * mt + m0];
                const lgh8 b1 = __builtin_bit_cast(lgh8, make_uint4(l1.x, l1.y, h1.x, h1.y));
                const lgh8 b2 = __builtin_bit_cast(lgh8, make_uint4(l2.x, l2.y, h2.x, h2.y));
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[nt], b1, acc[nt][mt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[nt], b1, acc[nt][mt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[nt], b2, acc[nt][mt], 0, 0, 0);
            }
        }
        if (c + 1 < nchunks) {
            stage_store(buf ^ 1, sv);
            __syncthreads();
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float* gv = grad + (size_t)vabs_of[mt >> 1] * a.Hl * a.Wl * a.L + 32 * nt0 + lane;
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 t;
                t.x = acc[nt][mt][4 * q + 0];
                t.y = acc[nt][mt][4 * q + 1];
                t.z = acc[nt][mt][4 * q + 2];
                t.w = acc[nt][mt][4 * q + 3];
                *reinterpret_cast<float4*>(&tr[wave][m0][32 * nt + 8 * q + 4 * hh]) = t;
            }
        __syncthreads();
        for (int m = 0; m < 32; ++m) {
            const float val = tr[wave][m][lane];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float wk = tap_w[32 * mt + m][k];
                if (wk != 0.0f) unsafeAtomicAdd(gv + tap_off[32 * mt + m][k], wk * val);
            }
        }
    }
}

void launch_latent_grad(const MlpArgs& a, const float* dy_stash, const StashLayout& lay, const float* w_cat, float* grad, int nvb,
                        hipStream_t st, const unsigned* dy_absmax) {
    const long long blocks = (long long)a.n_tiles * a.NS * (a.L / 256);
    if (dy_absmax && a.n_tiles >= 2) {
        const long long pairs = (long long)((a.n_tiles + 1) / 2) * a.NS * (a.L / 256);
        hipLaunchKernelGGL(latent_grad_h2_kernel<2>, dim3((unsigned)pairs), dim3(64 * LG_NW), 0, st, a, dy_stash, lay, w_cat, grad, nvb, dy_absmax);
    } else if (dy_absmax)
        hipLaunchKernelGGL(latent_grad_h2_kernel<1>, dim3((unsigned)blocks), dim3(64 * LG_NW), 0, st, a, dy_stash, lay, w_cat, grad, nvb, dy_absmax);
    else
        hipLaunchKernelGGL(latent_grad_kernel, dim3((unsigned)blocks), dim3(64 * LG_NW), 0, st, a, dy_stash, lay, w_cat, grad, nvb);
}

}  // namespace pny
