// Backward pass of the conditioned ResNet MLP for gfx950 (MI355X) -- SURVEY.md 8f-1: the gradient the reference gets from
// loss.backward() through PixelNeRFNet.forward / ResnetFC.forward (reference src/model/resnetfc.py:134-186,
// src/model/models.py:153-318) w.r.t. every MLP parameter.
//
// Three kernels, all exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), all on 64-sample tiles:
//   1. pny_mlp_kernel<.., STASH> (mlp.hip): the forward chain once more in the reference's operation order, writing
//      every GEMM's B operand X_l (relu'd activations, gathered latent, positional code) to an HBM stash in the LDS
//      operand layout [feature/4][sample] float4 (pny_common.h StashLayout).
//   2. pny_mlp_bwd_kernel (here): the dX chain.  Same transposed formulation and weight-stream ring as the forward:
//      dX^T[k][m] = W^T[k][n] dY^T[n][m] with the TRANSPOSED packed weights as the A operand and dY^T in the LDS
//      activation buffer as the B operand; the gradient of the residual stream stays in the accumulators (dh), a second
//      accumulator set carries the block-internal gradient.  relu masks come from the stash (x > 0); every dY_l a
//      weight gradient needs is written to a second stash in the same layout.
//   3. pny_dw_gemm_kernel (here): dW_l[n][k] = sum_samples dY_l[s][n] X_l[s][k] as an LDS-staged split-K GEMM over the
//      two stashes (256x256 output tile per workgroup, K = samples x views), bias gradients as column sums of dY_l in
//      the same pass; a deterministic reduction over the K splits writes the gradients in the state_dict's layout.
// Per-tile atomics into dW were rejected (1 MB of read-modify-write per layer per 64 samples).
#include <algorithm>
#include <cstring>
#include <vector>

#include "mlp_bwd_core.h"
#include "pny_rng.h"

namespace pny {

// ---------------------------------------------------------------------------------------------- chain kernel
__device__ __forceinline__ float bwd_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// act[feature/4][m] = acc (no activation): the B operand of the next transposed GEMM; optionally also to the dY stash.
// `amax` collects max |v| over everything written to the dY stash: the split-f16 weight-gradient GEMM scales dY by a power of two
// derived from it (pny_dw_gemm_h2_kernel).
template <int NT, int MT, bool TO_LDS, bool TO_GLOBAL>
__device__ __forceinline__ void store_plain(const f32x16 (&acc)[NT][MT], float4* __restrict__ act, const StashRef& g,
                                            int wave, int lane, float& amax) {
    constexpr int TMc = 32 * MT;
    const int m0 = lane & 31, hh = lane >> 5;
    const unsigned lo = acc_lane_off<NT, MT>(wave, lane);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = make_float4(acc[nt][mt][4 * q + 0], acc[nt][mt][4 * q + 1], acc[nt][mt][4 * q + 2],
                                             acc[nt][mt][4 * q + 3]);
                const int idx = (8 * NT * wave + 8 * nt + 2 * q + hh) * TMc + 32 * mt + m0;
                if (TO_LDS) act[idx] = v;
                if (TO_GLOBAL) {
                    stash_st(g, lo, quad_off<NT, MT>(nt, mt, q), v);
                    amax = fmaxf(fmaxf(amax, fabsf(v.x)), fabsf(v.y));
                    amax = fmaxf(fmaxf(amax, fabsf(v.z)), fabsf(v.w));
                }
            }
}

// Reverse of one pre-activation residual block (reference resnetfc.py:53-62: net = fc_0(relu(h)); h' = h + fc_1(relu(net))):
//   dnet = (fc_1^T dh') * [net > 0];   dh = dh' + (fc_0^T dnet) * [h > 0]   (times `scale` = 1/NS below the cross-view mean)
// dh' arrives in the accumulators `dh` and leaves as dh.  Written to the dY stash: dnet (dY of fc_0) and dh (dY of
// lin_z of this block / fc_1 of the previous one / lin_in).
template <class C>
__device__ __forceinline__ void block_bwd(f32x16 (&dh)[C::NT][C::MT], WRing<C::WDEPTH, C::NT>& ring, const WStream& ws,
                                          const WSeg& s_fc1t, const WSeg& s_fc0t, const WSeg& after, float4* act,
                                          const StashRef& x_h, const StashRef& x_net, const StashRef& dy_dnet, const StashRef& dy_dh,
                                          float scale, int wave, int lane, float& amax) {
    constexpr int NT = C::NT, MT = C::MT;
    f32x16 t[NT][MT];
    __syncthreads();  // every wave is done reading the buffer (previous GEMM)
    store_plain<NT, MT, true, false>(dh, act, dy_dh, wave, lane, amax);
    __syncthreads();
    acc_zero<NT, MT>(t);
    gemm_run<C>(t, ring, ws, s_fc1t, s_fc0t, act, lane);
    mask_by<NT, MT>(t, x_net, wave, lane);
    __syncthreads();
    store_plain<NT, MT, true, true>(t, act, dy_dnet, wave, lane, amax);
    __syncthreads();
    acc_zero<NT, MT>(t);
    gemm_run<C>(t, ring, ws, s_fc0t, after, act, lane);
    mask_by<NT, MT>(t, x_h, wave, lane);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dh[nt][mt][r] = (dh[nt][mt][r] + t[nt][mt][r]) * scale;
    store_plain<NT, MT, false, true>(dh, nullptr, dy_dh, wave, lane, amax);
}

template <class C>
__global__ __launch_bounds__(C::THREADS, C::WPS) void pny_mlp_bwd_kernel(const BwdArgs a) {
    constexpr int NT = C::NT, MT = C::MT, TMc = C::TM;
    static_assert(TMc == 64, "the stash layout is defined on 64-sample tiles");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float4* act = reinterpret_cast<float4*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = a.n_blocks;
    const int nvb = a.combine_layer < nb ? a.combine_layer : nb;
    const int npost = nb - nvb;
    const WStream ws = wstream_raw(a.w_base, a.w_bytes, lane);
    const WSeg s_out = wseg<NT>(ws, a.wT_out, D_IN_PAD / 8, 0, D_IN_PAD / 8, wave);
    auto fc1t = [&](int b) { return wseg<NT>(ws, a.wT_fc1[b], 64, 0, 64, wave); };
    auto fc0t = [&](int b) { return wseg<NT>(ws, a.wT_fc0[b], 64, 0, 64, wave); };
    WRing<C::WDEPTH, NT> ring;
    ring_fill(ring, ws, s_out);
    const float inv_ns = 1.0f / (float)a.NS;
    float amax = 0.f;

    for (long long tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        float* dy_rec = a.dy_stash + tile * a.lay.dy_tile;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x_stash + tile * a.lay.x_tile), 0,
                                                                            (int)(a.lay.x_tile * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(dy_rec, 0, (int)(a.lay.dy_tile * 4), 0x00020000);
        auto x_post = [&](int i) { return StashRef{xr, ((unsigned)a.lay.x_post + (unsigned)i * (unsigned)STASH_SLOT) * 4u}; };
        auto x_act = [&](int v, int i) {
            return StashRef{xr, ((unsigned)v * (unsigned)a.lay.x_view + (unsigned)a.lay.x_act + (unsigned)i * (unsigned)STASH_SLOT) * 4u};
        };
        float4* dy_draw = reinterpret_cast<float4*>(dy_rec + a.lay.dy_post);
        auto dy_post = [&](int i) { return StashRef{yr, ((unsigned)a.lay.dy_post + (unsigned)STASH_SMALL + (unsigned)i * (unsigned)STASH_SLOT) * 4u}; };
        auto dy_view = [&](int v, int i) { return StashRef{yr, ((unsigned)v * (unsigned)a.lay.dy_view + (unsigned)i * (unsigned)STASH_SLOT) * 4u}; };

        // ---- head: gradient w.r.t. lin_out's output through sigmoid / relu (reference models.py:312-317), as the B
        // operand of lin_out^T (d_out rows padded to 64) and as the dY of lin_out for the weight-gradient GEMM
        __syncthreads();
        for (int idx = tid; idx < (D_IN_PAD / 4) * TMc; idx += C::THREADS) {
            const int kg = idx / TMc, m = idx % TMc;
            const long long s = tile * TMc + m;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (s < a.n_points) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int o = 4 * kg + c;
                    if (o < a.d_out) {
                        const float g = a.d_out_grad[s * a.d_out + o];
                        if (a.yolo) {
                            v[c] = g;
                        } else {
                            const float y = a.out[s * a.d_out + o];
                            v[c] = o < 3 ? g * (y * (1.0f - y)) : (o == 3 ? (y > 0.f ? g : 0.f) : g);
                        }
                    }
                }
            }
            const float4 v4 = make_float4(v[0], v[1], v[2], v[3]);
            act[idx] = v4;
            dy_draw[idx] = v4;   // lin_out's dY
            amax = fmaxf(fmaxf(amax, fabsf(v[0])), fabsf(v[1]));
            amax = fmaxf(fmaxf(amax, fabsf(v[2])), fabsf(v[3]));
        }
        __syncthreads();
        f32x16 dh[NT][MT];
        acc_zero<NT, MT>(dh);
        gemm_run<C>(dh, ring, ws, s_out, fc1t(nb - 1), act, lane);
        mask_by<NT, MT>(dh, x_post(2 * npost), wave, lane);                 // relu(h_top) > 0
        store_plain<NT, MT, false, true>(dh, nullptr, dy_post(0), wave, lane, amax);  // dh_top: dY of the last block's fc_1

        // ---- post-combine blocks, last to first; the first of them also applies the 1/NS of the cross-view mean
        for (int b = nb - 1; b >= nvb; --b) {
            const int i = b - nvb;
            const WSeg after = b > nvb ? fc1t(b - 1) : (nvb > 0 ? fc1t(nvb - 1) : s_out);
            block_bwd<C>(dh, ring, ws, fc1t(b), fc0t(b), after, act, x_post(2 * i), x_post(2 * i + 1), dy_post(1 + 2 * i),
                         dy_post(2 + 2 * i), b == nvb ? inv_ns : 1.0f, wave, lane, amax);
        }
        // dhm: what every view's last per-view block receives (dh_top itself when there is no post-combine block)
        const StashRef dhm = npost > 0 ? dy_post(2) : dy_post(0);
        for (int v = 0; v < a.NS && nvb > 0; ++v) {
            if (v > 0) acc_load<NT, MT>(dh, dhm, wave, lane);
            for (int b = nvb - 1; b >= 0; --b) {
                const WSeg after = b > 0 ? fc1t(b - 1) : (v + 1 < a.NS ? fc1t(nvb - 1) : s_out);
                block_bwd<C>(dh, ring, ws, fc1t(b), fc0t(b), after, act, x_act(v, 2 * b), x_act(v, 2 * b + 1),
                             dy_view(v, 2 * b), dy_view(v, 2 * b + 1), 1.0f, wave, lane, amax);
            }
        }
    }
    if (a.dy_absmax) {   // non-negative floats order like their bit patterns
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if (lane == 0) {
            atomicMax(a.dy_absmax, __float_as_uint(amax));
            // f16-range guard: a non-finite gradient was written to the dY stash (the split-f16 consumers -- this chain's
            // scaled planes, the weight-gradient GEMM's scale -- cannot represent it): PNY_RANGE_GRADIENT
            if (!(amax < 3.0e38f)) range_report(a.range_flag, 2u);
        }
    }
}

void launch_mlp_bwd(const BwdArgs& a, int grid, hipStream_t st) {
    using C = Cfg<2, 2>;
    static bool attr_set[64] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    dev_ &= 63;
    if (!attr_set[dev_]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pny_mlp_bwd_kernel<C>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        attr_set[dev_] = true;
    }
    hipLaunchKernelGGL((pny_mlp_bwd_kernel<C>), dim3(grid), dim3(C::THREADS), C::LDS, st, a);
}

// ---------------------------------------------------------------------------------------------- weight-gradient GEMM
// C[n][k] = sum over (tile, view) of sum_m dY[kgA][m] X[kgX][m]: both operands arrive in the stash layout (sample
// index fastest within a feature quad), are loaded as whole 1-KB rows and TRANSPOSED on their way into LDS to
// [sample][feature] images with a row stride of 260 floats: the staging ds_write_b128 (8 lanes x 4 banks) and the
// fragment ds_read_b32 (32 consecutive features of one sample) are both conflict-free.  8 waves = 2 (rows) x 4 (cols),
// 128 x 64 per wave = 4 x 2 MFMA tiles.  The next tile's rows are fetched into registers underneath the MFMAs.
constexpr int DW_LD = 260;
constexpr int DW_TILE = 256;
constexpr size_t DW_LDS_BYTES = (size_t)2 * 64 * DW_LD * sizeof(float);

template <bool FULL>   // FULL: every item is a complete 256 x 256 tile (all 512-wide layers): predicate-free loop
__global__ __launch_bounds__(512, 2) void pny_dw_gemm_kernel(const DwJob* __restrict__ jobs, const DwItem* __restrict__ items,
                                                             const float* __restrict__ x_stash,
                                                             const float* __restrict__ dy_stash, long long x_tile,
                                                             long long dy_tile, float* __restrict__ partial,
                                                             float* __restrict__ bias_partial) {
    extern __shared__ __attribute__((aligned(16))) float dw_lds[];
    const DwItem it = items[blockIdx.x];
    const DwJob jb = jobs[it.job];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3, hh = lane >> 5, l31 = lane & 31;
    const int row0 = it.mt * DW_TILE, col0 = it.nt * DW_TILE;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    bool live_a[4], live_x[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) live_a[i] = row0 + wr * 128 + 32 * i < jb.a_rows;
#pragma unroll
    for (int j = 0; j < 2; ++j) live_x[j] = col0 + wc * 64 + 32 * j < jb.x_cols;

    // Pipeline over HALF tiles (32 samples): two LDS buffers of [32][260] x (A, X).  While the MFMAs of half h run on
    // buffer h & 1, half h + 1 (fetched into registers one step earlier) is written to the other buffer and the global
    // loads of half h + 2 are issued; one barrier per half.  No MFMA-idle staging phase (the single-buffered 64-sample
    // version spent 26 % of its time in it).
    constexpr int HB = 32 * DW_LD;                 // floats of one [32][DW_LD] image
    auto bufA = [&](int b) { return dw_lds + (b ? 2 * HB : 0); };
    auto bufX = [&](int b) { return dw_lds + (b ? 3 * HB : HB); };
    const int ms = tid & 31, kg_s = tid >> 5;      // staging: this thread's sample within the half, first quad row (0..15)
    float4 ra[4], rx[4];
    auto fetch = [&](int h) {
        const int tv = it.tv_lo + (h >> 1), half = h & 1;
        const int tile = tv / jb.n_views, v = tv - tile * jb.n_views;
        const float4* ga = reinterpret_cast<const float4*>(dy_stash + (long long)tile * dy_tile + jb.a_off + (long long)v * jb.a_view) + (row0 / 4) * 64 + 32 * half + ms;
        const float4* gx = reinterpret_cast<const float4*>(x_stash + (long long)tile * x_tile + jb.x_off + (long long)v * jb.x_view) + (col0 / 4) * 64 + 32 * half + ms;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kg = kg_s + 16 * i;
            ra[i] = (row0 + 4 * kg < jb.a_rows) ? ga[kg * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
            rx[i] = (col0 + 4 * kg < jb.x_cols) ? gx[kg * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&](int b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kg = kg_s + 16 * i;
            *reinterpret_cast<float4*>(bufA(b) + ms * DW_LD + 4 * kg) = ra[i];
            *reinterpret_cast<float4*>(bufX(b) + ms * DW_LD + 4 * kg) = rx[i];
        }
    };
    const int n_half = 2 * (it.tv_hi - it.tv_lo);
    if (n_half > 0) {
        fetch(0);
        stage(0);
        if (n_half > 1) fetch(1);
        __syncthreads();
    }
    for (int h = 0; h < n_half; ++h) {
        const int cb = h & 1;
        if (h + 1 < n_half) {
            stage(cb ^ 1);                     // half h + 1: every wave left that buffer at the previous barrier
            if (h + 2 < n_half) fetch(h + 2);
        }
        const float* pa = bufA(cb) + hh * DW_LD + wr * 128 + l31;
        const float* px = bufX(cb) + hh * DW_LD + wc * 64 + l31;
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {   // (an explicit LDS-read-ahead of one k-step measured the same: 13.9 vs 13.6 ms)
            float av[4], xv[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = pa[2 * ks * DW_LD + 32 * i];
#pragma unroll
            for (int j = 0; j < 2; ++j) xv[j] = px[2 * ks * DW_LD + 32 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (FULL || live_a[i]) {
                    bsum[i] += av[i];
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (FULL || live_x[j]) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], xv[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // partial tile: P[split][row][col], row stride = the job's padded column count
    float* P = partial + it.part_off;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (!live_a[i] || !live_x[j]) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wr * 128 + 32 * i + 8 * (r >> 2) + 4 * hh + (r & 3);
                const int col = col0 + wc * 64 + 32 * j + l31;
                if (row < jb.a_rows && col < jb.x_cols) P[(long long)row * jb.x_cols + col] = acc[i][j][r];
            }
        }
    if (it.nt == 0 && wc == 0) {  // column sums of dY (bias gradients): the two lane halves hold alternate samples
        float* B = bias_partial + it.bias_off;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s = bsum[i] + __shfl_xor(bsum[i], 32, 64);
            const int row = row0 + wr * 128 + 32 * i + l31;
            if (hh == 0 && live_a[i] && row < jb.a_rows) B[row] = s;
        }
    }
}

// ---- the same GEMM on the f16 matrix cores with split fp32 operands (the arithmetic of mlp_h2.hip: x = x1 + x2 in two f16
// planes, x1 y1 + x2 y1 + x1 y2 on v_mfma_f32_32x32x16_f16 with fp32 accumulation; 5.3x the matrix rate of the fp32 MFMA).
// Clipped tiles (lin_in's 64 columns, lin_out's 64 rows) run a predicated instantiation, as in the fp32 kernel.
//   * The contraction runs over SAMPLES, and an f16 fragment holds 8 consecutive k of one row: a staging thread loads, for one
//     feature quad, the 8 samples {c, c + 4, ..., c + 28} of a 32-sample half (lanes c = 0..3 adjacent: 64-byte segments),
//     i.e. an 8 x 4 block in registers, and writes per feature ONE 16-byte vector per plane -- the transpose happens in
//     registers.  Which 8 samples share a fragment is irrelevant as long as both operands agree (k is a dummy index).
//   * LDS image per operand and plane: [c = 0..3][feature 0..255 (+1 pad)] x 16 bytes; a fragment read is 32 consecutive
//     features of one c: 512 contiguous bytes.  Two buffers of (A, X) x 2 planes = 128.5 KiB, one barrier per half.
//   * Gradients span many orders of magnitude and f16 does not: dY is multiplied by a power of two that puts the launch's
//     max |dY| (tracked by the chain kernel, BwdArgs::dy_absmax) at 2^13..2^14, and the accumulators are multiplied by its
//     inverse on the way out -- exact.  Elements down to 2^-27 of the maximum keep all 22 bits, smaller ones an absolute
//     2^-38 of the maximum.
//   * Bias gradients (column sums of dY) are taken by the staging threads from the fp32 values.
typedef _Float16 dwh8 __attribute__((ext_vector_type(8)));
constexpr int DWH_CS = 257;                                  // 16-byte entries per c row (256 features + 1 pad)
constexpr int DWH_PLANE = 4 * DWH_CS;                        // entries per plane
constexpr int DWH_BUF = 2 * 2 * DWH_PLANE;                   // entries per buffer: (A, X) x 2 planes
constexpr size_t DWH_LDS_BYTES = (size_t)2 * DWH_BUF * 16;   // 131584

__device__ __forceinline__ void dwh_split2(float a, float b, unsigned& p0, unsigned& p1) {
    float ra, rb;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p0) : "v"(a), "v"(b));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(p0), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(p0), "v"(b));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p1) : "v"(ra), "v"(rb));
}

template <bool FULL>   // FULL: complete 256 x 256 tiles, predicate-free; otherwise rows / columns beyond the job are zero-filled and skipped
__global__ __launch_bounds__(512, 2) void pny_dw_gemm_h2_kernel(const DwJob* __restrict__ jobs, const DwItem* __restrict__ items,
                                                                const float* __restrict__ x_stash,
                                                                const float* __restrict__ dy_stash, long long x_tile,
                                                                long long dy_tile, float* __restrict__ partial,
                                                                float* __restrict__ bias_partial,
                                                                const unsigned* __restrict__ dy_absmax) {
    extern __shared__ __attribute__((aligned(16))) float dw_lds[];
    uint4* lds = reinterpret_cast<uint4*>(dw_lds);
    const DwItem it = items[blockIdx.x];
    const DwJob jb = jobs[it.job];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3, hh = lane >> 5, l31 = lane & 31;
    const int row0 = it.mt * DW_TILE, col0 = it.nt * DW_TILE;

    // scale = 2^(13 - floor(log2(max |dY|))), clamped to the normal range
    float scale = 1.0f, inv_scale = 1.0f;
    {
        const unsigned mb = *dy_absmax;
        int e = (int)((mb >> 23) & 0xffu) - 127;
        if (mb != 0u && e > -100 && e < 100) {
            scale = __uint_as_float((unsigned)(127 + 13 - e) << 23);
            inv_scale = __uint_as_float((unsigned)(127 - 13 + e) << 23);
        }
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging role of this thread: operand (waves 0..3: A = dY, waves 4..7: X), feature quad fq of the tile, sample group c
    const int op = wave >> 2;
    const int fq = (tid & 255) >> 2, c = tid & 3;
    const float sc = op == 0 ? scale : 1.0f;
    const bool want_bias = op == 0 && it.nt == 0;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    float4 r[8];
    // clipped tiles: quads beyond the job's extent load nothing (zeros); quads beyond the last 32-wide MFMA tile that holds
    // live data are not even written
    const int extent = op == 0 ? jb.a_rows - row0 : jb.x_cols - col0;
    const bool q_load = FULL || 4 * fq < extent, q_write = FULL || 4 * fq < ((extent + 31) & ~31);
    bool live_a[4], live_x[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) live_a[i] = FULL || row0 + wr * 128 + 32 * i < jb.a_rows;
#pragma unroll
    for (int j = 0; j < 2; ++j) live_x[j] = FULL || col0 + wc * 64 + 32 * j < jb.x_cols;
    auto fetch = [&](int h) {
        const int tv = it.tv_lo + (h >> 1), half = h & 1;
        const int tile = tv / jb.n_views, v = tv - tile * jb.n_views;
        const float* rec = op == 0 ? dy_stash + (long long)tile * dy_tile + jb.a_off + (long long)v * jb.a_view
                                   : x_stash + (long long)tile * x_tile + jb.x_off + (long long)v * jb.x_view;
        const float4* g = reinterpret_cast<const float4*>(rec) + ((op == 0 ? row0 : col0) / 4 + fq) * 64 + 32 * half + c;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = q_load ? g[4 * i] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stage = [&](int b) {
        uint4* img = lds + b * DWH_BUF + op * (2 * DWH_PLANE) + c * DWH_CS + 4 * fq;
        if (!q_write) return;
        if (want_bias) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {   // (asm: keeps the SLP vectoriser from packing these into v_pk_add_f32 beside the MFMAs)
                asm("v_add_f32 %0, %1, %0" : "+v"(bs[0]) : "v"(r[i].x));
                asm("v_add_f32 %0, %1, %0" : "+v"(bs[1]) : "v"(r[i].y));
                asm("v_add_f32 %0, %1, %0" : "+v"(bs[2]) : "v"(r[i].z));
                asm("v_add_f32 %0, %1, %0" : "+v"(bs[3]) : "v"(r[i].w));
            }
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (f == 0 ? r[i].x : f == 1 ? r[i].y : f == 2 ? r[i].z : r[i].w) * sc;
            uint4 p0, p1;
            dwh_split2(v[0], v[1], p0.x, p1.x);
            dwh_split2(v[2], v[3], p0.y, p1.y);
            dwh_split2(v[4], v[5], p0.z, p1.z);
            dwh_split2(v[6], v[7], p0.w, p1.w);
            img[f] = p0;
            img[DWH_PLANE + f] = p1;
        }
    };
    const int n_half = 2 * (it.tv_hi - it.tv_lo);
    if (n_half > 0) {
        fetch(0);
        stage(0);
        if (n_half > 1) fetch(1);
        __syncthreads();
    }
    auto mm = [&](int cb) {   // the MFMAs of one half on buffer cb
        const uint4* pa = lds + cb * DWH_BUF + hh * DWH_CS + wr * 128 + l31;
        const uint4* px = lds + cb * DWH_BUF + 2 * DWH_PLANE + hh * DWH_CS + wc * 64 + l31;
#pragma unroll
        for (int st = 0; st < 2; ++st) {       // 16 samples per step: this lane's 8 are group c = 2 st + hh
            dwh8 a1[4] = {}, a2[4] = {}, x1[2] = {}, x2[2] = {};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (FULL || live_a[i]) {
                    a1[i] = __builtin_bit_cast(dwh8, pa[2 * st * DWH_CS + 32 * i]);
                    a2[i] = __builtin_bit_cast(dwh8, pa[DWH_PLANE + 2 * st * DWH_CS + 32 * i]);
                }
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (FULL || live_x[j]) {
                    x1[j] = __builtin_bit_cast(dwh8, px[2 * st * DWH_CS + 32 * j]);
                    x2[j] = __builtin_bit_cast(dwh8, px[DWH_PLANE + 2 * st * DWH_CS + 32 * j]);
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (FULL || (live_a[i] && live_x[j])) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[i], x1[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (FULL || (live_a[i] && live_x[j])) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[i], x1[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (FULL || (live_a[i] && live_x[j])) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[i], x2[j], acc[i][j], 0, 0, 0);
        }
    };
    for (int h = 0; h < n_half; ++h) {
        const int cb = h & 1;
        if (h + 1 < n_half) {
            stage(cb ^ 1);                     // half h + 1: every wave left that buffer at the previous barrier
            if (h + 2 < n_half) fetch(h + 2);
        }
        mm(cb);
        __syncthreads();
    }
    float* P = partial + it.part_off;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (!FULL && !(live_a[i] && live_x[j])) continue;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int row = row0 + wr * 128 + 32 * i + 8 * (rr >> 2) + 4 * hh + (rr & 3);
                const int col = col0 + wc * 64 + 32 * j + l31;
                if (FULL || (row < jb.a_rows && col < jb.x_cols)) P[(long long)row * jb.x_cols + col] = acc[i][j][rr] * inv_scale;
            }
        }
    if (want_bias) {   // the four lanes c = 0..3 of a feature quad hold the sums of their own samples
        float* B = bias_partial + it.bias_off;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            float t = bs[f];
            t += __shfl_xor(t, 1, 64);
            t += __shfl_xor(t, 2, 64);
            if (c == 0 && (FULL || row0 + 4 * fq + f < jb.a_rows)) B[row0 + 4 * fq + f] = t;
        }
    }
}

// grad (+)= sum over splits of the job's partial tiles, in a fixed order (deterministic).
__global__ __launch_bounds__(256) void pny_dw_reduce_kernel(const DwTarget* __restrict__ targets, const float* __restrict__ partial,
                                                            const float* __restrict__ bias_partial, int accumulate) {
    const DwTarget t = targets[blockIdx.y];
    const long long n_w = (long long)t.rows * t.cols;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n_w) {
        if (!t.w) return;
        const int row = (int)(idx / t.cols), col = (int)(idx - (long long)row * t.cols);
        const float* p = partial + t.part_off + (long long)row * t.pcols + col;
        float s = 0.f;
        for (int k = 0; k < t.splits; ++k) s += p[(long long)k * t.prows * t.pcols];
        t.w[idx] = accumulate ? t.w[idx] + s : s;
    } else if (idx < n_w + t.rows) {
        const int row = (int)(idx - n_w);
        const float* p = bias_partial + t.bias_off + row;
        float s = 0.f;
        for (int k = 0; k < t.splits; ++k) s += p[(long long)k * t.prows];
        if (t.b0) t.b0[row] = accumulate ? t.b0[row] + s : s;
        if (t.b1) t.b1[row] = accumulate ? t.b1[row] + s : s;
    }
}

// items_dev: n_part clipped items first (lin_in's 64 columns, lin_out's 64 rows: mostly staging, few FLOPs), then n_full
// complete tiles.  The clipped ones run on `aux` (forked from and joined back into `st`) so that they share the chip with
// the bulk instead of forming a tail of 60 workgroups on 256 CUs.
void launch_dw_gemm(const DwJob* jobs_dev, const DwItem* items_dev, int n_part, int n_full, const float* x_stash,
                    const float* dy_stash, long long x_tile, long long dy_tile, float* partial, float* bias_partial, hipStream_t st,
                    hipStream_t aux, hipEvent_t ev_fork, hipEvent_t ev_join, const unsigned* dy_absmax) {
    static bool attr_set[64] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    dev_ &= 63;
    if (!attr_set[dev_]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pny_dw_gemm_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)DW_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pny_dw_gemm_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)DW_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pny_dw_gemm_h2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)DWH_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pny_dw_gemm_h2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)DWH_LDS_BYTES);
        attr_set[dev_] = true;
    }
    const bool fork = n_part > 0 && n_full > 0 && aux && ev_fork && ev_join;
    hipStream_t sp = st;
    if (fork && hipEventRecord(ev_fork, st) == hipSuccess && hipStreamWaitEvent(aux, ev_fork, 0) == hipSuccess) sp = aux;
    if (n_part > 0 && dy_absmax)   // split-f16 matrix path (dY scaled by the tracked maximum)
        hipLaunchKernelGGL(pny_dw_gemm_h2_kernel<false>, dim3(n_part), dim3(512), DWH_LDS_BYTES, sp, jobs_dev, items_dev, x_stash, dy_stash,
                           x_tile, dy_tile, partial, bias_partial, dy_absmax);
    else if (n_part > 0)
        hipLaunchKernelGGL(pny_dw_gemm_kernel<false>, dim3(n_part), dim3(512), DW_LDS_BYTES, sp, jobs_dev, items_dev, x_stash, dy_stash,
                           x_tile, dy_tile, partial, bias_partial);
    if (n_full > 0 && dy_absmax)
        hipLaunchKernelGGL(pny_dw_gemm_h2_kernel<true>, dim3(n_full), dim3(512), DWH_LDS_BYTES, st, jobs_dev, items_dev + n_part, x_stash,
                           dy_stash, x_tile, dy_tile, partial, bias_partial, dy_absmax);
    else if (n_full > 0)
        hipLaunchKernelGGL(pny_dw_gemm_kernel<true>, dim3(n_full), dim3(512), DW_LDS_BYTES, st, jobs_dev, items_dev + n_part, x_stash,
                           dy_stash, x_tile, dy_tile, partial, bias_partial);
    if (sp != st) {
        (void)hipEventRecord(ev_join, aux);
        (void)hipStreamWaitEvent(st, ev_join, 0);
    }
}

void launch_dw_reduce(const DwTarget* targets_dev, int n_targets, long long max_elems, const float* partial,
                      const float* bias_partial, int accumulate, hipStream_t st) {
    if (n_targets <= 0) return;
    hipLaunchKernelGGL(pny_dw_reduce_kernel, dim3((unsigned)((max_elems + 255) / 256), n_targets), dim3(256), 0, st,
                       targets_dev, partial, bias_partial, accumulate);
}

// ---------------------------------------------------------------------------------------------- composite backward
// Reverse of composite_kernel (render_kernels.hip; reference nerf.py:184-188, 229-250).  One wavefront per ray.
//   alpha_k = 1 - exp(-delta_k s_k), s_k = relu(sigma_k [+ noise]); A_k = 1 - alpha_k + 1e-10; T_k = prod_{j<k} A_j; w_k = alpha_k T_k
//   G_k = dL/dw_k = sum_c g_rgb_c (c_kc - [white]) + g_depth z_k + g_w_k
//   dL/dalpha_k = G_k T_k - (sum_{j>k} G_j w_j) / A_k      (cumprod backward for non-zero factors, as autograd computes it)
//   dL/ds_k = dL/dalpha_k delta_k exp(-delta_k s_k);  dL/ddelta_k = dL/dalpha_k s_k exp(-delta_k s_k)
// Outputs: d_samp (n,K,4) = gradient w.r.t. the model's per-sample outputs [rgb (after sigmoid), sigma (after relu)];
// d_z (n,K) (optional) = gradient w.r.t. the sample depths through delta and through depth = sum w z.

__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ rays, const float* __restrict__ z,
                                                            const float* __restrict__ samp, const float* __restrict__ noise,
                                                            long long n, int K, int white, const float* __restrict__ g_rgb,
                                                            const float* __restrict__ g_depth, const float* __restrict__ g_w,
                                                            float* __restrict__ d_samp, float* __restrict__ d_z) {
    extern __shared__ float cb_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long ray = (long long)blockIdx.x * 4 + wv;
    if (ray >= n) return;
    float* Tw = cb_lds + (size_t)wv * 2 * K;  // T_k, later w_k
    float* Ad = Tw + K;                       // alpha_k, later dL/ddelta_k
    const float far = rays[ray * 8 + 7];
    const float* zr = z + ray * K;
    const float4* sr = reinterpret_cast<const float4*>(samp) + ray * K;
    const float gr = g_rgb ? g_rgb[ray * 3 + 0] : 0.f, gg = g_rgb ? g_rgb[ray * 3 + 1] : 0.f,
                gb = g_rgb ? g_rgb[ray * 3 + 2] : 0.f;
    const float gd = g_depth ? g_depth[ray] : 0.f;
    const float wsub = white ? (gr + gg + gb) : 0.f;  // rgb + 1 - sum(w): every w_k carries -sum_c g_rgb_c

    float carry = 1.0f;
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + lane;
        const bool on = k < K;
        float alpha = 0.f;
        if (on) {
            const float zk = zr[k];
            const float znext = (k + 1 < K) ? zr[k + 1] : far;
            float sg = sr[k].w;
            if (noise) sg += noise[ray * K + k];
            alpha = 1.0f - expf(-(znext - zk) * fmaxf(sg, 0.f));
        }
        float pprod = on ? (1.0f - alpha + 1e-10f) : 1.0f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float up = __shfl_up(pprod, o, 64);
            if (lane >= o) pprod *= up;
        }
        float excl = __shfl_up(pprod, 1, 64);
        if (lane == 0) excl = 1.0f;
        if (on) {
            Tw[k] = carry * excl;
            Ad[k] = alpha;
        }
        carry *= __shfl(pprod, 63, 64);
    }
    // reverse pass: suffix sums of G_j w_j
    float suffix = 0.f;  // sum over samples beyond the current chunk
    const int nchunk = (K + 63) / 64;
    for (int c = nchunk - 1; c >= 0; --c) {
        const int k = c * 64 + lane;
        const bool on = k < K;
        float G = 0.f, w = 0.f, T = 0.f, alpha = 0.f, zk = 0.f, delta = 0.f, s = 0.f, sg_eff = 0.f;
        float4 sm = {0.f, 0.f, 0.f, 0.f};
        if (on) {
            zk = zr[k];
            const float znext = (k + 1 < K) ? zr[k + 1] : far;
            delta = znext - zk;
            sm = sr[k];
            sg_eff = sm.w + (noise ? noise[ray * K + k] : 0.f);
            s = fmaxf(sg_eff, 0.f);
            T = Tw[k];
            alpha = Ad[k];
            w = alpha * T;
            G = gr * sm.x + gg * sm.y + gb * sm.z - wsub + gd * zk + (g_w ? g_w[ray * K + k] : 0.f);
        }
        // inclusive suffix scan over the lanes of the chunk
        float incl = G * w;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float dn = __shfl_down(incl, o, 64);
            if (lane + o < 64) incl += dn;
        }
        const float excl = incl - G * w + suffix;   // sum_{j > k} G_j w_j
        suffix += __shfl(incl, 0, 64);
        if (on) {
            const float A = 1.0f - alpha + 1e-10f;
            const float dalpha = G * T - excl / A;
            const float e = expf(-delta * s);
            const float ds = dalpha * (delta * e);
            const float dsig = sg_eff > 0.f ? ds : 0.f;
            reinterpret_cast<float4*>(d_samp)[ray * K + k] = make_float4(gr * w, gg * w, gb * w, dsig);
            Tw[k] = w;
            Ad[k] = dalpha * (s * e);   // dL/ddelta_k
        }
    }
    if (d_z) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int k = lane; k < K; k += 64) d_z[ray * K + k] = gd * Tw[k] + (k > 0 ? Ad[k - 1] : 0.f) - Ad[k];
    }
}

void launch_composite_bwd(const float* rays, const float* z, const float* samp, const float* noise, long long n, int k,
                          int white, const float* g_rgb, const float* g_depth, const float* g_w, float* d_samp, float* d_z,
                          hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), (size_t)4 * 2 * k * sizeof(float), st,
                       rays, z, samp, noise, n, k, white, g_rgb, g_depth, g_w, d_samp, d_z);
}

// ---------------------------------------------------------------------------------------------- sample-depth gradients
// The reference centres the fine pass's depth samples on the coarse pass's depth WITHOUT detaching it
// (nerf.py:156-167, 296-298): the fine loss reaches the coarse MLP through the positions of those samples.
//   dL/dz_j = [composite: through the deltas and the depth sum]  +  d . dL/dp_j,   p_j = o + z_j d,
//   dL/dp = sum over views R^T (dL/dxr + dL/dxc):  xr = R p feeds the positional code (code.py:30-42) and, as
//   xc = xr + t, the projection uv (models.py:219-230) whose bilinear lookup (encoder.py:101) is differentiated as
//   grid_sampler does (zeros padding: out-of-range taps count as 0).
// Only the kfd depth samples per ray need it, so this is a separate small kernel over the dY stash of the fine pass
// (one wavefront per selected sample) and not part of the chain kernel:
//   dL/d(code input) = lin_in^T dh_in(0);   dL/d(ix, iy) = sum_b dh_in(b) . d interp(ZP_b) / d(ix, iy)
// with ZP_b the per-scene projected maps lin_z[b](latent) (the lookup is linear in the latent, api.hip
// ensure_projection), so the lin_z^T GEMMs are not needed.

// sel[ray * kfd + j] = index (ray * kt + position) of depth sample j in the sorted fine depths, or -1 when the sample
// was clamped to [near, far] (no gradient passes the clamp).  Re-creates the forward's draw (explicit or Philox).
__global__ void locate_depth_samples_kernel(const float* __restrict__ rays, const float* __restrict__ depth_c,
                                            const float* __restrict__ g, uint64_t seed, const float* __restrict__ z_fine,
                                            long long n, int kt, int kfd, float depth_std, int* __restrict__ sel) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * kfd) return;
    const long long ray = i / kfd;
    const float near = rays[ray * 8 + 6], far = rays[ray * 8 + 7];
    const float gg = g ? g[i] : normal_at(seed, STREAM_DEPTH, (uint64_t)i);
    const float zz = depth_c[ray] + gg * depth_std;
    const float zc = fmaxf(fminf(zz, far), near);
    const float* zr = z_fine + ray * kt;
    int lo = 0, hi = kt;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (zr[mid] < zc)
            lo = mid + 1;
        else
            hi = mid;
    }
    const bool inside = zz > near && zz < far;
    sel[i] = (inside && lo < kt && zr[lo] == zc) ? (int)(ray * kt + lo) : -1;
}

__global__ __launch_bounds__(256) void mlp_dz_kernel(const DzArgs a) {
    __shared__ float sh[4][HID];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wv;
    if (i >= a.n_sel) return;
    const int idx = a.sel[i];
    if (idx < 0) return;
    const long long loc = (long long)idx - a.p0;
    const long long tile = loc >> 6;
    const int m = (int)(loc & 63);
    const long long ray = idx / a.K;
    const float4* rr = reinterpret_cast<const float4*>(a.rays + ray * 8);
    const float4 r0 = rr[0], r1 = rr[1];
    const float zz = a.z[idx];
    const float d[3] = {r0.w, r1.x, r1.y};
    const float p[3] = {r0.x + zz * d[0], r0.y + zz * d[1], r0.z + zz * d[2]};
    const float* dy_rec = a.dy_stash + tile * a.lay.dy_tile;
    const int ncode = 3 + 6 * a.num_freqs;
    float acc = 0.f;
    // grouped scene (MlpArgs::obj_pts): the sample's object sees views vb .. vb + NS - 1 of the view list (wave-uniform)
    const int vb = a.obj_pts ? __builtin_amdgcn_readfirstlane((int)((long long)idx / a.obj_pts) * a.NS) : 0;
    for (int v = 0; v < a.NS; ++v) {
        const Cam cam = a.cams[vb + v];
        float xr[3], xc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            xr[k] = cam.w2c[4 * k + 0] * p[0] + cam.w2c[4 * k + 1] * p[1] + cam.w2c[4 * k + 2] * p[2];
            xc[k] = xr[k] + cam.w2c[4 * k + 3];
        }
        // ---- positional code: g_in = lin_in^T dh_in(0), then d sin(phase + x f) / dx = cos(.) f
        const float* dh0 = a.nvb > 0 ? dy_rec + (size_t)v * a.lay.dy_view + STASH_SLOT
                                     : dy_rec + a.lay.dy_post + STASH_SMALL + (size_t)(a.npost > 0 ? 2 : 0) * STASH_SLOT;
        {
            const float4 h0 = *reinterpret_cast<const float4*>(dh0 + ((size_t)(2 * lane) * 64 + m) * 4);
            const float4 h1 = *reinterpret_cast<const float4*>(dh0 + ((size_t)(2 * lane + 1) * 64 + m) * 4);
            *reinterpret_cast<float4*>(&sh[wv][8 * lane]) = h0;
            *reinterpret_cast<float4*>(&sh[wv][8 * lane + 4]) = h1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float gin = 0.f;
        if (lane < a.d_in)
            for (int n = 0; n < HID; ++n) gin += a.w_in[(size_t)n * a.d_in + lane] * sh[wv][n];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int dim = -1;
        float c = 0.f;
        if (lane < 3) {
            dim = lane;
            c = gin;
        } else if (lane < ncode) {
            const int e = lane - 3;
            const int fi = e / 6, ph = (e / 3) & 1;
            dim = e % 3;
            const float freq = a.freq_factor * (float)(1 << fi);
            const float arg = (ph ? 1.57079632679489661923f : 0.f) + xr[dim] * freq;
            c = gin * (cosf(arg) * freq);
        }
        float gx[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) gx[k] = bwd_wave_sum(dim == k ? c : 0.f);
        // ---- projection + bilinear lookup of the projected maps
        if (a.nvb > 0 && a.zp) {
            const float sgn = a.yolo ? 1.0f : -1.0f;
            float ux = sgn * xc[0] / xc[2], uy = sgn * xc[1] / xc[2];
            ux = ux * cam.fx + cam.cx;
            uy = uy * cam.fy + cam.cy;
            const float gxn = ux * a.sx - 1.0f, gyn = uy * a.sy - 1.0f;
            const float ix = ((gxn + 1.0f) / 2.0f) * (float)(a.Wl - 1);
            const float iy = ((gyn + 1.0f) / 2.0f) * (float)(a.Hl - 1);
            const float x0 = floorf(ix), y0 = floorf(iy);
            const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
            const float xs[4] = {x0, x1, x0, x1};
            const float ys[4] = {y0, y0, y1, y1};
            // d(bilinear weight of tap k)/d ix and /d iy: nw = (x1-ix)(y1-iy), ne = (ix-x0)(y1-iy), sw, se
            const float wx[4] = {-(y1 - iy), (y1 - iy), -(iy - y0), (iy - y0)};
            const float wy[4] = {-(x1 - ix), -(ix - x0), (x1 - ix), (ix - x0)};
            const bool cull = (a.yolo && !(xc[2] < 0.0f)) || !(ix == ix) || !(iy == iy);
            float six = 0.f, siy = 0.f;
            if (!cull) {
                const float* zv = a.zp + (size_t)(vb + v) * a.Hl * a.Wl * a.zp_stride;
                for (int b = 0; b < a.nvb; ++b) {
                    const float* dhb = dy_rec + (size_t)v * a.lay.dy_view + (size_t)(2 * b + 1) * STASH_SLOT;
                    const float4 h0 = *reinterpret_cast<const float4*>(dhb + ((size_t)(2 * lane) * 64 + m) * 4);
                    const float4 h1 = *reinterpret_cast<const float4*>(dhb + ((size_t)(2 * lane + 1) * 64 + m) * 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool ok = (xs[k] >= 0.f) && (xs[k] <= (float)(a.Wl - 1)) && (ys[k] >= 0.f) && (ys[k] <= (float)(a.Hl - 1));
                        if (!ok) continue;
                        const float* tp = zv + ((size_t)((int)ys[k] * a.Wl + (int)xs[k])) * a.zp_stride + b * HID + 8 * lane;
                        const float4 t0 = *reinterpret_cast<const float4*>(tp);
                        const float4 t1 = *reinterpret_cast<const float4*>(tp + 4);
                        const float dot = h0.x * t0.x + h0.y * t0.y + h0.z * t0.z + h0.w * t0.w + h1.x * t1.x + h1.y * t1.y +
                                          h1.z * t1.z + h1.w * t1.w;
                        six += dot * wx[k];
                        siy += dot * wy[k];
                    }
                }
            }
            six = bwd_wave_sum(six);
            siy = bwd_wave_sum(siy);
            const float dux = six * (a.sx * (float)(a.Wl - 1) * 0.5f), duy = siy * (a.sy * (float)(a.Hl - 1) * 0.5f);
            const float inv = 1.0f / xc[2];
            gx[0] += dux * sgn * cam.fx * inv;
            gx[1] += duy * sgn * cam.fy * inv;
            gx[2] += -(dux * sgn * cam.fx * xc[0] + duy * sgn * cam.fy * xc[1]) * inv * inv;
        }
        // dL/dp = R^T (dL/dxr + dL/dxc);  dL/dz = d . dL/dp
#pragma unroll
        for (int j = 0; j < 3; ++j)
            acc += d[j] * (cam.w2c[j] * gx[0] + cam.w2c[4 + j] * gx[1] + cam.w2c[8 + j] * gx[2]);
    }
    if (lane == 0) a.dz[idx] += acc;
}

// g_depth_out[ray] = g_depth_in[ray] (or 0) + sum over the ray's unclamped depth samples of dL/dz
// One wavefront per ray: lane j takes depth sample j (the loads of a ray's samples are in flight together -- a thread per ray
// walked them one dependent pair after the other, 0.3-0.45 ms on the critical path between a scene's fine and coarse chains), the
// sum is taken in sample order by lane 0 (the order of the serial loop: bit-identical to it).
__global__ __launch_bounds__(256) void depth_grad_gather_kernel(const int* __restrict__ sel, const float* __restrict__ dz,
                                                                const float* __restrict__ g_in, long long n, int kfd,
                                                                float* __restrict__ g_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long ray = (long long)blockIdx.x * 4 + wave;
    if (ray >= n) return;
    float s = g_in ? g_in[ray] : 0.f;
    for (int j0 = 0; j0 < kfd; j0 += 64) {
        float v = 0.f;
        bool live = false;
        if (j0 + lane < kfd) {
            const int idx = sel[ray * kfd + j0 + lane];
            live = idx >= 0;
            if (live) v = dz[idx];
        }
        const unsigned long long mask = __ballot(live);
        const int cnt = kfd - j0 < 64 ? kfd - j0 : 64;
        for (int j = 0; j < cnt; ++j) {   // every lane forms the same sum, in sample order
            const float vj = __shfl(v, j, 64);
            if ((mask >> j) & 1ull) s += vj;
        }
    }
    if (lane == 0) g_out[ray] = s;
}

void launch_locate_depth_samples(const float* rays, const float* depth_c, const float* g, uint64_t seed, const float* z_fine,
                                 long long n, int kt, int kfd, float depth_std, int* sel, hipStream_t st) {
    const long long tot = n * kfd;
    if (tot == 0) return;
    hipLaunchKernelGGL(locate_depth_samples_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, rays, depth_c, g, seed,
                       z_fine, n, kt, kfd, depth_std, sel);
}
void launch_mlp_dz(const DzArgs& a, hipStream_t st) {
    if (a.n_sel <= 0) return;
    hipLaunchKernelGGL(mlp_dz_kernel, dim3((unsigned)((a.n_sel + 3) / 4)), dim3(256), 0, st, a);
}
void launch_depth_grad_gather(const int* sel, const float* dz, const float* g_in, long long n, int kfd, float* g_out, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(depth_grad_gather_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, sel, dz, g_in, n, kfd, g_out);
}

// ---------------------------------------------------------------------------------------------- YOLO aggregation backward
// Reverse of yolo_aggregate_kernel (render_kernels.hip; reference yolo.py:96-114): p_k = sigmoid(o0_k), P = sum_k p_k,
//   out0 = max_k p_k,  out_i = sum_k p_k o_ik / (P + 1e-5), i = 1..6.
//   dL/do_ik = g_i p_k / (P + eps);  dL/dp_k = [k == argmax] g_0 + sum_i g_i (o_ik - out_i) / (P + eps);  dL/do0_k = dL/dp_k p_k (1 - p_k)
// One wavefront per (ray, anchor); the max's gradient goes to the first index that attains it (torch.max(dim)).
__global__ __launch_bounds__(256) void yolo_aggregate_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ g,
                                                                 long long n, int K, int na, float* __restrict__ d_raw) {
    const int lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n * na) return;
    const long long ray = item / na;
    const int a = (int)(item - ray * na);
    float ps = 0.f, pm = -1.f, v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int km = 0x7fffffff;
    for (int k = lane; k < K; k += 64) {
        const float* r = raw + ((ray * K + k) * na + a) * 7;
        const float p = 1.0f / (1.0f + expf(-r[0]));
        ps += p;
        if (p > pm) {
            pm = p;
            km = k;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] += r[1 + i] * p;
    }
    ps = bwd_wave_sum(ps);
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = bwd_wave_sum(v[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {   // (max, first index) reduction
        const float pmo = __shfl_xor(pm, o, 64);
        const int kmo = __shfl_xor(km, o, 64);
        if (pmo > pm || (pmo == pm && kmo < km)) {
            pm = pmo;
            km = kmo;
        }
    }
    const float den = ps + 1e-5f;
    const float* gi = g + item * 7;
    float gv[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) gv[i] = gi[i];
    for (int k = lane; k < K; k += 64) {
        const float* r = raw + ((ray * K + k) * na + a) * 7;
        float* d = d_raw + ((ray * K + k) * na + a) * 7;
        const float p = 1.0f / (1.0f + expf(-r[0]));
        float dp = (k == km) ? gv[0] : 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            d[1 + i] = gv[1 + i] * (p / den);
            dp += gv[1 + i] * ((r[1 + i] - v[i] / den) / den);
        }
        d[0] = dp * (p * (1.0f - p));
    }
}

void launch_yolo_aggregate_bwd(const float* raw, const float* g, long long n, int k, int na, float* d_raw, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(yolo_aggregate_bwd_kernel, dim3((unsigned)((n * na + 3) / 4)), dim3(256), 0, st, raw, g, n, k, na, d_raw);
}

}  // namespace pny
