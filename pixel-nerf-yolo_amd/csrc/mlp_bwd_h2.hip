// The backward dX chain on the f16 matrix cores with split fp32 operands (gfx950): pny_mlp_bwd_kernel (mlp_bwd.hip) with the
// matrix products of mlp_h2.hip.  What is differentiated, which tensors go to the dY stash and in which layout is unchanged
// (reference src/model/resnetfc.py:53-62,134-186, src/model/models.py:312-317; mlp_bwd.hip header); what changes:
//   * dX^T[k][m] = W^T[k][n] dY^T[n][m] with the TRANSPOSED weights as split-f16 images (api.hip pack_mlp, PACK_H2T) streaming
//     through the 2-step register ring of mlp_h2_core.h, and dY^T as two f16 planes in the LDS activation buffer
//     ([n / 8][plane][sample] x 16 bytes): x1 w1 + x2 w1 + x1 w2 on v_mfma_f32_32x32x16_f16, fp32 accumulation.
//   * Gradients have no fixed magnitude and f16 has 40 binades: every tile works in a SCALED domain.  sigma = the power of two
//     that puts the tile's largest head gradient (d loss / d lin_out's output) at 2^4..2^5; the chain is linear in dY between
//     the relu masks, so the accumulators simply hold sigma x the true values and every store to the dY stash multiplies by
//     1 / sigma (exact).  Headroom: the chain's values may grow 2^11-fold over the head's before an f16 plane overflows and
//     shrink 2^8-fold before the second plane's absolute 2^-25 costs relative precision (kaiming-scaled weights keep a
//     residual chain within a few binades; scenes pinned to F32 run pny_mlp_bwd_kernel).
//   * The running max |dY| for the weight-gradient GEMM's own scale (pny_dw_gemm_h2_kernel) is taken on the TRUE values.
// Compiled without SLP vectorisation like mlp_h2.hip (csrc/Makefile).
#include <cstring>

#include "mlp_bwd_core.h"
#include "mlp_h2_core.h"

namespace pny {

namespace {
constexpr int BH_LDS = h2::ACT_BYTES + 64;   // activation planes + 8 floats for the tile's max reduction

// acc (scaled domain) -> the two f16 planes of the LDS B operand (TO_LDS) and / or, times inv_sigma, the dY stash (TO_GLOBAL)
template <bool TO_LDS, bool TO_GLOBAL, class Acc>
__device__ __forceinline__ void bh_store(const Acc (&acc)[h2::NT][h2::MT], char* planes, const StashRef& g, float inv_sigma,
                                         int wave, int lane, float& amax) {
    using namespace h2;
    const int m0 = lane & 31, hh = lane >> 5;
    const unsigned lo = acc_lane_off<NT, MT>(wave, lane);
    // accumulator quad (nt, q) of this lane = features 32 NT wave + 32 nt + 8 q + 4 hh + 0..3: row 4 NT wave + 4 nt + q, half hh
    char* base = planes + (4 * NT * wave) * (2 * ROW_BYTES) + m0 * 16 + 8 * hh;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x0 = acc[nt][mt][4 * q + 0], x1 = acc[nt][mt][4 * q + 1], x2 = acc[nt][mt][4 * q + 2], x3 = acc[nt][mt][4 * q + 3];
                if (TO_LDS) {
                    h4 p0, p1;
                    split4(x0, x1, x2, x3, p0, p1);
                    char* s0 = base + (4 * nt + q) * (2 * ROW_BYTES) + 32 * mt * 16;
                    *reinterpret_cast<h4*>(s0) = p0;
                    *reinterpret_cast<h4*>(s0 + ROW_BYTES) = p1;
                }
                if (TO_GLOBAL) {
                    const float4 v = make_float4(x0 * inv_sigma, x1 * inv_sigma, x2 * inv_sigma, x3 * inv_sigma);
                    stash_st(g, lo, quad_off<NT, MT>(nt, mt, q), v);
                    amax = fmaxf(fmaxf(amax, fabsf(v.x)), fabsf(v.y));
                    amax = fmaxf(fmaxf(amax, fabsf(v.z)), fabsf(v.w));
                }
            }
}

// Reverse of one pre-activation residual block in the scaled domain (block_bwd of mlp_bwd.hip):
//   dnet = (fc_1^T dh') * [net > 0];   dh = (dh' + (fc_0^T dnet) * [h > 0]) * scale
// dh: the gradient of the residual stream as plain floats (it is never an MFMA accumulator inside a block: as f32x16 tiles the
// compiler kept the 16-register tuples in scratch and re-spilled a whole tuple after every element-wise update)
typedef float bh_acc[16];
__device__ __forceinline__ void bh_block(bh_acc (&dh)[h2::NT][h2::MT], H2Ring& ring, const WStream& ws, const H2Seg& s_fc1t,
                                         const H2Seg& s_fc0t, const H2Seg& after, char* planes, const StashRef& x_h,
                                         const StashRef& x_net, const StashRef& dy_dnet, const StashRef& dy_dh, float scale,
                                         float inv_sigma, int wave, int lane, float& amax) {
    using namespace h2;
    f32x16 t[NT][MT];
    __syncthreads();  // every wave is done reading the planes (previous GEMM)
    bh_store<true, false>(dh, planes, dy_dh, inv_sigma, wave, lane, amax);
    __syncthreads();
    h2zero<NT, MT>(t);
    h2gemm(t, ring, ws, s_fc1t, s_fc0t, planes, lane);
    mask_by<NT, MT>(t, x_net, wave, lane);
    __syncthreads();
    bh_store<true, true>(t, planes, dy_dnet, inv_sigma, wave, lane, amax);
    __syncthreads();
    h2zero<NT, MT>(t);
    h2gemm(t, ring, ws, s_fc0t, after, planes, lane);
    mask_by<NT, MT>(t, x_h, wave, lane);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dh[nt][mt][r] = (dh[nt][mt][r] + t[nt][mt][r]) * scale;
    bh_store<false, true>(dh, nullptr, dy_dh, inv_sigma, wave, lane, amax);
}
}  // namespace

__global__ __launch_bounds__(h2::THREADS, 2) void pny_mlp_bwd_h2_kernel(const BwdArgs a) {
    using namespace h2;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* planes = smem_raw;
    float* red = reinterpret_cast<float*>(smem_raw + ACT_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = a.n_blocks;
    const int nvb = a.combine_layer < nb ? a.combine_layer : nb;
    const int npost = nb - nvb;
    const WStream ws = wstream_raw(a.w_base, a.w_bytes, lane);
    const H2Seg s_out = h2seg(ws, a.h2T_out, D_IN_PAD / 16, wave);
    auto fc1t = [&](int b) { return h2seg(ws, a.h2T_fc1[b], HID / 16, wave); };
    auto fc0t = [&](int b) { return h2seg(ws, a.h2T_fc0[b], HID / 16, wave); };
    H2Ring ring;
    h2ring_fill(ring, ws, s_out);
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

        // ---- head: gradient w.r.t. lin_out's output through sigmoid / relu (reference models.py:312-317): the dY of lin_out
        // (true values, to the stash) and, scaled by the tile's sigma, the B operand of lin_out^T (d_out rows padded to 64)
        constexpr int NQ = (D_IN_PAD / 4) * TM / THREADS;   // head quads per thread (2)
        static_assert((D_IN_PAD / 4) * TM % THREADS == 0, "head mapping");
        float4 hv[NQ];
        float hmax = 0.f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int idx = tid + i * THREADS;
            const int kg = idx / TM, m = idx % TM;
            const long long s = tile * TM + m;
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
            hv[i] = make_float4(v[0], v[1], v[2], v[3]);
            dy_draw[idx] = hv[i];   // lin_out's dY
            hmax = fmaxf(fmaxf(hmax, fabsf(v[0])), fabsf(v[1]));
            hmax = fmaxf(fmaxf(hmax, fabsf(v[2])), fabsf(v[3]));
        }
        amax = fmaxf(amax, hmax);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) hmax = fmaxf(hmax, __shfl_xor(hmax, o, 64));
        __syncthreads();   // previous tile: every wave is done with `red` and with the planes
        if (lane == 0) red[wave] = hmax;
        __syncthreads();
        float tmax = red[0];
#pragma unroll
        for (int w = 1; w < THREADS / 64; ++w) tmax = fmaxf(tmax, red[w]);
        float sigma = 1.0f, inv_sigma = 1.0f;
        {
            const unsigned mb = __float_as_uint(tmax);
            const int e = (int)((mb >> 23) & 0xffu) - 127;
            if (mb != 0u && e > -100 && e < 100) {
                sigma = __uint_as_float((unsigned)(127 + 4 - e) << 23);
                inv_sigma = __uint_as_float((unsigned)(127 - 4 + e) << 23);
            }
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int idx = tid + i * THREADS;
            const int kg = idx / TM, m = idx % TM;
            h4 p0, p1;
            split4(hv[i].x * sigma, hv[i].y * sigma, hv[i].z * sigma, hv[i].w * sigma, p0, p1);
            char* s0 = planes + (kg >> 1) * (2 * ROW_BYTES) + m * 16 + 8 * (kg & 1);
            *reinterpret_cast<h4*>(s0) = p0;
            *reinterpret_cast<h4*>(s0 + ROW_BYTES) = p1;
        }
        __syncthreads();
        bh_acc dh[NT][MT];
        {
            f32x16 t[NT][MT];
            h2zero<NT, MT>(t);
            h2gemm(t, ring, ws, s_out, fc1t(nb - 1), planes, lane);
            mask_by<NT, MT>(t, x_post(2 * npost), wave, lane);                           // relu(h_top) > 0
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dh[nt][mt][r] = t[nt][mt][r];
        }
        bh_store<false, true>(dh, nullptr, dy_post(0), inv_sigma, wave, lane, amax);     // dh_top: dY of the last block's fc_1

        // ---- post-combine blocks, last to first; the first of them also applies the 1/NS of the cross-view mean
        for (int b = nb - 1; b >= nvb; --b) {
            const int i = b - nvb;
            const H2Seg after = b > nvb ? fc1t(b - 1) : (nvb > 0 ? fc1t(nvb - 1) : s_out);
            bh_block(dh, ring, ws, fc1t(b), fc0t(b), after, planes, x_post(2 * i), x_post(2 * i + 1), dy_post(1 + 2 * i),
                     dy_post(2 + 2 * i), b == nvb ? inv_ns : 1.0f, inv_sigma, wave, lane, amax);
        }
        // dhm: what every view's last per-view block receives (dh_top itself when there is no post-combine block)
        const StashRef dhm = npost > 0 ? dy_post(2) : dy_post(0);
        for (int v = 0; v < a.NS && nvb > 0; ++v) {
            if (v > 0) {   // the stash holds true values
                acc_load<NT, MT>(dh, dhm, wave, lane);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) dh[nt][mt][r] = dh[nt][mt][r] * sigma;
            }
            for (int b = nvb - 1; b >= 0; --b) {
                const H2Seg after = b > 0 ? fc1t(b - 1) : (v + 1 < a.NS ? fc1t(nvb - 1) : s_out);
                bh_block(dh, ring, ws, fc1t(b), fc0t(b), after, planes, x_act(v, 2 * b), x_act(v, 2 * b + 1), dy_view(v, 2 * b),
                         dy_view(v, 2 * b + 1), 1.0f, inv_sigma, wave, lane, amax);
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

void launch_mlp_bwd_h2(const BwdArgs& a, int grid, hipStream_t st) {
    static bool attr_set[64] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    dev_ &= 63;
    if (!attr_set[dev_]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pny_mlp_bwd_h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BH_LDS);
        attr_set[dev_] = true;
    }
    hipLaunchKernelGGL(pny_mlp_bwd_h2_kernel, dim3(grid), dim3(h2::THREADS), BH_LDS, st, a);
}

}  // namespace pny
