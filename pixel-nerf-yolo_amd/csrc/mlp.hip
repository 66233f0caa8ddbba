// Fused conditioned-MLP kernel for gfx950 (MI355X): everything PixelNeRFNet.forward does for a
// tile of 64 query samples in ONE launch --
//   world->camera transform, positional encoding, projection, bilinear latent gather
//   (reference src/model/models.py:153-276, src/model/code.py:30-42, src/model/encoder.py:79-108),
//   the ResnetFC chain with per-view latent injection and the cross-view mean
//   (reference src/model/resnetfc.py:134-186), sigmoid / relu head (models.py:312-317).
//
// Design (not a translation of the reference's addmm/relu/add op sequence):
//   * arithmetic is exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): the 1e-4 fp32 parity bar rules
//     out bf16 inputs; the roofline is the 157.3 TFLOP/s fp32 matrix peak.
//   * GEMMs are computed transposed, H^T[n][m] = W[n][k] X^T[k][m]: the weight matrix is the
//     A operand (pre-packed on the host in exact lane order, streamed from L2 with 16-byte
//     loads), the activations are the B operand.  The accumulator then has the sample index on
//     the lane and 4 consecutive features per register quad, which is exactly the 16-byte
//     k-group the next layer's B operand wants -> the epilogue is one ds_write_b128 per quad
//     and no transposes or shuffles exist anywhere in the chain.
//   * 8 waves per workgroup, wave w owns features [64w, 64w+64) of all 64 samples (2x2 tiles of
//     32x32).  The residual stream h never leaves the accumulators (fc_1 accumulates straight
//     into it); only relu(.) inputs travel through a 128 KiB LDS activation buffer.
//   * the bilinear latent gather writes the lin_z B operand straight into that LDS buffer
//     (channel-last latent, 16-byte loads, 128-byte lines per 8 lanes).
//   * one workgroup per CU (LDS bound), persistent over tiles; cross-view running sum lives in
//     a per-workgroup L2-resident scratch slab (128 KiB).
#include "pny_common.h"

namespace pny {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- accumulator <-> feature mapping of v_mfma_f32_32x32x2_f32 -------------------------------
// lane l = 32*hh + m0.  acc[nt][mt] register r holds
//     feature n = 64*wave + 32*nt + 8*(r>>2) + 4*hh + (r&3),  sample m = 32*mt + m0.
// LDS activation buffer: float4 act[kg][m], kg = feature/4, component = feature%4.
// B operand of k-iteration j (8 features): lane reads act[2j + hh][m]; its 4 components feed 4
// successive MFMAs.  A operand: packed so that lane reads float4 #lane of block (nt, j) holding
//     W[32*nt_global + m0][8j + 4hh + 0..3].

__device__ __forceinline__ void gemm_tile(f32x16 (&acc)[2][2], const float4* __restrict__ wp, int jtot, int jn,
                                          const float4* __restrict__ act, int lane) {
    const int m0 = lane & 31, hh = lane >> 5;
    const float4* w0 = wp + lane;                       // n-tile 0 of this wave
    const float4* w1 = wp + (size_t)jtot * 64 + lane;   // n-tile 1
    const float4* bp = act + hh * TM + m0;
    float4 a0 = w0[0], a1 = w1[0];
    for (int j = 0; j < jn; ++j) {
        const int jn1 = (j + 1 < jn) ? j + 1 : j;
        const float4 na0 = w0[(size_t)jn1 * 64];
        const float4 na1 = w1[(size_t)jn1 * 64];
        const float4 b0 = bp[(2 * j) * TM];
        const float4 b1 = bp[(2 * j) * TM + 32];
#define PNY_STEP(c)                                                                          \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b0.c, acc[0][0], 0, 0, 0);        \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b1.c, acc[0][1], 0, 0, 0);        \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b0.c, acc[1][0], 0, 0, 0);        \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b1.c, acc[1][1], 0, 0, 0);
        PNY_STEP(x)
        PNY_STEP(y)
        PNY_STEP(z)
        PNY_STEP(w)
#undef PNY_STEP
        a0 = na0;
        a1 = na1;
    }
}

// acc += bias[n] (broadcast over samples)
__device__ __forceinline__ void add_bias(f32x16 (&acc)[2][2], const float* __restrict__ bias, int wave, int lane) {
    const int hh = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = *reinterpret_cast<const float4*>(bias + 64 * wave + 32 * nt + 8 * q + 4 * hh);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[nt][mt][4 * q + 0] += b.x;
                acc[nt][mt][4 * q + 1] += b.y;
                acc[nt][mt][4 * q + 2] += b.z;
                acc[nt][mt][4 * q + 3] += b.w;
            }
        }
    }
}

__device__ __forceinline__ void set_bias(f32x16 (&acc)[2][2], const float* __restrict__ bias, int wave, int lane) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.f;
    add_bias(acc, bias, wave, lane);
}

// act[feature/4][m] = relu(acc): the next layer's B operand.
__device__ __forceinline__ void store_relu(const f32x16 (&acc)[2][2], float4* __restrict__ act, int wave, int lane) {
    const int m0 = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v;
                v.x = fmaxf(acc[nt][mt][4 * q + 0], 0.f);
                v.y = fmaxf(acc[nt][mt][4 * q + 1], 0.f);
                v.z = fmaxf(acc[nt][mt][4 * q + 2], 0.f);
                v.w = fmaxf(acc[nt][mt][4 * q + 3], 0.f);
                const int kg = 16 * wave + 8 * nt + 2 * q + hh;
                act[kg * TM + 32 * mt + m0] = v;
            }
}

// One pre-activation residual block (reference resnetfc.py:53-62):
//   net = fc_0(relu(h)); h = h + fc_1(relu(net))
__device__ __forceinline__ void res_block(f32x16 (&h)[2][2], const MlpWeights& w, int blk, float4* act, int wave,
                                          int lane) {
    f32x16 net[2][2];
    __syncthreads();
    store_relu(h, act, wave, lane);
    __syncthreads();
    set_bias(net, w.b_fc0[blk], wave, lane);
    gemm_tile(net, reinterpret_cast<const float4*>(w.w_fc0[blk]) + (size_t)(2 * wave) * 64 * 64, 64, 64, act, lane);
    __syncthreads();
    store_relu(net, act, wave, lane);
    __syncthreads();
    add_bias(h, w.b_fc1[blk], wave, lane);
    gemm_tile(h, reinterpret_cast<const float4*>(w.w_fc1[blk]) + (size_t)(2 * wave) * 64 * 64, 64, 64, act, lane);
}

__device__ __forceinline__ void load_point(const MlpArgs& a, long long s, float (&p)[3], float (&d)[3]) {
    if (a.mode == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            p[i] = a.xyz[s * 3 + i];
            d[i] = a.dirs[s * 3 + i];
        }
    } else {
        // points = o + z * d (reference nerf.py:191), view dir = ray dir (nerf.py:210)
        const long long ray = s / a.K;
        const float* r = a.rays + ray * 8;
        const float zz = a.z[s];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            d[i] = r[3 + i];
            p[i] = r[i] + zz * d[i];
        }
    }
}

// Positional-code entry e of the 48-row (42 valid) input column (reference code.py:30-42 layout:
// [x(3), then per frequency sin(f x)(3), sin(f x + pi/2)(3)], then view dirs (models.py:207)).
__device__ __forceinline__ float input_entry(int e, const float (&xr)[3], const float (&vd)[3], float freq_factor,
                                             int num_freqs) {
    const int ncode = 3 + 6 * num_freqs;
    if (e < 3) return xr[e];
    if (e < ncode) {
        const int idx = e - 3;
        const int fi = idx / 6, ph = (idx / 3) & 1, dim = idx % 3;
        const float freq = freq_factor * (float)(1 << fi);
        const float arg = (ph ? 1.57079632679489661923f : 0.f) + xr[dim] * freq;  // fp32 mul, then add
        return sinf(arg);
    }
    if (e < ncode + 3) return vd[e - ncode];
    return 0.f;
}

// Per (view, tile) prologue: B operand of lin_in into act k-groups 0..11, and the four bilinear
// taps of every sample into the tap table.
__device__ __forceinline__ void prologue(const MlpArgs& a, int v, long long tile, float4* act, int* tap_off,
                                         float* tap_w, int tid) {
    const int m = tid & 63, part = tid >> 6;
    long long s = tile * TM + m;
    if (s >= a.n_points) s = a.n_points - 1;
    float p[3], d[3];
    load_point(a, s, p, d);
    const Cam cam = a.cams[v];
    float xr[3], xc[3], vd[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        xr[i] = cam.w2c[4 * i + 0] * p[0] + cam.w2c[4 * i + 1] * p[1] + cam.w2c[4 * i + 2] * p[2];
        xc[i] = xr[i] + cam.w2c[4 * i + 3];
        vd[i] = cam.w2c[4 * i + 0] * d[0] + cam.w2c[4 * i + 1] * d[1] + cam.w2c[4 * i + 2] * d[2];
    }
    for (int g = part; g < D_IN_PAD / 4; g += 8) {
        float4 x4;
        x4.x = input_entry(4 * g + 0, xr, vd, a.freq_factor, a.num_freqs);
        x4.y = input_entry(4 * g + 1, xr, vd, a.freq_factor, a.num_freqs);
        x4.z = input_entry(4 * g + 2, xr, vd, a.freq_factor, a.num_freqs);
        x4.w = input_entry(4 * g + 3, xr, vd, a.freq_factor, a.num_freqs);
        act[g * TM + m] = x4;
    }
    if (part == 7) {
        // projection (reference models.py:219-230) and grid_sample coordinates
        // (encoder.py:97-98, align_corners=True, zeros padding)
        float ux, uy;
        if (!a.yolo) {
            ux = -xc[0] / xc[2];
            uy = -xc[1] / xc[2];
        } else {
            ux = xc[0] / xc[2];
            uy = xc[1] / xc[2];
        }
        ux = ux * cam.fx + cam.cx;
        uy = uy * cam.fy + cam.cy;
        const float gx = ux * a.sx - 1.0f, gy = uy * a.sy - 1.0f;
        const float ix = ((gx + 1.0f) / 2.0f) * (float)(a.Wl - 1);
        const float iy = ((gy + 1.0f) / 2.0f) * (float)(a.Hl - 1);
        const float x0 = floorf(ix), y0 = floorf(iy);
        const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
        float wgt[4] = {(x1 - ix) * (y1 - iy), (ix - x0) * (y1 - iy), (x1 - ix) * (iy - y0), (ix - x0) * (iy - y0)};
        const float xs[4] = {x0, x1, x0, x1};
        const float ys[4] = {y0, y0, y1, y1};
        const bool cull = a.yolo && !(xc[2] < 0.0f);  // models.py:224,254-264: z >= 0 (or NaN) -> zero latent
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = (xs[k] >= 0.f) && (xs[k] <= (float)(a.Wl - 1)) && (ys[k] >= 0.f) && (ys[k] <= (float)(a.Hl - 1));
            int off = 0;
            float wk = wgt[k];
            if (ok) {
                off = ((int)ys[k] * a.Wl + (int)xs[k]) * a.L;
            } else {
                wk = wk * 0.0f;  // out-of-range tap contributes 0 (NaN coordinates stay NaN, as in ATen)
            }
            if (cull || (a.yolo && (wk != wk))) wk = 0.0f;
            tap_off[k * TM + m] = off;
            tap_w[k * TM + m] = wk;
        }
    }
}

// Bilinear gather of latent channels [c0, c0 + 4*nq) of view v for all 64 samples into the
// B-operand layout act[(c - c0)/4][m].  A wave pass covers 8 samples x 8 channel quads: each
// group of lanes {l, l+8, .., l+56} reads one 128-byte line per tap, each 8-lane group writes
// 128 contiguous LDS bytes.
__device__ __forceinline__ void gather_latent(const MlpArgs& a, int v, int c0, int nq, float4* act,
                                              const int* tap_off, const float* tap_w, int wave, int lane) {
    const float* base = a.latent + (size_t)v * a.Hl * a.Wl * a.L + c0;
    const int ml = lane & 7, ql = lane >> 3;
    const int nqb = nq >> 3;
    for (int it = wave; it < 8 * nqb; it += 8) {
        const int m = (it & 7) * 8 + ml;
        const int q = (it >> 3) * 8 + ql;
        float4 acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float wk = tap_w[k * TM + m];
            const float4 t = *reinterpret_cast<const float4*>(base + tap_off[k * TM + m] + 4 * q);
            acc4.x += t.x * wk;
            acc4.y += t.y * wk;
            acc4.z += t.z * wk;
            acc4.w += t.w * wk;
        }
        act[q * TM + m] = acc4;
    }
}

__global__ __launch_bounds__(MLP_THREADS, 2) void pny_mlp_kernel(const MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float4* act = reinterpret_cast<float4*>(smem_raw);
    int* tap_off = reinterpret_cast<int*>(smem_raw + ACT_KG * TM * 16);
    float* tap_w = reinterpret_cast<float*>(smem_raw + ACT_KG * TM * 16 + 1024);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* slab = a.scratch + (size_t)blockIdx.x * (TM * HID) + (size_t)wave * (4 * 16 * 64) + lane;
    const int jz_tot = a.L / 8;

    for (long long tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        f32x16 h[2][2];
        const int n_view_blocks = a.combine_layer < a.n_blocks ? a.combine_layer : a.n_blocks;
        for (int v = 0; v < a.NS; ++v) {
            __syncthreads();
            prologue(a, v, tile, act, tap_off, tap_w, tid);
            __syncthreads();
            set_bias(h, a.w.b_in, wave, lane);
            gemm_tile(h, reinterpret_cast<const float4*>(a.w.w_in) + (size_t)(2 * wave) * (D_IN_PAD / 8) * 64,
                      D_IN_PAD / 8, D_IN_PAD / 8, act, lane);
            for (int blk = 0; blk < n_view_blocks; ++blk) {
                // x = x + lin_z[blk](z)  (reference resnetfc.py:176-182)
                for (int c0 = 0; c0 < a.L; c0 += 4 * ACT_KG) {
                    const int nch = (a.L - c0) < 4 * ACT_KG ? (a.L - c0) : 4 * ACT_KG;
                    __syncthreads();
                    gather_latent(a, v, c0, nch / 4, act, tap_off, tap_w, wave, lane);
                    __syncthreads();
                    gemm_tile(h,
                              reinterpret_cast<const float4*>(a.w.w_z[blk]) + ((size_t)(2 * wave) * jz_tot + c0 / 8) * 64,
                              jz_tot, nch / 8, act, lane);
                }
                add_bias(h, a.w.b_z[blk], wave, lane);
                res_block(h, a.w, blk, act, wave, lane);
            }
            if (a.NS > 1) {
                // running sum over views (reference util.py:489-499 combine_interleaved, mean)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float* p = slab + ((nt * 2 + mt) * 16 + r) * 64;
                            if (v == 0) {
                                *p = h[nt][mt][r];
                            } else if (v + 1 < a.NS) {
                                *p = *p + h[nt][mt][r];
                            } else {
                                h[nt][mt][r] = (*p + h[nt][mt][r]) / (float)a.NS;
                            }
                        }
            }
        }
        for (int blk = n_view_blocks; blk < a.n_blocks; ++blk) res_block(h, a.w, blk, act, wave, lane);

        // out = lin_out(relu(h)) (reference resnetfc.py:185) + output head (models.py:312-317)
        __syncthreads();
        store_relu(h, act, wave, lane);
        __syncthreads();
        for (int idx = tid; idx < a.d_out * TM; idx += MLP_THREADS) {
            const int o = idx >> 6, m = idx & 63;
            const float4* wrow = reinterpret_cast<const float4*>(a.w.w_out + (size_t)o * HID);
            float sum = 0.f;
#pragma unroll 8
            for (int kg = 0; kg < ACT_KG; ++kg) {
                const float4 x = act[kg * TM + m];
                const float4 ww = wrow[kg];
                sum += x.x * ww.x;
                sum += x.y * ww.y;
                sum += x.z * ww.z;
                sum += x.w * ww.w;
            }
            sum += a.w.b_out[o];
            if (!a.yolo) {
                if (o < 3)
                    sum = 1.0f / (1.0f + expf(-sum));
                else if (o == 3)
                    sum = fmaxf(sum, 0.f);
            }
            const long long s = tile * TM + m;
            if (s < a.n_points) a.out[s * a.d_out + o] = sum;
        }
    }
}

int mlp_max_grid() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
    return cus;  // one 128 KiB-LDS workgroup per CU
}

void launch_mlp(const MlpArgs& a, int grid, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pny_mlp_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(pny_mlp_kernel, dim3(grid), dim3(MLP_THREADS), MLP_LDS_BYTES, st, a);
}

}  // namespace pny
