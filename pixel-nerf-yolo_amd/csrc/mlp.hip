// Fused conditioned-MLP kernel for gfx950 (MI355X): everything PixelNeRFNet.forward does for a
// tile of query samples in ONE launch --
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
//   * a workgroup owns a tile of samples x all 512 features; each wave owns a feature slice of all
//     samples.  The residual stream h never leaves the accumulators (fc_1 accumulates straight
//     into it); only relu(.) inputs travel through the LDS activation buffer.
//   * the weights of all layers are one continuous stream per wave behind a static-slot register
//     ring that runs across layer boundaries (see gemm_run).
//   * the bilinear latent gather writes the lin_z B operand straight into the LDS buffer
//     (channel-last latent, 16-byte loads, 128-byte lines per 8 lanes), chunk-pipelined with the GEMM.
//   * persistent over tiles; the cross-view running sum lives in a per-workgroup slab accessed
//     non-temporally.
//   * two instantiations: ZP = false runs lin_z per sample as written in the reference; ZP = true
//     uses the per-scene projected latent (lin_z applied to every latent pixel once, api.hip
//     ensure_projection): bilinear interpolation and lin_z are both linear, so
//     lin_z(interp(latent)) = interp(lin_z(latent)) up to fp32 rounding, and the per-sample lin_z
//     GEMMs (29 % of the FLOPs at L = 512, 64 % at L = 1792) become a 512-channel gather that is
//     added to the residual stream in the epilogue that follows it anyway.
#include <cstdlib>
#include <cstring>

#include "mlp_core.h"

namespace pny {

// STASH (training, ZP = false only): every GEMM's B operand is also written to the activation stash of the tile
// (pny_common.h StashLayout) for the backward pass; outputs and arithmetic are those of the plain instantiation.
template <class C, bool ZP, bool STASH = false>
__global__ __launch_bounds__(C::THREADS, C::WPS) void pny_mlp_kernel(const MlpArgs a) {
    static_assert(!(STASH && ZP), "the stash variant runs the reference operation order");
    constexpr int NT = C::NT, MT = C::MT, TMc = C::TM;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float4* act = reinterpret_cast<float4*>(smem_raw);
    float4* tap_tab = reinterpret_cast<float4*>(smem_raw + ACT_KG * TMc * 16);  // 32 bytes per sample

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* slab = a.scratch + (size_t)blockIdx.x * (TMc * HID) + (size_t)wave * (NT * MT * 16 * 64) + 4 * lane;
    const int jz_tot = a.L / 8;
#ifdef PNY_STAMP
    StampCtx st;
    st.tr_n = 0;
    st.wave = wave;
    st.lane = lane;
    for (int i = 0; i < ST_N; ++i) st.acc[i] = 0;
    const unsigned long long st_start = stamp_now();
#endif

    const int n_view_blocks = a.combine_layer < a.n_blocks ? a.combine_layer : a.n_blocks;
    // weight segments of the stream (this wave's slices), in execution order:
    //   per view: lin_in, then per view-block: lin_z chunks (ZP: none), fc_0, fc_1; then the post-combine blocks
    const WStream ws = wstream(a, lane);
    const WSeg s_in = wseg<NT>(ws, a.w.w_in, D_IN_PAD / 8, 0, D_IN_PAD / 8, wave);
    auto zseg = [&](int blk, int c0) { return wseg<NT>(ws, a.w.w_z[blk], jz_tot, c0 / 8, GCH / 8, wave); };
    auto fc0seg = [&](int blk) { return wseg<NT>(ws, a.w.w_fc0[blk], 64, 0, 64, wave); };
    // segment that follows the per-view part of view v (after its last view-block)
    auto after_view = [&](int v) {
        if (v + 1 < a.NS) return s_in;
        if (n_view_blocks < a.n_blocks) return fc0seg(n_view_blocks);
        return s_in;  // next tile
    };
    WRing<C::WDEPTH, NT> ring;
    ring_fill(ring, ws, s_in);
#ifdef PNY_EXP_FOOT
    ring.exp_base = s_in.off;
    ring.exp_ctr = 0;
#endif

    // XCD-aware tile order: workgroups are dispatched round-robin over the 8 XCDs (XCD = blockIdx % 8), each with
    // its own L2.  XCD x walks the contiguous tile range [x*chunk, (x+1)*chunk): the 32 workgroups of an XCD then
    // work on neighbouring rays at any moment, whose samples project onto overlapping pixels of the (projected)
    // latent, instead of on every 8th ray.
#ifndef PNY_NO_XCD_ORDER
    const bool xcd_order = (gridDim.x & 7) == 0;
#else
    const bool xcd_order = false;
#endif
    const long long t_chunk = xcd_order ? (a.n_tiles + 7) / 8 : a.n_tiles;
    const long long t_first = xcd_order ? (long long)(blockIdx.x & 7) * t_chunk + (blockIdx.x >> 3) : blockIdx.x;
    const long long t_last = xcd_order ? ((long long)((blockIdx.x & 7) + 1) * t_chunk < a.n_tiles
                                              ? (long long)((blockIdx.x & 7) + 1) * t_chunk : (long long)a.n_tiles)
                                       : (long long)a.n_tiles;
    const int t_step = xcd_order ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    for (long long tile = t_first; tile < t_last; tile += t_step) {
        f32x16 h[NT][MT];
        float* const x_rec = STASH ? a.stash_x + tile * a.lay.x_tile : nullptr;
        const int vb = tile_view_base(a, tile * C::TM);   // grouped scene: the tile's object sees views vb .. vb + NS - 1
        for (int v = 0; v < a.NS; ++v) {
            float* const x_view = STASH ? x_rec + (size_t)v * a.lay.x_view : nullptr;
            auto act_slot = [&](int i) { return reinterpret_cast<float4*>(x_view + a.lay.x_act + (size_t)i * STASH_SLOT); };
            {
                BiasRegs<NT> bias;
                ST_BEGIN();
                bias_load<NT>(bias, a.w.b_in, wave, lane);  // b_in + b_z[0] (folded on the host)
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                prologue<C>(a, v, tile, act, tap_tab, tid);
                __syncthreads();
                if constexpr (STASH) {  // lin_in's B operand: act rows 0..15
                    float4* g = reinterpret_cast<float4*>(x_view + a.lay.x_in);
                    for (int i = tid; i < (D_IN_PAD / 4) * TMc; i += C::THREADS) g[i] = act[i];
                }
                bias_apply<NT, MT, false>(h, bias);
                ST_END(ST_PROLOGUE);
            }
            {
                ST_BEGIN();
                gemm_run<C>(h, ring, ws, s_in, n_view_blocks > 0 ? (ZP ? fc0seg(0) : zseg(0, 0)) : after_view(v), act, lane);
                ST_END(ST_GEMM);
            }
            for (int blk = 0; blk < n_view_blocks; ++blk) {
                const bool last = (blk == n_view_blocks - 1);
                if constexpr (ZP) {
                    // x = x + interp(lin_z[blk](latent map)): the 512 projected channels of this block are
                    // gathered into the whole LDS buffer (4 chunks, two in flight) and added to h by the
                    // block entry below (store_relu_addz), which needs the barrier that follows anyway.
                    constexpr int WIN = (GCH / 4) * TMc;
                    GatherTaps<C, 2> g;
                    ST_BEGIN();
                    gather_setup<C>(g, a.zp + (size_t)(vb + v) * a.Hl * a.Wl * a.zp_stride + blk * HID, tap_tab, wave, lane);
                    gather_issue<C, 0>(g, 0, wave);
                    gather_issue<C, 1>(g, GCH, wave);
                    __builtin_amdgcn_sched_barrier(0);
                    __syncthreads();  // every wave is done reading the buffer (previous GEMM)
                    gather_commit<C, 0, true>(g, act, wave, lane);
                    gather_issue<C, 0>(g, 2 * GCH, wave);
                    gather_commit<C, 1, true>(g, act + WIN, wave, lane);
                    gather_issue<C, 1>(g, 3 * GCH, wave);
                    gather_commit<C, 0, true>(g, act + 2 * WIN, wave, lane);
                    gather_commit<C, 1, true>(g, act + 3 * WIN, wave, lane);
                    ST_END(ST_GATHER);
                    res_block<C, true>(h, ring, ws, a.w, blk, last ? after_view(v) : fc0seg(blk + 1), act, wave, lane,
                                       (last && v > 0) ? slab : nullptr ST_PASS);
                } else {
                // x = x + lin_z[blk](z)  (reference resnetfc.py:176-182); bias folded upstream.
                // GCH-channel chunks: the taps of chunk c+1 load while the MFMAs of chunk c run; chunk c
                // lives in LDS window c % 4 (a window is rewritten 4 chunks = 3 barriers later).
                {
                    GatherTaps<C> g;
                    {
                        ST_BEGIN();
                        gather_setup<C>(g, a.latent + (size_t)(vb + v) * a.Hl * a.Wl * a.L, tap_tab, wave, lane);
                        gather_issue<C>(g, 0, wave);
                        __builtin_amdgcn_sched_barrier(0);
                        __syncthreads();  // every wave is done reading the buffer (previous GEMM)
                        ST_END(ST_GATHER);
                    }
                    for (int c0 = 0; c0 < a.L; c0 += GCH) {
                        float4* win = act + (size_t)((c0 / GCH) & 3) * (GCH / 4) * TMc;
                        const bool more = c0 + GCH < a.L;
                        {
                            ST_BEGIN();
                            gather_commit<C>(g, win, wave, lane);
                            if (more) gather_issue<C>(g, c0 + GCH, wave);
                            __builtin_amdgcn_sched_barrier(0);
                            __syncthreads();  // chunk visible to all waves
                            if constexpr (STASH) {  // lin_z's B operand (same for every view block): stash it once
                                if (blk == 0) {
                                    float4* g = reinterpret_cast<float4*>(x_view + a.lay.x_z) + (size_t)(c0 / 4) * TMc;
                                    for (int i = tid; i < (GCH / 4) * TMc; i += C::THREADS) g[i] = win[i];
                                }
                            }
                            ST_END(ST_GATHER);
                        }
                        ST_BEGIN();
                        gemm_run<C>(h, ring, ws, zseg(blk, c0), more ? zseg(blk, c0 + GCH) : fc0seg(blk), win, lane);
                        ST_END(ST_GEMM);
                    }
                }
                // the last per-view block also folds in the running sum over the views done so far
                // (reference util.py:489-499 combine_interleaved, mean over the NS views)
                if constexpr (STASH)
                    res_block<C, false, true>(h, ring, ws, a.w, blk, last ? after_view(v) : zseg(blk + 1, 0), act, wave,
                                              lane, (last && v > 0) ? slab : nullptr ST_PASS, act_slot(2 * blk),
                                              act_slot(2 * blk + 1));
                else
                res_block<C, false>(h, ring, ws, a.w, blk, last ? after_view(v) : zseg(blk + 1, 0), act, wave, lane,
                                    (last && v > 0) ? slab : nullptr ST_PASS);
                }
            }
            if (a.NS > 1) {
                ST_BEGIN();
                if (n_view_blocks == 0 && v > 0) {  // degenerate combine_layer = 0: no block to hide the fetch under
                    f32x16 t[NT][MT];
                    slab_load<NT, MT>(t, slab);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) h[nt][mt][r] = t[nt][mt][r] + h[nt][mt][r];
                }
                if (v + 1 < a.NS) {
                    slab_store<NT, MT>(h, slab);
                } else {
                    const float ns = (float)a.NS;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) h[nt][mt][r] = h[nt][mt][r] / ns;
                }
                ST_END(ST_HSUM);
            }
        }
        auto post_slot = [&](int i) { return reinterpret_cast<float4*>(x_rec + a.lay.x_post + (size_t)i * STASH_SLOT); };
        for (int blk = n_view_blocks; blk < a.n_blocks; ++blk) {
            if constexpr (STASH)
                res_block<C, false, true>(h, ring, ws, a.w, blk, blk + 1 < a.n_blocks ? fc0seg(blk + 1) : s_in, act, wave,
                                          lane, nullptr ST_PASS, post_slot(2 * (blk - n_view_blocks)),
                                          post_slot(2 * (blk - n_view_blocks) + 1));
            else
            res_block<C, false>(h, ring, ws, a.w, blk, blk + 1 < a.n_blocks ? fc0seg(blk + 1) : s_in, act, wave, lane,
                                nullptr ST_PASS);
        }

        // out = lin_out(relu(h)) (reference resnetfc.py:185) + output head (models.py:312-317)
        ST_BEGIN();
        __syncthreads();
        store_relu<NT, MT, STASH>(h, act, wave, lane, STASH ? post_slot(2 * (a.n_blocks - n_view_blocks)) : nullptr);
        __syncthreads();
        for (int idx = tid; idx < a.d_out * TMc; idx += C::THREADS) {
            const int o = idx / TMc, m = idx % TMc;
            const float4* wrow = reinterpret_cast<const float4*>(a.w.w_out + (size_t)o * HID);
            float sum = 0.f;
#pragma unroll 8
            for (int kg = 0; kg < ACT_KG; ++kg) {
                const float4 x = act[kg * TMc + m];
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
            const long long s = tile * TMc + m;
            if (s < a.n_points) a.out[s * a.d_out + o] = sum;
        }
        ST_END(ST_LINOUT);
    }
#ifdef PNY_STAMP
    st.acc[ST_TOTAL] = stamp_now() - st_start;
    if (lane == 0)
        for (int i = 0; i < ST_N; ++i) g_stamp_buf[((size_t)blockIdx.x * C::NW + wave) * ST_N + i] = st.acc[i];
#endif
}

// ------------------------------------------------------------------------------------- host side
int mlp_cu_count() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
    return cus;
}

// Kernel shape per launch.  PNYOLO_MLP_VARIANT forces "8x64", "16x64" or "8x32" (see Cfg); otherwise 8x64
// (fastest in steady state) unless the 32-sample shape finishes the launch sooner: launches that cannot give
// every CU a 64-sample tile (training / visualisation batches) or that end in a mostly empty last round.
static int forced_variant() {
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("PNYOLO_MLP_VARIANT");
        v = -1;
        if (e && !strcmp(e, "8x64")) v = MLP_8x64;
        if (e && !strcmp(e, "16x64")) v = MLP_16x64;
        if (e && !strcmp(e, "8x32")) v = MLP_8x32;
    }
    return v;
}
int mlp_pick_variant(long long n_points) {
    const int f = forced_variant();
    if (f >= 0) return f;
    // estimated duration in units of one 64-sample tile on a CU of its own (measured: a lone 32-sample tile 0.55,
    // two 32-sample workgroups sharing a CU 1.05 -- the steady-state handicap of the 8x32 shape)
    const long long cus = mlp_cu_count();
    const long long n64 = (n_points + 63) / 64, n32 = (n_points + 31) / 32;
    const double t64 = (double)((n64 + cus - 1) / cus);
    const long long full = n32 / (2 * cus), rem = n32 % (2 * cus);
    const double t32 = 1.05 * (double)full + (rem == 0 ? 0.0 : (rem <= cus ? 0.55 : 1.05));
    return t32 < t64 ? MLP_8x32 : MLP_8x64;
}
int mlp_tile_samples(int variant) { return variant == MLP_8x32 ? 32 : 64; }
int mlp_max_grid(int variant) { return mlp_cu_count() * (variant == MLP_8x32 ? 2 : 1); }
size_t mlp_scratch_floats() { return (size_t)mlp_cu_count() * 64 * HID; }  // same for every shape

template <class C, bool ZP>
static void launch_mlp_t(const MlpArgs& a, int grid, hipStream_t st) {
    static bool attr_set[64] = {};  // per device: function attributes are per device
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    dev_ &= 63;
    if (!attr_set[dev_]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pny_mlp_kernel<C, ZP>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        attr_set[dev_] = true;
    }
#ifdef PNY_STAMP
    static unsigned long long* dbuf = nullptr;
    const size_t nst = (size_t)grid * C::NW * ST_N;
    if (!dbuf) {
        (void)hipMalloc((void**)&dbuf, (size_t)1024 * 16 * ST_N * sizeof(unsigned long long));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &dbuf, sizeof(dbuf));
    }
    static unsigned long long* tbuf = nullptr;
    if (!tbuf) {
        (void)hipMalloc((void**)&tbuf, (size_t)TRACE_WAVES * TRACE_N * sizeof(unsigned long long));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &tbuf, sizeof(tbuf));
    }
    (void)hipMemsetAsync(tbuf, 0, (size_t)TRACE_WAVES * TRACE_N * sizeof(unsigned long long), st);
    (void)hipMemsetAsync(dbuf, 0, nst * sizeof(unsigned long long), st);
#endif
    hipLaunchKernelGGL((pny_mlp_kernel<C, ZP>), dim3(grid), dim3(C::THREADS), C::LDS, st, a);
#ifdef PNY_STAMP
    {
        std::vector<unsigned long long> hst(nst);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hst.data(), dbuf, nst * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sum[ST_N] = {0};
        for (size_t i = 0; i < nst; ++i) sum[i % ST_N] += (double)hst[i];
        static const char* names[ST_N] = {"total", "gemm", "gather", "prologue", "store+sync", "hsum", "lin_out",
                                          "(sync1", "write", "sync2)"};
        fprintf(stderr, "[pny stamp] waves=%d tile=%d tiles=%d grid=%d:", C::NW, C::TM, a.n_tiles, grid);
        for (int i = 0; i < ST_N; ++i) fprintf(stderr, " %s=%.1f%%", names[i], 100.0 * sum[i] / sum[0]);
        fprintf(stderr, " (mean wave cycles %.3g)\n", sum[0] / ((double)grid * C::NW));
        if (const char* tf = getenv("PNYOLO_TRACE_FILE")) {
            std::vector<unsigned long long> tr((size_t)TRACE_WAVES * TRACE_N);
            (void)hipMemcpy(tr.data(), tbuf, tr.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            if (FILE* f = fopen(tf, "a")) {
                for (int w = 0; w < C::NW; ++w) {
                    for (int e = 0; e < TRACE_N; ++e) fprintf(f, "%llu ", tr[(size_t)w * TRACE_N + e]);
                    fprintf(f, "\n");
                }
                fprintf(f, "#\n");
                fclose(f);
            }
        }
    }
#endif
}

template <class C>
static void launch_mlp_c(const MlpArgs& a, int grid, hipStream_t st) {
    if (a.zp)
        launch_mlp_t<C, true>(a, grid, st);
    else
        launch_mlp_t<C, false>(a, grid, st);
}

void launch_mlp_stash(const MlpArgs& a, int grid, hipStream_t st) {
    using C = Cfg<2, 2>;
    static bool attr_set[64] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    dev_ &= 63;
    if (!attr_set[dev_]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pny_mlp_kernel<C, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        attr_set[dev_] = true;
    }
    hipLaunchKernelGGL((pny_mlp_kernel<C, false, true>), dim3(grid), dim3(C::THREADS), C::LDS, st, a);
}

void launch_mlp(const MlpArgs& a, int variant, int grid, hipStream_t st) {
    switch (variant) {
        case MLP_16x64: launch_mlp_c<Cfg<1, 2>>(a, grid, st); break;
        case MLP_8x32: launch_mlp_c<Cfg<2, 1>>(a, grid, st); break;
        default: launch_mlp_c<Cfg<2, 2>>(a, grid, st); break;
    }
}

}  // namespace pny
