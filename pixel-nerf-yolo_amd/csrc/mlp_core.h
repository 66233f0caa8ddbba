// Device-side building blocks of the fused MLP kernels (gfx950): kernel shapes, the weight-stream ring, the
// transposed-GEMM step, epilogues, the prologue (positional code + projection) and the bilinear gather.
// Shared by the forward kernel (mlp.hip) and the backward chain (mlp_bwd.hip).  See mlp.hip's header comment
// for the design.
#pragma once
#include "pny_common.h"
#ifdef PNY_STAMP
#include <cstdio>
#include <vector>
#endif

namespace pny {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Kernel shape: a workgroup owns a tile of TM = 32*MT samples x 512 features; wave w owns the NT
// 32-feature n-tiles [NT*w, NT*w + NT) of all TM samples (NT x MT accumulator tiles of 32x32).
//   <NT=2, MT=2>:  8 waves, 256 VGPRs, 2 waves/SIMD, 64-sample tile, one workgroup per CU
//   <NT=1, MT=2>: 16 waves, 128 VGPRs, 4 waves/SIMD, 64-sample tile, one workgroup per CU
//   <NT=2, MT=1>:  8 waves, 128 VGPRs, 4 waves/SIMD, 32-sample tile, two workgroups per CU
//   (<NT=4, MT=1>: 4 waves x 256 VGPRs, two workgroups per CU without the 128-register squeeze, was measured too:
//    -2.7 % against <2,2> -- two workgroups per CU double the weight stream per sample, which costs more than the
//    overlap of their non-GEMM phases gains)
#ifndef PNY_WDEPTH
#define PNY_WDEPTH 4  // 8 re-measured with the low-spill build: -1.1 %
#endif
#ifndef PNY_WDEPTH32
#define PNY_WDEPTH32 4  // ring depth of the 8x32 shape (2 measured: -1.3 %)
#endif
template <int NT_, int MT_>
struct Cfg {
    static constexpr int NT = NT_, MT = MT_;
    static constexpr int TM = 32 * MT;       // samples (GEMM columns) per workgroup tile
    static constexpr int NW = 16 / NT;       // waves per workgroup
    static constexpr int THREADS = 64 * NW;
    static constexpr int WPS = (NT * MT == 4) ? 2 : 4;           // resident waves per SIMD (VGPR budget 512 / WPS)
    static constexpr int WDEPTH = (NT == 2 && MT == 1) ? PNY_WDEPTH32 : PNY_WDEPTH;  // weight-ring depth (k-iterations)
    static constexpr int LDS = ACT_KG * TM * 16 + 32 * TM;       // activations + tap table
};

// Diagnostic build only (-DPNY_STAMP, tools/stamp_build.sh): s_memtime brackets around the phases
// of a tile, summed per wave and dumped by launch_mlp, plus a raw event trace of workgroup 0.
// No stamp executes in the product build.
#ifdef PNY_STAMP
enum { ST_TOTAL = 0, ST_GEMM, ST_GATHER, ST_PROLOGUE, ST_STORE, ST_HSUM, ST_LINOUT, ST_SYNC1, ST_WRITE, ST_SYNC2, ST_N };
__device__ unsigned long long* g_stamp_buf;
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
constexpr int TRACE_N = 256;
constexpr int TRACE_WAVES = 16;
__device__ unsigned long long* g_trace_buf;
struct StampCtx {
    unsigned long long acc[ST_N];
    int tr_n;
    int wave, lane;
};
__device__ __forceinline__ void trace_ev(StampCtx& c) {
    if (blockIdx.x == 0 && c.tr_n < TRACE_N) {
        const unsigned long long t = stamp_now();
        if (c.lane == 0) g_trace_buf[c.wave * TRACE_N + c.tr_n] = t;
    }
    ++c.tr_n;
}
#define ST_BEGIN() const unsigned long long st_t0_ = stamp_now()
#define ST_END(cat) st.acc[cat] += stamp_now() - st_t0_
#define ST_ARG , StampCtx& st
#define ST_PASS , st
#define TRACE() trace_ev(st)
#else
#define ST_BEGIN()
#define ST_END(cat)
#define ST_ARG
#define ST_PASS
#define TRACE()
#endif

// ---- accumulator <-> feature mapping of v_mfma_f32_32x32x2_f32 -------------------------------
// lane l = 32*hh + m0.  acc[nt][mt] register r holds
//     feature n = 32*NT*wave + 32*nt + 8*(r>>2) + 4*hh + (r&3),  sample m = 32*mt + m0.
// LDS activation buffer: float4 act[kg][m], kg = feature/4, component = feature%4.
// B operand of k-iteration j (8 features): lane reads act[2j + hh][m]; its 4 components feed 4
// successive MFMAs.  A operand: packed so that lane reads float4 #lane of block (nt, j) holding
//     W[32*nt_global + m0][8j + 4hh + 0..3].

// One k-iteration (8 input features = 4 MFMA k-steps) of the wave's NT x MT tile.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT, int MT>
__device__ __forceinline__ void mfma_iter(f32x16 (&acc)[NT][MT], const f32x4 (&a)[NT], const float4 (&b)[MT]) {
#define PNY_STEP(c)                                                                                   \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                 \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                             \
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nt].c, b[mt].c, acc[nt][mt], 0, 0, 0);
    PNY_STEP(x)
    PNY_STEP(y)
    PNY_STEP(z)
    PNY_STEP(w)
#undef PNY_STEP
}

// The weights of all layers form ONE stream per wave: 13.7 MB per MLP cannot stay in the XCD's
// 4 MiB L2, and the 32 CUs of an XCD reach a layer together, so the first touch of every line is an
// Infinity-Cache access for everybody.  A register ring keeps the next WDEPTH k-iterations of
// fragments in flight and runs ACROSS layer boundaries: while a GEMM drains, the ring already fills
// with the head of the next layer's slice (WSeg next), so neither the epilogue/barrier phase nor the
// head of a GEMM waits on memory.
// Packed layout (api.hip pack_layer): k-iteration-major, [j][n-tile 0..15][lane] float4.
// The packed weights of both MLPs are ONE allocation, addressed through a raw buffer resource: a fragment
// load is buffer_load_dwordx4 with the resource in SGPRs, a loop-invariant lane offset (16*lane) in one
// VGPR and the fragment's byte offset in an SGPR -- the per-iteration address arithmetic is SALU only.
// (With per-lane 64-bit pointers it was ~6 VALU instructions per k-iteration, and VALU issue takes cycles
// from the fp32 MFMA pipe, tools/ubench/mfma_valu_coexec.hip.)  The wave index comes through
// readfirstlane so that segment offsets are provably wave-uniform.  Out-of-range reads return 0.
struct WSeg {  // this wave's slice of one packed layer: fragment j of its n-tile t at off + ((j*16 + t)*64 + lane)*16
    unsigned off;  // byte offset into the weight blob
    int jtot;      // (unused by the k-iteration-major layout; kept for segment bookkeeping)
    int jn;
};
struct WStream {
    __amdgpu_buffer_rsrc_t rsrc;
    const char* base;
    unsigned lane_off;
};
__device__ __forceinline__ WStream wstream(const MlpArgs& a, int lane) {
    WStream w;
    w.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w_base), 0, (int)a.w_bytes, 0x00020000);
    w.base = reinterpret_cast<const char*>(a.w_base);
    w.lane_off = 16u * (unsigned)lane;
    return w;
}
template <int NT>
__device__ __forceinline__ WSeg wseg(const WStream& ws, const float* packed, int jtot, int j0, int jn, int wave) {
    WSeg s;
    s.off = (unsigned)(reinterpret_cast<const char*>(packed) - ws.base) + (unsigned)((j0 * 16 + NT * wave) * 64) * 16u;
    s.jtot = jtot;
    s.jn = jn;
    return s;
}
// fragment (k-iteration j, local n-tile nt) of a segment, this lane's 16 bytes
__device__ __forceinline__ f32x4 wload(const WStream& ws, unsigned seg_off, int nt, int j) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.lane_off, seg_off + (unsigned)((j * 16 + nt) * 64) * 16u, 0);
    return __builtin_bit_cast(f32x4, v);
}

template <int D, int NT>
struct WRing {
    f32x4 f[D][NT];
#ifdef PNY_EXP_FOOT  // timing-only experiment: stream cyclically through the first PNY_EXP_FOOT k-iterations
    unsigned exp_base;  // (16 KiB each) of the packed blob instead of the real layers (wrong results)
    int exp_ctr;
#endif
};

template <int D, int NT>
__device__ __forceinline__ void ring_fill(WRing<D, NT>& r, const WStream& ws, const WSeg& s) {
#pragma unroll
    for (int d = 0; d + 1 < D; ++d) {  // slot D-1 is loaded by the segment's first step (gemm_run)
        const int j = d < s.jn ? d : s.jn - 1;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) r.f[d][nt] = wload(ws, s.off, nt, j);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) r.f[D - 1][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// acc += W_slice * act over segment `cur`; leaves the ring holding the first WDEPTH-1 fragments of
// `next`.  Ring slots are STATIC: the loop is unrolled by the depth; slot d is consumed by the MFMAs
// of k-iteration j+d, and slot d-1 (whose MFMAs were issued in the previous step, so no register is
// still being read) is refilled with fragment j+d-1+WDEPTH *while* slot d's MFMAs issue.  (A rotating
// ring makes the compiler copy registers that are destinations of in-flight loads, which costs an
// s_waitcnt vmcnt(0) per iteration -- measured: 15 % MFMA-pipe idle inside the loop.)
// sched_group_barrier spreads the step's loads / LDS reads / address arithmetic between the MFMAs
// so a wave that is alone on its SIMD (its partner waiting at a barrier) still issues them in the
// shadow of its own 64-cycle MFMAs instead of between MFMA blocks.
// Every segment length is a multiple of the depth (K padded accordingly on the host).
// Activation fragments (LDS) alternate between two static slots, one iteration ahead.
template <class C>
__device__ __forceinline__ void gemm_run(f32x16 (&acc)[C::NT][C::MT], WRing<C::WDEPTH, C::NT>& r, const WStream& ws,
                                         const WSeg& cur, const WSeg& next, const float4* __restrict__ act, int lane) {
    constexpr int NT = C::NT, MT = C::MT, TMc = C::TM, WDEPTH = C::WDEPTH;
    const int m0 = lane & 31, hh = lane >> 5;
    const float4* bp = act + hh * TMc + m0;
    const int jn = cur.jn, jl = jn - 1;
    float4 B[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) B[0][mt] = bp[32 * mt];
    for (int j = 0; j < jn; j += WDEPTH) {
#pragma unroll
        for (int d = 0; d < WDEPTH; ++d) {
            const int jd = j + d;
            const int j1 = (jd + 1 < jl) ? jd + 1 : jl;
            __builtin_amdgcn_sched_barrier(0);
            // refill the slot consumed one step ago: fragment (jd - 1) + WDEPTH of the stream
            const int dp = (d + WDEPTH - 1) % WDEPTH;
            const int jj = jd - 1 + WDEPTH;
            const bool in_cur = jj < jn;
            const int jx = in_cur ? jj : jj - jn;  // next.jn >= WDEPTH, so jx is in range
            const unsigned src = in_cur ? cur.off : next.off;
            // (at jd == 0 this loads fragment WDEPTH-1 of this very segment: on entry the ring holds
            //  fragments 0 .. WDEPTH-2 only, so there is no special case at segment boundaries)
#ifdef PNY_EXP_FOOT
            {
                (void)src;
                (void)jx;
                const int it = r.exp_ctr;
                r.exp_ctr = (it + 1 == PNY_EXP_FOOT) ? 0 : it + 1;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) r.f[dp][nt] = wload(ws, r.exp_base, nt, it);
            }
#else
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) r.f[dp][nt] = wload(ws, src, nt, jx);
#endif
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) B[(d + 1) & 1][mt] = bp[(2 * j1) * TMc + 32 * mt];
            mfma_iter<NT, MT>(acc, r.f[d], B[d & 1]);
            // issue pattern of the step: MFMA, then a few non-MFMA instructions, repeated
#pragma unroll
            for (int i = 0; i < 4 * NT * MT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // 1 MFMA
                if (i == 1 || i == 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
                if (i == 5 || i == 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);                        // <= 3 VALU/SALU
            }
        }
    }
}

// A lane's slice of a 512-entry bias: features 32*NT*wave + 32*nt + 8*q + 4*hh + 0..3.  Loaded
// before a barrier phase (bias_load) and applied after it, so its latency is not exposed.
template <int NT>
struct BiasRegs {
    float4 v[NT][4];
};

template <int NT>
__device__ __forceinline__ void bias_load(BiasRegs<NT>& b, const float* __restrict__ bias, int wave, int lane) {
    const int hh = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            b.v[nt][q] = *reinterpret_cast<const float4*>(bias + 32 * NT * wave + 32 * nt + 8 * q + 4 * hh);
}

template <int NT, int MT, bool ADD>
__device__ __forceinline__ void bias_apply(f32x16 (&acc)[NT][MT], const BiasRegs<NT>& b) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (ADD) {
                    acc[nt][mt][4 * q + 0] += b.v[nt][q].x;
                    acc[nt][mt][4 * q + 1] += b.v[nt][q].y;
                    acc[nt][mt][4 * q + 2] += b.v[nt][q].z;
                    acc[nt][mt][4 * q + 3] += b.v[nt][q].w;
                } else {
                    acc[nt][mt][4 * q + 0] = b.v[nt][q].x;
                    acc[nt][mt][4 * q + 1] = b.v[nt][q].y;
                    acc[nt][mt][4 * q + 2] = b.v[nt][q].z;
                    acc[nt][mt][4 * q + 3] = b.v[nt][q].w;
                }
            }
}

// relu as ONE v_max_f32 (fmaxf compiles to a canonicalising v_max x,x plus the max: 2 VALU per element,
// 128 per epilogue, on the issue port the partner wave's MFMAs share).  Same result as fmaxf(x, 0.f)
// for every input, NaN -> 0 included (IEEE mode: v_max returns the non-NaN operand).
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

// act[feature/4][m] = relu(acc): the next layer's B operand.  STASH (training): the same values also go to the
// activation stash in HBM in the SAME [feature/4][m] float4 layout (one 64-sample tile = one contiguous slot per
// tensor): a wave-instruction writes two 512-byte segments.  The backward chain (mlp_bwd.hip) reads them back as relu
// masks in accumulator layout, the weight-gradient GEMMs as their B operand.
template <int NT, int MT, bool STASH = false>
__device__ __forceinline__ void store_relu(const f32x16 (&acc)[NT][MT], float4* __restrict__ act, int wave, int lane,
                                           float4* __restrict__ stash = nullptr) {
    constexpr int TMc = 32 * MT;
    const int m0 = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v;
                v.x = relu1(acc[nt][mt][4 * q + 0]);
                v.y = relu1(acc[nt][mt][4 * q + 1]);
                v.z = relu1(acc[nt][mt][4 * q + 2]);
                v.w = relu1(acc[nt][mt][4 * q + 3]);
                const int kg = 8 * NT * wave + 8 * nt + 2 * q + hh;
                act[kg * TMc + 32 * mt + m0] = v;
                if constexpr (STASH) stash[kg * TMc + 32 * mt + m0] = v;
            }
}

// Projected-latent variant of the block entry: the LDS buffer holds the interpolated lin_z output in
// the same [feature/4][m] layout; every lane reads the quads of ITS accumulator elements, adds them
// to the residual stream and overwrites the same slots with relu(h) -- no other lane touches them.
template <int NT, int MT>
__device__ __forceinline__ void store_relu_addz(f32x16 (&acc)[NT][MT], float4* __restrict__ act, int wave, int lane) {
    constexpr int TMc = 32 * MT;
    const int m0 = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kg = 8 * NT * wave + 8 * nt + 2 * q + hh;
                float4* slot = act + kg * TMc + 32 * mt + m0;
                const float4 zv = *slot;
                acc[nt][mt][4 * q + 0] += zv.x;
                acc[nt][mt][4 * q + 1] += zv.y;
                acc[nt][mt][4 * q + 2] += zv.z;
                acc[nt][mt][4 * q + 3] += zv.w;
                float4 v;
                v.x = relu1(acc[nt][mt][4 * q + 0]);
                v.y = relu1(acc[nt][mt][4 * q + 1]);
                v.z = relu1(acc[nt][mt][4 * q + 2]);
                v.w = relu1(acc[nt][mt][4 * q + 3]);
                *slot = v;
            }
}

// Cross-view running sum slab (per workgroup, coalesced 16-byte accesses: register quad q of tile
// position p of lane l at float4 index (4p + q)*64 + l).  It is written and read with non-temporal
// accesses: 32 CUs x 128 KiB would otherwise evict the layer weights from the XCD's 4 MiB L2 three
// times per tile.
template <int NT, int MT>
__device__ __forceinline__ void slab_store(const f32x16 (&h)[NT][MT], float* slab) {
    f32x4* s4 = reinterpret_cast<f32x4*>(slab);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
                v.x = h[nt][mt][4 * q + 0];
                v.y = h[nt][mt][4 * q + 1];
                v.z = h[nt][mt][4 * q + 2];
                v.w = h[nt][mt][4 * q + 3];
                __builtin_nontemporal_store(v, s4 + ((nt * MT + mt) * 4 + q) * 64);
            }
}
template <int NT, int MT>
__device__ __forceinline__ void slab_load(f32x16 (&t)[NT][MT], const float* slab) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(slab);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = __builtin_nontemporal_load(s4 + ((nt * MT + mt) * 4 + q) * 64);
                t[nt][mt][4 * q + 0] = v.x;
                t[nt][mt][4 * q + 1] = v.y;
                t[nt][mt][4 * q + 2] = v.z;
                t[nt][mt][4 * q + 3] = v.w;
            }
}

// One pre-activation residual block (reference resnetfc.py:53-62):
//   net = fc_0(relu(h)); h = h + fc_1(relu(net))
// b_fc1 already contains the next block's lin_z bias (folded on the host, api.hip).
// `after` is the weight segment that follows this block in the stream.  With slab != nullptr the
// other views' running sum is fetched into the (then dead) net registers underneath the fc_1
// GEMM and added to h afterwards.  ADDZ: the LDS buffer holds this block's interpolated lin_z output
// (projected-latent variant), added to h on entry.
// STASH: relu(h) and relu(net) (the B operands of fc_0 and fc_1) are also written to stash_h / stash_net.
template <class C, bool ADDZ, bool STASH = false>
__device__ __forceinline__ void res_block(f32x16 (&h)[C::NT][C::MT], WRing<C::WDEPTH, C::NT>& ring, const WStream& ws,
                                          const MlpWeights& w, int blk, const WSeg& after, float4* act, int wave,
                                          int lane, const float* slab ST_ARG, float4* stash_h = nullptr,
                                          float4* stash_net = nullptr) {
    constexpr int NT = C::NT, MT = C::MT;
    f32x16 net[NT][MT];
    BiasRegs<NT> bias;
    const WSeg s_fc0 = wseg<NT>(ws, w.w_fc0[blk], 64, 0, 64, wave);
    const WSeg s_fc1 = wseg<NT>(ws, w.w_fc1[blk], 64, 0, 64, wave);
    {
        ST_BEGIN();
        bias_load<NT>(bias, w.b_fc0[blk], wave, lane);
        __builtin_amdgcn_sched_barrier(0);
#ifdef PNY_STAMP
        const unsigned long long f0 = stamp_now();
        TRACE();  // ev A: arrive at sync1 (end of previous GEMM)
        __syncthreads();
        const unsigned long long f1 = stamp_now();
        TRACE();  // ev B: past sync1
        if (ADDZ)
            store_relu_addz<NT, MT>(h, act, wave, lane);
        else
            store_relu<NT, MT>(h, act, wave, lane);
        const unsigned long long f2 = stamp_now();
        __syncthreads();
        const unsigned long long f3 = stamp_now();
        TRACE();  // ev C: past sync2 (GEMM fc0 starts)
        st.acc[ST_SYNC1] += f1 - f0;
        st.acc[ST_WRITE] += f2 - f1;
        st.acc[ST_SYNC2] += f3 - f2;
#else
        __syncthreads();
        if (ADDZ)
            store_relu_addz<NT, MT>(h, act, wave, lane);
        else
            store_relu<NT, MT, STASH>(h, act, wave, lane, stash_h);
        __syncthreads();
#endif
        bias_apply<NT, MT, false>(net, bias);
        ST_END(ST_STORE);
    }
    {
        ST_BEGIN();
        gemm_run<C>(net, ring, ws, s_fc0, s_fc1, act, lane);
        ST_END(ST_GEMM);
        TRACE();  // ev D: end of GEMM fc0
    }
    {
        ST_BEGIN();
        bias_load<NT>(bias, w.b_fc1[blk], wave, lane);
        __builtin_amdgcn_sched_barrier(0);
#ifdef PNY_STAMP
        const unsigned long long f0 = stamp_now();
        __syncthreads();
        const unsigned long long f1 = stamp_now();
        store_relu<NT, MT>(net, act, wave, lane);
        if (slab) slab_load<NT, MT>(net, slab);
        const unsigned long long f2 = stamp_now();
        __syncthreads();
        const unsigned long long f3 = stamp_now();
        st.acc[ST_SYNC1] += f1 - f0;
        st.acc[ST_WRITE] += f2 - f1;
        st.acc[ST_SYNC2] += f3 - f2;
#else
        __syncthreads();
        store_relu<NT, MT, STASH>(net, act, wave, lane, stash_net);
        if (slab) slab_load<NT, MT>(net, slab);
        __syncthreads();
#endif
        bias_apply<NT, MT, true>(h, bias);
        ST_END(ST_STORE);
    }
    {
        ST_BEGIN();
        gemm_run<C>(h, ring, ws, s_fc1, after, act, lane);
        ST_END(ST_GEMM);
    }
    if (slab) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[nt][mt][r] = net[nt][mt][r] + h[nt][mt][r];
    }
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
        // (a 64-bit software division per thread is ~100 instructions: 32-bit whenever the launch fits)
        const long long ray = a.idx32 ? (long long)((unsigned)s / (unsigned)a.K) : s / a.K;
        const float4* r = reinterpret_cast<const float4*>(a.rays + ray * 8);  // rows are 32 bytes (api.hip checks alignment)
        const float4 r0 = r[0], r1 = r[1];
        const float zz = a.z[s];
        d[0] = r0.w;
        d[1] = r1.x;
        d[2] = r1.y;
        p[0] = r0.x + zz * d[0];
        p[1] = r0.y + zz * d[1];
        p[2] = r0.z + zz * d[2];
    }
}

// First view of the object a tile's samples belong to (MlpArgs::obj_pts; 0 for an ungrouped scene).  Uniform per tile.
__device__ __forceinline__ int tile_view_base(const MlpArgs& a, long long first_sample) {
    if (a.obj_pts == 0) return 0;
    if (first_sample >= a.n_points) first_sample = a.n_points - 1;
    const long long obj = a.idx32 ? (long long)((unsigned)first_sample / (unsigned)a.obj_pts) : first_sample / a.obj_pts;
    return (int)obj * a.NS;
}

// Positional-code entry e of the 64-row (42 valid) input column (reference code.py:30-42 layout:
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

// Per (view, tile) prologue: B operand of lin_in into act k-groups 0..15, and the four bilinear
// taps of every sample into the tap table.
template <class C>
__device__ __forceinline__ void prologue(const MlpArgs& a, int v, long long tile, float4* act, float4* tap_tab,
                                         int tid) {
    constexpr int TMc = C::TM, NPART = C::THREADS / TMc;
    const int m = tid % TMc, part = tid / TMc;
    long long s = tile * TMc + m;
    if (s >= a.n_points) s = a.n_points - 1;
    float p[3], d[3];
    load_point(a, s, p, d);
    const Cam cam = a.cams[tile_view_base(a, tile * TMc) + v];
    float xr[3], xc[3], vd[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        xr[i] = cam.w2c[4 * i + 0] * p[0] + cam.w2c[4 * i + 1] * p[1] + cam.w2c[4 * i + 2] * p[2];
        xc[i] = xr[i] + cam.w2c[4 * i + 3];
        vd[i] = cam.w2c[4 * i + 0] * d[0] + cam.w2c[4 * i + 1] * d[1] + cam.w2c[4 * i + 2] * d[2];
    }
    for (int g = part; g < D_IN_PAD / 4; g += NPART) {
        float4 x4;
        x4.x = input_entry(4 * g + 0, xr, vd, a.freq_factor, a.num_freqs);
        x4.y = input_entry(4 * g + 1, xr, vd, a.freq_factor, a.num_freqs);
        x4.z = input_entry(4 * g + 2, xr, vd, a.freq_factor, a.num_freqs);
        x4.w = input_entry(4 * g + 3, xr, vd, a.freq_factor, a.num_freqs);
        act[g * TMc + m] = x4;
    }
    if (part == NPART - 1) {
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
        int offs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = (xs[k] >= 0.f) && (xs[k] <= (float)(a.Wl - 1)) && (ys[k] >= 0.f) && (ys[k] <= (float)(a.Hl - 1));
            offs[k] = 0;
            if (ok) {
                offs[k] = ((int)ys[k] * a.Wl + (int)xs[k]) * a.tap_stride;
            } else {
                wgt[k] = wgt[k] * 0.0f;  // out-of-range tap contributes 0 (NaN coordinates stay NaN, as in ATen)
            }
            if (cull || (a.yolo && (wgt[k] != wgt[k]))) wgt[k] = 0.0f;
        }
        // one 32-byte record per sample: {offset[4] (int bits), weight[4]}
        tap_tab[2 * m] = make_float4(__int_as_float(offs[0]), __int_as_float(offs[1]), __int_as_float(offs[2]),
                                     __int_as_float(offs[3]));
        tap_tab[2 * m + 1] = make_float4(wgt[0], wgt[1], wgt[2], wgt[3]);
    }
}

// The four bilinear taps of sample s in source view v: element offsets (pixel index x tap_stride) and weights, exactly as
// the prologue above computes them (projection models.py:219-230; grid_sample coordinates encoder.py:97-98,
// align_corners=True, zeros padding; YOLO-mode culling).  Used by kernels outside the forward chain (latent_grad.hip).
// v: index into the scene's view list (a grouped scene: tile_view_base + the object's view).
__device__ __forceinline__ void sample_taps(const MlpArgs& a, int v, long long s, int (&offs)[4], float (&wgt)[4]) {
    float p[3], d[3];
    load_point(a, s, p, d);
    const Cam cam = a.cams[v];
    float xc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        xc[i] = (cam.w2c[4 * i + 0] * p[0] + cam.w2c[4 * i + 1] * p[1] + cam.w2c[4 * i + 2] * p[2]) + cam.w2c[4 * i + 3];
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
    wgt[0] = (x1 - ix) * (y1 - iy);
    wgt[1] = (ix - x0) * (y1 - iy);
    wgt[2] = (x1 - ix) * (iy - y0);
    wgt[3] = (ix - x0) * (iy - y0);
    const float xs[4] = {x0, x1, x0, x1};
    const float ys[4] = {y0, y0, y1, y1};
    const bool cull = a.yolo && !(xc[2] < 0.0f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool ok = (xs[k] >= 0.f) && (xs[k] <= (float)(a.Wl - 1)) && (ys[k] >= 0.f) && (ys[k] <= (float)(a.Hl - 1));
        offs[k] = 0;
        if (ok)
            offs[k] = ((int)ys[k] * a.Wl + (int)xs[k]) * a.tap_stride;
        else
            wgt[k] = wgt[k] * 0.0f;
        if (cull || (a.yolo && (wgt[k] != wgt[k]))) wgt[k] = 0.0f;
    }
}

// Bilinear gather of the latent (reference encoder.py:101 F.grid_sample, written tap by tap) into the
// B-operand layout act[(c - c0)/4][m], pipelined with the lin_z GEMM in chunks of GCH channels:
// gather_issue() starts the 16-byte tap loads of a chunk into registers, gather_commit() combines
// the four taps and writes the chunk to its LDS window.  The loads of chunk c+1 are issued before the
// MFMAs of chunk c (the registers are the then-dead `net` accumulators' budget); because vmcnt
// retires in order, the first ring wait of that GEMM still covers them, so about one ring period of
// their latency hides.
// A wave pass covers 8 samples x 8 channel quads: the lanes {l, l+8, .., l+56} read one 128-byte
// line per tap, each 8-lane group writes 128 contiguous LDS bytes.  A lane keeps the same sample for
// the whole gather, so its four tap pointers / weights are set up once per (view, block).
constexpr int GCH = 128;  // channels per gather chunk = 16 k-iterations of the lin_z GEMM

template <class C, int NB = 1>
struct GatherTaps {
    static constexpr int NMB = C::TM / 8;           // sample blocks of 8
    static constexpr int QSTEP = C::NW / NMB;       // q-blocks (32 channels) covered per pass of all waves
    static constexpr int QPW = (GCH / 32) / QSTEP;  // q-blocks per wave per chunk
    static_assert(QSTEP >= 1 && QPW >= 1 && QPW * QSTEP * 32 == GCH, "gather mapping");
    const float* t[4];  // tap base pointers (view, pixel, + 4*ql), channel 0
    float w[4];
    float4 x[NB][QPW][4];  // NB chunks in flight
};

template <class C, int NB>
__device__ __forceinline__ void gather_setup(GatherTaps<C, NB>& g, const float* view_base, const float4* tap_tab,
                                             int wave, int lane) {
    constexpr int NMB = GatherTaps<C>::NMB;
    const int m = (wave % NMB) * 8 + (lane & 7);
    const float* base = view_base + 4 * (lane >> 3);
    float4 o = tap_tab[2 * m];
    const float4 w = tap_tab[2 * m + 1];
#if defined(PNY_EXP_GATHER) && PNY_EXP_GATHER == 1  // timing only: every tap reads pixel 0 (wrong results)
    o = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
    g.t[0] = base + __float_as_int(o.x);
    g.t[1] = base + __float_as_int(o.y);
    g.t[2] = base + __float_as_int(o.z);
    g.t[3] = base + __float_as_int(o.w);
    g.w[0] = w.x;
    g.w[1] = w.y;
    g.w[2] = w.z;
    g.w[3] = w.w;
}

template <class C, int B = 0, int NB>
__device__ __forceinline__ void gather_issue(GatherTaps<C, NB>& g, int c0, int wave) {
    constexpr int NMB = GatherTaps<C>::NMB, QSTEP = GatherTaps<C>::QSTEP;
#pragma unroll
    for (int i = 0; i < GatherTaps<C>::QPW; ++i) {
        const int qb = wave / NMB + i * QSTEP;  // q-block (8 quads = 32 channels) within the chunk
#if defined(PNY_EXP_GATHER) && PNY_EXP_GATHER == 2  // timing only: no tap loads (wrong results)
#pragma unroll
        for (int k = 0; k < 4; ++k) g.x[B][i][k] = make_float4(g.w[k], (float)c0, (float)qb, 1.f);
#else
#pragma unroll
        for (int k = 0; k < 4; ++k) g.x[B][i][k] = *reinterpret_cast<const float4*>(g.t[k] + c0 + 32 * qb);
#endif
    }
}

// FUSED = false: ATen's operation order (nw*w + ne*w + sw*w + se*w, every product and sum rounded), used by the
// reference-order variant.  FUSED = true: the same blend as 1 multiply + 3 fused multiply-adds per component (16
// instead of 28 VALU instructions per float4) for the projected-latent variant, which is held to the 1e-4
// tolerance and not to an operation order.
template <class C, int B = 0, bool FUSED = false, int NB>
__device__ __forceinline__ void gather_commit(const GatherTaps<C, NB>& g, float4* act_win, int wave, int lane) {
    constexpr int TMc = C::TM, NMB = GatherTaps<C>::NMB, QSTEP = GatherTaps<C>::QSTEP;
    const int m = (wave % NMB) * 8 + (lane & 7);
    float4* dst = act_win + (lane >> 3) * TMc + m;
#pragma unroll
    for (int i = 0; i < GatherTaps<C>::QPW; ++i) {
        const int qb = wave / NMB + i * QSTEP;
        const float4(&x)[4] = g.x[B][i];
        float4 r;
        if (FUSED) {
            r.x = __builtin_fmaf(x[3].x, g.w[3], __builtin_fmaf(x[2].x, g.w[2], __builtin_fmaf(x[1].x, g.w[1], x[0].x * g.w[0])));
            r.y = __builtin_fmaf(x[3].y, g.w[3], __builtin_fmaf(x[2].y, g.w[2], __builtin_fmaf(x[1].y, g.w[1], x[0].y * g.w[0])));
            r.z = __builtin_fmaf(x[3].z, g.w[3], __builtin_fmaf(x[2].z, g.w[2], __builtin_fmaf(x[1].z, g.w[1], x[0].z * g.w[0])));
            r.w = __builtin_fmaf(x[3].w, g.w[3], __builtin_fmaf(x[2].w, g.w[2], __builtin_fmaf(x[1].w, g.w[1], x[0].w * g.w[0])));
        } else {
            r.x = ((x[0].x * g.w[0] + x[1].x * g.w[1]) + x[2].x * g.w[2]) + x[3].x * g.w[3];
            r.y = ((x[0].y * g.w[0] + x[1].y * g.w[1]) + x[2].y * g.w[2]) + x[3].y * g.w[3];
            r.z = ((x[0].z * g.w[0] + x[1].z * g.w[1]) + x[2].z * g.w[2]) + x[3].z * g.w[3];
            r.w = ((x[0].w * g.w[0] + x[1].w * g.w[1]) + x[2].w * g.w[2]) + x[3].w * g.w[3];
        }
        dst[(size_t)(8 * qb) * TMc] = r;
    }
}


}  // namespace pny
