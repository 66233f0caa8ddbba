// The split-f16 GEMM machinery shared by the f16x2 forward kernel (mlp_h2.hip) and the f16x2 backward chain (mlp_bwd_h2.hip):
// tile constants, the per-wave weight stream of [16-k step][n-tile][plane][lane] x 16-byte fragments behind a static-slot
// register ring, the GEMM loop over the LDS activation planes, the fp32 -> two f16 planes split.  See mlp_h2.hip for the
// layouts and DESIGN.md 4.0 for the arithmetic.
#pragma once
#include "mlp_core.h"

namespace pny {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

namespace h2 {
#ifndef PNY_H2_WD
#define PNY_H2_WD 2  // measured: 4 is 7 % faster inside the GEMMs, but its 32 more ring registers spill in the gather phase (-17 % overall)
#endif
// Tile shape of the translation unit (a wave owns NT n-tiles of 32 features x MT m-tiles of 32 samples; NT * MT = 4):
//   2 x 2 (default): 8 waves, 64-sample tiles, one workgroup per CU (152 KiB of LDS) -- mlp_h2.hip, mlp_bwd_h2.hip;
//   4 x 1 (mlp_h2s.hip, -DPNY_H2_NT=4 -DPNY_H2_MT=1): 4 waves, 32-sample tiles, 65 KiB of LDS, TWO workgroups per CU whose
//   phases interleave (one's gather / epilogue under the other's GEMM) at twice the weight stream per sample.
#ifndef PNY_H2_NT
#define PNY_H2_NT 2
#endif
#ifndef PNY_H2_MT
#define PNY_H2_MT 2
#endif
constexpr int NT = PNY_H2_NT, MT = PNY_H2_MT, TM = 32 * MT, THREADS = 64 * (16 / NT), WD = PNY_H2_WD;  // WD = ring depth in 16-k steps
static_assert(NT * MT == 4 && 16 % NT == 0, "tile shape");
constexpr bool LDS_BIAS = (NT == 2 && MT == 2);   // the bias table lives in LDS (22 KiB); the split shape reads biases from global
constexpr int ROW_BYTES = TM * 16;          // one plane of one row (8 features x TM samples x f16)
constexpr int ACT_BYTES = 64 * 2 * ROW_BYTES;  // activation buffer: [row = feature / 8][plane][sample] x 16 bytes (128 KiB at TM = 64)
constexpr int TAP_BYTES = 32 * TM;             // tap table
constexpr int MAX_NB = LDS_BIAS ? 6 : MAX_BLOCKS;   // bias table: (1 + 2 n_blocks) x 512 floats must fit the 160 KiB with the rest
__host__ __device__ constexpr int lds_bytes(int n_blocks) { return ACT_BYTES + TAP_BYTES + (LDS_BIAS ? (1 + 2 * n_blocks) * HID * 4 : 0); }
using C = Cfg<NT, MT>;
}  // namespace h2

struct H2Seg {
    unsigned off;  // byte offset of fragment (step 0, this wave's first n-tile, plane 0) in the weight blob
    int jn;        // 16-k steps
};
__device__ __forceinline__ H2Seg h2seg(const WStream& ws, const float* packed, int jn, int wave) {
    H2Seg s;
    s.off = (unsigned)(reinterpret_cast<const char*>(packed) - ws.base) + (unsigned)((h2::NT * wave) * 2 * 64) * 16u;
    s.jn = jn;
    return s;
}
// cache policy of the weight-fragment loads (gfx940+ aux bits: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef PNY_H2_WAUX
#define PNY_H2_WAUX 0
#endif
// fragment (step j, local n-tile nt, plane p): this lane's 16 bytes = 8 halves W[32 nt_g + (l & 31)][16 j + 8 (l >> 5) + 0..7]
__device__ __forceinline__ h8 h2load(const WStream& ws, unsigned seg_off, int nt, int p, int j) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.lane_off, seg_off + (unsigned)(((j * 16 + nt) * 2 + p) * 64) * 16u, PNY_H2_WAUX);
    return __builtin_bit_cast(h8, v);
}
struct H2Ring {
    h8 f[h2::WD][h2::NT][2];
};
__device__ __forceinline__ void h2ring_fill(H2Ring& r, const WStream& ws, const H2Seg& s) {
#pragma unroll
    for (int d = 0; d + 1 < h2::WD; ++d) {
        const int j = d < s.jn ? d : s.jn - 1;
#pragma unroll
        for (int nt = 0; nt < h2::NT; ++nt)
#pragma unroll
            for (int p = 0; p < 2; ++p) r.f[d][nt][p] = h2load(ws, s.off, nt, p, j);
    }
#pragma unroll
    for (int nt = 0; nt < h2::NT; ++nt)
#pragma unroll
        for (int p = 0; p < 2; ++p) r.f[h2::WD - 1][nt][p] = h8{0, 0, 0, 0, 0, 0, 0, 0};
}

// acc += W_slice . act over segment `cur` (its 16-k steps a multiple of the ring depth); leaves the ring holding the first
// WD - 1 steps of `next`.  Per step and accumulator tile: x1 w1 + x2 w1 + x1 w2.  Static ring slots as in gemm_run
// (mlp_core.h): slot d is consumed by step j + d while slot d - 1 is refilled with step j + d - 1 + WD.
// Timing-only experiment (-DPNY_H2_EXP_MFMA16, wrong results): every 32x32x16 MFMA replaced by two 16x16x32 MFMAs on the
// same operand registers (the same FLOPs and operand traffic) -- what the chip's clock does with the other MFMA shape.
#ifdef PNY_H2_EXP_MFMA16
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 h2mfma(h8 a, h8 b, f32x16 c) {
    f32x4v c0 = {c[0], c[1], c[2], c[3]}, c1 = {c[4], c[5], c[6], c[7]};
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
    c[0] = c0[0]; c[1] = c0[1]; c[2] = c0[2]; c[3] = c0[3];
    c[4] = c1[0]; c[5] = c1[1]; c[6] = c1[2]; c[7] = c1[3];
    return c;
}
#define PNY_H2_MFMA_PER 2
#else
__device__ __forceinline__ f32x16 h2mfma(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
#define PNY_H2_MFMA_PER 1
#endif
struct NoSide {
    __device__ __forceinline__ void operator()() const {}
};
template <class Side = NoSide>
__device__ __forceinline__ void h2gemm(f32x16 (&acc)[h2::NT][h2::MT], H2Ring& r, const WStream& ws, const H2Seg& cur,
                                       const H2Seg& next, const char* planes, int lane, Side side = Side()) {
    using namespace h2;
    const int m0 = lane & 31, hh = lane >> 5;
    const char* bp = planes + hh * (2 * ROW_BYTES) + m0 * 16;   // row 2 j + hh, plane 0, sample m0
    const int jn = cur.jn, jl = jn - 1;
    h8 B[2][MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int p = 0; p < 2; ++p) B[0][mt][p] = *reinterpret_cast<const h8*>(bp + p * ROW_BYTES + 32 * mt * 16);
    for (int j = 0; j < jn; j += WD) {
        // `side` (loads whose results are needed after this GEMM: the first chunk of the next block's gather) is issued
        // from INSIDE the loop: placed in front of it the compiler sinks the loads behind the loop, to their first use
        if (j == WD) side();
#ifdef PNY_H2_EXP_LOCKSTEP   // experiment: a workgroup barrier every PNY_H2_EXP_LOCKSTEP steps keeps the two waves of a SIMD level
        if (j > 0 && (j % PNY_H2_EXP_LOCKSTEP) == 0) __builtin_amdgcn_s_barrier();
#endif
#pragma unroll
        for (int d = 0; d < WD; ++d) {
            const int jd = j + d;
            const int j1 = (jd + 1 < jl) ? jd + 1 : jl;
            const char* bj = bp + j1 * (4 * ROW_BYTES);
            __builtin_amdgcn_sched_barrier(0);
            const int dp = (d + WD - 1) % WD;
            const int jj = jd - 1 + WD;
            const bool in_cur = jj < jn;
            const int jx = in_cur ? jj : jj - jn;
            const unsigned src = in_cur ? cur.off : next.off;
            // Three groups of four INDEPENDENT MFMAs (x1 w1, then x2 w1, then x1 w2 over the four accumulator tiles),
            // fenced so the scheduler cannot regroup them by accumulator (dependent MFMAs back to back stall the matrix
            // pipe).  The weight loads of the step ride in the first group, the LDS reads of the next step's B fragments
            // in the other two.
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int p = 0; p < 2; ++p) r.f[dp][nt][p] = h2load(ws, src, nt, p, jx);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = h2mfma(r.f[d][nt][0], B[d & 1][mt][0], acc[nt][mt]);
#ifndef PNY_H2_NOSCHED
#pragma unroll
            for (int i = 0; i < NT * MT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, PNY_H2_MFMA_PER, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, NT * 2 / (NT * MT), 0);   // the step's NT x 2 weight loads
                __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 2; ++p) B[(d + 1) & 1][0][p] = *reinterpret_cast<const h8*>(bj + p * ROW_BYTES);   // (m-tile 0)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = h2mfma(r.f[d][nt][0], B[d & 1][mt][1], acc[nt][mt]);
#ifndef PNY_H2_NOSCHED
#pragma unroll
            for (int i = 0; i < NT * MT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, PNY_H2_MFMA_PER, 0);
                if (i < 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MT > 1) {
#pragma unroll
                for (int p = 0; p < 2; ++p) B[(d + 1) & 1][MT - 1][p] = *reinterpret_cast<const h8*>(bj + p * ROW_BYTES + 32 * 16);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = h2mfma(r.f[d][nt][1], B[d & 1][mt][0], acc[nt][mt]);
#ifndef PNY_H2_NOSCHED
#pragma unroll
            for (int i = 0; i < NT * MT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, PNY_H2_MFMA_PER, 0);
                if (MT > 1 && i < 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);
            }
#endif
            // keep this step's B fragments allocated until here: the fragments of step d + 1 (read from LDS during this
            // step) must not be given registers that MFMAs of this step still have to read
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int p = 0; p < 2; ++p) asm volatile("" ::"v"(B[d & 1][mt][p]));
        }
    }
}

// split 4 fp32 values into the two f16 planes (round to nearest)
// Two values per call: x1 pair = v_cvt_pk_f16_f32 (round to nearest even), residuals x - f32(x1) by v_fma_mix_f32 reading the
// f16 halves in place (exact, like the subtraction), x2 pair = v_cvt_pk_f16_f32 of the residuals: 4 VALU instructions
// per pair where the compiler's form of the C expression takes 8 (scalar convert, convert back, subtract, two packs).
__device__ __forceinline__ void split2(float a, float b, unsigned& p0, unsigned& p1) {
#ifdef PNY_H2_PLAIN_SPLIT
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    h2v q0, q1;
    q0[0] = (_Float16)a;
    q0[1] = (_Float16)b;
    q1[0] = (_Float16)(a - (float)q0[0]);
    q1[1] = (_Float16)(b - (float)q0[1]);
    p0 = __builtin_bit_cast(unsigned, q0);
    p1 = __builtin_bit_cast(unsigned, q1);
#else
    float ra, rb;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p0) : "v"(a), "v"(b));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(p0), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(p0), "v"(b));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p1) : "v"(ra), "v"(rb));
#endif
}
__device__ __forceinline__ void split4(float a, float b, float c, float d, h4& p0, h4& p1) {
    uint2 u0, u1;
    split2(a, b, u0.x, u1.x);
    split2(c, d, u0.y, u1.y);
    p0 = __builtin_bit_cast(h4, u0);
    p1 = __builtin_bit_cast(h4, u1);
}

template <int NT_, int MT_>
__device__ __forceinline__ void h2zero(f32x16 (&t)[NT_][MT_]) {
#pragma unroll
    for (int nt = 0; nt < NT_; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT_; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) t[nt][mt][r] = 0.f;
}

}  // namespace pny
