// The WIDE shape of the fused conditioned-MLP kernel on split-f16 operands (gfx950, MI355X): 4 waves x 512 registers, one wave
// per SIMD, v_mfma_f32_16x16x32_f16.
//
// Same function, same LDS activation image and the same three products per fp32 multiply as mlp_h2.hip (x1 w1 + x2 w1 + x1 w2,
// fp32 accumulation) -- what changes is who holds what:
//   * a workgroup is 4 waves; wave w owns features [128 w, 128 w + 128) of the 64 samples of the tile: 8 x 4 accumulator tiles
//     of 16 x 16 (128 registers for the residual stream h, 128 for net; the register file of a SIMD belongs to ONE wave);
//   * a B fragment read from LDS serves 8 n-tiles instead of 4: half the LDS read traffic per MFMA;
//   * v_mfma_f32_16x16x32_f16: the accumulator tile of a lane is ONE feature quad of ONE sample, so the 8-byte plane slots an
//     epilogue writes are 256 contiguous bytes per 16 lanes -- no bank conflict (the 32 x 32 layout's are 2-way) -- and the chip
//     holds a higher clock on this MFMA shape (MI355X_MICROARCH.md, DVFS give-back 7; measured here: profiles/r03_*);
//   * weights: one stream per wave, [32-k step][n-tile of 16][plane][lane] x 16 bytes (api.hip pack_layer_h3); a step's 16
//     fragments are consumed n-tile by n-tile (12 MFMAs each) and every fragment is refetched for the NEXT step right behind
//     its last use -- 7/8 of a step (1300 cycles) of latency cover with 64 registers and no separate ring.
// Render only (no STASH instantiation).  PNYOLO_H2_WIDE=1 selects it for full-size projected launches (api.hip run_mlp).
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <vector>

#include "mlp_h2_core.h"

namespace pny {

typedef float f32x4a __attribute__((ext_vector_type(4)));

// The same source in a second shape (mlp_h2n.hip: -DPNY_HW_NW=8): EIGHT waves of 256 registers, two per SIMD like mlp_h2.hip, a wave
// owning 64 features (4 x 4 tiles of 16 x 16) -- the 16 x 16 x 32 MFMA, its conflict-free plane writes and the n-tile-major
// weight refetch in the occupancy of the product kernel.
#ifndef PNY_HW_NW
#define PNY_HW_NW 4
#endif
#if PNY_HW_NW == 4
#define PNY_HW_KERNEL pny_mlp_h2w_kernel
#define PNY_HW_LAUNCH launch_mlp_h2w
#define PNY_HW_STAMPBUF PNY_HW_STAMPBUF
#else
#define PNY_HW_KERNEL pny_mlp_h2n_kernel
#define PNY_HW_LAUNCH launch_mlp_h2n
#define PNY_HW_STAMPBUF g_h2n_stamp_buf
#endif

// Diagnostic build only (-DPNY_H2_STAMP): s_memtime brackets around the phases of a tile (as in mlp_h2.hip)
#ifdef PNY_H2_STAMP
enum { HS_TOTAL = 0, HS_GEMM, HS_GATHER_WAIT, HS_GATHER, HS_EPI_WAIT, HS_EPI, HS_PROLOGUE, HS_LINOUT, HS_SLAB, HS_REAL, HS_N };
__device__ unsigned long long* PNY_HW_STAMPBUF;
__device__ __forceinline__ unsigned long long hwnow() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define HS_T0() hs_t_ = hwnow()
#define HS_LAP(cat)                            \
    {                                          \
        const unsigned long long n_ = hwnow(); \
        hs_acc[cat] += n_ - hs_t_;             \
        hs_t_ = n_;                            \
    }
#else
#define HS_T0()
#define HS_LAP(cat)
#endif

namespace hw {
constexpr int NW = PNY_HW_NW, THREADS = 64 * NW, TM = 64;
constexpr int NT = 32 / NW, MT = 4;           // 16 x 16 accumulator tiles per wave: (512 / NW) features x 64 samples
constexpr int FPG = 8 / NT;                   // next-step B fragments read per n-tile group
constexpr int SPW = 8 / NW;                   // 8-sample blocks a wave serves in the gather
static_assert(NW == 4 || NW == 8, "shape");
constexpr int ROW_BYTES = TM * 16;            // one plane of one row (8 features x TM samples x f16)
constexpr int ACT_BYTES = 64 * 2 * ROW_BYTES; // [row = feature / 8][plane][sample] x 16 bytes = 128 KiB
constexpr int TAP_BYTES = 32 * TM;
constexpr int MAX_NB = 6;
__host__ __device__ constexpr int lds_bytes(int n_blocks) { return ACT_BYTES + TAP_BYTES + (1 + 2 * n_blocks) * HID * 4; }
}  // namespace hw

struct HwSeg {
    unsigned off;  // byte offset of fragment (step 0, this wave's first n-tile, plane 0) in the weight blob
    int jn;        // 32-k steps (even)
};
__device__ __forceinline__ HwSeg hwseg(const WStream& ws, const float* packed, int jn, int wave) {
    HwSeg s;
    s.off = (unsigned)(reinterpret_cast<const char*>(packed) - ws.base) + (unsigned)((hw::NT * wave) * 2 * 64) * 16u;
    s.jn = jn;
    return s;
}
// fragment (step j, local n-tile nt, plane p): this lane's 8 halves W[16 nt_g + (l & 15)][32 j + 8 (l >> 4) + 0..7]
__device__ __forceinline__ h8 hwload(const WStream& ws, unsigned seg_off, int nt, int p, int j) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.lane_off, seg_off + (unsigned)(((j * 32 + nt) * 2 + p) * 64) * 16u, 0);
    return __builtin_bit_cast(h8, v);
}
// Weight fragments in flight: RD steps (of 32 k) ahead of the MFMAs.  One wave per SIMD has no partner wave to cover a
// fetch that misses L2 (7 % of them do, served by the Infinity Cache at 2-3x the latency): with RD = 1 a fragment is
// refetched 7/8 of a step (1.3 k cycles) before its next use and the GEMM phases run the matrix pipe at 78 %; RD = 2 gives
// 15/8 of a step for 64 more registers.
#ifndef PNY_HW_PF
#define PNY_HW_PF 0   // gather pieces of the next block fetched underneath the fc_1 GEMM (8-wave shape).  Measured: 0 -> 39.5, 1 -> 43.6, 2 -> 47.8, 4 -> 53.7 ms per launch (130 / 230 / 298 spilled registers, some inside the GEMM loops)
#endif
#ifndef PNY_HW_RD
#define PNY_HW_RD 1   // (2 measured on the 4-wave shape: 52.2 vs 46.2 ms per launch -- 509 spilled registers)
#endif
#ifndef PNY_HW_GD
#define PNY_HW_GD (PNY_HW_NW == 4 ? 3 : 4)   // gather pieces (16 registers per sample block served) in flight
#endif
struct HwRing {
    h8 f[PNY_HW_RD][hw::NT][2];   // slot d holds step j + d at the head of step j (rotating statically: steps are unrolled by 2)
};
__device__ __forceinline__ void hwring_fill(HwRing& r, const WStream& ws, const HwSeg& s) {
#pragma unroll
    for (int d = 0; d < PNY_HW_RD; ++d)
#pragma unroll
        for (int nt = 0; nt < hw::NT; ++nt)
#pragma unroll
            for (int p = 0; p < 2; ++p) r.f[d][nt][p] = hwload(ws, s.off, nt, p, d < s.jn ? d : s.jn - 1);
}

template <class Acc>
__device__ __forceinline__ void hwzero(Acc (&t)[hw::NT][hw::MT]) {
#pragma unroll
    for (int nt = 0; nt < hw::NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < hw::MT; ++mt) t[nt][mt] = f32x4a{0.f, 0.f, 0.f, 0.f};
}

// One 32-k step: B = the step's activation fragments (4 m-tiles x 2 planes), Bn receives the next step's (read from `bnext`).
// n-tile by n-tile: 12 MFMAs on the n-tile's two weight fragments (ring slot D), which are then refetched for step + RD.
template <int D>
__device__ __forceinline__ void hwstep(f32x4a (&acc)[hw::NT][hw::MT], HwRing& r, const h8 (&B)[hw::MT][2], h8 (&Bn)[hw::MT][2],
                                       const char* bnext, const WStream& ws, unsigned src, int jx) {
    using namespace hw;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r.f[D][nt][0], B[mt][0], acc[nt][mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r.f[D][nt][0], B[mt][1], acc[nt][mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r.f[D][nt][1], B[mt][0], acc[nt][mt], 0, 0, 0);
        // the next step's 8 activation fragments, FPG per n-tile group
#pragma unroll
        for (int i = 0; i < FPG; ++i) {
            const int f = nt * FPG + i;
            Bn[f >> 1][f & 1] = *reinterpret_cast<const h8*>(bnext + (f & 1) * ROW_BYTES + (f >> 1) * 256);
        }
        r.f[D][nt][0] = hwload(ws, src, nt, 0, jx);
        r.f[D][nt][1] = hwload(ws, src, nt, 1, jx);
#ifndef PNY_H2_NOSCHED
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, FPG, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
}

// acc += W_slice . act over segment `cur` (an even number of steps); leaves the ring holding the first RD steps of `next`.
// `side`: loads whose results are needed AFTER this GEMM (the next block's first gather pieces), issued from inside the loop --
// placed in front of it the compiler sinks them behind the loop, to their first use (mlp_h2.hip h2gemm).
struct HwNoSide {
    __device__ __forceinline__ void operator()() const {}
};
template <class Side = HwNoSide>
__device__ __forceinline__ void hwgemm(f32x4a (&acc)[hw::NT][hw::MT], HwRing& r, const WStream& ws, const HwSeg& cur, const HwSeg& next,
                                       const char* planes, int lane, Side side = Side()) {
    using namespace hw;
    constexpr int RD = PNY_HW_RD;
    const char* bp = planes + (lane >> 4) * (2 * ROW_BYTES) + (lane & 15) * 16;   // row 4 j + (lane >> 4), plane 0, sample lane & 15
    const int jn = cur.jn;
    h8 B0[MT][2], B1[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int p = 0; p < 2; ++p) B0[mt][p] = *reinterpret_cast<const h8*>(bp + p * ROW_BYTES + mt * 256);
    for (int j = 0; j < jn; j += 2) {
        if (j == 2) side();
        // step j consumes slot 0 (RD = 1) / slot 0 (RD = 2) and refetches it with step j + RD; step j + 1 the other way round
        const int ja = j + RD, jb = j + 1 + RD;
        const bool ina = ja < jn, inb = jb < jn;
        hwstep<0>(acc, r, B0, B1, bp + (j + 1) * (8 * ROW_BYTES), ws, ina ? cur.off : next.off, ina ? ja : ja - jn);
        const bool last = j + 2 >= jn;
        hwstep<RD - 1>(acc, r, B1, B0, bp + (last ? j + 1 : j + 2) * (8 * ROW_BYTES), ws, inb ? cur.off : next.off, inb ? jb : jb - jn);
    }
}

// Block entry / GEMM epilogue (mlp_h2.hip h2epilogue for the 16 x 16 accumulator layout): acc += bias (+ the staged fp32
// projection, ADDZ), then the planes of relu(acc) go to the slots of the lane's own feature quads.
// accumulator tile (nt, mt) of lane l = features 128 w + 16 nt + 4 (l >> 4) + 0..3 of sample 16 mt + (l & 15):
// row 16 w + 2 nt + (l >> 5), half (l >> 4) & 1.
template <bool ADDZ>
__device__ __forceinline__ void hwepilogue(f32x4a (&acc)[hw::NT][hw::MT], const float* bias, char* planes, int wave, int lane,
                                           unsigned* range_flag) {
    using namespace hw;
    const int fq = lane >> 4;
    const float* bl = bias + 16 * NT * wave + 4 * fq;
    char* base = planes + (2 * NT * wave + (fq >> 1)) * (2 * ROW_BYTES) + (lane & 15) * 16 + 8 * (fq & 1);
    float rmax = 0.f;   // f16-range guard (include/pnyolo.h pny_model_range_status)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float4 b = *reinterpret_cast<const float4*>(bl + 16 * nt);
        float2 za[MT], zb[MT];
        if (ADDZ) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const char* s0 = base + (2 * nt) * (2 * ROW_BYTES) + mt * 256;
                za[mt] = *reinterpret_cast<const float2*>(s0);
                zb[mt] = *reinterpret_cast<const float2*>(s0 + ROW_BYTES);
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            char* s0 = base + (2 * nt) * (2 * ROW_BYTES) + mt * 256;
            float x0 = acc[nt][mt][0] + b.x, x1 = acc[nt][mt][1] + b.y, x2 = acc[nt][mt][2] + b.z, x3 = acc[nt][mt][3] + b.w;
            if (ADDZ) {
                x0 += za[mt].x;
                x1 += za[mt].y;
                x2 += zb[mt].x;
                x3 += zb[mt].y;
            }
            acc[nt][mt] = f32x4a{x0, x1, x2, x3};
            const float r0 = relu1(x0), r1 = relu1(x1), r2 = relu1(x2), r3 = relu1(x3);
            rmax = fmaxf(fmaxf(rmax, fmaxf(r0, r1)), fmaxf(r2, r3));
            h4 p0, p1;
            split4(r0, r1, r2, r3, p0, p1);
            *reinterpret_cast<h4*>(s0) = p0;
            *reinterpret_cast<h4*>(s0 + ROW_BYTES) = p1;
        }
    }
    if (__builtin_expect(!(rmax < 65520.0f), 0)) range_report(range_flag, 1u);
}

// Cross-view running sum slab of the workgroup: tile t = nt * MT + mt of lane l at float4 index (wave * 32 + t) * 64 + l
struct HwSlab {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned lane_off;
};
__device__ __forceinline__ void hwslab_store(const f32x4a (&h)[hw::NT][hw::MT], const HwSlab& sl) {
    using namespace hw;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h[nt][mt]), sl.rsrc, sl.lane_off + (unsigned)((nt * MT + mt) * 64 * 16), 0, 2);
}
__device__ __forceinline__ void hwslab_load(f32x4a (&t)[hw::NT][hw::MT], const HwSlab& sl) {
    using namespace hw;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            t[nt][mt] = __builtin_bit_cast(f32x4a, __builtin_amdgcn_raw_buffer_load_b128(sl.rsrc, sl.lane_off, (unsigned)((nt * MT + mt) * 64 * 16), 2));
}

// Bilinear gather of the projected maps, staged in fp32 in the plane slots (see mlp_h2.hip h2gather_commit): a wave pass covers
// 8 samples x one 128-byte line (32 channels); a lane serves TWO samples (sample blocks 2 w and 2 w + 1), whose taps it re-reads
// from the tap table at every block (24 registers that would otherwise stay live across the GEMMs), and moves one line of
// both per piece: 8 tap loads of 16 bytes = 32 registers, pieces in flight as the caller chooses.
struct HwTaps {
    const float* t[hw::SPW][4];
    float w[hw::SPW][4];
};
__device__ __forceinline__ void hwgather_setup(HwTaps& g, const float* view_base, const float4* tap_tab, int wave, int lane) {
    const float* base = view_base + 4 * (lane >> 3);
#pragma unroll
    for (int i = 0; i < hw::SPW; ++i) {
        const int m = (hw::SPW * wave + i) * 8 + (lane & 7);
        const float4 o = tap_tab[2 * m], w = tap_tab[2 * m + 1];
        g.t[i][0] = base + __float_as_int(o.x);
        g.t[i][1] = base + __float_as_int(o.y);
        g.t[i][2] = base + __float_as_int(o.z);
        g.t[i][3] = base + __float_as_int(o.w);
        g.w[i][0] = w.x;
        g.w[i][1] = w.y;
        g.w[i][2] = w.z;
        g.w[i][3] = w.w;
    }
}
struct HwPiece {
    float4 x[hw::SPW][4];   // [sample][tap]
};
__device__ __forceinline__ void hwgather_issue(HwPiece& pc, const HwTaps& g, int c0) {
#pragma unroll
    for (int i = 0; i < hw::SPW; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) pc.x[i][k] = *reinterpret_cast<const float4*>(g.t[i][k] + c0);
}
// piece ln of the block (channels [32 ln, 32 ln + 32)): feature quad 8 ln + (lane >> 3) -> row 4 ln + (lane >> 4)
__device__ __forceinline__ void hwgather_commit(const HwPiece& pc, const HwTaps& g, char* planes, int ln, int wave, int lane) {
    using namespace hw;
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
        const int m = (SPW * wave + i) * 8 + (lane & 7);
        const float4(&x)[4] = pc.x[i];
        const float(&w)[4] = g.w[i];
        float2 lo, hi;
        lo.x = __builtin_fmaf(x[3].x, w[3], __builtin_fmaf(x[2].x, w[2], __builtin_fmaf(x[1].x, w[1], x[0].x * w[0])));
        lo.y = __builtin_fmaf(x[3].y, w[3], __builtin_fmaf(x[2].y, w[2], __builtin_fmaf(x[1].y, w[1], x[0].y * w[0])));
        hi.x = __builtin_fmaf(x[3].z, w[3], __builtin_fmaf(x[2].z, w[2], __builtin_fmaf(x[1].z, w[1], x[0].z * w[0])));
        hi.y = __builtin_fmaf(x[3].w, w[3], __builtin_fmaf(x[2].w, w[2], __builtin_fmaf(x[1].w, w[1], x[0].w * w[0])));
        char* s0 = planes + (4 * ln + (lane >> 4)) * (2 * ROW_BYTES) + m * 16 + 8 * ((lane >> 3) & 1);
        *reinterpret_cast<float2*>(s0) = lo;
        *reinterpret_cast<float2*>(s0 + ROW_BYTES) = hi;
    }
}

// per (view, tile) prologue (mlp_h2.hip h2prologue for 256 threads): lin_in's B operand as f16 planes in rows 0..7, tap table
__device__ __forceinline__ void hwprologue(const MlpArgs& a, int v, long long tile, char* planes, float4* tap_tab, int tid) {
    using namespace hw;
    constexpr int NPART = THREADS / TM;
    const int m = tid % TM, part = tid / TM;
    long long s = tile * TM + m;
    if (s >= a.n_points) s = a.n_points - 1;
    float p[3], d[3];
    load_point(a, s, p, d);
    const Cam cam = a.cams[tile_view_base(a, tile * TM) + v];
    float xr[3], xc[3], vd[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        xr[i] = cam.w2c[4 * i + 0] * p[0] + cam.w2c[4 * i + 1] * p[1] + cam.w2c[4 * i + 2] * p[2];
        xc[i] = xr[i] + cam.w2c[4 * i + 3];
        vd[i] = cam.w2c[4 * i + 0] * d[0] + cam.w2c[4 * i + 1] * d[1] + cam.w2c[4 * i + 2] * d[2];
    }
    float rmax = 0.f;
    for (int g = part; g < D_IN_PAD / 4; g += NPART) {
        h4 p0, p1;
        const float e0 = input_entry(4 * g + 0, xr, vd, a.freq_factor, a.num_freqs), e1 = input_entry(4 * g + 1, xr, vd, a.freq_factor, a.num_freqs);
        const float e2 = input_entry(4 * g + 2, xr, vd, a.freq_factor, a.num_freqs), e3 = input_entry(4 * g + 3, xr, vd, a.freq_factor, a.num_freqs);
        rmax = fmaxf(fmaxf(rmax, fmaxf(fabsf(e0), fabsf(e1))), fmaxf(fabsf(e2), fabsf(e3)));
        split4(e0, e1, e2, e3, p0, p1);
        char* s0 = planes + (g >> 1) * (2 * ROW_BYTES) + m * 16 + 8 * (g & 1);
        *reinterpret_cast<h4*>(s0) = p0;
        *reinterpret_cast<h4*>(s0 + ROW_BYTES) = p1;
    }
    if (__builtin_expect(!(rmax < 65520.0f), 0)) range_report(a.range_flag, 1u);
    if (part == NPART - 1) {
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
        const bool cull = a.yolo && !(xc[2] < 0.0f);
        int offs[4];
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
        tap_tab[2 * m] = make_float4(__int_as_float(offs[0]), __int_as_float(offs[1]), __int_as_float(offs[2]), __int_as_float(offs[3]));
        tap_tab[2 * m + 1] = make_float4(wgt[0], wgt[1], wgt[2], wgt[3]);
    }
}

__global__ __launch_bounds__(hw::THREADS) __attribute__((amdgpu_waves_per_eu(hw::NW / 4, hw::NW / 4))) void PNY_HW_KERNEL(const MlpArgs a) {
    using namespace hw;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* planes = smem_raw;
    float4* tap_tab = reinterpret_cast<float4*>(smem_raw + ACT_BYTES);
    float* bias_tab = reinterpret_cast<float*>(smem_raw + ACT_BYTES + TAP_BYTES);   // [b_in, b_fc0[0], b_fc1[0], b_fc0[1], ...][512]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const HwSlab slab = {__builtin_amdgcn_make_buffer_rsrc(a.scratch + (size_t)blockIdx.x * (TM * HID), 0, TM * HID * 4, 0x00020000),
                         (unsigned)((wave * (NT * MT * 64) + lane) * 16)};
    const int nb = a.n_blocks;
    const int nvb = a.combine_layer < nb ? a.combine_layer : nb;
    const WStream ws = wstream(a, lane);
    const HwSeg s_in = hwseg(ws, a.h3_in, D_IN_PAD / 32, wave);
    auto fc0seg = [&](int b) { return hwseg(ws, a.h3_fc0[b], HID / 32, wave); };
    auto fc1seg = [&](int b) { return hwseg(ws, a.h3_fc1[b], HID / 32, wave); };
    auto entry_bias = [&](int b) -> const float* { return bias_tab + (b == 0 ? 0 : 2 * b) * HID; };
    auto fc0_bias = [&](int b) -> const float* { return bias_tab + (1 + 2 * b) * HID; };
    HwRing ring;
    hwring_fill(ring, ws, s_in);
#ifdef PNY_H2_STAMP
    unsigned long long hs_acc[HS_N], hs_t_ = 0;
    for (int i = 0; i < HS_N; ++i) hs_acc[i] = 0;
    const unsigned long long hs_start = hwnow();
    const unsigned long long hs_real0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int i = tid; i < (1 + 2 * nb) * HID; i += THREADS) {
        const int vec = i / HID, f = i % HID;
        const float* src = vec == 0 ? a.w.b_in : ((vec & 1) ? a.w.b_fc0[(vec - 1) >> 1] : a.w.b_fc1[(vec - 2) >> 1]);
        bias_tab[i] = src[f];
    }

    const bool xcd_order = (gridDim.x & 7) == 0;   // see mlp.hip
    const long long t_chunk = xcd_order ? (a.n_tiles + 7) / 8 : a.n_tiles;
    const long long t_first = xcd_order ? (long long)(blockIdx.x & 7) * t_chunk + (blockIdx.x >> 3) : blockIdx.x;
    const long long t_last = xcd_order ? ((long long)((blockIdx.x & 7) + 1) * t_chunk < a.n_tiles
                                              ? (long long)((blockIdx.x & 7) + 1) * t_chunk : (long long)a.n_tiles)
                                       : (long long)a.n_tiles;
    const int t_step = xcd_order ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    for (long long tile = t_first; tile < t_last; tile += t_step) {
        f32x4a h[NT][MT];
        f32x4a net[NT][MT];
        // one residual block from "planes hold relu(h_in)" on: net = fc_0(.), h += fc_1(relu(net + b_fc0))
        // PREFETCH (8-wave shape): the first PNY_HW_GD pieces of the NEXT block's projection are fetched underneath the fc_1 GEMM, in
        // the registers of the then-dead `net` accumulators (16 registers a piece); the tap pointers are set up per view.
        constexpr bool PREFETCH = NW == 8 && PNY_HW_PF > 0;
        HwPiece pc[PNY_HW_GD];
        const float* zp_view = a.zp;   // (set per view below)
        auto block_tail = [&](int blk, const HwSeg& after, bool slab_in, int next_cb = -1) {
            HS_T0();
            hwzero(net);
            __syncthreads();
            HS_LAP(HS_EPI_WAIT);
            hwgemm(net, ring, ws, fc0seg(blk), fc1seg(blk), planes, lane);
            HS_LAP(HS_GEMM);
            __syncthreads();
            HS_LAP(HS_EPI_WAIT);
            hwepilogue<false>(net, fc0_bias(blk), planes, wave, lane, a.range_flag);
            if (slab_in) {
                hwslab_load(net, slab);   // the other views' running sum, in the registers of the now dead `net`
                HS_LAP(HS_EPI);
                __syncthreads();
                HS_LAP(HS_EPI_WAIT);
                hwgemm(h, ring, ws, fc1seg(blk), after, planes, lane);
                HS_LAP(HS_GEMM);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) h[nt][mt] = net[nt][mt] + h[nt][mt];
            } else if (PREFETCH && next_cb >= 0) {
                HS_LAP(HS_EPI);
                __syncthreads();
                HS_LAP(HS_EPI_WAIT);
                hwgemm(h, ring, ws, fc1seg(blk), after, planes, lane, [&]() {
                    HwTaps gp;   // (the tap pointers are re-read from the table here and again in the gather phase: kept across
                                 // the GEMMs they cost 12 registers where the loops have none to spare)
                    hwgather_setup(gp, zp_view, tap_tab, wave, lane);
#pragma unroll
                    for (int i = 0; i < PNY_HW_PF; ++i) hwgather_issue(pc[i], gp, next_cb + 32 * i);
                });
                HS_LAP(HS_GEMM);
            } else {
                HS_LAP(HS_EPI);
                __syncthreads();
                HS_LAP(HS_EPI_WAIT);
                hwgemm(h, ring, ws, fc1seg(blk), after, planes, lane);
                HS_LAP(HS_GEMM);
            }
        };
        for (int v = 0; v < a.NS; ++v) {
            const HwSeg after_view = v + 1 < a.NS ? s_in : (nvb < nb ? fc0seg(nvb) : s_in);
            HS_T0();
            __syncthreads();
            hwprologue(a, v, tile, planes, tap_tab, tid);
            hwzero(h);
            __syncthreads();
            zp_view = a.zp + (size_t)(tile_view_base(a, tile * hw::TM) + v) * a.Hl * a.Wl * a.zp_stride;
            if (PREFETCH) {   // block 0's first pieces travel underneath the lin_in GEMM
                HwTaps gp;
                hwgather_setup(gp, zp_view, tap_tab, wave, lane);
#pragma unroll
                for (int i = 0; i < PNY_HW_PF; ++i) hwgather_issue(pc[i], gp, 32 * i);
            }
            HS_LAP(HS_PROLOGUE);
            hwgemm(h, ring, ws, s_in, fc0seg(0), planes, lane);
            HS_LAP(HS_GEMM);
            for (int blk = 0; blk < nvb; ++blk) {
                // h += interp(lin_z[blk](latent map)): the block's 512 projected channels in 16 pieces of 32 (one 128-byte line
                // per tap and sample), PNY_HW_GD in flight
                const int cb = blk * HID;
                HS_T0();
                __syncthreads();  // every wave is done reading the planes (previous GEMM)
                HS_LAP(HS_GATHER_WAIT);
                HwTaps g;
                hwgather_setup(g, zp_view, tap_tab, wave, lane);
                if (PREFETCH) {
#pragma unroll
                    for (int i = PNY_HW_PF; i < PNY_HW_GD; ++i) hwgather_issue(pc[i], g, cb + 32 * i);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ln = 0; ln < 16; ++ln) {   // pieces 0 .. GD - 1 are in flight: commit, then refill the slot
                        hwgather_commit(pc[ln % PNY_HW_GD], g, planes, ln, wave, lane);
                        if (ln + PNY_HW_GD < 16) hwgather_issue(pc[ln % PNY_HW_GD], g, cb + 32 * (ln + PNY_HW_GD));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < PNY_HW_GD - 1; ++i) hwgather_issue(pc[i], g, cb + 32 * i);
#pragma unroll
                    for (int ln = 0; ln < 16; ++ln) {
                        if (ln + PNY_HW_GD - 1 < 16) hwgather_issue(pc[(ln + PNY_HW_GD - 1) % PNY_HW_GD], g, cb + 32 * (ln + PNY_HW_GD - 1));
                        __builtin_amdgcn_sched_barrier(0);
                        hwgather_commit(pc[ln % PNY_HW_GD], g, planes, ln, wave, lane);
                    }
                }
                HS_LAP(HS_GATHER);
                __syncthreads();  // projection visible
                HS_LAP(HS_GATHER_WAIT);
                hwepilogue<true>(h, entry_bias(blk), planes, wave, lane, a.range_flag);
                HS_LAP(HS_EPI);
                const bool last = blk + 1 == nvb;
                block_tail(blk, last ? after_view : fc0seg(blk + 1), last && v > 0, last ? -1 : cb + HID);
            }
            if (a.NS > 1) {
                if (v + 1 < a.NS) {
                    hwslab_store(h, slab);
                } else {
                    const float rns = 1.0f / (float)a.NS;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) h[nt][mt] = h[nt][mt] * rns;
                }
            }
        }
        for (int blk = nvb; blk < nb; ++blk) {
            HS_T0();
            __syncthreads();
            HS_LAP(HS_EPI_WAIT);
            hwepilogue<false>(h, entry_bias(blk), planes, wave, lane, a.range_flag);
            HS_LAP(HS_EPI);
            block_tail(blk, blk + 1 < nb ? fc0seg(blk + 1) : s_in, false);
        }
        // out = lin_out(relu(h + b_fc1[last])) (reference resnetfc.py:185) + output head (models.py:312-317)
        HS_T0();
        __syncthreads();
        hwepilogue<false>(h, entry_bias(nb), planes, wave, lane, a.range_flag);
        __syncthreads();
        for (int idx = tid; idx < a.d_out * TM; idx += THREADS) {
            const int o = idx / TM, m = idx % TM;
            const float4* wrow = reinterpret_cast<const float4*>(a.w.w_out + (size_t)o * HID);
            float sum = 0.f;
#pragma unroll 4
            for (int kg = 0; kg < HID / 8; ++kg) {
                const h8 x0 = *reinterpret_cast<const h8*>(planes + kg * (2 * ROW_BYTES) + m * 16);
                const h8 x1 = *reinterpret_cast<const h8*>(planes + kg * (2 * ROW_BYTES) + ROW_BYTES + m * 16);
                const float4 wa = wrow[2 * kg], wb = wrow[2 * kg + 1];
                sum += ((float)x0[0] + (float)x1[0]) * wa.x;
                sum += ((float)x0[1] + (float)x1[1]) * wa.y;
                sum += ((float)x0[2] + (float)x1[2]) * wa.z;
                sum += ((float)x0[3] + (float)x1[3]) * wa.w;
                sum += ((float)x0[4] + (float)x1[4]) * wb.x;
                sum += ((float)x0[5] + (float)x1[5]) * wb.y;
                sum += ((float)x0[6] + (float)x1[6]) * wb.z;
                sum += ((float)x0[7] + (float)x1[7]) * wb.w;
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
        HS_LAP(HS_LINOUT);
    }
#ifdef PNY_H2_STAMP
    hs_acc[HS_TOTAL] = hwnow() - hs_start;
    hs_acc[HS_REAL] = __builtin_amdgcn_s_memrealtime() - hs_real0;
    if (lane == 0)
        for (int i = 0; i < HS_N; ++i) PNY_HW_STAMPBUF[((size_t)blockIdx.x * NW + wave) * HS_N + i] = hs_acc[i];
#endif
}

#if PNY_HW_NW == 4
bool mlp_h2w_supports(int n_blocks, int combine_layer) { return n_blocks <= hw::MAX_NB && combine_layer >= 1; }
#endif

void PNY_HW_LAUNCH(const MlpArgs& a, int grid, hipStream_t st) {
    static bool attr_set[64] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    dev_ &= 63;
    if (!attr_set[dev_]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(PNY_HW_KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  hw::lds_bytes(hw::MAX_NB));
        attr_set[dev_] = true;
    }
#ifdef PNY_H2_STAMP
    static unsigned long long* dbuf = nullptr;
    const size_t nst = (size_t)grid * hw::NW * HS_N;
    if (!dbuf) {
        (void)hipMalloc((void**)&dbuf, (size_t)1024 * 8 * HS_N * sizeof(unsigned long long));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(PNY_HW_STAMPBUF), &dbuf, sizeof(dbuf));
    }
    (void)hipMemsetAsync(dbuf, 0, nst * sizeof(unsigned long long), st);
#endif
    hipLaunchKernelGGL(PNY_HW_KERNEL, dim3(grid), dim3(hw::THREADS), hw::lds_bytes(a.n_blocks), st, a);
#ifdef PNY_H2_STAMP
    {
        std::vector<unsigned long long> hst(nst);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hst.data(), dbuf, nst * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sum[HS_N] = {0};
        for (size_t i = 0; i < nst; ++i) sum[i % HS_N] += (double)hst[i];
        static const char* names[HS_N] = {"total", "gemm", "gather-barrier-wait", "gather", "epilogue-barrier-wait", "epilogue", "prologue", "lin_out", "slab", "realtime"};
        fprintf(stderr, "[h2w stamp, %d waves] tiles=%d grid=%d:", hw::NW, a.n_tiles, grid);
        for (int i = 0; i < HS_N; ++i) fprintf(stderr, " %s=%.1f%%", names[i], 100.0 * sum[i] / sum[0]);
        fprintf(stderr, " (mean wave cycles %.4g; in-kernel clock %.3f GHz)\n", sum[0] / ((double)grid * hw::NW), sum[0] / sum[HS_REAL] * 0.1);
    }
#endif
}

}  // namespace pny
