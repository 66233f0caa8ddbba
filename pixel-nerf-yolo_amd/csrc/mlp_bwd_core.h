// Device helpers shared by the two backward chain kernels (mlp_bwd.hip: fp32 MFMA; mlp_bwd_h2.hip: split f16): the transposed
// weight stream, stash records addressed through buffer resources, accumulator-layout loads and relu masks.
#pragma once
#include "mlp_core.h"

namespace pny {

__device__ __forceinline__ WStream wstream_raw(const float* base, unsigned bytes, int lane) {
    WStream w;
    w.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
    w.base = reinterpret_cast<const char*>(base);
    w.lane_off = 16u * (unsigned)lane;
    return w;
}

template <int NT, int MT>
__device__ __forceinline__ void acc_zero(f32x16 (&t)[NT][MT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) t[nt][mt][r] = 0.f;
}

// The two stashes of a tile are addressed through raw buffer resources (one per tile record, 32-bit byte offsets): a slot
// access in accumulator layout is the lane's own offset (one VGPR for the whole kernel) + the slot's offset + a
// compile-time constant per accumulator quad.  With 64-bit pointers every quad had its own address pair, and those spilled
// inside the GEMM loops.  Stores keep the whole offset in the VGPR operand and the constant 0 in soffset (see mlp_h2.hip
// stash_store: the form for which the compiler inserts the >64-bit store-data wait state).
struct StashRef {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned off;   // byte offset of the slot inside the record
};
__device__ __forceinline__ float4 stash_ld(const StashRef& r, unsigned lane_off, unsigned c) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r.rsrc, lane_off, r.off + c, 0);
    const f32x4 f = __builtin_bit_cast(f32x4, v);
    return make_float4(f.x, f.y, f.z, f.w);
}
__device__ __forceinline__ void stash_st(const StashRef& r, unsigned lane_off, unsigned c, const float4& v) {
    const f32x4 f = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f), r.rsrc, lane_off + r.off + c, 0, 0);
}
// byte offset of accumulator quad (nt, mt, q) relative to the lane's own offset acc_lane_off()
template <int NT, int MT>
__device__ __forceinline__ constexpr unsigned quad_off(int nt, int mt, int q) {
    return (unsigned)(((8 * nt + 2 * q) * (32 * MT) + 32 * mt) * 16);
}
template <int NT, int MT>
__device__ __forceinline__ unsigned acc_lane_off(int wave, int lane) {
    return (unsigned)(((8 * NT * wave + (lane >> 5)) * (32 * MT) + (lane & 31)) * 16);
}

// acc = (x > 0) ? acc : 0 with x = the stashed relu'd activation of the same element (accumulator layout).
template <int NT, int MT, class Acc>   // Acc = f32x16 (an MFMA accumulator tile) or float[16]
__device__ __forceinline__ void mask_by(Acc (&acc)[NT][MT], const StashRef& x, int wave, int lane) {
    const unsigned lo = acc_lane_off<NT, MT>(wave, lane);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float4 xv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xv[q] = stash_ld(x, lo, quad_off<NT, MT>(nt, mt, q));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[nt][mt][4 * q + 0] = xv[q].x > 0.f ? acc[nt][mt][4 * q + 0] : 0.f;
                acc[nt][mt][4 * q + 1] = xv[q].y > 0.f ? acc[nt][mt][4 * q + 1] : 0.f;
                acc[nt][mt][4 * q + 2] = xv[q].z > 0.f ? acc[nt][mt][4 * q + 2] : 0.f;
                acc[nt][mt][4 * q + 3] = xv[q].w > 0.f ? acc[nt][mt][4 * q + 3] : 0.f;
            }
        }
}

template <int NT, int MT, class Acc>
__device__ __forceinline__ void acc_load(Acc (&acc)[NT][MT], const StashRef& g, int wave, int lane) {
    const unsigned lo = acc_lane_off<NT, MT>(wave, lane);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = stash_ld(g, lo, quad_off<NT, MT>(nt, mt, q));
                acc[nt][mt][4 * q + 0] = v.x;
                acc[nt][mt][4 * q + 1] = v.y;
                acc[nt][mt][4 * q + 2] = v.z;
                acc[nt][mt][4 * q + 3] = v.w;
            }
}

}  // namespace pny
