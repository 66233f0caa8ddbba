// Counter-based random streams of the renderer (perf mode: draws generated in-kernel from a seed instead of being
// passed in).  Shared by the forward sampling kernels (render_kernels.hip) and the backward pass, which must
// reproduce the forward's depth-sample draws (mlp_bwd.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pny {

// ------------------------------------------------------------------ counter-based RNG (perf mode)
// Philox4x32-10 (Salmon et al., SC'11), written out from the published round function.
struct Philox {
    uint32_t key[2];
    __device__ Philox(uint64_t seed) {
        key[0] = (uint32_t)seed;
        key[1] = (uint32_t)(seed >> 32);
    }
    __device__ void draw(uint64_t index, uint32_t stream, uint32_t (&out)[4]) const {
        uint32_t c[4] = {(uint32_t)index, (uint32_t)(index >> 32), stream, 0x9E3779B9u};
        uint32_t k0 = key[0], k1 = key[1];
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
            const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
            const uint32_t n1 = (uint32_t)p1;
            const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
            const uint32_t n3 = (uint32_t)p0;
            c[0] = n0;
            c[1] = n1;
            c[2] = n2;
            c[3] = n3;
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        out[0] = c[0];
        out[1] = c[1];
        out[2] = c[2];
        out[3] = c[3];
    }
};
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }  // [0,1)
__device__ __forceinline__ float uniform_at(uint64_t seed, uint32_t stream, uint64_t idx) {
    uint32_t r[4];
    Philox(seed).draw(idx >> 2, stream, r);
    return u01(r[idx & 3]);
}
__device__ __forceinline__ float normal_at(uint64_t seed, uint32_t stream, uint64_t idx) {
    uint32_t r[4];
    Philox(seed).draw(idx >> 1, stream, r);
    const float u1 = 1.0f - u01(r[2 * (idx & 1)]);  // (0,1]
    const float u2 = u01(r[2 * (idx & 1) + 1]);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}
enum { STREAM_COARSE = 1, STREAM_FINE = 2, STREAM_FINE2 = 3, STREAM_DEPTH = 4 };


}  // namespace pny
