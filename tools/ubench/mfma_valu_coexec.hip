// Microbenchmark (MI355X): can fp32 MFMA (v_mfma_f32_32x32x2_f32) and fp32 VALU FMA sustain their rates
// at the same time on one SIMD?  Three kernels with the same grid: MFMA-only waves, VALU-only waves,
// and a mix (half the waves of each workgroup do MFMA, half do VALU FMAs).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_coexec.hip -o gpurun_out/coexec && ./coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>  // 0 = all waves MFMA, 1 = all waves VALU, 2 = waves 0-3 MFMA / 4-7 VALU, 3 = mix with 8 MFMA + 8 VALU waves
__global__ __launch_bounds__(1024) void k(float* out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    bool do_mfma = MODE == 0 || (MODE >= 2 && wave < nw / 2);
    float r = 0.f;
    if (do_mfma) {
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        float x = seed + threadIdx.x * 1e-3f, y = seed * 0.5f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
            }
        }
        for (int j = 0; j < 16; ++j) r += a0[j] + a1[j] + a2[j] + a3[j];
    } else {
        float acc[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j] = seed * j;
        float x = seed + threadIdx.x * 1e-3f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int j = 0; j < 32; ++j) acc[j] = __builtin_fmaf(acc[j], x, seed);  // 32 independent chains
            }
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) r += acc[j];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
double run(const char* name, int threads, int iters, double flop_mfma_per_wave_iter, double flop_valu_per_wave_iter) {
    float* out;
    hipMalloc(&out, 256 * 8 * 1024 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256;
    k<MODE><<<grid, threads>>>(out, iters, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, threads>>>(out, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const int nw = threads / 64;
    int wm = MODE == 0 ? nw : MODE == 1 ? 0 : nw / 2, wv = nw - wm;
    const double fm = (double)grid * wm * iters * flop_mfma_per_wave_iter, fv = (double)grid * wv * iters * flop_valu_per_wave_iter;
    printf("%-34s threads=%4d  %.3f ms   MFMA %.1f TF/s  VALU %.1f TF/s  total %.1f TF/s\n", name, threads, ms, fm / ms / 1e9,
           fv / ms / 1e9, (fm + fv) / ms / 1e9);
    hipFree(out);
    return ms;
}

int main() {
    const int iters = 4000;
    const double fm = 16.0 * 32 * 32 * 2 * 2;      // 16 MFMAs of 32x32x2 per iteration
    const double fv = 8.0 * 32 * 64 * 2;           // 8*32 wave-wide FMAs per iteration
    run<0>("MFMA only, 4 waves/CU", 256, iters, fm, fv);
    run<0>("MFMA only, 8 waves/CU", 512, iters, fm, fv);
    run<1>("VALU only, 4 waves/CU", 256, iters, fm, fv);
    run<1>("VALU only, 8 waves/CU", 512, iters, fm, fv);
    run<1>("VALU only, 16 waves/CU", 1024, iters, fm, fv);
    run<2>("mix: 4 MFMA + 4 VALU waves/CU", 512, iters, fm, fv);
    run<2>("mix: 8 MFMA + 8 VALU waves/CU", 1024, iters, fm, fv);
    return 0;
}
