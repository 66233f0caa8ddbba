// Numerical feasibility of fp32 GEMMs on the f16 matrix cores by operand splitting (gfx950):
//   x = x1 + x2, w = w1 + w2 with x1 = f16(x), x2 = f16(x - x1) (22 significant bits, denormal second planes),
//   x w ~ x1 w1 + x1 w2 + x2 w1   (three v_mfma_f32_32x32x16_f16 with fp32 accumulation per 16 k)
// against v_mfma_f32_32x32x2_f32 (the exact-fp32 path) and an fp64 host reference, on data with the magnitudes of the
// ResnetFC layers (x = relu(N(0,1)), w = N(0, sqrt(2/512)), K = 512).  Build: hipcc --offload-arch=gfx950 -O2 -o split_f16_check.bin split_f16_check.hip
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int M = 32, N = 32, K = 512;

__global__ void k_f32(const float* A, const float* B, float* D) {  // A (M,K) row-major, B (K,N) row-major
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 acc = {0};
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * N + r], acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * N + r] = acc[i];
}

__device__ inline void split(float x, _Float16& a, _Float16& b, int rtz) {
    if (rtz) {
        a = (_Float16)__builtin_amdgcn_cvt_pkrtz(x, 0.f)[0];
        const float rem = x - (float)a;
        b = (_Float16)__builtin_amdgcn_cvt_pkrtz(rem, 0.f)[0];
    } else {
        a = (_Float16)x;
        b = (_Float16)(x - (float)a);
    }
}

__global__ void k_split(const float* A, const float* B, float* D, int rtz, int nprod) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 acc = {0};
    for (int k = 0; k < K; k += 16) {
        h8 a1, a2, b1, b2;
        for (int j = 0; j < 8; ++j) {
            _Float16 p, q;
            split(A[r * K + k + 8 * h + j], p, q, rtz);
            a1[j] = p;
            a2[j] = q;
            split(B[(k + 8 * h + j) * N + r], p, q, rtz);
            b1[j] = p;
            b2[j] = q;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, acc, 0, 0, 0);
        if (nprod > 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b2, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * N + r] = acc[i];
}

int main() {
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int scen = 0; scen < 3; ++scen) {
        // scen 0: weights as A (w ~ N(0, 0.0625)), relu activations as B; 1: activations x 0.01 (small values); 2: x 50
        const float xs = scen == 0 ? 1.f : (scen == 1 ? 0.01f : 50.f);
        std::vector<float> A(M * K), B(K * N), D0(M * N), D1(M * N), D2(M * N), D3(M * N);
        for (auto& v : A) v = nd(rng) * 0.0625f;
        for (auto& v : B) v = std::fmax(nd(rng), 0.f) * xs;
        float *dA, *dB, *dD;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D0.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        auto run = [&](int which, std::vector<float>& out) {
            if (which == 0) hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, dA, dB, dD);
            if (which == 1) hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0, 3);
            if (which == 2) hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, dD, 1, 3);
            if (which == 3) hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0, 4);
            hipDeviceSynchronize();
            hipMemcpy(out.data(), dD, out.size() * 4, hipMemcpyDeviceToHost);
        };
        run(0, D0); run(1, D1); run(2, D2); run(3, D3);
        double e[4] = {0, 0, 0, 0}, mx = 0, sabs = 0;
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < N; ++j) {
                double ref = 0, sa = 0;
                for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * N + j]; sa += std::fabs((double)A[i * K + k] * B[k * N + j]); }
                mx = std::fmax(mx, std::fabs(ref)); sabs = std::fmax(sabs, sa);
                e[0] = std::fmax(e[0], std::fabs(D0[i * N + j] - ref));
                e[1] = std::fmax(e[1], std::fabs(D1[i * N + j] - ref));
                e[2] = std::fmax(e[2], std::fabs(D2[i * N + j] - ref));
                e[3] = std::fmax(e[3], std::fabs(D3[i * N + j] - ref));
            }
        printf("scenario %d (x scale %g): max|ref| %.3g, max sum|ab| %.3g | max abs err: f32 mfma %.3e | f16 split rtn 3-prod %.3e | rtz 3-prod %.3e | rtn 4-prod %.3e\n",
               scen, xs, mx, sabs, e[0], e[1], e[2], e[3]);
        hipFree(dA); hipFree(dB); hipFree(dD);
    }
    return 0;
}
