// Ray / sampling / compositing kernels of the renderer (gfx950).  These are HBM- and
// latency-trivial next to the MLP (~1 % of the reference's time); they exist so
// that a render call never leaves the device and never launches an ATen op.
#include "pny_common.h"
#include "pny_rng.h"

namespace pny {

// ------------------------------------------------------------------ sample_coarse
// reference nerf.py:104-121: t = linspace(0, 1-1/Kc, Kc)[k] + u/Kc ; z = near(1-t) + far t
__device__ __forceinline__ float lerp_depth(float near, float far, float t, int lindisp) {
    if (!lindisp) return near * (1.0f - t) + far * t;
    return 1.0f / (1.0f / near * (1.0f - t) + 1.0f / far * t);
}

__global__ void sample_coarse_kernel(const float* __restrict__ rays, long long n, int kc, int lindisp,
                                     const float* __restrict__ u, uint64_t seed, float step, float end,
                                     float* __restrict__ z) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * kc) return;
    const long long ray = i / kc;
    const int k = (int)(i - ray * kc);
    // torch.linspace(0, 1-step, Kc): start + k * ((end - start)/(Kc-1)), symmetric about the middle;
    // step and end are the reference's Python doubles rounded to fp32 on the host
    const float inc = kc > 1 ? end / (float)(kc - 1) : 0.f;
    const float lin = (k < kc / 2) ? (float)k * inc : end - (float)(kc - 1 - k) * inc;
    const float uu = u ? u[i] : uniform_at(seed, STREAM_COARSE, (uint64_t)i);
    const float t = lin + uu * step;
    z[i] = lerp_depth(rays[ray * 8 + 6], rays[ray * 8 + 7], t, lindisp);
}

void launch_sample_coarse(const float* rays, long long n, int kc, int lindisp, const float* u, uint64_t seed,
                          float* z, hipStream_t st) {
    const long long tot = n * kc;
    if (tot == 0) return;
    const double dstep = 1.0 / (double)kc;
    hipLaunchKernelGGL(sample_coarse_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, rays, n, kc,
                       lindisp, u, seed, (float)dstep, (float)(1.0 - dstep), z);
}

// ------------------------------------------------------------------ composite
// reference nerf.py:184-188, 229-250.  One wavefront per ray, one sample per lane (chunks of 64
// with a carried transmittance): alpha per lane, transmittance = exclusive prefix product by a
// log-step wavefront scan, weighted sums by butterfly reduction.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// noise (optional, (n, K)): the training-time draw added to sigma before the relu (reference nerf.py:231-232), already
// scaled by noise_std.
__global__ __launch_bounds__(256) void composite_kernel(const float* __restrict__ rays, const float* __restrict__ z,
                                                        const float* __restrict__ samp, const float* __restrict__ noise,
                                                        long long n, int K, int white, float* __restrict__ wout,
                                                        float* __restrict__ rgb, float* __restrict__ depth) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n) return;
    const float far = rays[ray * 8 + 7];
    const float* zr = z + ray * K;
    const float4* sr = reinterpret_cast<const float4*>(samp) + ray * K;
    float carry = 1.0f;  // T before the first sample of the chunk
    float ar = 0.f, ag = 0.f, ab = 0.f, ad = 0.f, aw = 0.f;
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + lane;
        const bool on = k < K;
        float alpha = 0.f, zk = 0.f;
        float4 s = {0.f, 0.f, 0.f, 0.f};
        if (on) {
            zk = zr[k];
            const float znext = (k + 1 < K) ? zr[k + 1] : far;
            s = sr[k];
            const float delta = znext - zk;
            const float sg = noise ? s.w + noise[ray * K + k] : s.w;
            alpha = 1.0f - expf(-delta * fmaxf(sg, 0.f));
        }
        // inclusive product scan of (1 - alpha + 1e-10)
        float pprod = on ? (1.0f - alpha + 1e-10f) : 1.0f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float up = __shfl_up(pprod, o, 64);
            if (lane >= o) pprod *= up;
        }
        float excl = __shfl_up(pprod, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float T = carry * excl;
        const float w = alpha * T;
        if (on && wout) wout[ray * K + k] = w;
        ar += w * s.x;
        ag += w * s.y;
        ab += w * s.z;
        ad += w * zk;
        aw += w;
        carry *= __shfl(pprod, 63, 64);
    }
    ar = wave_sum(ar);
    ag = wave_sum(ag);
    ab = wave_sum(ab);
    ad = wave_sum(ad);
    aw = wave_sum(aw);
    if (lane == 0) {
        if (white) {  // rgb + 1 - sum(w), reference nerf.py:247-250
            ar = ar + 1.0f - aw;
            ag = ag + 1.0f - aw;
            ab = ab + 1.0f - aw;
        }
        if (rgb) {
            rgb[ray * 3 + 0] = ar;
            rgb[ray * 3 + 1] = ag;
            rgb[ray * 3 + 2] = ab;
        }
        if (depth) depth[ray] = ad;
    }
}

void launch_composite(const float* rays, const float* z, const float* samp, long long n, int k, int white,
                      float* w, float* rgb, float* depth, hipStream_t st, const float* noise) {
    if (n == 0) return;
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, rays, z, samp, noise, n, k, white,
                       w, rgb, depth);
}

// Fine-pass sample selection (reference nerf.py:126-167, 291-301): importance samples from the coarse weights'
// cdf, depth samples around the coarse depth, concatenation with the coarse depths and an ascending sort.
// One WAVEFRONT per ray (lane = sample index): the per-ray rows are loaded and stored coalesced and live in LDS.
// The arithmetic that decides bins keeps a fixed sequential order -- total = ((w0 + w1) + w2) + ..., each
// normalised weight divided once, cdf as a running sum in index order (every lane accumulates the same
// LDS-broadcast values redundantly) -- because bin selection (searchsorted on the cdf) is discontinuous
// (hard part 5); the draws are searched in parallel (one binary search per lane) and the sort is a
// rank computation: a coarse depth moves up by the number of new depths below it, a new depth lands after the
// coarse depths <= it and the new depths before it (the coarse depths are ascending: one sample per stratum).
// An earlier one-ray-per-lane version took 0.29 ms regardless of the ray count (strided row accesses and a serial
// insertion sort through LDS); this one is ~10x faster and, for training-size batches, off the critical path.
constexpr int FINE_WAVES = 4;  // rays per workgroup

__device__ __forceinline__ void fine_wave_sync() {  // LDS traffic of ONE wavefront: order it, no s_barrier needed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(64 * FINE_WAVES) void sample_fine_kernel(
    const float* __restrict__ rays, const float* __restrict__ zc, const float* __restrict__ wts,
    const float* __restrict__ depth, long long n, int kc, int kf, int kfd, float depth_std, int lindisp,
    const float* __restrict__ u, const float* __restrict__ u2, const float* __restrict__ g, uint64_t seed,
    float* __restrict__ zout) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long ray = (long long)blockIdx.x * FINE_WAVES + wv;  // wave-uniform
    if (ray >= n) return;
    const int ktot = kc + kf, kimp = kf - kfd;
    float* cdf = lds + (size_t)wv * (3 * kc + 1 + kf + ktot);  // [kc + 1]
    float* q = cdf + kc + 1;                                   // [kc] weight + 1e-5, then normalised
    float* zcs = q + kc;                                       // [kc] coarse depths (ascending)
    float* zn = zcs + kc;                                      // [kf] new depths
    float* zo = zn + kf;                                       // [ktot] merged
    const float near = rays[ray * 8 + 6], far = rays[ray * 8 + 7];

    for (int k = lane; k < kc; k += 64) {
        q[k] = wts[ray * kc + k] + 1e-5f;
        zcs[k] = zc[ray * kc + k];
    }
    fine_wave_sync();
    // pdf / cdf: weights + 1e-5, normalised, running sum with a leading 0 (nerf.py:136-139)
    float tot = 0.f;
    for (int k = 0; k < kc; ++k) tot += q[k];
    fine_wave_sync();
    for (int k = lane; k < kc; k += 64) q[k] = q[k] / tot;
    fine_wave_sync();
    float run = 0.f;
    if (lane == 0) cdf[0] = 0.f;
    for (int k = 0; k < kc; ++k) {
        run += q[k];
        if (lane == (k & 63)) cdf[k + 1] = run;
    }
    fine_wave_sync();
    // importance samples (nerf.py:141-153)
    for (int i = lane; i < kimp; i += 64) {
        const long long idx = ray * kimp + i;
        const float uu = u ? u[idx] : uniform_at(seed, STREAM_FINE, (uint64_t)idx);
        const float vv = u2 ? u2[idx] : uniform_at(seed, STREAM_FINE2, (uint64_t)idx);
        // searchsorted(cdf, u, right=True): first index with cdf[index] > u, over kc+1 entries
        int lo = 0, hi = kc + 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] > uu)
                hi = mid;
            else
                lo = mid + 1;
        }
        float ind = (float)lo - 1.0f;
        ind = fmaxf(ind, 0.0f);
        const float t = (ind + vv) / (float)kc;
        zn[i] = lerp_depth(near, far, t, lindisp);
    }
    // depth samples (nerf.py:163-166): clamp(depth + g*std, near, far)
    for (int i = lane; i < kfd; i += 64) {
        const long long idx = ray * kfd + i;
        const float gg = g ? g[idx] : normal_at(seed, STREAM_DEPTH, (uint64_t)idx);
        float zz = depth[ray] + gg * depth_std;
        zz = fmaxf(fminf(zz, far), near);
        zn[kimp + i] = zz;
    }
    fine_wave_sync();
    // sort (nerf.py:301) by rank
    for (int e = lane; e < ktot; e += 64) {
        float v;
        int r;
        if (e < kc) {
            v = zcs[e];
            r = e;
            for (int j = 0; j < kf; ++j) r += (zn[j] < v) ? 1 : 0;
        } else {
            const int j0 = e - kc;
            v = zn[j0];
            int lo = 0, hi = kc;  // number of coarse depths <= v
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (zcs[mid] <= v)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            r = lo;
            for (int j = 0; j < kf; ++j) {
                const float o = zn[j];
                r += (o < v || (o == v && j < j0)) ? 1 : 0;
            }
        }
        zo[r] = v;
    }
    fine_wave_sync();
    for (int e = lane; e < ktot; e += 64) zout[ray * ktot + e] = zo[e];
}

void launch_sample_fine(const float* rays, const float* zc, const float* w, const float* depth, long long n, int kc,
                        int kf, int kfd, float depth_std, int lindisp, const float* u, const float* u2,
                        const float* g, uint64_t seed, float* zout, hipStream_t st) {
    if (n == 0) return;
    const size_t lds = (size_t)FINE_WAVES * (3 * kc + 1 + kf + kc + kf) * sizeof(float);
    static size_t max_set[64] = {};  // per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (lds > 48 * 1024 && lds > max_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sample_fine_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        max_set[dev] = lds;
    }
    hipLaunchKernelGGL(sample_fine_kernel, dim3((unsigned)((n + FINE_WAVES - 1) / FINE_WAVES)), dim3(64 * FINE_WAVES), lds,
                       st, rays, zc, w, depth, n, kc, kf, kfd, depth_std, lindisp, u, u2, g, seed, zout);
}

// ------------------------------------------------------------------ YOLO aggregation
// reference yolo.py:96-114: p = sigmoid(out[...,0]); [max_k p, sum_k p v / (sum_k p + 1e-5)].
// One wavefront per (ray, anchor); lanes stride over the K samples.
__global__ __launch_bounds__(256) void yolo_aggregate_kernel(const float* __restrict__ raw, long long n, int K,
                                                             int na, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n * na) return;
    const long long ray = item / na;
    const int a = (int)(item - ray * na);
    float ps = 0.f, pm = 0.f, v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = lane; k < K; k += 64) {
        const float* r = raw + ((ray * K + k) * na + a) * 7;
        const float p = 1.0f / (1.0f + expf(-r[0]));
        ps += p;
        pm = fmaxf(pm, p);
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] += r[1 + i] * p;
    }
    ps = wave_sum(ps);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pm = fmaxf(pm, __shfl_xor(pm, o, 64));
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = wave_sum(v[i]);
    if (lane == 0) {
        float* o = out + item * 7;
        o[0] = pm;
#pragma unroll
        for (int i = 0; i < 6; ++i) o[1 + i] = v[i] / (ps + 1e-5f);
    }
}

void launch_yolo_aggregate(const float* raw, long long n, int k, int na, float* out, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(yolo_aggregate_kernel, dim3((unsigned)((n * na + 3) / 4)), dim3(256), 0, st, raw, n, k, na,
                       out);
}

// ------------------------------------------------------------------ ray generation
// cam16 per image: [M 3x3 row-major (pixel-dir -> world), origin(3), fx, fy, cx, cy].
//   yolo=0 (reference util.py:115-145,240-278): d = normalize((x-cx)/fx, -(y-cy)/fy, -1), dir = R d
//   yolo=1 (reference util.py:808-876): d = Kinv [x+.49, y+.49, 1] (host passes Kinv entries in
//   fx,fy,cx,cy as 1/fx, 1/fy, -cx/fx, -cy/fy), dir = Einv[:3,:3] d, not normalised.
constexpr int GEN_RAYS_IMGS = 8;  // cameras per launch, passed by value in the kernel-argument segment (512 B):
struct GenRaysCams {              // no device staging buffer, no copy, nothing to free or synchronise
    float cam[GEN_RAYS_IMGS][16];
};

// Pixels [first, first + count) of the flattened (img, y, x) index space of images [img0, img0 + GEN_RAYS_IMGS).
__global__ void gen_rays_kernel(const GenRaysCams cams, int img0, int w, int h, float znear, float zfar, int yolo,
                                long long first, long long count, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const long long p = first + i;
    const long long per = (long long)w * h;
    const int img = (int)(p / per);
    const int pix = (int)(p - img * per);
    const int y = pix / w, x = pix - y * w;
    const float* c = cams.cam[img - img0];
    float d0, d1, d2;
    if (!yolo) {
        d0 = ((float)x - c[14]) / c[12];
        d1 = -(((float)y - c[15]) / c[13]);
        d2 = -1.0f;
        const float nrm = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
        d0 /= nrm;
        d1 /= nrm;
        d2 /= nrm;
    } else {
        const float px = (float)x + 0.49f, py = (float)y + 0.49f;
        d0 = c[12] * px + c[14];
        d1 = c[13] * py + c[15];
        d2 = 1.0f;
    }
    float4* o = reinterpret_cast<float4*>(out + i * 8);  // hipMalloc / torch allocations: rows are 32-byte aligned
    o[0] = make_float4(c[9], c[10], c[11], c[0] * d0 + c[1] * d1 + c[2] * d2);
    o[1] = make_float4(c[3] * d0 + c[4] * d1 + c[5] * d2, c[6] * d0 + c[7] * d1 + c[8] * d2, znear, zfar);
}

// cam16_host: b x 16 floats (host).  Rays [first, first + count) of the (b, h, w) pixel grid -> out (count, 8).
void launch_gen_rays(const float* cam16_host, int b, int w, int h, float znear, float zfar, int yolo, float* out,
                     hipStream_t st, long long first, long long count) {
    const long long per = (long long)w * h;
    if (count <= 0) return;
    for (int img0 = (int)(first / per); img0 < b && (long long)img0 * per < first + count; img0 += GEN_RAYS_IMGS) {
        GenRaysCams cams;
        const int nimg = (b - img0) < GEN_RAYS_IMGS ? (b - img0) : GEN_RAYS_IMGS;
        for (int i = 0; i < GEN_RAYS_IMGS; ++i)
            for (int k = 0; k < 16; ++k) cams.cam[i][k] = i < nimg ? cam16_host[(size_t)(img0 + i) * 16 + k] : 0.f;
        const long long lo = first > (long long)img0 * per ? first : (long long)img0 * per;
        const long long hi_img = (long long)(img0 + nimg) * per;
        const long long hi = first + count < hi_img ? first + count : hi_img;
        if (hi <= lo) continue;
        hipLaunchKernelGGL(gen_rays_kernel, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, st, cams, img0, w, h,
                           znear, zfar, yolo, lo, hi - lo, out + (lo - first) * 8);
    }
}

// ------------------------------------------------------------------ layout repack
// (n, c, hw) <-> (n, hw, c) through a 32x32 LDS tile so that both sides are coalesced.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        int rows, int cols) {
    __shared__ float tile[32][33];
    const size_t base = (size_t)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        if (r < rows && c < cols) tile[i][tx] = in[base + (size_t)r * cols + c];
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (r < rows && c < cols) out[base + (size_t)c * rows + r] = tile[tx][i];
    }
}

void launch_nchw_to_nhwc(const float* in, float* out, int n, int c, int hw, hipStream_t st) {
    hipLaunchKernelGGL(transpose_kernel, dim3((hw + 31) / 32, (c + 31) / 32, n), dim3(256), 0, st, in, out, c, hw);
}
void launch_nhwc_to_nchw(const float* in, float* out, int n, int c, int hw, hipStream_t st) {
    hipLaunchKernelGGL(transpose_kernel, dim3((c + 31) / 32, (hw + 31) / 32, n), dim3(256), 0, st, in, out, hw, c);
}

}  // namespace pny
