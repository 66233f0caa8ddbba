/* A plain-C host above the C ABI (include/pnyolo.h): what a cgo / JNI / FFI binding does, without Python or torch in the
 * process.  Reads one self-describing input file (weights by state_dict name, latent, cameras, rays, the renderer's draws),
 * renders through libpnyolo on HIP device memory it allocates itself (default stream), writes rgb / depth of both passes.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/abi_host.c -o abi_host \
 *       -L pixel-nerf-yolo_amd -lpnyolo -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/pixel-nerf-yolo_amd -Wl,-rpath,/opt/rocm/lib
 *   ./abi_host in.bin out.bin
 *
 * File layout (little endian): int32 magic 0x504e5931; pny_model_desc fields as 11 x 4 bytes (freq_factor as float);
 * int32 n_tensors, then per tensor: int32 name_len, name bytes, int32 ndim, int64 shape[ndim], float data[...];
 * int32 ns, L, hl, wl, W, H; float latent[ns*L*hl*wl] (NCHW); float poses[ns*16]; float focal[2]; float c[2];
 * int32 n_rays, kc, kf, kfd, white_bkgd; float depth_std; float rays[n*8]; float u_coarse[n*kc]; u_fine[n*(kf-kfd)];
 * u_fine2[n*(kf-kfd)]; g_depth[n*kfd].
 * tests/test_gpu_configs.py::test_plain_c_host_through_the_abi writes it, runs this program and compares with the Python path. */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pnyolo.h"

#define CHECK_PNY(call)                                                                  \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != 0) {                                                                  \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, pny_last_error());             \
            return 2;                                                                    \
        }                                                                                \
    } while (0)
#define CHECK_HIP(call)                                                                  \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_));                 \
            return 3;                                                                    \
        }                                                                                \
    } while (0)

static int rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

static float* rd_floats(FILE* f, size_t n) {
    float* p = (float*)malloc(n * sizeof(float) + 16);
    if (p && n && rd(f, p, n * sizeof(float))) {
        free(p);
        return NULL;
    }
    return p;
}

static int to_device(float** dev, const float* host, size_t n) {
    if (n == 0) {
        *dev = NULL;
        return 0;
    }
    CHECK_HIP(hipMalloc((void**)dev, n * sizeof(float)));
    CHECK_HIP(hipMemcpy(*dev, host, n * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]);
        return 1;
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    int32_t magic = 0;
    pny_model_desc desc;
    if (rd(f, &magic, 4) || magic != 0x504e5931 || rd(f, &desc, sizeof(desc))) return 1;
    if (pny_version() != PNY_ABI_VERSION) {
        fprintf(stderr, "library ABI %d, header %d\n", pny_version(), PNY_ABI_VERSION);
        return 1;
    }
    pny_model* m = NULL;
    CHECK_PNY(pny_model_create(&m, &desc));
    int32_t n_tensors = 0;
    if (rd(f, &n_tensors, 4)) return 1;
    for (int32_t t = 0; t < n_tensors; ++t) {
        int32_t len = 0, ndim = 0;
        char name[256];
        int64_t shape[8];
        size_t count = 1;
        if (rd(f, &len, 4) || len <= 0 || len >= (int32_t)sizeof(name) || rd(f, name, (size_t)len) || rd(f, &ndim, 4) || ndim < 0 || ndim > 8) return 1;
        name[len] = 0;
        if (ndim && rd(f, shape, (size_t)ndim * 8)) return 1;
        for (int i = 0; i < ndim; ++i) count *= (size_t)shape[i];
        float* data = rd_floats(f, count);
        if (!data) return 1;
        CHECK_PNY(pny_model_load_weights(m, name, data, shape, ndim));
        free(data);
    }
    CHECK_PNY(pny_model_finalize(m));
    int32_t dims[6];
    if (rd(f, dims, sizeof(dims))) return 1;
    const int ns = dims[0], L = dims[1], hl = dims[2], wl = dims[3], W = dims[4], H = dims[5];
    float* latent = rd_floats(f, (size_t)ns * L * hl * wl);
    float* poses = rd_floats(f, (size_t)ns * 16);
    float focal[2], c[2];
    if (!latent || !poses || rd(f, focal, 8) || rd(f, c, 8)) return 1;
    int32_t r[5];
    float depth_std = 0.f;
    if (rd(f, r, sizeof(r)) || rd(f, &depth_std, 4)) return 1;
    const int n = r[0], kc = r[1], kf = r[2], kfd = r[3];
    float* rays = rd_floats(f, (size_t)n * 8);
    float* u_c = rd_floats(f, (size_t)n * kc);
    float* u_f = rd_floats(f, (size_t)n * (kf - kfd));
    float* u_f2 = rd_floats(f, (size_t)n * (kf - kfd));
    float* g_d = rd_floats(f, (size_t)n * kfd);
    fclose(f);
    if (!rays || !u_c || !u_f || !u_f2 || !g_d) return 1;

    pny_scene* s = NULL;
    CHECK_PNY(pny_scene_create(&s, m));
    CHECK_PNY(pny_scene_set_cameras(s, poses, ns, focal, 1, c, 1, W, H));
    float *d_lat, *d_rays, *d_uc, *d_uf, *d_uf2, *d_gd, *d_out;
    if (to_device(&d_lat, latent, (size_t)ns * L * hl * wl) || to_device(&d_rays, rays, (size_t)n * 8) || to_device(&d_uc, u_c, (size_t)n * kc) ||
        to_device(&d_uf, u_f, (size_t)n * (kf - kfd)) || to_device(&d_uf2, u_f2, (size_t)n * (kf - kfd)) || to_device(&d_gd, g_d, (size_t)n * kfd))
        return 3;
    CHECK_PNY(pny_scene_set_latent(s, d_lat, ns, L, hl, wl, NULL));
    CHECK_HIP(hipMalloc((void**)&d_out, (size_t)n * 8 * sizeof(float)));   /* rgb_c (3n) | depth_c (n) | rgb_f (3n) | depth_f (n) */
    pny_render_opts o;
    memset(&o, 0, sizeof(o));
    o.n_coarse = kc;
    o.n_fine = kf;
    o.n_fine_depth = kfd;
    o.depth_std = depth_std;
    o.white_bkgd = r[4];
    o.u_coarse_dev = d_uc;
    o.u_fine_dev = d_uf;
    o.u_fine2_dev = d_uf2;
    o.g_depth_dev = d_gd;
    pny_render_out out;
    memset(&out, 0, sizeof(out));
    out.rgb_coarse = d_out;
    out.depth_coarse = d_out + 3 * (size_t)n;
    out.rgb_fine = d_out + 4 * (size_t)n;
    out.depth_fine = d_out + 7 * (size_t)n;
    CHECK_PNY(pny_render(s, d_rays, n, &o, &out, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float* host = (float*)malloc((size_t)n * 8 * sizeof(float));
    CHECK_HIP(hipMemcpy(host, d_out, (size_t)n * 8 * sizeof(float), hipMemcpyDeviceToHost));
    FILE* g = fopen(argv[2], "wb");
    if (!g || fwrite(host, sizeof(float), (size_t)n * 8, g) != (size_t)n * 8) return 1;
    fclose(g);
    pny_scene_destroy(s);
    pny_model_destroy(m);
    printf("rendered %d rays (%d + %d samples, %d views, L = %d) through the C ABI version %d\n", n, kc, kf, ns, L, pny_version());
    return 0;
}
