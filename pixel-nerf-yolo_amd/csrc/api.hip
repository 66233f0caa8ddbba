// C ABI of libpnyolo.so (include/pnyolo.h): handles, weight packing, per-scene state and the
// launch sequences of a query / render call.  Host-side C++; all arithmetic of the hot path is
// in the kernels (mlp.hip, render_kernels.hip, encoder.hip).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "api_internal.h"

namespace pny {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int hip_fail(hipError_t e, const char* what) {
    g_err = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what;
    return PNY_ERR_HIP;
}
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

}  // namespace pny

using namespace pny;

// A scene's state (latent, projected maps, workspace, cross-view slab) is written and read by the kernels of
// successive calls without any host synchronisation, which is only ordered if those calls share a stream.  When a
// call arrives on a DIFFERENT stream than the previous call on the same scene, the new stream is made to wait for
// everything the previous call enqueued (one event record + one stream wait, paid only on a stream switch).
namespace pny {
int enter_stream(pny_scene* s, hipStream_t st) {
    if (s->has_last_stream && s->last_stream != st) {
        if (!s->order_ev && hipEventCreateWithFlags(&s->order_ev, hipEventDisableTiming) != hipSuccess) {
            s->order_ev = nullptr;
            return hip_fail(hipGetLastError(), "hipEventCreate(stream order)");
        }
        if (s->order_ev_valid) {   // recorded right behind the previous call's work (mark_stream_point)
            PNY_HIP(hipStreamWaitEvent(st, s->order_ev, 0));
        } else if (hipEventRecord(s->order_ev, s->last_stream) == hipSuccess) {
            PNY_HIP(hipStreamWaitEvent(st, s->order_ev, 0));
        } else {
            (void)hipGetLastError();  // the previous stream no longer exists: its work has drained
        }
    }
    s->order_ev_valid = false;   // the call that enters enqueues new work
    s->last_stream = st;
    s->has_last_stream = true;
    return 0;
}
// Records the scene's order event NOW, behind the work just enqueued on its stream: a later call on another stream then waits
// for exactly this point and not for whatever else the first stream has been given in between (pny_scenes_encode: the scenes
// of a super-batch are encoded on one stream and rendered on one stream each).
int mark_stream_point(pny_scene* s, hipStream_t st) {
    if (!s->order_ev && hipEventCreateWithFlags(&s->order_ev, hipEventDisableTiming) != hipSuccess) {
        s->order_ev = nullptr;
        return hip_fail(hipGetLastError(), "hipEventCreate(stream order)");
    }
    PNY_HIP(hipEventRecord(s->order_ev, st));
    s->order_ev_valid = true;
    return 0;
}
}  // namespace pny

// ---------------------------------------------------------------------------------- packing
// A-operand order of v_mfma_f32_32x32x2_f32 for H^T = W X^T (see mlp.hip): for k-iteration j (8
// inputs), n-tile nt (32 output features), lane l, component r:
//     Wp[((j*NT + nt)*64 + l)*4 + r] = W[32 nt + (l & 31)][8 j + 4 (l >> 5) + r]   (0 beyond K)
// k-iteration-major: the 16 KiB that ALL waves of a workgroup need for iteration j are contiguous,
// so the 16 per-wave streams of a CU walk the same pages together (measured +1 % over n-tile-major,
// where each stream strides through its own 64 KiB region).
static void pack_layer(const float* W, int n_out, int k_in, int k_pad, std::vector<float>& dst) {
    const int J = k_pad / 8, NT = n_out / 32;
    const size_t base = dst.size();
    dst.resize(base + (size_t)NT * J * 64 * 4);
    float* o = dst.data() + base;
    for (int j = 0; j < J; ++j)
        for (int nt = 0; nt < NT; ++nt)
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r) {
                    const int n = 32 * nt + (l & 31), k = 8 * j + 4 * (l >> 5) + r;
                    *o++ = (k < k_in) ? W[(size_t)n * k_in + k] : 0.0f;
                }
}

// Split-f16 image of a layer for mlp_h2.hip: w = w1 + w2, w1 = f16(w), w2 = f16(w - w1) (round to nearest); per 16-k step
// [n-tile][plane][lane] x 8 halves, lane l holding W[32 nt + (l & 31)][16 j + 8 (l >> 5) + 0..7].  4 bytes per weight.
static void pack_layer_h2(const float* W, int n_out, int k_in, int k_pad, std::vector<float>& dst) {
    const int J = k_pad / 16, NT = n_out / 32;
    const size_t base = dst.size();
    dst.resize(base + (size_t)J * NT * 2 * 64 * 4);
    _Float16* o = reinterpret_cast<_Float16*>(dst.data() + base);
    for (int j = 0; j < J; ++j)
        for (int nt = 0; nt < NT; ++nt)
            for (int p = 0; p < 2; ++p)
                for (int l = 0; l < 64; ++l)
                    for (int r = 0; r < 8; ++r) {
                        const int n = 32 * nt + (l & 31), k = 16 * j + 8 * (l >> 5) + r;
                        const float w = (k < k_in) ? W[(size_t)n * k_in + k] : 0.0f;
                        const _Float16 w1 = (_Float16)w;
                        *o++ = p == 0 ? w1 : (_Float16)(w - (float)w1);
                    }
}

// The same split for mlp_h2w.hip (v_mfma_f32_16x16x32_f16): per 32-k step [n-tile of 16][plane][lane] x 8 halves, lane l holding
// W[16 nt + (l & 15)][32 j + 8 (l >> 4) + 0..7].
static void pack_layer_h3(const float* W, int n_out, int k_in, int k_pad, std::vector<float>& dst) {
    const int J = k_pad / 32, NT = n_out / 16;
    const size_t base = dst.size();
    dst.resize(base + (size_t)J * NT * 2 * 64 * 4);
    _Float16* o = reinterpret_cast<_Float16*>(dst.data() + base);
    for (int j = 0; j < J; ++j)
        for (int nt = 0; nt < NT; ++nt)
            for (int p = 0; p < 2; ++p)
                for (int l = 0; l < 64; ++l)
                    for (int r = 0; r < 8; ++r) {
                        const int n = 16 * nt + (l & 15), k = 32 * j + 8 * (l >> 4) + r;
                        const float w = (k < k_in) ? W[(size_t)n * k_in + k] : 0.0f;
                        const _Float16 w1 = (_Float16)w;
                        *o++ = p == 0 ? w1 : (_Float16)(w - (float)w1);
                    }
}

static const HostTensor* find(const pny_model* m, const std::string& name) {
    auto it = m->host.find(name);
    return it == m->host.end() ? nullptr : &it->second;
}

static int need(const pny_model* m, const std::string& name, std::vector<int64_t> shape, const HostTensor** out) {
    const HostTensor* t = find(m, name);
    if (!t) return fail(PNY_ERR_STATE, "missing weight tensor '" + name + "'");
    if (t->shape != shape) return fail(PNY_ERR_ARG, "weight tensor '" + name + "' has an unexpected shape");
    *out = t;
    return 0;
}

struct PackPlan {
    std::vector<float> blob;
    std::vector<std::pair<const float**, size_t>> fix;  // pointer slot -> offset in blob
    size_t add_plain(const std::vector<float>& v) {
        // keep every sub-buffer 64-byte aligned
        while (blob.size() % 16) blob.push_back(0.f);
        const size_t off = blob.size();
        blob.insert(blob.end(), v.begin(), v.end());
        return off;
    }
};

static int pack_mlp(pny_model* m, const std::string& pre, MlpWeights& w, MlpWeightsT& wt, PackPlan& plan) {
    const pny_model_desc& d = m->desc;
    const int d_in = 3 + 6 * d.num_freqs + 3;
    const int nvb = d.combine_layer < d.n_blocks ? d.combine_layer : d.n_blocks;
    const HostTensor* t = nullptr;
    int rc;
    auto packed = [&](const std::string& name, int k_in, int k_pad, const float** slot) -> int {
        if ((rc = need(m, name, {HID, k_in}, &t))) return rc;
        while (plan.blob.size() % 16) plan.blob.push_back(0.f);
        const size_t off = plan.blob.size();
        pack_layer(t->data.data(), HID, k_in, k_pad, plan.blob);
        plan.fix.push_back({slot, off});
        m->repack.push_back({PACK_A, name, "", off, nullptr, HID, k_in, k_pad, 0});
        return 0;
    };
    auto plain = [&](const std::string& name, std::vector<int64_t> shape, const float** slot) -> int {
        if ((rc = need(m, name, shape, &t))) return rc;
        const size_t off = plan.add_plain(t->data);
        plan.fix.push_back({slot, off});
        m->repack.push_back({PACK_COPY, name, "", off, nullptr, 0, 0, 0, (int)t->data.size()});
        return 0;
    };
    // `x = x + lin_z[b](z)` (resnetfc.py:176-182) happens right after lin_in (b = 0) or right after
    // the previous block's fc_1 (b > 0): its bias is folded into that layer's bias here, so the kernel
    // has one bias vector per GEMM chain link and no separate bias pass.
    auto plain_plus = [&](const std::string& name, const std::string& extra, const float** slot) -> int {
        const HostTensor* t2 = nullptr;
        if ((rc = need(m, name, {HID}, &t))) return rc;
        std::vector<float> sum = t->data;
        if (!extra.empty()) {
            if ((rc = need(m, extra, {HID}, &t2))) return rc;
            for (int i = 0; i < HID; ++i) sum[i] += t2->data[i];
        }
        const size_t off = plan.add_plain(sum);
        plan.fix.push_back({slot, off});
        m->repack.push_back({extra.empty() ? PACK_COPY : PACK_ADD2, name, extra, off, nullptr, 0, 0, 0, HID});
        return 0;
    };
    auto zbias = [&](int b) { return b < nvb ? pre + "lin_z." + std::to_string(b) + ".bias" : std::string(); };
    if ((rc = packed(pre + "lin_in.weight", d_in, D_IN_PAD, &w.w_in))) return rc;
    if ((rc = plain_plus(pre + "lin_in.bias", zbias(0), &w.b_in))) return rc;
    for (int b = 0; b < nvb; ++b) {
        const std::string p = pre + "lin_z." + std::to_string(b);
        if ((rc = packed(p + ".weight", d.d_latent, d.d_latent, &w.w_z[b]))) return rc;
        w.b_z[b] = nullptr;  // folded
    }
    for (int b = 0; b < d.n_blocks; ++b) {
        const std::string p = pre + "blocks." + std::to_string(b);
        if ((rc = packed(p + ".fc_0.weight", HID, HID, &w.w_fc0[b]))) return rc;
        if ((rc = plain(p + ".fc_0.bias", {HID}, &w.b_fc0[b]))) return rc;
        if ((rc = packed(p + ".fc_1.weight", HID, HID, &w.w_fc1[b]))) return rc;
        if ((rc = plain_plus(p + ".fc_1.bias", zbias(b + 1), &w.b_fc1[b]))) return rc;
    }
    if ((rc = plain(pre + "lin_out.weight", {d.d_out, HID}, &w.w_out))) return rc;
    if ((rc = plain(pre + "lin_out.bias", {d.d_out}, &w.b_out))) return rc;
    // transposed copies for the backward chain (dX^T = W^T dY^T, mlp_bwd.hip): same operand order, W^T as the matrix
    auto packedT = [&](const std::string& name, int n_out, int k_in, const float** slot) -> int {
        if ((rc = need(m, name, {n_out, k_in}, &t))) return rc;
        std::vector<float> wtr((size_t)n_out * k_in);
        for (int n = 0; n < n_out; ++n)
            for (int k = 0; k < k_in; ++k) wtr[(size_t)k * n_out + n] = t->data[(size_t)n * k_in + k];
        while (plan.blob.size() % 16) plan.blob.push_back(0.f);
        const size_t off = plan.blob.size();
        // W^T is (k_in x n_out): its rows (the GEMM's outputs) must be 512; its K (= n_out) is padded to a ring multiple
        pack_layer(wtr.data(), k_in, n_out, n_out == HID ? HID : D_IN_PAD, plan.blob);
        plan.fix.push_back({slot, off});
        m->repack.push_back({PACK_AT, name, "", off, nullptr, n_out, k_in, n_out == HID ? HID : D_IN_PAD, 0});
        return 0;
    };
    // split-f16 images for the f16x2 kernel (mlp_h2.hip)
    auto packed_h2 = [&](const std::string& name, int k_in, int k_pad, const float** slot) -> int {
        if ((rc = need(m, name, {HID, k_in}, &t))) return rc;
        for (float v : t->data)
            if (!(std::fabs(v) <= 65504.0f)) m->f16_weights_ok = false;   // out of the f16 range (or NaN): AUTO stays on fp32
        while (plan.blob.size() % 16) plan.blob.push_back(0.f);
        const size_t off = plan.blob.size();
        pack_layer_h2(t->data.data(), HID, k_in, k_pad, plan.blob);
        plan.fix.push_back({slot, off});
        m->repack.push_back({PACK_H2, name, "", off, nullptr, HID, k_in, k_pad, 0});
        return 0;
    };
    auto packed_h3 = [&](const std::string& name, int k_in, int k_pad, const float** slot) -> int {
        if ((rc = need(m, name, {HID, k_in}, &t))) return rc;
        while (plan.blob.size() % 16) plan.blob.push_back(0.f);
        const size_t off = plan.blob.size();
        pack_layer_h3(t->data.data(), HID, k_in, k_pad, plan.blob);
        plan.fix.push_back({slot, off});
        m->repack.push_back({PACK_H3, name, "", off, nullptr, HID, k_in, k_pad, 0});
        return 0;
    };
    if ((rc = packed_h3(pre + "lin_in.weight", d_in, D_IN_PAD, &wt.h3_in))) return rc;
    for (int b = 0; b < d.n_blocks; ++b) {
        const std::string p = pre + "blocks." + std::to_string(b);
        if ((rc = packed_h3(p + ".fc_0.weight", HID, HID, &wt.h3_fc0[b]))) return rc;
        if ((rc = packed_h3(p + ".fc_1.weight", HID, HID, &wt.h3_fc1[b]))) return rc;
    }
    if ((rc = packed_h2(pre + "lin_in.weight", d_in, D_IN_PAD, &wt.h2_in))) return rc;
    for (int b = 0; b < d.n_blocks; ++b) {
        const std::string p = pre + "blocks." + std::to_string(b);
        if ((rc = packed_h2(p + ".fc_0.weight", HID, HID, &wt.h2_fc0[b]))) return rc;
        if ((rc = packed_h2(p + ".fc_1.weight", HID, HID, &wt.h2_fc1[b]))) return rc;
    }
    // stacked transposed lin_z for the latent gradient (latent_grad.hip): W_cat[c][b * 512 + f] = lin_z[b].weight[f][c],
    // n-tile-major [c / 32][k-iteration][lane] float4 like the projection weights (encoder.hip build_pixel_linear)
    wt.wzT_cat = nullptr;
    if (nvb > 0) {
        const int Lc = d.d_latent, Kc = nvb * HID, Jc = Kc / 8;
        while (plan.blob.size() % 16) plan.blob.push_back(0.f);
        const size_t off = plan.blob.size();
        plan.blob.resize(off + (size_t)Lc * Kc);
        for (int b = 0; b < nvb; ++b) {
            const std::string name = pre + "lin_z." + std::to_string(b) + ".weight";
            if ((rc = need(m, name, {HID, Lc}, &t))) return rc;
            for (int nt = 0; nt < Lc / 32; ++nt)
                for (int jl = 0; jl < HID / 8; ++jl)
                    for (int l = 0; l < 64; ++l)
                        for (int r = 0; r < 4; ++r) {
                            const int c = 32 * nt + (l & 31), f = 8 * jl + 4 * (l >> 5) + r;
                            plan.blob[off + (((size_t)nt * Jc + b * (HID / 8) + jl) * 64 + l) * 4 + r] = t->data[(size_t)f * Lc + c];
                        }
            m->repack.push_back({PACK_NTT, name, "", off + (size_t)b * (HID / 8) * 64 * 4, nullptr, Lc, HID, Kc, 0});
        }
        plan.fix.push_back({&wt.wzT_cat, off});
    }
    // ... and of the TRANSPOSED matrices for the f16x2 backward chain (mlp_bwd_h2.hip): W^T is (k_in x n_out), K = n_out
    auto packed_h2T = [&](const std::string& name, int n_out, int k_in, const float** slot) -> int {
        if ((rc = need(m, name, {n_out, k_in}, &t))) return rc;
        std::vector<float> wtr((size_t)n_out * k_in);
        for (int n = 0; n < n_out; ++n)
            for (int k = 0; k < k_in; ++k) wtr[(size_t)k * n_out + n] = t->data[(size_t)n * k_in + k];
        for (float v : t->data)
            if (!(std::fabs(v) <= 65504.0f)) m->f16_weights_ok = false;
        const int k_pad = n_out == HID ? HID : D_IN_PAD;
        while (plan.blob.size() % 16) plan.blob.push_back(0.f);
        const size_t off = plan.blob.size();
        pack_layer_h2(wtr.data(), k_in, n_out, k_pad, plan.blob);
        plan.fix.push_back({slot, off});
        m->repack.push_back({PACK_H2T, name, "", off, nullptr, n_out, k_in, k_pad, 0});
        return 0;
    };
    if (d.d_out > D_IN_PAD) return fail(PNY_ERR_ARG, "d_out > 64");
    if ((rc = plain(pre + "lin_in.weight", {HID, d_in}, &wt.w_in_plain))) return rc;
    if ((rc = packedT(pre + "lin_out.weight", d.d_out, HID, &wt.wT_out))) return rc;
    for (int b = 0; b < d.n_blocks; ++b) {
        const std::string p = pre + "blocks." + std::to_string(b);
        if ((rc = packedT(p + ".fc_0.weight", HID, HID, &wt.wT_fc0[b]))) return rc;
        if ((rc = packedT(p + ".fc_1.weight", HID, HID, &wt.wT_fc1[b]))) return rc;
    }
    if ((rc = packed_h2T(pre + "lin_out.weight", d.d_out, HID, &wt.h2T_out))) return rc;
    for (int b = 0; b < d.n_blocks; ++b) {
        const std::string p = pre + "blocks." + std::to_string(b);
        if ((rc = packed_h2T(p + ".fc_0.weight", HID, HID, &wt.h2T_fc0[b]))) return rc;
        if ((rc = packed_h2T(p + ".fc_1.weight", HID, HID, &wt.h2T_fc1[b]))) return rc;
    }
    return 0;
}

// ---------------------------------------------------------------------------------- C ABI
extern "C" {

int pny_version(void) { return PNY_ABI_VERSION; }
const char* pny_last_error(void) { return g_err.c_str(); }

int pny_model_create(pny_model** out, const pny_model_desc* desc) {
    if (!out || !desc) return fail(PNY_ERR_ARG, "pny_model_create: null argument");
    if (desc->d_hidden != HID) return fail(PNY_ERR_ARG, "pny_model_create: d_hidden must be 512");
    if (desc->n_blocks < 1 || desc->n_blocks > MAX_BLOCKS) return fail(PNY_ERR_ARG, "pny_model_create: n_blocks out of range [1,8]");
    if (desc->combine_layer < 0) return fail(PNY_ERR_ARG, "pny_model_create: combine_layer < 0");
    if (desc->d_latent < 128 || desc->d_latent % 128) return fail(PNY_ERR_ARG, "pny_model_create: d_latent must be a positive multiple of 128");
    if (desc->d_out < 1 || desc->d_out > 64) return fail(PNY_ERR_ARG, "pny_model_create: d_out out of range [1,64]");
    if (3 + 6 * desc->num_freqs + 3 > D_IN_PAD || desc->num_freqs < 0) return fail(PNY_ERR_ARG, "pny_model_create: num_freqs too large (d_in must be <= 64)");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(PNY_ERR_NOGPU, "pny_model_create: no HIP device visible (this library has no CPU path)");
    if (desc->device < 0 || desc->device >= count) return fail(PNY_ERR_ARG, "pny_model_create: device ordinal out of range");
    PNY_HIP(hipSetDevice(desc->device));
    pny_model* m = new pny_model();
    m->desc = *desc;
    // f16-range guard word (pny_model_range_status): pinned, device-visible host memory; the kernels OR into it
    if (hipHostMalloc(reinterpret_cast<void**>(&m->range_flag), 64, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        delete m;
        return fail(PNY_ERR_HIP, "pny_model_create: hipHostMalloc(range flag) failed");
    }
    *m->range_flag = 0;
    *out = m;
    return PNY_OK;
}

void pny_model_destroy(pny_model* m) {
    if (!m) return;
    m->packed.release();
    m->repack_jobs.release();
    m->d_absmax.release();
    m->enc_batch_work.release();
    m->enc_batch_lat.release();
    for (int w = 0; w < 2; ++w) {
        m->dx_stash[w].release();
        m->ddy_stash[w].release();
        m->d_partial[w].release();
        m->d_bias[w].release();
        m->d_tables[w].release();
        m->d_stage[w].release();
    }
    for (auto& e : m->flush_ev)
        if (e) (void)hipEventDestroy(e);
    if (m->aux_stream) (void)hipStreamDestroy(m->aux_stream);
    if (m->aux_fork) (void)hipEventDestroy(m->aux_fork);
    if (m->aux_join) (void)hipEventDestroy(m->aux_join);
    m->enc.release();
    for (float* p : m->zproj_allocs) (void)hipFree(p);
    if (m->range_flag) (void)hipHostFree(m->range_flag);
    trunk_release(m->trunk);
    delete m;
}

int pny_model_load_weights(pny_model* m, const char* name, const float* data_host, const int64_t* shape, int ndim) {
    if (!m || !name || (!data_host && ndim > 0) || ndim < 0 || ndim > 4) return fail(PNY_ERR_ARG, "pny_model_load_weights: bad argument");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] < 0) return fail(PNY_ERR_ARG, "pny_model_load_weights: negative dimension");
        t.shape.push_back(shape[i]);
        n *= (size_t)shape[i];
    }
    t.data.assign(data_host, data_host + n);
    m->host[name] = std::move(t);
    m->finalized = false;
    return PNY_OK;
}

int pny_model_finalize(pny_model* m) {
    if (!m) return fail(PNY_ERR_ARG, "pny_model_finalize: null model");
    PNY_HIP(hipSetDevice(m->desc.device));
    PackPlan plan;
    int rc;
    m->repack.clear();
    m->repack_ready = false;
    PNY_HIP(hipDeviceSynchronize());  // a re-finalize must not overwrite weights a running kernel reads
    m->f16_weights_ok = true;
    if ((rc = pack_mlp(m, "mlp_coarse.", m->coarse, m->coarse_t, plan))) return rc;
    if (m->desc.has_fine && (rc = pack_mlp(m, "mlp_fine.", m->fine, m->fine_t, plan))) return rc;
    if (plan.blob.size() * sizeof(float) >= (1ull << 31)) return fail(PNY_ERR_ARG, "packed weights exceed the 2 GiB raw-buffer range");
    if ((rc = m->packed.reserve(plan.blob.size() * sizeof(float)))) return rc;
    PNY_HIP(hipMemcpy(m->packed.p, plan.blob.data(), plan.blob.size() * sizeof(float), hipMemcpyHostToDevice));
    for (auto& f : plan.fix) *f.first = m->packed.f() + f.second;
    if (!m->desc.has_fine) {
        m->fine = m->coarse;
        m->fine_t = m->coarse_t;
    }
    // encoder weights are optional (a scene may be fed through pny_scene_set_latent instead)
    // stacked lin_z maps for the projected-latent variant
    for (float* p : m->zproj_allocs) (void)hipFree(p);
    m->zproj_allocs.clear();
    m->has_zproj = false;
    {
        const int nvb = m->desc.combine_layer < m->desc.n_blocks ? m->desc.combine_layer : m->desc.n_blocks;
        if (nvb > 0) {
            for (int f = 0; f < (m->desc.has_fine ? 2 : 1); ++f) {
                const std::string pre = f ? "mlp_fine." : "mlp_coarse.";
                std::vector<const float*> mats;
                for (int b = 0; b < nvb; ++b) mats.push_back(find(m, pre + "lin_z." + std::to_string(b) + ".weight")->data.data());
                std::string err;
                if (!build_pixel_linear(mats.data(), nvb, HID, m->desc.d_latent, &m->zproj[f], &m->zproj_allocs, &err))
                    return fail(PNY_ERR_HIP, "latent projection weights: " + err);
                for (int b = 0; b < nvb; ++b)   // stacked along the output rows: block b owns n-tiles [16 b, 16 b + 16)
                    m->repack.push_back({PACK_NT, pre + "lin_z." + std::to_string(b) + ".weight", "", 0,
                                         m->zproj[f].w + (size_t)b * 16 * (m->desc.d_latent / 8) * 64 * 4, HID, m->desc.d_latent,
                                         m->desc.d_latent, 0});
            }
            if (!m->desc.has_fine) m->zproj[1] = m->zproj[0];
            m->has_zproj = true;
        }
    }
    ++m->generation;
    m->has_encoder = false;
    if (find(m, "encoder.model.conv1.weight")) {
        auto get = [&](const std::string& name, const float** data, std::vector<int64_t>* shape) -> bool {
            const HostTensor* t = find(m, name);
            if (!t) return false;
            *data = t->data.data();
            *shape = t->shape;
            return true;
        };
        std::string err;
        if (!m->enc.build(get, "encoder.model.", &err)) return fail(PNY_ERR_STATE, "encoder weights: " + err);
        m->has_encoder = true;
    }
    m->finalized = true;
    return PNY_OK;
}

int pny_model_bind_param(pny_model* m, const char* name, const float* param_dev) {
    if (!m || !name) return fail(PNY_ERR_ARG, "pny_model_bind_param: null argument");
    if (param_dev)
        m->params_dev[name] = param_dev;
    else
        m->params_dev.erase(name);
    m->repack_ready = false;
    return PNY_OK;
}

int pny_model_refresh(pny_model* m, pny_stream stream) {
    if (!m) return fail(PNY_ERR_ARG, "pny_model_refresh: null model");
    if (!m->finalized) return fail(PNY_ERR_STATE, "pny_model_refresh: call pny_model_finalize once first (it lays out the packed weights)");
    PNY_HIP(hipSetDevice(m->desc.device));
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (!m->repack_ready) {   // resolve names -> bound device pointers, upload the job table (once per binding)
        std::vector<PackJob> jobs;
        long long max_elems = 0;
        for (const RepackEntry& e : m->repack) {
            auto it = m->params_dev.find(e.name);
            if (it == m->params_dev.end()) return fail(PNY_ERR_STATE, "pny_model_refresh: no device pointer bound for '" + e.name + "'");
            PackJob j;
            memset(&j, 0, sizeof(j));
            j.kind = e.kind;
            j.src = it->second;
            if (!e.name2.empty()) {
                auto it2 = m->params_dev.find(e.name2);
                if (it2 == m->params_dev.end()) return fail(PNY_ERR_STATE, "pny_model_refresh: no device pointer bound for '" + e.name2 + "'");
                j.src2 = it2->second;
            }
            j.dst = e.dst_abs ? e.dst_abs : m->packed.f() + e.dst_off;
            j.n_out = e.n_out;
            j.k_in = e.k_in;
            j.k_pad = e.k_pad;
            if (e.kind == PACK_A || e.kind == PACK_NT || e.kind == PACK_H2 || e.kind == PACK_H3)
                j.count = (e.n_out / 32) * (e.k_pad / 8) * 64;       // 16-byte elements
            else if (e.kind == PACK_AT || e.kind == PACK_H2T)
                j.count = (e.k_in / 32) * (e.k_pad / 8) * 64;
            else if (e.kind == PACK_NTT)
                j.count = (e.n_out / 32) * (e.k_in / 8) * 64;
            else
                j.count = e.count;
            max_elems = std::max(max_elems, (long long)j.count);
            jobs.push_back(j);
        }
        if ((rc = m->repack_jobs.reserve(jobs.size() * sizeof(PackJob)))) return rc;
        PNY_HIP(hipMemcpy(m->repack_jobs.p, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice));
        m->n_repack_jobs = (int)jobs.size();
        m->repack_max_elems = max_elems;
        m->repack_ready = true;
    }
    // order this stream behind the last call of every scene that may still read the packed weights on another stream
    for (pny_scene* s : m->scenes) {
        if (s->has_last_stream && s->last_stream != st) {
            if (!s->order_ev && hipEventCreateWithFlags(&s->order_ev, hipEventDisableTiming) != hipSuccess) {
                s->order_ev = nullptr;
                return hip_fail(hipGetLastError(), "hipEventCreate(refresh order)");
            }
            if (hipEventRecord(s->order_ev, s->last_stream) == hipSuccess)
                PNY_HIP(hipStreamWaitEvent(st, s->order_ev, 0));
            else
                (void)hipGetLastError();
            s->last_stream = st;   // the scene's next call must in turn wait for this refresh
        } else if (!s->has_last_stream) {
            s->last_stream = st;
            s->has_last_stream = true;
        }
        // an order event recorded BEFORE this refresh (mark_stream_point) no longer covers the scene's stream: a later call
        // on another stream has to wait for the repack too, so enter_stream must record a fresh event behind it
        s->order_ev_valid = false;
    }
    launch_repack(reinterpret_cast<const PackJob*>(m->repack_jobs.p), m->n_repack_jobs, m->repack_max_elems, st, m->range_flag);
    PNY_HIP(hipGetLastError());
    ++m->generation;   // projected maps of every scene are stale
    return PNY_OK;
}

int pny_model_use_fine(pny_model* m, int enable) {
    if (!m) return fail(PNY_ERR_ARG, "pny_model_use_fine: null model");
    m->use_fine = enable != 0;
    return PNY_OK;
}

int pny_scene_create(pny_scene** out, pny_model* m) {
    if (!out || !m) return fail(PNY_ERR_ARG, "pny_scene_create: null argument");
    pny_scene* s = new pny_scene();
    s->m = m;
    m->scenes.push_back(s);
    if (const char* e = getenv("PNYOLO_PROJECTION")) {  // process-wide default: off | on | auto
        if (!strcmp(e, "off")) s->zp_mode = PNY_PROJECTION_OFF;
        if (!strcmp(e, "on")) s->zp_mode = PNY_PROJECTION_ON;
    }
    if (const char* e = getenv("PNYOLO_MLP_PRECISION")) {  // process-wide default: f32 | f16x2 | auto
        if (!strcmp(e, "f32")) s->precision = PNY_PRECISION_F32;
        if (!strcmp(e, "f16x2")) s->precision = PNY_PRECISION_F16X2;
    }
    *out = s;
    return PNY_OK;
}

void pny_scene_destroy(pny_scene* s) {
    if (!s) return;
    if (s->m) {
        auto& v = s->m->scenes;
        v.erase(std::remove(v.begin(), v.end(), s), v.end());
    }
    s->latent.release();
    s->work.release();
    s->scratch.release();
    s->enc_work.release();
    s->zp[0].release();
    s->zp[1].release();
    for (DevBuf* b : {&s->dy_absmax, &s->x_stash, &s->dy_stash, &s->dw_partial, &s->dw_bias, &s->dw_tables, &s->d_samp, &s->out_tmp, &s->dz_tmp,
                      &s->sel_tmp, &s->gdepth_tmp})
        b->release();
    for (auto e : s->ev) (void)hipEventDestroy(e);
    for (auto e : s->bev) (void)hipEventDestroy(e);
    s->table_stage.release();
    if (s->order_ev) (void)hipEventDestroy(s->order_ev);
    delete s;
}

int pny_scene_set_cameras(pny_scene* s, const float* poses, int ns, const float* focal, int nf, const float* c, int nc,
                          int width, int height) {
    if (!s || !poses || !focal || !c) return fail(PNY_ERR_ARG, "pny_scene_set_cameras: null argument");
    if (ns < 1 || ns > MAX_VIEWS) return fail(PNY_ERR_ARG, "pny_scene_set_cameras: ns out of range [1,16]");
    if ((nf != 1 && nf != ns) || (nc != 1 && nc != ns)) return fail(PNY_ERR_ARG, "pny_scene_set_cameras: focal / c count must be 1 or ns");
    if (width < 1 || height < 1) return fail(PNY_ERR_ARG, "pny_scene_set_cameras: bad image size");
    for (int v = 0; v < ns; ++v) {
        const float* P = poses + 16 * v;
        Cam& cm = s->cams[v];
        if (!s->m->desc.yolo) {
            // reference models.py:116-118: rot = R^T, trans = -(R^T t) via bmm (fp32)
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) cm.w2c[4 * i + j] = P[4 * j + i];
                float acc = 0.f;
                for (int j = 0; j < 3; ++j) acc += P[4 * j + i] * P[4 * j + 3];
                cm.w2c[4 * i + 3] = -acc;
            }
        } else {
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 4; ++j) cm.w2c[4 * i + j] = P[4 * i + j];
        }
        const float* f = focal + 2 * (nf == 1 ? 0 : v);
        const float* cc = c + 2 * (nc == 1 ? 0 : v);
        cm.fx = f[0];
        cm.fy = s->m->desc.yolo ? f[1] : -f[1];  // models.py:136-137
        cm.cx = cc[0];
        cm.cy = cc[1];
    }
    s->cam_ns = ns;
    s->width = width;
    s->height = height;
    s->have_cams = true;
    return PNY_OK;
}

int pny_scene_set_groups(pny_scene* s, int n_objs) {
    if (!s) return fail(PNY_ERR_ARG, "pny_scene_set_groups: null scene");
    if (n_objs < 1 || n_objs > MAX_VIEWS) return fail(PNY_ERR_ARG, "pny_scene_set_groups: n_objs out of range [1,16]");
    s->n_objs = n_objs;   // (the view count is checked against it when a call needs both: check_ready)
    return PNY_OK;
}

int pny_scene_set_latent(pny_scene* s, const float* latent_dev, int ns, int channels, int hl, int wl, pny_stream stream) {
    if (!s || !latent_dev) return fail(PNY_ERR_ARG, "pny_scene_set_latent: null argument");
    if (channels != s->m->desc.d_latent) return fail(PNY_ERR_ARG, "pny_scene_set_latent: channel count != model d_latent");
    if (ns < 1 || ns > MAX_VIEWS || hl < 1 || wl < 1) return fail(PNY_ERR_ARG, "pny_scene_set_latent: bad shape");
    if ((long long)hl * wl * channels >= (1ll << 31)) return fail(PNY_ERR_ARG, "pny_scene_set_latent: latent too large for 32-bit tap offsets");
    PNY_HIP(hipSetDevice(s->m->desc.device));
    int rc;
    if ((rc = enter_stream(s, (hipStream_t)stream))) return rc;
    if ((rc = s->latent.reserve((size_t)ns * channels * hl * wl * sizeof(float)))) return rc;
    launch_nchw_to_nhwc(latent_dev, s->latent.f(), ns, channels, hl * wl, (hipStream_t)stream);
    PNY_HIP(hipGetLastError());
    s->ns = ns;
    s->L = channels;
    s->hl = hl;
    s->wl = wl;
    s->have_latent = true;
    s->zp_valid[0] = s->zp_valid[1] = false;
    return PNY_OK;
}

int pny_scene_encode(pny_scene* s, const float* images_dev, int ns, int height, int width, pny_stream stream) {
    if (!s || !images_dev) return fail(PNY_ERR_ARG, "pny_scene_encode: null argument");
    if (!s->m->finalized) return fail(PNY_ERR_STATE, "pny_scene_encode: call pny_model_finalize first");
    if (!s->m->has_encoder) return fail(PNY_ERR_STATE, "pny_scene_encode: no encoder.model.* weights were loaded");
    if (s->m->desc.d_latent != 512) return fail(PNY_ERR_ARG, "pny_scene_encode: ResNet-34 trunk yields 512 channels; model d_latent differs");
    if (ns < 1 || ns > MAX_VIEWS || height < 32 || width < 32) return fail(PNY_ERR_ARG, "pny_scene_encode: bad shape");
    PNY_HIP(hipSetDevice(s->m->desc.device));
    int hl = 0, wl = 0;
    encoder_latent_size(height, width, &hl, &wl);
    if ((long long)hl * wl * 512 >= (1ll << 31)) return fail(PNY_ERR_ARG, "pny_scene_encode: latent too large for 32-bit tap offsets");
    int rc;
    if ((rc = enter_stream(s, (hipStream_t)stream))) return rc;
    if ((rc = s->latent.reserve((size_t)ns * 512 * hl * wl * sizeof(float)))) return rc;
    const bool pool = s->m->desc.enc_use_first_pool != 0;
    if ((rc = s->enc_work.reserve(encoder_workspace_bytes(ns, height, width, pool)))) return rc;
    std::string err;
    if (!encoder_forward(s->m->enc, images_dev, ns, height, width, pool, s->enc_work.f(), s->latent.f(), (hipStream_t)stream, &err))
        return fail(PNY_ERR_HIP, "pny_scene_encode: " + err);
    s->ns = ns;
    s->L = 512;
    s->hl = hl;
    s->wl = wl;
    s->have_latent = true;
    s->zp_valid[0] = s->zp_valid[1] = false;
    return PNY_OK;
}

int pny_scenes_encode(pny_scene** scenes, int n_scenes, const float* images_dev, int ns, int height, int width, pny_stream stream) {
    if (!scenes || n_scenes < 1 || !images_dev) return fail(PNY_ERR_ARG, "pny_scenes_encode: null argument");
    for (int i = 0; i < n_scenes; ++i)
        if (!scenes[i] || scenes[i]->m != scenes[0]->m) return fail(PNY_ERR_ARG, "pny_scenes_encode: scenes must share one model");
    if (n_scenes == 1) return pny_scene_encode(scenes[0], images_dev, ns, height, width, stream);
    pny_model* m = scenes[0]->m;
    if (!m->finalized) return fail(PNY_ERR_STATE, "pny_scenes_encode: call pny_model_finalize first");
    if (!m->has_encoder) return fail(PNY_ERR_STATE, "pny_scenes_encode: no encoder.model.* weights were loaded");
    if (m->desc.d_latent != 512) return fail(PNY_ERR_ARG, "pny_scenes_encode: ResNet-34 trunk yields 512 channels; model d_latent differs");
    if (ns < 1 || ns > MAX_VIEWS || height < 32 || width < 32) return fail(PNY_ERR_ARG, "pny_scenes_encode: bad shape");
    PNY_HIP(hipSetDevice(m->desc.device));
    int hl = 0, wl = 0;
    encoder_latent_size(height, width, &hl, &wl);
    if ((long long)hl * wl * 512 >= (1ll << 31)) return fail(PNY_ERR_ARG, "pny_scenes_encode: latent too large for 32-bit tap offsets");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const size_t lat_bytes = (size_t)ns * 512 * hl * wl * sizeof(float);
    for (int i = 0; i < n_scenes; ++i) {
        if ((rc = enter_stream(scenes[i], st))) return rc;
        if ((rc = scenes[i]->latent.reserve(lat_bytes))) return rc;
    }
    // ONE pass of the trunk over every scene's images (n_scenes x ns): 41 launches instead of 41 per scene; the images are
    // independent in an eval-mode trunk, so scene i's latent is the i-th slice of the result
    const bool pool = m->desc.enc_use_first_pool != 0;
    const int n_img = n_scenes * ns;
    if ((rc = m->enc_batch_work.reserve(encoder_workspace_bytes(n_img, height, width, pool)))) return rc;
    if ((rc = m->enc_batch_lat.reserve(lat_bytes * (size_t)n_scenes))) return rc;
    std::string err;
    if (!encoder_forward(m->enc, images_dev, n_img, height, width, pool, m->enc_batch_work.f(), m->enc_batch_lat.f(), st, &err))
        return fail(PNY_ERR_HIP, "pny_scenes_encode: " + err);
    for (int i = 0; i < n_scenes; ++i) {
        pny_scene* s = scenes[i];
        PNY_HIP(hipMemcpyAsync(s->latent.p, reinterpret_cast<const char*>(m->enc_batch_lat.p) + lat_bytes * (size_t)i, lat_bytes,
                               hipMemcpyDeviceToDevice, st));
        s->ns = ns;
        s->L = 512;
        s->hl = hl;
        s->wl = wl;
        s->have_latent = true;
        s->zp_valid[0] = s->zp_valid[1] = false;
    }
    for (int i = 0; i < n_scenes; ++i)
        if ((rc = mark_stream_point(scenes[i], st))) return rc;
    return PNY_OK;
}

int pny_scene_latent_shape(pny_scene* s, int* ns, int* channels, int* hl, int* wl) {
    if (!s || !s->have_latent) return fail(PNY_ERR_STATE, "pny_scene_latent_shape: no latent");
    if (ns) *ns = s->ns;
    if (channels) *channels = s->L;
    if (hl) *hl = s->hl;
    if (wl) *wl = s->wl;
    return PNY_OK;
}

int pny_scene_get_latent(pny_scene* s, float* latent_dev, pny_stream stream) {
    if (!s || !latent_dev) return fail(PNY_ERR_ARG, "pny_scene_get_latent: null argument");
    if (!s->have_latent) return fail(PNY_ERR_STATE, "pny_scene_get_latent: no latent");
    PNY_HIP(hipSetDevice(s->m->desc.device));
    if (int rc = enter_stream(s, (hipStream_t)stream)) return rc;
    launch_nhwc_to_nchw(s->latent.f(), latent_dev, s->ns, s->L, s->hl * s->wl, (hipStream_t)stream);
    PNY_HIP(hipGetLastError());
    return PNY_OK;
}

// 3x3 / 4x4 inverses on the host in double precision (reference uses torch.inverse on fp32).
static bool invert4(const float* m, double* inv) {
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = m[4 * i + j];
            a[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r)
            if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (std::fabs(a[piv][c]) < 1e-30) return false;
        if (piv != c)
            for (int j = 0; j < 8; ++j) std::swap(a[piv][j], a[c][j]);
        const double d = a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] /= d;
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                const double f = a[r][c];
                for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
            }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) inv[4 * i + j] = a[i][4 + j];
    return true;
}

int pny_gen_rays_range(const float* poses_host, int b, int width, int height, const float focal[2], const float c[2],
                       float z_near, float z_far, int yolo_mode, int64_t first_ray, int64_t n_rays, float* out_dev,
                       pny_stream stream) {
    if (!poses_host || !focal || !c || (!out_dev && n_rays > 0)) return fail(PNY_ERR_ARG, "pny_gen_rays: null argument");
    if (b < 0 || width < 1 || height < 1) return fail(PNY_ERR_ARG, "pny_gen_rays: bad shape");
    const int64_t total = (int64_t)b * width * height;
    if (first_ray < 0 || n_rays < 0 || first_ray + n_rays > total) return fail(PNY_ERR_ARG, "pny_gen_rays: ray range outside the (b, h, w) grid");
    if (n_rays == 0) return PNY_OK;
    if (reinterpret_cast<uintptr_t>(out_dev) & 15) return fail(PNY_ERR_ARG, "pny_gen_rays: out_dev must be 16-byte aligned");
    std::vector<float> cam((size_t)b * 16);
    for (int i = 0; i < b; ++i) {
        const float* P = poses_host + 16 * i;
        float* o = cam.data() + 16 * i;
        if (!yolo_mode) {
            for (int r = 0; r < 3; ++r) {
                for (int q = 0; q < 3; ++q) o[3 * r + q] = P[4 * r + q];
                o[9 + r] = P[4 * r + 3];
            }
            o[12] = focal[0];
            o[13] = focal[1];
            o[14] = c[0];
            o[15] = c[1];
        } else {
            double inv[16];
            if (!invert4(P, inv)) return fail(PNY_ERR_ARG, "pny_gen_rays: singular extrinsic matrix");
            for (int r = 0; r < 3; ++r) {
                for (int q = 0; q < 3; ++q) o[3 * r + q] = (float)inv[4 * r + q];
                o[9 + r] = (float)inv[4 * r + 3];
            }
            if (focal[0] == 0.f || focal[1] == 0.f) return fail(PNY_ERR_ARG, "pny_gen_rays: zero focal length");
            o[12] = (float)(1.0 / focal[0]);
            o[13] = (float)(1.0 / focal[1]);
            o[14] = (float)(-(double)c[0] / focal[0]);
            o[15] = (float)(-(double)c[1] / focal[1]);
        }
    }
    // the per-image parameter blocks travel as kernel arguments: nothing is staged, allocated or synchronised here
    launch_gen_rays(cam.data(), b, width, height, z_near, z_far, yolo_mode, out_dev, (hipStream_t)stream, first_ray, n_rays);
    PNY_HIP(hipGetLastError());
    return PNY_OK;
}

int pny_gen_rays(const float* poses_host, int b, int width, int height, const float focal[2], const float c[2],
                 float z_near, float z_far, int yolo_mode, float* out_dev, pny_stream stream) {
    if (b < 0 || width < 1 || height < 1) return fail(PNY_ERR_ARG, "pny_gen_rays: bad shape");
    return pny_gen_rays_range(poses_host, b, width, height, focal, c, z_near, z_far, yolo_mode, 0,
                              (int64_t)b * width * height, out_dev, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------------- MLP launch
namespace pny {
static unsigned range_bits(const pny_model* m) {
    return m->range_flag ? __atomic_load_n(m->range_flag, __ATOMIC_RELAXED) : 0u;
}
int check_ready(pny_scene* s, const char* who) {
    if (!s) return fail(PNY_ERR_ARG, std::string(who) + ": null scene");
    if (!s->m->finalized) return fail(PNY_ERR_STATE, std::string(who) + ": weights not finalized (pny_model_finalize)");
    // f16-range guard: scenes pinned to F32 neither cause nor suffer from it (weights beyond the f16 range are legal there)
    const unsigned bits = s->precision == PNY_PRECISION_F32 ? 0u : range_bits(s->m);
    if (bits) {
        if (bits & PNY_RANGE_WEIGHT) s->m->f16_weights_ok = false;   // AUTO scenes run F32 from here on
        return fail(PNY_ERR_RANGE, std::string(who) + ": an earlier F16X2 launch left the f16 range (" +
                                       std::string(bits & PNY_RANGE_ACTIVATION ? "activation " : "") +
                                       std::string(bits & PNY_RANGE_GRADIENT ? "gradient " : "") +
                                       std::string(bits & PNY_RANGE_WEIGHT ? "weight " : "") +
                                       "beyond +-65504 or not finite): its results are invalid.  Clear with pny_model_range_status(m, 0, 1), pin "
                                       "pny_scene_set_precision(s, PNY_PRECISION_F32) and repeat the call");
    }
    if (!s->have_latent) return fail(PNY_ERR_STATE, std::string(who) + ": scene has no latent (pny_scene_encode / pny_scene_set_latent)");
    if (!s->have_cams) return fail(PNY_ERR_STATE, std::string(who) + ": scene has no cameras (pny_scene_set_cameras)");
    if (s->cam_ns != s->ns) return fail(PNY_ERR_STATE, std::string(who) + ": camera count != latent view count");
    if (s->n_objs > 1 && s->ns % s->n_objs)
        return fail(PNY_ERR_STATE, std::string(who) + ": grouped scene: the view count is not a multiple of the object count");
    if (obj_views(s) > 1 && s->m->desc.combine_layer >= s->m->desc.n_blocks)
        return fail(PNY_ERR_ARG, std::string(who) + ": multi-view scene needs combine_layer < n_blocks");
    return 0;
}

int view_blocks(const pny_model_desc& d) { return d.combine_layer < d.n_blocks ? d.combine_layer : d.n_blocks; }
}  // namespace pny

// GEMM FLOPs (2 per MAC, unpadded) per query point: as the reference computes it (with_lin_z) or as the
// projected-latent variant executes it (lin_z moved to the per-scene projection).
static double mlp_flops_per_point(const pny_model_desc& d, int ns, bool with_lin_z) {
    const int d_in = 3 + 6 * d.num_freqs + 3;
    const int nvb = view_blocks(d);
    const double per_view = (double)d_in * HID + (with_lin_z ? (double)nvb * d.d_latent * HID : 0.0) + 2.0 * nvb * HID * HID;
    const double post = 2.0 * (d.n_blocks - nvb) * HID * HID + (double)HID * d.d_out;
    return 2.0 * (ns * per_view + post);
}

// Projected latent: zp[v][y][x][b*512 + n] = sum_k lin_z[b].weight[n][k] * latent[v][y][x][k], computed
// once per (scene latent, weights) on the caller's stream and cached.  AUTO uses it when the launch has
// at least twice as many points as the latent has pixels per view (the projection costs one lin_z per
// PIXEL instead of one per (sample, view); tiny training-size batches on large maps stay direct).
namespace pny {
int ensure_projection(pny_scene* s, int which, long long n_points, hipStream_t st, const float** zp, bool force) {
    *zp = nullptr;
    const pny_model* m = s->m;
    if (!m->has_zproj) return 0;
    if (!force && s->zp_mode == PNY_PROJECTION_OFF) return 0;
    // AUTO: with the f16x2 kernel available every launch is projected -- a lone 64-sample tile on it (0.5 ms) beats the
    // 32-sample fp32 shape on the unprojected latent (1.0 ms) even with the one-off projection of the scene (0.3 ms per MLP
    // at C2), and a ray's result then never depends on the size of the batch it is rendered in.  Without it (F32 scenes,
    // more than 6 blocks): project when the launch has at least 2x as many points as the latent has pixels per view.
    const bool h2_ok = s->precision != PNY_PRECISION_F32 && m->f16_weights_ok && mlp_h2_supports(m->desc.n_blocks, m->desc.combine_layer);
    if (!force && s->zp_mode == PNY_PROJECTION_AUTO && !h2_ok && n_points < 2ll * s->hl * s->wl) return 0;
    const int nvb = view_blocks(m->desc);
    if ((long long)s->hl * s->wl * nvb * HID >= (1ll << 31)) return 0;  // 32-bit tap offsets: stay direct
    if (s->zp_generation != m->generation) {
        s->zp_valid[0] = s->zp_valid[1] = false;
        s->zp_generation = m->generation;
    }
    if (!m->desc.has_fine) which = 0;
    if (!s->zp_valid[which]) {
        const long long npix = (long long)s->ns * s->hl * s->wl;
        int rc;
        if ((rc = s->zp[which].reserve((size_t)npix * nvb * HID * sizeof(float)))) return rc;
        // scenes whose projected launches run the f16x2 kernel project on the split-f16 matrix path too
        if (!run_pixel_linear(m->zproj[which], s->latent.f(), npix, s->zp[which].f(), st, h2_ok))
            return fail(PNY_ERR_HIP, "latent projection launch failed");
        s->zp_valid[which] = true;
    }
    *zp = s->zp[which].f();
    return 0;
}
}  // namespace pny

namespace pny {
int fill_mlp_args(pny_scene* s, int mode, const float* xyz, const float* dirs, const float* rays, const float* z, int K,
                  long long n_points, int coarse, float* out, MlpArgs* pa) {
    const pny_model_desc& d = s->m->desc;
    MlpArgs& a = *pa;
    memset(&a, 0, sizeof(a));
    const bool fine_w = !(coarse || !d.has_fine || !s->m->use_fine);
    a.w = fine_w ? s->m->fine : s->m->coarse;
    {
        const MlpWeightsT& wt = fine_w ? s->m->fine_t : s->m->coarse_t;
        a.h2_in = wt.h2_in;
        a.h3_in = wt.h3_in;
        for (int b = 0; b < d.n_blocks; ++b) {
            a.h2_fc0[b] = wt.h2_fc0[b];
            a.h2_fc1[b] = wt.h2_fc1[b];
            a.h3_fc0[b] = wt.h3_fc0[b];
            a.h3_fc1[b] = wt.h3_fc1[b];
        }
    }
    a.w_base = s->m->packed.f();
    a.w_bytes = (unsigned)s->m->packed.bytes;
    a.range_flag = s->m->range_flag;
    a.latent = s->latent.f();
    a.zp = nullptr;
    a.zp_stride = view_blocks(d) * HID;
    a.tap_stride = s->L;
    memcpy(a.cams, s->cams, sizeof(Cam) * (size_t)s->ns);
    a.xyz = xyz;
    a.dirs = dirs;
    a.rays = rays;
    a.z = z;
    a.out = out;
    a.n_points = n_points;
    a.K = K;
    a.mode = mode;
    a.NS = obj_views(s);
    a.obj_pts = 0;
    if (s->n_objs > 1) {   // grouped scene: equal consecutive shares, whole tiles per object (pny_scene_set_groups)
        if (n_points % s->n_objs || (n_points / s->n_objs) % 64)
            return fail(PNY_ERR_ARG, "grouped scene: every object's share of the samples must be the same multiple of 64");
        a.obj_pts = n_points / s->n_objs;
    }
    a.L = s->L;
    a.Hl = s->hl;
    a.Wl = s->wl;
    a.n_blocks = d.n_blocks;
    a.combine_layer = d.combine_layer;
    a.d_out = d.d_out;
    a.yolo = d.yolo;
    a.num_freqs = d.num_freqs;
    a.freq_factor = d.freq_factor;
    // latent_scaling / image_size in fp32 (reference encoder.py:97,170-172)
    const float lsx = (float)s->wl / ((float)s->wl - 1.0f) * 2.0f;
    const float lsy = (float)s->hl / ((float)s->hl - 1.0f) * 2.0f;
    a.sx = lsx / (float)s->width;
    a.sy = lsy / (float)s->height;
    const long long tiles = (n_points + 63) / 64;
    if (tiles > 0x7fffffffll) return fail(PNY_ERR_ARG, "too many points for one launch");
    a.n_tiles = (int)tiles;
    a.idx32 = (tiles * 64) < 0xffffffffll;
    if (mode == 1 && (reinterpret_cast<uintptr_t>(rays) & 15)) return fail(PNY_ERR_ARG, "rays must be 16-byte aligned");
    int rc;
    if ((rc = s->scratch.reserve(mlp_scratch_floats() * sizeof(float)))) return rc;
    a.scratch = s->scratch.f();
    return 0;
}
}  // namespace pny

static bool h2w_default() { return false; }   // (experimental until it wins: PNYOLO_H2_WIDE=1)

static int run_mlp(pny_scene* s, int mode, const float* xyz, const float* dirs, const float* rays, const float* z,
                   int K, long long n_points, int coarse, float* out, hipStream_t st, int stash_pass = -1) {
    if (n_points == 0) return 0;
    const pny_model_desc& d = s->m->desc;
    MlpArgs a;
    int rc;
    if ((rc = fill_mlp_args(s, mode, xyz, dirs, rays, z, K, n_points, coarse, out, &a))) return rc;
    const bool fine_w = !(coarse || !d.has_fine || !s->m->use_fine);
    if (stash_pass >= 0) {
        // Training forward (pny_scene_stash_next_render): the STASH instantiation (reference operation order, 64-sample
        // tiles) writes every GEMM operand into the tiles this pass takes from the model-level reservation; the backward
        // of the same reservation epoch then starts at the dX chain.
        pny_model* m = s->m;
        const int which = fine_w ? 1 : 0;
        pny_scene::StashedPass& sp = s->stashed[stash_pass];
        sp.valid = false;
        if (m->defer && obj_views(s) == m->defer_ns && m->defer_used[which] + a.n_tiles <= m->defer_cap[which]) {
            a.lay = stash_layout(d, obj_views(s), s->L);
            a.stash_x = m->dx_stash[which].f() + m->defer_used[which] * a.lay.x_tile;
            sp.valid = true;
            sp.epoch = m->defer_epoch;
            sp.which = which;
            sp.tile0 = m->defer_used[which];
            sp.tiles = a.n_tiles;
            sp.n_points = n_points;
            m->defer_used[which] += a.n_tiles;
            const int grid = std::min(mlp_max_grid(MLP_8x64), a.n_tiles);
            // f16x2 variant of the stashing forward (mlp_h2.hip, STASH instantiation): projected latent for the forward, the
            // raw latent gathered once more per view for lin_z's weight gradient; the same stash layout and contents
            bool use_h2 = false;
            if (s->precision != PNY_PRECISION_F32 && (m->f16_weights_ok || s->precision == PNY_PRECISION_F16X2) &&
                mlp_h2_supports(d.n_blocks, d.combine_layer) && s->L % 128 == 0) {
                if ((rc = ensure_projection(s, fine_w ? 1 : 0, n_points, st, &a.zp, true))) return rc;
                use_h2 = a.zp != nullptr;
                if (use_h2) a.tap_stride = a.zp_stride;
            }
            if (s->timing) {
                while ((int)s->ev.size() < s->ev_used + 2) {
                    hipEvent_t e;
                    PNY_HIP(hipEventCreate(&e));
                    s->ev.push_back(e);
                }
                PNY_HIP(hipEventRecord(s->ev[s->ev_used], st));
            }
            if (use_h2)
                launch_mlp_h2_stash(a, grid, st);
            else
                launch_mlp_stash(a, grid, st);
            PNY_HIP(hipGetLastError());
            if (s->timing) {
                PNY_HIP(hipEventRecord(s->ev[s->ev_used + 1], st));
                s->ev_used += 2;
            }
            s->last_flops += mlp_flops_per_point(d, obj_views(s), !use_h2) * (double)n_points;
            s->last_flops_ref += mlp_flops_per_point(d, obj_views(s), true) * (double)n_points;
            s->last_projected = use_h2;
            s->last_f16x2 = use_h2;
            s->last_launches += 1;
            return 0;
        }
        // no room in the reservation: plain forward, the backward recomputes
    }
    if ((rc = ensure_projection(s, fine_w ? 1 : 0, n_points, st, &a.zp))) return rc;
    a.tap_stride = a.zp ? a.zp_stride : s->L;
    int variant = mlp_pick_variant(n_points);
    // f16x2 kernel (split-f16 operands, mlp_h2.hip): every projected launch unless the scene is pinned to F32 -- one
    // arithmetic for all projected launches keeps a ray's result independent of the batch it is rendered in (ray
    // sharding stays bit-exact); a lone 64-sample h2 tile is also faster than the 32-sample fp32 shape it replaces.
    const bool use_h2 = a.zp && mlp_h2_supports(d.n_blocks, d.combine_layer) && s->precision != PNY_PRECISION_F32 &&
                        (s->m->f16_weights_ok || s->precision == PNY_PRECISION_F16X2);
    if (use_h2) variant = MLP_8x64;
    // Split shape of the f16x2 kernel (mlp_h2s.hip: 32-sample tiles, 4-wave workgroups, two per CU): the same arithmetic per
    // sample, bit for bit.  Measured (profiles/r02zk_split_sweep.log): a launch that gives every CU at most ONE 32-sample tile
    // takes 0.40-0.43 ms against 0.47-0.51 ms on 64-sample tiles; as soon as two workgroups share a CU the doubled weight
    // stream per sample costs more than the overlap of their phases returns (full C2 frame: 61.7 vs 39.9 ms per launch).
    // So: launches of at most 32 x CUs points.  PNYOLO_H2_SPLIT=0|1 overrides.
    bool use_h2s = use_h2 && n_points <= 32ll * mlp_max_grid(MLP_8x64);
    if (use_h2)
        if (const char* e = getenv("PNYOLO_H2_SPLIT")) use_h2s = atoi(e) != 0;
    // wide shape (mlp_h2w.hip: 4 waves x 512 registers, 16 x 16 x 32 MFMAs): launches that give every CU more than one tile
    bool use_h2w = use_h2 && !use_h2s && mlp_h2w_supports(d.n_blocks, d.combine_layer) && h2w_default();
    int h2w_mode = 1;   // 1: 4 waves x 512 registers (mlp_h2w.hip), 2: 8 waves x 256 (mlp_h2n.hip)
    if (use_h2 && !use_h2s)
        if (const char* e = getenv("PNYOLO_H2_WIDE")) {
            h2w_mode = atoi(e);
            use_h2w = h2w_mode != 0 && mlp_h2w_supports(d.n_blocks, d.combine_layer);
        }
    const int tm = use_h2s ? 32 : mlp_tile_samples(variant);
    const long long tiles = (n_points + tm - 1) / tm;
    if (tiles > 0x7fffffffll) return fail(PNY_ERR_ARG, "too many points for one launch");
    a.n_tiles = (int)tiles;
    a.idx32 = (tiles * tm) < 0xffffffffll;
    int grid = use_h2s ? 2 * mlp_max_grid(MLP_8x64) : mlp_max_grid(variant);
    if (const char* e = getenv("PNYOLO_GRID")) {  // diagnostic: fewer resident workgroups
        const int g = atoi(e);
        if (g > 0 && g < grid) grid = g;
    }
    if (tiles < grid) grid = (int)tiles;
    if (s->timing) {
        while ((int)s->ev.size() < s->ev_used + 2) {
            hipEvent_t e;
            PNY_HIP(hipEventCreate(&e));
            s->ev.push_back(e);
        }
        PNY_HIP(hipEventRecord(s->ev[s->ev_used], st));
    }
    if (use_h2s)
        launch_mlp_h2s(a, grid, st);
    else if (use_h2w && h2w_mode == 2)
        launch_mlp_h2n(a, grid, st);
    else if (use_h2w)
        launch_mlp_h2w(a, grid, st);
    else if (use_h2)
        launch_mlp_h2(a, grid, st);
    else
        launch_mlp(a, variant, grid, st);
    PNY_HIP(hipGetLastError());
    if (s->timing) {
        PNY_HIP(hipEventRecord(s->ev[s->ev_used + 1], st));
        s->ev_used += 2;
    }
    s->last_f16x2 = use_h2;
    s->last_flops += mlp_flops_per_point(d, obj_views(s), a.zp == nullptr) * (double)n_points;
    s->last_flops_ref += mlp_flops_per_point(d, obj_views(s), true) * (double)n_points;
    s->last_projected = a.zp != nullptr;
    s->last_launches += 1;
    return 0;
}

static void begin_call(pny_scene* s) {
    s->ev_used = 0;
    s->last_flops = 0.0;
    s->last_flops_ref = 0.0;
    s->last_launches = 0;
}

extern "C" {

int pny_query(pny_scene* s, const float* xyz_dev, const float* viewdirs_dev, int64_t n, int coarse, float* out_dev,
              pny_stream stream) {
    int rc;
    if ((rc = check_ready(s, "pny_query"))) return rc;
    if (n < 0 || (n > 0 && (!xyz_dev || !viewdirs_dev || !out_dev))) return fail(PNY_ERR_ARG, "pny_query: bad argument");
    PNY_HIP(hipSetDevice(s->m->desc.device));
    if ((rc = enter_stream(s, (hipStream_t)stream))) return rc;
    begin_call(s);
    return run_mlp(s, 0, xyz_dev, viewdirs_dev, nullptr, nullptr, 1, n, coarse, out_dev, (hipStream_t)stream);
}

int pny_sample_coarse(const float* rays_dev, int64_t n, int n_coarse, int lindisp, const float* u_dev, uint64_t seed,
                      float* z_dev, pny_stream stream) {
    if (n < 0 || n_coarse < 1 || (n > 0 && (!rays_dev || !z_dev))) return fail(PNY_ERR_ARG, "pny_sample_coarse: bad argument");
    launch_sample_coarse(rays_dev, n, n_coarse, lindisp, u_dev, seed, z_dev, (hipStream_t)stream);
    PNY_HIP(hipGetLastError());
    return PNY_OK;
}

int pny_composite(const float* rays_dev, const float* z_dev, const float* sample_dev, int64_t n, int k, int white_bkgd,
                  float* weights_dev, float* rgb_dev, float* depth_dev, pny_stream stream) {
    if (n < 0 || k < 1 || (n > 0 && (!rays_dev || !z_dev || !sample_dev))) return fail(PNY_ERR_ARG, "pny_composite: bad argument");
    launch_composite(rays_dev, z_dev, sample_dev, n, k, white_bkgd, weights_dev, rgb_dev, depth_dev, (hipStream_t)stream);
    PNY_HIP(hipGetLastError());
    return PNY_OK;
}

int pny_sample_fine(const float* rays_dev, const float* z_coarse_dev, const float* weights_dev, const float* depth_dev,
                    int64_t n, int n_coarse, int n_fine, int n_fine_depth, float depth_std, int lindisp,
                    const float* u_dev, const float* u2_dev, const float* g_dev, uint64_t seed, float* z_out_dev,
                    pny_stream stream) {
    if (n < 0 || n_coarse < 1 || n_fine < 0 || n_fine_depth < 0 || n_fine_depth > n_fine)
        return fail(PNY_ERR_ARG, "pny_sample_fine: bad sample counts");
    if (n > 0 && (!rays_dev || !z_coarse_dev || !weights_dev || !z_out_dev || (n_fine_depth > 0 && !depth_dev)))
        return fail(PNY_ERR_ARG, "pny_sample_fine: null argument");
    if ((size_t)(4 * n_coarse + 1 + 2 * n_fine) * 4 * sizeof(float) > 160 * 1024)
        return fail(PNY_ERR_ARG, "pny_sample_fine: n_coarse + n_fine too large for the LDS-resident sort");
    launch_sample_fine(rays_dev, z_coarse_dev, weights_dev, depth_dev, n, n_coarse, n_fine, n_fine_depth, depth_std,
                       lindisp, u_dev, u2_dev, g_dev, seed, z_out_dev, (hipStream_t)stream);
    PNY_HIP(hipGetLastError());
    return PNY_OK;
}

int pny_yolo_aggregate(const float* raw_dev, int64_t n, int k, int n_anchors, float* out_dev, pny_stream stream) {
    if (n < 0 || k < 1 || n_anchors < 1 || (n > 0 && (!raw_dev || !out_dev))) return fail(PNY_ERR_ARG, "pny_yolo_aggregate: bad argument");
    launch_yolo_aggregate(raw_dev, n, k, n_anchors, out_dev, (hipStream_t)stream);
    PNY_HIP(hipGetLastError());
    return PNY_OK;
}

int pny_render(pny_scene* s, const float* rays_dev, int64_t n, const pny_render_opts* o, const pny_render_out* out,
               pny_stream stream) {
    int rc;
    if ((rc = check_ready(s, "pny_render"))) return rc;
    if (!o || !out || n < 0 || (n > 0 && !rays_dev)) return fail(PNY_ERR_ARG, "pny_render: bad argument");
    if (s->m->desc.yolo || s->m->desc.d_out != 4) return fail(PNY_ERR_ARG, "pny_render: model is in YOLO mode (use pny_yolo_render)");
    if (o->n_coarse < 1 || o->n_fine < 0 || o->n_fine_depth < 0 || o->n_fine_depth > o->n_fine)
        return fail(PNY_ERR_ARG, "pny_render: bad sample counts");
    const int kimp = o->n_fine - o->n_fine_depth;
    const bool any_u = o->u_coarse_dev || o->u_fine_dev || o->u_fine2_dev || o->g_depth_dev;
    if (any_u) {
        if (!o->u_coarse_dev || (kimp > 0 && (!o->u_fine_dev || !o->u_fine2_dev)) || (o->n_fine_depth > 0 && !o->g_depth_dev))
            return fail(PNY_ERR_ARG, "pny_render: explicit random draws must be given for every stage or for none");
    }
    if (n == 0) return PNY_OK;
    PNY_HIP(hipSetDevice(s->m->desc.device));
    hipStream_t st = (hipStream_t)stream;
    if ((rc = enter_stream(s, st))) return rc;
    begin_call(s);
    const int kc = o->n_coarse, kt = o->n_coarse + o->n_fine;
    // workspace carve (floats): z_c, samp_c, w_c, rgb_c(3)+depth_c, z_f, samp_f
    size_t off = 0;
    auto carve = [&](size_t nfl) {
        size_t r = off;
        off += (nfl + 63) & ~(size_t)63;
        return r;
    };
    const size_t o_zc = carve((size_t)n * kc), o_sc = carve((size_t)n * kc * 4), o_wc = carve((size_t)n * kc);
    const size_t o_dc = carve((size_t)n), o_rc = carve((size_t)n * 3);
    const size_t o_zf = carve((size_t)n * kt), o_sf = carve((size_t)n * kt * 4);
    if ((rc = s->work.reserve(off * sizeof(float)))) return rc;
    float* W = s->work.f();
    float* zc = out->z_coarse ? out->z_coarse : W + o_zc;
    float* sc = out->sample_coarse ? out->sample_coarse : W + o_sc;
    float* wc = out->weights_coarse ? out->weights_coarse : W + o_wc;
    float* dc = out->depth_coarse ? out->depth_coarse : W + o_dc;
    float* rgbc = out->rgb_coarse ? out->rgb_coarse : W + o_rc;

    const bool stash = s->stash_next;
    s->stash_next = false;
    s->stashed[0].valid = s->stashed[1].valid = false;
    launch_sample_coarse(rays_dev, n, kc, o->lindisp, o->u_coarse_dev, o->seed, zc, st);
    if ((rc = run_mlp(s, 1, nullptr, nullptr, rays_dev, zc, kc, (long long)n * kc, 1, sc, st, stash ? 0 : -1))) return rc;
    launch_composite(rays_dev, zc, sc, n, kc, o->white_bkgd, wc, rgbc, dc, st, o->sigma_noise_coarse_dev);
    if (o->n_fine > 0) {
        float* zf = out->z_fine ? out->z_fine : W + o_zf;
        float* sf = out->sample_fine ? out->sample_fine : W + o_sf;
        if ((size_t)(4 * kc + 1 + 2 * o->n_fine) * 4 * sizeof(float) > 160 * 1024)
            return fail(PNY_ERR_ARG, "pny_render: n_coarse + n_fine too large for the LDS-resident sort");
        launch_sample_fine(rays_dev, zc, wc, dc, n, kc, o->n_fine, o->n_fine_depth, o->depth_std, o->lindisp,
                           o->u_fine_dev, o->u_fine2_dev, o->g_depth_dev, o->seed, zf, st);
        if ((rc = run_mlp(s, 1, nullptr, nullptr, rays_dev, zf, kt, (long long)n * kt, 0, sf, st, stash ? 1 : -1))) return rc;
        launch_composite(rays_dev, zf, sf, n, kt, o->white_bkgd, out->weights_fine, out->rgb_fine, out->depth_fine, st,
                         o->sigma_noise_fine_dev);
    }
    PNY_HIP(hipGetLastError());
    return PNY_OK;
}

int pny_yolo_render(pny_scene* s, const float* rays_dev, int64_t n, int n_coarse, const float* u_coarse_dev,
                    uint64_t seed, float* out_dev, float* raw_dev, pny_stream stream) {
    int rc;
    if ((rc = check_ready(s, "pny_yolo_render"))) return rc;
    const pny_model_desc& d = s->m->desc;
    if (!d.yolo || d.d_out % 7) return fail(PNY_ERR_ARG, "pny_yolo_render: model is not in YOLO mode");
    if (n < 0 || n_coarse < 1 || (n > 0 && (!rays_dev || !out_dev))) return fail(PNY_ERR_ARG, "pny_yolo_render: bad argument");
    if (n == 0) return PNY_OK;
    PNY_HIP(hipSetDevice(d.device));
    hipStream_t st = (hipStream_t)stream;
    if ((rc = enter_stream(s, st))) return rc;
    begin_call(s);
    const size_t nz = ((size_t)n * n_coarse + 63) & ~(size_t)63;
    if ((rc = s->work.reserve((nz + (size_t)n * n_coarse * d.d_out) * sizeof(float)))) return rc;
    float* z = s->work.f();
    float* raw = raw_dev ? raw_dev : s->work.f() + nz;
    launch_sample_coarse(rays_dev, n, n_coarse, 0, u_coarse_dev, seed, z, st);
    if ((rc = run_mlp(s, 1, nullptr, nullptr, rays_dev, z, n_coarse, (long long)n * n_coarse, 1, raw, st))) return rc;
    launch_yolo_aggregate(raw, n, n_coarse, d.d_out / 7, out_dev, st);
    PNY_HIP(hipGetLastError());
    return PNY_OK;
}

int pny_scene_enable_timing(pny_scene* s, int enable) {
    if (!s) return fail(PNY_ERR_ARG, "pny_scene_enable_timing: null scene");
    s->timing = enable != 0;
    return PNY_OK;
}

int pny_scene_set_projection(pny_scene* s, int mode) {
    if (!s) return fail(PNY_ERR_ARG, "pny_scene_set_projection: null scene");
    if (mode != PNY_PROJECTION_OFF && mode != PNY_PROJECTION_ON && mode != PNY_PROJECTION_AUTO)
        return fail(PNY_ERR_ARG, "pny_scene_set_projection: mode must be PNY_PROJECTION_{OFF,ON,AUTO}");
    s->zp_mode = mode;
    return PNY_OK;
}

int pny_scene_set_precision(pny_scene* s, int mode) {
    if (!s) return fail(PNY_ERR_ARG, "pny_scene_set_precision: null scene");
    if (mode != PNY_PRECISION_F32 && mode != PNY_PRECISION_F16X2 && mode != PNY_PRECISION_AUTO)
        return fail(PNY_ERR_ARG, "pny_scene_set_precision: mode must be PNY_PRECISION_{F32,F16X2,AUTO}");
    if ((mode == PNY_PRECISION_F32) != (s->precision == PNY_PRECISION_F32)) s->zp_valid[0] = s->zp_valid[1] = false;   // re-project in the new arithmetic
    s->precision = mode;
    return PNY_OK;
}

int pny_model_range_status(pny_model* m, unsigned* bits, int clear) {
    if (!m) return fail(PNY_ERR_ARG, "pny_model_range_status: null model");
    const unsigned b = range_bits(m);
    if (bits) *bits = b;
    if (b & PNY_RANGE_WEIGHT) m->f16_weights_ok = false;   // until the next finalize re-checks on the host
    if (clear && m->range_flag) __atomic_store_n(m->range_flag, 0u, __ATOMIC_RELAXED);
    return PNY_OK;
}

int pny_scene_last_precision(pny_scene* s, int* f16x2) {
    if (!s || !f16x2) return fail(PNY_ERR_ARG, "pny_scene_last_precision: null argument");
    *f16x2 = s->last_f16x2 ? 1 : 0;
    return PNY_OK;
}

int pny_scene_project(pny_scene* s, pny_stream stream) {
    int rc;
    if ((rc = check_ready(s, "pny_scene_project"))) return rc;
    if (!s->m->has_zproj) return PNY_OK;  // no per-view blocks: nothing to project
    if (s->zp_mode == PNY_PROJECTION_OFF) return fail(PNY_ERR_STATE, "pny_scene_project: projection is switched off for this scene");
    PNY_HIP(hipSetDevice(s->m->desc.device));
    if ((rc = enter_stream(s, (hipStream_t)stream))) return rc;
    const int keep = s->zp_mode;
    s->zp_mode = PNY_PROJECTION_ON;
    const float* zp = nullptr;
    rc = ensure_projection(s, 0, 0, (hipStream_t)stream, &zp);
    if (!rc && s->m->desc.has_fine && s->m->use_fine) rc = ensure_projection(s, 1, 0, (hipStream_t)stream, &zp);
    s->zp_mode = keep;
    if (rc) return rc;
    PNY_HIP(hipGetLastError());
    return PNY_OK;
}

int pny_scene_last_mlp_stats(pny_scene* s, double* flops, double* flops_reference, double* kernel_ms, int* launches,
                             int* projected) {
    if (!s) return fail(PNY_ERR_ARG, "pny_scene_last_mlp_stats: null scene");
    if (flops) *flops = s->last_flops;
    if (flops_reference) *flops_reference = s->last_flops_ref;
    if (projected) *projected = s->last_projected ? 1 : 0;
    if (launches) *launches = s->last_launches;
    if (kernel_ms) {
        double tot = 0.0;
        for (int i = 0; i + 1 < s->ev_used; i += 2) {
            PNY_HIP(hipEventSynchronize(s->ev[i + 1]));
            float ms = 0.f;
            PNY_HIP(hipEventElapsedTime(&ms, s->ev[i], s->ev[i + 1]));
            tot += ms;
        }
        *kernel_ms = s->timing ? tot : -1.0;
    }
    return PNY_OK;
}

}  // extern "C"
