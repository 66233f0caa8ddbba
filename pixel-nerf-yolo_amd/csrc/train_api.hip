// Training entry points of libpnyolo.so (include/pnyolo.h, "backward" section): the gradient the reference obtains with
// loss.backward() through NeRFRenderer.forward / PixelNeRFNet.forward (reference train/trainlib/PixelNerfTrainer.py:133-156,
// src/render/nerf.py:169-309, src/model/resnetfc.py:134-186).  Host-side C++: stash sizing, the weight-gradient GEMM
// work list and the launch sequence; all arithmetic is in mlp.hip (STASH forward) and mlp_bwd.hip.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "api_internal.h"

using namespace pny;

namespace pny {
StashLayout stash_layout(const pny_model_desc& d, int ns, int L) {
    const int nb = d.n_blocks, nvb = view_blocks(d), npost = nb - nvb;
    StashLayout l;
    l.x_in = 0;
    l.x_z = STASH_SMALL;
    l.x_act = STASH_SMALL + L * 64;
    l.x_view = l.x_act + 2 * nvb * STASH_SLOT;
    l.x_post = ns * l.x_view;
    l.x_tile = (long long)l.x_post + (long long)(2 * npost + 1) * STASH_SLOT;
    l.dy_view = 2 * nvb * STASH_SLOT;
    l.dy_post = ns * l.dy_view;
    l.dy_tile = (long long)l.dy_post + STASH_SMALL + (long long)(1 + 2 * npost) * STASH_SLOT;
    return l;
}
}  // namespace pny

namespace {

struct TrainPlan {
    StashLayout lay;
    std::vector<DwJob> jobs;
    std::vector<DwTarget> targets;  // same order as jobs
};

float* grad_of(const pny_model* m, const std::string& name) {
    auto it = m->grads.find(name);
    return it == m->grads.end() ? nullptr : it->second;
}

// Stash layout and the list of weight-gradient GEMMs of one MLP (pny_common.h StashLayout for the slot order).
TrainPlan build_plan(const pny_model* m, int ns, int L, const std::string& pre) {
    const pny_model_desc& d = m->desc;
    const int nb = d.n_blocks, nvb = view_blocks(d), npost = nb - nvb;
    const int d_in = 3 + 6 * d.num_freqs + 3;
    TrainPlan p;
    p.lay = stash_layout(d, ns, L);
    StashLayout& l = p.lay;
    auto xact = [&](int i) { return (long long)l.x_act + (long long)i * STASH_SLOT; };
    auto xpost = [&](int i) { return (long long)l.x_post + (long long)i * STASH_SLOT; };
    auto dyv = [&](int i) { return (long long)i * STASH_SLOT; };
    const long long draw = l.dy_post;
    auto dypost = [&](int i) { return (long long)l.dy_post + STASH_SMALL + (long long)i * STASH_SLOT; };
    const long long dhm = npost > 0 ? dypost(2) : dypost(0);
    auto add = [&](long long a_off, int a_view, int a_rows, long long x_off, int x_view, int x_cols, int n_views,
                   const std::string& wname, int rows_valid, int cols_valid, const std::string& b0, const std::string& b1) {
        DwJob j;
        j.a_off = a_off;
        j.a_view = a_view;
        j.a_rows = a_rows;
        j.x_off = x_off;
        j.x_view = x_view;
        j.x_cols = x_cols;
        j.n_views = n_views;
        DwTarget t;
        memset(&t, 0, sizeof(t));
        t.w = grad_of(m, wname);
        t.b0 = b0.empty() ? nullptr : grad_of(m, b0);
        t.b1 = b1.empty() ? nullptr : grad_of(m, b1);
        t.rows = rows_valid;
        t.cols = cols_valid;
        t.prows = a_rows;
        t.pcols = x_cols;
        p.jobs.push_back(j);
        p.targets.push_back(t);
    };
    auto blk = [&](int b) { return pre + "blocks." + std::to_string(b); };
    auto lz = [&](int b) { return pre + "lin_z." + std::to_string(b); };
    // lin_out: dY = d_raw (16-row slot, d_out rows valid), X = relu(h_top)
    add(draw, 0, D_IN_PAD, xpost(2 * npost), 0, HID, 1, pre + "lin_out.weight", d.d_out, HID, pre + "lin_out.bias", "");
    for (int b = nvb; b < nb; ++b) {  // post-combine blocks
        const int i = b - nvb;
        add(dypost(1 + 2 * i), 0, HID, xpost(2 * i), 0, HID, 1, blk(b) + ".fc_0.weight", HID, HID, blk(b) + ".fc_0.bias", "");
        add(b == nb - 1 ? dypost(0) : dypost(2 + 2 * (i + 1)), 0, HID, xpost(2 * i + 1), 0, HID, 1, blk(b) + ".fc_1.weight", HID, HID,
            blk(b) + ".fc_1.bias", "");
    }
    for (int b = 0; b < nvb; ++b) {  // per-view blocks: reduce over the views too
        add(dyv(2 * b), l.dy_view, HID, xact(2 * b), l.x_view, HID, ns, blk(b) + ".fc_0.weight", HID, HID, blk(b) + ".fc_0.bias", "");
        const bool last = b == nvb - 1;
        add(last ? dhm : dyv(2 * (b + 1) + 1), last ? 0 : l.dy_view, HID, xact(2 * b + 1), l.x_view, HID, ns,
            blk(b) + ".fc_1.weight", HID, HID, blk(b) + ".fc_1.bias", last ? std::string() : lz(b + 1) + ".bias");
        add(dyv(2 * b + 1), l.dy_view, HID, l.x_z, l.x_view, L, ns, lz(b) + ".weight", HID, L, "", "");
    }
    // lin_in: dY = the gradient at the first block's entry (dhm when the mean follows lin_in directly)
    add(nvb > 0 ? dyv(1) : dhm, nvb > 0 ? l.dy_view : 0, HID, l.x_in, l.x_view, D_IN_PAD, ns, pre + "lin_in.weight", HID, d_in,
        pre + "lin_in.bias", nvb > 0 ? lz(0) + ".bias" : std::string());
    return p;
}

// Split every job's (tile, view) range over workgroups so that the grid has ~4 workgroups per CU with equal work.
void build_items(TrainPlan& p, int n_tiles, int cus, std::vector<DwItem>& items, long long* part_floats, long long* bias_floats,
                 int* n_full) {
    items.clear();
    double total = 0.0;
    std::vector<int> otiles(p.jobs.size());
    for (size_t j = 0; j < p.jobs.size(); ++j) {
        const DwJob& jb = p.jobs[j];
        otiles[j] = ((jb.a_rows + 255) / 256) * ((jb.x_cols + 255) / 256);
        total += (double)otiles[j] * jb.n_views * n_tiles;
    }
    double items_per_cu = 4.0;   // PNYOLO_DW_ITEMS_PER_CU: tuning knob (finer items = shorter tail, more partial sums)
    if (const char* e = getenv("PNYOLO_DW_ITEMS_PER_CU")) {
        const double v = atof(e);
        if (v >= 1.0 && v <= 64.0) items_per_cu = v;
    }
    const double per_item = std::max(1.0, total / (items_per_cu * cus));
    long long poff = 0, boff = 0;
    for (size_t j = 0; j < p.jobs.size(); ++j) {
        const DwJob& jb = p.jobs[j];
        const int tv = jb.n_views * n_tiles;
        int splits = (int)std::lround((double)tv / per_item);
        splits = std::max(1, std::min(splits, tv));
        const int per = (tv + splits - 1) / splits;
        splits = (tv + per - 1) / per;
        DwTarget& t = p.targets[j];
        t.splits = splits;
        t.part_off = poff;
        t.bias_off = boff;
        const int mts = (jb.a_rows + 255) / 256, nts = (jb.x_cols + 255) / 256;
        for (int sp = 0; sp < splits; ++sp)
            for (int mt = 0; mt < mts; ++mt)
                for (int nt = 0; nt < nts; ++nt) {
                    DwItem it;
                    it.job = (int)j;
                    it.mt = mt;
                    it.nt = nt;
                    it.tv_lo = sp * per;
                    it.tv_hi = std::min(tv, (sp + 1) * per);
                    it.part_off = poff + (long long)sp * jb.a_rows * jb.x_cols;
                    it.bias_off = boff + (long long)sp * jb.a_rows;
                    items.push_back(it);
                }
        poff += (long long)splits * jb.a_rows * jb.x_cols;
        boff += (long long)splits * jb.a_rows;
    }
    // complete 256 x 256 tiles first (they run the predicate-free instantiation), clipped ones after; within each group
    // longest first, so that the tail of the grid is made of short items
    auto full = [&](const DwItem& it) {
        const DwJob& jb = p.jobs[it.job];
        return (it.mt + 1) * 256 <= jb.a_rows && (it.nt + 1) * 256 <= jb.x_cols;
    };
    std::stable_sort(items.begin(), items.end(), [&](const DwItem& a, const DwItem& b) {
        const bool fa = full(a), fb = full(b);
        if (fa != fb) return fb;   // clipped tiles (lin_in's 64 columns, lin_out's 64 rows) first: they are mostly staging
        return (a.tv_hi - a.tv_lo) > (b.tv_hi - b.tv_lo);
    });
    *n_full = 0;
    for (const DwItem& it : items) *n_full += full(it) ? 1 : 0;
    // XCD-aware placement of the complete tiles: the four output tiles (mt, nt) of one K split read the same dY rows and X
    // columns twice each.  Consecutive workgroup ids go round-robin to the 8 XCDs (each with its own L2), so the four tiles of a
    // split are given ids 8 apart -- same XCD, dispatched together -- and the second read of an operand slab is an L2 hit instead
    // of another trip over the fabric.  (Groups of other sizes, e.g. lin_z at L = 1792 with 2 x 7 tiles, keep list order.)
    if (!getenv("PNYOLO_DW_NO_XCD_GROUPS")) {
        const size_t first = items.size() - (size_t)*n_full;
        std::vector<DwItem> out(items.begin(), items.begin() + first);
        std::vector<std::vector<DwItem>> groups;   // pending groups of exactly 4 tiles
        auto flush = [&]() {
            if (groups.size() == 8) {
                for (int j = 0; j < 4; ++j)
                    for (int g = 0; g < 8; ++g) out.push_back(groups[g][j]);
            } else {
                for (auto& g : groups) out.insert(out.end(), g.begin(), g.end());
            }
            groups.clear();
        };
        size_t i = first;
        while (i < items.size()) {
            size_t j = i + 1;
            while (j < items.size() && j - i < 4 && items[j].job == items[i].job && items[j].tv_lo == items[i].tv_lo) ++j;
            if (j - i == 4) {
                groups.emplace_back(items.begin() + i, items.begin() + j);
                if (groups.size() == 8) flush();
            } else {
                flush();
                out.insert(out.end(), items.begin() + i, items.begin() + j);
            }
            i = j;
        }
        flush();
        items.swap(out);
    }
    *part_floats = poff;
    *bias_floats = boff;
}

// Matrix arithmetic of the backward's GEMMs (dX chain and weight gradients): the scene's precision setting (F32 pins the fp32
// MFMA; anything else = split f16: mlp_bwd_h2.hip pny_mlp_bwd_h2_kernel, mlp_bwd.hip pny_dw_gemm_h2_kernel); env
// PNYOLO_BWD_PRECISION=f32|f16x2 overrides (read at every call: tests vary it).
bool bwd_use_h2(const pny_scene* s) {
    if (const char* e = getenv("PNYOLO_BWD_PRECISION")) {
        if (!strcmp(e, "f32")) return false;
        if (!strcmp(e, "f16x2")) return true;
    }
    return s->precision != PNY_PRECISION_F32;
}

// A zeroed device word (or two) for the chain kernels' atomic max
int ensure_absmax(DevBuf& b, size_t bytes) {
    if (b.p) return 0;
    int rc;
    if ((rc = b.reserve(bytes))) return rc;
    PNY_HIP(hipMemset(b.p, 0, bytes));
    return 0;
}

size_t stash_budget_bytes() {
    size_t v = (size_t)16 << 30;  // both stashes together; PNYOLO_STASH_GB overrides (read at every call: tests vary it)
    if (const char* e = getenv("PNYOLO_STASH_GB")) {
        const double g = atof(e);
        if (g > 0.001) v = (size_t)(g * (double)((size_t)1 << 30));
    }
    return v;
}

// Weight-gradient GEMMs over n_tiles tiles of the two stashes + deterministic split reduction into the bound gradients.
// dy_absmax: the chain kernels' running max |dY| of these tiles (device), or null for the fp32 matrix path
int run_weight_grads(pny_model* m, TrainPlan& plan, int n_tiles, const float* x_stash, const float* dy_stash, DevBuf& partial,
                     DevBuf& bias, DevBuf& tables, PinnedStage& stage, int accumulate, hipStream_t st,
                     const unsigned* dy_absmax = nullptr) {
    if (!m->aux_stream) {
        PNY_HIP(hipStreamCreateWithFlags(&m->aux_stream, hipStreamNonBlocking));
        PNY_HIP(hipEventCreateWithFlags(&m->aux_fork, hipEventDisableTiming));
        PNY_HIP(hipEventCreateWithFlags(&m->aux_join, hipEventDisableTiming));
    }
    const int cus = mlp_max_grid(MLP_8x64);
    std::vector<DwItem> items;
    long long part_floats = 0, bias_floats = 0;
    int n_full = 0;
    build_items(plan, n_tiles, cus, items, &part_floats, &bias_floats, &n_full);
    int rc;
    if ((rc = partial.reserve((size_t)part_floats * sizeof(float)))) return rc;
    if ((rc = bias.reserve((size_t)bias_floats * sizeof(float)))) return rc;
    const size_t jb_bytes = plan.jobs.size() * sizeof(DwJob), it_bytes = items.size() * sizeof(DwItem),
                 tg_bytes = plan.targets.size() * sizeof(DwTarget);
    const size_t o_items = (jb_bytes + 255) & ~(size_t)255, o_targets = o_items + ((it_bytes + 255) & ~(size_t)255);
    const size_t total = o_targets + tg_bytes;
    // NOTE: a grown device table is only safe because growth happens before any kernel of this call uses it and after
    // the previous call's kernels (same stream) were enqueued: hipFree of DevBuf::reserve synchronises the device
    if ((rc = tables.reserve(total))) return rc;
    if ((rc = stage.prepare(total))) return rc;
    char* h = reinterpret_cast<char*>(stage.host);
    memcpy(h, plan.jobs.data(), jb_bytes);
    memcpy(h + o_items, items.data(), it_bytes);
    memcpy(h + o_targets, plan.targets.data(), tg_bytes);
    if ((rc = stage.upload(tables.p, total, st))) return rc;
    char* tb = reinterpret_cast<char*>(tables.p);
    launch_dw_gemm(reinterpret_cast<const DwJob*>(tb), reinterpret_cast<const DwItem*>(tb + o_items), (int)items.size() - n_full,
                   n_full, x_stash, dy_stash, plan.lay.x_tile, plan.lay.dy_tile, partial.f(), bias.f(), st, m->aux_stream,
                   m->aux_fork, m->aux_join, dy_absmax);
    PNY_HIP(hipGetLastError());
    long long max_elems = 0;
    for (const DwTarget& t : plan.targets) max_elems = std::max(max_elems, (long long)t.rows * t.cols + t.rows);
    launch_dw_reduce(reinterpret_cast<const DwTarget*>(tb + o_targets), (int)plan.targets.size(), max_elems, partial.f(), bias.f(),
                     accumulate, st);
    PNY_HIP(hipGetLastError());
    return 0;
}

// Backward of one MLP evaluation over n_points query points (mode 0: xyz / dirs; mode 1: rays + z with K samples per
// ray): d_out (n_points, d_out) -> bound parameter gradients.  Points are processed in chunks that fit the stash.
// dz_sel / dz_out (mode 1 only, optional): per ray kfd sample indices (ray * K + position, or -1) whose depth gradient
// through the MLP inputs is added to dz_out (n_points).
int mlp_backward(pny_scene* s, int mode, const float* xyz, const float* dirs, const float* rays, const float* z, int K,
                 long long n_points, int coarse, const float* d_out, int accumulate, hipStream_t st,
                 const int* dz_sel = nullptr, int kfd = 0, float* dz_out = nullptr,
                 const pny_scene::StashedPass* stashed = nullptr, const float* fwd_out = nullptr, bool immediate = false) {
    if (n_points == 0) return 0;
    pny_model* m = s->m;
    // forward already stashed by pny_render (same reservation epoch): no recompute, the tiles are in the model-level stash
    const bool have_x = !immediate && stashed && stashed->valid && m->defer && stashed->epoch == m->defer_epoch &&
                        stashed->n_points == n_points && fwd_out;
    const bool defer = m->defer && !immediate;
    if (s->n_objs > 1 && !have_x && !defer)   // (the chunked recompute path would split an object's share)
        return fail(PNY_ERR_STATE, "grouped scene: the backward needs the deferred stash (pny_model_defer_weight_grads)");
    const pny_model_desc& d = m->desc;
    const bool fine_w = !(coarse || !d.has_fine || !m->use_fine);
    const std::string pre = fine_w ? "mlp_fine." : "mlp_coarse.";
    TrainPlan plan = build_plan(m, obj_views(s), s->L, pre);
    const size_t tile_bytes = (size_t)(plan.lay.x_tile + plan.lay.dy_tile) * sizeof(float);
    long long max_tiles = (long long)(stash_budget_bytes() / tile_bytes);
    if (max_tiles < 1) return fail(PNY_ERR_ARG, "stash budget smaller than one tile");
    // chunk boundaries on whole rays (mode 1) so that sample -> ray indexing stays local to the chunk
    const long long unit = mode == 1 ? K : 1;
    long long chunk_pts = std::min(n_points, max_tiles * 64);
    chunk_pts = std::max(unit, chunk_pts / unit * unit);
    const int cus = mlp_max_grid(MLP_8x64);
    int rc;
    const long long chunk_tiles_max = (chunk_pts + 63) / 64;
    // Deferred mode: this call's tiles are appended to the model-level stash of its MLP; the weight-gradient GEMM runs
    // once over every scene's tiles at pny_model_flush_weight_grads
    const int which = fine_w ? 1 : 0;
    float* x_base = nullptr;
    float* dy_base = nullptr;
    if (have_x) {
        chunk_pts = n_points;
        x_base = m->dx_stash[stashed->which].f() + stashed->tile0 * plan.lay.x_tile;
        dy_base = m->ddy_stash[stashed->which].f() + stashed->tile0 * plan.lay.dy_tile;
    } else if (defer) {
        const long long tiles = (n_points + 63) / 64;
        if (obj_views(s) != m->defer_ns) return fail(PNY_ERR_STATE, "deferred weight gradients: scene view count differs from the reservation");
        if (m->defer_used[which] + tiles > m->defer_cap[which])
            return fail(PNY_ERR_STATE, "deferred weight gradients: more tiles than pny_model_defer_weight_grads reserved");
        chunk_pts = n_points;   // one chunk: the reservation was made against the budget
        x_base = m->dx_stash[which].f() + m->defer_used[which] * plan.lay.x_tile;
        dy_base = m->ddy_stash[which].f() + m->defer_used[which] * plan.lay.dy_tile;
        m->defer_used[which] += tiles;
    } else {
        if ((rc = s->x_stash.reserve((size_t)chunk_tiles_max * plan.lay.x_tile * sizeof(float)))) return rc;
        if ((rc = s->dy_stash.reserve((size_t)chunk_tiles_max * plan.lay.dy_tile * sizeof(float)))) return rc;
        x_base = s->x_stash.f();
        dy_base = s->dy_stash.f();
    }
    if ((rc = s->out_tmp.reserve((size_t)chunk_pts * d.d_out * sizeof(float)))) return rc;
    // running max |dY| for the split-f16 weight-gradient GEMM: per model and MLP in deferred mode (every scene's chain adds
    // to it, zeroed again by the flush), per scene otherwise (zeroed in front of every chunk's chain)
    const bool dw_h2 = bwd_use_h2(s);
    unsigned* absmax = nullptr;
    if (defer || have_x) {
        if (!dw_h2) m->defer_dw_f32 = true;
        if ((rc = ensure_absmax(m->d_absmax, 2 * sizeof(unsigned)))) return rc;
        absmax = reinterpret_cast<unsigned*>(m->d_absmax.p) + which;
    } else if (dw_h2) {
        if ((rc = ensure_absmax(s->dy_absmax, sizeof(unsigned)))) return rc;
        absmax = reinterpret_cast<unsigned*>(s->dy_absmax.p);
    }
    auto stamp = [&]() -> int {   // kernel timing for bench.py (pny_scene_enable_timing)
        if (!s->timing) return 0;
        if ((int)s->bev.size() <= s->bev_used) {
            hipEvent_t e;
            PNY_HIP(hipEventCreate(&e));
            s->bev.push_back(e);
        }
        PNY_HIP(hipEventRecord(s->bev[s->bev_used++], st));
        return 0;
    };
    {   // GEMM FLOPs (2 per MAC, unpadded) of the three kernels, per query point
        const int nvb_ = view_blocks(d), npost_ = d.n_blocks - nvb_, d_in_ = 3 + 6 * d.num_freqs + 3;
        const double per_view_f = (double)d_in_ * HID + (double)nvb_ * s->L * HID + 2.0 * nvb_ * HID * HID;
        const double post_f = 2.0 * npost_ * HID * HID + (double)HID * d.d_out;
        const double fwd = 2.0 * (obj_views(s) * per_view_f + post_f);
        const double chain = 2.0 * (obj_views(s) * 2.0 * nvb_ * HID * HID + 2.0 * npost_ * HID * HID + (double)HID * d.d_out);
        if (!have_x) s->bwd_flops[0] += fwd * (double)n_points;   // the forward already stashed: nothing is recomputed
        s->bwd_flops[1] += chain * (double)n_points;
        s->bwd_flops[2] += fwd * (double)n_points;   // every forward GEMM has one weight-gradient GEMM of the same size
    }
    const float* zp_maps = nullptr;
    if (dz_sel && view_blocks(d) > 0) {
        if ((rc = ensure_projection(s, fine_w ? 1 : 0, 0, st, &zp_maps, true))) return rc;
        if (!zp_maps) return fail(PNY_ERR_ARG, "sample-depth gradients need the projected latent maps (latent too large)");
    }
    for (long long p0 = 0; p0 < n_points; p0 += chunk_pts) {
        const long long np = std::min(chunk_pts, n_points - p0);
        const int n_tiles = (int)((np + 63) / 64);
        // 1. forward in the reference's operation order, stashing every GEMM's B operand
        MlpArgs a;
        if ((rc = fill_mlp_args(s, mode, mode == 0 ? xyz + 3 * p0 : nullptr, mode == 0 ? dirs + 3 * p0 : nullptr,
                                mode == 1 ? rays + (p0 / K) * 8 : nullptr, mode == 1 ? z + p0 : nullptr, K, np, coarse,
                                s->out_tmp.f(), &a)))
            return rc;
        a.stash_x = x_base;
        a.lay = plan.lay;
        const int grid = std::min(cus, n_tiles);
        if ((rc = stamp())) return rc;
        if (!have_x) {
            launch_mlp_stash(a, grid, st);
            PNY_HIP(hipGetLastError());
        }
        if ((rc = stamp())) return rc;
        // 2. dX chain
        const MlpWeightsT& wt = fine_w ? m->fine_t : m->coarse_t;
        BwdArgs b;
        memset(&b, 0, sizeof(b));
        b.wT_out = wt.wT_out;
        for (int i = 0; i < d.n_blocks; ++i) {
            b.wT_fc0[i] = wt.wT_fc0[i];
            b.wT_fc1[i] = wt.wT_fc1[i];
        }
        b.w_base = m->packed.f();
        b.w_bytes = (unsigned)m->packed.bytes;
        b.x_stash = x_base;
        b.dy_stash = dy_base;
        b.lay = plan.lay;
        b.out = have_x ? fwd_out : s->out_tmp.f();
        b.d_out_grad = d_out + p0 * d.d_out;
        b.n_points = np;
        b.n_tiles = n_tiles;
        b.NS = obj_views(s);
        b.n_blocks = d.n_blocks;
        b.combine_layer = d.combine_layer;
        b.d_out = d.d_out;
        b.yolo = d.yolo;
        b.dy_absmax = absmax;
        b.range_flag = m->range_flag;
        if (absmax && !(defer || have_x)) PNY_HIP(hipMemsetAsync(absmax, 0, sizeof(unsigned), st));
        if (dw_h2 && m->f16_weights_ok) {   // split-f16 chain (weights beyond the f16 range: fp32 chain)
            b.h2T_out = wt.h2T_out;
            for (int i = 0; i < d.n_blocks; ++i) {
                b.h2T_fc0[i] = wt.h2T_fc0[i];
                b.h2T_fc1[i] = wt.h2T_fc1[i];
            }
            launch_mlp_bwd_h2(b, grid, st);
        } else {
            launch_mlp_bwd(b, grid, st);
        }
        PNY_HIP(hipGetLastError());
        if ((rc = stamp())) return rc;
        // 2b. gradient w.r.t. the depths of the selected samples through the MLP inputs (fine pass of a render)
        if (dz_sel && mode == 1) {
            DzArgs dz;
            memset(&dz, 0, sizeof(dz));
            dz.dy_stash = dy_base;
            dz.lay = plan.lay;
            dz.sel = dz_sel + (p0 / K) * kfd;
            dz.n_sel = (int)((np / K) * kfd);
            dz.p0 = p0;
            dz.rays = rays;
            dz.z = z;
            dz.K = K;
            dz.w_in = wt.w_in_plain;
            dz.d_in = 3 + 6 * d.num_freqs + 3;
            dz.zp = zp_maps;
            dz.zp_stride = view_blocks(d) * HID;
            dz.NS = obj_views(s);
            dz.obj_pts = a.obj_pts;
            dz.Hl = s->hl;
            dz.Wl = s->wl;
            dz.nvb = view_blocks(d);
            dz.npost = d.n_blocks - view_blocks(d);
            dz.yolo = d.yolo;
            dz.num_freqs = d.num_freqs;
            dz.freq_factor = d.freq_factor;
            dz.sx = a.sx;
            dz.sy = a.sy;
            dz.dz = dz_out;
            memcpy(dz.cams, s->cams, sizeof(Cam) * (size_t)s->ns);
            launch_mlp_dz(dz, st);
            PNY_HIP(hipGetLastError());
        }
        // 2c. gradient w.r.t. the latent (encoder training): lin_z^T GEMM off the dY stash + scatter into the taps
        if (s->latent_grad && view_blocks(d) > 0) {
            if (!wt.wzT_cat) return fail(PNY_ERR_STATE, "latent gradient: transposed lin_z weights are missing");
            if (s->L % 256) return fail(PNY_ERR_ARG, "latent gradient: d_latent must be a multiple of 256");
            launch_latent_grad(a, dy_base, plan.lay, wt.wzT_cat, s->latent_grad, view_blocks(d), st, dw_h2 ? absmax : nullptr);
            PNY_HIP(hipGetLastError());
        }
        // 3. weight-gradient GEMMs over the two stashes + deterministic split reduction into the bound gradients
        if (!defer && !have_x &&
            (rc = run_weight_grads(m, plan, n_tiles, x_base, dy_base, s->dw_partial, s->dw_bias, s->dw_tables, s->table_stage,
                                   (accumulate || p0 > 0) ? 1 : 0, st, absmax)))
            return rc;
        if ((rc = stamp())) return rc;
    }
    return 0;
}

}  // namespace

extern "C" {

int pny_scene_bind_latent_grad(pny_scene* s, float* grad_dev) {
    if (!s) return fail(PNY_ERR_ARG, "pny_scene_bind_latent_grad: null scene");
    s->latent_grad = grad_dev;
    return PNY_OK;
}

int pny_model_bind_grad(pny_model* m, const char* name, float* grad_dev) {
    if (!m || !name) return fail(PNY_ERR_ARG, "pny_model_bind_grad: null argument");
    if (grad_dev)
        m->grads[name] = grad_dev;
    else
        m->grads.erase(name);
    return PNY_OK;
}

int pny_model_defer_weight_grads(pny_model* m, int enable, int ns, int64_t coarse_tiles, int64_t fine_tiles) {
    if (!m) return fail(PNY_ERR_ARG, "pny_model_defer_weight_grads: null model");
    m->defer = false;
    m->defer_used[0] = m->defer_used[1] = 0;
    ++m->defer_epoch;   // passes stashed under the previous reservation are no longer valid
    if (!enable) return PNY_OK;
    if (ns < 1 || ns > MAX_VIEWS || coarse_tiles < 0 || fine_tiles < 0) return fail(PNY_ERR_ARG, "pny_model_defer_weight_grads: bad argument");
    PNY_HIP(hipSetDevice(m->desc.device));
    TrainPlan plan = build_plan(m, ns, m->desc.d_latent, "mlp_coarse.");   // the layout does not depend on the MLP
    const size_t tile_bytes = (size_t)(plan.lay.x_tile + plan.lay.dy_tile) * sizeof(float);
    if ((size_t)(coarse_tiles + fine_tiles) * tile_bytes > stash_budget_bytes())
        return fail(PNY_ERR_ARG, "pny_model_defer_weight_grads: reservation exceeds the stash budget (PNYOLO_STASH_GB)");
    const int64_t tiles[2] = {coarse_tiles, fine_tiles};
    int rc;
    for (int w = 0; w < 2; ++w) {
        if ((rc = m->dx_stash[w].reserve((size_t)tiles[w] * plan.lay.x_tile * sizeof(float)))) return rc;
        if ((rc = m->ddy_stash[w].reserve((size_t)tiles[w] * plan.lay.dy_tile * sizeof(float)))) return rc;
        m->defer_cap[w] = tiles[w];
    }
    m->defer_ns = ns;
    m->defer = true;
    ++m->defer_epoch;
    return PNY_OK;
}

int pny_model_flush_weight_grads(pny_model* m, int accumulate, pny_stream stream) {
    if (!m) return fail(PNY_ERR_ARG, "pny_model_flush_weight_grads: null model");
    if (!m->defer) return fail(PNY_ERR_STATE, "pny_model_flush_weight_grads: not in deferred mode");
    PNY_HIP(hipSetDevice(m->desc.device));
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (!(accumulate & 16)) {   // (a coarse-only flush follows the fine-only one of the same step: keep its numbers)
        m->flush_flops = 0.0;
        m->flush_launches = 0;
    }
    for (int i = 0; i < 4; ++i)
        if (!m->flush_ev[i]) PNY_HIP(hipEventCreate(&m->flush_ev[i]));
    const pny_model_desc& d = m->desc;
    const int nvb_ = view_blocks(d), npost_ = d.n_blocks - nvb_, d_in_ = 3 + 6 * d.num_freqs + 3;
    const double per_view_f = (double)d_in_ * HID + (double)nvb_ * d.d_latent * HID + 2.0 * nvb_ * HID * HID;
    const double fwd = 2.0 * (m->defer_ns * per_view_f + 2.0 * npost_ * HID * HID + (double)HID * d.d_out);
    // bits 16 / 32 of `accumulate`: only mlp_coarse's / only mlp_fine's stash (0: both)
    const bool only_c = (accumulate & 16) != 0, only_f = (accumulate & 32) != 0;
    accumulate &= 1;
    for (int w = 0; w < 2; ++w) {
        if ((w == 0 && only_f) || (w == 1 && only_c)) continue;
        PNY_HIP(hipEventRecord(m->flush_ev[2 * w], st));
        if (m->defer_used[w] > 0) {
            TrainPlan plan = build_plan(m, m->defer_ns, d.d_latent, w ? "mlp_fine." : "mlp_coarse.");
            const unsigned* absmax = (m->d_absmax.p && !m->defer_dw_f32) ? reinterpret_cast<const unsigned*>(m->d_absmax.p) + w : nullptr;
            if ((rc = run_weight_grads(m, plan, (int)m->defer_used[w], m->dx_stash[w].f(), m->ddy_stash[w].f(), m->d_partial[w],
                                       m->d_bias[w], m->d_tables[w], m->d_stage[w], accumulate, st, absmax)))
                return rc;
            m->flush_flops += fwd * 64.0 * (double)m->defer_used[w];
            m->flush_launches += 1;
        }
        PNY_HIP(hipEventRecord(m->flush_ev[2 * w + 1], st));
        m->defer_used[w] = 0;
        if (m->d_absmax.p)   // the next step's chains start from 0
            PNY_HIP(hipMemsetAsync(reinterpret_cast<unsigned*>(m->d_absmax.p) + w, 0, sizeof(unsigned), st));
    }
    if (!only_f) m->defer_dw_f32 = false;   // (a fine-only flush is followed by the coarse-only one of the same step)
    return PNY_OK;
}

int pny_model_last_flush_stats(pny_model* m, double* flops, double* kernel_ms) {
    if (!m) return fail(PNY_ERR_ARG, "pny_model_last_flush_stats: null model");
    if (flops) *flops = m->flush_flops;
    if (kernel_ms) {
        *kernel_ms = 0.0;
        for (int w = 0; w < 2 && m->flush_ev[0]; ++w) {
            PNY_HIP(hipEventSynchronize(m->flush_ev[2 * w + 1]));
            float ms = 0.f;
            PNY_HIP(hipEventElapsedTime(&ms, m->flush_ev[2 * w], m->flush_ev[2 * w + 1]));
            *kernel_ms += ms;
        }
    }
    return PNY_OK;
}

int pny_scene_stash_next_render(pny_scene* s, int enable) {
    if (!s) return fail(PNY_ERR_ARG, "pny_scene_stash_next_render: null scene");
    if (enable && !s->m->defer) return fail(PNY_ERR_STATE, "pny_scene_stash_next_render: reserve the stash first (pny_model_defer_weight_grads)");
    s->stash_next = enable != 0;
    return PNY_OK;
}

int pny_query_backward(pny_scene* s, const float* xyz_dev, const float* viewdirs_dev, int64_t n, int coarse,
                       const float* d_out_dev, int accumulate, pny_stream stream) {
    int rc;
    if ((rc = check_ready(s, "pny_query_backward"))) return rc;
    if (n < 0 || (n > 0 && (!xyz_dev || !viewdirs_dev || !d_out_dev))) return fail(PNY_ERR_ARG, "pny_query_backward: bad argument");
    PNY_HIP(hipSetDevice(s->m->desc.device));
    if ((rc = enter_stream(s, (hipStream_t)stream))) return rc;
    s->bev_used = 0;
    s->bwd_flops[0] = s->bwd_flops[1] = s->bwd_flops[2] = 0.0;
    // immediate: a deferred reservation that a training pny_render left outstanding belongs to that render's backward; this
    // call's weight gradients go straight into the buffers bound now
    return mlp_backward(s, 0, xyz_dev, viewdirs_dev, nullptr, nullptr, 1, n, coarse, d_out_dev, accumulate & 1, (hipStream_t)stream,
                        nullptr, 0, nullptr, nullptr, nullptr, true);
}

int pny_composite_backward(const float* rays_dev, const float* z_dev, const float* sample_dev, int64_t n, int k, int white_bkgd,
                           const float* g_rgb_dev, const float* g_depth_dev, const float* g_weights_dev, float* d_sample_dev,
                           float* d_z_dev, pny_stream stream) {
    if (n < 0 || k < 1 || (n > 0 && (!rays_dev || !z_dev || !sample_dev || !d_sample_dev)))
        return fail(PNY_ERR_ARG, "pny_composite_backward: bad argument");
    if ((size_t)4 * 2 * k * sizeof(float) > 64 * 1024) return fail(PNY_ERR_ARG, "pny_composite_backward: too many samples per ray");
    launch_composite_bwd(rays_dev, z_dev, sample_dev, nullptr, n, k, white_bkgd, g_rgb_dev, g_depth_dev, g_weights_dev, d_sample_dev,
                         d_z_dev, (hipStream_t)stream);
    PNY_HIP(hipGetLastError());
    return PNY_OK;
}

int pny_render_backward(pny_scene* s, const float* rays_dev, int64_t n, const pny_render_opts* o, const pny_render_saved* sv,
                        const pny_render_grads* g, int accumulate, pny_stream stream) {
    int rc;
    if ((rc = check_ready(s, "pny_render_backward"))) return rc;
    if (!o || !sv || !g || n < 0 || (n > 0 && !rays_dev)) return fail(PNY_ERR_ARG, "pny_render_backward: bad argument");
    if (s->m->desc.yolo || s->m->desc.d_out != 4) return fail(PNY_ERR_ARG, "pny_render_backward: model is in YOLO mode");
    if (n == 0) return PNY_OK;
    const int kc = o->n_coarse, kt = o->n_coarse + o->n_fine;
    if (!sv->z_coarse || !sv->sample_coarse || (o->n_fine > 0 && (!sv->z_fine || !sv->sample_fine)))
        return fail(PNY_ERR_ARG, "pny_render_backward: the forward call's z / per-sample outputs are required");
    if ((size_t)4 * 2 * kt * sizeof(float) > 64 * 1024) return fail(PNY_ERR_ARG, "pny_render_backward: too many samples per ray");
    PNY_HIP(hipSetDevice(s->m->desc.device));
    hipStream_t st = (hipStream_t)stream;
    if ((rc = enter_stream(s, st))) return rc;
    s->bev_used = 0;
    s->bwd_flops[0] = s->bwd_flops[1] = s->bwd_flops[2] = 0.0;
    if ((rc = s->d_samp.reserve((size_t)n * kt * 4 * sizeof(float)))) return rc;
    const bool same_mlp = !s->m->desc.has_fine || !s->m->use_fine;  // both passes differentiate mlp_coarse
    const bool immediate = (accumulate & 2) != 0;   // ignore a deferred reservation that belongs to another forward
    // bits 4 / 8: only the fine / only the coarse pass of this backward (the caller runs every scene's fine pass, starts the
    // fine MLP's weight-gradient flush beside the coarse passes, then the coarse passes: pny_model_flush_weight_grads)
    const bool do_fine = (accumulate & 8) == 0, do_coarse = (accumulate & 4) == 0;
    bool first = true;
    const bool any_f = o->n_fine > 0 && (g->rgb_fine || g->depth_fine || g->weights_fine);
    const bool any_c = g->rgb_coarse || g->depth_coarse || g->weights_coarse;
    // The fine pass's depth samples are centred on the (attached) coarse depth (nerf.py:156-167, 296-298): the fine
    // loss reaches mlp_coarse through those samples' positions.  dL/dz of the fine samples = composite part + MLP-input
    // part; summed over a ray's unclamped depth samples it is an extra dL/d(depth_coarse).
    const int kfd = o->n_fine_depth;
    const bool depth_path = any_f && kfd > 0 && sv->depth_coarse;
    {   // a pass that was stashed by the forward but receives no gradient must not leave stale dY tiles for the flush
        pny_model* m = s->m;
        const long long dy_tile = stash_layout(m->desc, obj_views(s), s->L).dy_tile;
        auto zero_pass = [&](const pny_scene::StashedPass& sp) -> int {
            if (sp.valid && m->defer && !immediate && sp.epoch == m->defer_epoch)
                PNY_HIP(hipMemsetAsync(m->ddy_stash[sp.which].f() + sp.tile0 * dy_tile, 0, (size_t)sp.tiles * dy_tile * sizeof(float), st));
            return 0;
        };
        if (do_fine && !any_f && (rc = zero_pass(s->stashed[1]))) return rc;
        if (do_coarse && !(any_c || depth_path) && (rc = zero_pass(s->stashed[0]))) return rc;
    }
    const float* g_depth_c = g->depth_coarse;
    if (any_f && !do_fine) {   // coarse-only call: the fine pass of this backward ran in an earlier call and left the depth path's sum
        if (depth_path) g_depth_c = s->gdepth_tmp.f();
        first = false;
    }
    if (any_f && do_fine) {
        float* dz = nullptr;
        int* sel = nullptr;
        if (depth_path) {
            if ((rc = s->dz_tmp.reserve((size_t)n * kt * sizeof(float)))) return rc;
            if ((rc = s->sel_tmp.reserve((size_t)n * kfd * sizeof(int)))) return rc;
            if ((rc = s->gdepth_tmp.reserve((size_t)n * sizeof(float)))) return rc;
            dz = s->dz_tmp.f();
            sel = reinterpret_cast<int*>(s->sel_tmp.p);
            launch_locate_depth_samples(rays_dev, sv->depth_coarse, o->g_depth_dev, o->seed, sv->z_fine, n, kt, kfd, o->depth_std, sel, st);
        }
        launch_composite_bwd(rays_dev, sv->z_fine, sv->sample_fine, o->sigma_noise_fine_dev, n, kt, o->white_bkgd, g->rgb_fine,
                             g->depth_fine, g->weights_fine, s->d_samp.f(), dz, st);
        PNY_HIP(hipGetLastError());
        if ((rc = mlp_backward(s, 1, nullptr, nullptr, rays_dev, sv->z_fine, kt, (long long)n * kt, 0, s->d_samp.f(), accumulate & 1, st,
                               sel, kfd, dz, &s->stashed[1], sv->sample_fine, immediate)))
            return rc;
        if (depth_path) {
            launch_depth_grad_gather(sel, dz, g->depth_coarse, n, kfd, s->gdepth_tmp.f(), st);
            PNY_HIP(hipGetLastError());
            g_depth_c = s->gdepth_tmp.f();
        }
        first = false;
    }
    if (do_coarse && (any_c || depth_path)) {
        launch_composite_bwd(rays_dev, sv->z_coarse, sv->sample_coarse, o->sigma_noise_coarse_dev, n, kc, o->white_bkgd, g->rgb_coarse,
                             g_depth_c, g->weights_coarse, s->d_samp.f(), nullptr, st);
        PNY_HIP(hipGetLastError());
        if ((rc = mlp_backward(s, 1, nullptr, nullptr, rays_dev, sv->z_coarse, kc, (long long)n * kc, 1, s->d_samp.f(),
                               ((accumulate & 1) || (same_mlp && !first)) ? 1 : 0, st, nullptr, 0, nullptr, &s->stashed[0],
                               sv->sample_coarse, immediate)))
            return rc;
    }
    return PNY_OK;
}

int pny_yolo_render_backward(pny_scene* s, const float* rays_dev, int64_t n, int n_coarse, const float* u_coarse_dev, uint64_t seed,
                             const float* raw_dev, const float* g_out_dev, int accumulate, pny_stream stream) {
    int rc;
    if ((rc = check_ready(s, "pny_yolo_render_backward"))) return rc;
    const pny_model_desc& d = s->m->desc;
    if (!d.yolo || d.d_out % 7) return fail(PNY_ERR_ARG, "pny_yolo_render_backward: model is not in YOLO mode");
    if (n < 0 || n_coarse < 1 || (n > 0 && (!rays_dev || !raw_dev || !g_out_dev))) return fail(PNY_ERR_ARG, "pny_yolo_render_backward: bad argument");
    if (n == 0) return PNY_OK;
    PNY_HIP(hipSetDevice(d.device));
    hipStream_t st = (hipStream_t)stream;
    if ((rc = enter_stream(s, st))) return rc;
    s->bev_used = 0;
    s->bwd_flops[0] = s->bwd_flops[1] = s->bwd_flops[2] = 0.0;
    const size_t nz = ((size_t)n * n_coarse + 63) & ~(size_t)63;
    if ((rc = s->d_samp.reserve((size_t)n * n_coarse * d.d_out * sizeof(float)))) return rc;
    if ((rc = s->dz_tmp.reserve(nz * sizeof(float)))) return rc;
    float* z = s->dz_tmp.f();
    launch_sample_coarse(rays_dev, n, n_coarse, 0, u_coarse_dev, seed, z, st);   // the forward's depths (same draws)
    launch_yolo_aggregate_bwd(raw_dev, g_out_dev, n, n_coarse, d.d_out / 7, s->d_samp.f(), st);
    PNY_HIP(hipGetLastError());
    return mlp_backward(s, 1, nullptr, nullptr, rays_dev, z, n_coarse, (long long)n * n_coarse, 1, s->d_samp.f(), accumulate & 1, st,
                        nullptr, 0, nullptr, nullptr, nullptr, true);   // immediate (see pny_query_backward)
}

int pny_scene_last_backward_stats(pny_scene* s, double flops[3], double kernel_ms[3]) {
    if (!s) return fail(PNY_ERR_ARG, "pny_scene_last_backward_stats: null scene");
    for (int i = 0; i < 3; ++i) {
        if (flops) flops[i] = s->bwd_flops[i];
        if (kernel_ms) kernel_ms[i] = 0.0;
    }
    if (kernel_ms && s->timing) {
        // events come in groups of 4 per chunk: before the stash forward, after it, after the chain, after the GEMMs + reduce
        // (the chunk's sample-depth kernel, when present, is counted with the GEMMs)
        for (int i = 0; i + 3 < s->bev_used; i += 4) {
            PNY_HIP(hipEventSynchronize(s->bev[i + 3]));
            for (int k = 0; k < 3; ++k) {
                float ms = 0.f;
                PNY_HIP(hipEventElapsedTime(&ms, s->bev[i + k], s->bev[i + k + 1]));
                kernel_ms[k] += ms;
            }
        }
    }
    return PNY_OK;
}

}  // extern "C"
