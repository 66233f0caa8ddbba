// YOLO detection tail of the eval path, on the device so that
// eval_yolo / YoloTrainer.metric_step no longer round-trip every view through Python lists:
//   cells_to_bboxes   reference src/util/util.py:633-689  convert_cells_to_bboxes
//   nms               reference src/util/util.py:691-722  nms  (+ iou :575-611)
//   tp_fp_fn          reference src/util/util.py:765-802  calculate_tp_fp_fn
// The reference's semantics are kept exactly, including two accidents a drop-in must reproduce:
//   * nms() removes items from the list it is iterating, so the element that follows a removed one
//     is skipped (kept for this round);
//   * confidence / size filters compare Python floats (double), IoU tests compare fp32 tensors with
//     the threshold rounded to fp32.
// These are tiny, latency-bound, integer/compare workloads (<= a few thousand boxes per view): one
// workgroup, index lists in LDS, boxes read from L2.
#include "pny_common.h"

namespace pny {

// iou(box1, box2) of reference util.py:576-611 (is_pred=True), [x, y, w, h] boxes, fp32 op for op.
__device__ __forceinline__ float iou_xywh(const float* a, const float* b) {
    const float a_x1 = a[0] - a[2] / 2.0f, a_y1 = a[1] - a[3] / 2.0f;
    const float a_x2 = a[0] + a[2] / 2.0f, a_y2 = a[1] + a[3] / 2.0f;
    const float b_x1 = b[0] - b[2] / 2.0f, b_y1 = b[1] - b[3] / 2.0f;
    const float b_x2 = b[0] + b[2] / 2.0f, b_y2 = b[1] + b[3] / 2.0f;
    const float x1 = fmaxf(a_x1, b_x1), y1 = fmaxf(a_y1, b_y1);
    const float x2 = fminf(a_x2, b_x2), y2 = fminf(a_y2, b_y2);
    const float inter = fmaxf(x2 - x1, 0.0f) * fmaxf(y2 - y1, 0.0f);
    const float area_a = fabsf((a_x2 - a_x1) * (a_y2 - a_y1));
    const float area_b = fabsf((b_x2 - b_x1) * (b_y2 - b_y1));
    const float uni = area_a + area_b - inter;
    return inter / (uni + 1e-6f);
}

// ------------------------------------------------------------------ cells -> boxes
__global__ void cells_to_bboxes_kernel(const float* __restrict__ cells, int h, int w, int na, int is_pred,
                                       float ax0, float ay0, float ax1, float ay1, float ax2, float ay2,
                                       float ax3, float ay3, float inv_w, float inv_h, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h * w * na) return;
    const int a = i % na, x = (i / na) % w, y = i / (na * w);
    const int C = is_pred ? 7 : 6;
    const float* c = cells + (size_t)i * C;
    float bx = c[1], by = c[2], bw = c[3], bh = c[4], cls;
    if (is_pred) {
        const float aw = a == 0 ? ax0 : a == 1 ? ax1 : a == 2 ? ax2 : ax3;
        const float ah = a == 0 ? ay0 : a == 1 ? ay1 : a == 2 ? ay2 : ay3;
        bx = 1.0f / (1.0f + expf(-bx));
        by = 1.0f / (1.0f + expf(-by));
        bw = expf(bw) * aw;
        bh = expf(bh) * ah;
        // torch.argmax over the class logits (first maximum wins)
        int best = 0;
        float bv = c[5];
        for (int k = 1; k < C - 5; ++k)
            if (c[5 + k] > bv) {
                bv = c[5 + k];
                best = k;
            }
        cls = (float)best;
    } else {
        cls = c[5];
    }
    float* o = out + (size_t)i * 6;
    o[0] = cls;
    o[1] = c[0];
    o[2] = inv_w * (bx + (float)x);
    o[3] = inv_h * (by + (float)y);
    o[4] = inv_w * bw;
    o[5] = inv_h * bh;
}

// ------------------------------------------------------------------ nms
constexpr int NMS_THREADS = 256;
constexpr int NMS_MAX = 8192;  // boxes per call (LDS: 3 int/float arrays of this length)

// Block-wide order-preserving compaction of `keep` over positions [0, n): writes the kept source
// values src[p] to dst[0..m) and returns m.  dst may alias src only if processed in order (it does
// not here).  scan_buf: NMS_THREADS ints of LDS.
__device__ int block_compact(const int* src, const unsigned char* keep, int n, int* dst, int* scan_buf) {
    const int tid = threadIdx.x;
    const int per = (n + NMS_THREADS - 1) / NMS_THREADS;
    const int lo = tid * per, hi = min(n, lo + per);
    int cnt = 0;
    for (int p = lo; p < hi; ++p) cnt += keep[p] ? 1 : 0;
    scan_buf[tid] = cnt;
    __syncthreads();
    if (tid == 0) {  // 256-entry exclusive scan; trivial next to the rest
        int run = 0;
        for (int t = 0; t < NMS_THREADS; ++t) {
            const int v = scan_buf[t];
            scan_buf[t] = run;
            run += v;
        }
        scan_buf[NMS_THREADS] = run;
    }
    __syncthreads();
    int o = scan_buf[tid];
    for (int p = lo; p < hi; ++p)
        if (keep[p]) dst[o++] = src[p];
    const int total = scan_buf[NMS_THREADS];
    __syncthreads();
    return total;
}

// One workgroup.  boxes (n,6) = [class, score, x, y, w, h].  Writes the surviving boxes in the
// reference's output order to kept (n,6); meta[0] = count kept, meta[1] = boxes above the confidence
// threshold; hc[0] = highest confidence of all input boxes.
__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const float* __restrict__ boxes, int n, float iou_thr,
                                                          double conf_thr, float* __restrict__ kept,
                                                          int* __restrict__ meta, float* __restrict__ hc) {
    extern __shared__ int lds_i[];
    int* listA = lds_i;                     // [n]
    int* listB = lds_i + n;                 // [n]
    int* scan_buf = lds_i + 2 * n;          // [NMS_THREADS + 1]
    float* redf = reinterpret_cast<float*>(scan_buf + NMS_THREADS + 1);     // [NMS_THREADS]
    int* redi = reinterpret_cast<int*>(redf + NMS_THREADS);                  // [NMS_THREADS]
    unsigned char* flag = reinterpret_cast<unsigned char*>(redi + NMS_THREADS);  // [n]
    const int tid = threadIdx.x;

    // highest confidence over ALL boxes, count above threshold, filter (reference util.py:693-700)
    float mx = -INFINITY;
    int above = 0;
    for (int i = tid; i < n; i += NMS_THREADS) {
        const float* b = boxes + (size_t)i * 6;
        mx = fmaxf(mx, b[1]);
        const bool conf_ok = (double)b[1] > conf_thr;
        above += conf_ok ? 1 : 0;
        const double bw = (double)b[4], bh = (double)b[5];
        flag[i] = (conf_ok && 10e-4 < bw && bw < 10e4 && 10e-4 < bh && bh < 10e4) ? 1 : 0;
        listA[i] = i;
    }
    redf[tid] = mx;
    redi[tid] = above;
    __syncthreads();
    if (tid == 0) {
        float m2 = -INFINITY;
        int a2 = 0;
        for (int t = 0; t < NMS_THREADS; ++t) {
            m2 = fmaxf(m2, redf[t]);
            a2 += redi[t];
        }
        hc[0] = m2;
        meta[1] = a2;
    }
    __syncthreads();
    int m = block_compact(listA, flag, n, listB, scan_buf);  // listB = filtered indices, input order

    // stable sort by confidence, descending (Python sorted(..., reverse=True) keeps input order on ties)
    for (int p = tid; p < m; p += NMS_THREADS) {
        const int i = listB[p];
        const float ci = boxes[(size_t)i * 6 + 1];
        int rank = 0;
        for (int q = 0; q < m; ++q) {
            const float cq = boxes[(size_t)listB[q] * 6 + 1];
            rank += (cq > ci || (cq == ci && q < p)) ? 1 : 0;
        }
        listA[rank] = i;
    }
    __syncthreads();

    // greedy suppression with the reference's list semantics (util.py:709-720)
    int* cur = listA;
    int* nxt = listB;
    int n_out = 0;
    while (m > 0) {
        const int first = cur[0];
        if (tid < 6) kept[(size_t)n_out * 6 + tid] = boxes[(size_t)first * 6 + tid];
        ++n_out;
        const float* fb = boxes + (size_t)first * 6 + 2;
        for (int p = 1 + tid; p < m; p += NMS_THREADS)
            flag[p] = iou_xywh(fb, boxes + (size_t)cur[p] * 6 + 2) > iou_thr ? 1 : 0;
        __syncthreads();
        if (tid == 0) {
            // `for box in lst: if ...: lst.remove(box)`: removing shifts the tail left under the
            // iterator, so the element after a removed one is never examined in this round.
            // `lst.remove(box)` deletes the FIRST element that compares equal: an identical row that is still in the list
            // further up (it can only be there because the iterator skipped it) goes instead of the row at hand.  Rows are
            // sorted by confidence, so such a twin sits in the run of equal confidences right above.
            flag[0] = 0;
            int p = 1;
            while (p < m) {
                if (flag[p]) {
                    int drop = p;
                    const float* bp = boxes + (size_t)cur[p] * 6;
                    for (int q = p - 1; q >= 1; --q) {
                        const float* bq = boxes + (size_t)cur[q] * 6;
                        if (bq[1] != bp[1]) break;
                        // position q is still in the list iff it was kept so far in this round (flag 1 = keep, set below)
                        if (flag[q] == 1 && bq[0] == bp[0] && bq[2] == bp[2] && bq[3] == bp[3] && bq[4] == bp[4] && bq[5] == bp[5])
                            drop = q;
                    }
                    flag[p] = 1;
                    flag[drop] = 0;  // 0 = drop
                    if (p + 1 < m) flag[p + 1] = 1;
                    p += 2;
                } else {
                    flag[p] = 1;  // 1 = keep
                    p += 1;
                }
            }
        }
        __syncthreads();
        m = block_compact(cur, flag, m, nxt, scan_buf);
        int* t = cur;
        cur = nxt;
        nxt = t;
    }
    if (tid == 0) meta[0] = n_out;
}

static size_t nms_lds_bytes(int n) {
    return (size_t)(2 * n + NMS_THREADS + 1) * sizeof(int) + (size_t)NMS_THREADS * (sizeof(float) + sizeof(int)) + (size_t)n;
}

// ------------------------------------------------------------------ tp / fp / fn
// reference util.py:779-802 on the two suppressed lists.
__global__ __launch_bounds__(256) void match_kernel(const float* __restrict__ tgt, const int* __restrict__ tmeta,
                                                    const float* __restrict__ prd, const int* __restrict__ pmeta,
                                                    float match_iou, int* __restrict__ out) {
    __shared__ int cnt[3];
    const int nt = tmeta[0], np = pmeta[0];
    if (threadIdx.x < 3) cnt[threadIdx.x] = 0;
    __syncthreads();
    if (nt == 0) {
        if (threadIdx.x == 0) cnt[1] = np;
    } else if (np == 0) {
        if (threadIdx.x == 0) cnt[2] = nt;
    } else {
        for (int p = threadIdx.x; p < np; p += blockDim.x) {
            float best = -INFINITY;
            for (int t = 0; t < nt; ++t) best = fmaxf(best, iou_xywh(prd + (size_t)p * 6 + 2, tgt + (size_t)t * 6 + 2));
            atomicAdd(&cnt[best > match_iou ? 0 : 1], 1);
        }
        for (int t = threadIdx.x; t < nt; t += blockDim.x) {
            float best = -INFINITY;
            for (int p = 0; p < np; ++p) best = fmaxf(best, iou_xywh(tgt + (size_t)t * 6 + 2, prd + (size_t)p * 6 + 2));
            if (best < match_iou) atomicAdd(&cnt[2], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) out[threadIdx.x] = cnt[threadIdx.x];
}

}  // namespace pny

using namespace pny;

extern "C" {

int pny_cells_to_bboxes(const float* cells_dev, const float* anchors_host, int h, int w, int n_anchors,
                        int is_predictions, float* boxes_dev, pny_stream stream) {
    if (!cells_dev || !boxes_dev || h < 1 || w < 1 || n_anchors < 1 || n_anchors > 4 || (is_predictions && !anchors_host)) {
        set_error("pny_cells_to_bboxes: bad argument (1 <= n_anchors <= 4)");
        return PNY_ERR_ARG;
    }
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (anchors_host)
        for (int i = 0; i < 2 * n_anchors; ++i) a[i] = anchors_host[i];
    const int tot = h * w * n_anchors;
    // 1/w as the reference forms it: Python double 1/w rounded to fp32 == fp32 1/w (both correctly rounded)
    hipLaunchKernelGGL(cells_to_bboxes_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, cells_dev, h,
                       w, n_anchors, is_predictions, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7],
                       (float)(1.0 / (double)w), (float)(1.0 / (double)h), boxes_dev);
    PNY_HIP(hipGetLastError());
    return PNY_OK;
}

int pny_nms(const float* boxes_dev, int n, double iou_threshold, double threshold, float* kept_dev, int* meta_dev,
            float* highest_conf_dev, pny_stream stream) {
    if (n < 0 || n > NMS_MAX || !meta_dev || !highest_conf_dev || (n > 0 && (!boxes_dev || !kept_dev))) {
        set_error("pny_nms: bad argument (0 <= n <= 8192)");
        return PNY_ERR_ARG;
    }
    const size_t lds = nms_lds_bytes(n > 0 ? n : 1);
    static size_t max_set = 0;
    if (lds > max_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        max_set = lds;
    }
    hipLaunchKernelGGL(nms_kernel, dim3(1), dim3(NMS_THREADS), lds, (hipStream_t)stream, boxes_dev, n,
                       (float)iou_threshold, threshold, kept_dev, meta_dev, highest_conf_dev);
    PNY_HIP(hipGetLastError());
    return PNY_OK;
}

int pny_tp_fp_fn(const float* target_boxes_dev, int nt, const float* pred_boxes_dev, int np, double nms_iou,
                 double nms_threshold, double match_iou, int* out_dev, pny_stream stream) {
    if (nt < 0 || np < 0 || nt > NMS_MAX || np > NMS_MAX || !out_dev || (nt > 0 && !target_boxes_dev) ||
        (np > 0 && !pred_boxes_dev)) {
        set_error("pny_tp_fp_fn: bad argument");
        return PNY_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    // scratch: kept lists + meta + hc, stream-ordered
    const size_t fl = (size_t)(nt + np) * 6 + 16;
    float* scratch = nullptr;
    PNY_HIP(hipMallocAsync((void**)&scratch, fl * sizeof(float) + 64, st));
    float* kt = scratch;
    float* kp = scratch + (size_t)nt * 6;
    int* tmeta = reinterpret_cast<int*>(scratch + (size_t)(nt + np) * 6);
    int* pmeta = tmeta + 2;
    float* hcs = reinterpret_cast<float*>(pmeta + 2);
    int rc = pny_nms(target_boxes_dev, nt, nms_iou, nms_threshold, kt, tmeta, hcs, stream);
    if (!rc) rc = pny_nms(pred_boxes_dev, np, nms_iou, nms_threshold, kp, pmeta, hcs + 1, stream);
    if (!rc) {
        hipLaunchKernelGGL(match_kernel, dim3(1), dim3(256), 0, st, kt, tmeta, kp, pmeta, (float)match_iou, out_dev);
        if (hipGetLastError() != hipSuccess) rc = PNY_ERR_HIP;
    }
    (void)hipFreeAsync(scratch, st);
    return rc;
}

}  // extern "C"
