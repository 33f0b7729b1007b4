// Recall@K / IoU tables on the device: the kept rows of every query are already in HBM
// after fusion + NMS, so a validation run ends with its numbers without a python loop over queries.
//   mode 0  standalone_eval/evaluate_ego4d_nlq.py:41-60,93-103  numpy float64: inter and union clamped at 0,
//           plain division (0/0 = nan, x/0 = inf; `nan > thr` is False);
//   mode 1  standalone_eval/evaluate_mad.py:33-38,87-104        torch float32: the python doubles are rounded
//           to fp32 first, inter clamped, union not clamped, thresholds compared as fp32.
// `bools[:K].any()` == (index of the first row over the threshold) < K, so one pass over the rows serves
// every K.  Counts are exact integers (atomics), the divisions by the number of queries happen on the host in
// the reference's dtype.  Compiled with -ffp-contract=off.
#include "common.h"

namespace cone {

constexpr int kMaxThr = 8, kMaxTopk = 16;
struct RecallParams {
    double thr[kMaxThr];
    int topk[kMaxTopk];
    int n_thr, n_topk;
};

__global__ __launch_bounds__(256) void eval_recall_kernel(const double* __restrict__ rows,
                                                          const int* __restrict__ n, const double* __restrict__ gt,
                                                          int nq, int A, RecallParams p, int mode,
                                                          unsigned long long* hits, double* top1) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const int cnt = min(n[q], A);
    int first[kMaxThr];
#pragma unroll
    for (int t = 0; t < kMaxThr; ++t) first[t] = 0x7fffffff;
    const double g0 = gt[2 * q], g1 = gt[2 * q + 1];
    double iou0 = 0.0;
    for (int i = 0; i < cnt; ++i) {
        const double ps = rows[((size_t)q * A + i) * 5], pe = rows[((size_t)q * A + i) * 5 + 1];
        if (mode == 0) {
            const double inter = fmax(0.0, fmin(pe, g1) - fmax(ps, g0));
            const double uni = fmax(0.0, fmax(pe, g1) - fmin(ps, g0));
            const double iou = inter / uni;
            if (i == 0) iou0 = iou;
#pragma unroll
            for (int t = 0; t < kMaxThr; ++t)
                if (t < p.n_thr && iou > p.thr[t] && first[t] == 0x7fffffff) first[t] = i;
        } else {
            const float s = (float)ps, e = (float)pe, gs = (float)g0, ge = (float)g1;
            const float inter = fminf(e, ge) - fmaxf(s, gs);
            const float uni = fmaxf(e, ge) - fminf(s, gs);
            const float iou = (inter < 0.f ? 0.f : inter) / uni;      // clamp(min=0) keeps nan
            if (i == 0) iou0 = (double)iou;
#pragma unroll
            for (int t = 0; t < kMaxThr; ++t)
                if (t < p.n_thr && iou > (float)p.thr[t] && first[t] == 0x7fffffff) first[t] = i;
        }
    }
    top1[q] = cnt > 0 ? iou0 : __builtin_nan("");
    for (int t = 0; t < p.n_thr; ++t)
        for (int r = 0; r < p.n_topk; ++r)
            if (first[t] < p.topk[r]) atomicAdd(&hits[t * p.n_topk + r], 1ull);
}

// standalone_eval/evaluate_pre_filtered_window.py:45-66: target windows floor(start/S) .. ceil(end/S)
// (start, end in clips = seconds / clip_length, python doubles); hit if one of the first K ranked windows is among them.
__global__ __launch_bounds__(256) void eval_window_recall_kernel(const int* __restrict__ win_idx, int nq, int K,
                                                                 const double* __restrict__ gt, double clip_length,
                                                                 int S, RecallParams p, unsigned long long* hits) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const double start = gt[2 * q] / clip_length, end = gt[2 * q + 1] / clip_length;
    const double lo = floor(start / (double)S), hi = ceil(end / (double)S) + 1.0;
    int first = 0x7fffffff;
    for (int i = 0; i < K; ++i) {
        const int w = win_idx[(size_t)q * K + i];
        if (w < 0) break;
        if ((double)w >= lo && (double)w < hi) { first = i; break; }
    }
    for (int r = 0; r < p.n_topk; ++r)
        if (first < p.topk[r]) atomicAdd(&hits[r], 1ull);
}

static int fill_params(RecallParams& p, const double* thr, int n_thr, const int32_t* topk, int n_topk) {
    CONE_REQUIRE(n_thr >= 0 && n_thr <= kMaxThr && n_topk >= 1 && n_topk <= kMaxTopk && topk && (thr || n_thr == 0),
                 "eval: at most %d thresholds and %d K values (got %d, %d)", kMaxThr, kMaxTopk, n_thr, n_topk);
    p.n_thr = n_thr; p.n_topk = n_topk;
    for (int i = 0; i < n_thr; ++i) p.thr[i] = thr[i];
    for (int i = 0; i < n_topk; ++i) p.topk[i] = topk[i];
    return 0;
}

}  // namespace cone

extern "C" int cone_eval_recall(const double* rows, const int32_t* n, const double* gt, int nq, int max_after,
                                const double* thresholds, int n_thr, const int32_t* topk, int n_topk, int mode,
                                int64_t* hits, double* top1_iou, void* stream) {
    CONE_REQUIRE(rows && n && gt && hits && top1_iou && nq > 0 && max_after > 0 && (mode == 0 || mode == 1),
                 "eval_recall: bad arguments");
    cone::RecallParams p;
    int rc = cone::fill_params(p, thresholds, n_thr, topk, n_topk);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    CONE_CHECK_HIP(hipMemsetAsync(hits, 0, sizeof(int64_t) * n_thr * n_topk, s));
    hipLaunchKernelGGL(cone::eval_recall_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, rows, n, gt, nq, max_after,
                       p, mode, (unsigned long long*)hits, top1_iou);
    CONE_LAUNCH_CHECK();
    return 0;
}

extern "C" int cone_eval_window_recall(const int32_t* win_idx, int nq, int k, const double* gt, double clip_length,
                                       int slide, const int32_t* topk, int n_topk, int64_t* hits, void* stream) {
    CONE_REQUIRE(win_idx && gt && hits && nq > 0 && k > 0 && slide > 0 && clip_length > 0,
                 "eval_window_recall: bad arguments");
    cone::RecallParams p;
    int rc = cone::fill_params(p, nullptr, 0, topk, n_topk);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    CONE_CHECK_HIP(hipMemsetAsync(hits, 0, sizeof(int64_t) * n_topk, s));
    hipLaunchKernelGGL(cone::eval_window_recall_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, win_idx, nq, k, gt,
                       clip_length, slide, p, (unsigned long long*)hits);
    CONE_LAUNCH_CHECK();
    return 0;
}
