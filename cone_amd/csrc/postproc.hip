// Stage C: 4-decimal rounding, min-max score fusion, (st,ed) dict collapse, stable sort and greedy
// temporal NMS -- one workgroup per query, all arithmetic in fp64 exactly like the reference's
// Python floats (cone/inference.py:83,103-127,205-217; utils/temporal_nms.py:6-74).
//
// Exactness notes
//   * float(f"{x:.4f}") for an fp32 x: x*1e4 is exact in fp64 (24-bit x 14-bit significands), so
//     rint(x*1e4)/1e4 (round-half-even, correctly rounded division) is the same double that Python's
//     correctly-rounded format + parse produces.
//   * sum((a, b)) in Python is (0 + a) + b.
//   * sorted(..., reverse=True) is stable: rank = #{greater} + #{equal with smaller position}.
#include "common.h"

namespace cone {

constexpr int kMaxCand = 1024;

__device__ __forceinline__ double round4(double x) { return rint(x * 1e4) / 1e4; }

__device__ __forceinline__ double pseudo_iou(double s0, double e0, double s1, double e1) {
    const double lo = fmin(e0, e1) - fmax(s0, s1);
    const double inter = lo > 0.0 ? lo : 0.0;
    const double uni = fmax(e0, e1) - fmin(s0, s1);
    return uni == 0.0 ? 0.0 : inter / uni;
}

// Greedy NMS over `m` candidates already in score order (sidx[j] = candidate id at sorted position j).
// Block-cooperative; writes kept sorted positions.
__device__ void block_nms(const double* st, const double* ed, const int* sidx, int m, double thd,
                          int max_after, unsigned char* alive, int* kept_pos, int* kept_n_out) {
    const int tid = threadIdx.x;
    __shared__ int s_cur;
    __shared__ int s_kept;
    for (int j = tid; j < m; j += blockDim.x) alive[j] = 1;
    if (tid == 0) { s_kept = 0; s_cur = 0; }
    __syncthreads();
    if (m == 1) {  // utils/temporal_nms.py:38-39
        if (tid == 0) { kept_pos[0] = 0; *kept_n_out = 1; }
        __syncthreads();
        return;
    }
    int start = 0;
    while (true) {
        if (tid == 0) {
            int c = start;
            while (c < m && !alive[c]) ++c;
            s_cur = c;
            if (c < m && s_kept < max_after) kept_pos[s_kept++] = c; else s_cur = m;
        }
        __syncthreads();
        const int cur = s_cur;
        if (cur >= m) break;
        const double s0 = st[sidx[cur]], e0 = ed[sidx[cur]];
        for (int j = cur + 1 + tid; j < m; j += blockDim.x)
            if (alive[j] && pseudo_iou(s0, e0, st[sidx[j]], ed[sidx[j]]) > thd) alive[j] = 0;
        start = cur + 1;
        __syncthreads();
    }
    if (tid == 0) *kept_n_out = s_kept;
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void fuse_nms_kernel(const T* __restrict__ cand, const int64_t* __restrict__ cand_off,
                                                       const int* __restrict__ n_valid, int nq, int n_max,
                                                       double thd, int max_before, int max_after,
                                                       double* out_rows, int* out_n, int* out_idx) {
    __shared__ double c_st[kMaxCand], c_ed[kMaxCand], c_val[3][kMaxCand];  // val: 0 prop, 1 match, 2 fused
    __shared__ int u_first[kMaxCand];   // unique entries in first-insertion order: candidate id of the key
    __shared__ int u_last[kMaxCand];    //   candidate id whose values the dict holds for that key
    __shared__ int sidx[kMaxCand];      // unique-entry ids in score order
    // kept_pos | cidx (NMS scratch of a score type) share their 8 KiB with uval (the unique entries' values while that type
    // is ranked: dead before the NMS starts) -- 64 KiB of static LDS is the limit
    __shared__ __attribute__((aligned(8))) int ibuf[2 * kMaxCand];
    int* kept_pos = ibuf;
    int* cidx = ibuf + kMaxCand;
    double* uval = reinterpret_cast<double*>(ibuf);
    __shared__ unsigned char flag[kMaxCand];
    __shared__ double red[4][4];
    __shared__ int s_nu, s_kept_n;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(n_valid[q], n_max);
    const T* cq = cand + (cand_off ? (size_t)cand_off[q] : (size_t)q * n_max) * 4;

    // 1. float(f"{e:.4f}")
    for (int i = tid; i < n; i += 256) {
        c_st[i] = round4((double)cq[i * 4]);
        c_ed[i] = round4((double)cq[i * 4 + 1]);
        c_val[0][i] = round4((double)cq[i * 4 + 2]);
        c_val[1][i] = round4((double)cq[i * 4 + 3]);
    }
    __syncthreads();
    // 2. normalize_score on both lists + fused = (0 + a) + b
    double mn0 = INFINITY, mx0 = -INFINITY, mn1 = INFINITY, mx1 = -INFINITY;
    for (int i = tid; i < n; i += 256) {
        mn0 = fmin(mn0, c_val[0][i]); mx0 = fmax(mx0, c_val[0][i]);
        mn1 = fmin(mn1, c_val[1][i]); mx1 = fmax(mx1, c_val[1][i]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        mn0 = fmin(mn0, __shfl_xor(mn0, o, 64)); mx0 = fmax(mx0, __shfl_xor(mx0, o, 64));
        mn1 = fmin(mn1, __shfl_xor(mn1, o, 64)); mx1 = fmax(mx1, __shfl_xor(mx1, o, 64));
    }
    if (lane == 0) { red[wave][0] = mn0; red[wave][1] = mx0; red[wave][2] = mn1; red[wave][3] = mx1; }
    __syncthreads();
    mn0 = fmin(fmin(red[0][0], red[1][0]), fmin(red[2][0], red[3][0]));
    mx0 = fmax(fmax(red[0][1], red[1][1]), fmax(red[2][1], red[3][1]));
    mn1 = fmin(fmin(red[0][2], red[1][2]), fmin(red[2][2], red[3][2]));
    mx1 = fmax(fmax(red[0][3], red[1][3]), fmax(red[2][3], red[3][3]));
    for (int i = tid; i < n; i += 256) {
        const double a = (mn0 == mx0) ? c_val[0][i] : (c_val[0][i] - mn0) / (mx0 - mn0);
        const double b = (mn1 == mx1) ? c_val[1][i] : (c_val[1][i] - mn1) / (mx1 - mn1);
        c_val[2][i] = (0.0 + a) + b;
    }
    // 3. dict keyed by (st, ed): first occurrence fixes the position, last occurrence the values.  One pass per candidate
    // over the whole list (loads independent of each other: they pipeline, no early exit), the unique entries compacted in
    // candidate order by ballot + prefix count instead of a serial walk by one thread (a single query is latency-bound
    // here: 45 -> ~15 us for 100 candidates)
    __shared__ int s_cnt[4][kMaxCand / 256];
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        bool first = i < n;
        int last = i;
        if (i < n) {
            const double si = c_st[i], ei = c_ed[i];
            for (int j = 0; j < n; ++j) {
                const bool same = c_st[j] == si && c_ed[j] == ei;
                first = first && !(same && j < i);
                last = same && j > last ? j : last;
            }
        }
        flag[i0 + tid] = first ? 1 : 0;             // (kMaxCand is a multiple of 256: in bounds)
        if (first) kept_pos[i] = last;              // parked: the value-holder of the key whose first occurrence is i
        const unsigned long long b = __ballot(first);
        if (lane == 0) s_cnt[wave][i0 >> 8] = __popcll(b);
    }
    __syncthreads();
    int nu_acc = 0;
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        int base = nu_acc;
        for (int w = 0; w < wave; ++w) base += s_cnt[w][i0 >> 8];
        const bool first = i < n && flag[i];
        const unsigned long long b = __ballot(first);
        if (first) {
            const int u = base + __popcll(b & ((1ull << lane) - 1ull));
            u_first[u] = i;
            u_last[u] = kept_pos[i];
        }
        nu_acc += s_cnt[0][i0 >> 8] + s_cnt[1][i0 >> 8] + s_cnt[2][i0 >> 8] + s_cnt[3][i0 >> 8];
    }
    if (tid == 0) s_nu = nu_acc;
    __syncthreads();
    const int nu = s_nu;

    // 4. per score type: stable descending order of the unique entries, truncate, NMS
    // gridDim.y == 3 (few queries: the launch is latency-bound): the three score types of a query run in three workgroups,
    // each repeating steps 1 - 3 for itself (the same arithmetic: the same outputs as one workgroup walking all three)
    const int order[3] = {2, 0, 1};  // fused, proposal, matching  (cone/inference.py:152-164)
    const int t_lo = gridDim.y == 3 ? (int)blockIdx.y : 0, t_hi = gridDim.y == 3 ? t_lo + 1 : 3;
    for (int t = t_lo; t < t_hi; ++t) {
        const double* val = c_val[order[t]];
        for (int u = tid; u < nu; u += 256) uval[u] = val[u_last[u]];
        __syncthreads();
        for (int u = tid; u < nu; u += 256) {
            const double v = uval[u];
            int rank = 0;
            for (int w = 0; w < nu; ++w) {          // (every thread reads the same address: LDS broadcast, independent loads)
                const double x = uval[w];
                rank += (x > v) || (x == v && w < u);
            }
            sidx[rank] = u;
        }
        __syncthreads();
        int* keep_n = out_n + (size_t)t * nq + q;
        double* rows = out_rows + ((size_t)t * nq + q) * max_after * 5;
        int* oidx = out_idx + ((size_t)t * nq + q) * max_after;
        int kept;
        if (thd != -1.0) {
            const int m = min(nu, max_before);
            // positions use the key's first-occurrence candidate for (st, ed)
            for (int j = tid; j < m; j += 256) cidx[j] = u_first[sidx[j]];
            __syncthreads();
            block_nms(c_st, c_ed, cidx, m, thd, max_after, flag, kept_pos, &s_kept_n);
            kept = s_kept_n;
        } else {
            kept = min(nu, max_after);
            for (int j = tid; j < kept; j += 256) kept_pos[j] = j;
            __syncthreads();
        }
        for (int j = tid; j < kept; j += 256) {
            const int u = sidx[kept_pos[j]];
            const int kf = u_first[u], kl = u_last[u];
            rows[j * 5] = c_st[kf]; rows[j * 5 + 1] = c_ed[kf];
            rows[j * 5 + 2] = c_val[0][kl]; rows[j * 5 + 3] = c_val[1][kl]; rows[j * 5 + 4] = c_val[2][kl];
            oidx[j] = kf;
        }
        for (int j = kept + tid; j < max_after; j += 256) {     // rows past the kept ones: zeros / -1 (the caller need not pre-fill)
            rows[j * 5] = 0.0; rows[j * 5 + 1] = 0.0; rows[j * 5 + 2] = 0.0; rows[j * 5 + 3] = 0.0; rows[j * 5 + 4] = 0.0;
            oidx[j] = -1;
        }
        if (tid == 0) *keep_n = kept;
        __syncthreads();
    }
}

// temporal_nms on one list of [st, ed, score] (fp64).
__global__ __launch_bounds__(256) void temporal_nms_kernel(const double* __restrict__ pred, int n, double thd,
                                                           int max_after, int* keep_idx, int* keep_n) {
    __shared__ double c_st[kMaxCand], c_ed[kMaxCand];
    __shared__ int sidx[kMaxCand], kept_pos[kMaxCand];
    __shared__ unsigned char alive[kMaxCand];
    __shared__ int s_kept_n;
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += 256) { c_st[i] = pred[i * 3]; c_ed[i] = pred[i * 3 + 1]; }
    if (n == 1) {  // returned untouched
        if (tid == 0) { keep_idx[0] = 0; *keep_n = 1; }
        return;
    }
    for (int i = tid; i < n; i += 256) {
        const double v = pred[i * 3 + 2];
        int rank = 0;
        for (int w = 0; w < n; ++w) {
            const double x = pred[w * 3 + 2];
            rank += (x > v) || (x == v && w < i);
        }
        sidx[rank] = i;
    }
    __syncthreads();
    block_nms(c_st, c_ed, sidx, n, thd, max_after, alive, kept_pos, &s_kept_n);
    for (int j = tid; j < s_kept_n; j += 256) keep_idx[j] = sidx[kept_pos[j]];
    if (tid == 0) *keep_n = s_kept_n;
}

// HungarianMatcher cost for one target span per window (cone/matcher.py:61-95, cone/span_utils.py).
__global__ __launch_bounds__(256) void matcher_cost_kernel(const float* __restrict__ logits,
                                                           const float* __restrict__ spans,
                                                           const float* __restrict__ tgt, int B, int Nq, float cs,
                                                           float cg, float cc, float* cost, int* best) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float tc = tgt[2 * b], tw = tgt[2 * b + 1];
    const float t1 = __fsub_rn(tc, __fmul_rn(0.5f, tw)), t2 = __fadd_rn(tc, __fmul_rn(0.5f, tw));
    float bestv = INFINITY;
    int besti = 0;
    for (int n = 0; n < Nq; ++n) {
        const int i = b * Nq + n;
        const float l0 = logits[2 * i], l1 = logits[2 * i + 1];
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        const float prob = e0 / (e0 + e1);
        const float c = spans[2 * i], w = spans[2 * i + 1];
        const float l1d = fabsf(c - tc) + fabsf(w - tw);  // cdist p=1
        const float x1 = __fsub_rn(c, __fmul_rn(0.5f, w)), x2 = __fadd_rn(c, __fmul_rn(0.5f, w));
        const float a1 = x2 - x1, a2 = t2 - t1;
        const float inter = fmaxf(fminf(x2, t2) - fmaxf(x1, t1), 0.f);
        const float uni = a1 + a2 - inter;
        const float iou = inter / uni;
        const float enc = fmaxf(fmaxf(x2, t2) - fminf(x1, t1), 0.f);
        const float giou = iou - (enc - uni) / enc;
        const float v = cs * l1d + cg * (-giou) + cc * (-prob);
        cost[i] = v;
        if (v < bestv) { bestv = v; besti = n; }
    }
    best[b] = besti;
}

}  // namespace cone

template <typename T>
static int fuse_nms_launch(const T* cand, const int64_t* cand_off, const int32_t* n_valid, int nq, int n_max, double nms_thd,
                           int max_before, int max_after, double* out_rows, int32_t* out_n, int32_t* out_idx,
                           void* stream) {
    CONE_REQUIRE(cand && n_valid && out_rows && out_n && out_idx, "fuse_nms: null argument");
    CONE_REQUIRE(n_max >= 1 && n_max <= cone::kMaxCand, "fuse_nms: n_max=%d not in [1,%d]", n_max,
                 cone::kMaxCand);
    CONE_REQUIRE(max_after >= 1 && max_after <= cone::kMaxCand && max_before >= 1, "fuse_nms: bad limits");
    if (nq <= 0) return 0;
    // up to 128 queries (one workgroup each would leave most of the chip idle and the launch bound by ONE workgroup's serial
    // walk over the three score types): a workgroup per (query, score type).  The outputs do not depend on the choice
    const unsigned ty = nq <= 128 ? 3u : 1u;
    hipLaunchKernelGGL(cone::fuse_nms_kernel<T>, dim3(nq, ty), dim3(256), 0, (hipStream_t)stream, cand, cand_off, n_valid, nq,
                       n_max, nms_thd, max_before, max_after, out_rows, out_n, out_idx);
    CONE_LAUNCH_CHECK();
    return 0;
}

extern "C" int cone_fuse_nms(const float* cand, const int64_t* cand_off, const int32_t* n_valid, int nq, int n_max,
                             double nms_thd, int max_before, int max_after, double* out_rows, int32_t* out_n,
                             int32_t* out_idx, void* stream) {
    return fuse_nms_launch(cand, cand_off, n_valid, nq, n_max, nms_thd, max_before, max_after, out_rows, out_n, out_idx,
                           stream);
}

extern "C" int cone_fuse_nms_f64(const double* cand, const int64_t* cand_off, const int32_t* n_valid, int nq, int n_max,
                                 double nms_thd, int max_before, int max_after, double* out_rows, int32_t* out_n,
                                 int32_t* out_idx, void* stream) {
    return fuse_nms_launch(cand, cand_off, n_valid, nq, n_max, nms_thd, max_before, max_after, out_rows, out_n, out_idx,
                           stream);
}

extern "C" int cone_temporal_nms(const double* pred, int n, double nms_thd, int max_after, int32_t* keep_idx,
                                 int32_t* keep_n, void* stream) {
    CONE_REQUIRE(n >= 1 && n <= cone::kMaxCand, "temporal_nms: n=%d not in [1,%d]", n, cone::kMaxCand);
    hipLaunchKernelGGL(cone::temporal_nms_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, n, nms_thd,
                       max_after, keep_idx, keep_n);
    CONE_LAUNCH_CHECK();
    return 0;
}

extern "C" int cone_matcher_cost(const float* logits, const float* spans, const float* tgt, int B, int Nq,
                                 float cost_span, float cost_giou, float cost_class, float* cost,
                                 int32_t* best, void* stream) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(cone::matcher_cost_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       logits, spans, tgt, B, Nq, cost_span, cost_giou, cost_class, cost, best);
    CONE_LAUNCH_CHECK();
    return 0;
}
