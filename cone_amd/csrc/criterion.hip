// SetCriterion.forward of the reference (cone/model.py:213-425) with its HungarianMatcher (cone/matcher.py:37-106) as
// device kernels -- forward values only (no autograd; the reference's training loop is out of scope, this is the
// evaluation-side use: loss meters next to the predictions).
//
//   criterion_window_kernel : one thread per window.  Cost matrix C[n][j] = cost_span * L1(span_n, tgt_j)
//       - cost_giou * GIoU(span_n, tgt_j) - cost_class * softmax(logits_n)[foreground]  (cone/matcher.py:61-95), then
//       the exact minimum-cost assignment of min(Nq, T) pairs (scipy.optimize.linear_sum_assignment in the reference)
//       by dynamic programming over subsets (Nq, T <= 8: at most 256 states), then this window's share of every loss:
//       matched L1 / GIoU sums (loss_spans :266-293), weighted cross-entropy over all slots incl. the negative
//       window's (loss_labels :295-327), top-1 hits of the matched slots (class_error), saliency hinge sums
//       (loss_saliency :329-363).
//   criterion_reduce_kernel : sums the per-window shares in window order (deterministic) and applies the means.
#include "common.h"

namespace cone {

constexpr int CRIT_MAXN = 8;        // slots / targets per window
constexpr int CRIT_PART = 8;        // floats per window: l1, giou, n_matched, ce, n_correct, sal, neg_sal, (pad)

struct CritArgs {
    const float* logits; const float* spans;          // (B, Nq, 2)
    const float* tgt; const int* tgt_off;             // (sum T, 2) (center, width); (B + 1)
    const float* neg_logits;                          // (B, Nq, 2) or null
    const float* sal; int L;                          // (B, L) or null
    const int* pos_idx; const int* neg_idx; int P;    // (B, P)
    const float* neg_sal; int L2;                     // (B, L2) or null
    int B, Nq;
    float cost_span, cost_giou, cost_class, eos_coef, margin;
    int* assign;                                      // (B, Nq): matched target (index inside the window) or -1
    float* part;                                      // (B, CRIT_PART)
};

__device__ __forceinline__ float crit_giou(float c, float w, float tc, float tw) {     // cone/span_utils.py:25-41,62-122
    const float x1 = __fsub_rn(c, __fmul_rn(0.5f, w)), x2 = __fadd_rn(c, __fmul_rn(0.5f, w));
    const float t1 = __fsub_rn(tc, __fmul_rn(0.5f, tw)), t2 = __fadd_rn(tc, __fmul_rn(0.5f, tw));
    const float inter = fmaxf(__fsub_rn(fminf(x2, t2), fmaxf(x1, t1)), 0.f);
    const float uni = __fsub_rn(__fadd_rn(__fsub_rn(x2, x1), __fsub_rn(t2, t1)), inter);
    const float iou = __fdiv_rn(inter, uni);
    const float enc = fmaxf(__fsub_rn(fmaxf(x2, t2), fminf(x1, t1)), 0.f);
    return __fsub_rn(iou, __fdiv_rn(__fsub_rn(enc, uni), enc));
}

// -w[y] * log_softmax(l)[y] for the two-class head; y = 0 foreground (weight 1), 1 background (weight eos_coef)
__device__ __forceinline__ float crit_ce(float l0, float l1, int y, float eos) {
    const float m = fmaxf(l0, l1);
    const float lse = __fadd_rn(m, logf(__fadd_rn(expf(__fsub_rn(l0, m)), expf(__fsub_rn(l1, m)))));
    const float lp = __fsub_rn(y == 0 ? l0 : l1, lse);
    return -(y == 0 ? 1.0f : eos) * lp;
}

__global__ __launch_bounds__(64) void criterion_window_kernel(CritArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const int Nq = a.Nq;
    const int t0 = a.tgt ? a.tgt_off[b] : 0;
    const int T = a.tgt ? a.tgt_off[b + 1] - t0 : 0;
    float C[CRIT_MAXN][CRIT_MAXN];
    for (int n = 0; n < Nq; ++n) {
        const int i = b * Nq + n;
        const float l0 = a.logits[2 * i], l1 = a.logits[2 * i + 1];
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        const float prob = e0 / (e0 + e1);
        if (T == 0) continue;                                   // no targets: spans may be NULL (forward(outputs, None))
        const float c = a.spans[2 * i], w = a.spans[2 * i + 1];
        for (int j = 0; j < T; ++j) {
            const float tc = a.tgt[2 * (t0 + j)], tw = a.tgt[2 * (t0 + j) + 1];
            const float l1d = __fadd_rn(fabsf(__fsub_rn(c, tc)), fabsf(__fsub_rn(w, tw)));      // cdist p=1
            const float g = crit_giou(c, w, tc, tw);
            C[n][j] = __fadd_rn(__fadd_rn(__fmul_rn(a.cost_span, l1d), __fmul_rn(a.cost_giou, -g)),
                                __fmul_rn(a.cost_class, -prob));
        }
    }
    // ---- exact assignment: the smaller side is walked in order, the DP state is the set of used items of the other
    int match[CRIT_MAXN];                       // slot n -> target j or -1
    for (int n = 0; n < Nq; ++n) match[n] = -1;
    if (T > 0) {
        const bool rows_small = Nq <= T;        // walk slots (choose distinct targets) or targets (distinct slots)
        const int ns = rows_small ? Nq : T, nl = rows_small ? T : Nq;
        float best[1 << CRIT_MAXN];
        unsigned char from[1 << CRIT_MAXN];
        const int full = 1 << nl;
        best[0] = 0.f;
        for (int mask = 1; mask < full; ++mask) {
            const int k = __popc(mask);         // items of the small side placed so far
            best[mask] = INFINITY;
            from[mask] = 0;
            if (k > ns) continue;
            for (int j = 0; j < nl; ++j)
                if (mask & (1 << j)) {
                    const float prev = best[mask ^ (1 << j)];
                    const float cst = rows_small ? C[k - 1][j] : C[j][k - 1];
                    const float v = __fadd_rn(prev, cst);
                    if (v < best[mask]) { best[mask] = v; from[mask] = (unsigned char)j; }
                }
        }
        int bm = 0;
        float bv = INFINITY;
        for (int mask = 0; mask < full; ++mask)
            if (__popc(mask) == ns && best[mask] < bv) { bv = best[mask]; bm = mask; }
        for (int k = ns; k >= 1; --k) {
            const int j = from[bm];
            if (rows_small) match[k - 1] = j; else match[j] = k - 1;
            bm ^= 1 << j;
        }
    }
    // ---- this window's share of the losses
    float l1s = 0.f, gs = 0.f, ce = 0.f;
    int nm = 0, ncorrect = 0;
    for (int n = 0; n < Nq; ++n) {
        const int i = b * Nq + n;
        const float l0 = a.logits[2 * i], l1 = a.logits[2 * i + 1];
        const int y = match[n] >= 0 ? 0 : 1;
        ce = __fadd_rn(ce, crit_ce(l0, l1, y, a.eos_coef));
        if (match[n] >= 0) {
            const int j = t0 + match[n];
            const float c = a.spans[2 * i], w = a.spans[2 * i + 1];
            const float tc = a.tgt[2 * j], tw = a.tgt[2 * j + 1];
            l1s = __fadd_rn(l1s, __fadd_rn(fabsf(__fsub_rn(c, tc)), fabsf(__fsub_rn(w, tw))));
            gs = __fadd_rn(gs, __fsub_rn(1.0f, crit_giou(c, w, tc, tw)));
            ++nm;
            ncorrect += l0 >= l1;               // top-1 == foreground (topk keeps the lower index on a tie)
        }
        if (a.assign) a.assign[i] = match[n];
    }
    if (a.neg_logits)
        for (int n = 0; n < Nq; ++n) {
            const int i = b * Nq + n;
            ce = __fadd_rn(ce, crit_ce(a.neg_logits[2 * i], a.neg_logits[2 * i + 1], 1, a.eos_coef));
        }
    float sal = 0.f, nsal = 0.f;
    if (a.sal) {
        float nmax = -INFINITY;
        if (a.neg_sal)
            for (int t = 0; t < a.L2; ++t) nmax = fmaxf(nmax, a.neg_sal[(size_t)b * a.L2 + t]);
        for (int k = 0; k < a.P; ++k) {
            const float ps = a.sal[(size_t)b * a.L + a.pos_idx[b * a.P + k]];
            const float ns_ = a.sal[(size_t)b * a.L + a.neg_idx[b * a.P + k]];
            sal = __fadd_rn(sal, fmaxf(__fsub_rn(__fadd_rn(a.margin, ns_), ps), 0.f));
            if (a.neg_sal) nsal = __fadd_rn(nsal, fmaxf(__fsub_rn(__fadd_rn(a.margin, nmax), ps), 0.f));
        }
    }
    float* o = a.part + (size_t)b * CRIT_PART;
    o[0] = l1s; o[1] = gs; o[2] = (float)nm; o[3] = ce; o[4] = (float)ncorrect; o[5] = sal; o[6] = nsal; o[7] = 0.f;
}

// losses[0..4] = loss_span, loss_giou, loss_label, class_error, loss_saliency (window order, one thread: deterministic)
__global__ void criterion_reduce_kernel(const float* __restrict__ part, int B, int Nq, int has_neg, int P, int has_sal,
                                        float* __restrict__ losses) {
    if (blockIdx.x || threadIdx.x) return;
    double s[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < 7; ++k) s[k] += (double)part[(size_t)b * CRIT_PART + k];
    const double n = s[2];
    losses[0] = (float)(s[0] / (2.0 * n));                                     // F.l1_loss(...,'none').mean()
    losses[1] = (float)(s[1] / n);
    losses[2] = (float)(s[3] / ((double)B * Nq * (has_neg ? 2 : 1)));
    losses[3] = (float)(100.0 - 100.0 * s[4] / n);
    losses[4] = has_sal ? (float)(s[5] / ((double)B * P) * 2.0 + (double)s[6] / ((double)B * P) * 2.0) : 0.f;
}

// loss_adapter (cone/model.py:249-264): symmetric cross-entropy of sim / temperature against the diagonal
__global__ void adapter_nce_kernel(const float* __restrict__ sim, int n, float temperature, float* __restrict__ out) {
    if (blockIdx.x || threadIdx.x) return;
    double lr = 0, lc = 0;
    for (int i = 0; i < n; ++i) {
        float mr = -INFINITY, mc = -INFINITY;
        for (int j = 0; j < n; ++j) {
            mr = fmaxf(mr, sim[i * n + j] / temperature);
            mc = fmaxf(mc, sim[j * n + i] / temperature);
        }
        float sr = 0.f, sc = 0.f;
        for (int j = 0; j < n; ++j) {
            sr += expf(sim[i * n + j] / temperature - mr);
            sc += expf(sim[j * n + i] / temperature - mc);
        }
        const float d = sim[i * n + i] / temperature;
        lr += (double)(mr + logf(sr) - d);
        lc += (double)(mc + logf(sc) - d);
    }
    out[0] = (float)((lr / n + lc / n) / 2.0);
}

}  // namespace cone

extern "C" int cone_criterion_forward(const float* logits, const float* spans, const float* tgt, const int32_t* tgt_off,
                                      const float* neg_logits, const float* saliency, int L, const int32_t* pos_idx,
                                      const int32_t* neg_idx, int P, const float* neg_saliency, int L2, int B, int Nq,
                                      float cost_span, float cost_giou, float cost_class, float eos_coef,
                                      float saliency_margin, int32_t* assign, float* part, float* losses,
                                      void* stream) {
    using namespace cone;
    CONE_REQUIRE(logits && part && losses, "criterion: null argument");
    CONE_REQUIRE(B > 0 && Nq >= 1 && Nq <= CRIT_MAXN, "criterion: bad sizes B=%d Nq=%d", B, Nq);
    CONE_REQUIRE(!tgt || (spans && tgt_off), "criterion: targets need spans and offsets");
    CONE_REQUIRE(!saliency || (pos_idx && neg_idx && P >= 1 && L >= 1), "criterion: saliency needs the label pairs");
    hipStream_t s = (hipStream_t)stream;
    CritArgs a{logits, spans, tgt, tgt_off, neg_logits, saliency, L, pos_idx, neg_idx, P, neg_saliency, L2, B, Nq,
               cost_span, cost_giou, cost_class, eos_coef, saliency_margin, assign, part};
    hipLaunchKernelGGL(criterion_window_kernel, dim3((B + 63) / 64), dim3(64), 0, s, a);
    CONE_LAUNCH_CHECK();
    hipLaunchKernelGGL(criterion_reduce_kernel, dim3(1), dim3(64), 0, s, part, B, Nq, neg_logits != nullptr, P,
                       saliency != nullptr, losses);
    CONE_LAUNCH_CHECK();
    return 0;
}

extern "C" int cone_adapter_nce(const float* sim, int n, float temperature, float* loss, void* stream) {
    CONE_REQUIRE(sim && loss && n >= 1 && temperature > 0.f, "adapter_nce: bad argument");
    hipLaunchKernelGGL(cone::adapter_nce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sim, n, temperature, loss);
    CONE_LAUNCH_CHECK();
    return 0;
}
