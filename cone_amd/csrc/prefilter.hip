// Stage A: sliding-window pre-filter (cone/inference.py:276-299).
//
//   frame_score_kernel : frame_scores[q][f] = <vid[f], txt[q]>  -- the HBM-bound stream over the
//                        pre-extracted clip features (12.7 GB for the MAD-scale stress video).
//                        One wavefront streams whole rows with coalesced 16-B lane loads (1 KiB per
//                        wave-instruction), RPW rows in flight per wave, QG query vectors held in
//                        registers; per-(row,query) partial sums are combined with wave shuffles.
//   window_max_kernel  : win[q][i] = max(frame_scores[q][max((i-1)S,0) : min((i-1)S+W, ctx_l)])
//   topk_kernel        : first k entries of the stable descending sort of each score row.
#include "common.h"

namespace cone {

template <int VPL /* float4 per lane per row: dv = 256*VPL */, int QG, int RPW>
__global__ __launch_bounds__(256) void frame_score_kernel(const float* __restrict__ vid, int64_t ctx_l,
                                                          const float* __restrict__ txt, int q0, int nq,
                                                          float* __restrict__ fs) {
    constexpr int DV = 256 * VPL;
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    float4 q[QG][VPL];
#pragma unroll
    for (int g = 0; g < QG; ++g)
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int qi = min(q0 + g, nq - 1);
            q[g][v] = reinterpret_cast<const float4*>(txt + (size_t)qi * DV)[lane + 64 * v];
        }
    for (int64_t r0 = wave_id * RPW; r0 < ctx_l; r0 += n_waves * RPW) {
        float4 x[RPW][VPL];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int64_t row = min(r0 + r, ctx_l - 1);
#pragma unroll
            for (int v = 0; v < VPL; ++v)
                x[r][v] = reinterpret_cast<const float4*>(vid + row * DV)[lane + 64 * v];
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < VPL; ++v)
                    s += (x[r][v].x * q[g][v].x + x[r][v].y * q[g][v].y) +
                         (x[r][v].z * q[g][v].z + x[r][v].w * q[g][v].w);
                s = wave_sum(s);
                if (lane == 0 && r0 + r < ctx_l && q0 + g < nq) fs[(size_t)(q0 + g) * ctx_l + r0 + r] = s;
            }
    }
}

// One wave per window: coalesced reads of its <= W frame scores, wave max (exact, order-free).
__global__ __launch_bounds__(256) void window_max_kernel(const float* __restrict__ fs, int64_t ctx_l, int W,
                                                         int S, int64_t num_window, float* __restrict__ win) {
    const int q = blockIdx.y, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= num_window) return;
    const int64_t s = max((i - 1) * S, (int64_t)0);
    const int64_t e = min((i - 1) * S + W, ctx_l);
    const float* f = fs + (size_t)q * ctx_l;
    float m = -INFINITY;
    for (int64_t t = s + lane; t < e; t += 64) m = fmaxf(m, f[t]);
    m = wave_max(m);
    if (lane == 0) win[(size_t)q * num_window + i] = m;
}

// Stable descending top-k: pass p picks the largest (score, then lowest index) strictly after the
// previous pick in that order.  One workgroup per score row.
template <int NT>
__global__ __launch_bounds__(NT) void topk_kernel(const float* __restrict__ sc, int64_t n, int k,
                                                  int32_t* __restrict__ idx, float* __restrict__ val) {
    constexpr int NW = NT / 64;
    __shared__ float s_v[NW];
    __shared__ int s_i[NW];
    __shared__ float best_v;
    __shared__ int best_i;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = sc + (size_t)q * n;
    float last_v = INFINITY;
    int last_i = -1;
    for (int p = 0; p < k; ++p) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        // 4 independent loads in flight per thread: a pass over a 100k-window row is latency-, not bandwidth-bound
        for (int64_t j0 = tid; j0 < n; j0 += 4 * NT) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t j = j0 + (int64_t)u * NT;
                v[u] = j < n ? row[j] : -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t j = j0 + (int64_t)u * NT;
                const bool after = (v[u] < last_v) || (v[u] == last_v && (int)j > last_i);
                if (j < n && after && (v[u] > bv || (v[u] == bv && (int)j < bi))) { bv = v[u]; bi = (int)j; }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float v = s_v[0];
            int i = s_i[0];
            for (int w = 1; w < NW; ++w)
                if (s_v[w] > v || (s_v[w] == v && s_i[w] < i)) { v = s_v[w]; i = s_i[w]; }
            best_v = v; best_i = i;
            idx[(size_t)q * k + p] = i == 0x7fffffff ? -1 : i;
            if (val) val[(size_t)q * k + p] = v;
        }
        __syncthreads();
        last_v = best_v;
        last_i = best_i;
        __syncthreads();
    }
}

// ---- segmented forms: all queries of a split in three launches --------------------------------
// A group = one video and up to 4 of its queries (the clip rows are read once per group).
template <int VPL>
__global__ __launch_bounds__(256) void frame_score_groups_kernel(const float* __restrict__ arena,
                                                                 const float* __restrict__ cls,
                                                                 const int64_t* __restrict__ g_row0,
                                                                 const int* __restrict__ g_ctx_l,
                                                                 const int* __restrict__ g_q,
                                                                 const int64_t* __restrict__ q_fs_off,
                                                                 float* __restrict__ fs) {
    constexpr int DV = 256 * VPL, RPW = 4;
    const int g = blockIdx.y;
    const int ctx_l = g_ctx_l[g];
    const int lane = threadIdx.x & 63;
    const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * 4;
    if (wave_id * RPW >= ctx_l) return;
    const float* vid = arena + g_row0[g] * DV;
    int qi[4];
    float4 q[4][VPL];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        qi[j] = g_q[g * 4 + j];
        const int src = qi[j] >= 0 ? qi[j] : g_q[g * 4];
#pragma unroll
        for (int v = 0; v < VPL; ++v) q[j][v] = reinterpret_cast<const float4*>(cls + (size_t)src * DV)[lane + 64 * v];
    }
    for (int r0 = wave_id * RPW; r0 < ctx_l; r0 += n_waves * RPW) {
        float4 x[RPW][VPL];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = min(r0 + r, ctx_l - 1);
#pragma unroll
            for (int v = 0; v < VPL; ++v) x[r][v] = reinterpret_cast<const float4*>(vid + (size_t)row * DV)[lane + 64 * v];
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < VPL; ++v)
                    s += (x[r][v].x * q[j][v].x + x[r][v].y * q[j][v].y) + (x[r][v].z * q[j][v].z + x[r][v].w * q[j][v].w);
                s = wave_sum(s);
                if (lane == 0 && r0 + r < ctx_l && qi[j] >= 0) fs[q_fs_off[qi[j]] + r0 + r] = s;
            }
    }
}

__global__ __launch_bounds__(256) void window_max_seg_kernel(const float* __restrict__ fs,
                                                             const int64_t* __restrict__ q_fs_off,
                                                             const int64_t* __restrict__ q_win_off,
                                                             const int* __restrict__ q_ctx_l, int W, int S,
                                                             float* __restrict__ win) {
    const int q = blockIdx.y;
    const int ctx_l = q_ctx_l[q];
    const int nw = (ctx_l + S - 1) / S + 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nw) return;
    const int s = max((i - 1) * S, 0), e = min((i - 1) * S + W, ctx_l);
    const float* f = fs + q_fs_off[q];
    float m = -INFINITY;
    for (int t = s; t < e; ++t) m = fmaxf(m, f[t]);
    win[q_win_off[q] + i] = m;
}

__global__ __launch_bounds__(256) void topk_seg_kernel(const float* __restrict__ win,
                                                       const int64_t* __restrict__ q_win_off,
                                                       const int* __restrict__ q_ctx_l, int S, int k,
                                                       int32_t* __restrict__ idx) {
    __shared__ float s_v[4];
    __shared__ int s_i[4];
    __shared__ float best_v;
    __shared__ int best_i;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = (q_ctx_l[q] + S - 1) / S + 1;
    const float* row = win + q_win_off[q];
    float last_v = INFINITY;
    int last_i = -1;
    for (int p = 0; p < k; ++p) {
        if (p >= n) {   // fewer windows than k: pad with -1 (uniform branch)
            if (tid == 0) idx[(size_t)q * k + p] = -1;
            continue;
        }
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = tid; j < n; j += 256) {
            const float v = row[j];
            const bool after = (v < last_v) || (v == last_v && j > last_i);
            if (after && (v > bv || (v == bv && j < bi))) { bv = v; bi = j; }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float v = s_v[0];
            int i = s_i[0];
            for (int w = 1; w < 4; ++w)
                if (s_v[w] > v || (s_v[w] == v && s_i[w] < i)) { v = s_v[w]; i = s_i[w]; }
            best_v = v; best_i = i;
            idx[(size_t)q * k + p] = i == 0x7fffffff ? -1 : i;
        }
        __syncthreads();
        last_v = best_v;
        last_i = best_i;
        __syncthreads();
    }
}

template <int VPL>
static int launch_frame_scores(const float* vid, int64_t ctx_l, const float* txt, int nq, float* fs,
                               hipStream_t s) {
    constexpr int RPW = 4;
    int64_t blocks = (ctx_l + 4 * RPW - 1) / (4 * RPW);
    if (blocks > 256 * 8) blocks = 256 * 8;  // grid-stride: 8 workgroups per CU
    for (int q0 = 0; q0 < nq;) {
        const int rem = nq - q0;
        ProfScope ps(PK_FRAME_SCORE, ctx_l, 256 * VPL, rem >= 4 ? 4 : (rem >= 2 ? 2 : 1), nullptr, s);
        if (rem >= 4) {
            hipLaunchKernelGGL((frame_score_kernel<VPL, 4, RPW>), dim3((unsigned)blocks), dim3(256), 0, s, vid,
                               ctx_l, txt, q0, nq, fs);
            q0 += 4;
        } else if (rem >= 2) {
            hipLaunchKernelGGL((frame_score_kernel<VPL, 2, RPW>), dim3((unsigned)blocks), dim3(256), 0, s, vid,
                               ctx_l, txt, q0, nq, fs);
            q0 += 2;
        } else {
            hipLaunchKernelGGL((frame_score_kernel<VPL, 1, RPW>), dim3((unsigned)blocks), dim3(256), 0, s, vid,
                               ctx_l, txt, q0, nq, fs);
            q0 += 1;
        }
        CONE_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace cone

extern "C" int64_t cone_num_windows(int64_t ctx_l, int W) {
    const int S = W / 2;
    if (S <= 0 || ctx_l <= 0) return 0;
    return (ctx_l + S - 1) / S + 1;
}

extern "C" int cone_prefilter_scores(const float* vid, int64_t ctx_l, int dv, const float* txt, int nq, int W,
                                     int S, float* frame_scores, float* win_scores, void* stream) {
    CONE_REQUIRE(ctx_l > 0 && nq > 0 && W > 0 && S > 0, "prefilter: bad sizes ctx_l=%lld nq=%d W=%d S=%d",
                 (long long)ctx_l, nq, W, S);
    CONE_REQUIRE(dv == 256 || dv == 512 || dv == 768 || dv == 1024,
                 "prefilter: feature dim %d not in {256,512,768,1024}", dv);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (nq >= 8 && ctx_l < (1ll << 31)) {
        // Many queries over one video: frame_scores (nq, ctx_l) = txt . vid^T is a GEMM whose "weight"
        // operand is the clip arena itself ([N = ctx_l][K = dv], read once for all queries) -- the
        // fp32-MFMA tile instead of nq/4 VALU passes over the features (BASELINE config 5).
        cone::GemmArgs g{};
        g.A = txt; g.lda = dv; g.W = vid; g.ldw = dv; g.C = frame_scores; g.ldc = (int)ctx_l;
        g.M = nq; g.N = (int)ctx_l; g.K = dv;
        rc = cone::launch_gemm(g, s);
        if (rc) return rc;
        const int64_t nwg = (ctx_l + S - 1) / S + 1;
        hipLaunchKernelGGL(cone::window_max_kernel, dim3((unsigned)((nwg + 3) / 4), nq), dim3(256), 0, s,
                           frame_scores, ctx_l, W, S, nwg, win_scores);
        CONE_LAUNCH_CHECK();
        return 0;
    }
    switch (dv / 256) {
        case 1: rc = cone::launch_frame_scores<1>(vid, ctx_l, txt, nq, frame_scores, s); break;
        case 2: rc = cone::launch_frame_scores<2>(vid, ctx_l, txt, nq, frame_scores, s); break;
        case 3: rc = cone::launch_frame_scores<3>(vid, ctx_l, txt, nq, frame_scores, s); break;
        default: rc = cone::launch_frame_scores<4>(vid, ctx_l, txt, nq, frame_scores, s); break;
    }
    if (rc) return rc;
    const int64_t nw = (ctx_l + S - 1) / S + 1;
    hipLaunchKernelGGL(cone::window_max_kernel, dim3((unsigned)((nw + 3) / 4), nq), dim3(256), 0, s,
                       frame_scores, ctx_l, W, S, nw, win_scores);
    CONE_LAUNCH_CHECK();
    return 0;
}

extern "C" int cone_prefilter_batched(const float* arena, int dv, const float* cls, const int64_t* g_row0,
                                      const int32_t* g_ctx_l, const int32_t* g_q, int ng, int max_ctx_l,
                                      const int64_t* q_fs_off, const int64_t* q_win_off, const int32_t* q_ctx_l,
                                      int nq, int W, int S, float* frame_scores, float* win_scores, int k,
                                      int32_t* topk_idx, void* stream) {
    CONE_REQUIRE(arena && cls && g_row0 && g_ctx_l && g_q && q_fs_off && q_win_off && q_ctx_l && frame_scores &&
                     win_scores && topk_idx, "prefilter_batched: null argument");
    CONE_REQUIRE(ng > 0 && nq > 0 && max_ctx_l > 0 && W > 0 && S > 0 && k > 0, "prefilter_batched: bad sizes");
    CONE_REQUIRE(dv == 256 || dv == 512 || dv == 768 || dv == 1024,
                 "prefilter: feature dim %d not in {256,512,768,1024}", dv);
    hipStream_t s = (hipStream_t)stream;
    int64_t bx = ((int64_t)max_ctx_l + 15) / 16;
    if (bx > 2048) bx = 2048;
    dim3 grid((unsigned)bx, ng);
    {
        cone::ProfScope ps(cone::PK_FRAME_SCORE, max_ctx_l, dv, ng, nullptr, s);
        switch (dv / 256) {
            case 1: hipLaunchKernelGGL(cone::frame_score_groups_kernel<1>, grid, dim3(256), 0, s, arena, cls, g_row0, g_ctx_l, g_q, q_fs_off, frame_scores); break;
            case 2: hipLaunchKernelGGL(cone::frame_score_groups_kernel<2>, grid, dim3(256), 0, s, arena, cls, g_row0, g_ctx_l, g_q, q_fs_off, frame_scores); break;
            case 3: hipLaunchKernelGGL(cone::frame_score_groups_kernel<3>, grid, dim3(256), 0, s, arena, cls, g_row0, g_ctx_l, g_q, q_fs_off, frame_scores); break;
            default: hipLaunchKernelGGL(cone::frame_score_groups_kernel<4>, grid, dim3(256), 0, s, arena, cls, g_row0, g_ctx_l, g_q, q_fs_off, frame_scores); break;
        }
    }
    CONE_LAUNCH_CHECK();
    const int max_nw = (max_ctx_l + S - 1) / S + 1;
    hipLaunchKernelGGL(cone::window_max_seg_kernel, dim3((max_nw + 255) / 256, nq), dim3(256), 0, s, frame_scores,
                       q_fs_off, q_win_off, q_ctx_l, W, S, win_scores);
    CONE_LAUNCH_CHECK();
    hipLaunchKernelGGL(cone::topk_seg_kernel, dim3(nq), dim3(256), 0, s, win_scores, q_win_off, q_ctx_l, S, k,
                       topk_idx);
    CONE_LAUNCH_CHECK();
    return 0;
}

extern "C" int cone_topk_windows(const float* win_scores, int nq, int64_t num_window, int k, int32_t* idx,
                                 float* val, void* stream) {
    CONE_REQUIRE(nq > 0 && num_window > 0 && k > 0 && k <= num_window && num_window < 0x7fffffff,
                 "topk: bad sizes nq=%d num_window=%lld k=%d", nq, (long long)num_window, k);
    if (num_window > 4096)      // long video (MAD scale): 16 waves per row
        hipLaunchKernelGGL(cone::topk_kernel<1024>, dim3(nq), dim3(1024), 0, (hipStream_t)stream, win_scores,
                           num_window, k, idx, val);
    else
        hipLaunchKernelGGL(cone::topk_kernel<256>, dim3(nq), dim3(256), 0, (hipStream_t)stream, win_scores,
                           num_window, k, idx, val);
    CONE_LAUNCH_CHECK();
    return 0;
}
