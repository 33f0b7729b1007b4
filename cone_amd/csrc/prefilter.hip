// Stage A: sliding-window pre-filter (cone/inference.py:276-299).
//
//   frame_score_kernel : frame_scores[q][f] = <vid[f], txt[q]>  -- the HBM-bound stream over the
//                        pre-extracted clip features (12.7 GB for the MAD-scale stress video).
//                        One wavefront streams whole rows with coalesced 16-B lane loads (1 KiB per
//                        wave-instruction), RPW rows in flight per wave, QG query vectors held in
//                        registers; per-(row,query) partial sums are combined with wave shuffles.
//   window_max_kernel  : win[q][i] = max(frame_scores[q][max((i-1)S,0) : min((i-1)S+W, ctx_l)])
//   topk_kernel        : first k entries of the stable descending sort of each score row.
#include <mutex>

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

// Long score rows (MAD scale: 100 k windows x up to 64 queries): one workgroup per 64 consecutive windows of one query
// stages the (64 + 1) S + 1 frame scores they cover in LDS with coalesced loads (every frame score is read once from
// memory instead of once per window it belongs to) and each wave takes its windows' maxima from there.
constexpr int WM_WPB = 64;
__global__ __launch_bounds__(256) void window_max_tiled_kernel(const float* __restrict__ fs, int64_t ctx_l, int W, int S,
                                                               int64_t num_window, float* __restrict__ win) {
    extern __shared__ float wm_s[];
    const int q = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t w0 = (int64_t)blockIdx.x * WM_WPB;
    const int64_t f_lo = max((w0 - 1) * S, (int64_t)0);
    const int64_t w_hi = min(w0 + WM_WPB, num_window);                  // exclusive
    const int64_t f_hi = min((w_hi - 2) * S + W, ctx_l);                // end of the last window's range
    const float* f = fs + (size_t)q * ctx_l;
    const int n = (int)(f_hi - f_lo);
    for (int i = tid; i < n; i += 256) wm_s[i] = f[f_lo + i];
    __syncthreads();
    for (int64_t i = w0 + wave; i < w_hi; i += 4) {
        const int s = (int)(max((i - 1) * S, (int64_t)0) - f_lo);
        const int e = (int)(min((i - 1) * S + W, ctx_l) - f_lo);
        float m = -INFINITY;
        for (int t = s + lane; t < e; t += 64) m = fmaxf(m, wm_s[t]);
        m = wave_max(m);
        if (lane == 0) win[(size_t)q * num_window + i] = m;
    }
}

// Two-level stable top-k for long rows: level 1 -- one workgroup per chunk of TK_CH scores keeps the chunk in LDS and
// extracts its own stable top-k (k short passes over LDS instead of k passes over the whole row in memory); level 2 --
// topk_merge_kernel picks the k best of the (chunks x k) candidates with the same (score desc, index asc) order.  Chunks
// cover ascending index ranges and every list is (score desc, index asc), so the merged list is the row's stable
// descending order (the argument of parallel.merge_topk).
constexpr int TK_CH = 4096;
__global__ __launch_bounds__(256) void topk_chunk_kernel(const float* __restrict__ sc, int64_t n, int k,
                                                         float* __restrict__ cval, int* __restrict__ cidx, int n_chunks) {
    __shared__ float v_s[TK_CH];
    __shared__ float s_v[4];
    __shared__ int s_i[4];
    __shared__ float best_v;
    __shared__ int best_i;
    const int q = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)ch * TK_CH;
    const int m = (int)min((int64_t)TK_CH, n - base);
    const float* row = sc + (size_t)q * n + base;
    for (int i = tid; i < m; i += 256) v_s[i] = row[i];
    __syncthreads();
    float last_v = INFINITY;
    int last_i = -1;
    float* ov = cval + ((size_t)q * n_chunks + ch) * k;
    int* oi = cidx + ((size_t)q * n_chunks + ch) * k;
    for (int p = 0; p < k; ++p) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = tid; j < m; j += 256) {
            const float v = v_s[j];
            const bool after = (v < last_v) || (v == last_v && j > last_i);
            if (after && (v > bv || (v == bv && j < bi))) { bv = v; bi = j; }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov2 = __shfl_xor(bv, o, 64);
            const int oi2 = __shfl_xor(bi, o, 64);
            if (ov2 > bv || (ov2 == bv && oi2 < bi)) { bv = ov2; bi = oi2; }
        }
        if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float v = s_v[0];
            int i = s_i[0];
            for (int w = 1; w < 4; ++w)
                if (s_v[w] > v || (s_v[w] == v && s_i[w] < i)) { v = s_v[w]; i = s_i[w]; }
            best_v = v; best_i = i;
            ov[p] = i == 0x7fffffff ? -INFINITY : v;
            oi[p] = i == 0x7fffffff ? 0x7fffffff : (int)(base + i);
        }
        __syncthreads();
        last_v = best_v;
        last_i = best_i;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ cval, const int* __restrict__ cidx,
                                                         int n_cand, int k, int32_t* __restrict__ idx,
                                                         float* __restrict__ val) {
    __shared__ float s_v[4];
    __shared__ int s_i[4];
    __shared__ float best_v;
    __shared__ int best_i;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* cv = cval + (size_t)q * n_cand;
    const int* ci = cidx + (size_t)q * n_cand;
    float last_v = INFINITY;
    int last_i = -1;
    for (int p = 0; p < k; ++p) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = tid; j < n_cand; j += 256) {
            const float v = cv[j];
            const int gi = ci[j];
            const bool after = (v < last_v) || (v == last_v && gi > last_i);
            if (gi != 0x7fffffff && after && (v > bv || (v == bv && gi < bi))) { bv = v; bi = gi; }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov2 = __shfl_xor(bv, o, 64);
            const int oi2 = __shfl_xor(bi, o, 64);
            if (ov2 > bv || (ov2 == bv && oi2 < bi)) { bv = ov2; bi = oi2; }
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
            if (val) val[(size_t)q * k + p] = v;
        }
        __syncthreads();
        last_v = best_v;
        last_i = best_i;
        __syncthreads();
    }
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

// ---- many queries over one video: frame scores on the fp32 matrix cores --------------------------------------------
// fs[q][f] = <vid[f], txt[q]> for up to 64 queries at once is a skinny GEMM whose big operand is the clip arena itself
// (read ONCE for all queries): arithmetic intensity 0.5 * Q flop/B -- at Q = 64 the MFMA time (2.7 ms for the 12.7 GB
// MAD-scale video) and the HBM time (2.3 ms) are about equal.  A general GEMM tile wastes half of its 128 rows on 64
// queries and re-stages the arena through LDS; here
//   * D[query][frame] tiles of v_mfma_f32_16x16x4_f32: A = the query vectors from LDS (operand slabs [16 queries][16
//     channels], 16-B chunks XOR-swizzled: conflict-free ds_read_b128; 128 KiB for 64 x 512), B = the frames straight
//     from global memory into registers (lane = frame, float4 = channels 16 s + 4 lg .. : k slot lg of step (s, r) <->
//     channel 16 s + 4 lg + r on both operands), 128 channels (8 float4) at a time, the next 128 in flight meanwhile;
//   * a wave owns 16 frames per step of a grid-stride loop; accumulator register r of lane (li, lg) = fs[query 4 lg + r
//     of the tile][frame li]: 64-B row segments per store.
// Exact fp32 products and sums (another summation order than the streaming kernel: ~1e-7 relative).
typedef float pf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int pf_swz16(int row) { return (0x1230 >> (((row >> 2) & 3) * 4)) & 3; }

constexpr int MQ_NT = 768;     // 12 waves: the one workgroup a CU holds (128 KiB of LDS) runs three waves per SIMD

template <int QT /* query tiles of 16 */>
__global__ __launch_bounds__(MQ_NT, 3) void frame_score_mq_kernel(const float* __restrict__ vid, int64_t ctx_l, int dv,
                                                                const float* __restrict__ txt, int q0, int nq,
                                                                float* __restrict__ fs) {
    extern __shared__ __attribute__((aligned(16))) float qs[];          // [QT][dv / 16] slabs of [16 queries][16 floats]
    const int tid = threadIdx.x, lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    const int ns = dv >> 4;                                              // slabs per query tile
    for (int i = tid; i < QT * 16 * (dv >> 2); i += MQ_NT) {             // (query, float4 of its vector)
        const int qi = i / (dv >> 2), c4 = i % (dv >> 2);
        const int qg = q0 + qi;
        pf4 v = pf4{0.f, 0.f, 0.f, 0.f};
        if (qg < nq) v = *reinterpret_cast<const pf4*>(txt + (size_t)qg * dv + c4 * 4);
        const int row = qi & 15, s = c4 >> 2, ch = c4 & 3;
        *reinterpret_cast<pf4*>(qs + ((qi >> 4) * ns + s) * 256 + row * 16 + ((ch ^ pf_swz16(row)) << 2)) = v;
    }
    __syncthreads();
    const int rd = li * 16 + ((lg ^ pf_swz16(li)) << 2);
    const int64_t n_tiles = (ctx_l + 15) >> 4;
    const int64_t wave_id = (int64_t)blockIdx.x * (MQ_NT / 64) + (tid >> 6), n_waves = (int64_t)gridDim.x * (MQ_NT / 64);
    const int nchunk = dv >> 7;                                          // 128-channel chunks (8 slabs)
    for (int64_t t = wave_id; t < n_tiles; t += n_waves) {
        const int64_t f0 = t * 16;
        const int64_t frame = min(f0 + li, ctx_l - 1);
        const float* fp = vid + frame * dv + 4 * lg;
        pf4 acc[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) acc[qt] = pf4{0.f, 0.f, 0.f, 0.f};
        pf4 cur[8], nxt[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) cur[s] = *reinterpret_cast<const pf4*>(fp + 16 * s);
        for (int c = 0; c < nchunk; ++c) {
            if (c + 1 < nchunk) {
#pragma unroll
                for (int s = 0; s < 8; ++s) nxt[s] = *reinterpret_cast<const pf4*>(fp + 128 * (c + 1) + 16 * s);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const pf4 a = *reinterpret_cast<const pf4*>(qs + (qt * ns + c * 8 + s) * 256 + rd);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], cur[s][r], acc[qt], 0, 0, 0);
                }
            }
            if (c + 1 < nchunk) {
#pragma unroll
                for (int s = 0; s < 8; ++s) cur[s] = nxt[s];
            }
        }
        if (f0 + li < ctx_l) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qg = q0 + qt * 16 + 4 * lg + r;
                    if (qg < nq) fs[(size_t)qg * ctx_l + f0 + li] = acc[qt][r];
                }
        }
    }
}

static int launch_frame_scores_mq(const float* vid, int64_t ctx_l, int dv, const float* txt, int nq, float* fs,
                                  hipStream_t s) {
    // 64 queries per launch while their vectors fit the LDS next to nothing else (64 x 512 x 4 B = 128 KiB), else 32
    const int qpl = dv <= 512 ? 64 : 32;
    static std::once_flag once;
    static hipError_t attr_rc = hipSuccess;
    static int n_cu = 0;
    std::call_once(once, [] {
        attr_rc = hipFuncSetAttribute((const void*)frame_score_mq_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      128 * 1024);
        if (attr_rc == hipSuccess)
            attr_rc = hipFuncSetAttribute((const void*)frame_score_mq_kernel<2>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        int dev = 0;
        if (attr_rc == hipSuccess) attr_rc = hipGetDevice(&dev);
        if (attr_rc == hipSuccess) attr_rc = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    });
    CONE_CHECK_HIP(attr_rc);
    int64_t blocks = ((ctx_l + 15) / 16 + MQ_NT / 64 - 1) / (MQ_NT / 64);
    if (blocks > n_cu) blocks = n_cu;                                  // one workgroup per CU, grid-stride over 16-frame tiles
    for (int q0 = 0; q0 < nq;) {
        const int rem = nq - q0;
        const bool wide = qpl == 64 && rem > 32;                           // 4 query tiles, else 2
        ProfScope ps(PK_FRAME_SCORE, ctx_l, dv, rem < (wide ? 64 : 32) ? rem : (wide ? 64 : 32), nullptr, s);
        if (wide)
            hipLaunchKernelGGL(frame_score_mq_kernel<4>, dim3((unsigned)blocks), dim3(MQ_NT), (size_t)64 * dv * 4, s, vid,
                               ctx_l, dv, txt, q0, nq, fs);
        else
            hipLaunchKernelGGL(frame_score_mq_kernel<2>, dim3((unsigned)blocks), dim3(MQ_NT), (size_t)32 * dv * 4, s, vid,
                               ctx_l, dv, txt, q0, nq, fs);
        CONE_LAUNCH_CHECK();
        q0 += wide ? 64 : 32;
    }
    return 0;
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

int launch_window_max(const float* fs, int64_t ctx_l, int W, int S, int nq, float* win, hipStream_t s) {
    const int64_t nw = (ctx_l + S - 1) / S + 1;
    const size_t lds = (size_t)((WM_WPB + 1) * S + W + 1) * sizeof(float);
    if (nw >= 4 * WM_WPB && lds <= 48 * 1024) {
        hipLaunchKernelGGL(window_max_tiled_kernel, dim3((unsigned)((nw + WM_WPB - 1) / WM_WPB), nq), dim3(256), lds, s,
                           fs, ctx_l, W, S, nw, win);
    } else {
        hipLaunchKernelGGL(window_max_kernel, dim3((unsigned)((nw + 3) / 4), nq), dim3(256), 0, s, fs, ctx_l, W, S, nw, win);
    }
    CONE_LAUNCH_CHECK();
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
    if (nq >= 8) {
        // Many queries over one video: the clip arena is read once for up to 64 queries by the fp32-MFMA kernel
        // (BASELINE configs 3 / 5) instead of nq / 4 VALU passes over the features.
        rc = cone::launch_frame_scores_mq(vid, ctx_l, dv, txt, nq, frame_scores, s);
        if (rc) return rc;
        return cone::launch_window_max(frame_scores, ctx_l, W, S, nq, win_scores, s);
    }
    switch (dv / 256) {
        case 1: rc = cone::launch_frame_scores<1>(vid, ctx_l, txt, nq, frame_scores, s); break;
        case 2: rc = cone::launch_frame_scores<2>(vid, ctx_l, txt, nq, frame_scores, s); break;
        case 3: rc = cone::launch_frame_scores<3>(vid, ctx_l, txt, nq, frame_scores, s); break;
        default: rc = cone::launch_frame_scores<4>(vid, ctx_l, txt, nq, frame_scores, s); break;
    }
    if (rc) return rc;
    return cone::launch_window_max(frame_scores, ctx_l, W, S, nq, win_scores, s);
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

extern "C" size_t cone_topk_windows_workspace(int nq, int64_t num_window, int k) {
    if (num_window <= 2 * cone::TK_CH) return 0;
    const size_t n_chunks = (size_t)((num_window + cone::TK_CH - 1) / cone::TK_CH);
    return 2 * cone::align_up((size_t)nq * n_chunks * k * 4, 256);
}

extern "C" int cone_topk_windows_ws(const float* win_scores, int nq, int64_t num_window, int k, int32_t* idx, float* val,
                                    void* ws, size_t ws_bytes, void* stream) {
    CONE_REQUIRE(nq > 0 && num_window > 0 && k > 0 && k <= num_window && num_window < 0x7fffffff,
                 "topk: bad sizes nq=%d num_window=%lld k=%d", nq, (long long)num_window, k);
    const size_t need = cone_topk_windows_workspace(nq, num_window, k);
    if (need == 0 || k > cone::TK_CH / 4) return cone_topk_windows(win_scores, nq, num_window, k, idx, val, stream);
    CONE_REQUIRE(ws && ws_bytes >= need, "topk: workspace too small (%zu < %zu)", ws_bytes, need);
    const int n_chunks = (int)((num_window + cone::TK_CH - 1) / cone::TK_CH);
    float* cval = (float*)ws;
    int* cidx = (int*)((char*)ws + need / 2);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cone::topk_chunk_kernel, dim3(n_chunks, nq), dim3(256), 0, s, win_scores, num_window, k, cval, cidx,
                       n_chunks);
    CONE_LAUNCH_CHECK();
    hipLaunchKernelGGL(cone::topk_merge_kernel, dim3(nq), dim3(256), 0, s, cval, cidx, n_chunks * k, k, idx, val);
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
