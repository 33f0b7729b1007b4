// Stage A: sliding-window pre-filter (cone/inference.py:276-299).
//
//   frame_score_kernel : frame_scores[q][f] = <vid[f], txt[q]>  -- the HBM-bound stream over the
//                        pre-extracted clip features (12.7 GB for the MAD-scale stress video).
//                        One wavefront streams whole rows with coalesced 16-B lane loads (1 KiB per
//                        wave-instruction), RPW rows in flight per wave, QG query vectors held in
//                        registers; per-(row,query) partial sums are combined with wave shuffles.
//                        The wave keeps the running max of each half-block (S frames) it streams: the window max is
//                        fused (window_combine_kernel), the frame scores are written only on request.
//   topk_kernel        : first k entries of the stable descending sort of each score row.
#include <mutex>

#include "common.h"

namespace cone {

// The clip rows are read exactly once per launch: the stream's loads are NON-TEMPORAL (global_load_dwordx4 ... nt).  Measured
// on one box (tools/ab_variants.sh prefilter.hip, the 12.7 GB MAD-scale video, 1 query): 2.11 ms = 6.0 TB/s with ordinary loads,
// 1.89 - 1.96 ms = 6.5 - 6.7 TB/s with nt (0.75 -> 0.81 - 0.84 of the 8 TB/s peak): lines that will not be read again no longer
// displace each other through the L2 / Infinity Cache.  CONE_PF_NT = 0 restores ordinary loads for an A/B.  The matrix-core
// kernels for >= 8 queries (CONE_PF_NT_MQ) keep ordinary loads: there a lane-row's 128-B line is fetched by two consecutive
// instructions (64 B each: the MFMA operand layout), and with nt the second half loses its L1 hit -- measured 8 queries
// 2.38 -> 2.45 ms, the three-piece bf16 form 3.42 -> 3.70 ms, 64 queries unchanged.
#ifndef CONE_PF_NT
#define CONE_PF_NT 1
#endif
#ifndef CONE_PF_MQ_MIN
#define CONE_PF_MQ_MIN 5            // queries per video from which the matrix-core kernel takes over (A/B: 8 = round 5)
#endif
#ifndef CONE_PF_NT_MQ
#define CONE_PF_NT_MQ 0
#endif
__device__ __forceinline__ float4 pf_stream_ld(const float4* p) {
#if CONE_PF_NT
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
#else
    return *p;
#endif
}

// Fused frame scores + window max.  S = int(W/2), so window i = half-blocks i-1 and i (block h = frames [hS, (h+1)S)) plus,
// when W is odd, the first frame of block i+1: a wave keeps the running max of the half-block it streams in registers and
// writes ONE value per (query, half-block) -- hm -- and the block's first frame score -- fr; window_combine_kernel takes
// win[i] = max(hm[i-1], hm[i], fr[i+1]) from those (max is order-free: bit-identical to the max over the stored frame
// scores).  The (nq, ctx_l) frame-score matrix is written only when the caller asks for it (fs != nullptr): the
// reference needs it for nothing but this max (cone/inference.py:284-295).
//   WPH = 1: one wave per half-block, grid-stride (long videos: every wave streams contiguous 4-row groups of its blocks);
//   WPH = 4: the four waves of a workgroup share one half-block and combine through LDS (short videos: 4x the waves).
// <x, q> over one float4 of channels, the arithmetic PINNED: each pair sum is fma(first factors, second product), the two
// pair sums are added -- what hipcc's contraction made of (x.x q.x + x.y q.y) + (x.z q.z + x.w q.w) in the streaming kernels
// since round 1.  Left to -ffp-contract=fast the choice is the backend's, per kernel: the per-video kernel and the batched
// one (frame_score_groups_kernel) must produce the same bits (test_prefilter_batched_equals_per_video_path).
__device__ __forceinline__ float pf_dot4(const float4& x, const float4& q) {
#pragma clang fp contract(off)
    const float m1 = x.y * q.y, m3 = x.w * q.w;
    const float p01 = __builtin_fmaf(x.x, q.x, m1);
    const float p23 = __builtin_fmaf(x.z, q.z, m3);
    return p01 + p23;
}

// Sum of N per-lane values over the 64 lanes at once: the totals of wave_sum() -- the same butterfly (lane l adds its
// partner l ^ 32, then l ^ 16, ... l ^ 1: the same pairs, so the same bits) -- but a lane keeps only HALF of its values at
// each of the first log2(N) steps (the partner keeps the other half), so N totals cost N - 1 + (6 - log2 N) shuffles
// instead of 6 N.  The lane's result is the total of value index J(l) = the top log2(N) bits of l (bit 5 = the index's
// top bit); the 64 / N lanes that share those bits all hold it.  (Recursion on the array size: every index is static.)
template <int N, int O = 32>
__device__ __forceinline__ float wave_sum_multi(const float (&v)[N], int lane) {
    static_assert(N >= 1 && N <= 64 && (N & (N - 1)) == 0, "a power of two");
    if constexpr (N == 1) {
        float x = v[0];
#pragma unroll
        for (int o = O; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
        return x;
    } else {
        constexpr int H = N / 2;
        const bool up = (lane & O) != 0;                // this lane keeps the upper half of its values, its partner the lower
        float w[H];
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const float send = up ? v[j] : v[j + H];
            const float keep = up ? v[j + H] : v[j];
            w[j] = keep + __shfl_xor(send, O, 64);
        }
        return wave_sum_multi<H, O / 2>(w, lane);
    }
}

template <int VPL /* float4 per lane per row: dv = 256*VPL */, int QG, int RPW, int WPH>
__global__ __launch_bounds__(256) void frame_score_kernel(const float* __restrict__ vid, int64_t ctx_l, int S, int64_t nh,
                                                          const float* __restrict__ txt, int q0, int nq,
                                                          float* __restrict__ fs, float* __restrict__ hm,
                                                          float* __restrict__ fr) {
    constexpr int DV = 256 * VPL, UPB = 4 / WPH, NV = RPW * QG;
    static_assert((RPW & (RPW - 1)) == 0 && (QG & (QG - 1)) == 0 && NV <= 64, "row / query counts: powers of two");
    __shared__ float red[4][QG];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = wave % WPH;
    // the (row, query) pair whose total this lane ends up with (wave_sum_multi: value index r * QG + g = the lane's top bits)
    const int my_j = lane / (64 / NV), my_r = my_j / QG, my_g = my_j % QG;
    const bool writer = (lane & (64 / NV - 1)) == 0;                   // one of the 64 / NV lanes that hold the same total
    float4 q[QG][VPL];
#pragma unroll
    for (int g = 0; g < QG; ++g)
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int qi = min(q0 + g, nq - 1);
            q[g][v] = reinterpret_cast<const float4*>(txt + (size_t)qi * DV)[lane + 64 * v];
        }
    for (int64_t h = (int64_t)blockIdx.x * UPB + wave / WPH; h < nh; h += (int64_t)gridDim.x * UPB) {
        const int64_t r_lo = h * S;
        const int n = (int)min((int64_t)S, ctx_l - r_lo);            // frames of this half-block
        const float* base = vid + r_lo * DV;
        float m = -INFINITY;                                         // running max of this lane's (row slot, query)
        for (int j0 = sub * RPW; j0 < n; j0 += WPH * RPW) {
            float4 x[RPW][VPL];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = min(j0 + r, n - 1);
#pragma unroll
                for (int v = 0; v < VPL; ++v)
                    x[r][v] = pf_stream_ld(reinterpret_cast<const float4*>(base + (size_t)row * DV) + lane + 64 * v);
            }
            float part[NV];
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int g = 0; g < QG; ++g) {
                    float s = 0.f;
#pragma unroll
                    for (int v = 0; v < VPL; ++v) s += pf_dot4(x[r][v], q[g][v]);
                    part[r * QG + g] = s;
                }
            const float s = wave_sum_multi<NV>(part, lane);            // = wave_sum of (row j0 + my_r, query q0 + my_g)
            if (j0 + my_r < n) {
                m = fmaxf(m, s);
                if (writer && q0 + my_g < nq) {
                    if (fs) fs[(size_t)(q0 + my_g) * ctx_l + r_lo + j0 + my_r] = s;
                    if (my_r == 0 && j0 == 0) fr[(size_t)(q0 + my_g) * nh + h] = s;      // the block's first frame
                }
            }
        }
        // over the RPW row slots of a query: the lanes that differ in the index's top log2(RPW) bits (lane bits 5, 4, ...)
#pragma unroll
        for (int o = 32; o > 32 / RPW; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        const bool out_lane = writer && my_r == 0;                     // lane g * (64 / NV): query q0 + g
        if (WPH == 1) {
            if (out_lane && q0 + my_g < nq) hm[(size_t)(q0 + my_g) * nh + h] = m;
        } else {                                    // h is uniform over the workgroup: the barriers are too
            if (out_lane) red[wave][my_g] = m;
            __syncthreads();
            if (wave == 0 && lane < QG && q0 + lane < nq)
                hm[(size_t)(q0 + lane) * nh + h] = fmaxf(fmaxf(red[0][lane], red[1][lane]), fmaxf(red[2][lane], red[3][lane]));
            __syncthreads();
        }
    }
}

// win[q][i] = max(hm[q][i-1], hm[q][i], W odd ? fr[q][i+1] : -inf) over the half-blocks that exist (0 <= h < nh):
// window i covers frames [max((i-1)S, 0), min((i-1)S + W, ctx_l)), cone/inference.py:286-292.
__global__ __launch_bounds__(256) void window_combine_kernel(const float* __restrict__ hm, const float* __restrict__ fr,
                                                             int64_t nh, int odd, float* __restrict__ win) {
    const int q = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > nh) return;                                                  // num_window = nh + 1
    const float* a = hm + (size_t)q * nh;
    float m = -INFINITY;
    if (i >= 1) m = a[i - 1];
    if (i < nh) m = fmaxf(m, a[i]);
    if (odd && i + 1 < nh) m = fmaxf(m, fr[(size_t)q * nh + i + 1]);
    win[(size_t)q * (nh + 1) + i] = m;
}

// Two-level stable top-k for long rows: level 1 -- one workgroup per chunk of TK_CH scores extracts the chunk's own
// stable top-k; level 2 -- topk_merge_kernel picks the k best of the (chunks x k) candidates with the same (score desc,
// index asc) order.  Chunks cover ascending index ranges and every list is (score desc, index asc), so the merged list is
// the row's stable descending order (the argument of parallel.merge_topk).
//
// Both levels run the same barrier-free selection (tk_select): every thread holds its share of the values in registers,
// each WAVE extracts the top-k of its own quarter by k passes of (register scan, 6-step shuffle arg-max) -- no LDS traffic,
// no workgroup barrier inside the passes -- and wave 0 merges the four sorted lists (4 k candidates, again in registers)
// behind ONE barrier.  (Round 2 re-read the chunk from LDS in every pass behind three barriers: 156 us for 64 rows of
// 100 001 windows; this form: see profiles/README.md.)
constexpr int TK_CH = 4096;
constexpr int TK_PT = TK_CH / 256;      // values per thread
constexpr int TK_KMAX = 256;            // candidates per wave list held in LDS (4 lists)

// order of the selection: a precedes b iff a.v > b.v or (a.v == b.v and a.i < b.i); "after last" = strictly later
__device__ __forceinline__ bool tk_after(float v, int i, float lv, int li_) { return (v < lv) || (v == lv && i > li_); }
__device__ __forceinline__ bool tk_better(float v, int i, float bv, int bi) { return (v > bv) || (v == bv && i < bi); }

// One wave: k passes over PT (value, index) pairs per lane (index 0x7fffffff = empty slot); pass p's winner goes to
// out_v[p] / out_i[p] (lane 0 writes); returns nothing -- lists shorter than k are padded with (-inf, 0x7fffffff).
template <int PT>
__device__ __forceinline__ void tk_wave_select(const float (&v)[PT], const int (&ix)[PT], int k, float* out_v, int* out_i,
                                               int used = PT /* slots that can hold a value (wave-uniform) */) {
    const int lane = threadIdx.x & 63;
    float last_v = INFINITY;
    int last_i = -1;
    for (int p = 0; p < k; ++p) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < PT; ++u)
            if (u < used && ix[u] != 0x7fffffff && tk_after(v[u], ix[u], last_v, last_i) && tk_better(v[u], ix[u], bv, bi)) { bv = v[u]; bi = ix[u]; }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (tk_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { out_v[p] = bi == 0x7fffffff ? -INFINITY : bv; out_i[p] = bi; }
        if (bi == 0x7fffffff) {                       // (wave-uniform) nothing left: pad the rest of the list
            for (int r = p + 1 + lane; r < k; r += 64) { out_v[r] = -INFINITY; out_i[r] = 0x7fffffff; }
            break;
        }
        last_v = bv; last_i = bi;
    }
}

// Bitonic sort of one (value, index) pair per lane over the 64 lanes of a wave, best first (lane 0 = largest value, ties
// to the lower index); empty slots (index 0x7fffffff, value -inf) sink to the end.
__device__ __forceinline__ void tk_wave_sort(float& v, int& i) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1)
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const float pv = __shfl_xor(v, j, 64);
            const int pi = __shfl_xor(i, j, 64);
            const bool take_better = ((lane & j) == 0) == ((lane & k2) == 0);
            const bool p_better = tk_better(pv, pi, v, i), p_worse = tk_better(v, i, pv, pi);
            if (take_better ? p_better : p_worse) { v = pv; i = pi; }
        }
}

// Threshold selection for k <= 64 (the pre-filter's top-k is 20 - 30): sort the 64 per-lane bests; the k-th of them, T,
// cannot precede the k-th best of ALL values (the k best lane-bests are k distinct values at or ahead of T), so every
// member of the top-k is at or ahead of T; typically ~1.3 k values are (40 of 1 024 at k = 30).  They are compacted into
// `scr` (64 slots of this wave), sorted once more, and lanes 0 .. k-1 hold the answer: ~700 instructions instead of k
// passes over all PT values (5 100 at k = 30, PT = 16).  More than 64 survivors (values clustered in few lanes) -> false:
// the caller falls back to the pass-based selection.  Same total order, hence the same list, bit for bit.
template <int PT>
__device__ __forceinline__ bool tk_wave_select_fast(const float (&v)[PT], const int (&ix)[PT], int k, float* out_v, int* out_i,
                                                    float* scr_v, int* scr_i, int used = PT) {
    const int lane = threadIdx.x & 63;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < PT; ++u)
        if (u < used && ix[u] != 0x7fffffff && tk_better(v[u], ix[u], bv, bi)) { bv = v[u]; bi = ix[u]; }
    tk_wave_sort(bv, bi);
    const float tv = __shfl(bv, k - 1, 64);
    const int ti = __shfl(bi, k - 1, 64);           // 0x7fffffff: fewer than k lanes hold anything -> everything survives
    int c = 0;
#pragma unroll
    for (int u = 0; u < PT; ++u)
        c += (u < used && ix[u] != 0x7fffffff && !tk_better(tv, ti, v[u], ix[u])) ? 1 : 0;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const int total = __shfl(incl, 63, 64);
    if (total > 64) return false;
    int pos = incl - c;
#pragma unroll
    for (int u = 0; u < PT; ++u)
        if (u < used && ix[u] != 0x7fffffff && !tk_better(tv, ti, v[u], ix[u])) { scr_v[pos] = v[u]; scr_i[pos] = ix[u]; ++pos; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // this wave's ds_writes ahead of its ds_reads
    float cv = lane < total ? scr_v[lane] : -INFINITY;
    int ci = lane < total ? scr_i[lane] : 0x7fffffff;
    tk_wave_sort(cv, ci);
    if (lane < k) { out_v[lane] = ci == 0x7fffffff ? -INFINITY : cv; out_i[lane] = ci; }
    return true;
}

// Workgroup of 256: per-wave lists into LDS, then wave 0 merges the four lists into (gv, gi)[0 .. k) in global memory.
template <int PT>
__device__ __forceinline__ void tk_block_select(const float (&v)[PT], const int (&ix)[PT], int k, float* gv, int* gi,
                                                int idx_none) {
    __shared__ float l_v[4 * TK_KMAX];
    __shared__ int l_i[4 * TK_KMAX];
    __shared__ float s_v[4 * 64];
    __shared__ int s_i[4 * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (!(k <= 64 && tk_wave_select_fast<PT>(v, ix, k, l_v + wave * k, l_i + wave * k, s_v + wave * 64, s_i + wave * 64)))
        tk_wave_select<PT>(v, ix, k, l_v + wave * k, l_i + wave * k);
    __syncthreads();
    if (wave != 0) return;
    constexpr int MT = 4 * TK_KMAX / 64;            // merge slots per lane
    float mv[MT];
    int mi[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int e = lane + 64 * t;
        const bool ok = e < 4 * k;
        mv[t] = ok ? l_v[e] : -INFINITY;
        mi[t] = ok ? l_i[e] : 0x7fffffff;
    }
    // the merged list goes through the (now free) first list's LDS slots, then out with the caller's "none" index
    const int used = (4 * k + 63) / 64;
    if (!(k <= 64 && tk_wave_select_fast<MT>(mv, mi, k, l_v, l_i, s_v, s_i, used)))
        tk_wave_select<MT>(mv, mi, k, l_v, l_i, used);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // the list's ds_writes before the wave reads it back
    for (int r = lane; r < k; r += 64) {
        const int i = l_i[r];
        gv[r] = l_v[r];
        gi[r] = i == 0x7fffffff ? idx_none : i;
    }
}

__global__ __launch_bounds__(256) void topk_chunk_kernel(const float* __restrict__ sc, int64_t n, int k,
                                                         float* __restrict__ cval, int* __restrict__ cidx, int n_chunks) {
    const int q = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
    const int64_t base = (int64_t)ch * TK_CH;
    const int m = (int)min((int64_t)TK_CH, n - base);
    const float* row = sc + (size_t)q * n + base;
    float v[TK_PT];
    int ix[TK_PT];
#pragma unroll
    for (int u = 0; u < TK_PT; ++u) {
        const int j = u * 256 + tid;
        const float x = j < m ? row[j] : -INFINITY;
        const bool ok = j < m && x == x;                    // a NaN score is never selected (as in the one-level kernel)
        v[u] = ok ? x : -INFINITY;
        ix[u] = ok ? (int)(base + j) : 0x7fffffff;          // global window index: ascending with j
    }
    tk_block_select<TK_PT>(v, ix, k, cval + ((size_t)q * n_chunks + ch) * k, cidx + ((size_t)q * n_chunks + ch) * k, 0x7fffffff);
}

// level 2: up to 256 * TK_PT candidates per row in registers
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ cval, const int* __restrict__ cidx,
                                                         int n_cand, int k, int32_t* __restrict__ idx,
                                                         float* __restrict__ val) {
    __shared__ float o_v[TK_KMAX];
    const int q = blockIdx.x, tid = threadIdx.x;
    const float* cv = cval + (size_t)q * n_cand;
    const int* ci = cidx + (size_t)q * n_cand;
    float v[TK_PT];
    int ix[TK_PT];
#pragma unroll
    for (int u = 0; u < TK_PT; ++u) {
        const int j = u * 256 + tid;
        v[u] = j < n_cand ? cv[j] : -INFINITY;
        ix[u] = j < n_cand ? ci[j] : 0x7fffffff;
    }
    float* ov = val ? val + (size_t)q * k : o_v;
    tk_block_select<TK_PT>(v, ix, k, ov, idx + (size_t)q * k, -1);
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
__device__ __forceinline__ pf4 pf_mq_ld(const float* p) {       // a frame's 16-B piece of the once-read stream
#if CONE_PF_NT_MQ
    return __builtin_nontemporal_load(reinterpret_cast<const pf4*>(p));
#else
    return *reinterpret_cast<const pf4*>(p);
#endif
}

constexpr int MQ_NT = 768;     // 12 waves: the one workgroup a CU holds (128 KiB of LDS) runs three waves per SIMD

template <int QT /* query tiles of 16 */, bool FS /* also write the frame scores */>
__global__ __launch_bounds__(MQ_NT, 3) void frame_score_mq_kernel(const float* __restrict__ vid, int64_t ctx_l, int dv,
                                                                int S, int64_t nh, const float* __restrict__ txt, int q0,
                                                                int nq, float* __restrict__ fs, float* __restrict__ hm,
                                                                float* __restrict__ fr) {
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
    const int nchunk = dv >> 7;                                          // 128-channel chunks (8 slabs)
    constexpr int NW = MQ_NT / 64;
    // A wave owns whole half-blocks (frames [hS, (h+1)S)): ceil(S / 16) tiles of 16 frames, the last one partial (its spare
    // lanes re-read the block's last frame and are masked), running max per (query, lane) in registers, one value per
    // (query, half-block) out.  The waves of a workgroup take consecutive half-blocks, so their 4-B results of one query
    // fall into one cache line.  The stream is software-pipelined across tiles AND half-blocks: the first 128 channels of
    // the next tile are in flight under the last chunk of this one.
    const int64_t h_step = (int64_t)gridDim.x * NW;
    int64_t h = (int64_t)blockIdx.x * NW + (tid >> 6);
    if (h >= nh) return;
    int64_t r_lo = h * S, r_hi = min(r_lo + S, ctx_l);
    int64_t f0 = r_lo;
    const float* fp = vid + min(f0 + li, r_hi - 1) * dv + 4 * lg;
    pf4 cur[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) cur[s] = pf_mq_ld(fp + 16 * s);
    pf4 mx[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) mx[qt] = pf4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    while (true) {
        // where the stream goes after this tile
        int64_t h2 = h, f2 = f0 + 16, lo2 = r_lo, hi2 = r_hi;
        if (f2 >= r_hi) { h2 = h + h_step; lo2 = h2 * S; hi2 = min(lo2 + S, ctx_l); f2 = lo2; }
        const bool more = h2 < nh;
        const float* fp2 = more ? vid + min(f2 + li, hi2 - 1) * dv + 4 * lg : fp;
        pf4 acc[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) acc[qt] = pf4{0.f, 0.f, 0.f, 0.f};
        // Burst loads (the 8 x 64 B of a row's next 128 channels back to back: whole 128-B lines), compiler-scheduled MFMAs.
        // Measured on one box at 64 queries (round 3, tools/ab_variants.sh): this loop 3.76 - 3.79 ms; the same with the query
        // fragments software-pipelined and the MFMA order pinned 3.97; with a rolling per-slab prefetch instead of the bursts
        // 4.07 - 4.10.  PMC: 84 % MFMA-busy at an effective 1.85 GHz -- the exact-fp32 matrix pipe under a 3.4 TB/s stream
        // is power-limited, three waves per SIMD already cover each other's LDS and memory waits.
        for (int c = 0; c < nchunk; ++c) {
            const float* np = c + 1 < nchunk ? fp + 128 * (c + 1) : fp2;
            pf4 nxt[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) nxt[s] = pf_mq_ld(np + 16 * s);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const pf4 aa = *reinterpret_cast<const pf4*>(qs + (qt * ns + c * 8 + s) * 256 + rd);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[r], cur[s][r], acc[qt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) cur[s] = nxt[s];
        }
        const bool valid = f0 + li < r_hi;
        const bool first = f0 == r_lo && li == 0;       // lane li = 0 of the block's first tile = frame hS
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (valid) mx[qt][r] = fmaxf(mx[qt][r], acc[qt][r]);
                const int qg = q0 + qt * 16 + 4 * lg + r;
                if (FS && valid && qg < nq) fs[(size_t)qg * ctx_l + f0 + li] = acc[qt][r];
                if (first && qg < nq) fr[(size_t)qg * nh + h] = acc[qt][r];
            }
        }
        if (f0 + 16 >= r_hi) {                          // half-block done: max over its 16 frame lanes, one store per query
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = mx[qt][r];
                    v = fmaxf(v, __shfl_xor(v, 1, 64));
                    v = fmaxf(v, __shfl_xor(v, 2, 64));
                    v = fmaxf(v, __shfl_xor(v, 4, 64));
                    v = fmaxf(v, __shfl_xor(v, 8, 64));
                    const int qg = q0 + qt * 16 + 4 * lg + r;
                    if (li == 0 && qg < nq) hm[(size_t)qg * nh + h] = v;
                    mx[qt][r] = -INFINITY;
                }
        }
        if (!more) break;
        h = h2; f0 = f2; r_lo = lo2; r_hi = hi2; fp = fp2;
    }
}

static int prefilter_grid_setup(int* n_cu_out) {
    static DeviceOnce once;     // the many-query kernel's 128 KiB of LDS: opt-in once per device; its grid = one workgroup per CU
    const hipError_t rc = device_once(once, [] {
        const void* fns[6] = {(const void*)frame_score_mq_kernel<4, false>, (const void*)frame_score_mq_kernel<4, true>,
                              (const void*)frame_score_mq_kernel<2, false>, (const void*)frame_score_mq_kernel<2, true>,
                              (const void*)frame_score_mq_kernel<1, false>, (const void*)frame_score_mq_kernel<1, true>};
        hipError_t e = hipSuccess;
        for (int i = 0; i < 6 && e == hipSuccess; ++i)
            e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        return e;
    }, n_cu_out);
    return rc == hipSuccess ? 0 : -1;
}

static int launch_frame_scores_mq(const float* vid, int64_t ctx_l, int dv, int S, int64_t nh, const float* txt, int nq,
                                  float* fs, float* hm, float* fr, hipStream_t s) {
    // 64 queries per launch while their vectors fit the LDS next to nothing else (64 x 512 x 4 B = 128 KiB), else 32
    const int qpl = dv <= 512 ? 64 : 32;
    int n_cu = 0;
    if (prefilter_grid_setup(&n_cu)) { set_error("prefilter: raising the LDS limit of the many-query kernel failed"); return CONE_E_HIP; }
    int64_t blocks = (nh + MQ_NT / 64 - 1) / (MQ_NT / 64);
    if (blocks > n_cu) blocks = n_cu;                                  // one workgroup per CU, grid-stride over half-blocks
    for (int q0 = 0; q0 < nq;) {
        const int rem = nq - q0;
        const bool wide = qpl == 64 && rem > 32;                           // 4 query tiles, else 2, or 1 for <= 16 queries:
        const bool one = rem <= 16;                                        // half the matrix work of two, the stream is HBM-bound
        ProfScope ps(PK_FRAME_SCORE, ctx_l, dv, rem < (wide ? 64 : 32) ? rem : (wide ? 64 : 32), nullptr, s);
#define CONE_MQ_LAUNCH(QT, FS, QN)                                                                                      \
    hipLaunchKernelGGL((frame_score_mq_kernel<QT, FS>), dim3((unsigned)blocks), dim3(MQ_NT), (size_t)(QN) * dv * 4, s, vid, \
                       ctx_l, dv, S, nh, txt, q0, nq, fs, hm, fr)
        if (wide) { if (fs) CONE_MQ_LAUNCH(4, true, 64); else CONE_MQ_LAUNCH(4, false, 64); }
        else if (one) { if (fs) CONE_MQ_LAUNCH(1, true, 16); else CONE_MQ_LAUNCH(1, false, 16); }
        else { if (fs) CONE_MQ_LAUNCH(2, true, 32); else CONE_MQ_LAUNCH(2, false, 32); }
#undef CONE_MQ_LAUNCH
        CONE_LAUNCH_CHECK();
        q0 += wide ? 64 : 32;
    }
    return 0;
}

// ---- OPT-IN: many queries on the bf16 matrix cores, every fp32 product as six partial products of three-piece operands --
// At 64 queries the exact-fp32 kernel above is bound by the matrix pipe, not by the stream (32 FLOP per byte is past the
// ridge of 19.7: 3.7 ms where HBM needs 2.1).  gfx950's bf16 MFMA runs 16 x faster, and an fp32 product can ride on it without
// giving up fp32 accuracy (ffn_split.hip; tools/probe/split_bf16_probe.hip: the error of the six-product form equals the
// fp32 chain's): x = xh + xm + xl exactly (three bf16 pieces, 24 bits), x w ~= xl wh + xh wl + xm wm + xh wm + xm wh + xh wh.
//   * D[query][frame] tiles of v_mfma_f32_16x16x32_bf16: B = the frame's 32 channels of a k-step = the two float4 a lane
//     already streams (k slot (lg, j) <-> channel 32 t + 16 (j / 4) + 4 lg + j % 4), split into pieces in registers ONCE per
//     frame tile and used for all four query tiles; A = the query pieces as 1-KiB slabs [16 queries][4 lg][8 bf16] (a
//     lane's ds_read_b128: conflict-free).
//   * The query pieces of 64 queries x 512 channels are 192 KiB -- more than the LDS.  They are split ONCE per launch into an
//     image in the caller's workspace (pf_split_queries_kernel; L2-resident from then on), laid out [128-channel chunk][piece]
//     [query tile][k-step] in exactly the byte order of the LDS slabs, and the workgroup's twelve waves walk their frame tiles
//     in LOCK-STEP on the chunk index: the 48 KiB of chunk c + 2 stream in by LDS-DMA (four 1-KiB pieces per wave: no VGPRs,
//     no addressing) into a three-stage ring while chunk c is multiplied -- one raw barrier per chunk step.  (First cut: 32
//     queries per workgroup and two workgroups on one XCD per frame range, hoping for L2 hits on the second read: FETCH_SIZE
//     1.49 x the arena, 3.78 ms -- the waves' positions are megabytes apart and a 4 MB L2 cannot bridge that.)
// The running max per half window, the first-frame scores and the output format are those of the fp32 kernel.
typedef short pf_s8 __attribute__((ext_vector_type(8)));
typedef unsigned pf_u4 __attribute__((ext_vector_type(4)));
typedef unsigned pf_u2 __attribute__((ext_vector_type(2)));
typedef float pf_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 pf_b2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pf_pk(float a, float b) {       // two floats -> packed bf16 pair, round to nearest even
    return __builtin_bit_cast(unsigned, __builtin_convertvector(pf_f2{a, b}, pf_b2));
}
__device__ __forceinline__ void pf_split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = pf_pk(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = pf_pk(ra, rb);
    l = pf_pk(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ void pf_split8(const pf4& v0, const pf4& v1, pf_s8& h, pf_s8& m, pf_s8& l) {
    unsigned a[4], b[4], c[4];
    pf_split2(v0[0], v0[1], a[0], b[0], c[0]);
    pf_split2(v0[2], v0[3], a[1], b[1], c[1]);
    pf_split2(v1[0], v1[1], a[2], b[2], c[2]);
    pf_split2(v1[2], v1[3], a[3], b[3], c[3]);
    h = __builtin_bit_cast(pf_s8, pf_u4{a[0], a[1], a[2], a[3]});
    m = __builtin_bit_cast(pf_s8, pf_u4{b[0], b[1], b[2], b[3]});
    l = __builtin_bit_cast(pf_s8, pf_u4{c[0], c[1], c[2], c[3]});
}
#define PF_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0)
#define PF_GLDS16(src, dst) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

constexpr int MQ3_QT = 4;                           // query tiles of 16 per launch
constexpr int MQ3_CHUNK = 3 * MQ3_QT * 4 * 1024;    // bytes of one 128-channel chunk of the image: [piece][query tile][k-step] slabs
constexpr int MQ3_NBUF = 3;
size_t frame_scores_split_image_bytes(int dv) { return (size_t)(dv >> 7) * MQ3_CHUNK; }

// queries q0 .. q0 + 63 (zeros past nq) -> the image: chunk c = channels [128 c, 128 c + 128), slab (piece, qt, tt), inside a
// slab query row r, lane group lg, half u: 8 B = channels 32 t + 16 u + 4 lg + 0 .. 3 of step t = 4 c + tt
__global__ __launch_bounds__(256) void pf_split_queries_kernel(const float* __restrict__ txt, int dv, int q0, int nq,
                                                               char* __restrict__ img) {
    const int i = blockIdx.x * 256 + threadIdx.x;           // (query, float4 of its vector)
    if (i >= MQ3_QT * 16 * (dv >> 2)) return;
    const int qi = i / (dv >> 2), c4 = i % (dv >> 2);
    const int qg = q0 + qi;
    pf4 v = pf4{0.f, 0.f, 0.f, 0.f};
    if (qg < nq) v = *reinterpret_cast<const pf4*>(txt + (size_t)qg * dv + c4 * 4);
    unsigned h0, m0, l0, h1, m1, l1;
    pf_split2(v[0], v[1], h0, m0, l0);
    pf_split2(v[2], v[3], h1, m1, l1);
    const int t = c4 >> 3, u = (c4 >> 2) & 1, lgq = c4 & 3, row = qi & 15;
    const int c = t >> 2, tt = t & 3, qt = qi >> 4;
    char* base = img + (size_t)c * MQ3_CHUNK + ((qt * 4 + tt) << 10) + row * 64 + lgq * 16 + u * 8;
    constexpr int pstride = MQ3_QT * 4 * 1024;
    *reinterpret_cast<pf_u2*>(base) = pf_u2{h0, h1};
    *reinterpret_cast<pf_u2*>(base + pstride) = pf_u2{m0, m1};
    *reinterpret_cast<pf_u2*>(base + 2 * pstride) = pf_u2{l0, l1};
}

__global__ __launch_bounds__(MQ_NT, 3) void frame_score_mq3_kernel(const float* __restrict__ vid, int64_t ctx_l, int dv, int S,
                                                                 int64_t nh, const char* __restrict__ img, int q0, int nq,
                                                                 float* __restrict__ hm, float* __restrict__ fr) {
    constexpr int QT = MQ3_QT;
    extern __shared__ __attribute__((aligned(16))) char ring[];         // MQ3_NBUF stages of MQ3_CHUNK bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int nchunk = dv >> 7;
    constexpr int NW = MQ_NT / 64;
    constexpr int pstride = QT * 4 * 1024;
    const int rd = li * 64 + lg * 16;
    // tile slots of this workgroup = the tiles of its busiest wave (every wave takes every barrier; a wave out of tiles only
    // keeps the ring fed).  Half-blocks have ceil(S / 16) tiles, the video's last one possibly fewer.
    const int64_t h_step = (int64_t)gridDim.x * NW;
    const int tps = (S + 15) >> 4;
    const int tps_last = (int)((ctx_l - (nh - 1) * S + 15) >> 4);
    auto tiles_of = [&](int w) -> int64_t {
        const int64_t h0 = (int64_t)blockIdx.x * NW + w;
        if (h0 >= nh) return 0;
        const int64_t n = (nh - 1 - h0) / h_step + 1;
        const bool owns_last = (nh - 1 - h0) % h_step == 0;
        return n * tps - (owns_last ? tps - tps_last : 0);
    };
    int64_t slots = 0;
    for (int w = 0; w < NW; ++w) { const int64_t t = tiles_of(w); slots = t > slots ? t : slots; }
    if (slots == 0) return;
    const int64_t n_steps = slots * nchunk;
    // ring: the wave's four 1-KiB pieces of a chunk
    auto stream = [&](int64_t g) {                                      // chunk of global step g -> stage g % MQ3_NBUF
        const char* src = img + (size_t)(g % nchunk) * MQ3_CHUNK + (wave * 4 << 10) + lane * 16;
        char* dst = ring + (int)(g % MQ3_NBUF) * MQ3_CHUNK + (wave * 4 << 10);
#pragma unroll
        for (int i = 0; i < 4; ++i) PF_GLDS16(src + (i << 10), dst + (i << 10));
    };
    stream(0);
    if (n_steps > 1) stream(1);

    int64_t h = (int64_t)blockIdx.x * NW + wave;
    bool live = h < nh;
    int64_t r_lo = live ? h * S : 0, r_hi = live ? min(r_lo + S, ctx_l) : 1;
    int64_t f0 = r_lo;
    const float* fp = vid + min(f0 + li, r_hi - 1) * dv + 4 * lg;
    pf4 cur[8];
    if (live) {
#pragma unroll
        for (int s = 0; s < 8; ++s) cur[s] = pf_mq_ld(fp + 16 * s);
    }
    pf4 mx[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) mx[qt] = pf4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int64_t g = 0;
    for (int64_t slot = 0; slot < slots; ++slot) {
        int64_t h2 = h, f2 = f0 + 16, lo2 = r_lo, hi2 = r_hi;
        if (f2 >= r_hi) { h2 = h + h_step; lo2 = h2 * S; hi2 = min(lo2 + S, ctx_l); f2 = lo2; }
        const bool more = live && h2 < nh;
        const float* fp2 = more ? vid + min(f2 + li, hi2 - 1) * dv + 4 * lg : fp;
        pf4 acc[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) acc[qt] = pf4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nchunk; ++c, ++g) {
            // stage g has landed (this wave's pieces: everything but the 4 pieces of stage g + 1 and, with the loads of the
            // previous step consumed by the compiler's own waits, nothing else); the barrier makes all twelve shares visible
            // and tells that everybody is done with stage g - 1, which the pieces of stage g + 2 overwrite
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            pf4 nxt[8];
            if (live) {
                const float* np = c + 1 < nchunk ? fp + 128 * (c + 1) : fp2;
#pragma unroll
                for (int s = 0; s < 8; ++s) nxt[s] = pf_mq_ld(np + 16 * s);
            }
            if (g + 2 < n_steps) stream(g + 2);
            if (live) {
                const char* st = ring + (int)(g % MQ3_NBUF) * MQ3_CHUNK + rd;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    pf_s8 fh, fm, fl;
                    pf_split8(cur[2 * tt], cur[2 * tt + 1], fh, fm, fl);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        const char* sl = st + ((qt * 4 + tt) << 10);
                        const pf_s8 ah = *reinterpret_cast<const pf_s8*>(sl);
                        const pf_s8 am = *reinterpret_cast<const pf_s8*>(sl + pstride);
                        const pf_s8 al = *reinterpret_cast<const pf_s8*>(sl + 2 * pstride);
                        PF_MFMA(acc[qt], al, fh);                       // small terms first
                        PF_MFMA(acc[qt], ah, fl);
                        PF_MFMA(acc[qt], am, fm);
                        PF_MFMA(acc[qt], am, fh);
                        PF_MFMA(acc[qt], ah, fm);
                        PF_MFMA(acc[qt], ah, fh);
                    }
                }
#pragma unroll
                for (int s = 0; s < 8; ++s) cur[s] = nxt[s];
            }
        }
        if (!live) continue;
        const bool valid = f0 + li < r_hi;
        const bool first = f0 == r_lo && li == 0;       // lane li = 0 of the block's first tile = frame hS
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (valid) mx[qt][r] = fmaxf(mx[qt][r], acc[qt][r]);
                const int qg = q0 + qt * 16 + 4 * lg + r;
                if (first && qg < nq) fr[(size_t)qg * nh + h] = acc[qt][r];
            }
        }
        if (f0 + 16 >= r_hi) {                          // half-block done: max over its 16 frame lanes, one store per query
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = mx[qt][r];
                    v = fmaxf(v, __shfl_xor(v, 1, 64));
                    v = fmaxf(v, __shfl_xor(v, 2, 64));
                    v = fmaxf(v, __shfl_xor(v, 4, 64));
                    v = fmaxf(v, __shfl_xor(v, 8, 64));
                    const int qg = q0 + qt * 16 + 4 * lg + r;
                    if (li == 0 && qg < nq) hm[(size_t)qg * nh + h] = v;
                    mx[qt][r] = -INFINITY;
                }
        }
        if (!more) { live = false; continue; }
        h = h2; f0 = f2; r_lo = lo2; r_hi = hi2; fp = fp2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA may outlive the workgroup's LDS
}

bool frame_scores_split_supported(int dv, int nq) { return nq >= 8 && dv % 128 == 0; }

static int launch_frame_scores_mq3(const float* vid, int64_t ctx_l, int dv, int S, int64_t nh, const float* txt, int nq, float* hm,
                                   float* fr, char* img, hipStream_t s) {
    static DeviceOnce once;
    int n_cu = 0;
    if (device_once(once, [] {
            return hipFuncSetAttribute((const void*)frame_score_mq3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       MQ3_NBUF * MQ3_CHUNK);
        }, &n_cu) != hipSuccess) {
        set_error("prefilter: raising the LDS limit of the split many-query kernel failed");
        return CONE_E_HIP;
    }
    int64_t blocks = (nh + MQ_NT / 64 - 1) / (MQ_NT / 64);
    if (blocks > n_cu) blocks = n_cu;                                  // one workgroup per CU, grid-stride over half-blocks
    for (int q0 = 0; q0 < nq; q0 += 64) {
        const int rem = nq - q0;
        hipLaunchKernelGGL(pf_split_queries_kernel, dim3((unsigned)((MQ3_QT * 16 * (dv >> 2) + 255) / 256)), dim3(256), 0, s, txt, dv,
                           q0, nq, img);
        CONE_LAUNCH_CHECK();
        ProfScope ps(PK_FRAME_SCORE, ctx_l, dv, rem < 64 ? rem : 64, nullptr, s);
        hipLaunchKernelGGL(frame_score_mq3_kernel, dim3((unsigned)blocks), dim3(MQ_NT), MQ3_NBUF * MQ3_CHUNK, s, vid, ctx_l, dv, S, nh,
                           (const char*)img, q0, nq, hm, fr);
        CONE_LAUNCH_CHECK();
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
            // (ordinary loads: the other groups of this video read the same rows again -- a split's arena sits in the caches)
            for (int v = 0; v < VPL; ++v) x[r][v] = reinterpret_cast<const float4*>(vid + (size_t)row * DV)[lane + 64 * v];
        }
        float part[RPW * 4];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < VPL; ++v) s += pf_dot4(x[r][v], q[j][v]);
                part[r * 4 + j] = s;
            }
        // the 16 sums by one butterfly (wave_sum_multi): lane l ends up with the total of (row l / 16, query (l / 4) % 4)
        const float s = wave_sum_multi<RPW * 4>(part, lane);
        const int my_r = lane >> 4;
        int my_qi = qi[0];
#pragma unroll
        for (int j = 1; j < 4; ++j) my_qi = ((lane >> 2) & 3) == j ? qi[j] : my_qi;
        if ((lane & 3) == 0 && r0 + my_r < ctx_l && my_qi >= 0) fs[q_fs_off[my_qi] + r0 + my_r] = s;
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
    if (n <= 1024) {
        // a short row (an Ego4D video: ~22 windows): the stable descending RANK of every window by counting -- rank = #{greater}
        // + #{equal with a lower index}, window j goes to slot rank if rank < k -- one pass and one barrier instead of k selection
        // passes with three barriers each (23 -> 5 us for one query; the same list: the order is a total one).  NaN scores
        // (an all-zero adapted clip row: cone/inference.py:258 divides by an un-eps'd norm) are ordered like torch.sort orders
        // them -- ahead of every number, equal among themselves -- so the ranks stay a permutation and every slot is written
        __shared__ float s_row[1024];
        for (int j = tid; j < n; j += 256) s_row[j] = row[j];
        for (int p = n + tid; p < k; p += 256) idx[(size_t)q * k + p] = -1;        // fewer windows than k: pad with -1
        __syncthreads();
        for (int j = tid; j < n; j += 256) {
            const float v = s_row[j];
            const bool v_nan = v != v;
            int rank = 0;
            for (int i = 0; i < n; ++i) {           // (every thread reads the same address: LDS broadcast)
                const float x = s_row[i];
                const bool x_nan = x != x;
                rank += (x > v) || (x_nan && !v_nan) || ((x == v || (x_nan && v_nan)) && i < j);
            }
            if (rank < k) idx[(size_t)q * k + rank] = j;
        }
        return;
    }
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
static int launch_frame_scores(const float* vid, int64_t ctx_l, int S, int64_t nh, const float* txt, int nq, float* fs,
                               float* hm, float* fr, hipStream_t s) {
#ifndef CONE_PF_RPW
#define CONE_PF_RPW 4
#endif
#ifndef CONE_PF_WGS_PER_CU
#define CONE_PF_WGS_PER_CU 8
#endif
    constexpr int RPW = CONE_PF_RPW;
    // long videos: one wave per half-block, 8 workgroups per CU grid-striding; short ones: a workgroup per half-block
    const bool wide = nh < 4096;
    int64_t blocks = wide ? nh : (nh + 3) / 4;
    if (blocks > 256 * CONE_PF_WGS_PER_CU) blocks = 256 * CONE_PF_WGS_PER_CU;
    for (int q0 = 0; q0 < nq;) {
        const int rem = nq - q0;
        // 3 remaining queries ride a 4-query launch (the fourth slot repeats the last query and stores nothing): one pass over
        // the video instead of two; a query's bits do not depend on the launch it shares (pf_dot4, wave_sum_multi)
        const int qg = rem >= 3 ? 4 : (rem >= 2 ? 2 : 1);
        ProfScope ps(PK_FRAME_SCORE, ctx_l, 256 * VPL, qg, nullptr, s);
#define CONE_FS_LAUNCH(QG, WPH)                                                                                        \
    hipLaunchKernelGGL((frame_score_kernel<VPL, QG, RPW, WPH>), dim3((unsigned)blocks), dim3(256), 0, s, vid, ctx_l, S, \
                       nh, txt, q0, nq, fs, hm, fr)
        if (qg == 4) { if (wide) CONE_FS_LAUNCH(4, 4); else CONE_FS_LAUNCH(4, 1); }
        else if (qg == 2) { if (wide) CONE_FS_LAUNCH(2, 4); else CONE_FS_LAUNCH(2, 1); }
        else { if (wide) CONE_FS_LAUNCH(1, 4); else CONE_FS_LAUNCH(1, 1); }
#undef CONE_FS_LAUNCH
        q0 += qg;
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

extern "C" size_t cone_prefilter_scores_workspace(int64_t ctx_l, int nq, int W) {
    const int S = W / 2;
    if (S <= 0 || ctx_l <= 0 || nq <= 0) return 0;
    const size_t nh = (size_t)((ctx_l + S - 1) / S);
    return 2 * cone::align_up((size_t)nq * nh * sizeof(float), 256);
}

extern "C" size_t cone_prefilter_scores_split_workspace(int64_t ctx_l, int nq, int W, int dv) {
    const size_t base = cone_prefilter_scores_workspace(ctx_l, nq, W);
    return base ? base + cone::frame_scores_split_image_bytes(dv) : 0;
}

static int prefilter_scores_impl(const float* vid, int64_t ctx_l, int dv, const float* txt, int nq, int W, int S,
                                 float* frame_scores, float* win_scores, void* ws, size_t ws_bytes, void* stream, bool split) {
    CONE_REQUIRE(vid && txt && win_scores, "prefilter: null argument");
    CONE_REQUIRE(ctx_l > 0 && nq > 0 && W > 0 && S > 0 && S == W / 2, "prefilter: bad sizes ctx_l=%lld nq=%d W=%d S=%d",
                 (long long)ctx_l, nq, W, S);
    CONE_REQUIRE(dv == 256 || dv == 512 || dv == 768 || dv == 1024,
                 "prefilter: feature dim %d not in {256,512,768,1024}", dv);
    const size_t need = cone_prefilter_scores_workspace(ctx_l, nq, W);
    CONE_REQUIRE(ws && ws_bytes >= need, "prefilter: workspace too small (%zu < %zu)", ws_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    const int64_t nh = (ctx_l + S - 1) / S;
    float* hm = (float*)ws;
    float* fr = (float*)((char*)ws + need / 2);
    int rc;
    if (split && !frame_scores && cone::frame_scores_split_supported(dv, nq)) {
        // opt-in: the same stream on the bf16 matrix cores (three-piece operands, six partial products: fp32 accuracy); the
        // split query image lives behind the two score planes of the workspace
        CONE_REQUIRE(ws_bytes >= need + cone::frame_scores_split_image_bytes(dv), "prefilter (split): workspace too small (%zu < %zu)",
                     ws_bytes, need + cone::frame_scores_split_image_bytes(dv));
        rc = cone::launch_frame_scores_mq3(vid, ctx_l, dv, S, nh, txt, nq, hm, fr, (char*)ws + need, s);
    } else if (nq >= CONE_PF_MQ_MIN) {
        // Many queries over one video: the clip arena is read once for up to 64 queries by the fp32-MFMA kernel
        // (BASELINE configs 3 / 5) instead of nq / 4 VALU passes over the features.  From 5 queries on: they would be two
        // passes of the streaming kernel (3.9 ms on the 12.7 GB video), one 16-query tile of this one is a single pass.
        rc = cone::launch_frame_scores_mq(vid, ctx_l, dv, S, nh, txt, nq, frame_scores, hm, fr, s);
    } else {
        switch (dv / 256) {
            case 1: rc = cone::launch_frame_scores<1>(vid, ctx_l, S, nh, txt, nq, frame_scores, hm, fr, s); break;
            case 2: rc = cone::launch_frame_scores<2>(vid, ctx_l, S, nh, txt, nq, frame_scores, hm, fr, s); break;
            case 3: rc = cone::launch_frame_scores<3>(vid, ctx_l, S, nh, txt, nq, frame_scores, hm, fr, s); break;
            default: rc = cone::launch_frame_scores<4>(vid, ctx_l, S, nh, txt, nq, frame_scores, hm, fr, s); break;
        }
    }
    if (rc) return rc;
    hipLaunchKernelGGL(cone::window_combine_kernel, dim3((unsigned)((nh + 1 + 255) / 256), nq), dim3(256), 0, s, hm, fr, nh,
                       W & 1, win_scores);
    CONE_LAUNCH_CHECK();
    return 0;
}

extern "C" int cone_prefilter_scores(const float* vid, int64_t ctx_l, int dv, const float* txt, int nq, int W,
                                     int S, float* frame_scores, float* win_scores, void* ws, size_t ws_bytes,
                                     void* stream) {
    return prefilter_scores_impl(vid, ctx_l, dv, txt, nq, W, S, frame_scores, win_scores, ws, ws_bytes, stream, false);
}

extern "C" int cone_prefilter_scores_split(const float* vid, int64_t ctx_l, int dv, const float* txt, int nq, int W,
                                           int S, float* win_scores, void* ws, size_t ws_bytes, void* stream) {
    return prefilter_scores_impl(vid, ctx_l, dv, txt, nq, W, S, nullptr, win_scores, ws, ws_bytes, stream, true);
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
    const int64_t n_chunks = (num_window + cone::TK_CH - 1) / cone::TK_CH;
    // two-level selection: lists of <= TK_KMAX per wave, all candidates of a row in the merge workgroup's registers
    if (need == 0 || k > cone::TK_KMAX || n_chunks * k > 256 * cone::TK_PT)
        return cone_topk_windows(win_scores, nq, num_window, k, idx, val, stream);
    CONE_REQUIRE(ws && ws_bytes >= need, "topk: workspace too small (%zu < %zu)", ws_bytes, need);
    float* cval = (float*)ws;
    int* cidx = (int*)((char*)ws + need / 2);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cone::topk_chunk_kernel, dim3((unsigned)n_chunks, nq), dim3(256), 0, s, win_scores, num_window, k, cval,
                       cidx, (int)n_chunks);
    CONE_LAUNCH_CHECK();
    hipLaunchKernelGGL(cone::topk_merge_kernel, dim3(nq), dim3(256), 0, s, cval, cidx, (int)(n_chunks * k), k, idx, val);
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
