// Fused decoder cross-attention on the fp32 matrix cores (cone/transformer.py:308-311 -> nn.MultiheadAttention), the
// memory K / V projections folded into the 8 * NQ (slot, head) pairs exactly as in dec_cross.hip:
//
//   score[p][j] = qk_p . (mem_j + pos_j) (+ a per-pair constant that cancels in the softmax),  qk_p = W_k,h^T (q_p / sqrt(32))
//   out_p       = W_v,h (sum_j P[p][j] mem_j) + b_v,h
//
// dec_cross.hip runs the two contractions over the 256 channels (2 x 2 MFLOP per window and layer, 75 % of the kernel)
// on the VALU at ~27 % of its rate; here they are v_mfma_f32_16x16x4_f32 tiles (exact fp32, same results up to
// summation order), one 8-wave workgroup per window:
//   0. qk_p for the 40 pairs (VALU; thread = channel, W_k read coalesced), written to LDS as MFMA operand slabs
//      [pair tile][16 channels]: [16 pairs][16 floats], 16-B chunks XOR-swizzled (conflict-free ds_read_b128);
//   A. S^T tile = keys . qk^T: wave = one 16-key tile.  A operand = the tile's key rows straight from global memory
//      into registers (lane = key, float4 = 4 consecutive channels: k slot lg of step (q, r) <-> channel 16 q + 4 lg + r,
//      the same permutation on both operands), position rows of clip tokens added from the static sine table; B operand
//      = the qk slabs.  Accumulator register r of lane (li, lg) = score[key 4 lg + r][pair li];
//   B. softmax over keys: in-lane + two shuffles inside the tile, one LDS exchange across the key tiles (waves);
//      probabilities go to LDS as Pt[key][pair];
//   C. ctx^T = mem^T . Pt: wave = two 16-channel tiles x three pair tiles; A operand = memory values read by
//      channel (lane = channel, k slot = key: 4 keys x 64 B per load), B operand = Pt (lane = pair, k slot = key);
//   D. out = W_v,h ctx_p + b_v (VALU; thread = output channel, W_v^T read coalesced), as in dec_cross.hip.
// 48 of 40 pairs and ceil(L / 16) * 16 of L keys are multiplied (padding), 2 688 MFMAs per 101-token window.
#include <mutex>

#include "common.h"

namespace cone {

typedef float g2v __attribute__((ext_vector_type(2)));
typedef float g4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int dcm_swz16(int row) { return (0x1230 >> (((row >> 2) & 3) * 4)) & 3; }

template <int KTW, int NQ = 5>
struct DecCrossMfmaCfg {
    static constexpr int NP = 8 * NQ, NPT = (NP + 15) / 16, NPP = 16 * NPT;   // pairs, pair tiles, padded pairs (5 slots: 40, 3, 48)
    static constexpr int KP = 128 * KTW;               // key capacity: 8 waves x KTW tiles x 16
    static constexpr int CTX_LD = 260;                 // ctx row stride (floats): 16-B aligned rows, spread banks
    static constexpr int QK_FLOATS = NPT * 16 * 256;   // operand slabs of qk (48 KiB)
    static constexpr int A_FLOATS = NPP * CTX_LD > QK_FLOATS ? NPP * CTX_LD : QK_FLOATS;    // qk slabs, later ctx
    static constexpr int PT_FLOATS = KP * NPP;          // Pt[key][pair]; first the scaled queries, last stage D's partials
    static constexpr int RED_FLOATS = 2 * 8 * NPP;      // per-wave softmax maxima / sums
    static constexpr int LDS_FLOATS = A_FLOATS + PT_FLOATS + RED_FLOATS;
    static_assert(A_FLOATS >= 8 * NQ * 256, "region A holds stage D's partial sums");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
    // two workgroups per CU (<= 128 VGPRs) where the LDS allows it and the folds' accumulators (NQ float4 each) fit
    static constexpr int MIN_WAVES = (KTW == 1 && NQ <= 5) ? 4 : 2;
};

// NQ = decoder slots (num_queries, the first argument of the reference's training scripts): 5 = every shipped configuration;
// 3 / 8 / 10 are instantiated for the table form too (Moment-DETR's own default is 10) -- more pair tiles, more LDS (one
// workgroup per CU from 8 slots on), the same stages.
template <int KTW, bool POSTAB, int NQ = 5>
__global__ __launch_bounds__(512, (DecCrossMfmaCfg<KTW, NQ>::MIN_WAVES)) void dec_cross_mfma_kernel(const float* __restrict__ DQ,
                                                                 const float* __restrict__ XP,
                                                                 const float* __restrict__ X,
                                                                 const float* __restrict__ pos_rows,
                                                                 const int* __restrict__ vlen,
                                                                 const int* __restrict__ off,
                                                                 const float* __restrict__ Wk,
                                                                 const float* __restrict__ WvT,
                                                                 const float* __restrict__ bv, float* __restrict__ OUT,
                                                                 const float* __restrict__ QKS, float* __restrict__ QKS_OUT,
                                                                 const float* __restrict__ sal_w, const float* __restrict__ sal_b,
                                                                 float* __restrict__ sal, int sal_ld) {
    // sal != null (POSTAB only): the saliency head rides along -- sal[b][p] = <memory row of clip p, sal_w> + sal_b
    // (cone/model.py:119-122) from the very registers stage A holds the window's memory rows in, instead of a separate pass
    // over the 2 GB of memory rows (0.4 ms per 20 000-window step).  Entries of padded clips are the caller's zeros.
    // QKS != null: the qk operand slabs are window-independent (first decoder layer: tgt = 0, the queries are the
    // same for every window) and were written once by a one-workgroup launch of this kernel with QKS_OUT set.
    using C = DecCrossMfmaCfg<KTW, NQ>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* qkf = smem;                          // [NPT][16] slabs of [16 pairs][16 channels]; later ctx[NPP][260]
    float* Pt = smem + C::A_FLOATS;             // [KP][48]
    float* smax = Pt + C::PT_FLOATS;            // [8][48]
    float* ssum = smax + 8 * C::NPP;            // [8][48]
    const int b = blockIdx.x;
    const int t0 = off[b];
    const int L = min(off[b + 1] - t0, C::KP);
    const int nkt = (L + 15) >> 4;
    const int lv = POSTAB ? vlen[b] : 0;
    const float* __restrict__ KEYS = POSTAB ? X : XP;
    const float* __restrict__ prow = POSTAB ? pos_rows + (size_t)(lv * (lv - 1) / 2) * 256 : nullptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;

    // More than 5 slots: the window's query rows go through LDS (stage 0 reads them as wave-uniform broadcasts).  As scalar
    // operands 8 / 10 slots x 32 values are hoisted into more SGPRs than there are: 235 / 281 spilled to VGPR lanes, 448 / 745
    // v_readlane per wave.  Requested FIRST, so that the wait for them does not cover the key rows behind them.
#ifndef CONE_DCM_QLDS_MIN
#define CONE_DCM_QLDS_MIN 5
#endif
    constexpr bool QLDS = NQ > CONE_DCM_QLDS_MIN;
    constexpr int QREG = QLDS ? (NQ * 64 + 511) / 512 : 1;
    g4v qreg[QREG];
    if (QLDS && !QKS) {
        const g4v* qsrc = reinterpret_cast<const g4v*>(DQ + (size_t)b * NQ * 256);
#pragma unroll
        for (int j = 0; j < QREG; ++j) {
            const int i = tid + 512 * j;
            if (i < NQ * 64) qreg[j] = qsrc[i];
        }
    }
    // ---- this wave's key rows (A operand of stage A: lane = key, float4 = channels 16 q + 4 lg ..) are requested first:
    // their HBM / L2 latency runs under stage 0
    g4v xk[KTW][16];
#pragma unroll
    for (int w = 0; w < KTW; ++w) {
        const int kt = wave + 8 * w;
        if (kt < nkt && !QKS_OUT) {
            const int key = min(kt * 16 + li, L - 1);
            const float* kp = KEYS + (size_t)(t0 + key) * 256 + 4 * lg;
#pragma unroll
            for (int q = 0; q < 16; ++q) xk[w][q] = *reinterpret_cast<const g4v*>(kp + 16 * q);
            if (POSTAB && sal) {
                g4v d4 = g4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 16; ++q) d4 += xk[w][q] * *reinterpret_cast<const g4v*>(sal_w + 16 * q + 4 * lg);
                float d = (d4[0] + d4[1]) + (d4[2] + d4[3]);
                d += __shfl_xor(d, 16, 64);
                d += __shfl_xor(d, 32, 64);
                const int p = kt * 16 + li;
                if (lg == 0 && p < lv && p < sal_ld) sal[(size_t)b * sal_ld + p] = d + sal_b[0];
            }
            if (POSTAB && key < lv) {
                const float* pp = prow + (size_t)key * 256 + 4 * lg;
#pragma unroll
                for (int q = 0; q < 16; ++q) xk[w][q] += *reinterpret_cast<const g4v*>(pp + 16 * q);
            }
        }
    }

    if (QLDS && !QKS) {     // (Pt is free until stage B)
#pragma unroll
        for (int j = 0; j < QREG; ++j) {
            const int i = tid + 512 * j;
            if (i < NQ * 64) reinterpret_cast<g4v*>(Pt)[i] = qreg[j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (an LDS-only barrier: __syncthreads would wait out the key rows)
        __builtin_amdgcn_s_barrier();
    }
    if (QKS) {      // precomputed slabs: 48 KiB from the L2 instead of 256 KiB of W_k
        for (int i = tid; i < C::QK_FLOATS / 4; i += 512)
            reinterpret_cast<g4v*>(qkf)[i] = reinterpret_cast<const g4v*>(QKS)[i];
    } else {
    // rows NP .. NPP - 1 of the last pair tile's slabs = 0 (an odd slot count fills half of its last tile)
    if (C::NP < C::NPP) {
        for (int i = tid; i < 16 * 8 * 16; i += 512) {      // (q, row 8 .. 15, 16 floats) of the last pair tile
            const int q = i >> 7, r = 8 + ((i >> 4) & 7), c = i & 15;
            qkf[((C::NPT - 1) * 16 + q) * 256 + r * 16 + c] = 0.f;
        }
    }
    // ---- stage 0: qk[(s, h)][c] = sqrt(1/32) sum_d q[s][32 h + d] * Wk[32 h + d][c]: wave = head h, lane = channels 4 lane ..
    // + 3 -- a W_k row is ONE coalesced 1-KiB request per wave (a thread per channel made it four 256-B ones: 128 scalar loads
    // per thread, the L2 request rate of 20 000 windows x 256 KiB).  The window's query values are wave-uniform: they come
    // through the scalar cache straight into the FMAs' scalar operands.  Rows in groups of four, the next group requested
    // ahead of this group's FMAs.  Stored in slab (pair tile, c / 16), row pair % 16, physical chunk ((c / 4) % 4) ^ swz(row)
    {
        const float* __restrict__ qb = DQ + (size_t)b * NQ * 256 + wave * 32;
        const float* wb = Wk + (size_t)wave * 32 * 256;
        const unsigned l4 = 4u * lane;
        g4v a[NQ];
#pragma unroll
        for (int s = 0; s < NQ; ++s) a[s] = g4v{0.f, 0.f, 0.f, 0.f};
        // (rows two at a time, the next pair requested ahead of this pair's FMAs: the key rows of stage A are live -- 64 of
        // the 128 VGPRs two workgroups per CU leave a wave)
        g4v wn[2], wc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) wn[i] = *reinterpret_cast<const g4v*>(wb + i * 256 + l4);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
#pragma unroll
            for (int i = 0; i < 2; ++i) wc[i] = wn[i];
            if (g < 15) {
#pragma unroll
                for (int i = 0; i < 2; ++i) wn[i] = *reinterpret_cast<const g4v*>(wb + ((g + 1) * 2 + i) * 256 + l4);
            }
            if (QLDS) {
                const float* ql = Pt + wave * 32 + g * 2;       // this head's two query values of every slot: LDS broadcasts
#pragma unroll
                for (int s = 0; s < NQ; ++s) {
                    const g2v q2 = *reinterpret_cast<const g2v*>(ql + s * 256);
                    a[s] += wc[0] * q2[0];
                    a[s] += wc[1] * q2[1];
                }
            } else {
#pragma unroll
                for (int s = 0; s < NQ; ++s)
#pragma unroll
                    for (int i = 0; i < 2; ++i) a[s] += wc[i] * qb[s * 256 + g * 2 + i];
            }
        }
        const int c = 4 * lane;
#pragma unroll
        for (int s = 0; s < NQ; ++s) {
            const int p = s * 8 + wave, row = p & 15;
            *reinterpret_cast<g4v*>(qkf + ((p >> 4) * 16 + (c >> 4)) * 256 + row * 16 + ((((c >> 2) & 3) ^ dcm_swz16(row)) << 2)) =
                a[s] * 0.17677669529663687f;
        }
    }
    }
    __syncthreads();                                        // qk slabs complete; qs dead
    if (QKS_OUT) {  // slab-building launch (one workgroup): publish and stop
        for (int i = tid; i < C::QK_FLOATS / 4; i += 512)
            reinterpret_cast<g4v*>(QKS_OUT)[i] = reinterpret_cast<const g4v*>(qkf)[i];
        return;
    }

    // ---- stage A: scores of this wave's key tile(s) against the 48 pairs
    const int rd = li * 16 + ((lg ^ dcm_swz16(li)) << 2);   // this lane's 16-B chunk inside a slab (row = pair li)
    g4v sc[KTW][C::NPT];
#pragma unroll
    for (int w = 0; w < KTW; ++w) {
        const int kt = wave + 8 * w;
#pragma unroll
        for (int pt = 0; pt < C::NPT; ++pt) sc[w][pt] = g4v{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (kt < nkt) {
#pragma unroll
            for (int pt = 0; pt < C::NPT; ++pt) {
                g4v ch[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) ch[r] = g4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const g4v w4 = *reinterpret_cast<const g4v*>(qkf + (pt * 16 + q) * 256 + rd);
#pragma unroll
                    for (int r = 0; r < 4; ++r) ch[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(xk[w][q][r], w4[r], ch[r], 0, 0, 0);
                }
                g4v s4 = (ch[0] + ch[1]) + (ch[2] + ch[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * 16 + 4 * lg + r >= L) s4[r] = -INFINITY;       // key 4 lg + r of the tile is padding
                sc[w][pt] = s4;
            }
        }
    }

    // ---- stage B: softmax over the keys of each pair (lane li of pair tile pt): registers, lane groups, key tiles
#pragma unroll
    for (int pt = 0; pt < C::NPT; ++pt) {
        float m = -INFINITY;
#pragma unroll
        for (int w = 0; w < KTW; ++w)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, sc[w][pt][r]);
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (lg == 0) smax[wave * C::NPP + pt * 16 + li] = m;
    }
    __syncthreads();
    float inv[C::NPT];
#pragma unroll
    for (int pt = 0; pt < C::NPT; ++pt) {
        float m = smax[pt * 16 + li];
#pragma unroll
        for (int w = 1; w < 8; ++w) m = fmaxf(m, smax[w * C::NPP + pt * 16 + li]);
        const float m2 = m * 1.4426950408889634f;
        float l = 0.f;
#pragma unroll
        for (int w = 0; w < KTW; ++w)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(sc[w][pt][r], 1.4426950408889634f, -m2));   // exp(-inf) = 0
                sc[w][pt][r] = e;
                l += e;
            }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (lg == 0) ssum[wave * C::NPP + pt * 16 + li] = l;
    }
    __syncthreads();
#pragma unroll
    for (int pt = 0; pt < C::NPT; ++pt) {
        float l = ssum[pt * 16 + li];
#pragma unroll
        for (int w = 1; w < 8; ++w) l += ssum[w * C::NPP + pt * 16 + li];
        inv[pt] = 1.0f / l;
    }
#pragma unroll
    for (int w = 0; w < KTW; ++w) {
        const int kt = wave + 8 * w;
        if (kt < nkt) {
#pragma unroll
            for (int pt = 0; pt < C::NPT; ++pt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Pt[(kt * 16 + 4 * lg + r) * C::NPP + pt * 16 + li] = sc[w][pt][r] * inv[pt];
        }
    }
    __syncthreads();                                        // Pt complete (rows [L, 16 nkt) are zeros)

    // ---- stage C: ctx[p][c] = sum_j P[p][j] * mem[j][c]; wave = channels [32 w, 32 w + 32) x all pairs
    g4v acc[2][C::NPT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pt = 0; pt < C::NPT; ++pt) acc[t][pt] = g4v{0.f, 0.f, 0.f, 0.f};
    {
        const float* xb = X + (size_t)t0 * 256 + 32 * wave + li;
        const int nks = nkt * 4;
#pragma unroll 4
        for (int s = 0; s < nks; ++s) {
            const int key = min(4 * s + lg, L - 1);         // padding keys carry P = 0: any finite row will do
            const float a0 = xb[(size_t)key * 256], a1 = xb[(size_t)key * 256 + 16];
            const float* pr = Pt + (4 * s + lg) * C::NPP + li;
#pragma unroll
            for (int pt = 0; pt < C::NPT; ++pt) {
                const float pb = pr[pt * 16];
                acc[0][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, pb, acc[0][pt], 0, 0, 0);
                acc[1][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, pb, acc[1][pt], 0, 0, 0);
            }
        }
    }
    float* ctxs = qkf;                                      // the qk slabs are dead since stage A
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pt = 0; pt < C::NPT; ++pt) {
            const int p = pt * 16 + li;
            if (p < C::NP) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ctxs[p * C::CTX_LD + 32 * wave + 16 * t + 4 * lg + r] = acc[t][pt][r];
            }
        }
    __syncthreads();

    // ---- stage D: out[s][o] = sum_c WvT[c][o] * ctx[(s, o / 32)][c] + bv[o]: wave = the slice c in [32 w, 32 w + 32) of the
    // contraction, lane = outputs 4 lane .. + 3 (one head: a W_v^T row is one coalesced 1-KiB request per wave); the eight
    // partial sums meet in LDS
    {
        g4v o[NQ];
#pragma unroll
        for (int s = 0; s < NQ; ++s) o[s] = g4v{0.f, 0.f, 0.f, 0.f};
        const float* wb = WvT + (size_t)wave * 32 * 256;
        const unsigned l4 = 4u * lane;
        const float* crow = ctxs + (lane >> 3) * C::CTX_LD + 32 * wave;
        g4v wn[4], wc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wn[i] = *reinterpret_cast<const g4v*>(wb + i * 256 + l4);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wc[i] = wn[i];
            if (g < 7) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wn[i] = *reinterpret_cast<const g4v*>(wb + ((g + 1) * 4 + i) * 256 + l4);
            }
#pragma unroll
            for (int s = 0; s < NQ; ++s) {
                const g4v cx = *reinterpret_cast<const g4v*>(crow + s * 8 * C::CTX_LD + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i) o[s] += wc[i] * cx[i];
            }
        }
        __syncthreads();                                    // every wave is done reading ctx
        float* part = qkf;                                  // [8 waves][NQ][256] over the ctx rows
#pragma unroll
        for (int s = 0; s < NQ; ++s) *reinterpret_cast<g4v*>(part + (wave * NQ + s) * 256 + 4 * lane) = o[s];
        __syncthreads();
        for (int i = tid; i < NQ * 256; i += 512) {
            float v = part[i];
#pragma unroll
            for (int w = 1; w < 8; ++w) v += part[w * NQ * 256 + i];
            OUT[(size_t)b * NQ * 256 + i] = v + bv[i & 255];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dec_cross_x_kernel: the same computation with the window's memory rows read ONCE, at two workgroups per CU.
//
// Stage A holds the window in registers (lane = key: 16 keys x 256 channels per wave); stage C needs it by channel (lane =
// channel).  The two-read kernel above fetches the rows a second time from global memory for that -- from beyond the L2, in
// 64-byte pieces, latency-bound -- and the LDS-resident form below pays for its 110-KiB image with one workgroup per CU.
// Here the registers ARE the image and are transposed through LDS a channel QUARTER at a time: each wave parks the 64
// channels of its 16 keys (four ds_write_b128 per lane) in a [128 keys][64 + 16] tile whose 16-B chunks are XOR-swizzled by
// the key (2-way conflicts on the writes, none on stage C's ds_read_b32 of lanes (key 4 s + lg, channel 16 ct + li)), the
// eight waves multiply it as (channel tile ct = w % 4) x (key half w / 4) against all pair tiles, and the next quarter
// follows: 36 KiB of LDS beside Pt, 69 KiB per workgroup in all -- two per CU as before.  For the position term the rows must
// stay RAW in the registers (stage C multiplies memory, not memory + pos), so stage A walks the channels in the OUTER loop and
// forms memory + pos per 16-channel step in a temporary (the position row of that step is requested a step ahead); two
// partial chains per pair tile keep an accumulator two MFMAs apart.  The key halves of a channel tile meet in LDS (the
// second half adds to the first: two addends, order-independent).
constexpr int DCR_QK = 32 * 256 + 16 * 128;         // compact slabs: pair tiles 0, 1 [16 q][16 rows][16], tile 2 [16 q][8 rows][16]
constexpr int DCX_QS = 80;                          // row stride (floats) of the quarter tile: = 16 mod 32
constexpr int DCX_REGA = 128 * 48 + 128 * DCX_QS;   // Pt [128][48] + the quarter tile [128][80] (>= slabs, ctx, partials)
constexpr int DCX_LDS_FLOATS = DCX_REGA + 2 * 8 * 48;
static_assert(DCX_REGA >= DCR_QK && DCX_REGA >= 40 * 260 && DCX_REGA >= 8 * 5 * 256, "region A holds each of its tenants");

template <bool POSTAB>
__global__ __launch_bounds__(512, 4) void dec_cross_x_kernel(const float* __restrict__ DQ, const float* __restrict__ XP,
                                                             const float* __restrict__ X, const float* __restrict__ pos_rows,
                                                             const int* __restrict__ vlen, const int* __restrict__ off,
                                                             const float* __restrict__ Wk, const float* __restrict__ WvT,
                                                             const float* __restrict__ bv, float* __restrict__ OUT,
                                                             const float* __restrict__ QKS, float* __restrict__ QKS_OUT,
                                                             const float* __restrict__ sal_w, const float* __restrict__ sal_b,
                                                             float* __restrict__ sal, int sal_ld) {
    constexpr int NQ = 5, NPT = 3, NPP = 48, NP = 40, CTX_LD = 260;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* regA = smem;                         // slabs (compact, 40 KiB) -> Pt + quarter tile -> ctx -> stage D partials
    float* Pt = regA;                           // [128][48]
    float* Qt = regA + 128 * NPP;               // [128][DCX_QS]
    float* smax = regA + DCX_REGA;              // [8][48]
    float* ssum = smax + 8 * NPP;               // [8][48]
    const int b = blockIdx.x;
    const int t0 = off[b];
    const int L = min(off[b + 1] - t0, 128);
    const int nkt = (L + 15) >> 4;
    const int lv = POSTAB ? vlen[b] : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int rdA = li * 16 + ((lg ^ dcm_swz16(li)) << 2);
    const int li8 = li & 7;
    const int rdB = li8 * 16 + ((lg ^ dcm_swz16(li8)) << 2);
    auto slab = [](int pt, int q) { return pt < 2 ? (pt * 16 + q) * 256 : 32 * 256 + q * 128; };

    // ---- this wave's key rows, RAW (POSTAB: the memory rows; else memory + pos as given), requested first
    // (the first 128 channels ahead of stage 0, the other 128 behind it: stage A walks the channels in order, so the second
    // half has eight steps of MFMAs to land under, and stage 0 keeps the registers for a four-row look-ahead on W_k)
    g4v xk[16];
    const int my_key = min(wave * 16 + li, L - 1);
    const bool have = wave < nkt && !QKS_OUT;
    const float* kp = (POSTAB ? X : XP) + (size_t)(t0 + my_key) * 256 + 4 * lg;
    if (have) {
#pragma unroll
        for (int q = 0; q < 8; ++q) xk[q] = *reinterpret_cast<const g4v*>(kp + 16 * q);
    }

    // ---- stage 0: the folded-query slabs (compact layout)
    if (QKS) {
        for (int i = tid; i < DCR_QK / 4; i += 512) reinterpret_cast<g4v*>(regA)[i] = reinterpret_cast<const g4v*>(QKS)[i];
    } else {
        const float* __restrict__ qb = DQ + (size_t)b * NQ * 256 + wave * 32;
        const float* wb = Wk + (size_t)wave * 32 * 256;
        const unsigned l4 = 4u * lane;
        g4v a[NQ];
#pragma unroll
        for (int s = 0; s < NQ; ++s) a[s] = g4v{0.f, 0.f, 0.f, 0.f};
        g4v wn[4], wc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wn[i] = *reinterpret_cast<const g4v*>(wb + i * 256 + l4);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wc[i] = wn[i];
            if (g < 7) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wn[i] = *reinterpret_cast<const g4v*>(wb + ((g + 1) * 4 + i) * 256 + l4);
            }
            // (a fence per step: left alone, the scheduler either hoists W_k rows until the key rows spill, or -- with the loop
            // rolled -- sinks each step's loads to their use and every step waits out an L2 round trip)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int s = 0; s < NQ; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) a[s] += wc[i] * qb[s * 256 + g * 4 + i];
        }
        const int c = 4 * lane;
#pragma unroll
        for (int s = 0; s < NQ; ++s) {
            const int p = s * 8 + wave, row = p & 15;
            *reinterpret_cast<g4v*>(regA + slab(p >> 4, c >> 4) + row * 16 + ((((c >> 2) & 3) ^ dcm_swz16(row)) << 2)) =
                a[s] * 0.17677669529663687f;
        }
    }
    if (have) {                                             // the key rows' second half (see above)
        const float* kp2 = kp;
        asm volatile("" : "+v"(kp2));                       // not hoisted over stage 0 (the registers are W_k's there)
        const float __attribute__((address_space(1)))* kg = (const float __attribute__((address_space(1)))*)kp2;
#pragma unroll
        for (int q = 8; q < 16; ++q) xk[q] = *reinterpret_cast<const g4v __attribute__((address_space(1)))*>(kg + 16 * q);
    }
    __syncthreads();                                        // slabs complete
    if (QKS_OUT) {  // slab-building launch (one workgroup): publish and stop
        for (int i = tid; i < DCR_QK / 4; i += 512) reinterpret_cast<g4v*>(QKS_OUT)[i] = reinterpret_cast<const g4v*>(regA)[i];
        return;
    }

    // ---- stage A: scores of this wave's key tile against the 48 pairs, channels in the outer loop
    g4v sc[NPT];
    if (wave >= nkt) {
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) sc[pt] = g4v{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    } else {
        g4v ch[NPT][2];
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) { ch[pt][0] = g4v{0.f, 0.f, 0.f, 0.f}; ch[pt][1] = ch[pt][0]; }
        const int pk = max(min(my_key, lv - 1), 0);         // a text key re-reads a clip's position row and is masked out
        const float* pp = POSTAB ? pos_rows + ((size_t)(lv * (lv - 1) / 2) + pk) * 256 + 4 * lg : nullptr;
        const float pmask = my_key < lv ? 1.f : 0.f;
        g4v pn = g4v{0.f, 0.f, 0.f, 0.f};
        if (POSTAB) pn = *reinterpret_cast<const g4v*>(pp);
        g4v d4 = g4v{0.f, 0.f, 0.f, 0.f};                  // the saliency head rides along (see dec_cross_mfma_kernel)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            g4v t = xk[q];
            if (POSTAB && sal) d4 += t * *reinterpret_cast<const g4v*>(sal_w + 16 * q + 4 * lg);
            if (POSTAB) {
                t += pn * pmask;
                if (q < 15) pn = *reinterpret_cast<const g4v*>(pp + 16 * (q + 1));
            }
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                const g4v w4 = *reinterpret_cast<const g4v*>(regA + slab(pt, q) + (pt < 2 ? rdA : rdB));
#pragma unroll
                for (int r = 0; r < 4; ++r) ch[pt][r & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(t[r], w4[r], ch[pt][r & 1], 0, 0, 0);
            }
        }
        if (POSTAB && sal) {
            float d = (d4[0] + d4[1]) + (d4[2] + d4[3]);
            d += __shfl_xor(d, 16, 64);
            d += __shfl_xor(d, 32, 64);
            const int pc = wave * 16 + li;
            if (lg == 0 && pc < lv && pc < sal_ld) sal[(size_t)b * sal_ld + pc] = d + sal_b[0];
        }
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            g4v s4 = ch[pt][0] + ch[pt][1];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (wave * 16 + 4 * lg + r >= L) s4[r] = -INFINITY;       // key 4 lg + r of the tile is padding
            sc[pt] = s4;
        }
    }

    // ---- stage B: softmax over the keys of each pair
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        float m = fmaxf(fmaxf(sc[pt][0], sc[pt][1]), fmaxf(sc[pt][2], sc[pt][3]));
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (lg == 0) smax[wave * NPP + pt * 16 + li] = m;
    }
    __syncthreads();                                        // every wave is past stage A: the slabs are dead
    float inv[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        float m = smax[pt * 16 + li];
#pragma unroll
        for (int w = 1; w < 8; ++w) m = fmaxf(m, smax[w * NPP + pt * 16 + li]);
        const float m2 = m * 1.4426950408889634f;
        float l = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(sc[pt][r], 1.4426950408889634f, -m2));   // exp(-inf) = 0
            sc[pt][r] = e;
            l += e;
        }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (lg == 0) ssum[wave * NPP + pt * 16 + li] = l;
    }
    __syncthreads();
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        float l = ssum[pt * 16 + li];
#pragma unroll
        for (int w = 1; w < 8; ++w) l += ssum[w * NPP + pt * 16 + li];
        inv[pt] = 1.0f / l;
    }
    if (wave < nkt) {
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Pt[(wave * 16 + 4 * lg + r) * NPP + pt * 16 + li] = sc[pt][r] * inv[pt];
    }

    // ---- stage C, a channel quarter at a time: ctx[p][c] = sum_j P[p][j] * mem[j][c]
    // wave = (channel tile ct of the quarter, key half kh); quarter-tile chunk swizzle f(key) = (key / 2) % 4
    const int ct = wave & 3, kh = wave >> 2;
    const int krow = wave * 16 + li;                        // this lane's key row of the tile it parks
    const int fw = (krow >> 1) & 3;
    g4v acc[4][NPT];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        if (wave < nkt) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<g4v*>(Qt + krow * DCX_QS + (((4 * j + lg) ^ fw) << 2)) = xk[4 * qq + j];
        }
        __syncthreads();                                    // the quarter (and, the first time, Pt) is complete
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) acc[qq][pt] = g4v{0.f, 0.f, 0.f, 0.f};
        const int s0 = kh * 2 * nkt, s1 = s0 + 2 * nkt;     // this half's k-steps (4 keys each) of the 4 nkt
#pragma unroll 2
        for (int s = s0; s < s1; ++s) {
            const int key = 4 * s + lg;                     // rows [L, 16 nkt) are finite copies of the last row, their P is 0
            const float a0 = Qt[key * DCX_QS + (((4 * ct + (li >> 2)) ^ ((key >> 1) & 3)) << 2) + (li & 3)];
            const float* pr = Pt + key * NPP + li;
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
                acc[qq][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, pr[pt * 16], acc[qq][pt], 0, 0, 0);
        }
        __syncthreads();                                    // the quarter tile is free again
    }
    // the two key halves of a channel tile meet in the ctx rows: the first half stores, the second adds
    float* ctxs = regA;                                     // [40][260] (Pt and the tile are dead: barrier above)
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
        if (kh == hlf) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) {
                    const int p = pt * 16 + li;
                    if (p < NP) {
                        float* cp = ctxs + p * CTX_LD + 64 * qq + 16 * ct + 4 * lg;
                        g4v v = acc[qq][pt];
                        if (hlf) v += *reinterpret_cast<const g4v*>(cp);
                        *reinterpret_cast<g4v*>(cp) = v;
                    }
                }
        }
        __syncthreads();
    }

    // ---- stage D: out[s][o] = sum_c WvT[c][o] * ctx[(s, o / 32)][c] + bv[o] (as in dec_cross_mfma_kernel)
    {
        g4v o[NQ];
#pragma unroll
        for (int s = 0; s < NQ; ++s) o[s] = g4v{0.f, 0.f, 0.f, 0.f};
        const float* wb = WvT + (size_t)wave * 32 * 256;
        const unsigned l4 = 4u * lane;
        const float* crow = ctxs + (lane >> 3) * CTX_LD + 32 * wave;
        g4v wn[4], wc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wn[i] = *reinterpret_cast<const g4v*>(wb + i * 256 + l4);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wc[i] = wn[i];
            if (g < 7) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wn[i] = *reinterpret_cast<const g4v*>(wb + ((g + 1) * 4 + i) * 256 + l4);
            }
#pragma unroll
            for (int s = 0; s < NQ; ++s) {
                const g4v cx = *reinterpret_cast<const g4v*>(crow + s * 8 * CTX_LD + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i) o[s] += wc[i] * cx[i];
            }
        }
        __syncthreads();                                    // every wave is done reading ctx
        float* part = regA;                                 // [8 waves][5][256]
#pragma unroll
        for (int s = 0; s < NQ; ++s) *reinterpret_cast<g4v*>(part + (wave * NQ + s) * 256 + 4 * lane) = o[s];
        __syncthreads();
        for (int i = tid; i < NQ * 256; i += 512) {
            float v = part[i];
#pragma unroll
            for (int w = 1; w < 8; ++w) v += part[w * NQ * 256 + i];
            OUT[(size_t)b * NQ * 256 + i] = v + bv[i & 255];
        }
    }
}

template <bool POSTAB>
static int launch_x_one(const float* DQ, const float* XP, const float* X, const float* pos_rows, const int* vlen, const int* off,
                        const float* Wk, const float* WvT, const float* bv, float* OUT, int B, float* qk_slabs, hipStream_t s,
                        const float* sal_w, const float* sal_b, float* sal, int sal_ld) {
    static DeviceOnce once;     // the opt-in to > 64 KiB of LDS: once per device
    CONE_CHECK_HIP(device_once(once, [] {
        return hipFuncSetAttribute((const void*)dec_cross_x_kernel<POSTAB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   DCX_LDS_FLOATS * 4);
    }));
    if (qk_slabs) {     // window-independent queries: the (compact) operand slabs once, by one workgroup
        hipLaunchKernelGGL((dec_cross_x_kernel<POSTAB>), dim3(1), dim3(512), DCX_LDS_FLOATS * 4, s, DQ, XP, X, pos_rows, vlen, off,
                           Wk, WvT, bv, OUT, (const float*)nullptr, qk_slabs, (const float*)nullptr, (const float*)nullptr,
                           (float*)nullptr, 0);
        CONE_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((dec_cross_x_kernel<POSTAB>), dim3(B), dim3(512), DCX_LDS_FLOATS * 4, s, DQ, XP, X, pos_rows, vlen, off, Wk,
                       WvT, bv, OUT, (const float*)qk_slabs, (float*)nullptr, sal_w, sal_b, sal, sal_ld);
    CONE_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// dec_cross_res_kernel: the same computation with the window's memory rows read from HBM ONCE.
//
// The kernel above reads every memory row twice from beyond the L2 -- by key rows for the scores (stage A: lane = key), by
// channel columns for ctx = P . mem (stage C: lane = channel) -- 1.94 x the rows' bytes per launch (profiles/r03_pmc_*): the
// two contractions need the rows in TRANSPOSED register layouts, and two workgroups per CU x 32 CUs x 104 KB of rows do
// not stay in a 4 MB L2 between the stages.  Here the rows are LDS-resident for stage C: ONE persistent 8-wave workgroup
// per CU; per window the rows come in twice from the L2's point of view but once from HBM's, back to back -- as stage A's
// register operand (lane = key, as before) and, by LDS-DMA (global_load_lds_dwordx4: no VGPRs, one 1-KiB row per
// instruction, the 64-B block of a row swapped by the row's parity so that stage C's ds_read_b32 of lanes (key 4 s + lg,
// channel li) are conflict-free), into a [110][256] image that stage C reads by channel.  160 KiB of LDS hold that image
// (110 KiB) + ONE 46-KiB region that is, in turn, the folded-query slabs (stage A's B operand; the third pair tile keeps
// only its 8 real rows: 40 KiB), the probabilities Pt, the ctx rows, stage D's partial sums; every hand-over is a barrier
// that the stage boundaries need anyway.  Windows of up to 110 tokens (Ego4D: 90 clips + <= 20 text tokens); longer ones
// take the kernel above.
//
// STATUS (round 4, tools/dec_cross_bench.py, 20 000 windows): 2.39 / 1.95 ms per launch (per-window / shared queries) against
// 1.71 / 1.50 ms of the two-read kernel above -- it reads the rows once and is SLOWER, so it is opt-in ("dec_fold" 4) and
// the two-read kernel stays the default.  Why: one workgroup per CU (the image leaves no LDS for a second one) exposes
// every latency that two interleaved workgroups hide from each other -- 25 us per window where the matrix pipe needs 10:
// what is left after the hand pipelining below are the L2 round trips of the fold operands (2 - 2.5 us each under load) and
// the next window's key rows, which 256 VGPRs cannot hold together with W_v through stage C (26 spills: measured).
//
// With one workgroup per CU nothing else hides a stage's load latency (first cut, stages as above: 36 us per window, 10 of
// them in stage D's eight dependent groups of W_v loads), so the loop is software-pipelined by hand; every vector-memory
// operand of a stage is requested one or two stages ahead, into registers that are dead in between:
//   * W_v (stage D) right after stage A -- into the VGPRs that held the key rows -- landing under the softmax and stage C
//     (LDS + MFMA only);
//   * the first half of W_k and the next window's key rows at the end of stage D, the second half of W_k when the first has
//     been consumed (all of it ahead would need 128 + 64 + 20 live registers at the top of the loop: spills), the position
//     rows eight float4 at a time behind it; the row image when stage 0 has consumed its last load (vmcnt retires in order: a DMA issued any
//     earlier would hold up every ordinary load behind it) -- it lands under stage A.
// The two folds run on float4 columns (thread = 4 adjacent channels, wave = head resp. a 32-channel slice of the
// contraction): 32 coalesced 1-KiB row reads per wave instead of 128 scalar ones -- at most 63 vector-memory operations are
// outstanding per wave (vmcnt is 6 bits), the whole schedule stays below that.
constexpr int DCR_RMAX = 110;                       // rows of the LDS image
constexpr int DCR_ROWS = DCR_RMAX * 256;
constexpr int DCR_CTX_LD = 260;
constexpr int DCR_REGB = 40 * DCR_CTX_LD + 5 * 256 + 96;   // >= the slabs, Pt [128][48], ctx [40][260], stage D partials [8][5][256]
constexpr int DCR_LDS_FLOATS = DCR_ROWS + DCR_REGB + 2 * 8 * 48;
static_assert(DCR_REGB >= DCR_QK && DCR_REGB >= 128 * 48 && DCR_REGB >= 8 * 5 * 256, "region B holds each of its tenants");
static_assert(DCR_LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");

// Every barrier of the window loop orders LDS traffic only; __syncthreads() would also drain the vector-memory queue
// (vmcnt(0)) -- i.e. wait out the row DMA and every prefetch at every stage boundary.
#define DCR_BARRIER()                                          \
    {                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                          \
        asm volatile("" ::: "memory");                         \
    }
#define DCR_PIN() asm volatile("" ::: "memory")     // loads stay on their side of this point
// a row load as `uniform base (SGPR pair, made opaque right here) + 32-bit lane offset`: without the pin the compiler folds the
// lane offset into the base, hoists the per-lane 64-bit pointers of every 4-KiB step out of the window loop and spills them
__device__ __forceinline__ g4v dcr_ldg(const float* ubase, unsigned voff) {
    asm volatile("" : "+s"(ubase));
    // (through the asm the pointer has lost its address space: say "global" again, or the load is a flat_load)
    typedef const __attribute__((address_space(1))) g4v* gptr;
    return *(gptr)((const __attribute__((address_space(1))) float*)ubase + voff);
}
#define DCR_GLDS16(src, dst) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

template <bool POSTAB, bool SHARED>
__global__ __launch_bounds__(512, 1) void dec_cross_res_kernel(const float* __restrict__ DQ, const float* __restrict__ XP,
                                                               const float* __restrict__ X,
                                                               const float* __restrict__ pos_rows,
                                                               const int* __restrict__ vlen, const int* __restrict__ off,
                                                               const float* __restrict__ Wk, const float* __restrict__ WvT,
                                                               const float* __restrict__ bv, float* __restrict__ OUT,
                                                               const float* __restrict__ QKS, float* __restrict__ QKS_OUT,
                                                               int B) {
    constexpr int NQ = 5, NPT = 3, NPP = 48, NP = 40;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* rows = smem;                         // [110][256], 64-B block c / 16 swapped with its neighbour on odd rows
    float* regB = smem + DCR_ROWS;              // slabs -> Pt [128][48] -> ctx [40][260] -> stage D partials [8][5][256]
    float* smax = regB + DCR_REGB;              // [8][48]
    float* ssum = smax + 8 * NPP;               // [8][48]
    const float* __restrict__ KEYS = POSTAB ? X : XP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    // this lane's 16-B chunk inside a slab: row = pair li (tiles 0, 1); tile 2 holds rows 0 .. 7 only -- lanes li >= 8 re-read
    // pairs 32 .. 39 (their score columns are padding: finite, never used)
    const int rdA = li * 16 + ((lg ^ dcm_swz16(li)) << 2);
    const int li8 = li & 7;
    const int rdB = li8 * 16 + ((lg ^ dcm_swz16(li8)) << 2);
    auto slab = [](int pt, int q) { return pt < 2 ? (pt * 16 + q) * 256 : 32 * 256 + q * 128; };

    // ---- the two folds on float4 columns.  Stage 0: wave = head h, lane = channels 4 lane .. + 3:
    //   qk[(s, h)][c] = sqrt(1/32) sum_d q[s][32 h + d] Wk[32 h + d][c]       (q through the scalar cache: wave-uniform)
    // wk[d] = Wk[32 h + d][4 lane ..]: window-independent, requested ahead in two halves.
    // (every global address below is a wave-uniform base -- SGPRs -- plus ONE 32-bit lane offset: per-lane 64-bit pointers for
    // 32 + 32 + 16 + 16 rows would be hoisted out of the window loop and spilled)
    g4v wk[SHARED ? 1 : 32];
    const float* wkb = Wk + (size_t)wave * 32 * 256;            // uniform
    const unsigned l4 = 4u * lane;
    auto load_wk = [&](int d0) {
        if (!SHARED) {
#pragma unroll
            for (int d = d0; d < d0 + 16; ++d) wk[d] = dcr_ldg(wkb + d * 256, l4);
        }
    };
    g4v fa[SHARED ? 1 : NQ];
    auto fold_half = [&](int bw, int d0) {      // accumulate d0 .. d0 + 15 of window bw's queries into fa
        if (!SHARED) {
            const float* __restrict__ qb = DQ + (size_t)bw * NQ * 256 + wave * 32;
#pragma unroll
            for (int d = d0; d < d0 + 16; ++d) {
#pragma unroll
                for (int s = 0; s < NQ; ++s) fa[s] += wk[d] * qb[s * 256 + d];
            }
        }
    };
    auto fold_store = [&]() {
        if (!SHARED) {
            const int c = 4 * lane;
#pragma unroll
            for (int s = 0; s < NQ; ++s) {
                const int p = s * 8 + wave, row = p & 15;
                *reinterpret_cast<g4v*>(regB + slab(p >> 4, c >> 4) + row * 16 + ((((c >> 2) & 3) ^ dcm_swz16(row)) << 2)) =
                    fa[s] * 0.17677669529663687f;
            }
        }
    };
    if (QKS_OUT) {      // slab-building launch (one workgroup, window 0's queries; instantiated with SHARED = false)
        load_wk(0); load_wk(16);
#pragma unroll
        for (int s = 0; s < (SHARED ? 1 : NQ); ++s) fa[s] = g4v{0.f, 0.f, 0.f, 0.f};
        fold_half(0, 0); fold_half(0, 16);
        fold_store();
        __syncthreads();
        for (int i = tid; i < DCR_QK / 4; i += 512) reinterpret_cast<g4v*>(QKS_OUT)[i] = reinterpret_cast<const g4v*>(regB)[i];
        return;
    }
    int b = blockIdx.x;
    if (b >= B) return;

    // this wave's key tile of window `bw` -> registers (A operand of stage A: lane = key, float4 = channels 16 q + 4 lg ..);
    // the position rows ride in their own registers (requested later: L2-resident table) and are added ahead of stage A
    g4v xk[16];
    float pmask = 0.f;
    auto load_keys = [&](int bw) {
        const int t0 = off[bw];
        const int L = min(off[bw + 1] - t0, DCR_RMAX);
        const int key = min(wave * 16 + li, L - 1);
        const float* kb = KEYS + (size_t)t0 * 256;              // uniform
        const unsigned ko = (unsigned)key * 256u + 4u * lg;
#pragma unroll
        for (int q = 0; q < 16; ++q) xk[q] = dcr_ldg(kb + 16 * q, ko);
    };
    // xk += position rows of window bw (clip keys; a text key re-reads a clip's row and is masked out), eight float4 at a time
    auto add_pos = [&](int bw) {
        if (POSTAB) {
            const int t0 = off[bw];
            const int L = min(off[bw + 1] - t0, DCR_RMAX);
            const int key = min(wave * 16 + li, L - 1);
            const int lv = vlen[bw];
            const int pk = max(min(key, lv - 1), 0);
            const float* pb = pos_rows + (size_t)(lv * (lv - 1) / 2) * 256;        // uniform
            const unsigned po = (unsigned)pk * 256u + 4u * lg;
            pmask = key < lv ? 1.f : 0.f;
            g4v xa[8], xb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) xa[q] = dcr_ldg(pb + 16 * q, po);
            asm volatile("" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(xa[4]), "+v"(xa[5]), "+v"(xa[6]), "+v"(xa[7]));
#pragma unroll
            for (int q = 0; q < 8; ++q) xk[q] += xa[q] * pmask;
#pragma unroll
            for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(xk[q]));
            DCR_PIN();
#pragma unroll
            for (int q = 0; q < 8; ++q) xb[q] = dcr_ldg(pb + 16 * (q + 8), po);
            // all eight are requested before the first is used (near the register limit the scheduler otherwise walks them
            // one by one: load, wait, fma, load ...)
            asm volatile("" : "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3]), "+v"(xb[4]), "+v"(xb[5]), "+v"(xb[6]), "+v"(xb[7]));
#pragma unroll
            for (int q = 0; q < 8; ++q) xk[q + 8] += xb[q] * pmask;
        }
    };
    // stage D's operand: wv[j] = WvT[32 wave + j][4 lane ..] (window-independent), requested right after stage A
    g4v wv[32];
    const float* wvb = WvT + (size_t)wave * 32 * 256;           // uniform
    g4v qs[SHARED ? DCR_QK / 4 / 512 : 1];         // SHARED: the window-independent slabs, 5 float4 per thread
    auto load_slabs = [&]() {
        if (SHARED) {
#pragma unroll
            for (int i = 0; i < DCR_QK / 4 / 512; ++i) qs[i] = dcr_ldg(QKS + 2048 * i, 4u * tid);
        }
    };
    // prologue: the first window's operands (exposed once per workgroup)
    load_wk(0);
    load_slabs();
    load_keys(b);

    for (; b < B; b += (int)gridDim.x) {
        const int t0 = off[b];
        const int L = min(off[b + 1] - t0, DCR_RMAX);
        const int nkt = (L + 15) >> 4;

        // ---- stage 0: the folded-query slabs into region B (free: the previous window's stage D ended with a barrier)
#pragma unroll
        for (int s = 0; s < (SHARED ? 1 : NQ); ++s) fa[s] = g4v{0.f, 0.f, 0.f, 0.f};
        fold_half(b, 0);
#pragma unroll
        for (int s = 0; s < (SHARED ? 1 : NQ); ++s) asm volatile("" : "+v"(fa[s]));    // (the first half's FMAs are done HERE)
        DCR_PIN();
        load_wk(16);                                            // the second half of W_k into the registers of the first
        DCR_PIN();
        fold_half(b, 16);
        fold_store();
        if (SHARED) {
#pragma unroll
            for (int i = 0; i < DCR_QK / 4 / 512; ++i) reinterpret_cast<g4v*>(regB)[tid + 512 * i] = qs[i];
        }
        // the key rows requested a window ago are consumed HERE, ahead of the DMA: the compiler's waits for them cover loads
        // issued long ago and not the DMA below (a wait placed behind it would be a vmcnt(0))
        add_pos(b);
#pragma unroll
        for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(xk[q]) : : "memory");
        // ---- the window's rows -> LDS image (stage C), one row per instruction; wave w: slots w, w + 8, ...  ALWAYS 14
        // instructions per wave (slots past the window take a copy of its last row, slots past the image repeat slot 109
        // with the same bytes): a compile-time count keeps every later vmcnt wait counted
        {
            const float* xb = X + (size_t)t0 * 256;
#pragma unroll
            for (int i = 0; i < 14; ++i) {
                const int slot = min(wave + 8 * i, DCR_RMAX - 1);
                const int r = min(slot, L - 1);
                DCR_GLDS16(xb + (size_t)r * 256 + 4u * (lane ^ ((slot & 1) << 2)), rows + slot * 256);
            }
        }
        DCR_BARRIER();                                          // slabs complete

        // ---- stage A: scores of this wave's key tile against the 48 pairs
        g4v sc[NPT];
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) sc[pt] = g4v{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (wave < nkt) {
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                g4v ch[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) ch[r] = g4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const g4v w4 = *reinterpret_cast<const g4v*>(regB + slab(pt, q) + (pt < 2 ? rdA : rdB));
#pragma unroll
                    for (int r = 0; r < 4; ++r) ch[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(xk[q][r], w4[r], ch[r], 0, 0, 0);
                }
                g4v s4 = (ch[0] + ch[1]) + (ch[2] + ch[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (wave * 16 + 4 * lg + r >= L) s4[r] = -INFINITY;       // key 4 lg + r of the tile is padding
                sc[pt] = s4;
            }
        }
        // stage D's W_v: 32 requests into the registers that held the key rows, in flight under the softmax and stage C
        // (the next window's key rows as well -- 64 + 128 registers through stage C -- spills: measured)
        DCR_PIN();
#pragma unroll
        for (int j = 0; j < 32; ++j) wv[j] = dcr_ldg(wvb + j * 256, l4);
        DCR_PIN();

        // ---- stage B: softmax over the keys of each pair
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            float m = fmaxf(fmaxf(sc[pt][0], sc[pt][1]), fmaxf(sc[pt][2], sc[pt][3]));
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            if (lg == 0) smax[wave * NPP + pt * 16 + li] = m;
        }
        DCR_BARRIER();                                          // every wave is past stage A: the slabs are dead
        float inv[NPT];
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            float m = smax[pt * 16 + li];
#pragma unroll
            for (int w = 1; w < 8; ++w) m = fmaxf(m, smax[w * NPP + pt * 16 + li]);
            const float m2 = m * 1.4426950408889634f;
            float l = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(sc[pt][r], 1.4426950408889634f, -m2));   // exp(-inf) = 0
                sc[pt][r] = e;
                l += e;
            }
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            if (lg == 0) ssum[wave * NPP + pt * 16 + li] = l;
        }
        DCR_BARRIER();
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            float l = ssum[pt * 16 + li];
#pragma unroll
            for (int w = 1; w < 8; ++w) l += ssum[w * NPP + pt * 16 + li];
            inv[pt] = 1.0f / l;
        }
        float* Pt = regB;
        if (wave < nkt) {
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Pt[(wave * 16 + 4 * lg + r) * NPP + pt * 16 + li] = sc[pt][r] * inv[pt];
        }
        // this wave's rows of the LDS image have landed: everything but the requests issued after them (32 W_v)
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        DCR_BARRIER();                                          // Pt complete (rows [L, 16 nkt) are zeros), image complete

        // ---- stage C: ctx[p][c] = sum_j P[p][j] * mem[j][c]; wave = channels [32 w, 32 w + 32) x all pairs, from LDS
        g4v acc[2][NPT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[t][pt] = g4v{0.f, 0.f, 0.f, 0.f};
        {
            // key 4 s + lg: odd keys (lg odd) hold their 64-B blocks swapped pairwise
            const int c0 = 32 * wave + li + ((lg & 1) << 4), c1 = 32 * wave + li + (((lg & 1) ^ 1) << 4);
            const int nks = nkt * 4;
#pragma unroll 4
            for (int s = 0; s < nks; ++s) {
                const int key = min(4 * s + lg, L - 1);         // padding keys carry P = 0: any landed row will do
                const float a0 = rows[key * 256 + c0], a1 = rows[key * 256 + c1];
                const float* pr = Pt + (4 * s + lg) * NPP + li;
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) {
                    const float pb = pr[pt * 16];
                    acc[0][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, pb, acc[0][pt], 0, 0, 0);
                    acc[1][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, pb, acc[1][pt], 0, 0, 0);
                }
            }
        }
        DCR_BARRIER();                                          // Pt and the image are dead
        float* ctxs = regB;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                const int p = pt * 16 + li;
                if (p < NP) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) ctxs[p * DCR_CTX_LD + 32 * wave + 16 * t + 4 * lg + r] = acc[t][pt][r];
                }
            }
        DCR_BARRIER();

        // ---- stage D: out[s][o] = sum_c WvT[c][o] * ctx[(s, o / 32)][c] + bv[o]; wave = the slice c in [32 w, 32 w + 32) of
        // the contraction, lane = outputs 4 lane .. + 3 (one head); the eight partial sums meet in LDS
        {
            g4v o[NQ];
#pragma unroll
            for (int s = 0; s < NQ; ++s) o[s] = g4v{0.f, 0.f, 0.f, 0.f};
            const float* crow = ctxs + (lane >> 3) * DCR_CTX_LD + 32 * wave;
#pragma unroll
            for (int j4 = 0; j4 < 8; ++j4) {
#pragma unroll
                for (int s = 0; s < NQ; ++s) {
                    const g4v cx = *reinterpret_cast<const g4v*>(crow + s * 8 * DCR_CTX_LD + 4 * j4);
#pragma unroll
                    for (int u = 0; u < 4; ++u) o[s] += wv[4 * j4 + u] * cx[u];
                }
            }
            DCR_BARRIER();                                      // every wave is done reading ctx
            float* part = regB;                                 // [8 waves][5][256]
#pragma unroll
            for (int s = 0; s < NQ; ++s) *reinterpret_cast<g4v*>(part + (wave * NQ + s) * 256 + 4 * lane) = o[s];
            // the next window's first half of W_k and its key rows: into the registers W_v and the partial sums have just left
            DCR_PIN();
            load_wk(0);
            load_slabs();
            const int nb = b + (int)gridDim.x < B ? b + (int)gridDim.x : b;        // (after the last window: a re-read)
            load_keys(nb);
            DCR_PIN();
            DCR_BARRIER();
            for (int i = tid; i < NQ * 256; i += 512) {
                float v = part[i];
#pragma unroll
                for (int w = 1; w < 8; ++w) v += part[w * NQ * 256 + i];
                OUT[(size_t)b * NQ * 256 + i] = v + bv[i & 255];
            }
        }
        DCR_BARRIER();                                          // region B is free for the next window's slabs
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // no LDS-DMA may outlive the workgroup's LDS
}

template <bool POSTAB>
static int launch_res_one(const float* DQ, const float* XP, const float* X, const float* pos_rows, const int* vlen,
                          const int* off, const float* Wk, const float* WvT, const float* bv, float* OUT, int B,
                          float* qk_slabs, hipStream_t s) {
    static DeviceOnce once;     // the opt-in to > 64 KiB of LDS + the CU count that sizes the persistent grid: once per device
    int n_cu = 0;
    CONE_CHECK_HIP(device_once(once, [] {
        hipError_t rc = hipFuncSetAttribute((const void*)dec_cross_res_kernel<POSTAB, false>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, DCR_LDS_FLOATS * 4);
        if (rc == hipSuccess)
            rc = hipFuncSetAttribute((const void*)dec_cross_res_kernel<POSTAB, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     DCR_LDS_FLOATS * 4);
        return rc;
    }, &n_cu));
    const int grid = B < n_cu ? B : n_cu;
    if (qk_slabs) {     // window-independent queries: the (compact) operand slabs once, by one workgroup
        hipLaunchKernelGGL((dec_cross_res_kernel<POSTAB, false>), dim3(1), dim3(512), DCR_LDS_FLOATS * 4, s, DQ, XP, X, pos_rows,
                           vlen, off, Wk, WvT, bv, OUT, (const float*)nullptr, qk_slabs, B);
        CONE_LAUNCH_CHECK();
        hipLaunchKernelGGL((dec_cross_res_kernel<POSTAB, true>), dim3(grid), dim3(512), DCR_LDS_FLOATS * 4, s, DQ, XP, X, pos_rows,
                           vlen, off, Wk, WvT, bv, OUT, (const float*)qk_slabs, (float*)nullptr, B);
        CONE_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL((dec_cross_res_kernel<POSTAB, false>), dim3(grid), dim3(512), DCR_LDS_FLOATS * 4, s, DQ, XP, X, pos_rows,
                       vlen, off, Wk, WvT, bv, OUT, (const float*)nullptr, (float*)nullptr, B);
    CONE_LAUNCH_CHECK();
    return 0;
}

template <int KTW, bool POSTAB, int NQ = 5>
static int launch_mfma_one(const float* DQ, const float* XP, const float* X, const float* pos_rows, const int* vlen,
                           const int* off, const float* Wk, const float* WvT, const float* bv, float* OUT, int B,
                           float* qk_slabs, hipStream_t s, const float* sal_w = nullptr, const float* sal_b = nullptr,
                           float* sal = nullptr, int sal_ld = 0) {
    using C = DecCrossMfmaCfg<KTW, NQ>;
    static DeviceOnce once;     // the opt-in to > 64 KiB of LDS: once per device
    CONE_CHECK_HIP(device_once(once, [] {
        return hipFuncSetAttribute((const void*)dec_cross_mfma_kernel<KTW, POSTAB, NQ>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   C::LDS_FLOATS * 4);
    }));
    if (qk_slabs) {     // window-independent queries: the operand slabs once, by one workgroup (window 0's rows)
        hipLaunchKernelGGL((dec_cross_mfma_kernel<KTW, POSTAB, NQ>), dim3(1), dim3(512), C::LDS_FLOATS * 4, s, DQ, XP, X,
                           pos_rows, vlen, off, Wk, WvT, bv, OUT, (const float*)nullptr, qk_slabs, (const float*)nullptr,
                           (const float*)nullptr, (float*)nullptr, 0);
        CONE_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((dec_cross_mfma_kernel<KTW, POSTAB, NQ>), dim3(B), dim3(512), C::LDS_FLOATS * 4, s, DQ, XP, X,
                       pos_rows, vlen, off, Wk, WvT, bv, OUT, (const float*)qk_slabs, (float*)nullptr, sal_w, sal_b, sal, sal_ld);
    CONE_LAUNCH_CHECK();
    return 0;
}

// scratch of the shared-query operand slabs: the largest instantiated slot count's (10 slots: 5 pair tiles x 16 KiB)
size_t dec_cross_mfma_slab_floats() { return DecCrossMfmaCfg<1, 10>::QK_FLOATS; }

// the matrix-core forms: 5 slots everywhere (<= 256 tokens); 3 / 8 / 10 slots on the two-read kernel, table form (10 slots
// with windows of more than 128 tokens would need 166 KiB of LDS: the unfolded decoder runs them, up to 192 tokens)
bool dec_cross_mfma_supported(int nq, int Lmax, bool table) {
    if (nq == 5) return Lmax <= 256;        // (two key tiles per wave: 8 waves x 2 x 16 keys)
    if (!table) return false;
    return ((nq == 3 || nq == 8) && Lmax <= 256) || (nq == 10 && Lmax <= 128);
}

// qk_slabs != null: the NQ queries of every window are the SAME rows (DQ holds them for window 0 at least; first decoder
// layer) -- the folded-key operand is built once into that scratch (dec_cross_mfma_slab_floats() floats).
bool dec_cross_res_supported(int nq, int Lmax) { return nq == 5 && Lmax <= DCR_RMAX; }

int launch_dec_cross_mfma(const float* DQ, const float* XP, const float* X, const float* pos_rows, const int* vlen,
                          const int* off, const float* Wk, const float* WvT, const float* bv, float* OUT, int B, int nq,
                          int Lmax, float* qk_slabs, hipStream_t s, int form, const float* sal_w, const float* sal_b,
                          float* sal, int sal_ld) {
    CONE_REQUIRE(dec_cross_mfma_supported(nq, Lmax, !XP), "fused decoder cross-attention: nq=%d Lmax=%d unsupported", nq, Lmax);
    CONE_REQUIRE(XP || (pos_rows && vlen), "fused decoder cross-attention: needs memory+pos rows or the sine table");
    if (B <= 0) return 0;
    ProfScope ps(PK_DEC_CROSS, B, Lmax, nq, nullptr, s);
    if (nq != 5) {      // other slot counts: the two-read kernel, table form
#define CONE_DCM_NQ(N)                                                                                                        \
        if (nq == N)                                                                                                          \
            return Lmax <= 128 ? launch_mfma_one<1, true, N>(DQ, nullptr, X, pos_rows, vlen, off, Wk, WvT, bv, OUT, B, qk_slabs, s, \
                                                             sal_w, sal_b, sal, sal_ld)                                      \
                               : launch_mfma_one<2, true, N>(DQ, nullptr, X, pos_rows, vlen, off, Wk, WvT, bv, OUT, B, qk_slabs, s, \
                                                             sal_w, sal_b, sal, sal_ld);
        CONE_DCM_NQ(3)
        CONE_DCM_NQ(8)
#undef CONE_DCM_NQ
        return launch_mfma_one<1, true, 10>(DQ, nullptr, X, pos_rows, vlen, off, Wk, WvT, bv, OUT, B, qk_slabs, s, sal_w, sal_b,
                                            sal, sal_ld);
    }
    CONE_REQUIRE(!sal || (!XP && form != 4 && sal_w && sal_b && sal_ld > 0), "fused decoder cross-attention: the saliency "
                 "head rides only on the table form of the register-row kernels");
    // rows read once: registers transposed through LDS by channel quarters (two workgroups per CU).  Table form only: with a
    // precomputed memory + pos matrix the registers of stage A do not hold the rows stage C multiplies
    // -- and only where the queries are shared (first layer): with per-window queries the fold of stage 0 is VALU work this
    // kernel has no idle issue slots left to hide (1.62 ms against the two-read kernel's 1.50 on 20 000 windows; shared: 1.36
    // against 1.47)
    // (form 5: everywhere it can run -- the parity tests and tools/dec_cross_bench.py)
    if ((form == 5 || (form == 2 && qk_slabs)) && Lmax <= 128 && !XP)
        return launch_x_one<true>(DQ, nullptr, X, pos_rows, vlen, off, Wk, WvT, bv, OUT, B, qk_slabs, s, sal_w, sal_b, sal, sal_ld);
    const bool resident = form == 4;
    if (resident && dec_cross_res_supported(nq, Lmax)) {    // opt-in: rows LDS-resident, one HBM read per row (<= 110 tokens)
        if (XP) return launch_res_one<false>(DQ, XP, X, nullptr, nullptr, off, Wk, WvT, bv, OUT, B, qk_slabs, s);
        return launch_res_one<true>(DQ, nullptr, X, pos_rows, vlen, off, Wk, WvT, bv, OUT, B, qk_slabs, s);
    }
    if (XP) {
        if (Lmax <= 128) return launch_mfma_one<1, false>(DQ, XP, X, nullptr, nullptr, off, Wk, WvT, bv, OUT, B, qk_slabs, s);
        return launch_mfma_one<2, false>(DQ, XP, X, nullptr, nullptr, off, Wk, WvT, bv, OUT, B, qk_slabs, s);
    }
    if (Lmax <= 128)
        return launch_mfma_one<1, true>(DQ, nullptr, X, pos_rows, vlen, off, Wk, WvT, bv, OUT, B, qk_slabs, s, sal_w, sal_b, sal, sal_ld);
    return launch_mfma_one<2, true>(DQ, nullptr, X, pos_rows, vlen, off, Wk, WvT, bv, OUT, B, qk_slabs, s, sal_w, sal_b, sal, sal_ld);
}

}  // namespace cone
