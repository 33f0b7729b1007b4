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

template <int KTW>
struct DecCrossMfmaCfg {
    static constexpr int NP = 40, NPT = 3, NPP = 48;   // pairs, pair tiles, padded pairs
    static constexpr int KP = 128 * KTW;               // key capacity: 8 waves x KTW tiles x 16
    static constexpr int CTX_LD = 260;                 // ctx row stride (floats): 16-B aligned rows, spread banks
    static constexpr int QK_FLOATS = NPT * 16 * 256;   // operand slabs of qk (48 KiB)
    static constexpr int A_FLOATS = NPP * CTX_LD > QK_FLOATS ? NPP * CTX_LD : QK_FLOATS;    // qk slabs, later ctx
    static constexpr int PT_FLOATS = KP * NPP;          // Pt[key][pair]; first the scaled queries, last stage D's partials
    static constexpr int RED_FLOATS = 2 * 8 * NPP;      // per-wave softmax maxima / sums
    static constexpr int LDS_FLOATS = A_FLOATS + PT_FLOATS + RED_FLOATS;
};

template <int KTW, bool POSTAB>
__global__ __launch_bounds__(512, (KTW == 1 ? 4 : 2)) void dec_cross_mfma_kernel(const float* __restrict__ DQ,
                                                                 const float* __restrict__ XP,
                                                                 const float* __restrict__ X,
                                                                 const float* __restrict__ pos_rows,
                                                                 const int* __restrict__ vlen,
                                                                 const int* __restrict__ off,
                                                                 const float* __restrict__ Wk,
                                                                 const float* __restrict__ WvT,
                                                                 const float* __restrict__ bv, float* __restrict__ OUT,
                                                                 const float* __restrict__ QKS, float* __restrict__ QKS_OUT) {
    // QKS != null: the qk operand slabs are window-independent (first decoder layer: tgt = 0, the queries are the
    // same for every window) and were written once by a one-workgroup launch of this kernel with QKS_OUT set.
    using C = DecCrossMfmaCfg<KTW>;
    constexpr int NQ = 5;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* qkf = smem;                          // [3][16] slabs of [16 pairs][16 channels]; later ctx[48][260]
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
            if (POSTAB && key < lv) {
                const float* pp = prow + (size_t)key * 256 + 4 * lg;
#pragma unroll
                for (int q = 0; q < 16; ++q) xk[w][q] += *reinterpret_cast<const g4v*>(pp + 16 * q);
            }
        }
    }

    const int col = tid & 255, half = tid >> 8;
    if (QKS) {      // precomputed slabs: 48 KiB from the L2 instead of 256 KiB of W_k
        for (int i = tid; i < C::QK_FLOATS / 4; i += 512)
            reinterpret_cast<g4v*>(qkf)[i] = reinterpret_cast<const g4v*>(QKS)[i];
    } else {
    // rows 40 .. 47 of the last pair tile's slabs = 0
    for (int i = tid; i < 16 * 8 * 16; i += 512) {          // (q, row 8 .. 15, 16 floats) of pair tile 2
        const int q = i >> 7, r = 8 + ((i >> 4) & 7), c = i & 15;
        qkf[(2 * 16 + q) * 256 + r * 16 + c] = 0.f;
    }
    // ---- stage 0: qk[p][c] = sqrt(1/32) sum_d q[p][d] * Wk[h*32+d][c], thread = channel c (two thread groups split the
    // heads); the window's query values are wave-uniform: they come through the scalar cache (s_load) straight into the
    // FMAs' scalar operands -- no LDS staging, no LDS read per FMA pair.  Stored in slab (pair tile, c / 16), row
    // pair % 16, physical chunk ((c / 4) % 4) ^ swz(row)
    const float* __restrict__ qb = DQ + (size_t)b * NQ * 256;
    const int h0 = __builtin_amdgcn_readfirstlane(4 * half);
    for (int h = h0; h < h0 + 4; ++h) {
        g2v a[NQ];
#pragma unroll
        for (int s = 0; s < NQ; ++s) a[s] = g2v{0.f, 0.f};
        const float* wcol = Wk + (size_t)h * 32 * 256 + col;
        // the head's 32 rows of W_k in four groups of 8, the next group requested ahead of this group's FMAs
        float wn[8], wc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) wn[i] = wcol[i * 256];
#pragma unroll
        for (int d8 = 0; d8 < 4; ++d8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) wc[i] = wn[i];
            if (d8 < 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) wn[i] = wcol[((d8 + 1) * 8 + i) * 256];
            }
#pragma unroll
            for (int s = 0; s < NQ; ++s) {
                const float* qp = qb + s * 256 + h * 32 + d8 * 8;
#pragma unroll
                for (int i = 0; i < 8; i += 2)
                    a[s] = __builtin_elementwise_fma(g2v{qp[i], qp[i + 1]}, g2v{wc[i], wc[i + 1]}, a[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < NQ; ++s) {
            const int p = s * 8 + h, row = p & 15;
            qkf[((p >> 4) * 16 + (col >> 4)) * 256 + row * 16 + ((((col >> 2) & 3) ^ dcm_swz16(row)) << 2) + (col & 3)] =
                (a[s].x + a[s].y) * 0.17677669529663687f;
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

    // ---- stage D: out[s][o] = sum_c WvT[c][o] * ctx[(s, o/32)][c] + bv[o], thread = output column o; the two thread
    // groups split the c range (each W_v^T element is loaded once per window), partial sums meet in LDS
    {
        g2v o[NQ];
#pragma unroll
        for (int s = 0; s < NQ; ++s) o[s] = g2v{0.f, 0.f};
        const int h = col >> 5;
        const float* wcol = WvT + (size_t)half * 128 * 256 + col;
        const float* crow = ctxs + h * C::CTX_LD + half * 128;
#pragma unroll 4
        for (int c4 = 0; c4 < 32; ++c4) {
            const g2v w01 = {wcol[(c4 * 4 + 0) * 256], wcol[(c4 * 4 + 1) * 256]};
            const g2v w23 = {wcol[(c4 * 4 + 2) * 256], wcol[(c4 * 4 + 3) * 256]};
#pragma unroll
            for (int s = 0; s < NQ; ++s) {
                const g4v cx = *reinterpret_cast<const g4v*>(crow + s * 8 * C::CTX_LD + c4 * 4);
                o[s] = __builtin_elementwise_fma(cx.xy, w01, o[s]);
                o[s] = __builtin_elementwise_fma(cx.zw, w23, o[s]);
            }
        }
        float* red = Pt;                                    // Pt is dead since the end of stage C (barrier above)
        if (half == 1) {
#pragma unroll
            for (int s = 0; s < NQ; ++s) red[s * 256 + col] = o[s].x + o[s].y;
        }
        __syncthreads();
        if (half == 0) {
            const float bias = bv[col];
#pragma unroll
            for (int s = 0; s < NQ; ++s)
                OUT[(size_t)(b * NQ + s) * 256 + col] = ((o[s].x + o[s].y) + red[s * 256 + col]) + bias;
        }
    }
}

template <int KTW, bool POSTAB>
static int launch_mfma_one(const float* DQ, const float* XP, const float* X, const float* pos_rows, const int* vlen,
                           const int* off, const float* Wk, const float* WvT, const float* bv, float* OUT, int B,
                           float* qk_slabs, hipStream_t s) {
    using C = DecCrossMfmaCfg<KTW>;
    static DeviceOnce once;     // the opt-in to > 64 KiB of LDS: once per device
    CONE_CHECK_HIP(device_once(once, [] {
        return hipFuncSetAttribute((const void*)dec_cross_mfma_kernel<KTW, POSTAB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   C::LDS_FLOATS * 4);
    }));
    if (qk_slabs) {     // window-independent queries: the operand slabs once, by one workgroup (window 0's rows)
        hipLaunchKernelGGL((dec_cross_mfma_kernel<KTW, POSTAB>), dim3(1), dim3(512), C::LDS_FLOATS * 4, s, DQ, XP, X,
                           pos_rows, vlen, off, Wk, WvT, bv, OUT, (const float*)nullptr, qk_slabs);
        CONE_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((dec_cross_mfma_kernel<KTW, POSTAB>), dim3(B), dim3(512), C::LDS_FLOATS * 4, s, DQ, XP, X,
                       pos_rows, vlen, off, Wk, WvT, bv, OUT, (const float*)qk_slabs, (float*)nullptr);
    CONE_LAUNCH_CHECK();
    return 0;
}

size_t dec_cross_mfma_slab_floats() { return DecCrossMfmaCfg<1>::QK_FLOATS; }

// qk_slabs != null: the NQ queries of every window are the SAME rows (DQ holds them for window 0 at least; first decoder
// layer) -- the folded-key operand is built once into that scratch (dec_cross_mfma_slab_floats() floats).
int launch_dec_cross_mfma(const float* DQ, const float* XP, const float* X, const float* pos_rows, const int* vlen,
                          const int* off, const float* Wk, const float* WvT, const float* bv, float* OUT, int B, int nq,
                          int Lmax, float* qk_slabs, hipStream_t s) {
    CONE_REQUIRE(dec_cross_supported(nq, Lmax), "fused decoder cross-attention: nq=%d Lmax=%d unsupported", nq, Lmax);
    CONE_REQUIRE(XP || (pos_rows && vlen), "fused decoder cross-attention: needs memory+pos rows or the sine table");
    if (B <= 0) return 0;
    ProfScope ps(PK_DEC_CROSS, B, Lmax, nq, nullptr, s);
    if (XP) {
        if (Lmax <= 128) return launch_mfma_one<1, false>(DQ, XP, X, nullptr, nullptr, off, Wk, WvT, bv, OUT, B, qk_slabs, s);
        return launch_mfma_one<2, false>(DQ, XP, X, nullptr, nullptr, off, Wk, WvT, bv, OUT, B, qk_slabs, s);
    }
    if (Lmax <= 128) return launch_mfma_one<1, true>(DQ, nullptr, X, pos_rows, vlen, off, Wk, WvT, bv, OUT, B, qk_slabs, s);
    return launch_mfma_one<2, true>(DQ, nullptr, X, pos_rows, vlen, off, Wk, WvT, bv, OUT, B, qk_slabs, s);
}

}  // namespace cone
