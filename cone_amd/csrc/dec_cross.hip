// Fused decoder cross-attention (cone/transformer.py:308-311 -> nn.MultiheadAttention) for NQ query slots
// per window, with the memory K / V projections folded into the 8*NQ (slot, head) pairs:
//
//   score[p][j] = (q_p * s) . (W_k,h (mem_j + pos_j) + b_k,h)  =  qk_p . (mem_j + pos_j) + const_p
//       with qk_p = W_k,h^T (q_p * s)  (256-vector); const_p is the same for every key of the pair and
//       drops out of the softmax;
//   out_p = sum_j P[p][j] (W_v,h mem_j + b_v,h) = W_v,h (sum_j P[p][j] mem_j) + b_v,h       (sum_j P = 1).
//
// So the two M-row GEMMs that project every memory token to K and V for every decoder layer (12 % of the
// window model's FLOPs) are replaced by 4 small per-window stages, all VALU + LDS (no padding waste: 40 pairs
// and ~101 keys do not fill 32x32 MFMA tiles well):
//   0. qk_p for the 40 pairs            (thread = column c, W_k read coalesced)
//   A. scores: lane = key, wave = 5 pairs; mem+pos rows staged through LDS in 32-column chunks
//   B. softmax per pair (all keys of a pair live in one wave)
//   C. ctx_p = sum_j P[p][j] mem_j      (lane = 4 columns, wave = 5 pairs, mem rows read coalesced)
//   D. out = W_v,h ctx_p + b_v          (thread = output column, W_v^T read coalesced)
// Results equal the unfused path up to fp32 re-association (~1e-6 relative).
#include <mutex>

#include "common.h"

namespace cone {

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));   // (float4 struct copies lower to memcpy -> scratch)

template <int CTRL>
__device__ __forceinline__ float dppx(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dppx_rows(float v, float identity) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity),
                                                                 __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wsum(float v) {
    v += dppx<0xB1>(v); v += dppx<0x4E>(v); v += dppx<0x141>(v); v += dppx<0x140>(v);
    v += dppx_rows<0x142, 0xA>(v, 0.f);
    v += dppx_rows<0x143, 0xC>(v, 0.f);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wmax(float v) {
    v = fmaxf(v, dppx<0xB1>(v)); v = fmaxf(v, dppx<0x4E>(v)); v = fmaxf(v, dppx<0x141>(v)); v = fmaxf(v, dppx<0x140>(v));
    // quad_perm / mirrors with bound_ctrl read 0 for nothing here (all lanes valid); row broadcasts keep
    // the old value (-inf identity) on rows that do not receive
    v = fmaxf(v, dppx_rows<0x142, 0xA>(v, -INFINITY));
    v = fmaxf(v, dppx_rows<0x143, 0xC>(v, -INFINITY));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <int NQ, int NKL>
struct DecCrossCfg {
    static constexpr int NP = 8 * NQ;                 // (slot, head) pairs of one window
    static constexpr int NT = 512, NWV = NT / 64;     // 8 waves: 16 per CU at two workgroups (latency hiding)
    static constexpr int PPW = NP / NWV;              // pairs per wave
    static constexpr int PPWP = (PPW + 3) / 4 * 4;    //   padded to whole float4s in the P image
    static constexpr int NPP = NWV * PPWP;
    static constexpr int KP = 64 * NKL;               // key capacity
    static constexpr int TILE_LD = 36;                // 32 columns + 4 pad: conflict-free ds_read_b128 per key
    static constexpr int UNION = KP * (NPP > TILE_LD ? NPP : TILE_LD);
    static constexpr int LDS_FLOATS = NP * 32 + NP * 256 + UNION;
};

// POSTAB: the keys memory + pos are formed in the staging loads from the memory rows X and the static sine table
// (row (lv, p) of pos_rows for clip token p of a window with lv clips; text tokens carry no position, cone/model.py:106)
// instead of being read from a precomputed (M, 256) matrix XP.
template <int NQ, int NKL, bool POSTAB>
__global__ __launch_bounds__(512, 4) void dec_cross_kernel(const float* __restrict__ DQ,
                                                           const float* __restrict__ XP,
                                                           const float* __restrict__ X,
                                                           const float* __restrict__ pos_rows,
                                                           const int* __restrict__ vlen,
                                                           const int* __restrict__ off,
                                                           const float* __restrict__ Wk,
                                                           const float* __restrict__ WvT,
                                                           const float* __restrict__ bv, float* __restrict__ OUT) {
    using C = DecCrossCfg<NQ, NKL>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* qs = smem;                       // [NP][32]   scaled queries, pair p = s*8 + h
    float* qkf = smem + C::NP * 32;         // [NP][256]  folded keys-side queries; later ctx
    float* tile = qkf + C::NP * 256;        // [KP][36]   mem+pos chunk; later P [KP][NPP]
    const int b = blockIdx.x;
    const int t0 = off[b];
    const int L = min(off[b + 1] - t0, C::KP);
    const int lv = POSTAB ? vlen[b] : 0;
    const float* __restrict__ KEYS = POSTAB ? X : XP;
    const float* __restrict__ prow = POSTAB ? pos_rows + (size_t)(lv * (lv - 1) / 2) * 256 : nullptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- q * sqrt(1/32) into LDS as [pair = s*8+h][32]
    static_assert(C::NP % C::NWV == 0, "pairs must split evenly over the waves");
    for (int i = tid; i < NQ * 256; i += C::NT) {
        const int s = i >> 8, o = i & 255;
        qs[(s * 8 + (o >> 5)) * 32 + (o & 31)] = DQ[(size_t)(b * NQ + s) * 256 + o] * 0.17677669529663687f;
    }
    __syncthreads();

    // ---- stage 0: qkf[p][c] = sum_d qs[p][d] * Wk[h*32+d][c], thread = column c.  Packed fp32 FMAs
    // (v_pk_fma_f32): even / odd d accumulate in the two halves, summed at the end.
    const int col = tid & 255, half = tid >> 8;          // two thread groups split the heads (stage 0) / slots (D)
    for (int h = 4 * half; h < 4 * half + 4; ++h) {
        f2v a[NQ];
#pragma unroll
        for (int s = 0; s < NQ; ++s) a[s] = f2v{0.f, 0.f};
        const float* wcol = Wk + (size_t)h * 32 * 256 + col;
#pragma unroll 4
        for (int d4 = 0; d4 < 8; ++d4) {      // 16 W_k loads in flight per thread: this stage is L2-latency bound
            const f2v w01 = {wcol[(d4 * 4 + 0) * 256], wcol[(d4 * 4 + 1) * 256]};
            const f2v w23 = {wcol[(d4 * 4 + 2) * 256], wcol[(d4 * 4 + 3) * 256]};
#pragma unroll
            for (int s = 0; s < NQ; ++s) {
                const f4v q4 = *reinterpret_cast<const f4v*>(qs + (s * 8 + h) * 32 + d4 * 4);
                a[s] = __builtin_elementwise_fma(q4.xy, w01, a[s]);
                a[s] = __builtin_elementwise_fma(q4.zw, w23, a[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < NQ; ++s) qkf[(s * 8 + h) * 256 + col] = a[s].x + a[s].y;
    }

    // ---- stage A: scores.  lane = key (NKL keys per lane), wave = PPW pairs.
    f2v sc2[NKL][C::PPW];                                    // even / odd column partial sums
#pragma unroll
    for (int kk = 0; kk < NKL; ++kk)
#pragma unroll
        for (int pp = 0; pp < C::PPW; ++pp) sc2[kk][pp] = f2v{0.f, 0.f};
    constexpr int TPASS = C::KP / 64;                       // staging passes: 64 key rows per pass
    f4v pf[TPASS];
    const int srow = tid >> 3, spart = tid & 7;
#define DC_FETCH(ch_)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < TPASS; ++i) {                                                     \
        int r = srow + 64 * i;                                                                              \
        r = r < L ? r : L - 1;                                                                              \
        pf[i] = *reinterpret_cast<const f4v*>(KEYS + (size_t)(t0 + r) * 256 + (ch_) * 32 + spart * 4);   \
        if (POSTAB && r < lv) pf[i] += *reinterpret_cast<const f4v*>(prow + (size_t)r * 256 + (ch_) * 32 + spart * 4); \
    }
#define DC_STASH()                                                                                          \
    _Pragma("unroll") for (int i = 0; i < TPASS; ++i)                                                       \
        *reinterpret_cast<f4v*>(tile + (srow + 64 * i) * C::TILE_LD + spart * 4) = pf[i];
    DC_FETCH(0)
    DC_STASH()
    __syncthreads();                                        // tile(0) and qkf complete
    for (int ch = 0; ch < 8; ++ch) {
        if (ch < 7) { DC_FETCH(ch + 1) }                    // in flight during this chunk's FMAs
#pragma unroll 2
        for (int c4 = 0; c4 < 8; ++c4) {
            f4v xk[NKL];
#pragma unroll
            for (int kk = 0; kk < NKL; ++kk)
                xk[kk] = *reinterpret_cast<const f4v*>(tile + (lane + 64 * kk) * C::TILE_LD + c4 * 4);
#pragma unroll
            for (int pp = 0; pp < C::PPW; ++pp) {
                const f4v qv = *reinterpret_cast<const f4v*>(qkf + (wave * C::PPW + pp) * 256 + ch * 32 + c4 * 4);
#pragma unroll
                for (int kk = 0; kk < NKL; ++kk) {
                    sc2[kk][pp] = __builtin_elementwise_fma(xk[kk].xy, qv.xy, sc2[kk][pp]);
                    sc2[kk][pp] = __builtin_elementwise_fma(xk[kk].zw, qv.zw, sc2[kk][pp]);
                }
            }
        }
        __syncthreads();                                    // everyone is done reading this chunk
        if (ch < 7) {
            DC_STASH()
            __syncthreads();
        }
    }
#undef DC_FETCH
#undef DC_STASH

    // ---- stage B: softmax over the keys of each pair -> P[key][wave*PPWP + pp]  (aliases the tile)
    float* P = tile;
    float sc[NKL][C::PPW];
#pragma unroll
    for (int kk = 0; kk < NKL; ++kk)
#pragma unroll
        for (int pp = 0; pp < C::PPW; ++pp) sc[kk][pp] = sc2[kk][pp].x + sc2[kk][pp].y;
#pragma unroll
    for (int pp = 0; pp < C::PPW; ++pp) {
        float m = -INFINITY;
#pragma unroll
        for (int kk = 0; kk < NKL; ++kk) {
            if (lane + 64 * kk >= L) sc[kk][pp] = -INFINITY;
            m = fmaxf(m, sc[kk][pp]);
        }
        m = wmax(m);
        const float m2 = m * 1.4426950408889634f;
        float l = 0.f;
#pragma unroll
        for (int kk = 0; kk < NKL; ++kk) {
            sc[kk][pp] = __builtin_amdgcn_exp2f(fmaf(sc[kk][pp], 1.4426950408889634f, -m2));
            l += sc[kk][pp];
        }
        const float inv = 1.0f / wsum(l);
#pragma unroll
        for (int kk = 0; kk < NKL; ++kk) P[(lane + 64 * kk) * C::NPP + wave * C::PPWP + pp] = sc[kk][pp] * inv;
    }
    __syncthreads();

    // ---- stage C: ctx[p][c] = sum_j P[j][p] * mem[j][c]; lane = 4 columns, wave = its PPW pairs.  Every wave needs
    // every memory row, so the rows go through LDS once per workgroup (16-key chunks, double-buffered in the dead
    // qkf region, register prefetch of the next chunk) instead of eight times through the vector L1.
    f4v ctx[C::PPW];
#pragma unroll
    for (int pp = 0; pp < C::PPW; ++pp) ctx[pp] = f4v{0.f, 0.f, 0.f, 0.f};
    {
        constexpr int CK = 16;
        float* xs = qkf;                                     // [2][CK][256]
        const int nch = (L + CK - 1) / CK;
        const int xr = tid >> 5, xc = (tid & 31) * 4;        // 512 threads = 16 rows x 32 float4, two halves of a row
        f4v xa, xb;
#define DC_XFETCH(ch_)                                                                              \
    {                                                                                               \
        const int j = min((ch_) * CK + xr, L - 1);                                                  \
        const float* src = X + (size_t)(t0 + j) * 256 + xc;                                         \
        xa = *reinterpret_cast<const f4v*>(src);                                                    \
        xb = *reinterpret_cast<const f4v*>(src + 128);                                              \
    }
#define DC_XSTASH(buf_)                                                                             \
    {                                                                                               \
        *reinterpret_cast<f4v*>(xs + (buf_) * (CK * 256) + xr * 256 + xc) = xa;                     \
        *reinterpret_cast<f4v*>(xs + (buf_) * (CK * 256) + xr * 256 + xc + 128) = xb;               \
    }
        DC_XFETCH(0)
        DC_XSTASH(0)
        __syncthreads();
        for (int ch = 0; ch < nch; ++ch) {
            const int buf = ch & 1;
            if (ch + 1 < nch) DC_XFETCH(ch + 1)
            const float* xrow = xs + buf * (CK * 256) + lane * 4;
            const int jn = min(CK, L - ch * CK);
            for (int u = 0; u < jn; ++u) {
                const f4v x4 = *reinterpret_cast<const f4v*>(xrow + u * 256);
                f4v pv[C::PPWP / 4];
#pragma unroll
                for (int i = 0; i < C::PPWP / 4; ++i)
                    pv[i] = *reinterpret_cast<const f4v*>(P + (ch * CK + u) * C::NPP + wave * C::PPWP + 4 * i);
#pragma unroll
                for (int pp = 0; pp < C::PPW; ++pp) {
                    const float p = pv[pp >> 2][pp & 3];
                    const f2v p2 = {p, p};
                    ctx[pp].xy = __builtin_elementwise_fma(p2, x4.xy, ctx[pp].xy);
                    ctx[pp].zw = __builtin_elementwise_fma(p2, x4.zw, ctx[pp].zw);
                }
            }
            if (ch + 1 < nch) DC_XSTASH(buf ^ 1)             // last read of that buffer was before the previous barrier
            __syncthreads();
        }
#undef DC_XFETCH
#undef DC_XSTASH
    }
    float* ctxs = qkf;                                       // the chunk buffers are dead after the loop's last barrier
#pragma unroll
    for (int pp = 0; pp < C::PPW; ++pp)
        *reinterpret_cast<f4v*>(ctxs + (wave * C::PPW + pp) * 256 + lane * 4) = ctx[pp];
    __syncthreads();

    // ---- stage D: out[s][o] = sum_c WvT[c][o] * ctx[(s, o/32)][c] + bv[o], thread = output column o; the two thread
    // groups split the c range (each W_v^T element is loaded once per window), partial sums meet in LDS
    {
        f2v o[NQ];
#pragma unroll
        for (int s = 0; s < NQ; ++s) o[s] = f2v{0.f, 0.f};
        const int h = col >> 5;
        const float* wcol = WvT + (size_t)half * 128 * 256 + col;
        const float* crow = ctxs + h * 256 + half * 128;
#pragma unroll 4
        for (int c4 = 0; c4 < 32; ++c4) {
            const f2v w01 = {wcol[(c4 * 4 + 0) * 256], wcol[(c4 * 4 + 1) * 256]};
            const f2v w23 = {wcol[(c4 * 4 + 2) * 256], wcol[(c4 * 4 + 3) * 256]};
#pragma unroll
            for (int s = 0; s < NQ; ++s) {
                const f4v cx = *reinterpret_cast<const f4v*>(crow + s * 8 * 256 + c4 * 4);
                o[s] = __builtin_elementwise_fma(cx.xy, w01, o[s]);
                o[s] = __builtin_elementwise_fma(cx.zw, w23, o[s]);
            }
        }
        float* red = tile;                                   // P image is dead since the end of stage C
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

template <int NQ, int NKL, bool POSTAB>
static int launch_one(const float* DQ, const float* XP, const float* X, const float* pos_rows, const int* vlen,
                      const int* off, const float* Wk, const float* WvT, const float* bv, float* OUT, int B,
                      hipStream_t s) {
    using C = DecCrossCfg<NQ, NKL>;
    static DeviceOnce once;     // the opt-in to > 64 KiB of LDS: once per device
    CONE_CHECK_HIP(device_once(once, [] {
        return hipFuncSetAttribute((const void*)dec_cross_kernel<NQ, NKL, POSTAB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   C::LDS_FLOATS * 4);
    }));
    hipLaunchKernelGGL((dec_cross_kernel<NQ, NKL, POSTAB>), dim3(B), dim3(C::NT), C::LDS_FLOATS * 4, s, DQ, XP, X,
                       pos_rows, vlen, off, Wk, WvT, bv, OUT);
    CONE_LAUNCH_CHECK();
    return 0;
}

bool dec_cross_supported(int nq, int Lmax) { return nq == 5 && Lmax <= 192; }

int launch_dec_cross(const float* DQ, const float* XP, const float* X, const float* pos_rows, const int* vlen,
                     const int* off, const float* Wk, const float* WvT, const float* bv, float* OUT, int B, int nq,
                     int Lmax, hipStream_t s) {
    CONE_REQUIRE(dec_cross_supported(nq, Lmax), "fused decoder cross-attention: nq=%d Lmax=%d unsupported", nq, Lmax);
    CONE_REQUIRE(XP || (pos_rows && vlen), "fused decoder cross-attention: needs memory+pos rows or the sine table");
    if (B <= 0) return 0;
    ProfScope ps(PK_DEC_CROSS, B, Lmax, nq, nullptr, s);
    if (XP) {
        if (Lmax <= 128) return launch_one<5, 2, false>(DQ, XP, X, nullptr, nullptr, off, Wk, WvT, bv, OUT, B, s);
        return launch_one<5, 3, false>(DQ, XP, X, nullptr, nullptr, off, Wk, WvT, bv, OUT, B, s);
    }
    if (Lmax <= 128) return launch_one<5, 2, true>(DQ, nullptr, X, pos_rows, vlen, off, Wk, WvT, bv, OUT, B, s);
    return launch_one<5, 3, true>(DQ, nullptr, X, pos_rows, vlen, off, Wk, WvT, bv, OUT, B, s);
}

}  // namespace cone
