// WIDE form of the fused layer tail (ffn.hip) for launches of a few row groups:
//     x1  = LayerNorm_p(R + A Wo^T + bo)                       (PROJ; else x1 = X)
//     out = LayerNorm(x1 + W2 relu(W1 x1 + b1) + b2)           (cone/transformer.py:239-245, 308-316)
//
// In ffn.hip a wave owns 16 token rows for the WHOLE block: 9 216 exact-fp32 MFMAs in a row, 0.18 ms even with a SIMD to
// itself -- whatever the number of rows.  A launch of a few hundred rows (the four layer tails of the single-query path,
// the decoder tails of a 64-query video or of an 8-GPU rank's share) is bound by that serial chain while most CUs idle.
// Here ONE workgroup of 8 waves owns 16 rows and the waves split the block's OUTPUT elements:
//   projection : wave w computes channels [32 w, 32 w + 32) of A Wo^T  (pair g = w of ffn.hip),
//   GEMM1      : wave w computes the hidden chunks c = w, w + 8, ...   (16 hidden units each),
//   GEMM2      : wave w computes output tiles 2 w, 2 w + 1             (32 channels) over ALL hidden chunks,
// with the projected row (16 x 256) and the hidden tile (16 x ff, in the accumulator = B-operand layout) passing through
// LDS.  Every output element is accumulated by the SAME fma chain as in ffn.hip (same MFMA k-slot assignment, same order
// of steps, same partial chains and the same order of their final additions; the LayerNorm moments by the same routine on
// the same register layout), so the result is bit-identical to the 8-wave / 4-wave forms: a row's result does not depend
// on which form -- i.e. on how many rows -- it was computed with.  Weight fragments come straight from global memory /
// L2 in the operand layout (row = lane % 16, four consecutive k at 16 q + 4 (lane / 16)), the next unit's requested ahead.
// 1 152 MFMAs per wave instead of 9 216: ~40 us per launch of up to one workgroup per CU.
#include "common.h"

namespace cone {

typedef float f32x4w __attribute__((ext_vector_type(4)));

struct FfnWideArgs {
    const float* X; int ldx;
    const float* A; int lda; const float* R; int ldr; const int* r_idx; const float* R2;
    const float* Wo; const float* bo; const float* pg; const float* pb;
    const float* W1; const float* b1; const float* W2; const float* b2; const float* ln_g; const float* ln_b;
    float* OUT; int ldo; int M; const int* M_dev; int ff;
};

constexpr int FW_XLD = 260;     // row stride (floats) of the 16 x 256 exchange tile

// the same moments as ffn.hip's ffn_layernorm_regs: a token's 256 channels as v[16] (channel 16 t + 4 lg + r in v[t][r])
__device__ __forceinline__ void fw_layernorm_regs(f32x4w (&v)[16], float& rstd) {
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s1 += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 * (1.0f / 256.0f);
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[t][r] -= mean; s2 = fmaf(v[t][r], v[t][r], s2); }
    }
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    rstd = 1.0f / sqrtf(s2 * (1.0f / 256.0f) + 1e-5f);
}

#define FW_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0)

template <bool PROJ>
__global__ __launch_bounds__(512, 2) void ffn_wide_kernel(FfnWideArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* XS = smem;                               // [16 tokens][FW_XLD]: projected rows, later the block's output rows
    float* HS = smem + 16 * FW_XLD;                 // [ff / 16 chunks][64 lanes][4]: hidden tiles in accumulator layout
    int M = p.M;
    if (p.M_dev) { const int md = *p.M_dev; M = md < M ? md : M; }
    const int row0 = blockIdx.x * 16;
    if (row0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int ff = p.ff, nc = ff >> 4;
    const int my_row = row0 + li;
    const size_t ld_row = (size_t)(my_row < M ? my_row : M - 1);       // rows past M feed unstored outputs

    // ---- the block input x1 (every wave holds the 16 rows: xr[q][r] = row[token li][16 q + 4 lg + r])
    f32x4w xr[16];
    if (PROJ) {
        f32x4w ar[16];
        const float* ap = p.A + ld_row * p.lda + 4 * lg;
#pragma unroll
        for (int q = 0; q < 16; ++q) ar[q] = *reinterpret_cast<const f32x4w*>(ap + 16 * q);
        const float* rp = p.R + ld_row * p.ldr + 4 * lg;
        if (p.r_idx) {
            const int ix = p.r_idx[ld_row];
            rp = (ix >= 0 ? p.R + (size_t)ix * p.ldr : p.R2 + (size_t)(~ix) * p.ldr) + 4 * lg;
        }
        // pair g = wave: channels [32 g, 32 g + 32) of A Wo^T, two partial chains per tile as in ffn.hip.  All 32 weight
        // fragments are requested at once (a wave that waits for one L2 round trip per fragment spends its time waiting)
        const float* w_lo = p.Wo + (size_t)(32 * wave + li) * 256 + 4 * lg;          // rows 32 g + li
        const float* w_hi = w_lo + 16 * 256;                                          // rows 32 g + 16 + li
        f32x4w wl[16], wh[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            wl[q] = *reinterpret_cast<const f32x4w*>(w_lo + 16 * q);
            wh[q] = *reinterpret_cast<const f32x4w*>(w_hi + 16 * q);
        }
        const f32x4w r0 = *reinterpret_cast<const f32x4w*>(rp + 32 * wave);          // the residual of this wave's two tiles
        const f32x4w r1 = *reinterpret_cast<const f32x4w*>(rp + 32 * wave + 16);
        __builtin_amdgcn_sched_barrier(0);                  // the whole burst before the first MFMA
        f32x4w ha[2], hb[2];
        ha[0] = f32x4w{0.f, 0.f, 0.f, 0.f}; ha[1] = ha[0]; hb[0] = ha[0]; hb[1] = ha[0];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                FW_MFMA(ha[r & 1], wl[q][r], ar[q][r]);
                FW_MFMA(hb[r & 1], wh[q][r], ar[q][r]);
            }
        }
        // residual + projection of this wave's two tiles -> LDS; register r of lane (li, lg) = channel 16 t + 4 lg + r
        const f32x4w x0 = r0 + (ha[0] + ha[1]);
        const f32x4w x1 = r1 + (hb[0] + hb[1]);
        *reinterpret_cast<f32x4w*>(XS + li * FW_XLD + 32 * wave + 4 * lg) = x0;
        *reinterpret_cast<f32x4w*>(XS + li * FW_XLD + 32 * wave + 16 + 4 * lg) = x1;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 16; ++t)
            xr[t] = *reinterpret_cast<const f32x4w*>(XS + li * FW_XLD + 16 * t + 4 * lg) +
                    *reinterpret_cast<const f32x4w*>(p.bo + 16 * t + 4 * lg);
        float rstd;
        fw_layernorm_regs(xr, rstd);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f32x4w g4 = *reinterpret_cast<const f32x4w*>(p.pg + 16 * t + 4 * lg);
            const f32x4w b4 = *reinterpret_cast<const f32x4w*>(p.pb + 16 * t + 4 * lg);
#pragma unroll
            for (int r = 0; r < 4; ++r) xr[t][r] = xr[t][r] * rstd * g4[r] + b4[r];
        }
    } else {
        const float* xp = p.X + ld_row * p.ldx + 4 * lg;
#pragma unroll
        for (int q = 0; q < 16; ++q) xr[q] = *reinterpret_cast<const f32x4w*>(xp + 16 * q);
    }

    // ---- GEMM1: hidden chunks c = wave, wave + 8, ...: four partial chains over the 16 k-slabs, as ffn.hip's FFN_MM_A.
    // The 16 weight fragments of the NEXT chunk are in flight while this chunk's 64 MFMAs run.
    f32x4w r0 = xr[0], r1 = xr[1];                          // the block input of this wave's two output tiles (residual)
#pragma unroll
    for (int g = 1; g < 8; ++g)
        if (wave == g) { r0 = xr[2 * g]; r1 = xr[2 * g + 1]; }
    {
        f32x4w wA[16], wB[16], bA, bB;                      // ping-pong: no register copies, exact wait counts
        auto load_chunk = [&](f32x4w (&w)[16], f32x4w& bias, int c) {
            const bool live = c < nc;                       // past the end: every lane re-reads W1's first bytes (unused)
            const float* w1p = live ? p.W1 + (size_t)(16 * c + li) * 256 + 4 * lg : p.W1;
#pragma unroll
            for (int q = 0; q < 16; ++q) w[q] = *reinterpret_cast<const f32x4w*>(w1p + 16 * q);
            bias = *reinterpret_cast<const f32x4w*>(live ? p.b1 + 16 * c + 4 * lg : p.b1);
            __builtin_amdgcn_sched_barrier(0);              // keep the burst ahead of the MFMAs it overlaps
        };
        auto mm_chunk = [&](const f32x4w (&w)[16], const f32x4w& bias, int c) {
            f32x4w hp[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) hp[r] = f32x4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 16; ++q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) FW_MFMA(hp[r], w[q][r], xr[q][r]);
            }
            f32x4w h = (hp[0] + hp[1]) + (hp[2] + hp[3]) + bias;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = fmaxf(h[r], 0.f);
            *reinterpret_cast<f32x4w*>(HS + (size_t)c * 256 + lane * 4) = h;    // k slot lg of step r <-> hidden unit 16 c + 4 lg + r
            __builtin_amdgcn_sched_barrier(0);
        };
        // the look-ahead load is unconditional (past the end it reads one dummy line): behind a branch it makes the
        // compiler wait for ALL loads in flight before the first MFMA of the chunk
        load_chunk(wA, bA, wave);
        for (int c = wave; c < nc; c += 16) {
            load_chunk(wB, bB, c + 8);
            mm_chunk(wA, bA, c);
            if (c + 8 >= nc) break;
            load_chunk(wA, bA, c + 16);
            mm_chunk(wB, bB, c + 8);
        }
    }

    // ---- GEMM2: output tiles 2 w, 2 w + 1 over all hidden chunks in order, as ffn.hip's FFN_MM_Y; weight fragments by
    // groups of 8 chunks, the next group in flight (the first one across the barrier)
    const int t0 = 2 * wave;
    f32x4w y0 = f32x4w{0.f, 0.f, 0.f, 0.f}, y1 = y0;
    {
        const float* w2a = p.W2 + (size_t)(16 * t0 + li) * ff + 4 * lg;
        const float* w2b = w2a + (size_t)16 * ff;
        f32x4w uA[8], vA[8], uB[8], vB[8];
        auto load_group = [&](f32x4w (&u)[8], f32x4w (&v)[8], int c0) {
            const float* pa = c0 < nc ? w2a + 16 * c0 : p.W2;
            const float* pb = c0 < nc ? w2b + 16 * c0 : p.W2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                u[j] = *reinterpret_cast<const f32x4w*>(pa + 16 * j);
                v[j] = *reinterpret_cast<const f32x4w*>(pb + 16 * j);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mm_group = [&](const f32x4w (&u)[8], const f32x4w (&v)[8], int c0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4w hh = *reinterpret_cast<const f32x4w*>(HS + (size_t)(c0 + j) * 256 + lane * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    FW_MFMA(y0, u[j][r], hh[r]);
                    FW_MFMA(y1, v[j][r], hh[r]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        load_group(uA, vA, 0);
        __syncthreads();                                    // the hidden tile is complete
        for (int c0 = 0; c0 < nc; c0 += 16) {
            load_group(uB, vB, c0 + 8);
            mm_group(uA, vA, c0);
            if (c0 + 8 >= nc) break;
            load_group(uA, vA, c0 + 16);
            mm_group(uB, vB, c0 + 8);
        }
    }
    // ---- epilogue: + b2 + residual of this wave's tiles -> LDS, full rows back, LayerNorm (same routine, same layout)
    y0 = y0 + *reinterpret_cast<const f32x4w*>(p.b2 + 16 * t0 + 4 * lg) + r0;
    y1 = y1 + *reinterpret_cast<const f32x4w*>(p.b2 + 16 * (t0 + 1) + 4 * lg) + r1;
    __syncthreads();                                    // (PROJ: every wave is done reading the projected rows)
    *reinterpret_cast<f32x4w*>(XS + li * FW_XLD + 16 * t0 + 4 * lg) = y0;
    *reinterpret_cast<f32x4w*>(XS + li * FW_XLD + 16 * (t0 + 1) + 4 * lg) = y1;
    __syncthreads();
    f32x4w y[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) y[t] = *reinterpret_cast<const f32x4w*>(XS + li * FW_XLD + 16 * t + 4 * lg);
    float rstd;
    fw_layernorm_regs(y, rstd);
    if (my_row < M) {
        float* op = p.OUT + (size_t)my_row * p.ldo + 4 * lg;
#pragma unroll
        for (int t = 0; t < 16; ++t) {                  // each wave stores the two tiles it computed
            if ((t >> 1) == wave) {
                const f32x4w g = *reinterpret_cast<const f32x4w*>(p.ln_g + 16 * t + 4 * lg);
                const f32x4w be = *reinterpret_cast<const f32x4w*>(p.ln_b + 16 * t + 4 * lg);
                f32x4w o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = y[t][r] * rstd * g[r] + be[r];
                *reinterpret_cast<f32x4w*>(op + 16 * t) = o;
            }
        }
    }
}

#undef FW_MFMA

bool ffn_wide_supported(int ff) { return ff >= 128 && ff % 128 == 0 && (size_t)(16 * FW_XLD + ff * 16) * sizeof(float) <= 160 * 1024; }

template <bool PROJ>
static int launch_wide_t(const FfnWideArgs& a, hipStream_t s) {
    const size_t lds = (size_t)(16 * FW_XLD + a.ff * 16) * sizeof(float);
    static DeviceOnce once;
    CONE_CHECK_HIP(device_once(once, [] {
        return hipFuncSetAttribute((const void*)ffn_wide_kernel<PROJ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }));
    ProfScope ps(PROJ ? PK_FFN_PROJ : PK_FFN_FUSED, a.M, a.ff, 256, a.M_dev, s);
    hipLaunchKernelGGL((ffn_wide_kernel<PROJ>), dim3((unsigned)((a.M + 15) / 16)), dim3(512), lds, s, a);
    CONE_LAUNCH_CHECK();
    return 0;
}

int launch_ffn_wide(const float* X, int ldx, const float* W1, const float* b1, const float* W2, const float* b2,
                    const float* ln_g, const float* ln_b, float* OUT, int ldo, int M, const int* M_dev, int ff, hipStream_t s) {
    FfnWideArgs a{};
    a.X = X; a.ldx = ldx; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.ln_g = ln_g; a.ln_b = ln_b;
    a.OUT = OUT; a.ldo = ldo; a.M = M; a.M_dev = M_dev; a.ff = ff;
    return launch_wide_t<false>(a, s);
}

int launch_proj_ffn_wide(const float* A, int lda, const float* Wo, const float* bo, const float* R, int ldr, const float* pg,
                         const float* pb, const float* W1, const float* b1, const float* W2, const float* b2,
                         const float* ln_g, const float* ln_b, float* OUT, int ldo, int M, const int* M_dev, int ff,
                         hipStream_t s, const int* r_idx, const float* R2) {
    FfnWideArgs a{};
    a.A = A; a.lda = lda; a.Wo = Wo; a.bo = bo; a.R = R; a.ldr = ldr; a.pg = pg; a.pb = pb; a.r_idx = r_idx; a.R2 = R2;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.ln_g = ln_g; a.ln_b = ln_b;
    a.OUT = OUT; a.ldo = ldo; a.M = M; a.M_dev = M_dev; a.ff = ff;
    return launch_wide_t<true>(a, s);
}

}  // namespace cone
