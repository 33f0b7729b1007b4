// WIDE form of the fused layer tail (ffn.hip) for launches of a few row groups:
//     x1  = LayerNorm_p(R + A Wo^T + bo)                       (PROJ; else x1 = X)
//     out = LayerNorm(x1 + W2 relu(W1 x1 + b1) + b2)           (cone/transformer.py:239-245, 308-316)
//
// In ffn.hip a wave owns 16 token rows for the WHOLE block: 9 216 exact-fp32 MFMAs in a row, 0.15 ms even with a SIMD to
// itself -- whatever the number of rows.  A launch of a few hundred rows (the four layer tails of the single-query path,
// the decoder tails of a small batch) is bound by that serial chain while most CUs idle.
// Here ONE workgroup of 8 waves owns 16 rows and the waves split the block's OUTPUT elements:
//   projection : wave w computes channels [32 w, 32 w + 32) of A Wo^T  (pair g = w of ffn.hip),
//   GEMM1      : wave w computes the hidden chunks c = w, w + 8, ...   (16 hidden units each),
//   GEMM2      : wave w computes output tiles 2 w, 2 w + 1             (32 channels) over ALL hidden chunks,
// with the projected row (16 x 256) and the hidden tile (16 x ff, in the accumulator = B-operand layout) passing through
// LDS.  Every output element is accumulated by the SAME fma chain as in ffn.hip (same MFMA k-slot assignment, same order
// of steps, same partial chains and the same order of their final additions; the LayerNorm moments by the same routine on
// the same register layout), so the result is bit-identical to the 8-wave / 4-wave forms: a row's result does not depend
// on which form -- i.e. on how many rows -- it was computed with.
//
// Weight stream.  Every wave needs its OWN weights (no reuse inside the workgroup), as operand slabs [16 rows][16 floats]:
// 32 for the projection, 16 per hidden chunk, 2 per hidden chunk in GEMM2.  A slab is fetched by ONE LDS-DMA instruction
// (global_load_lds_dwordx4: lane -> row = lane / 4, 16-B chunk = lane % 4, i.e. 64 contiguous bytes per 4 adjacent lanes)
// into the wave's private 8-slab ring and read back as the MFMA operand (row = lane % 16, chunk = lane / 16) with
// gemm.hip's chunk XOR swizzle (conflict-free).  [Measured: the same bytes loaded straight into the operand layout --
// 16 B per lane, adjacent lanes 1 KiB apart -- cost 9 us per GEMM on top of 18 us of MFMA; one coalesced instruction per
// slab 1.3 us.]  The stream is a sequence of BLOCKS of 8 slab pairs -- projection (2 blocks), then per pass of 8 hidden chunks
// one GEMM1 block (this wave's chunk of the pass) and one GEMM2 block (its two output tiles over the pass's 8 chunks) -- through
// a 16-slab ring: pair u of every block lives in slots 2u, 2u + 1; step u multiplies pair u (already in registers), reads pair
// u + 1 from the ring and refills pair u's slots with pair u of the NEXT block: 14 KiB in flight per wave (round 5; an 8-slab
// ring with 6 KiB in flight left the wave waiting on the L2 for 23 of 54 us), one counted s_waitcnt vmcnt per pair.  The
// hidden tile passes through LDS one pass (8 chunks, 8 KiB) at a time -- GEMM2 accumulates the chunks in the same order as
// before, so the bits do not change -- which is what frees the LDS for the deeper ring.
#include "common.h"

namespace cone {

typedef float f32x4w __attribute__((ext_vector_type(4)));

struct FfnWideArgs {
    const float* X; int ldx;
    const float* A; int lda; const float* R; int ldr; const int* r_idx; const float* R2;
    const float* Wo; const float* bo; const float* pg; const float* pb;
    const float* W1; const float* b1; const float* W2; const float* b2; const float* ln_g; const float* ln_b;
    float* OUT; int ldo; int M; const int* M_dev; int ff; int m_off;
    float* OUT2; int ldo2;      // PRE (pre-norm tail): OUT = the un-normalised stream, OUT2 (may be null) = LayerNorm(OUT; ln_g, ln_b)
};

constexpr int FW_XLD = 260;     // row stride (floats) of the 16 x 256 exchange tile
constexpr int FW_RING = 16 * 256;   // floats per wave: 16 slabs of [16 rows][16 floats] = 8 slab pairs = one block
constexpr int FW_PASS = 8;          // hidden chunks per pass (one per wave in GEMM1)

#define FW_GLDS16(src, dst) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

__device__ __forceinline__ int fw_swz16(int row) { return (0x1230 >> (((row >> 2) & 3) * 4)) & 3; }

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
#define FW_SB() __builtin_amdgcn_sched_barrier(0)
// workgroup barrier for data written with ds_write: __syncthreads() would also drain the LDS-DMA ring (its fence waits
// for every LDS write in flight, vmcnt(0)); the ring slots are wave-private and ordered by the counted waits
#define FW_BARRIER()                                           \
    {                                                          \
        FW_SB();                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                          \
        FW_SB();                                               \
    }
// step u of a block: the NEXT pair (pair (u + 1) % 8: of this block, or pair 0 of the next one) has landed -- the six pairs
// issued after it may still be in flight -- and goes to registers
#define FW_STEP_BEGIN(u)                                       \
    {                                                          \
        FW_SB();                                               \
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      \
        n0 = *reinterpret_cast<const f32x4w*>(ring + (2 * (((u) + 1) & 7)) * 256 + rdo);     \
        n1 = *reinterpret_cast<const f32x4w*>(ring + (2 * (((u) + 1) & 7) + 1) * 256 + rdo); \
        FW_SB();                                               \
    }
// pair u's slots (read one step ago, consumed by the MFMAs above) take pair u of the next block
#define FW_STEP_END(src0, src1, u)                             \
    {                                                          \
        FW_SB();                                               \
        FW_GLDS16(src0, ring + (2 * (u)) * 256);               \
        FW_GLDS16(src1, ring + (2 * (u) + 1) * 256);           \
        c0 = n0; c1 = n1;                                      \
        FW_SB();                                               \
    }

// PRE (PROJ only; --pre_norm, as ffn.hip's PRE): the block input is LayerNorm_p(x1) of the un-normalised stream x1 = R + A Wo^T + bo,
// the output accumulators START from x1 + b2, OUT = the stream, OUT2 = LayerNorm(OUT) for the next consumer.
template <bool PROJ, bool PRE = false>
__global__ __launch_bounds__(512, 2) void ffn_wide_kernel(FfnWideArgs p) {
    static_assert(!PRE || PROJ, "the pre-norm form is the projecting tail");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ff = p.ff, nc = ff >> 4;
    float* XS = smem;                               // [16 tokens][FW_XLD]: projected rows, later the block's output rows
    float* HS = XS + 16 * FW_XLD;                   // [8 chunks of the pass][64 lanes][4]: hidden tiles in accumulator layout
    float* B1 = HS + FW_PASS * 256;                 // b1 (no ordinary global load inside the DMA-counted loops)
    int M = p.M;
    if (p.M_dev) { const int md = *p.M_dev - p.m_off; M = md < M ? md : M; }
    const int row0 = blockIdx.x * 16;
    if (row0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    float* ring = B1 + ff + wave * FW_RING;         // this wave's slab ring
    const int my_row = row0 + li;
    const size_t ld_row = (size_t)(my_row < M ? my_row : M - 1);       // rows past M feed unstored outputs
    const int npass = nc / FW_PASS;                 // passes of 8 hidden chunks (one GEMM1 chunk per wave and pass)

    // slab sources: uniform base + lane offset (row = lane / 4, source chunk = the one that lands in physical chunk lane % 4)
    const int drow = lane >> 2;
    const int dq = ((lane & 3) ^ fw_swz16(drow)) << 2;
    const int off256 = drow * 256 + dq;             // Wo, W1 (row stride 256)
    const int offff = drow * ff + dq;               // W2 (row stride ff)
    const int rdo = li * 16 + ((lg ^ fw_swz16(li)) << 2);       // operand read: row li, chunk lg
    const float* baseA = p.Wo + (size_t)(32 * wave) * 256;                                 // + 16 t * 256 + 16 q
    const float* baseB = p.W1 + (size_t)(16 * wave) * 256;                                 // + 128 i * 256 + 16 q
    const float* baseC = p.W2 + (size_t)(32 * wave) * ff;                                  // + 16 t * ff + 16 c
    auto srcA = [&](int q, int t) { return baseA + (t * 16 * 256 + 16 * q) + off256; };
    auto srcB = [&](int i, int q) { return baseB + ((size_t)i * (128 * 256) + 16 * q) + off256; };
    auto srcC = [&](int c, int t) { return baseC + ((size_t)t * 16 * ff + 16 * c) + offff; };

    for (int i = tid; i < ff; i += 512) B1[i] = p.b1[i];

    f32x4w c0, c1, n0, n1;                          // the current / next slab pair as MFMA operands
    // ---- the block input x1 (every wave holds the 16 rows: xr[q][r] = row[token li][16 q + 4 lg + r])
    f32x4w xr[16];
    f32x4w pre_b2a = f32x4w{0.f, 0.f, 0.f, 0.f}, pre_b2b = pre_b2a, pre_y0 = pre_b2a, pre_y1 = pre_b2a;
    if (PROJ) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { FW_GLDS16(srcA(j, 0), ring + (2 * j) * 256); FW_GLDS16(srcA(j, 1), ring + (2 * j + 1) * 256); }
        FW_SB();
        f32x4w ar[16];
        const float* ap = p.A + ld_row * p.lda + 4 * lg;
#pragma unroll
        for (int q = 0; q < 16; ++q) ar[q] = *reinterpret_cast<const f32x4w*>(ap + 16 * q);
        const float* rp = p.R + ld_row * p.ldr + 4 * lg;
        if (p.r_idx) {
            const int ix = p.r_idx[ld_row];
            rp = (ix >= 0 ? p.R + (size_t)ix * p.ldr : p.R2 + (size_t)(~ix) * p.ldr) + 4 * lg;
        }
        const f32x4w r0 = *reinterpret_cast<const f32x4w*>(rp + 32 * wave);          // the residual of this wave's two tiles
        const f32x4w r1 = *reinterpret_cast<const f32x4w*>(rp + 32 * wave + 16);
        if (PRE) {                                          // b2 of the two tiles: the start of their output accumulators
            pre_b2a = *reinterpret_cast<const f32x4w*>(p.b2 + 32 * wave + 4 * lg);
            pre_b2b = *reinterpret_cast<const f32x4w*>(p.b2 + 32 * wave + 16 + 4 * lg);
        }
        FW_SB();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the younger row loads are needed by the first MFMA anyway)
        c0 = *reinterpret_cast<const f32x4w*>(ring + rdo);
        c1 = *reinterpret_cast<const f32x4w*>(ring + 256 + rdo);
        // pair g = wave: channels [32 g, 32 g + 32) of A Wo^T, two partial chains per tile as in ffn.hip
        f32x4w ha[2], hb[2];
        ha[0] = f32x4w{0.f, 0.f, 0.f, 0.f}; ha[1] = ha[0]; hb[0] = ha[0]; hb[1] = ha[0];
#pragma unroll
        for (int q = 0; q < 16; ++q) {              // two blocks of 8 pairs
            FW_STEP_BEGIN(q & 7);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                FW_MFMA(ha[r & 1], c0[r], ar[q][r]);
                FW_MFMA(hb[r & 1], c1[r], ar[q][r]);
            }
            if (q < 8) { FW_STEP_END(srcA(q + 8, 0), srcA(q + 8, 1), q & 7); }
            else { FW_STEP_END(srcB(0, 2 * (q - 8)), srcB(0, 2 * (q - 8) + 1), q & 7); }      // GEMM1 block of pass 0
        }
        // residual + projection of this wave's two tiles -> LDS; register r of lane (li, lg) = channel 16 t + 4 lg + r
        const f32x4w x0 = r0 + (ha[0] + ha[1]);
        const f32x4w x1 = r1 + (hb[0] + hb[1]);
        *reinterpret_cast<f32x4w*>(XS + li * FW_XLD + 32 * wave + 4 * lg) = x0;
        *reinterpret_cast<f32x4w*>(XS + li * FW_XLD + 32 * wave + 16 + 4 * lg) = x1;
        FW_BARRIER();                                       // (also: b1 is in LDS)
#pragma unroll
        for (int t = 0; t < 16; ++t)
            xr[t] = *reinterpret_cast<const f32x4w*>(XS + li * FW_XLD + 16 * t + 4 * lg) +
                    *reinterpret_cast<const f32x4w*>(p.bo + 16 * t + 4 * lg);
        if (PRE) {      // the un-normalised stream is the residual: this wave's two output tiles start from x1 + b2
            f32x4w s0 = xr[0], s1 = xr[1];
#pragma unroll
            for (int g = 1; g < 8; ++g)
                if (wave == g) { s0 = xr[2 * g]; s1 = xr[2 * g + 1]; }
            pre_y0 = s0 + pre_b2a; pre_y1 = s1 + pre_b2b;
        }
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
#pragma unroll
        for (int j = 0; j < 8; ++j) { FW_GLDS16(srcB(0, 2 * j), ring + (2 * j) * 256); FW_GLDS16(srcB(0, 2 * j + 1), ring + (2 * j + 1) * 256); }
        FW_SB();
        const float* xp = p.X + ld_row * p.ldx + 4 * lg;
#pragma unroll
        for (int q = 0; q < 16; ++q) xr[q] = *reinterpret_cast<const f32x4w*>(xp + 16 * q);
        FW_SB();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        c0 = *reinterpret_cast<const f32x4w*>(ring + rdo);
        c1 = *reinterpret_cast<const f32x4w*>(ring + 256 + rdo);
        FW_BARRIER();                                       // b1 is in LDS
    }

    f32x4w r0 = xr[0], r1 = xr[1];                          // the block input of this wave's two output tiles (residual)
#pragma unroll
    for (int g = 1; g < 8; ++g)
        if (wave == g) { r0 = xr[2 * g]; r1 = xr[2 * g + 1]; }
    f32x4w y0 = f32x4w{0.f, 0.f, 0.f, 0.f}, y1 = y0;
    if (PRE) { y0 = pre_y0; y1 = pre_y1; }
    for (int s = 0; s < npass; ++s) {
        // ---- GEMM1 block: hidden chunk c = wave + 8 s: four partial chains over the 16 k-slabs, as ffn.hip's FFN_MM_A
        const int c = wave + FW_PASS * s;
        f32x4w hp[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) hp[r] = f32x4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            FW_STEP_BEGIN(u);
#pragma unroll
            for (int r = 0; r < 4; ++r) FW_MFMA(hp[r], c0[r], xr[2 * u][r]);
#pragma unroll
            for (int r = 0; r < 4; ++r) FW_MFMA(hp[r], c1[r], xr[2 * u + 1][r]);
            FW_STEP_END(srcC(FW_PASS * s + u, 0), srcC(FW_PASS * s + u, 1), u);       // this pass's GEMM2 block
        }
        f32x4w h = (hp[0] + hp[1]) + (hp[2] + hp[3]) + *reinterpret_cast<const f32x4w*>(B1 + 16 * c + 4 * lg);
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = fmaxf(h[r], 0.f);
        *reinterpret_cast<f32x4w*>(HS + wave * 256 + lane * 4) = h;        // k slot lg of step r <-> hidden unit 16 c + 4 lg + r
        FW_BARRIER();                                       // the pass's hidden tiles are complete
        // ---- GEMM2 block: output tiles 2 w, 2 w + 1 over the pass's chunks in order, as ffn.hip's FFN_MM_Y
        const bool more = s + 1 < npass;                    // past the end: dummy lines (keep the wait count exact)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4w hh = *reinterpret_cast<const f32x4w*>(HS + j * 256 + lane * 4);
            FW_STEP_BEGIN(j);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                FW_MFMA(y0, c0[r], hh[r]);
                FW_MFMA(y1, c1[r], hh[r]);
            }
            const float* s0 = more ? srcB(s + 1, 2 * j) : p.W2;
            const float* s1 = more ? srcB(s + 1, 2 * j + 1) : p.W2;
            FW_STEP_END(s0, s1, j);                         // the next pass's GEMM1 block
        }
        FW_BARRIER();                                       // every wave is done reading the hidden tiles
    }
    const int t0 = 2 * wave;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA may outlive the workgroup's LDS allocation
    // ---- epilogue: + b2 + residual of this wave's tiles -> LDS, full rows back, LayerNorm (same routine, same layout)
    if (!PRE) {
        y0 = y0 + *reinterpret_cast<const f32x4w*>(p.b2 + 16 * t0 + 4 * lg) + r0;
        y1 = y1 + *reinterpret_cast<const f32x4w*>(p.b2 + 16 * (t0 + 1) + 4 * lg) + r1;
    } else {
        if (my_row < M) {                               // the un-normalised stream: each wave stores the two tiles it computed
            float* sp = p.OUT + (size_t)my_row * p.ldo + 4 * lg;
            *reinterpret_cast<f32x4w*>(sp + 16 * t0) = y0;
            *reinterpret_cast<f32x4w*>(sp + 16 * (t0 + 1)) = y1;
        }
        if (!p.OUT2) return;
    }
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
        float* op = PRE ? p.OUT2 + (size_t)my_row * p.ldo2 + 4 * lg : p.OUT + (size_t)my_row * p.ldo + 4 * lg;
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

// ---------------------------------------------------------------------------------------------------------------------------
// SPREAD form of the projecting tail for a handful of row groups (the decoder tails of a small batch: 20 windows x 5 slots = 7
// groups).  In the wide form ONE CU walks a group's 9 216 MFMAs (31 us of matrix time + 20 us around it) while 249 CUs idle.
// Here the group's OUTPUT elements are spread over single-wave workgroups, every element by the same fma chain as above:
//   fs_proj_kernel : (16 tiles, groups)       channels [16 t, 16 t + 16) of R + A Wo^T          -> XP   (64 MFMAs per wave)
//   fs_g1_kernel   : (ff / 16 chunks, groups) x1 = LayerNorm_p(XP + bo) in every wave (the same routine on the same layout);
//                                              hidden chunk c = relu(W1[c] x1 + b1)              -> HG, x1 tiles -> X1
//   fs_g2_kernel   : (16 tiles, groups)       tile t of x1 + b2 + W2 HG, the chunks in order    -> YG   (4 ff / 16 MFMAs)
//   fs_ln_kernel   : (groups)                 LayerNorm of the full rows                        -> OUT
// Rows move between the launches through a 1.8 MB scratch of the caller's workspace (L2); a launch boundary is the only
// synchronisation.  Weights: a wave's slabs by LDS-DMA into its own ring, all requested at once (proj, g1) or two blocks of 16
// ahead (g2).  4 launches, ~25 us instead of 53 (measured: see DESIGN.md), chosen by the host for M <= 256 rows.
#ifndef CONE_FFN_SPREAD_GROUPS
#define CONE_FFN_SPREAD_GROUPS 64
#endif
constexpr int FS_MAX_GROUPS = CONE_FFN_SPREAD_GROUPS;

struct FfnSpreadBufs { float* XP; float* X1; float* HG; float* YG; };
// the launch's row count: the host bound, cut to the device-side count where there is one (as ffn_wide_kernel)
__device__ __forceinline__ int fs_rows(const FfnWideArgs& p) {
    int M = p.M;
    if (p.M_dev) { const int md = *p.M_dev - p.m_off; M = md < M ? md : M; }
    return M;
}

__global__ __launch_bounds__(64) void fs_proj_kernel(FfnWideArgs p, FfnSpreadBufs b) {
    __shared__ __attribute__((aligned(16))) float ring[16 * 256];
    const int t = blockIdx.x, g = blockIdx.y, lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
    const int M = fs_rows(p), my_row = g * 16 + li;
    if (g * 16 >= M) return;
    const size_t ld_row = (size_t)(my_row < M ? my_row : M - 1);
    const int drow = lane >> 2, dq = ((lane & 3) ^ fw_swz16(drow)) << 2;
    const int rdo = li * 16 + ((lg ^ fw_swz16(li)) << 2);
    const float* src = p.Wo + (size_t)(16 * t) * 256 + drow * 256 + dq;              // + 16 q: k-slab q of tile t
#pragma unroll
    for (int q = 0; q < 16; ++q) FW_GLDS16(src + 16 * q, ring + q * 256);           // the whole weight tile at once
    FW_SB();
    f32x4w ar[16];
    const float* ap = p.A + ld_row * p.lda + 4 * lg;
#pragma unroll
    for (int q = 0; q < 16; ++q) ar[q] = *reinterpret_cast<const f32x4w*>(ap + 16 * q);
    const float* rp = p.R + ld_row * p.ldr;
    if (p.r_idx) {                      // the residual rows gathered through a row index (first encoder layer, as the wide form)
        const int ix = p.r_idx[ld_row];
        rp = ix >= 0 ? p.R + (size_t)ix * p.ldr : p.R2 + (size_t)(~ix) * p.ldr;
    }
    const f32x4w r0 = *reinterpret_cast<const f32x4w*>(rp + 16 * t + 4 * lg);
    FW_SB();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4w ha[2];
    ha[0] = f32x4w{0.f, 0.f, 0.f, 0.f}; ha[1] = ha[0];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const f32x4w c = *reinterpret_cast<const f32x4w*>(ring + q * 256 + rdo);
#pragma unroll
        for (int r = 0; r < 4; ++r) FW_MFMA(ha[r & 1], c[r], ar[q][r]);
    }
    const f32x4w x0 = r0 + (ha[0] + ha[1]);
    *reinterpret_cast<f32x4w*>(b.XP + ((size_t)g * 16 + li) * 256 + 16 * t + 4 * lg) = x0;
}

template <bool PRE>
__global__ __launch_bounds__(64) void fs_g1_kernel(FfnWideArgs p, FfnSpreadBufs b) {
    __shared__ __attribute__((aligned(16))) float ring[16 * 256];
    const int c = blockIdx.x, g = blockIdx.y, lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
    if (g * 16 >= fs_rows(p)) return;
    const int drow = lane >> 2, dq = ((lane & 3) ^ fw_swz16(drow)) << 2;
    const int rdo = li * 16 + ((lg ^ fw_swz16(li)) << 2);
    const float* src = p.W1 + (size_t)(16 * c) * 256 + drow * 256 + dq;              // + 16 q: k-slab q of hidden chunk c
#pragma unroll
    for (int q = 0; q < 16; ++q) FW_GLDS16(src + 16 * q, ring + q * 256);           // the chunk's 16 k-slabs, under the LayerNorm
    FW_SB();
    f32x4w xr[16];
    const float* xp = b.XP + ((size_t)g * 16 + li) * 256 + 4 * lg;
#pragma unroll
    for (int t = 0; t < 16; ++t)
        xr[t] = *reinterpret_cast<const f32x4w*>(xp + 16 * t) + *reinterpret_cast<const f32x4w*>(p.bo + 16 * t + 4 * lg);
    if (PRE) {      // pre-norm: output tile t starts from the un-normalised stream + b2 (as ffn.hip's PRE): kept for fs_g2_kernel
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (t == c)
                *reinterpret_cast<f32x4w*>(b.X1 + ((size_t)g * 16 + li) * 256 + 16 * t + 4 * lg) =
                    xr[t] + *reinterpret_cast<const f32x4w*>(p.b2 + 16 * t + 4 * lg);
    }
    float rstd;
    fw_layernorm_regs(xr, rstd);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const f32x4w g4 = *reinterpret_cast<const f32x4w*>(p.pg + 16 * t + 4 * lg);
        const f32x4w b4 = *reinterpret_cast<const f32x4w*>(p.pb + 16 * t + 4 * lg);
#pragma unroll
        for (int r = 0; r < 4; ++r) xr[t][r] = xr[t][r] * rstd * g4[r] + b4[r];
    }
    if (!PRE) {
#pragma unroll
        for (int t = 0; t < 16; ++t)    // the block input of output tile t (its residual), kept for fs_g2_kernel
            if (t == c) *reinterpret_cast<f32x4w*>(b.X1 + ((size_t)g * 16 + li) * 256 + 16 * t + 4 * lg) = xr[t];
    }
    const f32x4w b1v = *reinterpret_cast<const f32x4w*>(p.b1 + 16 * c + 4 * lg);
    FW_SB();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4w hp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) hp[r] = f32x4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const f32x4w c0 = *reinterpret_cast<const f32x4w*>(ring + (2 * u) * 256 + rdo);
        const f32x4w c1 = *reinterpret_cast<const f32x4w*>(ring + (2 * u + 1) * 256 + rdo);
#pragma unroll
        for (int r = 0; r < 4; ++r) FW_MFMA(hp[r], c0[r], xr[2 * u][r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) FW_MFMA(hp[r], c1[r], xr[2 * u + 1][r]);
    }
    f32x4w h = (hp[0] + hp[1]) + (hp[2] + hp[3]) + b1v;
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = fmaxf(h[r], 0.f);
    *reinterpret_cast<f32x4w*>(b.HG + ((size_t)g * (p.ff >> 4) + c) * 256 + lane * 4) = h;
}

template <bool PRE>
__global__ __launch_bounds__(64) void fs_g2_kernel(FfnWideArgs p, FfnSpreadBufs b) {
    __shared__ __attribute__((aligned(16))) float ring[32 * 256];       // two blocks of 16 weight slabs
    const int ff = p.ff, nc = ff >> 4, nb = nc >> 4;
    const int t = blockIdx.x, g = blockIdx.y, lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
    if (g * 16 >= fs_rows(p)) return;
    const int drow = lane >> 2, dq = ((lane & 3) ^ fw_swz16(drow)) << 2;
    const int rdo = li * 16 + ((lg ^ fw_swz16(li)) << 2);
    const float* src = p.W2 + (size_t)(16 * t) * ff + drow * ff + dq;                // + 16 c: hidden chunk c of tile t
    const float* hg = b.HG + (size_t)g * nc * 256 + lane * 4;                        // + 256 c: chunk c in the accumulator layout
    const f32x4w b2v = *reinterpret_cast<const f32x4w*>(p.b2 + 16 * t + 4 * lg);
    const f32x4w r0 = *reinterpret_cast<const f32x4w*>(b.X1 + ((size_t)g * 16 + li) * 256 + 16 * t + 4 * lg);
    // Issue order (vector-memory results return in order): hidden tiles of block 0, weight blocks 0 and 1; then per block i:
    // hidden tiles of block i + 1, [the block's MFMAs], weight block i + 2.  At the top of block i everything but the 16 slabs
    // of weight block i + 1 has to be there: one counted wait.
    f32x4w hh0[16], hh1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) hh0[j] = *reinterpret_cast<const f32x4w*>(hg + j * 256);
    FW_SB();
#pragma unroll
    for (int j = 0; j < 16; ++j) FW_GLDS16(src + 16 * j, ring + j * 256);
    if (nb > 1) {
#pragma unroll
        for (int j = 0; j < 16; ++j) FW_GLDS16(src + 16 * (16 + j), ring + (16 + j) * 256);
    }
    f32x4w y = f32x4w{0.f, 0.f, 0.f, 0.f};
    if (PRE) y = r0;                    // (pre-norm: X1 holds the stream + b2, the accumulator's start)
    // (two blocks per trip so that the hidden tiles' two register sets are indexed statically)
#define FS_G2_BLOCK(HC, HN, CUR, blk)                                                                                     \
    {                                                                                                                     \
        FW_SB();                                                                                                          \
        if ((blk) + 1 < nb) { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }                                         \
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }                                                         \
        FW_SB();                                                                                                          \
        if ((blk) + 1 < nb) {                                                                                             \
            _Pragma("unroll") for (int j = 0; j < 16; ++j)                                                                \
                HN[j] = *reinterpret_cast<const f32x4w*>(hg + (16 * ((blk) + 1) + j) * 256);                              \
        }                                                                                                                 \
        FW_SB();                                                                                                          \
        const float* rb = ring + (CUR) * 16 * 256;                                                                        \
        _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                                  \
            const f32x4w cw = *reinterpret_cast<const f32x4w*>(rb + j * 256 + rdo);                                       \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) FW_MFMA(y, cw[r], HC[j][r]);                                    \
        }                                                                                                                 \
        if ((blk) + 2 < nb) {                                                                                             \
            FW_SB();                                                                                                      \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                /* the ring half has been read */           \
            _Pragma("unroll") for (int j = 0; j < 16; ++j)                                                                \
                FW_GLDS16(src + 16 * (16 * ((blk) + 2) + j), ring + ((CUR) * 16 + j) * 256);                              \
        }                                                                                                                 \
    }
    for (int blk = 0; blk < nb; blk += 2) {
        FS_G2_BLOCK(hh0, hh1, 0, blk);
        if (blk + 1 < nb) FS_G2_BLOCK(hh1, hh0, 1, blk + 1);
    }
#undef FS_G2_BLOCK
    if (!PRE) y = y + b2v + r0;
    *reinterpret_cast<f32x4w*>(b.YG + ((size_t)g * 16 + li) * 256 + 16 * t + 4 * lg) = y;
}

template <bool PRE>
__global__ __launch_bounds__(64) void fs_ln_kernel(FfnWideArgs p, FfnSpreadBufs b) {
    const int g = blockIdx.x, lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
    const int my_row = g * 16 + li, M = fs_rows(p);
    if (g * 16 >= M) return;
    f32x4w y[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) y[t] = *reinterpret_cast<const f32x4w*>(b.YG + ((size_t)g * 16 + li) * 256 + 16 * t + 4 * lg);
    if (PRE) {                          // the un-normalised stream; its LayerNorm only where a consumer wants it
        if (my_row < M) {
            float* sp = p.OUT + (size_t)my_row * p.ldo + 4 * lg;
#pragma unroll
            for (int t = 0; t < 16; ++t) *reinterpret_cast<f32x4w*>(sp + 16 * t) = y[t];
        }
        if (!p.OUT2) return;
    }
    float rstd;
    fw_layernorm_regs(y, rstd);
    if (my_row < M) {
        float* op = PRE ? p.OUT2 + (size_t)my_row * p.ldo2 + 4 * lg : p.OUT + (size_t)my_row * p.ldo + 4 * lg;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f32x4w gg = *reinterpret_cast<const f32x4w*>(p.ln_g + 16 * t + 4 * lg);
            const f32x4w be = *reinterpret_cast<const f32x4w*>(p.ln_b + 16 * t + 4 * lg);
            f32x4w o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = y[t][r] * rstd * gg[r] + be[r];
            *reinterpret_cast<f32x4w*>(op + 16 * t) = o;
        }
    }
}

#undef FW_STEP_BEGIN
#undef FW_STEP_END
#undef FW_SB
#undef FW_BARRIER
#undef FW_MFMA

static size_t fw_lds_bytes(int ff) { return (size_t)(16 * FW_XLD + FW_PASS * 256 + ff + 8 * FW_RING) * sizeof(float); }

bool ffn_wide_supported(int ff) { return ff >= 128 && ff % 128 == 0 && fw_lds_bytes(ff) <= 160 * 1024; }

template <bool PROJ, bool PRE = false>
static int launch_wide_t(const FfnWideArgs& a, hipStream_t s) {
    const size_t lds = fw_lds_bytes(a.ff);
    static DeviceOnce once;
    CONE_CHECK_HIP(device_once(once, [] {
        return hipFuncSetAttribute((const void*)ffn_wide_kernel<PROJ, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }));
    ProfScope ps(PROJ ? PK_FFN_PROJ_WIDE : PK_FFN_WIDE, a.M, a.ff, 256, a.M_dev, s, a.m_off);
    hipLaunchKernelGGL((ffn_wide_kernel<PROJ, PRE>), dim3((unsigned)((a.M + 15) / 16)), dim3(512), lds, s, a);
    CONE_LAUNCH_CHECK();
    return 0;
}

int launch_ffn_wide(const float* X, int ldx, const float* W1, const float* b1, const float* W2, const float* b2,
                    const float* ln_g, const float* ln_b, float* OUT, int ldo, int M, const int* M_dev, int ff, hipStream_t s,
                    int m_off) {
    FfnWideArgs a{};
    a.m_off = m_off;
    a.X = X; a.ldx = ldx; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.ln_g = ln_g; a.ln_b = ln_b;
    a.OUT = OUT; a.ldo = ldo; a.M = M; a.M_dev = M_dev; a.ff = ff;
    return launch_wide_t<false>(a, s);
}

int launch_proj_ffn_wide(const float* A, int lda, const float* Wo, const float* bo, const float* R, int ldr, const float* pg,
                         const float* pb, const float* W1, const float* b1, const float* W2, const float* b2,
                         const float* ln_g, const float* ln_b, float* OUT, int ldo, int M, const int* M_dev, int ff,
                         hipStream_t s, const int* r_idx, const float* R2, int m_off) {
    FfnWideArgs a{};
    a.m_off = m_off;
    a.A = A; a.lda = lda; a.Wo = Wo; a.bo = bo; a.R = R; a.ldr = ldr; a.pg = pg; a.pb = pb; a.r_idx = r_idx; a.R2 = R2;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.ln_g = ln_g; a.ln_b = ln_b;
    a.OUT = OUT; a.ldo = ldo; a.M = M; a.M_dev = M_dev; a.ff = ff;
    return launch_wide_t<true>(a, s);
}

// The pre-norm tail (ffn.hip's PRE) in the wide form: OUT = x1 + FFN(LN_p(x1)), x1 = R + A Wo^T + bo; OUT2 (may be null) = LN(OUT).
int launch_proj_ffn_prenorm_wide(const float* A, int lda, const float* Wo, const float* bo, const float* R, int ldr, const float* pg,
                                 const float* pb, const float* W1, const float* b1, const float* W2, const float* b2, float* OUT,
                                 int ldo, const float* n2g, const float* n2b, float* OUT2, int ldo2, int M, const int* M_dev, int ff,
                                 hipStream_t s, const int* r_idx, const float* R2) {
    CONE_REQUIRE(ffn_wide_supported(ff) && (!OUT2 || (n2g && n2b)) && (!r_idx || R2), "pre-norm wide tail: bad arguments");
    FfnWideArgs a{};
    a.A = A; a.lda = lda; a.Wo = Wo; a.bo = bo; a.R = R; a.ldr = ldr; a.pg = pg; a.pb = pb; a.r_idx = r_idx; a.R2 = R2;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.ln_g = n2g ? n2g : pg; a.ln_b = n2b ? n2b : pb;
    a.OUT = OUT; a.ldo = ldo; a.OUT2 = OUT2; a.ldo2 = ldo2; a.M = M; a.M_dev = M_dev; a.ff = ff;
    return launch_wide_t<true, true>(a, s);
}

// ---- the spread form (above): scratch = XP | X1 | YG (16 groups x 16 rows x 256) + HG (16 groups x ff / 16 tiles of 256)
size_t ffn_spread_scratch_floats(int ff) { return (size_t)FS_MAX_GROUPS * (3 * 16 * 256 + (size_t)(ff >> 4) * 256); }
bool ffn_spread_supported(int M, int ff) {
    return M > 0 && (M + 15) / 16 <= FS_MAX_GROUPS && ff >= 256 && ff % 256 == 0;
}
int launch_proj_ffn_spread(const float* A, int lda, const float* Wo, const float* bo, const float* R, int ldr, const float* pg,
                           const float* pb, const float* W1, const float* b1, const float* W2, const float* b2,
                           const float* ln_g, const float* ln_b, float* OUT, int ldo, int M, int ff, float* scratch,
                           hipStream_t s, const int* M_dev, const int* r_idx, const float* R2, bool pre, float* OUT2, int ldo2) {
    CONE_REQUIRE(ffn_spread_supported(M, ff) && scratch, "spread layer tail: unsupported size M=%d ff=%d", M, ff);
    CONE_REQUIRE(!r_idx || R2, "spread layer tail: a gathered residual needs both source matrices");
    CONE_REQUIRE(lda % 4 == 0 && ldr % 4 == 0 && ldo % 4 == 0, "spread layer tail: row strides must be multiples of 4");
    FfnWideArgs a{};
    a.A = A; a.lda = lda; a.Wo = Wo; a.bo = bo; a.R = R; a.ldr = ldr; a.pg = pg; a.pb = pb;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.ln_g = ln_g; a.ln_b = ln_b;
    a.OUT = OUT; a.ldo = ldo; a.M = M; a.ff = ff; a.M_dev = M_dev; a.r_idx = r_idx; a.R2 = R2; a.OUT2 = OUT2; a.ldo2 = ldo2;
    const int groups = (M + 15) / 16, nc = ff >> 4;
    FfnSpreadBufs b;
    b.XP = scratch; b.X1 = b.XP + (size_t)FS_MAX_GROUPS * 16 * 256; b.YG = b.X1 + (size_t)FS_MAX_GROUPS * 16 * 256;
    b.HG = b.YG + (size_t)FS_MAX_GROUPS * 16 * 256;
    ProfScope ps(PK_FFN_PROJ_WIDE, M, ff, 256, M_dev, s, 0);
    hipLaunchKernelGGL(fs_proj_kernel, dim3(16, groups), dim3(64), 0, s, a, b);
    if (pre) {
        hipLaunchKernelGGL(fs_g1_kernel<true>, dim3(nc, groups), dim3(64), 0, s, a, b);
        hipLaunchKernelGGL(fs_g2_kernel<true>, dim3(16, groups), dim3(64), 0, s, a, b);
        hipLaunchKernelGGL(fs_ln_kernel<true>, dim3(groups), dim3(64), 0, s, a, b);
    } else {
        hipLaunchKernelGGL(fs_g1_kernel<false>, dim3(nc, groups), dim3(64), 0, s, a, b);
        hipLaunchKernelGGL(fs_g2_kernel<false>, dim3(16, groups), dim3(64), 0, s, a, b);
        hipLaunchKernelGGL(fs_ln_kernel<false>, dim3(groups), dim3(64), 0, s, a, b);
    }
    CONE_LAUNCH_CHECK();
    return 0;
}

}  // namespace cone
