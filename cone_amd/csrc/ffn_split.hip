// Fused feed-forward block (ffn.hip's computation: out = LayerNorm(x + W2 relu(W1 x + b1) + b2)) with every fp32
// product carried by the bf16 matrix cores as SIX partial products of three-piece operands:
//     x = xh + xm + xl,  w = wh + wm + wl   (bf16 pieces: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): exact, 24 bits)
//     x w ~= xl wh + xh wl + xm wm + xh wm + xm wh + xh wh        (the three dropped products are <= 2^-24 |x w|)
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16.  Measured on this part (tools/probe/split_bf16_probe.hip, K = 256 and
// 1024, against float64): max |error| / sum |a b| = 1.5e-7 .. 2.4e-7, the exact-fp32 v_mfma_f32_16x16x4_f32 chain of
// ffn.hip: 2.0e-7 .. 2.1e-7 -- the same accuracy, at 16 / 6 = 2.7x the matrix-core rate.  OPT-IN (cone_model_set_option
// "split_bf16"): the default path stays on the fp32 MFMA.
//
// Orientation and operand maps are those of ffn.hip (a wave owns 16 token rows = the MFMA column index; accumulators feed
// the next product without lane movement), on v_mfma_f32_16x16x32_bf16: lane (li = l % 16, lg = l / 16) holds
// A[row li][k = 8 lg + j], B[k = 8 lg + j][col li], j = 0 .. 7, and D[row 4 lg + r][col li], r = 0 .. 3.  The k index is
// permuted identically on both operands so that operands come out of the registers they already live in:
//   GEMM1 (K = 256 channels, step s = 32 channels): k slot (lg, j) <-> channel 32 s + 16 (j / 4) + 4 lg + j % 4, i.e. the
//          two float4 x[token][32 s + 4 lg ..] and x[token][32 s + 16 + 4 lg ..] -- which are also the accumulator
//          layout of the output (channel 16 t + 4 lg + r), so the residual is rebuilt from the very same registers;
//   GEMM2 (K = 32 hidden units of a chunk): k slot (lg, j) <-> unit 16 (j / 4) + 4 lg + j % 4 = accumulator register
//          j % 4 of the chunk's GEMM1 tile j / 4.
// Weights are static: they are split and laid out ONCE (cone_ffn_split_pack) in exactly the order the kernel reads them,
// as 1-KiB operand slabs [lg][li][8 bf16] (a wave's ds_read_b128 of a slab is linear in the lane index: conflict-free),
// 48 slabs = 48 KiB per ring slot: slot 2 c = W1 image of hidden chunk c ([tile 2][step 8][piece 3] slabs), slot
// 2 c + 1 = W2 image ([channel tile 16][piece 3]).  The stream is linear, so a piece's LDS-DMA source is base + lane * 16.
#include <mutex>

#include "common.h"

namespace cone {

typedef float sp_f4 __attribute__((ext_vector_type(4)));
typedef float sp_f2 __attribute__((ext_vector_type(2)));
typedef short sp_s8 __attribute__((ext_vector_type(8)));
typedef unsigned sp_u4 __attribute__((ext_vector_type(4)));
typedef __bf16 sp_b2 __attribute__((ext_vector_type(2)));

constexpr int SP_ROWS = 128;                       // token rows per workgroup (8 waves x 16)
constexpr int SP_SLOT = 48 * 1024;                 // bytes per ring slot
constexpr int SP_NSLOT = 3;
constexpr int SP_NPIECE = 6;                       // 1-KiB LDS-DMA pieces per wave per slot (48 / 8)

#define SP_GLDS16(src, dst) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// two floats -> packed bf16 pair (round to nearest even: v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned sp_pk(float a, float b) {
    const sp_b2 v = __builtin_convertvector(sp_f2{a, b}, sp_b2);
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float sp_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float sp_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
// (a, b) -> the three packed piece pairs
__device__ __forceinline__ void sp_split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = sp_pk(a, b);
    const float ra = a - sp_lo(h), rb = b - sp_hi(h);
    m = sp_pk(ra, rb);
    l = sp_pk(ra - sp_lo(m), rb - sp_hi(m));
}
// eight fp32 values (two float4) -> three 8-element bf16 operands
__device__ __forceinline__ void sp_split8(const sp_f4& v0, const sp_f4& v1, sp_s8& h, sp_s8& m, sp_s8& l) {
    unsigned a[4], b[4], c[4];
    sp_split2(v0[0], v0[1], a[0], b[0], c[0]);
    sp_split2(v0[2], v0[3], a[1], b[1], c[1]);
    sp_split2(v1[0], v1[1], a[2], b[2], c[2]);
    sp_split2(v1[2], v1[3], a[3], b[3], c[3]);
    const sp_u4 uh = {a[0], a[1], a[2], a[3]}, um = {b[0], b[1], b[2], b[3]}, ul = {c[0], c[1], c[2], c[3]};
    h = __builtin_bit_cast(sp_s8, uh); m = __builtin_bit_cast(sp_s8, um); l = __builtin_bit_cast(sp_s8, ul);
}

struct FfnSplitArgs {
    const float* X; int ldx;                      // (M, 256) block input = residual
    const void* Wimg;                             // packed weight image: 2 * (ff / 32) slots of 48 KiB
    const float* b1; const float* b2;             // (ff), (256)
    const float* ln_g; const float* ln_b;         // (256)
    float* OUT; int ldo;
    int M; const int* M_dev;
    int ff;
    // PROJ: the block input is LayerNorm(R + A Wo^T + bo), computed here (ffn.hip's PROJ form): A (M, 256) attention rows,
    // R residual rows (r_idx != null: gathered, row i = R[r_idx[i]] or R2[~r_idx[i]]), Woimg = Wo's image (8 slots)
    const float* A; int lda; const float* R; int ldr; const int* r_idx; const float* R2;
    const void* Woimg; const float* bo; const float* pg; const float* pb;
    // QKV: the NEXT layer's q | k | v projection of the rows this kernel produces, computed from the registers that hold
    // them: Qimg = its weight image (NQ = n_qkv / 32 slots), qb its bias, QKV (M, n_qkv) its output
    const void* Qimg; const float* qb; float* QKV; int ldq; int n_qkv;
};

#define SP_SB() __builtin_amdgcn_sched_barrier(0)
#define SP_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0)
// the six partial products of one (weight fragment triple, activation triple), small terms first
#define SP_MM6(acc, wh, wm, wl, xh, xm, xl) \
    {                                        \
        SP_MFMA(acc, wl, xh);                \
        SP_MFMA(acc, wh, xl);                \
        SP_MFMA(acc, wm, xm);                \
        SP_MFMA(acc, wm, xh);                \
        SP_MFMA(acc, wh, xm);                \
        SP_MFMA(acc, wh, xh);                \
    }

template <bool PROJ, bool QKV>
__global__ __launch_bounds__(512, 2) void ffn_split_kernel(FfnSplitArgs p) {
    extern __shared__ __attribute__((aligned(16))) char sp_smem[];
    float* b1s = reinterpret_cast<float*>(sp_smem + SP_NSLOT * SP_SLOT);
    int M = p.M;
    if (p.M_dev) { const int md = *p.M_dev; M = md < M ? md : M; }
    const int n_tiles = (M + SP_ROWS - 1) / SP_ROWS;
    if ((int)blockIdx.x >= n_tiles) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int ff = p.ff, nc = ff >> 5;            // 32-unit hidden chunks
    constexpr int NP = PROJ ? 8 : 0;              // leading slots of the output projection (32 channels each)
    const int NQ = QKV ? p.n_qkv >> 5 : 0;        // trailing slots of the next layer's q | k | v projection
    const int G = NP + 2 * nc + NQ;               // ring slots per tile

    float* prm = b1s + ff;                        // b2, ln_g, ln_b
    for (int i = tid; i < (ff >> 2); i += 512)
        reinterpret_cast<sp_f4*>(b1s)[i] = reinterpret_cast<const sp_f4*>(p.b1)[i];
    if (tid < 64) {
        reinterpret_cast<sp_f4*>(prm)[tid] = reinterpret_cast<const sp_f4*>(p.b2)[tid];
        reinterpret_cast<sp_f4*>(prm + 256)[tid] = reinterpret_cast<const sp_f4*>(p.ln_g)[tid];
        reinterpret_cast<sp_f4*>(prm + 512)[tid] = reinterpret_cast<const sp_f4*>(p.ln_b)[tid];
        if (PROJ) {
            reinterpret_cast<sp_f4*>(prm + 768)[tid] = reinterpret_cast<const sp_f4*>(p.bo)[tid];
            reinterpret_cast<sp_f4*>(prm + 1024)[tid] = reinterpret_cast<const sp_f4*>(p.pg)[tid];
            reinterpret_cast<sp_f4*>(prm + 1280)[tid] = reinterpret_cast<const sp_f4*>(p.pb)[tid];
        }
    }
    float* qbs = prm + 1536;                      // q | k | v bias
    if (QKV)
        for (int i = tid; i < (p.n_qkv >> 2); i += 512)
            reinterpret_cast<sp_f4*>(qbs)[i] = reinterpret_cast<const sp_f4*>(p.qb)[i];

    // LDS-DMA: piece i of slot g (of the current tile; g >= G: the next tile's first slots, the same weights) = 1 KiB
    // at image offset g * 48 KiB + (6 wave + i) KiB, lane * 16 B inside it; destination = the same offset in ring slot
    // (sb + g) % 3.
    int sb = 0;
    const char* wimg = reinterpret_cast<const char*>(p.Wimg);
    const char* woimg = reinterpret_cast<const char*>(p.Woimg);
    const char* qimg = reinterpret_cast<const char*>(p.Qimg);
    auto stream_piece = [&](int g, int i) {
        const int gg = g < G ? g : g - G;
        char* dstp = sp_smem + ((sb + g) % SP_NSLOT) * SP_SLOT + (wave * SP_NPIECE + i) * 1024;
        const char* ub = (PROJ && gg < NP ? woimg + (size_t)gg * SP_SLOT
                          : (QKV && gg >= NP + 2 * nc ? qimg + (size_t)(gg - NP - 2 * nc) * SP_SLOT
                                                       : wimg + (size_t)(gg - NP) * SP_SLOT)) +
                         (size_t)(wave * SP_NPIECE + i) * 1024;
        asm volatile("" : "+s"(ub));
        SP_GLDS16(ub + (unsigned)(lane * 16), dstp);
    };
#define SP_SLOT_OF(g) (sp_smem + ((sb + (g)) % SP_NSLOT) * SP_SLOT)
#define SP_RD(slot, slab) (*reinterpret_cast<const sp_s8*>((slot) + (slab) * 1024 + lane * 16))
    // end of a slot: the next slot has landed (all but the pieces issued last, which belong to the slot after it), and
    // every wave is done with the slot that the next pieces will overwrite
#define SP_END_SLOT()                                                             \
    {                                                                             \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SP_NPIECE) : "memory");          \
        __builtin_amdgcn_s_barrier();                                             \
    }

#pragma unroll
    for (int i = 0; i < SP_NPIECE; ++i) stream_piece(0, i);
#pragma unroll
    for (int i = 0; i < SP_NPIECE; ++i) stream_piece(1, i);

    // a tile's rows (PROJ: the attention rows = B operand of the projection, else the block input) are requested from
    // the previous tile's epilogue -- once its operand pieces are dead, ahead of its stores -- and split at the tile's top
    sp_f4 xr[16];
    auto load_rows = [&](int tile) {
        const int row = tile * SP_ROWS + wave * 16 + li;
        const size_t lr = (size_t)(row < M ? row : M - 1);
        const float* xp = (PROJ ? p.A + lr * p.lda : p.X + lr * p.ldx) + 4 * lg;
#pragma unroll
        for (int q = 0; q < 16; ++q) xr[q] = *reinterpret_cast<const sp_f4*>(xp + 16 * q);
    };
    load_rows(blockIdx.x);
    bool first = true;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int my_row = tile * SP_ROWS + wave * 16 + li;
    // split once: xh / xm / xl [s] = B operand of GEMM1's step s (channels 32 s + 16 (j / 4) + 4 lg + j % 4)
    sp_s8 xh[8], xm[8], xl[8];
    const size_t ld_row = (size_t)(my_row < M ? my_row : M - 1);
#pragma unroll
    for (int s = 0; s < 8; ++s) sp_split8(xr[2 * s], xr[2 * s + 1], xh[s], xm[s], xl[s]);
    if (first) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SP_NPIECE) : "memory");
        __syncthreads();
        first = false;
    }
    sp_s8 f[2][3];                                  // ping-pong fragment sets (statically indexed: loops are unrolled)
    if (PROJ) {
        // ---- attention output projection: slot g = channels [32 g, 32 g + 32) of A Wo^T ([tile 2][step 8][piece 3])
        sp_f4 x1[16];
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            const char* sa = SP_SLOT_OF(g);
            sp_f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
            f[0][0] = SP_RD(sa, 0); f[0][1] = SP_RD(sa, 1); f[0][2] = SP_RD(sa, 2);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                SP_SB();
                asm volatile("" : "+v"(f[u & 1][0]), "+v"(f[u & 1][1]), "+v"(f[u & 1][2]));
                SP_SB();
                if (u < 15) {
                    f[(u + 1) & 1][0] = SP_RD(sa, (u + 1) * 3 + 0); f[(u + 1) & 1][1] = SP_RD(sa, (u + 1) * 3 + 1);
                    f[(u + 1) & 1][2] = SP_RD(sa, (u + 1) * 3 + 2);
                }
                SP_SB();
                if (u < 8) { SP_MM6(a0, f[u & 1][0], f[u & 1][1], f[u & 1][2], xh[u & 7], xm[u & 7], xl[u & 7]) }
                else { SP_MM6(a1, f[u & 1][0], f[u & 1][1], f[u & 1][2], xh[u & 7], xm[u & 7], xl[u & 7]) }
                if ((u & 1) && (u >> 1) < SP_NPIECE) stream_piece(g + 2, u >> 1);
            }
            SP_SB();
            x1[2 * g] = a0; x1[2 * g + 1] = a1;
            SP_END_SLOT()
        }
        // + bo + residual rows, LayerNorm: the block input, split for GEMM1 (the attention pieces are dead)
        {
            const float* rp = p.R + ld_row * p.ldr + 4 * lg;
            if (p.r_idx) {
                const int ix = p.r_idx[ld_row];
                rp = (ix >= 0 ? p.R + (size_t)ix * p.ldr : p.R2 + (size_t)(~ix) * p.ldr) + 4 * lg;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q)
                x1[q] += *reinterpret_cast<const sp_f4*>(rp + 16 * q) + *reinterpret_cast<const sp_f4*>(prm + 768 + 16 * q + 4 * lg);
        }
        float t1 = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) t1 += (x1[t][0] + x1[t][1]) + (x1[t][2] + x1[t][3]);
        t1 += __shfl_xor(t1, 16, 64);
        t1 += __shfl_xor(t1, 32, 64);
        const float mu = t1 * (1.0f / 256.0f);
        float t2 = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { x1[t][r] -= mu; t2 = fmaf(x1[t][r], x1[t][r], t2); }
        }
        t2 += __shfl_xor(t2, 16, 64);
        t2 += __shfl_xor(t2, 32, 64);
        const float rs = 1.0f / sqrtf(t2 * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const sp_f4 g4 = *reinterpret_cast<const sp_f4*>(prm + 1024 + 16 * t + 4 * lg);
            const sp_f4 b4 = *reinterpret_cast<const sp_f4*>(prm + 1280 + 16 * t + 4 * lg);
#pragma unroll
            for (int r = 0; r < 4; ++r) x1[t][r] = x1[t][r] * rs * g4[r] + b4[r];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) sp_split8(x1[2 * s], x1[2 * s + 1], xh[s], xm[s], xl[s]);
    }
    sp_f4 y[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) y[t] = sp_f4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < nc; ++c) {
        // Both products walk 16 units of (3 fragments, 6 MFMAs); the fragments of unit u + 1 are requested ahead of the
        // MFMAs of unit u (the empty asm is where the wait for them lands: after those MFMAs, before the next request).
        // ---- GEMM1: the chunk's two 16-unit tiles over the 256 channels (slot 2 c: [tile][step][piece] slabs), tile by
        // tile: bias + ReLU + split of tile 0 run under tile 1's MFMAs
        const char* sa = SP_SLOT_OF(NP + 2 * c);
        sp_f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        unsigned p0h[2], p0m[2], p0l[2];
        f[0][0] = SP_RD(sa, 0); f[0][1] = SP_RD(sa, 1); f[0][2] = SP_RD(sa, 2);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            SP_SB();
            asm volatile("" : "+v"(f[u & 1][0]), "+v"(f[u & 1][1]), "+v"(f[u & 1][2]));
            SP_SB();
            if (u < 15) {
                f[(u + 1) & 1][0] = SP_RD(sa, (u + 1) * 3 + 0); f[(u + 1) & 1][1] = SP_RD(sa, (u + 1) * 3 + 1);
                f[(u + 1) & 1][2] = SP_RD(sa, (u + 1) * 3 + 2);
            }
            SP_SB();
            if (u < 8) { SP_MM6(a0, f[u & 1][0], f[u & 1][1], f[u & 1][2], xh[u & 7], xm[u & 7], xl[u & 7]) }
            else { SP_MM6(a1, f[u & 1][0], f[u & 1][1], f[u & 1][2], xh[u & 7], xm[u & 7], xl[u & 7]) }
            if ((u & 1) && (u >> 1) < SP_NPIECE) stream_piece(NP + 2 * c + 2, u >> 1);
            if (u == 9) {       // tile 0 is complete: its bias + ReLU + split run under tile 1's MFMAs
                a0 += *reinterpret_cast<const sp_f4*>(b1s + 32 * c + 4 * lg);
#pragma unroll
                for (int r = 0; r < 4; ++r) a0[r] = fmaxf(a0[r], 0.f);
                sp_split2(a0[0], a0[1], p0h[0], p0m[0], p0l[0]);
                sp_split2(a0[2], a0[3], p0h[1], p0m[1], p0l[1]);
            }
        }
        SP_SB();
        SP_END_SLOT()
        // ---- GEMM2: all 256 output channels over the chunk's 32 hidden units (slot 2 c + 1: [channel tile][piece]); its
        // first fragments are requested ahead of tile 1's bias + ReLU + split
        const char* sw = SP_SLOT_OF(NP + 2 * c + 1);
        f[0][0] = SP_RD(sw, 0); f[0][1] = SP_RD(sw, 1); f[0][2] = SP_RD(sw, 2);
        SP_SB();
        // the B operand of GEMM2 (k slot (lg, j) <-> unit 16 (j / 4) + 4 lg + j % 4): tile 0's pieces, then tile 1's
        sp_s8 hh, hm, hl;
        {
            unsigned p1h[2], p1m[2], p1l[2];
            a1 += *reinterpret_cast<const sp_f4*>(b1s + 32 * c + 16 + 4 * lg);
#pragma unroll
            for (int r = 0; r < 4; ++r) a1[r] = fmaxf(a1[r], 0.f);
            sp_split2(a1[0], a1[1], p1h[0], p1m[0], p1l[0]);
            sp_split2(a1[2], a1[3], p1h[1], p1m[1], p1l[1]);
            hh = __builtin_bit_cast(sp_s8, sp_u4{p0h[0], p0h[1], p1h[0], p1h[1]});
            hm = __builtin_bit_cast(sp_s8, sp_u4{p0m[0], p0m[1], p1m[0], p1m[1]});
            hl = __builtin_bit_cast(sp_s8, sp_u4{p0l[0], p0l[1], p1l[0], p1l[1]});
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            SP_SB();
            asm volatile("" : "+v"(f[t & 1][0]), "+v"(f[t & 1][1]), "+v"(f[t & 1][2]));
            SP_SB();
            if (t < 15) {
                f[(t + 1) & 1][0] = SP_RD(sw, (t + 1) * 3 + 0); f[(t + 1) & 1][1] = SP_RD(sw, (t + 1) * 3 + 1);
                f[(t + 1) & 1][2] = SP_RD(sw, (t + 1) * 3 + 2);
            }
            SP_SB();
            SP_MM6(y[t], f[t & 1][0], f[t & 1][1], f[t & 1][2], hh, hm, hl)
            if ((t & 1) && (t >> 1) < SP_NPIECE) stream_piece(NP + 2 * c + 3, t >> 1);
        }
        SP_SB();
        SP_END_SLOT()
    }
    // ---- epilogue: + b2 + residual (x = xh + xm + xl exactly, in the accumulator layout), LayerNorm, store
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        // (opaque here: otherwise the unpacked halves computed by the split at the tile's top are kept alive across the
        // whole tile -- 80 spilled registers -- instead of being re-derived by two shifts)
        asm volatile("" : "+v"(xh[s]), "+v"(xm[s]), "+v"(xl[s]));
        const sp_u4 uh = __builtin_bit_cast(sp_u4, xh[s]), um = __builtin_bit_cast(sp_u4, xm[s]), ul = __builtin_bit_cast(sp_u4, xl[s]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {           // pair e: elements 2 e, 2 e + 1 of the step = (tile 2 s + e / 2, r = 2 (e % 2) ..)
            const float x0 = (sp_lo(uh[e]) + sp_lo(um[e])) + sp_lo(ul[e]);
            const float x1 = (sp_hi(uh[e]) + sp_hi(um[e])) + sp_hi(ul[e]);
            y[2 * s + (e >> 1)][2 * (e & 1)] += x0;
            y[2 * s + (e >> 1)][2 * (e & 1) + 1] += x1;
        }
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) y[t] += *reinterpret_cast<const sp_f4*>(prm + 16 * t + 4 * lg);
    SP_SB();
    // the operand pieces are dead: the next tile's rows travel under the LayerNorm and the stores (after the last tile a
    // valid tile is simply re-read, so that the register tile has one definition per iteration)
    load_rows(tile + (int)gridDim.x < n_tiles ? tile + (int)gridDim.x : tile);
    SP_SB();
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s1 += (y[t][0] + y[t][1]) + (y[t][2] + y[t][3]);
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 * (1.0f / 256.0f);
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { y[t][r] -= mean; s2 = fmaf(y[t][r], y[t][r], s2); }
    }
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float rstd = 1.0f / sqrtf(s2 * (1.0f / 256.0f) + 1e-5f);
    if (my_row < M) {
        float* op = p.OUT + (size_t)my_row * p.ldo + 4 * lg;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const sp_f4 g = *reinterpret_cast<const sp_f4*>(prm + 256 + 16 * t + 4 * lg);
            const sp_f4 be = *reinterpret_cast<const sp_f4*>(prm + 512 + 16 * t + 4 * lg);
            sp_f4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = y[t][r] * rstd * g[r] + be[r];
            *reinterpret_cast<sp_f4*>(op + 16 * t) = o;
            if (QKV) y[t] = o;
        }
    } else if (QKV) {       // rows past M feed unstored outputs: any finite values
#pragma unroll
        for (int t = 0; t < 16; ++t) y[t] = sp_f4{0.f, 0.f, 0.f, 0.f};
    }
    if (QKV) {
        // ---- the next layer's q | k | v projection of these rows, straight from the registers: split once more, then
        // NQ slots of 32 output channels each ([tile 2][step 8][piece 3] slabs), stored from the accumulators
#pragma unroll
        for (int s = 0; s < 8; ++s) sp_split8(y[2 * s], y[2 * s + 1], xh[s], xm[s], xl[s]);
        float* qrow = p.QKV + (size_t)my_row * p.ldq + 4 * lg;
        for (int g = 0; g < NQ; ++g) {
            const char* sa = SP_SLOT_OF(NP + 2 * nc + g);
            sp_f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
            f[0][0] = SP_RD(sa, 0); f[0][1] = SP_RD(sa, 1); f[0][2] = SP_RD(sa, 2);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                SP_SB();
                asm volatile("" : "+v"(f[u & 1][0]), "+v"(f[u & 1][1]), "+v"(f[u & 1][2]));
                SP_SB();
                if (u < 15) {
                    f[(u + 1) & 1][0] = SP_RD(sa, (u + 1) * 3 + 0); f[(u + 1) & 1][1] = SP_RD(sa, (u + 1) * 3 + 1);
                    f[(u + 1) & 1][2] = SP_RD(sa, (u + 1) * 3 + 2);
                }
                SP_SB();
                if (u < 8) { SP_MM6(a0, f[u & 1][0], f[u & 1][1], f[u & 1][2], xh[u & 7], xm[u & 7], xl[u & 7]) }
                else { SP_MM6(a1, f[u & 1][0], f[u & 1][1], f[u & 1][2], xh[u & 7], xm[u & 7], xl[u & 7]) }
                if ((u & 1) && (u >> 1) < SP_NPIECE) stream_piece(NP + 2 * nc + g + 2, u >> 1);
            }
            SP_SB();
            if (my_row < M) {
                *reinterpret_cast<sp_f4*>(qrow + 32 * g) = a0 + *reinterpret_cast<const sp_f4*>(qbs + 32 * g + 4 * lg);
                *reinterpret_cast<sp_f4*>(qrow + 32 * g + 16) = a1 + *reinterpret_cast<const sp_f4*>(qbs + 32 * g + 16 + 4 * lg);
            }
            SP_END_SLOT()
        }
    }
    sb = (sb + G) % SP_NSLOT;
    }   // tile loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------------------------
// C = X W^T + bias for K = 256 and any N % 32 == 0 (the q | k | v projection of the later encoder layers) with the same
// split operands: the projection phase of the kernel above on its own -- a wave's 16 x 256 rows stay in registers as bf16
// pieces, W streams as N / 32 slots ([tile 2][step 8][piece 3] slabs each), every slot's 32 output channels are stored
// straight from the accumulators (lane = token, 16 B = 4 consecutive channels).
struct RowsSplitArgs {
    const float* X; int ldx; const void* Wimg; const float* bias; float* C; int ldc; int M; const int* M_dev; int N;
};

__global__ __launch_bounds__(512, 2) void rows256_split_kernel(RowsSplitArgs p) {
    extern __shared__ __attribute__((aligned(16))) char sp_smem[];
    float* bs = reinterpret_cast<float*>(sp_smem + SP_NSLOT * SP_SLOT);
    int M = p.M;
    if (p.M_dev) { const int md = *p.M_dev; M = md < M ? md : M; }
    const int n_tiles = (M + SP_ROWS - 1) / SP_ROWS;
    if ((int)blockIdx.x >= n_tiles) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int G = p.N >> 5;                         // ring slots per tile
    for (int i = tid; i < (p.N >> 2); i += 512)
        reinterpret_cast<sp_f4*>(bs)[i] = p.bias ? reinterpret_cast<const sp_f4*>(p.bias)[i] : sp_f4{0.f, 0.f, 0.f, 0.f};
    int sb = 0;
    const char* wimg = reinterpret_cast<const char*>(p.Wimg);
    auto stream_piece = [&](int g, int i) {
        int gg = g;
        while (gg >= G) gg -= G;                    // the next tile's first slots (G may be 1 or 2)
        char* dstp = sp_smem + ((sb + g) % SP_NSLOT) * SP_SLOT + (wave * SP_NPIECE + i) * 1024;
        const char* ub = wimg + (size_t)gg * SP_SLOT + (size_t)(wave * SP_NPIECE + i) * 1024;
        asm volatile("" : "+s"(ub));
        SP_GLDS16(ub + (unsigned)(lane * 16), dstp);
    };
#pragma unroll
    for (int i = 0; i < SP_NPIECE; ++i) stream_piece(0, i);
#pragma unroll
    for (int i = 0; i < SP_NPIECE; ++i) stream_piece(1, i);
    bool first = true;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int my_row = tile * SP_ROWS + wave * 16 + li;
        const size_t ld_row = (size_t)(my_row < M ? my_row : M - 1);
        sp_s8 xh[8], xm[8], xl[8];
        {
            const float* xp = p.X + ld_row * p.ldx + 4 * lg;
            sp_f4 xr[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) xr[q] = *reinterpret_cast<const sp_f4*>(xp + 16 * q);
#pragma unroll
            for (int s = 0; s < 8; ++s) sp_split8(xr[2 * s], xr[2 * s + 1], xh[s], xm[s], xl[s]);
        }
        if (first) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SP_NPIECE) : "memory");
            __syncthreads();
            first = false;
        }
        float* crow = p.C + (size_t)my_row * p.ldc + 4 * lg;
        sp_s8 f[2][3];
        for (int g = 0; g < G; ++g) {
            const char* sa = SP_SLOT_OF(g);
            sp_f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
            f[0][0] = SP_RD(sa, 0); f[0][1] = SP_RD(sa, 1); f[0][2] = SP_RD(sa, 2);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                SP_SB();
                asm volatile("" : "+v"(f[u & 1][0]), "+v"(f[u & 1][1]), "+v"(f[u & 1][2]));
                SP_SB();
                if (u < 15) {
                    f[(u + 1) & 1][0] = SP_RD(sa, (u + 1) * 3 + 0); f[(u + 1) & 1][1] = SP_RD(sa, (u + 1) * 3 + 1);
                    f[(u + 1) & 1][2] = SP_RD(sa, (u + 1) * 3 + 2);
                }
                SP_SB();
                if (u < 8) { SP_MM6(a0, f[u & 1][0], f[u & 1][1], f[u & 1][2], xh[u & 7], xm[u & 7], xl[u & 7]) }
                else { SP_MM6(a1, f[u & 1][0], f[u & 1][1], f[u & 1][2], xh[u & 7], xm[u & 7], xl[u & 7]) }
                if ((u & 1) && (u >> 1) < SP_NPIECE) stream_piece(g + 2, u >> 1);
            }
            SP_SB();
            if (my_row < M) {
                *reinterpret_cast<sp_f4*>(crow + 32 * g) = a0 + *reinterpret_cast<const sp_f4*>(bs + 32 * g + 4 * lg);
                *reinterpret_cast<sp_f4*>(crow + 32 * g + 16) = a1 + *reinterpret_cast<const sp_f4*>(bs + 32 * g + 16 + 4 * lg);
            }
            SP_END_SLOT()
        }
        sb = (sb + G) % SP_NSLOT;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#undef SP_SB
#undef SP_MM6
#undef SP_MFMA
#undef SP_END_SLOT
#undef SP_RD
#undef SP_SLOT_OF

// three 48-KiB ring slots + b1 + six parameter rows within the CU's 160 KiB of LDS
bool ffn_split_supported(int ff) { return ff >= 64 && ff % 32 == 0 && ff <= 2048; }
size_t ffn_split_image_bytes(int ff) { return (size_t)2 * (ff / 32) * SP_SLOT; }
size_t ffn_split_proj_image_bytes() { return (size_t)8 * SP_SLOT; }
// does the fused tail + a q | k | v projection of n_qkv outputs fit the CU's LDS (ring + b1 + parameter rows + its bias)?
bool ffn_split_qkv_fits(int ff, int n_qkv) {
    return ffn_split_supported(ff) && n_qkv >= 32 && n_qkv % 32 == 0 &&
           (size_t)SP_NSLOT * SP_SLOT + (size_t)(ff + 6 * 256 + n_qkv) * sizeof(float) <= 160 * 1024;
}

template <bool PROJ, bool QKV>
static int launch_ffn_split_t(const FfnSplitArgs& a, hipStream_t s) {
    const size_t lds = (size_t)SP_NSLOT * SP_SLOT + (size_t)(a.ff + 6 * 256 + (QKV ? a.n_qkv : 0)) * sizeof(float);
    CONE_REQUIRE(lds <= 160 * 1024, "split-bf16 fused layer tail: %zu bytes of LDS (ff %d, q|k|v %d) exceed 160 KiB", lds, a.ff,
                 a.n_qkv);
    static DeviceOnce once;
    int n_cu = 0;
    CONE_CHECK_HIP(device_once(once, [] {
        return hipFuncSetAttribute((const void*)ffn_split_kernel<PROJ, QKV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }, &n_cu));
    const int tiles = (a.M + SP_ROWS - 1) / SP_ROWS;
    const int grid = tiles < n_cu ? tiles : n_cu;
    // FLOPs of a record = 4 M ff 256 (+ 2 M 256 256 with the projection); the fused q | k | v projection adds
    // 2 M n_qkv 256 = 4 M (n_qkv / 2) 256: it is booked as n_qkv / 2 extra hidden units
    ProfScope ps(PROJ ? PK_FFN_PROJ : PK_FFN_FUSED, a.M, a.ff + (QKV ? a.n_qkv / 2 : 0), 256, a.M_dev, s);
    hipLaunchKernelGGL((ffn_split_kernel<PROJ, QKV>), dim3((unsigned)grid), dim3(512), lds, s, a);
    CONE_LAUNCH_CHECK();
    return 0;
}

int launch_ffn_split(const float* X, int ldx, const void* Wimg, const float* b1, const float* b2, const float* ln_g,
                     const float* ln_b, float* OUT, int ldo, int M, const int* M_dev, int ff, hipStream_t s) {
    CONE_REQUIRE(ffn_split_supported(ff), "split-bf16 fused FFN: dim_feedforward=%d unsupported", ff);
    CONE_REQUIRE(X && Wimg && b1 && b2 && ln_g && ln_b && OUT, "split-bf16 fused FFN: null argument");
    CONE_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0, "split-bf16 fused FFN: row strides must be multiples of 4");
    if (M <= 0) return 0;
    FfnSplitArgs a{};
    a.X = X; a.ldx = ldx; a.Wimg = Wimg; a.b1 = b1; a.b2 = b2; a.ln_g = ln_g; a.ln_b = ln_b;
    a.OUT = OUT; a.ldo = ldo; a.M = M; a.M_dev = M_dev; a.ff = ff;
    return launch_ffn_split_t<false, false>(a, s);
}

int launch_proj_ffn_split(const float* A, int lda, const void* Woimg, const float* bo, const float* R, int ldr,
                          const float* pg, const float* pb, const void* Wimg, const float* b1, const float* b2,
                          const float* ln_g, const float* ln_b, float* OUT, int ldo, int M, const int* M_dev, int ff,
                          hipStream_t s, const int* r_idx, const float* R2, const void* Qimg, const float* qb, float* QKV,
                          int ldq, int n_qkv) {
    CONE_REQUIRE(ffn_split_supported(ff), "split-bf16 fused layer tail: dim_feedforward=%d unsupported", ff);
    CONE_REQUIRE(A && Woimg && bo && R && pg && pb && Wimg && b1 && b2 && ln_g && ln_b && OUT, "split-bf16 fused layer tail: null argument");
    CONE_REQUIRE(!r_idx || R2, "split-bf16 fused layer tail: a gathered residual needs both source matrices");
    CONE_REQUIRE(lda % 4 == 0 && ldr % 4 == 0 && ldo % 4 == 0, "split-bf16 fused layer tail: row strides must be multiples of 4");
    if (M <= 0) return 0;
    FfnSplitArgs a{};
    a.A = A; a.lda = lda; a.Woimg = Woimg; a.bo = bo; a.R = R; a.ldr = ldr; a.pg = pg; a.pb = pb; a.r_idx = r_idx; a.R2 = R2;
    a.Wimg = Wimg; a.b1 = b1; a.b2 = b2; a.ln_g = ln_g; a.ln_b = ln_b;
    a.OUT = OUT; a.ldo = ldo; a.M = M; a.M_dev = M_dev; a.ff = ff;
    if (Qimg) {
        CONE_REQUIRE(qb && QKV && n_qkv >= 32 && n_qkv % 32 == 0 && ldq % 4 == 0, "split-bf16 fused layer tail: bad q|k|v arguments");
        a.Qimg = Qimg; a.qb = qb; a.QKV = QKV; a.ldq = ldq; a.n_qkv = n_qkv;
        return launch_ffn_split_t<true, true>(a, s);
    }
    return launch_ffn_split_t<true, false>(a, s);
}

bool rows256_split_supported(int N) { return N >= 32 && N % 32 == 0 && N <= 3072; }
size_t rows256_split_image_bytes(int N) { return (size_t)(N / 32) * SP_SLOT; }

int launch_rows256_split(const float* X, int ldx, const void* Wimg, const float* bias, float* C, int ldc, int M,
                         const int* M_dev, int N, hipStream_t s) {
    CONE_REQUIRE(rows256_split_supported(N), "split-bf16 row GEMM: N=%d unsupported", N);
    CONE_REQUIRE(X && Wimg && C && ldx % 4 == 0 && ldc % 4 == 0, "split-bf16 row GEMM: bad argument");
    if (M <= 0) return 0;
    static DeviceOnce once;
    int n_cu = 0;
    CONE_CHECK_HIP(device_once(once, [] {
        return hipFuncSetAttribute((const void*)rows256_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   SP_NSLOT * SP_SLOT + 3072 * (int)sizeof(float));
    }, &n_cu));
    RowsSplitArgs a{X, ldx, Wimg, bias, C, ldc, M, M_dev, N};
    const int tiles = (M + SP_ROWS - 1) / SP_ROWS;
    const int grid = tiles < n_cu ? tiles : n_cu;
    ProfScope ps(PK_GEMM_ROWS16, M, N, 256, M_dev, s);
    hipLaunchKernelGGL(rows256_split_kernel, dim3((unsigned)grid), dim3(512),
                       (size_t)SP_NSLOT * SP_SLOT + (size_t)N * sizeof(float), s, a);
    CONE_LAUNCH_CHECK();
    return 0;
}

// ---- weight images (once per model).  One thread per 16-B fragment (8 bf16 of one piece): slot g, slab sl, lane l.
// W2 != null: W1 (ff, 256) and W2 (256, ff) -> 2 * (ff / 32) slots (slot 2 c: W1 rows of hidden chunk c, slot 2 c + 1: W2
// columns).  W2 == null: W1 is any (N = ff, 256) weight of a 256-channel product (the attention output projection): N / 32
// slots in the W1 slab order.
__global__ __launch_bounds__(256) void ffn_split_pack_kernel(const float* __restrict__ W1, const float* __restrict__ W2,
                                                             int ff, unsigned* __restrict__ img) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)(W2 ? 2 : 1) * (ff / 32) * 48 * 64;
    if (idx >= total) return;
    const int l = (int)(idx & 63);
    const int sl = (int)((idx >> 6) % 48);
    const int g = (int)(idx / (48 * 64));
    const int c = W2 ? g >> 1 : g, li = l & 15, lg = l >> 4;
    float v[8];
    int piece;
    if (!W2 || (g & 1) == 0) {      // W1 image: slab = tile * 24 + step * 3 + piece
        const int t = sl / 24, s = (sl % 24) / 3;
        piece = sl % 3;
        const float* row = W1 + (size_t)(32 * c + 16 * t + li) * 256;
        for (int j = 0; j < 8; ++j) v[j] = row[32 * s + 16 * (j >> 2) + 4 * lg + (j & 3)];
    } else {                        // W2 image: slab = channel tile * 3 + piece
        const int t = sl / 3;
        piece = sl % 3;
        const float* row = W2 + (size_t)(16 * t + li) * ff + 32 * c;
        for (int j = 0; j < 8; ++j) v[j] = row[16 * (j >> 2) + 4 * lg + (j & 3)];
    }
    unsigned out[4];
    for (int e = 0; e < 4; ++e) {
        unsigned h, m, lo;
        sp_split2(v[2 * e], v[2 * e + 1], h, m, lo);
        out[e] = piece == 0 ? h : (piece == 1 ? m : lo);
    }
    unsigned* dst = img + idx * 4;
    dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2]; dst[3] = out[3];
}

int launch_ffn_split_pack(const float* W1, const float* W2, int ff, void* img, hipStream_t s) {
    CONE_REQUIRE(W2 ? ffn_split_supported(ff) : (ff >= 32 && ff % 32 == 0), "split-bf16 weight image: %d rows unsupported", ff);
    CONE_REQUIRE(W1 && img, "split-bf16 weight image: null argument");
    const size_t total = (size_t)(W2 ? 2 : 1) * (ff / 32) * 48 * 64;
    hipLaunchKernelGGL(ffn_split_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W1, W2, ff,
                       reinterpret_cast<unsigned*>(img));
    CONE_LAUNCH_CHECK();
    return 0;
}

}  // namespace cone
