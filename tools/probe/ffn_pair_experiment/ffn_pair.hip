// The fused layer tail (ffn.hip:  LayerNorm(R + A Wo^T + bo)  ->  LayerNorm(x + W2 relu(W1 x + b1) + b2),
// cone/transformer.py:239-245, 308-316) on v_mfma_f32_32x32x2_f32: the throughput form for the full rounds of a launch.
//
// Why a second shape.  tools/probe/mfma_shape_rate.hip: the exact-fp32 MFMA runs at 64 FLOP / clock / SIMD in either shape, but
// every register-writing load between two MFMAs costs matrix-pipe issue time -- 16x16x4 with one ds_read_b128 per four MFMAs
// (what ffn.hip does) sustains 140 TFLOP/s of 157, 136 with one vector instruction on top; 32x32x2 needs HALF the operand
// loads, vector and scalar instructions per FLOP: 148 / 145.  tools/probe/mfma_shape_bits.hip: both shapes are the same
// chain of rounded FMAs in k-slot order, so a kernel of either shape reproduces the other's bits when it walks k the same way.
//
// The register problem and the pair split.  A wave of 32 tokens needs x^T (32 x 256) and Y^T (32 x 256): 256 registers,
// all there are at two waves per SIMD.  Here TWO waves share 32 tokens and split both contractions so that nothing is
// computed twice:
//   GEMM1  H^T[h][tok] = sum_k W1[h][k] x[tok][k]:  ffn.hip accumulates FOUR chains per hidden unit, chain r = the k with
//          k % 4 == r, combined as (c0 + c1) + (c2 + c3).  Wave `hf` of a pair runs chains 2 hf and 2 hf + 1 -- its half of x^T
//          is the k with (k % 4) / 2 == hf, 64 registers -- and the pair exchanges P_hf = c_{2hf} + c_{2hf+1} through LDS
//          (4 KiB per wave and 32-unit chunk): h = relu((P_0 + P_1) + b1) in both, the very same sum.
//   GEMM2  Y^T[ch][tok] += sum_h W2[ch][h] H^T[h][tok]:  wave hf owns the 128 output channels with (ch % 4) / 2 == hf (64
//          accumulator registers) -- exactly the channels whose x it holds, so residual adds stay register-to-register --
//          and walks the hidden units in ffn.hip's order (chunks of 16: r = 0 .. 3, k slots lg = 0 .. 3).
//   The output projection in front runs the full K per wave (the attention rows are only live there: 128 registers) for
//   the wave's 128 channels, two chains per channel as in ffn.hip.  LayerNorm: a token's moments are summed per (lg, r / 2)
//   class -- ffn.hip / ffn_wide.hip use the same association (ffn_layernorm_regs) -- so the partial sums of the two waves
//   and the two half-waves combine to identical bits.
// Result: bit-identical rows to ffn_fused_kernel / ffn_wide_kernel (tests/test_gpu_parity.py asserts it), which still run
// the ragged end of a launch, the small launches and the fused q|k|v form.
//
// Weights.  The operand order of every MFMA is fixed, so the weights are packed ONCE per layer into the exact image the
// LDS ring wants (launch_ffn_pair_pack: 32-KiB stages = [half of wave class 0 | half of class 1], a half = 16 slabs of
// 1 KiB, a slab = lane l's four A operands of four consecutive MFMAs at 16 l): the LDS-DMA source of a piece is 1 KiB of
// contiguous memory and the ds_read_b128 of a slab is lane-linear (conflict-free by construction).  One stage = 64 MFMAs
// per wave; stage g+2 streams in while stage g is multiplied (3-stage ring, 96 KiB; one counted s_waitcnt + s_barrier
// per stage).  Feed-forward stage (I, ph): the W1 operands of chunk I (32 hidden units), k half ph, and the W2 operands of
// chunk I - 1, output tiles 2 ph and 2 ph + 1: GEMM1 of chunk I and GEMM2 of chunk I - 1 interleave MFMA by MFMA.
#include <mutex>

#include "common.h"

namespace cone {

typedef float f32x4p __attribute__((ext_vector_type(4)));
typedef float f32x2p __attribute__((ext_vector_type(2)));
typedef float f32x16p __attribute__((ext_vector_type(16)));

#define FP_GLDS16(src, dst) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

#ifndef FP_X_NOEX
#define FP_X_NOEX 0
#endif
constexpr int FP_STAGE = 8192;          // floats per ring stage (32 KiB)
constexpr int FP_NST = 3;               // ring depth
constexpr int FP_NPIECE = 4;            // 1-KiB LDS-DMA pieces per wave and stage

struct FfnPairArgs {
    const float* img;                                 // launch_ffn_pair_pack's image
    const float* A; int lda; const float* R; int ldr; // attention rows, residual rows (PROJ)
    const int* r_idx; const float* R2;                // gathered residual (ffn.hip's FfnArgs)
    const float* X; int ldx;                          // block input (!PROJ)
    const float *bo, *pg, *pb, *b1, *b2, *ln_g, *ln_b;
    float* OUT; int ldo;
    int M; const int* M_dev; int ff;
};

// channel of register (q, comp) of lane half `hi` in wave class `hf`: comp = 2 a + e
__host__ __device__ inline int fp_channel(int hf, int hi, int q, int comp) {
    return 16 * q + 8 * (comp >> 1) + 4 * hi + 2 * hf + (comp & 1);
}
// output channel of M-row m of the wave's tile t: the channel of register rho = (m & 3) + 4 (m >> 3), half-wave (m >> 2) & 1
__host__ __device__ inline int fp_row_channel(int hf, int t, int m) {
    return fp_channel(hf, (m >> 2) & 1, 4 * t + (m >> 3), m & 3);
}

// value (q, comp) of a 32 x 128 register tile held as four accumulator tiles: q = 4 t + rho / 4, comp = rho % 4
#define FP_V(v, q, c) v[(q) >> 2][4 * ((q) & 3) + (c)]

// A token's moments over 256 channels: 64 registers x 2 half-waves x 2 waves.  Association (shared with ffn_layernorm_regs):
// T[lg][p] = sum over q of (x[16 q + 4 lg + 2 p] + x[.. + 1]) in q order; U[lg] = T[lg][0] + T[lg][1];
// total = (U[lg] + U[lg ^ 1]) + (U[lg ^ 2] + U[lg ^ 3]).  Here lg = 2 a + hi, p = hf.
__device__ __forceinline__ float fp_pair_total(float t0, float t1, float* buf, int wave, int lane) {
    reinterpret_cast<f32x2p*>(buf)[wave * 64 + lane] = f32x2p{t0, t1};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const f32x2p o = reinterpret_cast<const f32x2p*>(buf)[(wave ^ 4) * 64 + lane];
    const float u0 = t0 + o.x, u1 = t1 + o.y;
    const float v0 = u0 + __shfl_xor(u0, 32, 64), v1 = u1 + __shfl_xor(u1, 32, 64);
    return v0 + v1;
}
__device__ __forceinline__ void fp_layernorm(f32x16p (&v)[4], float& rstd, float* buf_a, float* buf_b, int wave, int lane) {
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { t0 += FP_V(v, q, 0) + FP_V(v, q, 1); t1 += FP_V(v, q, 2) + FP_V(v, q, 3); }
    const float mean = fp_pair_total(t0, t1, buf_a, wave, lane) * (1.0f / 256.0f);
    float c0 = 0.f, c1 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        FP_V(v, q, 0) -= mean; FP_V(v, q, 1) -= mean; FP_V(v, q, 2) -= mean; FP_V(v, q, 3) -= mean;
        c0 = fmaf(FP_V(v, q, 0), FP_V(v, q, 0), c0); c0 = fmaf(FP_V(v, q, 1), FP_V(v, q, 1), c0);
        c1 = fmaf(FP_V(v, q, 2), FP_V(v, q, 2), c1); c1 = fmaf(FP_V(v, q, 3), FP_V(v, q, 3), c1);
    }
    const float s2 = fp_pair_total(c0, c1, buf_b, wave, lane);
    rstd = 1.0f / sqrtf(s2 * (1.0f / 256.0f) + 1e-5f);
}

template <bool PROJ>
__global__ __launch_bounds__(512, 2) void ffn_pair_kernel(FfnPairArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ff = p.ff, ncw = ff >> 5;
    float* b1s = smem + FP_NST * FP_STAGE;
    float* prm = b1s + ff;                          // 6 x 256, permuted: [vector][hf][hi][q][comp]
    float* hx = prm + 6 * 256;                      // [wave][rho / 4][lane][4]: a wave's partial hidden tile
    float* lnx = hx + 8 * 1024;                     // two buffers of [wave][lane][2]
    int M = p.M;
    if (p.M_dev) { const int md = *p.M_dev; M = md < M ? md : M; }
    const int n_tiles = (M + 127) >> 7;
    if ((int)blockIdx.x >= n_tiles) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tg = wave & 3, hf = wave >> 2;
    const int l32 = lane & 31, hi = lane >> 5;
    constexpr int NP = PROJ ? 8 : 0;
    const int G = NP + 2 * (ncw + 1) + 1;           // stages per tile (the feed-forward block: a prologue stage + 2 ncw + 2 phases); the ring runs on across tiles

    for (int i = tid; i < (ff >> 2); i += 512) reinterpret_cast<f32x4p*>(b1s)[i] = reinterpret_cast<const f32x4p*>(p.b1)[i];
    for (int i = tid; i < (PROJ ? 6 : 3) * 256; i += 512) {
        const int v = i >> 8, x = i & 255;
        const int c = fp_channel(x >> 7, (x >> 6) & 1, (x >> 2) & 15, x & 3);
        const float* src = v == 0 ? p.b2 : v == 1 ? p.ln_g : v == 2 ? p.ln_b : v == 3 ? p.bo : v == 4 ? p.pg : p.pb;
        prm[i] = src[c];
    }
    const float* myprm = prm + (hf * 2 + hi) * 64;  // + vector * 256 + 4 q: this lane's four values of register group q

    int sb = 0;
    const unsigned lane_off = 16u * lane;           // bytes
    auto stream_piece = [&](int g, int i) {
        const int gg = g < G ? g : g - G;           // past the tile: the next tile's first stages (the same image)
        float* dstp = smem + ((sb + g) % FP_NST) * FP_STAGE + (4 * wave + i) * 256;
        // wave-uniform base (pinned in SGPRs at this point: hoisted, base + lane offset becomes a spilled 64-bit VGPR pair
        // per piece and every reload drains the DMA queue -- ffn.hip) + ONE 32-bit lane offset
        const char* ub = reinterpret_cast<const char*>(p.img + (size_t)gg * FP_STAGE + (4 * wave + i) * 256);
        asm volatile("" : "+s"(ub));
#ifndef FP_X_NODMA
        FP_GLDS16(ub + lane_off, dstp);
#endif
    };
#define FP_SB() __builtin_amdgcn_sched_barrier(0)
#ifdef FP_X_NOBAR
#define FP_X_BARRIER()
#else
#define FP_X_BARRIER() __builtin_amdgcn_s_barrier()
#endif
#define FP_STAGE_OF(g) (smem + ((sb + (g)) % FP_NST) * FP_STAGE + hf * 4096 + 4 * lane)
#define FP_RD(stg, sl) (*reinterpret_cast<const f32x4p*>((stg) + (sl) * 256))
#define FP_STREAM(g, u) { if ((u) & 1) stream_piece(g, (u) >> 1); }
#define FP_END_STAGE()                                                            \
    {                                                                             \
        FP_SB();                                                                  \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FP_NPIECE) : "memory");          \
        __builtin_amdgcn_s_barrier();                                             \
        FP_SB();                                                                  \
    }
    // MFMA issue order is pinned (ffn.hip): an accumulator is reused two MFMAs (128 cycles) later at the earliest
#define FP_MFMA(acc, a, b) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0); FP_SB(); }

#pragma unroll
    for (int i = 0; i < FP_NPIECE; ++i) stream_piece(0, i);
#pragma unroll
    for (int i = 0; i < FP_NPIECE; ++i) stream_piece(1, i);

    // register tiles (value (q, comp) of FP_V = channel fp_channel(hf, hi, q, comp) of token l32):
    //   xr = residual of the projection -> block input = B operand of GEMM1 and residual of the block
    //   at = the attention rows, all 256 channels: at[2 q + a][r] = A[tok][16 q + 8 a + 4 hi + r]  (PROJ)
    f32x16p xr[4];
    f32x4p at[PROJ ? 32 : 1];
    auto tile_row = [&](int tile) { const int row = tile * 128 + 32 * tg + l32; return (size_t)(row < M ? row : M - 1); };
    auto load_in = [&](int tile) {
        const size_t row = tile_row(tile);
        if (PROJ) {
            const float* ap = p.A + row * p.lda + 4 * hi;
#pragma unroll
            for (int i = 0; i < 32; ++i) at[i] = *reinterpret_cast<const f32x4p*>(ap + 8 * i);
        } else {
            const float* xp = p.X + row * p.ldx + 4 * hi + 2 * hf;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const f32x2p lo = *reinterpret_cast<const f32x2p*>(xp + 16 * q), hi2 = *reinterpret_cast<const f32x2p*>(xp + 16 * q + 8);
                FP_V(xr, q, 0) = lo.x; FP_V(xr, q, 1) = lo.y; FP_V(xr, q, 2) = hi2.x; FP_V(xr, q, 3) = hi2.y;
            }
        }
    };
    auto load_res = [&](int tile) {
        const size_t row = tile_row(tile);
        const float* rp = p.R + row * p.ldr;
        if (p.r_idx) {
            const int ix = p.r_idx[row];
            rp = ix >= 0 ? p.R + (size_t)ix * p.ldr : p.R2 + (size_t)(~ix) * p.ldr;
        }
        rp += 4 * hi + 2 * hf;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const f32x2p lo = *reinterpret_cast<const f32x2p*>(rp + 16 * q), hi2 = *reinterpret_cast<const f32x2p*>(rp + 16 * q + 8);
            FP_V(xr, q, 0) = lo.x; FP_V(xr, q, 1) = lo.y; FP_V(xr, q, 2) = hi2.x; FP_V(xr, q, 3) = hi2.y;
        }
    };
    load_in(blockIdx.x);
    bool first = true;
    f32x4p wa, na, va, nva;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int my_row = tile * 128 + 32 * tg + l32;
        if (first) {        // stage 0 and the parameter images are in LDS (later tiles: the previous tile's last barrier)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FP_NPIECE) : "memory");
            __syncthreads();
            first = false;
        }
        if (PROJ) {
            load_res(tile);
            // ---- output projection: output tile t of this wave (32 channels) = stages 2 t (k half 0) and 2 t + 1.  Slab sl of k half
            // kh: q = 8 kh + sl / 2, steps j: r = 2 (sl % 2) + j % 2, a = j / 2, chain j % 2 (ffn.hip's ha[r & 1])
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x16p pa, pb;
#pragma unroll
                for (int r = 0; r < 16; ++r) { pa[r] = 0.f; pb[r] = 0.f; }
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    const float* st = FP_STAGE_OF(2 * t + kh);
                    wa = FP_RD(st, 0);
#pragma unroll
                    for (int sl = 0; sl < 16; ++sl) {
                        FP_SB();
                        asm volatile("" : "+v"(wa));
                        FP_SB();
                        if (sl < 15) na = FP_RD(st, sl + 1);
                        FP_SB();
                        const int q = 8 * kh + (sl >> 1), r0 = 2 * (sl & 1);
                        FP_MFMA(pa, wa[0], at[2 * q][r0])
                        FP_MFMA(pb, wa[1], at[2 * q][r0 + 1])
                        FP_MFMA(pa, wa[2], at[2 * q + 1][r0])
                        FP_MFMA(pb, wa[3], at[2 * q + 1][r0 + 1])
                        if ((sl & 3) == 3) stream_piece(2 * t + kh + 2, sl >> 2);
                        if (sl < 15) wa = na;
                    }
                    if (kh == 1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) xr[t][r] += pa[r] + pb[r];
                    }
                    FP_END_STAGE()
                }
            }
            FP_SB();
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const f32x4p b4 = *reinterpret_cast<const f32x4p*>(myprm + 3 * 256 + 4 * q);
#pragma unroll
                for (int c = 0; c < 4; ++c) FP_V(xr, q, c) += b4[c];
            }
            float rstd;
            fp_layernorm(xr, rstd, lnx, lnx + 1024, wave, lane);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const f32x4p g4 = *reinterpret_cast<const f32x4p*>(myprm + 4 * 256 + 4 * q);
                const f32x4p b4 = *reinterpret_cast<const f32x4p*>(myprm + 5 * 256 + 4 * q);
#pragma unroll
                for (int c = 0; c < 4; ++c) FP_V(xr, q, c) = FP_V(xr, q, c) * rstd * g4[c] + b4[c];
            }
            FP_SB();
        }

        // ---- feed-forward block.  A chunk = 32 hidden units = 16 GEMM1 slabs (k groups q = 0 .. 15 of this wave's k class) and 16
        // GEMM2 slabs (output tile t = 2 (yy / 8) + yy % 2, hidden quarter (yy / 2) % 4 at yy = 0 .. 15).  The feed-forward stages of a
        // tile: a 2-unit prologue stage (GEMM1 slabs 0, 1 of chunk 0), then phases P = 0 .. 2 ncw + 1 of 8 units; unit (P, u) runs
        // GEMM1 slab x = 8 P + u + 2 (chunk x / 16) and GEMM2 slab y = 8 P + u - 16 (chunk y / 16), MFMA by MFMA.  GEMM1 runs two
        // units ahead so that a chunk's partial tile is complete after unit 5 of an odd phase: the pair exchanges it during unit
        // 6 (send) and unit 7 (receive), across the phase's ONE barrier, which sits at the top of unit 7 -- every wave holds its
        // last fragments of the stage in registers by then, so the stage is free, the next stage is visible, and unit 7 already
        // requests the first fragments of the next phase: no LDS round trip is exposed at a stage boundary.
        f32x16p y[4], hpa, hpb, h;
        f32x16p zero16;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) y[t] = zero16;
        f32x4p pown[4], oth[4];             // this wave's partial hidden tile between send and receive; the partner's
#define FP_MFMA0(acc, a, b) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, zero16, 0, 0, 0); FP_SB(); }
#define FP_RHO(u, j) ((2 * (((u) >> 1) & 1) + ((j) >> 1)) + 4 * (((j) & 1) + 2 * ((u) >> 2)))
#define FP_YT(ph, u) y[2 * (ph) + ((u) & 1)]
#define FP_G1(s)                                                                              \
        {                                                                                     \
            FP_MFMA(hpa, wa[0], FP_V(xr, s, 0)) FP_MFMA(hpb, wa[1], FP_V(xr, s, 1))           \
            FP_MFMA(hpa, wa[2], FP_V(xr, s, 2)) FP_MFMA(hpb, wa[3], FP_V(xr, s, 3))           \
        }
#define FP_G1_NEW(s)    /* first slab of a chunk: the chains start from zero */               \
        {                                                                                     \
            FP_MFMA0(hpa, wa[0], FP_V(xr, s, 0)) FP_MFMA0(hpb, wa[1], FP_V(xr, s, 1))         \
            FP_MFMA(hpa, wa[2], FP_V(xr, s, 2)) FP_MFMA(hpb, wa[3], FP_V(xr, s, 3))           \
        }
#define FP_G2(ph, u)                                                                          \
        {                                                                                     \
            FP_MFMA(FP_YT(ph, u), va[0], h[FP_RHO(u, 0)]) FP_MFMA(FP_YT(ph, u), va[1], h[FP_RHO(u, 1)]) \
            FP_MFMA(FP_YT(ph, u), va[2], h[FP_RHO(u, 2)]) FP_MFMA(FP_YT(ph, u), va[3], h[FP_RHO(u, 3)]) \
        }
#define FP_G12(ph, u, s)                                                                      \
        {                                                                                     \
            FP_MFMA(hpa, wa[0], FP_V(xr, s, 0)) FP_MFMA(FP_YT(ph, u), va[0], h[FP_RHO(u, 0)]) \
            FP_MFMA(hpb, wa[1], FP_V(xr, s, 1)) FP_MFMA(FP_YT(ph, u), va[1], h[FP_RHO(u, 1)]) \
            FP_MFMA(hpa, wa[2], FP_V(xr, s, 2)) FP_MFMA(FP_YT(ph, u), va[2], h[FP_RHO(u, 2)]) \
            FP_MFMA(hpb, wa[3], FP_V(xr, s, 3)) FP_MFMA(FP_YT(ph, u), va[3], h[FP_RHO(u, 3)]) \
        }
        // the partial tile out: P = c_{2hf} + c_{2hf+1} (kept in pown as well)
#define FP_SEND()                                                                             \
        {                                                                                     \
            float* dst_ = hx + wave * 1024 + 4 * lane;                                        \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                   \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) pown[i][c] = hpa[4 * i + c] + hpb[4 * i + c]; \
                *reinterpret_cast<f32x4p*>(dst_ + i * 256) = pown[i];                         \
            }                                                                                 \
        }
        // ... and the pair's tile in: h = relu((P_0 + P_1) + b1).  The partner's tile is requested ahead of unit 7's MFMAs, the sums
        // run behind them, b1 comes last (register budget: the three operands are never live together)
#define FP_RECV_ISSUE()                                                                       \
        {                                                                                     \
            const float* oth_ = hx + (wave ^ 4) * 1024 + 4 * lane;                            \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) oth[i] = *reinterpret_cast<const f32x4p*>(oth_ + i * 256); \
        }
#define FP_RECV(I)                                                                            \
        {                                                                                     \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) oth[i] += pown[i];                  \
            FP_SB();                                                                          \
            const float* bb_ = b1s + 32 * (I) + 4 * hi;                                       \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) pown[i] = *reinterpret_cast<const f32x4p*>(bb_ + 8 * i); \
            FP_SB();                                                                          \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                     \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) h[4 * i + c] = fmaxf(oth[i][c] + pown[i][c], 0.f); \
            FP_SB();                                                                          \
        }
        // One phase.  D1 / D2: GEMM1 / GEMM2 present (D1LIM: GEMM1 only in units < D1LIM); EX: the pair exchange of an odd phase;
        // NX2: the NEXT phase has GEMM2; NX1: ... GEMM1.  gi = this phase's stage index.
#define FP_PHASE(gi, PH, D1LIM, D2, EX, NX1, NX2, ICH)                                        \
        {                                                                                     \
            const float* st = FP_STAGE_OF(gi);                                                \
            const float* stn = FP_STAGE_OF((gi) + 1);                                         \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                   \
                const bool d1 = u < (D1LIM), n1 = u < 7 ? (u + 1 < (D1LIM)) : (NX1), n2 = u < 7 ? (D2) : (NX2); \
                FP_SB();                                                                      \
                if (d1 && (D2)) asm volatile("" : "+v"(wa), "+v"(va));                        \
                else if (d1) asm volatile("" : "+v"(wa));                                     \
                else if (D2) asm volatile("" : "+v"(va));                                     \
                FP_SB();                                                                      \
                if (u == 7) {       /* the stage switch: three of this phase's pieces are in flight */ \
                    asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");               \
                    FP_X_BARRIER();                                                           \
                    FP_SB();                                                                  \
                }                                                                             \
                {                                                                             \
                    const float* sn_ = u == 7 ? stn : st;                                     \
                    if (n1) na = FP_RD(sn_, (u + 1) & 7);                                     \
                    if (n2) nva = FP_RD(sn_, 8 + ((u + 1) & 7));                              \
                }                                                                             \
                if ((EX) && u == 7 && !FP_X_NOEX) FP_RECV_ISSUE()                             \
                FP_SB();                                                                      \
                if ((EX) && u == 6) {                                                         \
                    if (D2) FP_G2(PH, u)                                                      \
                    if (!FP_X_NOEX) FP_SEND()                                                 \
                    FP_SB();                                                                  \
                    if (d1) FP_G1_NEW((8 * (PH) + u + 2) & 15)                                \
                } else if (d1 && (D2)) FP_G12(PH, u, (8 * (PH) + u + 2) & 15)                 \
                else if (d1) FP_G1((8 * (PH) + u + 2) & 15)                                   \
                else if (D2) FP_G2(PH, u)                                                     \
                if (u & 1) stream_piece((gi) + 2, u >> 1);                                    \
                if ((EX) && u == 7 && !FP_X_NOEX) FP_RECV(ICH)                                \
                if (n1) wa = na;                                                              \
                if (n2) va = nva;                                                             \
            }                                                                                 \
        }
        {   // prologue stage: GEMM1 slabs 0 and 1 of chunk 0
            const float* st = FP_STAGE_OF(NP);
            const float* stn = FP_STAGE_OF(NP + 1);
            wa = FP_RD(st, 0);
            FP_SB();
            asm volatile("" : "+v"(wa));
            FP_SB();
            na = FP_RD(st, 1);
            FP_SB();
            FP_G1_NEW(0)
#pragma unroll
            for (int i = 0; i < FP_NPIECE; ++i) stream_piece(NP + 2, i);
            wa = na;
            FP_SB();
            asm volatile("" : "+v"(wa));
            FP_SB();
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(FP_NPIECE) : "memory");
            __builtin_amdgcn_s_barrier();
            FP_SB();
            na = FP_RD(stn, 0);
            FP_SB();
            FP_G1(1)
            wa = na;
        }
        FP_PHASE(NP + 1, 0, 8, false, false, true, false, 0)                 // P = 0
        FP_PHASE(NP + 2, 1, 8, false, true, true, true, 0)                   // P = 1: chunk 0 goes to the pair
        for (int I = 1; I + 1 < ncw; ++I) {
            FP_PHASE(NP + 1 + 2 * I, 0, 8, true, false, true, true, 0)
            FP_PHASE(NP + 2 + 2 * I, 1, 8, true, true, true, true, I)
        }
        FP_PHASE(NP + 1 + 2 * (ncw - 1), 0, 8, true, false, true, true, 0)
        FP_PHASE(NP + 2 + 2 * (ncw - 1), 1, 6, true, true, false, true, ncw - 1)   // the last chunk's GEMM1 ends at unit 5
        FP_PHASE(NP + 1 + 2 * ncw, 0, 0, true, false, false, true, 0)
        FP_PHASE(NP + 2 + 2 * ncw, 1, 0, true, false, false, false, 0)

        // ---- epilogue: + b2 + residual, LayerNorm, store.  The next tile's rows are requested ahead of the stores.
        FP_SB();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const f32x4p b4 = *reinterpret_cast<const f32x4p*>(myprm + 4 * q);
#pragma unroll
            for (int c = 0; c < 4; ++c) FP_V(y, q, c) = FP_V(y, q, c) + b4[c] + FP_V(xr, q, c);
        }
        FP_SB();
        load_in(tile + (int)gridDim.x < n_tiles ? tile + (int)gridDim.x : tile);
        FP_SB();
        float rstd;
        fp_layernorm(y, rstd, lnx, lnx + 1024, wave, lane);
        if (my_row < M) {
            float* op = p.OUT + (size_t)my_row * p.ldo + 4 * hi + 2 * hf;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const f32x4p g4 = *reinterpret_cast<const f32x4p*>(myprm + 256 + 4 * q);
                const f32x4p b4 = *reinterpret_cast<const f32x4p*>(myprm + 512 + 4 * q);
                f32x4p o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = FP_V(y, q, c) * rstd * g4[c] + b4[c];
                *reinterpret_cast<f32x2p*>(op + 16 * q) = f32x2p{o[0], o[1]};
                *reinterpret_cast<f32x2p*>(op + 16 * q + 8) = f32x2p{o[2], o[3]};
            }
        }
        sb = (sb + G) % FP_NST;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // no LDS-DMA may outlive the workgroup's LDS
}

// ---- the weight image.  One thread per float: stage g, wave class hf, slab sl, lane l, operand j.
__global__ void ffn_pair_pack_kernel(const float* __restrict__ Wo, const float* __restrict__ W1, const float* __restrict__ W2,
                                     int ff, float* __restrict__ img, int n_stages) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n_stages * FP_STAGE) return;
    const int j = idx & 3, l = (idx >> 2) & 63, sl = (idx >> 8) & 15, hf = (idx >> 12) & 1, g = (int)(idx >> 13);
    const int l32 = l & 31, hi = l >> 5, ncw = ff >> 5;
    const int np = Wo ? 8 : 0;
    float v = 0.f;
    if (g < np) {
        const int t = g >> 1, q = 8 * (g & 1) + (sl >> 1), r = 2 * (sl & 1) + (j & 1), a = j >> 1;
        v = Wo[(size_t)fp_row_channel(hf, t, l32) * 256 + 16 * q + 8 * a + 4 * hi + r];
    } else {
        const int gi = g - np;          // 0: the prologue stage (GEMM1 slabs 0, 1 of chunk 0); 1 + P: phase P
        if (gi == 0) {
            if (sl < 2) v = W1[(size_t)l32 * 256 + 16 * sl + 8 * (j >> 1) + 4 * hi + 2 * hf + (j & 1)];
        } else if (sl < 8) {
            const int x = 8 * (gi - 1) + sl + 2, I = x >> 4, q = x & 15, a = j >> 1, e = j & 1;
            if (I < ncw) v = W1[(size_t)(32 * I + l32) * 256 + 16 * q + 8 * a + 4 * hi + 2 * hf + e];
        } else {
            const int yv = 8 * (gi - 1) + (sl - 8) - 16;
            if (yv >= 0 && (yv >> 4) < ncw) {
                const int J = yv >> 4, yy = yv & 15, u = yy & 7;
                const int t = 2 * (yy >> 3) + (u & 1), cc = u >> 2, rh = (u >> 1) & 1, r = 2 * rh + (j >> 1), a = j & 1;
                v = W2[(size_t)fp_row_channel(hf, t, l32) * ff + 32 * J + 16 * cc + 8 * a + 4 * hi + r];
            }
        }
    }
    img[idx] = v;
}

bool ffn_pair_supported(int ff) { return ff >= 64 && ff % 32 == 0 && ff <= 4096; }
static int fp_stages(int ff, bool proj) { return (proj ? 8 : 0) + 2 * ((ff >> 5) + 1) + 1; }
size_t ffn_pair_image_bytes(int ff, bool proj) { return (size_t)fp_stages(ff, proj) * FP_STAGE * sizeof(float); }

int launch_ffn_pair_pack(const float* Wo, const float* W1, const float* W2, int ff, float* img, hipStream_t s) {
    CONE_REQUIRE(ffn_pair_supported(ff) && W1 && W2 && img, "pair tail image: dim_feedforward=%d unsupported or null argument", ff);
    const int n_stages = fp_stages(ff, Wo != nullptr);
    const size_t n = (size_t)n_stages * FP_STAGE;
    hipLaunchKernelGGL(ffn_pair_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Wo, W1, W2, ff, img, n_stages);
    CONE_LAUNCH_CHECK();
    return 0;
}

static size_t fp_lds_bytes(int ff) { return (size_t)(FP_NST * FP_STAGE + ff + 6 * 256 + 8 * 1024 + 2 * 1024) * sizeof(float); }

template <bool PROJ>
static int launch_pair_t(const FfnPairArgs& a, hipStream_t s) {
    const size_t lds = fp_lds_bytes(a.ff);
    CONE_REQUIRE(lds <= 160 * 1024, "pair tail: %zu bytes of LDS (ff %d) exceed 160 KiB", lds, a.ff);
    static DeviceOnce once;
    int n_cu = 0;
    CONE_CHECK_HIP(device_once(once, [] {
        return hipFuncSetAttribute((const void*)ffn_pair_kernel<PROJ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }, &n_cu));
    const int tiles = (a.M + 127) / 128;
    const int grid = tiles < n_cu ? tiles : n_cu;
    ProfScope ps(PROJ ? PK_FFN_PAIR_PROJ : PK_FFN_PAIR, a.M, a.ff, 256, a.M_dev, s);
    hipLaunchKernelGGL((ffn_pair_kernel<PROJ>), dim3((unsigned)grid), dim3(512), lds, s, a);
    CONE_LAUNCH_CHECK();
    return 0;
}

bool ffn_pair_fits(int ff) { return ffn_pair_supported(ff) && fp_lds_bytes(ff) <= 160 * 1024; }

int launch_ffn_pair(const float* X, int ldx, const float* img, const float* b1, const float* b2, const float* ln_g,
                    const float* ln_b, float* OUT, int ldo, int M, const int* M_dev, int ff, hipStream_t s) {
    CONE_REQUIRE(ffn_pair_fits(ff) && X && img && b1 && b2 && ln_g && ln_b && OUT, "pair FFN: unsupported shape or null argument");
    CONE_REQUIRE(ldx % 2 == 0 && ldo % 2 == 0, "pair FFN: row strides must be even");
    if (M <= 0) return 0;
    FfnPairArgs a{};
    a.img = img; a.X = X; a.ldx = ldx; a.b1 = b1; a.b2 = b2; a.ln_g = ln_g; a.ln_b = ln_b; a.OUT = OUT; a.ldo = ldo;
    a.M = M; a.M_dev = M_dev; a.ff = ff;
    return launch_pair_t<false>(a, s);
}

int launch_proj_ffn_pair(const float* A, int lda, const float* img, const float* bo, const float* R, int ldr, const float* pg,
                         const float* pb, const float* b1, const float* b2, const float* ln_g, const float* ln_b, float* OUT,
                         int ldo, int M, const int* M_dev, int ff, hipStream_t s, const int* r_idx, const float* R2) {
    CONE_REQUIRE(ffn_pair_fits(ff) && A && img && bo && R && pg && pb && b1 && b2 && ln_g && ln_b && OUT,
                 "pair layer tail: unsupported shape or null argument");
    CONE_REQUIRE(lda % 4 == 0 && ldr % 2 == 0 && ldo % 2 == 0, "pair layer tail: bad row strides");
    CONE_REQUIRE(!r_idx || R2, "pair layer tail: a gathered residual needs both source matrices");
    if (M <= 0) return 0;
    FfnPairArgs a{};
    a.img = img; a.A = A; a.lda = lda; a.R = R; a.ldr = ldr; a.r_idx = r_idx; a.R2 = R2; a.bo = bo; a.pg = pg; a.pb = pb;
    a.b1 = b1; a.b2 = b2; a.ln_g = ln_g; a.ln_b = ln_b; a.OUT = OUT; a.ldo = ldo; a.M = M; a.M_dev = M_dev; a.ff = ff;
    return launch_pair_t<true>(a, s);
}

}  // namespace cone
