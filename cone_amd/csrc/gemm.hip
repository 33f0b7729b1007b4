// fp32 GEMM on the gfx950 matrix cores:  C = epi((A [+ A2]) . W^T + bias)
//
// Every dense contraction of the Moment-DETR window model goes through this kernel
// (nn.Linear / in_proj / out_proj / FFN of cone/transformer.py:205-317, input projections
// and heads of cone/model.py:58-73,112-115, adapter MLP cone/model.py:428-440).
//
// Numerics: v_mfma_f32_32x32x2_f32 is an exact-fp32, k-ordered fma chain, so results agree with
// the reference's fp32 addmm to ~1e-7 relative (only the summation order differs).
//
// Tiling: 256 threads = 4 wavefronts, each owning a 64x64 output tile (2x2 MFMA tiles of 32x32,
// 64 accumulator VGPRs).  Block tile <BM,BN> in {<128,128>, <64,256>}; the latter owns whole
// 256-wide rows so that residual-add + LayerNorm fuse into the epilogue.
// K is consumed in BK=32 slabs staged through registers into a k-major LDS image
// [k][row] with an odd row stride: the transposing ds_write_b32 and the lane==row
// ds_read_b32 operand fetches are both bank-conflict free.  Global loads of slab t+1 are
// issued before the MFMAs of slab t (register prefetch); one LDS buffer, two barriers per slab,
// 2-4 workgroups per CU cover each other's barriers.
#include <mutex>

#include "common.h"
#include "rows_chain.h"

#ifndef CONE_GEMM_NT_STORE
#define CONE_GEMM_NT_STORE 0
#endif

namespace cone {

constexpr int BK = 32;

template <int BM, int BN, bool HAS_A2>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs p) {
    constexpr int WAVES_N = BN / 64;
    constexpr int LDA_S = BM + 1, LDB_S = BN + 1;
    constexpr int A_PASSES = BM / 32, B_PASSES = BN / 32;
    __shared__ float smem[BK * LDA_S + BK * LDB_S];
    float* As = smem;
    float* Bs = smem + BK * LDA_S;

    int M = p.M;
    if (p.M_dev) { int md = *p.M_dev; M = md < M ? md : M; }
    const int m0 = blockIdx.x * BM;
    if (m0 >= M) return;
    const int n0 = blockIdx.y * BN;
    const int N = p.N, K = p.K;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, lh = lane >> 5;

    // global -> register staging map: 8 lanes cover one 128-byte row slab
    const int lrow = tid >> 3;  // 0..31
    const int lkc = tid & 7;    // float4 index inside the slab

    float4 ra[A_PASSES], rb[B_PASSES];

    // Rows past M / N are clamped to the last valid row instead of predicated: they only feed
    // output elements that are never stored, and unconditional loads keep the address math scalar.
    auto gload = [&](int kt) {
        const int kof = kt * BK + lkc * 4;
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) {
            int row = m0 + lrow + 32 * ps;
            row = row < M ? row : M - 1;
            float4 v = *reinterpret_cast<const float4*>(p.A + (size_t)row * p.lda + kof);
            if (HAS_A2) {
                const int r2 = p.a2_mod ? row % p.a2_mod : row;
                const float4 w = *reinterpret_cast<const float4*>(p.A2 + (size_t)r2 * p.lda2 + kof);
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            ra[ps] = v;
        }
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps) {
            int n = n0 + lrow + 32 * ps;
            n = n < N ? n : N - 1;
            rb[ps] = *reinterpret_cast<const float4*>(p.W + (size_t)n * p.ldw + kof);
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) {
            float* d = As + (lkc * 4) * LDA_S + lrow + 32 * ps;
            d[0] = ra[ps].x; d[LDA_S] = ra[ps].y; d[2 * LDA_S] = ra[ps].z; d[3 * LDA_S] = ra[ps].w;
        }
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps) {
            float* d = Bs + (lkc * 4) * LDB_S + lrow + 32 * ps;
            d[0] = rb[ps].x; d[LDB_S] = rb[ps].y; d[2 * LDB_S] = rb[ps].z; d[3 * LDB_S] = rb[ps].w;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / BK;
    gload(0);
    const float* a_rd = As + lh * LDA_S + wm * 64 + li;
    const float* b_rd = Bs + lh * LDB_S + wn * 64 + li;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        swrite();
        __syncthreads();
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const float a0 = a_rd[(2 * kk) * LDA_S];
            const float a1 = a_rd[(2 * kk) * LDA_S + 32];
            const float b0 = b_rd[(2 * kk) * LDB_S];
            const float b1 = b_rd[(2 * kk) * LDB_S + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    // ------------------------------------------------------------------ epilogue
    const int flags = p.flags;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int n = n0 + wn * 64 + tn * 32 + li;
            const float bv = (p.bias && n < N) ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + tm * 32 + acc_row(r, lane);
                float v = acc[tm][tn][r] + bv;
                if (flags & EPI_RELU) v = fmaxf(v, 0.f);
                if ((flags & EPI_RESIDUAL) && m < M && n < N) v += p.R[(size_t)(p.r_mod ? m % p.r_mod : m) * p.ldr + n];
                acc[tm][tn][r] = v;
            }
        }

    if constexpr (BN == 256) {
        if (flags & EPI_LN) {
            // whole rows live in this block: 4 waves x 64 columns.  Two-pass moments, one
            // 32-row half at a time to keep register pressure down.
            float* red = smem;  // [32 rows][4 waves]
            const float g0 = p.ln_g[wn * 64 + li], g1 = p.ln_g[wn * 64 + 32 + li];
            const float be0 = p.ln_b[wn * 64 + li], be1 = p.ln_b[wn * 64 + 32 + li];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                float mean[16];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float s = half_sum(acc[tm][0][r] + acc[tm][1][r]);
                    if (li == 0) red[acc_row(r, lane) * 4 + wn] = s;
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* q = red + acc_row(r, lane) * 4;
                    mean[r] = ((q[0] + q[1]) + (q[2] + q[3])) * (1.0f / 256.0f);
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d0 = acc[tm][0][r] - mean[r];
                    const float d1 = acc[tm][1][r] - mean[r];
                    const float s = half_sum(d0 * d0 + d1 * d1);
                    if (li == 0) red[acc_row(r, lane) * 4 + wn] = s;
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* q = red + acc_row(r, lane) * 4;
                    const float var = ((q[0] + q[1]) + (q[2] + q[3])) * (1.0f / 256.0f);
                    const float rstd = 1.0f / sqrtf(var + 1e-5f);
                    acc[tm][0][r] = (acc[tm][0][r] - mean[r]) * rstd * g0 + be0;
                    acc[tm][1][r] = (acc[tm][1][r] - mean[r]) * rstd * g1 + be1;
                }
            }
        }
    }

#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int n = n0 + wn * 64 + tn * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + tm * 32 + acc_row(r, lane);
                if (m < M && n < N) p.C[(size_t)m * p.ldc + n] = acc[tm][tn][r];
            }
        }
}


// ------------------------------------------------------------------------------------------------
// Row-owning tile fed by LDS-DMA:  128 x 256 per workgroup, each of the 4 wavefronts owns 32 COMPLETE
// 256-wide rows (8 MFMA tiles, 128 accumulator registers; one wave per SIMD, 512-register budget).
//   * global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write): two 48 KiB buffers,
//     slab t+1 streams in while slab t is multiplied; the only wait is the vmcnt(0)+barrier that ends
//     a slab.  Each wave-instruction fills 8 rows x 128 B; LDS rows are unpadded and the 16-byte
//     chunk index is XOR-swizzled with (row>>1)&7 -- applied to the per-lane SOURCE address and to the
//     ds_read_b128 address (cdna_hip_programming.md rule 21) -- which makes every 16-lane service
//     group of the operand reads bank-conflict free.
//   * operands with ds_read_b128: four consecutive k per lane; the MFMA k index is permuted
//     (half-wave h takes k = 8s+4h+j), legal because A and B agree; it only reorders the fp32 sum.
//   * a wave holds whole rows, so bias / ReLU / residual / LayerNorm need no LDS and no barrier:
//     row moments are 8 in-lane adds, a 4-step DPP reduction and one cross-row exchange.
//   * optional second output C2 = C + ADD (the next layer's "x + pos" operand), so no GEMM on this
//     path needs a fused addend on its A operand.
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// Sum over the 32 lanes that share (lane >> 5); every lane of the half-wave gets the total.
__device__ __forceinline__ float half_sum_dpp(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror  -> all 16 lanes of a row hold the row sum
    v += __shfl_xor(v, 16, 64);  // the other row of this half-wave
    return v;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov_rows(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// Sum over all 64 lanes, returned wave-uniform.
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_mov<0xB1>(v);                 // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);                 // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);                // row_half_mirror
    v += dpp_mov<0x140>(v);                // row_mirror: every lane holds its 16-lane row sum
    v += dpp_mov_rows<0x142, 0xA>(v);      // row_bcast15 into rows 1,3
    v += dpp_mov_rows<0x143, 0xC>(v);      // row_bcast31 into rows 2,3: lanes 48-63 hold the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

constexpr int RT_BM = 128, RT_BN = 256;
constexpr int RT_EP_LD = RT_BN + 4;                           // epilogue image row stride (floats)
constexpr int RT_EP_FLOATS = 4 * 8 * RT_EP_LD;                // 4 waves x 8 rows at a time: 33 KiB

#define GLDS16(src, dst) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// RBK = k-depth of one LDS slab (32: one workgroup per CU, 96 KiB of slabs; 16: 48 KiB, two workgroups
// per CU so that one's prologue / epilogue / store drain overlaps the other's MFMAs).
template <int RBK>
struct RowTile {
    static constexpr int CH = RBK / 4;                               // 16-byte chunks per row
    static constexpr int ROWS_PER_INST = 64 / CH;                    // rows one wave-instruction fills
    static constexpr int A_INST = RT_BM / ROWS_PER_INST / 4;         // LDS-DMA instructions per wave, A
    static constexpr int B_INST = RT_BN / ROWS_PER_INST / 4;         //   ... W
    static constexpr int SLAB = (RT_BM + RT_BN) * RBK;               // floats
    static constexpr int NBUF = RBK == 16 ? 3 : 2;                   // slabs in the LDS ring
    static constexpr int LDS_FLOATS = (NBUF * SLAB > RT_EP_FLOATS) ? NBUF * SLAB : RT_EP_FLOATS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    // chunk swizzle: makes every 16-lane ds_read_b128 service group hit 16 distinct 16-byte slots
    __device__ static __forceinline__ int swz(int row) { return CH == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
};

template <int RBK>
__global__ __launch_bounds__(256, (RBK == 32 ? 1 : 2)) void gemm_rows_kernel(GemmArgs p) {
    using T = RowTile<RBK>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int M = p.M;
    if (p.M_dev) { int md = *p.M_dev; M = md < M ? md : M; }
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so the
    // column tiles of one 128-row A tile are given ids that differ by multiples of 8: they run close in time on
    // ONE XCD and the A tile is read from HBM once, not N/256 times (measured 8.2 GB -> 2.1 GB per FFN1 launch).
    const int nc = p.N / RT_BN;
    const int L = blockIdx.x, grp = L / (8 * nc), r = L % (8 * nc);
    const int m0 = (grp * 8 + (r & 7)) * RT_BM;
    if (m0 >= M) return;
    const int n0 = (r >> 3) * RT_BN;
    const int K = p.K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    // ---- LDS-DMA source offsets: lane L of an instruction -> row (L / CH), physical chunk (L % CH)
    const float* __restrict__ Ab = p.A + (size_t)m0 * p.lda;
    const float* __restrict__ Wb = p.W + (size_t)n0 * p.ldw;   // N % 256 == 0: no column clamp
    int a_off[T::A_INST], b_off[T::B_INST];
#pragma unroll
    for (int i = 0; i < T::A_INST; ++i) {
        const int row = (wave * T::A_INST + i) * T::ROWS_PER_INST + lane / T::CH;
        const int src_row = (m0 + row < M) ? row : (M - 1 - m0);   // rows past M feed unstored outputs
        a_off[i] = src_row * p.lda + (((lane % T::CH) ^ T::swz(row)) << 2);
    }
#pragma unroll
    for (int i = 0; i < T::B_INST; ++i) {
        const int row = (wave * T::B_INST + i) * T::ROWS_PER_INST + lane / T::CH;
        b_off[i] = row * p.ldw + (((lane % T::CH) ^ T::swz(row)) << 2);
    }
    // piece i of a slab: one wave-instruction (1 KiB) of this wave's share
    auto stream_piece = [&](int kt, int buf, int i) {
        float* base = smem + buf * T::SLAB;
        if (i < T::A_INST) GLDS16(Ab + a_off[i] + kt * RBK, base + (wave * T::A_INST + i) * 256);
        else GLDS16(Wb + b_off[i - T::A_INST] + kt * RBK, base + RT_BM * RBK + (wave * T::B_INST + i - T::A_INST) * 256);
    };
    constexpr int NPIECE = T::A_INST + T::B_INST;
    auto stream_slab = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) stream_piece(kt, buf, i);
    };

    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // operand read addresses: chunk (2s+lh) ^ swz(row); swz only depends on li for both operands
    constexpr int NS = RBK / 8;              // super-steps (8 k-values each) per slab
    constexpr int NU = NS * 4;               // units of {2 B fragments, 8 MFMAs}
    int cs[NS];
#pragma unroll
    for (int s4 = 0; s4 < NS; ++s4) cs[s4] = ((2 * s4 + lh) ^ T::swz(li)) << 2;
    const int a_row = (wave * 32 + li) * RBK;
    const int b_row = RT_BM * RBK + li * RBK;

    const int nk = K / RBK;
    // Ring of NBUF slabs.  With 3 slabs the LDS-DMA of slab kt+2 is issued while slab kt is multiplied and
    // is only waited for one slab later, with a COUNTED vmcnt and a raw s_barrier (a __syncthreads() would
    // drain it with vmcnt(0)): the DMA gets a full slab of MFMA time to land.
    constexpr int AHEAD = T::NBUF - 1;
    stream_slab(0, 0);
    if (AHEAD == 2 && nk > 1) stream_slab(1, 1);
    if (AHEAD == 2 && nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt % T::NBUF;
        const int nxt = (kt + AHEAD) % T::NBUF;
        const bool more = kt + AHEAD < nk;
        // The next slab's LDS-DMA pieces are issued one per unit, in the shadow of that unit's MFMAs,
        // instead of as one burst ahead of the first MFMA (in-order issue would drain the matrix pipe).
        constexpr int PPU = (NPIECE + (RBK / 8) * 4 - 1) / ((RBK / 8) * 4);   // pieces per unit
        const float* sb = smem + cur * T::SLAB;
        // NU units of {two B fragments, 8 MFMAs on two alternating accumulator tiles}.  The fragments of
        // unit u+1 are requested right before the MFMAs of unit u and consumed (empty asm = the
        // compiler's lgkmcnt(0)) right after them: their LDS latency hides under 512 matrix-pipe cycles
        // and the wait never covers a read that was only just issued.
        f32x4v a = *reinterpret_cast<const f32x4v*>(sb + a_row + cs[0]);
        f32x4v b0 = *reinterpret_cast<const f32x4v*>(sb + b_row + cs[0]);
        f32x4v b1 = *reinterpret_cast<const f32x4v*>(sb + b_row + 32 * RBK + cs[0]);
        f32x4v n0v = *reinterpret_cast<const f32x4v*>(sb + b_row + 2 * 32 * RBK + cs[0]);
        f32x4v n1v = *reinterpret_cast<const f32x4v*>(sb + b_row + 3 * 32 * RBK + cs[0]);
        f32x4v an = a;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int t = (u & 3) * 2;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b0[j], acc[t], 0, 0, 0);
                acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b1[j], acc[t + 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
#pragma unroll
                for (int pp = 0; pp < PPU; ++pp)
                    if (u * PPU + pp < NPIECE) stream_piece(kt + AHEAD, nxt, u * PPU + pp);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (u + 1 < NU) {
                asm volatile("" : "+v"(n0v), "+v"(n1v));     // fragments of unit u+1 have landed
                b0 = n0v; b1 = n1v;
                if (((u + 1) & 3) == 0) { asm volatile("" : "+v"(an)); a = an; }
                __builtin_amdgcn_sched_barrier(0);
                if (u + 2 < NU) {
                    const int s2 = (u + 2) >> 2, t2 = ((u + 2) & 3) * 2;
                    n0v = *reinterpret_cast<const f32x4v*>(sb + b_row + t2 * 32 * RBK + cs[s2]);
                    n1v = *reinterpret_cast<const f32x4v*>(sb + b_row + (t2 + 1) * 32 * RBK + cs[s2]);
                    if (t2 == 0) an = *reinterpret_cast<const f32x4v*>(sb + a_row + cs[s2]);
                }
            }
        }
        // slab kt+1 must have landed (all but the pieces of slab kt+2 just issued), then everyone is done
        // reading slab kt
        __builtin_amdgcn_sched_barrier(0);
        if (AHEAD == 2 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // ------------------------------------------------------------------ epilogue
    // The accumulator layout (a lane owns one column of 16 scattered rows) would cost 128 dword stores per
    // lane; instead each wave parks 8 complete rows at a time in LDS (free after the last slab) and walks
    // them with one float4 per lane: bias / ReLU / residual / LayerNorm / C2 act on a whole 1-KiB row per
    // wave-instruction (fully coalesced); row moments are a DPP wave reduction.
    const int flags = p.flags;
    const int mrow0 = m0 + wave * 32;
    float* ep = smem + wave * (8 * RT_EP_LD);
    const int c4 = lane * 4;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), g4 = bias4, be4 = bias4;
    if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + n0 + c4);
    if (flags & EPI_LN) {
        g4 = *reinterpret_cast<const float4*>(p.ln_g + c4);
        be4 = *reinterpret_cast<const float4*>(p.ln_b + c4);
    }
    const int rows_here = min(32, M - mrow0);   // wave-uniform, may be <= 0 for a wave past M
    __syncthreads();                            // every wave is done reading the last slab: LDS is reusable
#pragma unroll
    for (int g = 0; g < 4; ++g) {               // rows 8g .. 8g+7 of this wave = accumulator registers 4g .. 4g+3
        // the row image is private to the wave and a wave's LDS operations execute in order: no workgroup
        // barrier between the groups (the four waves drift apart through the epilogue instead of marching)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int rr4 = 0; rr4 < 4; ++rr4)
                ep[(rr4 + 4 * lh) * RT_EP_LD + t * 32 + li] = acc[t][4 * g + rr4];
        __builtin_amdgcn_wave_barrier();
        float4 v[8], rr[8], ad[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int i = min(8 * g + k, rows_here - 1);
            i = i < 0 ? 0 : i;
            const size_t m = (size_t)min(mrow0 + i, M - 1);
            if (flags & EPI_RESIDUAL) rr[k] = *reinterpret_cast<const float4*>(p.R + (p.r_mod ? m % (size_t)p.r_mod : m) * p.ldr + n0 + c4);
            if (p.C2) ad[k] = *reinterpret_cast<const float4*>(p.ADD + m * p.ldc + n0 + c4);
            v[k] = *reinterpret_cast<const float4*>(ep + k * RT_EP_LD + c4);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float4 x = v[k];
            x.x += bias4.x; x.y += bias4.y; x.z += bias4.z; x.w += bias4.w;
            if (flags & EPI_RELU) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
            if (flags & EPI_RESIDUAL) { x.x += rr[k].x; x.y += rr[k].y; x.z += rr[k].z; x.w += rr[k].w; }
            if (flags & EPI_LN) {   // N == 256 == BN: the wave holds the complete row
                const float mean = wave_sum_dpp((x.x + x.y) + (x.z + x.w)) * (1.0f / 256.0f);
                const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
                const float var = wave_sum_dpp((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / 256.0f);
                const float rstd = 1.0f / sqrtf(var + 1e-5f);
                x.x = d0 * rstd * g4.x + be4.x; x.y = d1 * rstd * g4.y + be4.y;
                x.z = d2 * rstd * g4.z + be4.z; x.w = d3 * rstd * g4.w + be4.w;
            }
            if (8 * g + k < rows_here) {
                const size_t m = (size_t)(mrow0 + 8 * g + k);
                *reinterpret_cast<float4*>(p.C + m * p.ldc + n0 + c4) = x;
                if (p.C2) {
                    x.x += ad[k].x; x.y += ad[k].y; x.z += ad[k].z; x.w += ad[k].w;
                    *reinterpret_cast<float4*>(p.C2 + m * p.ldc + n0 + c4) = x;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// 8-wave variant of the row-owning tile: same 128 x 256 x 16 slabs and LDS-DMA ring, but a wave owns 16 complete
// rows on v_mfma_f32_16x16x4_f32 (64 accumulators instead of 128 -> <= 128 VGPRs -> 16 waves per CU at two
// workgroups): while one wave walks its epilogue three others keep the matrix pipe busy.  Operand chunk swizzle
// f[(row >> 2) & 3], f = {0, 3, 2, 1}: conflict-free for the (row = lane % 16, chunk = lane / 16) ds_read_b128.
typedef float f32x4a __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int swz16(int row) { return (0x1230 >> (((row >> 2) & 3) * 4)) & 3; }

__global__ __launch_bounds__(512, 4) void gemm_rows16_kernel(GemmArgs p) {
    constexpr int RBK = 16, SLAB = (RT_BM + RT_BN) * RBK, NBUF = 3, NPIECE = 3;   // 24 KiB slab / 8 waves
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int M = p.M;
    if (p.M_dev) { int md = *p.M_dev; M = md < M ? md : M; }
    const int nc = p.N / RT_BN;
    const int L = blockIdx.x, grp = L / (8 * nc), r0 = L % (8 * nc);
    const int m0 = (grp * 8 + (r0 & 7)) * RT_BM;
    if (m0 >= M) return;
    const int n0 = (r0 >> 3) * RT_BN;
    const int K = p.K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;

    // LDS-DMA pieces: the slab image is 384 rows x 64 B; piece q (of 24) = rows 16q .. 16q+15; wave w issues pieces
    // 3w .. 3w+2 (pieces 0-7 = A rows, 8-23 = W rows); lane -> (row = lane / 4, physical chunk = lane % 4)
    const float* __restrict__ Ab = p.A + (size_t)m0 * p.lda;
    const float* __restrict__ Wb = p.W + (size_t)n0 * p.ldw;
    const float* src[NPIECE];
    int dst[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int q = wave * NPIECE + i;
        const int row = q * 16 + lane / 4;                        // row of the slab image
        const int chunk = (lane % 4) ^ swz16(row);                // source chunk that lands in physical chunk lane % 4
        if (q < 8) {
            const int sr = (m0 + row < M) ? row : (M - 1 - m0);   // rows past M feed unstored outputs
            src[i] = Ab + (size_t)sr * p.lda + (chunk << 2);
        } else {
            src[i] = Wb + (size_t)(row - RT_BM) * p.ldw + (chunk << 2);
        }
        dst[i] = q * 256;
    }
    auto stream_piece = [&](int kt, int buf, int i) { GLDS16(src[i] + kt * RBK, smem + buf * SLAB + dst[i]); };

    f32x4a acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = f32x4a{0.f, 0.f, 0.f, 0.f};
    const int csw = (lg ^ swz16(li)) << 2;                        // this lane's chunk (4 consecutive k) of row li
    const int a_row = (wave * 16 + li) * RBK + csw;
    const int b_row = RT_BM * RBK + li * RBK + csw;

    const int nk = K / RBK;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) stream_piece(0, 0, i);
    if (nk > 1) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) stream_piece(1, 1, i);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const float* sb = smem + (kt % NBUF) * SLAB;
        const int nxt = (kt + 2) % NBUF;
        const bool more = kt + 2 < nk;
        const f32x4a a = *reinterpret_cast<const f32x4a*>(sb + a_row);
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
            const f32x4a b0 = *reinterpret_cast<const f32x4a*>(sb + b_row + t * 16 * RBK);
            const f32x4a b1 = *reinterpret_cast<const f32x4a*>(sb + b_row + (t + 1) * 16 * RBK);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b0[j], acc[t], 0, 0, 0);
                acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b1[j], acc[t + 1], 0, 0, 0);
            }
            if (more && t / 2 < NPIECE) stream_piece(kt + 2, nxt, t / 2);
            if (t & 2) asm volatile("" ::: "memory");       // bound the operand prefetch depth (register budget: 128)
        }
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: accumulator (t, r) of lane (li, lg) = row 4 lg + r of the wave's 16, column 16 t + li.  Eight rows
    // at a time (lane groups 0-1, then 2-3) are parked in LDS and walked with one float4 per lane per row.
    const int flags = p.flags;
    const int mrow0 = m0 + wave * 16;
    float* ep = smem + wave * (8 * RT_EP_LD);
    const int c4 = lane * 4;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), g4 = bias4, be4 = bias4;
    if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + n0 + c4);
    if (flags & EPI_LN) {
        g4 = *reinterpret_cast<const float4*>(p.ln_g + c4);
        be4 = *reinterpret_cast<const float4*>(p.ln_b + c4);
    }
    const int rows_here = min(16, M - mrow0);
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        __builtin_amdgcn_wave_barrier();
        if ((lg >> 1) == g) {
#pragma unroll
            for (int t = 0; t < 16; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) ep[(4 * (lg & 1) + r) * RT_EP_LD + t * 16 + li] = acc[t][r];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {          // four rows at a time: 64 accumulators + 48 row registers stay < 128
            float4 v[4], rr[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int i = min(8 * g + 4 * hf + k, rows_here - 1);
                i = i < 0 ? 0 : i;
                const size_t m = (size_t)min(mrow0 + i, M - 1);
                if (flags & EPI_RESIDUAL) rr[k] = *reinterpret_cast<const float4*>(p.R + (p.r_mod ? m % (size_t)p.r_mod : m) * p.ldr + n0 + c4);
                v[k] = *reinterpret_cast<const float4*>(ep + (4 * hf + k) * RT_EP_LD + c4);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float4 x = v[k];
                x.x += bias4.x; x.y += bias4.y; x.z += bias4.z; x.w += bias4.w;
                if (flags & EPI_RELU) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
                if (flags & EPI_RESIDUAL) { x.x += rr[k].x; x.y += rr[k].y; x.z += rr[k].z; x.w += rr[k].w; }
                if (flags & EPI_LN) {
                    const float mean = wave_sum_dpp((x.x + x.y) + (x.z + x.w)) * (1.0f / 256.0f);
                    const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
                    const float var = wave_sum_dpp((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / 256.0f);
                    const float rstd = 1.0f / sqrtf(var + 1e-5f);
                    x.x = d0 * rstd * g4.x + be4.x; x.y = d1 * rstd * g4.y + be4.y;
                    x.z = d2 * rstd * g4.z + be4.z; x.w = d3 * rstd * g4.w + be4.w;
                }
                if (8 * g + 4 * hf + k < rows_here) {
                    const size_t m = (size_t)(mrow0 + 8 * g + 4 * hf + k);
#if CONE_GEMM_NT_STORE      // A/B (tools/ab_variants.sh): the output rows as non-temporal stores (1 KiB contiguous per wave):
                            // 6.48 -> 6.42 ms on the 2 M x 768 x 256 shape (1 %), off
                    __builtin_nontemporal_store(f32x4a{x.x, x.y, x.z, x.w}, reinterpret_cast<f32x4a*>(p.C + m * p.ldc + n0 + c4));
#else
                    *reinterpret_cast<float4*>(p.C + m * p.ldc + n0 + c4) = x;
#endif
                    if (p.C2) {
                        const float4 ad = *reinterpret_cast<const float4*>(p.ADD + m * p.ldc + n0 + c4);
                        x.x += ad.x; x.y += ad.y; x.z += ad.z; x.w += ad.w;
                        *reinterpret_cast<float4*>(p.C2 + m * p.ldc + n0 + c4) = x;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Small-M form of the row tile: a full 128 x 256 tile keeps one CU busy for >= 27 us of exact-fp32 MFMA whatever the row
// count (16.8 MFLOP at 256 FLOP / clock), so a launch over a few hundred or thousand rows -- the decoder / head / matching
// GEMMs of the single-query path, the context projections of one long video -- is bound by that one tile, and a launch of
// 258 tiles by the two CUs that run two.  Here a workgroup of 8 waves owns 16 rows x 256 columns, wave w the column tiles
// 2w, 2w + 1.  The 16 activation rows are staged once as operand slabs [16 rows][16 floats] (16-B chunks XOR-swizzled, the
// slab format of gemm_rows16_kernel); a wave's weight slabs -- nobody else needs them -- stream through its private ring of
// three k-steps (one coalesced global_load_lds per slab, a counted s_waitcnt vmcnt per step).  The accumulation runs the
// SAME chain as the 128-row tile -- for kt, for j: acc[t] = mfma(a[j], b[j], acc[t]) -- and the epilogue the same per-row
// code (one float4 per lane and row: bias, ReLU, residual, LayerNorm by wave_sum_dpp), so a row's result is bit-identical
// to the 128-row tile's: the choice depends on the host-known row bound only.
constexpr int RS_EP_LD = RT_BN + 4;
constexpr int RS_RING = 6 * 256;                                  // floats per wave: 3 k-steps x 2 column tiles

static size_t rs_lds_bytes(int K) {
    const int a_floats = 16 * K > 16 * RS_EP_LD ? 16 * K : 16 * RS_EP_LD;      // A slabs, later the epilogue image
    return (size_t)(a_floats + 8 * RS_RING) * sizeof(float);
}

__global__ __launch_bounds__(512, 2) void gemm_rows_small_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int M = p.M;
    if (p.M_dev) { int md = *p.M_dev - p.m_off; M = md < M ? md : M; }
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * RT_BN;
    if (m0 >= M) return;
    const int K = p.K, nk = K / 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    float* As = smem;                                             // [nk] slabs
    float* ep = smem;                                             // (after the main loop) [16][RS_EP_LD]
    float* ring = smem + (16 * K > 16 * RS_EP_LD ? 16 * K : 16 * RS_EP_LD) + wave * RS_RING;
    // slab sources: lane -> row = lane / 4, source chunk = the one that lands in physical chunk lane % 4
    const int drow = lane >> 2;
    const int dq = ((lane & 3) ^ swz16(drow)) << 2;
    const int arow = min(m0 + drow, M - 1);                       // rows past M feed unstored outputs
    const float* __restrict__ asrc = p.A + (size_t)arow * p.lda + dq;
    const float* __restrict__ wsrc = p.W + (size_t)(n0 + 32 * wave + drow) * p.ldw + dq;     // + 16 t * ldw + 16 kt
    const size_t wt = (size_t)16 * p.ldw;
    const int rdo = li * 16 + ((lg ^ swz16(li)) << 2);            // operand read: row li, chunk lg

    for (int kt = wave; kt < nk; kt += 8) GLDS16(asrc + 16 * kt, As + kt * 256);
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
        const bool live = kt < nk;                                // past the end: one dummy line (keeps the wait count exact)
        GLDS16(live ? wsrc + 16 * kt : p.W, ring + (2 * kt) * 256);
        GLDS16(live ? wsrc + wt + 16 * kt : p.W, ring + (2 * kt + 1) * 256);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");              // this wave's A slabs (older than the six W slabs)
    __builtin_amdgcn_s_barrier();                                 // every wave's A slabs
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");              // W slabs of k-step 0
    __builtin_amdgcn_sched_barrier(0);
    f32x4a acc[2];
    acc[0] = f32x4a{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
    f32x4a a = *reinterpret_cast<const f32x4a*>(As + rdo);
    f32x4a b0 = *reinterpret_cast<const f32x4a*>(ring + rdo), b1 = *reinterpret_cast<const f32x4a*>(ring + 256 + rdo);
    int slot = 0;                                                 // kt % 3
    for (int kt = 0; kt < nk; ++kt) {
        const int nslot = slot == 2 ? 0 : slot + 1;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");          // k-step kt + 1 has landed (kt + 2 may be in flight)
        const int kn = kt + 1 < nk ? kt + 1 : kt;
        const f32x4a an = *reinterpret_cast<const f32x4a*>(As + kn * 256 + rdo);
        const f32x4a bn0 = *reinterpret_cast<const f32x4a*>(ring + (2 * nslot) * 256 + rdo);
        const f32x4a bn1 = *reinterpret_cast<const f32x4a*>(ring + (2 * nslot + 1) * 256 + rdo);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b0[j], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b1[j], acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool more = kt + 3 < nk;                            // this step's slots (read one step ago) take k-step kt + 3
        GLDS16(more ? wsrc + 16 * (kt + 3) : p.W, ring + (2 * slot) * 256);
        GLDS16(more ? wsrc + wt + 16 * (kt + 3) : p.W, ring + (2 * slot + 1) * 256);
        a = an; b0 = bn0; b1 = bn1;
        slot = nslot;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // no LDS-DMA may outlive the workgroup's LDS allocation
    __syncthreads();                                              // every wave is done with the A slabs: epilogue image
    // accumulator (t, r) of lane (li, lg) = row 4 lg + r, column 32 wave + 16 t + li
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) ep[(4 * lg + r) * RS_EP_LD + 32 * wave + 16 * t + li] = acc[t][r];
    __syncthreads();
    const int flags = p.flags;
    const int c4 = lane * 4;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), g4 = bias4, be4 = bias4;
    if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + n0 + c4);
    if (flags & EPI_LN) {
        g4 = *reinterpret_cast<const float4*>(p.ln_g + c4);
        be4 = *reinterpret_cast<const float4*>(p.ln_b + c4);
    }
    float4 v[2], rr[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {                                 // wave w walks rows 2w, 2w + 1
        const size_t m = (size_t)min(m0 + 2 * wave + k, M - 1);
        if (flags & EPI_RESIDUAL) rr[k] = *reinterpret_cast<const float4*>(p.R + (p.r_mod ? (m + (size_t)p.m_off) % (size_t)p.r_mod : m) * p.ldr + n0 + c4);
        v[k] = *reinterpret_cast<const float4*>(ep + (2 * wave + k) * RS_EP_LD + c4);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float4 x = v[k];
        x.x += bias4.x; x.y += bias4.y; x.z += bias4.z; x.w += bias4.w;
        if (flags & EPI_RELU) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
        if (flags & EPI_RESIDUAL) { x.x += rr[k].x; x.y += rr[k].y; x.z += rr[k].z; x.w += rr[k].w; }
        if (flags & EPI_LN) {
            const float mean = wave_sum_dpp((x.x + x.y) + (x.z + x.w)) * (1.0f / 256.0f);
            const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
            const float var = wave_sum_dpp((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / 256.0f);
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
            x.x = d0 * rstd * g4.x + be4.x; x.y = d1 * rstd * g4.y + be4.y;
            x.z = d2 * rstd * g4.z + be4.z; x.w = d3 * rstd * g4.w + be4.w;
        }
        const int m = m0 + 2 * wave + k;
        if (m < M) *reinterpret_cast<float4*>(p.C + (size_t)m * p.ldc + n0 + c4) = x;
    }
}

// SPREAD form of the small row GEMM for a handful of row groups without a LayerNorm epilogue (the slot-row projections of a
// small batch: 100 rows x N = 768 is 7 workgroups of the form above, each streaming all of W through one CU): ONE WAVE per
// (16 rows, 16 columns), the same chain per element -- for kt, for j: acc = mfma(a[j], b[j], acc) -- and the same epilogue
// arithmetic (+ bias, ReLU, + residual) on the accumulator layout.  The A and W slabs arrive in passes of 8 k-slabs (128
// channels) through two LDS buffers: K <= 256 is both passes requested at once, a longer K (the text input projection: 768 or
// 512 channels) keeps two passes in flight ahead of the chain; K <= 1024.
constexpr int RSP_MAX_GROUPS = 64;
// ... and at most this many single-wave workgroups in all: every tile re-reads its row group's A slabs and its column tile's
// W slabs, so beyond ~4 workgroups per CU the workgroup-per-16-rows form wins (tools/gemm_spread_bench.py, N = 768: 5.7 us
// against 8.2 at 960 tiles, 8.5 against 8.4 at 1 536, 11.7 against 8.7 at 2 736; N = 256 never gets there: 64 groups = 1 024)
constexpr int RSP_MAX_TILES = 1024;
constexpr int RSP_PASS = 8;             // k-slabs per pass
__global__ __launch_bounds__(64) void gemm_rows_spread_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float As[2][RSP_PASS * 256];
    __shared__ __attribute__((aligned(16))) float Ws[2][RSP_PASS * 256];
    int M = p.M;
    if (p.M_dev) { int md = *p.M_dev - p.m_off; M = md < M ? md : M; }
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    if (m0 >= M) return;
    const int nk = p.K / 16;
    const int np = (nk + RSP_PASS - 1) / RSP_PASS;
    const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
    const int drow = lane >> 2;
    const int dq = ((lane & 3) ^ swz16(drow)) << 2;
    const int arow = min(m0 + drow, M - 1);
    const float* __restrict__ asrc = p.A + (size_t)arow * p.lda + dq;
    const float* __restrict__ wsrc = p.W + (size_t)(n0 + drow) * p.ldw + dq;
    const int rdo = li * 16 + ((lg ^ swz16(li)) << 2);
    // the epilogue's operands first (ordinary loads; vmcnt retires in order: they are behind no slab)
    const int col = n0 + li;
    const float bias = p.bias ? p.bias[col] : 0.f;
    const int flags = p.flags;
    float rr[4] = {0.f, 0.f, 0.f, 0.f};
    if (flags & EPI_RESIDUAL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t m = (size_t)min(m0 + 4 * lg + r, M - 1);
            rr[r] = p.R[(p.r_mod ? (m + (size_t)p.m_off) % (size_t)p.r_mod : m) * p.ldr + col];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    auto request = [&](int ps) {         // pass ps: its (up to) 8 slab pairs into buffer ps & 1
        const int k0 = ps * RSP_PASS, k1 = min(k0 + RSP_PASS, nk);
        for (int kt = k0; kt < k1; ++kt) {
            GLDS16(asrc + 16 * kt, As[ps & 1] + (kt - k0) * 256);
            GLDS16(wsrc + 16 * kt, Ws[ps & 1] + (kt - k0) * 256);
        }
    };
    request(0);
    if (np > 1) request(1);
    __builtin_amdgcn_sched_barrier(0);
    f32x4a acc = f32x4a{0.f, 0.f, 0.f, 0.f};
    for (int ps = 0; ps < np; ++ps) {
        // pass ps has landed: everything but the (complete, 16-request) pass behind it -- or everything
        __builtin_amdgcn_sched_barrier(0);
        if (ps + 1 < np && min(nk - (ps + 1) * RSP_PASS, RSP_PASS) == RSP_PASS)
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const int k0 = ps * RSP_PASS, k1 = min(k0 + RSP_PASS, nk);
        for (int kt = k0; kt < k1; ++kt) {
            const f32x4a a = *reinterpret_cast<const f32x4a*>(As[ps & 1] + (kt - k0) * 256 + rdo);
            const f32x4a b = *reinterpret_cast<const f32x4a*>(Ws[ps & 1] + (kt - k0) * 256 + rdo);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
        }
        if (ps + 2 < np) {              // this buffer's next pass: only once every read of it has returned
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            request(ps + 2);
        }
    }
    // accumulator r of lane (li, lg) = row 4 lg + r, column n0 + li
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float x = acc[r];
        x += bias;
        if (flags & EPI_RELU) x = fmaxf(x, 0.f);
        if (flags & EPI_RESIDUAL) x += rr[r];
        const int m = m0 + 4 * lg + r;
        if (m < M) p.C[(size_t)m * p.ldc + col] = x;
    }
}

// ---------------------------------------------------------------------------------------------------------
// rows_chain_kernel (rows_chain.h): up to four row-wise stages over the 16 rows of a workgroup, the rows on chip in between.
// A GEMM stage is gemm_rows_small_kernel's body on operand slabs that the PREVIOUS stage left in LDS (instead of LDS-DMA from
// global memory): the same accumulation chain -- for kt, for j: acc[t] = mfma(a[j], b[j], acc[t]) -- the same epilogue image
// and per-row code.  A LayerNorm stage is layernorm_kernel's arithmetic on a row held as one float4 per lane (wave w owns
// rows 2w, 2w + 1, exactly the rows it walks in a GEMM stage's epilogue); a head is rowdot_kernel's dot on those registers.
// Between stages the rows go back to LDS in the operand-slab format (row r, column c -> slab c / 16, 16-B chunk
// ((c / 4) % 4) ^ swz16(r)).  Stores and parameter loads count in vmcnt like the weight DMAs: every GEMM stage drains the
// queue before it starts its counted stream.
__device__ __forceinline__ void chain_rows_to_slabs(float* As, int r, int c4, const float4& x) {
    *reinterpret_cast<float4*>(As + (c4 >> 4) * 256 + r * 16 + ((((c4 >> 2) & 3) ^ swz16(r)) << 2)) = x;
}

__global__ __launch_bounds__(512, 2) void rows_chain_kernel(ChainArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int M = p.M;
    const int m0 = blockIdx.x * 16;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int K0 = p.st[0].kind == 0 ? p.st[0].K : 256;
    const int a_floats = 16 * K0 > 16 * RS_EP_LD ? 16 * K0 : 16 * RS_EP_LD;
    float* As = smem;                                             // operand slabs of the current GEMM stage's input
    float* ep = smem + a_floats;                                  // [16][RS_EP_LD] epilogue image (its own region here)
    float* ring = ep + 16 * RS_EP_LD + wave * RS_RING;
    const int drow = lane >> 2;
    const int dq = ((lane & 3) ^ swz16(drow)) << 2;
    const int rdo = li * 16 + ((lg ^ swz16(li)) << 2);            // operand read: row li, chunk lg
    const int c4 = lane * 4;
    float4 x[2];                                                  // this wave's rows 2w, 2w + 1 (columns c4 .. c4 + 3)
    bool in_regs = false;                                         // x holds the current rows (else: As holds them as slabs)

    if (p.st[0].kind == 0) {                                      // the input as operand slabs, by LDS-DMA (as the small form)
        const int nk0 = K0 / 16;
        const int arow = min(m0 + drow, M - 1);
        const float* __restrict__ asrc = p.A + (size_t)arow * p.lda + dq;
        for (int kt = wave; kt < nk0; kt += 8) GLDS16(asrc + 16 * kt, As + kt * 256);
    } else {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const size_t m = (size_t)min(m0 + 2 * wave + k, M - 1);
            x[k] = *reinterpret_cast<const float4*>(p.A + m * p.lda + c4);
        }
        in_regs = true;
    }

    for (int si = 0; si < p.n_stages; ++si) {
        const ChainStage& st = p.st[si];
        if (st.kind == 1) {
            // ---- LayerNorm (layernorm_kernel, dim = 256: one float4 per lane)
            const float4 gg = reinterpret_cast<const float4*>(st.ln_g)[lane];
            const float4 bb = reinterpret_cast<const float4*>(st.ln_b)[lane];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float4 v = x[k];
                float s = 0.f;
                s += (v.x + v.y) + (v.z + v.w);
                const float mean = wave_sum(s) / (float)256;
                float q = 0.f;
                {
                    const float a = v.x - mean, bq = v.y - mean, cq = v.z - mean, d = v.w - mean;
                    q += __builtin_fmaf(a, a, bq * bq) + __builtin_fmaf(cq, cq, d * d);
                }
                const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)256 + 1e-5f);
                float4 o;
                o.x = __builtin_fmaf((v.x - mean) * rstd, gg.x, bb.x);
                o.y = __builtin_fmaf((v.y - mean) * rstd, gg.y, bb.y);
                o.z = __builtin_fmaf((v.z - mean) * rstd, gg.z, bb.z);
                o.w = __builtin_fmaf((v.w - mean) * rstd, gg.w, bb.w);
                x[k] = o;
            }
        } else {
            // ---- GEMM stage: gemm_rows_small_kernel on the slabs in As
            const int K = st.K, nk = K / 16;
            if (in_regs) {
#pragma unroll
                for (int k = 0; k < 2; ++k) chain_rows_to_slabs(As, 2 * wave + k, c4, x[k]);
            }
            const float* __restrict__ wsrc = st.W + (size_t)(32 * wave + drow) * K + dq;      // + 16 t * K + 16 kt
            const size_t wt = (size_t)16 * K;
            __builtin_amdgcn_sched_barrier(0);
            if (si > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stores / parameter loads of the stage before
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                const bool live = kt < nk;
                GLDS16(live ? wsrc + 16 * kt : st.W, ring + (2 * kt) * 256);
                GLDS16(live ? wsrc + wt + 16 * kt : st.W, ring + (2 * kt + 1) * 256);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // (stage 0: this wave's input slabs, older than the W slabs)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's rows are in the slabs
            __builtin_amdgcn_s_barrier();                         // every wave's
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // W slabs of k-step 0
            __builtin_amdgcn_sched_barrier(0);
            f32x4a acc[2];
            acc[0] = f32x4a{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
            f32x4a a = *reinterpret_cast<const f32x4a*>(As + rdo);
            f32x4a b0 = *reinterpret_cast<const f32x4a*>(ring + rdo), b1 = *reinterpret_cast<const f32x4a*>(ring + 256 + rdo);
            int slot = 0;
            for (int kt = 0; kt < nk; ++kt) {
                const int nslot = slot == 2 ? 0 : slot + 1;
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                const int kn = kt + 1 < nk ? kt + 1 : kt;
                const f32x4a an = *reinterpret_cast<const f32x4a*>(As + kn * 256 + rdo);
                const f32x4a bn0 = *reinterpret_cast<const f32x4a*>(ring + (2 * nslot) * 256 + rdo);
                const f32x4a bn1 = *reinterpret_cast<const f32x4a*>(ring + (2 * nslot + 1) * 256 + rdo);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b0[j], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b1[j], acc[1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                const bool more = kt + 3 < nk;
                GLDS16(more ? wsrc + 16 * (kt + 3) : st.W, ring + (2 * slot) * 256);
                GLDS16(more ? wsrc + wt + 16 * (kt + 3) : st.W, ring + (2 * slot + 1) * 256);
                a = an; b0 = bn0; b1 = bn1;
                slot = nslot;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // accumulator (t, r) of lane (li, lg) = row 4 lg + r, column 32 wave + 16 t + li
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) ep[(4 * lg + r) * RS_EP_LD + 32 * wave + 16 * t + li] = acc[t][r];
            __syncthreads();                                      // the image is complete; every wave is done with the slabs
            const int flags = st.flags;
            float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (st.bias) bias4 = *reinterpret_cast<const float4*>(st.bias + c4);
            float4 v[2], rr[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const size_t m = (size_t)min(m0 + 2 * wave + k, M - 1);
                if (flags & EPI_RESIDUAL) rr[k] = *reinterpret_cast<const float4*>(st.R + m * st.ldr + c4);
                v[k] = *reinterpret_cast<const float4*>(ep + (2 * wave + k) * RS_EP_LD + c4);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float4 xx = v[k];
                xx.x += bias4.x; xx.y += bias4.y; xx.z += bias4.z; xx.w += bias4.w;
                if (flags & EPI_RELU) { xx.x = fmaxf(xx.x, 0.f); xx.y = fmaxf(xx.y, 0.f); xx.z = fmaxf(xx.z, 0.f); xx.w = fmaxf(xx.w, 0.f); }
                if (flags & EPI_RESIDUAL) { xx.x += rr[k].x; xx.y += rr[k].y; xx.z += rr[k].z; xx.w += rr[k].w; }
                x[k] = xx;
            }
        }
        in_regs = true;
        // ---- the stage's outputs: rows, head
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int m = m0 + 2 * wave + k;
            if (st.C && m < M) *reinterpret_cast<float4*>(st.C + (size_t)m * st.ldc + c4) = x[k];
            if (st.hout) {
                for (int n = 0; n < st.hnout; ++n) {
                    const float4 xv = x[k];
                    const float4 wv = reinterpret_cast<const float4*>(st.hw + n * 256)[lane];
                    float s = __builtin_fmaf(xv.x, wv.x, xv.y * wv.y) + __builtin_fmaf(xv.z, wv.z, xv.w * wv.w);
                    s = wave_sum(s) + st.hb[n];
                    if (st.hact == 1) s = 1.0f / (1.0f + expf(-s));
                    if (lane == 0 && m < M) st.hout[(size_t)m * st.hld + n] = s;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // no LDS-DMA may outlive the workgroup's LDS allocation
}


// The small form streams every weight byte from L2 once per 16 rows: 8.7 us per 256 tiles at K = 256 (13.3 at K = 512), i.e.
// tools/gemm_small_bench.py, N = 256, K = 256: 9 / 17.5 / 31 us at 4 096 / 8 192 / 16 384 rows against 38 - 41 us of the 128-row
// tile, behind from ~20 000 rows (57 vs 46 us at 32 768); N = 768: 24 us at 4 096 rows (768 tiles) against 40, level at 8 192.
#ifndef CONE_RS_MAX_WGS
#define CONE_RS_MAX_WGS 1280
#endif
constexpr int RS_MAX_WGS = CONE_RS_MAX_WGS;
// Two 128-row tiles on one CU run at about half speed each, so a launch of a few tiles more than a multiple of the CU count
// (258 tiles for the 33 000 context rows of a MAD-length video: 134 us against 80 us for 256) waits for the CUs that hold two.
// Up to this many small-form tiles of rows past the last full round are launched as the small form instead (same bits).
#ifndef CONE_RS_SPLIT_WGS
#define CONE_RS_SPLIT_WGS 256
#endif

int launch_gemm(const GemmArgs& a, hipStream_t s) {
    CONE_REQUIRE(a.K > 0 && a.K % BK == 0, "gemm: K=%d must be a multiple of %d", a.K, BK);
    CONE_REQUIRE(a.lda % 4 == 0 && a.ldw % 4 == 0, "gemm: lda/ldw must be multiples of 4");
    CONE_REQUIRE(!(a.flags & EPI_RESIDUAL) || a.R, "gemm: residual flag without R");
    if (a.M <= 0) return 0;
    if (a.flags & EPI_LN) CONE_REQUIRE(a.N == 256 && a.ln_g && a.ln_b, "gemm: LayerNorm epilogue needs N == 256");
    const bool rows_ok = a.N % 256 == 0 && !a.A2;
    // The tile family is a function of the SHAPE and of a.variant only (never of M or of process state), so a row
    // of C is computed by the same instruction sequence whatever batch it sits in: results are bit-identical
    // across batch compositions.  (The row tile's small-M form, chosen by the host-known row bound, runs the same fma
    // chains and the same per-row epilogue on 16-row tiles: same bits.)
    const bool use_rows = rows_ok && a.variant != GEMM_SQUARE;
    const bool need_rows = a.C2 != nullptr;
    CONE_REQUIRE(!need_rows || (rows_ok && a.ADD), "gemm: second output needs the row tile (N %% 256 == 0, no A2)");
    if (use_rows || need_rows) {
        static DeviceOnce once;     // the opt-in to > 64 KiB of LDS: once per device
        CONE_CHECK_HIP(device_once(once, [] {
            hipError_t rc = hipFuncSetAttribute((const void*)gemm_rows16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                RowTile<16>::LDS_BYTES);
            if (rc == hipSuccess)
                rc = hipFuncSetAttribute((const void*)gemm_rows_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         RowTile<16>::LDS_BYTES);
            if (rc == hipSuccess)
                rc = hipFuncSetAttribute((const void*)gemm_rows_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            return rc;
        }));
        int n_cu = 0;
        CONE_CHECK_HIP(device_once(once, [] { return hipSuccess; }, &n_cu));
        const bool waves8 = a.variant != GEMM_ROWS4;
        const bool small_ok = waves8 && !a.C2 && a.variant == GEMM_AUTO && rs_lds_bytes(a.K) <= 160 * 1024;
        if (small_ok && (int64_t)((a.M + 15) / 16) * (a.N / RT_BN) > RS_MAX_WGS && a.m_off == 0) {
            const int nc = a.N / RT_BN;
            const int64_t G = (int64_t)((a.M + RT_BM - 1) / RT_BM) * nc;
            const int64_t m1 = G / n_cu * n_cu / nc * RT_BM;          // rows of the full rounds of 128-row tiles
            const int64_t rem = a.M - m1;
            if (m1 > 0 && rem > 0 && (rem + 15) / 16 * nc <= CONE_RS_SPLIT_WGS) {
                GemmArgs h = a;
                h.M = (int)m1;
                h.variant = GEMM_ROWS8;                               // (the full rounds: no further split)
                if (int rc = launch_gemm(h, s)) return rc;
                GemmArgs t = a;
                t.A = a.A + (size_t)m1 * a.lda; t.C = a.C + (size_t)m1 * a.ldc; t.M = (int)rem; t.m_off = (int)m1;
                if (a.R && !a.r_mod) t.R = a.R + (size_t)m1 * a.ldr;
                return launch_gemm(t, s);
            }
        }
        if (small_ok && !(a.flags & EPI_LN) && a.K <= 1024 && (a.M + 15) / 16 <= RSP_MAX_GROUPS &&
            (a.M + 15) / 16 * (a.N / 16) <= RSP_MAX_TILES && !(a.flags & GEMM_NO_SPREAD)) {
            // a handful of row groups, no LayerNorm epilogue: one wave per 16 x 16 output tile (bit-identical; see the kernel)
            ProfScope ps(PK_GEMM_ROWS_SMALL, a.M, a.N, a.K, a.M_dev, s, a.m_off);
            hipLaunchKernelGGL(gemm_rows_spread_kernel, dim3((unsigned)(a.N / 16), (unsigned)((a.M + 15) / 16)), dim3(64), 0, s, a);
            CONE_LAUNCH_CHECK();
            return 0;
        }
        if (small_ok && (int64_t)((a.M + 15) / 16) * (a.N / RT_BN) <= RS_MAX_WGS) {
            // small-M form: the same chains on 16-row tiles spread over the CUs (bit-identical rows; see the kernel)
            ProfScope ps(PK_GEMM_ROWS_SMALL, a.M, a.N, a.K, a.M_dev, s, a.m_off);
            hipLaunchKernelGGL(gemm_rows_small_kernel, dim3((unsigned)((a.M + 15) / 16), (unsigned)(a.N / RT_BN)), dim3(512),
                               rs_lds_bytes(a.K), s, a);
            CONE_LAUNCH_CHECK();
            return 0;
        }
        const int row_tiles = (a.M + RT_BM - 1) / RT_BM;
        dim3 grid((unsigned)((row_tiles + 7) / 8 * 8 * (a.N / RT_BN)));      // 1-D, see the tile order in the kernel
        ProfScope ps(waves8 ? PK_GEMM_ROWS16 : PK_GEMM_ROWS, a.M, a.N, a.K, a.M_dev, s);
        if (waves8)
            hipLaunchKernelGGL(gemm_rows16_kernel, grid, dim3(512), RowTile<16>::LDS_BYTES, s, a);
        else
            hipLaunchKernelGGL(gemm_rows_kernel<16>, grid, dim3(256), RowTile<16>::LDS_BYTES, s, a);
        CONE_LAUNCH_CHECK();
        return 0;
    }
    if (a.flags & EPI_LN) {
        CONE_REQUIRE(a.N == 256 && a.ln_g && a.ln_b, "gemm: LayerNorm epilogue needs N == 256");
        CONE_REQUIRE(!a.A2, "gemm: LayerNorm epilogue with A2 is not instantiated");
        dim3 grid((a.M + 63) / 64, 1);
        ProfScope ps(PK_GEMM_64x256, a.M, a.N, a.K, a.M_dev, s);
        hipLaunchKernelGGL((gemm_f32_kernel<64, 256, false>), grid, dim3(256), 0, s, a);
    } else {
        dim3 grid((a.M + 127) / 128, (a.N + 127) / 128);
        ProfScope ps(a.A2 ? PK_GEMM_128x128_A2 : PK_GEMM_128x128, a.M, a.N, a.K, a.M_dev, s);
        if (a.A2) {
            CONE_REQUIRE(a.lda2 % 4 == 0, "gemm: lda2 must be a multiple of 4");
            hipLaunchKernelGGL((gemm_f32_kernel<128, 128, true>), grid, dim3(256), 0, s, a);
        } else {
            hipLaunchKernelGGL((gemm_f32_kernel<128, 128, false>), grid, dim3(256), 0, s, a);
        }
    }
    CONE_LAUNCH_CHECK();
    return 0;
}

bool rows_chain_supported(int M) { return M > 0 && (M + 15) / 16 <= RS_MAX_WGS; }
// launches of this many rows (K <= 1 024, no LayerNorm epilogue) take the spread form of the row GEMM
bool gemm_rows_spread_rows(int M) { return M > 0 && (M + 15) / 16 <= RSP_MAX_GROUPS; }

int launch_rows_chain(const ChainArgs& a, hipStream_t s) {
    CONE_REQUIRE(a.A && a.n_stages >= 1 && a.n_stages <= CHAIN_MAX_STAGES && a.lda % 4 == 0, "rows chain: bad arguments");
    if (a.M <= 0) return 0;
    CONE_REQUIRE(rows_chain_supported(a.M), "rows chain: %d rows are beyond the small-M regime", a.M);
    for (int i = 0; i < a.n_stages; ++i) {
        const ChainStage& st = a.st[i];
        if (st.kind == 0) {
            CONE_REQUIRE(st.W && st.K % 16 == 0 && st.K >= 16 && st.K <= 1024 && (i == 0 || st.K == 256),
                         "rows chain: stage %d: K=%d (stage 0: a multiple of 16 up to 1024; later stages: 256)", i, st.K);
            CONE_REQUIRE(!(st.flags & EPI_RESIDUAL) || (st.R && st.ldr % 4 == 0), "rows chain: residual flag without R");
            CONE_REQUIRE(!(st.flags & EPI_LN), "rows chain: a LayerNorm epilogue is a stage of its own (kind 1)");
        } else {
            CONE_REQUIRE(st.kind == 1 && st.ln_g && st.ln_b, "rows chain: stage %d: bad LayerNorm stage", i);
        }
        CONE_REQUIRE(!st.C || st.ldc % 4 == 0, "rows chain: output row stride must be a multiple of 4");
        CONE_REQUIRE(!st.hout || (st.hw && st.hb && st.hnout >= 1 && st.hnout <= 2), "rows chain: bad head");
    }
    CONE_REQUIRE(a.st[0].kind == 0 || a.lda >= 256, "rows chain: a LayerNorm first stage reads 256-wide rows");
    const int K0 = a.st[0].kind == 0 ? a.st[0].K : 256;
    const size_t lds = (size_t)((16 * K0 > 16 * RS_EP_LD ? 16 * K0 : 16 * RS_EP_LD) + 16 * RS_EP_LD + 8 * RS_RING) * sizeof(float);
    static DeviceOnce once;     // the opt-in to > 64 KiB of LDS: once per device
    CONE_CHECK_HIP(device_once(once, [] {
        return hipFuncSetAttribute((const void*)rows_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }));
    hipLaunchKernelGGL(rows_chain_kernel, dim3((unsigned)((a.M + 15) / 16)), dim3(512), lds, s, a);
    CONE_LAUNCH_CHECK();
    return 0;
}

}  // namespace cone
