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
#include "common.h"

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
                if ((flags & EPI_RESIDUAL) && m < M && n < N) v += p.R[(size_t)m * p.ldr + n];
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

int launch_gemm(const GemmArgs& a, hipStream_t s) {
    CONE_REQUIRE(a.K > 0 && a.K % BK == 0, "gemm: K=%d must be a multiple of %d", a.K, BK);
    CONE_REQUIRE(a.lda % 4 == 0 && a.ldw % 4 == 0, "gemm: lda/ldw must be multiples of 4");
    CONE_REQUIRE(!(a.flags & EPI_RESIDUAL) || a.R, "gemm: residual flag without R");
    if (a.M <= 0) return 0;
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

}  // namespace cone
