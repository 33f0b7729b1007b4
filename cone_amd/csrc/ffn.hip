// Fused position-wise feed-forward block of the post-norm transformer layers
//     out = LayerNorm(x + W2 relu(W1 x + b1) + b2)          (cone/transformer.py:241-245, 312-316)
// as ONE kernel on the fp32 matrix cores: the (M, ff) hidden activations never leave the CU.
//
// Unfused, the two linear layers are 76 % of the window model's GEMM time and move 4 * ff bytes per token row out
// to HBM and back in between (16 GB per encoder layer at 2 M rows, ff = 1024), through two launches whose
// 128 x 256 output tiles each pay a prologue and a store / LayerNorm epilogue per 16 (linear1) or 64 (linear2)
// k-slabs.  Here a workgroup owns 128 token rows for the whole block and pays them once per 128 slabs' worth of
// MFMAs.
//
// Orientation.  Everything is computed TRANSPOSED so that the hidden tile can be fed back to the matrix core
// straight from its accumulator registers (cdna_hip_programming.md section 3, "An accumulator tile as the next
// MFMA's operand": a following product that sums over the tile's ROW index takes it with no lane movement):
//   a wave owns 16 token rows (the MFMA column index j = lane % 16);
//   GEMM1:  H^T[h][j]  = sum_k  W1[h][k]  x[j][k]    A = W1 rows of a 16-unit hidden chunk (LDS), B = x^T (registers:
//                                                      the wave's 16 x 256 input tile is loaded once, 64 VGPRs);
//           accumulator register r of lane (li, lg) = H^T[h0 + 4 lg + r][token li];
//   GEMM2:  Y^T[n][j] += sum_h  W2[n][h]  H^T[h][j]  A = W2[:, chunk] (LDS), B = the accumulator of GEMM1 after bias +
//                                                      ReLU -- k slot lg of step r <-> hidden unit h0 + 4 lg + r;
//           accumulator (t, r) of lane (li, lg) = Y[token li][channel 16 t + 4 lg + r].
//   The k index of GEMM1 is permuted the same way on both operands (k slot lg of step (q, r) <-> 16 q + 4 lg + r), so
//   a lane's A fragment for four steps is ONE ds_read_b128 and its x fragment is the float4 x[token][16 q + 4 lg ..],
//   which is also exactly the residual that Y's accumulator (t = q, r) needs: the residual add is register-register.
//   Exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): results equal the two-GEMM path up to fp32 summation order.
//
// Weight stream.  Per 16-unit hidden chunk the workgroup needs W1[h0 .. h0+15][0..255] (16 KiB) and
// W2[0..255][h0 .. h0+15] (16 KiB).  Both are staged by LDS-DMA (global_load_lds_dwordx4, no VGPRs) as 16 + 16 slabs of
// [16 rows][16 floats] -- the operand-slab format of gemm.hip's row tile: unpadded 64-B rows, 16-B chunk XOR-swizzled
// on the SOURCE address and on the ds_read_b128 address (conflict-free for lane = (row, chunk)) -- into a 3-stage
// ring (96 KiB): chunk c+2 streams in while chunk c is multiplied, one counted s_waitcnt vmcnt + raw s_barrier per
// chunk (128 MFMAs per wave).  b1 is copied to LDS once so that no ordinary global load sits inside the loop (hipcc
// would drain the DMA queue for it, cdna_hip_programming.md section 5 "Three .s-level traps" (b)).
//
// One 8-wave workgroup per CU (100 KiB of LDS, <= 256 VGPRs): two waves per SIMD keep the matrix pipe fed across each
// other's GEMM1 -> GEMM2 hand-over; the prologue (x tile: 16 KiB per wave) and the epilogue (bias + residual +
// LayerNorm in registers: a token's 256 channels sit in 4 lanes x 64 registers, two shuffle steps per moment; 64-B
// row segments per store instruction) cost ~2 % of a tile's 8 192 MFMAs per wave.
#include <mutex>

#include "common.h"

namespace cone {

typedef float f32x4f __attribute__((ext_vector_type(4)));

#define FFN_GLDS16(src, dst) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

__device__ __forceinline__ int ffn_swz16(int row) { return (0x1230 >> (((row >> 2) & 3) * 4)) & 3; }

constexpr int FFN_ROWS = 128;                     // token rows per workgroup (8 waves x 16)
constexpr int FFN_STAGE = 2 * 16 * 256;           // floats per ring stage: W1 image (16 slabs) + W2 image (16 slabs)
constexpr int FFN_NST = 3;
constexpr int FFN_NPIECE = 4;                     // 1-KiB LDS-DMA pieces per wave per chunk (32 pieces / 8 waves)

struct FfnArgs {
    const float* X; int ldx;                      // (M, 256) input = residual
    const float* W1; const float* b1;             // (ff, 256), (ff)
    const float* W2; const float* b2;             // (256, ff), (256)
    const float* ln_g; const float* ln_b;         // (256)
    float* OUT; int ldo;                          // (M, 256)
    int M; const int* M_dev;                      // rows; *M_dev wins when non-null (grid sized by M)
    int ff;
};

__global__ __launch_bounds__(512, 2) void ffn_fused_kernel(FfnArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* b1s = smem + FFN_NST * FFN_STAGE;
    int M = p.M;
    if (p.M_dev) { const int md = *p.M_dev; M = md < M ? md : M; }
    const int m_tile = blockIdx.x * FFN_ROWS;
    if (m_tile >= M) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int ff = p.ff, nc = ff >> 4;

    // ---- this wave's input tile: x[token li][16 q + 4 lg .. + 3], q = 0 .. 15 (B operand of GEMM1 + the residual)
    const int my_row = m_tile + wave * 16 + li;
    const size_t ld_row = (size_t)(my_row < M ? my_row : M - 1);       // rows past M feed unstored outputs
    f32x4f xr[16];
    {
        const float* xp = p.X + ld_row * p.ldx + 4 * lg;
#pragma unroll
        for (int q = 0; q < 16; ++q) xr[q] = *reinterpret_cast<const f32x4f*>(xp + 16 * q);
    }
    for (int i = tid; i < (ff >> 2); i += 512)
        reinterpret_cast<f32x4f*>(b1s)[i] = reinterpret_cast<const f32x4f*>(p.b1)[i];

    // ---- LDS-DMA pieces of a chunk: 32 slabs of [16 rows][16 floats]; wave w issues slabs 4w .. 4w+3 (0-15 = W1, 16-31
    // = W2); lane -> (row = lane / 4, physical chunk = lane % 4), the source chunk is XOR-swizzled
    const float* src[FFN_NPIECE];
    int step[FFN_NPIECE], dst[FFN_NPIECE];
#pragma unroll
    for (int i = 0; i < FFN_NPIECE; ++i) {
        const int pid = wave * FFN_NPIECE + i;
        const int row = lane >> 2;
        const int ch = (lane & 3) ^ ffn_swz16(row);
        if (pid < 16) {     // W1[h0 + row][16 pid + 4 ch ..]; next chunk: 16 rows further
            src[i] = p.W1 + (size_t)row * 256 + 16 * pid + 4 * ch;
            step[i] = 16 * 256;
        } else {            // W2[16 (pid - 16) + row][h0 + 4 ch ..]; next chunk: 16 columns further
            src[i] = p.W2 + (size_t)(16 * (pid - 16) + row) * ff + 4 * ch;
            step[i] = 16;
        }
        dst[i] = pid * 256;
    }
    auto stream_piece = [&](int c, int i) {
        FFN_GLDS16(src[i] + (size_t)c * step[i], smem + (c % FFN_NST) * FFN_STAGE + dst[i]);
    };

    f32x4f y[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) y[t] = f32x4f{0.f, 0.f, 0.f, 0.f};
    const int rd = li * 16 + ((lg ^ ffn_swz16(li)) << 2);          // this lane's 16-B chunk inside a slab

#pragma unroll
    for (int i = 0; i < FFN_NPIECE; ++i) stream_piece(0, i);
    if (nc > 1) {
#pragma unroll
        for (int i = 0; i < FFN_NPIECE; ++i) stream_piece(1, i);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FFN_NPIECE) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                               // chunk 0 and the b1 image are in LDS
    for (int c = 0; c < nc; ++c) {
        const float* st = smem + (c % FFN_NST) * FFN_STAGE;
        const bool more = c + 2 < nc;
        // GEMM1: H^T chunk, four independent partial chains (one per k step of a slab)
        f32x4f hp[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) hp[r] = f32x4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const f32x4f w = *reinterpret_cast<const f32x4f*>(st + q * 256 + rd);
#pragma unroll
            for (int r = 0; r < 4; ++r) hp[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], xr[q][r], hp[r], 0, 0, 0);
            if (more && (q & 3) == 1) stream_piece(c + 2, q >> 2);  // the next-but-one chunk's pieces ride behind MFMAs
        }
        const f32x4f bb = *reinterpret_cast<const f32x4f*>(b1s + 16 * c + 4 * lg);
        f32x4f h = (hp[0] + hp[1]) + (hp[2] + hp[3]) + bb;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = fmaxf(h[r], 0.f);
        // GEMM2: Y^T += W2[:, chunk] . H^T chunk, two output tiles at a time (independent accumulators)
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
            const f32x4f w0 = *reinterpret_cast<const f32x4f*>(st + 4096 + t * 256 + rd);
            const f32x4f w1 = *reinterpret_cast<const f32x4f*>(st + 4096 + (t + 1) * 256 + rd);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[r], h[r], y[t], 0, 0, 0);
                y[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[r], h[r], y[t + 1], 0, 0, 0);
            }
        }
        // chunk c+1 must have landed (all but the pieces of chunk c+2 just issued), then everyone is done with chunk c
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FFN_NPIECE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: + b2 + residual, LayerNorm over the token's 256 channels (4 lanes x 64 registers), store
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const f32x4f b2v = *reinterpret_cast<const f32x4f*>(p.b2 + 16 * t + 4 * lg);
        y[t] = y[t] + b2v + xr[t];
        s1 += (y[t][0] + y[t][1]) + (y[t][2] + y[t][3]);
    }
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
            const f32x4f g = *reinterpret_cast<const f32x4f*>(p.ln_g + 16 * t + 4 * lg);
            const f32x4f be = *reinterpret_cast<const f32x4f*>(p.ln_b + 16 * t + 4 * lg);
            f32x4f o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = y[t][r] * rstd * g[r] + be[r];
            *reinterpret_cast<f32x4f*>(op + 16 * t) = o;
        }
    }
}

bool ffn_fused_supported(int ff) { return ff >= 32 && ff % 16 == 0 && ff <= 4096; }

int launch_ffn_fused(const float* X, int ldx, const float* W1, const float* b1, const float* W2, const float* b2,
                     const float* ln_g, const float* ln_b, float* OUT, int ldo, int M, const int* M_dev, int ff,
                     hipStream_t s) {
    CONE_REQUIRE(ffn_fused_supported(ff), "fused FFN: dim_feedforward=%d unsupported", ff);
    CONE_REQUIRE(X && W1 && b1 && W2 && b2 && ln_g && ln_b && OUT, "fused FFN: null argument");
    CONE_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0, "fused FFN: row strides must be multiples of 4");
    if (M <= 0) return 0;
    const size_t lds = (size_t)(FFN_NST * FFN_STAGE + ff) * sizeof(float);
    static std::once_flag once;     // the opt-in to > 64 KiB of LDS is a property of the code object: set it once
    static hipError_t attr_rc = hipSuccess;
    std::call_once(once, [] {
        attr_rc = hipFuncSetAttribute((const void*)ffn_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (FFN_NST * FFN_STAGE + 4096) * (int)sizeof(float));
    });
    CONE_CHECK_HIP(attr_rc);
    FfnArgs a{X, ldx, W1, b1, W2, b2, ln_g, ln_b, OUT, ldo, M, M_dev, ff};
    ProfScope ps(PK_FFN_FUSED, M, ff, 256, M_dev, s);
    hipLaunchKernelGGL(ffn_fused_kernel, dim3((unsigned)((M + FFN_ROWS - 1) / FFN_ROWS)), dim3(512), lds, s, a);
    CONE_LAUNCH_CHECK();
    return 0;
}

}  // namespace cone
