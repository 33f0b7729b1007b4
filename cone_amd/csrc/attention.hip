// Multi-head attention cores of the Moment-DETR window model (head_dim = 32).
//
//   enc_attn_kernel   : encoder self-attention (cone/transformer.py:233-240 -> nn.MultiheadAttention),
//                       one workgroup per (window, head) over the window's PACKED tokens
//                       (video clips then text tokens; padded keys simply do not exist, which is what
//                       the reference's -inf key_padding_mask amounts to).
//   small_attn_kernel : the decoder's attentions with Nq<=8 query slots per window
//                       (cone/transformer.py:296-311): self-attention over the slots and
//                       cross-attention to the window's memory tokens.
//
// enc_attn_kernel layout (exact-fp32 MFMA 32x32x2):
//   S^T = K . Q^T is computed with the KEY on the accumulator rows and the QUERY on the lanes, so a
//   lane owns one query column: the row softmax is an in-register reduction plus ONE cross-half
//   shuffle (lane ^ 32), and the probability registers are directly the A operand of O = P . V
//   (the MFMA k index may be permuted freely as long as A and B agree, and the accumulator row
//   pattern (r&3)+8(r>>2)+4(lane>>5) is used as that permutation).  K is staged in LDS as
//   [key][33] (odd stride: conflict-free lane==key reads), V as [key][32] (lane==d reads).
#include "common.h"

namespace cone {

constexpr float kQScale = 0.17677669529663687f;  // sqrt(1/32), applied to q after projection

// GATHER = true (first encoder layer with the layer-0 cache): q|k|v of a token are not read from packed (M, .)
// matrices but straight from the per-clip / per-token projection caches, q and k of a clip plus the static
// pos.W_qk^T row of its (window length, position) -- the gather that pack_l0_kernel would otherwise write out
// to HBM (6 KB per token written and read back) happens in the staging loads.  Same adds, same results.
struct L0Gather {
    const float* qkv_vid;   // (n_clips, 768)  q | k | v
    const float* qkv_txt;   // (n_tokens, 768)
    const float* pos_qk;    // (W(W+1)/2, 512) row lv(lv-1)/2 + p
    const int* vrow0;
    const int* vlen;
    const int* trow0;
};

static int g_attn16 = 1;   // test hook (cone_test_set_option "attn16"): 0 = the 32x32x2 kernel below
void set_attn16(int v) { g_attn16 = v != 0; }

template <int NKB, bool GATHER>
__global__ __launch_bounds__(256, 2) void enc_attn_kernel(const float* __restrict__ QK,  // (M,512): q | k
                                                       const float* __restrict__ V,   // (M,256)
                                                       float* __restrict__ OUT,       // (M,256)
                                                       const int* __restrict__ off, L0Gather g) {
    __shared__ float Ks[NKB * 32 * 33];
    __shared__ __attribute__((aligned(16))) float Vs[NKB * 32 * 32];
    const int b = blockIdx.x, head = blockIdx.y;
    const int t0 = off[b];
    const int L = off[b + 1] - t0;
    const int nkb = (L + 31) >> 5;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    int lv = 0, vr0 = 0, tr0 = 0, pbase = 0;
    if (GATHER) { lv = g.vlen[b]; vr0 = g.vrow0[b]; tr0 = g.trow0[b]; pbase = lv * (lv - 1) / 2; }
    // source row of token `tok` for column block `col` (0 = q, 256 = k, 512 = v) + its additive pos row (or null)
    auto src_of = [&](int tok, int col, const float*& add) -> const float* {
        if (tok < lv) {
            add = col < 512 ? g.pos_qk + (size_t)(pbase + tok) * 512 + col : nullptr;
            return g.qkv_vid + (size_t)(vr0 + tok) * 768 + col;
        }
        add = nullptr;
        return g.qkv_txt + (size_t)(tr0 + tok - lv) * 768 + col;
    };

    {   // stage K (transposing scalar writes, stride 33) and V (float4) for all keys of the window
        const int kr = tid >> 3, c = tid & 7;
        for (int key = kr; key < nkb * 32; key += 32) {
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (key < L) {
                if (GATHER) {
                    const float* add;
                    const float* kp_ = src_of(key, 256, add);
                    kv = *reinterpret_cast<const float4*>(kp_ + head * 32 + c * 4);
                    vv = *reinterpret_cast<const float4*>(kp_ + 256 + head * 32 + c * 4);
                    if (add) {
                        const float4 t = *reinterpret_cast<const float4*>(add + head * 32 + c * 4);
                        kv.x += t.x; kv.y += t.y; kv.z += t.z; kv.w += t.w;
                    }
                } else {
                    kv = *reinterpret_cast<const float4*>(QK + (size_t)(t0 + key) * 512 + 256 + head * 32 + c * 4);
                    vv = *reinterpret_cast<const float4*>(V + (size_t)(t0 + key) * 256 + head * 32 + c * 4);
                }
            }
            float* kd = Ks + key * 33 + c * 4;
            kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
            *reinterpret_cast<float4*>(Vs + key * 32 + c * 4) = vv;
        }
    }
    __syncthreads();

    for (int qb = wave; qb < nkb; qb += 4) {
        // this lane's query row, dims 16*lh .. 16*lh+15, pre-scaled
        float qv[16];
        {
            int qrow = qb * 32 + li;
            qrow = qrow < L ? qrow : L - 1;
            const float* qadd = nullptr;
            const float* qp = GATHER ? src_of(qrow, 0, qadd) + head * 32 + 16 * lh
                                     : QK + (size_t)(t0 + qrow) * 512 + head * 32 + 16 * lh;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 x = reinterpret_cast<const float4*>(qp)[u];
                if (GATHER && qadd) {
                    const float4 t = reinterpret_cast<const float4*>(qadd + head * 32 + 16 * lh)[u];
                    x.x += t.x; x.y += t.y; x.z += t.z; x.w += t.w;
                }
                qv[4 * u] = x.x * kQScale; qv[4 * u + 1] = x.y * kQScale;
                qv[4 * u + 2] = x.z * kQScale; qv[4 * u + 3] = x.w * kQScale;
            }
        }
        f32x16 sc[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kb][r] = 0.f;
            if (kb < nkb) {
                const float* kp = Ks + (kb * 32 + li) * 33 + 16 * lh;
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    sc[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[t], qv[t], sc[kb], 0, 0, 0);
            }
        }
        // softmax over keys: registers (half of each key block) + the other half-wave.  Only the last key
        // block can hold padding; exp(s - m) is one fma + one v_exp_f32 (exp2((s - m) * log2 e)).
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
            if (kb < nkb) {
                if (kb == nkb - 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kb * 32 + acc_row(r, lane) >= L) sc[kb][r] = -INFINITY;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, sc[kb][r]);
            }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float m2 = m * 1.4426950408889634f;
        float l = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
            if (kb < nkb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(sc[kb][r], 1.4426950408889634f, -m2));
                    sc[kb][r] = e;
                    l += e;
                }
            }
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
            if (kb < nkb) {
                const float* vp = Vs + (kb * 32 + 4 * lh) * 32 + li;
                // k-step t covers keys (t&3)+8(t>>2) (+4 on the upper half-wave): past the window's last key the
                // probabilities are exactly 0, so the ragged last block stops early (uniform branch)
                const int rem = L - kb * 32;
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    if ((t & 3) + 8 * (t >> 2) < rem)
                        o = __builtin_amdgcn_mfma_f32_32x32x2f32(sc[kb][t] * inv, vp[((t & 3) + 8 * (t >> 2)) * 32],
                                                                 o, 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qrow = qb * 32 + acc_row(r, lane);
            if (qrow < L) OUT[(size_t)(t0 + qrow) * 256 + head * 32 + li] = o[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// 16x16x4 variant (default): one wave per 16-query tile, workgroup = ceil(Lmax/16) waves.  A window of 101 tokens
// pads to 112 x 112 scores instead of 128 x 128 (-23 % MFMA and exp work), a wave keeps only 4 registers per key
// tile (7 waves x <= 96 VGPRs: twice the resident waves of the 32x32 kernel) and the per-wave critical path halves.
//   S^T tile: A = K (keys on accumulator rows 4g + r, g = lane / 16), B = Q^T (query = lane % 16); the head dim
//             is walked as d = 8 g + step so that a lane's 8 query values are two contiguous float4.
//   P.V     : the probability registers are the A operand again (k slot g <-> key 4g + r of the tile); V rows are
//             read at stride 36 floats, K is staged d-major [32][KP + 2]: both conflict-free for ds_read_b32.
typedef float f32x4m __attribute__((ext_vector_type(4)));

template <int NKT, bool GATHER>
__global__ __launch_bounds__(64 * NKT, 6) void enc_attn16_kernel(const float* __restrict__ QK, const float* __restrict__ V,
                                                             float* __restrict__ OUT, const int* __restrict__ off,
                                                             L0Gather g) {
    constexpr int KP = 16 * NKT, LDK = KP + 2, LDV = 36, NT = 64 * NKT;
    __shared__ float KsT[32 * LDK];
    __shared__ __attribute__((aligned(16))) float Vs[KP * LDV];
    const int b = blockIdx.x, head = blockIdx.y;
    const int t0 = off[b];
    const int L = off[b + 1] - t0;
    const int nkt = (L + 15) >> 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    int lv = 0, vr0 = 0, tr0 = 0, pbase = 0;
    if (GATHER) { lv = g.vlen[b]; vr0 = g.vrow0[b]; tr0 = g.trow0[b]; pbase = lv * (lv - 1) / 2; }
    auto src_of = [&](int tok, int col, const float*& add) -> const float* {
        if (tok < lv) {
            add = col < 512 ? g.pos_qk + (size_t)(pbase + tok) * 512 + col : nullptr;
            return g.qkv_vid + (size_t)(vr0 + tok) * 768 + col;
        }
        add = nullptr;
        return g.qkv_txt + (size_t)(tr0 + tok - lv) * 768 + col;
    };

    // this lane's query values d = 8 lg .. 8 lg + 7 (fetched before the K/V staging: independent round trips)
    const int q0 = wave * 16;
    float qv[8];
    if (q0 < L) {
        int qrow = q0 + li;
        qrow = qrow < L ? qrow : L - 1;
        const float* qadd = nullptr;
        const float* qp = GATHER ? src_of(qrow, 0, qadd) + head * 32 + 8 * lg
                                 : QK + (size_t)(t0 + qrow) * 512 + head * 32 + 8 * lg;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float4 x = reinterpret_cast<const float4*>(qp)[u];
            if (GATHER && qadd) {
                const float4 t = reinterpret_cast<const float4*>(qadd + head * 32 + 8 * lg)[u];
                x.x += t.x; x.y += t.y; x.z += t.z; x.w += t.w;
            }
            qv[4 * u] = x.x * kQScale; qv[4 * u + 1] = x.y * kQScale;
            qv[4 * u + 2] = x.z * kQScale; qv[4 * u + 3] = x.w * kQScale;
        }
    }
    {   // stage K (d-major) and V for all keys of the window; zero rows past L
        const int kr = tid >> 3, c = tid & 7;
        for (int key = kr; key < nkt * 16; key += NT / 8) {
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (key < L) {
                if (GATHER) {
                    const float* add;
                    const float* kp_ = src_of(key, 256, add);
                    kv = *reinterpret_cast<const float4*>(kp_ + head * 32 + c * 4);
                    vv = *reinterpret_cast<const float4*>(kp_ + 256 + head * 32 + c * 4);
                    if (add) {
                        const float4 t = *reinterpret_cast<const float4*>(add + head * 32 + c * 4);
                        kv.x += t.x; kv.y += t.y; kv.z += t.z; kv.w += t.w;
                    }
                } else {
                    kv = *reinterpret_cast<const float4*>(QK + (size_t)(t0 + key) * 512 + 256 + head * 32 + c * 4);
                    vv = *reinterpret_cast<const float4*>(V + (size_t)(t0 + key) * 256 + head * 32 + c * 4);
                }
            }
            KsT[(4 * c + 0) * LDK + key] = kv.x; KsT[(4 * c + 1) * LDK + key] = kv.y;
            KsT[(4 * c + 2) * LDK + key] = kv.z; KsT[(4 * c + 3) * LDK + key] = kv.w;
            *reinterpret_cast<float4*>(Vs + key * LDV + c * 4) = vv;
        }
    }
    __syncthreads();
    if (q0 >= L) return;

    f32x4m sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) sc[kt] = f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        const float* kp = KsT + (8 * lg + st) * LDK + li;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
            if (kt < nkt) sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[kt * 16], qv[st], sc[kt], 0, 0, 0);
    }
    // softmax over the keys of this lane's query: registers, then the four lane groups
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
        if (kt < nkt) {
            if (kt == nkt - 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * 16 + 4 * lg + r >= L) sc[kt][r] = -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, sc[kt][r]);
        }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float m2 = m * 1.4426950408889634f;
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
        if (kt < nkt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], 1.4426950408889634f, -m2));
                sc[kt][r] = e;
                l += e;
            }
        }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    f32x4m o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
        if (kt < nkt) {
            const float* vp = Vs + (kt * 16 + 4 * lg) * LDV + li;
            const int rem = L - kt * 16;        // k-step r covers keys r, 4 + r, 8 + r, 12 + r of the tile
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < rem) {
                    const float pr = sc[kt][r] * inv;
                    o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pr, vp[r * LDV], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pr, vp[r * LDV + 16], o1, 0, 0, 0);
                }
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qrow = q0 + 4 * lg + r;
        if (qrow < L) {
            float* dst = OUT + (size_t)(t0 + qrow) * 256 + head * 32 + li;
            dst[0] = o0[r];
            dst[16] = o1[r];
        }
    }
}

template <bool GATHER>
static int launch_enc_attn_t(const float* QK, const float* V, float* OUT, const int* off, int B, int Lmax,
                             const L0Gather& g, hipStream_t s) {
    CONE_REQUIRE(Lmax >= 1 && Lmax <= 192, "enc attention: window length %d not in [1,192]", Lmax);
    if (B <= 0) return 0;
    dim3 grid(B, 8), block(256);
    const int nkb = (Lmax + 31) / 32;
    ProfScope ps(PK_ENC_ATTN, B, Lmax, GATHER, nullptr, s);
    if (g_attn16) {
        const int nkt = max(6, (Lmax + 15) / 16);     // short batches ride on the 6-wave build (spare waves exit early)
#define CONE_ATTN16(N) case N: hipLaunchKernelGGL((enc_attn16_kernel<N, GATHER>), grid, dim3(64 * N), 0, s, QK, V, OUT, off, g); break;
        switch (nkt) {
            CONE_ATTN16(6)
            CONE_ATTN16(7) CONE_ATTN16(8) CONE_ATTN16(9) CONE_ATTN16(10) CONE_ATTN16(11) CONE_ATTN16(12)
        }
#undef CONE_ATTN16
        CONE_LAUNCH_CHECK();
        return 0;
    }
    if (nkb <= 4) hipLaunchKernelGGL((enc_attn_kernel<4, GATHER>), grid, block, 0, s, QK, V, OUT, off, g);
    else if (nkb == 5) hipLaunchKernelGGL((enc_attn_kernel<5, GATHER>), grid, block, 0, s, QK, V, OUT, off, g);
    else hipLaunchKernelGGL((enc_attn_kernel<6, GATHER>), grid, block, 0, s, QK, V, OUT, off, g);
    CONE_LAUNCH_CHECK();
    return 0;
}

int launch_enc_attn(const float* QK, const float* V, float* OUT, const int* off, int B, int Lmax,
                    hipStream_t s) {
    return launch_enc_attn_t<false>(QK, V, OUT, off, B, Lmax, L0Gather{}, s);
}

int launch_enc_attn_l0(const float* qkv_vid, const float* qkv_txt, const float* pos_qk, const int* vrow0,
                       const int* vlen, const int* trow0, float* OUT, const int* off, int B, int Lmax,
                       hipStream_t s) {
    return launch_enc_attn_t<true>(nullptr, nullptr, OUT, off, B, Lmax,
                                   L0Gather{qkv_vid, qkv_txt, pos_qk, vrow0, vlen, trow0}, s);
}

// Decoder attentions: NQ (<= 8) query slots per window, one wavefront per (window, head).
// off == nullptr : keys are the window's own NQ slot rows (self-attention, no mask);
// off != nullptr : keys are memory tokens off[b] .. off[b+1] (cross-attention, <= 192 keys).
constexpr int kSmallMaxKeys = 192;
__global__ __launch_bounds__(64) void small_attn_kernel(const float* __restrict__ Q, int ldq,
                                                        const float* __restrict__ K, int ldk,
                                                        const float* __restrict__ V, int ldv,
                                                        float* __restrict__ OUT, int ldo,
                                                        const int* __restrict__ off, int nq) {
    __shared__ float qs[8][32];
    __shared__ float ps[8][kSmallMaxKeys];
    const int b = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
    const int k0 = off ? off[b] : b * nq;
    const int L = off ? off[b + 1] - k0 : nq;
    if (lane < nq * 8) {
        const int qi = lane >> 3, c = lane & 7;
        const float4 x = *reinterpret_cast<const float4*>(Q + (size_t)(b * nq + qi) * ldq + head * 32 + c * 4);
        qs[qi][c * 4] = x.x * kQScale; qs[qi][c * 4 + 1] = x.y * kQScale;
        qs[qi][c * 4 + 2] = x.z * kQScale; qs[qi][c * 4 + 3] = x.w * kQScale;
    }
    __syncthreads();
    float sc[3][8];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const int j = lane + 64 * jj;
#pragma unroll
        for (int qi = 0; qi < 8; ++qi) sc[jj][qi] = -INFINITY;
        if (j < L) {
            float kv[32];
            const float* kp = K + (size_t)(k0 + j) * ldk + head * 32;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 x = reinterpret_cast<const float4*>(kp)[u];
                kv[4 * u] = x.x; kv[4 * u + 1] = x.y; kv[4 * u + 2] = x.z; kv[4 * u + 3] = x.w;
            }
#pragma unroll
            for (int qi = 0; qi < 8; ++qi)
                if (qi < nq) {
                    float a = 0.f;
#pragma unroll
                    for (int d = 0; d < 32; ++d) a = fmaf(qs[qi][d], kv[d], a);
                    sc[jj][qi] = a;
                }
        }
    }
#pragma unroll
    for (int qi = 0; qi < 8; ++qi)
        if (qi < nq) {
            const float m = wave_max(fmaxf(fmaxf(sc[0][qi], sc[1][qi]), sc[2][qi]));
            float e[3], l = 0.f;
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                e[jj] = (lane + 64 * jj < L) ? expf(sc[jj][qi] - m) : 0.f;
                l += e[jj];
            }
            const float inv = 1.0f / wave_sum(l);
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
                if (lane + 64 * jj < L) ps[qi][lane + 64 * jj] = e[jj] * inv;
        }
    __syncthreads();
    const int d = lane & 31, h = lane >> 5;
    float o[8];
#pragma unroll
    for (int qi = 0; qi < 8; ++qi) o[qi] = 0.f;
    for (int j = h; j < L; j += 2) {
        const float v = V[(size_t)(k0 + j) * ldv + head * 32 + d];
#pragma unroll
        for (int qi = 0; qi < 8; ++qi)
            if (qi < nq) o[qi] = fmaf(ps[qi][j], v, o[qi]);
    }
#pragma unroll
    for (int qi = 0; qi < 8; ++qi)
        if (qi < nq) {
            const float t = o[qi] + __shfl_xor(o[qi], 32, 64);
            if (h == 0) OUT[(size_t)(b * nq + qi) * ldo + head * 32 + d] = t;
        }
}

int launch_small_attn(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* OUT,
                      int ldo, const int* off, int B, int nq, int Lmax, hipStream_t s) {
    CONE_REQUIRE(nq >= 1 && nq <= 8, "decoder attention: num_queries=%d not in [1,8]", nq);
    CONE_REQUIRE(Lmax <= kSmallMaxKeys, "decoder attention: %d keys > %d", Lmax, kSmallMaxKeys);
    if (B <= 0) return 0;
    hipLaunchKernelGGL(small_attn_kernel, dim3(B, 8), dim3(64), 0, s, Q, ldq, K, ldk, V, ldv, OUT, ldo, off, nq);
    CONE_LAUNCH_CHECK();
    return 0;
}

}  // namespace cone
