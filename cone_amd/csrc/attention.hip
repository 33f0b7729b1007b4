// Multi-head attention cores of the Moment-DETR window model (head_dim = 32).
//
//   enc_attn16_kernel : encoder self-attention (cone/transformer.py:233-240 -> nn.MultiheadAttention),
//                       one workgroup per (window, head) over the window's PACKED tokens
//                       (video clips then text tokens; padded keys simply do not exist, which is what
//                       the reference's -inf key_padding_mask amounts to).
//   small_attn_kernel : the decoder's attentions with Nq<=8 query slots per window
//                       (cone/transformer.py:296-311): self-attention over the slots and
//                       cross-attention to the window's memory tokens.
//
// enc_attn16_kernel layout (exact-fp32 MFMA 16x16x4), one wave per 16-query tile, workgroup = ceil(Lmax/16) waves.
// What bounds it (round 3, tools/attn_bench.py, one box): the time is invariant to the instruction count (a form without
// the per-key-tile branches: 1 164 -> 723 instructions, 201 -> 14 branches: 2.74 vs 2.74 ms), to the LDS instruction count
// (round 2) and to the HBM traffic (gather mode reads a quarter of the bytes of the packed mode: same time) -- a
// workgroup's LIFETIME (load round trips -> stage -> barrier -> S^T -> softmax -> P.V -> store) times the four workgroups
// a CU holds (7 waves each, 32 wave slots) is what counts.  Hence: all loads of a workgroup in ONE round trip (was four
// serial ones: 2.79 -> 2.66 ms), every key tile always walked (no branches, zero rows + masked scores).  Two query tiles
// per wave (4-wave workgroups, five per CU instead of four 7-wave ones, all loads still in one round trip) measured
// 2.61 / 2.67 / 2.83 ms against 2.64 / 2.67 / 2.91 (packed / gather / pos-add): within 3 %, not kept.
//
//   a window of 101 tokens pads to 112 x 112 scores, a wave keeps 4 registers per key tile.
//   S^T tile: A = K (keys on accumulator rows 4g + r, g = lane / 16), B = Q^T (query = lane % 16): a lane owns one
//             query column, so the row softmax is an in-register reduction plus two cross-group shuffles, and the
//             probability registers are directly the A operand of O = P . V (k slot g <-> key 4g + r of the tile;
//             the MFMA k index may be permuted freely as long as A and B agree).  The head dim is walked as
//             d = 8 g + step so that a lane's 8 query values are two contiguous float4.
//   P.V     : V rows are read at stride 36 floats, K is staged d-major [32][KP + 2]: both conflict-free for ds_read_b32.
//             The keys of a tile sit PERMUTED in both images (key kk in row 4 (kk % 4) + kk / 4), so that P.V's MFMA r covers the
//             consecutive keys 4 r .. 4 r + 3 and is skipped when they lie past the window's end.
#include "common.h"

#ifndef CONE_ATTN_NT
#define CONE_ATTN_NT 0
#endif

namespace cone {

constexpr float kQScale = 0.17677669529663687f;  // sqrt(1/32), applied to q after projection
// the encoder kernels carry their scores in the log2 domain: q is scaled by sqrt(1/32) * log2(e), the softmax uses exp2
constexpr float kQScaleLog2 = 0.17677669529663687f * 1.4426950408889634f;


typedef float f32x4m __attribute__((ext_vector_type(4)));
typedef float f32x2m __attribute__((ext_vector_type(2)));

// Row softmax of one query column held as NKT x 4 score registers per lane (key 16 kt + 4 r + lg: the keys of a tile are
// PERMUTED on the way into the S^T operand -- row 4 lg + r of the accumulator is key 4 r + lg -- so that the four keys of
// P.V's MFMA r are CONSECUTIVE (4 r .. 4 r + 3) and the MFMAs of key quads past the window's end can be skipped), shared by both kernel
// forms (same operations in the same order: identical bits).  Vector instructions do not hide under the exact-fp32 MFMA on
// this part (tools/probe/mfma_valu_overlap.hip), so every one of them is on the critical path -- the count is what matters:
//   * the scores arrive in the log2 domain (log2(e) is folded into the q scale): exp2(s - m) is ONE packed subtract per two
//     scores (v_pk_add_f32) + the exponential, instead of an fma per score;
//   * the row sum runs on two-wide registers (v_pk_add_f32): 14 adds instead of 28;
//   * the probabilities are NOT normalised (28 multiplies): P.V runs on the raw exponentials and the 8 OUTPUT registers are
//     scaled by 1 / sum afterwards (the sum of output row 4 lg + r lives in the lane whose query is that row: one
//     ds_bpermute per r) -- attn_normalise().
// Only a tile that reaches past the window's last key is masked (wave-uniform test on the scalar L).  Leaves exp2(s - m) in
// sc, returns 1 / sum.
// lane ^ 16 / lane ^ 32 exchanges as ds_bpermute with the byte address computed from the lane id the caller already holds
// (__shfl_xor re-derives the lane id and bounds-checks it: ~10 vector instructions per call).
__device__ __forceinline__ float lane_xor(float v, int lane, int mask) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ mask) << 2, __float_as_int(v)));
}

template <int NKT>
__device__ __forceinline__ float attn_softmax(f32x4m (&sc)[NKT], int L, int lane, int lg) {
    const int lim = L - lg;                         // key 16 kt + 4 r + lg is real iff 16 kt + 4 r < lim
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (16 * (kt + 1) > L) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[kt][r] = (16 * kt + 4 * r < lim) ? sc[kt][r] : -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, sc[kt][r]);
    }
    m = fmaxf(m, lane_xor(m, lane, 16));
    m = fmaxf(m, lane_xor(m, lane, 32));
    const f32x2m m2 = {m, m};
    f32x2m l2 = {0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        f32x2m a = sc[kt].xy - m2, b = sc[kt].zw - m2;
        a.x = __builtin_amdgcn_exp2f(a.x); a.y = __builtin_amdgcn_exp2f(a.y);
        b.x = __builtin_amdgcn_exp2f(b.x); b.y = __builtin_amdgcn_exp2f(b.y);
        sc[kt].xy = a; sc[kt].zw = b;
        l2 += a;
        l2 += b;
    }
    float l = l2.x + l2.y;
    l += lane_xor(l, lane, 16);
    l += lane_xor(l, lane, 32);
    return 1.0f / l;
}

// o0 / o1 register r = output row (query) 4 lg + r of the tile; its 1 / sum sits in every lane whose li is that row.
__device__ __forceinline__ void attn_normalise(f32x4m& o0, f32x4m& o1, float inv, int lane, int lg) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float s = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane & 48) + 4 * lg + r) << 2, __float_as_int(inv)));
        o0[r] *= s;
        o1[r] *= s;
    }
}

// MODE: ATTN_PACKED / ATTN_GATHER / ATTN_POSADD (common.h).  In the two table modes the q | k rows of a clip token
// get the static row pos_qk[(lv, p)] added in the staging loads -- the adds the reference performs as
// ``q = k = src + pos`` ahead of in_proj (cone/transformer.py:237), moved behind the (linear) projection.
// Addressing: every load is `uniform base (SGPR pair) + 32-bit lane offset` (global_load ... saddr): the window's row bases
// are scalars (t0, vrow0[b], trow0[b], the table row of (lv, 0)), a lane contributes token * stride + column.  A text token
// reads the table's ZERO row (AttnSrc::pos_zero_row: the last row of cone_pos_tables) instead of skipping the add, so the
// adds are unconditional and no 64-bit pointer is ever selected per lane except the gather mode's clip / text source.
// Windows of 193 .. 256 tokens (NKT 13 .. 16: WINDOW_LENGTH is a user argument of the reference's scripts, README.md:94-98) need
// 57 .. 70 KiB for the two images: dynamic LDS (opt-in above 64 KiB), one workgroup of up to 1 024 threads; the shipped
// window lengths (NKT <= 12) keep their static images and their instruction sequence.
// MODE_ = MODE | 4 (--use_txt_pos on the table path): a text token adds ITS position row txt_pos_qk[trow0[b] + j] (this layer's
// image of LayerNorm(src_txt + position_embeddings), cone/model.py:106) instead of the zero row -- the same unconditional add
// from another uniform base; the instantiations without the bit are untouched.
template <int NKT, int MODE_>
__global__ __launch_bounds__(64 * NKT, (NKT > 12 ? 4 : 6)) void enc_attn16_kernel(AttnSrc a, float* __restrict__ OUT,
                                                                 const int* __restrict__ off) {
    constexpr int MODE = MODE_ & 3;
    constexpr bool TXT = (MODE_ & 4) != 0;
    constexpr int KP = 16 * NKT, LDK = KP + 2, LDV = 36, NT = 64 * NKT;
    constexpr bool DYN = NKT > 12;
    __shared__ float KsT_s[DYN ? 4 : 32 * LDK];
    __shared__ __attribute__((aligned(16))) float Vs_s[DYN ? 4 : KP * LDV];
    extern __shared__ __attribute__((aligned(16))) float attn_dyn[];
    float* const KsT = DYN ? attn_dyn : KsT_s;
    float* const Vs = DYN ? attn_dyn + 32 * LDK : Vs_s;
    const int b = blockIdx.y, head = blockIdx.x;        // the 8 heads of a window are dispatched together
    const int t0 = off[b];
    const int L = off[b + 1] - t0;
    if (L <= 0) return;                                 // an empty window owns no row (and L - 1 would index before the buffer)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const unsigned hc = head * 32;
    int lv = 0;
    // rows of token `tok`: q at qrow(tok), k at krow(tok), v at vrow(tok) (column hc of each), all as base + unsigned offset
    const float *qb = nullptr, *kb = nullptr, *vb = nullptr, *tb = nullptr, *pb = nullptr;
    unsigned sq = 768, sk = 768, sv = 768;
    int pbase = 0;
    if (MODE != ATTN_PACKED) {
        lv = a.vlen[b];
        pbase = lv * (lv - 1) / 2;
        pb = a.pos_qk + hc;
    }
    if (MODE == ATTN_GATHER) {
        // token tok is row tok of `qb` (a clip: tok < lv) or of `tb` (a text token; tb is biased by -lv rows)
        qb = a.qkv_vid + (size_t)a.vrow0[b] * 768 + hc;
        tb = a.qkv_txt + ((ptrdiff_t)a.trow0[b] - lv) * 768 + hc;
    } else {
        sq = a.ldq; sk = a.ldk; sv = a.ldv;
        qb = a.Q + (size_t)t0 * sq + hc; kb = a.K + (size_t)t0 * sk + hc; vb = a.V + (size_t)t0 * sv + hc;
    }
    const int zrow = a.pos_zero_row;
    const float* tpb = nullptr;             // TXT: text token `tok` (>= lv) adds row tok of tpb (biased by -lv rows, like tb)
    if (TXT) tpb = a.txt_pos_qk + ((ptrdiff_t)a.trow0[b] - lv) * 512 + hc;
    auto prow = [&](int row) -> const float* {
        if (TXT && row >= lv) return tpb + (unsigned)row * 512u;
        return pb + (unsigned)(row < lv ? pbase + row : zrow) * 512u;
    };

    // ONE memory round trip per workgroup: the query values, both staging passes (key rows kr and kr + NT/8: KP = 2 NT/8
    // exactly) and their position rows are all requested before anything is waited for.  Rows past the window re-read its
    // last row (their scores are masked, their probabilities exactly 0), a text token's position row is the table's zero
    // row: no load sits behind a branch.
    const int q0 = wave * 16;
    const int kr = tid >> 3;
    const unsigned c4 = (tid & 7) * 4;
    f32x4m qx[2], qt[2], kv[2], vv[2], kt_[2];
    {
        const int qrow = min(q0 + li, L - 1);
        const float* qp;
        if (MODE == ATTN_GATHER) qp = (qrow < lv ? qb : tb) + (unsigned)qrow * 768u + 8u * lg;
        else qp = qb + (unsigned)qrow * sq + 8u * lg;
        qx[0] = *reinterpret_cast<const f32x4m*>(qp);
        qx[1] = *reinterpret_cast<const f32x4m*>(qp + 4);
        if (MODE != ATTN_PACKED) {
            const float* pr = prow(qrow) + 8u * lg;
            qt[0] = *reinterpret_cast<const f32x4m*>(pr);
            qt[1] = *reinterpret_cast<const f32x4m*>(pr + 4);
        }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = min(kr + it * (NT / 8), L - 1);
        if (MODE == ATTN_GATHER) {
            const float* rp = (row < lv ? qb : tb) + (unsigned)row * 768u + c4;
            kv[it] = *reinterpret_cast<const f32x4m*>(rp + 256);
            vv[it] = *reinterpret_cast<const f32x4m*>(rp + 512);
        } else {
#if CONE_ATTN_NT        // A/B (tools/ab_variants.sh): the packed q | k | v rows are read once -- non-temporal key / value loads:
                        // 2.74 - 2.80 / 2.88 - 2.90 ms either way (tools/attn_bench.py), off
            kv[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4m*>(kb + (unsigned)row * sk + c4));
            vv[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4m*>(vb + (unsigned)row * sv + c4));
#else
            kv[it] = *reinterpret_cast<const f32x4m*>(kb + (unsigned)row * sk + c4);
            vv[it] = *reinterpret_cast<const f32x4m*>(vb + (unsigned)row * sv + c4);
#endif
        }
        if (MODE != ATTN_PACKED)
            kt_[it] = *reinterpret_cast<const f32x4m*>(prow(row) + 256u + c4);
    }
    float qv[8];
    {
        f32x4m x0 = qx[0], x1 = qx[1];
        if (MODE != ATTN_PACKED) { x0 += qt[0]; x1 += qt[1]; }
        x0 *= kQScaleLog2; x1 *= kQScaleLog2;
#pragma unroll
        for (int j = 0; j < 4; ++j) { qv[j] = x0[j]; qv[4 + j] = x1[j]; }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int key = kr + it * (NT / 8);
        f32x4m k4 = kv[it];
        if (MODE != ATTN_PACKED) k4 += kt_[it];
        // key kk of a tile sits in operand row 4 (kk % 4) + kk / 4, in both images
        const int pos = (key & ~15) + 4 * (key & 3) + ((key >> 2) & 3);
        float* kd = KsT + c4 * LDK + pos;
        kd[0] = k4[0]; kd[LDK] = k4[1]; kd[2 * LDK] = k4[2]; kd[3 * LDK] = k4[3];
        *reinterpret_cast<f32x4m*>(Vs + pos * LDV + c4) = vv[it];
    }
    __syncthreads();
    if (q0 >= L) return;

    // All NKT key tiles are always walked: the rows past the window's last key hold finite copies of its last row, their
    // scores are masked to -inf (probability exactly 0), so the extra MFMAs add exact zeros -- and the per-tile "does this
    // tile exist" branches are gone.  A batch pads to its own longest window (the launcher picks NKT).
    f32x4m sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) sc[kt] = f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        const float* kp = KsT + (8 * lg + st) * LDK + li;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
            sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[kt * 16], qv[st], sc[kt], 0, 0, 0);
    }
    const float inv = attn_softmax<NKT>(sc, L, lane, lg);
    f32x4m o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const float* vp = Vs + (kt * 16 + 4 * lg) * LDV + li;       // MFMA r: image rows 4 lg + r = keys 16 kt + 4 r + lg, lg = 0 .. 3
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // the last two tiles of a launch may reach past this window: a key quad entirely behind its end multiplies exact
            // zeros (wave-uniform test; earlier tiles are always walked: no branch)
            if (kt < NKT - 2 || 16 * kt + 4 * r < L) {
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[kt][r], vp[r * LDV], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[kt][r], vp[r * LDV + 16], o1, 0, 0, 0);
            }
        }
    }
    attn_normalise(o0, o1, inv, lane, lg);
    float* ob = OUT + (size_t)t0 * 256 + hc;
    const int qr0 = q0 + 4 * lg;
    const unsigned oo = (unsigned)qr0 * 256u + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (qr0 + r < L) {
            ob[oo + 256u * r] = o0[r];
            ob[oo + 256u * r + 16u] = o1[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// enc_attn_wave_kernel: ONE WAVE per (window, head), the head's K and V resident in REGISTERS, no LDS, no barrier -- a second,
// independent formulation of the same arithmetic (same MFMA operand assignment, same order of every sum: IDENTICAL BITS, which
// tests/test_gpu_parity.py asserts), kept as a cross-check and as the record of what bounds this kernel.  A wave loads its
// head's keys once as the A operand of S^T (lane = key, 8 registers per key tile) and its values once as the B operand of P.V
// (lane = channel, 8 registers per key tile), then walks the window's query tiles: 16 NKT MFMAs per tile back to back, the row
// softmax in registers in between.  20 NKT + ~50 VGPRs: two waves per SIMD up to NKT = 9.
//
// Measured (tools/attn_bench.py, 20 000 windows; every variant verified bit-identical): this form 2.72 / 2.97 / 3.13 ms (packed /
// gather / pos-add) against 2.71 / 2.83 / 2.94 of enc_attn16_kernel; persistent workgroups, the second wave of a SIMD started
// half an item late, the S^T MFMAs of tile i + 1 interleaved with the softmax of tile i (sched_group_barrier), the eight waves
// in lockstep phases: 2.67 - 2.9, 2.75, 2.75, 3.4 ms.  With every load served from cache and no store: 2.59 ms; one wave per
// SIMD instead of two: 2.85 ms.  The parts ADD: 1.0 ms of loads + 0.65 of softmax / stores + 1.1 of MFMA.  Why -- tools/probe/
// mfma_valu_overlap.hip: v_mfma_f32_16x16x4_f32 issues every 33 cycles alone, every 44 / 46 / 54 / 63 cycles with 2 / 4 / 8 / 12
// independent v_fma_f32 behind each MFMA in the same wave, and a VALU-only partner wave on the SIMD slows an MFMA-only wave the
// same way (54 / 67 / 76 cycles per MFMA at 4 / 8 / 12 VALU per MFMA): the exact-fp32 matrix instruction runs at the fp32
// VECTOR rate (64 FLOP / clock / SIMD) and vector work does not hide under it, from the same wave or from another.  The
// kernel's floor is therefore its MFMA time PLUS its vector time: 784 MFMAs x 32 cycles + ~1 900 vector instructions per
// (window, head) = 25 k + >= 6 k cycles per SIMD = 2.2 ms at the 2.1 GHz the chip holds here; both forms sit at 2.7.
template <int NKT, int MODE>
__global__ __launch_bounds__(256, 2) void enc_attn_wave_kernel(AttnSrc a, float* __restrict__ OUT,
                                                               const int* __restrict__ off) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int head = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t0 = off[b];
    const int L = off[b + 1] - t0;
    if (L <= 0) return;
    const int li = lane & 15, lg = lane >> 4;
    const unsigned hc = head * 32;
    // the addressing of enc_attn16_kernel: uniform bases + 32-bit lane offsets, a text token adds the table's zero row
    int lv = 0, pbase = 0;
    const float *qb = nullptr, *kb = nullptr, *vb = nullptr, *tb = nullptr, *pb = nullptr;
    unsigned sq = 768, sk = 768, sv = 768;
    if (MODE != ATTN_PACKED) { lv = a.vlen[b]; pbase = lv * (lv - 1) / 2; pb = a.pos_qk + hc; }
    if (MODE == ATTN_GATHER) {
        qb = a.qkv_vid + (size_t)a.vrow0[b] * 768 + hc;
        tb = a.qkv_txt + ((ptrdiff_t)a.trow0[b] - lv) * 768 + hc;
    } else {
        sq = a.ldq; sk = a.ldk; sv = a.ldv;
        qb = a.Q + (size_t)t0 * sq + hc; kb = a.K + (size_t)t0 * sk + hc; vb = a.V + (size_t)t0 * sv + hc;
    }
    const int zrow = a.pos_zero_row;
    auto qptr = [&](int tok) { return MODE == ATTN_GATHER ? (tok < lv ? qb : tb) + (unsigned)tok * 768u : qb + (unsigned)tok * sq; };
    auto kptr = [&](int tok) { return MODE == ATTN_GATHER ? (tok < lv ? qb : tb) + (unsigned)tok * 768u + 256 : kb + (unsigned)tok * sk; };
    auto vptr = [&](int tok) { return MODE == ATTN_GATHER ? (tok < lv ? qb : tb) + (unsigned)tok * 768u + 512 : vb + (unsigned)tok * sv; };
    auto pptr = [&](int tok) { return pb + (unsigned)(tok < lv ? pbase + tok : zrow) * 512u; };
    // query values of a tile: 8 per lane (query = li, channels 8 lg ..), position row added, scaled
    f32x4m qx[2], qt[2];
    auto load_q = [&](int q0) {
        const int qrow = min(q0 + li, L - 1);
        const float* qp = qptr(qrow) + 8 * lg;
        qx[0] = *reinterpret_cast<const f32x4m*>(qp);
        qx[1] = *reinterpret_cast<const f32x4m*>(qp + 4);
        if (MODE != ATTN_PACKED) {
            const float* ap = pptr(qrow) + 8 * lg;
            qt[0] = *reinterpret_cast<const f32x4m*>(ap);
            qt[1] = *reinterpret_cast<const f32x4m*>(ap + 4);
        }
    };
    load_q(0);
    // keys: kreg[kt][st] = K[key of operand row li][8 lg + st] (rows past the window: a copy of its last row, masked below)
    float kreg[NKT][8];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const int key = min(16 * kt + 4 * (li & 3) + (li >> 2), L - 1);     // operand row li = key 4 (li % 4) + li / 4 of the tile
        const float* kp_ = kptr(key) + 8 * lg;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4m k4 = *reinterpret_cast<const f32x4m*>(kp_ + 4 * u);
            if (MODE != ATTN_PACKED) k4 += *reinterpret_cast<const f32x4m*>(pptr(key) + 256 + 8 * lg + 4 * u);
#pragma unroll
            for (int j = 0; j < 4; ++j) kreg[kt][4 * u + j] = k4[j];
        }
    }
    // values: vreg[kt][r][dt] = V[key 16 kt + 4 r + lg][16 dt + li]
    float vreg[NKT][4][2];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* vp_ = vptr(min(16 * kt + 4 * r + lg, L - 1));
            vreg[kt][r][0] = vp_[li];
            vreg[kt][r][1] = vp_[16 + li];
        }
    for (int q0 = 0; q0 < L; q0 += 16) {
        float qv[8];
        {
            f32x4m x0 = qx[0], x1 = qx[1];
            if (MODE != ATTN_PACKED) { x0 += qt[0]; x1 += qt[1]; }
            x0 *= kQScaleLog2; x1 *= kQScaleLog2;
#pragma unroll
            for (int j = 0; j < 4; ++j) { qv[j] = x0[j]; qv[4 + j] = x1[j]; }
        }
        load_q(q0 + 16 < L ? q0 + 16 : q0);         // the next tile's query rows under this tile's MFMAs
        f32x4m sc[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) sc[kt] = f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 8; ++st)
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kreg[kt][st], qv[st], sc[kt], 0, 0, 0);
        const float inv = attn_softmax<NKT>(sc, L, lane, lg);
        f32x4m o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[kt][r], vreg[kt][r][0], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[kt][r], vreg[kt][r][1], o1, 0, 0, 0);
            }
        attn_normalise(o0, o1, inv, lane, lg);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qrow = q0 + 4 * lg + r;
            if (qrow < L) {
                float* dst = OUT + (size_t)(t0 + qrow) * 256 + hc + li;
                dst[0] = o0[r];
                dst[16] = o1[r];
            }
        }
    }
}

template <int MODE>
static int launch_enc_attn_t(const AttnSrc& a, float* OUT, const int* off, int B, int Lmax, hipStream_t s) {
    const int nkt = max(6, (Lmax + 15) / 16);     // short batches ride on the 6-wave build (spare waves exit early)
    if (a.form == 2 && nkt <= 9 && !(MODE & 4)) { // the register-resident cross-check form (same bits), on request
        dim3 wgrid(2, B);
#define CONE_ATTNW(N) case N: hipLaunchKernelGGL((enc_attn_wave_kernel<N, (MODE & 3)>), wgrid, dim3(256), 0, s, a, OUT, off); break;
        switch (nkt) { CONE_ATTNW(6) CONE_ATTNW(7) CONE_ATTNW(8) CONE_ATTNW(9) }
#undef CONE_ATTNW
        CONE_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid(8, B);
#define CONE_ATTN16(N) case N: hipLaunchKernelGGL((enc_attn16_kernel<N, MODE>), grid, dim3(64 * N), 0, s, a, OUT, off); break;
    // 13 .. 16 key tiles: both images in dynamic LDS (the opt-in above 64 KiB once per device)
#define CONE_ATTN16D(N)                                                                                                     \
    case N: {                                                                                                               \
        constexpr int bytes = (32 * (16 * N + 2) + 16 * N * 36) * 4;                                                        \
        static DeviceOnce once;                                                                                             \
        CONE_CHECK_HIP(device_once(once, [] {                                                                               \
            return hipFuncSetAttribute((const void*)enc_attn16_kernel<N, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); \
        }));                                                                                                                \
        hipLaunchKernelGGL((enc_attn16_kernel<N, MODE>), grid, dim3(64 * N), bytes, s, a, OUT, off);                        \
        break;                                                                                                              \
    }
    switch (nkt) {
        CONE_ATTN16(6)
        CONE_ATTN16(7) CONE_ATTN16(8) CONE_ATTN16(9) CONE_ATTN16(10) CONE_ATTN16(11) CONE_ATTN16(12)
        CONE_ATTN16D(13) CONE_ATTN16D(14) CONE_ATTN16D(15) CONE_ATTN16D(16)
    }
#undef CONE_ATTN16
#undef CONE_ATTN16D
    CONE_LAUNCH_CHECK();
    return 0;
}

int launch_enc_attn(int mode, const AttnSrc& a, float* OUT, const int* off, int B, int Lmax, hipStream_t s) {
    CONE_REQUIRE(Lmax >= 1 && Lmax <= 256, "enc attention: window length %d not in [1,256]", Lmax);
    if (B <= 0) return 0;
    ProfScope ps(PK_ENC_ATTN, B, Lmax, mode, nullptr, s);
    switch (mode) {
        case ATTN_PACKED:
            CONE_REQUIRE(a.Q && a.K && a.V, "enc attention: packed mode needs Q, K, V");
            return launch_enc_attn_t<ATTN_PACKED>(a, OUT, off, B, Lmax, s);
        case ATTN_GATHER:
            CONE_REQUIRE(a.qkv_vid && a.qkv_txt && a.pos_qk && a.vrow0 && a.vlen && a.trow0 && a.pos_zero_row >= 0,
                         "enc attention: gather mode needs the layer-0 caches");
            if (a.txt_pos_qk) return launch_enc_attn_t<ATTN_GATHER | 4>(a, OUT, off, B, Lmax, s);
            return launch_enc_attn_t<ATTN_GATHER>(a, OUT, off, B, Lmax, s);
        case ATTN_POSADD:
            CONE_REQUIRE(a.Q && a.K && a.V && a.pos_qk && a.vlen && a.pos_zero_row >= 0,
                         "enc attention: pos-add mode needs Q, K, V, pos_qk, vlen");
            if (a.txt_pos_qk) {
                CONE_REQUIRE(a.trow0, "enc attention: text position rows need trow0");
                return launch_enc_attn_t<ATTN_POSADD | 4>(a, OUT, off, B, Lmax, s);
            }
            return launch_enc_attn_t<ATTN_POSADD>(a, OUT, off, B, Lmax, s);
    }
    set_error("enc attention: unknown mode %d", mode);
    return CONE_E_INVALID;
}

// Decoder attentions: NQ (<= MQ = 8 or 16) query slots per window, one wavefront per (window, head).
// off == nullptr : keys are the window's own NQ slot rows (self-attention, no mask);
// off != nullptr : keys are memory tokens off[b] .. off[b+1] (cross-attention, <= 192 keys).
constexpr int kSmallMaxKeys = 192;
template <int MQ>
__global__ __launch_bounds__(64) void small_attn_kernel(const float* __restrict__ Q, int ldq,
                                                        const float* __restrict__ K, int ldk,
                                                        const float* __restrict__ V, int ldv,
                                                        float* __restrict__ OUT, int ldo,
                                                        const int* __restrict__ off, int nq) {
    __shared__ float qs[MQ][32];
    __shared__ float ps[MQ][kSmallMaxKeys];
    const int b = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
    const int k0 = off ? off[b] : b * nq;
    const int L = off ? off[b + 1] - k0 : nq;
    for (int i = lane; i < nq * 8; i += 64) {
        const int qi = i >> 3, c = i & 7;
        const float4 x = *reinterpret_cast<const float4*>(Q + (size_t)(b * nq + qi) * ldq + head * 32 + c * 4);
        qs[qi][c * 4] = x.x * kQScale; qs[qi][c * 4 + 1] = x.y * kQScale;
        qs[qi][c * 4 + 2] = x.z * kQScale; qs[qi][c * 4 + 3] = x.w * kQScale;
    }
    __syncthreads();
    float sc[3][MQ];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const int j = lane + 64 * jj;
#pragma unroll
        for (int qi = 0; qi < MQ; ++qi) sc[jj][qi] = -INFINITY;
        if (j < L) {
            float kv[32];
            const float* kp = K + (size_t)(k0 + j) * ldk + head * 32;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 x = reinterpret_cast<const float4*>(kp)[u];
                kv[4 * u] = x.x; kv[4 * u + 1] = x.y; kv[4 * u + 2] = x.z; kv[4 * u + 3] = x.w;
            }
#pragma unroll
            for (int qi = 0; qi < MQ; ++qi)
                if (qi < nq) {
                    float a = 0.f;
#pragma unroll
                    for (int d = 0; d < 32; ++d) a = fmaf(qs[qi][d], kv[d], a);
                    sc[jj][qi] = a;
                }
        }
    }
#pragma unroll
    for (int qi = 0; qi < MQ; ++qi)
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
    float o[MQ];
#pragma unroll
    for (int qi = 0; qi < MQ; ++qi) o[qi] = 0.f;
    for (int j = h; j < L; j += 2) {
        const float v = V[(size_t)(k0 + j) * ldv + head * 32 + d];
#pragma unroll
        for (int qi = 0; qi < MQ; ++qi)
            if (qi < nq) o[qi] = fmaf(ps[qi][j], v, o[qi]);
    }
#pragma unroll
    for (int qi = 0; qi < MQ; ++qi)
        if (qi < nq) {
            const float t = o[qi] + __shfl_xor(o[qi], 32, 64);
            if (h == 0) OUT[(size_t)(b * nq + qi) * ldo + head * 32 + d] = t;
        }
}

// Decoder SELF-attention over the NQ slots of a window (cone/transformer.py:296-305), one wavefront per WINDOW: lane = (head
// l / 8, four channels 4 (l % 8) .. of the head's 32), so a slot's 256-channel row is one coalesced 1-KiB load per operand;
// the 32-channel dot products are finished by three shuffle steps inside the head's eight lanes; softmax over NQ keys in
// registers; the output row again one coalesced store.  (small_attn_kernel spends a whole wavefront on every (window, head):
// 160 000 one-wave workgroups for 20 000 windows, 264 us; this form 20 000 waves.)
template <int NQ>
__global__ __launch_bounds__(256) void dec_self_attn_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K,
                                                            int ldk, const float* __restrict__ V, int ldv,
                                                            float* __restrict__ OUT, int ldo, int B) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    float4 q[NQ], k[NQ], v[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const size_t row = (size_t)b * NQ + i;
        q[i] = reinterpret_cast<const float4*>(Q + row * ldq)[lane];
        k[i] = reinterpret_cast<const float4*>(K + row * ldk)[lane];
        v[i] = reinterpret_cast<const float4*>(V + row * ldv)[lane];
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        float sc[NQ];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            float d = ((q[i].x * kQScale) * k[j].x + (q[i].y * kQScale) * k[j].y) + ((q[i].z * kQScale) * k[j].z + (q[i].w * kQScale) * k[j].w);
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            d += __shfl_xor(d, 4, 64);
            sc[j] = d;
            m = fmaxf(m, d);
        }
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < NQ; ++j) { sc[j] = expf(sc[j] - m); l += sc[j]; }
        const float inv = 1.0f / l;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const float p = sc[j] * inv;
            o.x = fmaf(p, v[j].x, o.x); o.y = fmaf(p, v[j].y, o.y); o.z = fmaf(p, v[j].z, o.z); o.w = fmaf(p, v[j].w, o.w);
        }
        reinterpret_cast<float4*>(OUT + ((size_t)b * NQ + i) * ldo)[lane] = o;
    }
}

int launch_small_attn(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* OUT,
                      int ldo, const int* off, int B, int nq, int Lmax, hipStream_t s) {
    CONE_REQUIRE(nq >= 1 && nq <= 16, "decoder attention: num_queries=%d not in [1,16]", nq);
    CONE_REQUIRE(Lmax <= kSmallMaxKeys, "decoder attention: %d keys > %d", Lmax, kSmallMaxKeys);
    if (B <= 0) return 0;
    if (!off && (ldq | ldk | ldv | ldo) % 4 == 0) {     // self-attention over the slots: the shipped slot count (5) and the
                                                        // other counts the folded cross-attention is instantiated for
#define CONE_DSA(N)                                                                                                            \
        if (nq == N) {                                                                                                         \
            hipLaunchKernelGGL(dec_self_attn_kernel<N>, dim3((B + 3) / 4), dim3(256), 0, s, Q, ldq, K, ldk, V, ldv, OUT, ldo, B); \
            CONE_LAUNCH_CHECK();                                                                                               \
            return 0;                                                                                                          \
        }
        CONE_DSA(5) CONE_DSA(3) CONE_DSA(8) CONE_DSA(10)
#undef CONE_DSA
    }
    if (nq <= 8)
        hipLaunchKernelGGL(small_attn_kernel<8>, dim3(B, 8), dim3(64), 0, s, Q, ldq, K, ldk, V, ldv, OUT, ldo, off, nq);
    else        // (more slots than every shipped configuration trains with: cone/scripts/train_*.sh take NUM_QUERIES as an argument)
        hipLaunchKernelGGL(small_attn_kernel<16>, dim3(B, 8), dim3(64), 0, s, Q, ldq, K, ldk, V, ldv, OUT, ldo, off, nq);
    CONE_LAUNCH_CHECK();
    return 0;
}

}  // namespace cone
