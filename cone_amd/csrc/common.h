// Shared host/device helpers for libcone_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <mutex>

#include "../../include/cone_hip.h"

namespace cone {

void set_error(const char* fmt, ...);

#define CONE_CHECK_HIP(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            cone::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                            __LINE__);                                                        \
            return CONE_E_HIP;                                                                \
        }                                                                                     \
    } while (0)

#define CONE_REQUIRE(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            cone::set_error(__VA_ARGS__); \
            return CONE_E_INVALID;       \
        }                                \
    } while (0)

#define CONE_LAUNCH_CHECK() CONE_CHECK_HIP(hipGetLastError())

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// One-time kernel setup PER DEVICE: the opt-in to more than 64 KiB of dynamic LDS (hipFuncSetAttribute) applies to the
// current device's copy of the code object, and persistent grids are sized by that device's CU count -- a process that
// drives several GPUs through this library (the C ABI takes arbitrary streams) must get both for each of them.
constexpr int CONE_MAX_DEVICES = 64;
struct DeviceOnce {
    std::mutex mu;
    bool done[CONE_MAX_DEVICES] = {};
    hipError_t rc[CONE_MAX_DEVICES] = {};
    int n_cu[CONE_MAX_DEVICES] = {};
};
template <typename F>
static inline hipError_t device_once(DeviceOnce& st, F&& setup, int* n_cu = nullptr) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= CONE_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(st.mu);
    if (!st.done[dev]) {
        st.rc[dev] = setup();
        if (st.rc[dev] == hipSuccess)
            st.rc[dev] = hipDeviceGetAttribute(&st.n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev);
        st.done[dev] = true;
    }
    if (n_cu) *n_cu = st.n_cu[dev];
    return st.rc[dev];
}

// ---------------------------------------------------------------- device helpers
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef __HIPCC__
// Sum / max over the 64 lanes of a wavefront (wave64 on gfx950).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Sum over the 32 lanes that share (lane >> 5).
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Row index inside a 32x32 MFMA accumulator tile: register r of lane l holds
// D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]   (cdna_hip_programming.md section 3).
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
#endif

// ---------------------------------------------------------------- opt-in launch timing (prof.hip)
// one kind per KERNEL: a record's time is that kernel's launch and nothing else (the layer tail's wide / 64-row forms and
// the row GEMM's small form are other kernels than the persistent 128-row forms and have their own kinds)
enum ProfKind { PK_GEMM_128x128 = 0, PK_GEMM_128x128_A2 = 1, PK_GEMM_64x256 = 2, PK_ENC_ATTN = 3, PK_FRAME_SCORE = 4,
                PK_GEMM_ROWS = 5, PK_GEMM_ROWS16 = 6, PK_DEC_CROSS = 7, PK_FFN_FUSED = 8, PK_FFN_PROJ = 9,
                PK_FFN_WIDE = 10, PK_FFN_PROJ_WIDE = 11, PK_FFN_FUSED_NW4 = 12, PK_FFN_PROJ_NW4 = 13, PK_GEMM_ROWS_SMALL = 14 };
bool prof_enabled();
struct ProfScope {
    // a_dev: device-side row count of the whole job (wins over `a` when smaller); a_off: the launch covers rows
    // a_off .. a_off + a of that job, so its rows are clamp(*a_dev - a_off, 0, a)
    ProfScope(int kind, int64_t a, int64_t b, int64_t c, const int* a_dev, hipStream_t s, int64_t a_off = 0);
    ~ProfScope();
    int idx_; hipStream_t s_; bool counted_;
};

// ---------------------------------------------------------------- GEMM launcher (gemm.hip)
enum { EPI_RELU = 1, EPI_RESIDUAL = 2, EPI_LN = 4,
       GEMM_NO_SPREAD = 8 };      // (not an epilogue: keeps a small launch on the workgroup-per-16-rows form; A/B tests)

struct GemmArgs {
    const float* A; int lda;              // (M,K) row-major, K contiguous
    const float* A2; int lda2; int a2_mod; // optional addend; row index = a2_mod ? row % a2_mod : row
    const float* W; int ldw;              // (N,K): torch Linear weight
    const float* bias;                    // (N) or null
    const float* R; int ldr;              // residual (M,N), EPI_RESIDUAL
    int r_mod;                            // > 0: the residual is row-periodic, row m reads R[m % r_mod] (a per-slot table)
    const float* ln_g; const float* ln_b; // EPI_LN (N == 256)
    float* C; int ldc;
    float* C2; const float* ADD;          // optional second output C2 = C + ADD (same ldc), row tile only
    int M; const int* M_dev;              // rows; *M_dev wins when non-null (grid sized by M)
    int m_off;                            // the rows are rows m_off .. m_off + M of a larger job: *M_dev and r_mod count from its row 0
    int N, K;
    int flags;
    int variant;                          // tile family: 0 automatic (GEMM_ROWS8 where the shape allows), else forced
};
// GEMM_SQUARE: register-staged 128x128 / 64x256 tiles; GEMM_ROWS4 / GEMM_ROWS8: the 128x256 row-owning LDS-DMA tile
// with 4 waves x 32 rows (32x32x2) / 8 waves x 16 rows (16x16x4).  The family is a function of the shape and of
// this field only (never of M or of process state): a row of C is computed by the same instruction sequence
// whatever batch it sits in.
enum { GEMM_AUTO = 0, GEMM_SQUARE = 1, GEMM_ROWS4 = 2, GEMM_ROWS8 = 3 };
int launch_gemm(const GemmArgs& a, hipStream_t s);

// ---------------------------------------------------------------- fused feed-forward block (ffn.hip)
// OUT = LayerNorm(X + W2 relu(W1 X + b1) + b2): X, OUT (M, 256); W1 (ff, 256); W2 (256, ff).  OUT may alias nothing
// the kernel reads (X is re-read as the residual from registers only, but other workgroups read other rows).
bool ffn_fused_supported(int ff);
int launch_ffn_fused(const float* X, int ldx, const float* W1, const float* b1, const float* W2, const float* b2,
                     const float* ln_g, const float* ln_b, float* OUT, int ldo, int M, const int* M_dev, int ff,
                     hipStream_t s);
// The same with the block input computed in the kernel too: X = LayerNorm_p(R + A Wo^T + bo) (attention output
// projection + residual + norm), i.e. everything of a transformer layer behind its attention in one launch.
int launch_proj_ffn_fused(const float* A, int lda, const float* Wo, const float* bo, const float* R, int ldr,
                          const float* pg, const float* pb, const float* W1, const float* b1, const float* W2,
                          const float* b2, const float* ln_g, const float* ln_b, float* OUT, int ldo, int M,
                          const int* M_dev, int ff, hipStream_t s, const int* r_idx = nullptr, const float* R2 = nullptr,
                          const float* Wq = nullptr, const float* qb = nullptr, float* QKV = nullptr, int ldq = 0,
                          int n_qkv = 0);
int launch_proj_ffn_prenorm(const float* A, int lda, const float* Wo, const float* bo, const float* R, int ldr, const float* pg,
                            const float* pb, const float* W1, const float* b1, const float* W2, const float* b2, float* OUT,
                            int ldo, const float* n2g, const float* n2b, float* OUT2, int ldo2, int M, const int* M_dev, int ff,
                            hipStream_t s, const int* r_idx = nullptr, const float* R2 = nullptr);
// pre-norm form of the layer tail: OUT = x1 + FFN(LayerNorm(x1; pg, pb)), x1 = R + A Wo^T + bo (un-normalised stream);
// OUT2 (may be null) = LayerNorm(OUT; n2g, n2b)
// Wq != null: the kernel also writes QKV (M, n_qkv) = OUT Wq^T + qb (the next layer's q | k | v projection) from the
// registers that hold OUT
bool ffn_fused_qkv_fits(int ff, int n_qkv);
// r_idx != null: residual row i is gathered -- r_idx[i] >= 0: row r_idx[i] of R, else row ~r_idx[i] of R2 (launch_row_index)

// ---------------------------------------------------------------- wide form for a few row groups (ffn_wide.hip)
// The same two computations with ONE workgroup per 16 rows whose eight waves split the output elements (projection and
// GEMM2 by output channels, GEMM1 by hidden units): every element's fma chain is ffn.hip's, results are bit-identical.
bool ffn_wide_supported(int ff);
// the spread form of the projecting tail for <= 64 row groups (ffn_wide.hip): 4 launches over single-wave workgroups, same bits
size_t ffn_spread_scratch_floats(int ff);
bool ffn_spread_supported(int M, int ff);
int launch_proj_ffn_spread(const float* A, int lda, const float* Wo, const float* bo, const float* R, int ldr, const float* pg,
                           const float* pb, const float* W1, const float* b1, const float* W2, const float* b2,
                           const float* ln_g, const float* ln_b, float* OUT, int ldo, int M, int ff, float* scratch,
                           hipStream_t s, const int* M_dev = nullptr, const int* r_idx = nullptr, const float* R2 = nullptr,
                           bool pre = false, float* OUT2 = nullptr, int ldo2 = 0);      // pre: the pre-norm tail (OUT = the stream, OUT2 = LN(OUT; ln_g, ln_b))
int launch_proj_ffn_prenorm_wide(const float* A, int lda, const float* Wo, const float* bo, const float* R, int ldr, const float* pg,
                                 const float* pb, const float* W1, const float* b1, const float* W2, const float* b2, float* OUT,
                                 int ldo, const float* n2g, const float* n2b, float* OUT2, int ldo2, int M, const int* M_dev, int ff,
                                 hipStream_t s, const int* r_idx = nullptr, const float* R2 = nullptr);
// m_off: the rows are rows m_off .. m_off + M of a larger job whose device-side row count is *M_dev
int launch_ffn_wide(const float* X, int ldx, const float* W1, const float* b1, const float* W2, const float* b2,
                    const float* ln_g, const float* ln_b, float* OUT, int ldo, int M, const int* M_dev, int ff, hipStream_t s,
                    int m_off = 0);
int launch_proj_ffn_wide(const float* A, int lda, const float* Wo, const float* bo, const float* R, int ldr, const float* pg,
                         const float* pb, const float* W1, const float* b1, const float* W2, const float* b2,
                         const float* ln_g, const float* ln_b, float* OUT, int ldo, int M, const int* M_dev, int ff,
                         hipStream_t s, const int* r_idx, const float* R2, int m_off = 0);

// ---------------------------------------------------------------- the same on the bf16 matrix cores (ffn_split.hip)
// fp32 products as six partial products of three-piece bf16 operands, fp32 accumulation: fp32-MFMA accuracy (measured),
// 2.7x its rate.  Wimg = the layer's W1 / W2 split and laid out once by launch_ffn_split_pack
// (ffn_split_image_bytes(ff) bytes).
bool ffn_split_supported(int ff);
size_t ffn_split_image_bytes(int ff);
int launch_ffn_split_pack(const float* W1, const float* W2, int ff, void* img, hipStream_t s);
int launch_ffn_split(const float* X, int ldx, const void* Wimg, const float* b1, const float* b2, const float* ln_g,
                     const float* ln_b, float* OUT, int ldo, int M, const int* M_dev, int ff, hipStream_t s);
// C (M, N) = X (M, 256) W^T + bias on the same split operands; Wimg = launch_ffn_split_pack(W, nullptr, N, ...)
bool rows256_split_supported(int N);
size_t rows256_split_image_bytes(int N);
int launch_rows256_split(const float* X, int ldx, const void* Wimg, const float* bias, float* C, int ldc, int M,
                         const int* M_dev, int N, hipStream_t s);
// launch_proj_ffn_fused's computation: Woimg = launch_ffn_split_pack(Wo, nullptr, 256, ...) (ffn_split_proj_image_bytes())
size_t ffn_split_proj_image_bytes();
bool ffn_split_qkv_fits(int ff, int n_qkv);
int launch_proj_ffn_split(const float* A, int lda, const void* Woimg, const float* bo, const float* R, int ldr,
                          const float* pg, const float* pb, const void* Wimg, const float* b1, const float* b2,
                          const float* ln_g, const float* ln_b, float* OUT, int ldo, int M, const int* M_dev, int ff,
                          hipStream_t s, const int* r_idx = nullptr, const float* R2 = nullptr,
                          const void* Qimg = nullptr, const float* qb = nullptr, float* QKV = nullptr, int ldq = 0,
                          int n_qkv = 0);
// Qimg != null: the kernel also writes QKV (M, n_qkv) = OUT Wq^T + qb (the next layer's q | k | v projection; Qimg =
// launch_ffn_split_pack(Wq, nullptr, n_qkv, ...)) from the registers that hold OUT

// ---------------------------------------------------------------- row kernels (rowops.hip)
int launch_layernorm(const float* x, int ldx, const float* g, const float* b, float* out, int ldo,
                     int64_t n_rows, const int* n_rows_dev, int dim, hipStream_t s, const int* src_row = nullptr);
// src_row != null: output row i normalises input row src_row[i]
int launch_l2norm(const float* x, int64_t n_rows, int dim, float eps, float* out, hipStream_t s, int clamp = 0);
// out[m][n] = act(<X[m], W[n]> + b[n]), n < nout <= 2, K = 256; act 0 none, 1 sigmoid
int launch_rowdot(const float* X, int ldx, const float* W, const float* b, float* out, int ldo,
                  int64_t n_rows, int nout, int act, hipStream_t s);

// ---------------------------------------------------------------- attention.hip
// Where the encoder self-attention reads the q | k | v rows of a window's tokens from:
//   ATTN_PACKED : packed (M, ld) matrices Q / K / V (row off[b] + token), q and k already carry the position term;
//   ATTN_GATHER : the layer-0 caches -- clip rows qkv_vid[vrow0[b] + p] (+ pos_qk row of (vlen[b], p) on q | k), text
//                 rows qkv_txt[trow0[b] + j]: the first layer's in_proj is never written per window;
//   ATTN_POSADD : packed Q / K / V = x W^T + b of the layer's input WITHOUT the position term; the kernel adds the
//                 static row pos_qk[(vlen[b], p)] = pos W_qk^T to q | k of clip tokens in its staging loads
//                 ((x + pos) W^T = x W^T + pos W^T): one N = 768 GEMM on x per layer, no x + pos matrix.
enum { ATTN_PACKED = 0, ATTN_GATHER = 1, ATTN_POSADD = 2 };
struct AttnSrc {
    const float* Q; const float* K; const float* V; int ldq, ldk, ldv;
    const float* qkv_vid; const float* qkv_txt; const float* pos_qk;    // pos_qk (R, 512), row lv (lv - 1) / 2 + p
    int pos_zero_row;                             // a row of pos_qk that is all zeros (cone_pos_tables: its last row): what a text token adds
    const int* vrow0; const int* vlen; const int* trow0;
    const float* txt_pos_qk;                      // --use_txt_pos on the table path: (n_txt, 512) rows of THIS layer, token row trow0[b] + j (needs trow0); NULL: the zero row
    int form;                                     // 0: the workgroup-per-(window, head) kernel; 2: one wave per (window, head), K / V in registers (same bits; <= 144 tokens)
};
int launch_enc_attn(int mode, const AttnSrc& src, float* OUT, const int* off, int B, int Lmax, hipStream_t s);
int launch_tile_rows(float* x, int period, int64_t n_rows, hipStream_t s);
int launch_tile_rows2(float* d0, const float* s0, float* d1, const float* s1, int period, int64_t n_rows, hipStream_t s);
// fused decoder cross-attention with the memory K/V projections folded in (dec_cross.hip).  Keys = memory + pos:
// either XP (M, 256) = memory + pos precomputed, or (XP == nullptr) memory rows X plus the static sine rows
// pos_rows[(vlen[b], p)] of clip tokens added in the staging loads.
bool dec_cross_supported(int nq, int Lmax);
int launch_dec_cross(const float* DQ, const float* XP, const float* X, const float* pos_rows, const int* vlen,
                     const int* off, const float* Wk, const float* WvT, const float* bv, float* OUT, int B, int nq,
                     int Lmax, hipStream_t s);
// the same on the matrix cores (dec_cross_mfma.hip): the two 256-channel contractions as v_mfma_f32_16x16x4_f32 tiles
// qk_slabs (dec_cross_mfma_slab_floats() floats of scratch, or null): the queries are the same rows for every window
// (first decoder layer): the folded-key operand is built once instead of per window
size_t dec_cross_mfma_slab_floats();
bool dec_cross_mfma_supported(int nq, int Lmax, bool table);     // slot counts 3 / 8 / 10 as well (table form)
// form 2 (default): windows of at most 128 tokens run dec_cross_x_kernel (rows read ONCE: the registers of stage A transposed
// through LDS by channel quarters for stage C; two workgroups per CU), longer ones the two-read kernel; form 3: always the
// two-read kernel; form 4 (opt-in): windows of at most 110 tokens on the LDS-resident persistent form (dec_cross_res_kernel:
// one workgroup per CU, measured slower)
bool dec_cross_res_supported(int nq, int Lmax);
int launch_dec_cross_mfma(const float* DQ, const float* XP, const float* X, const float* pos_rows, const int* vlen,
                          const int* off, const float* Wk, const float* WvT, const float* bv, float* OUT, int B, int nq,
                          int Lmax, float* qk_slabs, hipStream_t s, int form = 2, const float* sal_w = nullptr,
                          const float* sal_b = nullptr, float* sal = nullptr, int sal_ld = 0);
// sal != null (table form of the two-read kernel only): the launch also writes the saliency head of the window's clip rows,
// sal[b][p] = <memory row p, sal_w> + sal_b[0] for p < vlen[b] (sal_ld floats per window; other entries untouched)
int launch_small_attn(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* OUT,
                      int ldo, const int* off, int B, int nq, int Lmax, hipStream_t s);

// ---------------------------------------------------------------- window_ops.hip
int launch_scan_lengths(const int* vlen, const int* qlen /* may be null */, int B, int* off, hipStream_t s);
int launch_compact_index(const int* vlen, const int* voff, int Lv_pad, int* vidx, const int* qlen, const int* toff, int Lq_pad,
                         int* tidx, int B, hipStream_t s);
int launch_pack_pos(const float* vproj, const int* vrow0, const int* vlen, const float* tproj, const int* trow0,
                    const int* qlen, const int* off, const float* dim_t, float* X, float* POS, float* XP, int B,
                    int Lmax, hipStream_t s, const float* tpe = nullptr, const float* tpg = nullptr,
                    const float* tpb = nullptr);        // tpe != null: --use_txt_pos (embedding rows + its LayerNorm)
int launch_pos_rows(const float* dim_t, int max_v_l, float* out, hipStream_t s);
int launch_add_pos_rows(const float* MEM, const int* off, const int* vlen, const float* pos_rows, float* XP, int B, int Lmax,
                        hipStream_t s, const float* txt_pos = nullptr, const int* trow0 = nullptr);
int launch_txt_pos_rows(const float* tproj, const int* tok_index, const int* src_row, int mod, int n_emb, const float* tpe,
                        const float* tpg, const float* tpb, int n, const int* n_dev, float* out, hipStream_t s);     // XP = MEM + table row of (vlen[b], p) for clip tokens (the unfolded decoder's keys)
// X always; POS (sine rows) and QK / V (layer-0 q|k|v gathered from the caches) only when non-null
int launch_row_index(const int* vrow0, const int* vlen, const int* trow0, const int* qlen, const int* off, int* ridx,
                     int B, int Lmax, hipStream_t s);
int launch_pack_l0(const float* vproj, const int* vrow0, const int* vlen, const float* tproj, const int* trow0,
                   const int* qlen, const int* off, const float* dim_t, const float* qkv_vid, const float* qkv_txt,
                   const float* pos_qk, float* X, float* POS, float* QK, float* V, int B, int Lmax, hipStream_t s);
int launch_saliency(const float* MEM, const int* off, const int* vlen, const int* qlen, const float* w,
                    const float* bias, float* sal, int Lv_out, float* mem_tap, int Lq_out, int B, hipStream_t s);
int launch_proposal_mean(const float* vid, const int* vrow0, const int* vlen, const int* pad_len,
                         const float* spans, int B, int Nq, int dv, float* out, hipStream_t s);
int launch_cosine_match(const float* pf, const float* cls, const int* cls_row, int B, int Nq, int dv, float* match,
                        hipStream_t s);

}  // namespace cone
