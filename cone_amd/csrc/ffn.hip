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
// on the SOURCE address and on the ds_read_b128 address (conflict-free for lane = (row, chunk)) -- into a 4-stage
// ring (128 KiB): chunk c+2 streams in while chunk c is multiplied, one counted s_waitcnt vmcnt + raw s_barrier per
// chunk (128 MFMAs per wave).  b1 is copied to LDS once so that no ordinary global load sits inside the loop (hipcc
// would drain the DMA queue for it, cdna_hip_programming.md section 5 "Three .s-level traps" (b)).
//
// One PERSISTENT 8-wave workgroup per CU (132 KiB of LDS, <= 256 VGPRs) walks the 128-row tiles: two waves per SIMD
// keep the matrix pipe fed; GEMM1 of chunk i and GEMM2 of chunk i-1 share the units of one iteration (software
// pipelining across chunks), so bias + ReLU never stall a chunk; the epilogue is bias + residual + LayerNorm in
// registers (a token's 256 channels sit in 4 lanes x 64 registers, two shuffle steps per moment).
#include <mutex>

#include "common.h"

namespace cone {

typedef float f32x4f __attribute__((ext_vector_type(4)));

#define FFN_GLDS16(src, dst) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

__device__ __forceinline__ int ffn_swz16(int row) { return (0x1230 >> (((row >> 2) & 3) * 4)) & 3; }

constexpr int FFN_STAGE = 2 * 16 * 256;           // floats per ring stage: W1 image (16 slabs) + W2 image (16 slabs)
constexpr int FFN_NST = 4;                        // ring depth: a stage is refilled two barriers after its last read
// 1-KiB LDS-DMA pieces per wave per chunk: 32 pieces / NW waves (NW = 8: 4, NW = 4: 8)

struct FfnArgs {
    const float* X; int ldx;                      // (M, 256) block input = residual of the feed-forward block
    const float* W1; const float* b1;             // (ff, 256), (ff)
    const float* W2; const float* b2;             // (256, ff), (256)
    const float* ln_g; const float* ln_b;         // (256)
    float* OUT; int ldo;                          // (M, 256)
    int M; const int* M_dev;                      // rows; *M_dev wins when non-null (grid sized by M)
    int ff;
    // PROJ: the block input is itself  LayerNorm(R + A Wo^T + bo)  (attention output projection + residual + norm,
    // cone/transformer.py:239-241, 308-312), computed here instead of being read: A (M, 256) attention output,
    // R (M, 256) residual; X is unused.
    const float* A; int lda; const float* R; int ldr;
    const float* Wo; const float* bo; const float* pg; const float* pb;
    // r_idx != null: the residual rows are gathered: row i = R[r_idx[i]] (r_idx[i] >= 0) or R2[~r_idx[i]]
    const int* r_idx; const float* R2;
    // QKV: the NEXT layer's q | k | v projection of the rows this kernel produces (Wq (n_qkv, 256), qb), computed from the
    // registers that hold them and written to QKV (M, n_qkv): no second pass over the rows, no extra launch
    const float* Wq; const float* qb; float* QKV; int ldq; int n_qkv;
    // PRE (pre-norm layers, cone/transformer.py:248-260 / 319-342; PROJ only): OUT = x1 + W2 relu(W1 LN_p(x1) + b1) + b2 with
    // x1 = R + A Wo^T + bo -- the residual stream stays un-normalised, (pg, pb) is the norm AHEAD of the feed-forward block --
    // and OUT2 (may be null) = LayerNorm(OUT; ln_g, ln_b): what the next consumer reads (the next layer's norm1, the
    // encoder's / decoder's final norm)
    float* OUT2; int ldo2;
};

// LayerNorm over a token's 256 channels held as v[16] (channel 16 t + 4 lg + r in v[t][r]): 4 lanes x 64 registers.
__device__ __forceinline__ void ffn_layernorm_regs(f32x4f (&v)[16], float& rstd) {
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

// NW = waves per workgroup = 16-row groups per tile.  NW = 8 (128-row tiles, two waves per SIMD) is the throughput form.
// NW = 4 (64-row tiles, ONE wave per SIMD) is the small-M form: a wave runs the very same instruction sequence on its 16
// rows (bit-identical results), but has its SIMD's matrix pipe to itself -- a tile takes half the time -- and the tile
// grid is twice as fine; chosen by the launcher when the 128-row tiles would leave at least half of the CUs idle.
template <bool PROJ, bool QKV, int NW, bool PRE = false>
__global__ __launch_bounds__(64 * NW, NW / 4) void ffn_fused_kernel(FfnArgs p) {
    static_assert(!PRE || (PROJ && !QKV), "the pre-norm form is the layer tail");
    constexpr int FFN_ROWS = 16 * NW, FFN_NPIECE = 32 / NW, NT = 64 * NW, HW = NW / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* b1s = smem + FFN_NST * FFN_STAGE;
    int M = p.M;
    if (p.M_dev) { const int md = *p.M_dev; M = md < M ? md : M; }
    // Work items of the persistent loop: full tiles of FFN_ROWS rows, walked round-robin over the grid -- except that the
    // tiles of the LAST round, when they would occupy at most half of the workgroups, are cut into half tiles of 64 rows
    // (waves 0-3; waves 4-7 idle through the item): twice as many CUs finish the ragged end of a launch in 0.18 ms instead of
    // 0.30 (one wave per SIMD has the matrix pipe to itself).  The per-wave instruction sequence on a row group does not
    // change: results are bit-identical.  `act` (wave-uniform) = this wave holds rows of the item; an idle wave takes a
    // short side path (its share of the weight stream + every barrier, no MFMA) instead of the tile body.
    const int n_tiles = (M + FFN_ROWS - 1) / FFN_ROWS;
    if (n_tiles == 0) return;
    const int G_ = (int)gridDim.x;
    const int full_items = ((n_tiles - 1) / G_) * G_;                          // tiles of the complete rounds
    const int rem_rows = M - full_items * FFN_ROWS;                             // rows of the last round (> 0)
    const int rem_half = (rem_rows + 63) >> 6;
    const bool half_mode = NW == 8 && rem_half <= G_ && 2 * ((rem_rows + FFN_ROWS - 1) / FFN_ROWS) <= G_;
    const int n_items = full_items + (half_mode ? rem_half : (rem_rows + FFN_ROWS - 1) / FFN_ROWS);
    if ((int)blockIdx.x >= n_items) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // first row of this wave's 16-row group in work item `it` (M or more: the wave idles)
    auto wave_row0 = [&](int it) -> int {
        if (it < full_items || !half_mode) return it * FFN_ROWS + wave * 16;
        return wave < 4 ? full_items * FFN_ROWS + (it - full_items) * 64 + wave * 16 : M;
    };
    const int li = lane & 15, lg = lane >> 4;
    const int ff = p.ff, nc = ff >> 4;
    constexpr int NP = PROJ ? 8 : 0;              // leading chunks of the output projection (32 channels each)
    const int NQ = QKV ? p.n_qkv >> 5 : 0;        // trailing chunks of the next layer's q | k | v projection (32 channels each)
    const int G = NP + nc + NQ;                   // weight chunks per tile; the ring runs on across tiles

    // every per-channel parameter vector goes to LDS once per workgroup: b1 (ff), then 6 x 256: b2, ln_g, ln_b and
    // (PROJ) bo, pg, pb -- no ordinary global load remains inside the tile loop besides the tile's own rows
    float* prm = b1s + ff;
    for (int i = tid; i < (ff >> 2); i += NT)
        reinterpret_cast<f32x4f*>(b1s)[i] = reinterpret_cast<const f32x4f*>(p.b1)[i];
    if (tid < 64) {
        reinterpret_cast<f32x4f*>(prm)[tid] = reinterpret_cast<const f32x4f*>(p.b2)[tid];
        reinterpret_cast<f32x4f*>(prm + 256)[tid] = reinterpret_cast<const f32x4f*>(p.ln_g)[tid];
        reinterpret_cast<f32x4f*>(prm + 512)[tid] = reinterpret_cast<const f32x4f*>(p.ln_b)[tid];
        if (PROJ) {
            reinterpret_cast<f32x4f*>(prm + 768)[tid] = reinterpret_cast<const f32x4f*>(p.bo)[tid];
            reinterpret_cast<f32x4f*>(prm + 1024)[tid] = reinterpret_cast<const f32x4f*>(p.pg)[tid];
            reinterpret_cast<f32x4f*>(prm + 1280)[tid] = reinterpret_cast<const f32x4f*>(p.pb)[tid];
        }
    }
    float* qbs = prm + 1536;
    if (QKV)
        for (int i = tid; i < (p.n_qkv >> 2); i += NT)
            reinterpret_cast<f32x4f*>(qbs)[i] = reinterpret_cast<const f32x4f*>(p.qb)[i];

    // ---- LDS-DMA pieces.  A feed-forward chunk = 32 slabs of [16 rows][16 floats]; wave w issues slabs 4w .. 4w+3
    // (0-15 = W1, 16-31 = W2); lane -> (row = lane / 4, physical chunk = lane % 4), the source chunk is XOR-swizzled.
    // A projection chunk = 32 slabs too: Wo rows [32 g, 32 g + 16) in the first half of the stage, rows [32 g + 16,
    // 32 g + 32) in the second half, both in the W1 slab format -- every chunk of the kernel costs the same 4 pieces
    // per wave and one counted s_waitcnt serves all of it.
    // Addressing: all four pieces of a wave lie on one side (waves 0-3: W1 slabs 4w .. 4w+3, waves 4-7: W2 slabs), so a
    // piece's source is a wave-uniform base (SGPRs) + ONE per-lane offset + scalar multiples of the piece / chunk index.
    const int drow = lane >> 2;
    const int dch = (lane & 3) ^ ffn_swz16(drow);
    const bool w2side = wave >= HW;
    const float* fbase = w2side ? p.W2 : p.W1;                         // wave-uniform
    const int foff = w2side ? (16 * FFN_NPIECE * (wave - HW) + drow) * ff + 4 * dch   // W2[16 t + row][h0 + 4 ch ..], t = NPIECE (w - HW) + i
                            : drow * 256 + 16 * FFN_NPIECE * wave + 4 * dch;         // W1[h0 + row][16 q + 4 ch ..],  q = NPIECE w + i
    const int fpiece = w2side ? 16 * ff : 16;                          // + i * fpiece
    const int fchunk = w2side ? 16 : 16 * 256;                         // + c * fchunk
    // projection chunk g: Wo[32 g + 16 (pid / 16) + row][16 (pid % 16) + 4 ch ..], pid = 4 w + i
    const int poff = PROJ ? (16 * (wave / HW) + drow) * 256 + 16 * FFN_NPIECE * (wave % HW) + 4 * dch : 0;
    // piece i of chunk g of the current tile (projection chunks first, then the feed-forward chunks); g >= G = the
    // first chunks of the NEXT tile this workgroup will run -- the same weights, so the ring simply runs on (after
    // the last tile they land in stages nobody reads again).  sb = (chunks consumed before this tile) % FFN_NST.
    int sb = 0;
    auto stream_piece = [&](int g, int i) {
        float* dstp = smem + ((sb + g) % FFN_NST) * FFN_STAGE + (wave * FFN_NPIECE + i) * 256;
        const int gg = g < G ? g : g - G;
        // wave-uniform pointer + zero-extended 32-bit lane offset: the saddr + voffset form of global_load_lds (one
        // address VGPR for all pieces instead of a hoisted 64-bit VGPR pair per piece)
        // ONE scalar base (selected, not branched on) + ONE 32-bit lane offset; the empty asm pins the base in SGPRs at
        // this point: hoisted out of the tile loop, base + lane offset becomes a 64-bit VGPR pair per piece, spilled,
        // and every reload drains the DMA queue with a vmcnt(0)
        const bool qk = QKV && gg >= NP + nc;         // a q | k | v chunk: 32 rows of Wq in the projection's slab format
        const bool pj = (PROJ && gg < NP) || qk;
        const char* ub = reinterpret_cast<const char*>(
            qk ? p.Wq + ((size_t)(gg - NP - nc) * (32 * 256) + 16 * i)
               : (pj ? p.Wo + ((size_t)gg * (32 * 256) + 16 * i) : fbase + ((size_t)(gg - NP) * fchunk + (size_t)i * fpiece)));
        const unsigned vo = (unsigned)((pj ? poff : foff) * 4);
        asm volatile("" : "+s"(ub));
        FFN_GLDS16(ub + vo, dstp);
    };

    const int rd = li * 16 + ((lg ^ ffn_swz16(li)) << 2);          // this lane's 16-B chunk inside a slab
#define FFN_RD(stg, off) (*reinterpret_cast<const f32x4f*>((stg) + (off) + rd))
#define FFN_STAGE_OF(g) (smem + ((sb + (g)) % FFN_NST) * FFN_STAGE)
#define FFN_SB() __builtin_amdgcn_sched_barrier(0)
    // unit u of a chunk (8 units) issues the wave's share of the next-but-one chunk: every other unit at 4 pieces per wave,
    // every unit at 8
#define FFN_STREAM(g, u) { if (FFN_NPIECE == 8) stream_piece(g, u); else if ((u) & 1) stream_piece(g, (u) >> 1); }
    // a counted wait + barrier ends every chunk: the next chunk has landed (all but the 4 pieces issued last, which
    // belong to the chunk after it), and every wave is done reading the stage that the next pieces will overwrite
#define FFN_END_CHUNK()                                                           \
    {                                                                             \
        FFN_SB();                                                                 \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FFN_NPIECE) : "memory");         \
        __builtin_amdgcn_s_barrier();                                             \
        FFN_SB();                                                                 \
    }
    // MFMA issue order is pinned (a sched_barrier after every MFMA): hipcc otherwise regroups the MFMAs of a unit by
    // accumulator, and back-to-back MFMAs on one accumulator wait out the 40-cycle dependent latency of the 32-cycle
    // v_mfma_f32_16x16x4_f32 (measured: 139 -> 131 TFLOP/s).  In every sequence below an accumulator is reused at the
    // earliest two MFMAs (64 cycles) later.
#define FFN_MFMA(acc, a, b) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0); FFN_SB(); }
    // 8 MFMAs of a weight-slab pair against a register tile: four independent partial chains acc[0..3]
#define FFN_MM_A(acc, w0, w1, B, q)                                                   \
    {                                                                                 \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) FFN_MFMA(acc[r], w0[r], B[q][r])         \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) FFN_MFMA(acc[r], w1[r], B[(q) + 1][r])   \
    }
    // 8 MFMAs of GEMM2: two output tiles, alternating accumulators
#define FFN_MM_Y(w0, w1, hh, t)                                                       \
    {                                                                                 \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                               \
            FFN_MFMA(y[t], w0[r], hh[r])                                              \
            FFN_MFMA(y[(t) + 1], w1[r], hh[r])                                        \
        }                                                                             \
    }
    // 16 MFMAs: GEMM1 slab pair and GEMM2 tile pair interleaved (accumulator reuse distance 4)
#define FFN_MM_AY(acc, w0, w1, B, q, v0, v1, hh, t)                                   \
    {                                                                                 \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                               \
            FFN_MFMA(acc[r], w0[r], B[q][r])                                          \
            FFN_MFMA(y[(t) + (r & 1)], ((r & 1) ? v1 : v0)[r >> 1], hh[r >> 1])       \
        }                                                                             \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                               \
            FFN_MFMA(acc[r], w1[r], B[(q) + 1][r])                                    \
            FFN_MFMA(y[(t) + (r & 1)], ((r & 1) ? v1 : v0)[2 + (r >> 1)], hh[2 + (r >> 1)]) \
        }                                                                             \
    }

#pragma unroll
    for (int i = 0; i < FFN_NPIECE; ++i) stream_piece(0, i);
#pragma unroll
    for (int i = 0; i < FFN_NPIECE; ++i) stream_piece(1 < G ? 1 : 0, i);

    // Schedule.  Every chunk is walked as 8 units of 16 MFMAs (two slabs of each half of the stage).  The ds_read_b128
    // of unit u+1 are issued right ahead of the MFMAs of unit u and consumed (the empty asm at the top of unit u+1 is
    // where hipcc puts its lgkmcnt wait -- BEFORE the next reads are issued, so it never covers a read that was only
    // just issued) after them: LDS latency hides under 512 matrix-pipe cycles.  Loop bodies are straight-line code: the
    // ring is refilled unconditionally (the last iterations re-stream the final chunk into stages nobody reads again).
    f32x4f wa, wb, na, nb;      // W1-format fragments (current / next unit)
    f32x4f va, vb, nva, nvb;    // second-half fragments

    // Persistent workgroup: one per CU, walking tiles blockIdx.x, + gridDim.x, ...  A tile's fixed costs -- its
    // input rows (16 KiB per wave, 64-B row segments per load), the output stores, the dispatch of a fresh workgroup --
    // are not covered by another workgroup at this occupancy (measured 18 us per 128-row tile against 3.7 us per
    // chunk = 7 % at ff = 1024); here the next tile's rows are requested while the previous tile's stores drain and
    // the weight ring never restarts.
    // register tiles: v[q][r] = row[token li][16 q + 4 lg + r], q = 0 .. 15.  xr = block input: B operand of GEMM1 and
    // the residual of the block (PROJ: starts as the residual of the projection, becomes the block input); ar (PROJ) =
    // attention rows, B operand of the projection.  The NEXT tile's rows are requested from the epilogue, ahead of
    // this tile's stores (vmcnt retires in order: loads queued behind stores would wait for the stores' acks).
    f32x4f xr[16];
    f32x4f ar[PROJ ? 16 : 1];
    auto load_tile = [&](int tile) {
        const int row = wave_row0(tile) + li;
        const size_t ld_row = (size_t)(row < M ? row : M - 1);         // rows past M feed unstored outputs
        if (PROJ) {     // the attention rows; the residual rows follow at the top of the tile (load_res)
            const float* ap = p.A + ld_row * p.lda + 4 * lg;
#pragma unroll
            for (int q = 0; q < 16; ++q) ar[q] = *reinterpret_cast<const f32x4f*>(ap + 16 * q);
        } else {
            const float* xp = p.X + ld_row * p.ldx + 4 * lg;
#pragma unroll
            for (int q = 0; q < 16; ++q) xr[q] = *reinterpret_cast<const f32x4f*>(xp + 16 * q);
        }
    };
    // PROJ: the residual rows of a tile are requested at its top and consumed at the end of each projection chunk (they
    // land under the first chunk's MFMAs): at no point are the next tile's rows, this tile's outputs and both input
    // tiles live together
    auto load_res = [&](int tile) {
        const int row = wave_row0(tile) + li;
        const size_t ld_row = (size_t)(row < M ? row : M - 1);
        const float* rp = p.R + ld_row * p.ldr + 4 * lg;
        if (p.r_idx) {
            const int ix = p.r_idx[ld_row];
            rp = (ix >= 0 ? p.R + (size_t)ix * p.ldr : p.R2 + (size_t)(~ix) * p.ldr) + 4 * lg;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) xr[q] = *reinterpret_cast<const f32x4f*>(rp + 16 * q);
    };
    load_tile(blockIdx.x);
    bool first = true;
    for (int tile = blockIdx.x; tile < n_items; tile += gridDim.x) {
    const int my_row0 = wave_row0(tile);
    const int my_row = my_row0 + li;
    const bool act = my_row0 < M;                   // wave-uniform
    if (first) {        // chunk 0 and the b1 image are in LDS (later tiles: the previous tile's last barrier covers chunk 0)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FFN_NPIECE) : "memory");
        __syncthreads();
        first = false;
    }
    if (!act) {
        // this wave holds no row of the item (the upper waves of a half tile, the waves past a launch's last row): it keeps
        // its part of the protocol -- per chunk its pieces of the next-but-one chunk, the counted wait, the barrier: G of
        // each per item, exactly what a computing wave does -- and issues no MFMA, so that the waves that do compute have
        // the matrix pipes to themselves.  Only the last item of a workgroup can be short.
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int i = 0; i < FFN_NPIECE; ++i) stream_piece(g + 2, i);
            FFN_END_CHUNK()
        }
        break;
    }

    f32x4f y[16];
    if (PROJ) {
        load_res(tile);
        // ---- attention output projection + residual + LayerNorm: pair g computes channels [32 g, 32 g + 32) of
        // A Wo^T and adds them to xr[2 g], xr[2 g + 1] (the residual rows); fully unrolled: xr[] is indexed statically
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            const float* st = FFN_STAGE_OF(g);
            f32x4f ha[2], hb[2];        // two partial chains per tile: with both tiles interleaved, four MFMAs apart
            ha[0] = f32x4f{0.f, 0.f, 0.f, 0.f}; hb[0] = ha[0];
            ha[1] = ha[0]; hb[1] = ha[0];
            wa = FFN_RD(st, 0); wb = FFN_RD(st, 256); va = FFN_RD(st, 4096); vb = FFN_RD(st, 4096 + 256);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                FFN_SB();
                asm volatile("" : "+v"(wa), "+v"(wb), "+v"(va), "+v"(vb));
                FFN_SB();
                if (u < 7) {
                    na = FFN_RD(st, (2 * u + 2) * 256); nb = FFN_RD(st, (2 * u + 3) * 256);
                    nva = FFN_RD(st, 4096 + (2 * u + 2) * 256); nvb = FFN_RD(st, 4096 + (2 * u + 3) * 256);
                }
                FFN_SB();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    FFN_MFMA(ha[r & 1], wa[r], ar[2 * u][r])
                    FFN_MFMA(hb[r & 1], va[r], ar[2 * u][r])
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    FFN_MFMA(ha[r & 1], wb[r], ar[2 * u + 1][r])
                    FFN_MFMA(hb[r & 1], vb[r], ar[2 * u + 1][r])
                }
                FFN_STREAM(g + 2, u)                               // the next-but-one chunk's pieces ride behind MFMAs
                if (u < 7) { wa = na; wb = nb; va = nva; vb = nvb; }
            }
            xr[2 * g] += ha[0] + ha[1];
            xr[2 * g + 1] += hb[0] + hb[1];
            FFN_END_CHUNK()
        }
        FFN_SB();
#pragma unroll
        for (int t = 0; t < 16; ++t) xr[t] += *reinterpret_cast<const f32x4f*>(prm + 768 + 16 * t + 4 * lg);       // + bo
        if (PRE) {      // the un-normalised stream x1 is the block's residual: the output accumulators START from x1 + b2
                        // (no second copy of the tile in registers), then xr becomes the normalised block input
#pragma unroll
            for (int t = 0; t < 16; ++t) y[t] = xr[t] + *reinterpret_cast<const f32x4f*>(prm + 16 * t + 4 * lg);
        }
        float rstd;
        ffn_layernorm_regs(xr, rstd);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f32x4f g4 = *reinterpret_cast<const f32x4f*>(prm + 1024 + 16 * t + 4 * lg);
            const f32x4f b4 = *reinterpret_cast<const f32x4f*>(prm + 1280 + 16 * t + 4 * lg);
#pragma unroll
            for (int r = 0; r < 4; ++r) xr[t][r] = xr[t][r] * rstd * g4[r] + b4[r];
        }
        FFN_SB();
    }

    // ---- feed-forward block, software-pipelined across chunks: iteration i multiplies GEMM1 of chunk i (W1 half of
    // stage i) and GEMM2 of chunk i-1 (W2 half of stage i-1, hidden tile h of the previous iteration) in the same
    // units, so the GEMM1 -> bias/ReLU -> GEMM2 dependency spans a whole iteration instead of stalling every chunk.
    if (!PRE) {
#pragma unroll
        for (int t = 0; t < 16; ++t) y[t] = f32x4f{0.f, 0.f, 0.f, 0.f};
    }
    f32x4f h, hp[4];
    {   // iteration 0: GEMM1 of chunk 0 alone
        const float* st = FFN_STAGE_OF(NP);
#pragma unroll
        for (int r = 0; r < 4; ++r) hp[r] = f32x4f{0.f, 0.f, 0.f, 0.f};
        wa = FFN_RD(st, 0); wb = FFN_RD(st, 256);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            FFN_SB();
            asm volatile("" : "+v"(wa), "+v"(wb));
            FFN_SB();
            if (u < 7) { na = FFN_RD(st, (2 * u + 2) * 256); nb = FFN_RD(st, (2 * u + 3) * 256); }
            else { va = FFN_RD(st, 4096); vb = FFN_RD(st, 4096 + 256); }       // unit 0 of GEMM2(0), next iteration
            FFN_SB();
            FFN_MM_A(hp, wa, wb, xr, 2 * u)
            FFN_STREAM(NP + 2, u)
            if (u < 7) { wa = na; wb = nb; }
        }
        FFN_SB();
        h = (hp[0] + hp[1]) + (hp[2] + hp[3]) + *reinterpret_cast<const f32x4f*>(b1s + 4 * lg);
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = fmaxf(h[r], 0.f);
        FFN_END_CHUNK()
    }
    for (int i = 1; i < nc; ++i) {
        const float* s1 = FFN_STAGE_OF(NP + i);             // W1 half: chunk i
        const float* s2 = FFN_STAGE_OF(NP + i - 1);         // W2 half: chunk i - 1
#pragma unroll
        for (int r = 0; r < 4; ++r) hp[r] = f32x4f{0.f, 0.f, 0.f, 0.f};
        wa = FFN_RD(s1, 0); wb = FFN_RD(s1, 256);           // only now visible (the barrier above); GEMM2's first
#pragma unroll                                              // fragments were requested before it
        for (int u = 0; u < 8; ++u) {
            FFN_SB();
            if (u == 0) {
                // the first unit runs GEMM2 ahead of GEMM1: its fragments are already here, GEMM1's are in flight
                asm volatile("" : "+v"(va), "+v"(vb));
                FFN_SB();
                nva = FFN_RD(s2, 4096 + 2 * 256); nvb = FFN_RD(s2, 4096 + 3 * 256);
                FFN_SB();
                FFN_MM_Y(va, vb, h, 0)
                FFN_SB();
                asm volatile("" : "+v"(wa), "+v"(wb));
                FFN_SB();
                na = FFN_RD(s1, 2 * 256); nb = FFN_RD(s1, 3 * 256);
                FFN_SB();
                FFN_MM_A(hp, wa, wb, xr, 0)
            } else {
                asm volatile("" : "+v"(wa), "+v"(wb), "+v"(va), "+v"(vb));
                FFN_SB();
                if (u < 7) {
                    na = FFN_RD(s1, (2 * u + 2) * 256); nb = FFN_RD(s1, (2 * u + 3) * 256);
                    nva = FFN_RD(s2, 4096 + (2 * u + 2) * 256); nvb = FFN_RD(s2, 4096 + (2 * u + 3) * 256);
                } else {    // GEMM2's first fragments of the next iteration: W2 half of THIS chunk's stage
                    nva = FFN_RD(s1, 4096); nvb = FFN_RD(s1, 4096 + 256);
                }
                FFN_SB();
                FFN_MM_AY(hp, wa, wb, xr, 2 * u, va, vb, h, 2 * u)
            }
            FFN_STREAM(NP + i + 2, u)
            if (u < 7) { wa = na; wb = nb; }
            va = nva; vb = nvb;
        }
        FFN_SB();
        // hidden tile of chunk i for the next iteration (the last GEMM1 MFMAs finished under the last GEMM2 ones)
        h = (hp[0] + hp[1]) + (hp[2] + hp[3]) + *reinterpret_cast<const f32x4f*>(b1s + 16 * i + 4 * lg);
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = fmaxf(h[r], 0.f);
        FFN_END_CHUNK()
    }
    {   // last iteration: GEMM2 of chunk nc - 1 alone
        const float* st = FFN_STAGE_OF(NP + nc - 1);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            FFN_SB();
            asm volatile("" : "+v"(va), "+v"(vb));
            FFN_SB();
            if (u < 7) { nva = FFN_RD(st, 4096 + (2 * u + 2) * 256); nvb = FFN_RD(st, 4096 + (2 * u + 3) * 256); }
            FFN_SB();
            FFN_MM_Y(va, vb, h, 2 * u)
            if (u < 7) { va = nva; vb = nvb; }
        }
    }
    // ---- epilogue: + b2 + residual, LayerNorm over the token's 256 channels (4 lanes x 64 registers), store
    FFN_SB();
    if (!PRE) {
#pragma unroll
        for (int t = 0; t < 16; ++t) y[t] = y[t] + *reinterpret_cast<const f32x4f*>(prm + 16 * t + 4 * lg) + xr[t];
    }
    FFN_SB();
    // in flight under the LayerNorm + stores (unconditional -- after the last tile a valid tile is simply re-read --
    // so that the register tiles have ONE definition per iteration)
    load_tile(tile + (int)gridDim.x < n_items ? tile + (int)gridDim.x : tile);
    FFN_SB();
    if (PRE && my_row < M) {        // the un-normalised stream
        float* op = p.OUT + (size_t)my_row * p.ldo + 4 * lg;
#pragma unroll
        for (int t = 0; t < 16; ++t) *reinterpret_cast<f32x4f*>(op + 16 * t) = y[t];
    }
    float rstd;
    if (!PRE || p.OUT2) ffn_layernorm_regs(y, rstd);
    if (PRE) {
        if (p.OUT2 && my_row < M) {
            float* op = p.OUT2 + (size_t)my_row * p.ldo2 + 4 * lg;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const f32x4f g = *reinterpret_cast<const f32x4f*>(prm + 256 + 16 * t + 4 * lg);
                const f32x4f be = *reinterpret_cast<const f32x4f*>(prm + 512 + 16 * t + 4 * lg);
                f32x4f o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = y[t][r] * rstd * g[r] + be[r];
                *reinterpret_cast<f32x4f*>(op + 16 * t) = o;
            }
        }
    } else if (my_row < M) {
        float* op = p.OUT + (size_t)my_row * p.ldo + 4 * lg;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f32x4f g = *reinterpret_cast<const f32x4f*>(prm + 256 + 16 * t + 4 * lg);
            const f32x4f be = *reinterpret_cast<const f32x4f*>(prm + 512 + 16 * t + 4 * lg);
            f32x4f o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = y[t][r] * rstd * g[r] + be[r];
            *reinterpret_cast<f32x4f*>(op + 16 * t) = o;
            if (QKV) y[t] = o;
        }
    } else if (QKV) {       // rows past M feed unstored outputs: any finite values
#pragma unroll
        for (int t = 0; t < 16; ++t) y[t] = f32x4f{0.f, 0.f, 0.f, 0.f};
    }
    if (QKV) {
        // ---- the next layer's q | k | v projection of the tile's output rows (still in registers, in the B-operand
        // layout): NQ chunks of 32 output channels in the projection phase's format, stored from the accumulators
        float* qrow = p.QKV + (size_t)my_row * p.ldq + 4 * lg;
        for (int g = 0; g < NQ; ++g) {
            const float* st = FFN_STAGE_OF(NP + nc + g);
            f32x4f ha[2], hb[2];
            ha[0] = f32x4f{0.f, 0.f, 0.f, 0.f}; hb[0] = ha[0];
            ha[1] = ha[0]; hb[1] = ha[0];
            wa = FFN_RD(st, 0); wb = FFN_RD(st, 256); va = FFN_RD(st, 4096); vb = FFN_RD(st, 4096 + 256);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                FFN_SB();
                asm volatile("" : "+v"(wa), "+v"(wb), "+v"(va), "+v"(vb));
                FFN_SB();
                if (u < 7) {
                    na = FFN_RD(st, (2 * u + 2) * 256); nb = FFN_RD(st, (2 * u + 3) * 256);
                    nva = FFN_RD(st, 4096 + (2 * u + 2) * 256); nvb = FFN_RD(st, 4096 + (2 * u + 3) * 256);
                }
                FFN_SB();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    FFN_MFMA(ha[r & 1], wa[r], y[2 * u][r])
                    FFN_MFMA(hb[r & 1], va[r], y[2 * u][r])
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    FFN_MFMA(ha[r & 1], wb[r], y[2 * u + 1][r])
                    FFN_MFMA(hb[r & 1], vb[r], y[2 * u + 1][r])
                }
                FFN_STREAM(NP + nc + g + 2, u)
                if (u < 7) { wa = na; wb = nb; va = nva; vb = nvb; }
            }
            FFN_SB();
            if (my_row < M) {
                *reinterpret_cast<f32x4f*>(qrow + 32 * g) = (ha[0] + ha[1]) + *reinterpret_cast<const f32x4f*>(qbs + 32 * g + 4 * lg);
                *reinterpret_cast<f32x4f*>(qrow + 32 * g + 16) = (hb[0] + hb[1]) + *reinterpret_cast<const f32x4f*>(qbs + 32 * g + 16 + 4 * lg);
            }
            FFN_END_CHUNK()
        }
    }
    sb = (sb + G) % FFN_NST;
    }   // tile loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // no LDS-DMA may outlive the workgroup's LDS
}

#undef FFN_MM_A
#undef FFN_MM_Y
#undef FFN_MM_AY
#undef FFN_MFMA
#undef FFN_END_CHUNK
#undef FFN_SB
#undef FFN_STREAM
#undef FFN_STAGE_OF
#undef FFN_RD

bool ffn_fused_supported(int ff) { return ff >= 32 && ff % 16 == 0 && ff <= 4096; }

// Launches of at most this many 16-row groups (host bound) take the wide form (ffn_wide.hip: one workgroup per group and
// CU, the eight waves split the output elements): 45 - 60 us per round of 256 groups against 160 - 200 us of a wave's
// serial pass over the weights.  tools/tail_wide_bench.py, out_proj + LN + FFN: 52 / 62 / 124 / 183 us at 16 / 4 096 /
// 8 192 / 12 288 rows against 168 / 181 / 187 / 194 us; from the fourth round on (12 500 rows: 226 us) the row forms win.
#ifndef CONE_FFN_WIDE_GROUPS
#define CONE_FFN_WIDE_GROUPS 768
#endif
constexpr int FFN_WIDE_GROUPS = CONE_FFN_WIDE_GROUPS;
bool ffn_fused_qkv_fits(int ff, int n_qkv) {
    return ffn_fused_supported(ff) && n_qkv >= 32 && n_qkv % 32 == 0 &&
           (size_t)(FFN_NST * FFN_STAGE + ff + 6 * 256 + n_qkv) * sizeof(float) <= 160 * 1024;
}

template <bool PROJ, bool QKV, int NW, bool PRE = false>
static int launch_ffn_nw(const FfnArgs& a, hipStream_t s, int* n_cu_out) {
    const size_t lds = (size_t)(FFN_NST * FFN_STAGE + a.ff + 6 * 256 + (QKV ? a.n_qkv : 0)) * sizeof(float);
    CONE_REQUIRE(lds <= 160 * 1024, "fused layer tail: %zu bytes of LDS (ff %d, q|k|v %d) exceed 160 KiB", lds, a.ff, a.n_qkv);
    // once per device: the opt-in to > 64 KiB of LDS (a property of the code object) and the CU count that sizes
    // the persistent grid (one workgroup per CU: 132 KiB of LDS, 64 NW threads at <= 256 VGPRs)
    static DeviceOnce once;
    int n_cu = 0;
    CONE_CHECK_HIP(device_once(once, [] {
        return hipFuncSetAttribute((const void*)ffn_fused_kernel<PROJ, QKV, NW, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024);
    }, &n_cu));
    if (n_cu_out) { *n_cu_out = n_cu; return 0; }
    const int tiles = (a.M + 16 * NW - 1) / (16 * NW);
    const int grid = tiles < n_cu ? tiles : n_cu;
    // FLOPs of a record: 4 * M * ff * 256 for the block, + 2 * M * 256 * 256 with the projection
    // (+ 2 * M * n_qkv * 256 with the fused q | k | v projection: booked as n_qkv / 2 extra hidden units)
    ProfScope ps(NW == 8 ? (PROJ ? PK_FFN_PROJ : PK_FFN_FUSED) : (PROJ ? PK_FFN_PROJ_NW4 : PK_FFN_FUSED_NW4), a.M,
                 a.ff + (QKV ? a.n_qkv / 2 : 0), 256, a.M_dev, s);
    hipLaunchKernelGGL((ffn_fused_kernel<PROJ, QKV, NW, PRE>), dim3((unsigned)grid), dim3(64 * NW), lds, s, a);
    CONE_LAUNCH_CHECK();
    return 0;
}

// Tile height by the HOST-known row bound M (never by the device-side count): 64-row tiles (one wave per SIMD) when the
// 128-row tiles would occupy at most half of the CUs -- every 64-row tile then has a CU to itself and finishes in half the
// time; the rows' results do not depend on the choice (same per-wave instruction sequence).
template <bool PROJ, bool QKV>
static int launch_ffn_t(const FfnArgs& a, hipStream_t s) {
    int n_cu = 0;
    const int rc = launch_ffn_nw<PROJ, QKV, 8>(a, s, &n_cu);
    if (rc) return rc;
    const int tiles128 = (a.M + 127) / 128;
    if (!QKV && 2 * tiles128 <= n_cu) return launch_ffn_nw<PROJ, QKV, 4>(a, s, nullptr);
    return launch_ffn_nw<PROJ, QKV, 8>(a, s, nullptr);
}

// Rows past the last FULL round of the persistent grid (n_cu tiles of 128 rows) cost the row forms a partial round -- half
// a round or more (0.15 - 0.29 ms) whatever their number; up to FFN_SPLIT_GROUPS groups of them are handed to the wide form
// instead (52 / 124 us for <= 4 096 / 8 192 rows; same bits).  Returns the row count of the full rounds, or M (no split).
#ifndef CONE_FFN_SPLIT_GROUPS
#define CONE_FFN_SPLIT_GROUPS 512
#endif
static int ffn_full_round_rows(int M, int n_cu, int ff) {
    const int tiles = (M + 127) / 128;
    if (tiles <= n_cu || !ffn_wide_supported(ff)) return M;
    const int64_t full = (int64_t)(tiles / n_cu) * n_cu * 128;
    const int64_t rem = M - full;
    return rem > 0 && (rem + 15) / 16 <= CONE_FFN_SPLIT_GROUPS ? (int)full : M;
}

int launch_ffn_fused(const float* X, int ldx, const float* W1, const float* b1, const float* W2, const float* b2,
                     const float* ln_g, const float* ln_b, float* OUT, int ldo, int M, const int* M_dev, int ff,
                     hipStream_t s) {
    CONE_REQUIRE(ffn_fused_supported(ff), "fused FFN: dim_feedforward=%d unsupported", ff);
    CONE_REQUIRE(X && W1 && b1 && W2 && b2 && ln_g && ln_b && OUT, "fused FFN: null argument");
    CONE_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0, "fused FFN: row strides must be multiples of 4");
    if (M <= 0) return 0;
    if ((M + 15) / 16 <= FFN_WIDE_GROUPS && ffn_wide_supported(ff))     // a few row groups: the wide form (same bits)
        return launch_ffn_wide(X, ldx, W1, b1, W2, b2, ln_g, ln_b, OUT, ldo, M, M_dev, ff, s);
    FfnArgs a{};
    a.X = X; a.ldx = ldx; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.ln_g = ln_g; a.ln_b = ln_b;
    a.OUT = OUT; a.ldo = ldo; a.M = M; a.M_dev = M_dev; a.ff = ff;
    int n_cu = 0;
    if (int rc = launch_ffn_nw<false, false, 8>(a, s, &n_cu)) return rc;
    const int m1 = ffn_full_round_rows(M, n_cu, ff);
    if (m1 == M) return launch_ffn_t<false, false>(a, s);
    a.M = m1;
    if (int rc = launch_ffn_t<false, false>(a, s)) return rc;
    return launch_ffn_wide(X + (size_t)m1 * ldx, ldx, W1, b1, W2, b2, ln_g, ln_b, OUT + (size_t)m1 * ldo, ldo, M - m1, M_dev, ff,
                           s, m1);
}

int launch_proj_ffn_fused(const float* A, int lda, const float* Wo, const float* bo, const float* R, int ldr,
                          const float* pg, const float* pb, const float* W1, const float* b1, const float* W2,
                          const float* b2, const float* ln_g, const float* ln_b, float* OUT, int ldo, int M,
                          const int* M_dev, int ff, hipStream_t s, const int* r_idx, const float* R2, const float* Wq,
                          const float* qb, float* QKV, int ldq, int n_qkv) {
    CONE_REQUIRE(!r_idx || R2, "fused layer tail: a gathered residual needs both source matrices");
    CONE_REQUIRE(ffn_fused_supported(ff), "fused layer tail: dim_feedforward=%d unsupported", ff);
    CONE_REQUIRE(A && Wo && bo && R && pg && pb && W1 && b1 && W2 && b2 && ln_g && ln_b && OUT, "fused layer tail: null argument");
    CONE_REQUIRE(lda % 4 == 0 && ldr % 4 == 0 && ldo % 4 == 0, "fused layer tail: row strides must be multiples of 4");
    if (M <= 0) return 0;
    if (!Wq && (M + 15) / 16 <= FFN_WIDE_GROUPS && ffn_wide_supported(ff))     // a few row groups: the wide form (same bits)
        return launch_proj_ffn_wide(A, lda, Wo, bo, R, ldr, pg, pb, W1, b1, W2, b2, ln_g, ln_b, OUT, ldo, M, M_dev, ff, s, r_idx, R2);
    FfnArgs a{};
    a.A = A; a.lda = lda; a.Wo = Wo; a.bo = bo; a.R = R; a.ldr = ldr; a.pg = pg; a.pb = pb;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.ln_g = ln_g; a.ln_b = ln_b;
    a.OUT = OUT; a.ldo = ldo; a.M = M; a.M_dev = M_dev; a.ff = ff; a.r_idx = r_idx; a.R2 = R2;
    if (Wq) {
        CONE_REQUIRE(qb && QKV && n_qkv >= 32 && n_qkv % 32 == 0 && ldq % 4 == 0, "fused layer tail: bad q|k|v arguments");
        a.Wq = Wq; a.qb = qb; a.QKV = QKV; a.ldq = ldq; a.n_qkv = n_qkv;
        return launch_ffn_t<true, true>(a, s);
    }
    int n_cu = 0;
    if (int rc = launch_ffn_nw<true, false, 8>(a, s, &n_cu)) return rc;
    const int m1 = ffn_full_round_rows(M, n_cu, ff);
    if (m1 == M) return launch_ffn_t<true, false>(a, s);
    a.M = m1;
    if (int rc = launch_ffn_t<true, false>(a, s)) return rc;
    return launch_proj_ffn_wide(A + (size_t)m1 * lda, lda, Wo, bo, r_idx ? R : R + (size_t)m1 * ldr, ldr, pg, pb, W1, b1, W2, b2,
                                ln_g, ln_b, OUT + (size_t)m1 * ldo, ldo, M - m1, M_dev, ff, s, r_idx ? r_idx + m1 : nullptr, R2,
                                m1);
}

// The pre-norm layer tail (--pre_norm): OUT = x1 + FFN(LN(x1; pg, pb)), x1 = R + A Wo^T + bo; OUT2 (may be null) =
// LN(OUT; n2g, n2b).  The persistent 128-row kernel for every row count (the option is off in every shipped configuration:
// no wide / 64-row forms).  OUT may be R (in place).
int launch_proj_ffn_prenorm(const float* A, int lda, const float* Wo, const float* bo, const float* R, int ldr, const float* pg,
                            const float* pb, const float* W1, const float* b1, const float* W2, const float* b2, float* OUT,
                            int ldo, const float* n2g, const float* n2b, float* OUT2, int ldo2, int M, const int* M_dev, int ff,
                            hipStream_t s, const int* r_idx, const float* R2) {
    CONE_REQUIRE(!r_idx || R2, "pre-norm layer tail: a gathered residual needs both source matrices");
    CONE_REQUIRE(ffn_fused_supported(ff), "pre-norm layer tail: dim_feedforward=%d unsupported", ff);
    CONE_REQUIRE(A && Wo && bo && R && pg && pb && W1 && b1 && W2 && b2 && OUT && (!OUT2 || (n2g && n2b)),
                 "pre-norm layer tail: null argument");
    CONE_REQUIRE(lda % 4 == 0 && ldr % 4 == 0 && ldo % 4 == 0 && ldo2 % 4 == 0, "pre-norm layer tail: row strides must be multiples of 4");
    if (M <= 0) return 0;
    FfnArgs a{};
    a.A = A; a.lda = lda; a.Wo = Wo; a.bo = bo; a.R = R; a.ldr = ldr; a.pg = pg; a.pb = pb;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.ln_g = n2g ? n2g : pg; a.ln_b = n2b ? n2b : pb;
    a.OUT = OUT; a.ldo = ldo; a.OUT2 = OUT2; a.ldo2 = ldo2; a.M = M; a.M_dev = M_dev; a.ff = ff; a.r_idx = r_idx; a.R2 = R2;
    return launch_ffn_nw<true, false, 8, true>(a, s, nullptr);
}

}  // namespace cone
