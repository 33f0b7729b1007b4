// A chain of row-wise stages over few rows as ONE launch (gemm.hip): the decoder heads (decoder.norm -> class head; span MLP
// -> span head: cone/model.py:112-117) and the adapter pair of the proposal matching (cone/model.py:146-150) are 2 - 5
// launches of a few microseconds each on the single-query path and on a rank's share of a sharded split; here a workgroup
// owns 16 rows and walks the stages with the rows on chip.  Every stage runs the arithmetic of the launch it replaces --
// the fma chain and per-row epilogue of gemm_rows_small_kernel, the moments of layernorm_kernel, the dot of rowdot_kernel --
// so the results are bit-identical to the separate launches (and, through the small form's own identity, to the 128-row
// tiles a large batch runs).
#pragma once
#include <hip/hip_runtime.h>

namespace cone {

constexpr int CHAIN_MAX_STAGES = 4;

struct ChainStage {
    int kind;                              // 0: x <- epi(x W^T + bias), W (256, K); 1: x <- LayerNorm(x; ln_g, ln_b) (eps 1e-5)
    int K;                                 // kind 0: input width (stage 0: the width of A, a multiple of 16, <= 1024; later: 256)
    const float* W; const float* bias;     // kind 0
    int flags;                             // kind 0: EPI_RELU | EPI_RESIDUAL (common.h), as launch_gemm
    const float* R; int ldr;               // EPI_RESIDUAL: (M, 256) rows
    const float* ln_g; const float* ln_b;  // kind 1
    float* C; int ldc;                     // the stage's output rows (null: they stay on chip)
    // optional head on the stage's OUTPUT rows: hout[row * hld + n] = act(<x, hw[n]> + hb[n]), n < hnout <= 2; hact 1 = sigmoid
    const float* hw; const float* hb; float* hout; int hld; int hnout; int hact;
};

struct ChainArgs {
    const float* A; int lda;               // (M, K of stage 0) input rows
    int M; int n_stages;
    ChainStage st[CHAIN_MAX_STAGES];
};

// the row counts launch_gemm would run on its small form (16-row tiles): only there the chain is never slower than the
// launches it replaces (the same weight traffic per 16 rows)
bool rows_chain_supported(int M);
// rows of a launch that takes the SPREAD form of the row GEMM instead (K <= 256, no LayerNorm epilogue): there the separate
// launches beat the chain (one wave per output tile against one workgroup per 16 rows walking every stage)
bool gemm_rows_spread_rows(int M);
int launch_rows_chain(const ChainArgs& a, hipStream_t s);

}  // namespace cone
