/*
 * cone_hip.h -- C ABI of libcone_hip.so: the MI355X (gfx950) implementation of CONE's
 * coarse-to-fine inference hot path.
 *
 * The reference (houzhijian/CONE) has no FFI: the path is plain Python calling torch.nn.
 * Each entry point below therefore names the reference *Python* site it replaces
 * (file:line relative to the reference repository).  A maintainer binds these with
 * ctypes (cone_amd/_lib.py is that binding).
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer to fp32 / int32 / fp64 memory owned by the
 *     caller (torch tensors), row-major, densely packed unless a stride is given;
 *     cone_weights pointers may be host or device (copied once at cone_model_create);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on
 *     it, nothing synchronises, nothing allocates (scratch comes in through `ws`) -- except
 *     cone_model_create / cone_model_destroy, which allocate / free the handle's weight arena
 *     (hipMalloc + blocking copies) and therefore synchronise with the device;
 *   - return value 0 = success, negative = error (CONE_E_*), text in cone_last_error();
 *     nothing throws across the ABI;
 *   - a cone_model is immutable after creation (cone_model_set_option, a parity-test hook, is the
 *     one exception and must not race with calls on the handle): one handle may be used from several
 *     streams/threads concurrently as long as each call has its own workspace.  The library keeps no
 *     other mutable state besides the opt-in launch timer (cone_prof_*).
 */
#ifndef CONE_HIP_H
#define CONE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CONE_HIP_ABI_VERSION 8

#define CONE_E_INVALID (-1)  /* bad argument / unsupported shape */
#define CONE_E_HIP (-2)      /* a HIP runtime call failed        */
#define CONE_E_WORKSPACE (-3) /* workspace too small             */

#define CONE_MAX_LAYERS 8
#define CONE_MAX_PROJ 3
#define CONE_TABLE_MAX_V_L 255 /* longest window (clips) the handle's own position tables cover */
#define CONE_MAX_WINDOW_TOKENS 256 /* clips + text tokens of one window (WINDOW_LENGTH + max_q_l of the reference's scripts) */

typedef struct cone_model cone_model;

/* torch.nn.Linear: weight [out][in], bias [out] */
typedef struct { const float* w; const float* b; } cone_linear_w;
/* torch.nn.LayerNorm: weight, bias (eps 1e-5) */
typedef struct { const float* g; const float* b; } cone_ln_w;
/* torch.nn.MultiheadAttention: packed in_proj [3d][d] rows Wq|Wk|Wv, in_proj_bias [3d] */
typedef struct { const float* in_proj_w; const float* in_proj_b; cone_linear_w out_proj; } cone_mha_w;
typedef struct { cone_mha_w self_attn; cone_linear_w linear1, linear2; cone_ln_w norm1, norm2; } cone_enc_layer_w;
typedef struct {
    cone_mha_w self_attn, cross_attn;
    cone_linear_w linear1, linear2;
    cone_ln_w norm1, norm2, norm3;
} cone_dec_layer_w;

/* The tensors of CONE.state_dict() (cone/model.py:19-80) that inference reads. */
typedef struct {
    int32_t hidden_dim, nheads, dim_ff, enc_layers, dec_layers, num_queries;
    int32_t n_input_proj, t_dim, v_dim, has_adapter;   /* v_dim: APPEARANCE clip features (pre-filter, proposal matching,
                                                        * adapter: cone/model.py:80, 186-208)                             */
    int32_t v_motion_dim;                              /* MOTION clip features = the window model's video input
                                                        * (input_vid_proj: cone/model.py:67); equal to v_dim in every
                                                        * shipped script (one LMDB serves both: cone/ego4d_mad_dataloader.py:77) */
    cone_ln_w vid_proj_ln[CONE_MAX_PROJ]; cone_linear_w vid_proj[CONE_MAX_PROJ]; /* input_vid_proj.{i} */
    cone_ln_w txt_proj_ln[CONE_MAX_PROJ]; cone_linear_w txt_proj[CONE_MAX_PROJ]; /* input_txt_proj.{i} */
    cone_enc_layer_w enc[CONE_MAX_LAYERS];   /* transformer.encoder.layers.{i} */
    cone_dec_layer_w dec[CONE_MAX_LAYERS];   /* transformer.decoder.layers.{i} */
    cone_ln_w dec_norm;                      /* transformer.decoder.norm        */
    const float* query_embed;                /* query_embed.weight [Nq][d]      */
    cone_linear_w class_embed;               /* [2][d]                          */
    cone_linear_w span_embed[3];             /* span_embed.layers.{0,1,2}       */
    cone_linear_w saliency_proj;             /* [1][d]                          */
    cone_linear_w adapter[2];                /* adapter_layer.layers.{0,1}      */
    const float* pos_dim_t;                  /* [d] temperature**(2*(i//2)/d), cone/position_encoding.py:66-67 */
    /* (ABI 5) --use_txt_pos (cone/config.py:115, cone/model.py:106): the position term of text token t of a query is
     * TrainablePositionalEncoding(src_txt) = LayerNorm(src_txt[t] + position_embeddings[t]) (cone/position_encoding.py:10-32).
     * txt_pos_embed = txt_position_embed.position_embeddings.weight [txt_pos_rows = max_q_l][d], txt_pos_ln its LayerNorm;
     * NULL = the option is off (every shipped configuration): text tokens carry a zero position term.  (ABI 7) With it set the
     * forward runs the table path when it gets the tokens' own position rows (cone_layer0.txt_pos / txt_pos_qk from
     * cone_layer0_text_positions; cone_forward_windows builds them itself), otherwise the general path (x + pos materialised
     * per token). */
    const float* txt_pos_embed; int32_t txt_pos_rows; cone_ln_w txt_pos_ln;
    /* (ABI 5) --pre_norm (cone/config.py:120 -> normalize_before, cone/transformer.py:19-36): pre_norm != 0 = every layer
     * normalises its input (forward_pre) and the encoder ends with enc_norm = transformer.encoder.norm, which exists only
     * then.  0 (every shipped configuration) = post-norm.  (ABI 6) such a model runs the table path as well: the fused layer
     * tail in its pre-norm form, first-layer row caches of in_proj(norm1(row)), the folded decoder cross-attention. */
    int32_t pre_norm; cone_ln_w enc_norm;
    /* (ABI 8) The longest window, in clips, this checkpoint is run on = max_v_l of build_model (cone/model.py:468-486,
     * WINDOW_LENGTH of the reference's scripts): cone_model_create builds the handle's position tables for windows of up to
     * this many clips (row lv (lv - 1) / 2 + p: a shorter bound is a prefix of a longer one; 4 096 rows = 21 MB for 90 clips
     * with two encoder layers, 32 641 rows = 167 MB for 255).  0 = CONE_TABLE_MAX_V_L.  Longer windows still run, on the
     * general path (x + pos materialised), or on tables the caller brings (cone_pos_tables, cone_layer0). */
    int32_t table_max_v_l;
} cone_weights;

const char* cone_last_error(void);
int cone_abi_version(void);

/* build_model + load_state_dict (cone/model.py:468-521, cone/inference.py:525-527). */
int cone_model_create(const cone_weights* w, cone_model** out);
void cone_model_destroy(cone_model* m);

/* ------------------------------------------------------------------ stage A: pre-filter */

/* A2, cone/inference.py:254-258: out = adapter(x)+x, then (renorm != 0) out /= ||out||_2 (no eps), row-wise;
 * renorm == 0 is the single-video localizer's form (run_on_video/cone_localizator.py:135-138).
 * x,out (n_rows, v_dim).  ws >= cone_adapter_norm_workspace(m, n_rows) bytes. */
size_t cone_adapter_norm_workspace(const cone_model* m, int64_t n_rows);
int cone_adapter_norm(const cone_model* m, const float* x, int64_t n_rows, float* out, int renorm,
                      void* ws, size_t ws_bytes, void* stream);

/* Row-wise L2 normalisation.  clamp == 0: x / (||x||_2 + eps) = l2_normalize_np_array
 * (utils/basic_utils.py:97-99); clamp != 0: x / max(||x||_2, eps) = torch.nn.functional.normalize
 * (run_on_video/cone_localizator.py:129-133). */
int cone_l2_normalize_rows(const float* x, int64_t n_rows, int dim, float eps, int clamp, float* out, void* stream);

/* A3+A4, cone/inference.py:284-296: frame_scores[q][f] = <vid[f], txt[q]>;
 * win_scores[q][i] = max(frame_scores[q][max((i-1)S,0) : min((i-1)S+W, ctx_l)]),
 * num_window = ceil(ctx_l/S)+1, S = W/2.  vid (ctx_l,dv), txt (nq,dv), win_scores (nq,num_window).
 * The window max is fused into the frame-score stream (a running max per half window of S frames; each frame belongs
 * to two windows), so the (nq,ctx_l) frame-score matrix -- which the reference computes only to take this max -- is
 * OPTIONAL: frame_scores may be NULL (nothing of that size is written), or receives the scores as before.
 * ws >= cone_prefilter_scores_workspace(ctx_l, nq, W) bytes (two floats per query and half window).
 * Forms (chosen by nq): 1 - 4 queries stream the clip rows once with non-temporal loads, the query vectors in registers (a
 * query's bits do not depend on the others of the launch); 5 or more run fp32-MFMA tiles of 16 / 32 / 64 queries per pass
 * (another summation order: ~1e-7 relative). */
int64_t cone_num_windows(int64_t ctx_l, int W);
size_t cone_prefilter_scores_workspace(int64_t ctx_l, int nq, int W);
int cone_prefilter_scores(const float* vid, int64_t ctx_l, int dv, const float* txt, int nq,
                          int W, int S, float* frame_scores, float* win_scores, void* ws, size_t ws_bytes,
                          void* stream);

/* OPT-IN form of cone_prefilter_scores without the frame-score matrix: 8 or more queries over one video run on the bf16
 * matrix cores, every fp32 product as six partial products of three-piece bf16 operands with fp32 accumulation -- the
 * accuracy of the exact-fp32 MFMA chain (tools/probe/split_bf16_probe.hip) at a rate that leaves the 64-query stream
 * HBM-bound instead of matrix-pipe-bound.  Fewer than 8 queries take cone_prefilter_scores' own kernels (1 - 4: the streaming
 * kernel; 5 - 7: one 16-query tile of the exact-fp32 matrix-core kernel).
 * ws >= cone_prefilter_scores_split_workspace(...) bytes (the score planes + the split query image).  The default path
 * (cone_prefilter_scores, cone/inference.py's drop-in) stays exact fp32. */
size_t cone_prefilter_scores_split_workspace(int64_t ctx_l, int nq, int W, int dv);
int cone_prefilter_scores_split(const float* vid, int64_t ctx_l, int dv, const float* txt, int nq, int W, int S,
                                float* win_scores, void* ws, size_t ws_bytes, void* stream);

/* A4, cone/inference.py:297-299 (+ [:topk], cone/ego4d_mad_dataloader.py:146): the first k entries
 * of the descending sort of each row of win_scores (nq,num_window); ties -> lower window index
 * first (stable order).  idx (nq,k) int32, val (nq,k) fp32 (val may be NULL). */
int cone_topk_windows(const float* win_scores, int nq, int64_t num_window, int k,
                      int32_t* idx, float* val, void* stream);
/* The same with caller-owned scratch: rows longer than 8 192 windows (MAD scale) run as a two-level selection -- a
 * stable top-k per 4 096-window chunk (per-wave selection out of registers, no barrier inside the passes), then a merge of the chunk lists (same order, bit for bit) -- instead
 * of k passes over the whole row.  ws >= cone_topk_windows_workspace(...) bytes (0 for short rows). */
size_t cone_topk_windows_workspace(int nq, int64_t num_window, int k);
int cone_topk_windows_ws(const float* win_scores, int nq, int64_t num_window, int k, int32_t* idx, float* val,
                         void* ws, size_t ws_bytes, void* stream);

/* A3+A4 for a whole split in three launches.  The clip features of all videos sit back to back in
 * `arena` (rows, dv).  A group g = one video (rows g_row0[g] .. +g_ctx_l[g]) and up to 4 of its
 * queries g_q[g][0..3] (indices into cls (nq,dv); -1 = unused slot), so a video's rows are read once
 * per group.  Query q writes its frame scores at frame_scores + q_fs_off[q] (q_ctx_l[q] values) and its
 * window scores at win_scores + q_win_off[q]; topk_idx (nq,k) receives the first k windows of the
 * stable descending order, padded with -1 when the video has fewer than k windows.
 * max_ctx_l bounds every ctx_l (host-known, sizes the grid). */
int cone_prefilter_batched(const float* arena, int dv, const float* cls, const int64_t* g_row0,
                           const int32_t* g_ctx_l, const int32_t* g_q, int ng, int max_ctx_l,
                           const int64_t* q_fs_off, const int64_t* q_win_off, const int32_t* q_ctx_l,
                           int nq, int W, int S, float* frame_scores, float* win_scores, int k,
                           int32_t* topk_idx, void* stream);

/* A5, eval branch of StartEndDataset.__getitem__ + collate (cone/ego4d_mad_dataloader.py:144-159, 229-234, 305-344) as
 * index arithmetic: win_idx (nq, K) int32 holds the ranked window indices of every query, valid entries first (a video of
 * fewer than K windows: the tail is -1 and never read).  Row b of every output is (query row_q[b], rank slot row_slot[b])
 * -- n_rows rows in annotation order, then rank order: the shape of the list is host metadata, a query owns
 * min(K, ceil(ctx_l / S) + 1) windows (cone/inference.py:286-299, cone/ego4d_mad_dataloader.py:146) -- or, with
 * row_q == row_slot == NULL and n_rows == nq * K, the dense list (b / K, b % K).  Per-query metadata: q_ctx_l (clips of the
 * query's video), q_vid_off (first arena
 * row of that video), tok_off / tok_len (the query's text rows).  Outputs (n_rows) int32 each: vid_row0, vid_len,
 * video_start (window i covers clips [max(0, (i-1) S), min(ctx_l, (i-1) S + max_v_l)), S = max_v_l / 2), txt_row0,
 * txt_len, cls_row, and pad_len = the zero-padded clip length of the window's reference batch -- the longest window among
 * the eval_bsz consecutive queries of the SPLIT it belongs to (hazard H3: cone/model.py:186-199 divides by a length
 * clipped to it).  batch_pad (n_batches) int32 is that table, indexed by (q_base + query) / eval_bsz with q_base = the
 * first query's index in the split: derive_pad != 0 computes it from these windows (only the reference's when the
 * queries are whole reference batches -- the caller's responsibility), derive_pad == 0 reads the split's table. */
int cone_window_table(const int32_t* win_idx, int nq, int K, const int32_t* row_q, const int32_t* row_slot, int n_rows,
                      const int32_t* q_ctx_l, const int32_t* q_vid_off,
                      const int32_t* tok_off, const int32_t* tok_len, int q_base, int eval_bsz, int max_v_l,
                      int32_t* batch_pad, int derive_pad, int n_batches, int32_t* vid_row0, int32_t* vid_len,
                      int32_t* video_start, int32_t* pad_len, int32_t* txt_row0, int32_t* txt_len, int32_t* cls_row,
                      void* stream);

/* --------------------------------------------------------- stage B: intra-window model */

/* A6, cone/model.py:100-101 (input_vid_proj / input_txt_proj): row-wise LN->Linear->ReLU->LN->Linear.
 * which: 0 = video (v_motion_dim in), 1 = text (t_dim in).  x (n_rows,din), out (n_rows,d). */
size_t cone_project_workspace(const cone_model* m, int which, int64_t n_rows);
int cone_project_tokens(const cone_model* m, int which, const float* x, int64_t n_rows, float* out,
                        void* ws, size_t ws_bytes, void* stream);

/* Optional taps for parity bisecting; any pointer may be NULL. */
typedef struct {
    float* memory;     /* (B, Lv_pad+Lq_pad, d): encoder output, valid tokens only, rest 0 */
    float* hs;         /* (dec_layers, B, Nq, d): decoder outputs after decoder.norm         */
    float* aux_logits; /* (dec_layers-1, B, Nq, 2)                                          */
    float* aux_spans;  /* (dec_layers-1, B, Nq, 2)                                          */
} cone_taps;

/* A6-A11, CONE.forward (cone/model.py:82-128) on zero-padded tensors exactly as
 * prepare_batch_inputs delivers them: vid (B,Lv_pad,v_motion_dim), txt (B,Lq_pad,t_dim); masks are prefix
 * masks given as valid lengths vid_len[B], txt_len[B] (int32, device).
 * (ABI 6) The batch is compacted first -- the valid clip / token rows are gathered by the first LayerNorm of their input
 * projection, padding is never projected -- the projections and the first encoder layer's q | k | v rows run once per compact
 * row (device-side row counts), and the windows enter the packed forward below as (row0, len) pairs with those row caches:
 * the SAME launches as the eval driver's arena path (gathering first layer, the handle's position tables, fused layer tails,
 * folded decoder cross-attention), the same bits for the same windows.  Lv_pad + Lq_pad <= CONE_MAX_WINDOW_TOKENS.
 * Outputs: logits (B,Nq,2), spans (B,Nq,2) = sigmoid(center,width), saliency (B,Lv_pad; may be NULL: the
 * saliency head is then skipped -- cone/inference.py computes and never reads it, :54-59; likewise the heads and
 * decoder.norm of the intermediate decoder layers run only when taps ask for hs / aux_logits / aux_spans)
 * (entries at padded clips are written as 0; the reference leaves them unspecified/unused). */
size_t cone_forward_workspace(const cone_model* m, int B, int Lv_pad, int Lq_pad);
int cone_forward_windows(const cone_model* m, const float* vid, const int32_t* vid_len,
                         const float* txt, const int32_t* txt_len, int B, int Lv_pad, int Lq_pad,
                         float* logits, float* spans, float* saliency, const cone_taps* taps,
                         void* ws, size_t ws_bytes, void* stream);

/* Valid lengths of the reference's prefix masks (pad_sequences_1d, utils/tensor_utils.py:50-52: 1 = valid, (B, L) fp32):
 * len[b] = sum_j mask[b][j] as int32 -- the vid_len / txt_len arguments of the two entry points above. */
int cone_mask_lengths(const float* mask, int B, int L, int32_t* len, void* stream);

/* Same computation on already-projected token arenas (cone_project_tokens outputs), windows given
 * by index: window b = vproj rows [vid_row0[b], +vid_len[b]) followed by tproj rows
 * [txt_row0[b], +txt_len[b]).  This is how the eval driver avoids re-projecting the clips shared by
 * overlapping windows and the text replicated across a query's windows.
 * Lv_max/Lq_max bound the lengths (host-known); saliency is (B, Lv_max). */
/* Optional row caches (+ caller-owned position tables).  Everything ahead of the first attention is row-wise, and the
 * position term enters every attention through a LINEAR map, so it can be split off and tabulated:
 *   (x + pos) W^T + b = (x W^T + b) + pos W^T.
 * qkv_vid (n_clips, 3d) / qkv_txt (n_tokens, 3d) = cone_layer0_project of the projected arenas: the FIRST encoder
 *   layer's in_proj once per clip / per text token instead of once per window row (two M-row GEMMs, ~10 % of the
 *   FLOPs, become a gather inside the attention kernel);
 * pos_rows (R, d), R = cone_pos_table_rows(max_v_l): row Lv(Lv-1)/2 + p = PositionEmbeddingSine of clip p of a
 *   window with Lv valid clips (cone/position_encoding.py:51-72); the LAST row (R - 1) is all zeros: the position of a
 *   text token (cone/model.py:106), added like any other row;
 * pos_qk (enc_layers, R, 2d): pos_rows [W_q | W_k]^T of every encoder layer (no bias).
 * With the tables the later encoder layers run ONE N = 3d GEMM on x (the attention kernel adds the pos_qk row to
 * q | k in its staging loads) and the decoder's cross-attention forms its keys memory + pos from pos_rows: no
 * x + pos matrix is ever written.
 * (ABI 6) The model handle OWNS such tables (built at cone_model_create for windows of up to CONE_TABLE_MAX_V_L clips: row
 * lv (lv - 1) / 2 + p is the same row whatever the bound).  pos_rows / pos_qk == NULL -- or the whole cone_layer0 == NULL --
 * means "the handle's": every call runs the table path.  qkv_vid / qkv_txt == NULL (no row caches): the packed layer input is
 * written once and the first layer runs like the later ones (one N = 3d GEMM on x, position rows from the table) -- the same
 * bits as with the caches, ~10 % more FLOPs when windows share clips / text.  max_v_l describes caller-owned tables only. */
typedef struct {
    const float* qkv_vid; const float* qkv_txt;
    const float* pos_qk; const float* pos_rows;
    int32_t max_v_l;
    /* (ABI 7) --use_txt_pos checkpoints (cone/config.py:115) on the table path: the position term of a text token is a
     * per-TOKEN row (cone/model.py:106) -- txt_pos (n_txt, d) and its images under every encoder layer's [W_q | W_k],
     * txt_pos_qk (enc_layers, n_txt, 2d), both from cone_layer0_text_positions on the projected token arena (row i of
     * either belongs to token row i of txt_proj).  NULL for such a model: the general path (x + pos materialised).
     * Ignored by models without the option. */
    const float* txt_pos; const float* txt_pos_qk;
    int64_t n_txt;
} cone_layer0;
int64_t cone_pos_table_rows(int max_v_l);
int cone_pos_tables(const cone_model* m, int max_v_l, float* pos_rows, float* pos_qk, void* stream);
/* (ABI 6) ws >= cone_layer0_project_workspace(m, n_rows) bytes: 0 for a post-norm model; a --pre_norm model's first in_proj
 * reads norm1(row) (cone/transformer.py:250-252), normalised into the scratch first. */
size_t cone_layer0_project_workspace(const cone_model* m, int64_t n_rows);
int cone_layer0_project(const cone_model* m, const float* proj_rows, int64_t n_rows, float* qkv, void* ws, size_t ws_bytes,
                        void* stream);
/* (ABI 7) --use_txt_pos: the text-side position rows of the table path, once per token of the projected text arena (they
 * are shared by all windows of a query, like the row caches).  tok_index[i] = index of token row i inside its query (0-based);
 * txt_pos[i] = LayerNorm(txt_proj_rows[i] + position_embeddings[tok_index[i]]) (cone/position_encoding.py:21-31, eval: no
 * dropout), txt_pos_qk[l][i] = txt_pos[i] [W_q | W_k]_l^T (no bias) for every encoder layer l.  Error for a model without
 * txt_pos_embed. */
int cone_layer0_text_positions(const cone_model* m, const float* txt_proj_rows, const int32_t* tok_index, int64_t n_rows,
                               float* txt_pos, float* txt_pos_qk, void* stream);

/* Workspace: 6 KiB per token row (B * (Lv_max + Lq_max) rows); 13 KiB on the general path (--use_txt_pos without its position rows, A/B switches). */
size_t cone_forward_packed_workspace(const cone_model* m, int B, int Lv_max, int Lq_max,
                                     const cone_layer0* l0 /* as passed to cone_forward_packed */);
int cone_forward_packed(const cone_model* m, const float* vproj, const int32_t* vid_row0,
                        const int32_t* vid_len, const float* tproj, const int32_t* txt_row0,
                        const int32_t* txt_len, int B, int Lv_max, int Lq_max,
                        float* logits, float* spans, float* saliency, const cone_taps* taps,
                        const cone_layer0* l0 /* may be NULL */, void* ws, size_t ws_bytes, void* stream);

/* A12, CONE.forward_clip_matching (cone/model.py:130-152,178-210).  Gathered form:
 * proposal n of window b averages rows [s,e) of the window's clips vid[vid_row0[b] + ...] where
 * rows >= vid_len[b] count as the zero padding of a (pad_len[b])-long tensor (hazard H3):
 * mean = sum(rows s..min(e,vid_len)-1) / (min(e,pad_len[b]) - s).
 * cls (n_cls,dv) rows selected by cls_row[b]; spans (B,Nq,2) (center,width); match (B,Nq). */
size_t cone_clip_matching_workspace(const cone_model* m, int B);
int cone_clip_matching_gathered(const cone_model* m, const float* cls, const int32_t* cls_row,
                                const float* vid, const int32_t* vid_row0, const int32_t* vid_len,
                                const int32_t* pad_len, const float* spans, int B, float* match,
                                void* ws, size_t ws_bytes, void* stream);
/* Padded form with the reference's signature: cls (B,dv), vid (B,Lv_pad,dv). */
int cone_clip_matching(const cone_model* m, const float* cls, const float* vid, const int32_t* vid_len,
                       int Lv_pad, const float* spans, int B, float* match,
                       void* ws, size_t ws_bytes, void* stream);

/* A13, cone/inference.py:47-82: rows[b][n] = [st, ed, softmax(logits)[0], match] with
 * (st,ed) = (cxw->xx(spans) * duration[b] + video_start[b]) * clip_length in fp32, each window's
 * Nq rows sorted by proposal score (stable, descending) when sort != 0.  rows (B,Nq,4) fp32. */
int cone_compose_rows(const float* logits, const float* spans, const float* match,
                      const int32_t* duration, const int32_t* video_start, float clip_length,
                      int sort, int B, int Nq, float* rows, void* stream);

/* ------------------------------------------------------------ stage C: fusion + NMS */

/* A13 rounding + A14 + A15 for nq queries at once (cone/inference.py:83,103-127,205-217;
 * utils/temporal_nms.py:25-74), all in fp64 like the reference's Python floats:
 *   cand (nq,n_max,4) fp32 rows [st,ed,prop,match], n_valid[q] rows used per query -- or, cand_off != NULL (nq int64):
 *   the candidate rows of query q are rows cand_off[q] .. + n_valid[q] of ONE (rows, 4) matrix (the per-window rows of
 *   cone_compose_rows ARE that matrix: a query's windows are adjacent, no scatter into a padded layout);
 *   every number -> float(f"{x:.4f}"); min-max fusion; dict collapse on (st,ed);
 *   for score_idx in (2 fused, 0 proposal, 1 matching): stable sort desc, [:max_before],
 *   greedy NMS (pseudo-IoU, strict >), stop at max_after; nms_thd == -1 -> top max_after.
 * out_rows (3,nq,max_after,5) fp64 rows [st,ed,prop,match,fused]; out_n (3,nq) int32;
 * out_idx (3,nq,max_after) int32 = index into the query's cand rows (first occurrence of the key).  Every element of the
 * three outputs is written (rows past out_n: zeros, index -1): the buffers need no initialisation. */
int cone_fuse_nms(const float* cand, const int64_t* cand_off, const int32_t* n_valid, int nq, int n_max,
                  double nms_thd, int max_before, int max_after, double* out_rows, int32_t* out_n, int32_t* out_idx,
                  void* stream);

/* Same on fp64 candidate rows (e.g. rows that went through Python and are already rounded; the
 * rounding is idempotent on them). */
int cone_fuse_nms_f64(const double* cand, const int64_t* cand_off, const int32_t* n_valid, int nq, int n_max,
                      double nms_thd, int max_before, int max_after, double* out_rows, int32_t* out_n, int32_t* out_idx,
                      void* stream);

/* temporal_nms (utils/temporal_nms.py:25-74) on one list: pred (n,3) fp64 [st,ed,score];
 * keep_idx (max_after) indices into pred in output order, keep_n (1). */
int cone_temporal_nms(const double* pred, int n, double nms_thd, int max_after,
                      int32_t* keep_idx, int32_t* keep_n, void* stream);

/* A17, HungarianMatcher cost matrix for one target per window (cone/matcher.py:61-95):
 * C[b][n] = cost_span*L1(span_n, tgt_b) - cost_giou*GIoU - cost_class*softmax(logits)[0].
 * logits (B,Nq,2), spans (B,Nq,2) cxw, tgt (B,2) cxw, cost (B,Nq), best (B) = argmin_n. */
int cone_matcher_cost(const float* logits, const float* spans, const float* tgt, int B, int Nq,
                      float cost_span, float cost_giou, float cost_class, float* cost, int32_t* best,
                      void* stream);

/* SetCriterion.forward (cone/model.py:213-425) with its HungarianMatcher (cone/matcher.py:37-106): forward values of
 * one decoder layer's losses -- the matcher's exact assignment (<= 8 slots and <= 8 target spans per window, the
 * reference calls scipy's linear_sum_assignment), then loss_span = mean L1 of the matched (center, width) pairs,
 * loss_giou = mean (1 - GIoU), loss_label = mean over all slots (the negative window's too when neg_logits != NULL) of
 * the class-weighted cross-entropy (foreground = class 0 for matched slots, weight 1; background weight eos_coef),
 * class_error = 100 - top-1 accuracy of the matched slots, loss_saliency = hinge over the (pos, neg) clip pairs
 * (+ the negative window's maximum when neg_saliency != NULL), 0 when saliency == NULL.
 * logits, spans, neg_logits (B, Nq, 2); tgt (sum T, 2) (center, width), tgt_off (B + 1); tgt == NULL: no targets --
 * every slot is background and only loss_label is meaningful (:385-388).  saliency (B, L), pos_idx / neg_idx (B, P).
 * Outputs: assign (B, Nq) int32 (target index inside the window, -1 = unmatched; may be NULL), part (B, 8) scratch,
 * losses (5) = loss_span, loss_giou, loss_label, class_error, loss_saliency. */
int cone_criterion_forward(const float* logits, const float* spans, const float* tgt, const int32_t* tgt_off,
                           const float* neg_logits, const float* saliency, int L, const int32_t* pos_idx,
                           const int32_t* neg_idx, int P, const float* neg_saliency, int L2, int B, int Nq,
                           float cost_span, float cost_giou, float cost_class, float eos_coef, float saliency_margin,
                           int32_t* assign, float* part, float* losses, void* stream);
/* loss_adapter (cone/model.py:249-264): (CE(sim / T, diag) + CE(sim^T / T, diag)) / 2 for sim (n, n); loss (1). */
int cone_adapter_nce(const float* sim, int n, float temperature, float* loss, void* stream);

/* -------------------------------------------------------------------- metrics on the device
 * Recall@K / IoU counts from the kept rows (layout of cone_fuse_nms: rows (nq, max_after, 5) fp64 [st, ed, ...],
 * n (nq) valid rows) and one target span per query gt (nq, 2) fp64 seconds -- device pointers.  thresholds /
 * topk are HOST arrays (<= 8 / <= 16 entries).  hits (n_thr, n_topk) int64 on the device: number of queries
 * with a prediction of IoU > threshold among their first K rows; top1_iou (nq) fp64: IoU of the first row.
 * mode 0 = standalone_eval/evaluate_ego4d_nlq.py:41-60,93-103 (float64, union clamped at 0);
 * mode 1 = standalone_eval/evaluate_mad.py:33-38,87-104 (float32 arithmetic, thresholds as fp32). */
int cone_eval_recall(const double* rows, const int32_t* n, const double* gt, int nq, int max_after,
                     const double* thresholds, int n_thr, const int32_t* topk, int n_topk, int mode,
                     int64_t* hits, double* top1_iou, void* stream);
/* Window pre-filter recall, standalone_eval/evaluate_pre_filtered_window.py:45-66: win_idx (nq, k) int32 ranked
 * windows (-1 padded), gt (nq, 2) fp64 seconds; hits (n_topk) int64 = queries whose first K windows contain
 * one of floor(start/S) .. ceil(end/S), start/end in clips (seconds / clip_length), S = slide. */
int cone_eval_window_recall(const int32_t* win_idx, int nq, int k, const double* gt, double clip_length,
                            int slide, const int32_t* topk, int n_topk, int64_t* hits, void* stream);

/* -------------------------------------------------------------------- measurement
 * Opt-in per-launch timing with hipEvents on the launch stream (used by bench.py for the roofline
 * line).  cone_prof_enable(1) clears and starts recording, (0) stops.  After synchronising the
 * stream, cone_prof_collect fills up to max_rec records of 5 doubles {kind, a, b, c, milliseconds}:
 * kind 0/1/2 = GEMM tiles 128x128 / 128x128 with fused addend / 64x256 with fused LayerNorm, (a,b,c) =
 * (M rows actually processed, N, K); kind 3 = encoder attention (B windows, Lmax, source mode); kind 4 = frame-score
 * stream (ctx_l, dv, queries in the launch); kind 5/6 = 128x256 row-owning GEMM tile with 4 / 8 waves (M, N, K); kind 7 =
 * fused decoder cross-attention (B windows, Lmax, nq); kind 8 = fused feed-forward block on the persistent 128-row grid
 * (M rows, ff, 256): 4*M*ff*256 FLOPs; kind 9 = the same with the output projection: + 2*M*256*256.  One kind per KERNEL:
 * kinds 10 / 11 = kinds 8 / 9 computed by the wide form (ffn_wide_kernel: launches of a few row groups and the rows past the
 * last full round of a big launch -- a record of its own, with the rows it covered); 12 / 13 = the 64-row / 4-wave form;
 * 14 = the row GEMM's small form (16-row tiles).  Returns the record count.  Not thread-safe. */
int cone_prof_enable(int on);
int64_t cone_prof_collect(double* out, int64_t max_rec);

/* -------------------------------------------------------------------- test hooks
 * Thin entry points onto single kernels so that parity tests can bisect; not needed by a binding. */
/* A/B switches of ONE model handle for the parity tests (not thread-safe; set them before using the handle).
 * "dec_fold" (default 2): decoder cross-attention with the memory K/V projections folded into one kernel per layer, its
 *   two 256-channel contractions on the matrix cores (dec_cross_mfma.hip).  2 = the first layer (the same queries for
 *   every window, windows of at most 128 tokens, position tables) on the rows-once form -- every memory row is read from
 *   HBM once, kept in registers and handed to the second contraction through LDS a 64-channel quarter at a time, two
 *   workgroups per CU -- and every other launch on the two-read form; 3 = the two-read form everywhere; 5 = the rows-once
 *   form wherever it can run (per-window queries too: measured slower there, its query fold has no idle issue slots to
 *   hide in); 4 (opt-in) = windows of at most 110 tokens on the LDS-resident persistent form (one HBM read per row as
 *   well, but one workgroup per CU: measured 1.3 - 1.4 x slower); 1 = on the VALU (dec_cross.hip); 0 = two stacked K/V
 *   GEMMs + per-head attention.
 * "l0_gather" (default 1): with a cone_layer0 cache the first encoder layer's attention gathers q|k|v from the
 *   caches in its staging loads; 0 = a packing kernel writes them to the workspace first.  Bit-identical.
 * "pos_tables" (default 1): later encoder layers and the decoder keys take the position term from the static
 *   tables of cone_layer0; 0 = from an x + pos matrix written by the previous layer's GEMM epilogue.
 * "dec0_const" (default 1): the first decoder layer's self-attention block and cross-attention queries (tgt = 0:
 *   the same for every window) are computed for one window and replicated; 0 = for all windows.  Bit-identical.
 * "ffn_fused" (default 2): 1 = linear1 + ReLU + linear2 + residual + LayerNorm of every transformer layer as ONE kernel
 *   that keeps the (M, dim_feedforward) hidden rows on chip (ffn.hip); 2 = the attention output projection + residual +
 *   LayerNorm ahead of it in that kernel too (everything of a layer behind its attention: one launch, the layer's
 *   intermediate rows never leave the CU); 0 = GEMMs through (M, 256) / (M, ff) buffers.
 * "split_bf16" (default 0, OPT-IN): every transformer layer's tail (ffn_fused 2) on the bf16 matrix cores -- each fp32
 *   product as six partial products of three-piece bf16 operands (x = xh + xm + xl exactly), fp32 accumulation
 *   (ffn_split.hip).  Measured against float64: the same error as the exact-fp32 MFMA chain (1.5e-7 .. 2.4e-7 of
 *   sum |a b| vs 2.0e-7 .. 2.1e-7), at 1.65x its speed.  The default stays exact fp32.
 * "qkv_fused" (default 1): the next encoder layer's q | k | v projection is computed by the fused layer tail from the
 *   registers that hold its output rows: 1 = on the split_bf16 path (bit-identical to the separate launch), 2 = on the
 *   exact-fp32 path as well (measured neutral; another summation order than the GEMM launch), 0 = always its own launch.
 * "res_gather" (default 1, with ffn_fused 2 + l0_gather + pos_tables): the first encoder layer's residual rows are read
 *   by the fused layer tail straight from the projected clip / text rows through a row index; 0 = from a packed copy of
 *   the layer input written by a packing pass.  Bit-identical.
 * "rows_chain" (default 1): for few rows (the regime of the row GEMM's 16-row form) decoder.norm + class head + span MLP +
 *   span head of a decoder layer, and the adapter pair of the proposal matching (d = 256 features), run as ONE launch each
 *   with the rows on chip in between (rows_chain.h); 0 = the separate launches.  Bit-identical.
 * "ffn_spread" (default 1): a projecting layer tail of at most 1 024 rows (the decoder slot rows of a small batch, the token
 *   rows of a few windows) runs as four launches over single-wave workgroups (ffn_wide.hip, fs_*_kernel) instead of one CU
 *   per 16 rows walking the whole block, and a row GEMM of at most 1 024 rows without a LayerNorm epilogue as one wave per
 *   16 x 16 output tile; 0 = the wide form / the workgroup-per-16-rows form.  Bit-identical.
 * "gemm" (default 0 = by shape): tile family of every dense layer: 1 = register-staged 128x128 / 64x256 tiles,
 *   2 / 3 = 128x256 row-owning LDS-DMA tile with 4 waves x 32 rows (32x32x2) / 8 waves x 16 rows (16x16x4) -- all
 *   exact-fp32 fma chains per output element that walk k in different orders. */
int cone_model_set_option(cone_model* m, const char* name, int value);
/* C = epi((A [+ A2]) W^T + bias); flags: 1 relu, 2 add residual R, 4 LayerNorm(g,b) (N must be 256), 8 = keep a launch of a few
 * row groups on the workgroup-per-16-rows form (else: one wave per 16 x 16 output tile, same bits);
 * bits 8-9 force a tile family (1 = 128x128 / 64x256 register-staged tiles, 2 = 128x256 row-owning LDS-DMA tile
 * with 4 waves x 32 rows, 3 = the same tile with 8 waves x 16 rows, 0 = automatic = 3 where the shape allows).  Optional second output C2 = C + ADD (row tile only). */
int cone_test_gemm(const float* A, const float* A2, int a2_mod, const float* W, const float* bias,
                   const float* R, const float* ln_g, const float* ln_b, float* C, float* C2,
                   const float* ADD, int M, int N, int K, int flags, void* stream);
/* OUT = LayerNorm(X + W2 relu(W1 X + b1) + b2), the fused feed-forward block: X, OUT (M, 256), W1 (ff, 256), W2 (256, ff). */
int cone_test_ffn(const float* X, const float* W1, const float* b1, const float* W2, const float* b2,
                  const float* ln_g, const float* ln_b, float* OUT, int M, int ff, void* stream);
/* The same with the block input computed in the kernel: X = LayerNorm_p(R + A Wo^T + bo) -- the attention output projection,
 * its residual and norm (A, R (M, 256); Wo (256, 256)).  OUT may be R (in place). */
int cone_test_proj_ffn(const float* A, const float* Wo, const float* bo, const float* R, const float* pg, const float* pb,
                       const float* W1, const float* b1, const float* W2, const float* b2, const float* ln_g,
                       const float* ln_b, float* OUT, int M, int ff, void* stream);
/* cone_test_ffn on the bf16 matrix cores (ffn_split.hip): every fp32 product as six partial products of three-piece bf16
 * operands with fp32 accumulation.  img = scratch of cone_test_ffn_split_image_bytes(ff) bytes for the split weight image
 * (pack != 0: built from W1 / W2 first; 0: reused from an earlier call). */
size_t cone_test_ffn_split_image_bytes(int ff);
int cone_test_ffn_split(const float* X, const float* W1, const float* b1, const float* W2, const float* b2,
                        const float* ln_g, const float* ln_b, float* OUT, int M, int ff, void* img, int pack, void* stream);
/* C (M, N) = X (M, 256) W^T + bias (W (N, 256), N % 32 == 0) on the same split operands; img = scratch of
 * cone_test_rows_split_image_bytes(N) bytes. */
size_t cone_test_rows_split_image_bytes(int N);
int cone_test_rows_split(const float* X, const float* W, const float* bias, float* C, int M, int N, void* img, int pack,
                         void* stream);
/* The same block in its SPREAD form (ffn_wide.hip: four launches over single-wave workgroups; a handful of row groups only:
 * M <= CONE_FFN_SPREAD_MAX_ROWS); scratch of cone_test_proj_ffn_spread_scratch_bytes(ff) bytes.  Bit-identical rows. */
#define CONE_FFN_SPREAD_MAX_ROWS 1024
size_t cone_test_proj_ffn_spread_scratch_bytes(int ff);
int cone_test_proj_ffn_spread(const float* A, const float* Wo, const float* bo, const float* R, const float* pg, const float* pb,
                              const float* W1, const float* b1, const float* W2, const float* b2, const float* ln_g,
                              const float* ln_b, float* OUT, int M, int ff, void* scratch, void* stream);
/* cone_test_proj_ffn on the bf16 matrix cores; wo_img = scratch of cone_test_proj_split_image_bytes() bytes. */
size_t cone_test_proj_split_image_bytes(void);
int cone_test_proj_ffn_split(const float* A, const float* Wo, const float* bo, const float* R, const float* pg,
                             const float* pb, const float* W1, const float* b1, const float* W2, const float* b2,
                             const float* ln_g, const float* ln_b, float* OUT, int M, int ff, void* img, void* wo_img,
                             int pack, void* stream);
/* Encoder self-attention core (cone/transformer.py:239, 8 heads x 32) over windows of packed tokens off[b] .. off[b+1]:
 * mode 0: QKV (M, 768) = q | k | v rows that already carry the position term; mode 2: the same without it, the kernel adds
 * pos_qk[(vlen[b], p)] (R, 512) to q | k of clip token p; mode 1: q | k | v gathered from per-clip rows qkv_vid[vrow0[b] + p]
 * and per-text-token rows qkv_txt[trow0[b] + t] (+ the pos_qk row for clips).  OUT (M, 256) ahead of out_proj.
 * mode | 0x200 (windows of <= 144 tokens) runs the second formulation -- one wave per (window, head), K / V resident in
 * registers, no LDS: the same bits as the default workgroup-per-(window, head) kernel. */
int cone_test_enc_attn(int mode, const float* QKV, const float* qkv_vid, const float* qkv_txt, const float* pos_qk,
                       const int32_t* vrow0, const int32_t* vlen, const int32_t* trow0, const int32_t* off, float* OUT,
                       int B, int Lmax, int pos_zero_row /* index of an all-zero row of pos_qk (modes 1, 2) */, void* stream);
/* Fused decoder cross-attention of one layer (cone/transformer.py:308-311) with the memory K / V projections folded in:
 * DQ (B * nq, 256) projected queries (+ bias), X (M, 256) memory rows packed by off (B + 1), pos_rows / vlen = the sine
 * table and the clip count of each window (keys = memory + position row for clip tokens), Wk (256, 256) = rows
 * 256 .. 511 of in_proj_weight, WvT (256, 256) = W_v transposed, bv (256).  OUT (B * nq, 256) = attention output ahead
 * of out_proj.  variant = the "dec_fold" value (cone_model_set_option): 2 default policy, 3 two-read, 5 rows-once, 4 LDS-
 * resident, 1 VALU.  qk_slabs != NULL (matrix-core forms): every window has the SAME nq query rows, the folded operand is
 * built once into that scratch of cone_test_dec_cross_slab_floats() floats. */
int cone_test_dec_cross(const float* DQ, const float* X, const float* pos_rows, const int32_t* vlen, const int32_t* off,
                        const float* Wk, const float* WvT, const float* bv, float* OUT, int B, int nq, int Lmax,
                        int variant, float* qk_slabs, void* stream);
size_t cone_test_dec_cross_slab_floats(void);
int cone_test_layernorm(const float* x, const float* g, const float* b, float* out, int64_t n_rows,
                        int dim, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CONE_HIP_H */
