// Model handle, workspace carving and the launch sequences behind the C ABI (include/cone_hip.h).
//
// The window model runs on PACKED tokens: window b owns rows off[b] .. off[b+1] of every (M,*)
// activation matrix (its valid video clips, then its valid text tokens).  Padded keys never exist,
// so no masks are needed, and every dense layer is one tall GEMM over all windows of the batch.
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "rows_chain.h"

namespace cone {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct Linear { const float* w = nullptr; const float* b = nullptr; };
struct LNorm { const float* g = nullptr; const float* b = nullptr; };
struct Mha { const float* in_w = nullptr; const float* in_b = nullptr; Linear out; };
struct EncLayer { Mha sa; Linear l1, l2; LNorm n1, n2; };
struct DecLayer { Mha sa, ca; Linear l1, l2; LNorm n1, n2, n3; };

}  // namespace cone

struct cone_model {
    int d, heads, ff, n_enc, n_dec, nq, n_proj, dt, dv, dvm, has_adapter;     // dv: appearance clips (pre-filter, matching), dvm: motion clips (window model)
    float* arena = nullptr;  // every weight, one allocation
    cone::LNorm vproj_ln[CONE_MAX_PROJ], tproj_ln[CONE_MAX_PROJ];
    cone::Linear vproj[CONE_MAX_PROJ], tproj[CONE_MAX_PROJ];
    cone::EncLayer enc[CONE_MAX_LAYERS];
    cone::DecLayer dec[CONE_MAX_LAYERS];
    cone::LNorm dec_norm;
    const float* query_embed = nullptr;
    cone::Linear class_embed, span[3], saliency, adapter[2];
    const float* dim_t = nullptr;
    const float* txt_pos_emb = nullptr; int txt_pos_rows = 0; cone::LNorm txt_pos_ln;     // --use_txt_pos (NULL: off)
    int pre_norm = 0; cone::LNorm enc_norm;   // --pre_norm: normalize_before + the encoder's final LayerNorm
    // derived: the cross-attention K / V projections of all decoder layers stacked along N
    cone::Linear dec_k, dec_v;
    // derived: W_v^T of each decoder layer's cross-attention (256x256, [c][o]) for the fused cross-attention
    const float* dec_vT[CONE_MAX_LAYERS] = {};
    // derived: the first decoder layer's window-independent rows (tgt = 0, cone/transformer.py:66: its self-attention block
    // and its cross-attention queries depend on the checkpoint only): norm1 output (nq, 256) and the cross-attention query
    // projection (nq, 256), computed once at creation by the same kernels a step would run on ONE window's rows
    float* dec0_tgt1 = nullptr; float* dec0_dq = nullptr; float* dec0_scratch = nullptr;
    // derived: the slot-position term of the decoder's projections.  q = k = (tgt + query_embed) W^T + b is linear, so
    // (tgt + qe) W^T + b = tgt W^T + (qe W^T + b): per layer a (nq, 768) table [qe W_q^T + b_q | qe W_k^T + b_k | b_v] for the
    // self-attention in_proj (ONE N = 768 GEMM on tgt, the table as a row-periodic residual) and a (nq, 256) table
    // qe W_q^T + b_q for the cross-attention query projection -- the same move as the encoder's position tables
    float* dec_sa_tab[CONE_MAX_LAYERS] = {}; float* dec_ca_tab[CONE_MAX_LAYERS] = {};
    // derived (ABI 6; sized by cone_weights.table_max_v_l since ABI 8): the static position tables of this checkpoint (row
    // lv (lv - 1) / 2 + p: the same rows whatever the bound) -- every entry point runs the table path without the caller
    // building anything (cone_forward_windows, cone_forward_packed without a cone_layer0 or with caches only)
    float* tab_arena = nullptr; const float* tab_pos_rows = nullptr; const float* tab_pos_qk = nullptr; int tab_max_v_l = 0;
    // A/B switches of THIS handle (cone_model_set_option; parity tests only).  Defaults = the fast paths.
    int opt_dec_fold = 2;     // decoder memory K/V projections folded into the cross-attention kernel: 2 .. 5 = on the matrix
                              // cores (dec_cross_mfma.hip; which form: see launch_dec_cross_mfma), 1 = on the VALU
                              // (dec_cross.hip), 0 = K/V GEMMs + small_attn
    int opt_dec0_const = 1;   // first decoder layer's window-independent rows computed once and replicated
    int opt_l0_gather = 1;    // first encoder layer's attention gathers q|k|v from the layer-0 caches itself
    int opt_pos_tables = 1;   // later layers / decoder keys take the position term from the static tables
    int opt_gemm = 0;         // GEMM tile family forced for every dense layer (GEMM_AUTO = by shape)
    // derived (d = 256, ff % 32 == 0): the layer tails' weights split into bf16 pieces and laid out for ffn_split.hip
    char* split_img = nullptr;                          // one allocation: per layer [Wo image | FFN image]
    const void* enc_wo_img[CONE_MAX_LAYERS] = {}; const void* enc_ffn_img[CONE_MAX_LAYERS] = {};
    const void* enc_qkv_img[CONE_MAX_LAYERS] = {};
    const void* dec_wo_img[CONE_MAX_LAYERS] = {}; const void* dec_ffn_img[CONE_MAX_LAYERS] = {};
    int opt_qkv_fused = 1;    // the next encoder layer's q | k | v projection inside the fused layer tail (same launch): 1 = on
                              // the split_bf16 path (-0.3 ms), 2 = on the exact-fp32 path too (neutral), 0 = own launch
    int opt_split_bf16 = 0;   // OPT-IN: layer tails on the bf16 matrix cores (six partial products of three-piece operands,
                              // fp32 accumulation: fp32-MFMA accuracy); 0 = exact-fp32 MFMA (default)
    int opt_res_gather = 1;   // first encoder layer's residual rows gathered by the fused layer tail (no packed input copy)
    int opt_spread = 1;       // <= 16 row groups in a decoder tail: the spread form (four launches over single-wave workgroups)
    int opt_chain = 1;        // few rows: decoder.norm + class head + span MLP + span head, and the adapter pair of the proposal
                              // matching, as ONE launch each (rows_chain.h: the same arithmetic, bit-identical); 0 = separate launches
    int opt_ffn_fused = 2;    // 1: linear1 + ReLU + linear2 + residual + LayerNorm as one kernel (ffn.hip); 2: the attention
                              // output projection + residual + LayerNorm ahead of it in the same kernel as well; 0: GEMMs
};

namespace cone {

// ------------------------------------------------------------------------------ weights
struct ArenaBuilder {
    struct Item { const float* src; size_t n; const float** dst; };
    std::vector<Item> items;
    size_t total = 0;
    void add(const float* src, size_t n, const float** dst) {
        items.push_back({src, n, dst});
        total += align_up(n, 64);
    }
};

static int dec0_constants(cone_model* m, hipStream_t s);
static int build_pos_tables(const cone_model* m, int max_v_l, float* pos_rows, float* pos_qk, hipStream_t s);
static int64_t pos_table_rows(int max_v_l) { return (int64_t)max_v_l * (max_v_l + 1) / 2 + 1; }

static int build_model(const cone_weights* w, cone_model** out) {
    CONE_REQUIRE(w && out, "model_create: null argument");
    CONE_REQUIRE(w->hidden_dim == 256 && w->nheads == 8,
                 "model_create: unsupported model shape hidden_dim=%d nheads=%d -- the attention / layer-tail kernels of this "
                 "build are instantiated for hidden_dim 256 with 8 heads (head_dim 32: every shipped CONE configuration, "
                 "cone/config.py:101-104); also required: dim_feedforward a multiple of 128, num_queries <= 16, feature dims "
                 "multiples of 32 up to 1024, at most 256 tokens (clips + words) per window", w->hidden_dim, w->nheads);
    CONE_REQUIRE(w->dim_ff % 128 == 0 && w->dim_ff >= 128, "model_create: dim_feedforward=%d must be a multiple of 128", w->dim_ff);
    CONE_REQUIRE(w->enc_layers >= 1 && w->enc_layers <= CONE_MAX_LAYERS && w->dec_layers >= 1 &&
                     w->dec_layers <= CONE_MAX_LAYERS, "model_create: layer counts out of range");
    CONE_REQUIRE(w->num_queries >= 1 && w->num_queries <= 16, "model_create: num_queries=%d not in [1,16]", w->num_queries);
    CONE_REQUIRE(w->n_input_proj >= 1 && w->n_input_proj <= CONE_MAX_PROJ, "model_create: n_input_proj out of range");
    CONE_REQUIRE(w->t_dim % 32 == 0 && w->v_dim % 32 == 0 && w->v_motion_dim % 32 == 0 && w->t_dim <= 1024 && w->v_dim <= 1024 &&
                     w->v_motion_dim <= 1024 && w->t_dim > 0 && w->v_dim > 0 && w->v_motion_dim > 0,
                 "model_create: feature dims must be multiples of 32 and <= 1024 (t=%d v_appear=%d v_motion=%d)", w->t_dim,
                 w->v_dim, w->v_motion_dim);
    cone_model* m = new cone_model();
    m->d = 256; m->heads = 8; m->ff = w->dim_ff; m->n_enc = w->enc_layers; m->n_dec = w->dec_layers;
    m->nq = w->num_queries; m->n_proj = w->n_input_proj; m->dt = w->t_dim; m->dv = w->v_dim; m->dvm = w->v_motion_dim;
    m->has_adapter = w->has_adapter;
    const size_t d = 256, ff = m->ff;
    ArenaBuilder ab;
    auto lin = [&](const cone_linear_w& s, size_t nout, size_t nin, Linear& dst) {
        ab.add(s.w, nout * nin, &dst.w);
        ab.add(s.b, nout, &dst.b);
    };
    auto ln = [&](const cone_ln_w& s, size_t n, LNorm& dst) { ab.add(s.g, n, &dst.g); ab.add(s.b, n, &dst.b); };
    auto mha = [&](const cone_mha_w& s, Mha& dst) {
        ab.add(s.in_proj_w, 3 * d * d, &dst.in_w);
        ab.add(s.in_proj_b, 3 * d, &dst.in_b);
        lin(s.out_proj, d, d, dst.out);
    };
    for (int i = 0; i < m->n_proj; ++i) {
        ln(w->vid_proj_ln[i], i == 0 ? m->dvm : d, m->vproj_ln[i]);
        lin(w->vid_proj[i], d, i == 0 ? m->dvm : d, m->vproj[i]);
        ln(w->txt_proj_ln[i], i == 0 ? m->dt : d, m->tproj_ln[i]);
        lin(w->txt_proj[i], d, i == 0 ? m->dt : d, m->tproj[i]);
    }
    for (int i = 0; i < m->n_enc; ++i) {
        mha(w->enc[i].self_attn, m->enc[i].sa);
        lin(w->enc[i].linear1, ff, d, m->enc[i].l1);
        lin(w->enc[i].linear2, d, ff, m->enc[i].l2);
        ln(w->enc[i].norm1, d, m->enc[i].n1);
        ln(w->enc[i].norm2, d, m->enc[i].n2);
    }
    for (int i = 0; i < m->n_dec; ++i) {
        mha(w->dec[i].self_attn, m->dec[i].sa);
        mha(w->dec[i].cross_attn, m->dec[i].ca);
        lin(w->dec[i].linear1, ff, d, m->dec[i].l1);
        lin(w->dec[i].linear2, d, ff, m->dec[i].l2);
        ln(w->dec[i].norm1, d, m->dec[i].n1);
        ln(w->dec[i].norm2, d, m->dec[i].n2);
        ln(w->dec[i].norm3, d, m->dec[i].n3);
    }
    ln(w->dec_norm, d, m->dec_norm);
    ab.add(w->query_embed, (size_t)m->nq * d, &m->query_embed);
    lin(w->class_embed, 2, d, m->class_embed);
    lin(w->span_embed[0], d, d, m->span[0]);
    lin(w->span_embed[1], d, d, m->span[1]);
    lin(w->span_embed[2], 2, d, m->span[2]);
    lin(w->saliency_proj, 1, d, m->saliency);
    if (m->has_adapter) {
        lin(w->adapter[0], d, m->dv, m->adapter[0]);
        lin(w->adapter[1], m->dv, d, m->adapter[1]);
    }
    ab.add(w->pos_dim_t, d, &m->dim_t);
    if (w->txt_pos_embed) {     // --use_txt_pos
        if (w->txt_pos_rows < 1 || w->txt_pos_rows > 4096 || !w->txt_pos_ln.g || !w->txt_pos_ln.b) {
            delete m;
            set_error("model_create: txt_pos_embed needs txt_pos_rows in [1, 4096] (got %d) and its LayerNorm", w->txt_pos_rows);
            return CONE_E_INVALID;
        }
        m->txt_pos_rows = w->txt_pos_rows;
        ab.add(w->txt_pos_embed, (size_t)w->txt_pos_rows * d, &m->txt_pos_emb);
        ln(w->txt_pos_ln, d, m->txt_pos_ln);
    }
    if (w->pre_norm) {          // --pre_norm
        m->pre_norm = 1;
        ln(w->enc_norm, d, m->enc_norm);
    }
    for (auto& it : ab.items)
        if (!it.src) {
            delete m;
            set_error("model_create: a required weight pointer is null");
            return CONE_E_INVALID;
        }
    const size_t dec0_floats = (size_t)m->nq * (256 * 2 + 256 + 768 + 256) + 6 * 64 +
                               (size_t)m->n_dec * (align_up((size_t)m->nq * 768, 64) + align_up((size_t)m->nq * 256, 64));
    const size_t stacked = (size_t)m->n_dec * (d * d + d) * 2 + 4 * 64 + (size_t)m->n_dec * d * d + dec0_floats;
    hipError_t e = hipMalloc((void**)&m->arena, (ab.total + stacked) * sizeof(float));
    if (e != hipSuccess) {
        delete m;
        set_error("model_create: hipMalloc of %zu bytes failed: %s", (ab.total + stacked) * sizeof(float),
                  hipGetErrorString(e));
        return CONE_E_HIP;
    }
    size_t cur = 0;
    for (auto& it : ab.items) {
        e = hipMemcpy(m->arena + cur, it.src, it.n * sizeof(float), hipMemcpyDefault);
        if (e != hipSuccess) {
            (void)hipFree(m->arena);
            delete m;
            set_error("model_create: weight copy failed: %s", hipGetErrorString(e));
            return CONE_E_HIP;
        }
        *it.dst = m->arena + cur;
        cur += align_up(it.n, 64);
    }
    // stack W_k / W_v (rows [d:2d] / [2d:3d] of each layer's multihead_attn.in_proj) along N
    float* kw = m->arena + cur; cur += align_up((size_t)m->n_dec * d * d, 64);
    float* kb = m->arena + cur; cur += align_up((size_t)m->n_dec * d, 64);
    float* vw = m->arena + cur; cur += align_up((size_t)m->n_dec * d * d, 64);
    float* vb = m->arena + cur; cur += align_up((size_t)m->n_dec * d, 64);
    for (int i = 0; i < m->n_dec; ++i) {
        const float* iw = m->dec[i].ca.in_w;
        const float* ib = m->dec[i].ca.in_b;
        (void)hipMemcpy(kw + (size_t)i * d * d, iw + d * d, d * d * sizeof(float), hipMemcpyDeviceToDevice);
        (void)hipMemcpy(kb + (size_t)i * d, ib + d, d * sizeof(float), hipMemcpyDeviceToDevice);
        (void)hipMemcpy(vw + (size_t)i * d * d, iw + 2 * d * d, d * d * sizeof(float), hipMemcpyDeviceToDevice);
        e = hipMemcpy(vb + (size_t)i * d, ib + 2 * d, d * sizeof(float), hipMemcpyDeviceToDevice);
    }
    if (e != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(m->arena);
        delete m;
        set_error("model_create: stacking decoder K/V weights failed");
        return CONE_E_HIP;
    }
    m->dec_k = {kw, kb};
    m->dec_v = {vw, vb};
    {   // W_v^T per decoder layer (host transpose; 256 KiB each, once per model)
        std::vector<float> h(d * d), ht(d * d);
        for (int i = 0; i < m->n_dec; ++i) {
            float* dst = m->arena + cur; cur += d * d;
            e = hipMemcpy(h.data(), m->dec[i].ca.in_w + 2 * d * d, d * d * sizeof(float), hipMemcpyDeviceToHost);
            for (size_t o = 0; o < d; ++o)
                for (size_t c = 0; c < d; ++c) ht[c * d + o] = h[o * d + c];
            if (e == hipSuccess) e = hipMemcpy(dst, ht.data(), d * d * sizeof(float), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                (void)hipFree(m->arena);
                delete m;
                set_error("model_create: transposing decoder V weights failed: %s", hipGetErrorString(e));
                return CONE_E_HIP;
            }
            m->dec_vT[i] = dst;
        }
    }
    m->dec0_tgt1 = m->arena + cur; cur += align_up((size_t)m->nq * 256, 64);
    m->dec0_dq = m->arena + cur; cur += align_up((size_t)m->nq * 256, 64);
    m->dec0_scratch = m->arena + cur; cur += align_up((size_t)m->nq * (256 + 768 + 256), 64);
    for (int i = 0; i < m->n_dec; ++i) {
        m->dec_sa_tab[i] = m->arena + cur; cur += align_up((size_t)m->nq * 768, 64);
        m->dec_ca_tab[i] = m->arena + cur; cur += align_up((size_t)m->nq * 256, 64);
    }
    if (d == 256 && ffn_split_supported(m->ff)) {   // split-bf16 images of every layer tail (13 MB at ff = 1024; opt-in path)
        const size_t per = ffn_split_proj_image_bytes() + ffn_split_image_bytes(m->ff);
        const size_t qkv = rows256_split_image_bytes(768);
        e = hipMalloc((void**)&m->split_img, per * (size_t)(m->n_enc + m->n_dec) + qkv * (size_t)m->n_enc);
        char* ip = m->split_img;
        int rc = 0;
        for (int l = 0; l < m->n_enc && e == hipSuccess && rc == 0; ++l) {     // q | k | v projections (768 x 256)
            rc = launch_ffn_split_pack(m->enc[l].sa.in_w, nullptr, 768, ip, nullptr);
            m->enc_qkv_img[l] = ip;
            ip += qkv;
        }
        for (int i = 0; i < m->n_enc + m->n_dec && e == hipSuccess && rc == 0; ++i) {
            const bool enc = i < m->n_enc;
            const int l = enc ? i : i - m->n_enc;
            const float* wo = enc ? m->enc[l].sa.out.w : m->dec[l].ca.out.w;
            const Linear& l1 = enc ? m->enc[l].l1 : m->dec[l].l1;
            const Linear& l2 = enc ? m->enc[l].l2 : m->dec[l].l2;
            rc = launch_ffn_split_pack(wo, nullptr, 256, ip, nullptr);
            if (rc == 0) rc = launch_ffn_split_pack(l1.w, l2.w, m->ff, ip + ffn_split_proj_image_bytes(), nullptr);
            (enc ? m->enc_wo_img : m->dec_wo_img)[l] = ip;
            (enc ? m->enc_ffn_img : m->dec_ffn_img)[l] = ip + ffn_split_proj_image_bytes();
            ip += per;
        }
        if (e != hipSuccess || rc != 0 || hipDeviceSynchronize() != hipSuccess) {
            if (m->split_img) (void)hipFree(m->split_img);
            (void)hipFree(m->arena);
            delete m;
            if (rc == 0) set_error("model_create: building the split-bf16 weight images failed");
            return CONE_E_HIP;
        }
    }
    if (dec0_constants(m, nullptr) != 0 || hipDeviceSynchronize() != hipSuccess) {
        if (m->split_img) (void)hipFree(m->split_img);
        (void)hipFree(m->arena);
        delete m;
        return CONE_E_HIP;
    }
    {   // the handle's own position tables, for the window lengths this checkpoint is built for (ABI 8: cone_weights.
        // table_max_v_l; rows (256 + 512 per encoder layer) floats each: 21 MB at 90 clips, 167 MB at 255 with two layers)
        const int tab_l = (w->table_max_v_l >= 1 && w->table_max_v_l <= CONE_TABLE_MAX_V_L) ? w->table_max_v_l : CONE_TABLE_MAX_V_L;
        const size_t rows = (size_t)pos_table_rows(tab_l);
        e = hipMalloc((void**)&m->tab_arena, rows * (256 + 512 * (size_t)m->n_enc) * sizeof(float));
        int rc = e == hipSuccess ? 0 : CONE_E_HIP;
        if (rc == 0) rc = build_pos_tables(m, tab_l, m->tab_arena, m->tab_arena + rows * 256, nullptr);
        if (rc != 0 || hipDeviceSynchronize() != hipSuccess) {
            if (e != hipSuccess) set_error("model_create: hipMalloc of the position tables failed: %s", hipGetErrorString(e));
            if (m->tab_arena) (void)hipFree(m->tab_arena);
            if (m->split_img) (void)hipFree(m->split_img);
            (void)hipFree(m->arena);
            delete m;
            return CONE_E_HIP;
        }
        m->tab_pos_rows = m->tab_arena; m->tab_pos_qk = m->tab_arena + rows * 256; m->tab_max_v_l = tab_l;
    }
    *out = m;
    return 0;
}

// ------------------------------------------------------------------------------ workspace
struct Carver {
    char* base; size_t cap, cur = 0; bool ok = true;
    Carver(void* p, size_t n) : base((char*)p), cap(n) {}
    template <typename T> T* take(size_t n) {
        const size_t bytes = align_up(n * sizeof(T), 256);
        T* r = (T*)(base + cur);
        cur += bytes;
        if (cur > cap) ok = false;
        return r;
    }
};

static GemmArgs G(const cone_model* m, const float* A, int lda, const float* W, int ldw, const float* bias, float* C,
                  int ldc, int M, const int* M_dev, int N, int K, int flags = 0) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc;
    g.M = M; g.M_dev = M_dev; g.N = N; g.K = K; g.flags = flags;
    g.variant = m ? m->opt_gemm : GEMM_AUTO;
    if (m && !m->opt_spread) g.flags |= GEMM_NO_SPREAD;        // option ffn_spread = 0: no spread forms anywhere
    return g;
}

#define RUN(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

// Per-checkpoint constants of the decoder (cone/transformer.py:296-311).  (1) The slot-position tables of every layer (see
// cone_model).  (2) The first layer on tgt = 0 (:66): its self-attention over the nq slots (q | k | v = the table rows), out_proj
// + norm1, and its cross-attention queries -- no window enters, so the nq rows are constants too.  Computed with the kernels
// and the row count (nq) a step uses for one window, hence the same bits as the per-window path (dec0_const = 0).
static int dec0_constants(cone_model* m, hipStream_t s) {
    const int nq = m->nq;
    float* TGT = m->dec0_scratch;                       // (nq, 256) zeros
    float* QKV = TGT + (size_t)nq * 256;                // (nq, 768)
    float* DATT = QKV + (size_t)nq * 768;               // (nq, 256)
    CONE_CHECK_HIP(hipMemsetAsync(TGT, 0, (size_t)nq * 256 * sizeof(float), s));
    for (int l = 0; l < m->n_dec; ++l) {
        const DecLayer& dl = m->dec[l];
        // [qe W_q^T + b_q | qe W_k^T + b_k]  and  b_v rows (0 W_v^T + b_v)
        RUN(launch_gemm(G(m, m->query_embed, 256, dl.sa.in_w, 256, dl.sa.in_b, m->dec_sa_tab[l], 768, nq, nullptr, 512, 256), s));
        RUN(launch_gemm(G(m, TGT, 256, dl.sa.in_w + 512 * 256, 256, dl.sa.in_b + 512, m->dec_sa_tab[l] + 512, 768, nq, nullptr, 256, 256), s));
        RUN(launch_gemm(G(m, m->query_embed, 256, dl.ca.in_w, 256, dl.ca.in_b, m->dec_ca_tab[l], 256, nq, nullptr, 256, 256), s));
    }
    const DecLayer& d0 = m->dec[0];
    GemmArgs g = G(m, TGT, 256, d0.sa.in_w, 256, nullptr, QKV, 768, nq, nullptr, 768, 256, EPI_RESIDUAL);
    g.R = m->dec_sa_tab[0]; g.ldr = 768; g.r_mod = nq;
    RUN(launch_gemm(g, s));
    RUN(launch_small_attn(QKV, 768, QKV + 256, 768, QKV + 512, 768, DATT, 256, nullptr, 1, nq, nq, s));
    g = G(m, DATT, 256, d0.sa.out.w, 256, d0.sa.out.b, m->dec0_tgt1, 256, nq, nullptr, 256, 256, EPI_RESIDUAL | EPI_LN);
    g.R = TGT; g.ldr = 256; g.ln_g = d0.n1.g; g.ln_b = d0.n1.b;
    RUN(launch_gemm(g, s));
    g = G(m, m->dec0_tgt1, 256, d0.ca.in_w, 256, nullptr, m->dec0_dq, 256, nq, nullptr, 256, 256, EPI_RESIDUAL);
    g.R = m->dec_ca_tab[0]; g.ldr = 256; g.r_mod = nq;
    RUN(launch_gemm(g, s));
    return 0;
}

// rows (lv, p), p < lv <= max_v_l, then ONE all-zero row: the position term of a text token (cone/model.py:106), which the
// encoder attention adds unconditionally instead of branching on the token kind
static int build_pos_tables(const cone_model* m, int max_v_l, float* pos_rows, float* pos_qk, hipStream_t s) {
    const int64_t rows = pos_table_rows(max_v_l);
    RUN(launch_pos_rows(m->dim_t, max_v_l, pos_rows, s));
    CONE_CHECK_HIP(hipMemsetAsync(pos_rows + (size_t)(rows - 1) * 256, 0, 256 * sizeof(float), s));    // the zero row
    // pos W_q^T | pos W_k^T of every encoder layer, no bias (the bias travels with the clip / token rows): zero row -> zeros
    for (int l = 0; l < m->n_enc; ++l)
        RUN(launch_gemm(G(m, pos_rows, 256, m->enc[l].sa.in_w, 256, nullptr, pos_qk + (size_t)l * rows * 512, 512,
                          (int)rows, nullptr, 512, 256), s));
    return 0;
}

// First encoder layer's in_proj hoisted out of the window loop: q | k | v = in_proj(x) once per projected clip / text token
// (post-norm, cone/transformer.py:237-239), resp. in_proj(norm1(x)) (--pre_norm, :250-252; tmp = n rows of scratch).
static int layer0_rows(const cone_model* m, const float* rows, int n, const int* n_dev, float* qkv, float* tmp, hipStream_t s) {
    const float* a = rows;
    if (m->pre_norm) {
        RUN(launch_layernorm(rows, 256, m->enc[0].n1.g, m->enc[0].n1.b, tmp, 256, n, n_dev, 256, s));
        a = tmp;
    }
    return launch_gemm(G(m, a, 256, m->enc[0].sa.in_w, 256, m->enc[0].sa.in_b, qkv, 768, n, n_dev, 768, 256), s);
}

// input_{vid,txt}_proj: LN -> Linear -> ReLU (all but last) with the next LN fused into the GEMM epilogue.
static size_t project_ws_bytes(const cone_model* m, int which, int64_t n) {
    const size_t din = which == 0 ? m->dvm : m->dt;
    return align_up(n * din * 4, 256) + 2 * align_up(n * 256 * 4, 256);
}
// src_row != null: row i of the projection reads row src_row[i] of x (the valid rows of a zero-padded batch, compacted by the
// first LayerNorm's loads); n_dev != null: only the first *n_dev rows exist (device-side count, n bounds it)
static int project_tokens(const cone_model* m, int which, const float* x, int64_t n, float* out, void* ws,
                          size_t ws_bytes, hipStream_t s, const int* src_row = nullptr, const int* n_dev = nullptr) {
    CONE_REQUIRE(n < (1ll << 31), "project: too many rows");
    const int din = which == 0 ? m->dvm : m->dt;
    const LNorm* lns = which == 0 ? m->vproj_ln : m->tproj_ln;
    const Linear* lin = which == 0 ? m->vproj : m->tproj;
    Carver c(ws, ws_bytes);
    float* t0 = c.take<float>((size_t)n * din);
    float* ta = c.take<float>((size_t)n * 256);
    float* tb = c.take<float>((size_t)n * 256);
    if (!c.ok) { set_error("project: workspace too small (%zu < %zu)", ws_bytes, c.cur); return CONE_E_WORKSPACE; }
    RUN(launch_layernorm(x, din, lns[0].g, lns[0].b, t0, din, n, n_dev, din, s, src_row));
    const float* cur = t0;
    int K = din;
    for (int i = 0; i < m->n_proj; ++i) {
        const bool last = i == m->n_proj - 1;
        float* dst = last ? out : (cur == ta ? tb : ta);
        GemmArgs g = G(m, cur, K, lin[i].w, K, lin[i].b, dst, 256, (int)n, n_dev, 256, K, last ? 0 : EPI_RELU);
        if (!last) { g.flags |= EPI_LN; g.ln_g = lns[i + 1].g; g.ln_b = lns[i + 1].b; }
        RUN(launch_gemm(g, s));
        cur = dst;
        K = 256;
    }
    return 0;
}

// ------------------------------------------------------------------------------ packed forward
// Workspace of one batch of B windows (M = B * Lmax token rows at most).  Two layouts:
//   table path (layer-0 cache + position tables; the eval driver): X, X1 and ONE (M, ff) region that holds, in
//     turn, the layer's q|k|v (M, 768) + attention output (M, 256) and then the FFN hidden rows -- q|k|v are dead
//     once the attention has run, its output once out_proj has, and FFN1 writes only after that: 4 * (2 * 256 +
//     max(ff, 1024)) = 6 KiB per token row at ff = 1024;
//   general path (--use_txt_pos, and the A/B switches of the parity tests): additionally POS and XP = X + POS, and the
//     stacked decoder K / V rows when the cross-attention fold is off.
struct FwdBuffers {
    int* off; int* RIDX;
    float *X, *POS, *XP, *QKV, *ATT, *X1, *H, *KD, *VD;
    float *TGT, *TGT1, *TGT2, *DQK, *DV, *DATT, *DQ, *DH, *HS, *S1, *S2, *LG, *SP, *QKS, *SPR;
};
struct FwdPlan { bool tables, fold, dec_xp = false; };
// What a call runs on: the caller's cone_layer0 with the handle's position tables filled in where it brings none (ABI 6: a
// NULL cone_layer0, or one with the row caches only, still takes the table path), or nothing for --use_txt_pos (the general
// path: the caches / tables assume a zero text position term) and for windows longer than the handle's tables cover.
static const cone_layer0* effective_l0(const cone_model* m, const cone_layer0* l0, int Lv_max, cone_layer0* eff,
                                       bool own_txt = false) {
    // --use_txt_pos: the table path needs the tokens' own position rows (cone_layer0_text_positions) -- handed over in l0, or
    // (own_txt: the padded entry) built by the caller itself; without them the general path
    if (m->txt_pos_emb && !(own_txt || (l0 && l0->txt_pos && l0->txt_pos_qk))) return nullptr;
    *eff = l0 ? *l0 : cone_layer0{};
    if (!m->txt_pos_emb) eff->txt_pos = eff->txt_pos_qk = nullptr;
    if (!eff->qkv_vid || !eff->qkv_txt) eff->qkv_vid = eff->qkv_txt = nullptr;
    if (!eff->pos_rows || !eff->pos_qk) {                   // no (complete) tables of the caller's: the handle's
        if (Lv_max > m->tab_max_v_l) return nullptr;
        eff->pos_rows = m->tab_pos_rows; eff->pos_qk = m->tab_pos_qk; eff->max_v_l = m->tab_max_v_l;
    }
    return eff;
}
// have_tables / caches: what effective_l0 resolved to (position tables available; first-layer row caches given)
static FwdPlan plan_for(const cone_model* m, bool have_tables, bool caches, int Lmax) {
    FwdPlan p;
    p.tables = have_tables && m->opt_pos_tables && (!caches || m->opt_l0_gather);
    // --use_txt_pos on the table path: the decoder's keys memory + pos are written once behind the encoder (clip rows from the
    // table, text rows from the tokens' own position rows) and the cross-attention runs its x + pos form on them
    p.dec_xp = p.tables && m->txt_pos_emb != nullptr;
    p.fold = m->opt_dec_fold >= 2 ? dec_cross_mfma_supported(m->nq, Lmax, p.tables && !p.dec_xp)
                                  : (m->opt_dec_fold == 1 && dec_cross_supported(m->nq, Lmax));
    if (m->pre_norm) {  // the fused pre-norm path needs all of: tables, the fused layer tail, the matrix-core fold; else the
                        // general pre-norm path (plain LayerNorm / GEMM / attention launches)
        const bool fused = p.tables && p.fold && m->opt_dec_fold >= 2 && m->opt_ffn_fused >= 2 && ffn_fused_supported(m->ff);
        p.tables = p.fold = fused;
        if (!fused) p.dec_xp = false;
        return p;
    }
    // the unfolded decoder projects its keys from memory + pos rows: on the table path that matrix is written once behind the
    // encoder (launch_add_pos_rows) -- slot counts other than 5 keep the encoder's fast path.  The fold switched off BY OPTION
    // (parity tests) keeps meaning the whole general path
    if (!p.fold && !(m->opt_dec_fold && m->nq != 5)) p.tables = false;
    if (!p.tables) p.dec_xp = false;
    return p;
}
static FwdPlan plan_of(const cone_model* m, const cone_layer0* l0 /* effective_l0 */, int Lmax) {
    return plan_for(m, l0 && l0->pos_rows && l0->pos_qk, l0 && l0->qkv_vid, Lmax);
}
// effective_l0 + plan_of; a --use_txt_pos model that cannot take the table path (an A/B switch, a window too long for the
// x + pos form of the fold) drops to the WHOLE general path: its row caches do not carry the text position term
static const cone_layer0* resolve_l0(const cone_model* m, const cone_layer0* l0, int Lv_max, int Lmax, cone_layer0* eff,
                                     FwdPlan* plan) {
    const cone_layer0* e = effective_l0(m, l0, Lv_max, eff);
    *plan = plan_of(m, e, Lmax);
    if (m->txt_pos_emb && e && !plan->tables) { e = nullptr; *plan = plan_of(m, nullptr, Lmax); }
    return e;
}
static void carve_fwd(const cone_model* m, Carver& c, int B, int Lmax, const FwdPlan& p, FwdBuffers& f) {
    const size_t M = (size_t)B * Lmax, T = (size_t)B * m->nq, nd = m->n_dec;
    const size_t wide = m->ff > 1024 ? m->ff : 1024;
    f.off = c.take<int>(B + 1);
    f.RIDX = m->pre_norm ? c.take<int>(M) : nullptr;    // (post-norm keeps the row index in the X1 region, unused on that path)
    f.X = c.take<float>(M * 256); f.X1 = c.take<float>(M * 256);
    f.H = c.take<float>(M * wide);
    f.QKV = f.H; f.ATT = f.H + M * 768;            // aliases of the FFN hidden region (see above)
    f.POS = f.XP = f.KD = f.VD = nullptr;
    if (!p.tables) {
        f.POS = c.take<float>(M * 256); f.XP = c.take<float>(M * 256);
        f.QKV = c.take<float>(M * 768); f.ATT = c.take<float>(M * 256);
    }
    if (!p.fold) { f.KD = c.take<float>(M * 256 * nd); f.VD = c.take<float>(M * 256 * nd); }
    if ((!p.fold || p.dec_xp) && p.tables) f.XP = c.take<float>(M * 256);   // memory + pos for the unfolded decoder / the
                                                                            // x + pos form of the fold (launch_add_pos_rows)
    f.TGT = c.take<float>(T * 256); f.TGT1 = c.take<float>(T * 256); f.TGT2 = c.take<float>(T * 256);
    f.DQK = c.take<float>(T * 768); f.DV = nullptr; f.DATT = c.take<float>(T * 256);      // DQK: the slots' q | k | v
    f.DQ = c.take<float>(T * 256); f.DH = c.take<float>(T * m->ff);
    f.HS = c.take<float>(nd * T * 256); f.S1 = c.take<float>(nd * T * 256); f.S2 = c.take<float>(nd * T * 256);
    f.LG = c.take<float>(nd * T * 2); f.SP = c.take<float>(nd * T * 2);
    f.QKS = c.take<float>(dec_cross_mfma_slab_floats());
    f.SPR = ffn_spread_supported((int)T, m->ff) || ffn_spread_supported((int)M, m->ff)                       // the spread tail's rows
                ? c.take<float>(ffn_spread_scratch_floats(m->ff)) : nullptr;
}
static size_t fwd_ws_bytes(const cone_model* m, int B, int Lmax, const FwdPlan& p) {
    Carver c(nullptr, ~(size_t)0);
    FwdBuffers f;
    carve_fwd(m, c, B, Lmax, p, f);
    return c.cur;
}

// --pre_norm (cone/config.py:120 -> normalize_before, cone/transformer.py:19-36): every layer normalises its INPUT
// (forward_pre, :248-260 / :319-342), the residual stream stays un-normalised, and the encoder ends with its own LayerNorm.
// Off in every shipped configuration: built from the plain blocks (LayerNorm kernel, row GEMMs with residual epilogue, the
// packed encoder attention, the small decoder attentions) -- no fused tails, no caches.
static int forward_packed_prenorm(const cone_model* m, const float* vproj, const int* vrow0, const int* vlen,
                                  const float* tproj, const int* trow0, const int* qlen, int B, int Lv_max, int Lq_max,
                                  float* logits, float* spans, float* saliency, const cone_taps* taps, FwdBuffers& f,
                                  hipStream_t s) {
    const int Lmax = Lv_max + Lq_max, Mmax = B * Lmax, T = B * m->nq, nd = m->n_dec, ff = m->ff;
    const int* Mdev = f.off + B;
    RUN(launch_scan_lengths(vlen, qlen, B, f.off, s));
    RUN(launch_pack_pos(vproj, vrow0, vlen, tproj, trow0, qlen, f.off, m->dim_t, f.X, f.POS, f.XP, B, Lmax, s,
                        m->txt_pos_emb, m->txt_pos_ln.g, m->txt_pos_ln.b));
    GemmArgs g;
    for (int l = 0; l < m->n_enc; ++l) {
        const EncLayer& e = m->enc[l];
        RUN(launch_layernorm(f.X, 256, e.n1.g, e.n1.b, f.X1, 256, Mmax, Mdev, 256, s));                 // src2 = norm1(src)
        float* QK = f.QKV; float* V = f.QKV + (size_t)Mmax * 512;
        g = G(m, f.X1, 256, e.sa.in_w, 256, e.sa.in_b, QK, 512, Mmax, Mdev, 512, 256);                  // q | k = (src2 + pos) W^T
        g.A2 = f.POS; g.lda2 = 256;
        RUN(launch_gemm(g, s));
        RUN(launch_gemm(G(m, f.X1, 256, e.sa.in_w + 512 * 256, 256, e.sa.in_b + 512, V, 256, Mmax, Mdev, 256, 256), s));
        AttnSrc src{};
        src.Q = QK; src.K = QK + 256; src.V = V; src.ldq = src.ldk = 512; src.ldv = 256;
        RUN(launch_enc_attn(ATTN_PACKED, src, f.ATT, f.off, B, Lmax, s));
        g = G(m, f.ATT, 256, e.sa.out.w, 256, e.sa.out.b, f.X, 256, Mmax, Mdev, 256, 256, EPI_RESIDUAL);
        g.R = f.X; g.ldr = 256;                                                                        // src += attn Wo^T + bo
        RUN(launch_gemm(g, s));
        RUN(launch_layernorm(f.X, 256, e.n2.g, e.n2.b, f.X1, 256, Mmax, Mdev, 256, s));                 // src2 = norm2(src)
        RUN(launch_gemm(G(m, f.X1, 256, e.l1.w, 256, e.l1.b, f.H, ff, Mmax, Mdev, ff, 256, EPI_RELU), s));
        g = G(m, f.H, ff, e.l2.w, ff, e.l2.b, f.X, 256, Mmax, Mdev, 256, ff, EPI_RESIDUAL);
        g.R = f.X; g.ldr = 256;                                                                        // src += ffn(src2)
        RUN(launch_gemm(g, s));
    }
    RUN(launch_layernorm(f.X, 256, m->enc_norm.g, m->enc_norm.b, f.X1, 256, Mmax, Mdev, 256, s));       // memory = encoder.norm(src)
    const float* MEM = f.X1;
    // decoder keys / values of all layers: k = (memory + pos) W_k^T, v = memory W_v^T (cone/transformer.py:333-336)
    g = G(m, MEM, 256, m->dec_k.w, 256, m->dec_k.b, f.KD, 256 * nd, Mmax, Mdev, 256 * nd, 256);
    g.A2 = f.POS; g.lda2 = 256;
    RUN(launch_gemm(g, s));
    RUN(launch_gemm(G(m, MEM, 256, m->dec_v.w, 256, m->dec_v.b, f.VD, 256 * nd, Mmax, Mdev, 256 * nd, 256), s));
    CONE_CHECK_HIP(hipMemsetAsync(f.TGT, 0, (size_t)T * 256 * sizeof(float), s));                       // tgt = 0 (:66)
    for (int l = 0; l < nd; ++l) {
        const DecLayer& dl = m->dec[l];
        RUN(launch_layernorm(f.TGT, 256, dl.n1.g, dl.n1.b, f.TGT1, 256, T, nullptr, 256, s));           // tgt2 = norm1(tgt)
        g = G(m, f.TGT1, 256, dl.sa.in_w, 256, nullptr, f.DQK, 768, T, nullptr, 768, 256, EPI_RESIDUAL); // q | k | v, slot term from the table
        g.R = m->dec_sa_tab[l]; g.ldr = 768; g.r_mod = m->nq;
        RUN(launch_gemm(g, s));
        RUN(launch_small_attn(f.DQK, 768, f.DQK + 256, 768, f.DQK + 512, 768, f.DATT, 256, nullptr, B, m->nq, m->nq, s));
        g = G(m, f.DATT, 256, dl.sa.out.w, 256, dl.sa.out.b, f.TGT, 256, T, nullptr, 256, 256, EPI_RESIDUAL);
        g.R = f.TGT; g.ldr = 256;
        RUN(launch_gemm(g, s));
        RUN(launch_layernorm(f.TGT, 256, dl.n2.g, dl.n2.b, f.TGT1, 256, T, nullptr, 256, s));           // tgt2 = norm2(tgt)
        g = G(m, f.TGT1, 256, dl.ca.in_w, 256, nullptr, f.DQ, 256, T, nullptr, 256, 256, EPI_RESIDUAL);
        g.R = m->dec_ca_tab[l]; g.ldr = 256; g.r_mod = m->nq;
        RUN(launch_gemm(g, s));
        RUN(launch_small_attn(f.DQ, 256, f.KD + l * 256, 256 * nd, f.VD + l * 256, 256 * nd, f.DATT, 256, f.off, B, m->nq,
                              Lmax, s));
        g = G(m, f.DATT, 256, dl.ca.out.w, 256, dl.ca.out.b, f.TGT, 256, T, nullptr, 256, 256, EPI_RESIDUAL);
        g.R = f.TGT; g.ldr = 256;
        RUN(launch_gemm(g, s));
        RUN(launch_layernorm(f.TGT, 256, dl.n3.g, dl.n3.b, f.TGT1, 256, T, nullptr, 256, s));           // tgt2 = norm3(tgt)
        RUN(launch_gemm(G(m, f.TGT1, 256, dl.l1.w, 256, dl.l1.b, f.DH, ff, T, nullptr, ff, 256, EPI_RELU), s));
        g = G(m, f.DH, ff, dl.l2.w, ff, dl.l2.b, f.TGT, 256, T, nullptr, 256, ff, EPI_RESIDUAL);
        g.R = f.TGT; g.ldr = 256;
        RUN(launch_gemm(g, s));
        RUN(launch_layernorm(f.TGT, 256, m->dec_norm.g, m->dec_norm.b, f.HS + (size_t)l * T * 256, 256, T, nullptr, 256, s));
    }
    // heads (cone/model.py:112-117), all layers into the workspace, the last layer's rows out
    const int HT = nd * T;
    RUN(launch_rowdot(f.HS, 256, m->class_embed.w, m->class_embed.b, f.LG, 2, HT, 2, 0, s));
    RUN(launch_gemm(G(m, f.HS, 256, m->span[0].w, 256, m->span[0].b, f.S1, 256, HT, nullptr, 256, 256, EPI_RELU), s));
    RUN(launch_gemm(G(m, f.S1, 256, m->span[1].w, 256, m->span[1].b, f.S2, 256, HT, nullptr, 256, 256, EPI_RELU), s));
    RUN(launch_rowdot(f.S2, 256, m->span[2].w, m->span[2].b, f.SP, 2, HT, 2, 1, s));
    const size_t last = (size_t)(nd - 1) * T * 2;
    CONE_CHECK_HIP(hipMemcpyAsync(logits, f.LG + last, (size_t)T * 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
    CONE_CHECK_HIP(hipMemcpyAsync(spans, f.SP + last, (size_t)T * 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (taps) {
        if (taps->hs)
            CONE_CHECK_HIP(hipMemcpyAsync(taps->hs, f.HS, (size_t)nd * T * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (taps->aux_logits && nd > 1)
            CONE_CHECK_HIP(hipMemcpyAsync(taps->aux_logits, f.LG, last * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (taps->aux_spans && nd > 1)
            CONE_CHECK_HIP(hipMemcpyAsync(taps->aux_spans, f.SP, last * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    if (saliency || (taps && taps->memory))
        RUN(launch_saliency(MEM, f.off, vlen, qlen, m->saliency.w, m->saliency.b, saliency, Lv_max,
                            taps ? taps->memory : nullptr, Lq_max, B, s));
    return 0;
}

// --pre_norm on the table path (ABI 6): the same launches as the post-norm step -- position tables, ONE N = 768 GEMM per
// encoder layer, the fused layer tail in its pre-norm form (attention out_proj + residual, norm ahead of the feed-forward
// block, the block, residual; the NEXT consumer's LayerNorm -- the next layer's norm1, the encoder's / decoder's final norm --
// written as a second output), the folded decoder cross-attention.  What it does not have: the first layer's row caches (its
// in_proj reads norm1(x)) and the first decoder layer's constants.
// The pre-norm layer tail by row count, as the post-norm one: <= 1 024 rows the spread form, <= 12 288 the wide form, else the
// persistent 128-row kernel -- the same bits in every form (option ffn_spread = 0: the persistent kernel only).
static int tail_prenorm(const cone_model* m, const FwdBuffers& f, const float* A, const float* Wo, const float* bo, const float* R,
                        const float* pg, const float* pb, const float* W1, const float* b1, const float* W2, const float* b2,
                        float* OUT, const float* n2g, const float* n2b, float* OUT2, int M, const int* M_dev, int ff, hipStream_t s,
                        const int* r_idx = nullptr, const float* R2 = nullptr) {
    if (m->opt_spread && f.SPR && ffn_spread_supported(M, ff))
        return launch_proj_ffn_spread(A, 256, Wo, bo, R, 256, pg, pb, W1, b1, W2, b2, n2g ? n2g : pg, n2b ? n2b : pb, OUT, 256, M, ff,
                                      f.SPR, s, M_dev, r_idx, R2, true, OUT2, 256);
    if (m->opt_spread && (M + 15) / 16 <= 768 && ffn_wide_supported(ff))
        return launch_proj_ffn_prenorm_wide(A, 256, Wo, bo, R, 256, pg, pb, W1, b1, W2, b2, OUT, 256, n2g, n2b, OUT2, 256, M, M_dev,
                                            ff, s, r_idx, R2);
    return launch_proj_ffn_prenorm(A, 256, Wo, bo, R, 256, pg, pb, W1, b1, W2, b2, OUT, 256, n2g, n2b, OUT2, 256, M, M_dev, ff, s,
                                   r_idx, R2);
}

static int forward_packed_prenorm_fused(const cone_model* m, const float* vproj, const int* vrow0, const int* vlen,
                                        const float* tproj, const int* trow0, const int* qlen, int B, int Lv_max, int Lq_max,
                                        float* logits, float* spans, float* saliency, const cone_taps* taps, FwdBuffers& f,
                                        hipStream_t s, const cone_layer0* l0) {
    const int Lmax = Lv_max + Lq_max, Mmax = B * Lmax, T = B * m->nq, nd = m->n_dec, ff = m->ff;
    const int* Mdev = f.off + B;
    const size_t pos_rows_n = (size_t)cone_pos_table_rows(l0->max_v_l);
    RUN(launch_scan_lengths(vlen, qlen, B, f.off, s));
    // with the row caches (q | k | v = in_proj(norm1(row)) once per clip / token: layer0_rows) the first layer is a pure gather:
    // its attention reads the caches, its tail gathers the residual rows through a row index -- as on the post-norm path
    const bool caches = l0->qkv_vid && m->opt_l0_gather && m->opt_res_gather;
    float* Z = f.X1;                                                                                 // the normalised rows
    if (caches) {
        RUN(launch_row_index(vrow0, vlen, trow0, qlen, f.off, f.RIDX, B, Lmax, s));
    } else {
        RUN(launch_pack_l0(vproj, vrow0, vlen, tproj, trow0, qlen, f.off, m->dim_t, nullptr, nullptr, nullptr, f.X, nullptr,
                           nullptr, nullptr, B, Lmax, s));                                           // the residual stream
        RUN(launch_layernorm(f.X, 256, m->enc[0].n1.g, m->enc[0].n1.b, Z, 256, Mmax, Mdev, 256, s)); // norm1 of layer 0
    }
    for (int l = 0; l < m->n_enc; ++l) {    // cone/transformer.py:248-260
        const EncLayer& e = m->enc[l];
        const bool g0 = l == 0 && caches;
        AttnSrc src{};
        src.vlen = vlen; src.pos_zero_row = (int)pos_rows_n - 1;
        src.pos_qk = l0->pos_qk + (size_t)l * pos_rows_n * 512;
        if (l0->txt_pos_qk) { src.txt_pos_qk = l0->txt_pos_qk + (size_t)l * l0->n_txt * 512; src.trow0 = trow0; }   // --use_txt_pos
        if (g0) {
            src.qkv_vid = l0->qkv_vid; src.qkv_txt = l0->qkv_txt; src.vrow0 = vrow0; src.trow0 = trow0;
        } else {
            RUN(launch_gemm(G(m, Z, 256, e.sa.in_w, 256, e.sa.in_b, f.QKV, 768, Mmax, Mdev, 768, 256), s));   // q | k | v of src2
            src.Q = f.QKV; src.K = f.QKV + 256; src.V = f.QKV + 512; src.ldq = src.ldk = src.ldv = 768;
        }
        RUN(launch_enc_attn(g0 ? ATTN_GATHER : ATTN_POSADD, src, f.ATT, f.off, B, Lmax, s));
        const LNorm& nxt = l + 1 < m->n_enc ? m->enc[l + 1].n1 : m->enc_norm;
        RUN(tail_prenorm(m, f, f.ATT, e.sa.out.w, e.sa.out.b, g0 ? vproj : f.X, e.n2.g, e.n2.b, e.l1.w, e.l1.b, e.l2.w, e.l2.b, f.X,
                         nxt.g, nxt.b, Z, Mmax, Mdev, ff, s, g0 ? f.RIDX : nullptr, g0 ? tproj : nullptr));
    }
    const float* MEM = Z;                                                                            // encoder.norm(src)
    // --use_txt_pos: the keys memory + pos written once (text rows from the tokens' own position rows), x + pos form of the fold
    const bool xp = l0->txt_pos != nullptr;
    if (xp) RUN(launch_add_pos_rows(MEM, f.off, vlen, l0->pos_rows, f.XP, B, Lmax, s, l0->txt_pos, trow0));
    CONE_CHECK_HIP(hipMemsetAsync(f.TGT, 0, (size_t)T * 256 * sizeof(float), s));                    // tgt = 0 (:66)
    GemmArgs g;
    for (int l = 0; l < nd; ++l) {          // cone/transformer.py:319-342
        const DecLayer& dl = m->dec[l];
        RUN(launch_layernorm(f.TGT, 256, dl.n1.g, dl.n1.b, f.TGT1, 256, T, nullptr, 256, s));
        g = G(m, f.TGT1, 256, dl.sa.in_w, 256, nullptr, f.DQK, 768, T, nullptr, 768, 256, EPI_RESIDUAL);
        g.R = m->dec_sa_tab[l]; g.ldr = 768; g.r_mod = m->nq;
        RUN(launch_gemm(g, s));
        RUN(launch_small_attn(f.DQK, 768, f.DQK + 256, 768, f.DQK + 512, 768, f.DATT, 256, nullptr, B, m->nq, m->nq, s));
        g = G(m, f.DATT, 256, dl.sa.out.w, 256, dl.sa.out.b, f.TGT, 256, T, nullptr, 256, 256, EPI_RESIDUAL);
        g.R = f.TGT; g.ldr = 256;
        RUN(launch_gemm(g, s));
        RUN(launch_layernorm(f.TGT, 256, dl.n2.g, dl.n2.b, f.TGT1, 256, T, nullptr, 256, s));
        g = G(m, f.TGT1, 256, dl.ca.in_w, 256, nullptr, f.DQ, 256, T, nullptr, 256, 256, EPI_RESIDUAL);
        g.R = m->dec_ca_tab[l]; g.ldr = 256; g.r_mod = m->nq;
        RUN(launch_gemm(g, s));
        RUN(launch_dec_cross_mfma(f.DQ, xp ? f.XP : nullptr, MEM, xp ? nullptr : l0->pos_rows, vlen, f.off, dl.ca.in_w + 256 * 256,
                                  m->dec_vT[l], dl.ca.in_b + 512, f.DATT, B, m->nq, Lmax, nullptr, s, 3));
        RUN(tail_prenorm(m, f, f.DATT, dl.ca.out.w, dl.ca.out.b, f.TGT, dl.n3.g, dl.n3.b, dl.l1.w, dl.l1.b, dl.l2.w, dl.l2.b, f.TGT,
                         m->dec_norm.g, m->dec_norm.b, f.HS + (size_t)l * T * 256, T, nullptr, ff, s));
    }
    const int HT = nd * T;
    RUN(launch_rowdot(f.HS, 256, m->class_embed.w, m->class_embed.b, f.LG, 2, HT, 2, 0, s));
    RUN(launch_gemm(G(m, f.HS, 256, m->span[0].w, 256, m->span[0].b, f.S1, 256, HT, nullptr, 256, 256, EPI_RELU), s));
    RUN(launch_gemm(G(m, f.S1, 256, m->span[1].w, 256, m->span[1].b, f.S2, 256, HT, nullptr, 256, 256, EPI_RELU), s));
    RUN(launch_rowdot(f.S2, 256, m->span[2].w, m->span[2].b, f.SP, 2, HT, 2, 1, s));
    const size_t last = (size_t)(nd - 1) * T * 2;
    CONE_CHECK_HIP(hipMemcpyAsync(logits, f.LG + last, (size_t)T * 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
    CONE_CHECK_HIP(hipMemcpyAsync(spans, f.SP + last, (size_t)T * 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (taps) {
        if (taps->hs)
            CONE_CHECK_HIP(hipMemcpyAsync(taps->hs, f.HS, (size_t)nd * T * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (taps->aux_logits && nd > 1)
            CONE_CHECK_HIP(hipMemcpyAsync(taps->aux_logits, f.LG, last * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (taps->aux_spans && nd > 1)
            CONE_CHECK_HIP(hipMemcpyAsync(taps->aux_spans, f.SP, last * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    if (saliency || (taps && taps->memory))
        RUN(launch_saliency(MEM, f.off, vlen, qlen, m->saliency.w, m->saliency.b, saliency, Lv_max,
                            taps ? taps->memory : nullptr, Lq_max, B, s));
    return 0;
}

static int forward_packed(const cone_model* m, const float* vproj, const int* vrow0, const int* vlen,
                          const float* tproj, const int* trow0, const int* qlen, int B, int Lv_max, int Lq_max,
                          float* logits, float* spans, float* saliency, const cone_taps* taps, void* ws,
                          size_t ws_bytes, hipStream_t s, const cone_layer0* l0 = nullptr) {
    CONE_REQUIRE(B > 0 && Lv_max > 0 && Lq_max >= 0, "forward: bad sizes B=%d Lv=%d Lq=%d", B, Lv_max, Lq_max);
    const int Lmax = Lv_max + Lq_max;
    CONE_REQUIRE(Lmax <= CONE_MAX_WINDOW_TOKENS, "forward: window length %d + %d exceeds %d tokens", Lv_max, Lq_max,
                 CONE_MAX_WINDOW_TOKENS);
    // beyond 192 tokens only the default path exists (the 256-key forms of the encoder attention and of the folded cross-
    // attention): the A/B forms and the unfolded decoder stop at 192 keys
    CONE_REQUIRE(Lmax <= 192 || (m->opt_dec_fold >= 2 && dec_cross_mfma_supported(m->nq, Lmax, true) && !m->txt_pos_emb &&
                                 m->opt_pos_tables),
                 "forward: windows of %d tokens (> 192) run only on the default table path with 3 / 5 / 8 decoder slots", Lmax);
    CONE_REQUIRE((int64_t)B * Lmax < (1ll << 24), "forward: batch too large (B * L >= 2^24 tokens)");
    if (m->txt_pos_emb)
        CONE_REQUIRE(Lq_max <= m->txt_pos_rows, "forward: %d text tokens but txt_position_embed has %d rows (max_q_l)", Lq_max,
                     m->txt_pos_rows);
    cone_layer0 eff;
    FwdPlan plan;
    l0 = resolve_l0(m, l0, Lv_max, Lmax, &eff, &plan);
    const bool caches = l0 && l0->qkv_vid;
    if (l0)     // (the handle's own tables or the caller's: either must cover the longest window of the call)
        CONE_REQUIRE(l0->pos_qk && l0->max_v_l >= Lv_max,
                     "forward: position tables / layer-0 cache built for a shorter window (%d < %d clips)", l0->max_v_l, Lv_max);
    Carver c(ws, ws_bytes);
    FwdBuffers f;
    carve_fwd(m, c, B, Lmax, plan, f);
    if (!c.ok) { set_error("forward: workspace too small (%zu < %zu)", ws_bytes, c.cur); return CONE_E_WORKSPACE; }
    if (m->pre_norm && plan.tables)
        return forward_packed_prenorm_fused(m, vproj, vrow0, vlen, tproj, trow0, qlen, B, Lv_max, Lq_max, logits, spans, saliency,
                                            taps, f, s, l0);
    if (m->pre_norm)
        return forward_packed_prenorm(m, vproj, vrow0, vlen, tproj, trow0, qlen, B, Lv_max, Lq_max, logits, spans, saliency, taps,
                                      f, s);
    const int Mmax = B * Lmax, T = B * m->nq, nd = m->n_dec, ff = m->ff;
    const int* Mdev = f.off + B;
    const size_t pos_rows_n = l0 ? (size_t)cone_pos_table_rows(l0->max_v_l) : 0;

    RUN(launch_scan_lengths(vlen, qlen, B, f.off, s));
    const bool gather0 = caches && m->opt_l0_gather;
    // first layer entirely from the per-clip / per-token rows: attention gathers q|k|v, the fused layer tail gathers its
    // residual rows through a row index (kept in the X1 region, unused by that path) -- no packed copy of the input
    const bool gather_res = gather0 && plan.tables && m->opt_ffn_fused >= 2 && m->opt_res_gather && ffn_fused_supported(ff) &&
                            m->n_enc > 0;
    int* RIDX = reinterpret_cast<int*>(f.X1);
    if (gather_res) {
        RUN(launch_row_index(vrow0, vlen, trow0, qlen, f.off, RIDX, B, Lmax, s));
    } else if (caches) {
        // X (and, off the table path, POS): the first layer's attention gathers q|k|v from the caches itself; with
        // the gather switched off a packing pass writes them out first
        RUN(launch_pack_l0(vproj, vrow0, vlen, tproj, trow0, qlen, f.off, m->dim_t, l0->qkv_vid, l0->qkv_txt,
                           l0->pos_qk, f.X, f.POS, gather0 ? nullptr : f.QKV, gather0 ? nullptr : f.QKV + (size_t)Mmax * 512,
                           B, Lmax, s));
    } else if (plan.tables) {
        // no row caches (a caller that hands over projected rows only): the packed layer input, nothing else -- the first
        // layer then runs like the later ones (one N = 768 GEMM on x, position rows from the table)
        RUN(launch_pack_l0(vproj, vrow0, vlen, tproj, trow0, qlen, f.off, m->dim_t, nullptr, nullptr, nullptr, f.X, nullptr,
                           nullptr, nullptr, B, Lmax, s));
    } else {
        RUN(launch_pack_pos(vproj, vrow0, vlen, tproj, trow0, qlen, f.off, m->dim_t, f.X, f.POS, f.XP, B, Lmax, s,
                            m->txt_pos_emb, m->txt_pos_ln.g, m->txt_pos_ln.b));
    }

    bool qkv_fused = false;               // this layer's q | k | v rows were written by the previous layer's tail
    for (int l = 0; l < m->n_enc; ++l) {  // cone/transformer.py:233-246
        const EncLayer& e = m->enc[l];
        GemmArgs g;
        AttnSrc src{};
        int mode = ATTN_PACKED;
        if (l == 0 && gather0) {
            mode = ATTN_GATHER;
            src.qkv_vid = l0->qkv_vid; src.qkv_txt = l0->qkv_txt; src.pos_qk = l0->pos_qk;
            src.vrow0 = vrow0; src.vlen = vlen; src.trow0 = trow0; src.pos_zero_row = (int)pos_rows_n - 1;
            src.txt_pos_qk = l0->txt_pos_qk;                                   // (--use_txt_pos: this layer's = the first image)
        } else if (l == 0 && caches) {                     // packed by pack_l0: (M, 512) q|k then (M, 256) v
            src.Q = f.QKV; src.K = f.QKV + 256; src.V = f.QKV + (size_t)Mmax * 512;
            src.ldq = src.ldk = 512; src.ldv = 256;
        } else if (plan.tables) {
            // q | k | v = x W^T + b in ONE N = 768 GEMM on x; the attention adds pos W_qk^T of this layer from the
            // static table ((x + pos) W^T = x W^T + pos W^T): no x + pos matrix, no second A operand
            if (qkv_fused) {
                // written by the previous layer's fused tail from the registers that held its output rows
            } else if (m->opt_split_bf16 && m->split_img && l > 0)      // (l == 0: the kernel of cone_layer0_project, so that
                                                                        // a window's bits do not depend on who projected it)
                RUN(launch_rows256_split(f.X, 256, m->enc_qkv_img[l], e.sa.in_b, f.QKV, 768, Mmax, Mdev, 768, s));
            else
                RUN(launch_gemm(G(m, f.X, 256, e.sa.in_w, 256, e.sa.in_b, f.QKV, 768, Mmax, Mdev, 768, 256), s));
            mode = ATTN_POSADD;
            src.Q = f.QKV; src.K = f.QKV + 256; src.V = f.QKV + 512; src.ldq = src.ldk = src.ldv = 768;
            src.pos_qk = l0->pos_qk + (size_t)l * pos_rows_n * 512; src.vlen = vlen; src.pos_zero_row = (int)pos_rows_n - 1;
            if (l0->txt_pos_qk) { src.txt_pos_qk = l0->txt_pos_qk + (size_t)l * l0->n_txt * 512; src.trow0 = trow0; }
        } else {
            float* QK = f.QKV; float* V = f.QKV + (size_t)Mmax * 512;
            RUN(launch_gemm(G(m, f.XP, 256, e.sa.in_w, 256, e.sa.in_b, QK, 512, Mmax, Mdev, 512, 256), s));  // q | k = (x+pos) W^T
            RUN(launch_gemm(G(m, f.X, 256, e.sa.in_w + 512 * 256, 256, e.sa.in_b + 512, V, 256, Mmax, Mdev, 256, 256), s));
            src.Q = QK; src.K = QK + 256; src.V = V; src.ldq = src.ldk = 512; src.ldv = 256;
        }
        RUN(launch_enc_attn(mode, src, f.ATT, f.off, B, Lmax, s));
        const bool fuse_ffn = plan.tables && m->opt_ffn_fused && ffn_fused_supported(ff);
        if (fuse_ffn && m->opt_ffn_fused >= 2) {
            // everything behind the attention in ONE launch: norm2(x1 + ffn(x1)), x1 = norm1(x + attn Wo^T + bo); a
            // workgroup reads its 128 rows of x before it writes them, and nobody else touches them: in place
            const bool g0 = l == 0 && gather_res;
            if (m->opt_split_bf16 && m->split_img) {
                // the next encoder layer's q | k | v projection rides in the same launch (its input rows are this kernel's
                // output: no second pass over them); ATT and QKV are disjoint parts of the H region
                const bool next_qkv = m->opt_qkv_fused && l + 1 < m->n_enc && plan.tables && ffn_split_qkv_fits(ff, 768);
                RUN(launch_proj_ffn_split(f.ATT, 256, m->enc_wo_img[l], e.sa.out.b, g0 ? vproj : f.X, 256, e.n1.g, e.n1.b,
                                          m->enc_ffn_img[l], e.l1.b, e.l2.b, e.n2.g, e.n2.b, f.X, 256, Mmax, Mdev, ff, s,
                                          g0 ? RIDX : nullptr, g0 ? tproj : nullptr,
                                          next_qkv ? m->enc_qkv_img[l + 1] : nullptr, next_qkv ? m->enc[l + 1].sa.in_b : nullptr,
                                          next_qkv ? f.QKV : nullptr, 768, next_qkv ? 768 : 0));
                qkv_fused = next_qkv;
            } else {
                // (exact-fp32 kernel: measured neutral against the separate GEMM launch -- 44.3 vs 44.1 ms of kernel time per
                // step -- so only on request: qkv_fused = 2)
                const bool next_qkv = m->opt_qkv_fused >= 2 && l + 1 < m->n_enc && plan.tables && ffn_fused_qkv_fits(ff, 768);
                if (!next_qkv && m->opt_spread && f.SPR && ffn_spread_supported(Mmax, ff)) {
                    // a few windows (<= 1 024 token rows): the spread form, as for the decoder tails of a small batch
                    RUN(launch_proj_ffn_spread(f.ATT, 256, e.sa.out.w, e.sa.out.b, g0 ? vproj : f.X, 256, e.n1.g, e.n1.b, e.l1.w,
                                               e.l1.b, e.l2.w, e.l2.b, e.n2.g, e.n2.b, f.X, 256, Mmax, ff, f.SPR, s, Mdev,
                                               g0 ? RIDX : nullptr, g0 ? tproj : nullptr));
                    qkv_fused = false;
                    continue;
                }
                RUN(launch_proj_ffn_fused(f.ATT, 256, e.sa.out.w, e.sa.out.b, g0 ? vproj : f.X, 256, e.n1.g, e.n1.b, e.l1.w,
                                          e.l1.b, e.l2.w, e.l2.b, e.n2.g, e.n2.b, f.X, 256, Mmax, Mdev, ff, s,
                                          g0 ? RIDX : nullptr, g0 ? tproj : nullptr,
                                          next_qkv ? m->enc[l + 1].sa.in_w : nullptr, next_qkv ? m->enc[l + 1].sa.in_b : nullptr,
                                          next_qkv ? f.QKV : nullptr, 768, next_qkv ? 768 : 0));
                qkv_fused = next_qkv;
            }
            continue;
        }
        g = G(m, f.ATT, 256, e.sa.out.w, 256, e.sa.out.b, f.X1, 256, Mmax, Mdev, 256, 256, EPI_RESIDUAL | EPI_LN);
        g.R = f.X; g.ldr = 256; g.ln_g = e.n1.g; g.ln_b = e.n1.b;
        RUN(launch_gemm(g, s));                                                             // norm1(x + attn)
        if (fuse_ffn) {                                                                     // norm2(x + ffn), one kernel
            RUN(launch_ffn_fused(f.X1, 256, e.l1.w, e.l1.b, e.l2.w, e.l2.b, e.n2.g, e.n2.b, f.X, 256, Mmax, Mdev, ff, s));
        } else {
            RUN(launch_gemm(G(m, f.X1, 256, e.l1.w, 256, e.l1.b, f.H, ff, Mmax, Mdev, ff, 256, EPI_RELU), s));
            g = G(m, f.H, ff, e.l2.w, ff, e.l2.b, f.X, 256, Mmax, Mdev, 256, ff, EPI_RESIDUAL | EPI_LN);
            g.R = f.X1; g.ldr = 256; g.ln_g = e.n2.g; g.ln_b = e.n2.b;
            if (!plan.tables) { g.C2 = f.XP; g.ADD = f.POS; }   // x + pos for the next layer's q/k / the decoder's keys
            RUN(launch_gemm(g, s));                                                         // norm2(x + ffn)
        }
    }
    const float* MEM = f.X;

    // decoder (cone/transformer.py:296-317, 117-146).  Default: the memory K / V projections are folded into
    // the cross-attention kernel (dec_cross.hip); otherwise memory K/V for all layers in two GEMMs.
    const bool fold = plan.fold;
    const bool want_aux = taps && (taps->hs || taps->aux_logits || taps->aux_spans);
    bool sal_done = false;
    const int h0 = want_aux ? 0 : nd - 1;               // first decoder layer whose heads are needed
    const int HT = (nd - h0) * T;
    const size_t hoff = (size_t)h0 * T;
    // only the last layer's heads wanted (the eval pipeline): they write straight into the caller's logits / spans; with the
    // intermediate layers' too, all layers go to the workspace and the last layer's rows are copied out
    float* LGo = want_aux ? f.LG + hoff * 2 : logits;
    float* SPo = want_aux ? f.SP + hoff * 2 : spans;
    // (the chain only where launch_gemm itself would run its 16-row form, and only on the automatic tile family: the A/B
    // families walk k in other orders)
    const bool heads_chain = m->opt_chain && m->opt_gemm == GEMM_AUTO && rows_chain_supported(T) &&
                             !(m->opt_spread && gemm_rows_spread_rows(T));      // (few rows: the spread GEMM launches are faster)
    // the position rows come from the tables inside the cross-attention kernels -- unless text tokens carry their own
    // (--use_txt_pos): then memory + pos is written once (dec_xp) and the kernels run their x + pos form
    const bool dec_tab = plan.tables && !plan.dec_xp;
    if (plan.dec_xp) RUN(launch_add_pos_rows(MEM, f.off, vlen, l0->pos_rows, f.XP, B, Lmax, s, l0->txt_pos, trow0));
    if (!fold) {
        if (dec_tab) RUN(launch_add_pos_rows(MEM, f.off, vlen, l0->pos_rows, f.XP, B, Lmax, s));
        GemmArgs g = G(m, f.XP, 256, m->dec_k.w, 256, m->dec_k.b, f.KD, 256 * nd, Mmax, Mdev, 256 * nd, 256);
        RUN(launch_gemm(g, s));                                                             // k = (memory+pos) W_k^T
        RUN(launch_gemm(G(m, MEM, 256, m->dec_v.w, 256, m->dec_v.b, f.VD, 256 * nd, Mmax, Mdev, 256 * nd, 256), s));
    }
    if (!m->opt_dec0_const)          // tgt = 0 is only read by the first layer's own projections (constants otherwise)
        CONE_CHECK_HIP(hipMemsetAsync(f.TGT, 0, (size_t)T * 256 * sizeof(float), s));
    for (int l = 0; l < nd; ++l) {
        const DecLayer& dl = m->dec[l];
        // Layer 0 starts from tgt = 0 (cone/transformer.py:66): its self-attention block and its cross-attention
        // queries do not depend on the window.  They are computed for ONE window's nq rows by the same kernels
        // (rows of a GEMM are independent: identical bits) and replicated, instead of T = B*nq identical rows.
        const bool dec0 = l == 0 && m->opt_dec0_const;
        const int Tq = dec0 ? m->nq : T;
        GemmArgs g;
        if (dec0) {
            // the layer's self-attention block and cross-attention queries: per-checkpoint constants (dec0_constants, at
            // cone_model_create), replicated to the T rows of the batch in one launch
            RUN(launch_tile_rows2(f.TGT1, m->dec0_tgt1, f.DQ, m->dec0_dq, m->nq, T, s));
        } else {
            // q | k | v of the slots in ONE N = 768 GEMM on tgt, the slot-position term from the layer's table
            g = G(m, f.TGT, 256, dl.sa.in_w, 256, nullptr, f.DQK, 768, T, nullptr, 768, 256, EPI_RESIDUAL);
            g.R = m->dec_sa_tab[l]; g.ldr = 768; g.r_mod = m->nq;
            RUN(launch_gemm(g, s));
            RUN(launch_small_attn(f.DQK, 768, f.DQK + 256, 768, f.DQK + 512, 768, f.DATT, 256, nullptr, B, m->nq, m->nq, s));
            g = G(m, f.DATT, 256, dl.sa.out.w, 256, dl.sa.out.b, f.TGT1, 256, T, nullptr, 256, 256, EPI_RESIDUAL | EPI_LN);
            g.R = f.TGT; g.ldr = 256; g.ln_g = dl.n1.g; g.ln_b = dl.n1.b;
            RUN(launch_gemm(g, s));
            g = G(m, f.TGT1, 256, dl.ca.in_w, 256, nullptr, f.DQ, 256, T, nullptr, 256, 256, EPI_RESIDUAL);
            g.R = m->dec_ca_tab[l]; g.ldr = 256; g.r_mod = m->nq;
            RUN(launch_gemm(g, s));
        }
        if (fold && m->opt_dec_fold >= 2) {
            // the first layer's launch holds every memory row of the batch in registers anyway: the saliency head rides along
            // (table form of the default kernel; other forms: the separate pass at the end)
            const bool ride = l == 0 && saliency && dec_tab && (m->opt_dec_fold == 2 || m->opt_dec_fold == 3 || m->opt_dec_fold == 5);
            if (ride) {
                CONE_CHECK_HIP(hipMemsetAsync(saliency, 0, (size_t)B * Lv_max * sizeof(float), s));     // padded clips: 0
                sal_done = true;
            }
            RUN(launch_dec_cross_mfma(f.DQ, dec_tab ? nullptr : f.XP, MEM, dec_tab ? l0->pos_rows : nullptr, vlen,
                                      f.off, dl.ca.in_w + 256 * 256, m->dec_vT[l], dl.ca.in_b + 512, f.DATT, B, m->nq, Lmax,
                                      Tq != T ? f.QKS : nullptr, s,       // layer 0: the same queries for every window
                                      // the kernel form.  Default: rows-once for the first layer, two-read behind it -- by
                                      // LAYER, not by whether this batch shares its slabs (a one-window batch does not): a
                                      // window's bits must not depend on the batch it rides in
                                      m->opt_dec_fold == 2 ? (l == 0 ? 5 : 3) : m->opt_dec_fold,
                                      ride ? m->saliency.w : nullptr, ride ? m->saliency.b : nullptr, ride ? saliency : nullptr,
                                      ride ? Lv_max : 0));
        }
        else if (fold)
            RUN(launch_dec_cross(f.DQ, dec_tab ? nullptr : f.XP, MEM, dec_tab ? l0->pos_rows : nullptr, vlen,
                                 f.off, dl.ca.in_w + 256 * 256, m->dec_vT[l], dl.ca.in_b + 512, f.DATT, B, m->nq, Lmax, s));
        else
            RUN(launch_small_attn(f.DQ, 256, f.KD + l * 256, 256 * nd, f.VD + l * 256, 256 * nd, f.DATT, 256, f.off, B,
                                  m->nq, Lmax, s));
        if (m->opt_ffn_fused >= 2 && m->opt_split_bf16 && m->split_img) {
            RUN(launch_proj_ffn_split(f.DATT, 256, m->dec_wo_img[l], dl.ca.out.b, f.TGT1, 256, dl.n2.g, dl.n2.b,
                                      m->dec_ffn_img[l], dl.l1.b, dl.l2.b, dl.n3.g, dl.n3.b, f.TGT, 256, T, nullptr, ff, s));
        } else if (m->opt_ffn_fused >= 2 && ffn_fused_supported(ff) && m->opt_spread && f.SPR && ffn_spread_supported(T, ff)) {
            // a handful of slot rows (a single query's 20 windows = 7 row groups): the group's output elements spread over
            // single-wave workgroups in four launches instead of one CU walking the whole block -- the same bits
            RUN(launch_proj_ffn_spread(f.DATT, 256, dl.ca.out.w, dl.ca.out.b, f.TGT1, 256, dl.n2.g, dl.n2.b, dl.l1.w, dl.l1.b,
                                       dl.l2.w, dl.l2.b, dl.n3.g, dl.n3.b, f.TGT, 256, T, ff, f.SPR, s));
        } else if (m->opt_ffn_fused >= 2 && ffn_fused_supported(ff)) {
            RUN(launch_proj_ffn_fused(f.DATT, 256, dl.ca.out.w, dl.ca.out.b, f.TGT1, 256, dl.n2.g, dl.n2.b, dl.l1.w, dl.l1.b,
                                      dl.l2.w, dl.l2.b, dl.n3.g, dl.n3.b, f.TGT, 256, T, nullptr, ff, s));
        } else {
            g = G(m, f.DATT, 256, dl.ca.out.w, 256, dl.ca.out.b, f.TGT2, 256, T, nullptr, 256, 256, EPI_RESIDUAL | EPI_LN);
            g.R = f.TGT1; g.ldr = 256; g.ln_g = dl.n2.g; g.ln_b = dl.n2.b;
            RUN(launch_gemm(g, s));
            if (m->opt_ffn_fused && ffn_fused_supported(ff)) {
                RUN(launch_ffn_fused(f.TGT2, 256, dl.l1.w, dl.l1.b, dl.l2.w, dl.l2.b, dl.n3.g, dl.n3.b, f.TGT, 256, T, nullptr, ff, s));
            } else {
                RUN(launch_gemm(G(m, f.TGT2, 256, dl.l1.w, 256, dl.l1.b, f.DH, ff, T, nullptr, ff, 256, EPI_RELU), s));
                g = G(m, f.DH, ff, dl.l2.w, ff, dl.l2.b, f.TGT, 256, T, nullptr, 256, ff, EPI_RESIDUAL | EPI_LN);
                g.R = f.TGT2; g.ldr = 256; g.ln_g = dl.n3.g; g.ln_b = dl.n3.b;
                RUN(launch_gemm(g, s));
            }
        }
        // decoder.norm + heads on an intermediate layer only feed aux_outputs / the hs tap (unused by inference,
        // cone/inference.py:54-59): computed on request only
        if (l == nd - 1 || want_aux) {
            if (heads_chain) {
                // few rows: decoder.norm -> class head, span MLP -> span head of THIS layer's rows in one launch (rows_chain.h);
                // the normalised rows are written only when the hs tap asks for them
                ChainArgs ca{};
                ca.A = f.TGT; ca.lda = 256; ca.M = T; ca.n_stages = 3;
                const size_t ro = (size_t)(l - h0) * T * 2;
                ChainStage& c0 = ca.st[0];
                c0.kind = 1; c0.ln_g = m->dec_norm.g; c0.ln_b = m->dec_norm.b;
                c0.C = taps && taps->hs ? f.HS + (size_t)l * T * 256 : nullptr; c0.ldc = 256;
                c0.hw = m->class_embed.w; c0.hb = m->class_embed.b; c0.hout = LGo + ro; c0.hld = 2; c0.hnout = 2; c0.hact = 0;
                ChainStage& c1 = ca.st[1];
                c1.kind = 0; c1.K = 256; c1.W = m->span[0].w; c1.bias = m->span[0].b; c1.flags = EPI_RELU;
                ChainStage& c2 = ca.st[2];
                c2.kind = 0; c2.K = 256; c2.W = m->span[1].w; c2.bias = m->span[1].b; c2.flags = EPI_RELU;
                c2.hw = m->span[2].w; c2.hb = m->span[2].b; c2.hout = SPo + ro; c2.hld = 2; c2.hnout = 2; c2.hact = 1;
                RUN(launch_rows_chain(ca, s));
            } else {
                RUN(launch_layernorm(f.TGT, 256, m->dec_norm.g, m->dec_norm.b, f.HS + (size_t)l * T * 256, 256, T, nullptr,
                                     256, s));
            }
        }
    }
    // heads (cone/model.py:112-117); the last layer is the prediction
    if (!heads_chain) {
        RUN(launch_rowdot(f.HS + hoff * 256, 256, m->class_embed.w, m->class_embed.b, LGo, 2, HT, 2, 0, s));
        RUN(launch_gemm(G(m, f.HS + hoff * 256, 256, m->span[0].w, 256, m->span[0].b, f.S1, 256, HT, nullptr, 256, 256, EPI_RELU), s));
        RUN(launch_gemm(G(m, f.S1, 256, m->span[1].w, 256, m->span[1].b, f.S2, 256, HT, nullptr, 256, 256, EPI_RELU), s));
        RUN(launch_rowdot(f.S2, 256, m->span[2].w, m->span[2].b, SPo, 2, HT, 2, 1, s));
    }
    const size_t last = (size_t)(nd - 1) * T * 2;
    if (want_aux) {
        CONE_CHECK_HIP(hipMemcpyAsync(logits, f.LG + last, (size_t)T * 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
        CONE_CHECK_HIP(hipMemcpyAsync(spans, f.SP + last, (size_t)T * 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    if (taps) {
        if (taps->hs)
            CONE_CHECK_HIP(hipMemcpyAsync(taps->hs, f.HS, (size_t)nd * T * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (taps->aux_logits && nd > 1)
            CONE_CHECK_HIP(hipMemcpyAsync(taps->aux_logits, f.LG, last * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (taps->aux_spans && nd > 1)
            CONE_CHECK_HIP(hipMemcpyAsync(taps->aux_spans, f.SP, last * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    if ((saliency && !sal_done) || (taps && taps->memory))     // saliency == NULL: not wanted (cone/inference.py never reads it)
        RUN(launch_saliency(MEM, f.off, vlen, qlen, m->saliency.w, m->saliency.b, sal_done ? nullptr : saliency, Lv_max,
                            taps ? taps->memory : nullptr, Lq_max, B, s));
    return 0;
}

__global__ void iota_fill_kernel(int* a, int* b, int n, int stride, int v) {    // a[i] = i * stride, b[i] = v
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = i * stride; b[i] = v; }
}

}  // namespace cone

using namespace cone;

extern "C" const char* cone_last_error(void) { return g_err; }
extern "C" int cone_abi_version(void) { return CONE_HIP_ABI_VERSION; }

extern "C" int cone_model_create(const cone_weights* w, cone_model** out) { return build_model(w, out); }
extern "C" void cone_model_destroy(cone_model* m) {
    if (!m) return;
    if (m->arena) (void)hipFree(m->arena);
    if (m->split_img) (void)hipFree(m->split_img);
    if (m->tab_arena) (void)hipFree(m->tab_arena);
    delete m;
}

extern "C" size_t cone_adapter_norm_workspace(const cone_model* m, int64_t n_rows) {
    return align_up((size_t)n_rows * 256 * 4, 256) + align_up((size_t)n_rows * m->dv * 4, 256);
}
extern "C" int cone_adapter_norm(const cone_model* m, const float* x, int64_t n_rows, float* out, int renorm,
                                 void* ws, size_t ws_bytes, void* stream) {
    CONE_REQUIRE(m && x && out, "adapter_norm: null argument");
    CONE_REQUIRE(n_rows > 0 && n_rows < (1ll << 31), "adapter_norm: bad row count");
    hipStream_t s = (hipStream_t)stream;
    if (!m->has_adapter) {  // adapter_module == "none": features pass through (cone/inference.py:259-260)
        if (out != x)
            CONE_CHECK_HIP(hipMemcpyAsync(out, x, (size_t)n_rows * m->dv * 4, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    Carver c(ws, ws_bytes);
    float* h = c.take<float>((size_t)n_rows * 256);
    float* y = c.take<float>((size_t)n_rows * m->dv);
    if (!c.ok) { set_error("adapter_norm: workspace too small (%zu < %zu)", ws_bytes, c.cur); return CONE_E_WORKSPACE; }
    RUN(launch_gemm(G(m, x, m->dv, m->adapter[0].w, m->dv, m->adapter[0].b, h, 256, (int)n_rows, nullptr, 256, m->dv, EPI_RELU), s));
    GemmArgs g = G(m, h, 256, m->adapter[1].w, 256, m->adapter[1].b, renorm ? y : out, m->dv, (int)n_rows, nullptr, m->dv,
                   256, EPI_RESIDUAL);
    g.R = x; g.ldr = m->dv;
    RUN(launch_gemm(g, s));
    if (!renorm) return 0;                       // run_on_video/cone_localizator.py:135-138 keeps the raw sum
    return launch_l2norm(y, n_rows, m->dv, 0.f, out, s);
}

extern "C" int cone_l2_normalize_rows(const float* x, int64_t n_rows, int dim, float eps, int clamp, float* out,
                                      void* stream) {
    return launch_l2norm(x, n_rows, dim, eps, out, (hipStream_t)stream, clamp);
}

extern "C" size_t cone_project_workspace(const cone_model* m, int which, int64_t n_rows) {
    return project_ws_bytes(m, which, n_rows);
}
extern "C" int cone_project_tokens(const cone_model* m, int which, const float* x, int64_t n_rows, float* out,
                                   void* ws, size_t ws_bytes, void* stream) {
    CONE_REQUIRE(m && x && out && (which == 0 || which == 1), "project_tokens: bad argument");
    if (n_rows <= 0) return 0;
    return project_tokens(m, which, x, n_rows, out, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" size_t cone_forward_packed_workspace(const cone_model* m, int B, int Lv_max, int Lq_max,
                                                const cone_layer0* l0) {
    cone_layer0 eff;
    FwdPlan plan;
    resolve_l0(m, l0, Lv_max, Lv_max + Lq_max, &eff, &plan);
    return fwd_ws_bytes(m, B, Lv_max + Lq_max, plan);
}
extern "C" int cone_forward_packed(const cone_model* m, const float* vproj, const int32_t* vid_row0,
                                   const int32_t* vid_len, const float* tproj, const int32_t* txt_row0,
                                   const int32_t* txt_len, int B, int Lv_max, int Lq_max, float* logits,
                                   float* spans, float* saliency, const cone_taps* taps, const cone_layer0* l0,
                                   void* ws, size_t ws_bytes, void* stream) {
    CONE_REQUIRE(m && vproj && tproj && vid_row0 && vid_len && txt_row0 && txt_len && logits && spans,
                 "forward_packed: null argument");
    return forward_packed(m, vproj, vid_row0, vid_len, tproj, txt_row0, txt_len, B, Lv_max, Lq_max, logits, spans,
                          saliency, taps, ws, ws_bytes, (hipStream_t)stream, l0);
}

extern "C" int64_t cone_pos_table_rows(int max_v_l) { return pos_table_rows(max_v_l); }

extern "C" int cone_pos_tables(const cone_model* m, int max_v_l, float* pos_rows, float* pos_qk, void* stream) {
    CONE_REQUIRE(m && pos_rows && pos_qk && max_v_l >= 1 && max_v_l <= CONE_TABLE_MAX_V_L, "pos_tables: bad argument");
    return build_pos_tables(m, max_v_l, pos_rows, pos_qk, (hipStream_t)stream);
}

// --use_txt_pos on the table path: the tokens' own position rows and their images under every encoder layer's [W_q | W_k]
// (rows of a GEMM are independent: an arena row and the same token as a compact row of the padded entry get the same bits)
static int text_positions(const cone_model* m, const float* tproj, const int* tok_index, const int* src_row, int mod, int n,
                          const int* n_dev, float* txt_pos, float* txt_pos_qk, hipStream_t s) {
    RUN(launch_txt_pos_rows(tproj, tok_index, src_row, mod, m->txt_pos_rows, m->txt_pos_emb, m->txt_pos_ln.g, m->txt_pos_ln.b, n,
                            n_dev, txt_pos, s));
    for (int l = 0; l < m->n_enc; ++l)
        RUN(launch_gemm(G(m, txt_pos, 256, m->enc[l].sa.in_w, 256, nullptr, txt_pos_qk + (size_t)l * n * 512, 512, n, n_dev, 512,
                          256), s));
    return 0;
}
extern "C" int cone_layer0_text_positions(const cone_model* m, const float* txt_proj_rows, const int32_t* tok_index,
                                          int64_t n_rows, float* txt_pos, float* txt_pos_qk, void* stream) {
    CONE_REQUIRE(m && txt_proj_rows && tok_index && txt_pos && txt_pos_qk && n_rows < (1ll << 31), "layer0_text_positions: bad argument");
    CONE_REQUIRE(m->txt_pos_emb, "layer0_text_positions: the model has no txt_position_embed (--use_txt_pos)");
    if (n_rows <= 0) return 0;
    return text_positions(m, txt_proj_rows, tok_index, nullptr, 1, (int)n_rows, nullptr, txt_pos, txt_pos_qk, (hipStream_t)stream);
}

extern "C" size_t cone_layer0_project_workspace(const cone_model* m, int64_t n_rows) {
    return m && m->pre_norm ? align_up((size_t)n_rows * 256 * 4, 256) : 0;
}
extern "C" int cone_layer0_project(const cone_model* m, const float* proj_rows, int64_t n_rows, float* qkv, void* ws,
                                   size_t ws_bytes, void* stream) {
    CONE_REQUIRE(m && proj_rows && qkv && n_rows < (1ll << 31), "layer0_project: bad argument");
    if (n_rows <= 0) return 0;
    CONE_REQUIRE(ws_bytes >= cone_layer0_project_workspace(m, n_rows) && (ws || !m->pre_norm), "layer0_project: workspace too small");
    return layer0_rows(m, proj_rows, (int)n_rows, nullptr, qkv, (float*)ws, (hipStream_t)stream);
}

// The padded entry.  A zero-padded batch is first COMPACTED: the valid clip rows and the valid token rows are gathered by the
// first LayerNorm of their input projection (padding is never projected), every later row-wise step -- the projections, the
// first encoder layer's q | k | v rows -- runs on the compact rows with a device-side row count, and the windows enter
// forward_packed as (row0, len) pairs into them: from there on this IS the arena path of the eval driver (gathering first
// layer, position tables, fused layer tails, folded decoder), bit for bit.
struct PaddedCarve {
    int *voff, *toff, *vidx, *tidx;
    float *vp, *tp, *qv, *qt, *pt, *ptqk;
    char* pws; size_t pw;
};
static void carve_padded(const cone_model* m, Carver& c, int B, int Lv_pad, int Lq_pad, bool caches, bool txt_tables,
                         PaddedCarve& p) {
    const size_t nv = (size_t)B * Lv_pad, nt = (size_t)B * Lq_pad;
    p.voff = c.take<int>(B + 1); p.toff = c.take<int>(B + 1);
    p.vidx = c.take<int>(nv); p.tidx = c.take<int>(nt);
    p.vp = c.take<float>(nv * 256); p.tp = c.take<float>(nt * 256);
    p.qv = p.qt = p.pt = p.ptqk = nullptr;
    if (caches) { p.qv = c.take<float>(nv * 768); p.qt = c.take<float>(nt * 768); }
    if (txt_tables) { p.pt = c.take<float>(nt * 256); p.ptqk = c.take<float>((size_t)m->n_enc * nt * 512); }   // --use_txt_pos
    p.pw = project_ws_bytes(m, 0, nv);
    const size_t pt = project_ws_bytes(m, 1, nt);
    if (pt > p.pw) p.pw = pt;
    p.pws = c.take<char>(p.pw);
}
// What the padded entry runs on: the handle's tables, first-layer row caches it builds itself, and for a --use_txt_pos model
// the text position rows it builds itself (such a model off the table path: nothing of it -- the general path)
static FwdPlan padded_plan(const cone_model* m, int Lv_pad, int Lmax, bool* caches, bool* txt_tables) {
    cone_layer0 eff;
    const cone_layer0* l0 = effective_l0(m, nullptr, Lv_pad, &eff, true);
    *caches = l0 && m->opt_pos_tables && m->opt_l0_gather;
    FwdPlan p = plan_for(m, l0 != nullptr, *caches, Lmax);
    if (m->txt_pos_emb && !p.tables) { *caches = false; p = plan_for(m, false, false, Lmax); }
    *txt_tables = m->txt_pos_emb && p.tables;
    return p;
}
extern "C" size_t cone_forward_workspace(const cone_model* m, int B, int Lv_pad, int Lq_pad) {
    bool caches, txt_tables;
    const FwdPlan plan = padded_plan(m, Lv_pad, Lv_pad + Lq_pad, &caches, &txt_tables);
    Carver c(nullptr, ~(size_t)0);
    PaddedCarve p;
    carve_padded(m, c, B, Lv_pad, Lq_pad, caches, txt_tables, p);
    return c.cur + fwd_ws_bytes(m, B, Lv_pad + Lq_pad, plan);
}
extern "C" int cone_forward_windows(const cone_model* m, const float* vid, const int32_t* vid_len, const float* txt,
                                    const int32_t* txt_len, int B, int Lv_pad, int Lq_pad, float* logits,
                                    float* spans, float* saliency, const cone_taps* taps, void* ws,
                                    size_t ws_bytes, void* stream) {
    CONE_REQUIRE(m && vid && txt && vid_len && txt_len && logits && spans, "forward_windows: null argument");
    CONE_REQUIRE(B > 0 && Lv_pad > 0 && Lq_pad > 0, "forward_windows: bad sizes");
    CONE_REQUIRE((int64_t)B * Lv_pad < (1ll << 31) && (int64_t)B * Lq_pad < (1ll << 31), "forward_windows: batch too large");
    hipStream_t s = (hipStream_t)stream;
    const size_t nv = (size_t)B * Lv_pad, nt = (size_t)B * Lq_pad;
    bool caches, txt_tables;
    padded_plan(m, Lv_pad, Lv_pad + Lq_pad, &caches, &txt_tables);
    Carver c(ws, ws_bytes);
    PaddedCarve p;
    carve_padded(m, c, B, Lv_pad, Lq_pad, caches, txt_tables, p);
    if (!c.ok) { set_error("forward_windows: workspace too small (%zu < %zu)", ws_bytes, c.cur); return CONE_E_WORKSPACE; }
    // compact row lists of the valid clips / tokens: offsets (= the windows' first rows), source-row indices, device counts
    RUN(launch_scan_lengths(vid_len, nullptr, B, p.voff, s));
    RUN(launch_scan_lengths(txt_len, nullptr, B, p.toff, s));
    RUN(launch_compact_index(vid_len, p.voff, Lv_pad, p.vidx, txt_len, p.toff, Lq_pad, p.tidx, B, s));
    RUN(project_tokens(m, 0, vid, nv, p.vp, p.pws, p.pw, s, p.vidx, p.voff + B));
    RUN(project_tokens(m, 1, txt, nt, p.tp, p.pws, p.pw, s, p.tidx, p.toff + B));
    cone_layer0 l0{};
    if (caches) {   // the first encoder layer's in_proj once per compact row (cone_layer0_project's kernel)
        RUN(layer0_rows(m, p.vp, (int)nv, p.voff + B, p.qv, (float*)p.pws, s));     // (scratch: the projections are done with it)
        RUN(layer0_rows(m, p.tp, (int)nt, p.toff + B, p.qt, (float*)p.pws, s));
        l0.qkv_vid = p.qv; l0.qkv_txt = p.qt;
    }
    if (txt_tables) {   // a compact token row's index in its query = its column in the padded batch (masks are prefixes)
        RUN(text_positions(m, p.tp, nullptr, p.tidx, Lq_pad, (int)nt, p.toff + B, p.pt, p.ptqk, s));
        l0.txt_pos = p.pt; l0.txt_pos_qk = p.ptqk; l0.n_txt = (int64_t)nt;
    }
    return forward_packed(m, p.vp, p.voff, vid_len, p.tp, p.toff, txt_len, B, Lv_pad, Lq_pad, logits, spans, saliency,
                          taps, (char*)ws + c.cur, ws_bytes - c.cur, s, caches || txt_tables ? &l0 : nullptr);
}

extern "C" size_t cone_clip_matching_workspace(const cone_model* m, int B) {
    const size_t T = (size_t)B * m->nq;
    return 2 * align_up(T * m->dv * 4, 256) + align_up(T * 256 * 4, 256) + 3 * align_up((size_t)B * 4, 256);
}
extern "C" int cone_clip_matching_gathered(const cone_model* m, const float* cls, const int32_t* cls_row,
                                           const float* vid, const int32_t* vid_row0, const int32_t* vid_len,
                                           const int32_t* pad_len, const float* spans, int B, float* match,
                                           void* ws, size_t ws_bytes, void* stream) {
    CONE_REQUIRE(m && cls && vid && vid_row0 && vid_len && pad_len && spans && match, "clip_matching: null argument");
    if (B <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int T = B * m->nq, dv = m->dv;
    Carver c(ws, ws_bytes);
    float* pf = c.take<float>((size_t)T * dv);
    float* pa = c.take<float>((size_t)T * dv);
    float* h = c.take<float>((size_t)T * 256);
    if (!c.ok) { set_error("clip_matching: workspace too small (%zu < %zu)", ws_bytes, c.cur); return CONE_E_WORKSPACE; }
    RUN(launch_proposal_mean(vid, vid_row0, vid_len, pad_len, spans, B, m->nq, dv, pf, s));
    const float* feat = pf;
    if (m->has_adapter && dv == 256 && m->opt_chain && m->opt_gemm == GEMM_AUTO && rows_chain_supported(T) &&
        !(m->opt_spread && gemm_rows_spread_rows(T))) {
        ChainArgs ca{};     // few proposals: both adapter layers in one launch (rows_chain.h; the same arithmetic)
        ca.A = pf; ca.lda = dv; ca.M = T; ca.n_stages = 2;
        ca.st[0].kind = 0; ca.st[0].K = dv; ca.st[0].W = m->adapter[0].w; ca.st[0].bias = m->adapter[0].b; ca.st[0].flags = EPI_RELU;
        ca.st[1].kind = 0; ca.st[1].K = 256; ca.st[1].W = m->adapter[1].w; ca.st[1].bias = m->adapter[1].b;
        ca.st[1].flags = EPI_RESIDUAL; ca.st[1].R = pf; ca.st[1].ldr = dv; ca.st[1].C = pa; ca.st[1].ldc = dv;
        RUN(launch_rows_chain(ca, s));
        feat = pa;
    } else if (m->has_adapter) {
        RUN(launch_gemm(G(m, pf, dv, m->adapter[0].w, dv, m->adapter[0].b, h, 256, T, nullptr, 256, dv, EPI_RELU), s));
        GemmArgs g = G(m, h, 256, m->adapter[1].w, 256, m->adapter[1].b, pa, dv, T, nullptr, dv, 256, EPI_RESIDUAL);
        g.R = pf; g.ldr = dv;
        RUN(launch_gemm(g, s));
        feat = pa;
    }
    return launch_cosine_match(feat, cls, cls_row, B, m->nq, dv, match, s);
}
extern "C" int cone_clip_matching(const cone_model* m, const float* cls, const float* vid, const int32_t* vid_len,
                                  int Lv_pad, const float* spans, int B, float* match, void* ws, size_t ws_bytes,
                                  void* stream) {
    CONE_REQUIRE(m && ws, "clip_matching: null argument");
    if (B <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    Carver c(ws, ws_bytes);
    int* vrow0 = c.take<int>(B);
    int* padl = c.take<int>(B);
    if (!c.ok) { set_error("clip_matching: workspace too small"); return CONE_E_WORKSPACE; }
    hipLaunchKernelGGL(iota_fill_kernel, dim3((B + 255) / 256), dim3(256), 0, s, vrow0, padl, B, Lv_pad, Lv_pad);
    CONE_LAUNCH_CHECK();
    return cone_clip_matching_gathered(m, cls, nullptr, vid, vrow0, vid_len, padl, spans, B, match,
                                       (char*)ws + c.cur, ws_bytes - c.cur, stream);
}

extern "C" int cone_model_set_option(cone_model* m, const char* name, int value) {
    CONE_REQUIRE(m && name, "set_option: null argument");
    if (!strcmp(name, "dec_fold")) {
        CONE_REQUIRE(value >= 0 && value <= 5, "set_option: dec_fold %d not in [0, 5]", value);
        m->opt_dec_fold = value;
        return 0;
    }
    if (!strcmp(name, "l0_gather")) { m->opt_l0_gather = value != 0; return 0; }
    if (!strcmp(name, "dec0_const")) { m->opt_dec0_const = value != 0; return 0; }
    if (!strcmp(name, "pos_tables")) { m->opt_pos_tables = value != 0; return 0; }
    if (!strcmp(name, "ffn_fused")) {
        CONE_REQUIRE(value >= 0 && value <= 2, "set_option: ffn_fused %d not in [0, 2]", value);
        m->opt_ffn_fused = value;
        return 0;
    }
    if (!strcmp(name, "qkv_fused")) {
        CONE_REQUIRE(value >= 0 && value <= 2, "set_option: qkv_fused %d not in [0, 2]", value);
        m->opt_qkv_fused = value;
        return 0;
    }
    if (!strcmp(name, "split_bf16")) {
        CONE_REQUIRE(value == 0 || m->split_img, "set_option: split_bf16 needs hidden_dim 256 and dim_feedforward %% 32 == 0 (<= 2048)");
        m->opt_split_bf16 = value != 0;
        return 0;
    }
    if (!strcmp(name, "res_gather")) { m->opt_res_gather = value != 0; return 0; }
    if (!strcmp(name, "rows_chain")) { m->opt_chain = value != 0; return 0; }
    if (!strcmp(name, "ffn_spread")) { m->opt_spread = value != 0; return 0; }
    if (!strcmp(name, "gemm")) {
        CONE_REQUIRE(value >= GEMM_AUTO && value <= GEMM_ROWS8, "set_option: gemm tile family %d not in [0, 3]", value);
        m->opt_gemm = value;
        // the first decoder layer's constants come from these GEMMs: keep them what a step would compute
        if (dec0_constants(m, nullptr) != 0 || hipDeviceSynchronize() != hipSuccess) return CONE_E_HIP;
        return 0;
    }
    cone::set_error("set_option: unknown option '%s'", name);
    return CONE_E_INVALID;
}

extern "C" int cone_test_gemm(const float* A, const float* A2, int a2_mod, const float* W, const float* bias,
                              const float* R, const float* ln_g, const float* ln_b, float* C, float* C2,
                              const float* ADD, int M, int N, int K, int flags, void* stream) {
    GemmArgs g = G(nullptr, A, K, W, K, bias, C, N, M, nullptr, N, K, flags & 0xff);
    g.variant = (flags >> 8) & 3;          // test hook: bits 8-9 pick the tile family (0 automatic)
    g.A2 = A2; g.lda2 = K; g.a2_mod = a2_mod; g.R = R; g.ldr = N; g.ln_g = ln_g; g.ln_b = ln_b;
    g.C2 = C2; g.ADD = ADD;
    return launch_gemm(g, (hipStream_t)stream);
}
extern "C" int cone_test_ffn(const float* X, const float* W1, const float* b1, const float* W2, const float* b2,
                             const float* ln_g, const float* ln_b, float* OUT, int M, int ff, void* stream) {
    return launch_ffn_fused(X, 256, W1, b1, W2, b2, ln_g, ln_b, OUT, 256, M, nullptr, ff, (hipStream_t)stream);
}
extern "C" int cone_test_proj_ffn(const float* A, const float* Wo, const float* bo, const float* R, const float* pg,
                                  const float* pb, const float* W1, const float* b1, const float* W2, const float* b2,
                                  const float* ln_g, const float* ln_b, float* OUT, int M, int ff, void* stream) {
    return launch_proj_ffn_fused(A, 256, Wo, bo, R, 256, pg, pb, W1, b1, W2, b2, ln_g, ln_b, OUT, 256, M, nullptr, ff,
                                 (hipStream_t)stream);
}
extern "C" size_t cone_test_proj_ffn_spread_scratch_bytes(int ff) { return ffn_spread_scratch_floats(ff) * sizeof(float); }
extern "C" int cone_test_proj_ffn_spread(const float* A, const float* Wo, const float* bo, const float* R, const float* pg,
                                         const float* pb, const float* W1, const float* b1, const float* W2, const float* b2,
                                         const float* ln_g, const float* ln_b, float* OUT, int M, int ff, void* scratch,
                                         void* stream) {
    return launch_proj_ffn_spread(A, 256, Wo, bo, R, 256, pg, pb, W1, b1, W2, b2, ln_g, ln_b, OUT, 256, M, ff, (float*)scratch,
                                  (hipStream_t)stream);
}
extern "C" size_t cone_test_ffn_split_image_bytes(int ff) { return ffn_split_supported(ff) ? ffn_split_image_bytes(ff) : 0; }
extern "C" int cone_test_ffn_split(const float* X, const float* W1, const float* b1, const float* W2, const float* b2,
                                   const float* ln_g, const float* ln_b, float* OUT, int M, int ff, void* img, int pack,
                                   void* stream) {
    if (pack) {
        const int rc = launch_ffn_split_pack(W1, W2, ff, img, (hipStream_t)stream);
        if (rc) return rc;
    }
    return launch_ffn_split(X, 256, img, b1, b2, ln_g, ln_b, OUT, 256, M, nullptr, ff, (hipStream_t)stream);
}
extern "C" size_t cone_test_rows_split_image_bytes(int N) { return rows256_split_supported(N) ? rows256_split_image_bytes(N) : 0; }
extern "C" int cone_test_rows_split(const float* X, const float* W, const float* bias, float* C, int M, int N, void* img,
                                    int pack, void* stream) {
    if (pack) {
        const int rc = launch_ffn_split_pack(W, nullptr, N, img, (hipStream_t)stream);
        if (rc) return rc;
    }
    return launch_rows256_split(X, 256, img, bias, C, N, M, nullptr, N, (hipStream_t)stream);
}
extern "C" size_t cone_test_proj_split_image_bytes(void) { return ffn_split_proj_image_bytes(); }
extern "C" int cone_test_proj_ffn_split(const float* A, const float* Wo, const float* bo, const float* R, const float* pg,
                                        const float* pb, const float* W1, const float* b1, const float* W2,
                                        const float* b2, const float* ln_g, const float* ln_b, float* OUT, int M, int ff,
                                        void* img, void* wo_img, int pack, void* stream) {
    if (pack) {
        int rc = launch_ffn_split_pack(W1, W2, ff, img, (hipStream_t)stream);
        if (rc) return rc;
        rc = launch_ffn_split_pack(Wo, nullptr, 256, wo_img, (hipStream_t)stream);
        if (rc) return rc;
    }
    return launch_proj_ffn_split(A, 256, wo_img, bo, R, 256, pg, pb, img, b1, b2, ln_g, ln_b, OUT, 256, M, nullptr, ff,
                                 (hipStream_t)stream);
}
extern "C" int cone_test_enc_attn(int mode, const float* QKV, const float* qkv_vid, const float* qkv_txt,
                                  const float* pos_qk, const int32_t* vrow0, const int32_t* vlen, const int32_t* trow0,
                                  const int32_t* off, float* OUT, int B, int Lmax, int pos_zero_row, void* stream) {
    AttnSrc a{};
    a.pos_zero_row = pos_zero_row;
    a.Q = QKV; a.K = QKV ? QKV + 256 : nullptr; a.V = QKV ? QKV + 512 : nullptr; a.ldq = a.ldk = a.ldv = 768;
    a.qkv_vid = qkv_vid; a.qkv_txt = qkv_txt; a.pos_qk = pos_qk; a.vrow0 = vrow0; a.vlen = vlen; a.trow0 = trow0;
    a.form = (mode >> 8) & 3;                     // mode | 0x200: the wave-per-(window, head) form (enc_attn_wave_kernel)
    return launch_enc_attn(mode & 0xff, a, OUT, off, B, Lmax, (hipStream_t)stream);
}
extern "C" int cone_test_dec_cross(const float* DQ, const float* X, const float* pos_rows, const int32_t* vlen,
                                   const int32_t* off, const float* Wk, const float* WvT, const float* bv, float* OUT,
                                   int B, int nq, int Lmax, int variant, float* qk_slabs, void* stream) {
    if (variant == 1)
        return launch_dec_cross(DQ, nullptr, X, pos_rows, vlen, off, Wk, WvT, bv, OUT, B, nq, Lmax, (hipStream_t)stream);
    return launch_dec_cross_mfma(DQ, nullptr, X, pos_rows, vlen, off, Wk, WvT, bv, OUT, B, nq, Lmax, qk_slabs,
                                 (hipStream_t)stream, variant);
}
extern "C" size_t cone_test_dec_cross_slab_floats(void) { return dec_cross_mfma_slab_floats(); }
extern "C" int cone_test_layernorm(const float* x, const float* g, const float* b, float* out, int64_t n_rows,
                                   int dim, void* stream) {
    return launch_layernorm(x, dim, g, b, out, dim, n_rows, nullptr, dim, (hipStream_t)stream);
}
