// Per-window glue kernels of stage B: token packing + sine position encoding, saliency head,
// proposal pooling / cosine matching, row composition.
#include "common.h"

namespace cone {

// off[0..B] = exclusive prefix sum of (vlen[b] + qlen[b]); off[B] is the packed token count.  One workgroup of 16 waves:
// a wave owns a contiguous block of windows, reads it 64 at a time (coalesced), totals it, takes its base from the totals of
// the waves before it, then scans its block 64 at a time with lane shuffles (20 000 windows: 67 -> ~10 us at the step's head,
// where nothing else runs yet).
__global__ __launch_bounds__(1024) void scan_lengths_kernel(const int* __restrict__ vlen,
                                                            const int* __restrict__ qlen, int B, int* off) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (((B + 15) / 16 + 63) / 64) * 64;             // windows per wave, a multiple of 64
    const int b0 = min(wave * per, B), b1 = min(b0 + per, B);
    int s = 0;
    for (int b = b0 + lane; b < b1; b += 64) s += vlen[b] + (qlen ? qlen[b] : 0);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    int run = 0;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    if (tid == 1023) off[B] = run + wsum[15];
    for (int base = b0; base < b1; base += 64) {
        const int b = base + lane;
        const int v = b < b1 ? vlen[b] + (qlen ? qlen[b] : 0) : 0;
        int x = v;                                                // inclusive scan over the 64 lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (b < b1) off[b] = run + x - v;
        run += __shfl(x, 63, 64);
    }
}

int launch_scan_lengths(const int* vlen, const int* qlen, int B, int* off, hipStream_t s) {
    hipLaunchKernelGGL(scan_lengths_kernel, dim3(1), dim3(1024), 0, s, vlen, qlen, B, off);
    CONE_LAUNCH_CHECK();
    return 0;
}

// Source rows of the COMPACT row lists of a zero-padded batch (cone_forward_windows): valid row p of window b of a (B, Lpad, *)
// tensor is row b * Lpad + p; it becomes row off[b] + p of the compact list.  blockIdx.z: 0 = clips, 1 = text tokens.
__global__ __launch_bounds__(256) void compact_index_kernel(const int* __restrict__ vlen, const int* __restrict__ voff, int Lv_pad,
                                                            int* __restrict__ vidx, const int* __restrict__ qlen,
                                                            const int* __restrict__ toff, int Lq_pad, int* __restrict__ tidx) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    const bool txt = blockIdx.z != 0;
    const int len = min(txt ? qlen[b] : vlen[b], txt ? Lq_pad : Lv_pad);
    if (p >= len) return;
    if (txt) tidx[toff[b] + p] = b * Lq_pad + p;
    else vidx[voff[b] + p] = b * Lv_pad + p;
}

int launch_compact_index(const int* vlen, const int* voff, int Lv_pad, int* vidx, const int* qlen, const int* toff, int Lq_pad,
                         int* tidx, int B, hipStream_t s) {
    if (B <= 0) return 0;
    const int L = Lv_pad > Lq_pad ? Lv_pad : Lq_pad;
    hipLaunchKernelGGL(compact_index_kernel, dim3((L + 255) / 256, B, 2), dim3(256), 0, s, vlen, voff, Lv_pad, vidx, qlen, toff,
                       Lq_pad, tidx);
    CONE_LAUNCH_CHECK();
    return 0;
}

// Pack window b = [vproj rows vrow0[b] .. +vlen[b]) ++ [tproj rows trow0[b] .. +qlen[b]) into the
// token matrix X at off[b], and write the matching position rows:
//   video token p : PositionEmbeddingSine(normalize=True), cone/position_encoding.py:51-72:
//                   x = (p+1) / (vlen + 1e-6) * 2pi ;  pos[c] = c even ? sin(x / dim_t[c]) : cos(x / dim_t[c])
//   text token    : zeros (cone/model.py:106, use_txt_pos off); with --use_txt_pos (tpe != null) token t of the query gets
//                   LayerNorm(x + position_embeddings[t]) -- TrainablePositionalEncoding applied to the projected text rows
//                   (cone/position_encoding.py:18-32; eval: no dropout).
// One wavefront per token, float4 per lane (d = 256).
__device__ __forceinline__ float pack_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(256) void pack_pos_kernel(const float* __restrict__ vproj,
                                                       const int* __restrict__ vrow0,
                                                       const int* __restrict__ vlen,
                                                       const float* __restrict__ tproj,
                                                       const int* __restrict__ trow0,
                                                       const int* __restrict__ qlen,
                                                       const int* __restrict__ off,
                                                       const float* __restrict__ dim_t, float* X, float* POS,
                                                       float* XP, const float* __restrict__ tpe,
                                                       const float* __restrict__ tpg, const float* __restrict__ tpb) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int lv = vlen[b], lq = qlen[b];
    if (p >= lv + lq) return;
    const size_t dst = (size_t)(off[b] + p) * 256;
    float4 x, ps;
    if (p < lv) {
        x = reinterpret_cast<const float4*>(vproj + (size_t)(vrow0[b] + p) * 256)[lane];
        const float xe = __fmul_rn(__fdiv_rn((float)(p + 1), __fadd_rn((float)lv, 1e-6f)), 6.283185307179586f);
        const float4 dt = reinterpret_cast<const float4*>(dim_t)[lane];
        ps.x = sinf(__fdiv_rn(xe, dt.x));
        ps.y = cosf(__fdiv_rn(xe, dt.y));
        ps.z = sinf(__fdiv_rn(xe, dt.z));
        ps.w = cosf(__fdiv_rn(xe, dt.w));
    } else {
        x = reinterpret_cast<const float4*>(tproj + (size_t)(trow0[b] + p - lv) * 256)[lane];
        ps = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tpe) {      // wave-uniform
            const float4 e = reinterpret_cast<const float4*>(tpe + (size_t)(p - lv) * 256)[lane];
            float4 v = make_float4(x.x + e.x, x.y + e.y, x.z + e.z, x.w + e.w);
            const float mean = pack_wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 256.0f);
            v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
            const float var = pack_wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) * (1.0f / 256.0f);
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
            const float4 g = reinterpret_cast<const float4*>(tpg)[lane], bb = reinterpret_cast<const float4*>(tpb)[lane];
            ps = make_float4(v.x * rstd * g.x + bb.x, v.y * rstd * g.y + bb.y, v.z * rstd * g.z + bb.z, v.w * rstd * g.w + bb.w);
        }
    }
    reinterpret_cast<float4*>(X + dst)[lane] = x;
    reinterpret_cast<float4*>(POS + dst)[lane] = ps;
    // x + pos: the q/k operand of the first encoder layer (cone/transformer.py:237)
    reinterpret_cast<float4*>(XP + dst)[lane] = make_float4(x.x + ps.x, x.y + ps.y, x.z + ps.z, x.w + ps.w);
}

int launch_pack_pos(const float* vproj, const int* vrow0, const int* vlen, const float* tproj, const int* trow0,
                    const int* qlen, const int* off, const float* dim_t, float* X, float* POS, float* XP, int B,
                    int Lmax, hipStream_t s, const float* tpe, const float* tpg, const float* tpb) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(pack_pos_kernel, dim3((Lmax + 3) / 4, B), dim3(256), 0, s, vproj, vrow0, vlen, tproj,
                       trow0, qlen, off, dim_t, X, POS, XP, tpe, tpg, tpb);
    CONE_LAUNCH_CHECK();
    return 0;
}

// Rows of the sine position table for every (window length Lv, position p): row Lv(Lv-1)/2 + p.
__global__ __launch_bounds__(256) void pos_rows_kernel(const float* __restrict__ dim_t, int max_v_l, float* out) {
    const int lv = blockIdx.y + 1;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= lv) return;
    const float xe = __fmul_rn(__fdiv_rn((float)(p + 1), __fadd_rn((float)lv, 1e-6f)), 6.283185307179586f);
    const float4 dt = reinterpret_cast<const float4*>(dim_t)[lane];
    float4 ps;
    ps.x = sinf(__fdiv_rn(xe, dt.x)); ps.y = cosf(__fdiv_rn(xe, dt.y));
    ps.z = sinf(__fdiv_rn(xe, dt.z)); ps.w = cosf(__fdiv_rn(xe, dt.w));
    reinterpret_cast<float4*>(out + ((size_t)(lv * (lv - 1) / 2 + p)) * 256)[lane] = ps;
}

int launch_pos_rows(const float* dim_t, int max_v_l, float* out, hipStream_t s) {
    hipLaunchKernelGGL(pos_rows_kernel, dim3((max_v_l + 3) / 4, max_v_l), dim3(256), 0, s, dim_t, max_v_l, out);
    CONE_LAUNCH_CHECK();
    return 0;
}

// XP = MEM + pos for the UNFOLDED decoder on the table path (slot counts other than 5: the folded cross-attention kernels are
// specialised): clip token p of window b gets the table row (vlen[b], p), a text token nothing.  One wavefront per token.
__global__ __launch_bounds__(256) void add_pos_rows_kernel(const float* __restrict__ MEM, const int* __restrict__ off,
                                                           const int* __restrict__ vlen, const float* __restrict__ pos_rows,
                                                           float* __restrict__ XP, const float* __restrict__ txt_pos,
                                                           const int* __restrict__ trow0) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int t0 = off[b], L = off[b + 1] - t0, lv = vlen[b];
    if (p >= L) return;
    float4 x = reinterpret_cast<const float4*>(MEM + (size_t)(t0 + p) * 256)[lane];
    if (p < lv) {
        const float4 q = reinterpret_cast<const float4*>(pos_rows + ((size_t)(lv * (lv - 1) / 2 + p)) * 256)[lane];
        x.x += q.x; x.y += q.y; x.z += q.z; x.w += q.w;
    } else if (txt_pos) {       // --use_txt_pos: the token's own position row (cone_layer0_text_positions)
        const float4 q = reinterpret_cast<const float4*>(txt_pos + (size_t)(trow0[b] + p - lv) * 256)[lane];
        x.x += q.x; x.y += q.y; x.z += q.z; x.w += q.w;
    }
    reinterpret_cast<float4*>(XP + (size_t)(t0 + p) * 256)[lane] = x;
}

int launch_add_pos_rows(const float* MEM, const int* off, const int* vlen, const float* pos_rows, float* XP, int B, int Lmax,
                        hipStream_t s, const float* txt_pos, const int* trow0) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(add_pos_rows_kernel, dim3((Lmax + 3) / 4, B), dim3(256), 0, s, MEM, off, vlen, pos_rows, XP, txt_pos,
                       trow0);
    CONE_LAUNCH_CHECK();
    return 0;
}

// --use_txt_pos on the table path: out[i] = LayerNorm(tproj[i] + E[index of token row i in its query]) (cone/model.py:106,
// cone/position_encoding.py:21-31), one wave per token row -- the arithmetic of pack_pos_kernel's text branch.  The index
// comes from tok_index[i], or (the padded entry's compact rows) from the row's place in the padded batch: src_row[i] % mod.
__global__ __launch_bounds__(256) void txt_pos_rows_kernel(const float* __restrict__ tproj, const int* __restrict__ tok_index,
                                                           const int* __restrict__ src_row, int mod, int n_emb,
                                                           const float* __restrict__ tpe, const float* __restrict__ tpg,
                                                           const float* __restrict__ tpb, int n, const int* __restrict__ n_dev,
                                                           float* __restrict__ out) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= (n_dev ? min(*n_dev, n) : n)) return;
    int j = tok_index ? tok_index[i] : src_row[i] % mod;
    j = min(max(j, 0), n_emb - 1);      // (callers check the query length against the table: cone_forward_packed)
    const float4 x = reinterpret_cast<const float4*>(tproj + (size_t)i * 256)[lane];
    const float4 e = reinterpret_cast<const float4*>(tpe + (size_t)j * 256)[lane];
    float4 v = make_float4(x.x + e.x, x.y + e.y, x.z + e.z, x.w + e.w);
    const float mean = pack_wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 256.0f);
    v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
    const float var = pack_wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) * (1.0f / 256.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float4 g = reinterpret_cast<const float4*>(tpg)[lane], bb = reinterpret_cast<const float4*>(tpb)[lane];
    reinterpret_cast<float4*>(out + (size_t)i * 256)[lane] =
        make_float4(v.x * rstd * g.x + bb.x, v.y * rstd * g.y + bb.y, v.z * rstd * g.z + bb.z, v.w * rstd * g.w + bb.w);
}

int launch_txt_pos_rows(const float* tproj, const int* tok_index, const int* src_row, int mod, int n_emb, const float* tpe,
                        const float* tpg, const float* tpb, int n, const int* n_dev, float* out, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(txt_pos_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, s, tproj, tok_index, src_row, mod, n_emb, tpe, tpg,
                       tpb, n, n_dev, out);
    CONE_LAUNCH_CHECK();
    return 0;
}

// pack_pos_kernel + the first encoder layer's in_proj by gather: the q|k|v rows of a token are
//   video clip  : qkv_vid[clip] (= vproj W^T + b, computed once per CLIP) + pos_qk[Lv, p] (= pos W_qk^T, a
//                 static table) on the q|k part -- (x + pos) W^T = x W^T + pos W^T, rows independent;
//   text token  : qkv_txt[token] (once per TOKEN of the query, shared by its windows).
// Replaces two M-row GEMMs (10 % of the window model's FLOPs) by an HBM-bound gather.
__global__ __launch_bounds__(256) void pack_l0_kernel(const float* __restrict__ vproj,
                                                      const int* __restrict__ vrow0,
                                                      const int* __restrict__ vlen,
                                                      const float* __restrict__ tproj,
                                                      const int* __restrict__ trow0,
                                                      const int* __restrict__ qlen,
                                                      const int* __restrict__ off,
                                                      const float* __restrict__ dim_t,
                                                      const float* __restrict__ qkv_vid,
                                                      const float* __restrict__ qkv_txt,
                                                      const float* __restrict__ pos_qk, float* X, float* POS,
                                                      float* QK, float* V) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int lv = vlen[b], lq = qlen[b];
    if (p >= lv + lq) return;
    const size_t row = (size_t)(off[b] + p);
    float4 x, ps = make_float4(0.f, 0.f, 0.f, 0.f), q0 = ps, q1 = q0, vv = q0;
    if (p < lv) {
        const size_t src = (size_t)(vrow0[b] + p);
        x = reinterpret_cast<const float4*>(vproj + src * 256)[lane];
        if (POS) {      // null: every consumer of the position term reads the static tables instead
            const float xe = __fmul_rn(__fdiv_rn((float)(p + 1), __fadd_rn((float)lv, 1e-6f)), 6.283185307179586f);
            const float4 dt = reinterpret_cast<const float4*>(dim_t)[lane];
            ps.x = sinf(__fdiv_rn(xe, dt.x)); ps.y = cosf(__fdiv_rn(xe, dt.y));
            ps.z = sinf(__fdiv_rn(xe, dt.z)); ps.w = cosf(__fdiv_rn(xe, dt.w));
        }
        if (QK) {
            const float4* qs = reinterpret_cast<const float4*>(qkv_vid + src * 768);
            const float4* pq = reinterpret_cast<const float4*>(pos_qk + (size_t)(lv * (lv - 1) / 2 + p) * 512);
            const float4 a0 = qs[lane], a1 = qs[64 + lane], t0 = pq[lane], t1 = pq[64 + lane];
            q0 = make_float4(a0.x + t0.x, a0.y + t0.y, a0.z + t0.z, a0.w + t0.w);
            q1 = make_float4(a1.x + t1.x, a1.y + t1.y, a1.z + t1.z, a1.w + t1.w);
            vv = qs[128 + lane];
        }
    } else {
        const size_t src = (size_t)(trow0[b] + p - lv);
        x = reinterpret_cast<const float4*>(tproj + src * 256)[lane];
        ps = make_float4(0.f, 0.f, 0.f, 0.f);
        if (QK) {
            const float4* qs = reinterpret_cast<const float4*>(qkv_txt + src * 768);
            q0 = qs[lane]; q1 = qs[64 + lane]; vv = qs[128 + lane];
        }
    }
    reinterpret_cast<float4*>(X + row * 256)[lane] = x;
    if (POS) reinterpret_cast<float4*>(POS + row * 256)[lane] = ps;
    if (QK) {       // null: the attention kernel gathers q|k|v itself (ATTN_GATHER)
        reinterpret_cast<float4*>(QK + row * 512)[lane] = q0;
        reinterpret_cast<float4*>(QK + row * 512)[64 + lane] = q1;
        reinterpret_cast<float4*>(V + row * 256)[lane] = vv;
    }
}

// Source row of every packed token (the residual stream entering the first encoder layer is a pure gather of the
// projected clip / text rows): idx >= 0: vproj row idx; idx < 0: tproj row ~idx.  With it the fused layer tail reads its
// residual rows from the per-clip / per-token matrices itself and the 2 GB packed copy is never written.
__global__ __launch_bounds__(256) void row_index_kernel(const int* __restrict__ vrow0, const int* __restrict__ vlen,
                                                        const int* __restrict__ trow0, const int* __restrict__ qlen,
                                                        const int* __restrict__ off, int* __restrict__ ridx) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int lv = vlen[b];
    if (p >= lv + qlen[b]) return;
    ridx[off[b] + p] = p < lv ? vrow0[b] + p : ~(trow0[b] + p - lv);
}

int launch_row_index(const int* vrow0, const int* vlen, const int* trow0, const int* qlen, const int* off, int* ridx,
                     int B, int Lmax, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(row_index_kernel, dim3((Lmax + 255) / 256, B), dim3(256), 0, s, vrow0, vlen, trow0, qlen, off, ridx);
    CONE_LAUNCH_CHECK();
    return 0;
}

int launch_pack_l0(const float* vproj, const int* vrow0, const int* vlen, const float* tproj, const int* trow0,
                   const int* qlen, const int* off, const float* dim_t, const float* qkv_vid, const float* qkv_txt,
                   const float* pos_qk, float* X, float* POS, float* QK, float* V, int B, int Lmax, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(pack_l0_kernel, dim3((Lmax + 3) / 4, B), dim3(256), 0, s, vproj, vrow0, vlen, tproj, trow0,
                       qlen, off, dim_t, qkv_vid, qkv_txt, pos_qk, X, POS, QK, V);
    CONE_LAUNCH_CHECK();
    return 0;
}

// saliency_proj on the video part of memory (cone/model.py:119-122) scattered to (B, Lv_out);
// optional copy of the packed memory into the padded (B, Lv_out + Lq_out, 256) tap.
__global__ __launch_bounds__(256) void saliency_kernel(const float* __restrict__ MEM, const int* __restrict__ off,
                                                       const int* __restrict__ vlen,
                                                       const int* __restrict__ qlen, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* sal, int Lv_out,
                                                       float* mem_tap, int Lq_out) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int lv = vlen[b], lq = qlen[b];
    if (p < Lv_out) {
        float s = 0.f;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < lv) {
            x = reinterpret_cast<const float4*>(MEM + (size_t)(off[b] + p) * 256)[lane];
            const float4 wv = reinterpret_cast<const float4*>(w)[lane];
            s = wave_sum((x.x * wv.x + x.y * wv.y) + (x.z * wv.z + x.w * wv.w)) + bias[0];
        }
        if (lane == 0 && sal) sal[(size_t)b * Lv_out + p] = s;
        if (mem_tap) reinterpret_cast<float4*>(mem_tap + ((size_t)b * (Lv_out + Lq_out) + p) * 256)[lane] = x;
    } else if (mem_tap && p < Lv_out + Lq_out) {
        const int t = p - Lv_out;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < lq) x = reinterpret_cast<const float4*>(MEM + (size_t)(off[b] + lv + t) * 256)[lane];
        reinterpret_cast<float4*>(mem_tap + ((size_t)b * (Lv_out + Lq_out) + p) * 256)[lane] = x;
    }
}

int launch_saliency(const float* MEM, const int* off, const int* vlen, const int* qlen, const float* w,
                    const float* bias, float* sal, int Lv_out, float* mem_tap, int Lq_out, int B, hipStream_t s) {
    if (B <= 0) return 0;
    const int span = mem_tap ? Lv_out + Lq_out : Lv_out;
    hipLaunchKernelGGL(saliency_kernel, dim3((span + 3) / 4, B), dim3(256), 0, s, MEM, off, vlen, qlen, w, bias,
                       sal, Lv_out, mem_tap, Lq_out);
    CONE_LAUNCH_CHECK();
    return 0;
}

// Proposal pooling (cone/model.py:186-199): one wavefront per (window, slot).
//   dur = vlen;  (x1,x2) = (c - 0.5w, c + 0.5w) * dur;  s = max(floor(x1),0);  e = ceil(x2)
//   feat = sum(rows s .. min(e,vlen)-1) / (min(e,pad_len) - s)      (zero rows of the padded tensor: H3)
__global__ __launch_bounds__(256) void proposal_mean_kernel(const float* __restrict__ vid,
                                                            const int* __restrict__ vrow0,
                                                            const int* __restrict__ vlen,
                                                            const int* __restrict__ pad_len,
                                                            const float* __restrict__ spans, int n_prop, int Nq,
                                                            int dv, float* out) {
    const int pi = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pi >= n_prop) return;
    const int b = pi / Nq;
    const float c = spans[2 * pi], w = spans[2 * pi + 1];
    const float hw = __fmul_rn(0.5f, w);
    const float dur = (float)vlen[b];
    const float p1 = __fmul_rn(__fsub_rn(c, hw), dur), p2 = __fmul_rn(__fadd_rn(c, hw), dur);
    int s = (int)floorf(p1);
    s = s > 0 ? s : 0;
    const int e = (int)ceilf(p2);
    const int e_pad = min(e, pad_len[b]);
    const int e_val = min(e, vlen[b]);
    const float cnt = (float)(e_pad - s);  // <= 0 -> the reference's mean of an empty slice (NaN)
    const float* base = vid + (size_t)vrow0[b] * dv;
    const int nv = dv >> 2;
    for (int ch = lane; ch < nv; ch += 64) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        int r = s;
        // eight rows requested before the first is added (the adds keep the row order: same sums as the plain loop)
        for (; r + 8 <= e_val; r += 8) {
            float4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = reinterpret_cast<const float4*>(base + (size_t)(r + u) * dv)[ch];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a.x += x[u].x; a.y += x[u].y; a.z += x[u].z; a.w += x[u].w; }
        }
        for (; r < e_val; ++r) {
            const float4 x = reinterpret_cast<const float4*>(base + (size_t)r * dv)[ch];
            a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
        }
        if (e_pad - s <= 0) {
            a.x = a.y = a.z = a.w = __builtin_nanf("");
        } else {
            a.x = __fdiv_rn(a.x, cnt); a.y = __fdiv_rn(a.y, cnt); a.z = __fdiv_rn(a.z, cnt); a.w = __fdiv_rn(a.w, cnt);
        }
        reinterpret_cast<float4*>(out + (size_t)pi * dv)[ch] = a;
    }
}

int launch_proposal_mean(const float* vid, const int* vrow0, const int* vlen, const int* pad_len,
                         const float* spans, int B, int Nq, int dv, float* out, hipStream_t s) {
    const int n = B * Nq;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(proposal_mean_kernel, dim3((n + 3) / 4), dim3(256), 0, s, vid, vrow0, vlen, pad_len,
                       spans, n, Nq, dv, out);
    CONE_LAUNCH_CHECK();
    return 0;
}

// match[b][n] = < pf / ||pf||, cls / ||cls|| >   (cone/model.py:142,151-152; no eps).
__global__ __launch_bounds__(256) void cosine_match_kernel(const float* __restrict__ pf,
                                                           const float* __restrict__ cls,
                                                           const int* __restrict__ cls_row, int n_prop, int Nq,
                                                           int dv, float* match) {
    const int pi = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pi >= n_prop) return;
    const int b = pi / Nq;
    const float* p = pf + (size_t)pi * dv;
    const float* c = cls + (size_t)(cls_row ? cls_row[b] : b) * dv;
    float pp = 0.f, cc = 0.f;
    for (int ch = lane * 4; ch < dv; ch += 256) {
        const float4 x = *reinterpret_cast<const float4*>(p + ch);
        const float4 y = *reinterpret_cast<const float4*>(c + ch);
        pp += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
        cc += (y.x * y.x + y.y * y.y) + (y.z * y.z + y.w * y.w);
    }
    const float pn = sqrtf(wave_sum(pp)), cn = sqrtf(wave_sum(cc));
    float d = 0.f;
    for (int ch = lane * 4; ch < dv; ch += 256) {
        const float4 x = *reinterpret_cast<const float4*>(p + ch);
        const float4 y = *reinterpret_cast<const float4*>(c + ch);
        d += (__fdiv_rn(x.x, pn) * __fdiv_rn(y.x, cn) + __fdiv_rn(x.y, pn) * __fdiv_rn(y.y, cn)) +
             (__fdiv_rn(x.z, pn) * __fdiv_rn(y.z, cn) + __fdiv_rn(x.w, pn) * __fdiv_rn(y.w, cn));
    }
    d = wave_sum(d);
    if (lane == 0) match[pi] = d;
}

int launch_cosine_match(const float* pf, const float* cls, const int* cls_row, int B, int Nq, int dv, float* match,
                        hipStream_t s) {
    const int n = B * Nq;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(cosine_match_kernel, dim3((n + 3) / 4), dim3(256), 0, s, pf, cls, cls_row, n, Nq, dv,
                       match);
    CONE_LAUNCH_CHECK();
    return 0;
}

// A13, cone/inference.py:47-82.  One thread per window; every fp32 operation is rounded separately
// (no fma contraction) to track the reference's op-by-op torch arithmetic.
__global__ __launch_bounds__(256) void compose_rows_kernel(const float* __restrict__ logits,
                                                           const float* __restrict__ spans,
                                                           const float* __restrict__ match,
                                                           const int* __restrict__ duration,
                                                           const int* __restrict__ vstart, float clip_len,
                                                           int sort, int B, int Nq, float* rows) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float st[16], ed[16], pr[16], mt[16];
    int order[16];
    const float dur = (float)duration[b], vs = (float)vstart[b];
    for (int n = 0; n < Nq; ++n) {
        const float l0 = logits[(b * Nq + n) * 2], l1 = logits[(b * Nq + n) * 2 + 1];
        const float m = fmaxf(l0, l1);
        const float e0 = expf(__fsub_rn(l0, m)), e1 = expf(__fsub_rn(l1, m));
        pr[n] = __fdiv_rn(e0, __fadd_rn(e0, e1));
        const float c = spans[(b * Nq + n) * 2], w = spans[(b * Nq + n) * 2 + 1];
        const float hw = __fmul_rn(0.5f, w);
        st[n] = __fmul_rn(__fadd_rn(__fmul_rn(__fsub_rn(c, hw), dur), vs), clip_len);
        ed[n] = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(c, hw), dur), vs), clip_len);
        mt[n] = match[b * Nq + n];
        order[n] = n;
    }
    if (sort) {  // stable insertion sort, descending by proposal score
        for (int i = 1; i < Nq; ++i) {
            const int oi = order[i];
            int j = i - 1;
            while (j >= 0 && pr[order[j]] < pr[oi]) { order[j + 1] = order[j]; --j; }
            order[j + 1] = oi;
        }
    }
    for (int n = 0; n < Nq; ++n) {
        const int o = order[n];
        float* r = rows + ((size_t)b * Nq + n) * 4;
        r[0] = st[o]; r[1] = ed[o]; r[2] = pr[o]; r[3] = mt[o];
    }
}

}  // namespace cone

extern "C" int cone_compose_rows(const float* logits, const float* spans, const float* match,
                                 const int32_t* duration, const int32_t* video_start, float clip_length, int sort,
                                 int B, int Nq, float* rows, void* stream) {
    CONE_REQUIRE(Nq >= 1 && Nq <= 16, "compose_rows: Nq=%d not in [1,16]", Nq);
    if (B <= 0) return 0;
    hipLaunchKernelGGL(cone::compose_rows_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       logits, spans, match, duration, video_start, clip_length, sort, B, Nq, rows);
    CONE_LAUNCH_CHECK();
    return 0;
}

// A5 -- eval branch of StartEndDataset.__getitem__ + collate (cone/ego4d_mad_dataloader.py:144-159, 229-234, 305-344):
// the window's clip range, its text range and -- hazard H3 -- the zero-padded clip length of its reference batch (the
// longest window among the eval_bsz consecutive queries of the SPLIT that the query belongs to).  Row b of the window list
// is (query row_q[b], rank slot row_slot[b]) -- the list's shape is HOST metadata: a query owns min(K, ceil(ctx_l / S) + 1)
// windows, whatever the pre-filter ranks first (cone/inference.py:286-299 ranks every window of the video, the dataset takes
// the first K: cone/ego4d_mad_dataloader.py:146) -- or, with the maps null, the dense list (b / K, b % K) of a split whose
// videos all hold at least K windows.
namespace cone {

__global__ __launch_bounds__(256) void window_table_kernel(const int* __restrict__ win_idx, int B, int K,
                                                           const int* __restrict__ row_q,
                                                           const int* __restrict__ row_slot,
                                                           const int* __restrict__ q_ctx_l,
                                                           const int* __restrict__ q_vid_off,
                                                           const int* __restrict__ tok_off,
                                                           const int* __restrict__ tok_len, int q_base, int eval_bsz,
                                                           int W, int S, int* __restrict__ batch_max,
                                                           int* __restrict__ vid_row0, int* __restrict__ vid_len,
                                                           int* __restrict__ video_start, int* __restrict__ txt_row0,
                                                           int* __restrict__ txt_len, int* __restrict__ cls_row) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    int vlen = 0, bid = -1;
    if (b < B) {
        const int q = row_q ? row_q[b] : b / K;
        const int wi = win_idx[row_q ? q * K + row_slot[b] : b];
        int start = (wi - 1) * S;                   // window 0 is the half window ahead of the video (:229-234)
        int end = start + W;
        start = start < 0 ? 0 : start;
        const int cl = q_ctx_l[q];
        end = end < cl ? end : cl;
        vlen = end - start;
        vid_row0[b] = q_vid_off[q] + start;
        vid_len[b] = vlen;
        video_start[b] = start;
        txt_row0[b] = tok_off[q];
        txt_len[b] = tok_len[q];
        cls_row[b] = q;
        bid = (q + q_base) / eval_bsz;
    }
    if (!batch_max) return;
    // longest window of every reference batch: rows of a batch are consecutive, so a wave holds one or two batches -- one
    // wave maximum and ONE atomic per (wave, batch) instead of eval_bsz * K contended atomics per batch slot
    unsigned long long todo = __ballot(bid >= 0);
    while (todo) {
        const int leader = __builtin_amdgcn_readlane(bid, __builtin_ctzll(todo));
        const bool mine = bid == leader;
        int m = mine ? vlen : 0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == __builtin_ctzll(todo)) atomicMax(batch_max + leader, m);
        todo &= ~__ballot(mine);
    }
}

__global__ __launch_bounds__(256) void window_pad_kernel(int B, int K, const int* __restrict__ row_q, int q_base, int eval_bsz,
                                                         const int* __restrict__ batch_pad, int* __restrict__ pad_len) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) pad_len[b] = batch_pad[((row_q ? row_q[b] : b / K) + q_base) / eval_bsz];
}

}  // namespace cone

extern "C" int cone_window_table(const int32_t* win_idx, int nq, int K, const int32_t* row_q, const int32_t* row_slot,
                                 int n_rows, const int32_t* q_ctx_l,
                                 const int32_t* q_vid_off, const int32_t* tok_off, const int32_t* tok_len, int q_base,
                                 int eval_bsz, int max_v_l, int32_t* batch_pad, int derive_pad, int n_batches,
                                 int32_t* vid_row0, int32_t* vid_len, int32_t* video_start, int32_t* pad_len,
                                 int32_t* txt_row0, int32_t* txt_len, int32_t* cls_row, void* stream) {
    CONE_REQUIRE(win_idx && q_ctx_l && q_vid_off && tok_off && tok_len && batch_pad && vid_row0 && vid_len &&
                     video_start && pad_len && txt_row0 && txt_len && cls_row, "window_table: null argument");
    CONE_REQUIRE(K >= 1 && eval_bsz >= 1 && max_v_l >= 2 && q_base >= 0, "window_table: bad K / eval_bsz / max_v_l / q_base");
    CONE_REQUIRE((q_base + (nq > 0 ? nq - 1 : 0)) / eval_bsz < n_batches, "window_table: batch_pad holds %d batches, queries reach batch %d",
                 n_batches, (q_base + nq - 1) / eval_bsz);
    CONE_REQUIRE((row_q == nullptr) == (row_slot == nullptr), "window_table: row_q and row_slot come together");
    CONE_REQUIRE(row_q ? (n_rows >= 0 && (int64_t)n_rows <= (int64_t)nq * K) : n_rows == nq * K,
                 "window_table: n_rows=%d does not fit %d queries x %d windows", n_rows, nq, K);
    const int B = n_rows;
    if (B <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (derive_pad) CONE_CHECK_HIP(hipMemsetAsync(batch_pad, 0, sizeof(int32_t) * (size_t)n_batches, s));
    hipLaunchKernelGGL(cone::window_table_kernel, dim3((B + 255) / 256), dim3(256), 0, s, win_idx, B, K, row_q, row_slot, q_ctx_l, q_vid_off,
                       tok_off, tok_len, q_base, eval_bsz, max_v_l, max_v_l / 2, derive_pad ? batch_pad : nullptr, vid_row0,
                       vid_len, video_start, txt_row0, txt_len, cls_row);
    CONE_LAUNCH_CHECK();
    hipLaunchKernelGGL(cone::window_pad_kernel, dim3((B + 255) / 256), dim3(256), 0, s, B, K, row_q, q_base, eval_bsz, batch_pad, pad_len);
    CONE_LAUNCH_CHECK();
    return 0;
}
