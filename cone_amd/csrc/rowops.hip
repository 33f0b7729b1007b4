// Row-wise (HBM-bound) kernels: LayerNorm, L2 normalisation, tiny-N heads.
// One wavefront per row, float4 per lane, wave-shuffle reductions.
#include "common.h"

namespace cone {

// nn.LayerNorm over the last dim (eps 1e-5): cone/model.py:451 (input projections),
// cone/transformer.py:135-141 (decoder.norm).  dim is a multiple of 4, <= 1024.
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx,
                                                        const float* __restrict__ g,
                                                        const float* __restrict__ b, float* out, int ldo,
                                                        int64_t n_rows, const int* n_rows_dev, int dim,
                                                        const int* __restrict__ src_row) {
    if (n_rows_dev) { int64_t nd = *n_rows_dev; n_rows = nd < n_rows ? nd : n_rows; }
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float* xr = x + (src_row ? (int64_t)src_row[row] : row) * ldx;      // src_row: a gathering read (compaction)
    const int nv = dim >> 2;  // float4 per row
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            v[i] = reinterpret_cast<const float4*>(xr)[c];
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            // (every product is an operand of an EXPLICIT fma: under -ffp-contract=fast the backend otherwise picks, per kernel,
            // which product of a sum of two it fuses -- and rows_chain_kernel (gemm.hip) must reproduce these bits)
            const float a = v[i].x - mean, bq = v[i].y - mean, cq = v[i].z - mean, d = v[i].w - mean;
            q += __builtin_fmaf(a, a, bq * bq) + __builtin_fmaf(cq, cq, d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)dim + 1e-5f);
    float* orow = out + row * ldo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float4 gg = reinterpret_cast<const float4*>(g)[c];
            const float4 bb = reinterpret_cast<const float4*>(b)[c];
            float4 o;
            o.x = __builtin_fmaf((v[i].x - mean) * rstd, gg.x, bb.x);
            o.y = __builtin_fmaf((v[i].y - mean) * rstd, gg.y, bb.y);
            o.z = __builtin_fmaf((v[i].z - mean) * rstd, gg.z, bb.z);
            o.w = __builtin_fmaf((v[i].w - mean) * rstd, gg.w, bb.w);
            reinterpret_cast<float4*>(orow)[c] = o;
        }
    }
}

int launch_layernorm(const float* x, int ldx, const float* g, const float* b, float* out, int ldo,
                     int64_t n_rows, const int* n_rows_dev, int dim, hipStream_t s, const int* src_row) {
    CONE_REQUIRE(dim % 4 == 0 && dim <= 1024 && ldx % 4 == 0 && ldo % 4 == 0,
                 "layernorm: dim=%d must be a multiple of 4 and <= 1024", dim);
    if (n_rows <= 0) return 0;
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, x, ldx, g, b,
                       out, ldo, n_rows, n_rows_dev, dim, src_row);
    CONE_LAUNCH_CHECK();
    return 0;
}

// out = x / (||x||_2 + eps): utils/basic_utils.py:97-99 (eps=1e-5), cone/inference.py:257 (eps=0).
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, int64_t n_rows, int dim,
                                                     float eps, int clamp, float* out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float* xr = x + row * dim;
    const int nv = dim >> 2;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            v[i] = reinterpret_cast<const float4*>(xr)[c];
            s += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
    }
    // clamp = 0: x / (||x|| + eps) (l2_normalize_np_array);  clamp = 1: x / max(||x||, eps) (F.normalize)
    const float nrm = clamp ? fmaxf(sqrtf(wave_sum(s)), eps) : sqrtf(wave_sum(s)) + eps;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            float4 o;
            o.x = v[i].x / nrm; o.y = v[i].y / nrm; o.z = v[i].z / nrm; o.w = v[i].w / nrm;
            reinterpret_cast<float4*>(out + row * dim)[c] = o;
        }
    }
}

int launch_l2norm(const float* x, int64_t n_rows, int dim, float eps, float* out, hipStream_t s, int clamp) {
    CONE_REQUIRE(dim % 4 == 0 && dim <= 1024, "l2norm: dim=%d must be a multiple of 4 and <= 1024", dim);
    if (n_rows <= 0) return 0;
    hipLaunchKernelGGL(l2norm_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, x, n_rows, dim, eps,
                       clamp, out);
    CONE_LAUNCH_CHECK();
    return 0;
}

// dst rows [period, n_rows) <- rows (r % period) of the same matrix (256 floats per row): replicates the decoder's
// window-independent first-layer rows to every window.
__global__ __launch_bounds__(256) void tile_rows_kernel(float* x, int period, int64_t n_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)period + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    reinterpret_cast<float4*>(x + row * 256)[lane] = reinterpret_cast<const float4*>(x + (row % period) * 256)[lane];
}

int launch_tile_rows(float* x, int period, int64_t n_rows, hipStream_t s) {
    if (n_rows <= period) return 0;
    hipLaunchKernelGGL(tile_rows_kernel, dim3((unsigned)((n_rows - period + 3) / 4)), dim3(256), 0, s, x, period, n_rows);
    CONE_LAUNCH_CHECK();
    return 0;
}

// Two matrices in one launch, from separate `period`-row sources: dst{0,1} rows [0, n_rows) <- src{0,1}[r % period]
// (the first decoder layer's per-checkpoint constants, replicated to every window of a batch).
__global__ __launch_bounds__(256) void tile_rows2_kernel(float* __restrict__ d0, const float* __restrict__ s0,
                                                         float* __restrict__ d1, const float* __restrict__ s1, int period,
                                                         int64_t n_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int64_t sr = row % period;
    reinterpret_cast<float4*>(d0 + row * 256)[lane] = reinterpret_cast<const float4*>(s0 + sr * 256)[lane];
    reinterpret_cast<float4*>(d1 + row * 256)[lane] = reinterpret_cast<const float4*>(s1 + sr * 256)[lane];
}

int launch_tile_rows2(float* d0, const float* s0, float* d1, const float* s1, int period, int64_t n_rows, hipStream_t s) {
    if (n_rows <= 0) return 0;
    hipLaunchKernelGGL(tile_rows2_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, d0, s0, d1, s1, period, n_rows);
    CONE_LAUNCH_CHECK();
    return 0;
}

// Heads with 1-2 outputs over d=256 inputs: class_embed, last span_embed layer (+sigmoid)
// (cone/model.py:112-115).
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ X, int ldx,
                                                     const float* __restrict__ W,
                                                     const float* __restrict__ b, float* out, int ldo,
                                                     int64_t n_rows, int nout, int act) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float4 xv = reinterpret_cast<const float4*>(X + row * ldx)[lane];
    for (int n = 0; n < nout; ++n) {
        const float4 wv = reinterpret_cast<const float4*>(W + n * 256)[lane];
        float s = __builtin_fmaf(xv.x, wv.x, xv.y * wv.y) + __builtin_fmaf(xv.z, wv.z, xv.w * wv.w);   // (pinned: see layernorm_kernel)
        s = wave_sum(s) + b[n];
        if (act == 1) s = 1.0f / (1.0f + expf(-s));
        if (lane == 0) out[row * ldo + n] = s;
    }
}

int launch_rowdot(const float* X, int ldx, const float* W, const float* b, float* out, int ldo,
                  int64_t n_rows, int nout, int act, hipStream_t s) {
    if (n_rows <= 0) return 0;
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, X, ldx, W, b, out,
                       ldo, n_rows, nout, act);
    CONE_LAUNCH_CHECK();
    return 0;
}

// Valid lengths of prefix masks (utils/tensor_utils.py:50-52: 1 = valid): len[b] = (int) sum_j mask[b][j], one wave per row --
// what the host mirror computed with two torch launches per mask (sum, cast).
__global__ __launch_bounds__(256) void mask_lengths_kernel(const float* __restrict__ mask, int B, int L, int* __restrict__ len) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float s = 0.f;
    for (int j = lane; j < L; j += 64) s += mask[(size_t)b * L + j];
    s = wave_sum(s);
    if (lane == 0) len[b] = (int)s;
}

}  // namespace cone

extern "C" int cone_mask_lengths(const float* mask, int B, int L, int32_t* len, void* stream) {
    CONE_REQUIRE(mask && len && B >= 0 && L >= 0, "mask_lengths: bad argument");
    if (B == 0) return 0;
    hipLaunchKernelGGL(cone::mask_lengths_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, mask, B, L, len);
    CONE_LAUNCH_CHECK();
    return 0;
}
