// Which exact-fp32 MFMA shape sustains more of the 64 FLOP / clock / SIMD under the fused layer tail's instruction mix?
//   v_mfma_f32_16x16x4_f32 (8 passes, 2 048 FLOP) vs v_mfma_f32_32x32x2_f32 (16 passes, 4 096 FLOP): the same peak rate, but the
//   wide shape needs HALF the instructions per FLOP -- and every instruction between two MFMAs (the ds_read_b128 of the next A
//   fragments, the odd vector / scalar instruction) costs issue cycles that the matrix pipe does not hide on this part.
// One workgroup per CU, W waves per SIMD.  Per group of four MFMAs: one ds_read_b128 (the four A operands of the NEXT group:
// software-pipelined, as the kernels do), NV vector FMAs.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_shape_rate.hip -o tools/probe/mfma_shape_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, bool LDS>
__global__ __launch_bounds__(512) void small_shape(float* out, int iters, float a0) {
    __shared__ __attribute__((aligned(16))) float sm[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = i * 1e-6f;
    __syncthreads();
    f32x4 acc[16];
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[4] = {a0, a0 * 2, a0 * 3, a0 * 4};
    const float b = a0 * 0.5f + threadIdx.x * 1e-9f;
    const f32x4* ap = reinterpret_cast<const f32x4*>(sm) + (threadIdx.x & 63);
    f32x4 a = {a0, a0, a0, a0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            f32x4 an = a;
            if (LDS) { an = ap[((g + it) & 15) * 64]; __builtin_amdgcn_sched_barrier(0); }     // next group's fragments, ahead of this group's MFMAs
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[(4 * g + j) & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b, acc[(4 * g + j) & 15], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a = an;
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i & 3] = fmaf(v[i & 3], a0, 1.0f);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    for (int i = 0; i < 4; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, bool LDS>
__global__ __launch_bounds__(512) void wide_shape(float* out, int iters, float a0) {
    __shared__ __attribute__((aligned(16))) float sm[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = i * 1e-6f;
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float v[4] = {a0, a0 * 2, a0 * 3, a0 * 4};
    const float b = a0 * 0.5f + threadIdx.x * 1e-9f;
    const f32x4* ap = reinterpret_cast<const f32x4*>(sm) + (threadIdx.x & 63);
    f32x4 a = {a0, a0, a0, a0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {          // 8 groups x 4 wide MFMAs = the FLOPs of 16 groups x 4 small ones
            f32x4 an = a;
            if (LDS) { an = ap[((g + it) & 15) * 64]; __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b, acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a = an;
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i & 3] = fmaf(v[i & 3], a0, 1.0f);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int i = 0; i < 4; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// the A fragments straight from global memory (L1 / L2 hits), DEPTH groups ahead, instead of LDS
template <int NV, int DEPTH>
__global__ __launch_bounds__(512) void small_shape_glb(float* out, int iters, float a0, const float* wsrc) {
    f32x4 acc[16];
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[4] = {a0, a0 * 2, a0 * 3, a0 * 4};
    const float b = a0 * 0.5f + threadIdx.x * 1e-9f;
    const f32x4* ap = reinterpret_cast<const f32x4*>(wsrc) + (threadIdx.x & 63);
    f32x4 q[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) q[d] = ap[d * 64];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const f32x4 a = q[g % DEPTH];
            q[g % DEPTH] = ap[((g + it) & 15) * 64 + 1024 * (it & 3)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[(4 * g + j) & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b, acc[(4 * g + j) & 15], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i & 3] = fmaf(v[i & 3], a0, 1.0f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    for (int i = 0; i < 4; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, int threads, float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<256, threads>>>(out, 10, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<256, threads>>>(out, iters, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * (threads / 64) * iters * 64.0 * 2048.0;        // 64 small MFMAs' worth per iteration per wave
    printf("%-44s %2d waves/SIMD: %7.3f ms  %6.1f TFLOP/s\n", name, threads / 256, ms, flops / ms / 1e9);
}

template <typename K>
static void run_g(const char* name, K kern, int threads, float* out, int iters, const float* w) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<256, threads>>>(out, 10, 1e-3f, w);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<256, threads>>>(out, iters, 1e-3f, w);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * (threads / 64) * iters * 64.0 * 2048.0;
    printf("%-44s %2d waves/SIMD: %7.3f ms  %6.1f TFLOP/s\n", name, threads / 256, ms, flops / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * sizeof(float));
    float* w; hipMalloc(&w, 64 * 1024 * sizeof(float)); hipMemset(w, 0, 64 * 1024 * sizeof(float));
    const int it = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        run("16x16x4  MFMA only", small_shape<0, false>, 256, out, it);
        run("16x16x4  MFMA only", small_shape<0, false>, 512, out, it);
        run("32x32x2  MFMA only", wide_shape<0, false>, 256, out, it);
        run("32x32x2  MFMA only", wide_shape<0, false>, 512, out, it);
        run("16x16x4  + ds_read_b128 / 4", small_shape<0, true>, 256, out, it);
        run("16x16x4  + ds_read_b128 / 4", small_shape<0, true>, 512, out, it);
        run("32x32x2  + ds_read_b128 / 4", wide_shape<0, true>, 256, out, it);
        run("32x32x2  + ds_read_b128 / 4", wide_shape<0, true>, 512, out, it);
        run_g("16x16x4  + global_load_dwordx4 / 4 (4 ahead)", small_shape_glb<0, 4>, 512, out, it, w);
        run_g("16x16x4  + global_load_dwordx4 / 4 (8 ahead)", small_shape_glb<0, 8>, 512, out, it, w);
        run_g("16x16x4  + global_load_dwordx4 + 1 fma / 4 (8)", small_shape_glb<1, 8>, 512, out, it, w);
        run("16x16x4  + ds_read_b128 + 1 fma / 4", small_shape<1, true>, 256, out, it);
        run("16x16x4  + ds_read_b128 + 1 fma / 4", small_shape<1, true>, 512, out, it);
        run("32x32x2  + ds_read_b128 + 1 fma / 4", wide_shape<1, true>, 256, out, it);
        run("32x32x2  + ds_read_b128 + 1 fma / 4", wide_shape<1, true>, 512, out, it);
        run("16x16x4  + ds_read_b128 + 2 fma / 4", small_shape<2, true>, 512, out, it);
        run("32x32x2  + ds_read_b128 + 2 fma / 4", wide_shape<2, true>, 512, out, it);
        run("32x32x2  + ds_read_b128 + 4 fma / 4", wide_shape<4, true>, 256, out, it);
    }
    return 0;
}
