// Does VALU work hide under v_mfma_f32_16x16x4_f32 on gfx950?  One workgroup per CU.
//   same<NV>   : 1 wave per SIMD, NV independent v_fma_f32 issued after every MFMA (4 accumulator chains)
//   cross<NV>  : 2 waves per SIMD; waves 0-3 issue the MFMAs, waves 4-7 the same number of VALU instructions
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_valu_overlap.hip -o tools/probe/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, bool EXP>
__device__ __forceinline__ void valu_block(float (&v)[8], float c) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (EXP && (i & 3) == 3) v[i & 7] = __builtin_amdgcn_exp2f(v[i & 7]);
        else v[i & 7] = fmaf(v[i & 7], c, 1.0f);
    }
}

template <int NV, bool EXP>
__global__ __launch_bounds__(256) void same(float* out, int iters, float a0) {
    f32x4 acc[4];
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a0 * i;
    const float a = a0 + threadIdx.x * 1e-9f, b = a0 * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            valu_block<NV, EXP>(v, a0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same with integer VALU (v_add_u32 / v_xor) or LDS reads (ds_read_b32) as the filler
template <int NV, bool LDS>
__global__ __launch_bounds__(256) void same_int(float* out, int iters, float a0) {
    __shared__ float sm[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = i * 1e-6f;
    __syncthreads();
    f32x4 acc[4];
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned v[8];
    float w[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 7u + i; w[i] = 0.f; }
    const float a = a0 + threadIdx.x * 1e-9f, b = a0 * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (LDS) { float x = sm[(v[i & 7] + j) & 1023]; asm volatile("" : "+v"(x)); w[i & 7] = x; }
                else { v[i & 7] = (v[i & 7] + 0x9e3779b9u) ^ (unsigned)it; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    for (int i = 0; i < 8; ++i) s += (float)v[i] + w[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, bool EXP>
__global__ __launch_bounds__(512) void cross(float* out, int iters, float a0) {
    f32x4 acc[4];
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a0 * i;
    const float a = a0 + threadIdx.x * 1e-9f, b = a0 * 0.5f;
    if (threadIdx.x < 256) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { valu_block<NV, EXP>(v, a0); __builtin_amdgcn_sched_barrier(0); }
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <typename F>
static float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;                      // 320 000 MFMAs per wave = 10.24 M cycles of issue
#define RUN(K, NV, EXP, T)                                                                                          \
    {                                                                                                               \
        float ms = timeit([&] { hipLaunchKernelGGL((K<NV, EXP>), dim3(256), dim3(T), 0, 0, out, iters, 1.0001f); }); \
        printf("%-6s NV=%2d exp=%d: %8.3f ms  (%.1f cycles per MFMA at 2.4 GHz)\n", #K, NV, (int)EXP, ms,           \
               ms * 1e-3 * 2.4e9 / (iters * 16.0));                                                                 \
    }
    RUN(same, 0, false, 256) RUN(same, 2, false, 256) RUN(same, 4, false, 256) RUN(same, 6, false, 256)
    RUN(same, 8, false, 256) RUN(same, 12, false, 256) RUN(same, 4, true, 256) RUN(same, 8, true, 256)
    RUN(same_int, 2, false, 256) RUN(same_int, 4, false, 256) RUN(same_int, 8, false, 256)
    RUN(same_int, 1, true, 256) RUN(same_int, 2, true, 256) RUN(same_int, 4, true, 256)
    RUN(cross, 0, false, 512) RUN(cross, 4, false, 512) RUN(cross, 8, false, 512) RUN(cross, 12, false, 512)
    RUN(cross, 8, true, 512)
    return 0;
}
