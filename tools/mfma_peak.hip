// Calibration: what does a bare v_mfma_f32_32x32x2_f32 stream reach on this chip?
// variants: waves per SIMD (1,2), accumulators per wave (1,2,4,8), with/without LDS-read fillers.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    __shared__ __attribute__((aligned(16))) float sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = (float)i * 1e-9f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    const float* p = sm + (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            f32x4v v = *reinterpret_cast<const f32x4v*>(p + ((it & 7) << 8));
            asm volatile("" : "+v"(v));
            b += v[0];
        }
#pragma unroll
        for (int j = 0; j < 8 / NACC; ++j)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// The GEMM inner pattern: 8 MFMAs on 2 accumulators, then consume the fragments requested before them
// and request the next ones (one-in-flight prefetch), optional barrier every 8 units.
template <bool BARRIER>
__global__ __launch_bounds__(256, 2) void kg(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[12288];
    for (int i = threadIdx.x; i < 12288; i += 256) sm[i] = (float)(i & 1023) * 1e-6f;
    __syncthreads();
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float* p = sm + (threadIdx.x & 63) * 4;
    f32x4v a = *reinterpret_cast<const f32x4v*>(p);
    f32x4v b0 = *reinterpret_cast<const f32x4v*>(p + 256), b1 = *reinterpret_cast<const f32x4v*>(p + 512);
    f32x4v n0 = *reinterpret_cast<const f32x4v*>(p + 768), n1 = *reinterpret_cast<const f32x4v*>(p + 1024);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = (u & 3) * 2;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b0[j], acc[t], 0, 0, 0);
                acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b1[j], acc[t + 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+v"(n0), "+v"(n1));
            b0 = n0; b1 = n1;
            __builtin_amdgcn_sched_barrier(0);
            n0 = *reinterpret_cast<const f32x4v*>(p + ((u * 2 + it) & 7) * 1024);
            n1 = *reinterpret_cast<const f32x4v*>(p + ((u * 2 + 1 + it) & 7) * 1024 + 256);
        }
        if (BARRIER) __syncthreads();
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool BARRIER>
void rung(const char* name, int blocks_per_cu) {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 2500, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kg<BARRIER><<<grid, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kg<BARRIER><<<grid, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 64 * 4096.0;
    printf("%-28s waves/SIMD=%d  %.1f TF  (%.2f ms)\n", name, blocks_per_cu, flops / ms / 1e9, ms);
    hipFree(out);
}

template <int NACC, bool LDS>
void run(const char* name, int blocks_per_cu) {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 20000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, LDS><<<grid, 256>>>(out, 100, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, LDS><<<grid, 256>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 8 * 4096.0;
    printf("%-28s waves/SIMD=%d  %.1f TF  (%.2f ms)\n", name, blocks_per_cu, flops / ms / 1e9, ms);
    hipFree(out);
}

int main() {
    run<8, false>("8 acc, bare", 1);
    run<8, false>("8 acc, bare", 2);
    run<4, false>("4 acc, bare", 1);
    run<2, false>("2 acc, bare", 1);
    run<2, false>("2 acc, bare", 2);
    run<1, false>("1 acc (dependent), bare", 1);
    run<1, false>("1 acc (dependent), bare", 2);
    run<2, true>("2 acc + ds_read_b128/8mfma", 1);
    run<2, true>("2 acc + ds_read_b128/8mfma", 2);
    rung<false>("gemm pattern, no barrier", 1);
    rung<false>("gemm pattern, no barrier", 2);
    rung<true>("gemm pattern + barrier/64", 1);
    rung<true>("gemm pattern + barrier/64", 2);
    return 0;
}
