// What does a bare v_mfma_f32_16x16x32_bf16 stream reach when issued the way ffn_split.hip issues it: groups of six
// dependent MFMAs on one accumulator, 16 groups per "phase", 1 / 2 waves per SIMD, zero or random-ish operands?
// Build: hipcc --offload-arch=gfx950 -O3 mfma_bf16_rate.hip -o mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, int seed) {
    s8 a[3], b[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (short)(seed ? (0x3f80 + ((threadIdx.x * 7 + i * 13 + j * 29) & 0x7f)) : 0);
            b[i][j] = (short)(seed ? (0x3c00 + ((threadIdx.x * 5 + i * 11 + j * 31) & 0x7f)) : 0);
        }
    f4 acc[16];
    for (int t = 0; t < 16; ++t) acc[t] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int t = NACC == 1 ? 0 : u % NACC;
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc[t], 0, 0, 0);
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]));
        }
    }
    float s = 0.f;
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// the kernel's pattern: 24 resident B operands (8 steps x 3 pieces), A fragments double-buffered out of LDS
template <bool LDSA>
__global__ __launch_bounds__(512, 2) void k2(float* out, int iters, int seed) {
    __shared__ __attribute__((aligned(16))) char sm[49152];
    for (int i = threadIdx.x; i < 49152 / 4; i += 512) reinterpret_cast<int*>(sm)[i] = seed ? 0x3c003c10 + (i & 31) : 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    s8 b[24];
    for (int i = 0; i < 24; ++i)
        for (int j = 0; j < 8; ++j) b[i][j] = (short)(seed ? (0x3c00 + ((threadIdx.x * 5 + i * 11 + j * 31) & 0x7f)) : 0);
    s8 f[2][3];
    f4 a0 = {0, 0, 0, 0}, a1 = a0;
#define RD(slab) (*reinterpret_cast<const s8*>(sm + (slab) * 1024 + lane * 16))
    for (int it = 0; it < iters; ++it) {
        f[0][0] = RD(0); f[0][1] = RD(1); f[0][2] = RD(2);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+v"(f[u & 1][0]), "+v"(f[u & 1][1]), "+v"(f[u & 1][2]));
            __builtin_amdgcn_sched_barrier(0);
            if (LDSA && u < 15) { f[(u + 1) & 1][0] = RD((u + 1) * 3); f[(u + 1) & 1][1] = RD((u + 1) * 3 + 1); f[(u + 1) & 1][2] = RD((u + 1) * 3 + 2); }
            else if (u < 15) { f[(u + 1) & 1][0] = f[u & 1][0]; f[(u + 1) & 1][1] = f[u & 1][1]; f[(u + 1) & 1][2] = f[u & 1][2]; }
            __builtin_amdgcn_sched_barrier(0);
            const int s3 = (u & 7) * 3;
            if (u < 8) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[u & 1][2], b[s3], a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[u & 1][0], b[s3 + 2], a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[u & 1][1], b[s3 + 1], a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[u & 1][1], b[s3], a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[u & 1][0], b[s3 + 1], a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[u & 1][0], b[s3], a0, 0, 0, 0);
            } else {
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[u & 1][2], b[s3], a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[u & 1][0], b[s3 + 2], a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[u & 1][1], b[s3 + 1], a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[u & 1][1], b[s3], a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[u & 1][0], b[s3 + 1], a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[u & 1][0], b[s3], a1, 0, 0, 0);
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a0[1] + a0[2] + a0[3] + a1[0] + a1[1] + a1[2] + a1[3];
}

template <bool LDSA>
void run2(const char* name, int seed) {
    float* out;
    hipMalloc(&out, 512 * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k2<LDSA><<<256, 512>>>(out, 10, seed);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k2<LDSA><<<256, 512>>>(out, iters, seed);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)iters * 96 * 8 * 256;
    printf("%-34s waves/SIMD 2  %s operands: %.3f ms  %.0f TFLOP/s bf16 = %.1f cycles per MFMA per SIMD at 2.4 GHz\n", name,
           seed ? "random" : "zero  ", ms, mfmas * 16384.0 / (ms * 1e-3) / 1e12, 2.4e9 * ms * 1e-3 / (mfmas / 1024));
    hipFree(out);
}

template <int NACC>
void run(const char* name, int threads, int seed) {
    float* out;
    hipMalloc(&out, 512 * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<256, threads>>>(out, 10, seed);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<256, threads>>>(out, iters, seed);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)iters * 96 * (threads / 64) * 256;
    const double tf = mfmas * 16384.0 / (ms * 1e-3) / 1e12;
    printf("%-34s waves/SIMD %d  %s operands: %.3f ms  %.0f TFLOP/s bf16 = %.1f cycles per MFMA per SIMD at 2.4 GHz\n", name,
           threads / 256, seed ? "random" : "zero  ", ms, tf, 2.4e9 * ms * 1e-3 / (mfmas / 1024));
    hipFree(out);
}

int main() {
    for (int seed = 0; seed < 2; ++seed) {
        run<1>("6-chains on ONE accumulator", 256, seed);
        run<1>("6-chains on ONE accumulator", 512, seed);
        run<16>("6-chains over 16 accumulators", 256, seed);
        run<16>("6-chains over 16 accumulators", 512, seed);
        run2<false>("kernel pattern, A in registers", seed);
        run2<true>("kernel pattern, A from LDS", seed);
    }
    return 0;
}
