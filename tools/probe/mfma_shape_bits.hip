// Do v_mfma_f32_16x16x4_f32 and v_mfma_f32_32x32x2_f32 round the same way?  C = A B^T over K = 256 with both shapes, the k index
// walked in the same order (a 16x16x4 step adds products k .. k+3 to the accumulator, two 32x32x2 steps add k, k+1 and k+2,
// k+3): identical bits would mean an MFMA is a chain of rounded FMAs in k order and kernels of either shape are interchangeable.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_shape_bits.hip -o tools/probe/mfma_shape_bits
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A (32, 256), B (32, 256) row-major; C (32, 32) = A B^T
__global__ void small_shape(const float* A, const float* B, float* C) {
    const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
    for (int mt = 0; mt < 2; ++mt)
        for (int nt = 0; nt < 2; ++nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < 256; k += 4)      // A operand: lane (m = li, k slot lg); B operand: lane (n = li, k slot lg)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(16 * mt + li) * 256 + k + lg], B[(16 * nt + li) * 256 + k + lg], acc, 0, 0, 0);
            for (int r = 0; r < 4; ++r) C[(16 * mt + 4 * lg + r) * 32 + 16 * nt + li] = acc[r];
        }
}
__global__ void wide_shape(const float* A, const float* B, float* C) {
    const int lane = threadIdx.x, l32 = lane & 31, hi = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = 0; k < 256; k += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l32 * 256 + k + hi], B[l32 * 256 + k + hi], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l32] = acc[r];
}
int main() {
    const int n = 32 * 256;
    float *hA = (float*)malloc(n * 4), *hB = (float*)malloc(n * 4), h1[1024], h2[1024];
    srand(1);
    int bad_total = 0;
    for (int trial = 0; trial < 8; ++trial) {
        for (int i = 0; i < n; ++i) {
            hA[i] = (rand() / (float)RAND_MAX - 0.5f) * (trial & 1 ? 100.f : 1.f);
            hB[i] = (rand() / (float)RAND_MAX - 0.5f) * (trial & 2 ? 1e-3f : 1.f);
        }
        float *A, *B, *C1, *C2;
        hipMalloc(&A, n * 4); hipMalloc(&B, n * 4); hipMalloc(&C1, 4096); hipMalloc(&C2, 4096);
        hipMemcpy(A, hA, n * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB, n * 4, hipMemcpyHostToDevice);
        small_shape<<<1, 64>>>(A, B, C1);
        wide_shape<<<1, 64>>>(A, B, C2);
        hipMemcpy(h1, C1, 4096, hipMemcpyDeviceToHost); hipMemcpy(h2, C2, 4096, hipMemcpyDeviceToHost);
        int bad = 0; double ref_err1 = 0, ref_err2 = 0;
        for (int i = 0; i < 1024; ++i) {
            if (memcmp(&h1[i], &h2[i], 4)) ++bad;
            double r = 0; for (int k = 0; k < 256; ++k) r += (double)hA[(i / 32) * 256 + k] * hB[(i % 32) * 256 + k];
            ref_err1 += fabs(h1[i] - r); ref_err2 += fabs(h2[i] - r);
        }
        printf("trial %d: %d of 1024 outputs differ in bits; mean |err| vs float64: 16x16x4 %.3e, 32x32x2 %.3e\n", trial, bad,
               ref_err1 / 1024, ref_err2 / 1024);
        bad_total += bad;
    }
    printf(bad_total ? "DIFFERENT\n" : "IDENTICAL BITS\n");
    return 0;
}
