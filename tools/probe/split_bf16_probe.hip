// Accuracy probe: a 16 x 16 x K product through v_mfma_f32_16x16x32_bf16 with both operands split into three bf16 pieces
// (six partial products hh, hm, mh, mm, hl, lh accumulated in fp32) against float64, next to the exact-fp32
// v_mfma_f32_16x16x4_f32 chain.  Build: hipcc --offload-arch=gfx950 -O3 split_bf16_probe.hip -o split_bf16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned short bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFF + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
__device__ inline float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ inline void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    h = bf16_rne(x);
    const float r1 = x - bf16_f(h);
    m = bf16_rne(r1);
    const float r2 = r1 - bf16_f(m);
    l = bf16_rne(r2);
}

// A (16, K) row-major, B (K, 16) row-major, D (16, 16); one wave
__global__ void probe(const float* A, const float* B, float* D3, float* D2, float* D32, int K, int order) {
    const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
    f4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc32 = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 32) {
        s8 ah, am, al, bh, bm, bl;
        for (int j = 0; j < 8; ++j) {
            unsigned short h, m, l;
            split3(A[li * K + k0 + 8 * lg + j], h, m, l);
            ah[j] = h; am[j] = m; al[j] = l;
            split3(B[(k0 + 8 * lg + j) * 16 + li], h, m, l);
            bh[j] = h; bm[j] = m; bl[j] = l;
        }
        if (order == 0) {           // small terms first
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
        }
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc2, 0, 0, 0);       // two pieces, three products
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc2, 0, 0, 0);
        for (int s = 0; s < 8; ++s)
            acc32 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[li * K + k0 + 4 * s + lg], B[(k0 + 4 * s + lg) * 16 + li], acc32, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) {
        D3[(4 * lg + r) * 16 + li] = acc[r];
        D2[(4 * lg + r) * 16 + li] = acc2[r];
        D32[(4 * lg + r) * 16 + li] = acc32[r];
    }
}

int main() {
    for (int K : {256, 1024}) {
        for (int order = 0; order < 2; ++order) {
            double w3 = 0, w2 = 0, w32 = 0, scale = 0;
            for (int trial = 0; trial < 20; ++trial) {
                std::vector<float> A(16 * K), B(K * 16);
                srand(trial * 7 + K);
                for (auto& x : A) x = (float)((rand() / (double)RAND_MAX - 0.5) * 4.0 * ((trial & 1) ? 1e-3 : 1.0));
                for (auto& x : B) x = (float)((rand() / (double)RAND_MAX - 0.5) * 0.25);
                float *dA, *dB, *d3, *d2, *d32;
                hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4);
                hipMalloc(&d3, 1024); hipMalloc(&d2, 1024); hipMalloc(&d32, 1024);
                hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
                hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, d3, d2, d32, K, order);
                float h3[256], h2[256], h32[256];
                hipMemcpy(h3, d3, 1024, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, 1024, hipMemcpyDeviceToHost);
                hipMemcpy(h32, d32, 1024, hipMemcpyDeviceToHost);
                for (int i = 0; i < 16; ++i)
                    for (int j = 0; j < 16; ++j) {
                        double ref = 0, mag = 0;
                        for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * 16 + j]; mag += fabs((double)A[i * K + k] * B[k * 16 + j]); }
                        w3 = fmax(w3, fabs(h3[i * 16 + j] - ref) / mag);
                        w2 = fmax(w2, fabs(h2[i * 16 + j] - ref) / mag);
                        w32 = fmax(w32, fabs(h32[i * 16 + j] - ref) / mag);
                    }
                hipFree(dA); hipFree(dB); hipFree(d3); hipFree(d2); hipFree(d32);
            }
            printf("K=%d order=%d: max |err| / sum|a b|:  bf16x3 (6 products) %.3e   bf16x2 (3 products) %.3e   fp32 MFMA %.3e\n", K, order, w3, w2, w32);
        }
    }
    return 0;
}
