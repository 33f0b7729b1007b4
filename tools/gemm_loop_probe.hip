// Probe: main-loop rate of the row GEMM's instruction mix WITH its LDS-DMA streams, for two wave shapes:
//   V0  4 waves / workgroup, wave = 32 rows x 256 cols on v_mfma_f32_32x32x2_f32 (the shipped kernel's loop)
//   V1  8 waves / workgroup, wave = 16 rows x 256 cols on v_mfma_f32_16x16x4_f32 (twice the waves per SIMD)
// Same 128 x 256 x 16 slabs, 3-slab ring, counted vmcnt + raw barrier, 2 workgroups per CU; no epilogue, data is
// not meaningful (addresses are: A streams fresh rows from a large buffer, W re-reads a small L2-resident one).
// Build: hipcc -O3 --offload-arch=gfx950 tools/gemm_loop_probe.hip -o tools/gemm_loop_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define GLDS16(src, dst) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
constexpr int SLAB = (128 + 256) * 16;   // floats

template <int V>
__global__ __launch_bounds__(V ? 512 : 256, 2) void probe(const float* __restrict__ A, const float* __restrict__ W,
                                                        float* out, int tiles, int nk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = V ? 8 : 4;
    constexpr int NPIECE = 24 / NW;                       // 1-KiB LDS-DMA instructions per wave per slab
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x16 acc32[V ? 1 : 8];
    f32x4v acc16[V ? 16 : 1];
    for (auto& a : acc32) for (int r = 0; r < 16; ++r) a[r] = 0.f;
    for (auto& a : acc16) a = f32x4v{0.f, 0.f, 0.f, 0.f};
    // piece p of this wave: rows (wave * NPIECE + p) * 16 .. +16 of the 384-row slab image, lane -> (row, 16-B chunk)
    int src_off[NPIECE];
    for (int p = 0; p < NPIECE; ++p) {
        const int row = (wave * NPIECE + p) * 16 + lane / 4;
        src_off[p] = row < 128 ? row * 256 + (lane % 4) * 4            // A rows: lda = 256 floats
                               : (row - 128) * 256 + (lane % 4) * 4;    // W rows
    }
    auto stream = [&](const float* a_tile, int kt, int buf) {
        float* base = smem + buf * SLAB;
#pragma unroll
        for (int p = 0; p < NPIECE; ++p) {
            const int row0 = (wave * NPIECE + p) * 16;
            const float* src = (row0 < 128 ? a_tile : W) + src_off[p] + kt * 16;
            GLDS16(src, base + row0 * 16 + 0);
        }
    };
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const float* a_tile = A + (size_t)tile * 128 * 256;
        stream(a_tile, 0, 0);
        stream(a_tile, 1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            const float* sb = smem + (kt % 3) * SLAB;
            const bool more = kt + 2 < nk;
            if (V == 0) {
                const int li = lane & 31, lh = lane >> 5, sw = (li >> 2) & 3;
#pragma unroll
                for (int s4 = 0; s4 < 2; ++s4) {
                    f32x4v a = *reinterpret_cast<const f32x4v*>(sb + (wave * 32 + li) * 16 + (((2 * s4 + lh) ^ sw) << 2));
#pragma unroll
                    for (int t = 0; t < 8; t += 2) {
                        f32x4v b0 = *reinterpret_cast<const f32x4v*>(sb + 128 * 16 + (t * 32 + li) * 16 + (((2 * s4 + lh) ^ sw) << 2));
                        f32x4v b1 = *reinterpret_cast<const f32x4v*>(sb + 128 * 16 + ((t + 1) * 32 + li) * 16 + (((2 * s4 + lh) ^ sw) << 2));
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc32[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b0[j], acc32[t], 0, 0, 0);
                            acc32[t + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b1[j], acc32[t + 1], 0, 0, 0);
                        }
                        if (more && (s4 * 4 + t / 2) < NPIECE) {
                            const int p = s4 * 4 + t / 2, row0 = (wave * NPIECE + p) * 16;
                            GLDS16((row0 < 128 ? a_tile : W) + src_off[p] + (kt + 2) * 16,
                                   smem + ((kt + 2) % 3) * SLAB + row0 * 16);
                        }
                    }
                }
            } else {
                const int li = lane & 15, lg = lane >> 4;
                const int sw = (0x1230 >> (((li >> 2) & 3) * 4)) & 3;      // f = [0,3,2,1]: conflict-free b128 groups
                f32x4v a = *reinterpret_cast<const f32x4v*>(sb + (wave * 16 + li) * 16 + ((lg ^ sw) << 2));
#pragma unroll
                for (int t = 0; t < 16; t += 2) {
                    f32x4v b0 = *reinterpret_cast<const f32x4v*>(sb + 128 * 16 + (t * 16 + li) * 16 + ((lg ^ sw) << 2));
                    f32x4v b1 = *reinterpret_cast<const f32x4v*>(sb + 128 * 16 + ((t + 1) * 16 + li) * 16 + ((lg ^ sw) << 2));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc16[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b0[j], acc16[t], 0, 0, 0);
                        acc16[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b1[j], acc16[t + 1], 0, 0, 0);
                    }
                    if (more && t / 2 < NPIECE) {
                        const int p = t / 2, row0 = (wave * NPIECE + p) * 16;
                        GLDS16((row0 < 128 ? a_tile : W) + src_off[p] + (kt + 2) * 16,
                               smem + ((kt + 2) % 3) * SLAB + row0 * 16);
                    }
                }
            }
            if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    float s = 0.f;
    for (auto& a : acc32) for (int r = 0; r < 16; ++r) s += a[r];
    for (auto& a : acc16) s += a[0] + a[1] + a[2] + a[3];
    out[blockIdx.x * blockDim.x + tid] = s;
}

__global__ void fill_random(float* x, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        x[i] = ((h & 0xffffff) / 8388608.0f - 1.0f);      // uniform in [-1, 1): realistic toggling in the MFMA datapath
    }
}

static bool g_random = false;

template <int V>
void run(const char* name, int nk, int grid) {
    const int tiles = 15800;
    float *A, *W, *out;
    hipMalloc(&A, (size_t)tiles * 128 * 256 * 4 + (4 << 20));   // rows overlap for nk > 16: only the address stream matters
    hipMalloc(&W, 256 * 1024 * 4 * 4);
    hipMalloc(&out, (size_t)16384 * 512 * 4);
    hipMemset(A, 0, (size_t)tiles * 128 * 256 * 4 + (4 << 20));
    hipMemset(W, 0, 256 * 1024 * 4 * 4);
    if (g_random) {
        fill_random<<<4096, 256>>>(A, (size_t)tiles * 128 * 256 + (1 << 20), 1u);
        fill_random<<<256, 256>>>(W, 256 * 1024 * 4, 2u);
    }
    hipFuncSetAttribute((const void*)probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * SLAB * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<V><<<512, V ? 512 : 256, 3 * SLAB * 4>>>(A, W, out, 512, nk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<V><<<grid, V ? 512 : 256, 3 * SLAB * 4>>>(A, W, out, tiles, nk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)tiles * nk * 2.0 * 128 * 256 * 16;
    printf("%s %-36s grid=%5d nk=%2d  %.1f TF  (%.3f ms)  %s\n", g_random ? "random" : "zeros ", name, grid, nk, flops / ms / 1e9, ms, hipGetErrorString(hipGetLastError()));
    hipFree(A); hipFree(W); hipFree(out);
}

int main() {
  for (int rnd = 0; rnd < 2; ++rnd) {
    g_random = rnd;
    for (int nk : {16, 64})
        for (int grid : {512, 15800}) {      // persistent (2 workgroups per CU looping over tiles) vs one tile per workgroup
            run<0>("4 waves x 32x32x2 (32 rows/wave)", nk, grid);
            run<1>("8 waves x 16x16x4 (16 rows/wave)", nk, grid);
        }
  }
    return 0;
}
