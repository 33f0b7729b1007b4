// How many workgroups of a given shape does a CU of this part really hold?  Every workgroup spins for a fixed number
// of shader cycles; elapsed time / spin time = rounds = ceil(workgroups per CU / resident workgroups).
// Build: hipcc --offload-arch=gfx950 -O3 occupancy_probe.hip -o occupancy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int VG>
__global__ void spin(unsigned long long cycles, float* out, int lds_touch) {
    extern __shared__ float sm[];
    if (lds_touch) sm[threadIdx.x] = 1.f;
    float keep[VG];
    for (int i = 0; i < VG; ++i) keep[i] = threadIdx.x * 1e-3f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) {
        for (int i = 0; i < VG; ++i) asm volatile("" : "+v"(keep[i]));
    }
    float s = 0;
    for (int i = 0; i < VG; ++i) s += keep[i];
    if (s == 123.456f) out[0] = s;
}

template <int VG>
void run(int threads, size_t lds, int per_cu) {
    float* out; hipMalloc(&out, 4);
    hipFuncSetAttribute((const void*)spin<VG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const unsigned long long cyc = 200000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    spin<VG><<<256, threads, lds>>>(cyc, out, lds > 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    spin<VG><<<256 * per_cu, threads, lds>>>(cyc, out, lds > 0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms, ms1;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventRecord(e0);
    spin<VG><<<256, threads, lds>>>(cyc, out, lds > 0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms1, e0, e1);
    printf("threads %4d (%2d waves)  ~%3d VGPRs  LDS %6zu B  %2d workgroups per CU: %.3f ms = %.2f rounds of %.3f ms -> ~%.1f resident\n",
           threads, threads / 64, VG + 8, lds, per_cu, ms, ms / ms1, ms1, per_cu / (ms / ms1));
    hipFree(out);
}

int main() {
    run<32>(448, 30720, 24);
    run<32>(448, 0, 24);
    run<32>(512, 0, 24);
    run<32>(384, 0, 24);
    run<32>(256, 0, 24);
    run<32>(896, 61440, 24);
    run<32>(896, 0, 24);
    run<32>(64, 0, 48);
    run<100>(448, 0, 24);
    return 0;
}
