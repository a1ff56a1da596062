// How fast does v_mfma_f32_16x16x4_f32 really issue in launches shaped like the small-graph cell kernels (32 workgroups, 16 waves each, tens
// of microseconds)?  N independent-accumulator MFMAs per wave, timed with HIP events; prints cycles/instruction/SIMD assuming 2.4 GHz, i.e.
// the effective clock if the issue rate is the documented 32 cycles.   hipcc --offload-arch=gfx950 -O3 mfma_f32_rate.hip -o mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ __launch_bounds__(1024) void k(float* out, int n, float a, float b) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int blocks : {32, 256})
        for (int threads : {256, 1024})
            for (int n : {64, 256, 1024, 8192}) {
                for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, n, 1.f, 2.f);
                hipEventRecord(e0);
                const int reps = 20;
                for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, n, 1.f, 2.f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double us = ms * 1e3 / reps, per_simd = 4.0 * n * (threads / 64) / 4.0;      // MFMAs per SIMD per launch
                printf("blocks %3d threads %4d n %5d: %8.2f us per launch, %6.1f ns per MFMA per SIMD = %5.1f cycles at 2.4 GHz\n", blocks, threads, n, us,
                       us * 1e3 / per_simd, us * 1e3 / per_simd * 2.4);
            }
    return 0;
}
