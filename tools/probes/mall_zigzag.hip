// Probe: does a consumer that reads a plane in the REVERSE order of its producer's writes find the freshest part in the 256 MB Infinity Cache?
// producer: y = x + 1 over chunks in ascending order; consumer: z = y + 1 over chunks ascending (as every kernel of the step does) or descending.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mall_zigzag.hip -o /tmp/mall_zigzag && /tmp/mall_zigzag
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f4 = __attribute__((ext_vector_type(4))) float;
constexpr int THREADS = 256, PER = 8;          // 8 x 16 B per thread, 32 KiB per workgroup, contiguous
template <bool NT>
__global__ __launch_bounds__(THREADS) void add1(const f4* __restrict__ x, f4* __restrict__ y, long long n4, int reverse) {
    const long long nb = gridDim.x, b = reverse ? nb - 1 - blockIdx.x : blockIdx.x;
    const long long base = b * THREADS * PER + threadIdx.x;
    f4 v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = base + (long long)i * THREADS < n4 ? (NT ? __builtin_nontemporal_load(x + base + (long long)i * THREADS) : x[base + (long long)i * THREADS]) : f4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < PER; ++i)
        if (base + (long long)i * THREADS < n4) {
            if (NT) __builtin_nontemporal_store(v[i] + 1.f, y + base + (long long)i * THREADS); else y[base + (long long)i * THREADS] = v[i] + 1.f;
        }
}
int main() {
    for (int mb : {64, 128, 256, 514, 822, 1644}) {
        const long long n4 = (long long)mb * 1000000 / 16;
        f4 *a, *b, *c;
        hipMalloc(&a, n4 * 16); hipMalloc(&b, n4 * 16); hipMalloc(&c, n4 * 16);
        hipMemset(a, 0, n4 * 16);
        const int grid = (int)((n4 + THREADS * PER - 1) / (THREADS * PER));
        hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
        for (int nt = 1; nt >= 0; --nt)
        for (int reverse = 0; reverse < 2; ++reverse) {
            auto kern = nt ? add1<true> : add1<false>;
            float best_p = 1e9f, best_c = 1e9f;
            for (int it = 0; it < 12; ++it) {
                float tp, tc;
                hipEventRecord(s); hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), 0, 0, a, b, n4, 0); hipEventRecord(e); hipEventSynchronize(e); hipEventElapsedTime(&tp, s, e);
                hipEventRecord(s); hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), 0, 0, b, c, n4, reverse); hipEventRecord(e); hipEventSynchronize(e); hipEventElapsedTime(&tc, s, e);
                if (it >= 2) { best_p = tp < best_p ? tp : best_p; best_c = tc < best_c ? tc : best_c; }
            }
            printf("%5d MB planes  %s  consumer %s: producer %7.1f us, consumer %7.1f us  (%.2f TB/s)\n", mb, nt ? "non-temporal" : "default     ", reverse ? "descending" : "ascending ", best_p * 1e3, best_c * 1e3,
                   2.0 * mb / (best_c * 1e3));
        }
        hipFree(a); hipFree(b); hipFree(c);
    }
    return 0;
}
