// Would an LDS-staged "patch" SpMM beat the row-blocked gather?  Standalone probe on the bench's own unit: Y = S.X on the 224 x 224 queen
// grid (9 entries per interior row), B = 5 samples, rows of F = 512 floats (1.03 GB per launch).  A workgroup = a 4 x 8 patch of grid nodes
// x a chunk of 128 feature columns: the 6 x 10 halo of rows is staged in LDS once (60 rows for 32 outputs = 1.9 fetches per output row
// instead of the row-blocked kernel's 4.46, all but the first from L2), then every output row sums its 9 neighbours from LDS.
//   hipcc --offload-arch=gfx950 -O3 spmm_patch_probe.hip -o spmm_patch_probe && ./spmm_patch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int G = 224, N = G * G, F = 512, B = 5, TY = 4, TX = 8, FC = 128, HY = TY + 2, HX = TX + 2;
constexpr int TILES_Y = G / TY, TILES_X = G / TX, TILES = TILES_Y * TILES_X, CHUNKS = F / FC;

__global__ __launch_bounds__(256) void patch_kernel(const float* __restrict__ X, float* __restrict__ Y, int n_wg) {
    __shared__ f32x4 halo[HY * HX][FC / 4];
    // XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs; give each XCD a contiguous band of (tile, chunk, sample) work
    const int per_xcd = (n_wg + 7) / 8, w = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (w >= n_wg) return;
    const int chunk = w % CHUNKS, tile = (w / CHUNKS) % TILES, b = w / (CHUNKS * TILES);
    const int ty0 = (tile / TILES_X) * TY, tx0 = (tile % TILES_X) * TX;
    const float* Xb = X + (size_t)b * N * F + chunk * FC;
    for (int i = threadIdx.x; i < HY * HX * (FC / 4); i += 256) {
        const int hr = i / (FC / 4), q = i % (FC / 4), gy = ty0 + hr / HX - 1, gx = tx0 + hr % HX - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gy >= 0 && gy < G && gx >= 0 && gx < G) v = *reinterpret_cast<const f32x4*>(Xb + (size_t)(gy * G + gx) * F + 4 * q);
        halo[hr][q] = v;
    }
    __syncthreads();
    float* Yb = Y + (size_t)b * N * F + chunk * FC;
    for (int i = threadIdx.x; i < TY * TX * (FC / 4); i += 256) {
        const int r = i / (FC / 4), q = i % (FC / 4), ly = r / TX, lx = r % TX;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const f32x4 v = halo[(ly + dy) * HX + lx + dx][q];
                const float wgt = 0.111f + 0.001f * (dy * 3 + dx);
                s += wgt * v;
            }
        __builtin_nontemporal_store(s, reinterpret_cast<f32x4*>(Yb + (size_t)((ty0 + ly) * G + tx0 + lx) * F + 4 * q));
    }
}

__global__ void copy_kernel(const f32x4* __restrict__ X, f32x4* __restrict__ Y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(X[i], Y + i);
}

int main() {
    const size_t n = (size_t)B * N * F;
    float *X[2], *Y[2];
    for (int i = 0; i < 2; ++i) {
        hipMalloc(&X[i], n * 4);
        hipMalloc(&Y[i], n * 4);
        hipMemset(X[i], 0, n * 4);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int n_wg = TILES * CHUNKS * B, grid = (n_wg + 7) / 8 * 8;
    auto time = [&](auto launch, const char* what) {
        for (int i = 0; i < 4; ++i) launch(i & 1);
        hipEventRecord(e0);
        const int reps = 40;
        for (int i = 0; i < reps; ++i) launch(i & 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps;
        printf("%-28s %7.1f us per launch = %.2f TB/s of 2 x %.0f MB\n", what, us, 2.0 * n * 4 / us / 1e6, n * 4 / 1e6);
    };
    time([&](int i) { hipLaunchKernelGGL(patch_kernel, dim3(grid), dim3(256), 0, 0, X[i], Y[i], n_wg); }, "patch SpMM (4x8 x 128 cols)");
    time([&](int i) { hipLaunchKernelGGL(copy_kernel, dim3(4096), dim3(256), 0, 0, (const f32x4*)X[i], (f32x4*)Y[i], n / 4); }, "plain copy");
    printf("row-blocked kernel in the bench step: 214-216 us for the same unit (4.77 TB/s)\n");
    return 0;
}
