// LDS-staged "patch" SpMM on the bench's aggregation unit (224 x 224 queen grid, B = 5, F = 512): sweep of patch shape (TY x TX nodes) and
// column chunk (FC floats) -- which staging granularity gets closest to the chip's copy rate?  Successor of spmm_patch_probe.hip (round 3).
//   hipcc --offload-arch=gfx950 -O3 spmm_patch_sweep.hip -o spmm_patch_sweep && ./spmm_patch_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int G = 224, N = G * G, F = 512, B = 5;

template <int TY, int TX, int FC, int THREADS>
__global__ __launch_bounds__(THREADS) void patch_kernel(const float* __restrict__ X, float* __restrict__ Y, int n_wg) {
    constexpr int HY = TY + 2, HX = TX + 2, Q = FC / 4, TILES_X = G / TX, TILES = (G / TY) * TILES_X, CHUNKS = F / FC;
    extern __shared__ f32x4 halo[];                  // [HY * HX][Q]
    const int per_xcd = (n_wg + 7) / 8, w = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (w >= n_wg) return;
    const int chunk = w % CHUNKS, tile = (w / CHUNKS) % TILES, b = w / (CHUNKS * TILES);
    const int ty0 = (tile / TILES_X) * TY, tx0 = (tile % TILES_X) * TX;
    const float* Xb = X + (size_t)b * N * F + chunk * FC;
    for (int i = threadIdx.x; i < HY * HX * Q; i += THREADS) {
        const int hr = i / Q, q = i % Q, gy = ty0 + hr / HX - 1, gx = tx0 + hr % HX - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gy >= 0 && gy < G && gx >= 0 && gx < G) v = *reinterpret_cast<const f32x4*>(Xb + (size_t)(gy * G + gx) * F + 4 * q);
        halo[hr * Q + q] = v;
    }
    __syncthreads();
    float* Yb = Y + (size_t)b * N * F + chunk * FC;
    for (int i = threadIdx.x; i < TY * TX * Q; i += THREADS) {
        const int r = i / Q, q = i % Q, ly = r / TX, lx = r % TX;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) s += (0.111f + 0.001f * (dy * 3 + dx)) * halo[((ly + dy) * HX + lx + dx) * Q + q];
        __builtin_nontemporal_store(s, reinterpret_cast<f32x4*>(Yb + (size_t)((ty0 + ly) * G + tx0 + lx) * F + 4 * q));
    }
}

// the same with the halo of the NEXT chunk requested into registers while the current chunk is summed (one workgroup walks all chunks of a patch)
template <int TY, int TX, int FC, int THREADS>
__global__ __launch_bounds__(THREADS) void patch_pipe_kernel(const float* __restrict__ X, float* __restrict__ Y, int n_wg) {
    constexpr int HY = TY + 2, HX = TX + 2, Q = FC / 4, TILES_X = G / TX, TILES = (G / TY) * TILES_X, CHUNKS = F / FC;
    constexpr int PER = (HY * HX * Q + THREADS - 1) / THREADS;
    extern __shared__ f32x4 halo[];
    const int per_xcd = (n_wg + 7) / 8, w = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (w >= n_wg) return;
    const int tile = w % TILES, b = w / TILES;
    const int ty0 = (tile / TILES_X) * TY, tx0 = (tile % TILES_X) * TX;
    f32x4 nx[PER];
    auto request = [&](int chunk) {
        const float* Xb = X + (size_t)b * N * F + chunk * FC;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * THREADS;
            const int hr = i / Q, q = i % Q, gy = ty0 + hr / HX - 1, gx = tx0 + hr % HX - 1;
            nx[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < HY * HX * Q && gy >= 0 && gy < G && gx >= 0 && gx < G) nx[k] = *reinterpret_cast<const f32x4*>(Xb + (size_t)(gy * G + gx) * F + 4 * q);
        }
    };
    request(0);
    for (int chunk = 0; chunk < CHUNKS; ++chunk) {
        __syncthreads();                                 // the previous chunk's sums are done with the tile
#pragma unroll
        for (int k = 0; k < PER; ++k) { const int i = threadIdx.x + k * THREADS; if (i < HY * HX * Q) halo[i] = nx[k]; }
        __syncthreads();
        if (chunk + 1 < CHUNKS) request(chunk + 1);
        float* Yb = Y + (size_t)b * N * F + chunk * FC;
        for (int i = threadIdx.x; i < TY * TX * Q; i += THREADS) {
            const int r = i / Q, q = i % Q, ly = r / TX, lx = r % TX;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) s += (0.111f + 0.001f * (dy * 3 + dx)) * halo[((ly + dy) * HX + lx + dx) * Q + q];
            __builtin_nontemporal_store(s, reinterpret_cast<f32x4*>(Yb + (size_t)((ty0 + ly) * G + tx0 + lx) * F + 4 * q));
        }
    }
}

__global__ void copy_kernel(const f32x4* __restrict__ X, f32x4* __restrict__ Y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(X[i], Y + i);
}

float *Xs[2], *Ys[2];
hipEvent_t e0, e1;
template <class L>
void timeit(L launch, const char* what) {
    for (int i = 0; i < 4; ++i) launch(i & 1);
    hipEventRecord(e0);
    const int reps = 30;
    for (int i = 0; i < reps; ++i) launch(i & 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, n = (double)B * N * F;
    printf("%-52s %7.1f us = %.2f TB/s = %.3f of peak\n", what, us, 2.0 * n * 4 / us / 1e6, 2.0 * n * 4 / us / 1e6 / 8.0);
    fflush(stdout);
}
template <int TY, int TX, int FC, int THREADS>
void run() {
    constexpr int TILES = (G / TY) * (G / TX), CHUNKS = F / FC;
    const size_t lds = (size_t)(TY + 2) * (TX + 2) * FC * 4;
    char name[96];
    {
        const int n_wg = TILES * CHUNKS * B, grid = (n_wg + 7) / 8 * 8;
        auto k = patch_kernel<TY, TX, FC, THREADS>;
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        snprintf(name, sizeof name, "patch %dx%d x %3d cols, %3d thr, %5.1f KB LDS, %.2f fetch", TY, TX, FC, THREADS, lds / 1024.0, (TY + 2.0) * (TX + 2) / (TY * TX));
        timeit([&](int i) { hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, 0, Xs[i], Ys[i], n_wg); }, name);
    }
    if (CHUNKS > 1) {
        const int n_wg = TILES * B, grid = (n_wg + 7) / 8 * 8;
        auto k = patch_pipe_kernel<TY, TX, FC, THREADS>;
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        snprintf(name, sizeof name, "  pipelined over %d chunks", CHUNKS);
        timeit([&](int i) { hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, 0, Xs[i], Ys[i], n_wg); }, name);
    }
}

int main() {
    const size_t n = (size_t)B * N * F;
    for (int i = 0; i < 2; ++i) { hipMalloc(&Xs[i], n * 4); hipMalloc(&Ys[i], n * 4); hipMemset(Xs[i], 0, n * 4); }
    hipEventCreate(&e0); hipEventCreate(&e1);
    timeit([&](int i) { hipLaunchKernelGGL(copy_kernel, dim3(4096), dim3(256), 0, 0, (const f32x4*)Xs[i], (f32x4*)Ys[i], n / 4); }, "plain copy, grid-stride 4096 x 256");
    timeit([&](int i) { hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, 0, (const f32x4*)Xs[i], (f32x4*)Ys[i], n / 4); }, "plain copy, grid-stride 1024 x 256");
    run<4, 8, 128, 256>();
    run<4, 8, 256, 256>();
    run<4, 8, 128, 512>();
    run<2, 8, 256, 256>();
    run<2, 8, 512, 256>();
    run<4, 4, 256, 256>();
    run<4, 4, 512, 256>();
    run<2, 4, 512, 256>();
    run<8, 8, 128, 256>();
    run<8, 8, 64, 256>();
    run<4, 16, 128, 256>();
    run<2, 16, 256, 256>();
    run<2, 2, 512, 128>();
    printf("row-blocked kernel in the bench step: 211 us for the same unit (4.87 TB/s = 0.609)\n");
    return 0;
}
