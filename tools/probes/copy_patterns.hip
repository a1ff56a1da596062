// What does a read + write stream reach on this box, and does the access pattern matter?  Every kernel of the metric step runs at
// 5.2-5.4 TB/s of its own bytes, the grid-stride copy of spmm_patch_probe at 5.21 -- but torch.add on 1 GiB tensors was measured at
// 6.28 TB/s in round 2.  Variants of one 2 x 514 MB copy (the plane of the bench step: 5 x 50 176 rows of 512 floats):
//   hipcc --offload-arch=gfx950 -O3 copy_patterns.hip -o copy_patterns && ./copy_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr size_t ROWS = 5 * 50176, F = 512, N4 = ROWS * F / 4;

template <bool NT_ST, bool NT_LD>
__global__ void stride_copy(const f32x4* __restrict__ X, f32x4* __restrict__ Y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = NT_LD ? __builtin_nontemporal_load(X + i) : X[i];
        if (NT_ST) __builtin_nontemporal_store(v, Y + i); else Y[i] = v;
    }
}

// a workgroup copies one contiguous chunk of CH vectors (UNROLL loads in flight per lane); XCD: blockIdx dealt so that each XCD walks a band
template <int UNROLL, bool NT_ST, bool XCD>
__global__ __launch_bounds__(256) void chunk_copy(const f32x4* __restrict__ X, f32x4* __restrict__ Y, size_t n, int n_wg) {
    int w = blockIdx.x;
    if (XCD) {
        const int per = (n_wg + 7) / 8;
        w = (blockIdx.x % 8) * per + blockIdx.x / 8;
        if (w >= n_wg) return;
    }
    const size_t base = (size_t)w * 256 * UNROLL + threadIdx.x;
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = base + u * 256 < n ? X[base + u * 256] : f32x4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
        if (base + u * 256 < n) { if (NT_ST) __builtin_nontemporal_store(v[u], Y + base + u * 256); else Y[base + u * 256] = v[u]; }
}

// one wave = one 2 KiB row at a time (the node kernels' unit), rows dealt to waves round-robin over the grid, PF rows requested ahead
template <int PF, bool NT_ST>
__global__ __launch_bounds__(256) void row_copy(const f32x4* __restrict__ X, f32x4* __restrict__ Y, int rows) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6), waves = gridDim.x * 4;
    f32x4 v[PF][2];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int r = min(wave + p * waves, rows - 1);
        v[p][0] = X[(size_t)r * 128 + lane]; v[p][1] = X[(size_t)r * 128 + 64 + lane];
    }
    for (int r0 = wave; r0 < rows; r0 += PF * waves) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int r = r0 + p * waves, rn = min(r + PF * waves, rows - 1);
            const f32x4 a = v[p][0], b = v[p][1];
            v[p][0] = X[(size_t)rn * 128 + lane]; v[p][1] = X[(size_t)rn * 128 + 64 + lane];
            if (r < rows) {
                if (NT_ST) { __builtin_nontemporal_store(a, Y + (size_t)r * 128 + lane); __builtin_nontemporal_store(b, Y + (size_t)r * 128 + 64 + lane); }
                else { Y[(size_t)r * 128 + lane] = a; Y[(size_t)r * 128 + 64 + lane] = b; }
            }
        }
    }
}

__global__ void read_only(const f32x4* __restrict__ X, float* out, size_t n) {
    f32x4 s = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += X[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.f) *out = 1.f;
}
template <bool NT_ST>
__global__ void write_only(f32x4* __restrict__ Y, size_t n) {
    const f32x4 v = {1, 2, 3, 4};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (NT_ST) __builtin_nontemporal_store(v, Y + i); else Y[i] = v;
    }
}

int main() {
    f32x4 *X[3], *Y[3];
    float* out;
    hipMalloc(&out, 4);
    for (int i = 0; i < 3; ++i) {
        hipMalloc(&X[i], N4 * 16); hipMalloc(&Y[i], N4 * 16);
        hipMemset(X[i], 0, N4 * 16); hipMemset(Y[i], 0, N4 * 16);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch, const char* what, double streams) {
        for (int i = 0; i < 6; ++i) launch(i % 3);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int it = 60;
        for (int i = 0; i < it; ++i) launch(i % 3);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = 1e3 * ms / it;
        printf("%-64s %7.1f us = %.2f TB/s\n", what, us, streams * N4 * 16 / us / 1e6);
    };
#define L(k, g, b, ...) [&](int i) { hipLaunchKernelGGL(k, dim3(g), dim3(b), 0, 0, __VA_ARGS__); }
    time(L((stride_copy<true, false>), 4096, 256, X[i], Y[i], N4), "grid-stride 4096 x 256, nt store", 2);
    time(L((stride_copy<false, false>), 4096, 256, X[i], Y[i], N4), "grid-stride 4096 x 256, plain store", 2);
    time(L((stride_copy<true, true>), 4096, 256, X[i], Y[i], N4), "grid-stride 4096 x 256, nt load + nt store", 2);
    time(L((stride_copy<true, false>), 1024, 256, X[i], Y[i], N4), "grid-stride 1024 x 256, nt store", 2);
    time(L((stride_copy<true, false>), 2048, 512, X[i], Y[i], N4), "grid-stride 2048 x 512, nt store", 2);
    time(L((stride_copy<true, false>), 16384, 256, X[i], Y[i], N4), "grid-stride 16384 x 256, nt store", 2);
    { const int n_wg = (int)((N4 + 256 * 4 - 1) / (256 * 4));
      time(L((chunk_copy<4, false, false>), n_wg, 256, X[i], Y[i], N4, n_wg), "chunk 16 KiB per workgroup (torch-like), plain store", 2);
      time(L((chunk_copy<4, true, false>), n_wg, 256, X[i], Y[i], N4, n_wg), "chunk 16 KiB per workgroup, nt store", 2);
      time(L((chunk_copy<4, true, true>), n_wg, 256, X[i], Y[i], N4, n_wg), "chunk 16 KiB per workgroup, nt store, XCD bands", 2); }
    { const int n_wg = (int)((N4 + 256 * 8 - 1) / (256 * 8));
      time(L((chunk_copy<8, false, false>), n_wg, 256, X[i], Y[i], N4, n_wg), "chunk 32 KiB per workgroup, plain store", 2);
      time(L((chunk_copy<8, true, false>), n_wg, 256, X[i], Y[i], N4, n_wg), "chunk 32 KiB per workgroup, nt store", 2); }
    time(L((row_copy<2, true>), 2048, 256, X[i], Y[i], (int)ROWS), "row per wave, 2 rows ahead, 2048 x 256, nt store", 2);
    time(L((row_copy<4, true>), 2048, 256, X[i], Y[i], (int)ROWS), "row per wave, 4 rows ahead, 2048 x 256, nt store", 2);
    time(L((row_copy<4, false>), 2048, 256, X[i], Y[i], (int)ROWS), "row per wave, 4 rows ahead, 2048 x 256, plain store", 2);
    time(L((row_copy<2, true>), 256, 256, X[i], Y[i], (int)ROWS), "row per wave, 2 rows ahead, 256 x 256 (one wave per SIMD), nt", 2);
    time(L((row_copy<4, true>), 256, 256, X[i], Y[i], (int)ROWS), "row per wave, 4 rows ahead, 256 x 256 (one wave per SIMD), nt", 2);
    time(L(read_only, 4096, 256, X[i], out, N4), "read only, grid-stride", 1);
    time(L((write_only<true>), 4096, 256, Y[i], N4), "write only, nt", 1);
    time(L((write_only<false>), 4096, 256, Y[i], N4), "write only, plain", 1);
    return 0;
}
