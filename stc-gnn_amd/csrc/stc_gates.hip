// GRU gate math of STC_Cell (STC_GNN.py:68-78), the matrix-side Chebyshev set of the
// small category graph (STC_GNN.py:24-29) and the pointwise helpers -- gfx950.
// All of these are streaming, HBM-bound kernels: one pass over each operand.
#include "stc_common.h"

namespace {

constexpr int EW_THREADS = 256;

inline dim3 ew_grid(long long n) {
    long long blocks = (n + EW_THREADS - 1) / EW_THREADS;
    const long long cap = (long long)stc::kNumCu * 8;     // grid-stride beyond 2048 blocks
    return dim3((unsigned)(blocks < cap ? (blocks > 0 ? blocks : 1) : cap));
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ---- gates: U = sigmoid(G[:, :h]); Rg = sigmoid(G[:, h:]); CandIn = [Xt | Rg*H]
__global__ __launch_bounds__(EW_THREADS) void gru_gates_fwd_kernel(
    const float* __restrict__ G, const float* __restrict__ Xt, const float* __restrict__ H,
    float* __restrict__ U, float* __restrict__ Rg, float* __restrict__ CandIn,
    long long rows, int cin, int h, int pad) {
    const int L = cin + h + pad;
    const long long n = rows * L;
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS) {
        const long long r = e / L;
        const int l = (int)(e - r * L);
        if (l < cin) {
            CandIn[e] = Xt[r * cin + l];
        } else if (l >= cin + h) {
            CandIn[e] = 0.f;
        } else {
            const int k = l - cin;
            const float u = sigmoidf_(G[r * 2 * h + k]);
            const float g = sigmoidf_(G[r * 2 * h + h + k]);
            U[r * h + k] = u;
            Rg[r * h + k] = g;
            CandIn[e] = g * H[r * h + k];
        }
    }
}

__global__ __launch_bounds__(EW_THREADS) void gru_gates_bwd_kernel(
    const float* __restrict__ dCandIn, const float* __restrict__ dU, const float* __restrict__ H,
    const float* __restrict__ U, const float* __restrict__ Rg,
    float* __restrict__ dG, float* __restrict__ dXt, float* __restrict__ dH,
    long long rows, int cin, int h, int pad) {
    const int L = cin + h + pad;
    const long long n = rows * L;
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS) {
        const long long r = e / L;
        const int l = (int)(e - r * L);
        if (l < cin) {
            dXt[r * cin + l] = dCandIn[e];
        } else if (l < cin + h) {
            const int k = l - cin;
            const long long i = r * h + k;
            const float u = U[i], g = Rg[i], d = dCandIn[e];
            dG[r * 2 * h + k] = dU[i] * u * (1.f - u);
            dG[r * 2 * h + h + k] = d * H[i] * g * (1.f - g);
            dH[i] = d * g;
        }
    }
}

// ---- blend: Cand = tanh(Cpre); Hnew = (1-U)*H + U*Cand
__global__ __launch_bounds__(EW_THREADS) void gru_blend_fwd_kernel(
    const float* __restrict__ Cpre, const float* __restrict__ U, const float* __restrict__ H,
    float* __restrict__ Cand, float* __restrict__ Hnew, long long n) {
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS) {
        const float c = tanhf(Cpre[e]);
        const float u = U[e];
        Cand[e] = c;
        Hnew[e] = (1.f - u) * H[e] + u * c;
    }
}

__global__ __launch_bounds__(EW_THREADS) void gru_blend_bwd_kernel(
    const float* __restrict__ dHnew, const float* __restrict__ U, const float* __restrict__ H,
    const float* __restrict__ Cand, float* __restrict__ dCpre, float* __restrict__ dU, float* __restrict__ dH,
    long long n) {
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS) {
        const float g = dHnew[e], u = U[e], c = Cand[e];
        dCpre[e] = g * u * (1.f - c * c);
        dU[e] = g * (c - H[e]);
        dH[e] = g * (1.f - u);
    }
}

__global__ __launch_bounds__(EW_THREADS) void axpy_kernel(float a, const float* __restrict__ x, float* y, long long n) {
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS)
        y[e] = fmaf(a, x[e], y[e]);
}

__global__ __launch_bounds__(EW_THREADS) void concat2_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                              float* __restrict__ out, long long rows, int a, int b, int pad) {
    const int L = a + b + pad;
    const long long n = rows * L;
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS) {
        const long long r = e / L;
        const int l = (int)(e - r * L);
        out[e] = l < a ? A[r * a + l] : (l < a + b ? B[r * b + (l - a)] : 0.f);
    }
}

__global__ __launch_bounds__(EW_THREADS) void split2_kernel(const float* __restrict__ src, float* __restrict__ A,
                                                             float* __restrict__ B, long long rows, int a, int b, int pad) {
    const int L = a + b + pad;
    const long long n = rows * L;
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS) {
        const long long r = e / L;
        const int l = (int)(e - r * L);
        if (l < a) A[r * a + l] = src[e];
        else if (l < a + b) B[r * b + (l - a)] = src[e];
    }
}

// ---- dense Chebyshev set of the category graph (n <= 128, a handful of n x n products)
__global__ __launch_bounds__(EW_THREADS) void cheby_init_kernel(const float* __restrict__ G, int n, int K, float* __restrict__ T) {
    const int e = blockIdx.x * EW_THREADS + threadIdx.x;
    if (e >= n * n) return;
    const int i = e / n, j = e - i * n;
    T[e] = i == j ? 1.f : 0.f;
    if (K > 1) T[n * n + e] = G[e];
}

// Tk = (2G) Tkm1 - Tkm2, same operand order as the reference's torch.mm(2*G, T[-1]) - T[-2]
__global__ __launch_bounds__(EW_THREADS) void cheby_step_kernel(const float* __restrict__ G, const float* __restrict__ Tkm1,
                                                                 const float* __restrict__ Tkm2, float* __restrict__ Tk, int n) {
    const int e = blockIdx.x * EW_THREADS + threadIdx.x;
    if (e >= n * n) return;
    const int i = e / n, j = e - i * n;
    float s = 0.f;
    for (int m = 0; m < n; ++m) s = fmaf(2.f * G[i * n + m], Tkm1[m * n + j], s);
    Tk[e] = s - Tkm2[e];
}

// one reverse step k: dG += 2 dTk Tkm1^T ; dTkm1 += 2 G^T dTk ; dTkm2 -= dTk
__global__ __launch_bounds__(EW_THREADS) void cheby_bwd_step_kernel(const float* __restrict__ G, const float* __restrict__ Tkm1,
                                                                     const float* __restrict__ dTk, float* dTkm1, float* dTkm2,
                                                                     float* dG, int n) {
    const int e = blockIdx.x * EW_THREADS + threadIdx.x;
    if (e >= n * n) return;
    const int i = e / n, j = e - i * n;
    float sg = 0.f, st = 0.f;
    for (int m = 0; m < n; ++m) {
        sg = fmaf(dTk[i * n + m], Tkm1[j * n + m], sg);
        st = fmaf(G[m * n + i], dTk[m * n + j], st);
    }
    dG[e] += 2.f * sg;
    dTkm1[e] += 2.f * st;
    dTkm2[e] -= dTk[e];
}

__global__ __launch_bounds__(EW_THREADS) void cheby_bwd_final_kernel(const float* __restrict__ dT1, float* dG, int nn, int add) {
    const int e = blockIdx.x * EW_THREADS + threadIdx.x;
    if (e >= nn) return;
    dG[e] = add ? dG[e] + dT1[e] : dT1[e];
}

}  // namespace

#define STC_EW_PROLOGUE(name, n_expr, ...)                                             \
    STC_REQUIRE((n_expr) >= 0, STC_EINVAL, name ": negative size");                    \
    if ((n_expr) == 0) return STC_OK;                                                  \
    STC_REQUIRE(__VA_ARGS__, STC_EINVAL, name ": null pointer");                       \
    hipStream_t s = static_cast<hipStream_t>(stream)

extern "C" int stc_gru_gates_fwd_f32(const float* G, const float* Xt, const float* H, float* U, float* Rg, float* CandIn,
                                     int64_t rows, int32_t cin, int32_t h, int32_t pad, void* stream) {
    STC_REQUIRE(cin >= 0 && h >= 1 && pad >= 0, STC_EINVAL, "stc_gru_gates_fwd_f32: bad widths cin=%d h=%d pad=%d", cin, h, pad);
    STC_EW_PROLOGUE("stc_gru_gates_fwd_f32", rows, G && H && U && Rg && CandIn && (cin == 0 || Xt));
    const long long n = (long long)rows * (cin + h + pad);
    hipLaunchKernelGGL(gru_gates_fwd_kernel, ew_grid(n), dim3(EW_THREADS), 0, s, G, Xt, H, U, Rg, CandIn, (long long)rows, cin, h, pad);
    STC_LAUNCH_CHECK("stc_gru_gates_fwd_f32 launch");
    return STC_OK;
}

extern "C" int stc_gru_gates_bwd_f32(const float* dCandIn, const float* dU, const float* H, const float* U, const float* Rg,
                                     float* dG, float* dXt, float* dH, int64_t rows, int32_t cin, int32_t h, int32_t pad, void* stream) {
    STC_REQUIRE(cin >= 0 && h >= 1 && pad >= 0, STC_EINVAL, "stc_gru_gates_bwd_f32: bad widths cin=%d h=%d pad=%d", cin, h, pad);
    STC_EW_PROLOGUE("stc_gru_gates_bwd_f32", rows, dCandIn && dU && H && U && Rg && dG && dH && (cin == 0 || dXt));
    const long long n = (long long)rows * (cin + h + pad);
    hipLaunchKernelGGL(gru_gates_bwd_kernel, ew_grid(n), dim3(EW_THREADS), 0, s, dCandIn, dU, H, U, Rg, dG, dXt, dH, (long long)rows, cin, h, pad);
    STC_LAUNCH_CHECK("stc_gru_gates_bwd_f32 launch");
    return STC_OK;
}

extern "C" int stc_gru_blend_fwd_f32(const float* Cpre, const float* U, const float* H, float* Cand, float* Hnew,
                                     int64_t n, void* stream) {
    STC_EW_PROLOGUE("stc_gru_blend_fwd_f32", n, Cpre && U && H && Cand && Hnew);
    hipLaunchKernelGGL(gru_blend_fwd_kernel, ew_grid(n), dim3(EW_THREADS), 0, s, Cpre, U, H, Cand, Hnew, (long long)n);
    STC_LAUNCH_CHECK("stc_gru_blend_fwd_f32 launch");
    return STC_OK;
}

extern "C" int stc_gru_blend_bwd_f32(const float* dHnew, const float* U, const float* H, const float* Cand,
                                     float* dCpre, float* dU, float* dH, int64_t n, void* stream) {
    STC_EW_PROLOGUE("stc_gru_blend_bwd_f32", n, dHnew && U && H && Cand && dCpre && dU && dH);
    hipLaunchKernelGGL(gru_blend_bwd_kernel, ew_grid(n), dim3(EW_THREADS), 0, s, dHnew, U, H, Cand, dCpre, dU, dH, (long long)n);
    STC_LAUNCH_CHECK("stc_gru_blend_bwd_f32 launch");
    return STC_OK;
}

extern "C" int stc_axpy_f32(float a, const float* x, float* y, int64_t n, void* stream) {
    STC_EW_PROLOGUE("stc_axpy_f32", n, x && y);
    hipLaunchKernelGGL(axpy_kernel, ew_grid(n), dim3(EW_THREADS), 0, s, a, x, y, (long long)n);
    STC_LAUNCH_CHECK("stc_axpy_f32 launch");
    return STC_OK;
}

extern "C" int stc_concat2_f32(const float* A, const float* B, float* out, int64_t rows, int32_t a, int32_t b, int32_t pad, void* stream) {
    STC_REQUIRE(a >= 0 && b >= 0 && pad >= 0, STC_EINVAL, "stc_concat2_f32: negative width");
    const long long n = (long long)rows * (a + b + pad);
    STC_EW_PROLOGUE("stc_concat2_f32", n, out && (a == 0 || A) && (b == 0 || B));
    hipLaunchKernelGGL(concat2_kernel, ew_grid(n), dim3(EW_THREADS), 0, s, A, B, out, (long long)rows, a, b, pad);
    STC_LAUNCH_CHECK("stc_concat2_f32 launch");
    return STC_OK;
}

extern "C" int stc_split2_f32(const float* src, float* A, float* B, int64_t rows, int32_t a, int32_t b, int32_t pad, void* stream) {
    STC_REQUIRE(a >= 0 && b >= 0 && pad >= 0, STC_EINVAL, "stc_split2_f32: negative width");
    const long long n = (long long)rows * (a + b + pad);
    STC_EW_PROLOGUE("stc_split2_f32", n, src && (a == 0 || A) && (b == 0 || B));
    hipLaunchKernelGGL(split2_kernel, ew_grid(n), dim3(EW_THREADS), 0, s, src, A, B, (long long)rows, a, b, pad);
    STC_LAUNCH_CHECK("stc_split2_f32 launch");
    return STC_OK;
}

extern "C" int stc_cheby_dense_fwd_f32(const float* G, int32_t n, int32_t K, float* T, void* stream) {
    STC_REQUIRE(n >= 1 && n <= 128 && K >= 1 && K <= STC_MAX_K, STC_ELIMIT, "stc_cheby_dense_fwd_f32: n=%d (<=128) K=%d (<=%d)", n, K, STC_MAX_K);
    STC_REQUIRE(G && T, STC_EINVAL, "stc_cheby_dense_fwd_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nn = n * n;
    const dim3 grid((nn + EW_THREADS - 1) / EW_THREADS), block(EW_THREADS);
    hipLaunchKernelGGL(cheby_init_kernel, grid, block, 0, s, G, n, K, T);
    for (int k = 2; k < K; ++k)
        hipLaunchKernelGGL(cheby_step_kernel, grid, block, 0, s, G, T + (size_t)(k - 1) * nn, T + (size_t)(k - 2) * nn, T + (size_t)k * nn, n);
    STC_LAUNCH_CHECK("stc_cheby_dense_fwd_f32 launch");
    return STC_OK;
}

extern "C" int stc_cheby_dense_bwd_f32(const float* G, const float* T, float* dT, int32_t n, int32_t K, float* dG, void* stream) {
    STC_REQUIRE(n >= 1 && n <= 128 && K >= 1 && K <= STC_MAX_K, STC_ELIMIT, "stc_cheby_dense_bwd_f32: n=%d (<=128) K=%d (<=%d)", n, K, STC_MAX_K);
    STC_REQUIRE(G && T && dT && dG, STC_EINVAL, "stc_cheby_dense_bwd_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nn = n * n;
    const dim3 grid((nn + EW_THREADS - 1) / EW_THREADS), block(EW_THREADS);
    if (K < 2) {
        if (int rc = stc::hip_status(hipMemsetAsync(dG, 0, (size_t)nn * sizeof(float), s), "memset dG")) return rc;
        return STC_OK;
    }
    if (K > 2) if (int rc = stc::hip_status(hipMemsetAsync(dG, 0, (size_t)nn * sizeof(float), s), "memset dG")) return rc;
    for (int k = K - 1; k >= 2; --k)
        hipLaunchKernelGGL(cheby_bwd_step_kernel, grid, block, 0, s, G, T + (size_t)(k - 1) * nn, dT + (size_t)k * nn,
                           dT + (size_t)(k - 1) * nn, dT + (size_t)(k - 2) * nn, dG, n);
    hipLaunchKernelGGL(cheby_bwd_final_kernel, grid, block, 0, s, dT + nn, dG, nn, K > 2 ? 1 : 0);
    STC_LAUNCH_CHECK("stc_cheby_dense_bwd_f32 launch");
    return STC_OK;
}

// ---- library-wide ---------------------------------------------------------------
namespace stc {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace stc

extern "C" int stc_version(void) { return STC_ABI_VERSION; }
extern "C" const char* stc_last_error(void) { return stc::error_buffer(); }
