// Front end of the reference's learned graph generator (MGP_Gen.forward, STC_GNN.py:229-232 for the spatial branch and :237-240 for the
// category branch), and its autograd:
//
//     U = tanh(alpha X Wu),  V = tanh(alpha X Wv)                      X: (B, T, R, F) rows x features, Wu / Wv: (F, h)
//     P = sum_{b,t} U V^T   (R x R)                                    the einsum pair of :231 is  P - P^T
//     Ps = softmax(relu(P - P^T), -1)
//
// As torch operations this is ~60 launches of a few microseconds per training step at the SF-incidents shape (two matmuls, scalings, tanh,
// permuted copies for the einsum, transpose, subtraction, relu, softmax, and their backward nodes, for two branches) -- a tenth of the
// reference's full-model step once the cells run in a few launches each.  Here a branch is three launches forward and six backward:
//
//   stc_mgp_uv_fwd_f32       U, V in (R, K, h) layout (K = B T): the contraction over (k, j) of the pair product is then one plain matrix
//                            product  P = U' V'^T  on (R, K h) operands -- left to the library (a plain GEMM), as is dU' = dP V', dV' = dP^T U'
//   stc_mgp_softmax_fwd_f32  Ps from P: antisymmetric part, relu, row softmax -- one wave per row
//   stc_mgp_softmax_bwd_f32  dP from dPs: softmax backward, relu mask, antisymmetric part (two tiny launches: row dots, then the entries)
//   stc_mgp_uv_bwd_f32       dWu, dWv = alpha X^T [(1 - U^2) dU] as per-k partial sums (fixed-order sum by the caller: reproducible)
//
// The two branches differ only in which axis of the window is "rows": X is addressed as x[k][r][f] = X[k k_stride + r r_stride + f f_stride]
// (spatial branch: rows = nodes, features = categories; category branch: the transpose, :236).  Nothing here is bound by anything but launch
// count: the operands are a few MB.
#include "stc_common.h"

namespace {

constexpr int MG_THREADS = 256;

__global__ __launch_bounds__(MG_THREADS) void mgp_uv_fwd_kernel(const float* __restrict__ X, long long ks, long long rs, long long fs,
                                                                const float* __restrict__ Wu, const float* __restrict__ Wv, float alpha,
                                                                float* __restrict__ U, float* __restrict__ V, int K, int R, int F, int h) {
    const long long total = (long long)R * K * h;
    for (long long o = (long long)blockIdx.x * MG_THREADS + threadIdx.x; o < total; o += (long long)gridDim.x * MG_THREADS) {
        const int j = (int)(o % h);
        const long long rk = o / h;
        const int k = (int)(rk % K), r = (int)(rk / K);
        const float* x = X + k * ks + r * rs;
        float su = 0.f, sv = 0.f;
        for (int f = 0; f < F; ++f) {
            const float xv = x[f * fs];
            su = fmaf(xv, Wu[f * h + j], su);
            sv = fmaf(xv, Wv[f * h + j], sv);
        }
        U[o] = stc_tanh(alpha * su);
        V[o] = stc_tanh(alpha * sv);
    }
}

// one workgroup per k: partial[k][w][f][j] = alpha sum_r x[k][r][f] (1 - T^2) dT [r][k][j]   (w = 0: T = U, w = 1: T = V)
__global__ __launch_bounds__(MG_THREADS) void mgp_uv_bwd_kernel(const float* __restrict__ X, long long ks, long long rs, long long fs,
                                                                const float* __restrict__ U, const float* __restrict__ V,
                                                                const float* __restrict__ dU, const float* __restrict__ dV, float alpha,
                                                                float* __restrict__ partial, int K, int R, int F, int h) {
    const int k = blockIdx.x;
    const int outs = 2 * F * h;
    for (int o = threadIdx.x; o < outs; o += MG_THREADS) {
        const int j = o % h, f = (o / h) % F, w = o / (h * F);
        const float* T = w ? V : U;
        const float* dT = w ? dV : dU;
        const float* x = X + k * ks + f * fs;
        float s = 0.f;
        for (int r = 0; r < R; ++r) {
            const long long e = ((long long)r * K + k) * h + j;
            const float t = T[e];
            s = fmaf(x[r * rs], (1.f - t * t) * dT[e], s);
        }
        partial[(long long)k * outs + o] = alpha * s;
    }
}

// Ps[n][:] = softmax(relu(P[n][:] - P[:][n])): one wave per row n
__global__ __launch_bounds__(64) void mgp_softmax_fwd_kernel(const float* __restrict__ P, float* __restrict__ Ps, int R) {
    const int n = blockIdx.x, lane = threadIdx.x;
    float mx = 0.f;                                                    // relu: every entry >= 0
    for (int m = lane; m < R; m += 64) mx = fmaxf(mx, P[(long long)n * R + m] - P[(long long)m * R + n]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float sum = 0.f;
    for (int m = lane; m < R; m += 64) sum += expf(fmaxf(P[(long long)n * R + m] - P[(long long)m * R + n], 0.f) - mx);
    sum = stc_wave_sum(sum);
    const float inv = 1.f / sum;
    for (int m = lane; m < R; m += 64) Ps[(long long)n * R + m] = expf(fmaxf(P[(long long)n * R + m] - P[(long long)m * R + n], 0.f) - mx) * inv;
}

// rowdot[n] = sum_m dPs[n][m] Ps[n][m]
__global__ __launch_bounds__(64) void mgp_softmax_dot_kernel(const float* __restrict__ Ps, const float* __restrict__ dPs, float* __restrict__ rowdot, int R) {
    const int n = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int m = lane; m < R; m += 64) s = fmaf(dPs[(long long)n * R + m], Ps[(long long)n * R + m], s);
    s = stc_wave_sum(s);
    if (lane == 0) rowdot[n] = s;
}

// dD[n][m] = Ps[n][m] (dPs[n][m] - rowdot[n]) where P[n][m] - P[m][n] > 0, else 0;   dP[n][m] = dD[n][m] - dD[m][n]
__global__ __launch_bounds__(MG_THREADS) void mgp_softmax_bwd_kernel(const float* __restrict__ P, const float* __restrict__ Ps, const float* __restrict__ dPs,
                                                                     const float* __restrict__ rowdot, float* __restrict__ dP, int R) {
    const long long total = (long long)R * R;
    for (long long o = (long long)blockIdx.x * MG_THREADS + threadIdx.x; o < total; o += (long long)gridDim.x * MG_THREADS) {
        const int m = (int)(o % R), n = (int)(o / R);
        const long long t = (long long)m * R + n;
        const float d = P[o] - P[t];
        const float a = d > 0.f ? Ps[o] * (dPs[o] - rowdot[n]) : 0.f;
        const float b = -d > 0.f ? Ps[t] * (dPs[t] - rowdot[m]) : 0.f;
        dP[o] = a - b;
    }
}

int check_uv(const char* who, const void* X, const void* Wu, const void* Wv, int K, int R, int F, int h) {
    STC_REQUIRE(K >= 0 && R >= 0 && F >= 1 && h >= 1, STC_EINVAL, "%s: negative or empty size (K=%d R=%d F=%d h=%d)", who, K, R, F, h);
    STC_REQUIRE((long long)R * K * h < (1ll << 31) && (long long)2 * F * h < (1ll << 24), STC_ELIMIT, "%s: operand too large (R K h = %lld)", who, (long long)R * K * h);
    if (K == 0 || R == 0) return STC_OK;
    STC_REQUIRE(X && Wu && Wv, STC_EINVAL, "%s: null pointer", who);
    return STC_OK;
}

}  // namespace

extern "C" int stc_mgp_uv_fwd_f32(const float* X, int64_t k_stride, int64_t r_stride, int64_t f_stride, const float* Wu, const float* Wv, float alpha,
                                  float* U, float* V, int32_t K, int32_t R, int32_t F, int32_t h, void* stream) {
    if (int rc = check_uv("stc_mgp_uv_fwd_f32", X, Wu, Wv, K, R, F, h)) return rc;
    if (K == 0 || R == 0) return STC_OK;
    STC_REQUIRE(U && V && U != V, STC_EINVAL, "stc_mgp_uv_fwd_f32: null or aliasing result");
    const long long total = (long long)R * K * h;
    const int grid = (int)((total + MG_THREADS - 1) / MG_THREADS < 4096 ? (total + MG_THREADS - 1) / MG_THREADS : 4096);
    hipLaunchKernelGGL(mgp_uv_fwd_kernel, dim3(grid), dim3(MG_THREADS), 0, static_cast<hipStream_t>(stream), X, (long long)k_stride, (long long)r_stride,
                       (long long)f_stride, Wu, Wv, alpha, U, V, K, R, F, h);
    STC_LAUNCH_CHECK("stc_mgp_uv_fwd_f32 launch");
    return STC_OK;
}

extern "C" int stc_mgp_uv_bwd_f32(const float* X, int64_t k_stride, int64_t r_stride, int64_t f_stride, const float* U, const float* V,
                                  const float* dU, const float* dV, float alpha, float* partials, int32_t K, int32_t R, int32_t F, int32_t h, void* stream) {
    if (int rc = check_uv("stc_mgp_uv_bwd_f32", X, U, V, K, R, F, h)) return rc;
    if (K == 0) return STC_OK;
    STC_REQUIRE(partials && (R == 0 || (dU && dV)), STC_EINVAL, "stc_mgp_uv_bwd_f32: null pointer");
    hipLaunchKernelGGL(mgp_uv_bwd_kernel, dim3(K), dim3(MG_THREADS), 0, static_cast<hipStream_t>(stream), X, (long long)k_stride, (long long)r_stride,
                       (long long)f_stride, U, V, dU, dV, alpha, partials, K, R, F, h);
    STC_LAUNCH_CHECK("stc_mgp_uv_bwd_f32 launch");
    return STC_OK;
}

extern "C" int stc_mgp_softmax_fwd_f32(const float* P, float* Ps, int32_t R, void* stream) {
    STC_REQUIRE(R >= 0 && R <= 46340, STC_EINVAL, "stc_mgp_softmax_fwd_f32: R = %d outside 0..46340", R);
    if (R == 0) return STC_OK;
    STC_REQUIRE(P && Ps && P != Ps, STC_EINVAL, "stc_mgp_softmax_fwd_f32: null or aliasing pointer (every row reads a column of P)");
    hipLaunchKernelGGL(mgp_softmax_fwd_kernel, dim3(R), dim3(64), 0, static_cast<hipStream_t>(stream), P, Ps, R);
    STC_LAUNCH_CHECK("stc_mgp_softmax_fwd_f32 launch");
    return STC_OK;
}

extern "C" int stc_mgp_softmax_bwd_f32(const float* P, const float* Ps, const float* dPs, float* rowdot, float* dP, int32_t R, void* stream) {
    STC_REQUIRE(R >= 0 && R <= 46340, STC_EINVAL, "stc_mgp_softmax_bwd_f32: R = %d outside 0..46340", R);
    if (R == 0) return STC_OK;
    STC_REQUIRE(P && Ps && dPs && rowdot && dP && dP != P && dP != Ps && dP != dPs, STC_EINVAL, "stc_mgp_softmax_bwd_f32: null or aliasing pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(mgp_softmax_dot_kernel, dim3(R), dim3(64), 0, s, Ps, dPs, rowdot, R);
    STC_LAUNCH_CHECK("stc_mgp_softmax_bwd_f32 (row dots) launch");
    const long long total = (long long)R * R;
    const int grid = (int)((total + MG_THREADS - 1) / MG_THREADS < 4096 ? (total + MG_THREADS - 1) / MG_THREADS : 4096);
    hipLaunchKernelGGL(mgp_softmax_bwd_kernel, dim3(grid), dim3(MG_THREADS), 0, s, P, Ps, dPs, rowdot, dP, R);
    STC_LAUNCH_CHECK("stc_mgp_softmax_bwd_f32 launch");
    return STC_OK;
}
