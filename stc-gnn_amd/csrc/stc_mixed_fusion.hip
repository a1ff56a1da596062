// MixedFusion of the reference's learned graph generator (STC_GNN.py:246-261, used by MGP_Gen :210-243):
//
//     a = sigmoid(W_A vec(A) + b_A + W_P vec(P) + b_P)          W_A, W_P: (D, D) with D = n^2 (nn.Linear weights: row i = output i)
//     G = a * A + (1 - a) * P
//
// At the SF-incidents shape D = 10^4: the two weight matrices are 2 x 400 MB -- 99.99 % of the model's parameters -- and every training step
// streams them three times (forward, the transposed product of the backward, the outer-product gradients; Adam on top).  As library calls the
// two GEMVs, the two outer products and the transposed GEMV took ~1.0 ms of the 7.8 ms step (kernel stats of `bench.py --preset sf-learned`);
// the traffic bound at 5.5 TB/s is 0.15 ms forward + 0.22 ms backward.  HBM-bound streaming kernels:
//
//   forward   one wave per output row: both matrix rows streamed with 16-byte loads (non-temporal: read once), the two input vectors from
//             cache, fmaf chains, wave reduction, gate and mix in the epilogue (the gate is kept for the backward)
//   backward  dpre = dG (A - P) a (1 - a) (= db_A = db_P, a first tiny launch);  dW_A = dpre (x) A, dW_P = dpre (x) P  (written once, non-temporal);  dP = dG (1 - a) + W_P^T dpre
//             (and dA = dG a + W_A^T dpre when wanted): a wave owns a strip of 256 columns over a chunk of rows, keeps the column sums in
//             registers and leaves them as a partial per row chunk; a second launch adds the partials in a fixed order (no atomics:
//             bitwise reproducible).
#include "stc_common.h"

namespace {

using v4f = __attribute__((ext_vector_type(4))) float;
constexpr int MF_THREADS = 256, MF_WAVES = MF_THREADS / 64;
constexpr int MF_ROW_CHUNKS = 64;      // backward: row chunks (partials per column)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// D % 4 == 0, all pointers 16-byte aligned
__global__ __launch_bounds__(MF_THREADS) void mixed_fusion_fwd_kernel(const float* __restrict__ WA, const float* __restrict__ bA, const float* __restrict__ WP,
                                                                     const float* __restrict__ bP, const float* __restrict__ A, const float* __restrict__ P,
                                                                     float* __restrict__ gate, float* __restrict__ G, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, D4 = D >> 2;
    const v4f* A4 = reinterpret_cast<const v4f*>(A);
    const v4f* P4 = reinterpret_cast<const v4f*>(P);
    for (int i = blockIdx.x * MF_WAVES + wave; i < D; i += gridDim.x * MF_WAVES) {
        const v4f* wa = reinterpret_cast<const v4f*>(WA + (size_t)i * D);
        const v4f* wp = reinterpret_cast<const v4f*>(WP + (size_t)i * D);
        float sa = 0.f, sp = 0.f;
        int j = lane;
        for (; j + 64 < D4; j += 128) {                    // two pieces of each row in flight per lane
            const v4f a0 = __builtin_nontemporal_load(wa + j), a1 = __builtin_nontemporal_load(wa + j + 64);
            const v4f p0 = __builtin_nontemporal_load(wp + j), p1 = __builtin_nontemporal_load(wp + j + 64);
            const v4f x0 = A4[j], x1 = A4[j + 64], y0 = P4[j], y1 = P4[j + 64];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sa = fmaf(a0[c], x0[c], sa); sa = fmaf(a1[c], x1[c], sa);
                sp = fmaf(p0[c], y0[c], sp); sp = fmaf(p1[c], y1[c], sp);
            }
        }
        for (; j < D4; j += 64) {
            const v4f a0 = __builtin_nontemporal_load(wa + j), p0 = __builtin_nontemporal_load(wp + j), x0 = A4[j], y0 = P4[j];
#pragma unroll
            for (int c = 0; c < 4; ++c) { sa = fmaf(a0[c], x0[c], sa); sp = fmaf(p0[c], y0[c], sp); }
        }
        const float pre = (wave_sum(sa) + bA[i]) + (wave_sum(sp) + bP[i]);       // lin_A(A) + lin_P(P), as the reference adds them (:256-258)
        if (lane == 0) {
            const float a = stc_sigmoid(pre);
            gate[i] = a;
            G[i] = a * A[i] + (1.f - a) * P[i];
        }
    }
}

// dpre[i] = dG[i] (A[i] - P[i]) a[i] (1 - a[i]): the gradient of the gate's pre-activation = db_A = db_P
__global__ __launch_bounds__(MF_THREADS) void mixed_fusion_dpre_kernel(const float* __restrict__ A, const float* __restrict__ P, const float* __restrict__ gate,
                                                                      const float* __restrict__ dG, float* __restrict__ db, int D) {
    const int i = blockIdx.x * MF_THREADS + threadIdx.x;
    if (i < D) {
        const float a = gate[i];
        db[i] = dG[i] * (A[i] - P[i]) * a * (1.f - a);
    }
}

// wave (strip s of 256 columns, row chunk r): rows [r0, r1) of W_P (and W_A) summed into the strip's columns with weights dpre[i], the rows of
// dW_A / dW_P written; four rows per iteration (their loads issued together: a wave has 4 - 8 KiB in flight)
template <bool WANT_DA>
__global__ __launch_bounds__(MF_THREADS) void mixed_fusion_bwd_kernel(const float* __restrict__ WA, const float* __restrict__ WP, const float* __restrict__ A,
                                                                     const float* __restrict__ P, const float* __restrict__ dpre,
                                                                     float* __restrict__ dWA, float* __restrict__ dWP, float* __restrict__ partP,
                                                                     float* __restrict__ partA, int D, int rows_per) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), D4 = D >> 2;
    const int strips = (D4 + 63) / 64;
    const int unit = blockIdx.x * MF_WAVES + wave;          // unit = chunk * strips + strip
    if (unit >= strips * MF_ROW_CHUNKS) return;
    const int strip = unit % strips, chunk = unit / strips;
    const int j = strip * 64 + lane;                        // this lane's 16-byte piece of every row
    const bool live = j < D4;
    const int jj = live ? j : 0;                            // (lanes past the row's end load piece 0 and store nothing)
    const int r0 = chunk * rows_per, r1 = min(D, r0 + rows_per);      // multiples of 4
    const v4f a4 = reinterpret_cast<const v4f*>(A)[jj], p4 = reinterpret_cast<const v4f*>(P)[jj];
    const v4f* wp = reinterpret_cast<const v4f*>(WP) + jj;
    const v4f* wa = reinterpret_cast<const v4f*>(WA) + jj;
    v4f sp = {0.f, 0.f, 0.f, 0.f}, sa = {0.f, 0.f, 0.f, 0.f};
    for (int i = r0; i < r1; i += 4) {
        const v4f d4 = *reinterpret_cast<const v4f*>(dpre + i);                 // (wave-uniform address)
        v4f xp[4], xa[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xp[r] = __builtin_nontemporal_load(wp + (size_t)(i + r) * D4);
            if (WANT_DA) xa[r] = __builtin_nontemporal_load(wa + (size_t)(i + r) * D4);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = d4[r];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sp[c] = fmaf(xp[r][c], d, sp[c]);
                if (WANT_DA) sa[c] = fmaf(xa[r][c], d, sa[c]);
            }
            if (live && dWA) {                                   // (frozen weights: no gradient wanted -- dW_A, dW_P both null)
                __builtin_nontemporal_store(d * a4, reinterpret_cast<v4f*>(dWA) + (size_t)(i + r) * D4 + j);
                __builtin_nontemporal_store(d * p4, reinterpret_cast<v4f*>(dWP) + (size_t)(i + r) * D4 + j);
            }
        }
    }
    if (live) {
        reinterpret_cast<v4f*>(partP)[(size_t)chunk * D4 + j] = sp;
        if (WANT_DA) reinterpret_cast<v4f*>(partA)[(size_t)chunk * D4 + j] = sa;
    }
}

// dP[j] = dG[j] (1 - a[j]) + sum_chunks partP[chunk][j]  (chunks in order);  dA likewise with a[j]
__global__ __launch_bounds__(MF_THREADS) void mixed_fusion_finish_kernel(const float* __restrict__ gate, const float* __restrict__ dG,
                                                                        const float* __restrict__ partP, const float* __restrict__ partA,
                                                                        float* __restrict__ dP, float* __restrict__ dA, int D) {
    const int j = blockIdx.x * MF_THREADS + threadIdx.x;
    if (j >= D) return;
    const float a = gate[j], g = dG[j];
    float sp = 0.f, sa = 0.f;
    for (int c = 0; c < MF_ROW_CHUNKS; ++c) {
        sp += partP[(size_t)c * D + j];
        if (partA) sa += partA[(size_t)c * D + j];
    }
    dP[j] = g * (1.f - a) + sp;
    if (dA) dA[j] = g * a + sa;
}

}  // namespace

extern "C" size_t stc_mixed_fusion_workspace_bytes(int32_t D, int32_t want_dA) {
    return D < 1 ? 0 : (size_t)MF_ROW_CHUNKS * D * sizeof(float) * (want_dA ? 2 : 1);
}

extern "C" int stc_mixed_fusion_fwd_f32(const float* WA, const float* bA, const float* WP, const float* bP, const float* A, const float* P,
                                        float* gate, float* G, int32_t D, void* stream) {
    STC_REQUIRE(D >= 0, STC_EINVAL, "stc_mixed_fusion_fwd_f32: negative size");
    if (D == 0) return STC_OK;
    STC_REQUIRE(D % 4 == 0, STC_EUNSUPPORTED, "stc_mixed_fusion_fwd_f32: D = n^2 = %d must be a multiple of 4 (rows of whole 16-byte pieces)", D);
    STC_REQUIRE(WA && bA && WP && bP && A && P && gate && G, STC_EINVAL, "stc_mixed_fusion_fwd_f32: null pointer");
    STC_REQUIRE(stc::aligned16(WA) && stc::aligned16(WP) && stc::aligned16(A) && stc::aligned16(P), STC_EALIGN,
                "stc_mixed_fusion_fwd_f32: W_A / W_P / A / P must be 16-byte aligned");
    const int blocks = (D + MF_WAVES - 1) / MF_WAVES;
    hipLaunchKernelGGL(mixed_fusion_fwd_kernel, dim3(blocks < 8192 ? blocks : 8192), dim3(MF_THREADS), 0, static_cast<hipStream_t>(stream),
                       WA, bA, WP, bP, A, P, gate, G, D);
    STC_LAUNCH_CHECK("stc_mixed_fusion_fwd_f32 launch");
    return STC_OK;
}

extern "C" int stc_mixed_fusion_bwd_f32(const float* WA, const float* WP, const float* A, const float* P, const float* gate, const float* dG,
                                        float* dWA, float* dWP, float* db, float* dP, float* dA,
                                        void* workspace, size_t workspace_bytes, int32_t D, void* stream) {
    STC_REQUIRE(D >= 0, STC_EINVAL, "stc_mixed_fusion_bwd_f32: negative size");
    if (D == 0) return STC_OK;
    STC_REQUIRE(D % 4 == 0, STC_EUNSUPPORTED, "stc_mixed_fusion_bwd_f32: D = n^2 = %d must be a multiple of 4", D);
    STC_REQUIRE(WP && A && P && gate && dG && db && dP && (!dA || WA) && ((dWA == nullptr) == (dWP == nullptr)), STC_EINVAL,
                "stc_mixed_fusion_bwd_f32: null pointer (dW_A and dW_P may be NULL together: frozen weights)");
    STC_REQUIRE(stc::aligned16(WP) && (!dA || stc::aligned16(WA)) && stc::aligned16(A) && stc::aligned16(P) && stc::aligned16(dWA) && stc::aligned16(dWP) && stc::aligned16(db), STC_EALIGN,
                "stc_mixed_fusion_bwd_f32: matrices, A / P and db must be 16-byte aligned");
    STC_REQUIRE(workspace && stc::aligned16(workspace) && workspace_bytes >= stc_mixed_fusion_workspace_bytes(D, dA != nullptr), STC_EINVAL,
                "stc_mixed_fusion_bwd_f32: workspace null, misaligned or too small (%zu B)", workspace_bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* partP = static_cast<float*>(workspace);
    float* partA = dA ? partP + (size_t)MF_ROW_CHUNKS * D : nullptr;
    const int strips = (D / 4 + 63) / 64, units = strips * MF_ROW_CHUNKS;
    const int rows_per = ((D + 4 * MF_ROW_CHUNKS - 1) / (4 * MF_ROW_CHUNKS)) * 4;          // a multiple of 4 (D is one: every chunk holds whole groups of four rows)
    const dim3 grid((units + MF_WAVES - 1) / MF_WAVES), block(MF_THREADS), flat((D + MF_THREADS - 1) / MF_THREADS);
    hipLaunchKernelGGL(mixed_fusion_dpre_kernel, flat, block, 0, s, A, P, gate, dG, db, D);
    if (dA) hipLaunchKernelGGL(mixed_fusion_bwd_kernel<true>, grid, block, 0, s, WA, WP, A, P, db, dWA, dWP, partP, partA, D, rows_per);
    else hipLaunchKernelGGL(mixed_fusion_bwd_kernel<false>, grid, block, 0, s, WA, WP, A, P, db, dWA, dWP, partP, partA, D, rows_per);
    hipLaunchKernelGGL(mixed_fusion_finish_kernel, flat, block, 0, s, gate, dG, partP, partA, dP, dA, D);
    STC_LAUNCH_CHECK("stc_mixed_fusion_bwd_f32 launch");
    return STC_OK;
}

// ---- Adam over a large parameter (the harness step of Model_Trainer.py:71-87: torch.optim.Adam(lr, weight_decay), no amsgrad) ----------------
// The two MixedFusion matrices are 99.99 % of the learned-graph model; torch's fused multi-tensor Adam moves their 5.6 GB per step at 4.4 TB/s
// (1.26 of the SF step's 6.0 ms).  One streaming launch per tensor: p, g, m, v read once, p, m, v written once, 16-byte non-temporal accesses.
// The step count lives on the device (`step`: one float, already incremented by the caller) so that a captured HIP graph replays with the
// right bias corrections.
namespace {

constexpr int AD_THREADS = 256, AD_PIECES = 4;      // 16-byte pieces per lane and trip: 4 x 4 planes in flight

__global__ __launch_bounds__(AD_THREADS) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                          long long n4, const float* __restrict__ step, float lr, float beta1, float beta2,
                                                          float omb1, float omb2, float eps, float weight_decay) {      // omb = 1 - beta, formed in double
    const float t = *step;
    const float bc1 = 1.f - powf(beta1, t), bc2_sqrt = sqrtf(1.f - powf(beta2, t));
    const float step_size = lr / bc1, inv_bc2_sqrt = 1.f / bc2_sqrt;
    v4f* p4 = reinterpret_cast<v4f*>(p);
    const v4f* g4 = reinterpret_cast<const v4f*>(g);
    v4f* m4 = reinterpret_cast<v4f*>(m);
    v4f* v4 = reinterpret_cast<v4f*>(v);
    // a workgroup takes CONTIGUOUS runs of AD_PIECES x 4 KiB per plane and trip (pieces a whole grid apart per lane ran at 4.3 TB/s: sixteen
    // distant streams per lane)
    const long long run = (long long)AD_THREADS * AD_PIECES;
    for (long long i = (long long)blockIdx.x * run + threadIdx.x; i < n4; i += (long long)gridDim.x * run) {
        constexpr long long stride = AD_THREADS;
        v4f pp[AD_PIECES], gg[AD_PIECES], mm[AD_PIECES], vv[AD_PIECES];
#pragma unroll
        for (int k = 0; k < AD_PIECES; ++k) {
            const long long at = i + k * stride;
            if (at < n4) {
                pp[k] = __builtin_nontemporal_load(p4 + at); gg[k] = __builtin_nontemporal_load(g4 + at);
                mm[k] = __builtin_nontemporal_load(m4 + at); vv[k] = __builtin_nontemporal_load(v4 + at);
            }
        }
#pragma unroll
        for (int k = 0; k < AD_PIECES; ++k) {
            const long long at = i + k * stride;
            if (at < n4) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float grad = fmaf(weight_decay, pp[k][c], gg[k][c]);                  // L2 weight decay folded into the gradient
                    mm[k][c] = fmaf(omb1, grad - mm[k][c], mm[k][c]);                           // lerp(m, grad, 1 - beta1)
                    vv[k][c] = fmaf(beta2, vv[k][c], omb2 * grad * grad);
                    const float denom = fmaf(sqrtf(vv[k][c]), inv_bc2_sqrt, eps);
                    pp[k][c] -= step_size * (mm[k][c] / denom);
                }
                __builtin_nontemporal_store(pp[k], p4 + at);
                __builtin_nontemporal_store(mm[k], m4 + at);
                __builtin_nontemporal_store(vv[k], v4 + at);
            }
        }
    }
}

}  // namespace

extern "C" int stc_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, const float* step, double lr, double beta1, double beta2, double eps,
                            double weight_decay, void* stream) {
    STC_REQUIRE(n >= 0 && n % 4 == 0, STC_EINVAL, "stc_adam_f32: %lld elements (a multiple of 4)", (long long)n);
    if (n == 0) return STC_OK;
    STC_REQUIRE(p && g && m && v && step, STC_EINVAL, "stc_adam_f32: null pointer");
    STC_REQUIRE(stc::aligned16(p) && stc::aligned16(g) && stc::aligned16(m) && stc::aligned16(v), STC_EALIGN, "stc_adam_f32: tensors must be 16-byte aligned");
    STC_REQUIRE(p != g && p != m && p != v && m != v && g != m && g != v, STC_EINVAL, "stc_adam_f32: tensors must be distinct");
    STC_REQUIRE(lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0. && weight_decay >= 0., STC_EINVAL,
                "stc_adam_f32: lr %g, betas (%g, %g), eps %g, weight decay %g", lr, beta1, beta2, eps, weight_decay);
    const long long n4 = n / 4;
    const long long want = (n4 + (long long)AD_THREADS * AD_PIECES - 1) / ((long long)AD_THREADS * AD_PIECES);
    const int grid = (int)(want < 1 ? 1 : (want > 256 * 8 ? 256 * 8 : want));
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(AD_THREADS), 0, static_cast<hipStream_t>(stream), p, g, m, v, n4, step, (float)lr, (float)beta1, (float)beta2,
                       (float)(1. - beta1), (float)(1. - beta2), (float)eps, (float)weight_decay);
    STC_LAUNCH_CHECK("stc_adam_f32 launch");
    return STC_OK;
}
