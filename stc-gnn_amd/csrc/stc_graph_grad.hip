// Gradients of LEARNED graphs (the reference's MGP_Gen output: dense Gs, and Gc through its Chebyshev stack; STC_GNN.py:185-261 via autograd)
// from what the few-category cell kernels leave behind (stc_cell_small.hip): both are sums over EVERY cell step and sample of a training
// step, of small products whose operands are planes (cells * batch, N, F):
//     stc_graph_grad_f32   out[n][m]   = sum_g sum_f A[g][n][f] B[g][m][f]      dGs^T pieces: A = gradient of an aggregated slab, B = its source slab
//     stc_mix_grad_f32     out[fa][fb] = sum_g sum_n A[g][n][fa] B[g][n][fb]    Q = Z^T . dY, from which dT_c = < W_(ks, c), Q[c, :, d, :] >
// As library GEMMs these cost 2.5 ms of the 8 ms learned SF step (permuted copies of every operand so that the contraction over (g, f) becomes
// one GEMM dimension; a batched product plus a sum is slower still), and dT_c needs more than fp32 over 10^5 cancelling terms (fp32: the SF
// golden's dGc at 1.1e-5).  Here: one workgroup per chunk of g, a wave per few 16 x 16 output tiles, v_mfma_f32_16x16x4_f32 per g (<= 160 terms,
// exact fp32 products, fp32 sums), and the per-g results added into FLOAT64 running sums in registers; every workgroup writes its float64 partial
// (deterministic: no atomics), the caller adds the partials.  `cell0 / cell_step / n_sel`: the cells of one parameter set inside a buffer that holds
// every cell of a width (g = sel * batch + sample -> plane (cell0 + sel * cell_step) * batch + sample).
#include "stc_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
#ifndef STC_GG_THREADS                 // (probe builds override these: tools/probes/graph_grad_ab.sh)
#define STC_GG_THREADS 256              // round 6, same-box A/B on the reference's full model at the SF shape (threads, graph tiles per wave, mix tiles per wave,
#define STC_GG_NT_TPW 1                 // steps in flight): (1024, 4, 4, 4) mix 42 us / graph 62 us per launch -> (256, 1, 2, 8) 27 / 48: these launches are
#define STC_GG_TN_TPW 2                 // chains of L2 round trips, and four waves of 80 registers hide each other's better than sixteen of 128
#define STC_GG_TN_STEPS 8
#endif
constexpr int GG_THREADS = STC_GG_THREADS, GG_WAVES = GG_THREADS / 64;
constexpr int GG_NT_TPW = STC_GG_NT_TPW;          // output tiles per wave and tile group (blockIdx.y), graph product
constexpr int GG_TN_TPW = STC_GG_TN_TPW;          // output tiles per wave and tile group, mix product
constexpr int GG_TN_STEPS = STC_GG_TN_STEPS;      // contraction steps (of 4 rows) whose operands are in flight together

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct Sel {
    int cell0, cell_step, n_sel, batch;
};
__device__ __forceinline__ size_t plane_of(const Sel& s, int g) { return (size_t)(s.cell0 + (g / s.batch) * s.cell_step) * s.batch + g % s.batch; }

__global__ __launch_bounds__(GG_THREADS) void graph_grad_kernel(const float* __restrict__ A, const float* __restrict__ B, double* __restrict__ part, Sel sel,
                                                                int N, int F, long long chunk_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, kq = lane >> 4;
    const int T = (N + 15) >> 4, tiles = T * T, G = sel.n_sel * sel.batch;
    // the wave's output tiles (blockIdx.y: groups of GG_NT_TPW * GG_WAVES tiles); a tile past the end recomputes tile 0 and is not stored
    int it[GG_NT_TPW], jt[GG_NT_TPW];
    bool live[GG_NT_TPW];
#pragma unroll
    for (int t = 0; t < GG_NT_TPW; ++t) {
        const int tile = (blockIdx.y * GG_NT_TPW + t) * GG_WAVES + wave;
        live[t] = tile < tiles;
        it[t] = live[t] ? tile / T : 0;
        jt[t] = live[t] ? tile - it[t] * T : 0;
    }
    double sum[GG_NT_TPW][4];
#pragma unroll
    for (int t = 0; t < GG_NT_TPW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[t][r] = 0.0;
    for (int g = blockIdx.x; g < G; g += gridDim.x) {
        const float* Ag = A + plane_of(sel, g) * N * F;
        const float* Bg = B + plane_of(sel, g) * N * F;
        f32x4 acc[GG_NT_TPW];
        const float *ar[GG_NT_TPW], *br[GG_NT_TPW];
#pragma unroll
        for (int t = 0; t < GG_NT_TPW; ++t) {
            acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            // rows beyond N read the last row: they only reach output rows / columns that are not stored
            ar[t] = Ag + (size_t)min(16 * it[t] + j, N - 1) * F;
            br[t] = Bg + (size_t)min(16 * jt[t] + j, N - 1) * F;
        }
        for (int kb = 0; kb < (F + 15) / 16; ++kb) {                // (the same count of steps on every lane; quads past F contribute zeros)
            const int k0 = 16 * kb + 4 * kq;
            const bool ok = k0 < F;
            f32x4 a[GG_NT_TPW], b[GG_NT_TPW];
#pragma unroll
            for (int t = 0; t < GG_NT_TPW; ++t) {                   // every operand of the step requested before the first product
                a[t] = *reinterpret_cast<const f32x4*>(ar[t] + (ok ? k0 : 0));
                b[t] = *reinterpret_cast<const f32x4*>(br[t] + (ok ? k0 : 0));
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < GG_NT_TPW; ++t) acc[t] = mfma4(ok ? a[t][s] : 0.f, b[t][s], acc[t]);
        }
#pragma unroll
        for (int t = 0; t < GG_NT_TPW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sum[t][r] += (double)acc[t][r];
    }
    double* out = part + (size_t)blockIdx.x * chunk_stride;
#pragma unroll
    for (int t = 0; t < GG_NT_TPW; ++t)
        if (live[t]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * it[t] + 4 * kq + r, m = 16 * jt[t] + j;
                if (n < N && m < N) out[(size_t)n * N + m] = sum[t][r];
            }
        }
}

__global__ __launch_bounds__(GG_THREADS) void mix_grad_kernel(const float* __restrict__ A, const float* __restrict__ B, double* __restrict__ part, Sel sel, int N,
                                                              int Fa, int Fb, long long chunk_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, kq = lane >> 4;
    const int Ta = (Fa + 15) >> 4, Tb = (Fb + 15) >> 4, tiles = Ta * Tb, G = sel.n_sel * sel.batch;
    int it[GG_TN_TPW], jt[GG_TN_TPW];
    bool live[GG_TN_TPW];
#pragma unroll
    for (int t = 0; t < GG_TN_TPW; ++t) {
        const int tile = (blockIdx.y * GG_TN_TPW + t) * GG_WAVES + wave;
        live[t] = tile < tiles;
        it[t] = live[t] ? tile / Tb : 0;
        jt[t] = live[t] ? tile - it[t] * Tb : 0;
    }
    double sum[GG_TN_TPW][4];
#pragma unroll
    for (int t = 0; t < GG_TN_TPW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[t][r] = 0.0;
    for (int g = blockIdx.x; g < G; g += gridDim.x) {
        const float* Ag = A + plane_of(sel, g) * N * Fa;
        const float* Bg = B + plane_of(sel, g) * N * Fb;
        f32x4 acc[GG_TN_TPW];
        const float *ac[GG_TN_TPW], *bc[GG_TN_TPW];
#pragma unroll
        for (int t = 0; t < GG_TN_TPW; ++t) {
            acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            ac[t] = Ag + min(16 * it[t] + j, Fa - 1);                  // columns beyond Fa / Fb: clamped, not stored
            bc[t] = Bg + min(16 * jt[t] + j, Fb - 1);
        }
        for (int n0 = 0; n0 < N; n0 += 4 * GG_TN_STEPS) {          // GG_TN_STEPS steps of 4 rows per pass: every operand of the pass is requested
            float a[GG_TN_STEPS][GG_TN_TPW], b[GG_TN_STEPS][GG_TN_TPW];     // before its first product (one load -> product at a time is an L2 round trip each)
            bool ok[GG_TN_STEPS];
#pragma unroll
            for (int u = 0; u < GG_TN_STEPS; ++u) {
                const int n = n0 + 4 * u + kq;
                ok[u] = n < N;                                          // the rows ARE the contraction: rows past N must contribute zero
                const size_t ra = (size_t)(ok[u] ? n : 0) * Fa, rb = (size_t)(ok[u] ? n : 0) * Fb;
#pragma unroll
                for (int t = 0; t < GG_TN_TPW; ++t) {
                    a[u][t] = ac[t][ra];
                    b[u][t] = bc[t][rb];
                }
            }
#pragma unroll
            for (int u = 0; u < GG_TN_STEPS; ++u)
#pragma unroll
                for (int t = 0; t < GG_TN_TPW; ++t) acc[t] = mfma4(ok[u] ? a[u][t] : 0.f, b[u][t], acc[t]);
        }
#pragma unroll
        for (int t = 0; t < GG_TN_TPW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sum[t][r] += (double)acc[t][r];
    }
    double* out = part + (size_t)blockIdx.x * chunk_stride;
#pragma unroll
    for (int t = 0; t < GG_TN_TPW; ++t)
        if (live[t]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int fa = 16 * it[t] + 4 * kq + r, fb = 16 * jt[t] + j;
                if (fa < Fa && fb < Fb) out[(size_t)fa * Fb + fb] = sum[t][r];
            }
        }
}

int check_sel(const char* what, int32_t cell0, int32_t cell_step, int32_t n_sel, int32_t batch, int32_t n_chunks) {
    STC_REQUIRE(cell0 >= 0 && cell_step >= 1 && n_sel >= 0 && batch >= 1 && n_chunks >= 1 && n_chunks <= 65535, STC_EINVAL,
                "%s: cell0=%d cell_step=%d n_sel=%d batch=%d n_chunks=%d", what, cell0, cell_step, n_sel, batch, n_chunks);
    return STC_OK;
}

}  // namespace

extern "C" int stc_graph_grad_f32(const float* A, const float* B, double* partials, int32_t n_chunks, int32_t cell0, int32_t cell_step, int32_t n_sel,
                                  int32_t batch, int32_t N, int32_t F, int64_t chunk_stride, void* stream) {
    if (int rc = check_sel("stc_graph_grad_f32", cell0, cell_step, n_sel, batch, n_chunks)) return rc;
    STC_REQUIRE(N >= 1 && N <= 4096 && F >= 4 && F % 4 == 0, STC_EINVAL, "stc_graph_grad_f32: N=%d (1..4096), F=%d (a multiple of 4)", N, F);
    STC_REQUIRE(A && B && partials && stc::aligned16(A) && stc::aligned16(B), STC_EINVAL, "stc_graph_grad_f32: null or unaligned operand");
    STC_REQUIRE(chunk_stride == 0 || chunk_stride >= (int64_t)N * N, STC_EINVAL, "stc_graph_grad_f32: chunk stride %lld below the block of %d x %d", (long long)chunk_stride, N, N);
    const int T = (N + 15) / 16, groups = (T * T + GG_NT_TPW * GG_WAVES - 1) / (GG_NT_TPW * GG_WAVES);
    hipLaunchKernelGGL(graph_grad_kernel, dim3((unsigned)n_chunks, (unsigned)groups), dim3(GG_THREADS), 0, static_cast<hipStream_t>(stream), A, B, partials,
                       Sel{cell0, cell_step, n_sel, batch}, N, F, (long long)(chunk_stride ? chunk_stride : (int64_t)N * N));
    STC_LAUNCH_CHECK("stc_graph_grad_f32 launch");
    return STC_OK;
}

extern "C" int stc_mix_grad_f32(const float* A, const float* B, double* partials, int32_t n_chunks, int32_t cell0, int32_t cell_step, int32_t n_sel,
                                int32_t batch, int32_t N, int32_t Fa, int32_t Fb, int64_t chunk_stride, void* stream) {
    if (int rc = check_sel("stc_mix_grad_f32", cell0, cell_step, n_sel, batch, n_chunks)) return rc;
    STC_REQUIRE(N >= 1 && Fa >= 1 && Fb >= 1 && Fa <= 4096 && Fb <= 4096, STC_EINVAL, "stc_mix_grad_f32: N=%d Fa=%d Fb=%d", N, Fa, Fb);
    STC_REQUIRE(A && B && partials, STC_EINVAL, "stc_mix_grad_f32: null operand");
    STC_REQUIRE(chunk_stride == 0 || chunk_stride >= (int64_t)Fa * Fb, STC_EINVAL, "stc_mix_grad_f32: chunk stride %lld below the block of %d x %d", (long long)chunk_stride, Fa, Fb);
    const int tiles = ((Fa + 15) / 16) * ((Fb + 15) / 16), groups = (tiles + GG_TN_TPW * GG_WAVES - 1) / (GG_TN_TPW * GG_WAVES);
    hipLaunchKernelGGL(mix_grad_kernel, dim3((unsigned)n_chunks, (unsigned)groups), dim3(GG_THREADS), 0, static_cast<hipStream_t>(stream), A, B, partials,
                       Sel{cell0, cell_step, n_sel, batch}, N, Fa, Fb, (long long)(chunk_stride ? chunk_stride : (int64_t)Fa * Fb));
    STC_LAUNCH_CHECK("stc_mix_grad_f32 launch");
    return STC_OK;
}
