// Aggregation with a DENSE graph matrix on the exact-fp32 matrix cores of gfx950 (reference STC_GNN.py:37 with the learned dense Gs of
// STC_GNN.py:227-243, SURVEY F3: the reference's own spatial graph is dense; and the autograd of that product w.r.t. the features).
//
//     Y[b] = alpha * S . X[b] + beta * Y0[b]        S (n_rows, n_cols) row-major, X (batch, n_cols, F), Y / Y0 (batch, n_rows, F)
//
// With a full N x N pattern the CSR kernel (one wave per output row, every row gathering all N neighbour rows) is a dense matrix product
// done the hard way.  Here the contraction really is dense, so it goes to v_mfma_f32_16x16x4_f32 (fp32 operands and accumulator, bit-for-bit
// an fmaf chain: the reference's arithmetic, no split formats needed):
//   workgroup = 4 waves = 64 feature columns of one batch element; a wave owns 16 of them for a pass of up to 128 rows of S (32 accumulator
//   registers); S reaches the waves through LDS in blocks of 16 graph columns (8 KB, double-buffered, requested one block ahead into
//   registers), read back as one ds_read_b128 per row tile = the A operands of four MFMAs; the B operand is 4 x 16 of X per MFMA
//   (4-byte loads, 64-byte runs, requested one block ahead), reused by every row tile.  The contraction index a lane feeds to step s of a
//   block is k0 + 4 (lane / 16) + s on BOTH operands -- any bijection serves a sum.
// Bound by the fp32 matrix pipe (32 cycles per MFMA) -- when the launch fills the chip: the passes over the rows of S are workgroups of
// their own (blockIdx.y), and a pass is 128, 64 or 32 rows, the largest that still gives two workgroups per compute unit (BASELINE
// configuration 2, batch 32 x 4 column groups x N = 200: 128 workgroups of two passes each took 34 us, half the chip idle).
#include "stc_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int DA_THREADS = 256, DA_WAVES = DA_THREADS / 64;
constexpr int DA_MT_MAX = 8;        // row tiles of 16 per pass, at most: 128 rows of S per pass (32 accumulator registers)
constexpr int DA_KB = 16;           // graph columns per staged block
constexpr int DA_LD = DA_KB + 4;    // LDS row stride in floats: 16-byte reads of 16 consecutive rows fall on distinct banks

template <bool VEC>
__device__ __forceinline__ f32x4 load_s(const float* __restrict__ S, int row, int k, int n_rows, int n_cols) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < n_rows) {
        const float* p = S + (size_t)row * n_cols + k;
        if (VEC) {
            if (k < n_cols) v = *reinterpret_cast<const f32x4*>(p);                  // n_cols % 4 == 0: a whole quad is in range
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k + j < n_cols) v[j] = p[j];
        }
    }
    return v;
}

template <bool VEC, int DA_MT>
__global__ __launch_bounds__(DA_THREADS) void dense_agg_kernel(
    const float* __restrict__ S, int n_rows, int n_cols, const float* __restrict__ X, const float* __restrict__ Y0, float* __restrict__ Y,
    int F, int col_groups, float alpha, float beta) {
    __shared__ float tile[2][16 * DA_MT * DA_LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c = lane & 15, kq = lane >> 4;
    const int b = blockIdx.x / col_groups, col = ((blockIdx.x % col_groups) * DA_WAVES + wave) * 16 + c;
    const bool col_ok = col < F;                                     // (a wave past F still stages S and meets every barrier)
    const float* Xb = X + (size_t)b * n_cols * F + (col_ok ? col : 0);
    const int blocks = (n_cols + DA_KB - 1) / DA_KB;
    const int srow = t >> 1, sk = (t & 1) * 8;                       // staging: thread -> (row of the pass, 8 of the block's 16 columns)
    const bool stages = srow < 16 * DA_MT;                           // (passes of fewer than 128 rows: the upper threads stage nothing)
    {
        const int m0 = blockIdx.y * 16 * DA_MT;                      // this workgroup's pass over the rows of S
        const int srow_g = stages ? m0 + srow : n_rows;              // (n_rows: load_s returns zeros without touching memory)
        f32x4 acc[DA_MT];
#pragma unroll
        for (int mt = 0; mt < DA_MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int mts = min(DA_MT, (n_rows - m0 + 15) / 16);          // row tiles of this pass that exist (workgroup-uniform)
        f32x4 s0 = load_s<VEC>(S, srow_g, sk, n_rows, n_cols), s1 = load_s<VEC>(S, srow_g, sk + 4, n_rows, n_cols);
        float xb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xb[s] = (col_ok && 4 * kq + s < n_cols) ? Xb[(size_t)(4 * kq + s) * F] : 0.f;
        if (stages) {
            *reinterpret_cast<f32x4*>(&tile[0][srow * DA_LD + sk]) = s0;
            *reinterpret_cast<f32x4*>(&tile[0][srow * DA_LD + sk + 4]) = s1;
        }
        __syncthreads();
        for (int blk = 0; blk < blocks; ++blk) {
            const int kn = (blk + 1) * DA_KB;                          // next block (past the end: zeros, never used)
            float xn[4];
            if (blk + 1 < blocks) {
                s0 = load_s<VEC>(S, srow_g, kn + sk, n_rows, n_cols);
                s1 = load_s<VEC>(S, srow_g, kn + sk + 4, n_rows, n_cols);
#pragma unroll
                for (int s = 0; s < 4; ++s) xn[s] = (col_ok && kn + 4 * kq + s < n_cols) ? Xb[(size_t)(kn + 4 * kq + s) * F] : 0.f;
            }
            const float* cur = tile[blk & 1];
#pragma unroll
            for (int mt = 0; mt < DA_MT; ++mt) {
                if (mt < mts) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(&cur[(16 * mt + c) * DA_LD + 4 * kq]);
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], xb[s], acc[mt], 0, 0, 0);
                }
            }
            if (blk + 1 < blocks) {
                float* nxt = tile[(blk + 1) & 1];                      // last read two blocks ago: the barrier of the previous block covers it
                if (stages) {
                    *reinterpret_cast<f32x4*>(&nxt[srow * DA_LD + sk]) = s0;
                    *reinterpret_cast<f32x4*>(&nxt[srow * DA_LD + sk + 4]) = s1;
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) xb[s] = xn[s];
            }
            __syncthreads();
        }
        if (col_ok) {
#pragma unroll
            for (int mt = 0; mt < DA_MT; ++mt) {
                if (mt < mts) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = m0 + 16 * mt + 4 * kq + r;                 // D layout: lane (c, kq) holds rows 4 kq + r of column c
                        if (row < n_rows) {
                            const size_t o = ((size_t)b * n_rows + row) * F + col;
                            float v = alpha * acc[mt][r];
                            if (beta != 0.f) v = fmaf(beta, Y0[o], v);
                            Y[o] = v;
                        }
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" int stc_dense_agg_f32(const float* S, int32_t n_rows, int32_t n_cols, const float* X, const float* Y0, float* Y,
                                 int32_t batch, int32_t F, float alpha, float beta, void* stream) {
    STC_REQUIRE(n_rows >= 0 && n_cols >= 0 && batch >= 0 && F >= 0, STC_EINVAL,
                "stc_dense_agg_f32: negative size (n_rows=%d n_cols=%d batch=%d F=%d)", n_rows, n_cols, batch, F);
    if (n_rows == 0 || batch == 0 || F == 0) return STC_OK;
    STC_REQUIRE(S && Y && (n_cols == 0 || X), STC_EINVAL, "stc_dense_agg_f32: null S / X / Y");
    STC_REQUIRE(beta == 0.f || Y0, STC_EINVAL, "stc_dense_agg_f32: beta != 0 needs Y0");
    STC_REQUIRE(X != Y, STC_EINVAL, "stc_dense_agg_f32: X must not alias Y");
    STC_REQUIRE((long long)batch * n_rows * F < (1ll << 40) && (long long)n_rows * n_cols < (1ll << 31), STC_ELIMIT, "stc_dense_agg_f32: operand too large");
    const int col_groups = (F + 16 * DA_WAVES - 1) / (16 * DA_WAVES);
    const long long blocks = (long long)batch * col_groups;
    STC_REQUIRE(blocks < (1ll << 31), STC_ELIMIT, "stc_dense_agg_f32: %lld workgroups", blocks);
    const bool vec = n_cols % 4 == 0 && (reinterpret_cast<uintptr_t>(S) & 15) == 0;
    // rows of S per pass: 128 while that still gives ~two workgroups per compute unit, else 64, else 32
    int mt = DA_MT_MAX;
    while (mt > 2 && blocks * ((n_rows + 16 * mt - 1) / (16 * mt)) < 2 * stc::kNumCu) mt /= 2;
    const unsigned passes = (unsigned)((n_rows + 16 * mt - 1) / (16 * mt));
    STC_REQUIRE(passes <= 65535u, STC_ELIMIT, "stc_dense_agg_f32: %u row passes", passes);
    using Kernel = void (*)(const float*, int, int, const float*, const float*, float*, int, int, float, float);
    const Kernel kern = mt == 8 ? (vec ? dense_agg_kernel<true, 8> : dense_agg_kernel<false, 8>)
                      : mt == 4 ? (vec ? dense_agg_kernel<true, 4> : dense_agg_kernel<false, 4>)
                                : (vec ? dense_agg_kernel<true, 2> : dense_agg_kernel<false, 2>);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, passes), dim3(DA_THREADS), 0, static_cast<hipStream_t>(stream),
                       S, n_rows, n_cols, X, Y0, Y, F, col_groups, alpha, beta);
    STC_LAUNCH_CHECK("stc_dense_agg_f32 launch");
    return STC_OK;
}
