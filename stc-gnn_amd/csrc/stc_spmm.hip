// CSR SpMM / SDDMM for the spatial (1-mode) aggregation of STC-GNN on gfx950.
//
//   Y[b,i,:] = alpha * sum_j val[j] * X[b, col[j], :] + beta * Y0[b,i,:]
//
// Replaces torch.einsum('bncl,nm->bmcl', X, T_n(Gs)) (reference STC_GNN.py:37)
// and, with (alpha,beta) = (2,-1), one step of the Chebyshev recurrence of
// STC_GNN.py:28 applied on the feature side.  HBM-bound: a feature row is
// F = C*L contiguous floats (4 KiB at C=32, L=32) and is streamed with one
// 16-byte load per lane, 1 KiB per wave instruction.
//
// Layout of one launch (vector path):
//   workgroup  = SPMM_ROWS consecutive output rows of one batch element
//   CSR stage  = rowptr slice + the (col,val) segment of those rows -> LDS, one coalesced pass
//   wave       = one output row at a time; (col,val) read from LDS are wave-uniform and
//                moved to SGPRs (readfirstlane) so the row base address is scalar
//   lane       = VPT float4 chunks of the row, 4 neighbour rows in flight (16 loads/lane)
//   blocks     = remapped so each XCD walks a contiguous band of rows: the rows a band
//                gathers (its own +- the graph bandwidth) stay in that XCD's 4 MiB L2
#include "stc_common.h"

#include <cstdlib>

#ifndef STC_SPMM_DEFAULT_VARIANT
#define STC_SPMM_DEFAULT_VARIANT 10
#endif

namespace {

constexpr int SPMM_THREADS = 256;
constexpr int SPMM_WAVES = SPMM_THREADS / 64;
constexpr int SPMM_SEG_CAP = 1024;   // CSR entries staged in LDS per workgroup

__device__ __forceinline__ float uniform_f(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

__device__ __forceinline__ void fma4(float4& acc, float s, const float4& x) {
    acc.x = fmaf(s, x.x, acc.x);
    acc.y = fmaf(s, x.y, acc.y);
    acc.z = fmaf(s, x.z, acc.z);
    acc.w = fmaf(s, x.w, acc.w);
}

// ROWS consecutive output rows per workgroup; NT: non-temporal stores of Y (written once, never re-read by
// this launch: keeps the output stream from evicting the X rows the neighbours still need from L2).
// gridDim.z > 1 splits the feature row into column blocks of 64*VPT float4 (one 1 KiB piece per lane-row at
// VPT = 1): a tile's gather set (~3x its rows for a banded graph) then fits the CU's 32 KiB L1.
template <int VPT, int ROWS, bool NT>
__global__ __launch_bounds__(SPMM_THREADS) void spmm_wave_row_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ colidx, const float* __restrict__ val,
    int n_rows, int n_cols, const float4* __restrict__ X, const float4* Y0, float4* Y,
    int F4, float alpha, float beta, int n_tiles) {
    constexpr int SPMM_ROWS = ROWS;
    __shared__ int s_rp[SPMM_ROWS + 1];
    __shared__ int s_col[SPMM_SEG_CAP];
    __shared__ float s_val[SPMM_SEG_CAP];

    const int tile = stc_xcd_tile(blockIdx.x, n_tiles);
    if (tile < 0) return;                       // whole workgroup leaves together
    const int b = blockIdx.y;
    const int row0 = tile * SPMM_ROWS;
    const int nr = min(SPMM_ROWS, n_rows - row0);

    if ((int)threadIdx.x <= nr) s_rp[threadIdx.x] = rowptr[row0 + threadIdx.x];
    __syncthreads();
    const int seg0 = s_rp[0];
    const int seg_n = min(s_rp[nr] - seg0, SPMM_SEG_CAP);
    for (int t = threadIdx.x; t < seg_n; t += SPMM_THREADS) {
        s_col[t] = colidx[seg0 + t];
        s_val[t] = val[seg0 + t];
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const float4* Xb = X + (size_t)b * n_cols * F4;
    const size_t out_base = (size_t)b * n_rows * F4;

    for (int r = wave; r < nr; r += SPMM_WAVES) {
        const int js = s_rp[r] - seg0;
        const int je = s_rp[r + 1] - seg0;
        const size_t orow = out_base + (size_t)(row0 + r) * F4;
        for (int cb = blockIdx.z * 64 * VPT; cb < F4; cb += gridDim.z * 64 * VPT) {
            float4 acc[VPT];
#pragma unroll
            for (int p = 0; p < VPT; ++p) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);

            auto entry = [&](int j, int& c, float& v) {
                if (j < SPMM_SEG_CAP) {
                    c = s_col[j];
                    v = s_val[j];
                } else {                       // rows longer than the staged segment (dense graphs)
                    c = colidx[seg0 + j];
                    v = val[seg0 + j];
                }
                c = __builtin_amdgcn_readfirstlane(c);
                v = uniform_f(v);
            };

            int j = js;
            for (; j + 4 <= je; j += 4) {
                int c[4];
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) entry(j + u, c[u], v[u]);
                float4 x[4][VPT];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4* xr = Xb + (size_t)c[u] * F4;
#pragma unroll
                    for (int p = 0; p < VPT; ++p) {
                        const int ch = cb + lane + 64 * p;
                        x[u][p] = ch < F4 ? xr[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int p = 0; p < VPT; ++p) fma4(acc[p], v[u], x[u][p]);
            }
            for (; j < je; ++j) {
                int c;
                float v;
                entry(j, c, v);
                const float4* xr = Xb + (size_t)c * F4;
#pragma unroll
                for (int p = 0; p < VPT; ++p) {
                    const int ch = cb + lane + 64 * p;
                    if (ch < F4) fma4(acc[p], v, xr[ch]);
                }
            }
#pragma unroll
            for (int p = 0; p < VPT; ++p) {
                const int ch = cb + lane + 64 * p;
                if (ch < F4) {
                    float4 o = make_float4(alpha * acc[p].x, alpha * acc[p].y, alpha * acc[p].z, alpha * acc[p].w);
                    if (beta != 0.f) {
                        float4 y0;
                        if (NT) {   // read once (often the very line this thread overwrites): keep it out of the gather's L2
                            using v4f = __attribute__((ext_vector_type(4))) float;
                            const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(&Y0[orow + ch]));
                            y0 = make_float4(t[0], t[1], t[2], t[3]);
                        } else {
                            y0 = Y0[orow + ch];
                        }
                        o.x = fmaf(beta, y0.x, o.x);
                        o.y = fmaf(beta, y0.y, o.y);
                        o.z = fmaf(beta, y0.z, o.z);
                        o.w = fmaf(beta, y0.w, o.w);
                    }
                    if (NT) {
                        using v4f = __attribute__((ext_vector_type(4))) float;
                        __builtin_nontemporal_store(v4f{o.x, o.y, o.z, o.w}, reinterpret_cast<v4f*>(&Y[orow + ch]));
                    } else {
                        Y[orow + ch] = o;
                    }
                }
            }
        }
    }
}


// ---- row-blocked (BCSR 4x1) variant: one wave produces 4 consecutive output rows -----------------------------
// Same skeleton as spmm_wave_row_kernel (CSR segment staged in LDS, scalar column index, 16-byte streaming of
// the neighbour row, 4 entries = 16 loads in flight per lane), but every fetched neighbour row is accumulated
// into the 4 rows of the block with its 4 wave-uniform values.  (An LDS-staged tile variant -- distinct rows
// of 8 output rows copied to LDS per 1 KiB column block -- was measured at 234 us vs 128 us for the direct
// kernel: three dependent memory latencies per workgroup and too few bytes in flight; dropped.)
constexpr int BR = STC_SPMM_BLOCK_ROWS;
constexpr int BC_BLOCKS = 8;          // row blocks per workgroup (2 per wave)
constexpr int BC_CAP = 512;           // block entries staged in LDS per workgroup

template <int VPT>
__global__ __launch_bounds__(SPMM_THREADS) void spmm_bcsr_kernel(
    const int* __restrict__ blk_ptr, const int* __restrict__ blk_cols, const float* __restrict__ blk_vals,
    int n_rows, int n_cols, const float4* __restrict__ X, const float4* Y0, float4* Y,
    int F4, float alpha, float beta, int n_blocks, int n_tiles) {
    __shared__ int s_bp[BC_BLOCKS + 1];
    __shared__ int s_col[BC_CAP];
    __shared__ float s_val[BC_CAP * BR];
    using v4f = __attribute__((ext_vector_type(4))) float;

    const int tile = stc_xcd_tile(blockIdx.x, n_tiles);
    if (tile < 0) return;
    const int b = blockIdx.y;
    const int blk0 = tile * BC_BLOCKS;
    const int nb = min(BC_BLOCKS, n_blocks - blk0);
    if ((int)threadIdx.x <= nb) s_bp[threadIdx.x] = blk_ptr[blk0 + threadIdx.x];
    __syncthreads();
    const int seg0 = s_bp[0];
    const int seg_n = min(s_bp[nb] - seg0, BC_CAP);
    for (int t = threadIdx.x; t < seg_n; t += SPMM_THREADS) s_col[t] = blk_cols[seg0 + t];
    for (int t = threadIdx.x; t < seg_n * BR; t += SPMM_THREADS) s_val[t] = blk_vals[(size_t)seg0 * BR + t];
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const float4* Xb = X + (size_t)b * n_cols * F4;

    for (int bi = wave; bi < nb; bi += SPMM_WAVES) {
        const int js = s_bp[bi] - seg0, je = s_bp[bi + 1] - seg0;
        const int row_base = (blk0 + bi) * BR;
        const int rows_here = min(BR, n_rows - row_base);
        for (int cb = 0; cb < F4; cb += 64 * VPT) {
            float4 acc[BR][VPT];
#pragma unroll
            for (int r = 0; r < BR; ++r)
#pragma unroll
                for (int p = 0; p < VPT; ++p) acc[r][p] = make_float4(0.f, 0.f, 0.f, 0.f);

            auto entry = [&](int j, int& c, float (&v)[BR]) {
                if (j < BC_CAP) {
                    c = s_col[j];
#pragma unroll
                    for (int r = 0; r < BR; ++r) v[r] = s_val[j * BR + r];
                } else {                       // block lists longer than the staged segment
                    c = blk_cols[seg0 + j];
#pragma unroll
                    for (int r = 0; r < BR; ++r) v[r] = blk_vals[(size_t)(seg0 + j) * BR + r];
                }
                c = __builtin_amdgcn_readfirstlane(c);
#pragma unroll
                for (int r = 0; r < BR; ++r) v[r] = uniform_f(v[r]);
            };

            int j = js;
            for (; j + 4 <= je; j += 4) {
                int c[4];
                float v[4][BR];
#pragma unroll
                for (int u = 0; u < 4; ++u) entry(j + u, c[u], v[u]);
                float4 x[4][VPT];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4* xr = Xb + (size_t)c[u] * F4;
#pragma unroll
                    for (int p = 0; p < VPT; ++p) {
                        const int ch = cb + lane + 64 * p;
                        x[u][p] = ch < F4 ? xr[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < BR; ++r)
#pragma unroll
                        for (int p = 0; p < VPT; ++p) fma4(acc[r][p], v[u][r], x[u][p]);
            }
            for (; j < je; ++j) {
                int c;
                float v[BR];
                entry(j, c, v);
                const float4* xr = Xb + (size_t)c * F4;
#pragma unroll
                for (int p = 0; p < VPT; ++p) {
                    const int ch = cb + lane + 64 * p;
                    if (ch < F4) {
                        const float4 xv = xr[ch];
#pragma unroll
                        for (int r = 0; r < BR; ++r) fma4(acc[r][p], v[r], xv);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < BR; ++r) {
                if (r < rows_here) {
                    const size_t orow = ((size_t)b * n_rows + row_base + r) * F4;
#pragma unroll
                    for (int p = 0; p < VPT; ++p) {
                        const int ch = cb + lane + 64 * p;
                        if (ch < F4) {
                            float4 o = make_float4(alpha * acc[r][p].x, alpha * acc[r][p].y, alpha * acc[r][p].z, alpha * acc[r][p].w);
                            if (beta != 0.f) {
                                const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(&Y0[orow + ch]));
                                o.x = fmaf(beta, t[0], o.x); o.y = fmaf(beta, t[1], o.y);
                                o.z = fmaf(beta, t[2], o.z); o.w = fmaf(beta, t[3], o.w);
                            }
                            __builtin_nontemporal_store(v4f{o.x, o.y, o.z, o.w}, reinterpret_cast<v4f*>(&Y[orow + ch]));
                        }
                    }
                }
            }
        }
    }
}

// Any F, any alignment (SF shape: F = C*L = 85): lanes_per_row threads share a row.
__global__ __launch_bounds__(SPMM_THREADS) void spmm_generic_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ colidx, const float* __restrict__ val,
    int n_rows, int n_cols, const float* __restrict__ X, const float* Y0, float* Y,
    int F, float alpha, float beta, int lanes_per_row) {
    const int rows_per_block = SPMM_THREADS / lanes_per_row;
    const int r = threadIdx.x / lanes_per_row;
    const int lr = threadIdx.x % lanes_per_row;
    const int i = blockIdx.x * rows_per_block + r;
    if (i >= n_rows) return;
    const int b = blockIdx.y;
    const int js = rowptr[i], je = rowptr[i + 1];
    const float* Xb = X + (size_t)b * n_cols * F;
    const size_t orow = ((size_t)b * n_rows + i) * F;
    for (int f = lr; f < F; f += lanes_per_row) {
        float acc = 0.f;
        for (int j = js; j < je; ++j) acc = fmaf(val[j], Xb[(size_t)colidx[j] * F + f], acc);
        float o = alpha * acc;
        if (beta != 0.f) o = fmaf(beta, Y0[orow + f], o);
        Y[orow + f] = o;
    }
}

// out[j] (+)= alpha * sum_b <A[b,i,:], Bm[b,col[j],:]> ; one wave per stored entry.
__global__ __launch_bounds__(SPMM_THREADS) void sddmm_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ colidx, int n_rows, int n_cols,
    const float* __restrict__ A, const float* __restrict__ Bm, float* out,
    int batch, int F, float alpha, int accumulate) {
    const int i = blockIdx.x;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int js = rowptr[i], je = rowptr[i + 1];
    for (int j = js + wave; j < je; j += SPMM_WAVES) {
        const int c = colidx[j];
        float s = 0.f;
        for (int b = 0; b < batch; ++b) {
            const float* a = A + ((size_t)b * n_rows + i) * F;
            const float* bm = Bm + ((size_t)b * n_cols + c) * F;
            for (int f = lane; f < F; f += 64) s = fmaf(a[f], bm[f], s);
        }
        s = stc_wave_sum(s);
        if (lane == 0) out[j] = accumulate ? fmaf(alpha, s, out[j]) : alpha * s;
    }
}

int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace

extern "C" int stc_csr_spmm_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                                int32_t n_rows, int32_t n_cols,
                                const float* X, const float* Y0, float* Y,
                                int32_t batch, int32_t F, float alpha, float beta, void* stream) {
    STC_REQUIRE(n_rows >= 0 && n_cols >= 0 && batch >= 0 && F >= 0, STC_EINVAL,
                "stc_csr_spmm_f32: negative size (n_rows=%d n_cols=%d batch=%d F=%d)", n_rows, n_cols, batch, F);
    if (n_rows == 0 || batch == 0 || F == 0) return STC_OK;
    STC_REQUIRE(rowptr && Y, STC_EINVAL, "stc_csr_spmm_f32: null rowptr/Y");
    STC_REQUIRE(n_cols > 0 && X, STC_EINVAL, "stc_csr_spmm_f32: null X or n_cols == 0 with rows to produce");
    // colidx/val may be null for a graph without edges (rowptr all zero): they are then never read
    STC_REQUIRE(beta == 0.f || Y0, STC_EINVAL, "stc_csr_spmm_f32: beta != 0 needs Y0");
    STC_REQUIRE(batch <= 65535, STC_ELIMIT, "stc_csr_spmm_f32: batch %d > 65535 (grid.y)", batch);
    STC_REQUIRE(X != Y, STC_EINVAL, "stc_csr_spmm_f32: X must not alias Y");
    hipStream_t s = static_cast<hipStream_t>(stream);

    const bool vec = (F % 4 == 0) && stc::aligned16(X) && stc::aligned16(Y) && (Y0 == nullptr || stc::aligned16(Y0)) && F >= 64;
    if (vec) {
        const int F4 = F / 4;
        const float4* X4 = reinterpret_cast<const float4*>(X);
        const float4* Y04 = reinterpret_cast<const float4*>(Y0);
        float4* Y4 = reinterpret_cast<float4*>(Y);
        // STC_SPMM_VARIANT (A/B runs): 0 = whole rows, 8 rows per workgroup; 1 = 1 KiB column blocks (grid.z);
        // 2 = whole rows, 16 rows per workgroup; +10 = non-temporal stores
        static const int variant = [] { const char* e = std::getenv("STC_SPMM_VARIANT"); return e ? std::atoi(e) : STC_SPMM_DEFAULT_VARIANT; }();
        const bool nt = variant >= 10;
        const int shape = variant % 10;
#define STC_SPMM_LAUNCH(VPT_, ROWS_, NT_, GZ_)                                                                          \
    do {                                                                                                               \
        const int n_tiles = (n_rows + ROWS_ - 1) / ROWS_;                                                              \
        const int per = (n_tiles + stc::kNumXcd - 1) / stc::kNumXcd;                                                   \
        dim3 grid(per * stc::kNumXcd, batch, GZ_), block(SPMM_THREADS);                                                \
        hipLaunchKernelGGL((spmm_wave_row_kernel<VPT_, ROWS_, NT_>), grid, block, 0, s, rowptr, colidx, val, n_rows,   \
                           n_cols, X4, Y04, Y4, F4, alpha, beta, n_tiles);                                             \
    } while (0)
#define STC_SPMM_BY_NT(VPT_, ROWS_, GZ_) do { if (nt) STC_SPMM_LAUNCH(VPT_, ROWS_, true, GZ_); else STC_SPMM_LAUNCH(VPT_, ROWS_, false, GZ_); } while (0)
        if (shape == 1 && F4 > 64) {
            STC_SPMM_BY_NT(1, 8, (F4 + 63) / 64);
        } else if (shape == 2) {
            if (F4 <= 64) STC_SPMM_BY_NT(1, 16, 1); else if (F4 <= 128) STC_SPMM_BY_NT(2, 16, 1); else STC_SPMM_BY_NT(4, 16, 1);
        } else {
            if (F4 <= 64) STC_SPMM_BY_NT(1, 8, 1); else if (F4 <= 128) STC_SPMM_BY_NT(2, 8, 1); else STC_SPMM_BY_NT(4, 8, 1);
        }
#undef STC_SPMM_BY_NT
#undef STC_SPMM_LAUNCH
    } else {
        const int lanes = next_pow2(F) < SPMM_THREADS ? next_pow2(F) : SPMM_THREADS;
        const int rows_per_block = SPMM_THREADS / lanes;
        dim3 grid((n_rows + rows_per_block - 1) / rows_per_block, batch), block(SPMM_THREADS);
        hipLaunchKernelGGL(spmm_generic_kernel, grid, block, 0, s, rowptr, colidx, val, n_rows, n_cols, X, Y0, Y, F, alpha, beta, lanes);
    }
    STC_LAUNCH_CHECK("stc_csr_spmm_f32 launch");
    return STC_OK;
}


extern "C" int stc_bcsr_spmm_f32(const int32_t* blk_ptr, const int32_t* blk_cols, const float* blk_vals,
                                 int32_t n_rows, int32_t n_cols,
                                 const float* X, const float* Y0, float* Y,
                                 int32_t batch, int32_t F, float alpha, float beta, void* stream) {
    STC_REQUIRE(n_rows >= 0 && n_cols >= 0 && batch >= 0 && F >= 0, STC_EINVAL, "stc_bcsr_spmm_f32: negative size");
    if (n_rows == 0 || batch == 0 || F == 0) return STC_OK;
    STC_REQUIRE(blk_ptr && X && Y, STC_EINVAL, "stc_bcsr_spmm_f32: null pointer");
    STC_REQUIRE(beta == 0.f || Y0, STC_EINVAL, "stc_bcsr_spmm_f32: beta != 0 needs Y0");
    STC_REQUIRE(X != Y, STC_EINVAL, "stc_bcsr_spmm_f32: X must not alias Y");
    STC_REQUIRE(F % 4 == 0, STC_EINVAL, "stc_bcsr_spmm_f32: F=%d must be a multiple of 4", F);
    STC_REQUIRE(stc::aligned16(X) && stc::aligned16(Y) && (!Y0 || stc::aligned16(Y0)), STC_EALIGN,
                "stc_bcsr_spmm_f32: X / Y / Y0 must be 16-byte aligned");
    STC_REQUIRE(batch <= 65535, STC_ELIMIT, "stc_bcsr_spmm_f32: batch %d > 65535 (grid.y)", batch);
    const int F4 = F / 4;
    const int n_blocks = (n_rows + BR - 1) / BR;
    const int n_tiles = (n_blocks + BC_BLOCKS - 1) / BC_BLOCKS;
    const int per = (n_tiles + stc::kNumXcd - 1) / stc::kNumXcd;
    const dim3 grid(per * stc::kNumXcd, batch), block(SPMM_THREADS);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float4* X4 = reinterpret_cast<const float4*>(X);
    const float4* Y04 = reinterpret_cast<const float4*>(Y0);
    float4* Y4 = reinterpret_cast<float4*>(Y);
    if (F4 <= 64)
        hipLaunchKernelGGL(spmm_bcsr_kernel<1>, grid, block, 0, s, blk_ptr, blk_cols, blk_vals, n_rows, n_cols, X4, Y04, Y4, F4, alpha, beta, n_blocks, n_tiles);
    else if (F4 <= 128)
        hipLaunchKernelGGL(spmm_bcsr_kernel<2>, grid, block, 0, s, blk_ptr, blk_cols, blk_vals, n_rows, n_cols, X4, Y04, Y4, F4, alpha, beta, n_blocks, n_tiles);
    else
        hipLaunchKernelGGL(spmm_bcsr_kernel<4>, grid, block, 0, s, blk_ptr, blk_cols, blk_vals, n_rows, n_cols, X4, Y04, Y4, F4, alpha, beta, n_blocks, n_tiles);
    STC_LAUNCH_CHECK("stc_bcsr_spmm_f32 launch");
    return STC_OK;
}

extern "C" int stc_csr_sddmm_f32(const int32_t* rowptr, const int32_t* colidx,
                                 int32_t n_rows, int32_t n_cols,
                                 const float* A, const float* Bm, float* out,
                                 int32_t batch, int32_t F, float alpha, int32_t accumulate, void* stream) {
    STC_REQUIRE(n_rows >= 0 && n_cols >= 0 && batch >= 0 && F >= 0, STC_EINVAL, "stc_csr_sddmm_f32: negative size");
    if (n_rows == 0) return STC_OK;
    STC_REQUIRE(rowptr && colidx && out, STC_EINVAL, "stc_csr_sddmm_f32: null rowptr/colidx/out");
    STC_REQUIRE((batch == 0 || F == 0) || (A && Bm), STC_EINVAL, "stc_csr_sddmm_f32: null A/Bm");
    hipLaunchKernelGGL(sddmm_kernel, dim3(n_rows), dim3(SPMM_THREADS), 0, static_cast<hipStream_t>(stream),
                       rowptr, colidx, n_rows, n_cols, A, Bm, out, batch, F, alpha, accumulate);
    STC_LAUNCH_CHECK("stc_csr_sddmm_f32 launch");
    return STC_OK;
}
