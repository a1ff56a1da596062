// Gradient of the category graph through ONE BDG_Dif (reference STC_GNN.py:38-42 through autograd: the 2-mode product's second operand) for
// FEW categories on the exact-fp32 matrix cores of gfx950:
//
//     dT_c[p, d] = sum_r sum_o U_c[r, p, o] * dY[r, d, o],      U_c[r] = sum_n Z_n[r] . W[(n, c, :), :]      (r: nodes; p, d: categories)
//
// The matrix-core node backward (csrc/stc_node_mfma.hip) leaves dT_c to its caller.  The rows of floor(16 / C) consecutive nodes are one
// tile of rpt = floor(16 / C) C <= 16 rows (stc_hip/ops.py _node_pack): U_c of a tile is a 16 x (Ks L) x Ho product, its contribution to dT_c a
// 16 x Ho x 16 one whose DIAGONAL C x C blocks are the nodes' -- the off-diagonal ones pair rows of different nodes and are dropped by the
// final sum; lanes past the tile's rows read its last row of Z (never used) and zeros of dY.  As library
// GEMMs (Q_n = Z_n^T . dY over the rows, then a contraction with W) this took two launches of 41 us at BASELINE configuration 2's shape, where
// the node backward itself takes 22.
//
// One wave per tile: W as B operands in registers (the same for every tile), Z rows as A operands straight from HBM (16-byte loads; the
// contraction slot of step s is column kcol(s, kq) on both operands, as in stc_node_mfma.hip), U_c from accumulator to row layout through
// a per-wave LDS tile, dY rows as B operands (lane (d, kq) feeds o = (Ho / 4) kq + s: one contiguous run of its row).  dT_c stays in four
// registers per c across the wave's tiles; fixed-order sums over waves, workgroups and diagonal blocks (bitwise reproducible).  T_0 = I is a
// constant of the Chebyshev stack: dT_0 is written as zeros, as stc_bdg_node_bwd_f32 writes it, and costs nothing.
#include "stc_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int DT_THREADS = 256, DT_WAVES = DT_THREADS / 64;
constexpr int DT_MAX_GRID = 512;            // two workgroups per compute unit
constexpr int DT_MAX_K = 3;

struct DtArgs {
    const float* Z[DT_MAX_K];               // (tiles, 16, L) each
    const float* W;                         // (Ks * Kc * Lw, Ho)
    const float* dY;                        // (tiles, 16, Ho)
    float* partial;                         // (grid, Kc, 16, 16)
    int tiles, Lw, rpt, rows;               // rpt: rows per tile (<= 16; tiles are rpt rows apart); the last tile may hold fewer nodes
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// slab column that lane quarter q feeds at contraction step s (LQ = L / 4 steps per slab): whole float4s first, then the remainder columns
template <int LQ>
__host__ __device__ constexpr int kcol(int s, int q) {
    constexpr int N16 = LQ / 4;
    return s < 4 * N16 ? 16 * (s / 4) + 4 * q + (s % 4) : 16 * N16 + 4 * (s - 4 * N16) + q;
}

template <int K, int LQ, int OT>           // Ks = Kc = K, L = 4 LQ, Ho = 16 OT
__global__ __launch_bounds__(DT_THREADS) void mix_dt_kernel(DtArgs a) {
    constexpr int L = 4 * LQ, HO = 16 * OT, N16 = LQ / 4, NREM = LQ - 4 * N16, TS = HO + 4, RUN = HO / 4;
    __shared__ __attribute__((aligned(16))) float tile[DT_WAVES][K][16 * TS];
    __shared__ float red[DT_WAVES][K][256];
    const int t = threadIdx.x, lane = t & 63, j = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    float wr[K][K][LQ][OT];                 // B operands of the projection: W[(n, c, kcol(s, kq)), 16 ot + j], c >= 1; rows >= Lw (pad columns) are zero
#pragma unroll
    for (int n = 0; n < K; ++n)
#pragma unroll
        for (int c = 1; c < K; ++c)
#pragma unroll
            for (int s = 0; s < LQ; ++s) {
                const int l = kcol<LQ>(s, kq);
                const bool ok = l < a.Lw;
                const float* row = a.W + ((size_t)(n * K + c) * a.Lw + (ok ? l : 0)) * HO + j;
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) {
                    const float v = row[16 * ot];
                    wr[n][c][s][ot] = ok ? v : 0.f;
                }
            }

    f32x4 dT[K];
#pragma unroll
    for (int c = 0; c < K; ++c) dT[c] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tl = blockIdx.x * DT_WAVES + wave; tl < a.tiles; tl += gridDim.x * DT_WAVES) {
        const int lim = min(a.rpt, a.rows - tl * a.rpt);         // rows of this tile (the last one: what is left)
        const int jr = j < lim ? j : lim - 1;
        const bool live = j < lim;
        // A operands: this lane's row of every slab; B operands of the second product: its run of the dY row
        f32x4 zv[K][N16 > 0 ? N16 : 1];
        float zr[K][NREM > 0 ? NREM : 1];
#pragma unroll
        for (int n = 0; n < K; ++n) {
            const float* row = a.Z[n] + ((size_t)tl * a.rpt + jr) * L;
#pragma unroll
            for (int m = 0; m < N16; ++m) zv[n][m] = *reinterpret_cast<const f32x4*>(row + 16 * m + 4 * kq);
#pragma unroll
            for (int u = 0; u < NREM; ++u) zr[n][u] = row[16 * N16 + 4 * u + kq];
        }
        f32x4 dyv[RUN / 4];
        {
            const float* row = a.dY + ((size_t)tl * a.rpt + jr) * HO + RUN * kq;
#pragma unroll
            for (int q = 0; q < RUN / 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * q);
                dyv[q] = live ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // U_c = sum_n Z_n . W_{n,c}: accumulator layout (rows 4 kq + r, column 16 ot + j) -> the wave's LDS tile in row layout
#pragma unroll
        for (int c = 1; c < K; ++c) {
            f32x4 acc[OT];
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) acc[ot] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < K; ++n)
#pragma unroll
                for (int s = 0; s < LQ; ++s) {
                    const float av = s < 4 * N16 ? zv[n][s / 4][s % 4] : zr[n][s - 4 * N16];
#pragma unroll
                    for (int ot = 0; ot < OT; ++ot) acc[ot] = mfma4(av, wr[n][c][s][ot], acc[ot]);
                }
#pragma unroll
            for (int ot = 0; ot < OT; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) tile[wave][c][(4 * kq + r) * TS + 16 * ot + j] = acc[ot][r];
        }
        __builtin_amdgcn_wave_barrier();            // (LDS operations of one wave complete in order)
        // dT_c += U_c . dY^T: A = row j of U_c, slots o = RUN kq + s; B = row j of dY, same slots
#pragma unroll
        for (int c = 1; c < K; ++c)
#pragma unroll
            for (int q = 0; q < RUN / 4; ++q) {
                const f32x4 ua = *reinterpret_cast<const f32x4*>(&tile[wave][c][j * TS + RUN * kq + 4 * q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) dT[c] = mfma4(ua[e], dyv[q][e], dT[c]);
            }
        __builtin_amdgcn_wave_barrier();            // the next tile overwrites the LDS tile
    }

    // waves of the workgroup, in wave order: lane (d = j, kq), register r holds dT[p = 4 kq + r][d]
#pragma unroll
    for (int c = 0; c < K; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][c][(4 * kq + r) * 16 + j] = dT[c][r];
    __syncthreads();
    for (int e = t; e < K * 256; e += DT_THREADS) {
        const int c = e >> 8, i = e & 255;
        float s = red[0][c][i];
#pragma unroll
        for (int w = 1; w < DT_WAVES; ++w) s += red[w][c][i];
        a.partial[((size_t)blockIdx.x * K + c) * 256 + i] = s;
    }
}

// dTc[c][p][d] = sum over workgroups and over the 16 / C diagonal blocks of partial[g][c][b C + p][b C + d]: one wave per element, lane l
// takes the workgroups g = l, l + 64, .. in order, then the wave's shuffle tree -- a fixed order
__global__ __launch_bounds__(256) void mix_dt_reduce_kernel(const float* __restrict__ partial, int grid, int K, int C, float* __restrict__ dTc) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= K * C * C) return;
    const int c = e / (C * C), p = (e / C) % C, d = e % C;
    float s = 0.f;
    for (int g = lane; g < grid; g += 64)
        for (int b = 0; b < 16 / C; ++b) s += partial[((size_t)g * K + c) * 256 + (b * C + p) * 16 + b * C + d];
    s = stc_wave_sum(s);
    if (lane == 0) dTc[e] = s;
}

template <int K, int LQ, int OT>
int launch_dt(const DtArgs& a, int grid, hipStream_t s) {
    hipLaunchKernelGGL((mix_dt_kernel<K, LQ, OT>), dim3(grid), dim3(DT_THREADS), 0, s, a);
    STC_LAUNCH_CHECK("stc_mix_dt_f32 launch");
    return STC_OK;
}

}  // namespace

extern "C" size_t stc_mix_dt_workspace_bytes(int32_t Ks) {
    return Ks >= 1 && Ks <= DT_MAX_K ? (size_t)DT_MAX_GRID * Ks * 256 * sizeof(float) : 0;
}

extern "C" int stc_mix_dt_supported(int32_t Ks, int32_t Kc, int32_t C, int32_t L, int32_t Ho) {
    return Ks == Kc && Ks >= 1 && Ks <= DT_MAX_K && C >= 1 && C <= 16 && (L == 20 || L == 32) && (Ho == 16 || Ho == 32);
}

extern "C" int stc_mix_dt_f32(const float* const* Z, int32_t Ks, const float* W, const float* dY, float* dTc, void* workspace,
                              size_t workspace_bytes, int64_t rows, int32_t C, int32_t L, int32_t Lw, int32_t Ho, void* stream) {
    STC_REQUIRE(stc_mix_dt_supported(Ks, Ks, C, L, Ho), STC_EUNSUPPORTED,
                "stc_mix_dt_f32: Ks = Kc in 1..3, C in 1..16, L in (20, 32), Ho in (16, 32); got Ks=%d C=%d L=%d Ho=%d", Ks, C, L, Ho);
    const int rpt = (16 / C) * C;                  // rows per tile: floor(16 / C) whole nodes
    STC_REQUIRE(rows >= 0 && rows % C == 0 && rows < (1ll << 31), STC_EINVAL, "stc_mix_dt_f32: %lld rows are not whole nodes of %d categories", (long long)rows, C);
    STC_REQUIRE(Lw >= 1 && Lw <= L, STC_EINVAL, "stc_mix_dt_f32: Lw=%d outside 1..L=%d", Lw, L);
    STC_REQUIRE(dTc, STC_EINVAL, "stc_mix_dt_f32: null dTc");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (rows == 0 || Ks == 1) return stc::hip_status(hipMemsetAsync(dTc, 0, (size_t)Ks * C * C * sizeof(float), s), "memset dTc");
    STC_REQUIRE(Z && W && dY && workspace, STC_EINVAL, "stc_mix_dt_f32: null operand");
    STC_REQUIRE(workspace_bytes >= stc_mix_dt_workspace_bytes(Ks) && stc::aligned16(workspace), STC_EINVAL, "stc_mix_dt_f32: workspace too small or misaligned");
    DtArgs a{};
    for (int n = 0; n < Ks; ++n) {
        STC_REQUIRE(Z[n] && stc::aligned16(Z[n]), STC_EALIGN, "stc_mix_dt_f32: Z[%d] null or not 16-byte aligned", n);
        a.Z[n] = Z[n];
    }
    STC_REQUIRE(stc::aligned16(dY), STC_EALIGN, "stc_mix_dt_f32: dY not 16-byte aligned");
    a.W = W;
    a.dY = dY;
    a.partial = static_cast<float*>(workspace);
    a.tiles = (int)((rows + rpt - 1) / rpt);
    a.Lw = Lw;
    a.rpt = rpt;
    a.rows = (int)rows;
    const int want = (a.tiles + DT_WAVES - 1) / DT_WAVES;
    const int grid = want < DT_MAX_GRID ? want : DT_MAX_GRID;
    int rc = STC_EUNSUPPORTED;
#define DT_CASE(K_, LQ_, OT_) if (Ks == K_ && L == 4 * LQ_ && Ho == 16 * OT_) rc = launch_dt<K_, LQ_, OT_>(a, grid, s);
    DT_CASE(2, 5, 1) DT_CASE(2, 5, 2) DT_CASE(2, 8, 1) DT_CASE(2, 8, 2)
    DT_CASE(3, 5, 1) DT_CASE(3, 5, 2) DT_CASE(3, 8, 1) DT_CASE(3, 8, 2)
#undef DT_CASE
    if (rc != STC_OK) return rc;
    const int n_out = Ks * C * C;
    hipLaunchKernelGGL(mix_dt_reduce_kernel, dim3((n_out + 3) / 4), dim3(256), 0, s, a.partial, grid, Ks, C, dTc);
    STC_LAUNCH_CHECK("stc_mix_dt_f32 reduce launch");
    return STC_OK;
}
