// One STC_Cell step for FEW CATEGORIES (C <= 16; reference STC_GNN.py:65-79 with BDG_Dif :31-47 inside, Ks = Kc = 2, hidden 16), and its
// autograd: the SF-incidents shape (N = 100, C = 5) and its relatives, where the general path is ~15 launches of generic
// vector kernels per cell and direction and the step is bound by the host's launch rate.
//
// The phases of a cell, each reading its neighbours' rows from the previous one:
//   forward   0  stage the graph and the sample's [H | X] rows in LDS
//             1  Zg = S.[H | X]                          (gather over the CSR rows; LP = 16 + 4 XQ columns, zero padded)
//             2  gates: [H|X], Zg -> project (fp32 MFMA) -> category mix -> sigmoid -> U, R, R*H
//             3  Zc = S.(R*H)                            (the X part of S.[X | R*H] is Zg's)
//             4  candidate: [R*H|X], [Zc|Zg.x] -> project -> mix -> tanh -> blend -> Cand, Hnew
//   backward  1  dY = dHnew U (1 - Cand^2) -> mix^T -> dWc, dbc partials; dZc_0, dZc_1
//             2  d[R*H | X] = dZc_0 + S^T dZc_1; gate backward -> dYg, first share of dH, dX
//             3  dYg -> mix^T -> dWg, dbg partials; dZg_0, dZg_1
//             4  dH, dX += dZg_0 + S^T dZg_1
// Two forms of a launch (argument `phase`):
//   * phase 0 -- the whole cell in ONE launch, one workgroup per sample, workgroup barriers between the phases.  What it costs is latency,
//     not bandwidth (first version, everything gathered from L2: 61 / 119 us forward / backward at the SF shape against ~11 / 22 us of
//     matrix-pipe time).  Hence, when the sample fits (MODE 1 / 2): the graph and the gathered planes ([H | X], R*H; in the backward the
//     slab dZ_1) live in LDS, operands that must come from global memory are requested one row tile ahead, 16 waves (forward: a wave owns
//     one 16-column tile of the output; backward: four waves per row tile, one per slab and role), parameter-gradient partials to one row
//     per tile quad of the caller's buffer (no cross-wave reduction).  A dense learned graph (MODE 2) aggregates as a matrix product on the
//     staged plane.
//   * phase 1..4 with `splits` = G -- ONE phase per launch, the sample's row tiles dealt in contiguous ranges over G workgroups (nothing
//     staged; the launch boundary is the barrier): one workgroup per sample keeps only `batch` of the 256 compute units busy, and at batch 32
//     the short launches win for the backward (79 -> 4 x 13 us at the SF shape) and, for samples too large to stage, for the forward too.
//     Phases whose input is the workgroup's OWN rows share a launch: 5 = 1 + 2 and 6 = 3 + 4 (forward), 7 = 2 + 3 (backward, CSR graphs).
// Projections are project-then-mix:  V_kc = sum_ks Z_ks . W[(ks,kc,:)],  Y[(n,c')] = V_0 + sum_{kc>=1} sum_c T_kc[c,c'] V_kc[(n,c)] + b,
// on v_mfma_f32_16x16x4_f32 (fp32 operands and accumulator: an fmaf chain per element, no split format).  A row tile = the C rows of
// floor(16 / C) whole nodes, so the category mix stays inside a tile -- and is itself four matrix instructions on the accumulators (build_mix).
#include <atomic>

#include "stc_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int SF_THREADS = 1024;                                  // forward: 16 waves, <= 128 registers
constexpr int SB_THREADS = 1024, SB_WAVES = SB_THREADS / 64;      // backward: 16 waves = 4 quads, four waves per row tile (slab x role)
constexpr int SC_H = 16, SC_MAXC = 16;
// Chebyshev ORDER 3 (Ks = Kc = 3; round 6).  The third spatial slab is Z_2 = T_2(S) Z with T_2(S) = 2 S^2 - I -- which the reference forms on the
// MATRIX side (cheby_poly, STC_GNN.py:24-29) and so does the host here: for a graph this small 2 S^2 - I is one more sparse matrix (25 entries per
// row on an 8-neighbour grid), handed over as a SECOND CSR graph.  Z_2 is then an aggregation of the same input rows as Z_1, not of Z_1's: no
// phase is added, no dependency crosses workgroups that order 2 does not have, and the split forms (5, 6 / 1, 7, 4) are the same launches.
// What changes: three slabs per convolution (nine W blocks, two category mixes), workgroups of 8 waves forward (the nine W blocks are 72
// registers per lane: 256 instead of 128 available) and 12 waves backward (six per row tile: slab x role), nothing staged in LDS (MODE 0),
// fixed CSR graphs only (a learned dense Gs at order 3 stays on the general path).
template <int KS> struct WgShape {                                     // workgroup shapes per order
#ifndef STC_SC_FWD_THREADS             // (probe builds: tools/gpu_ab.sh sf)
#define STC_SC_FWD_THREADS SF_THREADS
#define STC_SC_BWD_THREADS SB_THREADS
#endif
    static constexpr int FWD_THREADS = KS == 2 ? STC_SC_FWD_THREADS : 512, FWD_WAVES = FWD_THREADS / 64;
    static constexpr int BWD_THREADS = KS == 2 ? STC_SC_BWD_THREADS : 768, BWD_WAVES = BWD_THREADS / 64;
    static constexpr int BWD_GROUPS = BWD_WAVES / (2 * KS);       // row tiles a workgroup convolves at once (4 / 2)
};

// Probe hooks: tools/probes/small_cell_phases.py builds this file with -DSTC_PROBE -DSC_STOP_AFTER=n (-DSC_MAX_TILES, -DSC_SKIP_ROLE) and times
// the truncated launches.  The library is built without STC_PROBE: the hooks then do not exist.
#ifdef STC_PROBE
#ifndef SC_STOP_AFTER
#define SC_STOP_AFTER 99
#endif
#define SC_PHASE_END(n) do { if (SC_STOP_AFTER <= (n)) return; } while (0)
#ifndef SC_MAX_TILES
#define SC_MAX_TILES (1 << 30)                     // the backward convolutions stop after this many row tiles
#endif
#ifndef SC_SKIP_ROLE
#define SC_SKIP_ROLE (-1)                          // the waves of this backward role (0: dZ, 1: dW) do nothing
#endif
#else
#define SC_PHASE_END(n) do { } while (0)
#define SC_MAX_TILES (1 << 30)
#define SC_SKIP_ROLE (-1)
#endif

// Gate nonlinearities on the hardware exp2 / rcp, as in the large-graph cell kernels (stc_common.h: accurate relative to the result for every
// argument): libm expf / tanhf and the IEEE division are ~30 vector instructions each, four per lane and tile, in loops that run at the sum of
// their vector and matrix cycles.
__device__ __forceinline__ float sigm(float v) { return stc_sigmoid(v); }
__device__ __forceinline__ float tanh_hw(float v) { return stc_tanh(v); }
// x / C for x < 65536 and C <= 16 with inv = ceil(2^31 / C), as the high word of the 64-bit product 2x * inv (a shift and one
// v_mul_hi_u32): exact, since x (inv C - 2^31) < 65536 * 16 < 2^31, and nothing wraps (2x inv < 2^49; inv = 2^31 for C = 1 still fits 32
// bits).  A 2^20 reciprocal in a 32-bit product wrapped from x = 4096 C on.  A runtime integer division is ~25 instructions and the gathers
// below would do two per item.
__host__ __device__ constexpr unsigned inv_c(int C) { return (unsigned)((0x80000000ull + (unsigned)C - 1) / (unsigned)C); }
__device__ __forceinline__ int div_c(int x, unsigned inv) { return (int)__umulhi((unsigned)x << 1, inv); }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
// p[idx] where ok, else 0 -- as an UNCONDITIONAL load of a clamped address plus a select.  Written as `ok ? p[idx] : 0` every such load
// becomes its own exec-masked block with a wait at the join: the launch's few dozen guarded operand loads per tile then run one L2 round
// trip after the other.  p[0] must be readable.  Offsets inside a sample are 32-bit (unsigned) everywhere in this file: with size_t
// indices every operand address is a chain of 64-bit VALU operations, and at four waves per SIMD those, not the matrix pipe, set the pace.
__device__ __forceinline__ float ld_if(const float* p, unsigned idx, bool ok) {
    const float v = p[ok ? idx : 0];
    return ok ? v : 0.f;
}

struct SmallGraph {
    const int32_t* rowptr;
    const int32_t* colidx;
    const float* val;
    int nnz;
};

__host__ __device__ constexpr int plane_stride(int xq) { return xq == 4 ? 36 : 28; }      // floats per staged [H | X] row: 16-byte reads of
constexpr int SQ = 20;                                                                     // 16 consecutive rows fall on distinct banks

// out[row][quad] = base + sum_e val[e] * fetch(colidx[e] * C + c, quad)  for the rows of one sample; fetch returns 4 columns of a source row.
template <int THREADS, int QUADS, class Fetch, class Base, class Store>
__device__ __forceinline__ void aggregate_rows(const int* __restrict__ gp, const int* __restrict__ gc, const float* __restrict__ gv, int NC, int C,
                                               unsigned invC, int row_lo, int row_hi, Fetch fetch, Base base, Store store) {
    (void)NC;
    for (int item = threadIdx.x; item < (row_hi - row_lo) * QUADS; item += THREADS) {       // the workgroup's own rows [row_lo, row_hi)
        const int row = row_lo + item / QUADS, q = item - (row - row_lo) * QUADS;
        const int n = div_c(row, invC), c = row - n * C;
        f32x4 s = base(row, q);
        const int e1 = gp[n + 1];
        for (int e = gp[n]; e < e1; ++e) {
            const float v = gv[e];
            const f32x4 x = fetch(gc[e] * C + c, q);
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] = fmaf(v, x[i], s[i]);
        }
        store(row, q, s);
    }
}

// ... over TWO graphs with a source each (order 3's transposed aggregation  d[in] = dZ_0 + S^T dZ_1 + T_2(S)^T dZ_2):
template <int THREADS, int QUADS, class Fetch, class Fetch2, class Base, class Store>
__device__ __forceinline__ void aggregate_rows2(const int* __restrict__ gp, const int* __restrict__ gc, const float* __restrict__ gv,
                                                const int* __restrict__ gp2, const int* __restrict__ gc2, const float* __restrict__ gv2, int C,
                                                unsigned invC, int row_lo, int row_hi, Fetch fetch, Fetch2 fetch2, Base base, Store store) {
    for (int item = threadIdx.x; item < (row_hi - row_lo) * QUADS; item += THREADS) {
        const int row = row_lo + item / QUADS, q = item - (row - row_lo) * QUADS;
        const int n = div_c(row, invC), c = row - n * C;
        f32x4 s = base(row, q);
        const int e1 = gp[n + 1];
        for (int e = gp[n]; e < e1; ++e) {
            const float v = gv[e];
            const f32x4 x = fetch(gc[e] * C + c, q);
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] = fmaf(v, x[i], s[i]);
        }
        const int f1 = gp2[n + 1];
        for (int e = gp2[n]; e < f1; ++e) {
            const float v = gv2[e];
            const f32x4 x = fetch2(gc2[e] * C + c, q);
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] = fmaf(v, x[i], s[i]);
        }
        store(row, q, s);
    }
}

// The same aggregation for a DENSE graph (the reference's learned Gs: the CSR is the full N x N pattern, `S` its values as a row-major matrix)
// as a matrix product on the staged plane: out^T tile = src^T . S^T, i.e. A = 16 columns of the source rows (LDS), B = 16 nodes' rows of S
// (one 16-byte global load = the lane's B operands of four steps), so that a lane ends up with 4 consecutive COLUMNS of one output row --
// the quad the CSR form's base / store callbacks take.  (Gathering 100 neighbour rows per output row from L2 instead costs ~100 us per phase.)
template <int THREADS, int QUADS, class Base, class Store>
__device__ __forceinline__ void aggregate_dense(const float* __restrict__ S, int N, int C, const float* src, int stride, int split, int splits, Base base,
                                                Store store) {
    constexpr int CT = (QUADS + 3) / 4;                          // 16-column tiles per (node, category) row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, kq = lane >> 4;
    const int rtiles = (N + 15) >> 4, per_rt = C * CT, pairs = (per_rt + 1) >> 1, items = rtiles * pairs;
    const bool vec = (N & 3) == 0;
    // a wave takes TWO column tiles of one row tile at a time: they share the B operand (the rows of S) and give the matrix pipe two
    // independent accumulators (one accumulator = a chain of dependent instructions, 40 cycles each instead of 32)
    for (int item = wave + (THREADS / 64) * split; item < items; item += (THREADS / 64) * splits) {
        const int rt = item / pairs, t0 = 2 * (item - rt * pairs), t1 = min(t0 + 1, per_rt - 1);
        const bool two = t0 + 1 < per_rt;
        const int c0 = t0 / CT, lb0 = t0 - c0 * CT, c1 = t1 / CT, lb1 = t1 - c1 * CT;
        const int node = 16 * rt + j;
        const bool node_ok = node < N;
        const unsigned srow = (unsigned)(node_ok ? node : 0) * N;
        // the source column this lane feeds as A operand (clamped: extra columns are not stored)
        const unsigned a0 = (unsigned)c0 * stride + min(16 * lb0 + j, 4 * QUADS - 1), a1 = (unsigned)c1 * stride + min(16 * lb1 + j, 4 * QUADS - 1);
        auto load_s = [&](int kb) {                              // S[node][16 kb + 4 kq .. + 3], unmasked (clamped); the mask is applied when it is consumed
            const int k0 = 16 * kb + 4 * kq;
            if (vec) return ld4(S + (k0 < N ? srow + k0 : srow));
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = S[srow + min(k0 + i, N - 1)];
            return v;
        };
        f32x4 acc0 = zero4(), acc1 = zero4(), bn = load_s(0);
        for (int kb = 0; kb < rtiles; ++kb) {
            f32x4 b = bn;
            if (kb + 1 < rtiles) bn = load_s(kb + 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = 16 * kb + 4 * kq + i;
                const float bv = node_ok && m < N ? b[i] : 0.f;
                const unsigned mrow = (unsigned)(min(m, N - 1) * C) * stride;
                acc0 = mfma4(src[mrow + a0], bv, acc0);
                acc1 = mfma4(src[mrow + a1], bv, acc1);
            }
        }
        if (node_ok) {
            const int q0 = 4 * lb0 + kq, q1 = 4 * lb1 + kq;
            if (q0 < QUADS) {
                const int row = node * C + c0;
                f32x4 s = base(row, q0);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] += acc0[i];
                store(row, q0, s);
            }
            if (two && q1 < QUADS) {
                const int row = node * C + c1;
                f32x4 s = base(row, q1);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] += acc1[i];
                store(row, q1, s);
            }
        }
    }
}

// The dense aggregation for the rows a workgroup OWNS (split form, nothing staged): out[node][c][0 .. 4 QUADS) = sum_m S[node][m] src[m][c][.] for the
// node tiles that cover the nodes [n_lo, n_hi), the source rows read from global memory (src: rows of `stride` floats per (node, category), `ncols`
// of them meaningful -- columns beyond read as zero), store(row, quad, base(row, quad) + sum) for the OWN rows [row_lo, row_hi) only.  Neighbouring
// workgroups share a boundary node tile and both compute it (what they do not own they drop), so a store callback may accumulate in place and a
// phase that needs the aggregate of its own rows only can follow in the same launch.
template <int THREADS, int QUADS, class Base, class Store>
__device__ __forceinline__ void aggregate_dense_own(const float* __restrict__ S, int N, int C, const float* src, int stride, int ncols, int n_lo, int n_hi,
                                                    int row_lo, int row_hi, Base base, Store store) {
    constexpr int CT = (QUADS + 3) / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, kq = lane >> 4;
    const int rtiles = (N + 15) >> 4, per_rt = C * CT, pairs = (per_rt + 1) >> 1;
    const int rt_lo = n_lo >> 4, rt_hi = (n_hi + 15) >> 4, items = (rt_hi - rt_lo) * pairs;
    const bool vec = (N & 3) == 0;
    for (int item = wave; item < items; item += THREADS / 64) {
        const int rt = rt_lo + item / pairs, t0 = 2 * (item - (rt - rt_lo) * pairs), t1 = min(t0 + 1, per_rt - 1);
        const bool two = t0 + 1 < per_rt;
        const int c0 = t0 / CT, lb0 = t0 - c0 * CT, c1 = t1 / CT, lb1 = t1 - c1 * CT;
        const int node = 16 * rt + j;
        const bool node_ok = node < N;
        const unsigned srow = (unsigned)(node_ok ? node : 0) * N;
        const int col0 = 16 * lb0 + j, col1 = 16 * lb1 + j;
        const bool ok0 = col0 < ncols, ok1 = col1 < ncols;
        const unsigned a0 = (unsigned)c0 * stride + (ok0 ? col0 : 0), a1 = (unsigned)c1 * stride + (ok1 ? col1 : 0);
        auto load_s = [&](int kb) {
            const int k0 = 16 * kb + 4 * kq;
            if (vec) return ld4(S + (k0 < N ? srow + k0 : srow));
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = S[srow + min(k0 + i, N - 1)];
            return v;
        };
        auto load_a = [&](int kb, float (&x0)[4], float (&x1)[4]) {         // the lane's source values of four steps (clamped rows; masked where consumed)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned mrow = (unsigned)(min(16 * kb + 4 * kq + i, N - 1) * C) * stride;
                x0[i] = src[mrow + a0];
                x1[i] = src[mrow + a1];
            }
        };
        f32x4 acc0 = zero4(), acc1 = zero4(), bn = load_s(0);
        float xn0[4], xn1[4];
        load_a(0, xn0, xn1);
        for (int kb = 0; kb < rtiles; ++kb) {
            const f32x4 b = bn;
            float x0[4], x1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { x0[i] = xn0[i]; x1[i] = xn1[i]; }
            if (kb + 1 < rtiles) {                                       // the next block's operands are requested before this block's products
                bn = load_s(kb + 1);
                load_a(kb + 1, xn0, xn1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = 16 * kb + 4 * kq + i;
                const float bv = node_ok && m < N ? b[i] : 0.f;
                acc0 = mfma4(ok0 ? x0[i] : 0.f, bv, acc0);
                acc1 = mfma4(ok1 ? x1[i] : 0.f, bv, acc1);
            }
        }
        if (node_ok) {
            const int q0 = 4 * lb0 + kq, q1 = 4 * lb1 + kq, r0 = node * C + c0, r1 = node * C + c1;
            if (q0 < QUADS && r0 >= row_lo && r0 < row_hi) {
                f32x4 s = base(r0, q0);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] += acc0[i];
                store(r0, q0, s);
            }
            if (two && q1 < QUADS && r1 >= row_lo && r1 < row_hi) {
                f32x4 s = base(r1, q1);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] += acc1[i];
                store(r1, q1, s);
            }
        }
    }
}

template <int THREADS>
__device__ __forceinline__ void stage_graph(const SmallGraph& g, int N, int* gp, int* gc, float* gv) {
    for (int i = threadIdx.x; i <= N; i += THREADS) gp[i] = g.rowptr[i];
    for (int i = threadIdx.x; i < g.nnz; i += THREADS) {
        gc[i] = g.colidx[i];
        gv[i] = g.val[i];
    }
}

// A operands of one lane for one slab of a row tile: the H block (4 steps from one 16-byte read) and the X block.
template <int XS>
struct AOp {
    f32x4 h;
    float x[XS];
};
template <int XS>
__device__ __forceinline__ AOp<XS> zero_op() {
    AOp<XS> a;
    a.h = zero4();
#pragma unroll
    for (int s = 0; s < XS; ++s) a.x[s] = 0.f;
    return a;
}
// lane (row, kq) of a slab stored as [h-part pointer, row stride ldh | x-part pointer, row stride ldx]; narrow X blocks are zero padded
// to 4 columns where `padded`, else bounded by cin.
template <int XQ>
__device__ __forceinline__ AOp<(XQ == 4 ? 4 : XQ)> load_op(const float* ph, int ldh, const float* px, int ldx, bool padded, int cin, int row, int kq) {
    constexpr int XS = XQ == 4 ? 4 : XQ;
    AOp<XS> a;
    a.h = ld4(ph + (unsigned)row * ldh + 4 * kq);
    if (XQ == 4) {
        const f32x4 v = ld4(px + (unsigned)row * ldx + 4 * kq);
#pragma unroll
        for (int s = 0; s < XS; ++s) a.x[s] = v[s];
    } else {
#pragma unroll
        for (int s = 0; s < XS; ++s) a.x[s] = ld_if(px, (unsigned)row * ldx + 4 * s + kq, padded || 4 * s + kq < cin);
    }
    return a;
}

// W (Ks*Kc*L, HO) in B-operand order for the forward products of output column `col`: step s < 4 of slab ks feeds l = cin + 4 kq + s (the
// H block: one 16-byte read of a row gives a lane its A operands of four steps; any bijection of the contraction index serves a sum),
// steps 4.. the X block (wide: l = 4 kq + s; narrow: l = 4 s + kq).
template <int KS, int KC, int XQ>
__device__ __forceinline__ void load_w_fwd(float (&Wr)[KS][KC][4 + (XQ == 4 ? 4 : XQ)], const float* __restrict__ W, int HO, int col, int cin, int kq) {
    constexpr int XS = XQ == 4 ? 4 : XQ;
    const int L = cin + SC_H;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const float* Wb = W + (unsigned)((ks * KC + kc) * L) * HO + col;
#pragma unroll
            for (int s = 0; s < 4; ++s) Wr[ks][kc][s] = Wb[(unsigned)(cin + 4 * kq + s) * HO];
#pragma unroll
            for (int s = 0; s < XS; ++s) {
                const int l = XQ == 4 ? 4 * kq + s : 4 * s + kq;
                Wr[ks][kc][4 + s] = ld_if(Wb, (unsigned)l * HO, l < cin);
            }
        }
}

// The category mix of a row tile as a matrix product on the tile itself: a tile holds whole nodes, so mixing the C rows of each node is
//   out = M . V,   M[i][k] = T[c(k)][c(i)] (forward;  T[c(i)][c(k)] for the transposed mix of the backward) when rows i, k belong to the same
// node, else 0 -- a 16 x 16 x 16 product = four v_mfma_f32_16x16x4_f32 whose B operands ARE the accumulator registers of V (contraction index
// of step s: 4 kq + s = the row that register s of lane (., kq) holds).  M's A operands are four registers per lane, built once per launch.
// (The first version mixed with a VALU loop over LDS: C dependent LDS reads per output element, ~1 us per tile, most of the phase.)
template <int KC>
__device__ __forceinline__ void build_mix(float (&M)[KC][4], const float* __restrict__ Tc, bool transposed, int rpt, int C, unsigned invC, int j, int kq) {
    const int ni = div_c(j, invC), ci = j - ni * C;
#pragma unroll
    for (int kc = 1; kc < KC; ++kc)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 4 * kq + s, nk = div_c(k, invC), ck = k - nk * C;
            const bool same = j < rpt && k < rpt && ni == nk;
            M[kc][s] = same ? (transposed ? Tc[(kc * C + ci) * C + ck] : Tc[(kc * C + ck) * C + ci]) : 0.f;
        }
#pragma unroll
    for (int s = 0; s < 4; ++s) M[0][s] = 0.f;                    // (T_0 = I: never multiplied)
}

// The row tiles tile0, tile0 + tstep, .. of a forward convolution for ONE 16-column tile of its output: V_kc in the accumulators (lane
// (j, kq): rows 4 kq .. 4 kq + 3 of the tile, column j), the category mix as four more matrix instructions per kc >= 1 into V_0's
// accumulator, epi(global row, y) per finished pre-activation (bias not added).  z0(row) / z1(row): the lane's A operands of slab 0 /
// slab 1 (z2: slab 2, order 3 only); the aggregated slabs (which the previous phase left in global memory) are requested one tile ahead.
template <int KS, int KC, int XQ, class Z0, class Z1, class Z2, class Epi>
__device__ __forceinline__ void fwd_conv(const float (&Wr)[KS][KC][4 + (XQ == 4 ? 4 : XQ)], const float (&M)[KC][4], int tile0, int tstep, int tiles,
                                         int rpt, int NC, int j, int kq, Z0 z0, Z1 z1, Z2 z2, Epi epi) {
    constexpr int XS = XQ == 4 ? 4 : XQ;
    // rows that do not exist (beyond the tile's nodes / the sample) read the tile's first row -- a row this workgroup owns, so a FINITE value
    // even when other workgroups are still writing theirs (fused phases): their products land in accumulator rows nobody stores, but the
    // mix multiplies them by zeros.  (A tile requested past the end reads an in-bounds row that nobody uses.)
    auto row_of = [&](int tile) { return j < rpt && tile * rpt + j < NC ? tile * rpt + j : min(tile * rpt, NC - 1); };
    AOp<XS> nxt = z1(row_of(tile0)), nxt2 = zero_op<XS>();
    if constexpr (KS == 3) nxt2 = z2(row_of(tile0));
    for (int tile = tile0; tile < tiles; tile += tstep) {
        const int row0 = tile * rpt;
        AOp<XS> a[KS];
        a[1] = nxt;
        nxt = z1(row_of(tile + tstep));
        if constexpr (KS == 3) {
            a[2] = nxt2;
            nxt2 = z2(row_of(tile + tstep));
        }
        a[0] = z0(row_of(tile));
        f32x4 acc[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) acc[kc] = zero4();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) acc[kc] = mfma4(a[ks].h[s], Wr[ks][kc][s], acc[kc]);
#pragma unroll
            for (int s = 0; s < XS; ++s)
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) acc[kc] = mfma4(a[ks].x[s], Wr[ks][kc][4 + s], acc[kc]);
        }
#pragma unroll
        for (int kc = 1; kc < KC; ++kc)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[0] = mfma4(M[kc][s], acc[kc][s], acc[0]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int lrow = 4 * kq + r, grow = row0 + lrow;
            if (lrow < rpt && grow < NC) epi(grow, acc[0][r]);
        }
    }
}

struct SmallFwd {
    SmallGraph g;
    const float *X, *H, *Tc, *Wg, *bg, *Wc, *bc;
    float *U, *R, *Cand, *Hnew, *RH, *Zg, *Zc;
    float *Z0, *Z0c, *Z1c;                       // optional (learned graphs: operands of the graph-gradient products): the slabs [H | X | 0],
                                                 // [R*H | X | 0] and [S.(R*H) | S.X | 0]
    int N, C, cin, rpt, tiles;
    int phase;                                   // 0: the whole cell in this launch (one workgroup per sample); 1..4: that phase only, the sample's
                                                 // rows split over gridDim.y workgroups -- the launch boundary is the barrier between phases
    SmallGraph g2;                               // order 3: CSR of T_2(S)^T = 2 (S^T)^2 - I in the forward's orientation
    float *Zg2, *Zc2;                            // order 3: T_2 . [H | X] (rows of LP floats) and T_2 . (R*H) (rows of 16)
};

template <int KC>
__host__ __device__ constexpr int fwd_lds_fixed() { return 0; }                                         // floats of LDS every launch needs

// MODE 0: graph and planes read from global memory; 1: CSR graph + planes staged in LDS; 2: dense graph (matrix-product aggregation), planes staged;
// 3: dense graph in the split form (phase != 0): nothing staged, a workgroup aggregates the node tiles that cover its own rows (aggregate_dense_own)
template <int KS, int KC, int XQ, int MODE>
__global__ __launch_bounds__(WgShape<KS>::FWD_THREADS) void small_fwd_kernel(SmallFwd a) {
    constexpr int LP = 16 + 4 * XQ, XS = XQ == 4 ? 4 : XQ, SP = plane_stride(XQ);
    constexpr bool STAGED = MODE == 1 || MODE == 2, DENSE = MODE >= 2;
    constexpr int SF_THREADS = WgShape<KS>::FWD_THREADS, SF_WAVES = WgShape<KS>::FWD_WAVES;      // (shadow the order-2 constants of the file)
    static_assert(KS == 2 || (KS == 3 && KC == 3 && MODE == 0), "order 3: Ks = Kc = 3, CSR graph, nothing staged");
    extern __shared__ __align__(16) float lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, j = lane & 15, kq = lane >> 4;
    const int C = a.C, N = a.N, NC = N * C, cin = a.cin;
    const unsigned invC = inv_c(C);
    float* P = lds + fwd_lds_fixed<KC>();                   // STAGED: [H | X | 0] rows of the sample, stride SP
    float* Q = P + (size_t)NC * SP;                         //         R*H rows, stride SQ
    int* gpl = reinterpret_cast<int*>(Q + (size_t)NC * SQ);
    int* gcl = gpl + ((N + 4) & ~3);
    float* gvl = reinterpret_cast<float*>(gcl + ((a.g.nnz + 3) & ~3));
    const int* gp = STAGED && !DENSE ? gpl : a.g.rowptr;
    const int* gc = STAGED && !DENSE ? gcl : a.g.colidx;
    const float* gv = STAGED && !DENSE ? gvl : a.g.val;
    const size_t r0 = (size_t)blockIdx.x * NC;
    const float* Xb = a.X + r0 * cin;
    const float* Hb = a.H + r0 * SC_H;
    float* Ub = a.U + r0 * SC_H;
    float* Rb = a.R + r0 * SC_H;
    float* RHb = a.RH + r0 * SC_H;           // (written in phase 2, read later in the launch: never through a __restrict__ / const path)
    float* Zgb = a.Zg + r0 * LP;
    float* Zcb = a.Zc + r0 * SC_H;
    float* Zg2b = KS == 3 ? a.Zg2 + r0 * LP : nullptr;
    float* Zc2b = KS == 3 ? a.Zc2 + r0 * SC_H : nullptr;

    // Split form (phase != 0): this launch runs ONE phase, with the sample's tiles / rows dealt over gridDim.y workgroups, so that a step of
    // few samples still fills the chip; the caller launches the phases in order (the launch boundary replaces the workgroup barrier).
    // phase 5 = phases 1 + 2, phase 6 = phases 3 + 4 in one launch: a workgroup OWNS a contiguous range of row tiles, and the gates /
    // candidate convolution of a tile needs the aggregate of its own rows only (the aggregation reads the neighbours' INPUT rows, complete
    // before the launch), so only the step from phase 2 to phase 3 -- R*H of the neighbours -- needs a launch boundary.
    const int phase = a.phase, split = blockIdx.y, splits = gridDim.y;
    auto runs = [&](int p) { return phase == 0 || phase == p || (phase == 5 && p <= 2) || (phase == 6 && p >= 3); };
    auto sync = [&](int after) { if (phase == 0 || (phase == 5 && after == 1) || (phase == 6 && after == 3)) __syncthreads(); };
    const int t_lo = (int)((long long)a.tiles * split / splits), t_hi = (int)((long long)a.tiles * (split + 1) / splits);
    const int row_lo = min(t_lo * a.rpt, NC), row_hi = min(t_hi * a.rpt, NC);
    const int n_lo = div_c(row_lo, invC), n_hi = div_c(row_hi + C - 1, invC);      // the nodes of the workgroup's own rows (whole nodes: rpt = a multiple of C)

    // 0: tables, graph and the sample's rows into LDS (the gates' W operands are requested first: in flight during phases 0 and 1)
    const int ct = wave & 1;                                 // gates: wave w owns column tile w % 2 (0: update, 1: reset)
    float Wg_r[KS][KC][4 + XS];
    if (runs(2)) load_w_fwd<KS, KC, XQ>(Wg_r, a.Wg, 2 * SC_H, 16 * ct + j, cin, kq);
    float M[KC][4];
    build_mix<KC>(M, a.Tc, false, a.rpt, C, invC, j, kq);
    if (STAGED) {
        if (!DENSE) stage_graph<SF_THREADS>(a.g, N, gpl, gcl, gvl);
        for (int item = t; item < NC * 4; item += SF_THREADS) {
            const int row = item >> 2, q = item & 3;
            st4(P + (unsigned)row * SP + 4 * q, ld4(Hb + (unsigned)row * SC_H + 4 * q));
            if (XQ == 4) st4(P + (unsigned)row * SP + 16 + 4 * q, ld4(Xb + (unsigned)row * SC_H + 4 * q));
        }
        if (XQ != 4)
            for (int item = t; item < NC * XQ; item += SF_THREADS) {
                const int row = item / XQ, q = item - row * XQ;
                f32x4 x = zero4();
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (4 * q + i < cin) x[i] = Xb[(unsigned)row * cin + 4 * q + i];
                st4(P + (unsigned)row * SP + 16 + 4 * q, x);
            }
    }
    if (phase == 0) __syncthreads();

    if (a.Z0 != nullptr && runs(1)) {                        // (learned graphs only)
        float* Z0b = a.Z0 + r0 * LP;
        for (int item = t; item < (row_hi - row_lo) * (LP / 4); item += SF_THREADS) {
            const int row = row_lo + item / (LP / 4), q = item - (row - row_lo) * (LP / 4);
            f32x4 x;
            if (STAGED) {
                x = ld4(P + (unsigned)row * SP + 4 * q);
            } else if (q < 4) {
                x = ld4(Hb + (unsigned)row * SC_H + 4 * q);
            } else if (XQ == 4) {
                x = ld4(Xb + (unsigned)row * SC_H + 4 * (q - 4));
            } else {
                x = zero4();
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (4 * (q - 4) + i < cin) x[i] = Xb[(unsigned)row * cin + 4 * (q - 4) + i];
            }
            st4(Z0b + (unsigned)row * LP + 4 * q, x);
        }
    }

    // 1: Zg = S.[H | X]
    if (runs(1)) {
        auto none = [](int, int) -> f32x4 { return zero4(); };
        auto put = [&](int row, int q, f32x4 s) { st4(Zgb + (unsigned)row * LP + 4 * q, s); };
        if (DENSE && STAGED) {
            aggregate_dense<SF_THREADS, LP / 4>(a.g.val, N, C, P, SP, 0, 1, none, put);
        } else if (DENSE) {                                  // the H block and the X block of the slab from their own planes
            aggregate_dense_own<SF_THREADS, 4>(a.g.val, N, C, Hb, SC_H, SC_H, n_lo, n_hi, row_lo, row_hi, none, put);
            aggregate_dense_own<SF_THREADS, XQ>(a.g.val, N, C, Xb, cin, cin, n_lo, n_hi, row_lo, row_hi, none,
                                                [&](int row, int q, f32x4 s) { st4(Zgb + (unsigned)row * LP + SC_H + 4 * q, s); });
        } else {
            auto input_rows = [&](int src, int q) -> f32x4 {
                    if (STAGED) return ld4(P + (unsigned)src * SP + 4 * q);
                    if (q < 4) return ld4(Hb + (unsigned)src * SC_H + 4 * q);
                    if (XQ == 4) return ld4(Xb + (unsigned)src * SC_H + 4 * (q - 4));
                    f32x4 x = zero4();
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (4 * (q - 4) + i < cin) x[i] = Xb[(unsigned)src * cin + 4 * (q - 4) + i];
                    return x;
                };
            aggregate_rows<SF_THREADS, LP / 4>(gp, gc, gv, NC, C, invC, row_lo, row_hi, input_rows, none, put);
            if constexpr (KS == 3)                           // Zg2 = T_2(S) . [H | X]: the same input rows through the second graph
                aggregate_rows<SF_THREADS, LP / 4>(a.g2.rowptr, a.g2.colidx, a.g2.val, NC, C, invC, row_lo, row_hi, input_rows, none,
                                                   [&](int row, int q, f32x4 s) { st4(Zg2b + (unsigned)row * LP + 4 * q, s); });
        }
    }
    sync(1);
    SC_PHASE_END(1);

    // 2: gates -- wave w: column tile w % 2 of the workgroup's row tiles t_lo + w / 2, t_lo + w / 2 + 8, ..
    if (runs(2)) {
        const float bias = a.bg ? a.bg[16 * ct + j] : 0.f;
        fwd_conv<KS, KC, XQ>(Wg_r, M, t_lo + (wave >> 1), SF_WAVES / 2, t_hi, a.rpt, NC, j, kq,
            [&](int row) { return STAGED ? load_op<XQ>(P, SP, P + 16, SP, true, cin, row, kq) : load_op<XQ>(Hb, SC_H, Xb, cin, false, cin, row, kq); },
            [&](int row) { return load_op<XQ>(Zgb, LP, Zgb + 16, LP, true, cin, row, kq); },
            [&](int row) { return load_op<XQ>(Zg2b, LP, Zg2b + 16, LP, true, cin, row, kq); },
            [&](int grow, float y) {
                const size_t e = (unsigned)grow * SC_H + j;
                const float g = sigm(y + bias);
                if (ct == 0) {
                    Ub[e] = g;
                } else {
                    const float rh = g * (STAGED ? P[(unsigned)grow * SP + j] : Hb[e]);
                    Rb[e] = g;
                    RHb[e] = rh;
                    if (STAGED) Q[(unsigned)grow * SQ + j] = rh;
                }
            });
    }
    sync(2);
    SC_PHASE_END(2);

    // 3: Zc = S.(R*H)   (the candidate's W operands in flight meanwhile)
    float Wc_r[KS][KC][4 + XS];
    if (runs(4)) load_w_fwd<KS, KC, XQ>(Wc_r, a.Wc, SC_H, j, cin, kq);
    if (runs(3)) {
        auto none = [](int, int) -> f32x4 { return zero4(); };
        auto put = [&](int row, int q, f32x4 s) { st4(Zcb + (unsigned)row * SC_H + 4 * q, s); };
        if (DENSE && STAGED)
            aggregate_dense<SF_THREADS, 4>(a.g.val, N, C, Q, SQ, 0, 1, none, put);
        else if (DENSE)
            aggregate_dense_own<SF_THREADS, 4>(a.g.val, N, C, RHb, SC_H, SC_H, n_lo, n_hi, row_lo, row_hi, none, put);
        else {
            auto rh_rows = [&](int src, int q) -> f32x4 { return STAGED ? ld4(Q + (unsigned)src * SQ + 4 * q) : ld4(RHb + (unsigned)src * SC_H + 4 * q); };
            aggregate_rows<SF_THREADS, 4>(gp, gc, gv, NC, C, invC, row_lo, row_hi, rh_rows, none, put);
            if constexpr (KS == 3)                           // Zc2 = T_2(S) . (R*H)
                aggregate_rows<SF_THREADS, 4>(a.g2.rowptr, a.g2.colidx, a.g2.val, NC, C, invC, row_lo, row_hi, rh_rows, none,
                                              [&](int row, int q, f32x4 s) { st4(Zc2b + (unsigned)row * SC_H + 4 * q, s); });
        }
    }
    sync(3);
    SC_PHASE_END(3);

    if (a.Z0c != nullptr && a.Z0 != nullptr && runs(4)) {    // (learned graphs only; R*H, Zc, Zg and Z0 are complete: two barriers / launches ago)
        const float* Z0b = a.Z0 + r0 * LP;
        float* Z0cb = a.Z0c + r0 * LP;
        float* Z1cb = a.Z1c + r0 * LP;
        for (int item = t; item < (row_hi - row_lo) * (LP / 4); item += SF_THREADS) {
            const int row = row_lo + item / (LP / 4), q = item - (row - row_lo) * (LP / 4);
            st4(Z0cb + (unsigned)row * LP + 4 * q, q < 4 ? ld4(RHb + (unsigned)row * SC_H + 4 * q) : ld4(Z0b + (unsigned)row * LP + 4 * q));
            st4(Z1cb + (unsigned)row * LP + 4 * q, q < 4 ? ld4(Zcb + (unsigned)row * SC_H + 4 * q) : ld4(Zgb + (unsigned)row * LP + 4 * q));
        }
    }

    // 4: candidate + blend
    if (runs(4)) {
        const float bias = a.bc ? a.bc[j] : 0.f;
        float* Cb = a.Cand + r0 * SC_H;
        float* Hn = a.Hnew + r0 * SC_H;
        fwd_conv<KS, KC, XQ>(Wc_r, M, t_lo + wave, SF_WAVES, t_hi, a.rpt, NC, j, kq,
            [&](int row) { return STAGED ? load_op<XQ>(Q, SQ, P + 16, SP, true, cin, row, kq) : load_op<XQ>(RHb, SC_H, Xb, cin, false, cin, row, kq); },
            [&](int row) { return load_op<XQ>(Zcb, SC_H, Zgb + 16, LP, true, cin, row, kq); },
            [&](int row) { return load_op<XQ>(Zc2b, SC_H, Zg2b + 16, LP, true, cin, row, kq); },
            [&](int grow, float y) {
                const size_t e = (unsigned)grow * SC_H + j;
                const float cd = tanh_hw(y + bias), u = Ub[e], hh = STAGED ? P[(unsigned)grow * SP + j] : Hb[e];
                Cb[e] = cd;
                Hn[e] = (1.f - u) * hh + u * cd;
            });
    }
}

// ------------------------------------------------------------------------------------------------------------------------ backward
// 16 waves: FOUR waves share a row tile, one per (slab ks, role): role 0 forms dZ_ks = dV . W[(ks, :, :)]^T (32 registers of W^T), role 1
// accumulates the slab's share of dW (32 accumulator registers) and, for slab 0, db.  All four form the mixed gradients dV_kc themselves,
// each in its own LDS tile.  Per wave that is ~80 registers, so four waves per SIMD hide each other's LDS / global latencies (8 waves
// carrying both slabs and both roles: 53 us for the gates convolution at the SF shape against 15 us of matrix-pipe time), and the
// two roles run the same number of matrix instructions per tile.
// W in B-operand order for dZ_ks: step st feeds the contraction index (kc, o) = (st / (HO/4), 4 (st % (HO/4)) + kq); output column tile
// lt = 0: the H block (l = cin + j), lt = 1: the X block (l = j < cin).
template <int KC, int OT>
__device__ __forceinline__ void load_w_bwd(float (&WT)[2][KC * 4 * OT], const float* __restrict__ W, int ks, int cin, int j, int kq) {
    constexpr int HO = 16 * OT, SPK = HO / 4;
    const int L = cin + SC_H;
#pragma unroll
    for (int lt = 0; lt < 2; ++lt) {
        const int l = lt == 0 ? cin + j : j;
        const bool ok = lt == 0 || j < cin;
#pragma unroll
        for (int st = 0; st < KC * SPK; ++st) {
            const int kc = st / SPK, o = 4 * (st % SPK) + kq;
            WT[lt][st] = ld_if(W, (unsigned)((ks * KC + kc) * L + l) * HO + o, ok);
        }
    }
}

struct Slab {                                    // one slab of a convolution's input as the backward reads it (global memory):
    const float* Ph;                             // [Ph (16 columns, row stride ldh) | Px (cin columns, row stride ldx)]
    int ldh;
    const float* Px;
    int ldx;
};

// Backward of one convolution over the row tiles of a sample, for the slab ks = wave % 2 and the role (wave / 2) % 2.  load_dy(global
// row, column) -> the raw operands of one element of dY per column tile (requested one tile ahead), form_dy(raw, column tile) -> dY.  A lane
// holds dY in ACCUMULATOR layout (rows 4 kq + s, column j), so the transposed category mix dV_kc = M . dY is four matrix instructions per
// kc >= 1 (build_mix) and dY / dV_kc are, as they stand in registers, the B operands of the dW products (contraction = the tile's rows in
// the order 4 kq + s): role 1 touches no LDS at all.  Role 0 needs dV_kc with rows on lanes (A operands of dZ): one pass through the
// wave's LDS tile dv[kc][16][HO + 1].  dZ_ks goes to `dZ` (slab 1: LDS when staged); the dW / db partial sums stay in registers across the
// wave's tiles and are added to the row of the parameter-gradient partials that the tile's four waves share (disjoint parts).
template <int KC, int XQ, int OT, class Raw, class LoadDy, class FormDy>
__device__ __forceinline__ void conv_bwd_phase(const Slab& z, const float* __restrict__ W, int ks, int role, const float (&M)[KC][4],
                                               float* __restrict__ dv, float* dZ, float* __restrict__ dump, float* __restrict__ dW, float* __restrict__ db,
                                               int rpt, int tiles, int tile0, int tstep, int NC, int cin, LoadDy load_dy, FormDy form_dy) {
    constexpr int LP = 16 + 4 * XQ, HO = 16 * OT, DS = HO + 1, SPK = HO / 4;
    const int t = threadIdx.x, lane = t & 63, j = lane & 15, kq = lane >> 4, L = cin + SC_H;
    const int quad = tile0, QUADS = tstep;                       // the wave's first tile and its stride (one launch per cell: t / 256 and 4)
    // Rows 4 kq + s of a tile as a lane addresses them: clamped into the sample (rows of a tile beyond its nodes or beyond the sample, and
    // whole tiles requested past the end, read the last row or a neighbouring tile's: values the mask replaces by zero) -- one v_min per row and tile instead of a
    // compare / select pair per LOAD: the vector unit, not the matrix pipe, paces these loops (a matrix instruction holds the issue port).
    struct Rows {
        unsigned r[4];
    };
    auto rows_of = [&](int tile) {
        Rows w;
#pragma unroll
        for (int s = 0; s < 4; ++s) w.r[s] = min((unsigned)(tile * rpt + 4 * kq + s), (unsigned)(NC - 1));
        return w;
    };
    bool in_tile[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) in_tile[s] = 4 * kq + s < rpt;
    auto mask_of = [&](int tile, int s) { return in_tile[s] && tile * rpt + 4 * kq + s < NC; };
    auto request = [&](const Rows& w, Raw (&raw)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) raw[s] = load_dy(w.r[s], j);
    };
    // dY (rows 4 kq + s, column 16 ot + j) and the mixed gradients dV_kc = M . dY of a tile, both in accumulator layout
    auto form_and_mix = [&](int tile, const Raw (&cur)[4], float (&dy)[OT][4], f32x4 (&dvk)[KC][OT]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool m = mask_of(tile, s);           // (a select, not a product: the row read instead may be one another workgroup is still
#pragma unroll                                         //  writing -- phase 7 -- i.e. anything, NaN included)
            for (int ot = 0; ot < OT; ++ot) dy[ot][s] = m ? form_dy(cur[s], ot) : 0.f;
        }
#pragma unroll
        for (int kc = 1; kc < KC; ++kc)
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                dvk[kc][ot] = zero4();
#pragma unroll
                for (int s = 0; s < 4; ++s) dvk[kc][ot] = mfma4(M[kc][s], dy[ot][s], dvk[kc][ot]);
            }
    };
    if (role == SC_SKIP_ROLE) return;
    Raw nxt[4];
    request(rows_of(quad), nxt);

    if (role == 0) {
        float WT[2][KC * SPK];
        load_w_bwd<KC, OT>(WT, W, ks, cin, j, kq);
        for (int tile = quad; tile < tiles; tile += QUADS) {
            const int row0 = tile * rpt;
            Raw cur[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) cur[s] = nxt[s];
            request(rows_of(tile + QUADS), nxt);
            float dy[OT][4];
            f32x4 dvk[KC][OT];
            form_and_mix(tile, cur, dy, dvk);
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int ot = 0; ot < OT; ++ot)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dv[(kc * 16 + 4 * kq + r) * DS + 16 * ot + j] = kc == 0 ? dy[ot][r] : dvk[kc][ot][r];
            __builtin_amdgcn_wave_barrier();
            f32x4 dz[2] = {zero4(), zero4()};
#pragma unroll
            for (int st = 0; st < KC * SPK; ++st) {
                const int kc = st / SPK, o = 4 * (st % SPK) + kq;
                const float av = dv[(kc * 16 + j) * DS + o];
#pragma unroll
                for (int lt = 0; lt < 2; ++lt) dz[lt] = mfma4(av, WT[lt][st], dz[lt]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (mask_of(tile, r)) {
                    const unsigned e = (unsigned)(row0 + 4 * kq + r) * LP + j;
                    dZ[e] = dz[0][r];
                    if (j < LP - 16) dZ[e + 16] = dz[1][r];
                    if (dump != nullptr) {                       // (slab 1 of a learned graph: kept for the graph-gradient product)
                        dump[e] = dz[0][r];
                        if (j < LP - 16) dump[e + 16] = dz[1][r];
                    }
                }
            __builtin_amdgcn_wave_barrier();    // the next tile overwrites dv
        }
        return;
    }
    // role 1: dW[(ks, kc, l), o] += sum_rows Z_ks[row, l] dV_kc[row, o] -- the rows are the contraction (4 steps: rows 4 kq + st)
    f32x4 dw[2][KC][OT];
    float dbv[OT];
#pragma unroll
    for (int lt = 0; lt < 2; ++lt)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) dw[lt][kc][ot] = zero4();
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) dbv[ot] = 0.f;
    // the A operands of a tile (rows 4 kq + st of Z_ks, column j), requested one tile ahead like dY.  Not masked: rows that do not exist meet
    // dY = 0, and columns >= cin of a narrow X block (read as column cin - 1) feed dW entries that are never stored.
    const unsigned jx = min(j, cin - 1);
    auto request_z = [&](const Rows& w, float (&az)[4][2]) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            az[st][0] = z.Ph[w.r[st] * (unsigned)z.ldh + j];
            az[st][1] = z.Px[w.r[st] * (unsigned)z.ldx + jx];
        }
    };
    float azn[4][2];
    request_z(rows_of(quad), azn);
    for (int tile = quad; tile < tiles; tile += QUADS) {
        Raw cur[4];
        float az[4][2];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            cur[s] = nxt[s];
            az[s][0] = azn[s][0];
            az[s][1] = azn[s][1];
        }
        const Rows w = rows_of(tile + QUADS);
        request(w, nxt);
        request_z(w, azn);
        float dy[OT][4];
        f32x4 dvk[KC][OT];
        form_and_mix(tile, cur, dy, dvk);
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) {
                    const float bv = kc == 0 ? dy[ot][st] : dvk[kc][ot][st];
                    if (kc == 0) dbv[ot] += bv;
#pragma unroll
                    for (int lt = 0; lt < 2; ++lt) dw[lt][kc][ot] = mfma4(az[st][lt], bv, dw[lt][kc][ot]);
                }
    }
    // lane (j, kq), register r: l = 4 kq + r of tile lt, o = 16 ot + j.  All the old values are requested first, then added and stored
    // (one read-modify-write after the other would be one L2 round trip each).
    {
        float old[2][KC][OT][4];
        auto at = [&](int lt, int kc, int ot, int r) {
            const int li = 4 * kq + r;
            return dW + (unsigned)((ks * KC + kc) * L + (lt == 0 ? cin + li : li)) * HO + 16 * ot + j;
        };
#pragma unroll
        for (int lt = 0; lt < 2; ++lt)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int ot = 0; ot < OT; ++ot)
#pragma unroll
                    for (int r = 0; r < 4; ++r) old[lt][kc][ot][r] = *at(lt, kc, ot, r);      // (rows l >= cin of the X block: some other row of W, readable, not stored)
#pragma unroll
        for (int lt = 0; lt < 2; ++lt)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int ot = 0; ot < OT; ++ot)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (lt == 0 || 4 * kq + r < cin) *at(lt, kc, ot, r) = old[lt][kc][ot][r] + dw[lt][kc][ot][r];
    }
    if (db != nullptr && ks == 0) {
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
            float v = dbv[ot];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (kq == 0) db[16 * ot + j] += v;
        }
    }
}

struct SmallBwd {
    SmallGraph g;                                // CSR of Gs (the transpose of the forward's)
    const float *X, *H, *Tc, *Wg, *Wc, *U, *R, *Cand, *RH, *Zg, *Zc, *dHnew;
    float *dX, *dH, *dP, *ws;
    float *dZ1c, *dZ1g, *dYg, *dYc;              // optional dumps for learned graphs: gradients of the two aggregated slabs, gate and candidate
                                                 // pre-activation gradients
    int N, C, cin, rpt, tiles, acc_x, acc_h, has_bg, has_bc;
    int phase;                                   // as in SmallFwd
    long long P;                                 // floats per row of dP: [dWg | dbg (32) | dWc | dbc (16)]; SB_WAVES / 4 rows per sample
    SmallGraph g2;                               // order 3: CSR of T_2(S) = 2 S^2 - I (the transpose of the forward's second graph)
    const float *Zg2, *Zc2;                      // order 3: the forward's third slabs
};

struct Raw3 {
    float d, u, c;
};
struct Raw2 {
    float a, b;
};

template <int KS, int KC>
__host__ __device__ constexpr int bwd_lds_fixed() { return WgShape<KS>::BWD_WAVES * KC * 16 * 33; }           // floats: the waves' dV tiles
__host__ __device__ constexpr int ws_row_floats(int KS, int LP) { return (2 * KS - 1) * LP + 32; }       // per sample row: dZ_0 | dZ_1 | dYg (32) | dZ_1s [| dZ_2 | dZ_2s]

template <int KS, int KC, int XQ, int MODE>      // MODE 3 (backward only): dense graph, nothing staged -- the split form of a learned graph's backward
__global__ __launch_bounds__(WgShape<KS>::BWD_THREADS) void small_bwd_kernel(SmallBwd a) {
    constexpr int LP = 16 + 4 * XQ;
    constexpr bool STAGED = MODE == 1 || MODE == 2, DENSE = MODE >= 2;
    constexpr int SB_THREADS = WgShape<KS>::BWD_THREADS;
    static_assert(KS == 2 || (KS == 3 && KC == 3 && MODE == 0), "order 3: Ks = Kc = 3, CSR graph, nothing staged");
    extern __shared__ __align__(16) float lds[];
    const int t = threadIdx.x, wave = t >> 6;
    const int C = a.C, N = a.N, NC = N * C, cin = a.cin, L = cin + SC_H;
    const unsigned invC = inv_c(C);
    float* dv = lds + wave * KC * 16 * 33;
    float* D1 = lds + bwd_lds_fixed<KS, KC>();               // STAGED: the slab dZ_1 of the convolution in flight, stride LP
    int* gpl = reinterpret_cast<int*>(D1 + (size_t)NC * LP);
    int* gcl = gpl + ((N + 4) & ~3);
    float* gvl = reinterpret_cast<float*>(gcl + ((a.g.nnz + 3) & ~3));
    const int* gp = STAGED && !DENSE ? gpl : a.g.rowptr;
    const int* gc = STAGED && !DENSE ? gcl : a.g.colidx;
    const float* gv = STAGED && !DENSE ? gvl : a.g.val;
    const size_t r0 = (size_t)blockIdx.x * NC;
    const float* Xb = a.X + r0 * cin;
    const float* Hb = a.H + r0 * SC_H;
    const float* Ub = a.U + r0 * SC_H;
    const float* Rb = a.R + r0 * SC_H;
    const float* Cb = a.Cand + r0 * SC_H;
    const float* RHb = a.RH + r0 * SC_H;
    const float* Zgb = a.Zg + r0 * LP;
    const float* Zcb = a.Zc + r0 * SC_H;
    const float* dHn = a.dHnew + r0 * SC_H;
    float* dXb = a.dX ? a.dX + r0 * cin : nullptr;
    float* dHb = a.dH ? a.dH + r0 * SC_H : nullptr;
    float* wsb = a.ws + (size_t)blockIdx.x * NC * ws_row_floats(KS, LP);
    const float* Zg2b = KS == 3 ? a.Zg2 + r0 * LP : nullptr;
    const float* Zc2b = KS == 3 ? a.Zc2 + r0 * SC_H : nullptr;
    float* dZ2 = wsb + (size_t)NC * (3 * LP + 32);                    // order 3: the third slab's gradients, candidate / gates convolution
    float* dZ2s = wsb + (size_t)NC * (4 * LP + 32);
    float* dZ0 = wsb;
    float* dZ1 = STAGED ? D1 : wsb + (size_t)NC * LP;
    float* dZ1s = STAGED ? D1 : wsb + (size_t)NC * (2 * LP + 32);     // the gates convolution's dZ_1: its own slab, so that a workgroup may run phase 3
                                                                      // while another still reads the candidate's dZ_1 in phase 2 (phase 7)
    float* dYg = a.dYg ? a.dYg + r0 * 32 : wsb + (size_t)NC * 2 * LP;
    // order 2: four waves per row tile (wave & 1 = slab, (wave >> 1) & 1 = role), four tiles at once; order 3: six per tile, two tiles at once
    const int unit = KS == 2 ? (wave & 3) : wave % 6, group = KS == 2 ? (wave >> 2) : wave / 6;
    const int ks = KS == 2 ? (unit & 1) : unit % 3, role = KS == 2 ? (unit >> 1) : unit / 3;
    // phase 7 = phases 2 + 3 in one launch: the gates convolution of a tile reads the gate pre-activation gradients of its OWN rows only, which
    // phase 2 has just formed (a workgroup owns a contiguous range of row tiles); phases 1 -> 2 and 3 -> 4 read the neighbours' dZ_1.
    const int phase = a.phase, split = blockIdx.y, splits = gridDim.y;
    auto runs = [&](int p) { return phase == 0 || phase == p || (phase == 7 && (p == 2 || p == 3)); };
    auto sync = [&](int after) { if (phase == 0 || (phase == 7 && after == 2)) __syncthreads(); };
    const int t_lo = (int)((long long)a.tiles * split / splits), t_hi = (int)((long long)a.tiles * (split + 1) / splits);
    const int row_lo = min(t_lo * a.rpt, NC), row_hi = min(t_hi * a.rpt, NC);
    const int n_lo = div_c(row_lo, invC), n_hi = div_c(row_hi + C - 1, invC);      // the nodes of the workgroup's own rows
    const int tile0 = t_lo + group, tstep = WgShape<KS>::BWD_GROUPS;
    float* dPw = a.dP + (((size_t)blockIdx.x * splits + split) * (SB_WAVES / 4) + group) * a.P;      // (rows per sample: stc_cell_small_param_rows, either order)
    float* dWg = dPw;
    float* dbg = dPw + (size_t)KS * KC * L * 32;
    float* dWc = dbg + 32;
    float* dbc = dWc + (size_t)KS * KC * L * 16;
    const int lane = t & 63;
    float M[KC][4];
    build_mix<KC>(M, a.Tc, true, a.rpt, C, invC, lane & 15, lane >> 4);
    if (STAGED && !DENSE) stage_graph<SB_THREADS>(a.g, N, gpl, gcl, gvl);
    if (phase == 0) __syncthreads();

    // 1: candidate convolution
    if (runs(1)) conv_bwd_phase<KC, XQ, 1, Raw3>(ks == 0 ? Slab{RHb, SC_H, Xb, cin} : (ks == 1 ? Slab{Zcb, SC_H, Zgb + SC_H, LP} : Slab{Zc2b, SC_H, Zg2b + SC_H, LP}),
        a.Wc, ks, role, M, dv, ks == 0 ? dZ0 : (ks == 1 ? dZ1 : dZ2),
        ks == 1 && a.dZ1c ? a.dZ1c + r0 * LP : nullptr, dWc,
        a.has_bc ? dbc : nullptr, a.rpt, min(t_hi, SC_MAX_TILES), tile0, tstep, NC, cin,
        [&](int grow, int col) {
            const size_t e = (unsigned)grow * SC_H + col;
            return Raw3{dHn[e], Ub[e], Cb[e]};
        },
        [](const Raw3& w, int) { return w.d * w.u * (1.f - w.c * w.c); });
    sync(1);                                     // the dZ slabs are complete
    SC_PHASE_END(1);

    // 2: d[R*H | X] = dZ_0 + S^T dZ_1, gate backward
    if (runs(2)) {
        auto from_dz1 = [&](int src, int q) -> f32x4 { return ld4(dZ1 + (unsigned)src * LP + 4 * q); };
        auto from_dz0 = [&](int row, int q) -> f32x4 { return ld4(dZ0 + (unsigned)row * LP + 4 * q); };
        auto gate_bwd =
        [&](int row, int q, f32x4 s) {
            if (q < 4) {
                const size_t e = (unsigned)row * SC_H + 4 * q;
                const f32x4 hh = ld4(Hb + e), rr = ld4(Rb + e), u = ld4(Ub + e), cd = ld4(Cb + e), d = ld4(dHn + e);
                f32x4 gu, gr, dh;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    gu[i] = d[i] * (cd[i] - hh[i]) * u[i] * (1.f - u[i]);
                    gr[i] = s[i] * hh[i] * rr[i] * (1.f - rr[i]);
                    dh[i] = fmaf(d[i], 1.f - u[i], s[i] * rr[i]);
                }
                if (a.dYc != nullptr) {
                    f32x4 yc;
#pragma unroll
                    for (int i = 0; i < 4; ++i) yc[i] = d[i] * u[i] * (1.f - cd[i] * cd[i]);
                    st4(a.dYc + (r0 + (unsigned)row) * SC_H + 4 * q, yc);
                }
                st4(dYg + (unsigned)row * 32 + 4 * q, gu);
                st4(dYg + (unsigned)row * 32 + 16 + 4 * q, gr);
                if (dHb) {
                    if (a.acc_h) {
                        const f32x4 o = ld4(dHb + e);
#pragma unroll
                        for (int i = 0; i < 4; ++i) dh[i] += o[i];
                    }
                    st4(dHb + e, dh);
                }
            } else if (dXb) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = 4 * (q - 4) + i;
                    if (col < cin) {
                        float* p = dXb + (unsigned)row * cin + col;
                        *p = a.acc_x ? *p + s[i] : s[i];
                    }
                }
            }
        };
        if (DENSE && !STAGED)
            aggregate_dense_own<SB_THREADS, LP / 4>(a.g.val, N, C, dZ1, LP, LP, n_lo, n_hi, row_lo, row_hi, from_dz0, gate_bwd);
        else if (DENSE)
            aggregate_dense<SB_THREADS, LP / 4>(a.g.val, N, C, dZ1, LP, split, splits, from_dz0, gate_bwd);
        else if constexpr (KS == 3)
            aggregate_rows2<SB_THREADS, LP / 4>(gp, gc, gv, a.g2.rowptr, a.g2.colidx, a.g2.val, C, invC, row_lo, row_hi, from_dz1,
                                                [&](int src, int q) -> f32x4 { return ld4(dZ2 + (unsigned)src * LP + 4 * q); }, from_dz0, gate_bwd);
        else
            aggregate_rows<SB_THREADS, LP / 4>(gp, gc, gv, NC, C, invC, row_lo, row_hi, from_dz1, from_dz0, gate_bwd);
    }
    sync(2);
    SC_PHASE_END(2);

    // 3: gates convolution
    if (runs(3)) conv_bwd_phase<KC, XQ, 2, Raw2>(ks == 0 ? Slab{Hb, SC_H, Xb, cin} : (ks == 1 ? Slab{Zgb, LP, Zgb + SC_H, LP} : Slab{Zg2b, LP, Zg2b + SC_H, LP}),
        a.Wg, ks, role, M, dv, ks == 0 ? dZ0 : (ks == 1 ? dZ1s : dZ2s),
        ks == 1 && a.dZ1g ? a.dZ1g + r0 * LP : nullptr, dWg,
        a.has_bg ? dbg : nullptr, a.rpt, min(t_hi, SC_MAX_TILES), tile0, tstep, NC, cin,
        [&](int grow, int col) { return Raw2{dYg[(unsigned)grow * 32 + col], dYg[(unsigned)grow * 32 + 16 + col]}; },
        [](const Raw2& w, int ot) { return ot == 0 ? w.a : w.b; });
    sync(3);
    SC_PHASE_END(3);

    // 4: d[H | X] += dZ_0 + S^T dZ_1
    if ((dHb || dXb) && runs(4)) {
        auto from_dz1 = [&](int src, int q) -> f32x4 { return ld4(dZ1s + (unsigned)src * LP + 4 * q); };
        auto from_dz0 = [&](int row, int q) -> f32x4 { return ld4(dZ0 + (unsigned)row * LP + 4 * q); };
        auto add_in =
            [&](int row, int q, f32x4 s) {
                if (q < 4) {
                    if (dHb) {
                        float* p = dHb + (unsigned)row * SC_H + 4 * q;
                        const f32x4 o = ld4(p);
#pragma unroll
                        for (int i = 0; i < 4; ++i) s[i] += o[i];
                        st4(p, s);
                    }
                } else if (dXb) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int col = 4 * (q - 4) + i;
                        if (col < cin) dXb[(unsigned)row * cin + col] += s[i];
                    }
                }
            };
        if (DENSE && !STAGED)
            aggregate_dense_own<SB_THREADS, LP / 4>(a.g.val, N, C, dZ1s, LP, LP, n_lo, n_hi, row_lo, row_hi, from_dz0, add_in);
        else if (DENSE)
            aggregate_dense<SB_THREADS, LP / 4>(a.g.val, N, C, dZ1s, LP, split, splits, from_dz0, add_in);
        else if constexpr (KS == 3)
            aggregate_rows2<SB_THREADS, LP / 4>(gp, gc, gv, a.g2.rowptr, a.g2.colidx, a.g2.val, C, invC, row_lo, row_hi, from_dz1,
                                                [&](int src, int q) -> f32x4 { return ld4(dZ2s + (unsigned)src * LP + 4 * q); }, from_dz0, add_in);
        else
            aggregate_rows<SB_THREADS, LP / 4>(gp, gc, gv, NC, C, invC, row_lo, row_hi, from_dz1, from_dz0, add_in);
    }
}

int xq_of(int cin) { return cin == SC_H ? 4 : (cin >= 1 && cin <= 4 ? 1 : 0); }
size_t graph_lds_bytes(int N, int nnz) { return (size_t)(((N + 4) & ~3) + 2 * ((nnz + 3) & ~3)) * 4; }
constexpr size_t SC_LDS_BUDGET = 156 * 1024;     // of the 160 KB of a compute unit

// Raise a kernel's dynamic-LDS cap only when a launch needs more than it was already granted ON THE CURRENT DEVICE (the attribute belongs to
// the device's copy of the function; the attribute call costs microseconds, a launch here is tens of them).  Devices beyond the table are
// granted on every launch.
constexpr int SC_MAX_DEVICES = 16;
struct Grants { std::atomic<size_t> per_device[SC_MAX_DEVICES]; };
template <class K>
hipError_t allow_lds_once(K kern, size_t bytes, Grants& grants) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    std::atomic<size_t>* granted = dev >= 0 && dev < SC_MAX_DEVICES ? &grants.per_device[dev] : nullptr;
    if (granted && bytes <= granted->load(std::memory_order_relaxed)) return hipSuccess;
    const hipError_t e = stc::allow_lds(kern, bytes);
    if (e == hipSuccess && granted) granted->store(bytes, std::memory_order_relaxed);
    return e;
}
Grants g_granted[2][2][4];                       // [direction][wide input][mode]
Grants g_granted3[2];                            // order 3, backward: [wide input]

}  // namespace

extern "C" int stc_cell_small_supported(int32_t Ks, int32_t Kc, int32_t C, int32_t cin, int32_t h) {
    return (Ks == 2 || Ks == 3) && Kc == Ks && C >= 1 && C <= SC_MAXC && h == SC_H && xq_of(cin) != 0;
}

extern "C" int stc_cell_small_param_rows(void) { return SB_WAVES / 4; }

extern "C" size_t stc_cell_small_workspace_bytes(int32_t n_nodes, int32_t C, int32_t cin, int32_t batch, int32_t Ks) {
    const int xq = xq_of(cin);
    if (xq == 0 || n_nodes < 0 || C < 0 || batch < 0 || (Ks != 2 && Ks != 3)) return 0;
    return (size_t)batch * n_nodes * C * ws_row_floats(Ks, 16 + 4 * xq) * sizeof(float);
}

#define SC_COMMON_CHECKS(name)                                                                                                        \
    STC_REQUIRE(n_nodes >= 0 && batch >= 0 && nnz >= 0, STC_EINVAL, name ": negative size (n_nodes=%d nnz=%d batch=%d)", n_nodes, nnz, batch); \
    STC_REQUIRE(stc_cell_small_supported(Ks, Kc, C, cin, SC_H), STC_EINVAL, name ": unsupported shape (Ks=%d Kc=%d C=%d cin=%d)", Ks, Kc, C, cin); \
    STC_REQUIRE((long long)n_nodes * C < 65536 && (long long)n_nodes * C * batch < (1ll << 26), STC_ELIMIT,                             \
                name ": %lld rows per sample, %d samples: not a small graph", (long long)n_nodes * C, batch);                           \
    if (n_nodes == 0 || batch == 0) return STC_OK;                                                                                       \
    STC_REQUIRE(Ks == 2 || (rowptr2 && (nnz2 == 0 || (colidx2 && val2)) && nnz2 >= 0 && !graph_is_dense), STC_EINVAL,                    \
                name ": order 3 takes the CSR of T_2(S) = 2 S^2 - I as a second graph (and no dense graph)");

extern "C" int stc_cell_small_fwd_f32(const int32_t* rowptr, const int32_t* colidx, const float* val, int32_t n_nodes, int32_t nnz,
                                      int32_t graph_is_dense, const int32_t* rowptr2, const int32_t* colidx2, const float* val2, int32_t nnz2,
                                      const float* X, int32_t cin, const float* H, const float* Tc, int32_t Ks, int32_t Kc, const float* Wg,
                                      const float* bg, const float* Wc, const float* bc, float* U, float* R, float* Cand, float* Hnew, float* RH,
                                      float* Zg, float* Zc, float* Zg2, float* Zc2, float* Z0, float* Z0c, float* Z1c, int32_t phase, int32_t splits,
                                      int32_t batch, int32_t C, void* stream) {
    SC_COMMON_CHECKS("stc_cell_small_fwd_f32")
    STC_REQUIRE(rowptr && (nnz == 0 || (colidx && val)) && X && H && Tc && Wg && Wc && U && R && Cand && Hnew && RH && Zg && Zc, STC_EINVAL,
                "stc_cell_small_fwd_f32: null operand");
    STC_REQUIRE(Hnew != H, STC_EINVAL, "stc_cell_small_fwd_f32: Hnew must not alias H (neighbour rows are read after the first rows are written)");
    const int xq = xq_of(cin);
    STC_REQUIRE(stc::aligned16(H) && stc::aligned16(U) && stc::aligned16(R) && stc::aligned16(RH) && stc::aligned16(Zg) && stc::aligned16(Zc) &&
                    (xq != 4 || stc::aligned16(X)) && stc::aligned16(Z0), STC_EINVAL, "stc_cell_small_fwd_f32: planes must be 16-byte aligned");
    const int npt = 16 / C, rpt = npt * C;
    STC_REQUIRE(phase >= 0 && phase <= 6 && splits >= 1 && splits <= 64 && (phase != 0 || splits == 1), STC_EINVAL,
                "stc_cell_small_fwd_f32: phase %d / splits %d (phase 0 = the whole cell, one workgroup per sample; 1..4 = one phase, 5 = 1 + 2, "
                "6 = 3 + 4 over `splits` workgroups)", phase, splits);
    STC_REQUIRE((Z0c == nullptr) == (Z1c == nullptr) && (Z0c == nullptr || Z0 != nullptr) && stc::aligned16(Z0c) && stc::aligned16(Z1c), STC_EINVAL,
                "stc_cell_small_fwd_f32: Z0c and Z1c come together, with Z0, 16-byte aligned");
    SmallFwd a{{rowptr, colidx, val, nnz}, X, H, Tc, Wg, bg, Wc, bc, U, R, Cand, Hnew, RH, Zg, Zc, Z0, Z0c, Z1c, n_nodes, C, cin, rpt,
               (n_nodes + npt - 1) / npt, phase, {rowptr2, colidx2, val2, nnz2}, Zg2, Zc2};
    if (Ks == 3) {                                           // order 3: global-memory form only (MODE 0), eight waves
        STC_REQUIRE(Zg2 && Zc2 && stc::aligned16(Zg2) && stc::aligned16(Zc2) && !Z0 && !Z0c, STC_EINVAL,
                    "stc_cell_small_fwd_f32: order 3 wants the planes Zg2, Zc2 (16-byte aligned) and has no learned-graph outputs");
        auto kern3 = xq == 4 ? small_fwd_kernel<3, 3, 4, 0> : small_fwd_kernel<3, 3, 1, 0>;
        hipLaunchKernelGGL(kern3, dim3((unsigned)batch, (unsigned)splits), dim3(WgShape<3>::FWD_THREADS), 0, static_cast<hipStream_t>(stream), a);
        STC_LAUNCH_CHECK("stc_cell_small_fwd_f32 (order 3) launch");
        return STC_OK;
    }
    const size_t fixed = (size_t)fwd_lds_fixed<2>() * 4, planes = (size_t)n_nodes * C * (plane_stride(xq) + SQ) * 4;
    const bool dense_graph = graph_is_dense && nnz == (long long)n_nodes * n_nodes;
    const bool dense = phase == 0 && dense_graph && fixed + planes <= SC_LDS_BUDGET;
    const size_t staged = fixed + planes + (dense ? 0 : graph_lds_bytes(n_nodes, nnz));
    const int mode = dense ? 2 : (phase != 0 && dense_graph ? 3 : (phase == 0 && staged <= SC_LDS_BUDGET ? 1 : 0));
    const size_t lds = mode == 1 || mode == 2 ? staged : fixed;
    auto kern = mode == 3 ? (xq == 4 ? small_fwd_kernel<2, 2, 4, 3> : small_fwd_kernel<2, 2, 1, 3>)
              : mode == 2 ? (xq == 4 ? small_fwd_kernel<2, 2, 4, 2> : small_fwd_kernel<2, 2, 1, 2>)
              : mode == 1 ? (xq == 4 ? small_fwd_kernel<2, 2, 4, 1> : small_fwd_kernel<2, 2, 1, 1>)
                          : (xq == 4 ? small_fwd_kernel<2, 2, 4, 0> : small_fwd_kernel<2, 2, 1, 0>);
    const hipError_t e = allow_lds_once(kern, lds, g_granted[0][xq == 4][mode]);
    if (e != hipSuccess) return stc::hip_status(e, "stc_cell_small_fwd_f32 LDS attribute");
    hipLaunchKernelGGL(kern, dim3((unsigned)batch, (unsigned)splits), dim3(WgShape<2>::FWD_THREADS), lds, static_cast<hipStream_t>(stream), a);
    STC_LAUNCH_CHECK("stc_cell_small_fwd_f32 launch");
    return STC_OK;
}

extern "C" int stc_cell_small_bwd_f32(const int32_t* rowptr, const int32_t* colidx, const float* val, int32_t n_nodes, int32_t nnz,
                                      int32_t graph_is_dense, const int32_t* rowptr2, const int32_t* colidx2, const float* val2, int32_t nnz2,
                                      const float* X, int32_t cin, const float* H, const float* Tc, int32_t Ks, int32_t Kc, const float* Wg,
                                      const float* Wc, const float* U, const float* R, const float* Cand, const float* RH, const float* Zg,
                                      const float* Zc, const float* Zg2, const float* Zc2, const float* dHnew,
                                      float* dX, int32_t accumulate_x, float* dH, int32_t accumulate_h, float* dparams, int64_t params_ld,
                                      int32_t has_bg, int32_t has_bc, float* dZ1c, float* dZ1g, float* dYg, float* dYc, void* workspace,
                                      size_t workspace_bytes, int32_t phase, int32_t splits, int32_t batch, int32_t C, void* stream) {
    SC_COMMON_CHECKS("stc_cell_small_bwd_f32")
    STC_REQUIRE(rowptr && (nnz == 0 || (colidx && val)) && X && H && Tc && Wg && Wc && U && R && Cand && RH && Zg && Zc && dHnew && dparams && workspace,
                STC_EINVAL, "stc_cell_small_bwd_f32: null operand");
    const int xq = xq_of(cin), L = cin + SC_H, LP = 16 + 4 * xq;
    const long long P = (long long)Ks * Kc * L * 48 + 48;
    STC_REQUIRE(params_ld >= P, STC_EINVAL, "stc_cell_small_bwd_f32: params_ld %lld < %lld floats per row", (long long)params_ld, P);
    STC_REQUIRE(workspace_bytes >= stc_cell_small_workspace_bytes(n_nodes, C, cin, batch, Ks), STC_EINVAL, "stc_cell_small_bwd_f32: workspace too small");
    STC_REQUIRE(stc::aligned16(H) && stc::aligned16(U) && stc::aligned16(R) && stc::aligned16(Cand) && stc::aligned16(RH) && stc::aligned16(Zg) &&
                    stc::aligned16(Zc) && stc::aligned16(dHnew) && stc::aligned16(workspace) && (dH == nullptr || stc::aligned16(dH)) &&
                    stc::aligned16(dYg),
                STC_EINVAL, "stc_cell_small_bwd_f32: planes must be 16-byte aligned");
    STC_REQUIRE(dH != dHnew && (const float*)dX != dHnew && (dX == nullptr || (const float*)dX != (const float*)dH), STC_EINVAL,
                "stc_cell_small_bwd_f32: dHnew, dX and dH must be distinct buffers");
    const int npt = 16 / C, rpt = npt * C;
    SmallBwd a{{rowptr, colidx, val, nnz}, X, H, Tc, Wg, Wc, U, R, Cand, RH, Zg, Zc, dHnew, dX, dH, dparams, static_cast<float*>(workspace),
               dZ1c, dZ1g, dYg, dYc, n_nodes, C, cin, rpt, (n_nodes + npt - 1) / npt, accumulate_x, accumulate_h, has_bg, has_bc, phase, params_ld,
               {rowptr2, colidx2, val2, nnz2}, Zg2, Zc2};
    STC_REQUIRE(((phase >= 0 && phase <= 4) || phase == 7) && splits >= 1 && splits <= 64 && (phase != 0 || splits == 1), STC_EINVAL,
                "stc_cell_small_bwd_f32: phase %d / splits %d (0 = the whole cell; 1..4 = one phase; 7 = 2 + 3)", phase, splits);
    if (Ks == 3) {                                           // order 3: global-memory form only (MODE 0), twelve waves
        STC_REQUIRE(Zg2 && Zc2 && stc::aligned16(Zg2) && stc::aligned16(Zc2) && !dZ1c && !dZ1g && !dYc, STC_EINVAL,
                    "stc_cell_small_bwd_f32: order 3 wants the planes Zg2, Zc2 (16-byte aligned) and leaves no learned-graph operands");
        auto kern3 = xq == 4 ? small_bwd_kernel<3, 3, 4, 0> : small_bwd_kernel<3, 3, 1, 0>;
        const size_t lds3 = (size_t)bwd_lds_fixed<3, 3>() * 4;
        const hipError_t e3 = allow_lds_once(kern3, lds3, g_granted3[xq == 4]);
        if (e3 != hipSuccess) return stc::hip_status(e3, "stc_cell_small_bwd_f32 LDS attribute");
        hipLaunchKernelGGL(kern3, dim3((unsigned)batch, (unsigned)splits), dim3(WgShape<3>::BWD_THREADS), lds3, static_cast<hipStream_t>(stream), a);
        STC_LAUNCH_CHECK("stc_cell_small_bwd_f32 (order 3) launch");
        return STC_OK;
    }
    const size_t fixed = (size_t)bwd_lds_fixed<2, 2>() * 4, planes = (size_t)n_nodes * C * LP * 4;
    const bool full = graph_is_dense && nnz == (long long)n_nodes * n_nodes;
    const bool dense = phase == 0 && full && fixed + planes <= SC_LDS_BUDGET;
    const size_t staged = fixed + planes + (dense ? 0 : graph_lds_bytes(n_nodes, nnz));
    const int mode = dense ? 2 : (phase != 0 && full ? 3 : (phase == 0 && staged <= SC_LDS_BUDGET ? 1 : 0));
    const size_t lds = mode == 1 || mode == 2 ? staged : fixed;
    auto kern = mode == 3 ? (xq == 4 ? small_bwd_kernel<2, 2, 4, 3> : small_bwd_kernel<2, 2, 1, 3>)
              : mode == 2 ? (xq == 4 ? small_bwd_kernel<2, 2, 4, 2> : small_bwd_kernel<2, 2, 1, 2>)
              : mode == 1 ? (xq == 4 ? small_bwd_kernel<2, 2, 4, 1> : small_bwd_kernel<2, 2, 1, 1>)
                          : (xq == 4 ? small_bwd_kernel<2, 2, 4, 0> : small_bwd_kernel<2, 2, 1, 0>);
    const hipError_t e = allow_lds_once(kern, lds, g_granted[1][xq == 4][mode]);
    if (e != hipSuccess) return stc::hip_status(e, "stc_cell_small_bwd_f32 LDS attribute");
    hipLaunchKernelGGL(kern, dim3((unsigned)batch, (unsigned)splits), dim3(WgShape<2>::BWD_THREADS), lds, static_cast<hipStream_t>(stream), a);
    STC_LAUNCH_CHECK("stc_cell_small_bwd_f32 launch");
    return STC_OK;
}
