// MFMA fast path of the BDG_Dif node kernel (reference STC_GNN.py:38-45 and its autograd) for
// gfx950: exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32: bit-for-bit an fmaf chain, 64 FLOP/clk/SIMD).
//
// Shapes: C = 16*NRB categories, Ho = 16*HB outputs, L = 4*LQ features per slab, Ks = Kc = K.
// One WAVE owns one node row (C x L per Chebyshev slab) at a time; waves stream independently
// through the nodes -- no barrier in the main loop.  Everything that is the same for every node
// (W, T_c) is laid out once per workgroup in LDS in MFMA *fragment order* (64 consecutive floats per
// k-step, so each operand fetch is one conflict-free ds_read_b32); everything per-node stays in
// registers:
//
//   forward   U = Z.W      A = Z rows straight from HBM (lane = category row, 16-byte loads),
//                          B = W fragments (LDS), accumulators = U tiles [NRB][K*HB]
//             Y = U_0 + sum_c T_c^T U_c     the U_c accumulators ARE the B operand (their register
//                          index is the contraction index c'), A = T_c fragments (LDS): no LDS trip
//   backward  Q_c = T_c dY in both orientations (operands swapped), dZ^T = W Q^T, dW += Z^T Q,
//             with the accumulators of one product again feeding the next as B operands;
//             dW/db live in registers across all nodes of a wave and are combined per workgroup
//             in a fixed order (bitwise reproducible), then summed by bdg_node_reduce_kernel.
//
// A k-step of the 16x16x4 MFMA takes lane quarter q = lane>>4 as its k index.  Since a sum over k
// does not care about order, lane (row, q) simply owns columns {16m+4q .. 16m+4q+3} of its row
// (one float4), used at steps 4m..4m+3: kcol(s,q) below.  Both operands are built with the same map.
#include "stc_node_frag.h"

namespace {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// slab column that lane quarter q contributes at k-step s (LQ = L/4 steps per slab)
template <int LQ>
__host__ __device__ constexpr int kcol(int s, int q) {
    constexpr int N16 = LQ / 4;
    return s < 4 * N16 ? 16 * (s / 4) + 4 * q + (s % 4) : 16 * N16 + 4 * (s - 4 * N16) + q;
}

// --------------------------------------------------------------------------------------- forward
template <int NRB, int K, int LQ>
struct ZFrag {      // one node's A operand: this lane's row of every slab / row block
    static constexpr int N16 = LQ / 4, NREM = LQ - 4 * (LQ / 4);
    float4 v4[K][NRB][AtLeast1<N16>::v];
    float r[K][NRB][AtLeast1<NREM>::v];

    __device__ __forceinline__ void load(const ZPtrs& Z, int node, int j, int q) {
        constexpr int C = 16 * NRB, L = 4 * LQ;
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                const float* row = Z.p[n] + ((size_t)node * C + 16 * rb + j) * L;
#pragma unroll
                for (int m = 0; m < N16; ++m) v4[n][rb][m] = *reinterpret_cast<const float4*>(row + 16 * m + 4 * q);
#pragma unroll
                for (int u = 0; u < NREM; ++u) r[n][rb][u] = row[16 * N16 + 4 * u + q];
            }
    }
    // FEW categories (NRB = 1, Cr <= 16 rows per node): the node's rows start at r0 = node * Cr; lanes past the node's rows read its last row
    // (finite; what they produce is never stored and meets zero rows / columns of T_c)
    __device__ __forceinline__ void load_cr(const ZPtrs& Z, size_t r0, int jr, int q) {
        static_assert(NRB == 1, "ragged row tiles: one row block");
        constexpr int L = 4 * LQ;
#pragma unroll
        for (int n = 0; n < K; ++n) {
            const float* row = Z.p[n] + (r0 + jr) * L;
#pragma unroll
            for (int m = 0; m < N16; ++m) v4[n][0][m] = *reinterpret_cast<const float4*>(row + 16 * m + 4 * q);
#pragma unroll
            for (int u = 0; u < NREM; ++u) r[n][0][u] = row[16 * N16 + 4 * u + q];
        }
    }
    // value this lane contributes at k-step s of slab n, row block rb (s is a compile-time constant after unrolling)
    __device__ __forceinline__ float at(int n, int rb, int s) const {
        if (s < 4 * N16) {
            const float4 v = v4[n][rb][s / 4 < N16 ? s / 4 : 0];
            return (s % 4 == 0) ? v.x : (s % 4 == 1) ? v.y : (s % 4 == 2) ? v.z : v.w;
        }
        return r[n][rb][s - 4 * N16 < NREM && s >= 4 * N16 ? s - 4 * N16 : 0];
    }
};

template <int NRB, int HB, int K, int LQ, int EPI>
__global__ __launch_bounds__(MF_THREADS, ((NRB * K * LQ <= 32) ? 3 : 2)) void node_fwd_mfma_kernel(
    ZPtrs Z, const float* __restrict__ Tc, const float* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ Y, int nodes, int Lw, FwdEpi epi, int Cr) {
    constexpr int C = 16 * NRB, Ho = 16 * HB, L = 4 * LQ, NCB = K * HB;
    constexpr int HID = 16;            // hidden width the fused epilogues are built for
    // RAG: the plain node kernel at NRB = 1 takes nodes of Cr <= 16 categories (rows of a node start at node * Cr; T_c is (Cr, Cr) and enters the
    // fragments zero padded): every C <= 16 runs here, and the host packs 16 / C nodes of few categories into one such node (stc_hip/ops.py)
    constexpr bool RAG = NRB == 1 && EPI == EPI_NONE;
    const int CR = RAG ? Cr : C;
    static_assert(EPI == EPI_NONE || (EPI == EPI_GATES && HB == 2) || (EPI == EPI_BLEND && HB == 1), "epilogue needs hidden = 16");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wf = smem;                              // [K][LQ][NCB][64]   B fragments of the projection
    float* Tf = smem + K * LQ * NCB * 64;          // [K-1][NRB rb][NRB kb][4][64]   A fragments of the mix
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, q = lane >> 4;

    for (int idx = tid; idx < K * LQ * NCB * 64; idx += MF_THREADS) {
        const int ll = idx & 63;
        int t = idx >> 6;
        const int cb = t % NCB; t /= NCB;
        const int s = t % LQ, n = t / LQ;
        const int kc = kcol<LQ>(s, ll >> 4);
        const int c = cb / HB, o = (cb % HB) * 16 + (ll & 15);
        Wf[idx] = kc < Lw ? W[((size_t)(n * K + c) * Lw + kc) * Ho + o] : 0.f;   // pad columns [Lw, L) contribute nothing
    }
    for (int idx = tid; idx < (K - 1) * NRB * NRB * 4 * 64; idx += MF_THREADS) {
        const int ll = idx & 63;
        int t = idx >> 6;
        const int st = t & 3; t >>= 2;
        const int kb = t % NRB; t /= NRB;
        const int rb = t % NRB, c1 = t / NRB;
        // A[i = d][k = c'] = T_c[c'][d]  with c' = 16kb + 4q + st, d = 16rb + i
        const int kk = 16 * kb + 4 * (ll >> 4) + st, dd = 16 * rb + (ll & 15);
        Tf[idx] = (kk < CR && dd < CR) ? Tc[(size_t)(c1 + 1) * CR * CR + kk * CR + dd] : 0.f;
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = gridDim.x * MF_WAVES;
    float bv[HB];
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) bv[hb] = bias ? bias[16 * hb + j] : 0.f;

    int node = blockIdx.x * MF_WAVES + wave;
    ZFrag<NRB, K, LQ> cur, nxt;
    const int jr = j < CR ? j : CR - 1;
    auto load_node = [&](ZFrag<NRB, K, LQ>& z, int nd) {
        if constexpr (RAG) z.load_cr(Z, (size_t)nd * CR, jr, q);
        else z.load(Z, nd, j, q);
    };
    if (node < nodes) load_node(cur, node);
    while (node < nodes) {
        const int next_node = node + nw;
        if (next_node < nodes) load_node(nxt, next_node);        // software prefetch: lands while this node computes
        // epilogue operands in accumulator layout (row 16rb + 4q + r, column j): needed only after the MFMAs
        float hv[NRB][4], uv[NRB][4];
        if (EPI != EPI_NONE) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t e = ((size_t)node * C + 16 * rb + 4 * q + r) * HID + j;
                    hv[rb][r] = epi.H[e];
                    if (EPI == EPI_BLEND) uv[rb][r] = epi.U[e];
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        const int lo = opaque(lane);

        f32x4 acc[NRB][NCB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

        // project: U[rb][cb] += A(Z) * B(W)
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int s = 0; s < LQ; ++s) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    const float b = Wf[((n * LQ + s) * NCB + cb) * 64 + lo];
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) acc[rb][cb] = mfma16(cur.at(n, rb, s), b, acc[rb][cb]);
                }
            }

        // mix: Y[rb][hb] = U_0[rb][hb] + sum_{c>=1} sum_kb T_c^T[rb][kb] U_c[kb][hb]   (U_c read from its accumulators)
#pragma unroll
        for (int c1 = 0; c1 < K - 1; ++c1)
#pragma unroll
            for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) {
                        const float a = Tf[((((c1 * NRB + rb) * NRB + kb) * 4 + t)) * 64 + lo];
#pragma unroll
                        for (int hb = 0; hb < HB; ++hb)
                            acc[rb][hb] = mfma16(a, acc[kb][(c1 + 1) * HB + hb][t], acc[rb][hb]);
                    }

        // accumulator layout: lane (j, q), register r  <->  row 4q + r, column j of the 16 x 16 tile
        if (EPI == EPI_NONE) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (!RAG || 4 * q + r < CR) Y[((size_t)node * CR + 16 * rb + 4 * q + r) * Ho + 16 * hb + j] = acc[rb][hb][r] + bv[hb];
        } else if (EPI == EPI_GATES) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t row = (size_t)node * C + 16 * rb + 4 * q + r;
                    const float u = sigmoid_f(acc[rb][0][r] + bv[0]);
                    const float g = sigmoid_f(acc[rb][HB - 1][r] + bv[HB - 1]);
                    epi.U_out[row * HID + j] = u;
                    epi.R_out[row * HID + j] = g;
                    epi.CandIn[row * L + epi.cin + j] = g * hv[rb][r];
                }
            // the Xt columns and the zero padding of CandIn come from this lane's own row of slab 0
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                float* crow = epi.CandIn + ((size_t)node * C + 16 * rb + j) * L;
#pragma unroll
                for (int s = 0; s < LQ; ++s) {
                    const int col = kcol<LQ>(s, q);
                    if (col < epi.cin) crow[col] = cur.at(0, rb, s);
                    else if (col >= epi.cin + HID) crow[col] = 0.f;
                }
            }
        } else {
            float hn[NRB][4];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t e = ((size_t)node * C + 16 * rb + 4 * q + r) * HID + j;
                    const float c = tanhf(acc[rb][0][r] + bv[0]);
                    const float u = uv[rb][r];
                    hn[rb][r] = (1.f - u) * hv[rb][r] + u * c;
                    epi.Cand[e] = c;
                    epi.Hnew[e] = hn[rb][r];
                }
            store_state_copies<NRB>(epi, (size_t)node * C, j, q, hn);
        }
        cur = nxt;
        node = next_node;
    }
}

// --------------------------------------------------------------------------------------- backward
// rough register need of the prefetching schedule; above ~210 the kernel runs one wave per SIMD without prefetch
template <int NRB, int HB, int K, int LQ>
struct BwdPlan {
    static constexpr int LB = (4 * LQ + 15) / 16;
    static constexpr int regs = K * LB * K * HB * 4 + 2 * (2 * NRB * HB * 4) + K * LB * NRB * 4 + 2 * NRB * HB * 4 + 4 * NRB + 24;
    static constexpr bool prefetch = regs <= 232;
};

template <int NRB, int HB, int K, int LQ, int PRO>
__global__ __launch_bounds__(MF_THREADS, (BwdPlan<NRB, HB, K, LQ>::prefetch ? 2 : 1)) void node_bwd_mfma_kernel(
    ZPtrs Z, const float* __restrict__ Tc, const float* __restrict__ W, const float* __restrict__ dY,
    DZPtrs dZ, float* __restrict__ partial, int nodes, int want_db, int Lw, BwdPro pro, int Cr) {
    constexpr int C = 16 * NRB, Ho = 16 * HB, L = 4 * LQ, LB = (L + 15) / 16;
    constexpr bool PF = BwdPlan<NRB, HB, K, LQ>::prefetch && PRO == PRO_NONE;
    constexpr bool RAG = NRB == 1 && PRO == PRO_NONE;      // nodes of Cr <= 16 categories, as in the forward kernel
    const int CR = RAG ? Cr : C;
    constexpr int nTf = (K - 1) * NRB * NRB * 4 * 64;
    constexpr int nWf = K * LB * K * HB * 4 * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* TfA = smem;                 // [K-1][NRB rb][NRB kb][4][64]      T_c[16rb+i][16kb+4q+t]
    float* WfD = smem + nTf;           // [K n][LB][K c][HB][4][64]         W[(n,c,16lb+i)][16hb+4q+t]
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, q = lane >> 4;

    for (int idx = tid; idx < nTf; idx += MF_THREADS) {
        const int ll = idx & 63;
        int t = idx >> 6;
        const int st = t & 3; t >>= 2;
        const int kb = t % NRB; t /= NRB;
        const int rb = t % NRB, c1 = t / NRB;
        const int rr = 16 * rb + (ll & 15), cc = 16 * kb + 4 * (ll >> 4) + st;
        TfA[idx] = (rr < CR && cc < CR) ? Tc[(size_t)(c1 + 1) * CR * CR + rr * CR + cc] : 0.f;
    }
    for (int idx = tid; idx < nWf; idx += MF_THREADS) {
        const int ll = idx & 63;
        int t = idx >> 6;
        const int st = t & 3; t >>= 2;
        const int hb = t % HB; t /= HB;
        const int c = t % K; t /= K;
        const int lb = t % LB, n = t / LB;
        const int l = 16 * lb + (ll & 15);
        WfD[idx] = l < Lw ? W[((size_t)(n * K + c) * Lw + l) * Ho + 16 * hb + 4 * (ll >> 4) + st] : 0.f;
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = gridDim.x * MF_WAVES;

    f32x4 dWt[K][LB][K][HB];          // dW tiles: rows l = 16lb + 4q + r, columns o = 16hb + j
    float dbp[HB];
#pragma unroll
    for (int n = 0; n < K; ++n)
#pragma unroll
        for (int lb = 0; lb < LB; ++lb)
#pragma unroll
            for (int c = 0; c < K; ++c)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) dWt[n][lb][c][hb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) dbp[hb] = 0.f;

    int node = blockIdx.x * MF_WAVES + wave;
    DyFrag<NRB, HB> g, gn;
    auto load_dy = [&](DyFrag<NRB, HB>& f, int nd) {
        if constexpr (RAG) f.load_cr(dY, (size_t)nd * CR, CR, j, q);      // rows past the node's: zeros (they meet every product as a factor)
        else f.load(dY, nd, j, q);
    };
    if (PF && node < nodes) load_dy(g, node);
    while (node < nodes) {
        const int next_node = node + nw;
        const size_t r0 = (size_t)node * CR;
        if constexpr (PRO == PRO_GATES || PRO == PRO_GATES_CAND) load_gates_grad<NRB, HB, L, PRO == PRO_GATES_CAND>(g, pro, node, j, q);
        else if constexpr (PRO == PRO_BLEND) load_blend_grad<NRB>(g, pro, node, j, q);
        else if (!PF) load_dy(g, node);
        // this node's Z columns for the dW product (needed last: in flight during the Q and dZ phases)
        float za[K][LB][NRB][4];
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int lb = 0; lb < LB; ++lb) {
                const bool ok = 16 * lb + j < L;
                const float* col = Z.p[n] + r0 * L + 16 * lb + (ok ? j : 0);
#pragma unroll
                for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int zrow = 16 * kb + 4 * q + t;        // (RAG: rows past the node's read its last row -- finite, and Q_c is zero there)
                        const float zv = col[(size_t)(RAG && zrow >= CR ? CR - 1 : zrow) * L];
                        za[n][lb][kb][t] = ok ? zv : 0.f;
                    }
            }
        if (PF && next_node < nodes) load_dy(gn, next_node);
        if (PF) __builtin_amdgcn_sched_barrier(0);
        const int lo = opaque(lane);

#pragma unroll
        for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
                dbp[hb] += (g.d[kb][hb][0] + g.d[kb][hb][1]) + (g.d[kb][hb][2] + g.d[kb][hb][3]);

        // ---- Q_c^T tiles (rows o, columns c'): B operands of dZ.   Q_0^T = dY^T is g.v
        f32x4 Qv[AtLeast1<K - 1>::v][NRB][HB];
#pragma unroll
        for (int c1 = 0; c1 < K - 1; ++c1) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) Qv[c1][rb][hb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) {
                        const float tf = TfA[(((c1 * NRB + rb) * NRB + kb) * 4 + t) * 64 + lo];
#pragma unroll
                        for (int hb = 0; hb < HB; ++hb) Qv[c1][rb][hb] = mfma16(g.d[kb][hb][t], tf, Qv[c1][rb][hb]);   // dY^T . T_c^T
                    }
        }

        // ---- dZ_n^T tile (rows l, columns c') = sum_{c,o} W[(n,c,l)][o] * Q_c[c'][o]
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int lb = 0; lb < LB; ++lb) {
                f32x4 z[NRB];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) z[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < K; ++c)
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const float wf = WfD[((((n * LB + lb) * K + c) * HB + hb) * 4 + t) * 64 + lo];
#pragma unroll
                            for (int rb = 0; rb < NRB; ++rb) {
                                const float qv = c == 0 ? g.v[rb][hb][t] : Qv[c > 0 ? c - 1 : 0][rb][hb][t];
                                z[rb] = mfma16(wf, qv, z[rb]);
                            }
                        }
                if (16 * lb + 4 * q < L && (!RAG || j < CR)) {
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb)
                        *reinterpret_cast<float4*>(dZ.p[n] + (r0 + 16 * rb + j) * L + 16 * lb + 4 * q) =
                            make_float4(z[rb][0], z[rb][1], z[rb][2], z[rb][3]);
                }
            }

        // ---- Q_c tiles (rows c', columns o): B operands of dW.   Q_0 = dY is g.d
        f32x4 Qd[AtLeast1<K - 1>::v][NRB][HB];
#pragma unroll
        for (int c1 = 0; c1 < K - 1; ++c1) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) Qd[c1][rb][hb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) {
                        const float tf = TfA[(((c1 * NRB + rb) * NRB + kb) * 4 + t) * 64 + lo];
#pragma unroll
                        for (int hb = 0; hb < HB; ++hb) Qd[c1][rb][hb] = mfma16(tf, g.d[kb][hb][t], Qd[c1][rb][hb]);   // T_c . dY
                    }
        }

        // ---- dW_{n,c} tile (rows l, columns o) += sum_{c'} Z_n[c'][l] * Q_c[c'][o]
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int lb = 0; lb < LB; ++lb)
#pragma unroll
                for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int c = 0; c < K; ++c)
#pragma unroll
                            for (int hb = 0; hb < HB; ++hb) {
                                const float qd = c == 0 ? g.d[kb][hb][t] : Qd[c > 0 ? c - 1 : 0][kb][hb][t];
                                dWt[n][lb][c][hb] = mfma16(za[n][lb][kb][t], qd, dWt[n][lb][c][hb]);
                            }
        if (PF) g = gn;
        node = next_node;
    }

    combine_dw<K, LB, HB>(smem, dWt, dbp, partial, Lw, want_db);
}

template <int NRB, int HB, int K, int LQ, int EPI = EPI_NONE>
int launch_fwd(const float* const* Z, const float* Tc, const float* W, const float* bias, float* Y,
               long long nodes, int Lw, hipStream_t stream, FwdEpi epi = FwdEpi{}, int Cr = 16 * NRB) {
    constexpr int NCB = K * HB;
    const size_t lds = (size_t)(K * LQ * NCB * 64 + (K - 1) * NRB * NRB * 4 * 64) * sizeof(float);
    if (lds > stc::kMaxLdsBytes) return STC_NOT_HANDLED;
    auto kern = node_fwd_mfma_kernel<NRB, HB, K, LQ, EPI>;
    if (int rc = stc::hip_status(stc::allow_lds(kern, lds), "hipFuncSetAttribute(node fwd mfma)")) return rc;
    static const int resident = stc::resident_blocks(kern, MF_THREADS, lds, 2);   // persistent grid = what fits at once
    ZPtrs zp{};
    for (int n = 0; n < K; ++n) zp.p[n] = Z[n];
    const long long want = (nodes + MF_WAVES - 1) / MF_WAVES;
    const int grid = (int)(want < resident ? want : resident);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MF_THREADS), lds, stream, zp, Tc, W, bias, Y, (int)nodes, Lw, epi, Cr);
    STC_LAUNCH_CHECK("node_fwd_mfma launch");
    return STC_OK;
}

template <int NRB, int HB, int K, int LQ, int PRO = PRO_NONE>
int launch_bwd(const float* const* Z, const float* Tc, const float* W, const float* dY, float* const* dZ,
               float* partial, int* n_partials, int want_db, long long nodes, int Lw, hipStream_t stream, BwdPro pro = BwdPro{}, int Cr = 16 * NRB) {
    constexpr int L = 4 * LQ, Ho = 16 * HB, LB = (L + 15) / 16, nW = K * K * L * Ho;
    const size_t frag = (size_t)((K - 1) * NRB * NRB * 4 * 64 + K * LB * K * HB * 4 * 64);
    const size_t slabs = (size_t)MF_WAVES * (nW + Ho);
    const size_t lds = (frag > slabs ? frag : slabs) * sizeof(float);
    if (lds > stc::kMaxLdsBytes) return STC_NOT_HANDLED;
    auto kern = node_bwd_mfma_kernel<NRB, HB, K, LQ, PRO>;
    if (int rc = stc::hip_status(stc::allow_lds(kern, lds), "hipFuncSetAttribute(node bwd mfma)")) return rc;
    static const int resident = stc::resident_blocks(kern, MF_THREADS, lds, 1);
    ZPtrs zp{};
    DZPtrs dzp{};
    for (int n = 0; n < K; ++n) { zp.p[n] = Z[n]; dzp.p[n] = dZ[n]; }
    const long long want = (nodes + MF_WAVES - 1) / MF_WAVES;
    int grid = resident < MF_BWD_MAX_GRID ? resident : MF_BWD_MAX_GRID;
    if (want < grid) grid = (int)want;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MF_THREADS), lds, stream, zp, Tc, W, dY, dzp, partial, (int)nodes, want_db, Lw, pro, Cr);
    STC_LAUNCH_CHECK("node_bwd_mfma launch");
    *n_partials = grid;
    return STC_OK;
}

bool fast_path_shape(int Ks, int Kc, int C, int L, int Ho, long long nodes) {
    return Ks == Kc && Ks >= 1 && Ks <= 3 && (C == 16 || C == 32 || C == 64) && (Ho == 16 || Ho == 32) &&
           (L == 20 || L == 32) && nodes > 0 && nodes < (1ll << 31) / C;
}

}  // namespace

// C in {16,32,64} x Ho in {16,32} x K in {1,2,3} x L in {20,32}
#define STC_MF_CASE(NRB_, HB_, CALL)                                                                    \
    if (C == 16 * NRB_ && Ho == 16 * HB_) {                                                             \
        if (Ks == 1 && L == 20) return CALL(NRB_, HB_, 1, 5);                                           \
        if (Ks == 1 && L == 32) return CALL(NRB_, HB_, 1, 8);                                           \
        if (Ks == 2 && L == 20) return CALL(NRB_, HB_, 2, 5);                                           \
        if (Ks == 2 && L == 32) return CALL(NRB_, HB_, 2, 8);                                           \
        if (Ks == 3 && L == 20) return CALL(NRB_, HB_, 3, 5);                                           \
        if (Ks == 3 && L == 32) return CALL(NRB_, HB_, 3, 8);                                           \
    }
#define STC_MF_DISPATCH(CALL)                                                                           \
    STC_MF_CASE(1, 1, CALL) STC_MF_CASE(1, 2, CALL) STC_MF_CASE(2, 1, CALL)                             \
    STC_MF_CASE(2, 2, CALL) STC_MF_CASE(4, 1, CALL) STC_MF_CASE(4, 2, CALL)

int stc_node_fwd_mfma(const float* const* Z, int Ks, const float* Tc, int Kc, const float* W, const float* bias,
                      float* Y, long long nodes, int C, int L, int Lw, int Ho, hipStream_t stream) {
    // few categories (1 <= C < 16): the one-row-block kernels on nodes of Cr = C rows (RAG)
    const int Cr = C;
    if (C >= 1 && C < 16) C = 16;
    if (!fast_path_shape(Ks, Kc, C, L, Ho, nodes)) return STC_NOT_HANDLED;
    if (!all_aligned16(Z, Ks) || !stc::aligned16(Y)) return STC_NOT_HANDLED;
#define FWD_CALL(a, b, c, d) launch_fwd<a, b, c, d>(Z, Tc, W, bias, Y, nodes, Lw, stream, FwdEpi{}, Cr)
    STC_MF_DISPATCH(FWD_CALL)
#undef FWD_CALL
    return STC_NOT_HANDLED;
}

// ---- fused cell epilogues: hidden width 16, C in {16,32,64}, L in {20,32}, K in {1,2,3}
#define STC_MF_EPI_CASE(NRB_, CALL)                                                                     \
    if (C == 16 * NRB_) {                                                                               \
        if (K == 1 && L == 20) return CALL(NRB_, 1, 5);                                                 \
        if (K == 1 && L == 32) return CALL(NRB_, 1, 8);                                                 \
        if (K == 2 && L == 20) return CALL(NRB_, 2, 5);                                                 \
        if (K == 2 && L == 32) return CALL(NRB_, 2, 8);                                                 \
        if (K == 3 && L == 20) return CALL(NRB_, 3, 5);                                                 \
        if (K == 3 && L == 32) return CALL(NRB_, 3, 8);                                                 \
    }

int stc_cell_fused_shape_ok(int K, int C, int L, int h) {
    return K >= 1 && K <= 3 && (C == 16 || C == 32 || C == 64) && (L == 20 || L == 32) && h == 16;
}

int stc_cell_gates_fwd_mfma(const float* const* Z, int K, const float* Tc, const float* W, const float* bias,
                            const float* H, float* U, float* R, float* CandIn,
                            long long nodes, int C, int L, int Lw, int cin, hipStream_t stream) {
    if (!stc_cell_fused_shape_ok(K, C, L, 16) || nodes <= 0 || nodes >= (1ll << 31) / C || !all_aligned16(Z, K)) return STC_NOT_HANDLED;
    FwdEpi epi{};
    epi.H = H; epi.U_out = U; epi.R_out = R; epi.CandIn = CandIn; epi.cin = cin;
#define GATES_CALL(a, c, d) launch_fwd<a, 2, c, d, EPI_GATES>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi)
    STC_MF_EPI_CASE(1, GATES_CALL) STC_MF_EPI_CASE(2, GATES_CALL) STC_MF_EPI_CASE(4, GATES_CALL)
#undef GATES_CALL
    return STC_NOT_HANDLED;
}

int stc_cell_blend_fwd_mfma(const float* const* Z, int K, const float* Tc, const float* W, const float* bias,
                            const float* U, const float* H, float* Cand, float* Hnew, const StcStateCopies* copies,
                            long long nodes, int C, int L, int Lw, hipStream_t stream) {
    if (!stc_cell_fused_shape_ok(K, C, L, 16) || nodes <= 0 || nodes >= (1ll << 31) / C || !all_aligned16(Z, K)) return STC_NOT_HANDLED;
    FwdEpi epi{};
    epi.H = H; epi.U = U; epi.Cand = Cand; epi.Hnew = Hnew;
    set_state_copies(epi, copies);
#define BLEND_CALL(a, c, d) launch_fwd<a, 1, c, d, EPI_BLEND>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi)
    STC_MF_EPI_CASE(1, BLEND_CALL) STC_MF_EPI_CASE(2, BLEND_CALL) STC_MF_EPI_CASE(4, BLEND_CALL)
#undef BLEND_CALL
    return STC_NOT_HANDLED;
}

int stc_cell_gates_bwd_mfma(const float* const* Z, int K, const float* Tc, const float* W,
                            const float* dCandIn, const float* dU, const float* H, const float* U, const float* R, const float* dH_in, const float* Cand,
                            float* const* dZ, float* dXt, float* dH, float* partial, int* n_partials, int want_db,
                            long long nodes, int C, int L, int Lw, int cin, int dh_scaled, hipStream_t stream) {
    if (!stc_cell_fused_shape_ok(K, C, L, 16) || nodes <= 0 || nodes >= (1ll << 31) / C || !all_aligned16(Z, K)) return STC_NOT_HANDLED;
    for (int n = 0; n < K; ++n)
        if (!stc::aligned16(dZ[n])) return STC_NOT_HANDLED;
    if (!(stc::aligned16(dCandIn) && (!dU || stc::aligned16(dU)) && (!Cand || stc::aligned16(Cand)) && stc::aligned16(H) && stc::aligned16(U) && stc::aligned16(R) &&
          stc::aligned16(dH) && (!dH_in || stc::aligned16(dH_in))))
        return STC_NOT_HANDLED;
    BwdPro pro{};
    pro.Cand = Cand;
    pro.dCandIn = dCandIn; pro.dU = dU; pro.H = H; pro.U = U; pro.R = R; pro.dH_in = dH_in; pro.dXt = dXt; pro.dH = dH; pro.cin = cin; pro.dh_scaled = dh_scaled;
#define GBWD_CALL(a, c, d) (Cand ? launch_bwd<a, 2, c, d, PRO_GATES_CAND>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro) \
                                 : launch_bwd<a, 2, c, d, PRO_GATES>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro))
    STC_MF_EPI_CASE(1, GBWD_CALL) STC_MF_EPI_CASE(2, GBWD_CALL) STC_MF_EPI_CASE(4, GBWD_CALL)
#undef GBWD_CALL
    return STC_NOT_HANDLED;
}

int stc_cell_cand_bwd_mfma(const float* const* Z, int K, const float* Tc, const float* W,
                           const float* dHnew, const float* U, const float* Cand,
                           float* const* dZ, float* partial, int* n_partials, int want_db,
                           long long nodes, int C, int L, int Lw, hipStream_t stream) {
    if (!stc_cell_fused_shape_ok(K, C, L, 16) || nodes <= 0 || nodes >= (1ll << 31) / C || !all_aligned16(Z, K)) return STC_NOT_HANDLED;
    for (int n = 0; n < K; ++n)
        if (!stc::aligned16(dZ[n])) return STC_NOT_HANDLED;
    if (!(stc::aligned16(dHnew) && stc::aligned16(U) && stc::aligned16(Cand))) return STC_NOT_HANDLED;
    BwdPro pro{};
    pro.dH_in = dHnew; pro.U = U; pro.Cand = Cand;
#define CBWD_CALL(a, c, d) launch_bwd<a, 1, c, d, PRO_BLEND>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro)
    STC_MF_EPI_CASE(1, CBWD_CALL) STC_MF_EPI_CASE(2, CBWD_CALL) STC_MF_EPI_CASE(4, CBWD_CALL)
#undef CBWD_CALL
    return STC_NOT_HANDLED;
}

int stc_node_bwd_mfma_max_partials() { return MF_BWD_MAX_GRID; }

int stc_node_bwd_mfma(const float* const* Z, int Ks, const float* Tc, int Kc, const float* W, const float* dY,
                      float* const* dZ, float* partial, int* n_partials, int want_db,
                      long long nodes, int C, int L, int Lw, int Ho, hipStream_t stream) {
    const int Cr = C;
    if (C >= 1 && C < 16) C = 16;
    if (!fast_path_shape(Ks, Kc, C, L, Ho, nodes)) return STC_NOT_HANDLED;
    if (!all_aligned16(Z, Ks) || !stc::aligned16(dY)) return STC_NOT_HANDLED;
    for (int n = 0; n < Ks; ++n)
        if (!stc::aligned16(dZ[n])) return STC_NOT_HANDLED;
#define BWD_CALL(a, b, c, d) launch_bwd<a, b, c, d>(Z, Tc, W, dY, dZ, partial, n_partials, want_db, nodes, Lw, stream, BwdPro{}, Cr)
    STC_MF_DISPATCH(BWD_CALL)
#undef BWD_CALL
    return STC_NOT_HANDLED;
}
