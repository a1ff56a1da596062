// bf16-storage node kernels of BDG_Dif (reference STC_GNN.py:38-45 and its autograd) for gfx950: BASELINE.json's
// configuration 5 (N = 50 176, C = 64, bf16 -- "MFMA W_k projection").
//
// Feature slabs Z_n, the output Y and the gradients dY, dZ_n are stored in bf16; W, bias, T_c, dW, db are fp32 (master
// weights and their gradients); every product runs on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  A bf16 feature
// row IS a matrix-core operand: lane (x = lane & 15, g = lane >> 4) loads the 16 bytes holding columns 8g..8g+7 of row x
// of a 16-row block -- one fully coalesced 1 KiB wave load per 16 rows of L = 32 -- and hands that register quad to the
// MFMA as A[x][8g + e].  Nothing is split or converted on the way in.
//
// Register layout of the instruction:  A[x][8g + e]   B[8g + e][x]   D[4g + r][x]  (e = 0..7, r = 0..3).
// An accumulator tile pair (2p, 2p+1) packed to bf16 is an operand whose contraction slot (g, e) stands for row
// pair_row(g, e) = 16 (e >> 2) + 4g + (e & 3) of the 32-row pair; as B it reads P[row][x], as A it reads P^T[x][row].
//
// What is different from the fp32 kernels (stc_node_x3.hip), because stores of single bf16 elements would waste the bus:
//   * forward: the category mix is computed TRANSPOSED, Y^T = sum_c U_c^T T_c (T_0 = I is a table like the others; the
//     projected tiles U_c feed it as A operands straight from their accumulators), so a lane ends up with four consecutive
//     output columns of one row: 8-byte stores, 512 contiguous bytes per wave instruction for Ho = 16;
//   * backward: dY and Z are loaded once, in row layout; the accumulator-layout copies the contractions over the category
//     axis need (dY as Q_0, Z^T for dW) are made by one MFMA against a 0/1 selector (exact: 1.0 * bf16 in fp32, packed
//     back without rounding) instead of strided element loads.
// dW / db accumulate in fp32 registers across all nodes of a wave and are combined in fixed order (bitwise reproducible).
//
// Shapes: C = 32 * NB2 in {32, 64}, Ho = 16 * HB in {16, 32}, L in {16, 32} bf16 per row (Lw <= L real columns, pad columns
// must hold finite values), Ks = Kc = K <= 3.  Everything else: STC_EUNSUPPORTED.
#include "stc_node_frag.h"

namespace {

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
typedef unsigned short bf16_t;          // storage only

struct BPtrs { const bf16_t* p[STC_MAX_K]; };
struct BDPtrs { bf16_t* p[STC_MAX_K]; };

#define kZero4 (f32x4{0.f, 0.f, 0.f, 0.f})
#define kZeroU4 (u32x4{0u, 0u, 0u, 0u})

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {     // v_cvt_pk_bf16_f32 (RNE): a in the low half
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ u32x4 pack8(const f32x4 a, const f32x4 b) {      // slots 0..3 from a, 4..7 from b
    return u32x4{pk_bf16(a[0], a[1]), pk_bf16(a[2], a[3]), pk_bf16(b[0], b[1]), pk_bf16(b[2], b[3])};
}
__device__ __forceinline__ u32x2 pack4(const f32x4 a) { return u32x2{pk_bf16(a[0], a[1]), pk_bf16(a[2], a[3])}; }
__device__ __forceinline__ f32x4 mma(const u32x4 a, const u32x4 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__host__ __device__ constexpr int pair_row(int g, int e) { return 16 * (e >> 2) + 4 * g + (e & 3); }

__device__ __forceinline__ void put_frag(u32x4* tab, int frag, int lane, const float (&v)[8]) {
    tab[frag * 64 + lane] = u32x4{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
}

// columns 8g..8g+7 of row `row` of a (rows, L) bf16 matrix; columns >= L read as zero (L = 16: lanes g >= 2 hold zeros)
template <int L>
__device__ __forceinline__ u32x4 load_row8(const bf16_t* __restrict__ base, size_t row, int g) {
    if (L == 32 || 8 * g < L) return *reinterpret_cast<const u32x4*>(base + row * L + 8 * g);
    return kZeroU4;
}

// 0/1 selector as a B operand: B[slot 8g + e][x] = (8g + e == 16 blk + x).  A row-layout operand times it gives the
// accumulator-layout tile of columns 16 blk .. 16 blk + 15:  D[4g + r][x] = A[4g + r][16 blk + x].
__device__ __forceinline__ u32x4 selector(int blk, int x, int g) {
    u32x4 s = kZeroU4;
    const int e = 16 * blk + x - 8 * g;             // the one slot of this lane that is set, if in 0..7
    if (e >= 0 && e < 8) s[e >> 1] = (e & 1) ? 0x3F800000u : 0x00003F80u;
    return s;
}

// --------------------------------------------------------------------------------------- forward
template <int NB2, int HB, int K, int L>
__global__ __launch_bounds__(MF_THREADS, (NB2 == 1 ? 2 : 1)) void node_fwd_bf16_kernel(
    BPtrs Z, const float* __restrict__ Tc, const float* __restrict__ W, const float* __restrict__ bias,
    bf16_t* __restrict__ Y, int nodes, int Lw) {
    constexpr int NRB = 2 * NB2, C = 32 * NB2, Ho = 16 * HB, NCB = K * HB;
    constexpr int nWx = K * NCB, nTx = K * NB2 * NRB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* Wx = reinterpret_cast<u32x4*>(smem_raw);        // [K n][NCB = (c, hb)]   B: W[(n, c, l = slot)][o = 16 hb + x]
    u32x4* Tx = Wx + nWx * 64;                              // [K c][NB2 p][NRB db]   B: T_c[c' = 32 p + pair_row][d = 16 db + x]
    const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, g = lane >> 4;

    for (int idx = tid; idx < nWx * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, cb = f % NCB, n = f / NCB;
        const int c = cb / HB, o = (cb % HB) * 16 + (ll & 15), gg = ll >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int l = 8 * gg + e;
            v[e] = l < Lw ? W[((size_t)(n * K + c) * Lw + l) * Ho + o] : 0.f;      // pad columns contribute nothing
        }
        put_frag(Wx, f, ll, v);
    }
    for (int idx = tid; idx < nTx * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, db = f % NRB, p = (f / NRB) % NB2, c = f / (NRB * NB2), gg = ll >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int cp = 32 * p + pair_row(gg, e), d = 16 * db + (ll & 15);
            v[e] = c == 0 ? (cp == d ? 1.f : 0.f) : Tc[(size_t)c * C * C + cp * C + d];      // T_0 = I (STC_GNN.py:26)
        }
        put_frag(Tx, f, ll, v);
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = gridDim.x * MF_WAVES;
    f32x4 bv[HB];                                        // bias of the lane's four output columns 16 hb + 4g + r
#pragma unroll
    for (int hb = 0; hb < HB; ++hb)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[hb][r] = bias ? bias[16 * hb + 4 * g + r] : 0.f;

    int node = blockIdx.x * MF_WAVES + wave;
    u32x4 cur[K][NRB], nxt[K][NRB];
    auto load_rows = [&](u32x4 (&z)[K][NRB], int nd) {
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) z[n][rb] = load_row8<L>(Z.p[n], (size_t)nd * C + 16 * rb + x, g);
    };
    if (node < nodes) load_rows(cur, node);
    while (node < nodes) {
        const int next_node = node + nw;
        if (next_node < nodes) load_rows(nxt, next_node);        // software prefetch: lands while this node computes
        __builtin_amdgcn_sched_barrier(0);
        const int lo = opaque(lane);

        // project: U_c[rb][(c, hb)] (rows c' = 16 rb + 4g + r, column o = 16 hb + x) = sum_n Z_n rows (A) . W_{n,c} (B)
        f32x4 acc[NRB][NCB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = kZero4;
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const u32x4 w = Wx[(n * NCB + cb) * 64 + lo];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc[rb][cb] = mma(cur[n][rb], w, acc[rb][cb]);
            }

        // mix, transposed: Y^T[hb][db] (rows o = 16 hb + 4g + r, column d = 16 db + x) = sum_c U_c^T (A, from accumulators) . T_c (B)
        f32x4 yT[HB][NRB];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
#pragma unroll
            for (int db = 0; db < NRB; ++db) yT[hb][db] = bv[hb];
#pragma unroll
        for (int c = 0; c < K; ++c)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int p = 0; p < NB2; ++p) {
                    const u32x4 u = pack8(acc[2 * p][c * HB + hb], acc[2 * p + 1][c * HB + hb]);
#pragma unroll
                    for (int db = 0; db < NRB; ++db) yT[hb][db] = mma(u, Tx[((c * NB2 + p) * NRB + db) * 64 + lo], yT[hb][db]);
                }
#pragma unroll
        for (int db = 0; db < NRB; ++db)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
                *reinterpret_cast<u32x2*>(Y + ((size_t)node * C + 16 * db + x) * Ho + 16 * hb + 4 * g) = pack4(yT[hb][db]);
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) cur[n][rb] = nxt[n][rb];
        node = next_node;
    }
}

// --------------------------------------------------------------------------------------- backward
// Per node, with Q_0 = dY and Q_c = T_c dY (c >= 1):
//   gd (tiles rows d, cols o)     = dY rows . selector                     accumulator-layout copy of dY (exact)
//   Qv_c (rows o, cols c')        = dY^T . T_c^T          A = gd as operand (slots = rows d), B = T_c table
//   dZ_n^T (rows l, cols c')      = sum_c W_{n,c} . Q_c^T A = W table (slots = o; natural order for c = 0, pair order else),
//                                                         B = the dY rows themselves (c = 0) / Qv_c accumulators
//   Qd_c (rows c', cols o)        = T_c . dY              A = the same T_c table, B = gd
//   za (rows c', cols l)          = Z rows . selector                      accumulator-layout copy of Z_n (exact)
//   dW_{n,c} (rows l, cols o)    += Z_n^T . Q_c            A = za as operand (slots = rows c'), B = gd / Qd_c as operands
template <int NB2, int K>
struct BwdWaves { static constexpr int v = (NB2 == 1 && K <= 2) ? 2 : 1; };

template <int NB2, int HB, int K, int L>
__global__ __launch_bounds__(MF_THREADS, (BwdWaves<NB2, K>::v)) void node_bwd_bf16_kernel(
    BPtrs Z, const float* __restrict__ Tc, const float* __restrict__ W, const bf16_t* __restrict__ dY,
    BDPtrs dZ, float* __restrict__ partial, int nodes, int want_db, int Lw) {
    constexpr int NRB = 2 * NB2, C = 32 * NB2, Ho = 16 * HB, LB = (L + 15) / 16;
    constexpr int nTB = (K - 1) * NRB * NB2, nWA = K * LB * K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* TB = reinterpret_cast<u32x4*>(smem_raw);     // [K-1][NRB rb][NB2 p]   T_c[16 rb + x][32 p + pair_row]
    u32x4* WA = TB + nTB * 64;                           // [K n][LB][K c]         A: W[(n, c, 16 lb + x)][o(slot)]
    const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, g = lane >> 4;

    for (int idx = tid; idx < nTB * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, p = f % NB2, rb = (f / NB2) % NRB, c1 = f / (NB2 * NRB), gg = ll >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = Tc[(size_t)(c1 + 1) * C * C + (16 * rb + (ll & 15)) * C + 32 * p + pair_row(gg, e)];
        put_frag(TB, f, ll, v);
    }
    for (int idx = tid; idx < nWA * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, c = f % K, lb = (f / K) % LB, n = f / (K * LB), gg = ll >> 4;
        const int l = 16 * lb + (ll & 15);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int o = c == 0 ? 8 * gg + e : pair_row(gg, e);      // Q_0 comes as dY rows (natural order), Q_c from accumulators
            v[e] = (o < Ho && l < Lw) ? W[((size_t)(n * K + c) * Lw + l) * Ho + o] : 0.f;
        }
        put_frag(WA, f, ll, v);
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = gridDim.x * MF_WAVES;
    u32x4 sel[2];
    sel[0] = selector(0, x, g);
    sel[1] = selector(1, x, g);

    f32x4 dWt[K][LB][K][HB];          // dW tiles: rows l = 16 lb + 4g + r, columns o = 16 hb + x
    float dbp[HB];
#pragma unroll
    for (int n = 0; n < K; ++n)
#pragma unroll
        for (int lb = 0; lb < LB; ++lb)
#pragma unroll
            for (int c = 0; c < K; ++c)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) dWt[n][lb][c][hb] = kZero4;
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) dbp[hb] = 0.f;

    for (int node = blockIdx.x * MF_WAVES + wave; node < nodes; node += nw) {
        const size_t r0 = (size_t)node * C;
        u32x4 dyr[NRB], zr[K][NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) dyr[rb] = load_row8<Ho>(dY, r0 + 16 * rb + x, g);
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) zr[n][rb] = load_row8<L>(Z.p[n], r0 + 16 * rb + x, g);
        const int lo = opaque(lane);

        // ---- dY in accumulator layout (rows d = 16 kb + 4g + r, column o = 16 hb + x), db, and as operands
        u32x4 gd[HB][NB2];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
#pragma unroll
            for (int p = 0; p < NB2; ++p) {
                const f32x4 t0 = mma(dyr[2 * p], sel[hb], kZero4), t1 = mma(dyr[2 * p + 1], sel[hb], kZero4);
                dbp[hb] += ((t0[0] + t0[1]) + (t0[2] + t0[3])) + ((t1[0] + t1[1]) + (t1[2] + t1[3]));
                gd[hb][p] = pack8(t0, t1);
            }

        // ---- B operands of dZ per c: Q_0^T = the dY rows; Q_c^T from the Qv_c tiles (rows o, columns c')
        u32x4 qb[K][NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) qb[0][rb] = dyr[rb];
#pragma unroll
        for (int c1 = 0; c1 < K - 1; ++c1)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                f32x4 Qv[2] = {kZero4, kZero4};
#pragma unroll
                for (int p = 0; p < NB2; ++p) {
                    const u32x4 t = TB[((c1 * NRB + rb) * NB2 + p) * 64 + lo];
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) Qv[hb] = mma(gd[hb][p], t, Qv[hb]);
                }
                qb[c1 + 1][rb] = pack8(Qv[0], Qv[1]);
            }

        // ---- dZ_n^T tiles (rows l, columns c'): four consecutive columns of one gradient row per lane
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int lb = 0; lb < LB; ++lb) {
                f32x4 z[NRB];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) z[rb] = kZero4;
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    const u32x4 w = WA[((n * LB + lb) * K + c) * 64 + lo];
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) z[rb] = mma(w, qb[c][rb], z[rb]);
                }
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
                    *reinterpret_cast<u32x2*>(dZ.p[n] + (r0 + 16 * rb + x) * L + 16 * lb + 4 * g) = pack4(z[rb]);
            }

        // ---- Qd_c tiles (rows c', columns o) as operands: slots = rows c' of the tile pair p
        u32x4 qd[AtLeast1<K - 1>::v][HB][NB2];
#pragma unroll
        for (int c1 = 0; c1 < K - 1; ++c1) {
            f32x4 Qd[NRB][HB];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) Qd[rb][hb] = kZero4;
#pragma unroll
                for (int p = 0; p < NB2; ++p) {
                    const u32x4 t = TB[((c1 * NRB + rb) * NB2 + p) * 64 + lo];
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) Qd[rb][hb] = mma(t, gd[hb][p], Qd[rb][hb]);
                }
            }
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int p = 0; p < NB2; ++p) qd[c1][hb][p] = pack8(Qd[2 * p][hb], Qd[2 * p + 1][hb]);
        }

        // ---- dW_{n,c} tile (rows l, columns o) += Z_n^T . Q_c
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int lb = 0; lb < LB; ++lb)
#pragma unroll
                for (int p = 0; p < NB2; ++p) {
                    const u32x4 a = pack8(mma(zr[n][2 * p], sel[lb], kZero4), mma(zr[n][2 * p + 1], sel[lb], kZero4));
#pragma unroll
                    for (int c = 0; c < K; ++c)
#pragma unroll
                        for (int hb = 0; hb < HB; ++hb)
                            dWt[n][lb][c][hb] = mma(a, c == 0 ? gd[hb][p] : qd[c > 0 ? c - 1 : 0][hb][p], dWt[n][lb][c][hb]);
                }
    }
    combine_dw<K, LB, HB>(reinterpret_cast<float*>(smem_raw), dWt, dbp, partial, Lw, want_db);
}

// --------------------------------------------------------------------------------------- host side
template <int NB2, int HB, int K, int L>
int launch_fwd(const void* const* Z, const float* Tc, const float* W, const float* bias, void* Y, long long nodes, int Lw, hipStream_t stream) {
    constexpr int NRB = 2 * NB2, NCB = K * HB;
    const size_t lds = (size_t)(K * NCB + K * NB2 * NRB) * 64 * 16;
    auto kern = node_fwd_bf16_kernel<NB2, HB, K, L>;
    if (int rc = stc::hip_status(stc::allow_lds(kern, lds), "hipFuncSetAttribute(node fwd bf16)")) return rc;
    static const int resident = stc::resident_blocks(kern, MF_THREADS, lds, 2);   // persistent grid = what fits at once
    BPtrs zp{};
    for (int n = 0; n < K; ++n) zp.p[n] = static_cast<const bf16_t*>(Z[n]);
    const long long want = (nodes + MF_WAVES - 1) / MF_WAVES;
    const int grid = (int)(want < resident ? want : resident);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MF_THREADS), lds, stream, zp, Tc, W, bias, static_cast<bf16_t*>(Y), (int)nodes, Lw);
    STC_LAUNCH_CHECK("node_fwd_bf16 launch");
    return STC_OK;
}

template <int NB2, int HB, int K, int L>
int launch_bwd(const void* const* Z, const float* Tc, const float* W, const void* dY, void* const* dZ,
               float* partial, int* n_partials, int want_db, long long nodes, int Lw, hipStream_t stream) {
    constexpr int NRB = 2 * NB2, Ho = 16 * HB, LB = (L + 15) / 16, nW = K * K * L * Ho;
    const size_t frag = (size_t)((K - 1) * NRB * NB2 + K * LB * K) * 64 * 16;
    const size_t slabs = (size_t)MF_WAVES * (nW + Ho) * sizeof(float);
    const size_t lds = frag > slabs ? frag : slabs;
    if (lds > stc::kMaxLdsBytes) return stc::fail(STC_EUNSUPPORTED, "stc_bdg_node_bwd_bf16: %zu B of LDS for the dW combine exceed the CU", lds);
    auto kern = node_bwd_bf16_kernel<NB2, HB, K, L>;
    if (int rc = stc::hip_status(stc::allow_lds(kern, lds), "hipFuncSetAttribute(node bwd bf16)")) return rc;
    static const int resident = stc::resident_blocks(kern, MF_THREADS, lds, BwdWaves<NB2, K>::v);
    BPtrs zp{};
    BDPtrs dzp{};
    for (int n = 0; n < K; ++n) { zp.p[n] = static_cast<const bf16_t*>(Z[n]); dzp.p[n] = static_cast<bf16_t*>(dZ[n]); }
    const long long want = (nodes + MF_WAVES - 1) / MF_WAVES;
    int grid = resident < MF_BWD_MAX_GRID ? resident : MF_BWD_MAX_GRID;
    if (want < grid) grid = (int)want;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MF_THREADS), lds, stream, zp, Tc, W, static_cast<const bf16_t*>(dY), dzp, partial, (int)nodes, want_db, Lw);
    STC_LAUNCH_CHECK("node_bwd_bf16 launch");
    *n_partials = grid;
    return STC_OK;
}

bool bf16_shape(int Ks, int Kc, int C, int L, int Lw, int Ho, long long nodes) {
    return Ks == Kc && Ks >= 1 && Ks <= 3 && (C == 32 || C == 64) && (Ho == 16 || Ho == 32) && (L == 16 || L == 32) &&
           Lw >= 1 && Lw <= L && nodes >= 0 && nodes < (1ll << 31) / C;
}

}  // namespace

#define STC_BF16_CASES(NB2_, HB_, CALL)                                                                 \
    if (C == 32 * NB2_ && Ho == 16 * HB_) {                                                             \
        if (Ks == 1 && L == 16) return CALL(NB2_, HB_, 1, 16);                                          \
        if (Ks == 1 && L == 32) return CALL(NB2_, HB_, 1, 32);                                          \
        if (Ks == 2 && L == 16) return CALL(NB2_, HB_, 2, 16);                                          \
        if (Ks == 2 && L == 32) return CALL(NB2_, HB_, 2, 32);                                          \
        if (Ks == 3 && L == 16) return CALL(NB2_, HB_, 3, 16);                                          \
        if (Ks == 3 && L == 32) return CALL(NB2_, HB_, 3, 32);                                          \
    }
#define STC_BF16_DISPATCH(CALL) STC_BF16_CASES(1, 1, CALL) STC_BF16_CASES(1, 2, CALL) STC_BF16_CASES(2, 1, CALL) STC_BF16_CASES(2, 2, CALL)

extern "C" int stc_bdg_node_bf16_supported(int32_t Ks, int32_t Kc, int32_t C, int32_t L, int32_t Ho) {
    return bf16_shape(Ks, Kc, C, L, L, Ho, 0) ? 1 : 0;
}

extern "C" int stc_bdg_node_fwd_bf16(const void* const* Z, int32_t Ks, const float* Tc, int32_t Kc,
                                     const float* W, const float* bias, void* Y,
                                     int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t Ho, void* stream) {
    if (!bf16_shape(Ks, Kc, C, L, Lw, Ho, nodes))
        return stc::fail(STC_EUNSUPPORTED, "stc_bdg_node_fwd_bf16: shape Ks=%d Kc=%d C=%d L=%d Lw=%d Ho=%d not on the bf16 path", Ks, Kc, C, L, Lw, Ho);
    if (nodes == 0) return STC_OK;
    STC_REQUIRE(Z && W && Y && (Kc == 1 || Tc), STC_EINVAL, "stc_bdg_node_fwd_bf16: null Z/W/Y/Tc");
    for (int n = 0; n < Ks; ++n)
        STC_REQUIRE(Z[n] && stc::aligned16(Z[n]), STC_EALIGN, "stc_bdg_node_fwd_bf16: Z[%d] null or not 16-byte aligned", n);
    STC_REQUIRE(stc::aligned16(Y), STC_EALIGN, "stc_bdg_node_fwd_bf16: Y not 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
#define FWD_CALL(a, b, c, d) launch_fwd<a, b, c, d>(Z, Tc, W, bias, Y, nodes, Lw, s)
    STC_BF16_DISPATCH(FWD_CALL)
#undef FWD_CALL
    return stc::fail(STC_EUNSUPPORTED, "stc_bdg_node_fwd_bf16: no kernel for this shape");
}

extern "C" int stc_bdg_node_bwd_bf16(const void* const* Z, int32_t Ks, const float* Tc, int32_t Kc,
                                     const float* W, const void* dY,
                                     void* const* dZ, float* dW, float* db,
                                     void* workspace, size_t workspace_bytes,
                                     int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t Ho, void* stream) {
    if (!bf16_shape(Ks, Kc, C, L, Lw, Ho, nodes))
        return stc::fail(STC_EUNSUPPORTED, "stc_bdg_node_bwd_bf16: shape Ks=%d Kc=%d C=%d L=%d Lw=%d Ho=%d not on the bf16 path", Ks, Kc, C, L, Lw, Ho);
    STC_REQUIRE(W && dW && (Kc == 1 || Tc), STC_EINVAL, "stc_bdg_node_bwd_bf16: null W/dW/Tc");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nW = Ks * Kc * Lw * Ho;
    if (nodes == 0) {
        if (int rc = stc::hip_status(hipMemsetAsync(dW, 0, (size_t)nW * sizeof(float), s), "memset dW")) return rc;
        if (db) if (int rc = stc::hip_status(hipMemsetAsync(db, 0, (size_t)Ho * sizeof(float), s), "memset db")) return rc;
        return STC_OK;
    }
    STC_REQUIRE(Z && dY && dZ, STC_EINVAL, "stc_bdg_node_bwd_bf16: null Z/dY/dZ");
    for (int n = 0; n < Ks; ++n)
        STC_REQUIRE(Z[n] && dZ[n] && stc::aligned16(Z[n]) && stc::aligned16(dZ[n]), STC_EALIGN,
                    "stc_bdg_node_bwd_bf16: Z[%d] / dZ[%d] null or not 16-byte aligned", n, n);
    STC_REQUIRE(stc::aligned16(dY), STC_EALIGN, "stc_bdg_node_bwd_bf16: dY not 16-byte aligned");
    STC_REQUIRE(workspace && stc::aligned16(workspace), STC_EALIGN, "stc_bdg_node_bwd_bf16: workspace null or not 16-byte aligned");
    STC_REQUIRE(workspace_bytes >= stc_bdg_node_bwd_workspace_bytes(Ks, Kc, C, L, Ho, 0), STC_EINVAL,
                "stc_bdg_node_bwd_bf16: workspace of %zu B is too small", workspace_bytes);
    float* partial = static_cast<float*>(workspace);
    int n_parts = 0;
    auto run = [&]() -> int {
#define BWD_CALL(a, b, c, d) launch_bwd<a, b, c, d>(Z, Tc, W, dY, dZ, partial, &n_parts, db != nullptr, nodes, Lw, s)
        STC_BF16_DISPATCH(BWD_CALL)
#undef BWD_CALL
        return stc::fail(STC_EUNSUPPORTED, "stc_bdg_node_bwd_bf16: no kernel for this shape");
    };
    if (int rc = run()) return rc;
    return stc_node_reduce_partials(partial, n_parts, nW, Ho, dW, db, s);
}
