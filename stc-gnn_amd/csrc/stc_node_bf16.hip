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

#include <cstdlib>

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

// --------------------------------------------------------------------------------------- planar cell: gates forward
// The gates convolution of an STC_Cell (reference STC_GNN.py:68-75) on planar bf16 inputs, K = 2: the [Xt | H] row is read
// as two (nodes, C, 16) planes -- lanes g < 2 take 8 columns of the X plane, g >= 2 of the H plane, which together ARE the
// 32-slot operand -- once for the cell's own rows (X, H) and once for their aggregations (SX = S.X, SH = S.H).  The
// transposed mix leaves lane (x, g) with columns 4g..4g+3 of BOTH gates of category row x, so the epilogue is element-wise
// in registers: U = sigmoid(.), R = sigmoid(.), RH = R * H, each an 8-byte store.
// NARROW (layer 0): the input plane has cin = Lw - 16 <= 4 columns; the slab is read as [state (16) | input (cin) | 0] and
// W's rows are permuted to match (stc_wrow_swapped), so the reference's state_dict layout is untouched.
// POST: the candidate convolution's projection in post-aggregation form (A = sum_c T_c^T ([Xt | RH] Wc_{0,c}) + bc, Bm
// likewise with Wc_{1,c}; the caller finishes Y = A + S.Bm) runs as a second stage of the same launch: Xt is still in the
// row registers and RH is moved from the epilogue's layout to row layout with four ds_bpermute per 16 rows.
__device__ __forceinline__ float fast_sigmoid(float v) { return stc_sigmoid(v); }      // hardware exp2 / rcp (stc_common.h)
__device__ __forceinline__ float fast_tanh(float v) { return stc_tanh(v); }
__device__ __forceinline__ f32x4 unpack4(const u32x2 v) {
    return f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u)};
}
// 8 columns of one row of a planar slab: wide = [P0 (16) | P1 (16)]; narrow = [P1 (16) | P0 (cin) | 0]
// ONCE: the planes are streamed once by this launch (non-temporal policy, stc_common.h); false where the wave reads a plane again (the
// gates forward takes H a second time for its epilogue: with the policy on the first read that one missed -- 842 -> 918 us)
template <int NARROW, bool ONCE = true>
__device__ __forceinline__ u32x4 load_planar8(const bf16_t* __restrict__ P0, const bf16_t* __restrict__ P1, size_t row, int g, int cin) {
    if (!NARROW) {
        const u32x4* p = reinterpret_cast<const u32x4*>((g < 2 ? P0 : P1) + row * 16 + 8 * (g & 1));
        return ONCE ? stc_ld_once(p) : *p;
    }
    // branch-free: every lane issues the same five loads (valid addresses, cache hits for the lanes that discard them) and
    // selects afterwards -- divergent, cin-dependent branches around 2-byte loads serialised their latencies (the narrow
    // kernels ran 1.4-1.6x slower than the wide ones on fewer bytes)
    const u32x4* ps = reinterpret_cast<const u32x4*>(P1 + row * 16 + 8 * (g & 1));
    const u32x4 st = ONCE ? stc_ld_once(ps) : *ps;
    const bf16_t* q = P0 + row * cin;
    const unsigned v0 = q[0], v1 = q[cin > 1 ? 1 : 0], v2 = q[cin > 2 ? 2 : 0], v3 = q[cin > 3 ? 3 : 0];
    const u32x4 in = {v0 | (cin > 1 ? v1 << 16 : 0u), cin > 2 ? (v2 | (cin > 3 ? v3 << 16 : 0u)) : 0u, 0u, 0u};
    return g < 2 ? st : (g == 2 ? in : kZeroU4);
}

struct GatesFwd {
    const bf16_t *X, *H, *SX, *SH;     // planes: X / SX (nodes, C, cin), H / SH (nodes, C, 16)
    bf16_t *U, *R, *RH;                // out (nodes, C, 16)
    const float *Wc, *bc;              // POST: candidate weights (4 Lw, 16), bias (16) or null
    bf16_t *A, *Bm;                    // POST out (nodes, C, 16)
};

template <int NB2, int NARROW, int POST>
__global__ __launch_bounds__(MF_THREADS, (NB2 == 1 ? 2 : 1)) void cell_gates_fwd_bf16_kernel(
    GatesFwd a, const float* __restrict__ Tc, const float* __restrict__ W, const float* __restrict__ bias, int nodes, int Lw) {
    constexpr int K = 2, HB = 2, NRB = 2 * NB2, C = 32 * NB2, NCB = K * HB;
    constexpr int nWx = K * NCB, nTx = K * NB2 * NRB, nWp = POST ? K * K : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* Wx = reinterpret_cast<u32x4*>(smem_raw);        // [K n][NCB = (c, hb)]   B: W[(n, c, l = slot)][o = 16 hb + x]
    u32x4* Tx = Wx + nWx * 64;                              // [K c][NB2 p][NRB db]   B: T_c[c' = 32 p + pair_row][d = 16 db + x]
    u32x4* Wp = Tx + nTx * 64;                              // POST: [K n][K c]       B: Wc[(n, c, l = slot)][o = x]
    const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, g = lane >> 4;
    const int cin = Lw - 16;

    for (int idx = tid; idx < (nWx + nWp) * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, gg = ll >> 4;
        const bool post = f >= nWx;
        const int fp = f - nWx;
        const int n = post ? fp / K : f / NCB, c = post ? fp % K : (f % NCB) / HB;
        const int Ho = post ? 16 : 32, o = (post ? 0 : ((f % NCB) % HB) * 16) + (ll & 15);
        const float* Wsrc = post ? a.Wc : W;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int l = 8 * gg + e;
            const int wl = NARROW ? stc_wrow_swapped(l, cin) : l;
            v[e] = (wl >= 0 && wl < Lw) ? Wsrc[((size_t)(n * K + c) * Lw + wl) * Ho + o] : 0.f;
        }
        put_frag(Wx, f < nWx ? f : nWx + nTx + fp, ll, v);
    }
    for (int idx = tid; idx < nTx * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, db = f % NRB, p = (f / NRB) % NB2, c = f / (NRB * NB2), gg = ll >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int cp = 32 * p + pair_row(gg, e), d = 16 * db + (ll & 15);
            v[e] = c == 0 ? (cp == d ? 1.f : 0.f) : Tc[(size_t)c * C * C + cp * C + d];
        }
        put_frag(Tx, f, ll, v);
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = gridDim.x * MF_WAVES;
    f32x4 bv[HB], bcv = kZero4;
#pragma unroll
    for (int hb = 0; hb < HB; ++hb)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[hb][r] = bias ? bias[16 * hb + 4 * g + r] : 0.f;
    if (POST && a.bc)
#pragma unroll
        for (int r = 0; r < 4; ++r) bcv[r] = a.bc[4 * g + r];

    int node = blockIdx.x * MF_WAVES + wave;
    u32x4 cur[K][NRB], nxt[K][NRB];
    auto load_rows = [&](u32x4 (&z)[K][NRB], int nd) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const size_t row = (size_t)nd * C + 16 * rb + x;
            z[0][rb] = load_planar8<NARROW, false>(a.X, a.H, row, g, cin);     // (H comes again below, for the epilogue)
            z[1][rb] = load_planar8<NARROW>(a.SX, a.SH, row, g, cin);
        }
    };
    if (node < nodes) load_rows(cur, node);
    while (node < nodes) {
        const int next_node = node + nw;
        if (next_node < nodes) load_rows(nxt, next_node);
        u32x2 hq[NRB];                                          // H in the epilogue's layout: row 16 db + x, columns 4g..4g+3
#pragma unroll
        for (int db = 0; db < NRB; ++db) hq[db] = stc_ld_once(reinterpret_cast<const u32x2*>(a.H + ((size_t)node * C + 16 * db + x) * 16 + 4 * g));
        __builtin_amdgcn_sched_barrier(0);
        const int lo = opaque(lane);

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
        u32x2 rhp[NRB];
#pragma unroll
        for (int db = 0; db < NRB; ++db) {
            const size_t e = ((size_t)node * C + 16 * db + x) * 16 + 4 * g;
            const f32x4 hh = unpack4(hq[db]);
            f32x4 u, rg, rh;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                u[r] = fast_sigmoid(yT[0][db][r]);
                rg[r] = fast_sigmoid(yT[1][db][r]);
                rh[r] = rg[r] * hh[r];
            }
            rhp[db] = pack4(rh);
            stc_st_once(reinterpret_cast<u32x2*>(a.U + e), pack4(u));
            stc_st_once(reinterpret_cast<u32x2*>(a.R + e), pack4(rg));
            if (a.RH) stc_st_once(reinterpret_cast<u32x2*>(a.RH + e), rhp[db]);      // (optional with POST: the one-launch backward re-forms R*H)
        }
        if constexpr (POST) {
            // the candidate's input row [Xt | RH] (narrow: [RH | x | 0]) as an operand: RH columns 8j..8j+7 of row x sit with lanes (x, 2j), (x, 2j+1)
            const int j = NARROW ? (g & 1) : (g - 2 < 0 ? 0 : g - 2);
            const int s0 = 4 * (x + 32 * j), s1 = s0 + 64;          // byte addresses of the two source lanes
            u32x4 zc[NRB];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                const u32x4 rr = {(unsigned)__builtin_amdgcn_ds_bpermute(s0, (int)rhp[rb][0]), (unsigned)__builtin_amdgcn_ds_bpermute(s0, (int)rhp[rb][1]),
                                  (unsigned)__builtin_amdgcn_ds_bpermute(s1, (int)rhp[rb][0]), (unsigned)__builtin_amdgcn_ds_bpermute(s1, (int)rhp[rb][1])};
                zc[rb] = (NARROW ? g < 2 : g >= 2) ? rr : cur[0][rb];
            }
            f32x4 pa[K][NRB][K];
#pragma unroll
            for (int n = 0; n < K; ++n)
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    const u32x4 w = Wp[(n * K + c) * 64 + lo];
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) pa[n][rb][c] = mma(zc[rb], w, kZero4);
                }
#pragma unroll
            for (int n = 0; n < K; ++n) {
                f32x4 yA[NRB];
#pragma unroll
                for (int db = 0; db < NRB; ++db) yA[db] = n == 0 ? bcv : kZero4;
#pragma unroll
                for (int c = 0; c < K; ++c)
#pragma unroll
                    for (int p = 0; p < NB2; ++p) {
                        const u32x4 u = pack8(pa[n][2 * p][c], pa[n][2 * p + 1][c]);
#pragma unroll
                        for (int db = 0; db < NRB; ++db) yA[db] = mma(u, Tx[((c * NB2 + p) * NRB + db) * 64 + lo], yA[db]);
                    }
#pragma unroll
                for (int db = 0; db < NRB; ++db)
                    stc_st_once(reinterpret_cast<u32x2*>((n == 0 ? a.A : a.Bm) + ((size_t)node * C + 16 * db + x) * 16 + 4 * g), pack4(yA[db]));
            }
        }
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

// PL = 1: planar slabs (Z.p[n] = columns 0..15, Z.q[n] = columns 16..31 of slab n, (nodes, C, 16) planes each) and planar
// gradients (dZ.p[n], dZ.q[n] likewise).  PL = 2: planar with a narrow input plane -- Z.q[n] = the 16-wide state plane,
// Z.p[n] = the (nodes, C, cin) input plane, slab columns [state | input | 0], W rows permuted; only dZ.q[n] is produced.
// PRO = 1 (planar gates convolution of an STC_Cell, Ho = 32): dY is not read but formed per row from the GRU's saved planes
// (autograd of STC_GNN.py:71-78):  dY = [dHnew (Cand - H) U (1-U) | dRH H R (1-R)],  and the by-product
// dH = dRH R + dHnew (1-U) -- what the previous state is owed outside the convolution -- is written on the way.
struct GatesPro { const bf16_t *dRH, *Cand, *H, *U, *R, *dHnew; bf16_t* dH; };
struct BPtrs2 { const bf16_t* p[STC_MAX_K]; const bf16_t* q[STC_MAX_K]; };
struct BDPtrs2 { bf16_t* p[STC_MAX_K]; bf16_t* q[STC_MAX_K]; };

__device__ __forceinline__ void unpack8(const u32x4 v, float (&r)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[2 * i] = __uint_as_float(v[i] << 16); r[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u); }
}
__device__ __forceinline__ u32x4 pack8f(const float (&r)[8]) {
    return u32x4{pk_bf16(r[0], r[1]), pk_bf16(r[2], r[3]), pk_bf16(r[4], r[5]), pk_bf16(r[6], r[7])};
}

template <int NB2, int HB, int K, int L, int PL = 0, int PRO = 0, int WAVES = BwdWaves<NB2, K>::v>
__global__ __launch_bounds__(MF_THREADS, WAVES) void node_bwd_bf16_kernel(
    BPtrs2 Z, const float* __restrict__ Tc, const float* __restrict__ W, const bf16_t* __restrict__ dY,
    BDPtrs2 dZ, float* __restrict__ partial, int nodes, int want_db, int Lw, GatesPro pro) {
    static_assert(!PL || (L == 32 && K == 2), "planar slabs are 16 + 16 columns, K = 2");
    static_assert(!PRO || (PL && HB == 2), "the gates prologue belongs to the planar gates convolution");
    constexpr int NRB = 2 * NB2, C = 32 * NB2, Ho = 16 * HB, LB = (L + 15) / 16;
    const int cin = Lw - 16;                            // PL = 2: width of the narrow input plane
    constexpr int nTB = (K - 1) * NRB * NB2, nWA = K * LB * K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* TB = reinterpret_cast<u32x4*>(smem_raw);     // [K-1][NRB rb][NB2 p]   T_c[16 rb + x][32 p + pair_row]
    u32x4* WA = TB + nTB * 64;                           // [K n][LB][K c]         A: W[(n, c, 16 lb + x)][o(slot)]
    const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, g = lane >> 4;
    // PRO, dH == null: the state's share dRH R + dHnew (1-U) is parked here by the prologue ([wave][NRB][16 rows][16 columns] fp32,
    // written by the lanes that computed it, read back by the lanes that own those columns of the H plane's gradient tile: same
    // wave, program order) instead of being re-formed at the tile's store from four more loads per row block -- at one wave per SIMD
    // (C = 64) those late loads went back to HBM (8.7 GB per launch against 6.7 GB algorithmic)
    float* stash = reinterpret_cast<float*>(WA + nWA * 64) + (size_t)(tid >> 6) * NRB * 256;

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
            const int wl = PL == 2 ? stc_wrow_swapped(l, cin) : l;
            v[e] = (o < Ho && wl >= 0 && wl < Lw) ? W[((size_t)(n * K + c) * Lw + wl) * Ho + o] : 0.f;
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

    // One wave per SIMD (C = 64 with the prologue: > 256 registers) has no partner wave to hide the loads behind, so the next
    // node's slabs and prologue planes are requested before this node's arithmetic starts (software prefetch, raw registers).
    constexpr bool PF = PRO != 0 && WAVES == 1;
    u32x4 zr[K][NRB], znx[PF ? K : 1][PF ? NRB : 1];
    u32x4 pr[PRO ? 6 : 1][PRO ? NRB : 1], pnx[PF ? 6 : 1][PF ? NRB : 1];
    auto fetch_z = [&](auto& z, int nd) {
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                if constexpr (PL != 0) z[n][rb] = load_planar8<PL == 2>(Z.p[n], Z.q[n], (size_t)nd * C + 16 * rb + x, g, cin);
                else z[n][rb] = load_row8<L>(Z.p[n], (size_t)nd * C + 16 * rb + x, g);
            }
    };
    auto fetch_pro = [&](auto& q, int nd) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const size_t e = ((size_t)nd * C + 16 * rb + x) * 16 + 8 * (g & 1);       // lanes g and g + 2 read the same 8 columns
            q[0][rb] = stc_ld_once(reinterpret_cast<const u32x4*>(pro.dHnew + e));
            q[1][rb] = stc_ld_once(reinterpret_cast<const u32x4*>(pro.Cand + e));
            q[2][rb] = stc_ld_once(reinterpret_cast<const u32x4*>(pro.H + e));
            q[3][rb] = stc_ld_once(reinterpret_cast<const u32x4*>(pro.U + e));
            q[4][rb] = stc_ld_once(reinterpret_cast<const u32x4*>(pro.dRH + e));
            q[5][rb] = stc_ld_once(reinterpret_cast<const u32x4*>(pro.R + e));
        }
    };
    int node = blockIdx.x * MF_WAVES + wave;
    if constexpr (PF) { if (node < nodes) { fetch_z(zr, node); fetch_pro(pr, node); } }
    for (; node < nodes; node += nw) {
        const size_t r0 = (size_t)node * C;
        u32x4 dyr[NRB];
        if constexpr (!PF) {
            fetch_z(zr, node);
            if constexpr (PRO != 0) fetch_pro(pr, node);
        } else {
            if (node + nw < nodes) { fetch_z(znx, node + nw); fetch_pro(pnx, node + nw); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PRO != 0) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                float gn[8], cd[8], hh[8], uu[8], dr[8], rr[8], gy[8], dh[8];
                unpack8(pr[0][rb], gn); unpack8(pr[1][rb], cd); unpack8(pr[2][rb], hh);
                unpack8(pr[3][rb], uu); unpack8(pr[4][rb], dr); unpack8(pr[5][rb], rr);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float gu = gn[i] * (cd[i] - hh[i]) * uu[i] * (1.f - uu[i]);
                    const float gr = dr[i] * hh[i] * rr[i] * (1.f - rr[i]);
                    gy[i] = g < 2 ? gu : gr;
                    dh[i] = fmaf(dr[i], rr[i], gn[i] * (1.f - uu[i]));
                }
                dyr[rb] = pack8f(gy);
                if (g < 2) {
                    if (pro.dH) {
                        stc_st_once(reinterpret_cast<u32x4*>(pro.dH + (r0 + 16 * rb + x) * 16 + 8 * g), pack8f(dh));
                    } else {                                    // park the share: row x, columns 8g .. 8g+7 of block rb
                        float4* slot = reinterpret_cast<float4*>(stash + (rb * 16 + x) * 16 + 8 * g);
                        slot[0] = make_float4(dh[0], dh[1], dh[2], dh[3]);
                        slot[1] = make_float4(dh[4], dh[5], dh[6], dh[7]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) dyr[rb] = load_row8<Ho>(dY, r0 + 16 * rb + x, g);
        }
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
                if constexpr (PRO != 0) {
                    // dH == null: the previous state's other share, dRH R + dHnew (1-U), is added to its H-plane gradient here, in
                    // the tile's own layout (row x, columns 4g..4g+3: four 8-byte loads of lines the prologue just brought in), so
                    // the state's gradient gets ONE plane from this cell instead of two
                    if (n == 0 && lb == (PL == 1 ? 1 : 0) && !pro.dH) {
#pragma unroll
                        for (int rb = 0; rb < NRB; ++rb) {
                            const float4 sh = *reinterpret_cast<const float4*>(stash + (rb * 16 + x) * 16 + 4 * g);
                            z[rb][0] += sh.x; z[rb][1] += sh.y; z[rb][2] += sh.z; z[rb][3] += sh.w;
                        }
                    }
                }
                if constexpr (PL == 1) {                        // planar gradient slabs: block lb of the row goes to plane lb
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb)
                        stc_st_once(reinterpret_cast<u32x2*>((lb == 0 ? dZ.p[n] : dZ.q[n]) + (r0 + 16 * rb + x) * 16 + 4 * g), pack4(z[rb]));
                } else if constexpr (PL == 2) {                 // narrow input plane: only the state plane's gradient is wanted
                    if (lb == 0) {
#pragma unroll
                        for (int rb = 0; rb < NRB; ++rb)
                            stc_st_once(reinterpret_cast<u32x2*>(dZ.q[n] + (r0 + 16 * rb + x) * 16 + 4 * g), pack4(z[rb]));
                    }
                } else {
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb)
                        *reinterpret_cast<u32x2*>(dZ.p[n] + (r0 + 16 * rb + x) * L + 16 * lb + 4 * g) = pack4(z[rb]);
                }
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
        if constexpr (PF) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
                for (int n = 0; n < K; ++n) zr[n][rb] = znx[n][rb];
#pragma unroll
                for (int q = 0; q < 6; ++q) pr[q][rb] = pnx[q][rb];
            }
        }
    }
    combine_dw<K, LB, HB>(reinterpret_cast<float*>(smem_raw), dWt, dbp, partial, Lw, want_db, PL == 2 ? cin : -1);
}

// --------------------------------------------------------------------------------------- post-aggregation backward (K = 2, Ho = 16)
// Candidate convolution in the form Y = A + S.Bm (see stc_node_x3.hip): from the input row [X | X2] (planar) and the two
// gradients dA = dY, dB = S^T dY, produce the input's gradient and dW / db in one pass:
//   per weight set n (dY_0 = dA, dY_1 = dB):  Q^n_0 = dY_n,  Q^n_1 = T_1 dY_n
//   d[X | X2]^T (rows l, cols c') = sum_n sum_c W_{n,c} (Q^n_c)^T        dW_{n,c} (rows l, cols o) += [X | X2]^T Q^n_c
// bf16-native: the row [dA | dB] (lanes g < 2 take dA, g >= 2 take dB) is one 32-slot operand -- "Ho = 32 with hb = n" --
// so the structure is the generic backward's with the weight set in the place of the column block.
template <int NB2, int NARROW>
__global__ __launch_bounds__(MF_THREADS, (NB2 == 1 ? 2 : 1)) void node_post_bwd_bf16_kernel(
    const bf16_t* __restrict__ X, const bf16_t* __restrict__ X2, const float* __restrict__ Tc, const float* __restrict__ W,
    const bf16_t* __restrict__ dA, const bf16_t* __restrict__ dB, bf16_t* __restrict__ dX, bf16_t* __restrict__ dX2,
    float* __restrict__ partial, int nodes, int want_db, int Lw) {
    constexpr int K = 2, NRB = 2 * NB2, C = 32 * NB2, LB = 2;
    constexpr int nTB = NRB * NB2, nWA = LB * K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* TB = reinterpret_cast<u32x4*>(smem_raw);     // [NRB rb][NB2 p]   T_1[16 rb + x][32 p + pair_row]
    u32x4* WA = TB + nTB * 64;                           // [LB][K c]         A: W[(n(slot), c, 16 lb + x)][o(slot)]
    const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, g = lane >> 4;
    const int cin = Lw - 16;

    for (int idx = tid; idx < nTB * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, p = f % NB2, rb = f / NB2, gg = ll >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = Tc[(size_t)C * C + (16 * rb + (ll & 15)) * C + 32 * p + pair_row(gg, e)];
        put_frag(TB, f, ll, v);
    }
    for (int idx = tid; idx < nWA * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, c = f % K, lb = f / K, gg = ll >> 4;
        const int l = 16 * lb + (ll & 15);
        const int wl = NARROW ? stc_wrow_swapped(l, cin) : l;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = c == 0 ? gg >> 1 : e >> 2;                         // the [dA | dB] row: natural order; accumulators: pair order
            const int o = c == 0 ? 8 * (gg & 1) + e : 4 * gg + (e & 3);
            v[e] = (wl >= 0 && wl < Lw) ? W[((size_t)(n * K + c) * Lw + wl) * 16 + o] : 0.f;
        }
        put_frag(WA, f, ll, v);
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = gridDim.x * MF_WAVES;
    u32x4 sel[2];
    sel[0] = selector(0, x, g);
    sel[1] = selector(1, x, g);
    f32x4 dWt[K][LB][K][1];
    float dbp[1] = {0.f};
#pragma unroll
    for (int n = 0; n < K; ++n)
#pragma unroll
        for (int lb = 0; lb < LB; ++lb)
#pragma unroll
            for (int c = 0; c < K; ++c) dWt[n][lb][c][0] = kZero4;

    for (int node = blockIdx.x * MF_WAVES + wave; node < nodes; node += nw) {
        const size_t r0 = (size_t)node * C;
        u32x4 dyr[NRB], zr[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            dyr[rb] = load_planar8<0>(dA, dB, r0 + 16 * rb + x, g, 16);
            zr[rb] = NARROW ? load_planar8<1>(X2, X, r0 + 16 * rb + x, g, cin) : load_planar8<0>(X, X2, r0 + 16 * rb + x, g, 16);
        }
        const int lo = opaque(lane);

        u32x4 gd[K][NB2];                                       // dY_n in accumulator layout (rows d, column o = x) as operands
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int p = 0; p < NB2; ++p) {
                const f32x4 t0 = mma(dyr[2 * p], sel[n], kZero4), t1 = mma(dyr[2 * p + 1], sel[n], kZero4);
                if (n == 0) dbp[0] += ((t0[0] + t0[1]) + (t0[2] + t0[3])) + ((t1[0] + t1[1]) + (t1[2] + t1[3]));      // the bias sits on A only
                gd[n][p] = pack8(t0, t1);
            }
        u32x4 qb1[NRB];                                         // (T_1 dY_n)^T, both n, as the B operand of the c = 1 step
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            f32x4 Qv[K] = {kZero4, kZero4};
#pragma unroll
            for (int p = 0; p < NB2; ++p) {
                const u32x4 t = TB[(rb * NB2 + p) * 64 + lo];
#pragma unroll
                for (int n = 0; n < K; ++n) Qv[n] = mma(gd[n][p], t, Qv[n]);
            }
            qb1[rb] = pack8(Qv[0], Qv[1]);
        }
#pragma unroll
        for (int lb = 0; lb < LB; ++lb) {
            if (NARROW && lb == 1) continue;                    // the narrow input plane needs no gradient
            const u32x4 w0 = WA[(lb * K + 0) * 64 + lo], w1 = WA[(lb * K + 1) * 64 + lo];
            bf16_t* dst = lb == 0 ? dX : dX2;
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                const f32x4 z = mma(w1, qb1[rb], mma(w0, dyr[rb], kZero4));
                stc_st_once(reinterpret_cast<u32x2*>(dst + (r0 + 16 * rb + x) * 16 + 4 * g), pack4(z));
            }
        }
        u32x4 qd[K][NB2];                                       // T_1 dY_n (rows c', columns o) as operands
#pragma unroll
        for (int n = 0; n < K; ++n) {
            f32x4 Qd[NRB];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                Qd[rb] = kZero4;
#pragma unroll
                for (int p = 0; p < NB2; ++p) Qd[rb] = mma(TB[(rb * NB2 + p) * 64 + lo], gd[n][p], Qd[rb]);
            }
#pragma unroll
            for (int p = 0; p < NB2; ++p) qd[n][p] = pack8(Qd[2 * p], Qd[2 * p + 1]);
        }
#pragma unroll
        for (int lb = 0; lb < LB; ++lb)
#pragma unroll
            for (int p = 0; p < NB2; ++p) {
                const u32x4 a = pack8(mma(zr[2 * p], sel[lb], kZero4), mma(zr[2 * p + 1], sel[lb], kZero4));
#pragma unroll
                for (int n = 0; n < K; ++n)
#pragma unroll
                    for (int c = 0; c < K; ++c) dWt[n][lb][c][0] = mma(a, c == 0 ? gd[n][p] : qd[n][p], dWt[n][lb][c][0]);
            }
    }
    combine_dw<K, LB, 1>(reinterpret_cast<float*>(smem_raw), dWt, dbp, partial, Lw, want_db, NARROW ? cin : -1);
}

// dY = dHnew * U * (1 - Cand^2): the blend backward alone, for a state whose gradient arrives as a finished tensor
__global__ __launch_bounds__(256) void blend_bwd_bf16_kernel(const u32x4* __restrict__ dHnew, const u32x4* __restrict__ U,
                                                             const u32x4* __restrict__ Cand, u32x4* __restrict__ dY, long long n8) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        float gn[8], u[8], c[8];
        unpack8(dHnew[i], gn); unpack8(U[i], u); unpack8(Cand[i], c);
#pragma unroll
        for (int k = 0; k < 8; ++k) gn[k] = gn[k] * u[k] * (1.f - c[k] * c[k]);
        dY[i] = pack8f(gn);
    }
}

// --------------------------------------------------------------------------------------- the whole planar cell backward in ONE launch
// Candidate backward (post-aggregation form, node_post_bwd_bf16_kernel) and gates backward with the gate / blend prologue (node_bwd_bf16_kernel,
// PRO) back to back on the same node, as stc_cell_bwd_x3.hip does for fp32 planes (reference STC_GNN.py:65-79 through autograd):
//   * dY = dHnew U (1 - Cand^2) and R*H are formed here from planes the gate prologue holds anyway (rounded to bf16 once, as the planes the
//     two-launch path stores): the forward need not store R*H, the state-gradient sum need not write dY for this cell's sake;
//   * the R*H plane's gradient goes from the candidate's tile to the gate prologue through a per-wave LDS tile, in fp32 -- never written;
//   * the candidate's X-side gradient is parked in a lane-private LDS slot and added to the gates' X tile: one plane for the source, not two;
//   * X is read once, H once (the prologue's H columns come from the lanes that hold the H half of the [X | H] row: one ds_bpermute per
//     register instead of a second read of the plane).
// 9 planes in (X, H, S.X, S.H, U, R, Cand, dHnew, dBm), 4 out, against 13 + 6 of the two launches.  One wave per SIMD, the next node's 28
// registers per row block requested a node ahead.
struct CellBwdB16 {
    const bf16_t *X, *H, *SX, *SH, *U, *R, *Cand, *dHnew, *dBm;        // X, SX: (nodes, C, cin) when narrow; the rest (nodes, C, 16)
    const float *Tc, *Wg, *Wc;                                          // (2, C, C); (4 Lw, 32); (4 Lw, 16)
    bf16_t *dX, *dSX, *dH, *dSH;                                        // dX, dSX: wide input only
    float *partial_g, *partial_c;
    int nodes, want_dbg, want_dbc, Lw;
};

__device__ __forceinline__ u32x4 from_lane(const u32x4 v, int lane_bytes) {
    return u32x4{(unsigned)__builtin_amdgcn_ds_bpermute(lane_bytes, (int)v[0]), (unsigned)__builtin_amdgcn_ds_bpermute(lane_bytes, (int)v[1]),
                 (unsigned)__builtin_amdgcn_ds_bpermute(lane_bytes, (int)v[2]), (unsigned)__builtin_amdgcn_ds_bpermute(lane_bytes, (int)v[3])};
}

template <int NB2, int NARROW>
__global__ __launch_bounds__(MF_THREADS, 1) void cell_bwd_bf16_kernel(CellBwdB16 a) {
    constexpr int K = 2, NRB = 2 * NB2, C = 32 * NB2, LB = 2;
    constexpr int RHB = NARROW ? 0 : 1;                 // block of the slab row that is the state plane (H for the gates, R*H for the candidate)
    constexpr int nTB = NRB * NB2, nWG = K * LB * K, nWC = LB * K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* TB = reinterpret_cast<u32x4*>(smem_raw);     // [NRB rb][NB2 p]        T_1[16 rb + x][32 p + pair_row]
    u32x4* WG = TB + nTB * 64;                           // [K n][LB][K c]         A: Wg[(n, c, 16 lb + x)][o(slot)]
    u32x4* WC = WG + nWG * 64;                           // [LB][K c]              A: Wc[(n(slot), c, 16 lb + x)][o(slot)]
    const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, g = lane >> 4;
    const int cin = a.Lw - 16;
    float* mine = reinterpret_cast<float*>(WC + nWC * 64) + (size_t)(tid >> 6) * 3 * NRB * 256;
    float* stash_h = mine;                              // [NRB][16 rows][16 columns]: dRH R + dHnew (1 - U), what H is owed outside the convolutions
    float* stash_x = mine + NRB * 256;                  // the candidate's X-side gradient tile, lane-private slots
    float* tile_rh = mine + 2 * NRB * 256;              // the R*H plane's gradient: candidate tile layout -> prologue layout

    for (int idx = tid; idx < nTB * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, p = f % NB2, rb = f / NB2, gg = ll >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = a.Tc[(size_t)C * C + (16 * rb + (ll & 15)) * C + 32 * p + pair_row(gg, e)];
        put_frag(TB, f, ll, v);
    }
    for (int idx = tid; idx < nWG * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, c = f % K, lb = (f / K) % LB, n = f / (K * LB), gg = ll >> 4;
        const int l = 16 * lb + (ll & 15);
        const int wl = NARROW ? stc_wrow_swapped(l, cin) : l;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int o = c == 0 ? 8 * gg + e : pair_row(gg, e);      // Q_0 comes as dY rows (natural order), Q_1 from accumulators
            v[e] = (wl >= 0 && wl < a.Lw) ? a.Wg[((size_t)(n * K + c) * a.Lw + wl) * 32 + o] : 0.f;
        }
        put_frag(WG, f, ll, v);
    }
    for (int idx = tid; idx < nWC * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, c = f % K, lb = f / K, gg = ll >> 4;
        const int l = 16 * lb + (ll & 15);
        const int wl = NARROW ? stc_wrow_swapped(l, cin) : l;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = c == 0 ? gg >> 1 : e >> 2;                         // the [dY | dBm] row: natural order; accumulators: pair order
            const int o = c == 0 ? 8 * (gg & 1) + e : 4 * gg + (e & 3);
            v[e] = (wl >= 0 && wl < a.Lw) ? a.Wc[((size_t)(n * K + c) * a.Lw + wl) * 16 + o] : 0.f;
        }
        put_frag(WC, f, ll, v);
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = gridDim.x * MF_WAVES;
    u32x4 sel[2];
    sel[0] = selector(0, x, g);
    sel[1] = selector(1, x, g);
    f32x4 dWg[K][LB][K][2], dWc[K][LB][K][1];
    float dbg[2] = {0.f, 0.f}, dbc[1] = {0.f};
#pragma unroll
    for (int n = 0; n < K; ++n)
#pragma unroll
        for (int lb = 0; lb < LB; ++lb)
#pragma unroll
            for (int c = 0; c < K; ++c) { dWg[n][lb][c][0] = kZero4; dWg[n][lb][c][1] = kZero4; dWc[n][lb][c][0] = kZero4; }

    // (the [S.X | S.H] rows are used last -- the dW products of slab 1 -- and are requested at the top of their OWN node, not a node ahead:
    //  16 registers less across the node at C = 64, where the kernel holds 512)
    struct Ops { u32x4 zx[NRB], gn[NRB], cd[NRB], uu[NRB], rr[NRB], bm[NRB]; };
    auto load_ops = [&](Ops& o, int nd) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const size_t row = (size_t)nd * C + 16 * rb + x, e = row * 16 + 8 * (g & 1);
            // wide: the [X | H] piece of the lane; narrow: H columns 8 (g & 1) .. in EVERY lane (the input's cin columns: per node, below)
            if constexpr (NARROW) o.zx[rb] = stc_ld_once(reinterpret_cast<const u32x4*>(a.H + e));
            else o.zx[rb] = load_planar8<0>(a.X, a.H, row, g, 16);
            o.gn[rb] = stc_ld_once(reinterpret_cast<const u32x4*>(a.dHnew + e));
            o.cd[rb] = stc_ld_once(reinterpret_cast<const u32x4*>(a.Cand + e));
            o.uu[rb] = stc_ld_once(reinterpret_cast<const u32x4*>(a.U + e));
            o.rr[rb] = stc_ld_once(reinterpret_cast<const u32x4*>(a.R + e));
            o.bm[rb] = stc_ld_once(reinterpret_cast<const u32x4*>(a.dBm + e));
        }
    };
    const bool holds_h = g >= 2;                            // wide: lanes whose piece of the [X | H] row is the H half (their own prologue columns)
    Ops cur, nxt;
    int node = blockIdx.x * MF_WAVES + wave;
    if (node < a.nodes) load_ops(cur, node);
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0) before the loop (see node_fwd_x3_kernel)
    while (node < a.nodes) {
        const int next_node = node + nw;
        const size_t r0 = (size_t)node * C;
        u32x4 zs[NRB], xin[NARROW ? NRB : 1];                   // narrow: the input plane's columns [x (cin) | 0] as lane group g = 2 carries them
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            zs[rb] = load_planar8<NARROW>(a.SX, a.SH, r0 + 16 * rb + x, g, cin);
            if constexpr (NARROW) {
                const bf16_t* q = a.X + (r0 + 16 * rb + x) * cin;
                const unsigned v0 = q[0], v1 = q[cin > 1 ? 1 : 0], v2 = q[cin > 2 ? 2 : 0], v3 = q[cin > 3 ? 3 : 0];
                xin[rb] = u32x4{v0 | (cin > 1 ? v1 << 16 : 0u), cin > 2 ? (v2 | (cin > 3 ? v3 << 16 : 0u)) : 0u, 0u, 0u};
            }
        }
        if (next_node < a.nodes) load_ops(nxt, next_node);      // software prefetch: one wave per SIMD has no partner to hide the loads behind
        __builtin_amdgcn_sched_barrier(0);
        const int lo = opaque(lane);
        // H, columns 8 (g & 1) .. of row 16 rb + x, in every lane: from the lanes that hold the H half of the [X | H] row
        auto h_cols = [&](int rb) {
            if constexpr (NARROW) return cur.zx[rb];
            else { const u32x4 other = from_lane(cur.zx[rb], (lane ^ 32) << 2); return holds_h ? cur.zx[rb] : other; }
        };
        // the [X | H] row as the lane's operand piece: wide = what was loaded; narrow = [H | x | 0]
        auto xh_row = [&](int rb) {
            if constexpr (NARROW) return g < 2 ? cur.zx[rb] : (g == 2 ? xin[rb] : kZeroU4);
            else return cur.zx[rb];
        };

        // =========================================================== candidate convolution: dA = dY, dBm given
        {
            u32x4 dyr[NRB], zc[NRB];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                float gn[8], cd[8], uu[8], rr[8], hh[8], dy[8], rh[8];
                unpack8(cur.gn[rb], gn); unpack8(cur.cd[rb], cd); unpack8(cur.uu[rb], uu); unpack8(cur.rr[rb], rr); unpack8(h_cols(rb), hh);
#pragma unroll
                for (int i = 0; i < 8; ++i) { dy[i] = gn[i] * uu[i] * (1.f - cd[i] * cd[i]); rh[i] = rr[i] * hh[i]; }
                const u32x4 dyp = pack8f(dy), rhp = pack8f(rh);
                dyr[rb] = g < 2 ? dyp : cur.bm[rb];               // the [dY | dBm] row
                zc[rb] = NARROW ? (g < 2 ? rhp : (g == 2 ? xin[NARROW ? rb : 0] : kZeroU4)) : (g < 2 ? cur.zx[rb] : rhp);      // [X | R*H]; narrow: [R*H | x | 0]
            }
            u32x4 gd[K][NB2];
#pragma unroll
            for (int n = 0; n < K; ++n)
#pragma unroll
                for (int p = 0; p < NB2; ++p) {
                    const f32x4 t0 = mma(dyr[2 * p], sel[n], kZero4), t1 = mma(dyr[2 * p + 1], sel[n], kZero4);
                    if (n == 0) dbc[0] += ((t0[0] + t0[1]) + (t0[2] + t0[3])) + ((t1[0] + t1[1]) + (t1[2] + t1[3]));      // the bias sits on A only
                    gd[n][p] = pack8(t0, t1);
                }
            u32x4 qb1[NRB];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                f32x4 Qv[K] = {kZero4, kZero4};
#pragma unroll
                for (int p = 0; p < NB2; ++p) {
                    const u32x4 t = TB[(rb * NB2 + p) * 64 + lo];
#pragma unroll
                    for (int n = 0; n < K; ++n) Qv[n] = mma(gd[n][p], t, Qv[n]);
                }
                qb1[rb] = pack8(Qv[0], Qv[1]);
            }
#pragma unroll
            for (int lb = 0; lb < LB; ++lb) {
                if (NARROW && lb != RHB) continue;              // the narrow input plane needs no gradient
                const u32x4 w0 = WC[(lb * K + 0) * 64 + lo], w1 = WC[(lb * K + 1) * 64 + lo];
                float* dst = lb == RHB ? tile_rh : stash_x;
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    const f32x4 z = mma(w1, qb1[rb], mma(w0, dyr[rb], kZero4));
                    *reinterpret_cast<float4*>(dst + (rb * 16 + x) * 16 + 4 * g) = make_float4(z[0], z[1], z[2], z[3]);
                }
            }
            u32x4 qd[K][NB2];
#pragma unroll
            for (int n = 0; n < K; ++n) {
                f32x4 Qd[NRB];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    Qd[rb] = kZero4;
#pragma unroll
                    for (int p = 0; p < NB2; ++p) Qd[rb] = mma(TB[(rb * NB2 + p) * 64 + lo], gd[n][p], Qd[rb]);
                }
#pragma unroll
                for (int p = 0; p < NB2; ++p) qd[n][p] = pack8(Qd[2 * p], Qd[2 * p + 1]);
            }
#pragma unroll
            for (int lb = 0; lb < LB; ++lb)
#pragma unroll
                for (int p = 0; p < NB2; ++p) {
                    const u32x4 za = pack8(mma(zc[2 * p], sel[lb], kZero4), mma(zc[2 * p + 1], sel[lb], kZero4));
#pragma unroll
                    for (int n = 0; n < K; ++n)
#pragma unroll
                        for (int c = 0; c < K; ++c) dWc[n][lb][c][0] = mma(za, c == 0 ? gd[n][p] : qd[n][p], dWc[n][lb][c][0]);
                }
        }
        __builtin_amdgcn_wave_barrier();                        // (LDS operations of one wave complete in order: the tile is readable)

        // =========================================================== gate + blend backward, then the gates convolution
        u32x4 dyr[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const float4 d0 = *reinterpret_cast<const float4*>(tile_rh + (rb * 16 + x) * 16 + 8 * (g & 1));
            const float4 d1 = *reinterpret_cast<const float4*>(tile_rh + (rb * 16 + x) * 16 + 8 * (g & 1) + 4);
            const float dr[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
            float gn[8], cd[8], uu[8], rr[8], hh[8], gy[8], dh[8];
            unpack8(cur.gn[rb], gn); unpack8(cur.cd[rb], cd); unpack8(cur.uu[rb], uu); unpack8(cur.rr[rb], rr); unpack8(h_cols(rb), hh);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float gu = gn[i] * (cd[i] - hh[i]) * uu[i] * (1.f - uu[i]);
                const float gr = dr[i] * hh[i] * rr[i] * (1.f - rr[i]);
                gy[i] = g < 2 ? gu : gr;
                dh[i] = fmaf(dr[i], rr[i], gn[i] * (1.f - uu[i]));
            }
            dyr[rb] = pack8f(gy);
            if (g < 2) {
                float4* slot = reinterpret_cast<float4*>(stash_h + (rb * 16 + x) * 16 + 8 * g);
                slot[0] = make_float4(dh[0], dh[1], dh[2], dh[3]);
                slot[1] = make_float4(dh[4], dh[5], dh[6], dh[7]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        u32x4 gd[2][NB2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int p = 0; p < NB2; ++p) {
                const f32x4 t0 = mma(dyr[2 * p], sel[hb], kZero4), t1 = mma(dyr[2 * p + 1], sel[hb], kZero4);
                dbg[hb] += ((t0[0] + t0[1]) + (t0[2] + t0[3])) + ((t1[0] + t1[1]) + (t1[2] + t1[3]));
                gd[hb][p] = pack8(t0, t1);
            }
        u32x4 qb[K][NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            qb[0][rb] = dyr[rb];
            f32x4 Qv[2] = {kZero4, kZero4};
#pragma unroll
            for (int p = 0; p < NB2; ++p) {
                const u32x4 t = TB[(rb * NB2 + p) * 64 + lo];
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) Qv[hb] = mma(gd[hb][p], t, Qv[hb]);
            }
            qb[1][rb] = pack8(Qv[0], Qv[1]);
        }
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int lb = 0; lb < LB; ++lb) {
                if (NARROW && lb != RHB) continue;              // narrow input plane: only the state plane's gradient is wanted
                f32x4 z[NRB];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) z[rb] = kZero4;
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    const u32x4 w = WG[((n * LB + lb) * K + c) * 64 + lo];
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) z[rb] = mma(w, qb[c][rb], z[rb]);
                }
                if (n == 0) {                                   // the H plane's tile: + the prologue's share; the X plane's: + the candidate's
                    const float* add = lb == RHB ? stash_h : stash_x;
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) {
                        const float4 sh = *reinterpret_cast<const float4*>(add + (rb * 16 + x) * 16 + 4 * g);
                        z[rb][0] += sh.x; z[rb][1] += sh.y; z[rb][2] += sh.z; z[rb][3] += sh.w;
                    }
                }
                bf16_t* dst = lb == RHB ? (n == 0 ? a.dH : a.dSH) : (n == 0 ? a.dX : a.dSX);
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) stc_st_once(reinterpret_cast<u32x2*>(dst + (r0 + 16 * rb + x) * 16 + 4 * g), pack4(z[rb]));
            }
        u32x4 qd[2][NB2];
        {
            f32x4 Qd[NRB][2];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                Qd[rb][0] = kZero4; Qd[rb][1] = kZero4;
#pragma unroll
                for (int p = 0; p < NB2; ++p) {
                    const u32x4 t = TB[(rb * NB2 + p) * 64 + lo];
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) Qd[rb][hb] = mma(t, gd[hb][p], Qd[rb][hb]);
                }
            }
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int p = 0; p < NB2; ++p) qd[hb][p] = pack8(Qd[2 * p][hb], Qd[2 * p + 1][hb]);
        }
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int lb = 0; lb < LB; ++lb)
#pragma unroll
                for (int p = 0; p < NB2; ++p) {
                    const u32x4 z0 = n == 0 ? xh_row(2 * p) : zs[2 * p], z1 = n == 0 ? xh_row(2 * p + 1) : zs[2 * p + 1];
                    const u32x4 za = pack8(mma(z0, sel[lb], kZero4), mma(z1, sel[lb], kZero4));
#pragma unroll
                    for (int c = 0; c < K; ++c)
#pragma unroll
                        for (int hb = 0; hb < 2; ++hb) dWg[n][lb][c][hb] = mma(za, c == 0 ? gd[hb][p] : qd[hb][p], dWg[n][lb][c][hb]);
                }
        __builtin_amdgcn_wave_barrier();                        // the next node overwrites the wave's LDS slots
        cur = nxt;
        node = next_node;
    }
    combine_dw<K, LB, 2>(reinterpret_cast<float*>(smem_raw), dWg, dbg, a.partial_g, a.Lw, a.want_dbg, NARROW ? cin : -1);
    combine_dw<K, LB, 1>(reinterpret_cast<float*>(smem_raw), dWc, dbc, a.partial_c, a.Lw, a.want_dbc, NARROW ? cin : -1);
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
    BPtrs2 zp{};
    BDPtrs2 dzp{};
    for (int n = 0; n < K; ++n) { zp.p[n] = static_cast<const bf16_t*>(Z[n]); dzp.p[n] = static_cast<bf16_t*>(dZ[n]); }
    const long long want = (nodes + MF_WAVES - 1) / MF_WAVES;
    int grid = resident < MF_BWD_MAX_GRID ? resident : MF_BWD_MAX_GRID;
    if (want < grid) grid = (int)want;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MF_THREADS), lds, stream, zp, Tc, W, static_cast<const bf16_t*>(dY), dzp, partial, (int)nodes, want_db, Lw, GatesPro{});
    STC_LAUNCH_CHECK("node_bwd_bf16 launch");
    *n_partials = grid;
    return STC_OK;
}

template <int NB2, int NARROW, int POST>
int launch_gates_fwd(const GatesFwd& a, const float* Tc, const float* W, const float* bias, long long nodes, int Lw, hipStream_t stream) {
    constexpr int NRB = 2 * NB2;
    const size_t lds = (size_t)(2 * 4 + 2 * NB2 * NRB + (POST ? 4 : 0)) * 64 * 16;
    auto kern = cell_gates_fwd_bf16_kernel<NB2, NARROW, POST>;
    if (int rc = stc::hip_status(stc::allow_lds(kern, lds), "hipFuncSetAttribute(gates fwd bf16)")) return rc;
    static const int resident = stc::resident_blocks(kern, MF_THREADS, lds, 2);
    const long long want = (nodes + MF_WAVES - 1) / MF_WAVES;
    const int grid = (int)(want < resident ? want : resident);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MF_THREADS), lds, stream, a, Tc, W, bias, (int)nodes, Lw);
    STC_LAUNCH_CHECK("cell_gates_fwd_bf16 launch");
    return STC_OK;
}

template <int NB2, int PL, int WAVES = BwdWaves<NB2, 2>::v>
int launch_gates_bwd(const BPtrs2& zp, const BDPtrs2& dzp, const GatesPro& pro, const float* Tc, const float* W,
                     float* partial, int* n_partials, int want_db, long long nodes, int Lw, hipStream_t stream) {
    constexpr int NRB = 2 * NB2, K = 2, LB = 2, nW = K * K * 32 * 32;
    const size_t frag = (size_t)(NRB * NB2 + K * LB * K) * 64 * 16 + (size_t)MF_WAVES * NRB * 256 * sizeof(float);     // tables + the dH stash
    const size_t slabs = (size_t)MF_WAVES * (nW + 32) * sizeof(float);
    const size_t lds = frag > slabs ? frag : slabs;
    auto kern = node_bwd_bf16_kernel<NB2, 2, 2, 32, PL, 1, WAVES>;
    if (int rc = stc::hip_status(stc::allow_lds(kern, lds), "hipFuncSetAttribute(gates bwd bf16)")) return rc;
    static const int resident = stc::resident_blocks(kern, MF_THREADS, lds, WAVES);
    const long long want = (nodes + MF_WAVES - 1) / MF_WAVES;
    int grid = resident < MF_BWD_MAX_GRID ? resident : MF_BWD_MAX_GRID;
    if (want < grid) grid = (int)want;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MF_THREADS), lds, stream, zp, Tc, W, static_cast<const bf16_t*>(nullptr), dzp, partial, (int)nodes, want_db, Lw, pro);
    STC_LAUNCH_CHECK("cell_gates_bwd_bf16 launch");
    *n_partials = grid;
    return STC_OK;
}

template <int NB2, int NARROW>
int launch_post_bwd(const bf16_t* X, const bf16_t* X2, const float* Tc, const float* W, const bf16_t* dA, const bf16_t* dB, bf16_t* dX, bf16_t* dX2,
                    float* partial, int* n_partials, int want_db, long long nodes, int Lw, hipStream_t stream) {
    constexpr int NRB = 2 * NB2, nW = 4 * 32 * 16;
    const size_t frag = (size_t)(NRB * NB2 + 4) * 64 * 16;
    const size_t slabs = (size_t)MF_WAVES * (nW + 16) * sizeof(float);
    const size_t lds = frag > slabs ? frag : slabs;
    auto kern = node_post_bwd_bf16_kernel<NB2, NARROW>;
    if (int rc = stc::hip_status(stc::allow_lds(kern, lds), "hipFuncSetAttribute(post bwd bf16)")) return rc;
    static const int resident = stc::resident_blocks(kern, MF_THREADS, lds, NB2 == 1 ? 2 : 1);
    const long long want = (nodes + MF_WAVES - 1) / MF_WAVES;
    int grid = resident < MF_BWD_MAX_GRID ? resident : MF_BWD_MAX_GRID;
    if (want < grid) grid = (int)want;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MF_THREADS), lds, stream, X, X2, Tc, W, dA, dB, dX, dX2, partial, (int)nodes, want_db, Lw);
    STC_LAUNCH_CHECK("node_post_bwd_bf16 launch");
    *n_partials = grid;
    return STC_OK;
}

bool bf16_shape(int Ks, int Kc, int C, int L, int Lw, int Ho, long long nodes) {
    return Ks == Kc && Ks >= 1 && Ks <= 3 && (C == 32 || C == 64) && (Ho == 16 || Ho == 32) && (L == 16 || L == 32) &&
           Lw >= 1 && Lw <= L && nodes >= 0 && nodes < (1ll << 31) / C;
}

template <int NB2, int NARROW>
int launch_cell_bwd(const CellBwdB16& a, int* n_partials, hipStream_t stream) {
    constexpr int NRB = 2 * NB2, nW = 4 * 32 * 32;
    const size_t frag = (size_t)(NRB * NB2 + 8 + 4) * 64 * 16 + (size_t)MF_WAVES * 3 * NRB * 256 * sizeof(float);     // tables + the three per-wave tiles
    const size_t slabs = (size_t)MF_WAVES * (nW + 32) * sizeof(float);
    const size_t lds = frag > slabs ? frag : slabs;
    auto kern = cell_bwd_bf16_kernel<NB2, NARROW>;
    if (int rc = stc::hip_status(stc::allow_lds(kern, lds), "hipFuncSetAttribute(cell bwd bf16)")) return rc;
    static const int resident = stc::resident_blocks(kern, MF_THREADS, lds, 1);
    const long long want = (a.nodes + MF_WAVES - 1) / MF_WAVES;
    int grid = resident < MF_BWD_MAX_GRID ? resident : MF_BWD_MAX_GRID;
    if (want < grid) grid = (int)want;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MF_THREADS), lds, stream, a);
    STC_LAUNCH_CHECK("cell_bwd_bf16 launch");
    *n_partials = grid;
    return STC_OK;
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

// ---- planar STC_Cell on bf16 planes (K = 2, hidden 16)
extern "C" int stc_cell_planar_bf16_supported(int32_t Ks, int32_t Kc, int32_t C, int32_t h) {
    return (Ks == 2 && Kc == 2 && (C == 32 || C == 64) && h == 16) ? 1 : 0;
}

extern "C" int stc_cell_gates_fwd_planar_bf16(const void* X, const void* H, const void* SX, const void* SH,
                                              const float* Tc, const float* W, const float* bias,
                                              void* U, void* Rg, void* RH,
                                              const float* Wc, const float* bc, void* A, void* Bm,
                                              int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream) {
    STC_REQUIRE(stc_cell_planar_bf16_supported(2, 2, C, h), STC_EUNSUPPORTED, "stc_cell_gates_fwd_planar_bf16: C=%d h=%d not on the bf16 planar path", C, h);
    const int cin = Lw - 16;
    STC_REQUIRE(cin == 16 || (cin >= 1 && cin <= 4), STC_EUNSUPPORTED, "stc_cell_gates_fwd_planar_bf16: input plane of %d columns (16, or 1..4)", cin);
    STC_REQUIRE(nodes >= 0 && nodes < (1ll << 31) / C, STC_ELIMIT, "stc_cell_gates_fwd_planar_bf16: nodes=%lld", (long long)nodes);
    if (nodes == 0) return STC_OK;
    STC_REQUIRE(X && H && SX && SH && Tc && W && U && Rg && (RH || A), STC_EINVAL, "stc_cell_gates_fwd_planar_bf16: null pointer (RH may be NULL only with the fused candidate projection)");
    STC_REQUIRE((A == nullptr) == (Bm == nullptr) && (A == nullptr) == (Wc == nullptr), STC_EINVAL, "stc_cell_gates_fwd_planar_bf16: Wc, A, Bm go together");
    STC_REQUIRE(stc::aligned16(H) && stc::aligned16(SH) && stc::aligned16(U) && stc::aligned16(Rg) && stc::aligned16(RH) &&
                    (cin != 16 || (stc::aligned16(X) && stc::aligned16(SX))) && (!A || (stc::aligned16(A) && stc::aligned16(Bm))),
                STC_EALIGN, "stc_cell_gates_fwd_planar_bf16: planes must be 16-byte aligned");
    const GatesFwd a{static_cast<const bf16_t*>(X), static_cast<const bf16_t*>(H), static_cast<const bf16_t*>(SX), static_cast<const bf16_t*>(SH),
                     static_cast<bf16_t*>(U), static_cast<bf16_t*>(Rg), static_cast<bf16_t*>(RH), Wc, bc, static_cast<bf16_t*>(A), static_cast<bf16_t*>(Bm)};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool narrow = cin != 16, post = A != nullptr;
#define GF(NB2_) (narrow ? (post ? launch_gates_fwd<NB2_, 1, 1>(a, Tc, W, bias, nodes, Lw, s) : launch_gates_fwd<NB2_, 1, 0>(a, Tc, W, bias, nodes, Lw, s)) \
                         : (post ? launch_gates_fwd<NB2_, 0, 1>(a, Tc, W, bias, nodes, Lw, s) : launch_gates_fwd<NB2_, 0, 0>(a, Tc, W, bias, nodes, Lw, s)))
    return C == 32 ? GF(1) : GF(2);
#undef GF
}

// planar form: X / SX (nodes, C, cin), H / SH and everything else (nodes, C, 16); dZ = {dX, dSX, dH plane, dSH} (the first two
// unused and may be null for a narrow input plane); dH: the previous state's share from the gates prologue, or null to have it
// added into dZ[2] (one gradient plane for the previous state instead of two)
extern "C" int stc_cell_gates_bwd_planar_bf16(const void* X, const void* H, const void* SX, const void* SH,
                                              const float* Tc, const float* W,
                                              const void* dCandIn, const void* Cand, const void* U, const void* Rg, const void* dHnew,
                                              void* const* dZ, float* dW, float* db, void* dH,
                                              void* workspace, size_t workspace_bytes,
                                              int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream) {
    STC_REQUIRE(stc_cell_planar_bf16_supported(2, 2, C, h), STC_EUNSUPPORTED, "stc_cell_gates_bwd_planar_bf16: C=%d h=%d not on the bf16 planar path", C, h);
    const int cin = Lw - 16;
    STC_REQUIRE(cin == 16 || (cin >= 1 && cin <= 4), STC_EUNSUPPORTED, "stc_cell_gates_bwd_planar_bf16: input plane of %d columns (16, or 1..4)", cin);
    STC_REQUIRE(nodes >= 0 && nodes < (1ll << 31) / C, STC_ELIMIT, "stc_cell_gates_bwd_planar_bf16: nodes=%lld", (long long)nodes);
    STC_REQUIRE(W && dW && Tc, STC_EINVAL, "stc_cell_gates_bwd_planar_bf16: null W/dW/Tc");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nW = 4 * Lw * 32;
    if (nodes == 0) {
        if (int rc = stc::hip_status(hipMemsetAsync(dW, 0, (size_t)nW * sizeof(float), s), "memset dW")) return rc;
        if (db) if (int rc = stc::hip_status(hipMemsetAsync(db, 0, 32 * sizeof(float), s), "memset db")) return rc;
        return STC_OK;
    }
    const bool narrow = cin != 16;
    STC_REQUIRE(X && H && SX && SH && dCandIn && Cand && U && Rg && dHnew && dZ && dZ[2] && dZ[3] && (narrow || (dZ[0] && dZ[1])), STC_EINVAL,
                "stc_cell_gates_bwd_planar_bf16: null pointer");
    for (const void* q : {H, SH, dCandIn, Cand, U, Rg, dHnew, (const void*)dH, (const void*)dZ[2], (const void*)dZ[3]})       // (null dH is aligned)
        STC_REQUIRE(stc::aligned16(q), STC_EALIGN, "stc_cell_gates_bwd_planar_bf16: planes must be 16-byte aligned");
    if (!narrow) STC_REQUIRE(stc::aligned16(X) && stc::aligned16(SX) && stc::aligned16(dZ[0]) && stc::aligned16(dZ[1]), STC_EALIGN,
                             "stc_cell_gates_bwd_planar_bf16: planes must be 16-byte aligned");
    STC_REQUIRE(workspace && stc::aligned16(workspace) && workspace_bytes >= stc_bdg_node_bwd_workspace_bytes(2, 2, C, 32, 32, 0), STC_EINVAL,
                "stc_cell_gates_bwd_planar_bf16: workspace null, misaligned or too small (%zu B)", workspace_bytes);
    auto B = [](const void* q) { return static_cast<const bf16_t*>(q); };
    auto M = [](void* q) { return static_cast<bf16_t*>(q); };
    BPtrs2 zp{};
    BDPtrs2 dzp{};
    zp.p[0] = B(X); zp.p[1] = B(SX); zp.q[0] = B(H); zp.q[1] = B(SH);
    dzp.p[0] = narrow ? nullptr : M(dZ[0]); dzp.p[1] = narrow ? nullptr : M(dZ[1]); dzp.q[0] = M(dZ[2]); dzp.q[1] = M(dZ[3]);
    const GatesPro pro{B(dCandIn), B(Cand), B(H), B(U), B(Rg), B(dHnew), M(dH)};
    float* partial = static_cast<float*>(workspace);
    int n_parts = 0;
    // C = 64: one wave per SIMD with software prefetch (a 256-register two-wave build spilled: 2313 vs 1840 us per launch, not kept)
    const int rc = C == 32 ? (narrow ? launch_gates_bwd<1, 2>(zp, dzp, pro, Tc, W, partial, &n_parts, db != nullptr, nodes, Lw, s)
                                     : launch_gates_bwd<1, 1>(zp, dzp, pro, Tc, W, partial, &n_parts, db != nullptr, nodes, Lw, s))
                           : (narrow ? launch_gates_bwd<2, 2, 1>(zp, dzp, pro, Tc, W, partial, &n_parts, db != nullptr, nodes, Lw, s)
                                     : launch_gates_bwd<2, 1, 1>(zp, dzp, pro, Tc, W, partial, &n_parts, db != nullptr, nodes, Lw, s));
    if (rc != STC_OK) return rc;
    return stc_node_reduce_partials(partial, n_parts, nW, 32, dW, db, s);
}

// ---- the whole backward of a planar cell step on bf16 planes in one launch (cell_bwd_bf16_kernel)
// (C = 64 with a narrow input plane: the kernel does not fit the register file -- 62 scratch accesses per node -- and is not dispatched;
//  the schedule's layer-0 cells keep the two launches there)
extern "C" int stc_cell_bwd_planar_bf16_supported(int32_t C, int32_t Lw, int32_t h) {
    if (!stc_cell_planar_bf16_supported(2, 2, C, h)) return 0;
    const int cin = Lw - h;
    return (cin == 16 || (C == 32 && cin >= 1 && cin <= 4)) ? 1 : 0;
}

extern "C" int stc_cell_bwd_planar_bf16(const void* X, const void* H, const void* SX, const void* SH,
                                        const float* Tc, const float* Wg, const float* Wc,
                                        const void* U, const void* Rg, const void* Cand, const void* dHnew, const void* dBm,
                                        void* dX, void* dSX, void* dH, void* dSH,
                                        float* dWg, float* dbg, float* dWc, float* dbc,
                                        void* workspace, size_t workspace_bytes,
                                        int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream) {
    STC_REQUIRE(stc_cell_bwd_planar_bf16_supported(C, Lw, h), STC_EUNSUPPORTED,
                "stc_cell_bwd_planar_bf16: C=%d h=%d input width %d is not built (C = 32: 16 or 1..4 columns; C = 64: 16)", C, h, Lw - h);
    const int cin = Lw - 16;
    STC_REQUIRE(nodes >= 0 && nodes < (1ll << 31) / C, STC_ELIMIT, "stc_cell_bwd_planar_bf16: nodes=%lld", (long long)nodes);
    STC_REQUIRE(Wg && Wc && dWg && dWc && Tc, STC_EINVAL, "stc_cell_bwd_planar_bf16: null W/dW/Tc");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nWg = 4 * Lw * 32, nWc = 4 * Lw * 16;
    if (nodes == 0) {
        if (int rc = stc::hip_status(hipMemsetAsync(dWg, 0, (size_t)nWg * sizeof(float), s), "memset dWg")) return rc;
        if (int rc = stc::hip_status(hipMemsetAsync(dWc, 0, (size_t)nWc * sizeof(float), s), "memset dWc")) return rc;
        if (dbg) if (int rc = stc::hip_status(hipMemsetAsync(dbg, 0, 32 * sizeof(float), s), "memset dbg")) return rc;
        if (dbc) if (int rc = stc::hip_status(hipMemsetAsync(dbc, 0, 16 * sizeof(float), s), "memset dbc")) return rc;
        return STC_OK;
    }
    const bool narrow = cin != 16;
    STC_REQUIRE(X && H && SX && SH && U && Rg && Cand && dHnew && dBm && dH && dSH && (narrow || (dX && dSX)), STC_EINVAL, "stc_cell_bwd_planar_bf16: null pointer");
    for (const void* q : {H, SH, U, Rg, Cand, dHnew, dBm, (const void*)dH, (const void*)dSH})
        STC_REQUIRE(stc::aligned16(q), STC_EALIGN, "stc_cell_bwd_planar_bf16: planes must be 16-byte aligned");
    if (!narrow) STC_REQUIRE(stc::aligned16(X) && stc::aligned16(SX) && stc::aligned16(dX) && stc::aligned16(dSX), STC_EALIGN,
                             "stc_cell_bwd_planar_bf16: planes must be 16-byte aligned");
    const size_t bytes_g = stc_bdg_node_bwd_workspace_bytes(2, 2, C, 32, 32, 0);
    STC_REQUIRE(workspace && stc::aligned16(workspace) && workspace_bytes >= bytes_g + stc_bdg_node_bwd_workspace_bytes(2, 2, C, 32, 16, 0), STC_EINVAL,
                "stc_cell_bwd_planar_bf16: workspace null, misaligned or too small (%zu B)", workspace_bytes);
    auto B = [](const void* q) { return static_cast<const bf16_t*>(q); };
    auto M = [](void* q) { return static_cast<bf16_t*>(q); };
    float* partial_g = static_cast<float*>(workspace);
    float* partial_c = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + bytes_g);
    const CellBwdB16 a{B(X), B(H), B(SX), B(SH), B(U), B(Rg), B(Cand), B(dHnew), B(dBm), Tc, Wg, Wc, M(dX), M(dSX), M(dH), M(dSH),
                       partial_g, partial_c, (int)nodes, dbg != nullptr, dbc != nullptr, Lw};
    int n_parts = 0;
    const int rc = C == 32 ? (narrow ? launch_cell_bwd<1, 1>(a, &n_parts, s) : launch_cell_bwd<1, 0>(a, &n_parts, s))
                           : launch_cell_bwd<2, 0>(a, &n_parts, s);
    if (rc != STC_OK) return rc;
    if (int r2 = stc_node_reduce_partials(partial_g, n_parts, nWg, 32, dWg, dbg, s)) return r2;
    return stc_node_reduce_partials(partial_c, n_parts, nWc, 16, dWc, dbc, s);
}

// post-aggregation backward of the candidate convolution on planes: wide -- X (columns 0..15), X2 (16..31), gradients dX, dX2;
// narrow (Lw = 16 + cin, cin <= 4) -- X = the 16-wide plane, X2 = the (nodes, C, cin) input plane, gradient for X only (dX2 null)
extern "C" int stc_bdg_node_post_bwd_bf16(const void* X, const void* X2, const float* Tc, const float* W, const void* dA, const void* dB,
                                          void* dX, void* dX2, float* dW, float* db, void* workspace, size_t workspace_bytes,
                                          int64_t nodes, int32_t C, int32_t Lw, int32_t Ho, void* stream) {
    STC_REQUIRE((C == 32 || C == 64) && Ho == 16, STC_EUNSUPPORTED, "stc_bdg_node_post_bwd_bf16: C=%d Ho=%d not on the bf16 planar path", C, Ho);
    const int cin = Lw - 16;
    STC_REQUIRE(cin == 16 || (cin >= 1 && cin <= 4), STC_EUNSUPPORTED, "stc_bdg_node_post_bwd_bf16: second plane of %d columns (16, or 1..4)", cin);
    STC_REQUIRE(nodes >= 0 && nodes < (1ll << 31) / C, STC_ELIMIT, "stc_bdg_node_post_bwd_bf16: nodes=%lld", (long long)nodes);
    STC_REQUIRE(W && dW && Tc, STC_EINVAL, "stc_bdg_node_post_bwd_bf16: null W/dW/Tc");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nW = 4 * Lw * 16;
    if (nodes == 0) {
        if (int rc = stc::hip_status(hipMemsetAsync(dW, 0, (size_t)nW * sizeof(float), s), "memset dW")) return rc;
        if (db) if (int rc = stc::hip_status(hipMemsetAsync(db, 0, 16 * sizeof(float), s), "memset db")) return rc;
        return STC_OK;
    }
    const bool narrow = cin != 16;
    STC_REQUIRE(X && X2 && dA && dB && dX && (narrow ? dX2 == nullptr : dX2 != nullptr), STC_EINVAL,
                "stc_bdg_node_post_bwd_bf16: null pointer (the narrow input plane gets no gradient: dX2 must be null there)");
    STC_REQUIRE(stc::aligned16(X) && stc::aligned16(dA) && stc::aligned16(dB) && stc::aligned16(dX) && (narrow || (stc::aligned16(X2) && stc::aligned16(dX2))),
                STC_EALIGN, "stc_bdg_node_post_bwd_bf16: planes must be 16-byte aligned");
    STC_REQUIRE(workspace && stc::aligned16(workspace) && workspace_bytes >= stc_bdg_node_bwd_workspace_bytes(2, 2, C, 32, 16, 0), STC_EINVAL,
                "stc_bdg_node_post_bwd_bf16: workspace null, misaligned or too small (%zu B)", workspace_bytes);
    auto B = [](const void* q) { return static_cast<const bf16_t*>(q); };
    float* partial = static_cast<float*>(workspace);
    int n_parts = 0;
#define PB(NB2_, NAR_) launch_post_bwd<NB2_, NAR_>(B(X), B(X2), Tc, W, B(dA), B(dB), static_cast<bf16_t*>(dX), static_cast<bf16_t*>(dX2), partial, &n_parts, db != nullptr, nodes, Lw, s)
    const int rc = C == 32 ? (narrow ? PB(1, 1) : PB(1, 0)) : (narrow ? PB(2, 1) : PB(2, 0));
#undef PB
    if (rc != STC_OK) return rc;
    return stc_node_reduce_partials(partial, n_parts, nW, 16, dW, db, s);
}

extern "C" int stc_gru_blend_bwd_bf16(const void* dHnew, const void* U, const void* Cand, void* dCpre, int64_t n, void* stream) {
    STC_REQUIRE(n >= 0 && n % 8 == 0, STC_EINVAL, "stc_gru_blend_bwd_bf16: n=%lld must be a non-negative multiple of 8", (long long)n);
    if (n == 0) return STC_OK;
    STC_REQUIRE(dHnew && U && Cand && dCpre, STC_EINVAL, "stc_gru_blend_bwd_bf16: null pointer");
    STC_REQUIRE(stc::aligned16(dHnew) && stc::aligned16(U) && stc::aligned16(Cand) && stc::aligned16(dCpre), STC_EALIGN, "stc_gru_blend_bwd_bf16: misaligned operand");
    const long long n8 = n / 8;
    const long long blocks = (n8 + 255) / 256;
    hipLaunchKernelGGL(blend_bwd_bf16_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const u32x4*>(dHnew), static_cast<const u32x4*>(U), static_cast<const u32x4*>(Cand), static_cast<u32x4*>(dCpre), n8);
    STC_LAUNCH_CHECK("stc_gru_blend_bwd_bf16 launch");
    return STC_OK;
}
