// Internal: register-fragment types shared by the two matrix-core node kernels (stc_node_mfma.hip: fp32 MFMA,
// stc_node_x3.hip: bf16 MFMA on exactly split fp32 operands).  Both use the accumulator layout of the 16 x 16
// MFMAs of gfx950 -- lane (x = lane & 15, g = lane >> 4), register r  <->  tile row 4g + r, tile column x -- so the
// per-node gradient fragments, the fused cell prologue / epilogue operands and the dW combine are common.
#pragma once
#include "stc_common.h"
#include "stc_node_mfma.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int MF_THREADS = 256;
constexpr int MF_WAVES = MF_THREADS / 64;
constexpr int MF_BWD_MAX_GRID = 512;     // partial rows the caller's workspace holds

// q: second plane of a slab in the planar layout ([Xt | H] kept as two contiguous (nodes, C, 16) planes); null otherwise
struct ZPtrs { const float* p[STC_MAX_K]; const float* q[STC_MAX_K]; };
struct DZPtrs { float* p[STC_MAX_K]; float* q[STC_MAX_K]; };        // q: second plane of a gradient slab (planar layout)

template <int N>
struct AtLeast1 { static constexpr int v = N > 0 ? N : 1; };

// A value the optimiser cannot see through: keeps LDS fragment fetches inside the node loop instead of
// being hoisted into ~80 registers (which costs a wave of occupancy per SIMD).
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// Fused epilogues of the two convolutions of an STC_Cell (reference STC_GNN.py:71-78), hidden width 16:
//   EPI_GATES (Ho = 32): U = sigmoid(y[:, :16]), R = sigmoid(y[:, 16:]), CandIn = [Xt | R*H | 0-pad]  -- the
//              gate pre-activations are never written; Xt is copied from this wave's own A fragment of slab 0
//   EPI_BLEND (Ho = 16): Cand = tanh(y), Hnew = (1-U)*H + U*Cand
enum { EPI_NONE = 0, EPI_GATES = 1, EPI_BLEND = 2 };
// BLEND can also drop the new state straight into the [Xt | H | pad] input rows of the cells that consume it (columns
// [off, off+16) of rows of ld floats), so those cells need no concat pass (reference STC_GNN.py:68 torch.cat).
struct StateCopy { float* p; int ld, off; };
struct FwdEpi {
    const float* H;       // (nodes, C, 16) previous state
    const float* U;       // BLEND in : update gate
    float* U_out;         // GATES out: update gate
    float* R_out;         // GATES out: reset gate
    float* CandIn;        // GATES out: (nodes, C, L) input rows of the candidate convolution
    float* Cand;          // BLEND out: tanh(candidate)
    float* Hnew;          // BLEND out: new state
    int cin;              // GATES: width of Xt inside a row (the H part starts there)
    StateCopy also[2];    // BLEND out, optional: further destinations of Hnew
    const float* side_src;   // BLEND, optional, belongs to also[0]: (nodes, C, side_cin) values for its columns [0, side_cin);
    int side_cin;            //   its pad columns [side_cin + 16, ld) are zeroed -- the consumer's row is then complete
    float* zmax;             // fp16 x 2 planar forms, optional: (2 K, 256) slots, zero-filled by the caller, that receive max |plane| of the
                             //   launch's 2 K input planes (rows in the launch order of Z): the scales of the backward's dW products
};

inline void set_state_copies(FwdEpi& epi, const StcStateCopies* c) {
    if (!c) return;
    for (int k = 0; k < 2; ++k) epi.also[k] = StateCopy{c->dst[k], c->ld[k], c->off[k]};
    epi.side_src = c->dst[0] ? c->side_src : nullptr;
    epi.side_cin = c->side_cin;
}

// accumulator-layout lanes (row 4g + r, column x of each 16-row block) write the extra copies of the new state
template <int NRB>
__device__ __forceinline__ void store_state_copies(const FwdEpi& epi, size_t row0, int x, int g, const float (&hn)[NRB][4]) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (epi.also[k].p) {
            float* base = epi.also[k].p + epi.also[k].off + x;
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) base[(row0 + 16 * rb + 4 * g + r) * epi.also[k].ld] = hn[rb][r];
        }
    if (epi.side_src && x < epi.also[0].ld - 16) {
        const int col = x < epi.side_cin ? x : x + 16;
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t row = row0 + 16 * rb + 4 * g + r;
                epi.also[0].p[row * epi.also[0].ld + col] = x < epi.side_cin ? epi.side_src[row * epi.side_cin + x] : 0.f;
            }
    }
}

template <int NRB, int HB>
struct DyFrag {     // one node's dY in both register layouts
    f32x4 d[NRB][HB];   // component t: dY[16kb + 4g + t][16hb + x]        (tile rows d, columns o; lane = o)
    f32x4 v[NRB][HB];   // dY[16rb + x][16hb + 4g + 0..3]                   (tile rows o, columns d; lane = d)
    __device__ __forceinline__ void load(const float* __restrict__ dY, int node, int j, int q) {
        constexpr int C = 16 * NRB, Ho = 16 * HB;
        const size_t r0 = (size_t)node * C;
#pragma unroll
        for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
#pragma unroll
                for (int t = 0; t < 4; ++t) d[kb][hb][t] = dY[(r0 + 16 * kb + 4 * q + t) * Ho + 16 * hb + j];
                const float4 x = *reinterpret_cast<const float4*>(dY + (r0 + 16 * kb + j) * Ho + 16 * hb + 4 * q);
                v[kb][hb] = f32x4{x.x, x.y, x.z, x.w};
            }
    }
    // nodes of Cr <= 16 rows (NRB = 1; rows of the node start at r0): rows past the node's enter as zeros
    __device__ __forceinline__ void load_cr(const float* __restrict__ dY, size_t r0, int Cr, int j, int q) {
        static_assert(NRB == 1, "ragged row tiles: one row block");
        constexpr int Ho = 16 * HB;
        const int jr = j < Cr ? j : Cr - 1;
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = 4 * q + t;
                const float val = dY[(r0 + (row < Cr ? row : Cr - 1)) * Ho + 16 * hb + j];
                d[0][hb][t] = row < Cr ? val : 0.f;
            }
            const float4 x = *reinterpret_cast<const float4*>(dY + (r0 + jr) * Ho + 16 * hb + 4 * q);
            v[0][hb] = j < Cr ? f32x4{x.x, x.y, x.z, x.w} : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
};

// Prologue of the gates convolution's backward (autograd of reference STC_GNN.py:71-75, hidden 16): instead of
// reading a precomputed dY, build it per node from the gradient of [Xt | R*H | pad] and the saved gates,
//   dY[:, :16] = dU * U * (1 - U),   dY[:, 16:] = dCandIn[:, cin:cin+16] * H * R * (1 - R),
// and emit the two by-products dXt = dCandIn[:, :cin] (optional: the caller may read those columns in place) and
// dH = dCandIn[h part] * R + dH_in on the way, so the separate gate-backward pass (and the dG round trip through HBM)
// disappears.  With dh_scaled, dH_in is the gradient of the new state itself and is taken times (1 - U) -- the state's
// share of the GRU blend (reference STC_GNN.py:78) -- so the blend backward need not write that product.
//
// With Cand given (and dU null), dH_in is the gradient dHnew of the cell's new state and the prologue also forms the
// blend's two other products itself: dU = dHnew * (Cand - H) and the state share dHnew * (1 - U) (dh_scaled implied).
// PRO_BLEND is the candidate convolution's counterpart (hidden 16, Ho = 16): dY = dHnew * U * (1 - Cand^2) is formed per
// node, so the blend backward pass (stc_gru_blend_bwd_f32) is not run at all.
enum { PRO_NONE = 0, PRO_GATES = 1, PRO_BLEND = 2, PRO_GATES_CAND = 3 };     // _CAND: dU formed here from (dHnew, Cand, H)
struct BwdPro {
    const float *dCandIn, *dU, *H, *U, *R, *dH_in;   // (nodes,C,L) and (nodes,C,16) operands; dH_in may be null / alias dH
    const float* Cand;                                // tanh(candidate): PRO_BLEND, and PRO_GATES when dU is to be formed here
    float *dXt, *dH;                                  // (nodes,C,cin) or null, (nodes,C,16)
    int cin;
    int dh_scaled;                                    // dH_in enters as dH_in * (1 - U)
    const float* zmax;                                // fp16 x 2, optional: (2 K, 256) slots of max |plane| of the launch's input planes (rows in the
                                                      //   launch order of Z, as the forward launch left them): scales of the dW products' activation operands
};

// FOLD (planar kernels): the state's share dH is not written to HBM; it is parked in a lane-private LDS slot (stash[kb * 64 + lane],
// same lane writes and reads: no barrier) and becomes the initial value of the H plane's gradient tile, whose register layout --
// lane (j, q) = row 16kb + j, columns 4q .. 4q+3 -- is exactly this one: one plane less written here and one addend less read by the
// state-gradient SpMM, at no live register across the kernel's body (keeping the eight values in registers spilled at the 256 cap).
template <int NRB, int HB, int L, bool FROM_CAND, bool FOLD = false>      // FROM_CAND is a compile-time switch: a run-time branch here would
__device__ __forceinline__ void load_gates_grad(DyFrag<NRB, HB>& g, const BwdPro& p, int node, int j, int q, float4* stash = nullptr) {      // split the load batches
    static_assert(HB == 2, "gates prologue needs Ho = 2 * 16");
    constexpr int C = 16 * NRB, HID = 16;
    const size_t r0 = (size_t)node * C;
#pragma unroll
    for (int kb = 0; kb < NRB; ++kb) {
        // accumulator-style layout: rows 16kb + 4q + t, column j
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const size_t row = r0 + 16 * kb + 4 * q + t, e = row * HID + j;
            const float u = p.U[e], r = p.R[e];
            const float du_ = FROM_CAND ? p.dH_in[e] * (p.Cand[e] - p.H[e]) : p.dU[e];
            g.d[kb][0][t] = du_ * u * (1.f - u);
            g.d[kb][1][t] = p.dCandIn[row * L + p.cin + j] * p.H[e] * r * (1.f - r);
        }
        // row-on-lane layout: row 16kb + j, columns 4q .. 4q+3; this is also where dXt and dH are produced
        const size_t row = r0 + 16 * kb + j, e = row * HID + 4 * q;
        const float4 u = *reinterpret_cast<const float4*>(p.U + e), r = *reinterpret_cast<const float4*>(p.R + e);
        const float4 hh = *reinterpret_cast<const float4*>(p.H + e);
        float4 du, own = make_float4(0.f, 0.f, 0.f, 0.f);        // own: what H is owed besides the reset-gate path
        if (FROM_CAND) {
            const float4 gn = *reinterpret_cast<const float4*>(p.dH_in + e), cd = *reinterpret_cast<const float4*>(p.Cand + e);
            du = make_float4(gn.x * (cd.x - hh.x), gn.y * (cd.y - hh.y), gn.z * (cd.z - hh.z), gn.w * (cd.w - hh.w));
            own = make_float4(gn.x * (1.f - u.x), gn.y * (1.f - u.y), gn.z * (1.f - u.z), gn.w * (1.f - u.w));
        } else {
            du = *reinterpret_cast<const float4*>(p.dU + e);
            if (p.dH_in) {
                const float4 o = *reinterpret_cast<const float4*>(p.dH_in + e);
                own = p.dh_scaled ? make_float4(o.x * (1.f - u.x), o.y * (1.f - u.y), o.z * (1.f - u.z), o.w * (1.f - u.w)) : o;
            }
        }
        const float* cr = p.dCandIn + row * L;
        float4 d;
        if ((p.cin & 3) == 0) d = *reinterpret_cast<const float4*>(cr + p.cin + 4 * q);
        else d = make_float4(cr[p.cin + 4 * q], cr[p.cin + 4 * q + 1], cr[p.cin + 4 * q + 2], cr[p.cin + 4 * q + 3]);
        g.v[kb][0] = f32x4{du.x * u.x * (1.f - u.x), du.y * u.y * (1.f - u.y), du.z * u.z * (1.f - u.z), du.w * u.w * (1.f - u.w)};
        g.v[kb][1] = f32x4{d.x * hh.x * r.x * (1.f - r.x), d.y * hh.y * r.y * (1.f - r.y), d.z * hh.z * r.z * (1.f - r.z), d.w * hh.w * r.w * (1.f - r.w)};
        const float4 dh = make_float4(d.x * r.x + own.x, d.y * r.y + own.y, d.z * r.z + own.z, d.w * r.w + own.w);
        if constexpr (FOLD) stash[kb * 64 + (q * 16 + j)] = dh;
        else *reinterpret_cast<float4*>(p.dH + e) = dh;
        if (p.dXt)
            for (int c0 = 4 * q; c0 < p.cin; c0 += 16)
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    if (c0 + x < p.cin) p.dXt[row * p.cin + c0 + x] = cr[c0 + x];
    }
}

// dY of the candidate convolution from the blend (reference STC_GNN.py:76-78): dHnew * U * (1 - Cand^2), both layouts
template <int NRB>
__device__ __forceinline__ void load_blend_grad(DyFrag<NRB, 1>& g, const BwdPro& p, int node, int j, int q) {
    constexpr int C = 16 * NRB, HID = 16;
    const size_t r0 = (size_t)node * C;
#pragma unroll
    for (int kb = 0; kb < NRB; ++kb) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const size_t e = (r0 + 16 * kb + 4 * q + t) * HID + j;
            const float c = p.Cand[e];
            g.d[kb][0][t] = p.dH_in[e] * p.U[e] * (1.f - c * c);
        }
        const size_t e = (r0 + 16 * kb + j) * HID + 4 * q;
        const float4 gn = *reinterpret_cast<const float4*>(p.dH_in + e), u = *reinterpret_cast<const float4*>(p.U + e);
        const float4 c = *reinterpret_cast<const float4*>(p.Cand + e);
        g.v[kb][0] = f32x4{gn.x * u.x * (1.f - c.x * c.x), gn.y * u.y * (1.f - c.y * c.y), gn.z * u.z * (1.f - c.z * c.z), gn.w * u.w * (1.f - c.w * c.w)};
    }
}

// Planar layout with a narrow input plane (layer 0: cin <= 4 input columns): the slab is read as [state plane (16) | input
// plane (cin) | pad], i.e. with the reference's [Xt | H] column order swapped.  W keeps its reference row order; this maps a
// slab column to the W row it multiplies (-1: padding).
__host__ __device__ __forceinline__ int stc_wrow_swapped(int col, int cin) {
    return col < 16 ? cin + col : (col < 16 + cin ? col - 16 : -1);
}

// End of a backward kernel: the workgroup's four waves hold dW tiles (rows l = 16lb + 4g + r, columns o = 16hb + x) and
// db partial sums in registers; combine them through LDS in a fixed order (bitwise reproducible) into ONE partial row
// [dW in W layout | db] per workgroup, which bdg_node_reduce_kernel then sums over workgroups.  (WAVES: waves per workgroup.)
// Called once per flush segment of the fp16 x 2 kernels (normally: once): segments after the first add to the row.
template <int K, int LB>
struct PlaneUnscale {     // fp16 x 2: 1 / (activation scale) of the plane that is block lb of slab n (the A operands of the dW tiles [n][lb][.][.])
    float v[K][LB];
    __device__ __forceinline__ PlaneUnscale() {
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int lb = 0; lb < LB; ++lb) v[n][lb] = 1.f;
    }
};

template <int K, int LB, int HB, int WAVES = MF_WAVES>
__device__ __forceinline__ void combine_dw(float* smem, const f32x4 (&dWt)[K][LB][K][HB], const float (&dbp)[HB],
                                           float* __restrict__ partial, int Lw, int want_db, int swapped_cin = -1,
                                           float unscale0 = 1.f, float unscale1 = 1.f, float db_unscale = 1.f,      // scaled operand formats: factor of the tiles of block c = 0 / c >= 1
                                           const PlaneUnscale<K, LB>& pu = PlaneUnscale<K, LB>(),
                                           bool accumulate = false) {                        // a flush segment after the first (stc_x3_frag.h: RunScale): add to the row
    constexpr int Ho = 16 * HB;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nW = K * K * Lw * Ho;
    __syncthreads();                                   // every wave is done with the fragment tables
    float* slab = smem + (size_t)wave * (nW + Ho);
#pragma unroll
    for (int n = 0; n < K; ++n)
#pragma unroll
        for (int lb = 0; lb < LB; ++lb)
#pragma unroll
            for (int c = 0; c < K; ++c)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int col = 16 * lb + 4 * q + r;
                        const int l = swapped_cin < 0 ? col : stc_wrow_swapped(col, swapped_cin);      // slab column -> W row
                        if (l >= 0 && l < Lw) slab[((n * K + c) * Lw + l) * Ho + 16 * hb + j] = (dWt[n][lb][c][hb][r] * (c == 0 ? unscale0 : unscale1)) * pu.v[n][lb];      // (one factor at a time: their product alone may underflow)
                    }
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
        float v = dbp[hb];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (q == 0) slab[nW + 16 * hb + j] = v * db_unscale;
    }
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * (nW + Ho);
    for (int e = tid; e < nW + Ho; e += WAVES * 64) {
        float s = smem[e];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) s += smem[(size_t)w * (nW + Ho) + e];
        if (e >= nW && !want_db) s = 0.f;
        out[e] = accumulate ? out[e] + s : s;
    }
}

inline bool all_aligned16(const float* const* p, int n) {
    for (int i = 0; i < n; ++i)
        if (!stc::aligned16(p[i])) return false;
    return true;
}

}  // namespace
