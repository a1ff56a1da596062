// One STC_Cell step of a SMALL graph in one launch (reference STC_GNN.py:65-79 with BDG_Dif :31-47 inside, Ks = 2), and its autograd in
// one launch: the SF-incidents shape (N = 100, C = 5, hidden 16), where a cell is ~15 launches of a few microseconds on
// the general path and the step is bound by the host's launch rate.
//
// One workgroup = one sample.  The sample's planes (N*C rows of 16 or cin floats: 32 KB at the SF shape) stay in L1 / L2; the phases of
// the cell follow each other inside the launch, separated by workgroup barriers where a phase reads its neighbours' rows:
//   forward   1  Zg = S.[H | X]                          (gather over the CSR rows; LP = 16 + 4 XQ columns, zero padded)
//             2  gates: [H|X], Zg -> project (fp32 MFMA) -> category mix -> sigmoid -> U, R, R*H
//             3  Zc = S.(R*H)                            (the X part of S.[X | R*H] is Zg's)
//             4  candidate: [R*H|X], [Zc|Zg.x] -> project -> mix -> tanh -> blend -> Cand, Hnew
//   backward  1  dY = dHnew U (1 - Cand^2) -> mix^T -> dWc, dbc partials; dZc_0, dZc_1
//             2  d[R*H | X] = dZc_0 + S^T dZc_1; gate backward -> dYg, first share of dH, dX
//             3  dYg -> mix^T -> dWg, dbg partials; dZg_0, dZg_1
//             4  dH, dX += dZg_0 + S^T dZg_1
// Projections are project-then-mix:  V_kc = sum_ks Z_ks . W[(ks,kc,:)],  Y[(n,c')] = V_0 + sum_{kc>=1} sum_c T_kc[c,c'] V_kc[(n,c)] + b,
// on v_mfma_f32_16x16x4_f32 (fp32 operands and accumulator: an fmaf chain per element, no split format).  A row tile = the C rows of
// floor(16 / C) whole nodes, so the category mix stays inside a wave: the V_kc accumulators go through a per-wave LDS tile.  W lives in
// registers in operand order for the whole phase (64 registers for the gates).  Parameter gradients are accumulated per SAMPLE into
// (batch, P) partials (a workgroup owns its row: deterministic, no atomics); the caller sums them over the batch once per backward pass.
#include "stc_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int SC_THREADS = 512, SC_WAVES = SC_THREADS / 64, SC_H = 16, SC_KS = 2;
constexpr int SC_MAXC = 16, SC_MAXKC = 3;

// Probe hook (tools/probes/small_cell_phases.py builds this file with -DSC_STOP_AFTER=n and times the truncated launches; the
// library is built without it: the condition is a compile-time false).
#ifndef SC_STOP_AFTER
#define SC_STOP_AFTER 99
#endif
#define SC_PHASE_END(n) do { if (SC_STOP_AFTER <= (n)) return; } while (0)

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

struct SmallGraph {
    const int32_t* rowptr;
    const int32_t* colidx;
    const float* val;
};

// out[row][quad] = base + sum_e val[e] * fetch(colidx[e] * C + c, quad)  for the rows of one sample; fetch returns 4 columns of a source row.
template <class Fetch, class Base, class Store>
__device__ __forceinline__ void aggregate_rows(const SmallGraph& g, int NC, int C, int quads, Fetch fetch, Base base, Store store) {
    for (int item = threadIdx.x; item < NC * quads; item += SC_THREADS) {
        const int row = item / quads, q = item - row * quads;
        const int n = row / C, c = row - n * C;
        f32x4 s = base(row, q);
        const int e1 = g.rowptr[n + 1];
        for (int e = g.rowptr[n]; e < e1; ++e) {
            const float v = g.val[e];
            const f32x4 x = fetch(g.colidx[e] * C + c, q);
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] = fmaf(v, x[i], s[i]);
        }
        store(row, q, s);
    }
}

// The two slabs of a convolution's input as the kernel reads them: slab 0 = [P0h (16 columns, row stride 16) | P0x (cin columns, row
// stride cin)], slab 1 = [P1h (row stride ld1h) | P1x (row stride ld1x, zero padded to 4 XQ columns)].
struct Slabs {
    const float* P0h;
    const float* P0x;
    const float* P1h;
    int ld1h;
    const float* P1x;
    int ld1x;
};

// W (Ks*Kc*L, HO) in B-operand order for the forward products: step s < 4 of slab ks feeds l = cin + 4 kq + s (the H block: one
// 16-byte load of a row gives a lane its A operands of four steps; any bijection of the contraction index serves a sum), steps 4.. the X
// block (wide: l = 4 kq + s; narrow: l = 4 s + kq).
template <int KC, int XQ, int CT>
__device__ __forceinline__ void load_w_fwd(float (&Wr)[SC_KS][KC][CT][4 + (XQ == 4 ? 4 : XQ)], const float* __restrict__ W, int cin, int j, int kq) {
    constexpr int XS = XQ == 4 ? 4 : XQ, HO = 16 * CT;
    const int L = cin + SC_H;
#pragma unroll
    for (int ks = 0; ks < SC_KS; ++ks)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const float* Wb = W + (size_t)((ks * KC + kc) * L) * HO + ct * 16 + j;
#pragma unroll
                for (int s = 0; s < 4; ++s) Wr[ks][kc][ct][s] = Wb[(size_t)(cin + 4 * kq + s) * HO];
#pragma unroll
                for (int s = 0; s < XS; ++s) {
                    const int l = XQ == 4 ? 4 * kq + s : 4 * s + kq;
                    Wr[ks][kc][ct][4 + s] = l < cin ? Wb[(size_t)l * HO] : 0.f;
                }
            }
}

// One row tile of a forward convolution: acc[kc][ct] (lane (j, kq): rows 4 kq .. 4 kq + 3 of the tile, column 16 ct + j) = V_kc, then the
// category mix through the wave's LDS tile; epi(local row, global row, r, y[CT]) receives the finished pre-activations (bias not added).
template <int KC, int XQ, int CT, class Epi>
__device__ __forceinline__ void project_tile(const Slabs& z, const float (&Wr)[SC_KS][KC][CT][4 + (XQ == 4 ? 4 : XQ)], const float* __restrict__ Tl,
                                             float* __restrict__ ms, int row0, int RPT, int NC, int C, int cin, int j, int kq, Epi epi) {
    constexpr int XS = XQ == 4 ? 4 : XQ, MS = 16 * CT + 1;
    const int row = row0 + j;
    const bool ok = j < RPT && row < NC;
    f32x4 acc[KC][CT];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[kc][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < SC_KS; ++ks) {
        f32x4 ah = {0.f, 0.f, 0.f, 0.f};
        float ax[XS];
#pragma unroll
        for (int s = 0; s < XS; ++s) ax[s] = 0.f;
        if (ok) {
            if (ks == 0) {
                ah = ld4(z.P0h + (size_t)row * SC_H + 4 * kq);
                if (XQ == 4) {
                    const f32x4 v = ld4(z.P0x + (size_t)row * SC_H + 4 * kq);
#pragma unroll
                    for (int s = 0; s < XS; ++s) ax[s] = v[s];
                } else {
#pragma unroll
                    for (int s = 0; s < XS; ++s)
                        if (4 * s + kq < cin) ax[s] = z.P0x[(size_t)row * cin + 4 * s + kq];
                }
            } else {
                ah = ld4(z.P1h + (size_t)row * z.ld1h + 4 * kq);
                if (XQ == 4) {
                    const f32x4 v = ld4(z.P1x + (size_t)row * z.ld1x + 4 * kq);
#pragma unroll
                    for (int s = 0; s < XS; ++s) ax[s] = v[s];
                } else {
#pragma unroll
                    for (int s = 0; s < XS; ++s) ax[s] = z.P1x[(size_t)row * z.ld1x + 4 * s + kq];
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[kc][ct] = mfma4(ah[s], Wr[ks][kc][ct][s], acc[kc][ct]);
#pragma unroll
        for (int s = 0; s < XS; ++s)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[kc][ct] = mfma4(ax[s], Wr[ks][kc][ct][4 + s], acc[kc][ct]);
    }
#pragma unroll
    for (int kc = 1; kc < KC; ++kc)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) ms[((kc - 1) * 16 + 4 * kq + r) * MS + 16 * ct + j] = acc[kc][ct][r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int lrow = 4 * kq + r, grow = row0 + lrow;
        if (lrow < RPT && grow < NC) {
            const int nl = lrow / C, cp = lrow - nl * C;
            float y[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) y[ct] = acc[0][ct][r];
#pragma unroll
            for (int kc = 1; kc < KC; ++kc)
                for (int c = 0; c < C; ++c) {
                    const float tt = Tl[(kc * C + c) * C + cp];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) y[ct] = fmaf(tt, ms[((kc - 1) * 16 + nl * C + c) * MS + 16 * ct + j], y[ct]);
                }
            epi(grow, y);
        }
    }
    __builtin_amdgcn_wave_barrier();            // the next tile overwrites ms
}

struct SmallFwd {
    SmallGraph g;
    const float *X, *H, *Tc, *Wg, *bg, *Wc, *bc;
    float *U, *R, *Cand, *Hnew, *RH, *Zg, *Zc;
    int N, C, cin, rpt, tiles;
};

template <int KC, int XQ>
__global__ __launch_bounds__(SC_THREADS) void small_fwd_kernel(SmallFwd a) {
    constexpr int LP = 16 + 4 * XQ, XS = XQ == 4 ? 4 : XQ;
    __shared__ float Tl[SC_MAXKC * SC_MAXC * SC_MAXC];
    __shared__ float mixs[SC_WAVES][(KC - 1) * 16 * 33];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, j = lane & 15, kq = lane >> 4;
    const int C = a.C, NC = a.N * C, cin = a.cin;
    const size_t r0 = (size_t)blockIdx.x * NC;
    const float* Xb = a.X + r0 * cin;
    const float* Hb = a.H + r0 * SC_H;
    float* Ub = a.U + r0 * SC_H;
    float* Rb = a.R + r0 * SC_H;
    float* RHb = a.RH + r0 * SC_H;           // (written in phase 2, read in 3 and 4: never through a __restrict__ / const path)
    float* Zgb = a.Zg + r0 * LP;
    float* Zcb = a.Zc + r0 * SC_H;
    for (int i = t; i < KC * C * C; i += SC_THREADS) Tl[i] = a.Tc[i];

    // 1: Zg = S.[H | X]
    aggregate_rows(a.g, NC, C, LP / 4,
        [&](int src, int q) -> f32x4 {
            if (q < 4) return ld4(Hb + (size_t)src * SC_H + 4 * q);
            if (XQ == 4) return ld4(Xb + (size_t)src * SC_H + 4 * (q - 4));
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * (q - 4) + i < cin) x[i] = Xb[(size_t)src * cin + 4 * (q - 4) + i];
            return x;
        },
        [](int, int) -> f32x4 { return f32x4{0.f, 0.f, 0.f, 0.f}; },
        [&](int row, int q, f32x4 s) { st4(Zgb + (size_t)row * LP + 4 * q, s); });
    __syncthreads();
    SC_PHASE_END(1);

    // 2: gates
    {
        float Wr[SC_KS][KC][2][4 + XS];
        load_w_fwd<KC, XQ, 2>(Wr, a.Wg, cin, j, kq);
        const float bu = a.bg ? a.bg[j] : 0.f, br = a.bg ? a.bg[SC_H + j] : 0.f;
        const Slabs z{Hb, Xb, Zgb, LP, Zgb + SC_H, LP};
        for (int tile = wave; tile < a.tiles; tile += SC_WAVES)
            project_tile<KC, XQ, 2>(z, Wr, Tl, mixs[wave], tile * a.rpt, a.rpt, NC, C, cin, j, kq, [&](int grow, const float (&y)[2]) {
                const size_t e = (size_t)grow * SC_H + j;
                const float u = sigm(y[0] + bu), rr = sigm(y[1] + br);
                Ub[e] = u;
                Rb[e] = rr;
                RHb[e] = rr * Hb[e];
            });
    }
    __syncthreads();
    SC_PHASE_END(2);

    // 3: Zc = S.(R*H)
    aggregate_rows(a.g, NC, C, 4, [&](int src, int q) -> f32x4 { return ld4(RHb + (size_t)src * SC_H + 4 * q); },
                   [](int, int) -> f32x4 { return f32x4{0.f, 0.f, 0.f, 0.f}; }, [&](int row, int q, f32x4 s) { st4(Zcb + (size_t)row * SC_H + 4 * q, s); });
    __syncthreads();
    SC_PHASE_END(3);

    // 4: candidate + blend
    {
        float Wr[SC_KS][KC][1][4 + XS];
        load_w_fwd<KC, XQ, 1>(Wr, a.Wc, cin, j, kq);
        const float bcj = a.bc ? a.bc[j] : 0.f;
        const Slabs z{RHb, Xb, Zcb, SC_H, Zgb + SC_H, LP};
        float* Cb = a.Cand + r0 * SC_H;
        float* Hn = a.Hnew + r0 * SC_H;
        for (int tile = wave; tile < a.tiles; tile += SC_WAVES)
            project_tile<KC, XQ, 1>(z, Wr, Tl, mixs[wave], tile * a.rpt, a.rpt, NC, C, cin, j, kq, [&](int grow, const float (&y)[1]) {
                const size_t e = (size_t)grow * SC_H + j;
                const float cd = tanhf(y[0] + bcj), u = Ub[e];
                Cb[e] = cd;
                Hn[e] = (1.f - u) * Hb[e] + u * cd;
            });
    }
}

// ------------------------------------------------------------------------------------------------------------------------ backward
// W in B-operand order for dZ = dV . W^T: step st feeds the contraction index (kc, o) = (st / (HO/4), 4 (st % (HO/4)) + kq); output
// column tile lt = 0: the H block (l = cin + j), lt = 1: the X block (l = j < cin).
template <int KC, int OT>
__device__ __forceinline__ void load_w_bwd(float (&WT)[SC_KS][2][KC * 4 * OT], const float* __restrict__ W, int cin, int j, int kq) {
    constexpr int HO = 16 * OT, SPK = HO / 4;
    const int L = cin + SC_H;
#pragma unroll
    for (int ks = 0; ks < SC_KS; ++ks)
#pragma unroll
        for (int lt = 0; lt < 2; ++lt) {
            const int l = lt == 0 ? cin + j : j;
            const bool ok = lt == 0 || j < cin;
#pragma unroll
            for (int st = 0; st < KC * SPK; ++st) {
                const int kc = st / SPK, o = 4 * (st % SPK) + kq;
                WT[ks][lt][st] = ok ? W[(size_t)((ks * KC + kc) * L + l) * HO + o] : 0.f;
            }
        }
}

// Backward of one convolution over the row tiles of a sample.  form_dy(local row, global row, quad) -> 4 columns of dY per column tile;
// the mixed gradients dV_kc live in the wave's LDS tile dv[kc][16][HO + 1]; dZ_0 / dZ_1 go to the workspace slabs, the dW / db partial
// sums stay in registers across the wave's tiles.
template <int KC, int XQ, int OT, class FormDy>
__device__ __forceinline__ void conv_bwd_phase(const Slabs& z, const float* __restrict__ W, const float* __restrict__ Tl, float* __restrict__ dv,
                                               float* __restrict__ red, float* __restrict__ dZ0, float* __restrict__ dZ1, float* __restrict__ dW,
                                               float* __restrict__ db, int rpt, int tiles, int NC, int C, int cin, FormDy form_dy) {
    constexpr int LP = 16 + 4 * XQ, HO = 16 * OT, DS = HO + 1, SPK = HO / 4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, j = lane & 15, kq = lane >> 4, L = cin + SC_H;
    float WT[SC_KS][2][KC * SPK];
    load_w_bwd<KC, OT>(WT, W, cin, j, kq);
    f32x4 dw[SC_KS][2][KC][OT];
    float dbv[OT];
#pragma unroll
    for (int ks = 0; ks < SC_KS; ++ks)
#pragma unroll
        for (int lt = 0; lt < 2; ++lt)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) dw[ks][lt][kc][ot] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) dbv[ot] = 0.f;

    for (int tile = wave; tile < tiles; tile += SC_WAVES) {
        const int row0 = tile * rpt;
        {   // dY of the tile, row layout: lane -> (row lane / 4, columns 4 (lane % 4) .. + 3 of every column tile)
            const int rr = lane >> 2, qd = lane & 3, grow = row0 + rr;
            const bool ok = rr < rpt && grow < NC;
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                const f32x4 d = ok ? form_dy(grow, qd, ot) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) dv[rr * DS + 16 * ot + 4 * qd + i] = d[i];
            }
        }
        __builtin_amdgcn_wave_barrier();
        // mix^T: dV_kc[(n, c)] = sum_c' T_kc[c, c'] dY[(n, c')]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int lrow = 4 * kq + r;
            const int nl = lrow / C, c = lrow - nl * C;
            const bool ok = lrow < rpt && row0 + lrow < NC;
#pragma unroll
            for (int kc = 1; kc < KC; ++kc) {
                float s[OT];
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) s[ot] = 0.f;
                if (ok)
                    for (int cp = 0; cp < C; ++cp) {
                        const float tt = Tl[(kc * C + c) * C + cp];
#pragma unroll
                        for (int ot = 0; ot < OT; ++ot) s[ot] = fmaf(tt, dv[(nl * C + cp) * DS + 16 * ot + j], s[ot]);
                    }
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) dv[(kc * 16 + lrow) * DS + 16 * ot + j] = s[ot];
            }
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) dbv[ot] += dv[lrow * DS + 16 * ot + j];
        }
        __builtin_amdgcn_wave_barrier();
        // dZ_ks = sum_kc dV_kc . W[(ks, kc, :)]^T
        {
            f32x4 dz[SC_KS][2];
#pragma unroll
            for (int ks = 0; ks < SC_KS; ++ks)
#pragma unroll
                for (int lt = 0; lt < 2; ++lt) dz[ks][lt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < KC * SPK; ++st) {
                const int kc = st / SPK, o = 4 * (st % SPK) + kq;
                const float av = dv[(kc * 16 + j) * DS + o];
#pragma unroll
                for (int ks = 0; ks < SC_KS; ++ks)
#pragma unroll
                    for (int lt = 0; lt < 2; ++lt) dz[ks][lt] = mfma4(av, WT[ks][lt][st], dz[ks][lt]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lrow = 4 * kq + r, grow = row0 + lrow;
                if (lrow < rpt && grow < NC) {
                    dZ0[(size_t)grow * LP + j] = dz[0][0][r];
                    dZ1[(size_t)grow * LP + j] = dz[1][0][r];
                    if (j < LP - 16) {
                        dZ0[(size_t)grow * LP + 16 + j] = dz[0][1][r];
                        dZ1[(size_t)grow * LP + 16 + j] = dz[1][1][r];
                    }
                }
            }
        }
        // dW[(ks, kc, l), o] += sum_rows Z_ks[row, l] dV_kc[row, o]: the rows are the contraction (4 steps of 4 rows)
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int lrow = 4 * st + kq, grow = row0 + lrow;
            const bool ok = lrow < rpt && grow < NC;
            float az[SC_KS][2];
            az[0][0] = ok ? z.P0h[(size_t)grow * SC_H + j] : 0.f;
            az[0][1] = ok && j < cin ? z.P0x[(size_t)grow * cin + j] : 0.f;
            az[1][0] = ok ? z.P1h[(size_t)grow * z.ld1h + j] : 0.f;
            az[1][1] = ok && j < cin ? z.P1x[(size_t)grow * z.ld1x + j] : 0.f;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) {
                    const float bv = dv[(kc * 16 + lrow) * DS + 16 * ot + j];
#pragma unroll
                    for (int ks = 0; ks < SC_KS; ++ks)
#pragma unroll
                        for (int lt = 0; lt < 2; ++lt) dw[ks][lt][kc][ot] = mfma4(az[ks][lt], bv, dw[ks][lt][kc][ot]);
                }
        }
        __builtin_amdgcn_wave_barrier();        // the next tile overwrites dv
    }

    // the waves' partial sums, added in wave order (deterministic), then into the sample's row of the parameter-gradient partials
    constexpr int NW = SC_KS * 2 * KC * OT * 4;
    for (int w = 0; w < SC_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int ks = 0; ks < SC_KS; ++ks)
#pragma unroll
                for (int lt = 0; lt < 2; ++lt)
#pragma unroll
                    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float* p = red + ((((ks * 2 + lt) * KC + kc) * OT + ot) * 4 + r) * 64 + lane;
                                *p = w == 0 ? dw[ks][lt][kc][ot][r] : *p + dw[ks][lt][kc][ot][r];
                            }
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                float* p = red + (NW + ot) * 64 + lane;
                *p = w == 0 ? dbv[ot] : *p + dbv[ot];
            }
        }
        __syncthreads();
    }
    for (int i = t; i < NW * 64; i += SC_THREADS) {
        const int ln = i & 63, idx = i >> 6, r = idx & 3, ot = (idx >> 2) % OT, kc = (idx / (4 * OT)) % KC, lt = (idx / (4 * OT * KC)) & 1,
                  ks = idx / (8 * OT * KC);
        const int li = 4 * (ln >> 4) + r, o = 16 * ot + (ln & 15);
        if (lt == 1 && li >= cin) continue;
        const int l = lt == 0 ? cin + li : li;
        dW[(size_t)((ks * KC + kc) * L + l) * HO + o] += red[i];
    }
    if (db != nullptr)
        for (int o = t; o < HO; o += SC_THREADS) {
            const float* p = red + (NW + o / 16) * 64 + (o & 15);
            db[o] += (p[0] + p[16]) + (p[32] + p[48]);
        }
    __syncthreads();                             // red and dv are reused by the next phase
}

struct SmallBwd {
    SmallGraph g;                                // CSR of Gs (the transpose of the forward's)
    const float *X, *H, *Tc, *Wg, *Wc, *U, *R, *Cand, *RH, *Zg, *Zc, *dHnew;
    float *dX, *dH, *dP, *ws;
    int N, C, cin, rpt, tiles, acc_x, acc_h, has_bg, has_bc;
    long long P;                                 // floats per sample in dP: [dWg | dbg (32) | dWc | dbc (16)]
};

template <int KC, int XQ>
__global__ __launch_bounds__(SC_THREADS) void small_bwd_kernel(SmallBwd a) {
    constexpr int LP = 16 + 4 * XQ;
    __shared__ float Tl[SC_MAXKC * SC_MAXC * SC_MAXC];
    __shared__ float dvs[SC_WAVES][KC * 16 * 33];
    __shared__ float red[(SC_KS * 2 * KC * 2 * 4 + 2) * 64];
    const int t = threadIdx.x, wave = t >> 6;
    const int C = a.C, NC = a.N * C, cin = a.cin, L = cin + SC_H;
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
    float* wsb = a.ws + (size_t)blockIdx.x * NC * (2 * LP + 32);
    float* dZ0 = wsb;
    float* dZ1 = wsb + (size_t)NC * LP;
    float* dYg = wsb + (size_t)NC * 2 * LP;
    float* dPb = a.dP + (size_t)blockIdx.x * a.P;
    float* dWg = dPb;
    float* dbg = dPb + (size_t)SC_KS * KC * L * 32;
    float* dWc = dbg + 32;
    float* dbc = dWc + (size_t)SC_KS * KC * L * 16;
    for (int i = t; i < KC * C * C; i += SC_THREADS) Tl[i] = a.Tc[i];
    __syncthreads();

    // 1: candidate convolution
    conv_bwd_phase<KC, XQ, 1>(Slabs{RHb, Xb, Zcb, SC_H, Zgb + SC_H, LP}, a.Wc, Tl, dvs[wave], red, dZ0, dZ1, dWc, a.has_bc ? dbc : nullptr, a.rpt,
                              a.tiles, NC, C, cin, [&](int grow, int qd, int) -> f32x4 {
                                  const size_t e = (size_t)grow * SC_H + 4 * qd;
                                  const f32x4 d = ld4(dHn + e), u = ld4(Ub + e), cd = ld4(Cb + e);
                                  f32x4 y;
#pragma unroll
                                  for (int i = 0; i < 4; ++i) y[i] = d[i] * u[i] * (1.f - cd[i] * cd[i]);
                                  return y;
                              });
    // (conv_bwd_phase ends on a workgroup barrier: the dZ slabs are complete)
    SC_PHASE_END(1);

    // 2: d[R*H | X] = dZ_0 + S^T dZ_1, gate backward
    aggregate_rows(a.g, NC, C, LP / 4, [&](int src, int q) -> f32x4 { return ld4(dZ1 + (size_t)src * LP + 4 * q); },
        [&](int row, int q) -> f32x4 { return ld4(dZ0 + (size_t)row * LP + 4 * q); },
        [&](int row, int q, f32x4 s) {
            if (q < 4) {
                const size_t e = (size_t)row * SC_H + 4 * q;
                const f32x4 hh = ld4(Hb + e), rr = ld4(Rb + e), u = ld4(Ub + e), cd = ld4(Cb + e), d = ld4(dHn + e);
                f32x4 gu, gr, dh;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    gu[i] = d[i] * (cd[i] - hh[i]) * u[i] * (1.f - u[i]);
                    gr[i] = s[i] * hh[i] * rr[i] * (1.f - rr[i]);
                    dh[i] = fmaf(d[i], 1.f - u[i], s[i] * rr[i]);
                }
                st4(dYg + (size_t)row * 32 + 4 * q, gu);
                st4(dYg + (size_t)row * 32 + 16 + 4 * q, gr);
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
                        float* p = dXb + (size_t)row * cin + col;
                        *p = a.acc_x ? *p + s[i] : s[i];
                    }
                }
            }
        });
    __syncthreads();
    SC_PHASE_END(2);

    // 3: gates convolution
    conv_bwd_phase<KC, XQ, 2>(Slabs{Hb, Xb, Zgb, LP, Zgb + SC_H, LP}, a.Wg, Tl, dvs[wave], red, dZ0, dZ1, dWg, a.has_bg ? dbg : nullptr, a.rpt, a.tiles,
                              NC, C, cin, [&](int grow, int qd, int ot) -> f32x4 { return ld4(dYg + (size_t)grow * 32 + 16 * ot + 4 * qd); });

    SC_PHASE_END(3);
    // 4: d[H | X] += dZ_0 + S^T dZ_1
    if (dHb || dXb)
        aggregate_rows(a.g, NC, C, LP / 4, [&](int src, int q) -> f32x4 { return ld4(dZ1 + (size_t)src * LP + 4 * q); },
            [&](int row, int q) -> f32x4 { return ld4(dZ0 + (size_t)row * LP + 4 * q); },
            [&](int row, int q, f32x4 s) {
                if (q < 4) {
                    if (dHb) {
                        float* p = dHb + (size_t)row * SC_H + 4 * q;
                        const f32x4 o = ld4(p);
#pragma unroll
                        for (int i = 0; i < 4; ++i) s[i] += o[i];
                        st4(p, s);
                    }
                } else if (dXb) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int col = 4 * (q - 4) + i;
                        if (col < cin) dXb[(size_t)row * cin + col] += s[i];
                    }
                }
            });
}

int xq_of(int cin) { return cin == SC_H ? 4 : (cin >= 1 && cin <= 4 ? 1 : 0); }

}  // namespace

extern "C" int stc_cell_small_supported(int32_t Ks, int32_t Kc, int32_t C, int32_t cin, int32_t h) {
    return Ks == SC_KS && Kc == 2 && C >= 1 && C <= SC_MAXC && h == SC_H && xq_of(cin) != 0;
}

extern "C" size_t stc_cell_small_workspace_bytes(int32_t n_nodes, int32_t C, int32_t cin, int32_t batch) {
    const int xq = xq_of(cin);
    if (xq == 0 || n_nodes < 0 || C < 0 || batch < 0) return 0;
    return (size_t)batch * n_nodes * C * (2 * (16 + 4 * xq) + 32) * sizeof(float);
}

#define SC_COMMON_CHECKS(name)                                                                                                        \
    STC_REQUIRE(n_nodes >= 0 && batch >= 0, STC_EINVAL, name ": negative size (n_nodes=%d batch=%d)", n_nodes, batch);                 \
    STC_REQUIRE(stc_cell_small_supported(SC_KS, Kc, C, cin, SC_H), STC_EINVAL, name ": unsupported shape (Kc=%d C=%d cin=%d)", Kc, C, cin); \
    STC_REQUIRE((long long)n_nodes * C * batch < (1ll << 26), STC_ELIMIT, name ": %lld rows: not a small graph", (long long)n_nodes * C * batch); \
    if (n_nodes == 0 || batch == 0) return STC_OK;

extern "C" int stc_cell_small_fwd_f32(const int32_t* rowptr, const int32_t* colidx, const float* val, int32_t n_nodes, const float* X, int32_t cin,
                                      const float* H, const float* Tc, int32_t Kc, const float* Wg, const float* bg, const float* Wc,
                                      const float* bc, float* U, float* R, float* Cand, float* Hnew, float* RH, float* Zg, float* Zc,
                                      int32_t batch, int32_t C, void* stream) {
    SC_COMMON_CHECKS("stc_cell_small_fwd_f32")
    STC_REQUIRE(rowptr && colidx && val && X && H && Tc && Wg && Wc && U && R && Cand && Hnew && RH && Zg && Zc, STC_EINVAL,
                "stc_cell_small_fwd_f32: null operand");
    STC_REQUIRE(Hnew != H, STC_EINVAL, "stc_cell_small_fwd_f32: Hnew must not alias H (neighbour rows are read after the first rows are written)");
    const int xq = xq_of(cin);
    STC_REQUIRE(stc::aligned16(H) && stc::aligned16(U) && stc::aligned16(R) && stc::aligned16(RH) && stc::aligned16(Zg) && stc::aligned16(Zc) &&
                    (xq != 4 || stc::aligned16(X)), STC_EINVAL, "stc_cell_small_fwd_f32: planes must be 16-byte aligned");
    const int npt = 16 / C, rpt = npt * C;
    SmallFwd a{{rowptr, colidx, val}, X, H, Tc, Wg, bg, Wc, bc, U, R, Cand, Hnew, RH, Zg, Zc, n_nodes, C, cin, rpt, (n_nodes + npt - 1) / npt};
    auto kern = xq == 4 ? small_fwd_kernel<2, 4> : small_fwd_kernel<2, 1>;
    hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(SC_THREADS), 0, static_cast<hipStream_t>(stream), a);
    STC_LAUNCH_CHECK("stc_cell_small_fwd_f32 launch");
    return STC_OK;
}

extern "C" int stc_cell_small_bwd_f32(const int32_t* rowptr, const int32_t* colidx, const float* val, int32_t n_nodes, const float* X, int32_t cin,
                                      const float* H, const float* Tc, int32_t Kc, const float* Wg, const float* Wc, const float* U,
                                      const float* R, const float* Cand, const float* RH, const float* Zg, const float* Zc, const float* dHnew,
                                      float* dX, int32_t accumulate_x, float* dH, int32_t accumulate_h, float* dparams, int64_t params_ld,
                                      int32_t has_bg, int32_t has_bc, void* workspace, size_t workspace_bytes, int32_t batch, int32_t C,
                                      void* stream) {
    SC_COMMON_CHECKS("stc_cell_small_bwd_f32")
    STC_REQUIRE(rowptr && colidx && val && X && H && Tc && Wg && Wc && U && R && Cand && RH && Zg && Zc && dHnew && dparams && workspace, STC_EINVAL,
                "stc_cell_small_bwd_f32: null operand");
    const int xq = xq_of(cin), L = cin + SC_H;
    const long long P = (long long)SC_KS * Kc * L * 48 + 48;
    STC_REQUIRE(params_ld >= P, STC_EINVAL, "stc_cell_small_bwd_f32: params_ld %lld < %lld floats per sample", (long long)params_ld, P);
    STC_REQUIRE(workspace_bytes >= stc_cell_small_workspace_bytes(n_nodes, C, cin, batch), STC_EINVAL, "stc_cell_small_bwd_f32: workspace too small");
    STC_REQUIRE(stc::aligned16(H) && stc::aligned16(U) && stc::aligned16(R) && stc::aligned16(Cand) && stc::aligned16(RH) && stc::aligned16(Zg) &&
                    stc::aligned16(Zc) && stc::aligned16(dHnew) && stc::aligned16(workspace) && (dH == nullptr || stc::aligned16(dH)),
                STC_EINVAL, "stc_cell_small_bwd_f32: planes must be 16-byte aligned");
    STC_REQUIRE(dH != dHnew && (const float*)dX != dHnew, STC_EINVAL, "stc_cell_small_bwd_f32: dHnew must not alias an output");
    const int npt = 16 / C, rpt = npt * C;
    SmallBwd a{{rowptr, colidx, val}, X, H, Tc, Wg, Wc, U, R, Cand, RH, Zg, Zc, dHnew, dX, dH, dparams, static_cast<float*>(workspace),
               n_nodes, C, cin, rpt, (n_nodes + npt - 1) / npt, accumulate_x, accumulate_h, has_bg, has_bc, params_ld};
    auto kern = xq == 4 ? small_bwd_kernel<2, 4> : small_bwd_kernel<2, 1>;
    hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(SC_THREADS), 0, static_cast<hipStream_t>(stream), a);
    STC_LAUNCH_CHECK("stc_cell_small_bwd_f32 launch");
    return STC_OK;
}
