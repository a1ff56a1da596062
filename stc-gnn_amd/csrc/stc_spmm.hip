// CSR / row-blocked CSR SpMM and SDDMM for the spatial (1-mode) aggregation of STC-GNN on gfx950.
//
//   Y[b,i,:] = alpha * sum_j val[j] * X[b, col[j], :] + beta * Y0[b,i,:]
//
// Replaces torch.einsum('bncl,nm->bmcl', X, T_n(Gs)) (reference STC_GNN.py:37) and, with (alpha,beta) = (2,-1),
// one step of the Chebyshev recurrence of STC_GNN.py:28 applied on the feature side.  HBM-bound: a feature row
// is F = C*L contiguous floats (4 KiB at C=32, L=32), streamed with one 16-byte load per lane, 1 KiB per wave
// instruction.
//
// Layout of one launch (vector path, both kernels):
//   workgroup  = a run of consecutive output rows of one batch element
//   CSR stage  = row-pointer slice + the (col,val) segment of those rows -> LDS, one coalesced pass
//   wave       = one output row (CSR) or one block of 4 rows (BCSR) at a time; (col,val) read from LDS are
//                wave-uniform and moved to SGPRs (readfirstlane) so the neighbour-row base address is scalar
//   lane       = VPT float4 pieces of the row, 4 neighbour rows in flight (16 loads per lane)
//   blocks     = remapped so each XCD walks a contiguous band of rows: the rows a band gathers (its own +-
//                the graph bandwidth) stay in that XCD's 4 MiB L2
//   output     = non-temporal stores (written once; keeps the output stream from evicting the X rows the
//                neighbours still need from L2); Y0 read non-temporally for the same reason
//
// Epilogues: the GRU blend of the forward (EP_BLEND: Cand = tanh(A + S.Bm), Hnew = (1-U) H + U Cand, reference STC_GNN.py:76-78) and the
// state-gradient sums of the backward (EP_SUM / EP_SUM2: Y = sum of addend planes + alpha S.(X [+ X2]), optionally with the blend backward
// dY = Y U (1 - Cand^2) and the launch's max |Y| on the way) run on the accumulators, so neither intermediate is written to HBM.
#include "stc_common.h"

#ifndef STC_BCSR_DEFAULT_BLOCKS
#define STC_BCSR_DEFAULT_BLOCKS 2
#endif

namespace {

constexpr int SPMM_THREADS = 256;
constexpr int SPMM_WAVES = SPMM_THREADS / 64;
constexpr int SPMM_ROWS = 8;         // CSR kernel: output rows per workgroup
constexpr int SPMM_SEG_CAP = 1024;   // CSR entries staged in LDS per workgroup

using v4f = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float uniform_f(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

__device__ __forceinline__ void fma4(float4& acc, float s, const float4& x) {
    acc.x = fmaf(s, x.x, acc.x);
    acc.y = fmaf(s, x.y, acc.y);
    acc.z = fmaf(s, x.z, acc.z);
    acc.w = fmaf(s, x.w, acc.w);
}

__device__ __forceinline__ float4 nt_load4(const float4* p) {
    const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(t[0], t[1], t[2], t[3]);
}
__device__ __forceinline__ void nt_store4(float4* p, const float4& v) {
    __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f*>(p));
}

enum { EP_PLAIN = 0, EP_BLEND = 3, EP_SUM = 4, EP_SUM2 = 5 };

struct EpiArgs {
    const float4* Y0;                       // base term (PLAIN: optional, scaled by beta; BLEND: A)
    float4* Y;                              // PLAIN / SUM output
    float alpha, beta;                      // PLAIN (SUM: alpha)
    int C, L, cin, h;                       // row geometry: rows of C categories x h = 16 floats (L = h, cin = 0)
    const float *H, *U;                     // BLEND inputs, (rows*C, h)
    // BLEND (forward, post-aggregation candidate convolution): Y0 = A (bias included), rows of C*h floats;
    // Cand = tanh(A + S.Bm), Hnew = (1-U)*H + U*Cand, plus the state copies of stc_cell_blend_fwd_f32
    float *Cand, *Hnew;                     // (rows*C, h)
    float* copy[2]; int copy_ld[2], copy_off[2];
    const float* side_src; int side_cin;    // belongs to copy[0]
    // SUM / SUM2 (gradient of a state from its consumers' pieces): Y = sum_i add[i] + S.(X [+ X2]), rows of C*h floats;
    // add[i] = columns [off, off+h) of rows of ld floats (a contiguous plane: ld = h, off = 0)
    const float4* X2;                       // SUM2: second gathered operand (same batch stride as X)
    const float* add[STC_SPMM_SUM_MAX_ADD]; int add_ld[STC_SPMM_SUM_MAX_ADD], add_off[STC_SPMM_SUM_MAX_ADD], n_add;
    float add_scale[STC_SPMM_SUM_MAX_ADD];   // SUM: Y = sum_i add_scale[i] add[i] + alpha S.(X [+ X2])
    const float *gU, *gCand; float* dYout;  // SUM, optional: also dY = Y * U * (1 - Cand^2), the blend backward of the cell that owns the state
    unsigned* amax; int n_amax;             // SUM, optional: max |Y| of the launch as float bits, spread over n_amax slots (zero before the launch)
};

// SUM / SUM2 with ep.amax: every wave leaves the largest |Y| it produced in one of the slots (atomic max on the bits of a non-negative
// float, which order like the values) -- the consumer of Y (stc_cell_bwd_planar_f32, fp16 x 2 operand format) takes the maximum over the
// slots as the launch's gradient scale, without another pass over Y and without a host round trip.  Slots spread the atomics over lines.
__device__ __forceinline__ void publish_amax(const EpiArgs& ep, float m) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) {
        const unsigned slot = (blockIdx.x + blockIdx.y * gridDim.x + (threadIdx.x >> 6) * 7u) % (unsigned)ep.n_amax;
        atomicMax(ep.amax + slot, __float_as_uint(m));
    }
}

// the GRU blend on one 16-byte piece of a (row, category) state row, plus its copies (EP_BLEND)
__device__ __forceinline__ float tanh_fast(float v) { return stc_tanh(v); }      // hardware exp2 / rcp, polynomial below 1/4 (stc_common.h)
__device__ __forceinline__ void blend_piece(const EpiArgs& a, size_t rowg, int ch, size_t o, const float4& v, const float4& u, const float4& hh) {
    const float4 c = make_float4(tanh_fast(v.x), tanh_fast(v.y), tanh_fast(v.z), tanh_fast(v.w));
    const float4 hn = make_float4((1.f - u.x) * hh.x + u.x * c.x, (1.f - u.y) * hh.y + u.y * c.y,
                                  (1.f - u.z) * hh.z + u.z * c.z, (1.f - u.w) * hh.w + u.w * c.w);
    nt_store4(reinterpret_cast<float4*>(a.Cand + 4 * o), c);
    nt_store4(reinterpret_cast<float4*>(a.Hnew + 4 * o), hn);
    const int q4 = ch & 3;
    const size_t e = rowg * a.C + (ch >> 2);
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (a.copy[k]) {
            float* dst = a.copy[k] + e * a.copy_ld[k] + a.copy_off[k] + 4 * q4;
            if (((a.copy_ld[k] | a.copy_off[k]) & 3) == 0) {
                *reinterpret_cast<float4*>(dst) = hn;
            } else {
                dst[0] = hn.x; dst[1] = hn.y; dst[2] = hn.z; dst[3] = hn.w;
            }
        }
    if (a.side_src) {                   // complete the consumer's row: its input columns and its zero padding
        float* drow = a.copy[0] + e * a.copy_ld[0];
        if (q4 == 0)
            for (int col = 0; col < a.side_cin; ++col) drow[col] = a.side_src[e * a.side_cin + col];
        if (q4 == 3)
            for (int col = a.side_cin + a.h; col < a.copy_ld[0]; ++col) drow[col] = 0.f;
    }
}

// rowg = b*n_rows + row; ch = float4 index inside the row; acc = sum_j val_j X[col_j] for that piece
template <int MODE>
__device__ __forceinline__ void epilogue(const EpiArgs& a, size_t rowg, int F4, int ch, const float4& acc, float& amax) {     // amax: running max |Y| (SUM forms)
    const size_t o = rowg * F4 + ch;
    if (MODE == EP_PLAIN) {
        float4 r = make_float4(a.alpha * acc.x, a.alpha * acc.y, a.alpha * acc.z, a.alpha * acc.w);
        if (a.beta != 0.f) {
            const float4 y0 = nt_load4(a.Y0 + o);
            r.x = fmaf(a.beta, y0.x, r.x); r.y = fmaf(a.beta, y0.y, r.y);
            r.z = fmaf(a.beta, y0.z, r.z); r.w = fmaf(a.beta, y0.w, r.w);
        }
        nt_store4(a.Y + o, r);
        return;
    }
    if (MODE == EP_SUM || MODE == EP_SUM2) {
        float4 y = make_float4(a.alpha * acc.x, a.alpha * acc.y, a.alpha * acc.z, a.alpha * acc.w);
        const size_t e = rowg * a.C + (ch >> 2);                  // h = 16: piece ch = 4 * category + quarter
        const int q4 = ch & 3;
        for (int i = 0; i < a.n_add; ++i) {
            const float4 t = nt_load4(reinterpret_cast<const float4*>(a.add[i] + e * a.add_ld[i] + a.add_off[i] + 4 * q4));
            const float sc = a.add_scale[i];
            y.x = fmaf(sc, t.x, y.x); y.y = fmaf(sc, t.y, y.y); y.z = fmaf(sc, t.z, y.z); y.w = fmaf(sc, t.w, y.w);
        }
        nt_store4(a.Y + o, y);
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(y.x), fabsf(y.y))), fmaxf(fabsf(y.z), fabsf(y.w)));
        if (a.dYout) {
            const float4 u = nt_load4(reinterpret_cast<const float4*>(a.gU + 4 * o)), c = nt_load4(reinterpret_cast<const float4*>(a.gCand + 4 * o));
            nt_store4(reinterpret_cast<float4*>(a.dYout) + o,
                      make_float4(y.x * u.x * (1.f - c.x * c.x), y.y * u.y * (1.f - c.y * c.y), y.z * u.z * (1.f - c.z * c.z), y.w * u.w * (1.f - c.w * c.w)));
        }
        return;
    }
    // EP_BLEND: rows are (category, h) with h = 16: piece ch = 4 * category + quarter
    const float4 y0 = nt_load4(a.Y0 + o);
    const float4 v = make_float4(acc.x + y0.x, acc.y + y0.y, acc.z + y0.z, acc.w + y0.w);
    const float4 u = nt_load4(reinterpret_cast<const float4*>(a.U + 4 * o)), hh = nt_load4(reinterpret_cast<const float4*>(a.H + 4 * o));
    blend_piece(a, rowg, ch, o, v, u, hh);
}

// ---- CSR: one wave per output row ------------------------------------------------------------------------------
template <int VPT, int MODE>
__global__ __launch_bounds__(SPMM_THREADS) void spmm_wave_row_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ colidx, const float* __restrict__ val,
    int n_rows, int n_cols, const float4* __restrict__ X, int F4, int n_tiles, EpiArgs ep) {
    __shared__ int s_rp[SPMM_ROWS + 1];
    __shared__ int s_col[SPMM_SEG_CAP];
    __shared__ float s_val[SPMM_SEG_CAP];

    const int tile = stc_xcd_tile(blockIdx.x, n_tiles);
    if (tile < 0) return;                       // whole workgroup leaves together
    float wmax = 0.f;                           // SUM forms: largest |Y| this lane produced
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

    for (int r = wave; r < nr; r += SPMM_WAVES) {
        const int js = s_rp[r] - seg0;
        const int je = s_rp[r + 1] - seg0;
        const size_t rowg = (size_t)b * n_rows + row0 + r;
        for (int cb = 0; cb < F4; cb += 64 * VPT) {
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
                        if (MODE == EP_SUM2 && ch < F4) {
                            const float4 t = (ep.X2 + (size_t)b * n_cols * F4 + (size_t)c[u] * F4)[ch];
                            x[u][p].x += t.x; x[u][p].y += t.y; x[u][p].z += t.z; x[u][p].w += t.w;
                        }
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
                    if (ch < F4) {
                        float4 xv = xr[ch];
                        if (MODE == EP_SUM2) {
                            const float4 t = (ep.X2 + (size_t)b * n_cols * F4 + (size_t)c * F4)[ch];
                            xv.x += t.x; xv.y += t.y; xv.z += t.z; xv.w += t.w;
                        }
                        fma4(acc[p], v, xv);
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < VPT; ++p) {
                const int ch = cb + lane + 64 * p;
                if (ch < F4) epilogue<MODE>(ep, rowg, F4, ch, acc[p], wmax);
            }
        }
    }
    if ((MODE == EP_SUM || MODE == EP_SUM2) && ep.amax) publish_amax(ep, wmax);
}

// ---- row-blocked (BCSR 4x1): one wave produces 4 consecutive output rows -----------------------------------------
// Same skeleton, but every fetched neighbour row is accumulated into the 4 rows of the block with its 4
// wave-uniform values: a row shared by several rows of the block is fetched once (18 instead of 36 fetches per 4
// rows of the 8-neighbour grid; the CSR kernel is bound by exactly that L2 -> CU gather traffic).  (An LDS-staged
// tile variant -- distinct rows of 8 output rows copied to LDS per 1 KiB column block -- measured 234 us vs 128 us
// for the CSR kernel: three dependent memory latencies per workgroup, too few bytes in flight; dropped.)
// BLOCKS row blocks per workgroup: 8 (two per wave), 4 (one per wave) or 2 (two waves share a block and split its
// column blocks).  Fewer rows per workgroup keep the set of rows all resident workgroups of an XCD are gathering
// (their "window") small enough for that XCD's 4 MiB L2: at 32 rows per workgroup the window is ~12 MB and
// FETCH_SIZE shows every X row fetched 1.8 times from beyond L2.
constexpr int BR = STC_SPMM_BLOCK_ROWS;
constexpr int BC_CAP = 512;           // block entries staged in LDS per workgroup

// PIPE = 1: the gather is software-pipelined -- batches of PU = 6 neighbour rows (the 8-neighbour grid's interior blocks list
// 18 rows: three full batches, no remainder; graph.py pads every block's list to a multiple of 6 with zero-weight repeats
// of its last column), the NEXT batch's row loads issued before the current batch is accumulated, so that a wave always
// has 6 KiB in flight instead of alternating between "4 loads outstanding" and "none while it multiplies"; the old form
// also walked a remainder (18 = 4 x 4 + 2) one exposed round trip per row.
constexpr int PU = 6;

template <int VPT, int MODE, int BC_BLOCKS, int PIPE = 0, int FULL = 0>      // FULL: F4 is a multiple of the columns the waves of a row block cover
__global__ __launch_bounds__(SPMM_THREADS) void spmm_bcsr_kernel(
    const int* __restrict__ blk_ptr, const int* __restrict__ blk_cols, const float* __restrict__ blk_vals,
    int n_rows, int n_cols, const float4* __restrict__ X, int F4, int n_blocks, int n_tiles, EpiArgs ep) {
    static_assert(BR == 4, "the pipelined gather reads one float4 of values per entry");
    __shared__ int s_bp[BC_BLOCKS + 1];
    __shared__ __attribute__((aligned(16))) int s_col[BC_CAP];
    __shared__ __attribute__((aligned(16))) float s_val[BC_CAP * BR];

    const int tile = stc_xcd_tile(blockIdx.x, n_tiles);
    if (tile < 0) return;
    float wmax = 0.f;                           // SUM forms: largest |Y| this lane produced
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

    constexpr int WPB = BC_BLOCKS >= SPMM_WAVES ? 1 : SPMM_WAVES / BC_BLOCKS;      // waves sharing one row block
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const float4* Xb = X + (size_t)b * n_cols * F4;

    for (int bi = wave / WPB; bi < nb; bi += SPMM_WAVES / WPB) {
        const int js = s_bp[bi] - seg0, je = s_bp[bi + 1] - seg0;
        const int row_base = (blk0 + bi) * BR;
        const int rows_here = min(BR, n_rows - row_base);
        for (int cb = (wave % WPB) * 64 * VPT; cb < F4; cb += WPB * 64 * VPT) {
            float4 acc[BR][VPT];
#pragma unroll
            for (int r = 0; r < BR; ++r)
#pragma unroll
                for (int p = 0; p < VPT; ++p) acc[r][p] = make_float4(0.f, 0.f, 0.f, 0.f);
            // EP_BLEND: the epilogue's own operands (A, U, H pieces of the block's rows) are requested BEFORE the gather so
            // that they arrive under it instead of as a second exposed round trip at the end
            constexpr int PR = MODE == EP_BLEND ? BR : 1;
            float4 pa[PR][VPT], pu[PR][VPT], ph[PR][VPT];
            if (MODE == EP_BLEND) {
#pragma unroll
                for (int r = 0; r < PR; ++r)
#pragma unroll
                    for (int p = 0; p < VPT; ++p) {
                        const int ch = cb + lane + 64 * p;
                        if (r < rows_here && ch < F4) {
                            const size_t o = ((size_t)b * n_rows + row_base + r) * F4 + ch;
                            pa[r][p] = nt_load4(ep.Y0 + o);
                            pu[r][p] = nt_load4(reinterpret_cast<const float4*>(ep.U + 4 * o));
                            ph[r][p] = nt_load4(reinterpret_cast<const float4*>(ep.H + 4 * o));
                        }
                    }
            }

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
            if constexpr (PIPE) {
                // rows of one batch: PU neighbour rows, this lane's VPT pieces of each (second gathered operand summed in).
                // The six column indices come out of LDS in three 8-byte reads issued together (one wait, not six).
                constexpr bool full_cols = FULL != 0;                     // every lane's pieces exist: no per-load column guard
                auto issue = [&](int j0, float4 (&x)[PU][VPT]) {
                    const int2* cp = reinterpret_cast<const int2*>(s_col + j0);
                    const int2 c01 = cp[0], c23 = cp[1], c45 = cp[2];
                    const int cc[PU] = {c01.x, c01.y, c23.x, c23.y, c45.x, c45.y};
#pragma unroll
                    for (int u = 0; u < PU; ++u) {
                        const int c = __builtin_amdgcn_readfirstlane(cc[u]);
                        const float4* xr = Xb + (size_t)c * F4;
#pragma unroll
                        for (int p = 0; p < VPT; ++p) {
                            const int ch = cb + lane + 64 * p;
                            if (full_cols) {
                                x[u][p] = xr[ch];
                                if (MODE == EP_SUM2) {
                                    const float4 t = (ep.X2 + (size_t)b * n_cols * F4 + (size_t)c * F4)[ch];
                                    x[u][p].x += t.x; x[u][p].y += t.y; x[u][p].z += t.z; x[u][p].w += t.w;
                                }
                            } else {
                                x[u][p] = ch < F4 ? xr[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
                                if (MODE == EP_SUM2 && ch < F4) {
                                    const float4 t = (ep.X2 + (size_t)b * n_cols * F4 + (size_t)c * F4)[ch];
                                    x[u][p].x += t.x; x[u][p].y += t.y; x[u][p].z += t.z; x[u][p].w += t.w;
                                }
                            }
                        }
                    }
                };
                auto consume = [&](int j0, const float4 (&x)[PU][VPT]) {
                    const float4* vp = reinterpret_cast<const float4*>(s_val + (size_t)j0 * BR);      // BR = 4 values per entry: one 16-byte read each
                    float4 vv[PU];
#pragma unroll
                    for (int u = 0; u < PU; ++u) vv[u] = vp[u];
#pragma unroll
                    for (int u = 0; u < PU; ++u) {
                        const float v[BR] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
                        for (int r = 0; r < BR; ++r)
#pragma unroll
                            for (int p = 0; p < VPT; ++p) fma4(acc[r][p], v[r], x[u][p]);
                    }
                };
                // (a workgroup whose lists outgrow the staged segment -- not a sparse graph any more -- or whose list starts at an
                // odd entry -- an unpadded plan: the 8-byte LDS reads want an even index -- takes the remainder loop)
                const int jfull = (je <= BC_CAP && (js & 1) == 0) ? js + (je - js) / PU * PU : js;
                if (j < jfull) {
                    float4 xa[PU][VPT], xb[PU][VPT];
                    issue(j, xa);
                    while (true) {
                        if (j + PU < jfull) issue(j + PU, xb);
                        consume(j, xa);
                        j += PU;
                        if (j >= jfull) break;
                        if (j + PU < jfull) issue(j + PU, xa);
                        consume(j, xb);
                        j += PU;
                        if (j >= jfull) break;
                    }
                }
            } else
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
                        if (MODE == EP_SUM2 && ch < F4) {
                            const float4 t = (ep.X2 + (size_t)b * n_cols * F4 + (size_t)c[u] * F4)[ch];
                            x[u][p].x += t.x; x[u][p].y += t.y; x[u][p].z += t.z; x[u][p].w += t.w;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < BR; ++r)
#pragma unroll
                        for (int p = 0; p < VPT; ++p) fma4(acc[r][p], v[u][r], x[u][p]);
            }
            for (; j < je; ++j) {                 // remainder (PIPE: lists that are not a multiple of PU, i.e. unpadded plans)
                int c;
                float v[BR];
                entry(j, c, v);
                const float4* xr = Xb + (size_t)c * F4;
#pragma unroll
                for (int p = 0; p < VPT; ++p) {
                    const int ch = cb + lane + 64 * p;
                    if (ch < F4) {
                        float4 xv = xr[ch];
                        if (MODE == EP_SUM2) {
                            const float4 t = (ep.X2 + (size_t)b * n_cols * F4 + (size_t)c * F4)[ch];
                            xv.x += t.x; xv.y += t.y; xv.z += t.z; xv.w += t.w;
                        }
#pragma unroll
                        for (int r = 0; r < BR; ++r) fma4(acc[r][p], v[r], xv);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < BR; ++r) {
                if (r < rows_here) {
                    const size_t rowg = (size_t)b * n_rows + row_base + r;
#pragma unroll
                    for (int p = 0; p < VPT; ++p) {
                        const int ch = cb + lane + 64 * p;
                        if (ch < F4) {
                            if (MODE == EP_BLEND) {
                                const int rp = MODE == EP_BLEND ? r : 0;
                                const float4 y = make_float4(acc[r][p].x + pa[rp][p].x, acc[r][p].y + pa[rp][p].y,
                                                             acc[r][p].z + pa[rp][p].z, acc[r][p].w + pa[rp][p].w);
                                blend_piece(ep, rowg, ch, rowg * F4 + ch, y, pu[rp][p], ph[rp][p]);
                            } else {
                                epilogue<MODE>(ep, rowg, F4, ch, acc[r][p], wmax);
                            }
                        }
                    }
                }
            }
        }
    }
    if ((MODE == EP_SUM || MODE == EP_SUM2) && ep.amax) publish_amax(ep, wmax);
}

// ---- row-blocked, NARROW rows (F4 = F/4 in {1, 2, 4, 8, 16} float4 per row: the layer-0 input plane, C = 32 floats = 128 bytes) --------
// A 64-lane wave would leave all but F4 lanes idle on such rows, so it is split into G = 64 / F4 lane groups and each group
// produces its OWN block of 4 rows: 4 G rows per wave.  Block lists differ per group, so column / values are per-lane loads
// (identical within a group: one L1 transaction each) and the loop runs to the longest list with the finished groups masked.
// Same arithmetic order per row as spmm_bcsr_kernel (block list order, fmaf chain), plain epilogue Y = alpha S.X + beta Y0.
template <int F4>
__global__ __launch_bounds__(SPMM_THREADS) void spmm_bcsr_narrow_kernel(
    const int* __restrict__ blk_ptr, const int* __restrict__ blk_cols, const float4* __restrict__ blk_vals,
    int n_rows, int n_cols, const float4* __restrict__ X, const float4* __restrict__ Y0, float4* __restrict__ Y,
    int n_blocks, float alpha, float beta) {
    constexpr int G = 64 / F4;                       // row blocks per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / F4, ch = lane % F4;
    const int blk = (blockIdx.x * SPMM_WAVES + wave) * G + grp;
    const int b = blockIdx.y;
    if (blk >= n_blocks) return;                     // (no barriers in this kernel)
    const int js = blk_ptr[blk], je = blk_ptr[blk + 1];
    const float4* Xb = X + (size_t)b * n_cols * F4 + ch;
    float4 acc[BR];
#pragma unroll
    for (int r = 0; r < BR; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    int j = js;
    for (; j + 3 <= je; j += 3) {                    // three neighbour rows in flight per lane
        int c[3]; float4 v[3], x[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) { c[u] = blk_cols[j + u]; v[u] = blk_vals[j + u]; }
#pragma unroll
        for (int u = 0; u < 3; ++u) x[u] = Xb[(size_t)c[u] * F4];
#pragma unroll
        for (int u = 0; u < 3; ++u) { fma4(acc[0], v[u].x, x[u]); fma4(acc[1], v[u].y, x[u]); fma4(acc[2], v[u].z, x[u]); fma4(acc[3], v[u].w, x[u]); }
    }
    for (; j < je; ++j) {
        const int c = blk_cols[j];
        const float4 v = blk_vals[j], x = Xb[(size_t)c * F4];
        fma4(acc[0], v.x, x); fma4(acc[1], v.y, x); fma4(acc[2], v.z, x); fma4(acc[3], v.w, x);
    }
#pragma unroll
    for (int r = 0; r < BR; ++r) {
        const int row = blk * BR + r;
        if (row < n_rows) {
            const size_t o = ((size_t)b * n_rows + row) * F4 + ch;
            float4 y = make_float4(alpha * acc[r].x, alpha * acc[r].y, alpha * acc[r].z, alpha * acc[r].w);
            if (beta != 0.f) {
                const float4 y0 = Y0[o];
                y.x = fmaf(beta, y0.x, y.x); y.y = fmaf(beta, y0.y, y.y); y.z = fmaf(beta, y0.z, y.z); y.w = fmaf(beta, y0.w, y.w);
            }
            Y[o] = y;
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

struct GraphArgs {      // either form of the same matrix; BCSR is used when blk_ptr is given
    const int32_t *rowptr, *colidx; const float* val;
    const int32_t *blk_ptr, *blk_cols; const float* blk_vals;
};

template <int MODE>
int launch_vector(const char* who, const GraphArgs& g, int n_rows, int n_cols, const float* X, int batch, int F,
                  const EpiArgs& ep, hipStream_t s) {
    const int F4 = F / 4;
    const float4* X4 = reinterpret_cast<const float4*>(X);
    if (g.blk_ptr) {
        const int n_blocks = (n_rows + BR - 1) / BR;
        constexpr int blocks = STC_BCSR_DEFAULT_BLOCKS;      // row blocks per workgroup (2: two waves share a block; 4 and 8 measured slower)
        const int n_tiles = (n_blocks + blocks - 1) / blocks;
        const int per = (n_tiles + stc::kNumXcd - 1) / stc::kNumXcd;
        const dim3 grid(per * stc::kNumXcd, batch), block(SPMM_THREADS);
        // The pipelined gather is used where it measured faster on MI355X (profiles/r02/c_kbench_spmm.txt): the plain product
        // Y = alpha S.X with full column blocks (F=1024, B=1: 99 -> 86 us; F=512, B=5: 218 -> 210 us).  With a Y0 operand or an
        // epilogue (sum / blend forms) the extra live registers cost what the pipelining gains (equal or slower): old loop.
        // (measured again for the state-gradient sum with ONE gathered operand, which is what the cell graph launches now: 27.1 vs 27.3 ms)
        const bool pipe = MODE == EP_PLAIN && (ep.Y0 == nullptr || ep.beta == 0.f);
#define STC_BCSR_GO(VPT_, BLK_) do { if (pipe && BLK_ == 2 && F4 % (128 * VPT_) == 0) hipLaunchKernelGGL((spmm_bcsr_kernel<VPT_, MODE, 2, (MODE == EP_PLAIN), (MODE == EP_PLAIN)>), grid, block, 0, s, g.blk_ptr, g.blk_cols, g.blk_vals, \
                                                   n_rows, n_cols, X4, F4, n_blocks, n_tiles, ep); \
                                     else hipLaunchKernelGGL((spmm_bcsr_kernel<VPT_, MODE, BLK_, 0, 0>), grid, block, 0, s, g.blk_ptr, g.blk_cols, g.blk_vals, \
                                                   n_rows, n_cols, X4, F4, n_blocks, n_tiles, ep); } while (0)
        if (blocks == 2) {            // two waves per block: each covers every other column block of 64*VPT float4
            if (F4 <= 128) STC_BCSR_GO(1, 2); else if (F4 <= 256) STC_BCSR_GO(2, 2); else STC_BCSR_GO(4, 2);
        } else if (blocks == 4) {
            if (F4 <= 64) STC_BCSR_GO(1, 4); else if (F4 <= 128) STC_BCSR_GO(2, 4); else STC_BCSR_GO(4, 4);
        } else {
            if (F4 <= 64) STC_BCSR_GO(1, 8); else if (F4 <= 128) STC_BCSR_GO(2, 8); else STC_BCSR_GO(4, 8);
        }
#undef STC_BCSR_GO
    } else {
        const int n_tiles = (n_rows + SPMM_ROWS - 1) / SPMM_ROWS;
        const int per = (n_tiles + stc::kNumXcd - 1) / stc::kNumXcd;
        const dim3 grid(per * stc::kNumXcd, batch), block(SPMM_THREADS);
        if (F4 <= 64)
            hipLaunchKernelGGL((spmm_wave_row_kernel<1, MODE>), grid, block, 0, s, g.rowptr, g.colidx, g.val, n_rows, n_cols, X4, F4, n_tiles, ep);
        else if (F4 <= 128)
            hipLaunchKernelGGL((spmm_wave_row_kernel<2, MODE>), grid, block, 0, s, g.rowptr, g.colidx, g.val, n_rows, n_cols, X4, F4, n_tiles, ep);
        else
            hipLaunchKernelGGL((spmm_wave_row_kernel<4, MODE>), grid, block, 0, s, g.rowptr, g.colidx, g.val, n_rows, n_cols, X4, F4, n_tiles, ep);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return stc::hip_status(e, who);
    return STC_OK;
}

int check_fused(const char* who, const GraphArgs& g, int n_rows, int n_cols, const float* X, const float* Y0,
                int batch, int C, int cin, int h, int pad) {
    STC_REQUIRE(n_rows >= 0 && n_cols >= 0 && batch >= 0 && C >= 1 && cin >= 0 && h >= 1 && pad >= 0, STC_EINVAL, "%s: bad sizes", who);
    STC_REQUIRE((cin + h + pad) % 4 == 0, STC_EINVAL, "%s: row width cin+h+pad = %d must be a multiple of 4", who, cin + h + pad);
    STC_REQUIRE((long long)C * (cin + h + pad) >= 64, STC_ELIMIT, "%s: node row of %d floats is too narrow for the vector kernels", who, C * (cin + h + pad));
    STC_REQUIRE(batch <= 65535, STC_ELIMIT, "%s: batch %d > 65535 (grid.y)", who, batch);
    if (n_rows == 0 || batch == 0) return STC_OK;
    // the column / value arrays may be null for a graph without edges (pointer arrays all zero: they are never read)
    STC_REQUIRE(g.blk_ptr || g.rowptr, STC_EINVAL, "%s: neither graph form given", who);
    STC_REQUIRE(X && Y0 && n_cols > 0, STC_EINVAL, "%s: null X / Y0", who);
    STC_REQUIRE(stc::aligned16(X) && stc::aligned16(Y0), STC_EALIGN, "%s: X / Y0 must be 16-byte aligned", who);
    return STC_OK;
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
        EpiArgs ep{};
        ep.Y0 = reinterpret_cast<const float4*>(Y0);
        ep.Y = reinterpret_cast<float4*>(Y);
        ep.alpha = alpha;
        ep.beta = beta;
        const GraphArgs g{rowptr, colidx, val, nullptr, nullptr, nullptr};
        return launch_vector<EP_PLAIN>("stc_csr_spmm_f32 launch", g, n_rows, n_cols, X, batch, F, ep, s);
    }
    const int lanes = next_pow2(F) < SPMM_THREADS ? next_pow2(F) : SPMM_THREADS;
    const int rows_per_block = SPMM_THREADS / lanes;
    dim3 grid((n_rows + rows_per_block - 1) / rows_per_block, batch), block(SPMM_THREADS);
    hipLaunchKernelGGL(spmm_generic_kernel, grid, block, 0, s, rowptr, colidx, val, n_rows, n_cols, X, Y0, Y, F, alpha, beta, lanes);
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
    STC_REQUIRE(F % 4 == 0 && F >= 4, STC_EINVAL, "stc_bcsr_spmm_f32: F=%d must be a positive multiple of 4", F);
    STC_REQUIRE(stc::aligned16(X) && stc::aligned16(Y) && (!Y0 || stc::aligned16(Y0)), STC_EALIGN,
                "stc_bcsr_spmm_f32: X / Y / Y0 must be 16-byte aligned");
    STC_REQUIRE(batch <= 65535, STC_ELIMIT, "stc_bcsr_spmm_f32: batch %d > 65535 (grid.y)", batch);
    const int F4 = F / 4;
    if (F4 <= 16 && (F4 & (F4 - 1)) == 0) {            // narrow rows (<= 256 bytes): several row blocks per wave
        const int n_blocks = (n_rows + BR - 1) / BR;
        hipStream_t s = static_cast<hipStream_t>(stream);
#define STC_NARROW_GO(F4_) hipLaunchKernelGGL((spmm_bcsr_narrow_kernel<F4_>), dim3((n_blocks + SPMM_WAVES * (64 / F4_) - 1) / (SPMM_WAVES * (64 / F4_)), batch), \
                                              dim3(SPMM_THREADS), 0, s, blk_ptr, blk_cols, reinterpret_cast<const float4*>(blk_vals), n_rows, n_cols, \
                                              reinterpret_cast<const float4*>(X), reinterpret_cast<const float4*>(Y0), reinterpret_cast<float4*>(Y), n_blocks, alpha, beta)
        switch (F4) { case 1: STC_NARROW_GO(1); break; case 2: STC_NARROW_GO(2); break; case 4: STC_NARROW_GO(4); break;
                      case 8: STC_NARROW_GO(8); break; default: STC_NARROW_GO(16); break; }
#undef STC_NARROW_GO
        STC_LAUNCH_CHECK("stc_bcsr_spmm_f32 (narrow rows) launch");
        return STC_OK;
    }
    EpiArgs ep{};
    ep.Y0 = reinterpret_cast<const float4*>(Y0);
    ep.Y = reinterpret_cast<float4*>(Y);
    ep.alpha = alpha;
    ep.beta = beta;
    const GraphArgs g{nullptr, nullptr, nullptr, blk_ptr, blk_cols, blk_vals};
    return launch_vector<EP_PLAIN>("stc_bcsr_spmm_f32 launch", g, n_rows, n_cols, X, batch, F, ep, static_cast<hipStream_t>(stream));
}

extern "C" int stc_spmm_blend_fwd_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                                      const int32_t* blk_ptr, const int32_t* blk_cols, const float* blk_vals,
                                      int32_t n_rows, int32_t n_cols, const float* Bm, const float* A,
                                      const float* U, const float* H, float* Cand, float* Hnew,
                                      float* copy0, int32_t copy0_ld, int32_t copy0_off, const float* side_src, int32_t side_cin,
                                      float* copy1, int32_t copy1_ld, int32_t copy1_off,
                                      int32_t batch, int32_t C, int32_t h, void* stream) {
    const GraphArgs g{rowptr, colidx, val, blk_ptr, blk_cols, blk_vals};
    STC_REQUIRE(h == 16, STC_EUNSUPPORTED, "stc_spmm_blend_fwd_f32: hidden width %d (the blend epilogue is built for 16)", h);
    if (int rc = check_fused("stc_spmm_blend_fwd_f32", g, n_rows, n_cols, Bm, A, batch, C, 0, h, 0)) return rc;
    if (n_rows == 0 || batch == 0) return STC_OK;
    STC_REQUIRE(U && H && Cand && Hnew, STC_EINVAL, "stc_spmm_blend_fwd_f32: null pointer");
    STC_REQUIRE(stc::aligned16(U) && stc::aligned16(H) && stc::aligned16(Cand) && stc::aligned16(Hnew), STC_EALIGN,
                "stc_spmm_blend_fwd_f32: operands must be 16-byte aligned");
    STC_REQUIRE(Bm != Cand && Bm != Hnew, STC_EINVAL, "stc_spmm_blend_fwd_f32: outputs must not alias Bm (its rows are gathered by other rows)");
    STC_REQUIRE((!copy0 || (copy0_off >= 0 && copy0_off + h <= copy0_ld)) && (!copy1 || (copy1_off >= 0 && copy1_off + h <= copy1_ld)),
                STC_EINVAL, "stc_spmm_blend_fwd_f32: state copy columns [off, off+%d) do not fit the row width", h);
    STC_REQUIRE(!side_src || (copy0 && side_cin >= 0 && side_cin == copy0_off), STC_EINVAL,
                "stc_spmm_blend_fwd_f32: side_src needs copy0 with copy0_off == side_cin");
    STC_REQUIRE((!copy0 || (reinterpret_cast<uintptr_t>(copy0) & 3u) == 0) && (!copy1 || (reinterpret_cast<uintptr_t>(copy1) & 3u) == 0),
                STC_EALIGN, "stc_spmm_blend_fwd_f32: misaligned copy destination");
    if (copy0 && ((copy0_ld | copy0_off) & 3) == 0) STC_REQUIRE(stc::aligned16(copy0), STC_EALIGN, "stc_spmm_blend_fwd_f32: copy0 not 16-byte aligned");
    if (copy1 && ((copy1_ld | copy1_off) & 3) == 0) STC_REQUIRE(stc::aligned16(copy1), STC_EALIGN, "stc_spmm_blend_fwd_f32: copy1 not 16-byte aligned");
    EpiArgs ep{};
    ep.Y0 = reinterpret_cast<const float4*>(A);
    ep.C = C; ep.L = h; ep.cin = 0; ep.h = h;
    ep.U = U; ep.H = H; ep.Cand = Cand; ep.Hnew = Hnew;
    ep.copy[0] = copy0; ep.copy_ld[0] = copy0_ld; ep.copy_off[0] = copy0_off;
    ep.copy[1] = copy1; ep.copy_ld[1] = copy1_ld; ep.copy_off[1] = copy1_off;
    ep.side_src = copy0 ? side_src : nullptr; ep.side_cin = side_cin;
    return launch_vector<EP_BLEND>("stc_spmm_blend_fwd_f32 launch", g, n_rows, n_cols, Bm, batch, C * h, ep, static_cast<hipStream_t>(stream));
}

extern "C" int stc_spmm_sum_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                                const int32_t* blk_ptr, const int32_t* blk_cols, const float* blk_vals,
                                int32_t n_rows, int32_t n_cols, const float* X, const float* X2, float alpha,
                                int32_t n_add, const float* const* add, const int32_t* add_ld, const int32_t* add_off, const float* add_scale,
                                float* Y, const float* U, const float* Cand, float* dY,
                                float* amax, int32_t n_amax,
                                int32_t batch, int32_t C, int32_t h, void* stream) {
    const GraphArgs g{rowptr, colidx, val, blk_ptr, blk_cols, blk_vals};
    STC_REQUIRE(h == 16, STC_EUNSUPPORTED, "stc_spmm_sum_f32: hidden width %d (built for 16)", h);
    STC_REQUIRE(!amax || n_amax >= 1, STC_EINVAL, "stc_spmm_sum_f32: amax with %d slots", n_amax);
    STC_REQUIRE(n_add >= 0 && n_add <= STC_SPMM_SUM_MAX_ADD && (n_add == 0 || (add && add_ld && add_off)), STC_EINVAL,
                "stc_spmm_sum_f32: 0..%d addends, got %d", STC_SPMM_SUM_MAX_ADD, n_add);
    if (int rc = check_fused("stc_spmm_sum_f32", g, n_rows, n_cols, X, Y, batch, C, 0, h, 0)) return rc;      // (Y checked as the aligned "Y0" operand)
    if (n_rows == 0 || batch == 0) return STC_OK;
    STC_REQUIRE(X != Y && X2 != Y, STC_EINVAL, "stc_spmm_sum_f32: Y must not alias a gathered operand");
    STC_REQUIRE(!X2 || stc::aligned16(X2), STC_EALIGN, "stc_spmm_sum_f32: X2 not 16-byte aligned");
    EpiArgs ep{};
    ep.Y = reinterpret_cast<float4*>(Y);
    ep.C = C; ep.L = h; ep.cin = 0; ep.h = h;
    ep.X2 = reinterpret_cast<const float4*>(X2);
    STC_REQUIRE(!dY || (U && Cand && stc::aligned16(U) && stc::aligned16(Cand) && stc::aligned16(dY) && dY != Y), STC_EINVAL,
                "stc_spmm_sum_f32: dY needs U and Cand (16-byte aligned, not aliasing Y)");
    ep.gU = U; ep.gCand = Cand; ep.dYout = dY;
    ep.amax = reinterpret_cast<unsigned*>(amax); ep.n_amax = n_amax;
    ep.alpha = alpha;
    ep.n_add = n_add;
    for (int i = 0; i < n_add; ++i) {
        STC_REQUIRE(add[i] && add_off[i] >= 0 && add_off[i] + h <= add_ld[i] && ((add_ld[i] | add_off[i]) & 3) == 0 && stc::aligned16(add[i]), STC_EINVAL,
                    "stc_spmm_sum_f32: addend %d (ld %d, off %d) must be non-null, 16-byte aligned, with ld and off multiples of 4", i, add_ld[i], add_off[i]);
        STC_REQUIRE(add[i] != Y, STC_EINVAL, "stc_spmm_sum_f32: Y must not alias an addend");
        ep.add[i] = add[i]; ep.add_ld[i] = add_ld[i]; ep.add_off[i] = add_off[i]; ep.add_scale[i] = add_scale ? add_scale[i] : 1.f;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    return X2 ? launch_vector<EP_SUM2>("stc_spmm_sum_f32 launch", g, n_rows, n_cols, X, batch, C * h, ep, s)
              : launch_vector<EP_SUM>("stc_spmm_sum_f32 launch", g, n_rows, n_cols, X, batch, C * h, ep, s);
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
