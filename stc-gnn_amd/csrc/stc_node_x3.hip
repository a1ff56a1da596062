// Split-operand matrix-core path of the BDG_Dif node kernel (reference STC_GNN.py:38-45 and its autograd) for gfx950.
//
// The fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at the fp32 vector rate, 1/16 of the bf16 MFMA, and the node kernels
// built on it (stc_node_mfma.hip) are bound by it, not by HBM.  Here every fp32 operand a is split EXACTLY into three
// bf16 pieces, a = a_h + a_m + a_l (8 + 8 + 8 significant bits; each remainder is exact in fp32), and a product
// sum is accumulated in fp32 from the six piece products of weight >= 2^-16,
//     A.B ~= Al.Bh + Ah.Bl + Am.Bm + Am.Bh + Ah.Bm + Ah.Bh      (dropped: 2^-24 |a||b| and below, the fp32 rounding level)
// on v_mfma_f32_16x16x32_bf16 (16 cycles for 16x16x32): 6/16 of the matrix time of the fp32 instruction.  bf16 x bf16
// products are exact in the fp32 accumulator, so the result differs from an fmaf chain only by summation order and
// the dropped 2^-24 terms (tools/probes/mfma_bf16_layout.hip: 8.5e-8 vs 1.4e-7 for the fmaf chain, against fp64).
//
// Register layout of v_mfma_f32_16x16x32_bf16, lane (x = lane & 15, g = lane >> 4):
//     A operand: A[x][8g + e], e = 0..7      B operand: B[8g + e][x]      D: D[4g + r][x], r = 0..3
// The contraction slot 8g + e may stand for any contraction index as long as both operands agree.  Two maps are used:
//     rows of a feature slab:   slot (g, e) = column 8g + e                 (8 contiguous floats of the lane's row)
//     accumulator fed operands: slot (g, e) = row 16 (e >> 2) + 4g + (e & 3) of a PAIR of 16-row tiles, which is exactly
//                               what lane (x, g) holds of those two tiles -- accumulators feed the next product from
//                               registers, no LDS trip (as in the fp32 kernel).
// Everything that is the same for every node (W, T_c) is split once per workgroup into LDS in fragment order (three
// 16-byte pieces per lane, conflict-free ds_read_b128).  One wave owns one node at a time; no barrier in the node loop.
//
// Shapes: C = 32 * NB2 categories, Ho = 16 * HB outputs, L <= 32 features per slab (one 32-wide step, zero padded),
// Ks = Kc = K.  Other shapes stay on the fp32 MFMA / VALU kernels.
#include "stc_x3_frag.h"
// Cache policy of this file's streams (stc_common.h: stc_ld_once / stc_st_once), measured: whole-line accesses carry the non-temporal
// policy (Row8::load_pieces, put_plane, the backward kernels' row-layout planes); where a lane's two 16-byte pieces of a row come from two
// load instructions that share every 128-byte line (Row8::load, load_planes) or results leave as 4-byte accumulator-layout stores (half a
// line per instruction), the default policy stays -- with the non-temporal one the gates forward ran 5 % (loads) and 17 % (stores) slower.

namespace {

// columns 8g .. 8g+7 of a feature row of L floats (L a multiple of 4, <= 32); columns >= L read as zero
template <int L>
struct Row8 {
    f32x4 a, b;
    __device__ __forceinline__ void load(const float* __restrict__ row, int g) {
        a = kZero4; b = kZero4;
        if (L == 32 || 8 * g < L) a = *reinterpret_cast<const f32x4*>(row + 8 * g);
        if (L == 32 || 8 * g + 4 < L) b = *reinterpret_cast<const f32x4*>(row + 8 * g + 4);
    }
    // planar layout: columns 0..15 of the row live in one (rows, 16) plane, 16..31 in another
    __device__ __forceinline__ void load_planes(const float* __restrict__ xrow, const float* __restrict__ hrow, int g) {
        static_assert(L == 32, "planar rows are 16 + 16 columns");
        const float* src = (g < 2 ? xrow : hrow) + 8 * (g & 1);
        a = *reinterpret_cast<const f32x4*>(src);
        b = *reinterpret_cast<const f32x4*>(src + 4);
    }
    // planar layout, PIECE order: this lane takes the 16-byte piece g of the row from EACH plane (slots 0..3 = X columns 4g.., slots 4..7 =
    // H columns 4g..), so that one load instruction covers 16 whole rows (1 KiB, whole 128-byte lines) and can carry the non-temporal
    // policy; the weight tables are filled in the same slot order (piece_slot_row).
    __device__ __forceinline__ void load_pieces(const float* __restrict__ xrow, const float* __restrict__ hrow, int g) {
        static_assert(L == 32, "planar rows are 16 + 16 columns");
        a = stc_ld_once(reinterpret_cast<const f32x4*>(xrow + 4 * g));
        b = stc_ld_once(reinterpret_cast<const f32x4*>(hrow + 4 * g));
    }
    // planar layout with a narrow input plane: columns 0..15 from the (rows, 16) state plane, 16..16+cin-1 from the
    // (rows, cin) input plane, cin <= 4; everything else zero
    __device__ __forceinline__ void load_planes_narrow(const float* __restrict__ prow, const float* __restrict__ xrow, int cin, int g) {
        a = kZero4; b = kZero4;
        if (g < 2) {
            a = *reinterpret_cast<const f32x4*>(prow + 8 * g);
            b = *reinterpret_cast<const f32x4*>(prow + 8 * g + 4);
        } else if (g == 2) {
            const float x0 = xrow[0], x1 = cin > 1 ? xrow[1] : 0.f, x2 = cin > 2 ? xrow[2] : 0.f, x3 = cin > 3 ? xrow[3] : 0.f;
            a = f32x4{x0, x1, x2, x3};
        }
    }
    __device__ __forceinline__ float at(int e) const { return e < 4 ? a[e & 3] : b[e & 3]; }
    __device__ __forceinline__ void fma(float v, const Row8& o) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = fmaf(v, o.a[i], a[i]); b[i] = fmaf(v, o.b[i], b[i]); }
    }
    __device__ __forceinline__ void store(float* __restrict__ row, int g) const {
        if (L == 32 || 8 * g < L) *reinterpret_cast<f32x4*>(row + 8 * g) = a;
        if (L == 32 || 8 * g + 4 < L) *reinterpret_cast<f32x4*>(row + 8 * g + 4) = b;
    }
};

// --------------------------------------------------------------------------------------- forward
// PL = 1: planar inputs (Z.p[n] = columns 0..15, Z.q[n] = columns 16..31 of slab n, each (nodes, C, 16)); with EPI_GATES the
// candidate's input then is planar too: its X plane is Xt itself and only the R*H plane (epi.CandIn, 16 wide) is written.
// PL = 2 (L = 20): planar with a narrow input plane -- Z.p[n] = the STATE plane (nodes, C, 16), Z.q[n] = the input plane
// (nodes, C, cin), cin = Lw - 16 <= 4; slab columns are [state | input | pad] and W's rows are permuted to match
// POST = 1 (planar gates kernel only): the candidate convolution's projection runs as a second stage of the same launch.
// Its input [Xt | R*H] is already with the wave -- Xt in the row fragments it loaded, R*H in its accumulator-layout epilogue
// registers, brought to row layout through a per-wave LDS tile -- so the post-aggregation pair A, Bm (stc_bdg_node_post_fwd_f32)
// is written without re-reading Xt and R*H from HBM.
struct PostArgs { const float* Wc; const float* bc; float* A; float* Bm; };

// Row of the [X | H] slab that contraction slot e of lane group gg carries: natural order 8 gg + e, or Row8::load_pieces' order.
__host__ __device__ constexpr int piece_slot_row(bool pieces, int gg, int e) { return !pieces ? 8 * gg + e : (e < 4 ? 4 * gg + e : 16 + 4 * gg + (e - 4)); }

// F: operand format (stc_x3_frag.h).  FmtH2 (two fp16 pieces, three products): W and T_c are normalised per workgroup at table-fill time
// (W's blocks c = 0 carry sW sT, blocks c >= 1 carry sW, T_c carries sT: projection and category mix then meet in one accumulator with the
// common factor sW sT, taken out in the epilogue); activations carry one power of two per NODE, from the node's own maximum (the
// reference's einsum is scale-free, two fp16 pieces are not: stc_x3_frag.h), taken out in the same epilogue.
template <int NB2, int HB, int K, int L, int EPI, int PL = 0, int POST = 0, class F = FmtB3>
__global__ __launch_bounds__(MF_THREADS, (NB2 == 1 ? 2 : 1)) void node_fwd_x3_kernel(
    ZPtrs Z, const float* __restrict__ Tc, const float* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ Y, int nodes, int Lw, FwdEpi epi, PostArgs post) {
    using Op = typename F::Op;
    constexpr int NP = F::NP;
    static_assert(!POST || (PL != 0 && EPI == EPI_GATES && K == 2), "the fused candidate projection belongs to the planar gates kernel");
    constexpr int KL = K;                               // slabs read from HBM
    constexpr int NRB = 2 * NB2, C = 32 * NB2, Ho = 16 * HB, NCB = K * HB;
    constexpr int HID = 16;
    constexpr int nWx = K * NCB, nTx = (K - 1) * NRB * NB2;
    static_assert(EPI == EPI_NONE || (EPI == EPI_GATES && HB == 2) || (EPI == EPI_BLEND && HB == 1), "epilogue needs hidden = 16");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* Wx = reinterpret_cast<u32x4*>(smem_raw);        // [K n][NCB]          B: W[(n, c, l = slot)][o = 16 hb + x]
    u32x4* Tx = Wx + nWx * NP * 64;                         // [K-1][NRB rb][NB2]  A: T_c[c' = 32 p + pair_row][d = 16 rb + x]
    constexpr int nWp = POST ? K * K : 0;                   // POST: [K n][K c]    B: Wc[(n, c, l = slot)][o = x]   (Ho = 16)
    constexpr int TRS = 20;                                 // row stride of the transpose tiles (16 + 4: conflict-free row reads)
    u32x4* Wp = Tx + nTx * NP * 64;
    float* tr = reinterpret_cast<float*>(Wp + nWp * NP * 64);  // POST: [wave][NRB][16 rows][TRS]  R*H, accumulator -> row layout
    // Planar gates kernel: the epilogue wants the previous state H in accumulator layout, and the H plane is already with the wave in row
    // layout (it is one of the planes of slab 0).  Loading it a second time from global memory re-fetched 0.74 of a plane per launch from
    // beyond L2 (rocprofv3 FETCH_SIZE: 2 435 MB against the 2 055 MB of the four input planes -- the rows a wave loaded one node ago have
    // left the XCD's 4 MiB L2 by the time it asks again): the rows go through a per-wave LDS tile instead, row layout -> accumulator layout.
    constexpr bool HT = EPI == EPI_GATES && PL != 0;
    constexpr bool PIECES = PL == 1;                        // two full planes per slab: piece-order loads (Row8::load_pieces)
    float* th = tr + (POST ? MF_WAVES * NRB * 16 * TRS : 0);    // HT: [wave][NRB][16 rows][TRS]
    const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, g = lane >> 4;

    float sT = 1.f, sW = 1.f, sWp = 1.f;                    // FmtH2: powers of two from the tables' own maxima (same in every workgroup)
    if constexpr (F::SCALED) {
        float* scratch = reinterpret_cast<float*>(smem_raw);
        if (K > 1) {
            sT = clamp_mix_scale(pow2_scale(block_absmax(Tc + (size_t)C * C, (K - 1) * C * C, scratch, MF_THREADS), STC_T_TARGET_FWD));      // (W's block 0 carries sT as well)
        }
        sW = pow2_scale(block_absmax(W, K * K * Lw * Ho, scratch, MF_THREADS), STC_W_TARGET);
        if (POST) sWp = pow2_scale(block_absmax(post.Wc, K * K * Lw * 16, scratch, MF_THREADS), STC_W_TARGET);
    }

    for (int idx = tid; idx < nWx * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, cb = f % NCB, n = f / NCB;
        const int c = cb / HB, o = (cb % HB) * 16 + (ll & 15), gg = ll >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int l = piece_slot_row(PIECES, gg, e);
            const int wl = PL == 2 ? stc_wrow_swapped(l, Lw - 16) : l;                // PL = 2: slab columns are [state | input | pad]
            v[e] = (wl >= 0 && wl < Lw) ? W[((size_t)(n * K + c) * Lw + wl) * Ho + o] : 0.f;       // pad columns contribute nothing
        }
        F::put(Wx, f, ll, v, c == 0 ? sW * sT : sW);
    }
    for (int idx = tid; idx < nWp * 64; idx += MF_THREADS) {      // the candidate's weights, same slab-column order (Ho = 16)
        const int ll = idx & 63, f = idx >> 6, c = f % K, n = f / K, gg = ll >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int l = piece_slot_row(PIECES, gg, e);
            const int wl = PL == 2 ? stc_wrow_swapped(l, Lw - 16) : l;
            v[e] = (wl >= 0 && wl < Lw) ? post.Wc[((size_t)(n * K + c) * Lw + wl) * 16 + (ll & 15)] : 0.f;
        }
        F::put(Wp, f, ll, v, c == 0 ? sWp * sT : sWp);
    }
    for (int idx = tid; idx < nTx * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, p = f % NB2, rb = (f / NB2) % NRB, c1 = f / (NB2 * NRB), gg = ll >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = Tc[(size_t)(c1 + 1) * C * C + (32 * p + pair_row(gg, e)) * C + 16 * rb + (ll & 15)];
        F::put(Tx, f, ll, v, sT);
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = gridDim.x * MF_WAVES;
    const float inv = inv_pow2(uniform_bits(sW * sT)), invp = inv_pow2(uniform_bits(sWp * sT));       // (1 for FmtB3; scalar registers)
    float bv[HB];
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) bv[hb] = bias ? bias[16 * hb + x] : 0.f;

    int node = blockIdx.x * MF_WAVES + wave;
    constexpr bool ASC = F::SCALED && PL != 0;              // activation scales (planar forms: the only ones that run the fp16 x 2 format)
    static_assert(!F::SCALED || PL != 0, "the fp16 x 2 format is instantiated for the planar forms only");
    float pmax[KL], qmax[KL];
#pragma unroll
    for (int n = 0; n < KL; ++n) { pmax[n] = 0.f; qmax[n] = 0.f; }
    Row8<L> cur[KL][NRB], nxt[KL][NRB];
    auto load_rows = [&](Row8<L> (&z)[KL][NRB], int nd) {
#pragma unroll
        for (int n = 0; n < KL; ++n)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                if constexpr (PL == 1) z[n][rb].load_pieces(Z.p[n] + ((size_t)nd * C + 16 * rb + x) * 16, Z.q[n] + ((size_t)nd * C + 16 * rb + x) * 16, g);
                else if constexpr (PL == 2) z[n][rb].load_planes_narrow(Z.p[n] + ((size_t)nd * C + 16 * rb + x) * 16,
                                                                        Z.q[n] + ((size_t)nd * C + 16 * rb + x) * (Lw - 16), Lw - 16, g);
                else z[n][rb].load(Z.p[n] + ((size_t)nd * C + 16 * rb + x) * L, g);
            }
    };
    if (node < nodes) load_rows(cur, node);
    // The first node's rows are waited for HERE: left pending into the loop, the compiler's wait-count pass merges them into the loop's state
    // and makes every iteration wait for the rows it has just requested for the NEXT node right away (vmcnt counts in order).
    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0); expcnt / lgkmcnt untouched
    while (node < nodes) {
        const int next_node = node + nw;
        // epilogue operands in accumulator layout (row 16rb + 4g + r, column x): needed only after the MFMAs.  Requested BEFORE the next
        // node's rows: vmcnt counts in order, so the other way round the epilogue would wait for the whole prefetch
        float hv[NRB][4], uv[NRB][4], side[NRB][4];
        // EPI_GATES: lane x < L - 16 also writes one column of CandIn outside the R*H block, in the same row layout:
        // column x of Xt (re-read from slab 0, an L2 hit) while x < cin, else the zero of pad column x + 16
        const bool has_side = EPI == EPI_GATES && !PL && x < L - HID;
        if constexpr (HT) {
            if constexpr (PIECES) {                                // every lane holds the piece g of its H row
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) *reinterpret_cast<f32x4*>(th + ((wave * NRB + rb) * 16 + x) * TRS + 4 * g) = cur[0][rb].b;
            } else if (g < 2) {                                    // PL = 2: [H | x | pad]
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    float* dst = th + ((wave * NRB + rb) * 16 + x) * TRS + 8 * (g & 1);
                    *reinterpret_cast<f32x4*>(dst) = cur[0][rb].a;
                    *reinterpret_cast<f32x4*>(dst + 4) = cur[0][rb].b;
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[rb][r] = th[((wave * NRB + rb) * 16 + 4 * g + r) * TRS + x];
        } else if (EPI != EPI_NONE) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t row = (size_t)node * C + 16 * rb + 4 * g + r, e = row * HID + x;
                    hv[rb][r] = epi.H[e];
                    if (EPI == EPI_BLEND) uv[rb][r] = epi.U[e];
                    if (EPI == EPI_GATES && !PL) side[rb][r] = (has_side && x < epi.cin) ? Z.p[0][row * L + x] : 0.f;
                }
        }
        if (next_node < nodes) load_rows(nxt, next_node);        // software prefetch: lands while this node computes
        __builtin_amdgcn_sched_barrier(0);
        const int lo = opaque(lane);

        // FmtH2: this node's activation scale sz = 2^k, from the maximum over every row the wave holds for it (stc_x3_frag.h); the wave's
        // running per-plane maxima go to the launch's slots after the loop (what the backward's dW products scale by)
        float sz = 1.f, invn = inv, invpn = invp;
        if constexpr (ASC) {
            float nm = 0.f;
#pragma unroll
            for (int n = 0; n < KL; ++n)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    const float ma = absmax4(cur[n][rb].a), mb = absmax4(cur[n][rb].b);
                    if constexpr (PL == 1) { pmax[n] = max_nonneg(pmax[n], ma); qmax[n] = max_nonneg(qmax[n], mb); }
                    else { pmax[n] = max_nonneg(pmax[n], g < 2 ? max_nonneg(ma, mb) : 0.f); qmax[n] = max_nonneg(qmax[n], g < 2 ? 0.f : ma); }
                    nm = vmax3_acc(nm, ma, mb);
                }
            sz = pow2_scale(wave_max_nonneg(nm), STC_ACT_TARGET_FWD);
            const float isz = inv_pow2(sz);
            invn = pow2_mul(inv, isz); invpn = pow2_mul(invp, isz);
        }

        f32x4 acc[NRB][NCB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = kZero4;

        // project: U[rb][cb] += Z_n rows (A) * W_{n,c} (B)
#pragma unroll
        for (int n = 0; n < K; ++n) {
            Op za[NRB];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                const Row8<L>& zr = cur[n][rb];
                if constexpr (ASC) za[rb] = F::split(zr.a * sz, zr.b * sz);
                else za[rb] = F::split(zr.a, zr.b);
            }
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const Op w = F::get(Wx, n * NCB + cb, lo);
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc[rb][cb] = F::mm(za[rb], w, acc[rb][cb]);
            }
        }

        // mix: Y[rb][hb] = U_0[rb][hb] + sum_{c>=1} T_c^T[rb][:] U_c[:][hb]     (U_c straight from its accumulators)
#pragma unroll
        for (int c1 = 0; c1 < K - 1; ++c1)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int p = 0; p < NB2; ++p) {
                    const Op u = F::split(acc[2 * p][(c1 + 1) * HB + hb], acc[2 * p + 1][(c1 + 1) * HB + hb]);
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) {
                        const Op t = F::get(Tx, (c1 * NRB + rb) * NB2 + p, lo);
                        acc[rb][hb] = F::mm(t, u, acc[rb][hb]);
                    }
                }

        if (EPI == EPI_NONE) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Y[((size_t)node * C + 16 * rb + 4 * g + r) * Ho + 16 * hb + x] = F::SCALED ? fmaf(acc[rb][hb][r], invn, bv[hb]) : acc[rb][hb][r] + bv[hb];
        } else if (EPI == EPI_GATES) {
            // Planar results leave in ROW layout, non-temporally: an accumulator-layout store instruction covers half a 128-byte line of each of
            // four rows, the row-layout one sixteen whole rows (the H tile is free once hv is read: it is the transposition scratch).  Timing
            // probe with the same bytes: gates forward 714 -> 700 us, and the blend aggregation that reads these planes next 569 -> 548 us.
            auto put_plane = [&](float* __restrict__ plane, const float (&v)[NRB][4]) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) th[((wave * NRB + rb) * 16 + 4 * g + r) * TRS + x] = v[rb][r];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
                    stc_st_once(reinterpret_cast<f32x4*>(plane + ((size_t)node * C + 16 * rb + x) * HID + 4 * g),
                                *reinterpret_cast<const f32x4*>(th + ((wave * NRB + rb) * 16 + x) * TRS + 4 * g));
                __builtin_amdgcn_wave_barrier();                // the next plane overwrites the tile
            };
            if constexpr (HT) {
                float uu[NRB][4], gt[NRB][4];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) uu[rb][r] = fast_sigmoid(F::SCALED ? fmaf(acc[rb][0][r], invn, bv[0]) : acc[rb][0][r] + bv[0]);
                put_plane(epi.U_out, uu);
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gt[rb][r] = fast_sigmoid(F::SCALED ? fmaf(acc[rb][HB - 1][r], invn, bv[HB - 1]) : acc[rb][HB - 1][r] + bv[HB - 1]);
                put_plane(epi.R_out, gt);
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gt[rb][r] *= hv[rb][r];                     // R*H
                if constexpr (POST) {
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) tr[((wave * NRB + rb) * 16 + 4 * g + r) * TRS + x] = gt[rb][r];
                    if (epi.CandIn) {                               // (the R*H plane is optional with POST)
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int rb = 0; rb < NRB; ++rb)
                            stc_st_once(reinterpret_cast<f32x4*>(epi.CandIn + ((size_t)node * C + 16 * rb + x) * HID + 4 * g),
                                        *reinterpret_cast<const f32x4*>(tr + ((wave * NRB + rb) * 16 + x) * TRS + 4 * g));
                    }
                } else {
                    put_plane(epi.CandIn, gt);
                }
            } else {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t row = (size_t)node * C + 16 * rb + 4 * g + r;
                    const float u = fast_sigmoid(F::SCALED ? fmaf(acc[rb][0][r], invn, bv[0]) : acc[rb][0][r] + bv[0]);
                    const float gate = fast_sigmoid(F::SCALED ? fmaf(acc[rb][HB - 1][r], invn, bv[HB - 1]) : acc[rb][HB - 1][r] + bv[HB - 1]);
                    epi.U_out[row * HID + x] = u;
                    epi.R_out[row * HID + x] = gate;
                    epi.CandIn[row * L + epi.cin + x] = gate * hv[rb][r];
                }
            }
            if constexpr (POST) {
                // ---- candidate projection on [Xt | R*H] (PL = 1) / [R*H | x | pad] (PL = 2): A = sum_c T_c^T (. Wc_{0,c}) + bc, Bm likewise with Wc_{1,c}
                __builtin_amdgcn_wave_barrier();
                Op zc[NRB];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    // this lane's row of R*H, columns 8(g&1)..: EVERY lane reads the tile and the row fragments are chosen by VALUE.
                    // (Branching between "a register of cur" and "a load from the tile" made the compiler select between two
                    // ADDRESSES instead: cur[0] was stored to scratch on every node -- 64 B x 64 lanes x 250 880 nodes = 1.03 GB of
                    // dead stores per launch, the 1.39x HBM traffic rocprofv3 showed for this kernel, profiles/r02/.)
                    f32x4 a4, b4;
                    if constexpr (PIECES) {                                               // piece order: [Xt columns 4g.. | R*H columns 4g..]
                        a4 = cur[0][rb].a;
                        b4 = *reinterpret_cast<const f32x4*>(tr + ((wave * NRB + rb) * 16 + x) * TRS + 4 * g);
                    } else {
                        const float* trow = tr + ((wave * NRB + rb) * 16 + x) * TRS + 8 * (g & 1);
                        const f32x4 ta = *reinterpret_cast<const f32x4*>(trow), tb = *reinterpret_cast<const f32x4*>(trow + 4);
                        const f32x4 xa = cur[0][rb].a;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {                                     // [R*H | the narrow input columns | zero]
                            a4[i] = g < 2 ? ta[i] : (g == 2 ? xa[i] : 0.f);
                            b4[i] = g < 2 ? tb[i] : 0.f;
                        }
                    }
                    if constexpr (ASC) zc[rb] = F::split(a4 * sz, b4 * sz);          // (|R*H| <= |H|: the node's scale covers it)
                    else zc[rb] = F::split(a4, b4);
                }
                f32x4 pa[K][NRB][K];
#pragma unroll
                for (int n = 0; n < K; ++n)
#pragma unroll
                    for (int c = 0; c < K; ++c) {
                        const Op w = F::get(Wp, n * K + c, lo);
#pragma unroll
                        for (int rb = 0; rb < NRB; ++rb) pa[n][rb][c] = F::mm(zc[rb], w, kZero4);
                    }
#pragma unroll
                for (int n = 0; n < K; ++n)
#pragma unroll
                    for (int p = 0; p < NB2; ++p) {
                        const Op u = F::split(pa[n][2 * p][1], pa[n][2 * p + 1][1]);
#pragma unroll
                        for (int rb = 0; rb < NRB; ++rb) {
                            const Op t = F::get(Tx, rb * NB2 + p, lo);
                            pa[n][rb][0] = F::mm(t, u, pa[n][rb][0]);
                        }
                    }
                const float bcv = post.bc ? post.bc[x] : 0.f;
                float ov[NRB][4];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[rb][r] = F::SCALED ? fmaf(pa[0][rb][0][r], invpn, bcv) : pa[0][rb][0][r] + bcv;
                put_plane(post.A, ov);
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[rb][r] = F::SCALED ? pa[1][rb][0][r] * invpn : pa[1][rb][0][r];
                put_plane(post.Bm, ov);
            }
            if (has_side) {
                const int scol = x < epi.cin ? x : x + HID;
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        epi.CandIn[((size_t)node * C + 16 * rb + 4 * g + r) * L + scol] = side[rb][r];
            }
        } else {
            float hn[NRB][4];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t e = ((size_t)node * C + 16 * rb + 4 * g + r) * HID + x;
                    const float c = fast_tanh(F::SCALED ? fmaf(acc[rb][0][r], invn, bv[0]) : acc[rb][0][r] + bv[0]);
                    const float u = uv[rb][r];
                    hn[rb][r] = (1.f - u) * hv[rb][r] + u * c;
                    epi.Cand[e] = c;
                    epi.Hnew[e] = hn[rb][r];
                }
            store_state_copies<NRB>(epi, (size_t)node * C, x, g, hn);
        }
#pragma unroll
        for (int n = 0; n < KL; ++n)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) cur[n][rb] = nxt[n][rb];
        node = next_node;
    }
    if constexpr (ASC) {
        if (epi.zmax) {                                          // rows [p planes of slab 0..K-1 | q planes of slab 0..K-1], launch order of Z
            const int slot = blockIdx.x * MF_WAVES + wave;
#pragma unroll
            for (int n = 0; n < KL; ++n) { leave_max(epi.zmax + n * STC_ACT_SLOTS, slot, pmax[n]); leave_max(epi.zmax + (KL + n) * STC_ACT_SLOTS, slot, qmax[n]); }
        }
    }
}

// --------------------------------------------------------------------------------------- backward
// Per node, with Q_0 = dY and Q_c = T_c dY (c >= 1), all contractions on the split-operand MFMA:
//   Qv_c (rows o, cols c')  = dY^T . T_c^T           A = dY in accumulator layout (slots = rows d), B = T_c table
//   dZ_n^T (rows l, cols c') = sum_{c,o} W . Q_c^T    A = W table (slots = (c, o) blocks of 16), B = dY rows / Qv accumulators
//   Qd_c (rows c', cols o)  = T_c . dY               A = T_c table (same table), B = dY in accumulator layout
//   dW_{n,c} (rows l, cols o) += Z_n^T . Q_c         A = Z columns (slots = rows c'), B = dY / Qd accumulators
// dW / db stay in registers across all nodes of a wave; fixed-order combine at the end (combine_dw).
template <int NB2, int HB, int K, int L, int PL = 0>
struct NodeIn {      // what one node contributes from HBM: its dY fragments and its Z columns
    static constexpr int NRB = 2 * NB2, LB = (L + 15) / 16;
    DyFrag<NRB, HB> g;
    float za[K][LB][NRB][4];     // Z_n[16kb + 4g + t][16lb + x]
    __device__ __forceinline__ void load_z(const ZPtrs& Z, int node, int x, int gq, int cinx = 0) {
        constexpr int C = 32 * NB2;
        const size_t r0 = (size_t)node * C;
        if constexpr (PL == 2) {        // [state plane (16) | input plane (cinx) | pad]: block 0 from p, column x < cinx of block 1 from q
#pragma unroll
            for (int n = 0; n < K; ++n)
#pragma unroll
                for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const size_t row = r0 + 16 * kb + 4 * gq + t;
                        za[n][0][kb][t] = Z.p[n][row * 16 + x];
                        za[n][1][kb][t] = x < cinx ? Z.q[n][row * cinx + x] : 0.f;
                    }
            return;
        }
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int lb = 0; lb < LB; ++lb) {
                const bool ok = 16 * lb + x < L;
                constexpr int LD = PL == 1 ? 16 : L;             // planar: block lb of the row is plane lb, 16 floats per row
                const float* col = PL == 1 ? (lb == 0 ? Z.p[n] : Z.q[n]) + r0 * 16 + x : Z.p[n] + r0 * L + 16 * lb + (ok ? x : 0);
#pragma unroll
                for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float zv = col[(size_t)(16 * kb + 4 * gq + t) * LD];
                        za[n][lb][kb][t] = ok ? zv : 0.f;
                    }
            }
    }
};

// One-wave kernels with a gate / blend prologue (order 3, C = 64): what the prologue reads per node, in ROW layout only (lane (x, g): row
// 16kb + x, columns 4g .. 4g+3; six 16-byte loads per 16 rows), so that the NEXT node's planes can be requested a node ahead in 24-48
// registers.  With one wave per SIMD nobody else covers a load: rocprofv3 showed the order-3 gates backward 62 % parked on memory at 16 %
// matrix-pipe busy.  The prologue's products are formed once, in row layout, and reach the accumulator layout (row 16kb + 4g + t, column x)
// through a per-wave LDS tile -- the old prologue loaded and multiplied everything a second time in that layout (48 four-byte loads).
template <int NRB>
struct GateRows {
    f32x4 u[NRB], r[NRB], h[NRB], c[NRB], gn[NRB], dci[NRB];
    template <bool GATES>
    __device__ __forceinline__ void load(const BwdPro& p, int node, int x, int g) {
        constexpr int C = 16 * NRB;
#pragma unroll
        for (int kb = 0; kb < NRB; ++kb) {
            const size_t e = ((size_t)node * C + 16 * kb + x) * 16 + 4 * g;
            u[kb] = stc_ld_once(reinterpret_cast<const f32x4*>(p.U + e));
            c[kb] = stc_ld_once(reinterpret_cast<const f32x4*>(p.Cand + e));
            gn[kb] = stc_ld_once(reinterpret_cast<const f32x4*>(p.dH_in + e));
            if constexpr (GATES) {
                r[kb] = stc_ld_once(reinterpret_cast<const f32x4*>(p.R + e));
                h[kb] = stc_ld_once(reinterpret_cast<const f32x4*>(p.H + e));
                dci[kb] = stc_ld_once(reinterpret_cast<const f32x4*>(p.dCandIn + e));      // planar: the R*H plane's gradient, rows of 16
            }
        }
    }
};

// dY of the gates convolution (PRO_GATES_CAND, HB = 2) or of the candidate convolution (PRO_BLEND, HB = 1) from the row-layout planes, both
// register layouts; the gates form also leaves the state's share dH = dRH R + dHnew (1 - U) in the stash (FOLD) or in p.dH.
template <int NRB, int HB, bool FOLD, int TRS>
__device__ __forceinline__ void form_dy(DyFrag<NRB, HB>& gq, const GateRows<NRB>& w, const BwdPro& p, int node, int x, int g, float4* stash, float* tile) {
    constexpr int C = 16 * NRB;
#pragma unroll
    for (int kb = 0; kb < NRB; ++kb) {
        f32x4 dh;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (HB == 2) {
                const float u = w.u[kb][i], r = w.r[kb][i], hh = w.h[kb][i], d = w.dci[kb][i], gn = w.gn[kb][i];
                gq.v[kb][0][i] = gn * (w.c[kb][i] - hh) * u * (1.f - u);
                gq.v[kb][1][i] = d * hh * r * (1.f - r);
                dh[i] = d * r + gn * (1.f - u);
            } else {
                const float cc = w.c[kb][i];
                gq.v[kb][0][i] = w.gn[kb][i] * w.u[kb][i] * (1.f - cc * cc);
            }
        }
        if constexpr (HB == 2) {
            if constexpr (FOLD) stash[kb * 64 + (g * 16 + x)] = make_float4(dh[0], dh[1], dh[2], dh[3]);
            else stc_st_once(reinterpret_cast<f32x4*>(p.dH + ((size_t)node * C + 16 * kb + x) * 16 + 4 * g), dh);
        }
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) *reinterpret_cast<f32x4*>(tile + ((hb * NRB + kb) * 16 + x) * TRS + 4 * g) = gq.v[kb][hb];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
#pragma unroll
            for (int t = 0; t < 4; ++t) gq.d[kb][hb][t] = tile[((hb * NRB + kb) * 16 + 4 * g + t) * TRS + x];
    __builtin_amdgcn_wave_barrier();
}

// C = 32 with K <= 2 fits two waves per SIMD (<= 256 registers): the partner wave hides load and LDS latency, so no
// software prefetch; the larger shapes run one wave per SIMD and prefetch the next node's operands instead.
template <int NB2, int K>
struct BwdSched { static constexpr int waves = (NB2 == 1 && K <= 2) ? 2 : 1; };

// ACC (planar, PL = 1): the X-side gradient planes dZ.p[n] already hold another convolution's gradients for the same planes (the
// candidate's) and this kernel ADDS its own: the tile's accumulators start from the stored values -- the source then gets one plane per
// Chebyshev order from the cell instead of two.
// F = FmtH2 (two fp16 pieces, three products; see cell_bwd_x3_kernel): the gradient fragments are taken into a space scaled by sg = 2^k
// (per node, from the node's own maximum: stc_x3_frag.h, RunScale) right after they are loaded / formed, W's blocks of c = 0 carry sW sT, those of c >= 1
// carry sW and the T_c tables sT, so every (c, o) block of a dZ contraction arrives with the factor sg sT sW, taken out at the store.
struct BwdArgs {       // the kernel's one parameter (read afresh per pass on the fp16 x 2 format: stc_x3_frag.h, kernargs_fresh)
    ZPtrs Z; const float *Tc, *W, *dY; DZPtrs dZ; float* partial; int nodes, want_db, Lw; BwdPro pro;
};
template <int NB2, int HB, int K, int L, int PRO, int PL = 0, int FOLD = 0, int ACC = 0, class F = FmtB3>      // FOLD: see load_gates_grad (planar gates backward only)
__global__ __launch_bounds__(MF_THREADS, (BwdSched<NB2, K>::waves)) void node_bwd_x3_kernel(BwdArgs args_in_kernarg_segment) {
    using Op = typename F::Op;
    constexpr int NP = F::NP;
    constexpr int NRB = 2 * NB2, C = 32 * NB2, Ho = 16 * HB, LB = (L + 15) / 16;
    constexpr int NBK = K * HB, S = (NBK + 1) / 2;             // (c, o) blocks of 16 and 32-wide steps over them
    constexpr int nTB = (K - 1) * NRB * NB2, nWA = K * LB * S;
    constexpr bool PF = PRO == PRO_NONE && BwdSched<NB2, K>::waves == 1;      // prefetch the next node's operands
    // ... also with a gate / blend prologue on planar inputs: row-layout planes a node ahead (GateRows), dY through an LDS tile
    constexpr bool PFG = (PRO == PRO_GATES_CAND || PRO == PRO_BLEND) && PL != 0 && BwdSched<NB2, K>::waves == 1;
    constexpr int TRS = 20;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* TB = reinterpret_cast<u32x4*>(smem_raw);     // [K-1][NRB rb][NB2 p]   T_c[16rb + x][32p + pair_row]
    u32x4* WA = TB + nTB * NP * 64;                      // [K n][LB][S]           W[(n, c, 16lb + x)][16hb + 4g + (e&3)], block 2s + (e>>2)
    static_assert(!FOLD || ((PRO == PRO_GATES || PRO == PRO_GATES_CAND) && PL != 0), "FOLD belongs to the planar gates backward");
    const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, g = lane >> 4;
    float4* stash = reinterpret_cast<float4*>(WA + nWA * NP * 64) + (tid >> 6) * NRB * 64;    // FOLD: [wave][NRB][64 lanes], lane-private
    float* dy_tile = reinterpret_cast<float*>(reinterpret_cast<float4*>(WA + nWA * NP * 64) + (FOLD ? MF_WAVES * NRB * 64 : 0))
                     + (tid >> 6) * (HB * NRB * 16 * TRS);                                    // PFG: [wave][HB][NRB][16 rows][TRS]

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = gridDim.x * MF_WAVES;
    // fp16 x 2: the kernel body runs in PASSES.  A node whose gradients exceed the sums' scale by more than 2^12 ends its wave's pass (stc_x3_frag.h:
    // RunScale): computed to its end with its sums muted and nothing new stored, then the wave leaves the loop at the latch.  After the
    // combine the body runs again while any wave of the workgroup has nodes left -- tables refilled, operands requested afresh, sums empty,
    // the partial row added to; nothing is live across passes but the node index.
    int node = blockIdx.x * MF_WAVES + wave;
    // The body is written once and compiled TWICE (see cell_bwd_x3_kernel): the first pass as straight-line code, later passes -- rare -- in a
    // loop with their arguments read afresh from the kernarg segment.
    auto run_pass = [&](const int pass, const BwdArgs ka) __attribute__((always_inline)) {
    const ZPtrs& Z = ka.Z;
    const DZPtrs& dZ = ka.dZ;
    const BwdPro& pro = ka.pro;
    const float* __restrict__ Tc = ka.Tc;
    const float* __restrict__ W = ka.W;
    const float* __restrict__ dY = ka.dY;
    float* __restrict__ partial = ka.partial;
    const int nodes = ka.nodes, want_db = ka.want_db, Lw = ka.Lw;
    float sT = 1.f, sW = 1.f;                            // FmtH2: table scales from the tables' own maxima
    if constexpr (F::SCALED) {
        float* scratch = reinterpret_cast<float*>(smem_raw);
        if (K > 1) {
            sT = clamp_mix_scale(pow2_scale(block_absmax(Tc + (size_t)C * C, (K - 1) * C * C, scratch, MF_THREADS), STC_T_TARGET_BWD));
        }
        sW = pow2_scale(block_absmax(W, K * K * Lw * Ho, scratch, MF_THREADS), STC_W_TARGET);
    }
    RunScale rs;                                         // FmtH2: gradient scales (stc_x3_frag.h: per node + the wave's reference for the sums over nodes)
    // activation operands of the dW products: one scale per plane and launch, from the slots the forward launch filled (block lb of slab n =
    // row lb K + n; see cell_bwd_x3_kernel)
    float sz[K][LB];
#pragma unroll
    for (int n = 0; n < K; ++n)
#pragma unroll
        for (int lb = 0; lb < LB; ++lb) sz[n][lb] = (F::SCALED && PL != 0) ? plane_scale(pro.zmax, lb * K + n) : 1.f;

    for (int idx = tid; idx < nTB * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, p = f % NB2, rb = (f / NB2) % NRB, c1 = f / (NB2 * NRB), gg = ll >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = Tc[(size_t)(c1 + 1) * C * C + (16 * rb + (ll & 15)) * C + 32 * p + pair_row(gg, e)];
        F::put(TB, f, ll, v, sT);
    }
    for (int idx = tid; idx < nWA * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, s = f % S, lb = (f / S) % LB, n = f / (S * LB), gg = ll >> 4;
        const int l = 16 * lb + (ll & 15);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int b = 2 * s + (e >> 2), c = b / HB, hb = b % HB;
            const int wl = PL == 2 ? stc_wrow_swapped(l, Lw - 16) : l;
            v[e] = (b < NBK && wl >= 0 && wl < Lw) ? W[((size_t)(n * K + c) * Lw + wl) * Ho + 16 * hb + 4 * gg + (e & 3)] * (F::SCALED && c == 0 ? sT : 1.f) : 0.f;
        }
        F::put(WA, f, ll, v, sW);
    }
    __syncthreads();

    const float kz = uniform_bits(sT * sW), ikz = inv_pow2(kz);      // (1 for FmtB3; scalar registers)

    f32x4 dWt[K][LB][K][HB];          // dW tiles: rows l = 16lb + 4g + r, columns o = 16hb + x
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

    static_assert(PL != 1 || L == 32, "planar rows are 16 + 16 columns");
    static_assert(PL != 2 || L == 20, "narrow planar rows are 16 + cin columns padded to 20");
    NodeIn<NB2, HB, K, L, PL> in, nx;
    GateRows<NRB> rows_cur, rows_nxt;
    if (PF && node < nodes) { in.g.load(dY, node, x, g); in.load_z(Z, node, x, g, Lw - 16); }      // (the width matters to PL = 2 only)
    if (PFG && node < nodes) { rows_cur.template load<PRO == PRO_GATES_CAND>(pro, node, x, g); in.load_z(Z, node, x, g, Lw - 16); }
    if (PF || PFG) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) before the loop: see node_fwd_x3_kernel
    bool skipped = false;                                   // this node ends the wave's pass: the next pass starts AT it
    int resume = -1;
    while (node < nodes) {
        const int next_node = node + nw;
        const size_t r0 = (size_t)node * C;
        if constexpr (PFG) {
            if (next_node < nodes) { rows_nxt.template load<PRO == PRO_GATES_CAND>(pro, next_node, x, g); nx.load_z(Z, next_node, x, g, Lw - 16); }
            __builtin_amdgcn_sched_barrier(0);
            form_dy<NRB, HB, FOLD != 0, TRS>(in.g, rows_cur, pro, node, x, g, stash, dy_tile);
        }
        else if constexpr (PRO == PRO_GATES || PRO == PRO_GATES_CAND) {
            // planar: dCandIn is the R*H plane's gradient alone, (nodes, C, 16) with the state columns at offset 0
            load_gates_grad<NRB, HB, (PL ? 16 : L), PRO == PRO_GATES_CAND, FOLD != 0>(in.g, pro, node, x, g, stash);
            in.load_z(Z, node, x, g, Lw - 16);
        }
        else if constexpr (PRO == PRO_BLEND) { load_blend_grad<NRB>(in.g, pro, node, x, g); in.load_z(Z, node, x, g, Lw - 16); }
        else if (!PF) { in.g.load(dY, node, x, g); in.load_z(Z, node, x, g, Lw - 16); }
        if (PF && next_node < nodes) { nx.g.load(dY, next_node, x, g); nx.load_z(Z, next_node, x, g, Lw - 16); }
        if (PF) __builtin_amdgcn_sched_barrier(0);
        const int lo = opaque(lane);
        float sg = 1.f, sh = 1.f;                              // FmtH2: this node's gradient scale a_n, and a / a_n for the sums over nodes
        if constexpr (F::SCALED) {                             // into the node's own scale: everything below is linear in the gradient
            float m = 0.f;
#pragma unroll
            for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) m = max_nonneg(m, absmax4(in.g.v[kb][hb]));
            sg = rs.node(wave_max_bits(m), sh, skipped);
            // (what a stopped node's prologue has stored already -- the state's share -- is stored again, alike, when the node is redone)
#pragma unroll
            for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) { in.g.d[kb][hb] *= sg; in.g.v[kb][hb] *= sg; }
        }
        // (a skipped node stores zeros, or what the plane held: whatever it stores is stored again when it is redone -- no branch at the stores)
        const float ikz_sg = F::SCALED ? (skipped ? 0.f : pow2_mul(ikz, inv_pow2(sg))) : 1.f;
        const DyFrag<NRB, HB>& gr = in.g;

#pragma unroll
        for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
                dbp[hb] += sh * ((gr.d[kb][hb][0] + gr.d[kb][hb][1]) + (gr.d[kb][hb][2] + gr.d[kb][hb][3]));

        // dY in accumulator layout as an operand: slots = rows d of the tile pair p
        Op gd[HB][NB2];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
#pragma unroll
            for (int p = 0; p < NB2; ++p) gd[hb][p] = F::split(gr.d[2 * p][hb], gr.d[2 * p + 1][hb]);

        // ---- Qv_c tiles (rows o, columns c')
        f32x4 Qv[AtLeast1<K - 1>::v][NRB][HB];
#pragma unroll
        for (int c1 = 0; c1 < K - 1; ++c1)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) Qv[c1][rb][hb] = kZero4;
#pragma unroll
                for (int p = 0; p < NB2; ++p) {
                    const Op t = F::get(TB, (c1 * NRB + rb) * NB2 + p, lo);
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) Qv[c1][rb][hb] = F::mm(gd[hb][p], t, Qv[c1][rb][hb]);
                }
            }

        // ---- B operands of dZ: step s covers the (c, o) blocks 2s, 2s+1; block (c, hb) of Q_c^T for columns c' = 16rb + x
        Op qb[S][NRB];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                f32x4 blk[2];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int b = 2 * s + h2, c = b / HB, hb = b % HB;
                    blk[h2] = b >= NBK ? kZero4 : (c == 0 ? gr.v[rb][hb] : Qv[c > 0 ? c - 1 : 0][rb][hb]);
                }
                qb[s][rb] = F::split(blk[0], blk[1]);
            }

        // ---- dZ_n^T tile (rows l, columns c')
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int lb = 0; lb < LB; ++lb) {
                f32x4 z[NRB];
                constexpr int HLB = PL == 2 ? 0 : 1;            // the block of the row that is the H (state) plane
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    if (FOLD && n == 0 && lb == HLB) {          // the tile starts from the state's share parked by the prologue
                        const float4 sh = stash[rb * 64 + lane];
                        z[rb] = f32x4{sh.x, sh.y, sh.z, sh.w};
                        if constexpr (F::SCALED) z[rb] *= pow2_mul(sg, kz);      // (parked unscaled)
                    } else if (ACC && PL == 1 && lb == 0 && !F::SCALED) {     // ... or from what the plane already holds
                        z[rb] = stc_ld_once(reinterpret_cast<const f32x4*>(dZ.p[n] + (r0 + 16 * rb + x) * 16 + 4 * g));
                    } else {
                        z[rb] = kZero4;
                    }
                }
                // ACC (fp16 x 2): what the plane already holds is requested HERE, one tile ahead of its use -- the matrix products of the tile cover
                // the latency -- and lives for this tile only.  Requested for all K planes at the top of the node (24 registers through the whole
                // body) the order-3 kernel spilled seven plane addresses and re-read them from scratch on every node: 301 -> 383 us per launch.
                f32x4 held[NRB];
                if constexpr (F::SCALED && ACC != 0 && PL == 1) {
                    if (lb == 0) {
#pragma unroll
                        for (int rb = 0; rb < NRB; ++rb) held[rb] = stc_ld_once(reinterpret_cast<const f32x4*>(dZ.p[n] + (r0 + 16 * rb + x) * 16 + 4 * g));
                    }
                }
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const Op w = F::get(WA, (n * LB + lb) * S + s, lo);
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) z[rb] = F::mm(w, qb[s][rb], z[rb]);
                }
                if constexpr (F::SCALED) {                      // out of the scaled space (+ what the plane already holds)
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) {
                        if (ACC && PL == 1 && lb == 0) z[rb] = z[rb] * ikz_sg + held[rb];
                        else z[rb] *= ikz_sg;
                    }
                }
                if constexpr (PL == 1) {                        // planar gradient slabs: block lb of the row goes to plane lb
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb)
                        stc_st_once(reinterpret_cast<f32x4*>((lb == 0 ? dZ.p[n] : dZ.q[n]) + (r0 + 16 * rb + x) * 16 + 4 * g), z[rb]);
                } else if constexpr (PL == 2) {                 // narrow input plane: only the state plane's gradient is wanted
                    if (lb == 0) {
#pragma unroll
                        for (int rb = 0; rb < NRB; ++rb)
                            stc_st_once(reinterpret_cast<f32x4*>(dZ.p[n] + (r0 + 16 * rb + x) * 16 + 4 * g), z[rb]);
                    }
                } else if (16 * lb + 4 * g < L) {
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb)
                        *reinterpret_cast<f32x4*>(dZ.p[n] + (r0 + 16 * rb + x) * L + 16 * lb + 4 * g) = z[rb];
                }
            }

        // ---- Qd_c tiles (rows c', columns o), then as operands: slots = rows c' of the tile pair p
        Op qd[AtLeast1<K - 1>::v][HB][NB2];
#pragma unroll
        for (int c1 = 0; c1 < K - 1; ++c1) {
            f32x4 Qd[NRB][HB];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) Qd[rb][hb] = kZero4;
#pragma unroll
                for (int p = 0; p < NB2; ++p) {
                    const Op t = F::get(TB, (c1 * NRB + rb) * NB2 + p, lo);
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) Qd[rb][hb] = F::mm(t, gd[hb][p], Qd[rb][hb]);
                }
            }
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int p = 0; p < NB2; ++p) qd[c1][hb][p] = F::split(Qd[2 * p][hb], Qd[2 * p + 1][hb]);
        }

        // ---- dW_{n,c} tile (rows l, columns o) += Z_n^T . Q_c
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int lb = 0; lb < LB; ++lb)
#pragma unroll
                for (int p = 0; p < NB2; ++p) {
                    const float (&zc)[NRB][4] = in.za[n][lb];
                    Op a;
                    if constexpr (F::SCALED) {
                        const float zs = pow2_mul(sz[n][lb], sh);                                // (a skipped node: sh = RunScale::mute())
                        a = F::split(f32x4{zc[2 * p][0], zc[2 * p][1], zc[2 * p][2], zc[2 * p][3]} * zs, f32x4{zc[2 * p + 1][0], zc[2 * p + 1][1], zc[2 * p + 1][2], zc[2 * p + 1][3]} * zs);
                    }
                    else a = F::split(f32x4{zc[2 * p][0], zc[2 * p][1], zc[2 * p][2], zc[2 * p][3]},
                                      f32x4{zc[2 * p + 1][0], zc[2 * p + 1][1], zc[2 * p + 1][2], zc[2 * p + 1][3]});
#pragma unroll
                    for (int c = 0; c < K; ++c)
#pragma unroll
                        for (int hb = 0; hb < HB; ++hb)
                            dWt[n][lb][c][hb] = F::mm(a, c == 0 ? gd[hb][p] : qd[c > 0 ? c - 1 : 0][hb][p], dWt[n][lb][c][hb]);
                }
        if (PF) in = nx;
        if constexpr (PFG) {
            rows_cur = rows_nxt;
#pragma unroll
            for (int n = 0; n < K; ++n)
#pragma unroll
                for (int lb = 0; lb < LB; ++lb)
#pragma unroll
                    for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
                        for (int t = 0; t < 4; ++t) in.za[n][lb][kb][t] = nx.za[n][lb][kb][t];
        }
        resume = skipped ? node : resume;                    // the wave's pass ends AT this node: no exit of its own, the loop condition does it
        node = skipped ? 0x7fffffff : next_node;
    }
    if (resume >= 0) node = resume;
    const float isg = rs.unscale();                                             // dW tiles of block c carry the wave's final gradient scale (c = 0) or that times sT (c >= 1); db the scale
    PlaneUnscale<K, LB> pu;
#pragma unroll
    for (int n = 0; n < K; ++n)
#pragma unroll
        for (int lb = 0; lb < LB; ++lb) pu.v[n][lb] = inv_pow2(sz[n][lb]);
    combine_dw<K, LB, HB>(reinterpret_cast<float*>(smem_raw), dWt, dbp, partial, Lw, want_db, PL == 2 ? Lw - 16 : -1, isg, isg / sT, isg, pu, pass > 0);
    };
    run_pass(0, args_in_kernarg_segment);
    if constexpr (F::SCALED) {
        const int n_all = args_in_kernarg_segment.nodes;
        // (the barrier also means: every wave is done with the combine's slabs before the tables are filled again)
        for (int pass = 1; __syncthreads_or(node < n_all); ++pass) run_pass(pass, kernargs_fresh<BwdArgs>());
    }
}

// --------------------------------------------------------------------------------------- post-aggregation form (K = 2)
// The aggregation acts on the node axis and the projection / category mix on the other two, so they commute:
//     Y = sum_c T_c^T (X W_{0,c}) + S . sum_c T_c^T (X W_{1,c})  =  A + S.Bm        (reference STC_GNN.py:35-45 reassociated)
// i.e. the SpMM can run AFTER the node kernel, on rows of C*Ho floats instead of C*L -- half the bytes for the candidate
// convolution (Ho = 16 against L = 32) -- and the slab Z_1 = S.X is never formed, stored or re-read.  Backward of this
// form: dA = dY, dBm = S^T dY (one narrow SpMM by the caller), and the kernel below turns (X, dA, dBm) into dX and dW
// directly; there is no second gradient slab and no SpMM after it.
//   per slab n (dY_0 = dA, dY_1 = dBm):  Q^n_0 = dY_n, Q^n_1 = T_1 dY_n
//   dX^T (rows l, cols c') = sum_n sum_{c,o} W[(n,c,l)][o] Q^n_c[c'][o]        dW_{n,c} (rows l, cols o) += X^T Q^n_c
// forward of the post-aggregation form: one input slab X, the two weight sets of the Chebyshev orders kept apart
//     A = sum_c T_c^T (X W_{0,c}) + b      Bm = sum_c T_c^T (X W_{1,c})           (nodes, C, Ho) each
template <int NB2, int HB, int L, int PL = 0>       // PL = 1: X is two planes, X (columns 0..15) and X2 (columns 16..31); PL = 2 (L = 20): X =
                                                   // the 16-wide plane, X2 = the narrow input plane (cin = Lw - 16), columns [X | X2 | pad]
__global__ __launch_bounds__(MF_THREADS, (NB2 == 1 ? 2 : 1)) void node_fwd2_x3_kernel(
    const float* __restrict__ X, const float* __restrict__ X2, const float* __restrict__ Tc, const float* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ A, float* __restrict__ Bm, int nodes, int Lw) {
    constexpr int K = 2, NRB = 2 * NB2, C = 32 * NB2, Ho = 16 * HB, NCB = K * HB;
    constexpr int nWx = K * NCB, nTx = NRB * NB2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* Wx = reinterpret_cast<u32x4*>(smem_raw);        // [K n][NCB]     B: W[(n, c, l = slot)][o = 16 hb + x]
    u32x4* Tx = Wx + nWx * 3 * 64;                          // [NRB rb][NB2]  A: T_1[c' = 32 p + pair_row][d = 16 rb + x]
    const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, g = lane >> 4;

    for (int idx = tid; idx < nWx * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, cb = f % NCB, n = f / NCB;
        const int c = cb / HB, o = (cb % HB) * 16 + (ll & 15), gg = ll >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int l = 8 * gg + e;
            const int wl = PL == 2 ? stc_wrow_swapped(l, Lw - 16) : l;
            v[e] = (wl >= 0 && wl < Lw) ? W[((size_t)(n * K + c) * Lw + wl) * Ho + o] : 0.f;
        }
        put_frag(Wx, f, ll, v);
    }
    for (int idx = tid; idx < nTx * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, p = f % NB2, rb = f / NB2, gg = ll >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = Tc[(size_t)C * C + (32 * p + pair_row(gg, e)) * C + 16 * rb + (ll & 15)];
        put_frag(Tx, f, ll, v);
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = gridDim.x * MF_WAVES;
    float bv[HB];
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) bv[hb] = bias ? bias[16 * hb + x] : 0.f;

    int node = blockIdx.x * MF_WAVES + wave;
    Row8<L> cur[NRB], nxt[NRB];
    auto load_rows = [&](Row8<L> (&z)[NRB], int nd) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            if constexpr (PL == 1) z[rb].load_planes(X + ((size_t)nd * C + 16 * rb + x) * 16, X2 + ((size_t)nd * C + 16 * rb + x) * 16, g);
            else if constexpr (PL == 2) z[rb].load_planes_narrow(X + ((size_t)nd * C + 16 * rb + x) * 16, X2 + ((size_t)nd * C + 16 * rb + x) * (Lw - 16), Lw - 16, g);
            else z[rb].load(X + ((size_t)nd * C + 16 * rb + x) * L, g);
        }
    };
    if (node < nodes) load_rows(cur, node);
    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0) before the loop: see node_fwd_x3_kernel
    while (node < nodes) {
        const int next_node = node + nw;
        if (next_node < nodes) load_rows(nxt, next_node);
        __builtin_amdgcn_sched_barrier(0);
        const int lo = opaque(lane);

        f32x4 acc[K][NRB][NCB];
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[n][rb][cb] = kZero4;
        X3 za[NRB];                                              // the rows are split once for both weight sets
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) za[rb] = split8(cur[rb].a, cur[rb].b);
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const X3 w = get_frag(Wx, n * NCB + cb, lo);
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc[n][rb][cb] = mma6(za[rb], w, acc[n][rb][cb]);
            }
        // mix per weight set: [rb][hb] += T_1^T [.][HB + hb]
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int p = 0; p < NB2; ++p) {
                    const X3 u = split8(acc[n][2 * p][HB + hb], acc[n][2 * p + 1][HB + hb]);
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) {
                        const X3 t = get_frag(Tx, rb * NB2 + p, lo);
                        acc[n][rb][hb] = mma6(t, u, acc[n][rb][hb]);
                    }
                }
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t o = ((size_t)node * C + 16 * rb + 4 * g + r) * Ho + 16 * hb + x;
                    A[o] = acc[0][rb][hb][r] + bv[hb];
                    Bm[o] = acc[1][rb][hb][r];
                }
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) cur[rb] = nxt[rb];
        node = next_node;
    }
}

struct Bwd2Args {      // the kernel's one parameter (read afresh per pass on the fp16 x 2 format: stc_x3_frag.h, kernargs_fresh)
    const float *X, *X2, *Tc, *W, *dA, *dB; float *dX, *dX2, *partial; int nodes, want_db, Lw; const float *zmax_x, *zmax_x2;
};
template <int NB2, int HB, int L, int PL = 0, class F = FmtB3>      // F: operand format, scales as in node_bwd_x3_kernel
__global__ __launch_bounds__(MF_THREADS, (NB2 == 1 ? 2 : 1)) void node_bwd2_x3_kernel(Bwd2Args args_in_kernarg_segment) {
    // zmax_x / zmax_x2 (fp16 x 2, optional): 256 slots each of max |X| / max |X2| (a row of the slots a forward launch filled; R*H takes H's): the
    // scales of the dW products' activation operands.
    using Op = typename F::Op;
    constexpr int NP = F::NP;
    constexpr int K = 2, NRB = 2 * NB2, C = 32 * NB2, Ho = 16 * HB, LB = (L + 15) / 16;
    constexpr int NBK = K * HB, S = (NBK + 1) / 2;
    constexpr int nTB = NRB * NB2, nWA = K * LB * S;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* TB = reinterpret_cast<u32x4*>(smem_raw);     // [NRB rb][NB2 p]        T_1[16rb + x][32p + pair_row]
    u32x4* WA = TB + nTB * NP * 64;                      // [K n][LB][S]           W[(n, c, 16lb + x)][16hb + 4g + (e&3)], block 2s + (e>>2)
    const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = gridDim.x * MF_WAVES;
    // fp16 x 2: the kernel body runs in PASSES.  A node whose gradients exceed the sums' scale by more than 2^12 ends its wave's pass (stc_x3_frag.h:
    // RunScale): computed to its end with its sums muted and nothing new stored, then the wave leaves the loop at the latch.  After the
    // combine the body runs again while any wave of the workgroup has nodes left -- tables refilled, operands requested afresh, sums empty,
    // the partial row added to; nothing is live across passes but the node index.
    int node = blockIdx.x * MF_WAVES + wave;
    // The body is written once and compiled TWICE (see cell_bwd_x3_kernel): the first pass as straight-line code, later passes -- rare -- in a
    // loop with their arguments read afresh from the kernarg segment.
    auto run_pass = [&](const int pass, const Bwd2Args ka) __attribute__((always_inline)) {
    const float* __restrict__ X = ka.X; const float* __restrict__ X2 = ka.X2; const float* __restrict__ Tc = ka.Tc; const float* __restrict__ W = ka.W;
    const float* __restrict__ dA = ka.dA; const float* __restrict__ dB = ka.dB;
    float* __restrict__ dX = ka.dX; float* __restrict__ dX2 = ka.dX2; float* __restrict__ partial = ka.partial;
    const int nodes = ka.nodes, want_db = ka.want_db, Lw = ka.Lw;
    const float* __restrict__ zmax_x = ka.zmax_x; const float* __restrict__ zmax_x2 = ka.zmax_x2;
    float sT = 1.f, sW = 1.f;
    if constexpr (F::SCALED) {
        float* scratch = reinterpret_cast<float*>(smem_raw);
        sT = clamp_mix_scale(pow2_scale(block_absmax(Tc + (size_t)C * C, C * C, scratch, MF_THREADS), STC_T_TARGET_BWD));
        sW = pow2_scale(block_absmax(W, K * K * Lw * Ho, scratch, MF_THREADS), STC_W_TARGET);
    }
    RunScale rs;                                         // FmtH2: gradient scales (stc_x3_frag.h: per node + the wave's reference for the sums over nodes)
    float sz[LB];
#pragma unroll
    for (int lb = 0; lb < LB; ++lb) sz[lb] = (F::SCALED && PL != 0) ? plane_scale(lb == 0 ? zmax_x : zmax_x2, 0) : 1.f;

    for (int idx = tid; idx < nTB * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, p = f % NB2, rb = f / NB2, gg = ll >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = Tc[(size_t)C * C + (16 * rb + (ll & 15)) * C + 32 * p + pair_row(gg, e)];
        F::put(TB, f, ll, v, sT);
    }
    for (int idx = tid; idx < nWA * 64; idx += MF_THREADS) {
        const int ll = idx & 63, f = idx >> 6, s = f % S, lb = (f / S) % LB, n = f / (S * LB), gg = ll >> 4;
        const int l = 16 * lb + (ll & 15);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int b = 2 * s + (e >> 2), c = b / HB, hb = b % HB;
            const int wl = PL == 2 ? stc_wrow_swapped(l, Lw - 16) : l;
            v[e] = (b < NBK && wl >= 0 && wl < Lw) ? W[((size_t)(n * K + c) * Lw + wl) * Ho + 16 * hb + 4 * gg + (e & 3)] * (F::SCALED && c == 0 ? sT : 1.f) : 0.f;
        }
        F::put(WA, f, ll, v, sW);
    }
    __syncthreads();

    const float ikz = inv_pow2(uniform_bits(sT * sW));
    f32x4 dWt[K][LB][K][HB];
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

    bool skipped = false;                                   // this node ends the wave's pass: the next pass starts AT it
    int resume = -1;
    for (; node < nodes; resume = skipped ? node : resume, node = skipped ? 0x7fffffff : node + nw) {
        const size_t r0 = (size_t)node * C;
        DyFrag<NRB, HB> gr[K];
        gr[0].load(dA, node, x, g);
        gr[1].load(dB, node, x, g);
        float ikz_sg = ikz, sh = 1.f;
        if constexpr (F::SCALED) {
            float m = 0.f;
#pragma unroll
            for (int n = 0; n < K; ++n)
#pragma unroll
                for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) m = max_nonneg(m, absmax4(gr[n].v[kb][hb]));
            const float sg = rs.node(wave_max_bits(m), sh, skipped);
            ikz_sg = skipped ? 0.f : pow2_mul(ikz, inv_pow2(sg));      // (a skipped node stores zeros: it is stored again when it is redone)
#pragma unroll
            for (int n = 0; n < K; ++n)
#pragma unroll
                for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) { gr[n].d[kb][hb] *= sg; gr[n].v[kb][hb] *= sg; }
        }
        float za[LB][NRB][4];                                   // X[16kb + 4g + t][16lb + x]
#pragma unroll
        for (int lb = 0; lb < LB; ++lb) {
            const bool ok = 16 * lb + x < L;
            constexpr int LD = PL == 1 ? 16 : L;
            const int cinx = Lw - 16;
            const float* col = PL == 1 ? (lb == 0 ? X : X2) + r0 * 16 + x : X + r0 * L + 16 * lb + (ok ? x : 0);
#pragma unroll
            for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if constexpr (PL == 2) {
                        const size_t row = r0 + 16 * kb + 4 * g + t;
                        za[lb][kb][t] = lb == 0 ? X[row * 16 + x] : (x < cinx ? X2[row * cinx + x] : 0.f);
                    } else {
                        const float zv = col[(size_t)(16 * kb + 4 * g + t) * LD];
                        za[lb][kb][t] = ok ? zv : 0.f;
                    }
                }
        }
        const int lo = opaque(lane);

#pragma unroll
        for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)                       // the bias sits on A only
                dbp[hb] += sh * ((gr[0].d[kb][hb][0] + gr[0].d[kb][hb][1]) + (gr[0].d[kb][hb][2] + gr[0].d[kb][hb][3]));

        Op gd[K][HB][NB2];
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int p = 0; p < NB2; ++p) gd[n][hb][p] = F::split(gr[n].d[2 * p][hb], gr[n].d[2 * p + 1][hb]);

        // ---- B operands of dX: per slab n, step s covers the (c, o) blocks 2s, 2s+1 of (Q^n_c)^T
        Op qb[K][S][NRB];
#pragma unroll
        for (int n = 0; n < K; ++n) {
            f32x4 Qv[NRB][HB];                                   // (T_1 dY_n)^T tiles (rows o, columns c')
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) Qv[rb][hb] = kZero4;
#pragma unroll
                for (int p = 0; p < NB2; ++p) {
                    const Op t = F::get(TB, rb * NB2 + p, lo);
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) Qv[rb][hb] = F::mm(gd[n][hb][p], t, Qv[rb][hb]);
                }
            }
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    f32x4 blk[2];
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int b = 2 * s + h2, c = b / HB, hb = b % HB;
                        blk[h2] = b >= NBK ? kZero4 : (c == 0 ? gr[n].v[rb][hb] : Qv[rb][hb]);
                    }
                    qb[n][s][rb] = F::split(blk[0], blk[1]);
                }
        }

        // ---- dX^T tile (rows l, columns c'): both slabs' weights against both slabs' gradients
#pragma unroll
        for (int lb = 0; lb < LB; ++lb) {
            f32x4 z[NRB];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) z[rb] = kZero4;
#pragma unroll
            for (int n = 0; n < K; ++n)
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const Op w = F::get(WA, (n * LB + lb) * S + s, lo);
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) z[rb] = F::mm(w, qb[n][s][rb], z[rb]);
                }
            if constexpr (F::SCALED) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) z[rb] *= ikz_sg;
            }
            if constexpr (PL == 1) {                            // planar gradient: columns 0..15 -> dX, 16..31 -> dX2
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
                    stc_st_once(reinterpret_cast<f32x4*>((lb == 0 ? dX : dX2) + (r0 + 16 * rb + x) * 16 + 4 * g), z[rb]);
            } else if constexpr (PL == 2) {                     // narrow input plane: only the 16-wide plane's gradient is wanted
                if (lb == 0) {
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb)
                        stc_st_once(reinterpret_cast<f32x4*>(dX + (r0 + 16 * rb + x) * 16 + 4 * g), z[rb]);
                }
            } else if (16 * lb + 4 * g < L) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
                    *reinterpret_cast<f32x4*>(dX + (r0 + 16 * rb + x) * L + 16 * lb + 4 * g) = z[rb];
            }
        }

        // ---- Q^n_1 tiles (rows c', columns o) as operands, then dW_{n,c} += X^T Q^n_c (the X columns are split once)
        Op qd[K][HB][NB2];
#pragma unroll
        for (int n = 0; n < K; ++n) {
            f32x4 Qd[NRB][HB];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) Qd[rb][hb] = kZero4;
#pragma unroll
                for (int p = 0; p < NB2; ++p) {
                    const Op t = F::get(TB, rb * NB2 + p, lo);
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) Qd[rb][hb] = F::mm(t, gd[n][hb][p], Qd[rb][hb]);
                }
            }
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int p = 0; p < NB2; ++p) qd[n][hb][p] = F::split(Qd[2 * p][hb], Qd[2 * p + 1][hb]);
        }
#pragma unroll
        for (int lb = 0; lb < LB; ++lb)
#pragma unroll
            for (int p = 0; p < NB2; ++p) {
                const float (&zc)[NRB][4] = za[lb];
                Op a;
                if constexpr (F::SCALED) {
                    const float zs = pow2_mul(sz[lb], sh);                                       // (a skipped node: sh = RunScale::mute())
                    a = F::split(f32x4{zc[2 * p][0], zc[2 * p][1], zc[2 * p][2], zc[2 * p][3]} * zs, f32x4{zc[2 * p + 1][0], zc[2 * p + 1][1], zc[2 * p + 1][2], zc[2 * p + 1][3]} * zs);
                }
                else a = F::split(f32x4{zc[2 * p][0], zc[2 * p][1], zc[2 * p][2], zc[2 * p][3]},
                                  f32x4{zc[2 * p + 1][0], zc[2 * p + 1][1], zc[2 * p + 1][2], zc[2 * p + 1][3]});
#pragma unroll
                for (int n = 0; n < K; ++n)
#pragma unroll
                    for (int c = 0; c < K; ++c)
#pragma unroll
                        for (int hb = 0; hb < HB; ++hb)
                            dWt[n][lb][c][hb] = F::mm(a, c == 0 ? gd[n][hb][p] : qd[n][hb][p], dWt[n][lb][c][hb]);
            }
    }
    if (resume >= 0) node = resume;
    const float isg = rs.unscale();
    PlaneUnscale<K, LB> pu;
#pragma unroll
    for (int n = 0; n < K; ++n)
#pragma unroll
        for (int lb = 0; lb < LB; ++lb) pu.v[n][lb] = inv_pow2(sz[lb]);
    combine_dw<K, LB, HB>(reinterpret_cast<float*>(smem_raw), dWt, dbp, partial, Lw, want_db, PL == 2 ? Lw - 16 : -1, isg, isg / sT, isg, pu, pass > 0);
    };
    run_pass(0, args_in_kernarg_segment);
    if constexpr (F::SCALED) {
        const int n_all = args_in_kernarg_segment.nodes;
        // (the barrier also means: every wave is done with the combine's slabs before the tables are filled again)
        for (int pass = 1; __syncthreads_or(node < n_all); ++pass) run_pass(pass, kernargs_fresh<Bwd2Args>());
    }
}

// --------------------------------------------------------------------------------------- host side
template <int NB2, int HB, int K, int L, int EPI = EPI_NONE, int PL = 0, int POST = 0, class F = FmtB3>
int launch_fwd(const float* const* Z, const float* Tc, const float* W, const float* bias, float* Y,
               long long nodes, int Lw, hipStream_t stream, FwdEpi epi = FwdEpi{}, PostArgs post = PostArgs{}) {
    constexpr int NRB = 2 * NB2, NCB = K * HB;
    const size_t lds = (size_t)(K * NCB + (K - 1) * NRB * NB2 + (POST ? K * K : 0)) * F::NP * 64 * 16
                       + ((POST ? 1 : 0) + (EPI == EPI_GATES && PL != 0 ? 1 : 0)) * (size_t)MF_WAVES * NRB * 16 * 20 * 4;      // R*H tile (POST), H tile (planar gates)
    if (lds > stc::kMaxLdsBytes) return STC_NOT_HANDLED;
    auto kern = node_fwd_x3_kernel<NB2, HB, K, L, EPI, PL, POST, F>;
    if (int rc = stc::hip_status(stc::allow_lds(kern, lds), "hipFuncSetAttribute(node fwd x3)")) return rc;
    static const int resident = stc::resident_blocks(kern, MF_THREADS, lds, 2);   // persistent grid = what fits at once
    ZPtrs zp{};
    for (int n = 0; n < K; ++n) { zp.p[n] = Z[n]; if (PL) zp.q[n] = Z[K + n]; }      // planar: Z = {X planes, H planes}
    const long long want = (nodes + MF_WAVES - 1) / MF_WAVES;
    const int grid = (int)(want < resident ? want : resident);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MF_THREADS), lds, stream, zp, Tc, W, bias, Y, (int)nodes, Lw, epi, post);
    STC_LAUNCH_CHECK("node_fwd_x3 launch");
    return STC_OK;
}

template <int NB2, int HB, int K, int L, int PRO = PRO_NONE, int PL = 0, int FOLD = 0, int ACC = 0, class F = FmtB3>
int launch_bwd(const float* const* Z, const float* Tc, const float* W, const float* dY, float* const* dZ,
               float* partial, int* n_partials, int want_db, long long nodes, int Lw, hipStream_t stream, BwdPro pro = BwdPro{}) {
    constexpr int NRB = 2 * NB2, Ho = 16 * HB, LB = (L + 15) / 16, NBK = K * HB, S = (NBK + 1) / 2, nW = K * K * L * Ho;
    constexpr bool PFG = (PRO == PRO_GATES_CAND || PRO == PRO_BLEND) && PL != 0 && BwdSched<NB2, K>::waves == 1;      // as in the kernel
    const size_t frag = (size_t)((K - 1) * NRB * NB2 + K * LB * S) * F::NP * 64 * 16 + (FOLD ? (size_t)MF_WAVES * NRB * 64 * 16 : 0)
                        + (PFG ? (size_t)MF_WAVES * HB * NRB * 16 * 20 * 4 : 0);
    const size_t slabs = (size_t)MF_WAVES * (nW + Ho) * sizeof(float);
    const size_t lds = frag > slabs ? frag : slabs;
    if (lds > stc::kMaxLdsBytes) return STC_NOT_HANDLED;
    auto kern = node_bwd_x3_kernel<NB2, HB, K, L, PRO, PL, FOLD, ACC, F>;
    if (int rc = stc::hip_status(stc::allow_lds(kern, lds), "hipFuncSetAttribute(node bwd x3)")) return rc;
    static const int resident = stc::resident_blocks(kern, MF_THREADS, lds, BwdSched<NB2, K>::waves);
    ZPtrs zp{};
    DZPtrs dzp{};
    for (int n = 0; n < K; ++n) { zp.p[n] = Z[n]; dzp.p[n] = dZ[n]; if (PL) { zp.q[n] = Z[K + n]; dzp.q[n] = dZ[K + n]; } }
    const long long want = (nodes + MF_WAVES - 1) / MF_WAVES;
    int grid = resident < MF_BWD_MAX_GRID ? resident : MF_BWD_MAX_GRID;
    if (want < grid) grid = (int)want;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MF_THREADS), lds, stream, BwdArgs{zp, Tc, W, dY, dzp, partial, (int)nodes, want_db, Lw, pro});
    STC_LAUNCH_CHECK("node_bwd_x3 launch");
    *n_partials = grid;
    return STC_OK;
}

bool x3_shape(int Ks, int Kc, int C, int L, int Ho, long long nodes) {
    return Ks == Kc && Ks >= 1 && Ks <= (C == 64 ? 2 : 3) && (C == 32 || C == 64) && (Ho == 16 || Ho == 32) &&
           (L == 20 || L == 32) && nodes > 0 && nodes < (1ll << 31) / C;        // C = 64, K = 3 would spill: fp32 MFMA path
}

}  // namespace

// C = 32: Ho in {16,32} x K in {1,2,3} x L in {20,32};  C = 64: K in {1,2}
#define STC_X3_CASE12(NB2_, HB_, CALL)                                                                  \
    if (C == 32 * NB2_ && Ho == 16 * HB_) {                                                             \
        if (Ks == 1 && L == 20) return CALL(NB2_, HB_, 1, 20);                                          \
        if (Ks == 1 && L == 32) return CALL(NB2_, HB_, 1, 32);                                          \
        if (Ks == 2 && L == 20) return CALL(NB2_, HB_, 2, 20);                                          \
        if (Ks == 2 && L == 32) return CALL(NB2_, HB_, 2, 32);                                          \
    }
#define STC_X3_CASE3(NB2_, HB_, CALL)                                                                   \
    if (C == 32 * NB2_ && Ho == 16 * HB_) {                                                             \
        if (Ks == 3 && L == 20) return CALL(NB2_, HB_, 3, 20);                                          \
        if (Ks == 3 && L == 32) return CALL(NB2_, HB_, 3, 32);                                          \
    }
#define STC_X3_DISPATCH(CALL)                                                                           \
    STC_X3_CASE12(1, 1, CALL) STC_X3_CASE12(1, 2, CALL) STC_X3_CASE12(2, 1, CALL) STC_X3_CASE12(2, 2, CALL) \
    STC_X3_CASE3(1, 1, CALL) STC_X3_CASE3(1, 2, CALL)

int stc_node_fwd_x3(const float* const* Z, int Ks, const float* Tc, int Kc, const float* W, const float* bias,
                    float* Y, long long nodes, int C, int L, int Lw, int Ho, hipStream_t stream) {
    if (!x3_shape(Ks, Kc, C, L, Ho, nodes)) return STC_NOT_HANDLED;
    if (!all_aligned16(Z, Ks) || !stc::aligned16(Y)) return STC_NOT_HANDLED;
#define FWD_CALL(a, b, c, d) launch_fwd<a, b, c, d>(Z, Tc, W, bias, Y, nodes, Lw, stream)
    STC_X3_DISPATCH(FWD_CALL)
#undef FWD_CALL
    return STC_NOT_HANDLED;
}

int stc_node_bwd_x3(const float* const* Z, int Ks, const float* Tc, int Kc, const float* W, const float* dY,
                    float* const* dZ, float* partial, int* n_partials, int want_db,
                    long long nodes, int C, int L, int Lw, int Ho, hipStream_t stream) {
    if (!x3_shape(Ks, Kc, C, L, Ho, nodes)) return STC_NOT_HANDLED;
    if (!all_aligned16(Z, Ks) || !stc::aligned16(dY)) return STC_NOT_HANDLED;
    for (int n = 0; n < Ks; ++n)
        if (!stc::aligned16(dZ[n])) return STC_NOT_HANDLED;
#define BWD_CALL(a, b, c, d) launch_bwd<a, b, c, d>(Z, Tc, W, dY, dZ, partial, n_partials, want_db, nodes, Lw, stream)
    STC_X3_DISPATCH(BWD_CALL)
#undef BWD_CALL
    return STC_NOT_HANDLED;
}

// ---- fused cell epilogues / prologue: hidden width 16
#define STC_X3_EPI_CASE(NB2_, CALL)                                                                     \
    if (C == 32 * NB2_) {                                                                               \
        if (K == 1 && L == 20) return CALL(NB2_, 1, 20);                                                \
        if (K == 1 && L == 32) return CALL(NB2_, 1, 32);                                                \
        if (K == 2 && L == 20) return CALL(NB2_, 2, 20);                                                \
        if (K == 2 && L == 32) return CALL(NB2_, 2, 32);                                                \
    }
#define STC_X3_EPI_CASE3(CALL)                                                                          \
    if (C == 32) {                                                                                      \
        if (K == 3 && L == 20) return CALL(1, 3, 20);                                                   \
        if (K == 3 && L == 32) return CALL(1, 3, 32);                                                   \
    }

static bool x3_cell_shape(int K, int C, int L, long long nodes) {
    return K >= 1 && K <= (C == 64 ? 2 : 3) && (C == 32 || C == 64) && (L == 20 || L == 32) && nodes > 0 && nodes < (1ll << 31) / C;
}

int stc_cell_gates_fwd_x3(const float* const* Z, int K, const float* Tc, const float* W, const float* bias,
                          const float* H, float* U, float* R, float* CandIn,
                          long long nodes, int C, int L, int Lw, int cin, hipStream_t stream) {
    if (!x3_cell_shape(K, C, L, nodes) || !all_aligned16(Z, K)) return STC_NOT_HANDLED;
    FwdEpi epi{};
    epi.H = H; epi.U_out = U; epi.R_out = R; epi.CandIn = CandIn; epi.cin = cin;
#define GATES_CALL(a, c, d) launch_fwd<a, 2, c, d, EPI_GATES>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi)
    STC_X3_EPI_CASE(1, GATES_CALL) STC_X3_EPI_CASE(2, GATES_CALL) STC_X3_EPI_CASE3(GATES_CALL)
#undef GATES_CALL
    return STC_NOT_HANDLED;
}

int stc_cell_blend_fwd_x3(const float* const* Z, int K, const float* Tc, const float* W, const float* bias,
                          const float* U, const float* H, float* Cand, float* Hnew, const StcStateCopies* copies,
                          long long nodes, int C, int L, int Lw, hipStream_t stream) {
    if (!x3_cell_shape(K, C, L, nodes) || !all_aligned16(Z, K)) return STC_NOT_HANDLED;
    FwdEpi epi{};
    epi.H = H; epi.U = U; epi.Cand = Cand; epi.Hnew = Hnew;
    set_state_copies(epi, copies);
#define BLEND_CALL(a, c, d) launch_fwd<a, 1, c, d, EPI_BLEND>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi)
    STC_X3_EPI_CASE(1, BLEND_CALL) STC_X3_EPI_CASE(2, BLEND_CALL) STC_X3_EPI_CASE3(BLEND_CALL)
#undef BLEND_CALL
    return STC_NOT_HANDLED;
}

int stc_cell_gates_bwd_x3(const float* const* Z, int K, const float* Tc, const float* W,
                          const float* dCandIn, const float* dU, const float* H, const float* U, const float* R, const float* dH_in, const float* Cand,
                          float* const* dZ, float* dXt, float* dH, float* partial, int* n_partials, int want_db,
                          long long nodes, int C, int L, int Lw, int cin, int dh_scaled, hipStream_t stream) {
    if (!x3_cell_shape(K, C, L, nodes) || !all_aligned16(Z, K)) return STC_NOT_HANDLED;
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
    STC_X3_EPI_CASE(1, GBWD_CALL) STC_X3_EPI_CASE(2, GBWD_CALL) STC_X3_EPI_CASE3(GBWD_CALL)
#undef GBWD_CALL
    return STC_NOT_HANDLED;
}

int stc_cell_cand_bwd_x3(const float* const* Z, int K, const float* Tc, const float* W,
                         const float* dHnew, const float* U, const float* Cand,
                         float* const* dZ, float* partial, int* n_partials, int want_db,
                         long long nodes, int C, int L, int Lw, hipStream_t stream) {
    if (!x3_cell_shape(K, C, L, nodes) || !all_aligned16(Z, K)) return STC_NOT_HANDLED;
    for (int n = 0; n < K; ++n)
        if (!stc::aligned16(dZ[n])) return STC_NOT_HANDLED;
    if (!(stc::aligned16(dHnew) && stc::aligned16(U) && stc::aligned16(Cand))) return STC_NOT_HANDLED;
    BwdPro pro{};
    pro.dH_in = dHnew; pro.U = U; pro.Cand = Cand;
#define CBWD_CALL(a, c, d) launch_bwd<a, 1, c, d, PRO_BLEND>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro)
    STC_X3_EPI_CASE(1, CBWD_CALL) STC_X3_EPI_CASE(2, CBWD_CALL) STC_X3_EPI_CASE3(CBWD_CALL)
#undef CBWD_CALL
    return STC_NOT_HANDLED;
}

template <int NB2, int HB, int L, int PL = 0, class F = FmtB3>
static int launch_bwd2(const float* X, const float* X2, const float* Tc, const float* W, const float* dA, const float* dB, float* dX, float* dX2,
                       float* partial, int* n_partials, int want_db, long long nodes, int Lw, hipStream_t stream,
                       const float* zmax_x = nullptr, const float* zmax_x2 = nullptr) {
    constexpr int K = 2, NRB = 2 * NB2, Ho = 16 * HB, LB = (L + 15) / 16, NBK = K * HB, S = (NBK + 1) / 2, nW = K * K * L * Ho;
    const size_t frag = (size_t)(NRB * NB2 + K * LB * S) * F::NP * 64 * 16;
    const size_t slabs = (size_t)MF_WAVES * (nW + Ho) * sizeof(float);
    const size_t lds = frag > slabs ? frag : slabs;
    if (lds > stc::kMaxLdsBytes) return STC_NOT_HANDLED;
    auto kern = node_bwd2_x3_kernel<NB2, HB, L, PL, F>;
    if (int rc = stc::hip_status(stc::allow_lds(kern, lds), "hipFuncSetAttribute(node bwd2 x3)")) return rc;
    static const int resident = stc::resident_blocks(kern, MF_THREADS, lds, NB2 == 1 ? 2 : 1);
    const long long want = (nodes + MF_WAVES - 1) / MF_WAVES;
    int grid = resident < MF_BWD_MAX_GRID ? resident : MF_BWD_MAX_GRID;
    if (want < grid) grid = (int)want;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MF_THREADS), lds, stream, Bwd2Args{X, X2, Tc, W, dA, dB, dX, dX2, partial, (int)nodes, want_db, Lw, zmax_x, zmax_x2});
    STC_LAUNCH_CHECK("node_bwd2_x3 launch");
    *n_partials = grid;
    return STC_OK;
}

template <int NB2, int HB, int L, int PL = 0>
static int launch_fwd2(const float* X, const float* X2, const float* Tc, const float* W, const float* bias, float* A, float* Bm,
                       long long nodes, int Lw, hipStream_t stream) {
    constexpr int K = 2, NRB = 2 * NB2, NCB = K * HB;
    const size_t lds = (size_t)(K * NCB + NRB * NB2) * 3 * 64 * 16;
    auto kern = node_fwd2_x3_kernel<NB2, HB, L, PL>;
    if (int rc = stc::hip_status(stc::allow_lds(kern, lds), "hipFuncSetAttribute(node fwd2 x3)")) return rc;
    static const int resident = stc::resident_blocks(kern, MF_THREADS, lds, NB2 == 1 ? 2 : 1);
    const long long want = (nodes + MF_WAVES - 1) / MF_WAVES;
    const int grid = (int)(want < resident ? want : resident);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MF_THREADS), lds, stream, X, X2, Tc, W, bias, A, Bm, (int)nodes, Lw);
    STC_LAUNCH_CHECK("node_fwd2_x3 launch");
    return STC_OK;
}

int stc_node_post_shape_ok(int K, int C, int L, int Ho) {
    return K == 2 && (C == 32 || C == 64) && (L == 20 || L == 32) && Ho == 16;     // Ho < L: where the narrow SpMM pays
}

// fmt == STC_FMT_F16X2: fp16 x 2 operand format for the planar forms at C = 64 (what the two-launch cell backward runs; C = 32 runs the
// one-launch kernel and keeps bf16 x 3 here: its two-waves-per-SIMD build has no registers to spare); else bf16 x 3
int stc_node_post_bwd_x3(const float* X, const float* X2, const float* Tc, const float* W, const float* dA, const float* dB, float* dX, float* dX2,
                         float* partial, int* n_partials, int want_db, int fmt,
                         const float* zmax_x, const float* zmax_x2,
                         long long nodes, int C, int L, int Lw, int Ho, hipStream_t stream) {
    if (!stc_node_post_shape_ok(2, C, L, Ho) || nodes <= 0 || nodes >= (1ll << 31) / C) return STC_NOT_HANDLED;
    if (!(stc::aligned16(X) && stc::aligned16(dA) && stc::aligned16(dB) && stc::aligned16(dX) && (!X2 || L == 20 || stc::aligned16(X2)))) return STC_NOT_HANDLED;
    if (X2 && L == 20) {                        // narrow planar rows: gradient of the 16-wide plane only (dX); the input plane gets none
        if (Lw - 16 < 1 || Lw - 16 > 4) return STC_NOT_HANDLED;
        if (C == 32) return launch_bwd2<1, 1, 20, 2>(X, X2, Tc, W, dA, dB, dX, nullptr, partial, n_partials, want_db, nodes, Lw, stream);
        if (C == 64 && fmt == STC_FMT_F16X2) return launch_bwd2<2, 1, 20, 2, FmtH2>(X, X2, Tc, W, dA, dB, dX, nullptr, partial, n_partials, want_db, nodes, Lw, stream, zmax_x, zmax_x2);
        if (C == 64) return launch_bwd2<2, 1, 20, 2>(X, X2, Tc, W, dA, dB, dX, nullptr, partial, n_partials, want_db, nodes, Lw, stream);
        return STC_NOT_HANDLED;
    }
    if (X2) {                                   // planar rows (16 + 16 columns): input planes X, X2 and gradient planes dX, dX2
        if (L != 32 || !dX2 || !stc::aligned16(dX2)) return STC_NOT_HANDLED;
        if (C == 32) return launch_bwd2<1, 1, 32, 1>(X, X2, Tc, W, dA, dB, dX, dX2, partial, n_partials, want_db, nodes, Lw, stream);
        if (C == 64 && fmt == STC_FMT_F16X2) return launch_bwd2<2, 1, 32, 1, FmtH2>(X, X2, Tc, W, dA, dB, dX, dX2, partial, n_partials, want_db, nodes, Lw, stream, zmax_x, zmax_x2);
        if (C == 64) return launch_bwd2<2, 1, 32, 1>(X, X2, Tc, W, dA, dB, dX, dX2, partial, n_partials, want_db, nodes, Lw, stream);
        return STC_NOT_HANDLED;
    }
#define B2_CALL(a, b, d) launch_bwd2<a, b, d>(X, nullptr, Tc, W, dA, dB, dX, nullptr, partial, n_partials, want_db, nodes, Lw, stream)
    if (C == 32 && L == 20) return B2_CALL(1, 1, 20);
    if (C == 32 && L == 32) return B2_CALL(1, 1, 32);
    if (C == 64 && L == 20) return B2_CALL(2, 1, 20);
    if (C == 64 && L == 32) return B2_CALL(2, 1, 32);
#undef B2_CALL
    return STC_NOT_HANDLED;
}

int stc_node_post_fwd_x3(const float* X, const float* X2, const float* Tc, const float* W, const float* bias, float* A, float* Bm,
                         long long nodes, int C, int L, int Lw, int Ho, hipStream_t stream) {
    if (!stc_node_post_shape_ok(2, C, L, Ho) || nodes <= 0 || nodes >= (1ll << 31) / C) return STC_NOT_HANDLED;
    if (!(stc::aligned16(X) && stc::aligned16(A) && stc::aligned16(Bm) && (!X2 || L == 20 || stc::aligned16(X2)))) return STC_NOT_HANDLED;
    if (X2 && L == 20) {                        // narrow planar rows: X = the 16-wide plane, X2 = the input plane (Lw - 16 <= 4 columns)
        if (Lw - 16 < 1 || Lw - 16 > 4) return STC_NOT_HANDLED;
        if (C == 32) return launch_fwd2<1, 1, 20, 2>(X, X2, Tc, W, bias, A, Bm, nodes, Lw, stream);
        if (C == 64) return launch_fwd2<2, 1, 20, 2>(X, X2, Tc, W, bias, A, Bm, nodes, Lw, stream);
        return STC_NOT_HANDLED;
    }
    if (X2) {
        if (L != 32) return STC_NOT_HANDLED;
        if (C == 32) return launch_fwd2<1, 1, 32, 1>(X, X2, Tc, W, bias, A, Bm, nodes, Lw, stream);
        if (C == 64) return launch_fwd2<2, 1, 32, 1>(X, X2, Tc, W, bias, A, Bm, nodes, Lw, stream);
        return STC_NOT_HANDLED;
    }
#define F2_CALL(a, d) launch_fwd2<a, 1, d>(X, nullptr, Tc, W, bias, A, Bm, nodes, Lw, stream)
    if (C == 32 && L == 20) return F2_CALL(1, 20);
    if (C == 32 && L == 32) return F2_CALL(1, 32);
    if (C == 64 && L == 20) return F2_CALL(2, 20);
    if (C == 64 && L == 32) return F2_CALL(2, 32);
#undef F2_CALL
    return STC_NOT_HANDLED;
}

// ---- planar cell inputs (K = 2, rows of 16 + 16 columns): Z = {X plane, S.X plane, H plane, S.H plane} in launch order {p[0], p[1], q[0], q[1]}
template <class F>
static int gates_fwd_planar_go(const float* X, const float* H, const float* SX, const float* SH, const float* Tc, const float* W, const float* bias,
                               const FwdEpi& epi, const PostArgs& post, bool fused, int cin, long long nodes, int C, int Lw, hipStream_t stream) {
    if (cin == 16) {
        const float* Z[4] = {X, SX, H, SH};
        if (!all_aligned16(Z, 4)) return STC_NOT_HANDLED;
        if (C == 32 && fused) return launch_fwd<1, 2, 2, 32, EPI_GATES, 1, 1, F>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi, post);
        if (C == 32) return launch_fwd<1, 2, 2, 32, EPI_GATES, 1, 0, F>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi);
        if (C == 64 && fused) return launch_fwd<2, 2, 2, 32, EPI_GATES, 1, 1, F>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi, post);
        if (C == 64) return launch_fwd<2, 2, 2, 32, EPI_GATES, 1, 0, F>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi);
        return STC_NOT_HANDLED;
    }
    const float* Z[4] = {H, SH, X, SX};          // narrow input: the STATE plane leads, columns [H | Xt | pad]
    if (!stc::aligned16(H) || !stc::aligned16(SH)) return STC_NOT_HANDLED;
    if (C == 32 && fused) return launch_fwd<1, 2, 2, 20, EPI_GATES, 2, 1, F>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi, post);
    if (C == 32) return launch_fwd<1, 2, 2, 20, EPI_GATES, 2, 0, F>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi);
    if (C == 64 && fused) return launch_fwd<2, 2, 2, 20, EPI_GATES, 2, 1, F>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi, post);
    if (C == 64) return launch_fwd<2, 2, 2, 20, EPI_GATES, 2, 0, F>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi);
    return STC_NOT_HANDLED;
}

int stc_cell_gates_fwd_planar_x3(const float* X, const float* H, const float* SX, const float* SH, const float* Tc, const float* W,
                                 const float* bias, float* U, float* R, float* RH,
                                 const float* Wc, const float* bc, float* A, float* Bm, int fmt, float* zmax,
                                 long long nodes, int C, int Lw, hipStream_t stream) {
    const int cin = Lw - 16;
    const PostArgs post{Wc, bc, A, Bm};
    const bool fused = A != nullptr;             // also the candidate's projection (A, Bm) in this launch
    if (fused && !(Wc && Bm && stc::aligned16(A) && stc::aligned16(Bm))) return STC_NOT_HANDLED;
    if (!x3_cell_shape(2, C, cin == 16 ? 32 : 20, nodes) || !(cin == 16 || (cin >= 1 && cin <= 4))) return STC_NOT_HANDLED;
    FwdEpi epi{};
    epi.H = H; epi.U_out = U; epi.R_out = R; epi.CandIn = RH; epi.cin = cin; epi.zmax = zmax;
    return fmt == STC_FMT_F16X2 ? gates_fwd_planar_go<FmtH2>(X, H, SX, SH, Tc, W, bias, epi, post, fused, cin, nodes, C, Lw, stream)
                                : gates_fwd_planar_go<FmtB3>(X, H, SX, SH, Tc, W, bias, epi, post, fused, cin, nodes, C, Lw, stream);
}

template <class F>
static int gates_bwd_planar_go(const float* const* Z, float* const* dZ, const float* Tc, const float* W, bool narrow, bool fold, const BwdPro& pro,
                               float* partial, int* n_partials, int want_db, long long nodes, int C, int Lw, hipStream_t stream) {
    if (narrow) {
        // (C = 32, narrow: the fp16 x 2 build of this instantiation spills 80-96 B per lane at its 256-register cap for two waves per SIMD, and
        // it is not a launch the default path takes -- C = 32 runs the one-launch cell backward -- so it stays on the bf16 x 3 format)
        if (C == 32 && fold) return launch_bwd<1, 2, 2, 20, PRO_GATES_CAND, 2, 1, 0, FmtB3>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro);
        if (C == 64 && fold) return launch_bwd<2, 2, 2, 20, PRO_GATES_CAND, 2, 1, 0, F>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro);
        if (C == 32) return launch_bwd<1, 2, 2, 20, PRO_GATES_CAND, 2, 0, 0, FmtB3>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro);
        if (C == 64) return launch_bwd<2, 2, 2, 20, PRO_GATES_CAND, 2, 0, 0, F>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro);
        return STC_NOT_HANDLED;
    }
    // fold: the state's share from the gate prologue is folded into the H plane's gradient dZ[2] (FOLD)
    if (C == 32 && fold) return launch_bwd<1, 2, 2, 32, PRO_GATES_CAND, 1, 1, 0, F>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro);
    if (C == 64 && fold) return launch_bwd<2, 2, 2, 32, PRO_GATES_CAND, 1, 1, 0, F>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro);
    if (C == 32) return launch_bwd<1, 2, 2, 32, PRO_GATES_CAND, 1, 0, 0, F>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro);
    if (C == 64) return launch_bwd<2, 2, 2, 32, PRO_GATES_CAND, 1, 0, 0, F>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro);
    return STC_NOT_HANDLED;
}

// fmt: STC_FMT_F16X2 or STC_FMT_BF16X3 (stc_x3_frag.h)
int stc_cell_gates_bwd_planar_x3(const float* X, const float* H, const float* SX, const float* SH, const float* Tc, const float* W,
                                 const float* dCandIn, const float* Cand, const float* U, const float* R, const float* dHnew,
                                 float* const* dZ, float* dH, float* partial, int* n_partials, int want_db, int fmt, const float* zmax,
                                 long long nodes, int C, int Lw, hipStream_t stream) {
    const int cin = Lw - 16;
    if (!x3_cell_shape(2, C, cin == 16 ? 32 : 20, nodes) || !(cin == 16 || (cin >= 1 && cin <= 4))) return STC_NOT_HANDLED;
    if (!(stc::aligned16(dCandIn) && stc::aligned16(Cand) && stc::aligned16(U) && stc::aligned16(R) && stc::aligned16(dHnew) && (!dH || stc::aligned16(dH))))
        return STC_NOT_HANDLED;
    BwdPro pro{};       // dCandIn: the gradient of the R*H plane, (nodes, C, 16): the state columns sit at offset 0
    pro.Cand = Cand; pro.dCandIn = dCandIn; pro.H = H; pro.U = U; pro.R = R; pro.dH_in = dHnew; pro.dH = dH; pro.cin = 0; pro.dh_scaled = 1;
    pro.zmax = zmax;
    const float* Zw[4] = {X, SX, H, SH};
    const float* Zn[4] = {H, SH, X, SX};           // narrow input plane: the state plane leads; only d H plane (dZ[2]) and d SH plane (dZ[3]) are produced
    float* dZn[4] = {dZ[2], dZ[3], nullptr, nullptr};
    const bool narrow = cin != 16;
    if (narrow) {
        if (!(stc::aligned16(H) && stc::aligned16(SH) && dZ[2] && dZ[3] && stc::aligned16(dZ[2]) && stc::aligned16(dZ[3]))) return STC_NOT_HANDLED;
    } else {
        if (!all_aligned16(Zw, 4)) return STC_NOT_HANDLED;
        for (int i = 0; i < 4; ++i)
            if (!dZ[i] || !stc::aligned16(dZ[i])) return STC_NOT_HANDLED;
    }
    return fmt == STC_FMT_F16X2 ? gates_bwd_planar_go<FmtH2>(narrow ? Zn : Zw, narrow ? dZn : dZ, Tc, W, narrow, dH == nullptr, pro, partial, n_partials, want_db, nodes, C, Lw, stream)
                : gates_bwd_planar_go<FmtB3>(narrow ? Zn : Zw, narrow ? dZn : dZ, Tc, W, narrow, dH == nullptr, pro, partial, n_partials, want_db, nodes, C, Lw, stream);
}

// ---- planar cell convolutions of Chebyshev order K = 3 (C = 32, hidden 16).  Zx[n] / Zh[n] = T_n(S) applied to the plane on the
// X side / the H side of the [X | H] row (n = 0: the plane itself), each (nodes, C, 16); a narrow input (layer 0: Lw - 16 = 1..4
// columns) has Zx[n] (nodes, C, Lw - 16) and the 16-wide side leads inside the kernels, as in the K = 2 planar launches.
// mode 1 = gates convolution with the sigmoid / R*H epilogue (prologue: gate + blend backward), mode 2 = candidate convolution
// on [X | R*H] with the tanh + GRU-blend epilogue (prologue: blend backward).  Same templates as everything above.
int stc_cell_planar_k_shape_ok(int K, int C, int h) { return K == 3 && C == 32 && h == 16; }

int stc_cell_conv_fwd_planar_k_x3(const float* const* Zx, const float* const* Zh, int K, const float* Tc, const float* W, const float* bias, int mode,
                                  const float* H, const float* Uin, float* U, float* R, float* RH, float* Cand, float* Hnew, int fmt, float* zmax,
                                  long long nodes, int C, int Lw, hipStream_t stream) {
    const int cin = Lw - 16;
    if (!stc_cell_planar_k_shape_ok(K, C, 16) || !x3_cell_shape(K, C, cin == 16 ? 32 : 20, nodes) || !(cin == 16 || (cin >= 1 && cin <= 4))) return STC_NOT_HANDLED;
    if (!all_aligned16(Zh, K) || (cin == 16 && !all_aligned16(Zx, K))) return STC_NOT_HANDLED;
    FwdEpi epi{};
    if (mode == 1) { epi.H = H; epi.U_out = U; epi.R_out = R; epi.CandIn = RH; epi.cin = cin; }
    else { epi.H = H; epi.U = Uin; epi.Cand = Cand; epi.Hnew = Hnew; }
    epi.zmax = zmax;
    const float* Z[6];
    for (int n = 0; n < 3; ++n) { Z[n] = cin == 16 ? Zx[n] : Zh[n]; Z[3 + n] = cin == 16 ? Zh[n] : Zx[n]; }
    if (fmt == STC_FMT_F16X2) {
        if (mode == 1) return cin == 16 ? launch_fwd<1, 2, 3, 32, EPI_GATES, 1, 0, FmtH2>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi)
                                        : launch_fwd<1, 2, 3, 20, EPI_GATES, 2, 0, FmtH2>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi);
        return cin == 16 ? launch_fwd<1, 1, 3, 32, EPI_BLEND, 1, 0, FmtH2>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi)
                         : launch_fwd<1, 1, 3, 20, EPI_BLEND, 2, 0, FmtH2>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi);
    }
    if (mode == 1) return cin == 16 ? launch_fwd<1, 2, 3, 32, EPI_GATES, 1>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi)
                                    : launch_fwd<1, 2, 3, 20, EPI_GATES, 2>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi);
    return cin == 16 ? launch_fwd<1, 1, 3, 32, EPI_BLEND, 1>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi)
                     : launch_fwd<1, 1, 3, 20, EPI_BLEND, 2>(Z, Tc, W, bias, nullptr, nodes, Lw, stream, epi);
}

template <class F>
static int conv_bwd_planar_k_go(const float* const* Z, float* const* dZ, const float* Tc, const float* W, int mode, int cin, bool fold, int accumulate_x,
                                const BwdPro& pro, float* partial, int* n_partials, int want_db, long long nodes, int Lw, hipStream_t stream) {
    if (mode == 1) {
        if (accumulate_x) return launch_bwd<1, 2, 3, 32, PRO_GATES_CAND, 1, 1, 1, F>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro);
        if (fold)       // the state's share from the gate prologue is folded into dZh[0]
            return cin == 16 ? launch_bwd<1, 2, 3, 32, PRO_GATES_CAND, 1, 1, 0, F>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro)
                             : launch_bwd<1, 2, 3, 20, PRO_GATES_CAND, 2, 1, 0, F>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro);
        return cin == 16 ? launch_bwd<1, 2, 3, 32, PRO_GATES_CAND, 1, 0, 0, F>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro)
                         : launch_bwd<1, 2, 3, 20, PRO_GATES_CAND, 2, 0, 0, F>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro);
    }
    return cin == 16 ? launch_bwd<1, 1, 3, 32, PRO_BLEND, 1, 0, 0, F>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro)
                     : launch_bwd<1, 1, 3, 20, PRO_BLEND, 2, 0, 0, F>(Z, Tc, W, nullptr, dZ, partial, n_partials, want_db, nodes, Lw, stream, pro);
}

// fmt: STC_FMT_F16X2 or STC_FMT_BF16X3 (stc_x3_frag.h)
int stc_cell_conv_bwd_planar_k_x3(const float* const* Zx, const float* const* Zh, int K, const float* Tc, const float* W, int mode,
                                  const float* dRH, const float* Cand, const float* U, const float* R, const float* dHnew,
                                  float* const* dZx, float* const* dZh, float* dH, float* partial, int* n_partials, int want_db,
                                  long long nodes, int C, int Lw, int accumulate_x, int fmt, const float* zmax, hipStream_t stream) {
    const int cin = Lw - 16;
    if (!stc_cell_planar_k_shape_ok(K, C, 16) || !x3_cell_shape(K, C, cin == 16 ? 32 : 20, nodes) || !(cin == 16 || (cin >= 1 && cin <= 4))) return STC_NOT_HANDLED;
    if (!all_aligned16(Zh, K) || (cin == 16 && !all_aligned16(Zx, K))) return STC_NOT_HANDLED;
    const float* Z[6];
    float* dZ[6];
    for (int n = 0; n < 3; ++n) {
        if (!dZh[n] || !stc::aligned16(dZh[n]) || (cin == 16 && (!dZx[n] || !stc::aligned16(dZx[n])))) return STC_NOT_HANDLED;
        Z[n] = cin == 16 ? Zx[n] : Zh[n]; Z[3 + n] = cin == 16 ? Zh[n] : Zx[n];
        dZ[n] = cin == 16 ? dZx[n] : dZh[n]; dZ[3 + n] = cin == 16 ? dZh[n] : nullptr;      // a narrow input plane gets no gradient
    }
    if (!(stc::aligned16(Cand) && stc::aligned16(U) && stc::aligned16(dHnew))) return STC_NOT_HANDLED;
    BwdPro pro{};
    pro.zmax = zmax;
    if (mode == 1) {
        if (!(stc::aligned16(dRH) && stc::aligned16(R) && (!dH || stc::aligned16(dH)))) return STC_NOT_HANDLED;
        pro.Cand = Cand; pro.dCandIn = dRH; pro.H = Zh[0]; pro.U = U; pro.R = R; pro.dH_in = dHnew; pro.dH = dH; pro.cin = 0; pro.dh_scaled = 1;
        if (accumulate_x && (dH || cin != 16)) return STC_NOT_HANDLED;      // wide, folded form only: what the cell graph runs
    } else {
        pro.dH_in = dHnew; pro.U = U; pro.Cand = Cand;
    }
    return fmt == STC_FMT_F16X2 ? conv_bwd_planar_k_go<FmtH2>(Z, dZ, Tc, W, mode, cin, dH == nullptr, accumulate_x, pro, partial, n_partials, want_db, nodes, Lw, stream)
                : conv_bwd_planar_k_go<FmtB3>(Z, dZ, Tc, W, mode, cin, dH == nullptr, accumulate_x, pro, partial, n_partials, want_db, nodes, Lw, stream);
}
