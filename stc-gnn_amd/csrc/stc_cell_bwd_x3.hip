// The whole backward of one planar STC_Cell step (reference STC_GNN.py:65-79 through autograd; K = 2, C = 32, hidden 16) in ONE
// launch on the split-operand matrix cores of gfx950: the candidate convolution's backward in post-aggregation form
// (node_bwd2_x3_kernel) and the gates convolution's backward with the gate / blend backward as its prologue
// (node_bwd_x3_kernel<.., PRO_GATES_CAND, PL, FOLD>) run back to back on the same node while its operands are with the wave.
//
// What the fusion removes from HBM per cell step and sample (planes of nodes x C x 16 floats):
//   * dY = dHnew * U * (1 - Cand^2) is formed here from three planes the gate prologue reads anyway (the candidate kernel read it);
//   * the R*H plane's gradient goes from the candidate's dX tile to the gate prologue through a per-wave LDS tile -- never written;
//   * the candidate's X-side gradient is parked in a lane-private LDS slot and the gates' X tile STARTS from it: one plane instead of
//     two for the source's gradient sum;
//   * X and R*H are not read a second time (R*H = R.H is formed from planes the prologue holds, so the forward need not store it).
// 13 planes (9 in: X, H, S.X, S.H, U, R, Cand, dHnew, dBm; 4 out: dX, dS.X, dH, dS.H) instead of 19, and one addend less in the
// state-gradient sum that follows.
//
// Registers: both convolutions' dW tiles persist across the nodes of a wave (64 + 32 registers) and the gates phase alone fills the 256
// registers of two waves per SIMD; at that budget the fused body spilled 72 registers to scratch (scratch stores go through to HBM:
// 4.6 GB per launch, more than the fusion saves).  The kernel therefore runs ONE wave per SIMD with the full 512-register file and hides
// HBM latency itself: the next node's 120 operand registers are requested before the current node is computed (software prefetch).
// Measured alternatives (MI355X, 250 880 nodes; HISTORY.md section 3c): splitting the node's work between a candidate wave and a gates wave
// per SIMD with an LDS hand-over (two waves per SIMD, 234 registers each) runs at the same 1.98 ms per launch; forcing MFMA / VALU
// interleaving with sched_group_barrier pipelines ended in scratch spills (2.8 ms).
#include "stc_x3_frag.h"

namespace {

constexpr int CB_WAVES = 4, CB_THREADS = CB_WAVES * 64;
constexpr int CB_TRS = 20;                      // row stride of the transpose tile (16 + 4: conflict-free column reads)

struct CellBwdArgs {
    const float *X, *H, *SX, *SH;               // PL = 1: (nodes, C, 16) planes; PL = 2: X, SX are the narrow input planes (nodes, C, cin)
    const float *U, *R, *Cand, *dHnew, *dBm;    // (nodes, C, 16)
    const float *Tc, *Wg, *Wc;                  // (2, C, C); (4 Lw, 32); (4 Lw, 16)
    float *dX, *dSX, *dH, *dSH;                 // gradient planes (dX, dSX: PL = 1 only)
    float *partial_g, *partial_c;               // one row [dW | db] per workgroup for each convolution
    int nodes, want_dbg, want_dbc, Lw;
    const float* zmax;                          // fp16 x 2 format, optional: (4, 256) slots of max |plane| as the gates forward left them (launch order of its
                                                //   planes: PL = 1 {X, S.X, H, S.H}, PL = 2 {H, S.H, x, S.x}): scales of the dW products' activation operands
};

// per-wave LDS block: [stash_h 2 x 64 float4][stash_x 2 x 64 float4][CB_TILES transposition tiles of 32 x CB_TRS floats]
constexpr int CB_TILES = 5;                     // dY, dBm, R*H (candidate phase), the two halves of the gates' dY
constexpr int CB_TILE_FLOATS = 32 * CB_TRS;
constexpr int CB_WAVE_BYTES = 2 * 64 * 16 + 2 * 64 * 16 + CB_TILES * CB_TILE_FLOATS * 4;
constexpr int CB_TABLE_FRAGS = 2 + 8 + 4;       // T_1 (2), gates W (K LB S = 8), candidate W (K LB = 4)
template <class F>
constexpr size_t cb_lds_bytes() { return (size_t)CB_TABLE_FRAGS * F::NP * 64 * 16 + (size_t)CB_WAVES * CB_WAVE_BYTES; }

// ACCX / ACCH: the X-side (dX, dS.X) / H-side (dH, dS.H) gradient planes already hold another cell's gradients for the same state -- its
// other consumer's -- and this launch ADDS its own: the state then owns ONE direct and ONE aggregated plane, and the state-gradient SpMM
// that follows gathers one operand instead of two (8 -> 6 planes there, 2 more streamed reads here, where HBM is not the bound).
// F: operand format of the matrix-core products (stc_x3_frag.h): FmtB3 = three bf16 pieces / six products, FmtH2 = two fp16 pieces / three
// products with every operand class scaled by powers of two (stc_x3_frag.h): gradient fragments per node (a_n, from the node's own maximum;
// the dW / db sums over nodes at the wave's reference scale), activation operands of the dW products per plane (a.zmax), and the tables
// normalised per workgroup: block c = 0 of W carries sW sT, blocks c >= 1 carry sW and T_1 carries sT, so both halves of a contraction over
// (c, o) arrive with the same factor a_n sT sW, which the tile's store takes out again.
template <class F, int L, int PL, int ACCX = 0, int ACCH = 0>        // PL = 1: L = 32, rows [X | H];  PL = 2: L = 20, rows [H | x (cin = Lw - 16 <= 4) | pad], W rows permuted to match
__global__ __launch_bounds__(CB_THREADS, 1) void cell_bwd_x3_kernel(CellBwdArgs args_in_kernarg_segment) {
    using Op = typename F::Op;
    constexpr int NP = F::NP;
    constexpr int K = 2, NRB = 2, C = 32, LB = 2, HID = 16;
    constexpr int RHB = PL == 1 ? 1 : 0;          // block of the row that is the state plane (H for the gates, R*H for the candidate)
    constexpr int LBD = PL == 1 ? 2 : 1;          // blocks of the row whose gradient is wanted (a narrow input plane gets none)
    static_assert((PL == 1 && L == 32) || (PL == 2 && L == 20), "planar rows are 16 + 16 or 16 + cin columns");
    static_assert(PL == 1 || !ACCX, "a narrow input plane gets no gradient");
    constexpr bool AHEAD = !(ACCX && ACCH);       // W fragments fetched one product group ahead (12 registers; the both-sides variant has none to spare)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* TB = reinterpret_cast<u32x4*>(smem_raw);     // [NRB rb]          T_1[16rb + x][pair_row]
    u32x4* WG = TB + 2 * NP * 64;                        // [K n][LB][S = 2]  Wg[(n, c, 16lb + x)][16hb + 4g + (e&3)], block 2s + (e>>2) = (c, hb)
    u32x4* WC = WG + 8 * NP * 64;                        // [K n][LB]         Wc[(n, c, 16lb + x)][4g + (e&3)], block (e>>2) = c
    const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* mine = reinterpret_cast<unsigned char*>(WC + 4 * NP * 64) + (size_t)wave * CB_WAVE_BYTES;
    float4* stash_h = reinterpret_cast<float4*>(mine);               // the state's share of the gate prologue, lane-private
    float4* stash_x = stash_h + 2 * 64;                               // the candidate's X-side gradient, lane-private
    float* tiles = reinterpret_cast<float*>(stash_x + 2 * 64);        // row-on-lane -> accumulator layout (to_acc below)
    const int nw = gridDim.x * CB_WAVES;
    // fp16 x 2: the kernel body runs in PASSES.  A node whose gradients exceed the sums' scale by more than 2^12 -- the candidate's or the gates' -- ends its wave's pass (stc_x3_frag.h:
    // RunScale): computed to its end with its sums muted and nothing new stored, then the wave leaves the loop at the latch.  After the
    // combine the body runs again while any wave of the workgroup has nodes left -- tables refilled, operands requested afresh, sums empty,
    // the partial row added to; nothing is live across passes but the node index and whether that node's candidate phase is already in the rows.
    int node = blockIdx.x * CB_WAVES + wave;
    bool cand_done = false;           // the pass starts at a node whose candidate phase is in the partial rows already (its gates phase ended the previous pass)
    // The body is written once and compiled TWICE: the first pass as straight-line code (with the loop over passes around it the node loop of
    // every launch ran 1.9 % slower -- 1 214 against 1 191 us, same box -- through worse register allocation, with the arguments kept live
    // across it 4 %), later passes -- rare -- in a loop, their arguments read afresh from the kernarg segment (stc_x3_frag.h).
    auto run_pass = [&](const int pass, const CellBwdArgs a) __attribute__((always_inline)) {
    const int cin = a.Lw - 16;
    // FmtH2: table scales from the tables' own maxima (same in every workgroup), gradient scale from the producer's slots
    float sT = 1.f, sWg = 1.f, sWc = 1.f;
    if constexpr (F::SCALED) {
        float* scratch = reinterpret_cast<float*>(smem_raw);
        sT = clamp_mix_scale(pow2_scale(block_absmax(a.Tc + (size_t)C * C, C * C, scratch, CB_THREADS), STC_T_TARGET_BWD));      // (W's block 0 carries sT as well)
        sWg = pow2_scale(block_absmax(a.Wg, 4 * a.Lw * 32, scratch, CB_THREADS), STC_W_TARGET);
        sWc = pow2_scale(block_absmax(a.Wc, 4 * a.Lw * 16, scratch, CB_THREADS), STC_W_TARGET);
    }
    RunScale rc, rg;                                     // gradient scales of the candidate / the gates phase (stc_x3_frag.h: per node + the wave's reference)
    // activation operands of the dW products (sums over nodes: one scale per plane and launch); block lb of slab n is row lb K + n of the slots.
    // R*H takes the H plane's scale (|R*H| <= |H|): block RHB of slab 0, where H itself sits.
    float sz[K][LB];
#pragma unroll
    for (int n = 0; n < K; ++n)
#pragma unroll
        for (int lb = 0; lb < LB; ++lb) sz[n][lb] = F::SCALED ? plane_scale(a.zmax, lb * K + n) : 1.f;

    for (int idx = tid; idx < 2 * 64; idx += CB_THREADS) {
        const int ll = idx & 63, rb = idx >> 6, gg = ll >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = a.Tc[(size_t)C * C + (16 * rb + (ll & 15)) * C + pair_row(gg, e)];
        F::put(TB, rb, ll, v, sT);
    }
    for (int idx = tid; idx < 8 * 64; idx += CB_THREADS) {
        const int ll = idx & 63, f = idx >> 6, s = f % 2, lb = (f / 2) % LB, n = f / (2 * LB), gg = ll >> 4;
        const int l = 16 * lb + (ll & 15);
        const int wl = PL == 2 ? stc_wrow_swapped(l, cin) : l;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int b = 2 * s + (e >> 2), c = b / 2, hb = b % 2;
            v[e] = (wl >= 0 && wl < a.Lw) ? a.Wg[((size_t)(n * K + c) * a.Lw + wl) * 32 + 16 * hb + 4 * gg + (e & 3)] : 0.f;
        }
        F::put(WG, f, ll, v, s == 0 ? sWg * sT : sWg);              // step s holds the blocks of c = s
    }
    for (int idx = tid; idx < 4 * 64; idx += CB_THREADS) {
        const int ll = idx & 63, f = idx >> 6, lb = f % LB, n = f / LB, gg = ll >> 4;
        const int l = 16 * lb + (ll & 15);
        const int wl = PL == 2 ? stc_wrow_swapped(l, cin) : l;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = e >> 2;
            v[e] = (wl >= 0 && wl < a.Lw) ? a.Wc[((size_t)(n * K + c) * a.Lw + wl) * 16 + 4 * gg + (e & 3)] * (F::SCALED && c == 0 ? sT : 1.f) : 0.f;
        }
        F::put(WC, f, ll, v, sWc);
    }
    __syncthreads();

    f32x4 dWg[K][LB][K][2];            // gates dW tiles: rows l = 16lb + 4g + r, columns o = 16hb + x
    float dbg[2] = {0.f, 0.f}, dbc[1] = {0.f};
#pragma unroll
    for (int n = 0; n < K; ++n)
#pragma unroll
        for (int lb = 0; lb < LB; ++lb)
#pragma unroll
            for (int c = 0; c < K; ++c)
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) dWg[n][lb][c][hb] = kZero4;
    f32x4 dWc[K][LB][K][1];            // candidate dW tiles: rows l, columns o = x
#pragma unroll
    for (int n = 0; n < K; ++n)
#pragma unroll
        for (int lb = 0; lb < LB; ++lb)
#pragma unroll
            for (int c = 0; c < K; ++c) dWc[n][lb][c][0] = kZero4;

    const float* P0[K] = {PL == 1 ? a.X : a.H, PL == 1 ? a.SX : a.SH};          // block 0 of slab n
    const float* P1[K] = {PL == 1 ? a.H : a.X, PL == 1 ? a.SH : a.SX};          // block 1 of slab n (PL = 2: the narrow plane)

    // One node's operands: the slab columns in accumulator layout (row 16kb + 4g + t, column x: the A operands of the dW products) and the
    // gate planes in row-on-lane layout (row 16kb + x, columns 4g .. 4g+3).  Everything element-wise is formed ONCE, in row layout, and
    // taken to the accumulator layout through per-wave LDS tiles (to_acc): the kernel used to load the five gate planes a second time in
    // accumulator layout (40 four-byte loads per node, half of the node's load instructions) and to repeat the gate / blend arithmetic there.
    struct Ops {
        float zg[K][LB][NRB][4];
        f32x4 uv[NRB], rv[NRB], cv[NRB], gv[NRB], hv[NRB], bv[NRB];
    };
    // rows 16kb + x of a pair of 16-row blocks, columns 4g .. 4g+3  ->  rows 16kb + 4g + t, column x.  Same wave writes and reads (LDS
    // operations of a wave complete in order); stride CB_TRS: conflict-free both ways.
    auto to_acc = [&](int which, const f32x4 (&v)[NRB], f32x4 (&d)[NRB]) {
        float* t = tiles + which * CB_TILE_FLOATS;
#pragma unroll
        for (int kb = 0; kb < NRB; ++kb) *reinterpret_cast<f32x4*>(t + (16 * kb + x) * CB_TRS + 4 * g) = v[kb];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) d[kb][i] = t[(16 * kb + 4 * g + i) * CB_TRS + x];
    };
    float* const dP[K][LB] = {{PL == 1 ? a.dX : a.dH, PL == 1 ? a.dH : nullptr}, {PL == 1 ? a.dSX : a.dSH, PL == 1 ? a.dSH : nullptr}};
    constexpr bool ACC[LB] = {PL == 1 ? ACCX != 0 : ACCH != 0, PL == 1 ? ACCH != 0 : false};      // per block of the row
    auto load_ops = [&](Ops& o, int nd) {
        const size_t r0 = (size_t)nd * C;
#pragma unroll
        for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const size_t row = r0 + 16 * kb + 4 * g + t, e = row * HID + x;
#pragma unroll
                for (int n = 0; n < K; ++n) {
                    o.zg[n][0][kb][t] = P0[n][e];
                    if constexpr (PL == 1) o.zg[n][1][kb][t] = P1[n][e];
                    else o.zg[n][1][kb][t] = x < cin ? P1[n][row * cin + x] : 0.f;
                }
            }
#pragma unroll
        for (int kb = 0; kb < NRB; ++kb) {
            const size_t e = (r0 + 16 * kb + x) * HID + 4 * g;
            o.uv[kb] = stc_ld_once(reinterpret_cast<const f32x4*>(a.U + e));
            o.rv[kb] = stc_ld_once(reinterpret_cast<const f32x4*>(a.R + e));
            o.cv[kb] = stc_ld_once(reinterpret_cast<const f32x4*>(a.Cand + e));
            o.gv[kb] = stc_ld_once(reinterpret_cast<const f32x4*>(a.dHnew + e));
            o.hv[kb] = stc_ld_once(reinterpret_cast<const f32x4*>(a.H + e));
            o.bv[kb] = stc_ld_once(reinterpret_cast<const f32x4*>(a.dBm + e));
        }
    };
    // the two T_1 fragments are used four times per node (both orientations, both convolutions): kept in registers for the whole kernel
    const Op tb0 = F::get(TB, 0, lane), tb1 = F::get(TB, 1, lane);
    // FmtH2 factors (all powers of two; 1 for FmtB3): what a tile of each phase carries besides sg, and their inverses
    const float kc = uniform_bits(sT * sWc), kg = uniform_bits(sT * sWg), ikc = inv_pow2(kc), ikg = inv_pow2(kg);      // (scalar registers)
    Ops cur, nxt, nx2;                            // operands two nodes ahead: ~40 MB in flight chip-wide instead of 20
    if (node < a.nodes) load_ops(cur, node);
    if (node + nw < a.nodes) load_ops(nxt, node + nw);
    // The first node's loads are waited for HERE: left pending into the loop, the compiler's wait-count pass merges them with the loop's own
    // state and makes every iteration wait for the loads it has just issued for the NEXT node (vmcnt counts in order) -- no prefetch at all.
    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0), expcnt / lgkmcnt untouched
    int resume = -1;
    while (node < a.nodes) {
        const bool mute_c = cand_done;                             // this node's candidate sums are in the partial rows already (first node of a pass only)
        bool skip_c = false, skip_g = false;                       // the node is too large for the candidate / the gates sums: it ends the pass and is redone
        const int next_node = node + nw;
        // ACCX / ACCH: what the gradient planes already hold (tile layout: row 16rb + x, columns 4g .. 4g+3).  Requested for THIS node, before
        // the next node's operands (vmcnt counts in order); first used in the gates phase, microseconds from here.
        f32x4 old[K][LB][NRB];
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int lb = 0; lb < LBD; ++lb)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
                    if (ACC[lb]) old[n][lb][rb] = stc_ld_once(reinterpret_cast<const f32x4*>(dP[n][lb] + ((size_t)node * C + 16 * rb + x) * HID + 4 * g));
        if (next_node + nw < a.nodes) load_ops(nx2, next_node + nw);      // software prefetch, two nodes ahead
        __builtin_amdgcn_sched_barrier(0);
        const size_t r0 = (size_t)node * C;
        const auto& zg = cur.zg;
        const auto &uv = cur.uv, &rv = cur.rv, &cv = cur.cv, &gv = cur.gv, &hv = cur.hv, &bv = cur.bv;
        const int lo = opaque(lane);

        // =========================================================== candidate convolution (post-aggregation form): dA = dY, dBm given
        f32x4 drh[NRB];                                            // gradient of the R*H plane, row-on-lane layout
        float a_c = 1.f, sh_c = 1.f;                               // FmtH2: this node's gradient scale in the candidate phase, and a / a_n for the sums over nodes
        {
            DyFrag<NRB, 1> gr[K];
            f32x4 rh_v[NRB], rh_d[NRB];                             // R*H, the candidate's second input plane (re-formed, not stored by the forward)
            {
                f32x4 v0[NRB], v1[NRB], d0[NRB], d1[NRB];
#pragma unroll
                for (int kb = 0; kb < NRB; ++kb) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v0[kb][i] = gv[kb][i] * uv[kb][i] * (1.f - cv[kb][i] * cv[kb][i]);      // dY = dHnew * U * (1 - Cand^2)
                        rh_v[kb][i] = rv[kb][i] * hv[kb][i];
                    }
                    v1[kb] = bv[kb];
                }
                if constexpr (F::SCALED) {                         // into the node's own scale: everything below is linear in (dY, dBm)
                    float m = 0.f;
#pragma unroll
                    for (int kb = 0; kb < NRB; ++kb) m = vmax3_acc(m, absmax4(v0[kb]), absmax4(v1[kb]));
                    const int k_was = rc.k;
                    a_c = rc.node(wave_max_bits(m), sh_c, skip_c);
                    if (mute_c) { rc.k = k_was; sh_c = RunScale::mute(); }               // (a muted node adds nothing: it sets no reference either)
#pragma unroll
                    for (int kb = 0; kb < NRB; ++kb) { v0[kb] *= a_c; v1[kb] *= a_c; }
                }
                to_acc(0, v0, d0);
                to_acc(1, v1, d1);
                to_acc(2, rh_v, rh_d);
#pragma unroll
                for (int kb = 0; kb < NRB; ++kb) {
                    gr[0].v[kb][0] = v0[kb]; gr[1].v[kb][0] = v1[kb];
                    gr[0].d[kb][0] = d0[kb]; gr[1].d[kb][0] = d1[kb];
                }
            }
#pragma unroll
            for (int kb = 0; kb < NRB; ++kb)
                dbc[0] += sh_c * ((gr[0].d[kb][0][0] + gr[0].d[kb][0][1]) + (gr[0].d[kb][0][2] + gr[0].d[kb][0][3]));
            Op gd[K];
#pragma unroll
            for (int n = 0; n < K; ++n) gd[n] = F::split(gr[n].d[0][0], gr[n].d[1][0]);
            // B operands of dX: per weight set n, the (c, o) blocks of (Q^n_c)^T for columns c' = 16rb + x
            Op qb[K][NRB];
#pragma unroll
            for (int n = 0; n < K; ++n)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    const f32x4 qv = F::mm(gd[n], rb == 0 ? tb0 : tb1, kZero4);      // (T_1 dY_n)^T tile
                    qb[n][rb] = F::split(gr[n].v[rb][0], qv);
                }
            Op wc;                                                  // fragments fetched one step ahead of the products that use them
            if (AHEAD) wc = F::get(WC, 0, lo);
#pragma unroll
            for (int lb = 0; lb < LBD; ++lb) {
                f32x4 z[NRB] = {kZero4, kZero4};
#pragma unroll
                for (int n = 0; n < K; ++n) {
                    const Op w = AHEAD ? wc : F::get(WC, n * LB + lb, lo);
                    const int nn = n + 1 < K ? n + 1 : 0, nlb = n + 1 < K ? lb : lb + 1;
                    if (AHEAD && nlb < LBD) wc = F::get(WC, nn * LB + nlb, lo);
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) z[rb] = F::mm(w, qb[n][rb], z[rb]);
                }
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    if (lb == RHB) {                               // out of the scaled space: the prologue works on plain values
                        if constexpr (F::SCALED) drh[rb] = z[rb] * pow2_mul(ikc, inv_pow2(a_c)); else drh[rb] = z[rb];
                    } else {                                       // the X plane's share: becomes the start of a gates tile (factor kg here, the gates' scale there)
                        if constexpr (F::SCALED) z[rb] *= pow2_mul(pow2_mul(ikc, inv_pow2(a_c)), kg);
                        stash_x[rb * 64 + lane] = make_float4(z[rb][0], z[rb][1], z[rb][2], z[rb][3]);
                    }
                }
            }
            // dWc_{n,c} (rows l, columns o) += [X | R*H]^T Q^n_c
            Op qd[K];
#pragma unroll
            for (int n = 0; n < K; ++n) {
                const f32x4 q0 = F::mm(tb0, gd[n], kZero4), q1 = F::mm(tb1, gd[n], kZero4);
                qd[n] = F::split(q0, q1);
            }
#pragma unroll
            for (int lb = 0; lb < LB; ++lb) {
                f32x4 c0, c1;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    c0[t] = lb == RHB ? rh_d[0][t] : zg[0][lb][0][t];
                    c1[t] = lb == RHB ? rh_d[1][t] : zg[0][lb][1][t];
                }
                Op za;
                if constexpr (F::SCALED) za = F::split(c0 * pow2_mul(sz[0][lb], sh_c), c1 * pow2_mul(sz[0][lb], sh_c)); else za = F::split(c0, c1);      // (muted / skipped: sh_c = RunScale::mute())
#pragma unroll
                for (int n = 0; n < K; ++n)
#pragma unroll
                    for (int c = 0; c < K; ++c) dWc[n][lb][c][0] = F::mm(za, c == 0 ? gd[n] : qd[n], dWc[n][lb][c][0]);
            }
        }

        // =========================================================== gate + blend backward (prologue of the gates convolution)
        DyFrag<NRB, 2> gr;
        float a_g = 1.f, sh_g = 1.f;                               // FmtH2: this node's gradient scale in the gates phase, and a / a_n
        {
            f32x4 v0[NRB], v1[NRB], d0[NRB], d1[NRB];
#pragma unroll
            for (int kb = 0; kb < NRB; ++kb) {
                f32x4 own;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float u = uv[kb][i], r = rv[kb][i], h = hv[kb][i], d = drh[kb][i];
                    v0[kb][i] = gv[kb][i] * (cv[kb][i] - h) * u * (1.f - u);
                    v1[kb][i] = d * h * r * (1.f - r);
                    own[i] = d * r + gv[kb][i] * (1.f - u);        // what H is owed directly: reset-gate path + its share of the blend
                    if constexpr (F::SCALED) own[i] *= kg;          // the H plane's gates tile starts from it (times the gates' scale, below)
                }
                stash_h[kb * 64 + lane] = make_float4(own[0], own[1], own[2], own[3]);
            }
            if constexpr (F::SCALED) {                             // the gates' dY into the node's own scale
                float m = 0.f;
#pragma unroll
                for (int kb = 0; kb < NRB; ++kb) m = vmax3_acc(m, absmax4(v0[kb]), absmax4(v1[kb]));
                const int k_was = rg.k;
                a_g = rg.node(wave_max_bits(m), sh_g, skip_g);
                // skip_g alone: the candidate phase of this node is in its sums, nothing is stored -- the node is redone with that phase muted
                if (skip_c) { rg.k = k_was; skip_g = true; sh_g = RunScale::mute(); }    // (skipped as a whole: no reference from it, no stores)
#pragma unroll
                for (int kb = 0; kb < NRB; ++kb) { v0[kb] *= a_g; v1[kb] *= a_g; }
            }
            to_acc(3, v0, d0);
            to_acc(4, v1, d1);
#pragma unroll
            for (int kb = 0; kb < NRB; ++kb) {
                gr.v[kb][0] = v0[kb]; gr.v[kb][1] = v1[kb];
                gr.d[kb][0] = d0[kb]; gr.d[kb][1] = d1[kb];
            }
            __builtin_amdgcn_wave_barrier();
        }

        // =========================================================== gates convolution (slab form on the planes), as node_bwd_x3_kernel
#pragma unroll
        for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
                dbg[hb] += sh_g * ((gr.d[kb][hb][0] + gr.d[kb][hb][1]) + (gr.d[kb][hb][2] + gr.d[kb][hb][3]));
        Op gd[2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) gd[hb] = F::split(gr.d[0][hb], gr.d[1][hb]);
        f32x4 Qv[NRB][2];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const Op& t = rb == 0 ? tb0 : tb1;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) Qv[rb][hb] = F::mm(gd[hb], t, kZero4);
        }
        Op qb[2][NRB];                                             // step s: blocks (c = s, hb = 0), (c = s, hb = 1)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            qb[0][rb] = F::split(gr.v[rb][0], gr.v[rb][1]);
            qb[1][rb] = F::split(Qv[rb][0], Qv[rb][1]);
        }
        Op wg;
        if (AHEAD) wg = F::get(WG, 0, lo);
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int lb = 0; lb < LBD; ++lb) {
                f32x4 z[NRB];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    if (n == 0 && lb == RHB) {                     // the H plane's tile starts from the prologue's share
                        const float4 sh = stash_h[rb * 64 + lane];
                        z[rb] = f32x4{sh.x, sh.y, sh.z, sh.w};
                        if constexpr (F::SCALED) z[rb] *= a_g;      // (parked with the factor kg only)
                    } else if (n == 0) {                           // the X plane's tile from the candidate's share (PL = 1)
                        const float4 sh = stash_x[rb * 64 + lane];
                        z[rb] = f32x4{sh.x, sh.y, sh.z, sh.w};
                        if constexpr (F::SCALED) z[rb] *= a_g;
                    } else {
                        z[rb] = kZero4;
                    }
                    if (!F::SCALED && ACC[lb]) z[rb] += old[n][lb][rb];          // the other consumer's gradients of the same plane
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const Op w = AHEAD ? wg : F::get(WG, (n * LB + lb) * 2 + s, lo);
                    if (AHEAD) {                                               // the next fragment in this loop nest's order: (n, lb, s) -> s, lb, n
                        const int ns = s + 1 < 2 ? s + 1 : 0, nlb = s + 1 < 2 ? lb : (lb + 1 < LBD ? lb + 1 : 0), nn = (s + 1 < 2 || lb + 1 < LBD) ? n : n + 1;
                        if (nn < K) wg = F::get(WG, (nn * LB + nlb) * 2 + ns, lo);
                    }
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) z[rb] = F::mm(w, qb[s][rb], z[rb]);
                }
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    if constexpr (F::SCALED) {                     // out of the scaled space (+ the other consumer's gradients of the same plane)
                        // (a skipped node stores zeros, or what the plane held: whatever it stores is stored again when it is redone -- no branch here)
                        const float ikg_n = skip_g ? 0.f : pow2_mul(ikg, inv_pow2(a_g));
                        if (ACC[lb]) z[rb] = z[rb] * ikg_n + old[n][lb][rb];
                        else z[rb] *= ikg_n;
                    }
                    stc_st_once(reinterpret_cast<f32x4*>(dP[n][lb] + (r0 + 16 * rb + x) * HID + 4 * g), z[rb]);
                }
            }
        Op qd[2];
        {
            f32x4 Qd[NRB][2];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                const Op& t = rb == 0 ? tb0 : tb1;
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) Qd[rb][hb] = F::mm(t, gd[hb], kZero4);
            }
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) qd[hb] = F::split(Qd[0][hb], Qd[1][hb]);
        }
#pragma unroll
        for (int n = 0; n < K; ++n)
#pragma unroll
            for (int lb = 0; lb < LB; ++lb) {
                const float (&zc)[NRB][4] = zg[n][lb];
                Op za;
                if constexpr (F::SCALED) {
                    const float zs = pow2_mul(sz[n][lb], sh_g);                                    // (skipped: sh_g = RunScale::mute())
                    za = F::split(f32x4{zc[0][0], zc[0][1], zc[0][2], zc[0][3]} * zs, f32x4{zc[1][0], zc[1][1], zc[1][2], zc[1][3]} * zs);
                }
                else za = F::split(f32x4{zc[0][0], zc[0][1], zc[0][2], zc[0][3]}, f32x4{zc[1][0], zc[1][1], zc[1][2], zc[1][3]});
#pragma unroll
                for (int c = 0; c < K; ++c)
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) dWg[n][lb][c][hb] = F::mm(za, c == 0 ? gd[hb] : qd[hb], dWg[n][lb][c][hb]);
            }
        cur = nxt;
        nxt = nx2;
        // skip_g: the wave's pass ends AT this node -- no exit of its own, the loop condition does it
        cand_done = skip_g && !skip_c;
        resume = skip_g ? node : resume;
        node = skip_g ? 0x7fffffff : next_node;
    }
    if (resume >= 0) node = resume;

    // dW tiles of block c carry the wave's final gradient scale (c = 0) or that times sT (c = 1: Q_1 = T_1 dY); db carries the scale
    const float isg_g = rg.unscale(), isg_c = rc.unscale();
    PlaneUnscale<K, LB> pug, puc;
#pragma unroll
    for (int n = 0; n < K; ++n)
#pragma unroll
        for (int lb = 0; lb < LB; ++lb) { pug.v[n][lb] = inv_pow2(sz[n][lb]); puc.v[n][lb] = inv_pow2(sz[0][lb]); }      // the candidate's input is slab 0, [X | R*H], for both weight sets
    combine_dw<K, LB, 2, CB_WAVES>(reinterpret_cast<float*>(smem_raw), dWg, dbg, a.partial_g, a.Lw, a.want_dbg, PL == 2 ? cin : -1, isg_g, isg_g / sT, isg_g, pug, pass > 0);
    combine_dw<K, LB, 1, CB_WAVES>(reinterpret_cast<float*>(smem_raw), dWc, dbc, a.partial_c, a.Lw, a.want_dbc, PL == 2 ? cin : -1, isg_c, isg_c / sT, isg_c, puc, pass > 0);
    };
    run_pass(0, args_in_kernarg_segment);
    if constexpr (F::SCALED) {
        const int nodes = args_in_kernarg_segment.nodes;
        // (the barrier also means: every wave is done with the combine's slabs before the tables are filled again)
        for (int pass = 1; __syncthreads_or(node < nodes); ++pass) run_pass(pass, kernargs_fresh<CellBwdArgs>());
    }
}

template <class F, int L, int PL, int ACCX = 0, int ACCH = 0>
int launch_cell_bwd(const CellBwdArgs& a, int* n_partials, hipStream_t stream) {
    const size_t slabs = (size_t)CB_WAVES * (4 * L * 32 + 32) * sizeof(float);
    const size_t lds = cb_lds_bytes<F>() > slabs ? cb_lds_bytes<F>() : slabs;
    static_assert(cb_lds_bytes<F>() <= stc::kMaxLdsBytes, "tables + per-wave tiles must fit the CU's LDS");
    if (lds > stc::kMaxLdsBytes) return STC_NOT_HANDLED;
    auto kern = cell_bwd_x3_kernel<F, L, PL, ACCX, ACCH>;
    if (int rc = stc::hip_status(stc::allow_lds(kern, lds), "hipFuncSetAttribute(cell bwd x3)")) return rc;
    static const int resident = stc::resident_blocks(kern, CB_THREADS, lds, 1);      // (one static per instantiation)
    const long long want = (a.nodes + CB_WAVES - 1) / CB_WAVES;
    int grid = resident < MF_BWD_MAX_GRID ? resident : MF_BWD_MAX_GRID;
    if (want < grid) grid = (int)want;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(CB_THREADS), lds, stream, a);
    STC_LAUNCH_CHECK("cell_bwd_x3 launch");
    *n_partials = grid;
    return STC_OK;
}

}  // namespace

// The form that accumulates on BOTH sides lives in a translation unit of its own (stc_cell_bwd_x3_acc2.hip = this file with
// STC_CB_ACC2_UNIT defined): this file is compiled with -mllvm -amdgpu-mfma-vgpr-form=1 (Makefile), and that LLVM pass crashes on that one
// instantiation.  The argument block crosses as an opaque pointer (the struct sits in each unit's anonymous namespace, same definition).
int stc_cell_bwd_x3_acc2(const void* args, int fmt, int* n_partials, hipStream_t stream);

#if defined(STC_CB_ACC2_UNIT)

int stc_cell_bwd_x3_acc2(const void* args, int fmt, int* n_partials, hipStream_t stream) {
    const CellBwdArgs& a = *static_cast<const CellBwdArgs*>(args);
    return fmt == STC_FMT_F16X2 ? launch_cell_bwd<FmtH2, 32, 1, 1, 1>(a, n_partials, stream) : launch_cell_bwd<FmtB3, 32, 1, 1, 1>(a, n_partials, stream);
}

#else

int stc_cell_bwd_planar_shape_ok(int C, int h) { return C == 32 && h == 16; }

template <class F>
static int dispatch_cell_bwd(const CellBwdArgs& a, int cin, int accumulate_x, int accumulate_h, int* n_partials, hipStream_t stream) {
    if (cin == 16) {
        if (accumulate_x && accumulate_h) return stc_cell_bwd_x3_acc2(&a, F::SCALED ? STC_FMT_F16X2 : STC_FMT_BF16X3, n_partials, stream);
        if (accumulate_x) return launch_cell_bwd<F, 32, 1, 1, 0>(a, n_partials, stream);
        if (accumulate_h) return launch_cell_bwd<F, 32, 1, 0, 1>(a, n_partials, stream);
        return launch_cell_bwd<F, 32, 1>(a, n_partials, stream);
    }
    return accumulate_h ? launch_cell_bwd<F, 20, 2, 0, 1>(a, n_partials, stream) : launch_cell_bwd<F, 20, 2>(a, n_partials, stream);
}

// fmt: STC_FMT_F16X2 (two fp16 pieces; every operand class scaled by powers of two the kernel finds itself, the activation planes of the dW
// products from zmax, the slots the gates forward left) or STC_FMT_BF16X3 (three bf16 pieces: fp32's range, no scales).
int stc_cell_bwd_planar_x3(const float* X, const float* H, const float* SX, const float* SH, const float* Tc, const float* Wg, const float* Wc,
                           const float* U, const float* R, const float* Cand, const float* dHnew, const float* dBm,
                           float* dX, float* dSX, float* dH, float* dSH, float* partial_g, float* partial_c, int* n_partials,
                           int want_dbg, int want_dbc, int accumulate_x, int accumulate_h, int fmt, const float* zmax,
                           long long nodes, int C, int Lw, hipStream_t stream) {
    const int cin = Lw - 16;
    if (!stc_cell_bwd_planar_shape_ok(C, 16) || nodes <= 0 || nodes >= (1ll << 31) / C || !(cin == 16 || (cin >= 1 && cin <= 4))) return STC_NOT_HANDLED;
    const float* wide[] = {H, SH, U, R, Cand, dHnew, dBm, dH, dSH};
    if (!all_aligned16(wide, 9)) return STC_NOT_HANDLED;
    CellBwdArgs a{X, H, SX, SH, U, R, Cand, dHnew, dBm, Tc, Wg, Wc, dX, dSX, dH, dSH, partial_g, partial_c, (int)nodes, want_dbg, want_dbc, Lw, zmax};
    if (cin == 16) {
        const float* more[] = {X, SX, dX, dSX};
        if (!dX || !dSX || !all_aligned16(more, 4)) return STC_NOT_HANDLED;
    } else if (accumulate_x) {
        return STC_NOT_HANDLED;       // a narrow input plane gets no gradient
    }
    return fmt == STC_FMT_F16X2 ? dispatch_cell_bwd<FmtH2>(a, cin, accumulate_x, accumulate_h, n_partials, stream)
                                : dispatch_cell_bwd<FmtB3>(a, cin, accumulate_x, accumulate_h, n_partials, stream);
}

#endif      // STC_CB_ACC2_UNIT
