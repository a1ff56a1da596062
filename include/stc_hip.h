/*
 * stc_hip.h -- C ABI of libstc_hip.so: the MI355X (gfx950) kernels behind the
 * STC-GNN multi-graph message-passing path.
 *
 * The reference (underdoc-wang/STC-GNN) is pure Python and has no FFI; the
 * "interface each entry point replaces" is therefore a span of torch ops in
 * framework/STC_GNN.py, cited per function.  The Python host
 * (stc-gnn_amd/stc_hip/_lib.py) binds these with ctypes and is the only
 * caller; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (tensor.data_ptr());
 *     nothing is allocated, freed or retained; buffers must stay alive until the
 *     stream has run the launch
 *   - all tensors are dense, row-major, contiguous fp32 unless stated;
 *     CSR indices are int32
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); launches
 *     are asynchronous; functions are re-entrant and keep no state
 *   - return 0 on success; STC_EINVAL (-1) bad argument / shape, STC_EALIGN (-2)
 *     misaligned pointer, STC_ELIMIT (-3) size beyond a kernel limit; > 0 is a
 *     hipError_t from the launch.  stc_last_error() gives a thread-local text.
 *   - sizes of zero elements are legal and launch nothing
 */
#ifndef STC_HIP_H
#define STC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STC_ABI_VERSION 33
#define STC_MAX_K 4          /* highest Chebyshev order (Ks, Kc) the node kernels accept */

/* Operand formats of the split-operand matrix-core kernels (C = 32 / 64, hidden 16).  Every fp32 operand is split into low-precision
 * pieces whose products are exact in the fp32 accumulator:
 *   STC_FMT_BF16X3  three bf16 pieces, six products  -- fp32's range, error at the fp32 rounding level (8.5e-8 on 32-term dot products);
 *   STC_FMT_F16X2   two fp16 pieces, three products  -- half the matrix instructions; error 1.7-2.9e-7 (an fp32 fmaf chain: 1.3-2.4e-7).
 * fp16 has a 5-bit exponent, so F16X2 kernels normalise EVERY operand class by powers of two (exact) -- the reference's einsum is
 * scale-free (STC_GNN.py:37-42) and so are they, for magnitudes 2^-100 .. 2^100:
 *   tables      weight and category-graph tables from their own maxima, inside the kernel;
 *   gradients   one scale per NODE, from the maximum over the node's gradient fragments, found by the wave that owns the node (what the gate
 *               prologues form can sit tens of binades below the state gradient); the dW / db sums over nodes run at a per-wave reference
 *               scale that follows the running maximum (no argument: nothing about the gradient's magnitude has to be known);
 *   activations forward: one scale per NODE, from the maximum over the node's rows of all input planes, found by the wave that owns the
 *               node (no argument); backward: the dW products sum over nodes, so they take one scale per input PLANE and launch, from
 *               the maxima the forward launch left in device memory (act_amax arguments: STC_ACT_AMAX_SLOTS floats per plane, ZERO before
 *               the forward launch, each wave leaves its own maximum by an atomic max; NULL in the backward = unscaled, i.e. plane maxima
 *               assumed within [2^-3, 65504)). */
#define STC_FMT_BF16X3 0
#define STC_FMT_F16X2 1
#define STC_ACT_AMAX_SLOTS 256   /* floats per plane row of an act_amax buffer */

#define STC_OK 0
#define STC_EINVAL (-1)
#define STC_EALIGN (-2)
#define STC_ELIMIT (-3)
#define STC_EUNSUPPORTED (-4) /* shape outside a fused fast path: use the unfused entry points */

int stc_version(void);
const char* stc_last_error(void);
/* Ceiling of the node / cell kernel dispatch, process-wide (tests, A/B runs): 0 = every path (default: split-operand matrix-core
 * kernels, then the fp32-MFMA kernels, then the generic ones), 1 = no split-operand kernels, 2 = generic kernels only.  The entry
 * points read no environment variables; apart from this setting and the thread-local error text the library keeps no state. */
int stc_set_dispatch_level(int32_t level);

/* ---- spatial aggregation -------------------------------------------------
 * Y[b,i,:] = alpha * sum_{j in row i} val[j] * X[b, colidx[j], :] + beta * Y0[b,i,:]
 *
 * X (batch, n_cols, F), Y0/Y (batch, n_rows, F).  Y0 may be NULL when beta == 0
 * and may alias Y (in-place epilogue).  With CSR(Gs^T) this is the reference's
 * 1-mode product torch.einsum('bncl,nm->bmcl', X, Gs) of STC_GNN.py:37 (F = C*L);
 * with alpha=2, beta=-1, Y0=Z_{k-2} it is one step of the Chebyshev recurrence
 * of STC_GNN.py:28 applied to the features; with CSR(Gs) it is the backward of both.
 */
int stc_csr_spmm_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                     int32_t n_rows, int32_t n_cols,
                     const float* X, const float* Y0, float* Y,
                     int32_t batch, int32_t F, float alpha, float beta, void* stream);

/* Same product on the row-blocked form of a FIXED graph (BCSR, STC_SPMM_BLOCK_ROWS x 1 blocks): the host
 * groups the rows 4 at a time and lists, per block, the distinct columns its rows touch (blk_ptr
 * (n_blocks+1), blk_cols) with the 4 values of each column, one per row of the block, zero where a row has
 * no such entry (blk_vals, nnzb x 4).  One wave produces the 4 rows of a block together, so a neighbour
 * row shared by several of them is fetched once: the 8-neighbour grid needs 18 instead of 36 row fetches
 * per 4 output rows (the direct kernel is bound by exactly that L2 -> CU gather traffic).  A graph without
 * locality degenerates to the same number of fetches as CSR.  Y0 / alpha / beta as stc_csr_spmm_f32. */
#define STC_SPMM_BLOCK_ROWS 4
int stc_bcsr_spmm_f32(const int32_t* blk_ptr, const int32_t* blk_cols, const float* blk_vals,
                      int32_t n_rows, int32_t n_cols,
                      const float* X, const float* Y0, float* Y,
                      int32_t batch, int32_t F, float alpha, float beta, void* stream);

/* Same product on the PATCH form of a fixed graph whose rows cluster AND whose clusters gather runs of consecutive rows (lattices in their
 * node numbering, grids renumbered by reverse Cuthill-McKee; not irregular meshes, whose scattered 1 KiB requests HBM serves slowly): the host groups the rows into
 * patches of up to STC_PATCH_ROWS output rows whose entries touch at most STC_PATCH_MAX_SRC distinct source rows together
 * (stc_hip/graph.py _patch_plan: 4 x 8 tiles of a lattice, greedy clusters otherwise; 1.85 source rows per output row on the 8-neighbour grid, against 4.5 row fetches through L1 / L2
 * for the row-blocked form).  One workgroup per (patch, batch element) copies the source rows into LDS, one 1 KiB column chunk at a
 * time with the next chunk already requested, and forms the patch's output rows out of LDS: the kernel runs at the rate of a plain
 * copy of X and Y (csrc/stc_spmm_patch.hip).  Per patch p:
 *   patch_src  (n_patches, STC_PATCH_MAX_SRC)         the source rows (columns of the matrix) it gathers: position q of its list at
 *                                                     [q % 4][q / 4] (wave q % 4 of the workgroup stages it), positions past the
 *                                                     end of the list repeat its first source row
 *   patch_rows (n_patches, STC_PATCH_ROWS)            its output rows, -1 = unused slot; every row of the matrix is in exactly one patch
 *   patch_cnt  (n_patches, STC_PATCH_ROWS)            entries of each output row
 *   patch_idx  (n_patches, STC_PATCH_ROWS, width) u8  per entry: position q of its column in the patch's source list
 *   patch_val  (n_patches, STC_PATCH_ROWS, width)     per entry: its value; entries in the row's CSR order, the tail of the row
 *                                                     filled with zero-weight repeats of its last entry
 * width: 4, 8, 12, 16, 24 or 32 (<= STC_PATCH_MAX_WIDTH).  F: a multiple of 256 (STC_EUNSUPPORTED otherwise: use the row-blocked form).
 * Each row's sum runs over its entries in CSR order, one fmaf each: results equal stc_csr_spmm_f32 / stc_bcsr_spmm_f32 bit for bit.
 * Y0 / alpha / beta as stc_csr_spmm_f32 (Y0 may alias Y). */
#define STC_PATCH_ROWS 32
#define STC_PATCH_MAX_SRC 64
#define STC_PATCH_MAX_WIDTH 32
int stc_patch_spmm_f32(const int32_t* patch_src, const int32_t* patch_rows, const int32_t* patch_cnt,
                       const uint8_t* patch_idx, const float* patch_val, int32_t n_patches, int32_t width,
                       int32_t n_rows, int32_t n_cols, const float* X, const float* Y0, float* Y,
                       int32_t batch, int32_t F, float alpha, float beta, void* stream);

/* ---- bf16 storage (BASELINE.json configuration 5: N = 50 176, C = 64, bf16) ------------------------------------
 * The same two products with the feature rows stored in bf16: X, Y0, Y are bf16 (2 bytes per element, passed as
 * void*), the graph values, every product and the row sums are fp32, and the result is rounded to bf16 once (round to
 * nearest even).  A bf16 value widens to fp32 exactly, so these equal the fp32 entry points run on the same
 * bf16-valued inputs up to that one final rounding.  F must be a multiple of 8 (16-byte pieces); 16-byte aligned
 * operands.  The reference has no bf16 behaviour of its own (it builds an fp32 identity, STC_GNN.py:26). */
int stc_csr_spmm_bf16(const int32_t* rowptr, const int32_t* colidx, const float* val,
                      int32_t n_rows, int32_t n_cols,
                      const void* X, const void* Y0, void* Y,
                      int32_t batch, int32_t F, float alpha, float beta, void* stream);
int stc_bcsr_spmm_bf16(const int32_t* blk_ptr, const int32_t* blk_cols, const float* blk_vals,
                       int32_t n_rows, int32_t n_cols,
                       const void* X, const void* Y0, void* Y,
                       int32_t batch, int32_t F, float alpha, float beta, void* stream);

/* stc_patch_spmm_f32 on bf16 rows (X, Y0, Y bf16, F a multiple of 512; values and sums fp32, one rounding at the store: equal to
 * stc_bcsr_spmm_bf16 bit for bit).  The same plan arrays. */
int stc_patch_spmm_bf16(const int32_t* patch_src, const int32_t* patch_rows, const int32_t* patch_cnt,
                        const uint8_t* patch_idx, const float* patch_val, int32_t n_patches, int32_t width,
                        int32_t n_rows, int32_t n_cols, const void* X, const void* Y0, void* Y,
                        int32_t batch, int32_t F, float alpha, float beta, void* stream);

/* Two-ring patch aggregation (csrc/stc_spmm_ring2.hip; ABI v27): the gradient of a planar cell's state from its pieces and the transpose
 * aggregation of the candidate's gradient in ONE launch -- what stc_spmm_sum_f32 (with U, Cand, dY) followed by the plain aggregation of dY
 * compute in two, without the dY plane:
 *     Y = sum_k add[k] + S.(A [+ A2])        Z = S.(Y * U * (1 - Cand^2))            planes (batch, n_rows, C, 16) fp32, contiguous
 * S (square, n_rows x n_rows) comes as a host-built plan over the patches of the patch form (stc_patch_spmm_f32):
 *   l2_rows  (n_patches, STC_RING2_SECOND)                 the rows staged per patch: every column the first ring's rows touch, padded with repeats
 *   l1_rows  (n_patches, STC_RING2_FIRST)                  the first ring: the patch's own rows (bit 30 set) and every column they touch; -1 = empty slot
 *   int_rows (n_patches, STC_RING2_INTERIOR)               the patch's own rows, -1 = empty; every row of S is in exactly one patch
 *   t1 (n_patches, STC_RING2_FIRST, STC_RING2_WIDTH, 2)    per entry of a first-ring row: (512 x position of its column in l2_rows, value bits);
 *   t2 (n_patches, STC_RING2_INTERIOR, STC_RING2_WIDTH, 2) per entry of an own row: (512 x slot of its column in l1_rows, value bits); rows
 *                                                          shorter than the width end in zero-valued repeats of their last entry
 * Rows of more than STC_RING2_WIDTH entries, patches whose rings exceed the sizes: no plan (the two launches remain).  n_add <= 5. */
#define STC_RING2_INTERIOR 32
#define STC_RING2_FIRST 64
#define STC_RING2_SECOND 96
#define STC_RING2_WIDTH 8
int stc_ring2_sum_f32(const int32_t* l2_rows, const int32_t* l1_rows, const int32_t* int_rows, const int32_t* t1, const int32_t* t2,
                      int32_t n_patches, int32_t n_rows,
                      const float* A, const float* A2, int32_t n_add, const float* const* add,
                      const float* U, const float* Cand, float* Y, float* Z,
                      int32_t batch, int32_t C, int32_t h, void* stream);
/* The forward counterpart on the same plan format (built for Gs^T): stc_spmm_blend_fwd_f32 (without state copies) and the plain aggregation
 * of the new state that follows it, in one launch -- the new state is aggregated out of LDS instead of being read back:
 *     Cand = tanh(A + S.Bm)      Hnew = (1 - U) H + U Cand      SHnew = S.Hnew          (reference STC_GNN.py:76-78, then :37 of the next cell) */
int stc_ring2_blend_f32(const int32_t* l2_rows, const int32_t* l1_rows, const int32_t* int_rows, const int32_t* t1, const int32_t* t2,
                        int32_t n_patches, int32_t n_rows,
                        const float* Bm, const float* A, const float* U, const float* H,
                        float* Cand, float* Hnew, float* SHnew,
                        int32_t batch, int32_t C, int32_t h, void* stream);
/* Two CHAINED aggregations on the same plan format (ABI v28): the feature-side Chebyshev recurrence of order 3 (reference STC_GNN.py:24-29
 * applied to the features, :37; BASELINE configuration 4) and its transpose, each one launch instead of two:
 *     V = alpha1 S.(A [+ A2]) + sum_k add1[k]          Z = alpha2 S.V + sum_k scale0[k] add0[k]
 *   forward   T_1 = S.X (V, stored), T_2 = 2 S.T_1 - X:                       alpha1 = 1, no add1, alpha2 = 2, add0 = {X}, scale0 = {-1}
 *   backward  d0 - d2 + S^T (d1 + 2 S^T d2)  (Clenshaw form, plan of S^T):    A = d2, alpha1 = 2, add1 = {d1}, V = NULL, alpha2 = 1, add0 = {d0.., d2..}, scale0 = {+1.., -1..}
 * V may be NULL (not stored).  n_add1 <= 2, 1 <= n_add0 <= 5; scale0 NULL = all +1.  V is formed for the patch's first ring (1.9 x redundant) and never
 * read back: what the two launches it replaces (stc_spmm_sum_f32 / stc_patch_spmm_f32 twice) pass through HBM. */
int stc_ring2_chain_f32(const int32_t* l2_rows, const int32_t* l1_rows, const int32_t* int_rows, const int32_t* t1, const int32_t* t2,
                        int32_t n_patches, int32_t n_rows,
                        const float* A, const float* A2, float alpha1, int32_t n_add1, const float* const* add1, float* V,
                        float alpha2, int32_t n_add0, const float* const* add0, const float* scale0, float* Z,
                        int32_t batch, int32_t C, int32_t h, void* stream);

/* bf16-storage node kernel (2-mode product + concat + projection + bias, STC_GNN.py:38-45) and its backward: the slabs
 * Z_n (nodes, C, L), the output Y / its gradient dY (nodes, C, Ho) and the slab gradients dZ_n are bf16 (void*); Tc, W,
 * bias, dW, db are fp32 (master weights: rounded to bf16 once per launch for the matrix cores).  Every contraction runs
 * on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; intermediate tiles that feed a second contraction (U_c = sum_n
 * Z_n W_{n,c} in the forward, Q_c = T_c dY in the backward) are rounded to bf16 in between.  T_0 is taken as the identity
 * (STC_GNN.py:26) whatever Tc[0] holds.  No dTc (the bf16 path is for fixed category graphs).
 * Shapes: Ks = Kc <= 3, C in {32, 64}, L in {16, 32}, Ho in {16, 32} (stc_bdg_node_bf16_supported), else
 * STC_EUNSUPPORTED.  Lw <= L real columns: pad columns must hold finite values and get zero gradient.
 * workspace as stc_bdg_node_bwd_f32 (stc_bdg_node_bwd_workspace_bytes(Ks, Kc, C, L, Ho, 0)). */
int stc_bdg_node_bf16_supported(int32_t Ks, int32_t Kc, int32_t C, int32_t L, int32_t Ho);
int stc_bdg_node_fwd_bf16(const void* const* Z, int32_t Ks, const float* Tc, int32_t Kc,
                          const float* W, const float* bias, void* Y,
                          int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t Ho, void* stream);
int stc_bdg_node_bwd_bf16(const void* const* Z, int32_t Ks, const float* Tc, int32_t Kc,
                          const float* W, const void* dY,
                          void* const* dZ, float* dW, float* db,
                          void* workspace, size_t workspace_bytes,
                          int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t Ho, void* stream);

/* The planar STC_Cell (see "planar cell inputs" below) on bf16 planes, K = 2, hidden 16, C in {32, 64}: every state, gate
 * and gradient plane is (nodes, C, 16) bf16 -- 32-byte rows -- the input plane of layer 0 is (nodes, C, cin) with cin in
 * 1..4; weights, biases and their gradients stay fp32.  Same mathematics and argument meaning as the _f32 entry points of
 * the same names (stc_cell_gates_fwd/bwd_planar_f32 with its fused candidate projection, stc_bdg_node_post_bwd_f32 with
 * X2, stc_spmm_blend_fwd_f32 without state copies, stc_spmm_sum_f32 with contiguous addends, stc_gru_blend_bwd_f32 in its
 * dCpre-only form); sums are fp32, each stored plane is rounded to bf16 once.  One extension: stc_cell_gates_bwd_planar_bf16
 * accepts dH == NULL and then adds the previous state's share from the GRU prologue (dRH R + dHnew (1-U)) into dZ[2], the
 * H plane's gradient, so that the state receives one gradient plane from the cell instead of two.  stc_cell_gates_fwd_planar_bf16: RH may be
 * NULL (not written) when the fused candidate projection is on (Wc, A, Bm given): stc_cell_bwd_planar_bf16 re-forms R*H. */
int stc_cell_planar_bf16_supported(int32_t Ks, int32_t Kc, int32_t C, int32_t h);
int stc_cell_gates_fwd_planar_bf16(const void* X, const void* H, const void* SX, const void* SH,
                                   const float* Tc, const float* W, const float* bias,
                                   void* U, void* Rg, void* RH,
                                   const float* Wc, const float* bc, void* A, void* Bm,
                                   int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream);
int stc_cell_gates_bwd_planar_bf16(const void* X, const void* H, const void* SX, const void* SH,
                                   const float* Tc, const float* W,
                                   const void* dCandIn, const void* Cand, const void* U, const void* Rg, const void* dHnew,
                                   void* const* dZ, float* dW, float* db, void* dH,
                                   void* workspace, size_t workspace_bytes,
                                   int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream);
/* The whole backward of a planar cell step on bf16 planes in ONE launch (counterpart of stc_cell_bwd_planar_f32 without its accumulate
 * flags; reference STC_GNN.py:65-79 through autograd): candidate backward in post-aggregation form + gate / blend backward + gates
 * backward per node, with dY = dHnew U (1 - Cand^2) and R*H formed inside (rounded to bf16 once, as the stored planes of the two
 * launches above are), the R*H plane's gradient handed over in fp32 through LDS, the candidate's X-side gradient and the state's
 * share of the gate prologue added into dX / dH.  9 planes in, 4 out (13 + 6 for stc_bdg_node_post_bwd_bf16 +
 * stc_cell_gates_bwd_planar_bf16).  Narrow input (Lw - h in 1..4): X, SX are (nodes, C, Lw - h); dX, dSX are not produced (NULL).
 * stc_cell_bwd_planar_bf16_supported(C, Lw, h): which shapes are built (C = 32: both input widths; C = 64: the wide input).
 * workspace >= stc_bdg_node_bwd_workspace_bytes(2, 2, C, 32, 32, 0) + stc_bdg_node_bwd_workspace_bytes(2, 2, C, 32, 16, 0) bytes. */
int stc_cell_bwd_planar_bf16_supported(int32_t C, int32_t Lw, int32_t h);
int stc_cell_bwd_planar_bf16(const void* X, const void* H, const void* SX, const void* SH,
                             const float* Tc, const float* Wg, const float* Wc,
                             const void* U, const void* Rg, const void* Cand, const void* dHnew, const void* dBm,
                             void* dX, void* dSX, void* dH, void* dSH,
                             float* dWg, float* dbg, float* dWc, float* dbc,
                             void* workspace, size_t workspace_bytes,
                             int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream);
int stc_bdg_node_post_bwd_bf16(const void* X, const void* X2, const float* Tc, const float* W, const void* dA, const void* dB,
                               void* dX, void* dX2, float* dW, float* db, void* workspace, size_t workspace_bytes,
                               int64_t nodes, int32_t C, int32_t Lw, int32_t Ho, void* stream);
int stc_spmm_blend_fwd_bf16(const int32_t* rowptr, const int32_t* colidx, const float* val,
                            const int32_t* blk_ptr, const int32_t* blk_cols, const float* blk_vals,
                            int32_t n_rows, int32_t n_cols, const void* Bm, const void* A,
                            const void* U, const void* H, void* Cand, void* Hnew,
                            int32_t batch, int32_t C, int32_t h, void* stream);
int stc_spmm_sum_bf16(const int32_t* rowptr, const int32_t* colidx, const float* val,
                      const int32_t* blk_ptr, const int32_t* blk_cols, const float* blk_vals,
                      int32_t n_rows, int32_t n_cols, const void* X, const void* X2,
                      int32_t n_add, const void* const* add,
                      void* Y, const void* U, const void* Cand, void* dY,
                      int32_t batch, int32_t C, int32_t h, void* stream);
int stc_gru_blend_bwd_bf16(const void* dHnew, const void* U, const void* Cand, void* dCpre, int64_t n, void* stream);
/* Output head on bf16 state rows (h = 16): H / dH bf16, y / dy and w, b, dwb fp32; otherwise as stc_head_fwd/bwd_f32
 * (workspace: stc_head_bwd_workspace_bytes(16)). */
int stc_head_fwd_bf16(const void* H, const float* w, const float* b, float* y, int64_t rows, int32_t h, void* stream);
int stc_head_bwd_bf16(const void* H, const float* w, const float* y, const float* dy, void* dH, float* dwb,
                      void* workspace, size_t workspace_bytes, int64_t rows, int32_t h, void* stream);

/* The same product with a DENSE graph matrix (the reference's learned Gs is dense: STC_GNN.py:227-243; this is :37 and its autograd
 * w.r.t. the features as they stand in the reference):  Y[b] = alpha * S x X[b] + beta * Y0[b],  S (n_rows, n_cols) row-major,
 * X (batch, n_cols, F), Y0 / Y (batch, n_rows, F); Y0 may be NULL when beta == 0 and may alias Y.  Runs on the exact-fp32 matrix
 * cores (v_mfma_f32_16x16x4_f32: an fmaf chain per output element), any sizes. */
int stc_dense_agg_f32(const float* S, int32_t n_rows, int32_t n_cols, const float* X, const float* Y0, float* Y,
                      int32_t batch, int32_t F, float alpha, float beta, void* stream);

/* out[j] (+)= alpha * sum_b < A[b,i,:], Bm[b,colidx[j],:] >   for j in row i
 * A (batch, n_rows, F), Bm (batch, n_cols, F), out (nnz).  Gradient of the
 * 1-mode product w.r.t. the graph values on a fixed pattern (autograd of :37);
 * with the full N x N pattern it yields the dense dGs of the learned graph. */
int stc_csr_sddmm_f32(const int32_t* rowptr, const int32_t* colidx,
                      int32_t n_rows, int32_t n_cols,
                      const float* A, const float* Bm, float* out,
                      int32_t batch, int32_t F, float alpha, int32_t accumulate, void* stream);

/* ---- category graph ------------------------------------------------------
 * T (K, n, n): T_0 = I, T_1 = G, T_k = (2G) T_{k-1} - T_{k-2}   (STC_GNN.py:24-29,
 * matrix side, same order as the reference).  n <= 128. */
int stc_cheby_dense_fwd_f32(const float* G, int32_t n, int32_t K, float* T, void* stream);
/* dG from dT (K,n,n); dT is used as scratch and destroyed. */
int stc_cheby_dense_bwd_f32(const float* G, const float* T, float* dT, int32_t n, int32_t K,
                            float* dG, void* stream);

/* ---- node kernel: 2-mode product + concat + projection + bias -------------
 * Y[r,d,:] = bias + sum_{n<Ks} sum_{c<Kc} sum_{c'} Tc[c][c',d] * ( Z_n[r,c',:] . W_{n,c} )
 * r over `nodes` = batch*N rows; Z_n (nodes, C, L); Tc (Kc, C, C); W (Ks*Kc*Lw, Ho)
 * with row blocks n-major, c-minor (STC_GNN.py:35-41); Y (nodes, C, Ho).
 * Replaces STC_GNN.py:38-45 without materialising the K*K*L concat.
 * Z is a HOST array of Ks device pointers.
 * Lw <= L: the slabs may carry L - Lw trailing pad columns per row (the host pads L = in + hidden
 * up to a multiple of 4 so that every row is 16-byte aligned: 17 -> 20); W has Lw rows per block,
 * pad columns are ignored in the forward and receive zero gradient. */
int stc_bdg_node_fwd_f32(const float* const* Z, int32_t Ks, const float* Tc, int32_t Kc,
                         const float* W, const float* bias, float* Y,
                         int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t Ho, void* stream);

/* Backward of the node kernel.  dZ: HOST array of Ks device pointers (nodes,C,L),
 * overwritten.  dW (Ks*Kc*Lw, Ho), db (Ho) or NULL, dTc (Kc,C,C) or NULL: overwritten.
 * workspace: >= stc_bdg_node_bwd_workspace_bytes(...) bytes, 16-byte aligned. */
size_t stc_bdg_node_bwd_workspace_bytes(int32_t Ks, int32_t Kc, int32_t C, int32_t L, int32_t Ho,
                                        int32_t want_dTc);
int stc_bdg_node_bwd_f32(const float* const* Z, int32_t Ks, const float* Tc, int32_t Kc,
                         const float* W, const float* dY,
                         float* const* dZ, float* dW, float* db, float* dTc,
                         void* workspace, size_t workspace_bytes,
                         int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t Ho, void* stream);

/* ---- GRU gate math (STC_GNN.py:68-78) --------------------------------------
 * gates:  U = sigmoid(G[:, :h]); Rg = sigmoid(G[:, h:]); CandIn = [Xt | Rg*H | 0-pad]
 *         G (rows, 2h), Xt (rows, cin), H/U/Rg (rows, h), CandIn/dCandIn (rows, cin+h+pad) */
int stc_gru_gates_fwd_f32(const float* G, const float* Xt, const float* H,
                          float* U, float* Rg, float* CandIn,
                          int64_t rows, int32_t cin, int32_t h, int32_t pad, void* stream);
/* dH_in (rows, h) or NULL: a gradient already owed to H by another consumer (the blend), added into dH
 * so that autograd does not need a separate accumulation pass; may alias dH. */
int stc_gru_gates_bwd_f32(const float* dCandIn, const float* dU, const float* H,
                          const float* U, const float* Rg, const float* dH_in,
                          float* dG, float* dXt, float* dH,
                          int64_t rows, int32_t cin, int32_t h, int32_t pad, void* stream);
/* blend:  Cand = tanh(Cpre); Hnew = (1-U)*H + U*Cand      (n = rows*h elements) */
int stc_gru_blend_fwd_f32(const float* Cpre, const float* U, const float* H,
                          float* Cand, float* Hnew, int64_t n, void* stream);
/* dCpre = dHnew*U*(1-Cand^2), dU = dHnew*(Cand-H), dH = dHnew*(1-U); dU and dH may be NULL when the consumer forms those
 * products itself (stc_cell_gates_bwd_f32, Cand form); H may be NULL when dU is */
int stc_gru_blend_bwd_f32(const float* dHnew, const float* U, const float* H, const float* Cand,
                          float* dCpre, float* dU, float* dH, int64_t n, void* stream);

/* ---- post-aggregation form of a Ks = Kc = 2 convolution -----------------------------------------------------------
 * The aggregation acts on the node axis, projection and category mix on the other two, so STC_GNN.py:35-45 reassociates to
 *     Y = A + Gs^T x Bm,   A = sum_c T_c^T (X W_{0,c}),  Bm = sum_c T_c^T (X W_{1,c}):
 * the SpMM runs on rows of C*Ho floats instead of C*L (half the bytes when Ho = 16, L = 32) and the slab Gs^T x X is never
 * formed.  Backward: dA = dY, dBm = Gs x dY (one narrow stc_csr/bcsr_spmm_f32 by the caller), then this kernel turns
 * (X, dA, dBm) into dX (nodes, C, L), dW (4*Lw, Ho) and db directly -- no second gradient slab, no SpMM after it.
 * Shapes: C in {32, 64}, L in {20, 32}, Ho = 16 (stc_bdg_node_post_supported), else STC_EUNSUPPORTED.
 * X2 != NULL (L = 32 only): planar input -- X holds columns 0..15 and X2 columns 16..31 of the rows, each (nodes, C, 16);
 * the backward then writes the gradient as the two planes dX, dX2 as well (dX2 != NULL exactly when X2 != NULL).
 * workspace as stc_bdg_node_bwd_f32 (stc_bdg_node_bwd_workspace_bytes(2, 2, C, L, Ho, 0)). */
int stc_bdg_node_post_supported(int32_t Ks, int32_t Kc, int32_t C, int32_t L, int32_t Ho);
/* forward: X (nodes, C, L) -> A = sum_c T_c^T (X W_{0,c}) + bias and Bm = sum_c T_c^T (X W_{1,c}), (nodes, C, Ho) each;
 * the caller finishes Y = A + Gs^T x Bm with stc_csr/bcsr_spmm_f32 (Y0 = A) or, for the STC_Cell's candidate
 * convolution, with stc_spmm_blend_fwd_f32 (the GRU blend of STC_GNN.py:76-78 in the SpMM's epilogue). */
int stc_bdg_node_post_fwd_f32(const float* X, const float* X2, const float* Tc, const float* W, const float* bias,
                              float* A, float* Bm,
                              int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t Ho, void* stream);
/* Y = A + S x Bm on rows of C*h floats (h = 16) with the blend as epilogue: Cand = tanh(Y), Hnew = (1-U)*H + U*Cand,
 * plus the optional state copies / side columns of stc_cell_blend_fwd_f32.  Graph as either form (see stc_spmm_bwd_*). */
int stc_spmm_blend_fwd_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                           const int32_t* blk_ptr, const int32_t* blk_cols, const float* blk_vals,
                           int32_t n_rows, int32_t n_cols, const float* Bm, const float* A,
                           const float* U, const float* H, float* Cand, float* Hnew,
                           float* copy0, int32_t copy0_ld, int32_t copy0_off, const float* side_src, int32_t side_cin,
                           float* copy1, int32_t copy1_ld, int32_t copy1_off,
                           int32_t batch, int32_t C, int32_t h, void* stream);
int stc_bdg_node_post_bwd_f32(const float* X, const float* X2, const float* Tc, const float* W, const float* dA, const float* dB,
                              float* dX, float* dX2, float* dW, float* db,
                              int32_t operand_format,                      /* STC_FMT_*: the planar forms at C = 64 take STC_FMT_F16X2, everything else runs bf16 x 3 */
                              const float* act_amax_x, const float* act_amax_x2,      /* fp16 x 2, optional: STC_ACT_AMAX_SLOTS floats each whose maximum is
                                                                                         max |X| / max |X2| (a row of a forward launch's act_amax; R*H takes H's) */
                              void* workspace, size_t workspace_bytes,
                              int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t Ho, void* stream);

/* ---- fused cell convolutions (STC_GNN.py:69-78) ------------------------------------
 * The two BDG_Dif of an STC_Cell with the gate math folded into the node kernel's epilogue, so the
 * pre-activations never go to HBM:
 *   gates: [U | R] = sigmoid(node_fwd(Z; W, bias)),  CandIn = [Xt | R*H | 0-pad]   (Xt = first cin columns of Z[0])
 *   blend: Cand = tanh(node_fwd(Z; W, bias)),        Hnew = (1-U)*H + U*Cand
 * Z, Tc, W, bias, L, Lw as stc_bdg_node_fwd_f32 (W (Ks*Kc*Lw, 2h) resp. (Ks*Kc*Lw, h)); H/U/R/Cand/Hnew
 * (nodes, C, h); CandIn (nodes, C, L).  Only the MFMA shapes are fused: stc_cell_fused_supported() tells
 * (h = 16, C in {16,32,64}, L in {20,32}, Ks = Kc <= 3); otherwise these return STC_EUNSUPPORTED and the
 * caller runs stc_bdg_node_fwd_f32 + stc_gru_gates_fwd_f32 / stc_gru_blend_fwd_f32. */
int stc_cell_fused_supported(int32_t Ks, int32_t Kc, int32_t C, int32_t L, int32_t h);
int stc_cell_gates_fwd_f32(const float* const* Z, int32_t Ks, const float* Tc, int32_t Kc,
                           const float* W, const float* bias, const float* H,
                           float* U, float* Rg, float* CandIn,
                           int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t h, int32_t cin, void* stream);
/* Backward of the gates convolution with the gate backward as its prologue (autograd of STC_GNN.py:69-75):
 *   dG = [dU*U*(1-U) | dCandIn[:, cin:cin+h]*H*Rg*(1-Rg)] is formed per node inside the kernel and never stored;
 *   outputs dZ / dW / db as stc_bdg_node_bwd_f32 for dY = dG, plus dXt = dCandIn[:, :cin] and
 *   dH = dCandIn[h part]*Rg + dH_in (dH_in may be NULL and may alias dH; with dH_in_scaled != 0 it enters as
 *   dH_in*(1-U), i.e. dH_in is the gradient of the new state and the blend backward need not write its (1-U) share).
 *   dXt may be NULL when the caller reads dCandIn[:, :cin] in place (stc_split2_f32's addA_ld).
 *   Give EITHER dU or Cand: with Cand (tanh of the candidate, saved by the forward) dH_in is the gradient dHnew of the
 *   cell's new state and the kernel forms dU = dHnew*(Cand-H) and the state share dHnew*(1-U) itself (STC_GNN.py:78).
 * dCandIn (nodes, C, L); dU/H/U/Rg/dH (nodes, C, h); dXt (nodes, C, cin).  Fused shapes only (else STC_EUNSUPPORTED). */
int stc_cell_gates_bwd_f32(const float* const* Z, int32_t Ks, const float* Tc, int32_t Kc, const float* W,
                           const float* dCandIn, const float* dU, const float* H, const float* U, const float* Rg,
                           const float* Cand, const float* dH_in, int32_t dH_in_scaled, float* const* dZ, float* dW, float* db, float* dXt, float* dH,
                           void* workspace, size_t workspace_bytes,
                           int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t h, int32_t cin, void* stream);
/* Backward of the candidate convolution with the blend backward as its prologue (autograd of STC_GNN.py:76-78):
 *   dY = dHnew*U*(1-Cand^2) is formed per node inside the kernel; outputs dZ / dW / db as stc_bdg_node_bwd_f32.
 * dHnew/U/Cand (nodes, C, h).  Together with the Cand form of stc_cell_gates_bwd_f32 this replaces stc_gru_blend_bwd_f32. */
int stc_cell_cand_bwd_f32(const float* const* Z, int32_t Ks, const float* Tc, int32_t Kc, const float* W,
                          const float* dHnew, const float* U, const float* Cand,
                          float* const* dZ, float* dW, float* db,
                          void* workspace, size_t workspace_bytes,
                          int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t h, void* stream);
/* Optional state copies (torch.cat of STC_GNN.py:68 done by the producer): besides Hnew, the new state is written into
 * columns [off, off+h) of rows of ld floats of up to two more buffers -- the [Xt | H | pad] input rows of the cells that
 * consume it, which then need no stc_concat2_f32 pass.  copy0 may also have its other columns completed: with side_src
 * (nodes, C, side_cin; side_cin == copy0_off) columns [0, side_cin) are copied from it and the pad columns
 * [side_cin+h, copy0_ld) zeroed.  Null copy pointers = none. */
int stc_cell_blend_fwd_f32(const float* const* Z, int32_t Ks, const float* Tc, int32_t Kc,
                           const float* W, const float* bias, const float* U, const float* H,
                           float* Cand, float* Hnew,
                           float* copy0, int32_t copy0_ld, int32_t copy0_off, const float* side_src, int32_t side_cin,
                           float* copy1, int32_t copy1_ld, int32_t copy1_off,
                           int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t h, void* stream);

/* ---- planar cell inputs (Ks = Kc = 2, cin = h = 16) ---------------------------------------------------------------
 * The [Xt | H] rows of STC_GNN.py:68 kept as two contiguous (nodes, C, h) planes instead of one concatenated row: a state
 * tensor then IS the X plane of the next layer's cell and the H plane of the next step's cell (no concat, no copies), and
 * its aggregation S x state is computed once (narrow SpMM) and shared by both.  X/H: the planes; SX/SH: their aggregations.
 * Forward: U, Rg (nodes, C, h) and the R*H plane RH -- the candidate convolution's input is (X, RH), see
 * stc_bdg_node_post_fwd_f32's X2.  Backward: as the Cand form of stc_cell_gates_bwd_f32 with dCandIn (nodes, C, h) the
 * gradient of the R*H plane (dX2 of stc_bdg_node_post_bwd_f32); the gradient slabs come out planar too:
 * dZ = {d X plane, d SX plane, d H plane, d SH plane}, (nodes, C, h) each, plus dH = the state's share from the gates
 * (dH = NULL: the kernel adds that share into the H plane's gradient dZ[2] itself -- one plane less to write and to sum).
 * The gradient of a state then is  sum of its consumers' direct planes + S^T (sum of their S planes): stc_spmm_sum_f32.
 * Narrow input (layer 0): Lw = cin + h with cin in 1..4 -- X / SX are (nodes, C, cin); the kernels read the slab as
 * [H | Xt | pad] (W's rows permuted inside) and the backward produces only dZ[2], dZ[3] (the input needs no gradient;
 * dZ[0], dZ[1] may be NULL).  stc_bdg_node_post_fwd/bwd_f32 take the same narrow form: X = the 16-wide plane, X2 = the
 * (nodes, C, cin) plane, L = 20, gradient for X only. */
int stc_cell_planar_supported(int32_t Ks, int32_t Kc, int32_t C, int32_t h);
/* A != NULL: the same launch also runs the candidate convolution's projection on [Xt | R*H], which the wave still
 * holds in registers: A, Bm (nodes, C, h) as stc_bdg_node_post_fwd_f32 would give for weights Wc (4*Lw, h) and bias bc --
 * that launch and its re-read of Xt and R*H are then not needed.  RH may then be NULL (not written): stc_cell_bwd_planar_f32
 * forms R*H itself; stc_bdg_node_post_bwd_f32 needs the plane. */
int stc_cell_gates_fwd_planar_f32(const float* X, const float* H, const float* SX, const float* SH,
                                  const float* Tc, const float* W, const float* bias,
                                  float* U, float* Rg, float* RH,
                                  const float* Wc, const float* bc, float* A, float* Bm,
                                  int32_t operand_format,
                                  float* act_amax,      /* STC_FMT_F16X2, optional: (4, STC_ACT_AMAX_SLOTS) floats, ZERO before the launch, that receive the
                                                           maxima of |X|, |SX|, |H|, |SH| (wide input) or |H|, |SH|, |X|, |SX| (narrow): act_amax of the backward */
                                  int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream);
int stc_cell_gates_bwd_planar_f32(const float* X, const float* H, const float* SX, const float* SH,
                                  const float* Tc, const float* W,
                                  const float* dCandIn, const float* Cand, const float* U, const float* Rg, const float* dHnew,
                                  float* const* dZ, float* dW, float* db, float* dH,
                                  int32_t operand_format,                      /* STC_FMT_* */
                                  const float* act_amax,                       /* fp16 x 2, optional: what stc_cell_gates_fwd_planar_f32 left for these planes */
                                  void* workspace, size_t workspace_bytes,
                                  int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream);

/* The whole backward of one planar cell step (autograd of STC_GNN.py:65-79 for the forward pair stc_cell_gates_fwd_planar_f32 with
 * its fused candidate projection + stc_spmm_blend_fwd_f32) in ONE launch: stc_bdg_node_post_bwd_f32 on (X, R*H; dA = dY, dB = dBm) and
 * stc_cell_gates_bwd_planar_f32 (dH = NULL form) back to back per node, with
 *   dY = dHnew * U * (1 - Cand^2) formed in the kernel (dBm = S^T dY is the caller's narrow SpMM on the dY plane it already has),
 *   R*H formed from Rg and H (the forward need not store that plane),
 *   the R*H plane's gradient handed from the candidate to the gate prologue inside the wave (never written),
 *   dX = the candidate's PLUS the gates' gradient of the X plane (one plane for the source's gradient sum instead of two).
 * In: X, H, SX, SH as above, gates Rg / U / Cand, dHnew and dBm (nodes, C, h); Wg (4*Lw, 2h), Wc (4*Lw, h).
 * Out: dX, dSX, dH (the H plane's gradient incl. the prologue's share), dSH (nodes, C, h); dWg, dbg, dWc, dbc (db* may be NULL).
 * accumulate_x / accumulate_h != 0: dX, dSX / dH, dSH already hold the gradients another cell computed for the SAME state (its other
 * consumer) and this launch adds its own to them, so the state owns one direct and one aggregated plane and stc_spmm_sum_f32 gathers one
 * operand for it instead of two.
 * Narrow input (Lw - h in 1..4): dX, dSX are not produced (may be NULL).  stc_cell_bwd_planar_supported() tells whether (C, h)
 * is built (C = 32, h = 16); workspace >= stc_cell_bwd_planar_workspace_bytes(C, Lw, h) bytes, 16-byte aligned.
 * operand_format: STC_FMT_F16X2 or STC_FMT_BF16X3 (see "operand formats" above).  act_amax (fp16 x 2, optional): the
 * (4, STC_ACT_AMAX_SLOTS) plane maxima the forward launch (stc_cell_gates_fwd_planar_f32) left for the same X, H, SX, SH: the scales of
 * the dW products' activation operands. */
int stc_cell_bwd_planar_supported(int32_t C, int32_t h);
size_t stc_cell_bwd_planar_workspace_bytes(int32_t C, int32_t Lw, int32_t h);
int stc_cell_bwd_planar_f32(const float* X, const float* H, const float* SX, const float* SH,
                            const float* Tc, const float* Wg, const float* Wc,
                            const float* U, const float* Rg, const float* Cand, const float* dHnew, const float* dBm,
                            float* dX, float* dSX, float* dH, float* dSH,
                            float* dWg, float* dbg, float* dWc, float* dbc,
                            int32_t accumulate_x, int32_t accumulate_h,
                            int32_t operand_format, const float* act_amax,
                            void* workspace, size_t workspace_bytes,
                            int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream);

/* ---- planar cell convolutions of Chebyshev order K = 3 (C = 32, h = 16) ----------------------------------------------
 * The planar form above for BDG_Dif.cheby_poly order 3 (STC_GNN.py:24-29, 35-39; BASELINE configuration 4).  Zx[n] / Zh[n],
 * n < K: T_n(S) applied to the plane on the X side / the H side of the [X | H] row -- the plane itself, S x plane and
 * 2 S (S x plane) - plane, the feature-side Chebyshev recurrence -- each (nodes, C, h); a narrow input (layer 0) has Zx[n]
 * (nodes, C, Lw - h) with Lw - h in 1..4.  A state's three planes are computed once and shared by the cells that consume it.
 *   gates_fwd: U, Rg, RH = R*H as stc_cell_gates_fwd_planar_f32.            W (K*K*Lw, 2h)
 *   cand_fwd:  candidate convolution on [X | R*H] (Zh[n] = T_n(S) R*H) with the tanh + GRU blend epilogue of
 *              stc_cell_blend_fwd_f32: Cand, Hnew from U and the previous state H.      W (K*K*Lw, h)
 *   *_bwd:     gradients of all 2K planes (dZx[n], dZh[n]; dZx may be NULL for a narrow input) and of W, b; the gate /
 *              blend backward run as prologues exactly as in stc_cell_gates_bwd_planar_f32 / stc_cell_cand_bwd_f32.
 * gates_bwd with accumulate_x != 0 (wide input, dH = NULL): dZx[n] already hold the candidate convolution's gradients of the same X-side
 * planes (cand_bwd's dZx) and the gates' are ADDED to them, so the source receives one plane per order from the cell.
 * The gradient of a plane's source is  d0 - d2 + S^T (d1 + 2 S^T d2)  (Clenshaw form of sum_n T_n(S)^T d_n): two
 * stc_spmm_sum_f32 launches with alpha / add_scale.  stc_cell_planar_k_supported() tells whether (K, C, h) is built. */
int stc_cell_planar_k_supported(int32_t K, int32_t C, int32_t h);
int stc_cell_gates_fwd_planar_k_f32(const float* const* Zx, const float* const* Zh, int32_t K, const float* Tc, const float* W, const float* bias,
                                    float* U, float* Rg, float* RH, int32_t operand_format,
                                    float* act_amax,      /* STC_FMT_F16X2, optional: (2 K, STC_ACT_AMAX_SLOTS) zero floats that receive max |plane| of
                                                             Zx[0..K-1], Zh[0..K-1] (wide input; Zh first for a narrow one): act_amax of the matching backward */
                                    int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream);
int stc_cell_cand_fwd_planar_k_f32(const float* const* Zx, const float* const* Zh, int32_t K, const float* Tc, const float* W, const float* bias,
                                   const float* U, const float* H, float* Cand, float* Hnew, int32_t operand_format, float* act_amax,
                                   int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream);
int stc_cell_gates_bwd_planar_k_f32(const float* const* Zx, const float* const* Zh, int32_t K, const float* Tc, const float* W,
                                    const float* dRH, const float* Cand, const float* U, const float* Rg, const float* dHnew,
                                    float* const* dZx, float* const* dZh, float* dW, float* db, float* dH, int32_t accumulate_x,
                                    int32_t operand_format,                     /* STC_FMT_* */
                                    const float* act_amax,                      /* fp16 x 2, optional: as left by stc_cell_gates_fwd_planar_k_f32 */
                                    void* workspace, size_t workspace_bytes, int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream);
int stc_cell_cand_bwd_planar_k_f32(const float* const* Zx, const float* const* Zh, int32_t K, const float* Tc, const float* W,
                                   const float* dHnew, const float* U, const float* Cand,
                                   float* const* dZx, float* const* dZh, float* dW, float* db,
                                   int32_t operand_format,                      /* STC_FMT_* */
                                   const float* act_amax,                       /* fp16 x 2, optional: as left by stc_cell_cand_fwd_planar_k_f32 */
                                   void* workspace, size_t workspace_bytes, int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream);

/* Y = sum_i add_scale[i] add[i] + alpha S x (X [+ X2]) on rows of C*h floats (h = 16): the gradient of a state from the pieces
 * its consumers left -- direct planes as addends (add[i]: columns [add_off[i], add_off[i]+h) of rows of add_ld[i] floats,
 * multiples of 4; a contiguous plane is ld = h, off = 0), aggregated planes X, X2 through the transposed graph.
 * n_add <= STC_SPMM_SUM_MAX_ADD, X2 may be NULL, add_scale may be NULL (all +1).  alpha / add_scale serve the order-3 form
 * (d0 - d2 + S^T (d1 + 2 S^T d2)); alpha = 1 and no scales give the order-2 sum.
 * dY != NULL: the epilogue also writes dY = Y*U*(1-Cand^2), the blend backward (STC_GNN.py:76-78) of the cell that owns the
 * state, from that cell's saved U and Cand -- the gradient is then not read again just to form it.
 * amax != NULL: n_amax floats, ZERO before the launch; afterwards their maximum is max |Y| over the launch (each wave leaves its own
 * maximum in one slot by an atomic max), at no extra pass over Y.  (Until ABI v21 the cell backward kernels took it as their gradient
 * scale; since v22 they find their gradient maxima themselves, per node.) */
#define STC_SPMM_SUM_MAX_ADD 8
int stc_spmm_sum_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                     const int32_t* blk_ptr, const int32_t* blk_cols, const float* blk_vals,
                     int32_t n_rows, int32_t n_cols, const float* X, const float* X2, float alpha,
                     int32_t n_add, const float* const* add, const int32_t* add_ld, const int32_t* add_off, const float* add_scale,
                     float* Y, const float* U, const float* Cand, float* dY,
                     float* amax, int32_t n_amax,
                     int32_t batch, int32_t C, int32_t h, void* stream);

/* ---- output head (STC_GNN.py:182-183, 206) -----------------------------------
 * The reference applies Linear(h, h/2) then Linear(h/2, 1) with NO nonlinearity in between, then a sigmoid:
 * that is one affine map h -> 1.  The host folds the two layers into w (h) and b (1 element, device) with
 * two tiny torch matmuls (which also route the gradient back to both layers) and these kernels do the
 * streaming part:   y[r] = sigmoid( <H[r,:], w> + b[0] )            H (rows, h), y (rows)
 * backward:         g = dy * y * (1 - y);  dH[r,:] = g[r] * w;  dw = sum_r g[r] H[r,:];  db = sum_r g[r]
 * dwb (h + 1 floats: dw | db) is overwritten; workspace >= stc_head_bwd_workspace_bytes(h), 16-byte aligned. */
int stc_head_fwd_f32(const float* H, const float* w, const float* b, float* y,
                     int64_t rows, int32_t h, void* stream);
size_t stc_head_bwd_workspace_bytes(int32_t h);
int stc_head_bwd_f32(const float* H, const float* w, const float* y, const float* dy,
                     float* dH, float* dwb, void* workspace, size_t workspace_bytes,
                     int64_t rows, int32_t h, void* stream);

/* ---- small graphs: one STC_Cell step in ONE launch (ABI v18) ------------------
 * Reference STC_GNN.py:65-79 (STC_Cell.forward, with BDG_Dif :31-47 for the gates and the candidate) and its autograd, for graphs small
 * enough that a sample's planes stay in cache (the SF-incidents shape: N = 100, C = 5): one workgroup per sample runs the aggregation,
 * the gates convolution, sigmoid, reset, the second aggregation, the candidate convolution, tanh and the GRU blend in sequence, on the
 * exact-fp32 matrix cores.  Ks = 2, Kc = 2, hidden 16, C <= 16, cin = 16 or 1..4 (stc_cell_small_supported).
 *   X (batch, N, C, cin), H (batch, N, C, 16): Xt and Ht_1;  CSR (rowptr, colidx, val) of Gs^T (forward) / of Gs (backward), fixed graphs;
 *   Tc (Kc, C, C): Chebyshev stack of the category graph;  Wg (Ks*Kc*(cin+16), 32), bg (32) | NULL, Wc (.., 16), bc (16) | NULL in the
 *   reference's layout;  outputs U, R, Cand, Hnew (batch, N, C, 16) and, kept for the backward, RH = R*H, Zg = Gs^T x [H | Xt | 0]
 *   (batch, N*C, 16 + 4*XQ floats per row, XQ = 4 for cin = 16 else 1) and Zc = Gs^T x RH (batch, N*C, 16).
 * When the sample fits the compute unit's LDS (graph of nnz entries + the planes: the SF shape does) the gathers run from LDS; larger
 * samples take the same kernels with the gathers served by L2.
 * Backward: dHnew -> dX / dH (NULL = not wanted; accumulate_x / accumulate_h: ADD to what the buffer holds, the state's other consumer
 * having written it) and the parameter gradients ADDED into the rows of sample b of dparams (batch * R, params_ld),
 * R = stc_cell_small_param_rows() (one row per wave of the sample's workgroup):
 *   [dWg (Ks*Kc*L*32) | dbg (32) | dWc (Ks*Kc*L*16) | dbc (16)], L = cin + 16 -- partial sums, no atomics, no cross-wave reduction: the
 * caller zeroes the buffer once per backward pass, every cell of a layer adds to it, and one sum over the rows finishes the gradient.
 * workspace: stc_cell_small_workspace_bytes(N, C, cin, batch), 16-byte aligned.
 * Learned graphs (ABI v19): the gradients of Gs and Gc are products over ALL cells of a step, so the launches only leave their operands --
 * Z0, Z0c, Z1c (forward, optional, laid out like Zg): the slabs [H | Xt | 0], [R*H | Xt | 0] and [Gs^T x (R*H) | Gs^T x Xt | 0];
 * dZ1c / dZ1g (backward, optional, like Zg): the gradients of the candidate's / gates' aggregated slab; dYg (batch, N*C, 32) / dYc
 * (batch, N*C, 16), optional: the gate / candidate pre-activation gradients -- and stc_graph_grad_f32 / stc_mix_grad_f32 form
 * dGs^T = sum dZ1 x Z0 and dT_c = sum V_c x dY with a few stacked products per backward pass (stc_hip/small.py).  NULL = not wanted.
 * graph_is_dense != 0: the caller vouches that the CSR is the FULL n x n pattern with columns in order (nnz = n*n), i.e. val is a dense
 * row-major matrix (graph.full_pattern): the aggregations then run as matrix products instead of row gathers -- on the staged planes with
 * phase = 0; in the split form (phase != 0; since round 5) over the node tiles that cover the workgroup's own rows, sources read from
 * global memory, so that the pairs 5 / 6 / 7 work for dense graphs too (a boundary node tile is formed by both neighbours; each stores its own rows).
 * phase / splits (ABI v20; fused phases v21): phase = 0, splits = 1: the whole cell step in this launch, one workgroup per sample (`batch` of
 * the chip's 256 compute units work).  phase != 0, splits = G: the sample's row tiles are dealt in CONTIGUOUS ranges over G workgroups and
 * the launch runs ONLY phase 1..4 (forward: aggregate, gates, aggregate R*H, candidate; backward: candidate convolution,
 * transpose-aggregate + gate backward, gates convolution, transpose-aggregate) or, v21, a pair of phases whose second half reads only the
 * workgroup's own rows: forward 5 = 1 + 2, 6 = 3 + 4; backward 7 = 2 + 3 (dense graphs too, since round 5).
 * The caller launches the phases in order -- (5, 6) and (1, 7, 4), or one by one -- the launch boundaries being the barriers; same
 * buffers, same results.  The workspace holds three gradient slabs and the gate gradients per row (stc_cell_small_workspace_bytes).
 * dparams then has R * G rows per sample (row (b * G + g) * R + r).
 * Chebyshev ORDER 3 (ABI v32: Ks = Kc = 3; `Ks` and the second-graph arguments are new in both entry points, `Ks` in the workspace query): the
 * third spatial slab is Z_2 = T_2(S) Z with T_2(S) = 2 S^2 - I formed on the MATRIX side, as the reference's cheby_poly forms it (STC_GNN.py:24-29):
 * (rowptr2, colidx2, val2, nnz2) is the CSR of T_2 in the orientation of the launch -- forward: 2 (Gs^T)^2 - I, backward: 2 Gs^2 - I -- built by the
 * caller (stc_hip/graph.py CsrGraph.second_order; 25 entries per row on an 8-neighbour grid).  Z_2 aggregates the same input rows as Z_1, so the
 * phases and the split forms are those of order 2.  Zg2 (like Zg) and Zc2 (like Zc) hold T_2.[H | Xt | 0] and T_2.(R*H) for the backward.  Fixed CSR
 * graphs only (graph_is_dense = 0, no learned-graph operands), nothing staged in LDS; order 2: pass NULL / 0 for the new pointers. */
int stc_cell_small_supported(int32_t Ks, int32_t Kc, int32_t C, int32_t cin, int32_t h);
size_t stc_cell_small_workspace_bytes(int32_t n_nodes, int32_t C, int32_t cin, int32_t batch, int32_t Ks);
int stc_cell_small_param_rows(void);
int stc_cell_small_fwd_f32(const int32_t* rowptr, const int32_t* colidx, const float* val, int32_t n_nodes, int32_t nnz,
                           int32_t graph_is_dense, const int32_t* rowptr2, const int32_t* colidx2, const float* val2, int32_t nnz2,
                           const float* X, int32_t cin, const float* H, const float* Tc, int32_t Ks, int32_t Kc,
                           const float* Wg, const float* bg, const float* Wc, const float* bc,
                           float* U, float* R, float* Cand, float* Hnew, float* RH, float* Zg, float* Zc, float* Zg2, float* Zc2, float* Z0,
                           float* Z0c, float* Z1c, int32_t phase, int32_t splits, int32_t batch, int32_t C, void* stream);
int stc_cell_small_bwd_f32(const int32_t* rowptr, const int32_t* colidx, const float* val, int32_t n_nodes, int32_t nnz,
                           int32_t graph_is_dense, const int32_t* rowptr2, const int32_t* colidx2, const float* val2, int32_t nnz2,
                           const float* X, int32_t cin, const float* H, const float* Tc, int32_t Ks, int32_t Kc,
                           const float* Wg, const float* Wc, const float* U, const float* R, const float* Cand,
                           const float* RH, const float* Zg, const float* Zc, const float* Zg2, const float* Zc2, const float* dHnew,
                           float* dX, int32_t accumulate_x, float* dH, int32_t accumulate_h,
                           float* dparams, int64_t params_ld, int32_t has_bg, int32_t has_bc,
                           float* dZ1c, float* dZ1g, float* dYg, float* dYc,
                           void* workspace, size_t workspace_bytes, int32_t phase, int32_t splits, int32_t batch, int32_t C, void* stream);

/* Gradients of learned graphs from the planes the cell launches left (ABI v20): sums over every selected cell step and sample, g = sel * batch
 * + sample -> plane (cell0 + sel * cell_step) * batch + sample of buffers holding (cells, batch, N, F):
 *   stc_graph_grad_f32:  partials[chunk][n][m]   = sum_{g = chunk, chunk + n_chunks, ..} sum_f A[g][n][f] * B[g][m][f]      (F % 4 == 0)
 *   stc_mix_grad_f32:    partials[chunk][fa][fb] = sum_g sum_n A[g][n][fa] * B[g][n][fb]                                   
 * fp32 products and per-g sums on the matrix cores, FLOAT64 accumulation over g; partials (n_chunks, ..) in float64 are WRITTEN (no
 * atomics: the caller adds them).  chunk_stride (ABI v30): doubles between the blocks of consecutive chunks, 0 = dense (N * N / Fa * Fb) -- with
 * a stride the blocks of SEVERAL products sit side by side in one (n_chunks, total) buffer and one sum over its rows adds them all.  The dGs^T of a learned dense Gs is  sum [dZ1g x Z0 + dZ1c x Z0c]  (graph form); dT_c[c][d] =
 * sum_{ks, l, o} W[(ks, c, l), o] * Q_ks[c, l, d, o]  with  Q_ks = Z_ks^T . dY  (mix form), per convolution and parameter set. */
int stc_graph_grad_f32(const float* A, const float* B, double* partials, int32_t n_chunks, int32_t cell0, int32_t cell_step, int32_t n_sel,
                       int32_t batch, int32_t N, int32_t F, int64_t chunk_stride, void* stream);
int stc_mix_grad_f32(const float* A, const float* B, double* partials, int32_t n_chunks, int32_t cell0, int32_t cell_step, int32_t n_sel,
                     int32_t batch, int32_t N, int32_t Fa, int32_t Fb, int64_t chunk_stride, void* stream);

/* Gradient of the category graph through ONE BDG_Dif for FEW categories (ABI v33; reference STC_GNN.py:38-42 through autograd), on the
 * exact-fp32 matrix cores:  dTc[c][p][d] = sum_r sum_o U_c[r][p][o] * dY[r][d][o],  U_c[r] = sum_n Z_n[r] . W[(n, c, :), :]  (r: nodes).
 * C <= 16 (stc_mix_dt_supported): the rows of floor(16 / C) consecutive nodes are one tile, as the host packs them for the matrix-core node
 * kernels (stc_hip/ops.py _node_pack) -- whose backward (stc_bdg_node_bwd_f32 on a block-diagonal category graph, dTc = NULL) leaves dT_c
 * to this entry point.  Z[n] (rows, L) for n < Ks with rows = nodes * C (any node count: the last tile may hold fewer), L in {20, 32}, columns >= Lw
 * are padding; W (Ks * Ks * Lw, Ho), Ho in {16, 32}; dY (rows, Ho); dTc (Ks, C, C) is WRITTEN (dTc[0] = 0: T_0 = I is a constant, as in
 * stc_bdg_node_bwd_f32).  Fixed-order sums: bitwise reproducible.
 * workspace: stc_mix_dt_workspace_bytes(Ks), 16-byte aligned. */
int stc_mix_dt_supported(int32_t Ks, int32_t Kc, int32_t C, int32_t L, int32_t Ho);
size_t stc_mix_dt_workspace_bytes(int32_t Ks);
int stc_mix_dt_f32(const float* const* Z, int32_t Ks, const float* W, const float* dY, float* dTc, void* workspace,
                   size_t workspace_bytes, int64_t rows, int32_t C, int32_t L, int32_t Lw, int32_t Ho, void* stream);

/* ---- MixedFusion of the learned graph generator (reference STC_GNN.py:246-261, called by MGP_Gen :227-243) ----------------------------
 *   gate = sigmoid(W_A vec(A) + b_A + W_P vec(P) + b_P),   G = gate * A + (1 - gate) * P        D = n^2 entries; W_A, W_P (D, D) row-major
 *                                                                                               (nn.Linear weights), D % 4 == 0
 * Forward: both matrices streamed once (one wave per output row); `gate` is kept for the backward.  Backward, from dG: db (D) =
 * dG (A - P) gate (1 - gate) -- the gradient of both biases --, dW_A = db (x) vec(A), dW_P = db (x) vec(P) (D, D, written once),
 * dP = dG (1 - gate) + W_P^T db and, when dA is not NULL, dA = dG gate + W_A^T db.  The column sums go through per-chunk partials
 * in the workspace and are added in a fixed order: bitwise reproducible.  dW_A and dW_P may be NULL together (frozen weights: the two
 * (D, D) outer products are then not written).  Replaces two GEMVs (forward) and two outer products + one or two transposed GEMVs
 * (backward) of the autograd graph of :253-260. */
size_t stc_mixed_fusion_workspace_bytes(int32_t D, int32_t want_dA);
int stc_mixed_fusion_fwd_f32(const float* WA, const float* bA, const float* WP, const float* bP, const float* A, const float* P,
                             float* gate, float* G, int32_t D, void* stream);
int stc_mixed_fusion_bwd_f32(const float* WA, const float* WP, const float* A, const float* P, const float* gate, const float* dG,
                             float* dWA, float* dWP, float* db, float* dP, float* dA,
                             void* workspace, size_t workspace_bytes, int32_t D, void* stream);
/* ---- front end of the learned graph generator (ABI v31; reference STC_GNN.py:229-232 spatial branch, :237-240 category branch) -----------
 *   U = tanh(alpha X Wu), V = tanh(alpha X Wv);   P = sum_k U_k V_k^T (R x R);   Ps = softmax(relu(P - P^T), -1)        (the einsum pair of :231)
 * X is addressed as x[k][r][f] = X[k k_stride + r r_stride + f f_stride] (floats), k = 0..K-1 the (sample, time) slices, r the R rows, f the F
 * features: the spatial branch takes rows = nodes / features = categories, the category branch the transposed window (:236) -- same kernels.
 *   stc_mgp_uv_fwd_f32       U, V (R, K, h): the pair product is then the plain matrix product P = U' V'^T on (R, K h) operands (the caller's GEMM)
 *   stc_mgp_uv_bwd_f32       partials (K, 2, F, h): [k][0] = alpha x_k^T [(1 - U^2) dU]_k, [k][1] likewise for V; the caller sums over k
 *                            (fixed order: reproducible) -> dWu, dWv.  X gets no gradient (it is the data window).
 *   stc_mgp_softmax_fwd_f32  Ps (R, R) from P (R, R)
 *   stc_mgp_softmax_bwd_f32  dP (R, R) from dPs: softmax backward, relu mask (gradient where P - P^T > 0), antisymmetric part; rowdot: R floats
 *                            of scratch.  dP must not alias an input.
 * Replaces ~60 torch launches per training step of the reference's full model at the SF shape by 18. */
int stc_mgp_uv_fwd_f32(const float* X, int64_t k_stride, int64_t r_stride, int64_t f_stride, const float* Wu, const float* Wv, float alpha,
                       float* U, float* V, int32_t K, int32_t R, int32_t F, int32_t h, void* stream);
int stc_mgp_uv_bwd_f32(const float* X, int64_t k_stride, int64_t r_stride, int64_t f_stride, const float* U, const float* V,
                       const float* dU, const float* dV, float alpha, float* partials, int32_t K, int32_t R, int32_t F, int32_t h, void* stream);
int stc_mgp_softmax_fwd_f32(const float* P, float* Ps, int32_t R, void* stream);
int stc_mgp_softmax_bwd_f32(const float* P, const float* Ps, const float* dPs, float* rowdot, float* dP, int32_t R, void* stream);
/* Adam on one large fp32 parameter (ABI v29; the optimizer of the reference's harness step, Model_Trainer.py:71-87: torch.optim.Adam with
 * L2 weight decay, no amsgrad), one streaming launch: the two MixedFusion matrices above are 2 x 400 MB at the SF shape.
 *     g' = g + weight_decay p;   m = m + (1 - beta1)(g' - m);   v = beta2 v + (1 - beta2) g'^2
 *     p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 * t = *step, a DEVICE float the caller has already incremented (so a captured launch replays with the right bias corrections).
 * n a multiple of 4, all tensors 16-byte aligned and distinct.  The hyper-parameters come as doubles: 1 - beta is formed in double, as torch forms it. */
int stc_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, const float* step, double lr, double beta1, double beta2, double eps,
                 double weight_decay, void* stream);


/* ---- small helpers ---------------------------------------------------------
 * y += a*x over n elements (Chebyshev backward g_{k-2} -= g_k) */
int stc_axpy_f32(float a, const float* x, float* y, int64_t n, void* stream);
/* out (rows, a+b+pad) = [A (rows,a) | B (rows,b) | zeros]   (torch.cat of STC_GNN.py:68) and its inverse */
int stc_concat2_f32(const float* A, const float* B, float* out,
                    int64_t rows, int32_t a, int32_t b, int32_t pad, void* stream);
/* split: A = src[:, :a] (+ addA + addA2), B = src[:, a:a+b] (+ addB + addB2); the addends may be NULL and may alias A/B
 * (accumulate in place).  A or B may be NULL when that half is not wanted.
 * addA_ld: row stride of addA in floats (0 = a, i.e. dense): lets addA be the first a columns of a wider buffer. */
int stc_split2_f32(const float* src, const float* addA, const float* addB, float* A, float* B,
                   int64_t rows, int32_t a, int32_t b, int32_t pad, int32_t addA_ld,
                   const float* addA2, const float* addB2, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STC_HIP_H */
