// Internal: MFMA fast path of the node kernels (stc_node_mfma.hip), tried first by the C entry
// points in stc_node.hip.  Returns STC_OK when it launched, STC_NOT_HANDLED when the shape is
// outside the fast path (the generic VALU kernels then run), or an error code.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

constexpr int STC_NOT_HANDLED = 1 << 20;

// optional extra destinations of the new state of stc_cell_blend_fwd_f32 (host view; null pointer = none)
struct StcStateCopies {
    float* dst[2]; int ld[2], off[2];
    const float* side_src; int side_cin;        // belongs to dst[0]
};

int stc_node_fwd_mfma(const float* const* Z, int Ks, const float* Tc, int Kc, const float* W, const float* bias,
                      float* Y, long long nodes, int C, int L, int Lw, int Ho, hipStream_t stream);

// partial: workspace of >= stc_node_bwd_mfma_partials() * (Ks*Kc*L*Ho + Ho) floats; on success
// *n_partials tells the caller how many per-workgroup partial rows to reduce.
int stc_node_bwd_mfma_max_partials();
int stc_node_bwd_mfma(const float* const* Z, int Ks, const float* Tc, int Kc, const float* W, const float* dY,
                      float* const* dZ, float* partial, int* n_partials, int want_db,
                      long long nodes, int C, int L, int Lw, int Ho, hipStream_t stream);

// Fused STC_Cell epilogues (hidden width 16 only): STC_NOT_HANDLED outside the MFMA shapes.
int stc_cell_fused_shape_ok(int K, int C, int L, int h);
int stc_cell_gates_fwd_mfma(const float* const* Z, int K, const float* Tc, const float* W, const float* bias,
                            const float* H, float* U, float* R, float* CandIn,
                            long long nodes, int C, int L, int Lw, int cin, hipStream_t stream);
int stc_cell_blend_fwd_mfma(const float* const* Z, int K, const float* Tc, const float* W, const float* bias,
                            const float* U, const float* H, float* Cand, float* Hnew, const StcStateCopies* copies,
                            long long nodes, int C, int L, int Lw, hipStream_t stream);
int stc_cell_gates_bwd_mfma(const float* const* Z, int K, const float* Tc, const float* W,
                            const float* dCandIn, const float* dU, const float* H, const float* U, const float* R, const float* dH_in, const float* Cand,
                            float* const* dZ, float* dXt, float* dH, float* partial, int* n_partials, int want_db,
                            long long nodes, int C, int L, int Lw, int cin, int dh_scaled, hipStream_t stream);

// Split-operand bf16 MFMA versions of the same five launches (stc_node_x3.hip; C in {32, 64}), tried before the fp32
// MFMA ones.  Same contract and return values.
int stc_node_fwd_x3(const float* const* Z, int Ks, const float* Tc, int Kc, const float* W, const float* bias,
                    float* Y, long long nodes, int C, int L, int Lw, int Ho, hipStream_t stream);
int stc_node_bwd_x3(const float* const* Z, int Ks, const float* Tc, int Kc, const float* W, const float* dY,
                    float* const* dZ, float* partial, int* n_partials, int want_db,
                    long long nodes, int C, int L, int Lw, int Ho, hipStream_t stream);
int stc_cell_gates_fwd_x3(const float* const* Z, int K, const float* Tc, const float* W, const float* bias,
                          const float* H, float* U, float* R, float* CandIn,
                          long long nodes, int C, int L, int Lw, int cin, hipStream_t stream);
int stc_cell_blend_fwd_x3(const float* const* Z, int K, const float* Tc, const float* W, const float* bias,
                          const float* U, const float* H, float* Cand, float* Hnew, const StcStateCopies* copies,
                          long long nodes, int C, int L, int Lw, hipStream_t stream);
int stc_cell_gates_bwd_x3(const float* const* Z, int K, const float* Tc, const float* W,
                          const float* dCandIn, const float* dU, const float* H, const float* U, const float* R, const float* dH_in, const float* Cand,
                          float* const* dZ, float* dXt, float* dH, float* partial, int* n_partials, int want_db,
                          long long nodes, int C, int L, int Lw, int cin, int dh_scaled, hipStream_t stream);

// Candidate convolution's backward with the blend backward as its prologue (hidden 16): dY = dHnew * U * (1 - Cand^2).
int stc_cell_cand_bwd_mfma(const float* const* Z, int K, const float* Tc, const float* W,
                           const float* dHnew, const float* U, const float* Cand,
                           float* const* dZ, float* partial, int* n_partials, int want_db,
                           long long nodes, int C, int L, int Lw, hipStream_t stream);
int stc_cell_cand_bwd_x3(const float* const* Z, int K, const float* Tc, const float* W,
                         const float* dHnew, const float* U, const float* Cand,
                         float* const* dZ, float* partial, int* n_partials, int want_db,
                         long long nodes, int C, int L, int Lw, hipStream_t stream);

// Post-aggregation form of a K = 2 convolution, Y = A + S.Bm (stc_node_x3.hip): backward from (X, dA = dY, dBm = S^T dY).
int stc_node_post_shape_ok(int K, int C, int L, int Ho);
int stc_node_post_bwd_x3(const float* X, const float* X2, const float* Tc, const float* W, const float* dA, const float* dB, float* dX, float* dX2,
                         float* partial, int* n_partials, int want_db, int fmt,
                         const float* zmax_x, const float* zmax_x2,      // fp16 x 2: activation-maximum slots of the two input planes
                         long long nodes, int C, int L, int Lw, int Ho, hipStream_t stream);
int stc_node_post_fwd_x3(const float* X, const float* X2, const float* Tc, const float* W, const float* bias, float* A, float* Bm,
                         long long nodes, int C, int L, int Lw, int Ho, hipStream_t stream);

// Planar cell inputs (stc_node_x3.hip): the [Xt | H] rows as two contiguous (nodes, C, 16) planes each, K = 2.
int stc_cell_gates_fwd_planar_x3(const float* X, const float* H, const float* SX, const float* SH, const float* Tc, const float* W,
                                 const float* bias, float* U, float* R, float* RH,
                                 const float* Wc, const float* bc, float* A, float* Bm,      // A != null: + the candidate's projection (C = 32)
                                 int fmt,                                                   // STC_FMT_*: operand format of the matrix-core products
                                 float* zmax,                                               // fp16 x 2, optional: (4, 256) slots that receive max |input plane|
                                 long long nodes, int C, int Lw, hipStream_t stream);
int stc_cell_gates_bwd_planar_x3(const float* X, const float* H, const float* SX, const float* SH, const float* Tc, const float* W,
                                 const float* dCandIn, const float* Cand, const float* U, const float* R, const float* dHnew,
                                 float* const* dZ, float* dH, float* partial, int* n_partials, int want_db, int fmt, const float* zmax,
                                 long long nodes, int C, int Lw, hipStream_t stream);

// The whole backward of a planar K = 2 cell step in one launch (stc_cell_bwd_x3.hip; C = 32): candidate (post-aggregation form) and
// gates convolutions back to back per node.  partial_g / partial_c: per-workgroup rows [dW | db] of the two convolutions.
int stc_cell_bwd_planar_shape_ok(int C, int h);
int stc_cell_bwd_planar_x3(const float* X, const float* H, const float* SX, const float* SH, const float* Tc, const float* Wg, const float* Wc,
                           const float* U, const float* R, const float* Cand, const float* dHnew, const float* dBm,
                           float* dX, float* dSX, float* dH, float* dSH, float* partial_g, float* partial_c, int* n_partials,
                           int want_dbg, int want_dbc, int accumulate_x, int accumulate_h, int fmt, const float* zmax,
                           long long nodes, int C, int Lw, hipStream_t stream);

// Planar cell convolutions of order K = 3 (stc_node_x3.hip): Zx[n] / Zh[n] = T_n(S) of the X-side / H-side plane; mode 1 gates, 2 candidate.
int stc_cell_planar_k_shape_ok(int K, int C, int h);
int stc_cell_conv_fwd_planar_k_x3(const float* const* Zx, const float* const* Zh, int K, const float* Tc, const float* W, const float* bias, int mode,
                                  const float* H, const float* Uin, float* U, float* R, float* RH, float* Cand, float* Hnew, int fmt, float* zmax,
                                  long long nodes, int C, int Lw, hipStream_t stream);
int stc_cell_conv_bwd_planar_k_x3(const float* const* Zx, const float* const* Zh, int K, const float* Tc, const float* W, int mode,
                                  const float* dRH, const float* Cand, const float* U, const float* R, const float* dHnew,
                                  float* const* dZx, float* const* dZh, float* dH, float* partial, int* n_partials, int want_db,
                                  long long nodes, int C, int Lw, int accumulate_x, int fmt, const float* zmax, hipStream_t stream);

// Fixed-order reduction of the backward kernels' per-workgroup partial rows [dW (nW) | db (Ho)] into dW, db (db may be null).
int stc_node_reduce_partials(const float* partial, int n_parts, int nW, int Ho, float* dW, float* db, hipStream_t stream);
