"""Autograd operators of the STC-GNN hot path, each a thin host sequence of HIP launches.

    bdg_dif      BDG_Dif.forward  (reference STC_GNN.py:31-47)  = SpMM hops + node kernel
    gru_gates    split / sigmoid / reset*H / second concat (STC_GNN.py:71-75)
    gru_blend    tanh + GRU blend (STC_GNN.py:76-78)
    concat2      torch.cat([Xt, Ht_1], -1) (STC_GNN.py:68)
    cheby_dense  BDG_Dif.cheby_poly on the small category graph (STC_GNN.py:24-29)

Decomposition of one BDG_Dif (K = Ks = Kc in the reference, kept separate here):

    Z_0 = X,  Z_1 = Gs^T X,  Z_k = 2 Gs^T Z_{k-1} - Z_{k-2}         K-1 CSR SpMM launches
    Y   = node(Z_0..Z_{K-1}; T_c(Gc), W, b)                         one fused node kernel

i.e. the Chebyshev recurrence runs on the FEATURES (never forms Gs^2; equal math because
polynomials of Gs commute with Gs) and the K*K*L concat is never written.  Backward:

    dZ_k, dW, db, dT_c = node_bwd(dY)
    for k = K-1 .. 2:   dZ_{k-1} += 2 Gs dZ_k ;  dZ_{k-2} -= dZ_k ;  dGs += 2 <Z_{k-1}, dZ_k>
    dX = dZ_0 + Gs dZ_1 ;  dGs += <Z_0, dZ_1>

All launches go to ``kernels()`` -- the ctypes front of libstc_hip.so.  There is no other
implementation in the product: without the built library or without a ROCm device the
first call raises ``StcError``.
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .graph import SpatialOperand
from .small import small_graph_supported, stc_small_graph

_kernels = None


def kernels():
    """The kernel set every operator launches through (HIP; created on first use)."""
    global _kernels
    if _kernels is None:
        from ._lib import HipKernels
        _kernels = HipKernels()
    return _kernels


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------- Chebyshev set of Gc
class _ChebyDense(Function):
    @staticmethod
    def forward(ctx, G, K: int):
        G = _c(G)
        T = G.new_empty((K,) + tuple(G.shape))
        kernels().cheby_dense_fwd(G, K, T)
        ctx.save_for_backward(G, T)
        return T

    @staticmethod
    @once_differentiable
    def backward(ctx, dT):
        G, T = ctx.saved_tensors
        dG = torch.empty_like(G)
        kernels().cheby_dense_bwd(G, T, dT.contiguous().clone(), dG)    # dT is consumed as scratch
        return dG, None


def cheby_dense(G: torch.Tensor, K: int) -> torch.Tensor:
    """(K, n, n) stack T_0..T_{K-1} of a small dense graph, matrix side as the reference."""
    if G.dim() != 2 or G.shape[0] != G.shape[1]:
        raise ValueError(f'category graph must be square, got {tuple(G.shape)}')
    return _ChebyDense.apply(G, K)


# ----------------------------------------------------------------------------- BDG_Dif
_NODE_PACK = True        # few categories (C <= 16): floor(16 / C) nodes per row tile of the matrix-core node kernels (``_node_pack``; tests flip it)
_PACK_SPLIT_ROWS = 1 << 18   # rows from which a node count that the tile does not divide is worth a second launch for its remainder


def _spatial_slabs(X, fwd_val, op: SpatialOperand, Ks: int):
    """[Z_0 = X, Z_1 = Gs^T X, Z_k = 2 Gs^T Z_{k-1} - Z_{k-2}]: the Ks-1 SpMM launches of one BDG_Dif."""
    k = kernels()
    B, N, C, L = X.shape
    F = C * L
    Zs = [X]
    for order in range(1, Ks):
        Zk = torch.empty_like(X)
        if order == 1:
            k.csr_spmm(op.fwd_rowptr, op.fwd_colidx, fwd_val, N, N, X.view(B, N, F), None, Zk.view(B, N, F), 1.0, 0.0, plan=op.fwd_plan)
        else:
            k.csr_spmm(op.fwd_rowptr, op.fwd_colidx, fwd_val, N, N, Zs[-1].view(B, N, F),
                       Zs[-2].view(B, N, F), Zk.view(B, N, F), 2.0, -1.0, plan=op.fwd_plan)
        Zs.append(Zk)
    return Zs


def _node_pack(X: torch.Tensor, Tc: torch.Tensor, Ks: int, Ho: int) -> int:
    """How many nodes one row tile of the matrix-core node kernels takes when the categories are few (C <= 16); 0: the route does not apply.
    The fp32-MFMA node kernels (csrc/stc_node_mfma.hip) work on tiles of 16 category rows and take a node of Cr <= 16 rows per tile; the node
    kernel is node-local (reference STC_GNN.py:38-45: the 2-mode product and the projection touch one node's C x L rows), so floor(16 / C)
    consecutive nodes ARE one node of p C categories whose category graph is block-diagonal -- same rows in memory, T_c repeated on the
    diagonal (``_block_diag``).  BASELINE configuration 2 (C = 8: two nodes per tile) otherwise runs the generic vector kernel: 35 / 181 us
    forward / backward for 13 / 32 + 8 on the matrix cores; the SF shape's C = 5 packs three nodes into 15 rows.  1: one node per tile
    (C > 8, or a small node count that p does not divide); 0: C > 16, a shape the matrix-core kernels do not take, bf16 rows, the switch
    off.  When p does not divide a LARGE node count the last (count mod p) nodes run one per tile in a second launch (``_packed_spans``)."""
    B, N, C, L = X.shape
    if not _NODE_PACK or X.dtype != torch.float32 or C > 16:
        return 0
    Kc = Tc.shape[0]
    if Ks != Kc or not 1 <= Ks <= 3 or Ho not in (16, 32) or L not in (20, 32):      # = fast_path_shape of csrc/stc_node_mfma.hip
        return 0
    p = 16 // C
    # a node count that p does not divide: the last nodes run one per tile in a launch of their own (``_packed_spans``) -- worth two more small
    # launches only when the packed part is large (N = 50 176, C = 5, 8 samples: 38.8 -> 59.3 samples/s; the SF shape's 3 200 nodes at order 3
    # with learned graphs: 10.1 ms with one node per tile, 11.2 with the split)
    return p if (B * N) % p == 0 or B * N * C >= _PACK_SPLIT_ROWS else 1


def _packed_spans(R: int, p: int):
    """[(first node, node count, nodes per tile)]: the nodes that fill whole tiles of p, then the remainder one node per tile."""
    main = R - R % p
    return [(0, main, p)] + ([(main, R - main, 1)] if main < R else []) if main else [(0, R, 1)]


def _cell_pack(k, R: int, C: int, Ks: int, Kc: int, L: int, h: int) -> int:
    """Nodes per row tile for the FUSED cell kernels of the per-cell path (gate math in the node kernels' epilogues / prologues): 1 where
    they take C itself, 16 / C where C divides 16 and the node count (the rows of 16 / C nodes are then exactly one tile: ``_node_pack``),
    0 where neither holds (the composed sequence: node kernel, gate kernel, node kernel, blend kernel)."""
    if k.cell_fused_supported(Ks, Kc, C, L, h):
        return 1
    if _NODE_PACK and C < 16 and 16 % C == 0 and R % (16 // C) == 0 and k.cell_fused_supported(Ks, Kc, 16, L, h):
        return 16 // C
    return 0


def _block_diag(Tc: torch.Tensor, p: int) -> torch.Tensor:
    """(Kc, C, C) -> (Kc, pC, pC) with T_c on the diagonal blocks: the category graph of p nodes taken as one."""
    Kc, C, _ = Tc.shape
    out = Tc.new_zeros(Kc, p, C, p, C)
    out.diagonal(dim1=1, dim2=3).copy_(Tc.unsqueeze(-1).expand(Kc, C, C, p))       # (diagonal view: (Kc, C, C, p))
    return out.view(Kc, p * C, p * C)


def _bdg_forward(X, W, b, Tc, fwd_val, op: SpatialOperand, Ks: int):
    """Launch sequence of one BDG_Dif forward on raw tensors; returns (Y, [Z_0..Z_{Ks-1}])."""
    k = kernels()
    B, N, C, L = X.shape
    Ho = W.shape[1]
    Zs = _spatial_slabs(X, fwd_val, op, Ks)
    Y = X.new_empty(B, N, C, Ho)
    p = max(1, _node_pack(X, Tc, Ks, Ho))
    R = B * N
    for lo, n, q in _packed_spans(R, p):
        k.bdg_node_fwd([z.view(R, C, L)[lo:lo + n].view(n // q, q * C, L) for z in Zs], Tc if q == 1 else _block_diag(Tc, q), W, b,
                       Y.view(R, C, Ho)[lo:lo + n].view(n // q, q * C, Ho))
    return Y, Zs


def _mix_grad(Zs, dY, W, Kc: int) -> torch.Tensor:
    """dT_c[p, d] = sum_r sum_o U_c[r, p, o] dY[r, d, o] with U_c = sum_n Z_n W_{n,c} (autograd of STC_GNN.py:38 w.r.t. the category graph) for few
    categories: ``stc_mix_dt_f32`` on the tiles of 16 rows the packed node kernels run on.  (As library GEMMs -- Q_n = Z_n^T . dY over the rows,
    then a contraction with W -- two launches of 41 us at configuration 2's shape, 64 workgroups each.)"""
    B, N, C, L = Zs[0].shape
    dTc = W.new_empty(Kc, C, C)
    kernels().mix_dT([z.view(B * N, C, L) for z in Zs], W, dY.view(B * N, C, W.shape[1]), dTc)
    return dTc


def _bdg_backward_slabs(dY, Zs, W, Tc, op: SpatialOperand, Ks: int, has_bias: bool, need_Tc: bool, need_val: bool, gates=None, cand=None, pack: int = 1):
    """Node-kernel backward and every hop of the Chebyshev recurrence but the last.

    Returns (g, dW, db | None, dTc | None, dval | None) with g = [g_0, g_1, ...] such that
    dX = g_0 + Gs.g_1 (g_1 absent for Ks = 1): the caller performs that last product, plain or with an
    element-wise consumer fused into its epilogue.
    """
    k = kernels()
    B, N, C, L = Zs[0].shape
    Ho = W.shape[1]
    F = C * L
    dZ = [torch.empty_like(Zs[0]) for _ in range(Ks)]
    dW = torch.empty_like(W)
    db = W.new_empty(Ho) if has_bias else None
    dTc = torch.empty_like(Tc) if need_Tc else None
    # pack > 1 (the fused prologue kernels on few categories, ``_cell_pack``): `pack` nodes are one node of pack * C categories, block-diagonal T_c
    rows = lambda ts: [t.view(B * N // pack, pack * C, t.shape[-1]) for t in ts]
    Tcp = Tc if pack == 1 else _block_diag(Tc, pack)
    if gates is not None:       # dY = gate pre-activation gradient, formed inside the kernel from (dCandIn, dU, H, U, R)
        dCandIn, Cand, H, U, Rg, dHnew, dH = gates   # dU = dHnew * (Cand - H); dH = dCandIn[h part] * R + dHnew * (1 - U); dXt stays in dCandIn
        dCandIn, Cand, H, U, Rg, dHnew, dH = rows((dCandIn, Cand, H, U, Rg, dHnew, dH))
        k.cell_gates_bwd(rows(Zs), Tcp, W, dCandIn, None, H, U, Rg, dHnew, rows(dZ), dW, db, None, dH, dH_in_scaled=True, Cand=Cand)
    elif cand is not None:      # dY = dHnew * U * (1 - Cand^2), formed inside the kernel
        k.cell_cand_bwd(rows(Zs), Tcp, W, *rows(cand), rows(dZ), dW, db)
    else:
        dY = _c(dY)
        rows = lambda ts: [t.view(B * N, C, t.shape[-1]) for t in ts]
        p = _node_pack(Zs[0], Tc, Ks, Ho)
        if p >= 1:
            # few categories: floor(16 / C) nodes per row tile of the matrix-core kernel (``_node_pack``); that kernel leaves dT_c to the caller
            R = B * N
            for i, (lo, n, q) in enumerate(_packed_spans(R, p)):
                dW_i, db_i = (dW, db) if i == 0 else (torch.empty_like(dW), None if db is None else torch.empty_like(db))
                k.bdg_node_bwd([z.view(R, C, L)[lo:lo + n].view(n // q, q * C, L) for z in Zs], Tc if q == 1 else _block_diag(Tc, q), W,
                               dY.view(R, C, Ho)[lo:lo + n].view(n // q, q * C, Ho), [z.view(R, C, L)[lo:lo + n].view(n // q, q * C, L) for z in dZ],
                               dW_i, db_i, None)
                if i:                                                    # (the remainder's share of the parameter gradients)
                    dW.add_(dW_i)
                    if db is not None:
                        db.add_(db_i)
            if need_Tc:
                dTc = _mix_grad(Zs, dY, W, Tc.shape[0])
        else:
            k.bdg_node_bwd(rows(Zs), Tc, W, dY.view(B * N, C, Ho), rows(dZ), dW, db, dTc)
    dval = torch.zeros_like(op.fwd_val) if need_val else None
    v3 = lambda t: t.view(B, N, F)
    dense = op.nnz == N * N                                        # learned dense Gs: the "pattern" is the whole matrix

    def values_grad(A, Bm, alpha):
        """dval[i, j] += alpha * sum_b <A[b, i, :], Bm[b, j, :]> (gradient of the 1-mode product w.r.t. the graph values)."""
        if dense:
            # on the full pattern this is ONE dense GEMM, (N, B*F) x (B*F, N): a plain library GEMM (rocBLAS through torch),
            # not a sampled product -- one wave per stored entry (stc_csr_sddmm_f32) took 524 us per call at the SF shape,
            # 62 % of the learned-graph train step
            a2 = A.transpose(0, 1).reshape(N, B * F)
            b2 = Bm.transpose(0, 1).reshape(N, B * F)
            dval.view(N, N).addmm_(a2, b2.t(), alpha=alpha)
        else:
            k.csr_sddmm(op.fwd_rowptr, op.fwd_colidx, N, N, A, Bm, dval, alpha, True)

    for order in range(Ks - 1, 1, -1):
        k.csr_spmm(op.bwd_rowptr, op.bwd_colidx, op.bwd_val, N, N, v3(dZ[order]), v3(dZ[order - 1]),
                   v3(dZ[order - 1]), 2.0, 1.0, plan=op.bwd_plan)
        if dZ[order].dtype == torch.bfloat16:
            dZ[order - 2].sub_(dZ[order])                        # bf16 slabs, Ks >= 3 only: a torch stream op (no bf16 axpy entry point)
        else:
            k.axpy(-1.0, dZ[order], dZ[order - 2])
        if need_val:
            values_grad(v3(dZ[order]), v3(Zs[order - 1]), 2.0)
    if Ks > 1 and need_val:
        values_grad(v3(dZ[1]), v3(Zs[0]), 1.0)
    return dZ[:2], dW, db, dTc, dval


def _bdg_backward(dY, Zs, W, Tc, op: SpatialOperand, Ks: int, has_bias: bool, need_X: bool, need_Tc: bool, need_val: bool):
    """Launch sequence of one BDG_Dif backward; returns (dX | None, dW, db | None, dTc | None, dval | None)."""
    g, dW, db, dTc, dval = _bdg_backward_slabs(dY, Zs, W, Tc, op, Ks, has_bias, need_Tc, need_val)
    if Ks > 1 and need_X:
        B, N, C, L = g[0].shape
        v3 = lambda t: t.view(B, N, C * L)
        kernels().csr_spmm(op.bwd_rowptr, op.bwd_colidx, op.bwd_val, N, N, v3(g[1]), v3(g[0]), v3(g[0]), 1.0, 1.0, plan=op.bwd_plan)
    return (g[0] if need_X else None), dW, db, dTc, dval


class _BdgDif(Function):
    @staticmethod
    def forward(ctx, X, W, b, Tc, fwd_val, op: SpatialOperand, Ks: int):
        X, W, Tc, fwd_val = _c(X), _c(W), _c(Tc), _c(fwd_val)
        Y, Zs = _bdg_forward(X, W, b, Tc, fwd_val, op, Ks)
        ctx.save_for_backward(W, Tc, *Zs)
        ctx.op = op
        ctx.has_bias = b is not None
        ctx.Ks = Ks
        return Y

    @staticmethod
    @once_differentiable
    def backward(ctx, dY):
        W, Tc, *Zs = ctx.saved_tensors
        need_X, _, _, need_Tc, need_val = ctx.needs_input_grad[:5]
        dX, dW, db, dTc, dval = _bdg_backward(dY, Zs, W, Tc, ctx.op, ctx.Ks, ctx.has_bias, need_X, need_Tc, need_val)
        return dX, dW, db, dTc, dval, None, None


def bdg_dif(X: torch.Tensor, op: SpatialOperand, Tc: torch.Tensor, W: torch.Tensor,
            b: Optional[torch.Tensor], Ks: int, pad: int = 0) -> torch.Tensor:
    """Y (B,N,C,Ho) of one bi-dimensional graph diffusion; ``Tc`` from ``cheby_dense``.

    ``pad``: the last ``pad`` columns of X's feature axis are zero padding added by the caller
    (``concat2(..., pad)`` / ``gru_gates(..., pad)``) to make rows 16-byte aligned; W keeps its
    reference shape (Ks*Kc*(L-pad), Ho).
    """
    if X.dim() != 4:
        raise ValueError(f'BDG_Dif input must be (B,N,C,L), got {tuple(X.shape)}')
    B, N, C, L = X.shape
    Kc = Tc.shape[0]
    if N != op.n:
        raise ValueError(f'X has {N} nodes, the spatial graph {op.n}')
    if Tc.shape[1] != C:
        raise ValueError(f'X has {C} categories, the category graph {Tc.shape[1]}')
    if W.shape[0] != Ks * Kc * (L - pad):
        raise ValueError(f'W has {W.shape[0]} rows, expected Ks*Kc*L = {Ks * Kc * (L - pad)}')
    if X.dtype == torch.bfloat16:
        # bf16 feature rows (BASELINE configuration 5): fixed graphs only, shapes of the bf16 matrix-core kernels, no fallback
        if Tc.requires_grad or op.fwd_val.requires_grad:
            raise ValueError('BDG_Dif on bfloat16 features needs fixed graphs (no gradient for Gs / Gc)')
        if not kernels().node_bf16_supported(Ks, Kc, C, L, W.shape[1]):
            raise ValueError(f'BDG_Dif on bfloat16 features: Ks=Kc<=3, C in (32, 64), L in (16, 32), Ho in (16, 32); got Ks={Ks} Kc={Kc} C={C} L={L} Ho={W.shape[1]}')
    return _BdgDif.apply(X, W, b, Tc, op.fwd_val, op, Ks)


# ----------------------------------------------------------------------------- GRU gate math
class _GruGates(Function):
    @staticmethod
    def forward(ctx, G, Xt, H, pad):
        G, Xt, H = _c(G), _c(Xt), _c(H)
        U = torch.empty_like(H)
        Rg = torch.empty_like(H)
        CandIn = H.new_empty(H.shape[:-1] + (Xt.shape[-1] + H.shape[-1] + pad,))
        kernels().gru_gates_fwd(G, Xt, H, U, Rg, CandIn)
        ctx.save_for_backward(H, U, Rg)
        ctx.cin = Xt.shape[-1]
        ctx.pad = pad
        ctx.xshape = Xt.shape
        return U, CandIn

    @staticmethod
    @once_differentiable
    def backward(ctx, dU, dCandIn):
        H, U, Rg = ctx.saved_tensors
        dU = torch.zeros_like(U) if dU is None else _c(dU)
        dCandIn = H.new_zeros(H.shape[:-1] + (ctx.cin + H.shape[-1] + ctx.pad,)) if dCandIn is None else _c(dCandIn)
        dG = H.new_empty(H.shape[:-1] + (2 * H.shape[-1],))
        dXt = H.new_empty(ctx.xshape)
        dH = torch.empty_like(H)
        kernels().gru_gates_bwd(dCandIn, dU, H, U, Rg, dG, dXt, dH)
        return dG, dXt, dH, None


def gru_gates(G, Xt, H, pad: int = 0):
    """(update, cat[Xt, reset*H, zeros(pad)]) from the gate pre-activations G (.., 2h)."""
    return _GruGates.apply(G, Xt, H, pad)


class _GruBlend(Function):
    @staticmethod
    def forward(ctx, Cpre, U, H):
        Cpre, U, H = _c(Cpre), _c(U), _c(H)
        Cand = torch.empty_like(H)
        Hnew = torch.empty_like(H)
        kernels().gru_blend_fwd(Cpre, U, H, Cand, Hnew)
        ctx.save_for_backward(U, H, Cand)
        return Hnew

    @staticmethod
    @once_differentiable
    def backward(ctx, dHnew):
        U, H, Cand = ctx.saved_tensors
        dCpre, dU, dH = torch.empty_like(H), torch.empty_like(H), torch.empty_like(H)
        kernels().gru_blend_bwd(_c(dHnew), U, H, Cand, dCpre, dU, dH)
        return dCpre, dU, dH


def gru_blend(Cpre, U, H):
    """(1-U)*H + U*tanh(Cpre)."""
    return _GruBlend.apply(Cpre, U, H)


class _Concat2(Function):
    @staticmethod
    def forward(ctx, A, Bm, pad):
        A, Bm = _c(A), _c(Bm)
        out = A.new_empty(A.shape[:-1] + (A.shape[-1] + Bm.shape[-1] + pad,))
        kernels().concat2(A, Bm, out)
        ctx.shapes = (A.shape, Bm.shape)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d):
        sa, sb = ctx.shapes
        dA, dB = d.new_empty(sa), d.new_empty(sb)
        kernels().split2(_c(d), dA, dB)
        return dA, dB, None


def concat2(A, Bm, pad: int = 0):
    """cat([A, B, zeros(pad)], dim=-1) for tensors that agree on every leading dimension."""
    if A.shape[:-1] != Bm.shape[:-1]:
        raise ValueError(f'concat2: leading shapes differ: {tuple(A.shape)} vs {tuple(Bm.shape)}')
    return _Concat2.apply(A, Bm, pad)


# ----------------------------------------------------------------------------- output head
class _Head(Function):
    @staticmethod
    def forward(ctx, H, w, b):
        H, w, b = _c(H), _c(w), _c(b)
        y = H.new_empty(H.shape[:-1], dtype=torch.float32)          # bf16 state rows: the prediction stays fp32
        kernels().head_fwd(H, w, b, y)
        ctx.save_for_backward(H, w, y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        H, w, y = ctx.saved_tensors
        dH = torch.empty_like(H)
        dwb = w.new_empty(w.numel() + 1)
        kernels().head_bwd(H, w, y, _c(dy), dH, dwb)
        return dH, dwb[:-1], dwb[-1:]


def head(H: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """sigmoid(<H[..., :], w> + b): the reference's Linear(h, h/2) -> Linear(h/2, 1) -> sigmoid with the two
    layers folded by the caller (no nonlinearity between them); returns H's shape without the last axis."""
    if w.shape != (H.shape[-1],) or b.shape != (1,):
        raise ValueError(f'head: w {tuple(w.shape)} / b {tuple(b.shape)} do not match feature width {H.shape[-1]}')
    return _Head.apply(H, w, b)


# ----------------------------------------------------------------------------- whole STC_Cell
class _StcCell(Function):
    """One STC_Cell step (reference STC_GNN.py:65-79) as a single autograd node.

    Same kernels as the composed path (concat2 -> bdg -> gru_gates -> bdg -> gru_blend); what it adds is
    the backward bookkeeping: the three gradients owed to H (blend, reset gate, concat) and the two owed
    to Xt are summed inside the gate / split kernels instead of by separate autograd accumulation passes,
    and nothing but the slabs the backward really needs is kept (Z of both convolutions, U, R, Cand, H).
    """

    @staticmethod
    def forward(ctx, Xt, H, Wg, bg, Wc, bc, Tc, fwd_val, op: SpatialOperand, Ks: int):
        k = kernels()
        Xt, H, Wg, Wc, Tc, fwd_val = _c(Xt), _c(H), _c(Wg), _c(Wc), _c(Tc), _c(fwd_val)
        cin, h = Xt.shape[-1], H.shape[-1]
        pad = -(cin + h) % 4                      # rows padded to 16 bytes (17 -> 20); W keeps its reference shape
        lead = H.shape[:-1]
        XH = H.new_empty(lead + (cin + h + pad,))
        k.concat2(Xt, H, XH)
        U, Rg, CandIn = torch.empty_like(H), torch.empty_like(H), torch.empty_like(XH)
        Cand, Hnew = torch.empty_like(H), torch.empty_like(H)
        B, N, C, L = XH.shape
        pack = _cell_pack(k, B * N, C, Ks, Tc.shape[0], L, h)
        if pack:
            # gate math in the node kernels' epilogues: the pre-activations never go to HBM (few categories: `pack` nodes per row tile)
            rows = lambda ts: [t.view(B * N // pack, pack * C, t.shape[-1]) for t in ts]
            Tcp = Tc if pack == 1 else _block_diag(Tc, pack)
            Zg = _spatial_slabs(XH, fwd_val, op, Ks)
            k.cell_gates_fwd(rows(Zg), Tcp, Wg, bg, *rows((H, U, Rg, CandIn)))
            Zc = _spatial_slabs(CandIn, fwd_val, op, Ks)
            k.cell_blend_fwd(rows(Zc), Tcp, Wc, bc, *rows((U, H, Cand, Hnew)))
        else:
            G, Zg = _bdg_forward(XH, Wg, bg, Tc, fwd_val, op, Ks)
            k.gru_gates_fwd(G, Xt, H, U, Rg, CandIn)
            del G
            Cpre, Zc = _bdg_forward(CandIn, Wc, bc, Tc, fwd_val, op, Ks)
            k.gru_blend_fwd(Cpre, U, H, Cand, Hnew)
        ctx.save_for_backward(H, U, Rg, Cand, Wg, Wc, Tc, *Zg, *Zc)
        ctx.op, ctx.Ks, ctx.cin = op, Ks, cin
        ctx.bias = (bg is not None, bc is not None)
        return Hnew

    @staticmethod
    @once_differentiable
    def backward(ctx, dHnew):
        k = kernels()
        H, U, Rg, Cand, Wg, Wc, Tc, *Z = ctx.saved_tensors
        Ks, op, cin = ctx.Ks, ctx.op, ctx.cin
        Zg, Zc = Z[:Ks], Z[Ks:]
        need_Xt, need_H = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_Tc, need_val = ctx.needs_input_grad[6], ctx.needs_input_grad[7]
        B, N, C, L = Zc[0].shape
        v3 = lambda t: t.view(B, N, C * L)
        bwd = (op.bwd_rowptr, op.bwd_colidx, op.bwd_val, op.bwd_plan)
        # the gate backward can run as the prologue of the gates convolution's node backward (dG is never stored); it then
        # also takes over two pure data movements: the state's (1 - U) share of the blend (read from dHnew in place) and
        # dXt = d[x part] (left in place, added by the final split straight from the candidate gradient's rows)
        pack = 0 if (need_Tc or need_val) else _cell_pack(k, B * N, C, Ks, Tc.shape[0], L, H.shape[-1])
        pro = pack > 0
        dHnew = _c(dHnew)
        dH = torch.empty_like(H)
        dXt = H.new_empty(H.shape[:-1] + (cin,))
        if pro:         # the blend backward (dCpre, dU, the state share) is formed inside the two node backward kernels
            dCpre = dU = dG = None
            g, dWc, dbc, dTc, dval = _bdg_backward_slabs(None, Zc, Wc, Tc, op, Ks, ctx.bias[1], False, False, cand=(dHnew, U, Cand), pack=pack)
        else:
            dCpre, dU = torch.empty_like(H), torch.empty_like(H)
            k.gru_blend_bwd(dHnew, U, H, Cand, dCpre, dU, dH)                     # dH = dHnew * (1 - U)
            dG = H.new_empty(H.shape[:-1] + (2 * H.shape[-1],))
            # candidate convolution: d[Xt | R*H] = g0 + Gs.g1, consumed by the gate backward
            g, dWc, dbc, dTc, dval = _bdg_backward_slabs(dCpre, Zc, Wc, Tc, op, Ks, ctx.bias[1], need_Tc, need_val)
        gates_pro, dci = None, g[0]
        if Ks > 1:
            k.csr_spmm(*bwd[:3], N, N, v3(g[1]), v3(g[0]), v3(g[0]), 1.0, 1.0, plan=op.bwd_plan)
        if pro:
            gates_pro = (dci, Cand, H, U, Rg, dHnew, dH)
        else:
            k.gru_gates_bwd(dci, dU, H, U, Rg, dG, dXt, dH, dH_in=dH)
        # gates convolution: d[Xt | H] = g0 + Gs.g1, split and added to what Xt and H are already owed
        g, dWg, dbg, dTc2, dval2 = _bdg_backward_slabs(dG, Zg, Wg, Tc, op, Ks, ctx.bias[0], need_Tc, need_val, gates=gates_pro, pack=max(1, pack))
        if need_Xt or need_H:
            if Ks > 1:
                k.csr_spmm(*bwd[:3], N, N, v3(g[1]), v3(g[0]), v3(g[0]), 1.0, 1.0, plan=op.bwd_plan)
            if pro:
                k.split2(g[0], dXt, dH, addA=dci, addB=dH, addA_ld=L)             # + d[x part] of the candidate, read in place
            else:
                k.split2(g[0], dXt, dH, addA=dXt, addB=dH)                        # + the concat's share, in place
        if need_Tc:
            dTc = dTc + dTc2
        if need_val:
            dval = dval + dval2
        return (dXt if need_Xt else None), (dH if need_H else None), dWg, dbg, dWc, dbc, dTc, dval, None, None


def stc_cell(Xt, H, op: SpatialOperand, Tc, Wg, bg, Wc, bc, Ks: int):
    """Ht of one STC_Cell step; gates / candidate BDG_Dif parameters in the reference's shapes."""
    if Xt.dim() != 4 or H.dim() != 4 or Xt.shape[:-1] != H.shape[:-1]:
        raise ValueError(f'stc_cell: Xt {tuple(Xt.shape)} and H {tuple(H.shape)} must be (B,N,C,*) with equal leading shape')
    B, N, C, h = H.shape
    L = Xt.shape[-1] + h
    Kc = Tc.shape[0]
    if N != op.n or Tc.shape[1] != C:
        raise ValueError(f'stc_cell: graphs are for N={op.n}, C={Tc.shape[1]}; got N={N}, C={C}')
    if Wg.shape != (Ks * Kc * L, 2 * h) or Wc.shape != (Ks * Kc * L, h):
        raise ValueError(f'stc_cell: W shapes {tuple(Wg.shape)}, {tuple(Wc.shape)} do not match Ks*Kc*L={Ks * Kc * L}, h={h}')
    return _StcCell.apply(Xt, H, Wg, bg, Wc, bc, Tc, op.fwd_val, op, Ks)


# ----------------------------------------------------------------------------- a whole schedule of cells as ONE autograd node
# Which forms the cell-graph executor uses.  Module constants (tests flip them to reach the forms other shapes still take -- C = 64 runs the
# two-launch backward, K = 3 the order-3 planar cells, other widths the interleaved rows); not environment switches.
_CELL_GRAPH = True       # encoder + decoder as ONE autograd node (False: one node per cell, ``_StcCell``)
_FUSE_POST = True        # candidate projection as a second stage of the planar gates forward
_PLANAR = True           # cells with 16 + 16-column inputs read them as two planes (no concat, shared S.state)
_POST_AGG = True         # candidate convolution as Y = A + S.Bm (narrow SpMM after the node kernel)
_PLANAR_K3 = True        # Chebyshev order 3: planar cells on three planes per side (T_0, T_1, T_2 of S)
_ACC_PLANES = True       # one-launch cell backward: a state's second consumer adds into the first one's planes
_SMALL = True            # small graphs (N*C rows per sample fit the caches, C <= 16): one launch per cell step and direction
_RING2 = True            # state gradient + transpose aggregation of dY in one launch where the graph has a two-ring plan (no dY plane)
_RING2_FWD = True        # ... and the forward's blend + aggregation of the new state (stc_ring2_blend_f32)


def cell_graph_supported(op: SpatialOperand, Tc, Ks: int, C: int, h: int, x_widths, dtype=torch.float32) -> bool:
    """Whether ``stc_cell_graph`` can run a schedule: matrix-core cell kernels for every row width that occurs, hidden 16,
    graphs that need no gradient (``csr-fixed`` mode).  ``dtype`` = storage type of the state tensors: bfloat16 runs the
    all-planar bf16 kernel set (Ks = Kc = 2; inputs 16 or 1..4 columns wide)."""
    if not _CELL_GRAPH or h != 16:
        return False
    k = kernels()
    if _SMALL and small_graph_supported(k, op, Tc, Ks, C, h, x_widths, dtype):
        return True                                                   # small graphs, fixed or learned: one launch per cell step (stc_hip/small.py)
    if Tc.requires_grad or op.fwd_val.requires_grad:
        return False
    if dtype == torch.bfloat16:
        return (_PLANAR and _POST_AGG and _FUSE_POST and Ks == 2 and Tc.shape[0] == 2 and k.bf16.cell_planar_supported(Ks, 2, C, h)
                and all(w == h or 1 <= w <= 4 for w in x_widths))
    return all(k.cell_fused_supported(Ks, Tc.shape[0], C, w + h + (-(w + h)) % 4, h) for w in set(x_widths))


def _alias_slice(base: torch.Tensor, i: int) -> torch.Tensor:
    """base[i] as a tensor of its own that shares the storage WITHOUT being a view of ``base`` in autograd's books: the
    slice is saved for backward while ``base`` is returned as the Function's output, which view tracking forbids."""
    t = base.new_empty(0)
    t.set_(base.untyped_storage(), base.storage_offset() + i * base.stride(0), base.shape[1:], base.stride()[1:])
    return t


def _amplification(op: SpatialOperand, Ks: int) -> float:
    """What one BDG_Dif of order Ks can amplify a state by: the graph's largest absolute row sum to the power Ks - 1 (T_2(S) = 2 S^2 - I has
    row sums of up to 2 r^2 + 1; order 3 on a graph with row sums of 8 puts the reference's own fp32 noise at 3.5e-4 on the prediction)."""
    return float(op.row_sum_bound) ** max(1, Ks - 1)


class _StcCellGraph(Function):
    """Encoder + decoder (any DAG of STC_Cells whose inputs are other cells' states) as one autograd node.

    Per cell the kernels are those of ``_StcCell``'s fused path.  What owning the whole schedule adds:
      * planar cells (inputs of 16 + 16 columns, i.e. every cell above layer 0): the [Xt | H] row of reference
        STC_GNN.py:68 is never built -- the kernels read the two state tensors as two planes, and the aggregation S.state is
        formed ONCE per state (narrow SpMM) for all the cells that consume it (``stc_cell_gates_fwd/bwd_planar_f32``);
      * interleaved cells (layer 0: 1 + 16 columns padded to 20): no concat pass either -- the producer's blend epilogue
        writes the new state straight into their input rows (state copies of ``stc_spmm_blend_fwd_f32`` / ``stc_cell_blend_fwd_f32``);
      * the candidate convolution in post-aggregation form where the kernels exist (``_POST_AGG``);
      * no autograd accumulation passes: a state consumed by two cells (next step, next layer) gets its two gradient
        contributions summed inside the consumers' final split (``stc_split2_f32`` addA2 / addB2), in schedule order.
    schedule[j] = (stack, ('ext', i) | ('cell', k), ('ext', i) | ('cell', k)): parameter set, source of Xt, source of H.
    """

    @staticmethod
    def forward(ctx, op: SpatialOperand, Ks: int, schedule, outputs, n_ext: int, Tc, fwd_val, *tensors):
        k = kernels().for_graph(_amplification(op, Ks))              # (a heavy graph: the 24-bit operand format, _lib.HEAVY_ROW_SUM)
        ext = [_c(t) for t in tensors[:n_ext]]
        bf16 = ext[0].dtype == torch.bfloat16                       # bf16 state planes: the all-planar bf16 kernel set
        if bf16:
            k = k.bf16
        flat = tensors[n_ext:]
        stacks = [tuple(None if p is None else _c(p) for p in flat[i:i + 4]) for i in range(0, len(flat), 4)]   # (Wg, bg, Wc, bc)
        Tc, fwd_val = _c(Tc), _c(fwd_val)
        h = 16
        n_cells = len(schedule)
        width = lambda src: ext[src[1]].shape[-1] if src[0] == 'ext' else h
        cin = [width(x) for _, x, _ in schedule]
        consumers = [[] for _ in range(n_cells)]
        for j, (_, x, hs) in enumerate(schedule):
            if hs[0] == 'cell':
                consumers[hs[1]].append((j, 'h'))
            if x[0] == 'cell':
                consumers[x[1]].append((j, 'x'))
        ref = ext[0]
        B, N, C = ref.shape[:3]
        rows = lambda ts: [t.view(B * N, C, t.shape[-1]) for t in ts]
        source = lambda src: ext[src[1]] if src[0] == 'ext' else state[src[1]]
        planar_ok = (_PLANAR and _POST_AGG and Ks == 2 and k.cell_planar_supported(Ks, Tc.shape[0], C, h)
                     and k.node_post_supported(Ks, Tc.shape[0], C, 2 * h, h))
        post20 = bool(planar_ok) and k.node_post_supported(Ks, Tc.shape[0], C, 20, h)
        # order 3 (BASELINE configuration 4): planar cells on the three Chebyshev planes of each side, candidate in slab-planar form
        planar_k = bool(_PLANAR and _PLANAR_K3 and not bf16 and Ks == 3 and Tc.shape[0] == 3 and k.cell_planar_k_supported(Ks, C, h))
        # 16 + 16 columns, or (layer 0) a narrow input plane of 1..4 columns beside the 16 state columns
        planar = [bool((planar_ok and (cin[j] == h or (post20 and 1 <= cin[j] <= 4))) or (planar_k and (cin[j] == h or 1 <= cin[j] <= 4)))
                  for j in range(n_cells)]
        if bf16 and not all(planar):
            raise ValueError('stc_cell_graph: bfloat16 states need an all-planar schedule (Ks = 2, inputs 16 or 1..4 columns wide)')
        XH, agg = {}, {}

        def rows_of(j):                                             # input rows of an interleaved cell, allocated at first touch
            if j not in XH:
                L = cin[j] + h + (-(cin[j] + h)) % 4
                XH[j] = ref.new_empty(B, N, C, L)
            return XH[j]

        def aggregated(src):                                        # S.source, once per source tensor
            if src not in agg:
                t = source(src)
                out, w = torch.empty_like(t), t.shape[-1]
                k.csr_spmm(op.fwd_rowptr, op.fwd_colidx, fwd_val, N, N, t.view(B, N, C * w), None, out.view(B, N, C * w), 1.0, 0.0,
                           plan=op.fwd_plan)
                agg[src] = out
            return agg[src]

        def cheb_planes(t):                                         # [t, S.t, 2 S.(S.t) - t]: the feature-side recurrence, order 3
            w = t.shape[-1]
            v3 = lambda a: a.view(B, N, C * w)
            s1, s2 = torch.empty_like(t), torch.empty_like(t)
            if _RING2_FWD and w == h and not bf16 and op.fwd_ring2 is not None and hasattr(k, 'ring2_chain') and k.ring2_fits(B, N, C, h):
                # both aggregations in one launch: S.t for a patch's first ring is formed in LDS and aggregated from there (stc_ring2_chain_f32)
                k.ring2_chain(op.fwd_rowptr, op.fwd_colidx, fwd_val, op.fwd_ring2, t, None, 1.0, [], s1, 2.0, [(t, -1.0)], s2)
                return [t, s1, s2]
            k.csr_spmm(op.fwd_rowptr, op.fwd_colidx, fwd_val, N, N, v3(t), None, v3(s1), 1.0, 0.0, plan=op.fwd_plan)
            k.csr_spmm(op.fwd_rowptr, op.fwd_colidx, fwd_val, N, N, v3(s1), v3(t), v3(s2), 2.0, -1.0, plan=op.fwd_plan)
            return [t, s1, s2]

        def planes_of(src):                                         # once per source tensor, shared by every cell that consumes it
            if src not in agg:
                agg[src] = cheb_planes(source(src))
            return agg[src]

        # fp16 x 2 operand format: every planar forward launch leaves the maxima of its input planes in a row of slots (one zero fill per
        # forward pass); the matching backward launch scales the activation operands of its dW products by them (_lib.act_amax_buffer)
        zmax_all = k.act_amax_buffer(ref, n_cells, 2, 2 * Ks) if (not bf16 and any(planar)) else None

        def act_slots(j, which=0):                                  # which: 0 = the gates convolution's planes, 1 = the candidate's (order 3)
            return {} if zmax_all is None else dict(act_amax=zmax_all[j, which])

        state = [None] * n_cells                                    # plain (B,N,C,h) new state of every cell
        out_stack = ref.new_empty(len(outputs), B, N, C, h)         # the requested states are produced in place, stacked
        out_slot = {j: i for i, j in enumerate(outputs)}
        if len(out_slot) != len(outputs):
            raise ValueError('stc_cell_graph: duplicate output cells')
        saved, n_saved = [], []
        for j, (s_id, x, hs) in enumerate(schedule):
            Wg, bg, Wc, bc = stacks[s_id]
            Hprev = source(hs)
            # where else the new state goes: straight into the input rows of the INTERLEAVED cells that consume it
            copies, side, late_copies, late_rows = [], None, [], []
            for (d, role) in consumers[j]:
                if planar[d]:
                    continue                                        # planar consumers read the state tensor itself
                Xd = rows_of(d)
                view = Xd.view(B * N, C, Xd.shape[-1])
                if role == 'x':
                    copies.append((view, 0))
                elif schedule[d][1][0] == 'ext':                    # H part + the consumer's external X part and pad columns
                    if side is None:
                        copies.insert(0, (view, cin[d]))
                        side = ext[schedule[d][1][1]].view(B * N, C, cin[d])
                    else:
                        late_rows.append(d)
                elif Xd.shape[-1] > cin[d] + h:
                    late_rows.append(d)                             # pad columns to zero: not a case the kernel handles
                else:
                    copies.append((view, cin[d]))
            first = 1 if side is not None else 0
            while len(copies) > 2:                                  # the kernel takes two destinations; the rest by torch
                late_copies.append(copies.pop(len(copies) - 1 if len(copies) - 1 >= first else first))
            U, Rg, Cand = torch.empty_like(Hprev), torch.empty_like(Hprev), torch.empty_like(Hprev)
            Hnew = _alias_slice(out_stack, out_slot[j]) if j in out_slot else torch.empty_like(Hprev)
            if planar[j] and planar_k:
                Zx, Zh, RH = planes_of(x), planes_of(hs), torch.empty_like(Hprev)
                k.cell_gates_fwd_planar_k(rows(Zx), rows(Zh), Tc, Wg, bg, *rows((U, Rg, RH)), **act_slots(j, 0))
                Zr = cheb_planes(RH)                                # the candidate's H side: T_n(S) of R*H (its X side is Zx again)
                k.cell_cand_fwd_planar_k(rows(Zx), rows(Zr), Tc, Wc, bc, *rows((U, Hprev, Cand, Hnew)), **act_slots(j, 1))
                saved += [Hprev, U, Rg, Cand, *Zx, *Zh[1:], *Zr]
                n_saved.append(-12)                                 # negative count: planar cell (12: order 3, slab-planar candidate)
            elif planar[j]:
                Xp, SXp, SHp = source(x), aggregated(x), aggregated(hs)
                fused_post = _FUSE_POST and k.cell_planar_post_fused(C)
                # one-launch backward (stc_cell_bwd_planar_f32) forms R*H itself: with the fused projection the plane is not stored at all
                one_bwd = fused_post and (k.cell_bwd_planar_supported(C, h, cin[j]) if bf16 else k.cell_bwd_planar_supported(C, h))
                RH = None if one_bwd else torch.empty_like(Hprev)
                A, Bm = torch.empty_like(Hprev), torch.empty_like(Hprev)
                if fused_post:                                        # the candidate's projection rides in the gates launch
                    k.cell_gates_fwd_planar(*rows((Xp, Hprev, SXp, SHp)), Tc, Wg, bg, *rows((U, Rg)), None if RH is None else RH.view(B * N, C, h),
                                            post=(Wc, bc, *rows((A, Bm))), **act_slots(j))
                else:
                    k.cell_gates_fwd_planar(*rows((Xp, Hprev, SXp, SHp)), Tc, Wg, bg, *rows((U, Rg, RH)), **act_slots(j))
                    lead, second = (Xp, RH) if cin[j] == h else (RH, Xp)     # narrow input plane: the 16-wide plane leads
                    k.node_post_fwd(*rows((lead,)), Tc, Wc, bc, *rows((A, Bm)), X2=second.view(B * N, C, second.shape[-1]))
                # the blend and the aggregation of the new state in one launch where the graph has a two-ring plan and some planar cell will
                # ask for S.Hnew (stc_ring2_blend_f32: the new state is summed out of LDS instead of being read back by a launch of its own; on ring-bounded
                # clusters it measured 792 us against 548 + 203 for the two launches: tiles only)
                if (_RING2_FWD and not bf16 and op.fwd_ring2 is not None and not op.ring2_clusters and not copies and side is None and hasattr(k, 'ring2_blend') and k.ring2_fits(B, N, C, h)
                        and any(planar[d] and not planar_k for d, _ in consumers[j])):
                    SHn = torch.empty_like(Hprev)
                    k.ring2_blend(op.fwd_rowptr, op.fwd_colidx, fwd_val, op.fwd_ring2, Bm, A, U, Hprev, Cand, Hnew, SHn)
                    agg[('cell', j)] = SHn
                else:
                    k.spmm_blend_fwd(op.fwd_rowptr, op.fwd_colidx, fwd_val, op.fwd_plan, Bm, A, U, Hprev, Cand, Hnew, copies=copies, side=side)
                del A, Bm
                saved += [Hprev, U, Rg, Cand, Xp, SXp, SHp] + ([] if RH is None else [RH])
                n_saved.append(-7 if RH is None else -8)            # negative count: planar cell (-7: no R*H plane, one-launch backward)
            else:
                Xj = rows_of(j)
                L = Xj.shape[-1]
                if x[0] == 'ext' and hs[0] == 'ext':
                    k.concat2(ext[x[1]], Hprev, Xj)
                elif x[0] == 'cell' and hs[0] == 'ext':             # X part came from its producer; complete the row
                    Xj[..., cin[j]:cin[j] + h].copy_(Hprev)
                    if L > cin[j] + h:
                        Xj[..., cin[j] + h:].zero_()
                # (H part from a cell: its producer also wrote an external X part and the pad columns, see above)
                CandIn = torch.empty_like(Xj)
                Zg = _spatial_slabs(Xj, fwd_val, op, Ks)
                k.cell_gates_fwd(rows(Zg), Tc, Wg, bg, *rows((Hprev, U, Rg, CandIn)))
                post = _POST_AGG and k.node_post_supported(Ks, Tc.shape[0], C, L, h)
                if post:
                    # candidate convolution as Y = A + S.Bm: project first, aggregate C*h-float rows, blend in the SpMM's epilogue
                    Zc = [CandIn]
                    A, Bm = torch.empty_like(Hprev), torch.empty_like(Hprev)
                    k.node_post_fwd(*rows((CandIn,)), Tc, Wc, bc, *rows((A, Bm)))
                    k.spmm_blend_fwd(op.fwd_rowptr, op.fwd_colidx, fwd_val, op.fwd_plan, Bm, A, U, Hprev, Cand, Hnew, copies=copies, side=side)
                    del A, Bm
                else:
                    Zc = _spatial_slabs(CandIn, fwd_val, op, Ks)
                    k.cell_blend_fwd(rows(Zc), Tc, Wc, bc, *rows((U, Hprev, Cand, Hnew)), copies=copies, side=side)
                saved += [Hprev, U, Rg, Cand, *Zg, *Zc]
                n_saved.append(4 + len(Zg) + len(Zc))
            for buf, off in late_copies:
                buf[..., off:off + h].copy_(Hnew.view(B * N, C, h))
            for d in late_rows:
                Xd, xs = rows_of(d), schedule[d][1]
                if xs[0] == 'ext':
                    k.concat2(ext[xs[1]], Hnew, Xd)
                else:
                    Xd[..., cin[d]:cin[d] + h].copy_(Hnew)
                    Xd[..., cin[d] + h:].zero_()
            state[j] = Hnew
        ctx.save_for_backward(Tc, *[p for st in stacks for p in st if p is not None], *saved)
        ctx.meta = (op, Ks, schedule, tuple(outputs), cin, [tuple(p is not None for p in st) for st in stacks], (B, N, C), n_saved)
        ctx.zmax_all = zmax_all
        # the saved states of the output cells ALIAS out_stack's storage without sharing its autograd version counter: the
        # returned stack is read-only for its consumers; its version is checked again in backward
        ctx.out_stack_ref, ctx.out_stack_version = weakref.ref(out_stack), out_stack._version      # (weak: no output -> ctx -> output cycle)
        return out_stack

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_stack):
        k = kernels().for_graph(_amplification(ctx.meta[0], ctx.meta[1]))
        bf16_planes = grad_stack.dtype == torch.bfloat16
        if bf16_planes:
            k = k.bf16
        op, Ks, schedule, outputs, cin, present, (B, N, C), n_saved = ctx.meta
        stack = ctx.out_stack_ref()
        if stack is not None and stack._version != ctx.out_stack_version:
            raise RuntimeError('stc_cell_graph: the returned state stack was modified in place after the forward pass; the states saved for '
                               'backward share its storage (treat the stack as read-only, or clone it before editing)')
        sv = list(ctx.saved_tensors)
        Tc = sv.pop(0)
        stacks = []
        for pres in present:
            st = [sv.pop(0) if p else None for p in pres]
            stacks.append(st)
        cells, at = [], 0
        for cnt in n_saved:
            cells.append(sv[at:at + abs(cnt)])
            at += abs(cnt)
        h = 16
        rows = lambda ts: [t.view(B * N, C, t.shape[-1]) for t in ts]
        bwd = (op.bwd_rowptr, op.bwd_colidx, op.bwd_val)
        G = {}                                                       # cell -> gradient its state is owed so far
        grad_stack = _c(grad_stack)
        for i, j in enumerate(outputs):
            G[j] = grad_stack[i]                                     # read-only here: sums go to fresh buffers
        acc = [[None] * 4 for _ in stacks]
        # Parameter gradients of the planar cells: every cell writes its (dWg, dbg, dWc, dbc) into its own row of ONE buffer per parameter
        # set, summed once at the end -- instead of four accumulation passes per cell (176 five-microsecond launches per metric step).
        rows_of_set = {}

        def grads_for(s_id):
            Wg_, bg_, Wc_, bc_ = stacks[s_id]
            sizes = [Wg_.numel(), 0 if bg_ is None else 2 * h, Wc_.numel(), 0 if bc_ is None else h]
            if s_id not in rows_of_set:
                n = sum(1 for sc in schedule if sc[0] == s_id)
                rows_of_set[s_id] = [Wg_.new_zeros(n, sum(sizes)), 0, sizes]
            buf, i, _ = rows_of_set[s_id]
            rows_of_set[s_id][1] = i + 1
            parts = buf[i].split(sizes)
            return (parts[0].view_as(Wg_), None if bg_ is None else parts[1], parts[2].view_as(Wc_), None if bc_ is None else parts[3])

        def add_to(slot, i, t):
            if t is not None:
                slot[i] = t if slot[i] is None else slot[i].add_(t)

        def narrow_transpose_aggregation(dY):                        # dBm = S^T dY on rows of C*h floats
            dBm = torch.empty_like(dY)
            k.csr_spmm(*bwd, N, N, dY.view(B, N, C * h), None, dBm.view(B, N, C * h), 1.0, 0.0, plan=op.bwd_plan)
            return dBm

        # Planar consumers leave PIECES of a state's gradient instead of a finished tensor: direct planes (what the state
        # is owed as a plane of their inputs) and aggregated planes (what its aggregation S.state is owed).  Aggregation
        # being linear, the state's gradient is  sum(direct) + S^T sum(aggregated): ONE narrow SpMM per state with the
        # sums in its gather / epilogue -- instead of a wide transpose SpMM and a split pass per consuming cell.
        pieces = {}

        def leave(kid, direct, aggregated):
            pc = pieces.setdefault(kid, dict(direct=[], agg=[]))
            pc['direct'] += [(t, 0) for t in direct]
            pc['agg'].append(aggregated)

        # fp16 x 2 operand format: the backward launches scale the activation operands of their dW products by the plane maxima the forward
        # launches left (gradient scales they find themselves, per node)
        zmax_all = ctx.zmax_all

        def act_slots(j, which=0):                                   # what cell j's forward launches left
            return {} if zmax_all is None else dict(act_amax=zmax_all[j, which])

        def owed(kid, blend=None):
            """The gradient of state ``kid``; with ``blend`` = (U, Cand) of its cell also dY = gradient * U * (1 - Cand^2)."""
            base = G.pop(kid, None)                                  # from interleaved consumers / the outputs: a finished tensor
            pc = pieces.pop(kid, None)
            if pc is None:
                if blend is None:
                    return base
                dY = torch.empty_like(base)
                k.gru_blend_bwd(base, blend[0], None, blend[1], dY, None, None)
                return base, dY
            add = pc['direct'] + ([(base, 0)] if base is not None else [])
            aggs = pc['agg']
            while len(add) > 5:                                      # more consumers than the kernel takes addends for: pre-sum
                (a, ao), (b_, bo) = add.pop(), add.pop()
                add.append((a[..., ao:ao + h] + b_[..., bo:bo + h], 0))
            while len(aggs) > 2:
                aggs = [aggs[0] + aggs[1]] + aggs[2:]
            out = aggs[0].new_empty(B, N, C, h)
            dY = torch.empty_like(out) if blend is not None else None
            k.spmm_sum(*bwd, op.bwd_plan, aggs[0], aggs[1] if len(aggs) > 1 else None, add, out,
                       blend=None if blend is None else (blend[0], blend[1], dY))
            return out if blend is None else (out, dY)

        def owed_ring2(kid, U_, Cand_):
            """(gradient of state ``kid``, S^T (gradient * U * (1 - Cand^2))) in one launch where the graph has a two-ring plan and the state's
            pieces are whole planes; None: the two launches (``owed`` with its blend epilogue, then the narrow aggregation)."""
            pc = pieces.get(kid)
            if pc is None or op.bwd_ring2 is None or not hasattr(k, 'ring2_sum') or not k.ring2_fits(B, N, C, h):
                return None
            base = G.get(kid)
            add = [t for t, off in pc['direct'] if off == 0 and t.shape[-1] == h] + ([base] if base is not None else [])
            # (the forms that fit the register file: one aggregated plane, up to two addends -- 590 / 680 us for the 555 + 185 they replace; with a
            #  second aggregated plane the kernel spills and takes 840 - 1 150 us: the two launches stay)
            if len(add) != len(pc['direct']) + (base is not None) or len(add) > 2 or len(pc['agg']) != 1:
                return None
            G.pop(kid, None)
            pieces.pop(kid)
            aggs = pc['agg']
            out, dBm_ = aggs[0].new_empty(B, N, C, h), aggs[0].new_empty(B, N, C, h)
            k.ring2_sum(*bwd, op.bwd_ring2, aggs[0], aggs[1] if len(aggs) > 1 else None, add, U_, Cand_, out, dBm_)
            return out, dBm_

        # Order 3: a consumer leaves direct planes d0 and the gradients d1, d2 of the S / T_2(S) planes; the source's gradient is
        #   sum d0 - sum d2 + S^T (sum d1 + 2 S^T sum d2)          (Clenshaw form of sum_n T_n(S)^T d_n)
        # = two narrow SpMMs with the sums in their gather / epilogue (alpha and signed addends of stc_spmm_sum_f32).
        def leave3(kid, d0, d1, d2):
            pc = pieces.setdefault(kid, dict(d0=[], d1=[], d2=[]))
            pc['d0'] += list(d0); pc['d1'] += list(d1); pc['d2'] += list(d2)

        def clenshaw(d0, d1, d2, blend=None):
            """sum d0 - sum d2 + S^T (sum d1 + 2 S^T sum d2) from lists of planes (d2 non-empty); with ``blend`` = (U, Cand) also
            dY = result * U * (1 - Cand^2) from the second launch's epilogue."""
            while len(d2) > 2:                                       # the kernel gathers two operands: pre-sum the rest
                d2 = [d2[0] + d2[1]] + d2[2:]
            if (_RING2 and blend is None and not bf16_planes and op.bwd_ring2 is not None and hasattr(k, 'ring2_chain') and k.ring2_fits(B, N, C, h) and len(d1) <= 2
                    and 1 <= len(d0) + len(d2) <= k.RING2_MAX_ADD):
                # both transpose aggregations in one launch (stc_ring2_chain_f32): the inner sum d1 + 2 S^T d2 never leaves the chip
                out = d2[0].new_empty(B, N, C, h)
                k.ring2_chain(*bwd, op.bwd_ring2, d2[0], d2[1] if len(d2) > 1 else None, 2.0, list(d1), None, 1.0,
                              [(a, 1.0) for a in d0] + [(a, -1.0) for a in d2], out)
                return out
            t = d2[0].new_empty(B, N, C, h)
            k.spmm_sum(*bwd, op.bwd_plan, d2[0], d2[1] if len(d2) > 1 else None, [(a, 0) for a in d1], t, alpha=2.0)
            adds = [(a, 0) for a in d0] + [(a, 0, -1.0) for a in d2]
            while len(adds) > 8:
                (a, _), (b_, _) = adds.pop(0), adds.pop(0)
                adds.insert(0, (a + b_, 0))
            out = t.new_empty(B, N, C, h)
            dY = torch.empty_like(out) if blend is not None else None
            k.spmm_sum(*bwd, op.bwd_plan, t, None, adds, out, blend=None if blend is None else (blend[0], blend[1], dY))
            return out if blend is None else (out, dY)

        def owed3(kid, blend=None):
            base = G.pop(kid, None)
            pc = pieces.pop(kid, None)
            if pc is None:
                if blend is None:
                    return base
                dY = torch.empty_like(base)
                k.gru_blend_bwd(base, blend[0], None, blend[1], dY, None, None)
                return base, dY
            return clenshaw(pc['d0'] + ([base] if base is not None else []), pc['d1'], pc['d2'], blend)

        for j in range(len(schedule) - 1, -1, -1):
            if j not in G and j not in pieces:
                continue                                             # nothing downstream depends on this cell
            s_id, x, hs = schedule[j]
            Wg, bg, Wc, bc = stacks[s_id]
            Hprev, U, Rg, Cand, *rest = cells[j]
            if n_saved[j] == -12:                                    # order-3 planar cell
                Zx, Zh = rest[:3], [Hprev] + rest[3:5]
                wide = cin[j] == h
                new = lambda: torch.empty_like(Hprev)
                dWg, dbg, dWc, dbc = grads_for(s_id)
                Zr = rest[5:8]                                       # slab-planar candidate
                dHnew = owed3(j)
                dXc, dR = ([new(), new(), new()] if wide else [None] * 3), [new(), new(), new()]
                k.cell_cand_bwd_planar_k(rows(Zx), rows(Zr), Tc, Wc, *rows((dHnew, U, Cand)),
                                         [None if t is None else t.view(B * N, C, h) for t in dXc], rows(dR), dWc, dbc, **act_slots(j, 1))
                # gradient of the R*H plane from its three Chebyshev planes
                dRH = clenshaw([dR[0]], [dR[1]], [dR[2]])
                del dR
                fold = getattr(k, 'folds_dH', False)                  # the kernel adds the prologue's share into the H plane's gradient
                # slab-planar candidate on a wide input: the gates' X-side gradients are ADDED into the candidate's three planes by the
                # kernel (accumulate_x), so the source gets one plane per order from this cell and its Clenshaw sums need no pre-sum
                into = fold and wide
                dXg = dXc if into else ([new(), new(), new()] if wide else [None] * 3)
                dHg, dH = [new(), new(), new()], (None if fold else new())
                k.cell_gates_bwd_planar_k(rows(Zx), rows(Zh), Tc, Wg, *rows((dRH, Cand, U, Rg, dHnew)),
                                          [None if t is None else t.view(B * N, C, h) for t in dXg], rows(dHg), dWg, dbg,
                                          None if fold else dH.view(B * N, C, h), accumulate_x=into, **act_slots(j, 0))
                if into:
                    dXc = [None] * 3
                if wide and x[0] == 'cell':
                    leave3(x[1], [t for t in (dXg[0], dXc[0]) if t is not None], [t for t in (dXg[1], dXc[1]) if t is not None],
                           [t for t in (dXg[2], dXc[2]) if t is not None])
                if hs[0] == 'cell':
                    leave3(hs[1], (dHg[0],) if fold else (dHg[0], dH), (dHg[1],), (dHg[2],))
                continue                                             # (parameter gradients: rows of the set's buffer, summed at the end)
            post_form = n_saved[j] < 0 or (len(rest) == Ks + 1 and Ks > 1)     # candidate backward starts from dY = dHnew * U * (1 - Cand^2)
            if Ks == 3 and j in pieces:                              # an interleaved cell whose state order-3 planar cells consumed
                G[j] = owed3(j)
            dBm = None
            if n_saved[j] == -7 and _RING2 and not bf16_planes:     # state gradient + S^T dY in one launch, no dY plane (stc_ring2_sum_f32)
                fused = owed_ring2(j, U, Cand)
                if fused is not None:
                    dHnew, dBm = fused
                    dY = None
            if dBm is None:
                dHnew, dY = owed(j, (U, Cand)) if post_form else (owed(j), None)
            dH = None if (n_saved[j] == -7 or (n_saved[j] < 0 and getattr(k, 'folds_dH', False))) else torch.empty_like(Hprev)
            if n_saved[j] == -7:                                     # planar cell, candidate + gates backward in ONE launch
                Xp, SXp, SHp = rest
                if dBm is None:
                    dBm = narrow_transpose_aggregation(dY)
                del dY                                               # (the kernel re-forms dY from dHnew, U, Cand)
                wide = cin[j] == h
                new = lambda: torch.empty_like(Hprev)
                dWg, dbg, dWc, dbc = grads_for(s_id)
                # A state has two consumers (next step as H, next layer as X): the first one processed writes the state's direct and
                # aggregated gradient planes, the second ADDS into them (accumulate_x / accumulate_h), so the state-gradient SpMM gathers
                # one operand instead of two and reads one direct plane instead of two.
                taken = set()

                def planes_of_state(src):
                    if src[0] != 'cell':
                        return new(), new(), False                   # an external tensor: gradients computed, nobody owed
                    pc = pieces.get(src[1])
                    if _ACC_PLANES and not bf16_planes and pc is not None and pc.get('own') is not None and src[1] not in taken:
                        taken.add(src[1])
                        return pc['own'][0], pc['own'][1], True
                    d, a_ = new(), new()
                    leave(src[1], (d,), a_)
                    if pieces[src[1]].get('own') is None:
                        pieces[src[1]]['own'] = (d, a_)
                        taken.add(src[1])
                    return d, a_, False

                dXd, dSX, acc_x = planes_of_state(x) if wide else (None, None, False)
                dHd, dSH, acc_h = planes_of_state(hs)
                k.cell_bwd_planar(*rows((Xp, Hprev, SXp, SHp)), Tc, Wg, Wc, *rows((U, Rg, Cand, dHnew, dBm)),
                                  [None if t is None else t.view(B * N, C, h) for t in (dXd, dSX, dHd, dSH)], dWg, dbg, dWc, dbc,
                                  **(dict(accumulate_x=acc_x, accumulate_h=acc_h, **act_slots(j)) if not bf16_planes else {}))
                continue                                             # (parameter gradients: rows of the set's buffer, summed at the end)
            if n_saved[j] < 0:                                       # planar cell: inputs and gradients as planes
                Xp, SXp, SHp, RH = rest
                wide = cin[j] == h                                   # else: narrow input plane (layer 0), which needs no gradient
                post_kw, gates_kw = {}, {}
                if zmax_all is not None:
                    # the candidate's input planes are (X, R*H): X's maximum as the gates forward left it, R*H rides on H's (|R*H| <= |H|).  Slot
                    # rows of that launch: wide {X, S.X, H, S.H}, narrow {H, S.H, x, S.x}; the post kernel takes (16-wide plane, other plane).
                    zr = zmax_all[j, 0]
                    post_kw.update(act_amax=(zr[0], zr[2]))
                    gates_kw.update(act_amax=zr)
                dBm = narrow_transpose_aggregation(dY)
                dRH = torch.empty_like(Hprev)
                dWg, dbg, dWc, dbc = grads_for(s_id)
                dHd, dSH = torch.empty_like(Hprev), torch.empty_like(Hprev)
                if wide:
                    dXc, dXd, dSX = (torch.empty_like(Hprev) for _ in range(3))
                    k.node_post_bwd(*rows((Xp,)), Tc, Wc, *rows((dY, dBm, dXc)), dWc, dbc, X2=RH.view(B * N, C, h), dX2=dRH.view(B * N, C, h), **post_kw)
                    planes = rows((dXd, dSX, dHd, dSH))
                else:
                    k.node_post_bwd(*rows((RH,)), Tc, Wc, *rows((dY, dBm, dRH)), dWc, dbc, X2=Xp.view(B * N, C, cin[j]), **post_kw)
                    planes = [None, None] + rows((dHd, dSH))
                del dY, dBm
                fold = getattr(k, 'folds_dH', False)                 # the kernel adds the prologue's share into the H plane's gradient
                k.cell_gates_bwd_planar(*rows((Xp, Hprev, SXp, SHp)), Tc, Wg, *rows((dRH, Cand, U, Rg, dHnew)), planes, dWg, dbg,
                                        None if fold else dH.view(B * N, C, h), **gates_kw)
                if wide and x[0] == 'cell':
                    leave(x[1], (dXd, dXc), dSX)                     # as the X plane: gates' and candidate's direct shares
                if hs[0] == 'cell':
                    leave(hs[1], (dHd,) if fold else (dHd, dH), dSH)  # as the H plane: direct share + what the gates prologue owes it
                continue                                             # (parameter gradients: rows of the set's buffer, summed at the end)
            else:
                Zg, Zc = rest[:Ks], rest[Ks:]
                L = Zc[0].shape[-1]
                if len(Zc) == 1 and Ks > 1:                         # the forward ran this convolution as Y = A + S.Bm (no Z_1 slab)
                    dBm = narrow_transpose_aggregation(dY)
                    dci, dWc = torch.empty_like(Zc[0]), torch.empty_like(Wc)
                    dbc = Wc.new_empty(h) if bc is not None else None
                    k.node_post_bwd(*rows((Zc[0],)), Tc, Wc, *rows((dY, dBm, dci)), dWc, dbc)
                else:                                               # slab form, blend backward in the node kernel's prologue
                    g, dWc, dbc, _, _ = _bdg_backward_slabs(None, Zc, Wc, Tc, op, Ks, bc is not None, False, False, cand=(dHnew, U, Cand))
                    if Ks > 1:
                        k.csr_spmm(*bwd, N, N, g[1].view(B, N, C * L), g[0].view(B, N, C * L), g[0].view(B, N, C * L), 1.0, 1.0, plan=op.bwd_plan)
                    dci = g[0]
                # gates convolution, gate + blend backward in its prologue
                g, dWg, dbg, _, _ = _bdg_backward_slabs(None, Zg, Wg, Tc, op, Ks, bg is not None, False, False,
                                                        gates=(dci, Cand, Hprev, U, Rg, dHnew, dH))
            v3 = lambda t: t.view(B, N, C * L)
            need_x, need_h = x[0] == 'cell', hs[0] == 'cell'
            if need_x or need_h:
                if Ks > 1:
                    k.csr_spmm(*bwd, N, N, v3(g[1]), v3(g[0]), v3(g[0]), 1.0, 1.0, plan=op.bwd_plan)
                dXt = Hprev.new_empty(Hprev.shape[:-1] + (cin[j],))
                same = need_x and need_h and x[1] == hs[1]
                owedA = G.get(x[1]) if need_x else None
                owedB = G.get(hs[1]) if (need_h and not same) else None
                k.split2(g[0], dXt, dH, addA=dci, addB=dH, addA_ld=L, addA2=owedA, addB2=owedB)
                if same:
                    G[x[1]] = dXt.add_(dH)
                else:
                    if need_x:
                        G[x[1]] = dXt
                    if need_h:
                        G[hs[1]] = dH
            for i, t in enumerate((dWg, dbg, dWc, dbc)):
                add_to(acc[s_id], i, t)
        for s_id, (buf, used, sizes) in rows_of_set.items():
            Wg_, bg_, Wc_, bc_ = stacks[s_id]
            parts = (buf[0] if buf.shape[0] == 1 else buf.sum(0)).split(sizes)
            for i, t in enumerate((parts[0].view_as(Wg_), None if bg_ is None else parts[1], parts[2].view_as(Wc_), None if bc_ is None else parts[3])):
                add_to(acc[s_id], i, t)
        flat = []
        for st, a in zip(stacks, acc):
            for p, gsum in zip(st, a):
                flat.append(None if p is None else (gsum if gsum is not None else torch.zeros_like(p)))
        n_ext = len(ctx.needs_input_grad) - 7 - len(flat)
        return (None,) * 7 + (None,) * n_ext + tuple(flat)


def stc_cell_graph(op: SpatialOperand, Tc, Ks: int, schedule, outputs, ext, stacks):
    """Run a schedule of STC_Cells (see ``_StcCellGraph``).  ``ext``: external (B,N,C,*) tensors (inputs, initial states;
    they get no gradient); ``stacks``: [(Wg, bg, Wc, bc)] parameter sets; returns the new states of the ``outputs`` cells
    stacked along a new leading axis, (len(outputs), B, N, C, h) -- written in place by the kernels, no stack pass."""
    k = kernels()
    if _SMALL and small_graph_supported(k, op, Tc, Ks, ext[0].shape[2], 16, {ext[x[1]].shape[-1] if x[0] == 'ext' else 16 for _, x, _ in schedule},
                                        ext[0].dtype):
        return stc_small_graph(k, op, Tc, Ks, schedule, outputs, ext, stacks)
    flat = [p for st in stacks for p in st]
    return _StcCellGraph.apply(op, Ks, list(schedule), list(outputs), len(ext), Tc, op.fwd_val, *ext, *flat)


# ---- MixedFusion of the learned graph generator ---------------------------------------------------------------------------------------
class _MixedFusion(torch.autograd.Function):
    """G = gate * A + (1 - gate) * P with gate = sigmoid(lin_A(vec A) + lin_P(vec P)) (reference STC_GNN.py:253-260) in one streaming launch per
    direction (``stc_mixed_fusion_fwd/bwd_f32``): the two (n^2, n^2) weight matrices are read once forward and once backward, their
    gradients written once."""

    @staticmethod
    def forward(ctx, A, P, WA, bA, WP, bP):
        k = kernels()
        A_, P_ = A.detach().contiguous(), P.detach().contiguous()
        gate, G = k.mixed_fusion_fwd(WA.detach(), bA.detach(), WP.detach(), bP.detach(), A_, P_)
        ctx.save_for_backward(A_, P_, WA, WP, gate)
        return G

    @staticmethod
    def backward(ctx, dG):
        A, P, WA, WP, gate = ctx.saved_tensors
        want_dA = ctx.needs_input_grad[0]
        need = ctx.needs_input_grad
        dWA, dWP, db, dP, dA = kernels().mixed_fusion_bwd(WA.detach(), WP.detach(), A, P, gate, dG.contiguous(), want_dA, want_dW=need[2] or need[4])
        return (dA if need[0] else None, dP if need[1] else None, dWA if need[2] else None, db if need[3] else None,
                dWP if need[4] else None, db if need[5] else None)


def mixed_fusion_supported(A: torch.Tensor, P: torch.Tensor, *params: torch.Tensor) -> bool:
    """Whether ``mixed_fusion`` takes these operands: float32 on a GPU, contiguous weights, 16-byte aligned, n^2 a multiple of 4."""
    ts = (A, P) + params
    return (all(t.is_cuda and t.dtype == torch.float32 for t in ts) and all(t.is_contiguous() for t in params)
            and all(t.data_ptr() % 16 == 0 for t in ts)               # (a slice / a view into a flat parameter buffer may not be: torch ops then)
            and kernels().mixed_fusion_supported(A.numel()))


def mixed_fusion(A, P, WA, bA, WP, bP):
    return _MixedFusion.apply(A, P, WA, bA, WP, bP)


# ---- front end of the learned graph generator --------------------------------------------------------------------------------------------
class _MgpUV(torch.autograd.Function):
    """U, V = tanh(alpha X Wu), tanh(alpha X Wv) (reference STC_GNN.py:229-230 / :237-238) in (rows, slices, hidden) layout, one launch; the data
    window X gets no gradient."""

    @staticmethod
    def forward(ctx, X, Wu, Wv, rows_axis, alpha):
        X_ = X.detach().contiguous()
        U, V = kernels().mgp_uv_fwd(X_, rows_axis, Wu.detach().contiguous(), Wv.detach().contiguous(), alpha)
        ctx.save_for_backward(X_, U, V)
        ctx.meta = (rows_axis, alpha)
        return U, V

    @staticmethod
    @once_differentiable
    def backward(ctx, dU, dV):
        X_, U, V = ctx.saved_tensors
        dWu, dWv = kernels().mgp_uv_bwd(X_, ctx.meta[0], U, V, dU.contiguous(), dV.contiguous(), ctx.meta[1])
        return None, dWu, dWv, None, None


class _MgpSoftmax(torch.autograd.Function):
    """softmax(relu(P - P^T), -1) (reference STC_GNN.py:231-232: the einsum pair is P - P^T) in one launch per direction."""

    @staticmethod
    def forward(ctx, P):
        P_ = P.detach().contiguous()
        Ps = kernels().mgp_softmax_fwd(P_)
        ctx.save_for_backward(P_, Ps)
        return Ps

    @staticmethod
    @once_differentiable
    def backward(ctx, dPs):
        P_, Ps = ctx.saved_tensors
        return kernels().mgp_softmax_bwd(P_, Ps, dPs.contiguous())


def mgp_front_supported(X: torch.Tensor, Wu: torch.Tensor, Wv: torch.Tensor) -> bool:
    """Whether ``mgp_front`` takes these operands: a float32 window (B, T, N, C) and float32 (F, h) parameters on a GPU, no gradient wanted for X."""
    return (X.is_cuda and X.dim() == 4 and all(t.is_cuda and t.dtype == torch.float32 for t in (X, Wu, Wv)) and not X.requires_grad
            and Wu.dim() == 2 and Wu.shape == Wv.shape and hasattr(kernels(), 'mgp_uv_fwd'))


def mgp_front(X, Wu, Wv, rows_axis: int, alpha: float, reduce=None):
    """The learned graph of one branch of ``MGP_Gen.forward``: softmax(relu(P - P^T)) with P = sum over (sample, time) of U V^T.  ``reduce``: applied
    to P before the softmax (the batch-sum all-reduce of a sharded batch)."""
    U, V = _MgpUV.apply(X, Wu, Wv, rows_axis, alpha)
    P = U.flatten(1) @ V.flatten(1).t()                                # one plain GEMM over the (slice, hidden) axis; its autograd is two more
    if reduce is not None:
        P = reduce(P)
    return _MgpSoftmax.apply(P)
