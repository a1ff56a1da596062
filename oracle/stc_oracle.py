"""CPU oracle for the STC-GNN hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, as plain functions over torch CPU tensors, the arithmetic
of the reference's multi-graph message-passing path so that the HIP kernels
have something to be checked against.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product (``stc-gnn_amd/``) never does.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
(``/root/reference/framework/STC_GNN.py``) in the build container and stores
its inputs/outputs/gradients as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function below against them (and, when the reference directory is
present, against the live reference on random shapes).

Every function follows the reference op for op (same contraction order, same
concat order, matrix-side Chebyshev recurrence) so that its rounding behaviour
is the reference's.  The differences are deliberate and listed here:

* dtype-generic: the identity in ``cheby_poly`` follows ``G.dtype`` (the
  reference builds a float32 ``eye`` whatever the input, STC_GNN.py:26, so its
  fp64 run crashes).  With float32 inputs the result is bit-identical.
* functional: parameters come from a flat ``state_dict`` with the reference's
  keys (``encoder.cell_list.0.gates.W`` ...), there are no ``nn.Module``s.
* ``bdg_dif_sparse`` is an ADDITION (no reference counterpart): the same
  layer with the spatial graph given as a sparse matrix and the Chebyshev
  recurrence applied on the feature side.  It exists for sizes where the
  reference's dense N x N formulation cannot be run (N = 50 176) and is
  itself checked against ``bdg_dif`` on the dense form of the same matrix.

Shapes use the reference's names: B batch, T time, N nodes, C categories,
L = in + hidden feature width, Ho output width, K Chebyshev order.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# BDG_Dif  (reference: framework/STC_GNN.py:5-47)
# --------------------------------------------------------------------------
def cheby_poly(G: Tensor, order: int) -> List[Tensor]:
    """[T_0, T_1, ...] with T_0 = I, T_1 = G, T_k = (2G) T_{k-1} - T_{k-2}.

    Reference: STC_GNN.py:24-29.  The list always has at least two entries;
    callers use only the first ``order`` of them (order 1 uses I alone).
    """
    eye = torch.eye(G.shape[0], dtype=G.dtype, device=G.device)
    polys = [eye, G]
    for _ in range(2, order):
        polys.append(torch.mm(2 * G, polys[-1]) - polys[-2])
    return polys


def bdg_dif(X: Tensor, Gs: Tensor, Gc: Tensor, W: Tensor, b: Optional[Tensor],
            Ks: int, Kc: int) -> Tensor:
    """Bi-dimensional graph diffusion convolution, dense graphs.

    Reference: STC_GNN.py:31-47.  ``X`` (B,N,C,L); ``Gs`` (N,N); ``Gc`` (C,C);
    ``W`` (Ks*Kc*L, Ho) with row blocks ordered n-major, c-minor
    (STC_GNN.py:35-41); ``b`` (Ho,) or None.  The 1-mode product is the
    TRANSPOSED aggregation ``T_n(Gs)^T . X[b]`` (einsum 'bncl,nm->bmcl').
    """
    Ts = cheby_poly(Gs, Ks)
    Tc = cheby_poly(Gc, Kc)
    feats = []
    for n in range(Ks):
        for c in range(Kc):
            mode1 = torch.einsum('bncl,nm->bmcl', X, Ts[n])
            mode2 = torch.einsum('bmcl,cd->bmdl', mode1, Tc[c])
            feats.append(mode2)
    stacked = torch.cat(feats, dim=-1)
    out = torch.einsum('bmdk,kh->bmdh', stacked, W)
    if b is not None:
        out = out + b
    return out


def bdg_dif_sparse(X: Tensor, GsT_sparse: Tensor, Gc: Tensor, W: Tensor,
                   b: Optional[Tensor], Ks: int, Kc: int) -> Tensor:
    """Same layer, spatial graph sparse, Chebyshev applied to the features.

    NOT in the reference (which only runs dense graphs).
    ``GsT_sparse`` is the sparse form of ``Gs^T`` (N,N): Z_1 = Gs^T X,
    Z_k = 2 Gs^T Z_{k-1} - Z_{k-2}.  Algebraically equal to ``bdg_dif`` on the
    dense matrix because polynomials of Gs commute with Gs.
    """
    B, N, C, L = X.shape
    flat = X.permute(1, 0, 2, 3).reshape(N, B * C * L)
    zs = [flat]
    if Ks > 1:
        zs.append(torch.sparse.mm(GsT_sparse, flat))
    for _ in range(2, Ks):
        zs.append(2 * torch.sparse.mm(GsT_sparse, zs[-1]) - zs[-2])
    Tc = cheby_poly(Gc, Kc)
    feats = []
    for n in range(Ks):
        Zn = zs[n].reshape(N, B, C, L).permute(1, 0, 2, 3)
        for c in range(Kc):
            feats.append(torch.einsum('bmcl,cd->bmdl', Zn, Tc[c]))
    stacked = torch.cat(feats, dim=-1)
    out = torch.einsum('bmdk,kh->bmdh', stacked, W)
    if b is not None:
        out = out + b
    return out


# --------------------------------------------------------------------------
# STC_Cell  (reference: framework/STC_GNN.py:51-79)
# --------------------------------------------------------------------------
def stc_cell(Gs: Tensor, Gc: Tensor, Xt: Tensor, Ht_1: Tensor,
             gates_W: Tensor, gates_b: Optional[Tensor],
             candi_W: Tensor, candi_b: Optional[Tensor],
             Ks: int, Kc: int, conv=bdg_dif) -> Tensor:
    """GRU-style co-evolution cell.  Reference: STC_GNN.py:65-79.

    gates = BDG(cat[Xt, Ht_1]) -> split into (update, reset) pre-activations,
    candidate = tanh(BDG(cat[Xt, reset * Ht_1])),
    Ht = (1 - update) * Ht_1 + update * candidate.
    """
    assert Xt.dim() == 4 and Ht_1.dim() == 4
    hidden = Ht_1.shape[-1]
    xh = torch.cat([Xt, Ht_1], dim=-1)
    pre = conv(xh, Gs, Gc, gates_W, gates_b, Ks, Kc)
    u_pre, r_pre = torch.split(pre, hidden, dim=-1)
    update = torch.sigmoid(u_pre)
    reset = torch.sigmoid(r_pre)
    cand_in = torch.cat([Xt, reset * Ht_1], dim=-1)
    cand = torch.tanh(conv(cand_in, Gs, Gc, candi_W, candi_b, Ks, Kc))
    return (1.0 - update) * Ht_1 + update * cand


def _cell_params(sd: Dict[str, Tensor], prefix: str):
    return (sd[prefix + '.gates.W'], sd.get(prefix + '.gates.b'),
            sd[prefix + '.candi.W'], sd.get(prefix + '.candi.b'))


# --------------------------------------------------------------------------
# STC_Encoder / STC_Decoder  (reference: STC_GNN.py:83-135, 139-172)
# --------------------------------------------------------------------------
def stc_encoder(Gs: Tensor, Gc: Tensor, X_seq: Tensor, sd: Dict[str, Tensor],
                prefix: str, num_layers: int, Ks: int, Kc: int, hidden: int,
                H0: Optional[Sequence[Tensor]] = None,
                return_all_layers: bool = True, conv=bdg_dif
                ) -> Tuple[List[Tensor], List[Tensor]]:
    """Layer-major, then time.  Reference: STC_GNN.py:97-123.

    ``X_seq`` (B,T,N,C,in).  Layer l consumes the stacked outputs of layer
    l-1; initial states are zeros (STC_GNN.py:60-63, 125-129).
    """
    assert X_seq.dim() == 5
    B, T, N, C, _ = X_seq.shape
    if H0 is None:
        H0 = [X_seq.new_zeros(B, N, C, hidden) for _ in range(num_layers)]
    seqs, lasts = [], []
    cur = X_seq
    for l in range(num_layers):
        gw, gb, cw, cb = _cell_params(sd, f'{prefix}.cell_list.{l}')
        h = H0[l]
        outs = []
        for t in range(T):
            h = stc_cell(Gs, Gc, cur[:, t], h, gw, gb, cw, cb, Ks, Kc, conv=conv)
            outs.append(h)
        cur = torch.stack(outs, dim=1)
        seqs.append(cur)
        lasts.append(h)
    if not return_all_layers:
        seqs, lasts = seqs[-1:], lasts[-1:]
    return seqs, lasts


def stc_decoder(Gs: Tensor, Gc: Tensor, Xt: Tensor, H0: Sequence[Tensor],
                sd: Dict[str, Tensor], prefix: str, num_layers: int,
                Ks: int, Kc: int, conv=bdg_dif) -> Tuple[Tensor, List[Tensor]]:
    """One step through the layer stack.  Reference: STC_GNN.py:154-166."""
    assert Xt.dim() == 4
    states = []
    cur = Xt
    for l in range(num_layers):
        gw, gb, cw, cb = _cell_params(sd, f'{prefix}.cell_list.{l}')
        cur = stc_cell(Gs, Gc, cur, H0[l], gw, gb, cw, cb, Ks, Kc, conv=conv)
        states.append(cur)
    return cur, states


# --------------------------------------------------------------------------
# MGP_Gen / MixedFusion  (reference: STC_GNN.py:210-243, 246-261)
# --------------------------------------------------------------------------
def mixed_fusion(A: Tensor, P: Tensor, wA: Tensor, bA: Tensor, wP: Tensor,
                 bP: Tensor) -> Tensor:
    """G = a*A + (1-a)*P, a = sigmoid(Lin_A(vec A) + Lin_P(vec P)).

    Reference: STC_GNN.py:253-261.
    """
    assert A.dim() == 2 and P.dim() == 2
    n = A.shape[0]
    gate = torch.sigmoid(torch.nn.functional.linear(A.reshape(n * n), wA, bA)
                         + torch.nn.functional.linear(P.reshape(n * n), wP, bP))
    gate = gate.reshape(n, n)
    return gate * A + (1 - gate) * P


def _antisym_softmax(U: Tensor, V: Tensor) -> Tensor:
    """softmax(relu(U V^T - V U^T)) summed over batch and time (STC_GNN.py:231-232)."""
    P = torch.einsum('btnh,btmh->nm', U, V) - torch.einsum('btmh,btnh->mn', V, U)
    return torch.softmax(torch.relu(P), dim=-1)


def mgp_gen(X_seq: Tensor, As: Tensor, Ac: Tensor, sd: Dict[str, Tensor],
            prefix: str = 'mix_graph_pair', alpha: float = 3.0
            ) -> Tuple[Tensor, Tensor]:
    """Learned mixed graph pair (Gs, Gc).  Reference: STC_GNN.py:227-243.

    Batch-coupled: the pre-activation sums over b and t.
    """
    p = prefix
    Us = torch.tanh(alpha * torch.matmul(X_seq, sd[f'{p}.params_S.Wu']))
    Vs = torch.tanh(alpha * torch.matmul(X_seq, sd[f'{p}.params_S.Wv']))
    Ps = _antisym_softmax(Us, Vs)
    Gs = mixed_fusion(As, Ps, sd[f'{p}.aggreg_S.lin_A.weight'], sd[f'{p}.aggreg_S.lin_A.bias'],
                      sd[f'{p}.aggreg_S.lin_P.weight'], sd[f'{p}.aggreg_S.lin_P.bias'])
    Xc = X_seq.transpose(2, 3)
    Uc = torch.tanh(alpha * torch.matmul(Xc, sd[f'{p}.params_C.Wu']))
    Vc = torch.tanh(alpha * torch.matmul(Xc, sd[f'{p}.params_C.Wv']))
    Pc = _antisym_softmax(Uc, Vc)
    Gc = mixed_fusion(Ac, Pc, sd[f'{p}.aggreg_C.lin_A.weight'], sd[f'{p}.aggreg_C.lin_A.bias'],
                      sd[f'{p}.aggreg_C.lin_P.weight'], sd[f'{p}.aggreg_C.lin_P.bias'])
    return Gs, Gc


# --------------------------------------------------------------------------
# STCGNN  (reference: STC_GNN.py:175-207)
# --------------------------------------------------------------------------
def out_head(H: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """sigmoid(Lin(h//2 -> 1)(Lin(h -> h//2)(H))), no inner nonlinearity.

    Reference: STC_GNN.py:182-183, 206.
    """
    y = torch.nn.functional.linear(H, sd['out_proj.0.weight'], sd.get('out_proj.0.bias'))
    y = torch.nn.functional.linear(y, sd['out_proj.1.weight'], sd.get('out_proj.1.bias'))
    return torch.sigmoid(y)


def encdec_forward(X_seq: Tensor, Gs: Tensor, Gc: Tensor, sd: Dict[str, Tensor],
                   Ks: int, Kc: int, hidden: int, num_layers: int,
                   out_horizon: int, conv=bdg_dif) -> Tensor:
    """Encoder -> autoregressive decoder -> head, graphs given.

    Reference: STC_GNN.py:189-207 (everything after MGP_Gen).  ``X_seq``
    (B,T,N,C); returns (B,horizon,N,C).
    """
    assert X_seq.dim() == 4
    x5 = X_seq.unsqueeze(-1)
    _, states = stc_encoder(Gs, Gc, x5, sd, 'encoder', num_layers, Ks, Kc, hidden, conv=conv)
    dec_in = states[-1]
    outs = []
    for _ in range(out_horizon):
        dec_in, states = stc_decoder(Gs, Gc, dec_in, states, sd, 'decoder',
                                     num_layers, Ks, Kc, conv=conv)
        outs.append(dec_in)
    stacked = torch.stack(outs, dim=1)
    return out_head(stacked, sd).squeeze(-1)


def stcgnn_forward(X_seq: Tensor, As: Tensor, Ac: Tensor, sd: Dict[str, Tensor],
                   Ks: int, Kc: int, hidden: int, num_layers: int,
                   out_horizon: int) -> Tensor:
    """Full model: MGP_Gen, then ``encdec_forward``.  Reference: STC_GNN.py:185-207."""
    Gs, Gc = mgp_gen(X_seq, As, Ac, sd)
    return encdec_forward(X_seq, Gs, Gc, sd, Ks, Kc, hidden, num_layers, out_horizon)


# --------------------------------------------------------------------------
# ComboLoss  (reference: framework/Model_Trainer.py:9-23)
# --------------------------------------------------------------------------
def combo_loss(y_pred: Tensor, y_true: Tensor) -> Tensor:
    """mean BCE + per-sample Dice, averaged over the batch.

    Reference: Model_Trainer.py:14-23.
    """
    bce = torch.nn.functional.binary_cross_entropy(y_pred, y_true, reduction='mean')
    B = y_pred.shape[0]
    num = 2 * (y_pred * y_true).reshape(B, -1).sum(-1)
    den = (y_pred + y_true).reshape(B, -1).sum(-1)
    return bce + torch.mean(1 - num / den)


# --------------------------------------------------------------------------
# Synthetic inputs of SURVEY section 8(d1): shared by tests and bench baseline
# --------------------------------------------------------------------------
def queen_grid_adjacency(H: int, W: int) -> Tuple[Tensor, Tensor]:
    """COO (row, col) of the H x W 8-neighbour grid, node id = h*W + w.

    Reproduces ``s_adj`` of data/SF-incidents-4h.npz for 10 x 10 (binary,
    symmetric, zero diagonal).  Entries are emitted row-major, columns ascending.
    """
    hh = torch.arange(H).repeat_interleave(W)
    ww = torch.arange(W).repeat(H)
    rows, cols = [], []
    for dh in (-1, 0, 1):
        for dw in (-1, 0, 1):
            if dh == 0 and dw == 0:
                continue
            nh, nw = hh + dh, ww + dw
            ok = (nh >= 0) & (nh < H) & (nw >= 0) & (nw < W)
            rows.append((hh * W + ww)[ok])
            cols.append((nh * W + nw)[ok])
    r = torch.cat(rows)
    c = torch.cat(cols)
    order = torch.argsort(r * (H * W) + c)
    return r[order], c[order]
