"""CPU emulation of the C-ABI kernels of ``include/stc_hip.h``.  TEST INFRASTRUCTURE ONLY.

One method per exported kernel, same argument meaning, computing with torch
CPU ops in float64-free plain fp32 (or whatever dtype the tensors carry, so
the tests can also run the whole decomposition in fp64).  It serves two ends:

* per-kernel reference for the ``-m gpu`` parity tests (HIP kernel vs this, on
  the same seeded inputs);
* a stand-in kernel set that the CPU tests inject into the host orchestration
  (``stc_hip.ops``) to prove, against the oracle's autograd, that the
  decomposition the HIP path uses -- feature-side Chebyshev recurrence,
  project-then-mix node kernel, hand-derived backward -- is the reference's
  math.  The product never imports this module and has no CPU path of its own.

Reference lines each kernel accounts for are cited per method.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch

Tensor = torch.Tensor


def _expand_rows(rowptr: Tensor) -> Tensor:
    counts = (rowptr[1:] - rowptr[:-1]).long()
    return torch.repeat_interleave(torch.arange(counts.numel()), counts)


class _Bf16Emulated:
    """CPU stand-in for ``HipKernels.bf16``: every bfloat16 tensor argument (also inside lists / tuples) is widened to fp32,
    the fp32 twin runs, and the results are rounded back into the caller's bf16 tensors -- i.e. "fp32 math on bf16-valued
    planes, one rounding per stored plane", the contract of the bf16 kernels (their extra internal roundings are what the
    GPU tests bound)."""

    name = 'emulated-cpu-bf16'
    folds_dH = True

    def __init__(self, em):
        self.em = em

    def cell_bwd_planar_supported(self, Cc, h, cin=16):
        """As ``_Bf16Planar``: the one-launch backward on bf16 planes exists for C = 32 (both input widths) and for C = 64 with the wide input."""
        return self.em.cell_bwd_planar_supported(Cc, h) and (cin == h or Cc == 32)

    def __getattr__(self, name):
        fn = getattr(self.em, name)
        pairs = []

        def widen(a):
            if isinstance(a, torch.Tensor) and a.dtype == torch.bfloat16:
                w = a.float()
                pairs.append((a, w))
                return w
            if isinstance(a, (list, tuple)):
                return type(a)(widen(v) for v in a)
            return a

        def call(*args, **kw):
            out = fn(*[widen(a) for a in args], **{k: widen(v) for k, v in kw.items()})
            for orig, w in pairs:
                orig.copy_(w)
            pairs.clear()
            return out
        return call


def _pow2_scale(amax: Tensor, t: int) -> Tensor:
    """2^k with amax 2^k in [2^(t-1), 2^t), elementwise; 1 where amax is zero / not finite; |k| <= 100 (csrc/stc_x3_frag.h: pow2_scale)."""
    a = amax.detach().double()
    _, e = torch.frexp(a)                                   # a = m 2^e, m in [1/2, 1)
    k = (t - e).clamp(-100, 100).double()
    return torch.where((a > 0) & torch.isfinite(a), torch.pow(torch.tensor(2.0, dtype=torch.float64), k), torch.ones_like(a))


def _q2(x: Tensor, s) -> Tensor:
    """x as the fp16 x 2 operand format carries it: (h + l) / s with h = fp16(x s), l = fp16(x s - h) (round to nearest even, subnormals kept,
    overflow to infinity) -- 22 significant bits above the subnormal floor of the SCALED value, an absolute floor of 2^-25 / s below it."""
    y = (x.double() * s).float()
    h = y.half().float()
    l = (y - h).half().float()
    return ((h.double() + l.double()) / s).to(x.dtype)


class EmulatedKernels:
    """Drop-in for ``stc_hip._lib.HipKernels`` on CPU tensors.

    ``operand_format='f16x2'``: the planar cell kernels' matrix products take their operands as the fp16 x 2 format of the HIP kernels carries
    them (csrc/stc_x3_frag.h: weight / category tables normalised to a maximum in [1/2, 1); gradient operands scaled into [2^3, 2^4) from the
    launch's gradient maximum; activations scaled per NODE in the forward and per PLANE in the backward's dW products) -- products and sums
    stay exact, so what the CPU suite sees is the format's representation error, floors included.  ``act_scales=False`` feeds activations
    unscaled, as the round-3 kernels did: the tests use it to show that the scale sweep would catch the absolute floor."""

    name = 'emulated-cpu'
    ACT_AMAX_SLOTS = 256

    def __init__(self, operand_format: Optional[str] = None, act_scales: bool = True):
        assert operand_format in (None, 'f16x2')
        self.fmt = operand_format
        self.act_scales = act_scales
        if operand_format == 'f16x2':
            self.operand_format = 1                          # what stc_hip.ops looks at (FMT_F16X2): amax / act_amax plumbing on

    def act_amax_buffer(self, like, *lead):
        return torch.zeros(*lead, self.ACT_AMAX_SLOTS, dtype=torch.float32) if self.fmt == 'f16x2' else None

    HEAVY_ROW_SUM = 24.0

    def for_graph(self, row_sum_bound):
        """As ``HipKernels.for_graph``: heavy graphs leave the fp16 x 2 format for the 24-bit one (here: the exact twin)."""
        if self.fmt != 'f16x2' or not (row_sum_bound > self.HEAVY_ROW_SUM):
            return self
        if getattr(self, '_b3_view', None) is None:
            self._b3_view = EmulatedKernels()
        return self._b3_view

    # ---- fp16 x 2 format emulation helpers (planar entry points only: the kernels that run the format)
    def _tables(self, *ts, backward=False):
        """Weight tables as the kernels hold them: normalised by their own maximum into [2^3, 2^4) (STC_W_TARGET)."""
        if self.fmt != 'f16x2':
            return ts
        return tuple(None if t is None else _q2(t, _pow2_scale(t.abs().max(), 4)) for t in ts)

    def _mix_tables(self, Tc, backward=False):
        """Category-mix tables T_c, c >= 1: maximum into [2^3, 2^4) in the forward, [2^1, 2^2) in the backward (STC_T_TARGET_*)."""
        if self.fmt != 'f16x2':
            return Tc
        out = Tc.clone()
        if Tc.shape[0] > 1:
            out[1:] = _q2(Tc[1:], _pow2_scale(Tc[1:].abs().max(), 2 if backward else 4))
        return out

    def _node_scaled(self, planes):
        """Forward: one scale per node (row of the (R, C, w) planes), from the maximum over all planes of the launch."""
        if self.fmt != 'f16x2':
            return list(planes)
        if not self.act_scales:
            return [_q2(p, 1.0) for p in planes]
        m = torch.stack([p.abs().amax(dim=(1, 2)) for p in planes]).amax(0)
        s = _pow2_scale(m, 5).view(-1, 1, 1)                          # STC_ACT_TARGET_FWD
        return [_q2(p, s) for p in planes]

    def _plane_scaled(self, planes, act_amax, rows):
        """Backward: one scale per plane and launch, from the slots the forward launch left (row ``rows[i]`` for plane i)."""
        if self.fmt != 'f16x2':
            return list(planes)
        if not self.act_scales or act_amax is None:
            return [_q2(p, 1.0) for p in planes]
        return [_q2(p, _pow2_scale(act_amax[r].max(), 6)) for p, r in zip(planes, rows)]

    #: waves a backward launch deals its nodes to (node r -> wave r % GRAD_WAVES, in order of r): 256 workgroups x 4 waves on the MI355X; the
    #: CPU tests lower it so that several nodes share a wave's sums
    GRAD_WAVES = 1024

    def _grad_scaled(self, grads, planes=(), plane_scales=()):
        """Backward: the gradient fragments of a launch take one scale per NODE (row of the (R, C, w) planes), a_n = 2^kn from the node's own
        maximum over all of them.  The dW products sum over the nodes a wave owns at the wave's REFERENCE scale a = 2^k, set by the first non-zero
        node: a node joins with its activation operand ``planes[i]`` (plane scale ``plane_scales[i]``) times a / a_n = 2^j, up to 2^8 there and up to
        2^4 more inside a_n; a node more than 2^12 above the reference ends the wave's pass -- the sums so far are added to the partial row and the
        node becomes the reference of the next pass (csrc/stc_x3_frag.h: RunScale).  Sums are exact here, so a pass boundary only shows in the
        scales the operands are represented at."""
        if self.fmt != 'f16x2':
            return list(grads), list(planes)
        m = torch.stack([g.abs().amax(dim=(1, 2)) for g in grads]).amax(0).double()
        R = m.numel()
        _, e = torch.frexp(m)
        kn = (4 - e).clamp(-100, 100)                                  # node maximum 2^kn in [2^3, 2^4)
        kn = torch.where(torch.isfinite(m), kn, torch.full_like(kn, -100))
        nw = min(self.GRAD_WAVES, R)
        EMPTY = 120
        k = torch.full((nw,), EMPTY, dtype=kn.dtype)
        a_exp, shift_exp = torch.zeros(R, dtype=torch.float64), torch.zeros(R, dtype=torch.float64)
        for t in range(0, R, nw):
            idx = torch.arange(t, min(t + nw, R))
            w = idx - t
            knt, zero = kn[idx], m[idx] == 0
            j = k[w] - knt
            fresh = (j > 12) & ~zero                                   # first gradient of the wave, or a node that ends its pass: the new reference
            k[w] = torch.where(fresh, knt, k[w])
            j = torch.where(fresh | zero, torch.zeros_like(j), j)
            up = (j - 8).clamp(min=0)
            a_exp[idx] = torch.where(zero, torch.zeros_like(j), knt + up).double()
            shift_exp[idx] = (j - up).clamp(min=-100).double()
        two = torch.tensor(2.0, dtype=torch.float64)
        a_n, shift = torch.pow(two, a_exp).view(-1, 1, 1), torch.pow(two, shift_exp).view(-1, 1, 1)
        return [_q2(g, a_n) for g in grads], [_q2(p, sp * shift) for p, sp in zip(planes, plane_scales)]

    def _plane_scales(self, act_amax, rows):
        if not self.act_scales or act_amax is None:
            return [1.0 for _ in rows]
        return [_pow2_scale(act_amax[r].max(), 6) for r in rows]                  # STC_ACT_TARGET_BWD

    @staticmethod
    def _leave_maxima(act_amax, planes):
        if act_amax is not None:
            for i, p in enumerate(planes):
                act_amax[i, 0] = torch.maximum(act_amax[i, 0], p.abs().max().float())      # any slot may hold the maximum

    @property
    def bf16(self):
        return _Bf16Emulated(self)

    # ---- stc_csr_spmm_f32: 1-mode product + Chebyshev epilogue (STC_GNN.py:28, 37)
    def csr_spmm(self, rowptr, colidx, val, n_rows, n_cols, X, Y0, Y, alpha, beta, plan=None):
        if X.dtype == torch.bfloat16:
            return self.csr_spmm_bf16(rowptr, colidx, val, n_rows, n_cols, X, Y0, Y, alpha, beta)
        B, nc, F = X.shape
        assert nc == n_cols and Y.shape == (B, n_rows, F)
        rows = _expand_rows(rowptr)
        acc = torch.zeros(B, n_rows, F, dtype=X.dtype)
        contrib = X[:, colidx.long(), :] * val.to(X.dtype)[None, :, None]
        acc.index_add_(1, rows, contrib)
        out = alpha * acc
        if beta != 0.0:
            out = out + beta * Y0
        Y.copy_(out)

    # ---- stc_csr/bcsr_spmm_bf16: the same product on bf16-stored rows, fp32 sums, one rounding at the end
    def csr_spmm_bf16(self, rowptr, colidx, val, n_rows, n_cols, X, Y0, Y, alpha, beta, plan=None):
        out = torch.empty(Y.shape, dtype=torch.float32)
        self.csr_spmm(rowptr, colidx, val, n_rows, n_cols, X.float(), None if Y0 is None else Y0.float(), out, alpha, beta)
        Y.copy_(out.to(torch.bfloat16))

    # ---- stc_ring2_sum_f32: state-gradient sum + blend backward + the transpose aggregation of dY, without the dY plane
    RING2_MAX_ADD = 5

    @staticmethod
    def ring2_fits(B, n, Cc, h) -> bool:
        return h == 16 and (Cc * h) % 128 == 0 and B * n * (Cc * h // 4) < (1 << 28) and B <= 65535

    def ring2_sum(self, rowptr, colidx, val, ring2, X, X2, addends, U, Cand, Y, Z):
        B, n, Cc, h = Y.shape
        v3 = lambda t: t.reshape(B, n, Cc * h)
        agg = torch.empty(B, n, Cc * h, dtype=Y.dtype)
        self.csr_spmm(rowptr, colidx, val, n, n, v3(X if X2 is None else X + X2), None, agg, 1.0, 0.0)
        dh = agg.view(B, n, Cc, h)
        for t in addends:
            dh = dh + t
        Y.copy_(dh)
        dY = dh * U * (1 - Cand * Cand)
        out = torch.empty(B, n, Cc * h, dtype=Y.dtype)
        self.csr_spmm(rowptr, colidx, val, n, n, v3(dY), None, out, 1.0, 0.0)
        Z.copy_(out.view(B, n, Cc, h))

    # ---- stc_ring2_blend_f32: the GRU blend on Y = A + S.Bm and the aggregation of the new state (STC_GNN.py:76-78, then :37 of the next cell)
    def ring2_blend(self, rowptr, colidx, val, ring2, Bm, A, U, H, Cand, Hnew, SHnew):
        B, n, Cc, h = H.shape
        self.spmm_blend_fwd(rowptr, colidx, val, None, Bm, A, U, H, Cand, Hnew)
        out = torch.empty(B, n, Cc * h, dtype=H.dtype)
        self.csr_spmm(rowptr, colidx, val, n, n, Hnew.reshape(B, n, Cc * h), None, out, 1.0, 0.0)
        SHnew.copy_(out.view(B, n, Cc, h))

    # ---- stc_ring2_chain_f32: two chained aggregations -- the order-3 feature recurrence (STC_GNN.py:24-29 on the feature side, :37) and its transpose
    def ring2_chain(self, rowptr, colidx, val, ring2, X, X2, alpha1, add1, V, alpha2, add0, Z):
        B, n, Cc, h = Z.shape
        v3 = lambda t: t.reshape(B, n, Cc * h)
        first = torch.empty(B, n, Cc * h, dtype=Z.dtype)
        self.csr_spmm(rowptr, colidx, val, n, n, v3(X if X2 is None else X + X2), None, first, float(alpha1), 0.0)
        for t in add1:
            first = first + v3(t)
        if V is not None:
            V.copy_(first.view(B, n, Cc, h))
        out = torch.empty(B, n, Cc * h, dtype=Z.dtype)
        self.csr_spmm(rowptr, colidx, val, n, n, first.contiguous(), None, out, float(alpha2), 0.0)
        for t, scale in add0:
            out = out + scale * v3(t)
        Z.copy_(out.view(B, n, Cc, h))

    # ---- stc_csr_sddmm_f32: gradient of the 1-mode product w.r.t. the graph values (autograd of :37)
    def csr_sddmm(self, rowptr, colidx, n_rows, n_cols, A, Bm, out, alpha, accumulate):
        rows = _expand_rows(rowptr)
        dots = (A[:, rows, :] * Bm[:, colidx.long(), :]).sum(dim=(0, 2))
        if accumulate:
            out.add_(alpha * dots)
        else:
            out.copy_(alpha * dots)

    # ---- stc_cheby_dense_fwd/bwd_f32: matrix-side Chebyshev set of the small C x C graph (STC_GNN.py:24-29)
    def cheby_dense_fwd(self, G, K, T):
        n = G.shape[0]
        T[0].copy_(torch.eye(n, dtype=G.dtype))
        if K > 1:
            T[1].copy_(G)
        for k in range(2, K):
            T[k].copy_(torch.mm(2 * G, T[k - 1]) - T[k - 2])

    def cheby_dense_bwd(self, G, T, dT, dG):
        """dT (K,n,n) is consumed (used as scratch); dG overwritten."""
        K = T.shape[0]
        dG.zero_()
        for k in range(K - 1, 1, -1):
            dG.add_(2 * torch.mm(dT[k], T[k - 1].t()))
            dT[k - 1].add_(2 * torch.mm(G.t(), dT[k]))
            dT[k - 2].sub_(dT[k])
        if K > 1:
            dG.add_(dT[1])

    # ---- stc_bdg_node_fwd_f32: 2-mode product + concat + projection + bias (STC_GNN.py:38-45)
    def bdg_node_fwd(self, Zs: Sequence[Tensor], Tc: Tensor, W: Tensor, bias: Optional[Tensor], Y: Tensor):
        """Y[r,d,:] = bias + sum_{n,c} sum_{c'} Tc[c][c',d] * (Z_n[r,c',:] @ W_{n,c}).

        Project-then-mix order: U_c = sum_n Z_n W_{n,c}, then Y = sum_c Tc[c]^T U_c.
        """
        if Zs[0].dtype == torch.bfloat16:
            return self.bdg_node_fwd_bf16(Zs, Tc, W, bias, Y)
        Ks, Kc = len(Zs), Tc.shape[0]
        R, C, L = Zs[0].shape
        Ho = W.shape[1]
        Lw = W.shape[0] // (Ks * Kc)          # slab columns [Lw, L) are padding: ignored
        Wv = W.view(Ks, Kc, Lw, Ho)
        out = torch.zeros(R, C, Ho, dtype=W.dtype)
        for c in range(Kc):
            U = torch.zeros(R, C, Ho, dtype=W.dtype)
            for n in range(Ks):
                U += Zs[n][..., :Lw] @ Wv[n, c]
            out += torch.einsum('pd,rpo->rdo', Tc[c], U)
        if bias is not None:
            out += bias
        Y.copy_(out)

    # ---- stc_bdg_node_fwd/bwd_bf16: the same kernels on bf16 slabs with the hardware path's rounding points
    # (weights and T_c rounded to bf16 once; U_c / Q_c rounded between the two contractions; fp32 sums; one final rounding)
    def node_bf16_supported(self, Ks, Kc, Cc, L, Ho):
        return Ks == Kc and 1 <= Ks <= 3 and Cc in (32, 64) and L in (16, 32) and Ho in (16, 32)

    def bdg_node_fwd_bf16(self, Zs, Tc, W, bias, Y):
        bf = torch.bfloat16
        Ks, Kc = len(Zs), Tc.shape[0]
        Ho = W.shape[1]
        Lw = W.shape[0] // (Ks * Kc)
        Wv = W.to(bf).float().view(Ks, Kc, Lw, Ho)
        out = torch.zeros(*Zs[0].shape[:2], Ho)
        for c in range(Kc):
            U = sum(Zs[n][..., :Lw].float() @ Wv[n, c] for n in range(Ks)).to(bf).float()
            out += U if c == 0 else torch.einsum('pd,rpo->rdo', Tc[c].to(bf).float(), U)
        if bias is not None:
            out += bias
        Y.copy_(out.to(bf))

    def bdg_node_bwd_bf16(self, Zs, Tc, W, dY, dZs, dW, db):
        bf = torch.bfloat16
        Ks, Kc = len(Zs), Tc.shape[0]
        L = Zs[0].shape[-1]
        Ho = W.shape[1]
        Lw = W.shape[0] // (Ks * Kc)
        Wv = W.to(bf).float().view(Ks, Kc, Lw, Ho)
        g = dY.float()
        Q = [g if c == 0 else torch.einsum('pd,rdo->rpo', Tc[c].to(bf).float(), g).to(bf).float() for c in range(Kc)]
        dWv = torch.zeros(Ks, Kc, Lw, Ho)
        for n in range(Ks):
            acc = torch.zeros(*Zs[n].shape[:2], L)
            for c in range(Kc):
                acc[..., :Lw] += Q[c] @ Wv[n, c].t()
                dWv[n, c] = torch.einsum('rpl,rpo->lo', Zs[n][..., :Lw].float(), Q[c])
            dZs[n].copy_(acc.to(bf))
        dW.copy_(dWv.view(Ks * Kc * Lw, Ho))
        if db is not None:
            db.copy_(g.sum(dim=(0, 1)))

    # ---- stc_bdg_node_bwd_f32: autograd of the above
    def bdg_node_bwd(self, Zs, Tc, W, dY, dZs, dW, db, dTc):
        if Zs[0].dtype == torch.bfloat16:
            assert dTc is None
            return self.bdg_node_bwd_bf16(Zs, Tc, W, dY, dZs, dW, db)
        Ks, Kc = len(Zs), Tc.shape[0]
        R, C, L = Zs[0].shape
        Ho = W.shape[1]
        Lw = W.shape[0] // (Ks * Kc)
        Wv = W.view(Ks, Kc, Lw, Ho)
        dWv = torch.zeros_like(Wv)
        # Q_c[r,c',:] = sum_d Tc[c][c',d] dY[r,d,:]
        Q = [torch.einsum('pd,rdo->rpo', Tc[c], dY) for c in range(Kc)]
        for n in range(Ks):
            acc = torch.zeros(R, C, L, dtype=W.dtype)            # pad columns get zero gradient
            for c in range(Kc):
                acc[..., :Lw] += Q[c] @ Wv[n, c].t()
                dWv[n, c] = torch.einsum('rpl,rpo->lo', Zs[n][..., :Lw], Q[c])
            dZs[n].copy_(acc)
        dW.copy_(dWv.view(Ks * Kc * Lw, Ho))
        if db is not None:
            db.copy_(dY.sum(dim=(0, 1)))
        if dTc is not None:
            for c in range(Kc):
                U = torch.zeros(R, C, Ho, dtype=W.dtype)
                for n in range(Ks):
                    U += Zs[n][..., :Lw] @ Wv[n, c]
                dTc[c].copy_(torch.einsum('rpo,rdo->pd', U, dY))

    # ---- stc_mix_dt_f32: the category graph's gradient through one BDG_Dif for few categories (what the packed matrix-core backward leaves out)
    @staticmethod
    def mix_dT_supported(Ks, Kc, Cc, L, Ho) -> bool:
        return Ks == Kc and 1 <= Ks <= 3 and 1 <= Cc <= 16 and L in (20, 32) and Ho in (16, 32)

    def mix_dT(self, Zs, W, dY, dTc):
        Ks = len(Zs)
        R, C, L = Zs[0].shape
        Ho = W.shape[1]
        Lw = W.shape[0] // (Ks * Ks)
        assert self.mix_dT_supported(Ks, Ks, C, L, Ho)
        Wv = W.view(Ks, Ks, Lw, Ho)
        dTc[0].zero_()                                                # T_0 = I is a constant of the Chebyshev stack
        for c in range(1, Ks):
            U = torch.zeros(R, C, Ho, dtype=W.dtype)
            for n in range(Ks):
                U += Zs[n][..., :Lw] @ Wv[n, c]
            # (the sum over the rows in float64: the kernel's is blocked -- per wave, workgroup, then a tree -- where a flat fp32 sum over 1e5 rows drifts)
            dTc[c].copy_(torch.einsum('rpo,rdo->pd', U.double(), dY.double()).to(dTc.dtype))

    # ---- stc_cell_small_fwd/bwd_f32: one STC_Cell step of a small graph per launch (STC_GNN.py:65-79 and its autograd)
    SMALL_MAX_ROWS = 65535
    SMALL_PREFERRED_ROWS = 65535
    SMALL_STAGED_ROWS = 640      # N*C rows per sample that a compute unit's LDS stages (above: every gather from L2; dense graphs go to the general path)

    def cell_small_supported(self, Ks, Kc, Cc, cin, h, n_nodes=0) -> bool:
        return Ks in (2, 3) and Kc == Ks and 1 <= Cc <= 16 and h == 16 and (cin == 16 or 1 <= cin <= 4) and n_nodes * Cc <= self.SMALL_MAX_ROWS

    @staticmethod
    def cell_small_zg_width(cin) -> int:
        return 32 if cin == 16 else 20

    @staticmethod
    def cell_small_params(Ks, Kc, cin, h=16) -> int:
        return Ks * Kc * (cin + h) * 3 * h + 3 * h

    cell_small_param_rows = 4          # rows of the parameter-gradient partials per sample and split (the twin adds to the first)

    @staticmethod
    def cell_small_splits(batch: int, rows: int = 0) -> int:
        g = max(1, min(8, 256 // max(1, batch)))
        while g > 1 and rows and rows < 48 * g:
            g //= 2
        return g

    def _small_agg(self, rowptr, colidx, val, T):
        """S.T over the node axis of T (B, N, C, w)."""
        B, N, Cc, w = T.shape
        out = torch.empty(B, N, Cc * w, dtype=T.dtype)
        self.csr_spmm(rowptr, colidx, val, N, N, T.reshape(B, N, Cc * w), None, out, 1.0, 0.0)
        return out.view(B, N, Cc, w)

    def cell_small_fwd(self, rowptr, colidx, val, X, H, Tc, Wg, bg, Wc, bc, U, R, Cand, Hnew, RH, Zg, Zc, checked=True, Z0=None, splits=1, Z0c=None,
                       Z1c=None, graph2=None, Zg2=None, Zc2=None):
        B, N, Cc, h = H.shape
        cin = X.shape[-1]
        order3 = Tc.shape[0] == 3                 # the third slab through the second graph T_2(S) = 2 S^2 - I (formed on the matrix side, as the kernels take it)
        assert order3 == (graph2 is not None) and (not order3 or (Zg2 is not None and Zc2 is not None and Z0 is None))
        rows = lambda t: t.reshape(B * N, Cc, t.shape[-1])
        XH = torch.cat([X, H], -1)
        SXH = self._small_agg(rowptr, colidx, val, XH)
        G = torch.empty(B * N, Cc, 2 * h, dtype=H.dtype)
        slabs_g = [rows(XH), rows(SXH)]
        if order3:
            S2XH = self._small_agg(*graph2, XH)
            slabs_g.append(rows(S2XH))
            Zg2.zero_()
            Zg2v = Zg2.view(B, N, Cc, Zg2.shape[-1])
            Zg2v[..., :h] = S2XH[..., cin:]
            Zg2v[..., h:h + cin] = S2XH[..., :cin]
        self.bdg_node_fwd(slabs_g, Tc, Wg, bg, G)
        G = G.view(B, N, Cc, 2 * h)
        U.copy_(torch.sigmoid(G[..., :h]))
        R.copy_(torch.sigmoid(G[..., h:]))
        RH.copy_(R * H)
        SRH = self._small_agg(rowptr, colidx, val, RH)
        Zg.zero_()
        Zgv = Zg.view(B, N, Cc, Zg.shape[-1])
        Zgv[..., :h] = SXH[..., cin:]
        Zgv[..., h:h + cin] = SXH[..., :cin]
        Zc.copy_(SRH.reshape(B, N * Cc, h))
        if Z0 is not None:
            Z0.zero_()
            Z0v = Z0.view(B, N, Cc, Z0.shape[-1])
            Z0v[..., :h] = H
            Z0v[..., h:h + cin] = X
        if Z0c is not None:
            for dst, hpart in ((Z0c, RH), (Z1c, SRH)):
                dst.copy_(Zg if dst is Z1c else Z0)
                dst.view(B, N, Cc, dst.shape[-1])[..., :h] = hpart
        Y = torch.empty(B * N, Cc, h, dtype=H.dtype)
        slabs_c = [rows(torch.cat([X, RH], -1)), rows(torch.cat([SXH[..., :cin], SRH], -1))]
        if order3:
            S2RH = self._small_agg(*graph2, RH)
            Zc2.copy_(S2RH.reshape(B, N * Cc, h))
            slabs_c.append(rows(torch.cat([S2XH[..., :cin], S2RH], -1)))
        self.bdg_node_fwd(slabs_c, Tc, Wc, bc, Y)
        Cand.copy_(torch.tanh(Y.view(B, N, Cc, h)))
        Hnew.copy_((1.0 - U) * H + U * Cand)

    def cell_small_bwd(self, rowptr, colidx, val, X, H, Tc, Wg, Wc, U, R, Cand, RH, Zg, Zc, dHnew, dX, accumulate_x, dH, accumulate_h,
                       dparams, has_bg, has_bc, checked=True, dZ1c=None, dZ1g=None, dYg=None, splits=1, dYc=None, graph2=None, Zg2=None, Zc2=None):
        B, N, Cc, h = H.shape
        cin = X.shape[-1]
        L = cin + h
        Kc = Ks = Tc.shape[0]
        order3 = Ks == 3
        assert order3 == (graph2 is not None) and (not order3 or (Zg2 is not None and Zc2 is not None and dZ1c is None))
        Zgv = Zg.view(B, N, Cc, Zg.shape[-1])
        SX, SH, SRH = Zgv[..., h:h + cin], Zgv[..., :h], Zc.view(B, N, Cc, h)
        if order3:
            Zg2v = Zg2.view(B, N, Cc, Zg2.shape[-1])
            S2X, S2H, S2RH = Zg2v[..., h:h + cin], Zg2v[..., :h], Zc2.view(B, N, Cc, h)
        nW = Ks * Kc * L
        first = dparams[::self.cell_small_param_rows * splits]       # (a view: row 0 of every sample's group)
        assert first.shape[0] == B
        dWg, dbg = first[:, :nW * 2 * h], first[:, nW * 2 * h:nW * 2 * h + 2 * h]
        dWc, dbc = first[:, nW * 2 * h + 2 * h:nW * 3 * h + 2 * h], first[:, nW * 3 * h + 2 * h:nW * 3 * h + 3 * h]

        def conv_bwd(Z0, Z1, W, dY, dW_rows, db_rows, dump=None, Z2=None):
            """d[Z0] + S^T d[Z1] (+ T_2(S)^T d[Z2] at order 3) of one convolution (B, N, C, L); parameter gradients per sample; ``dump``: receives
            d[Z1] in [H | X | 0] order."""
            out = torch.empty(B, N, Cc, L, dtype=H.dtype)
            if dump is not None:
                dump.zero_()
            for b in range(B):
                dZ = [torch.empty(N, Cc, L, dtype=H.dtype) for _ in range(Ks)]
                dW, db = torch.empty_like(W), torch.empty(W.shape[1], dtype=H.dtype)
                self.bdg_node_bwd([Z0[b], Z1[b]] + ([Z2[b]] if order3 else []), Tc, W, dY[b], dZ, dW, db, None)
                dW_rows[b] += dW.reshape(-1)
                if db_rows is not None:
                    db_rows[b] += db
                if dump is not None:
                    dv_ = dump[b].view(N, Cc, dump.shape[-1])
                    dv_[..., :h] = dZ[1][..., cin:]
                    dv_[..., h:h + cin] = dZ[1][..., :cin]
                back = torch.empty(1, N, Cc * L, dtype=H.dtype)
                self.csr_spmm(rowptr, colidx, val, N, N, dZ[1].reshape(1, N, Cc * L), dZ[0].reshape(1, N, Cc * L), back, 1.0, 1.0)
                if order3:
                    back2 = torch.empty_like(back)
                    self.csr_spmm(*graph2, N, N, dZ[2].reshape(1, N, Cc * L), back, back2, 1.0, 1.0)
                    back = back2
                out[b] = back.view(N, Cc, L)
            return out

        dCpre = dHnew * U * (1.0 - Cand * Cand)
        if dYc is not None:
            dYc.copy_(dCpre.reshape(B, N * Cc, h))
        dCI = conv_bwd(torch.cat([X, RH], -1), torch.cat([SX, SRH], -1), Wc, dCpre, dWc, dbc if has_bc else None, dump=dZ1c,
                       Z2=torch.cat([S2X, S2RH], -1) if order3 else None)
        dRH, dXc = dCI[..., cin:], dCI[..., :cin]
        dGu = dHnew * (Cand - H) * U * (1.0 - U)
        dGr = dRH * H * R * (1.0 - R)
        dXH = conv_bwd(torch.cat([X, H], -1), torch.cat([SX, SH], -1), Wg, torch.cat([dGu, dGr], -1), dWg, dbg if has_bg else None, dump=dZ1g,
                       Z2=torch.cat([S2X, S2H], -1) if order3 else None)
        if dYg is not None:
            dYg.copy_(torch.cat([dGu, dGr], -1).reshape(B, N * Cc, 2 * h))
        if dH is not None:
            g = dHnew * (1.0 - U) + dRH * R + dXH[..., cin:]
            dH.copy_(dH + g if accumulate_h else g)
        if dX is not None:
            g = dXc + dXH[..., :cin]
            dX.copy_(dX + g if accumulate_x else g)

    # ---- stc_graph_grad_f32 / stc_mix_grad_f32: the graph-gradient products of learned graphs, float64 sums over cells and samples
    @staticmethod
    def _selected(t, cell0, cell_step, n_sel, N):
        sel = t[cell0:cell0 + (n_sel - 1) * cell_step + 1:cell_step] if n_sel else t[:0]
        return sel.reshape(sel.shape[0] * sel.shape[1], N, -1).double()           # (g, N, C * width)

    GRAD_CHUNKS = 3

    def grad_partials(self, like, total, chunks=None):
        return torch.full((self.GRAD_CHUNKS if chunks is None else chunks, total), float('nan'), dtype=torch.float64)

    @staticmethod
    def _into(into, res):
        """As the kernels: every chunk's block is WRITTEN (here: the whole sum in chunk 0, zeros in the others)."""
        part, off = into
        part[:, off:off + res.numel()] = 0
        part[0, off:off + res.numel()] = res.reshape(-1).double()

    def graph_grad(self, A, Bm, cell0, cell_step, n_sel, N, into=None):
        res = torch.einsum('gnf,gmf->nm', self._selected(A, cell0, cell_step, n_sel, N), self._selected(Bm, cell0, cell_step, n_sel, N))
        return res if into is None else self._into(into, res)

    def mix_grad(self, A, Bm, cell0, cell_step, n_sel, N, into=None):
        res = torch.einsum('gna,gnb->ab', self._selected(A, cell0, cell_step, n_sel, N), self._selected(Bm, cell0, cell_step, n_sel, N))
        return res if into is None else self._into(into, res)

    # ---- stc_bdg_node_post_bwd_f32: Y = A + S.Bm (Ks = Kc = 2); backward from (X, dA, dBm)
    def node_post_supported(self, Ks, Kc, Cc, L, Ho) -> bool:
        return Ks == 2 and Kc == 2

    # planar forms: the [Xt | H] rows as two (R, C, 16) planes -- emulated by concatenating them
    def cell_planar_supported(self, Ks, Kc, Cc, h) -> bool:
        return Ks == 2 and Kc == 2 and h == 16

    def cell_planar_post_fused(self, Cc) -> bool:
        return True

    def cell_gates_fwd_planar(self, X, H, SX, SH, Tc, W, bias, U, Rg, RH, post=None, act_amax=None):
        cin, h = X.shape[-1], H.shape[-1]                        # cin = h, or 1..4 (narrow input plane, layer 0)
        self._leave_maxima(act_amax, (X, SX, H, SH) if cin == h else (H, SH, X, SX))      # rows in the launch order of the planes
        CandIn = torch.empty(H.shape[:-1] + (cin + h,), dtype=W.dtype)
        Xq, Hq, SXq, SHq = self._node_scaled((X, H, SX, SH))
        (Wq,), Tq = self._tables(W), self._mix_tables(Tc)
        self.cell_gates_fwd([torch.cat([Xq, Hq], -1), torch.cat([SXq, SHq], -1)], Tq, Wq, bias, H, U, Rg, CandIn)
        CandIn[..., :cin] = X                                   # (the row's X part is the input itself, not its operand representation)
        if RH is not None:                                      # optional with post=: cell_bwd_planar forms R*H itself
            RH.copy_(CandIn[..., cin:])
        if post is not None:                                   # + the candidate's projection on [Xt | R*H]
            Wc, bc, A, Bm = post
            if self.fmt == 'f16x2':                             # the node's scale covers [Xt | R*H] too (|R*H| <= |H|)
                Xc, RHc = (self._node_scaled((X, H, SX, SH, CandIn[..., cin:].contiguous()))[i] for i in (0, 4)) if self.act_scales else \
                          (_q2(X, 1.0), _q2(CandIn[..., cin:], 1.0))
                (Wcq,) = self._tables(Wc)
                self.node_post_fwd(torch.cat([Xc, RHc], -1), Tq, Wcq, bc, A, Bm)
            else:
                self.node_post_fwd(CandIn, Tc, Wc, bc, A, Bm)

    def cell_gates_bwd_planar(self, X, H, SX, SH, Tc, W, dRH, Cand, U, Rg, dHnew, dZs, dW, db, dH, act_amax=None):
        cin, h = X.shape[-1], H.shape[-1]
        if self.fmt == 'f16x2':
            # As the kernel: the gate prologue runs in fp32 on the planes themselves (dG and the state's own share are exact); the matrix
            # products take dG, the tables and -- the dW products -- the planes, each as its format carries it.
            order = (0, 1, 2, 3) if cin == h else (2, 3, 0, 1)                # rows of the slots: wide {X, S.X, H, S.H}, narrow {H, S.H, x, S.x}
            (Wq,), Tq = self._tables(W, backward=True), self._mix_tables(Tc, backward=True)
            dG = torch.cat([dHnew * (Cand - H) * U * (1 - U), dRH * H * Rg * (1 - Rg)], -1)
            (dGq,), (Xq, SXq, Hq, SHq) = self._grad_scaled((dG,), (X, SX, H, SH), self._plane_scales(act_amax, order))
            own = dRH * Rg + dHnew * (1 - U)
            rows = [torch.empty(H.shape[:-1] + (cin + h,), dtype=W.dtype) for _ in range(2)]
            EmulatedKernels().bdg_node_bwd([torch.cat([Xq, Hq], -1), torch.cat([SXq, SHq], -1)], Tq, Wq, dGq, rows, dW, None, None)
            if db is not None:
                db.copy_(dG.sum(dim=(0, 1)))                                  # (summed from the fp32 fragments in the kernel)
            fold = dH is None
            dZs[2].copy_(rows[0][..., cin:] + (own if fold else 0)); dZs[3].copy_(rows[1][..., cin:])
            if dZs[0] is not None:
                dZs[0].copy_(rows[0][..., :cin]); dZs[1].copy_(rows[1][..., :cin])
            if not fold:
                dH.copy_(own)
            return
        fold = dH is None                                                   # bf16 kernels: the prologue's share goes into dZs[2]
        if fold:
            dH = torch.empty_like(H)
        rows = [torch.empty(H.shape[:-1] + (cin + h,), dtype=W.dtype) for _ in range(2)]
        dCandIn = torch.cat([torch.zeros_like(X), dRH], -1)               # only the R*H part is read
        self.cell_gates_bwd([torch.cat([X, H], -1), torch.cat([SX, SH], -1)], Tc, W, dCandIn, None, H, U, Rg, dHnew, rows, dW, db, None, dH,
                            dH_in_scaled=True, Cand=Cand)
        dZs[2].copy_(rows[0][..., cin:] + (dH if fold else 0)); dZs[3].copy_(rows[1][..., cin:])  # d H plane, d SH plane
        if dZs[0] is not None:
            dZs[0].copy_(rows[0][..., :cin]); dZs[1].copy_(rows[1][..., :cin])  # d X plane, d SX plane

    # stc_cell_bwd_planar_f32: candidate (post-aggregation form) + gates backward of one planar cell step, composed from the two twins
    def cell_bwd_planar_supported(self, Cc, h) -> bool:
        return h == 16 and os.environ.get('STC_FUSE_CELL_BWD', '1') != '0'

    def cell_bwd_planar(self, X, H, SX, SH, Tc, Wg, Wc, U, Rg, Cand, dHnew, dBm, dZs, dWg, dbg, dWc, dbc, accumulate_x=False, accumulate_h=False,
                        act_amax=None):
        cin, h = X.shape[-1], H.shape[-1]
        old = [None if (z is None or not acc) else z.clone() for z, acc in zip(dZs, (accumulate_x, accumulate_x, accumulate_h, accumulate_h))]
        dY = dHnew * U * (1 - Cand * Cand)
        RH, dRH = Rg * H, torch.empty_like(H)
        xr, hr = (0, 2) if cin == h else (2, 0)                   # slot rows of the X plane and of the H plane (R*H rides on H's)
        post_amax = None if act_amax is None else ((act_amax[xr], act_amax[hr]) if cin == h else (act_amax[hr], act_amax[xr]))
        if cin == h:
            dXc = torch.empty_like(H)
            self.node_post_bwd(X, Tc, Wc, dY, dBm, dXc, dWc, dbc, X2=RH, dX2=dRH, act_amax=post_amax)
        else:
            self.node_post_bwd(RH, Tc, Wc, dY, dBm, dRH, dWc, dbc, X2=X, act_amax=post_amax)
        self.cell_gates_bwd_planar(X, H, SX, SH, Tc, Wg, dRH, Cand, U, Rg, dHnew, dZs, dWg, dbg, None, act_amax=act_amax)
        if cin == h:
            dZs[0].add_(dXc)
        for z, o in zip(dZs, old):
            if o is not None:
                z.add_(o)

    def spmm_sum(self, rowptr, colidx, val, plan, X, X2, addends, Y, blend=None, alpha=1.0, amax=None):
        B, n, Cc, h = Y.shape
        src = X if X2 is None else X + X2
        self.csr_spmm(rowptr, colidx, val, n, n, src.reshape(B, n, Cc * h), None, Y.view(B, n, Cc * h), float(alpha), 0.0)
        for ent in addends:
            t, off = ent[0], ent[1]
            Y += (ent[2] if len(ent) > 2 else 1.0) * t[..., off:off + h]
        if amax is not None:                                       # any slot may hold the maximum
            amax[0] = torch.maximum(amax[0], Y.abs().max())
        if blend is not None:
            U, Cand, dY = blend
            dY.copy_(Y * U * (1 - Cand * Cand))

    # planar cell convolutions of order K (stc_cell_*_planar_k_f32): composed from the slab-form twins on concatenated planes
    def cell_planar_k_supported(self, K, Cc, h) -> bool:
        return K == 3 and h == 16

    @staticmethod
    def _cat_planes(Zx, Zh):
        return [torch.cat([x, hh], -1) for x, hh in zip(Zx, Zh)]            # reference column order [X | H]

    def _planes_k_fwd(self, Zx, Zh, Tc, W, act_amax):
        """Order-3 forward launches: slots in launch order (Zx first for a wide input, Zh first for a narrow one), operands in their format."""
        wide = Zx[0].shape[-1] == Zh[0].shape[-1]
        self._leave_maxima(act_amax, (list(Zx) + list(Zh)) if wide else (list(Zh) + list(Zx)))
        q = self._node_scaled(list(Zx) + list(Zh))
        (Wq,), Tq = self._tables(W), self._mix_tables(Tc)
        return q[:len(Zx)], q[len(Zx):], Tq, Wq

    def cell_gates_fwd_planar_k(self, Zx, Zh, Tc, W, bias, U, Rg, RH, act_amax=None):
        cin, h = Zx[0].shape[-1], Zh[0].shape[-1]
        CandIn = torch.empty(Zh[0].shape[:-1] + (cin + h,), dtype=W.dtype)
        Zxq, Zhq, Tq, Wq = self._planes_k_fwd(Zx, Zh, Tc, W, act_amax)
        self.cell_gates_fwd(self._cat_planes(Zxq, Zhq), Tq, Wq, bias, Zh[0], U, Rg, CandIn)
        RH.copy_(CandIn[..., cin:])

    def cell_cand_fwd_planar_k(self, Zx, Zh, Tc, W, bias, U, H, Cand, Hnew, act_amax=None):
        Zxq, Zhq, Tq, Wq = self._planes_k_fwd(Zx, Zh, Tc, W, act_amax)
        self.cell_blend_fwd(self._cat_planes(Zxq, Zhq), Tq, Wq, bias, U, H, Cand, Hnew)

    def _split_planes(self, rows, dZx, dZh, cin):
        for n, r in enumerate(rows):
            dZh[n].copy_(r[..., cin:])
            if dZx[n] is not None:
                dZx[n].copy_(r[..., :cin])

    def cell_gates_bwd_planar_k(self, Zx, Zh, Tc, W, dRH, Cand, U, Rg, dHnew, dZx, dZh, dW, db, dH, accumulate_x=False, act_amax=None):
        cin, h = Zx[0].shape[-1], Zh[0].shape[-1]
        before = [z.clone() for z in dZx] if accumulate_x else None              # the candidate's gradients already in the X-side planes
        fold = dH is None                                                   # the prologue's share goes into dZh[0]
        if fold:
            dH = torch.empty_like(Zh[0])
        rows = [torch.empty(Zh[0].shape[:-1] + (cin + h,), dtype=W.dtype) for _ in Zh]
        dCandIn = torch.cat([torch.zeros_like(Zx[0]), dRH], -1)               # only the R*H part is read
        self.cell_gates_bwd(self._cat_planes(Zx, Zh), Tc, W, dCandIn, None, Zh[0], U, Rg, dHnew, rows, dW, db, None, dH, dH_in_scaled=True, Cand=Cand)
        self._split_planes(rows, dZx, dZh, cin)
        if fold:
            dZh[0] += dH
        if accumulate_x:
            for z, b in zip(dZx, before):
                z += b

    def cell_cand_bwd_planar_k(self, Zx, Zh, Tc, W, dHnew, U, Cand, dZx, dZh, dW, db, act_amax=None):
        cin, h = Zx[0].shape[-1], Zh[0].shape[-1]
        rows = [torch.empty(Zh[0].shape[:-1] + (cin + h,), dtype=W.dtype) for _ in Zh]
        self.cell_cand_bwd(self._cat_planes(Zx, Zh), Tc, W, dHnew, U, Cand, rows, dW, db)
        self._split_planes(rows, dZx, dZh, cin)

    def node_post_fwd(self, X, Tc, W, bias, A, Bm, X2=None):
        if X2 is not None:                                     # planar: 16 + 16 -> [X | X2]; narrow -> reference order [X2 (input) | X (16-wide)]
            X = torch.cat([X, X2], -1) if X2.shape[-1] == X.shape[-1] else torch.cat([X2, X], -1)
        Lw = W.shape[0] // 4
        for n, out in enumerate((A, Bm)):
            acc = torch.zeros_like(out)
            for c in range(2):
                P = X[..., :Lw] @ W[(n * 2 + c) * Lw:(n * 2 + c + 1) * Lw]            # (R, C, Ho)
                acc += P if c == 0 else torch.einsum('pd,rpo->rdo', Tc[c], P)          # T_c^T P
            out.copy_(acc + (bias if (bias is not None and n == 0) else 0))

    def spmm_blend_fwd(self, rowptr, colidx, val, plan, Bm, A, U, H, Cand, Hnew, copies=(), side=None):
        B, n, Cc, h = H.shape
        Y = torch.empty_like(A)
        self.csr_spmm(rowptr, colidx, val, n, n, Bm.view(B, n, Cc * h), A.view(B, n, Cc * h), Y.view(B, n, Cc * h), 1.0, 1.0)
        self.gru_blend_fwd(Y, U, H, Cand, Hnew)
        flat = Hnew.reshape(B * n, Cc, h)
        for buf, off in copies:
            buf[..., off:off + h].copy_(flat)
        if side is not None:
            buf, off = copies[0]
            buf[..., :off].copy_(side)
            buf[..., off + h:].zero_()

    def node_post_bwd(self, X, Tc, W, dA, dB, dX, dW, db, X2=None, dX2=None, act_amax=None):
        if X2 is not None and self.fmt == 'f16x2':             # operands in their format, then the exact twin
            (dAq, dBq), (Xq, X2q) = self._grad_scaled((dA, dB), (X, X2), self._plane_scales(
                None if act_amax is None else torch.stack([a.reshape(-1) for a in act_amax]), (0, 1)))
            (Wq,), Tq = self._tables(W, backward=True), self._mix_tables(Tc, backward=True)
            EmulatedKernels().node_post_bwd(Xq, Tq, Wq, dAq, dBq, dX, dW, db, X2=X2q, dX2=dX2)
            return
        if X2 is not None:                                     # planar: compute on the concatenated rows, hand back the planes
            w, w2 = X.shape[-1], X2.shape[-1]
            full = torch.empty(X.shape[:-1] + (w + w2,), dtype=W.dtype)
            if w2 == w:
                self.node_post_bwd(torch.cat([X, X2], -1), Tc, W, dA, dB, full, dW, db)
                dX.copy_(full[..., :w]); dX2.copy_(full[..., w:])
            else:                                                  # narrow: reference order [X2 | X]; gradient for the 16-wide plane only
                self.node_post_bwd(torch.cat([X2, X], -1), Tc, W, dA, dB, full, dW, db)
                dX.copy_(full[..., w2:])
            return
        Lw = W.shape[0] // 4
        dW.zero_()
        dX.zero_()
        for n, dYn in enumerate((dA, dB)):                     # slab n sees X with the weight blocks (n, c) and gradient dY_n
            for c in range(2):
                Q = dYn if c == 0 else torch.einsum('pd,rdo->rpo', Tc[c], dYn)        # Q_c = T_c dY_n
                Wnc = W[(n * 2 + c) * Lw:(n * 2 + c + 1) * Lw]                         # (Lw, Ho)
                dX[..., :Lw] += torch.einsum('rpo,lo->rpl', Q, Wnc)
                dW[(n * 2 + c) * Lw:(n * 2 + c + 1) * Lw] += torch.einsum('rpl,rpo->lo', X[..., :Lw], Q)
        if db is not None:
            db.copy_(dA.sum((0, 1)))

    # ---- stc_cell_gates/blend_fwd_f32: node kernel + gate math in its epilogue (STC_GNN.py:69-78)
    def cell_fused_supported(self, Ks, Kc, Cc, L, h) -> bool:
        return Ks == Kc                 # the twin composes the unfused kernels, any shape

    def cell_gates_fwd(self, Zs, Tc, W, bias, H, U, Rg, CandIn):
        h = H.shape[-1]
        cin = W.shape[0] // (len(Zs) * Tc.shape[0]) - h
        G = torch.empty(H.shape[:-1] + (2 * h,), dtype=W.dtype)
        self.bdg_node_fwd(Zs, Tc, W, bias, G)
        self.gru_gates_fwd(G, Zs[0][..., :cin], H, U, Rg, CandIn)

    def cell_cand_bwd(self, Zs, Tc, W, dHnew, U, Cand, dZs, dW, db):
        self.bdg_node_bwd(Zs, Tc, W, dHnew * U * (1 - Cand * Cand), dZs, dW, db, None)

    def cell_gates_bwd(self, Zs, Tc, W, dCandIn, dU, H, U, Rg, dH_in, dZs, dW, db, dXt, dH, dH_in_scaled=False, Cand=None):
        dG = torch.empty(H.shape[:-1] + (2 * H.shape[-1],), dtype=W.dtype)
        if Cand is not None:                                  # dH_in = dHnew: both blend products are formed here
            dU, dH_in_scaled = dH_in * (Cand - H), True
        if dXt is None:                                       # not wanted: the caller reads dCandIn[..., :cin] in place
            cin = W.shape[0] // (len(Zs) * Tc.shape[0]) - H.shape[-1]
            dXt = torch.empty(H.shape[:-1] + (cin,), dtype=W.dtype)
        extra = None if dH_in is None else (dH_in * (1 - U) if dH_in_scaled else dH_in.clone())
        self.gru_gates_bwd(dCandIn, dU, H, U, Rg, dG, dXt, dH, dH_in=extra)
        self.bdg_node_bwd(Zs, Tc, W, dG, dZs, dW, db, None)

    def cell_blend_fwd(self, Zs, Tc, W, bias, U, H, Cand, Hnew, copies=(), side=None):
        Cpre = torch.empty_like(H)
        self.bdg_node_fwd(Zs, Tc, W, bias, Cpre)
        self.gru_blend_fwd(Cpre, U, H, Cand, Hnew)
        h = H.shape[-1]
        for buf, off in copies:                                # the new state dropped into its consumers' input rows
            buf[..., off:off + h].copy_(Hnew)
        if side is not None:
            buf, off = copies[0]
            buf[..., :off].copy_(side)
            buf[..., off + h:].zero_()

    # ---- stc_gru_gates_fwd/bwd_f32: split + sigmoids + reset*H + second concat (STC_GNN.py:71-75)
    def gru_gates_fwd(self, G, Xt, H, U, Rg, CandIn):
        h = H.shape[-1]
        cin = Xt.shape[-1]
        U.copy_(torch.sigmoid(G[..., :h]))
        Rg.copy_(torch.sigmoid(G[..., h:]))
        CandIn[..., :cin].copy_(Xt)
        CandIn[..., cin:cin + h].copy_(Rg * H)
        CandIn[..., cin + h:].zero_()                      # optional zero padding up to the row width

    def gru_gates_bwd(self, dCandIn, dU, H, U, Rg, dG, dXt, dH, dH_in=None):
        h = H.shape[-1]
        cin = dXt.shape[-1]
        dRH = dCandIn[..., cin:cin + h]
        dG[..., :h].copy_(dU * U * (1 - U))
        dG[..., h:].copy_(dRH * H * Rg * (1 - Rg))
        dXt.copy_(dCandIn[..., :cin])
        dH.copy_(dRH * Rg + (dH_in if dH_in is not None else 0))

    # ---- stc_gru_blend_fwd/bwd_f32: tanh + GRU blend (STC_GNN.py:76-78)
    def gru_blend_fwd(self, Cpre, U, H, Cand, Hnew):
        Cand.copy_(torch.tanh(Cpre))
        Hnew.copy_((1.0 - U) * H + U * Cand)

    def gru_blend_bwd(self, dHnew, U, H, Cand, dCpre, dU, dH):
        dCpre.copy_(dHnew * U * (1 - Cand * Cand))
        if dU is not None:
            dU.copy_(dHnew * (Cand - H))
        if dH is not None:
            dH.copy_(dHnew * (1 - U))

    # ---- stc_head_fwd/bwd_f32: the two bias-ful Linears of the output head folded into one map + sigmoid (STC_GNN.py:182-183, 206)
    def head_fwd(self, H, w, b, y):
        y.copy_(torch.sigmoid(H.to(w.dtype) @ w + b))          # bf16 state rows (stc_head_fwd_bf16): fp32 arithmetic, fp32 y

    def head_bwd(self, H, w, y, dy, dH, dwb):
        H = H.to(w.dtype)
        g = dy * y * (1 - y)
        dH.copy_(g.unsqueeze(-1) * w)
        dwb[:-1].copy_((g.unsqueeze(-1) * H).reshape(-1, H.shape[-1]).sum(0))
        dwb[-1] = g.sum()

    # ---- stc_axpy_f32: y += a*x (Chebyshev recurrence backward, g_{k-2} -= g_k)
    def axpy(self, a, x, y):
        y.add_(x, alpha=a)

    # ---- stc_concat2_f32 / split: cat([A,B],-1) (STC_GNN.py:68) and its backward
    def concat2(self, A, Bm, out):
        a, b = A.shape[-1], Bm.shape[-1]
        out[..., :a].copy_(A)
        out[..., a:a + b].copy_(Bm)
        out[..., a + b:].zero_()

    def split2(self, src, A, Bm, addA=None, addB=None, addA_ld=0, addA2=None, addB2=None):
        a, b = A.shape[-1], Bm.shape[-1]
        A.copy_(src[..., :a] + (addA[..., :a] if addA is not None else 0) + (addA2 if addA2 is not None else 0))
        Bm.copy_(src[..., a:a + b] + (addB if addB is not None else 0) + (addB2 if addB2 is not None else 0))
