"""A schedule of STC_Cells with FEW CATEGORIES (C <= 16: the SF-incidents shape and its relatives) as one autograd node.

The SF-incidents shape (N = 100, C = 5, hidden 16) is bound by the host's launch rate on the general path (~15 launches
per cell and direction).  ``stc_cell_small_fwd/bwd_f32`` run a whole cell step -- both aggregations, both convolutions, the gate math
and the blend (reference STC_GNN.py:65-79) -- in one launch with one workgroup per sample; this module is the bookkeeping around them,
with the same interface as ``ops.stc_cell_graph``:

  * states, gates and the saved aggregates of all cells live in a handful of stacked buffers (one allocation per kind and pass);
  * a state's gradient is ONE buffer that its consumer cells write / add to in place (``accumulate_x`` / ``accumulate_h`` of the
    backward kernel), in reverse schedule order -- no autograd accumulation passes;
  * parameter gradients are partial sums (one row per sample and wave) that every cell of a parameter set adds to; one sum over the
    rows per backward pass.
Learned graphs (the reference's own mode: dense Gs from MGP_Gen, Gc through its Chebyshev stack): the gradients of Gs and Gc are sums over
ALL cells of a step, so the launches only leave their operands (the slab [H | X | 0], the gradients of the two aggregated slabs, the gate
pre-activation gradients) and the backward forms  dGs^T = sum_cells dZ_1 x Z_0  and  dT_c = sum_cells V_c x dY  as a few stacked
products per parameter set at the end -- not per cell.
"""
from __future__ import annotations

import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .graph import SpatialOperand

H16 = 16


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def small_graph_supported(k, op: SpatialOperand, Tc, Ks: int, C: int, h: int, x_widths, dtype=torch.float32) -> bool:
    """Fixed graphs, or the reference's learned ones: a learned spatial graph must be dense (values = the full N x N pattern, as
    ``graph.dense_operand`` builds it) -- its gradient is formed as a dense product."""
    if dtype != torch.float32 or not hasattr(k, 'cell_small_supported') or (op.fwd_val.requires_grad and op.nnz != op.n * op.n):
        return False
    if Ks == 3 and (op.source is None or op.fwd_val.requires_grad or op.nnz == op.n * op.n):
        return False                                                # order 3: T_2(S) as a second CSR graph -- fixed sparse graphs only
    if op.nnz == op.n * op.n and op.n * C > k.SMALL_STAGED_ROWS:
        return False                                                # a dense graph aggregates as a matrix product on the STAGED plane: the sample must fit the LDS
    if op.n * C > k.SMALL_PREFERRED_ROWS or (C == 16 and op.n * C > 4096):
        return False                                                # (16 categories on large graphs: the general path's fp32-MFMA node kernels win, 13.4 vs 18.3 ms)
    return all(k.cell_small_supported(Ks, Tc.shape[0], C, w, h, op.n) for w in set(x_widths))


def _alias(base: torch.Tensor) -> torch.Tensor:
    """The same storage as a tensor of its own (not a view in autograd's books): saved for backward while ``base`` is the node's output."""
    t = base.new_empty(0)
    t.set_(base.untyped_storage(), base.storage_offset(), base.shape, base.stride())
    return t


def _layout(schedule, cin, n_cells, outputs):
    """Where every cell's tensors live: slot in the output stack or the inner-state buffer, position inside its width group (cells whose
    input is 16 columns wide / 1..4 columns wide keep their aggregates in two buffers of different row widths)."""
    out_slot = {j: i for i, j in enumerate(outputs)}
    if len(out_slot) != len(outputs):
        raise ValueError('stc_cell_graph: duplicate output cells')
    pos, counts, inner_slot, nxt = [None] * n_cells, [0, 0], {}, 0
    # inside a width group the cells of ONE parameter set sit side by side (set by set, each in schedule order): the products over a set's
    # cells -- the learned graphs' dT_c through stc_mix_dt_f32 -- then take one contiguous run of planes (in schedule order the decoder's two
    # layers interleave)
    for j in sorted(range(n_cells), key=lambda j: (schedule[j][0], j)):
        wide = cin[j] == H16
        pos[j] = (wide, counts[wide])
        counts[wide] += 1
    for j in range(n_cells):
        if j not in out_slot:
            inner_slot[j] = nxt
            nxt += 1
    return out_slot, inner_slot, pos, counts


class _StcSmallGraph(Function):
    """schedule[j] = (stack, ('ext', i) | ('cell', k), ('ext', i) | ('cell', k)): parameter set, source of Xt, source of H."""

    @staticmethod
    def forward(ctx, k, op: SpatialOperand, Ks: int, schedule, outputs, n_ext: int, Tc, fwd_val, *tensors):
        ext = [_c(t) for t in tensors[:n_ext]]
        flat = tensors[n_ext:]
        stacks = [tuple(None if p is None else _c(p) for p in flat[i:i + 4]) for i in range(0, len(flat), 4)]   # (Wg, bg, Wc, bc)
        Tc, fwd_val = _c(Tc), _c(fwd_val)
        n_cells = len(schedule)
        ref = ext[0]
        B, N, C = ref.shape[:3]
        cin = [ext[x[1]].shape[-1] if x[0] == 'ext' else H16 for _, x, _ in schedule]
        for t in ext:                                                # (the per-cell launches below skip the wrapper's checks)
            if t.shape[:3] != (B, N, C) or t.dtype != torch.float32 or t.device != ref.device:
                raise ValueError(f'stc_cell_graph: external tensors must be float32 (B, N, C, *) on one device, got {tuple(t.shape)} {t.dtype}')
        if N != op.n or Tc.shape[1:] != (C, C):
            raise ValueError(f'stc_cell_graph: graphs are for N={op.n}, C={Tc.shape[1]}; got N={N}, C={C}')
        for (s_id, _, hs), w in zip(schedule, cin):
            Wg, bg, Wc, bc = stacks[s_id]
            rows = Ks * Tc.shape[0] * (w + H16)
            if Wg.shape != (rows, 2 * H16) or Wc.shape != (rows, H16) or (hs[0] == 'ext' and ext[hs[1]].shape[-1] != H16):
                raise ValueError(f'stc_cell_graph: parameter set {s_id} does not fit an input of {w} + {H16} columns')
        out_slot, inner_slot, pos, counts = _layout(schedule, cin, n_cells, outputs)
        out_stack = ref.new_empty(len(outputs), B, N, C, H16)        # the requested states are produced in place, stacked
        inner = ref.new_empty(max(1, n_cells - len(outputs)), B, N, C, H16)
        planes = ref.new_empty(6 if Ks == 3 else 5, n_cells, B, N, C, H16)   # U, R, Cand, R*H, S.(R*H) of every cell (order 3: + T_2(S).(R*H))
        learned = bool(ctx.needs_input_grad[6] or ctx.needs_input_grad[7])
        # per width group: the aggregate Zg of every cell and, for learned graphs, the slabs Z0, Z0c, Z1c the graph-gradient products read
        widths = (k.cell_small_zg_width(1), k.cell_small_zg_width(H16))
        if Ks == 3 and learned:
            raise ValueError('stc_cell_graph: the small-graph cell kernels take learned graphs at Chebyshev order 2 only')
        # (order 3: slot 1 holds the third slab T_2(S).[H | Xt | 0]; learned graphs -- order 2 only -- keep Z0, Z0c, Z1c in slots 1..3)
        slabs = [ref.new_empty(4 if learned else (2 if Ks == 3 else 1), max(1, counts[w]), B, N * C, widths[w]) for w in (0, 1)]
        g2 = op.source.second_order(ref.device) if Ks == 3 else None
        out_alias = _alias(out_stack)
        state = [out_alias[out_slot[j]] if j in out_slot else inner[inner_slot[j]] for j in range(n_cells)]
        U, R, Cand, RH, Zc, *Zc2 = (p.unbind(0) for p in planes.unbind(0))
        source = lambda src: ext[src[1]] if src[0] == 'ext' else state[src[1]]
        # Few samples: a cell step runs as a few launches over several workgroups per sample (each owning a contiguous range of row tiles)
        # instead of one launch with one workgroup per sample -- at the SF shape (batch 32) backward 79 -> ~45 us per cell in three
        # launches, forward 35 -> 24 us in two.  Dense learned graphs split alike: their aggregations are matrix products, in the forward over
        # the node tiles that cover a workgroup's own rows (so that the phase pairs still share a launch), in the backward over all workgroups.
        splits = k.cell_small_splits(B, N * C)
        fwd_splits = splits
        for j, (s_id, x, hs) in enumerate(schedule):
            Wg, bg, Wc, bc = stacks[s_id]
            w, i = pos[j]
            extra = dict(Z0=slabs[w][1, i], Z0c=slabs[w][2, i], Z1c=slabs[w][3, i]) if learned else {}
            if Ks == 3:
                extra = dict(graph2=(g2['fwd2_rowptr'], g2['fwd2_colidx'], g2['fwd2_val']), Zg2=slabs[w][1, i], Zc2=Zc2[0][j].view(B, N * C, H16))
            k.cell_small_fwd(op.fwd_rowptr, op.fwd_colidx, fwd_val, source(x), source(hs), Tc, Wg, bg, Wc, bc, U[j], R[j], Cand[j], state[j], RH[j],
                             slabs[w][0, i], Zc[j].view(B, N * C, H16), checked=False, splits=fwd_splits, **extra)
        ctx.save_for_backward(Tc, out_alias, inner, planes, slabs[0], slabs[1], *ext, *[p for st in stacks for p in st if p is not None])
        ctx.meta = (k, op, Ks, list(schedule), tuple(outputs), cin, [tuple(p is not None for p in st) for st in stacks], (B, N, C), len(ext), splits)
        ctx.out_stack_ref, ctx.out_stack_version = weakref.ref(out_stack), out_stack._version
        return out_stack

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_stack):
        k, op, Ks, schedule, outputs, cin, present, (B, N, C), n_ext, splits = ctx.meta
        stack = ctx.out_stack_ref()
        if stack is not None and stack._version != ctx.out_stack_version:
            raise RuntimeError('stc_cell_graph: the returned state stack was modified in place after the forward pass; the states saved for '
                               'backward share its storage (treat the stack as read-only, or clone it before editing)')
        Tc, out_alias, inner, planes, slabs_n, slabs_w, *rest = ctx.saved_tensors
        slabs = (slabs_n, slabs_w)
        need_Tc, need_val = ctx.needs_input_grad[6], ctx.needs_input_grad[7]
        learned = bool(need_Tc or need_val)
        ext, rest = rest[:n_ext], rest[n_ext:]
        stacks = []
        for pres in present:
            st = []
            for p in pres:
                st.append(rest.pop(0) if p else None)
            stacks.append(st)
        n_cells = len(schedule)
        out_slot, inner_slot, pos, counts = _layout(schedule, cin, n_cells, outputs)
        state = [out_alias[out_slot[j]] if j in out_slot else inner[inner_slot[j]] for j in range(n_cells)]
        # learned graphs: what the launches leave for the graph-gradient products, per width group (zeros where a cell is never reached):
        # the gradients of the two aggregated slabs, the gate and candidate pre-activation gradients
        dslab = [Tc.new_zeros(2, max(1, counts[w]), B, N * C, slabs[w].shape[-1]) for w in (0, 1)] if learned else None
        dyg = [Tc.new_zeros(max(1, counts[w]), B, N * C, 2 * H16) for w in (0, 1)] if learned else None
        dyc = [Tc.new_zeros(max(1, counts[w]), B, N * C, H16) for w in (0, 1)] if learned else None
        U, R, Cand, RH, Zc, *Zc2 = (p.unbind(0) for p in planes.unbind(0))
        source = lambda src: ext[src[1]] if src[0] == 'ext' else state[src[1]]
        Kc = Tc.shape[0]
        g2 = op.source.second_order(Tc.device) if Ks == 3 else None
        P = max(k.cell_small_params(Ks, Kc, w) for w in cin)
        dP = Tc.new_zeros(len(stacks), B * splits * k.cell_small_param_rows, P)   # parameter-gradient partials, every cell adds to its set's rows
        G = Tc.new_empty(n_cells, B, N, C, H16)                      # gradient owed to every cell's state
        owed = [False] * n_cells
        grad_stack = _c(grad_stack)
        for i, j in enumerate(outputs):
            G[j].copy_(grad_stack[i])
            owed[j] = True
        Gv, dPv = G.unbind(0), dP.unbind(0)
        for j in range(n_cells - 1, -1, -1):
            if not owed[j]:
                continue                                             # nothing downstream depends on this cell
            s_id, x, hs = schedule[j]
            Wg, bg, Wc, bc = stacks[s_id]
            dX = dH = None
            acc_x = acc_h = False
            late = None
            if hs[0] == 'cell':
                dH, acc_h = Gv[hs[1]], owed[hs[1]]
                owed[hs[1]] = True
            if x[0] == 'cell':
                if hs[0] == 'cell' and hs[1] == x[1]:                # one state as both inputs: the kernel's two outputs must not alias
                    dX, late = torch.empty_like(Gv[x[1]]), x[1]
                else:
                    dX, acc_x = Gv[x[1]], owed[x[1]]
                    owed[x[1]] = True
            w, i = pos[j]
            extra = dict(dZ1c=dslab[w][0, i], dZ1g=dslab[w][1, i], dYg=dyg[w][i], dYc=dyc[w][i]) if learned else {}
            if Ks == 3:
                extra = dict(graph2=(g2['bwd2_rowptr'], g2['bwd2_colidx'], g2['bwd2_val']), Zg2=slabs[w][1, i], Zc2=Zc2[0][j].view(B, N * C, H16))
            k.cell_small_bwd(op.bwd_rowptr, op.bwd_colidx, op.bwd_val, source(x), source(hs), Tc, Wg, Wc, U[j], R[j], Cand[j], RH[j], slabs[w][0, i],
                             Zc[j].view(B, N * C, H16), Gv[j], dX, acc_x, dH, acc_h, dPv[s_id], bg is not None, bc is not None, checked=False,
                             splits=splits, **extra)
            if late is not None:
                Gv[late].add_(dX)
        sums = dP.sum(1)                                             # (sets, P)
        flat = []
        for s_id, st in enumerate(stacks):
            w = next((cin[j] for j, sc in enumerate(schedule) if sc[0] == s_id), None)
            if w is None:
                flat += [None if p is None else torch.zeros_like(p) for p in st]
                continue
            nW, row = Ks * Kc * (w + H16), sums[s_id]
            dWg, dbg = row[:nW * 32].view(nW, 32), row[nW * 32:nW * 32 + 32]
            dWc, dbc = row[nW * 32 + 32:nW * 48 + 32].view(nW, H16), row[nW * 48 + 32:nW * 48 + 48]
            flat += [dWg, dbg if st[1] is not None else None, dWc, dbc if st[3] is not None else None]
        dT = dS = None
        if learned:
            dT, dS = _graph_gradients(k, Tc, Ks, stacks, schedule, cin, pos, counts, (B, N, C), slabs, dslab, dyg, dyc, need_Tc, need_val)
        return (None,) * 6 + (dT, dS) + (None,) * n_ext + tuple(flat)


def _graph_gradients(k, Tc, Ks, stacks, schedule, cin, pos, counts, dims, slabs, dslab, dyg, dyc, need_Tc, need_val):
    """(dT_c, d fwd_val) of a learned-graph backward pass from what the cell launches left (module docstring), per width group (index 0:
    narrow inputs, 1: 16-column inputs).  slabs[w] = (Zg, Z0, Z0c, Z1c), dslab[w] = (dZ1c, dZ1g) of every cell of the group, in the kernels'
    column order [H (16) | X (cin) | 0]; W's rows are re-ordered to match.  The sums over cells and samples run in ``graph_grad`` /
    ``mix_grad`` (stc_graph_grad_f32 / stc_mix_grad_f32: fp32 matrix products per plane, float64 accumulation):
      d fwd_val = sum_cells [dZ1g x Z0 + dZ1c x Z0c]                                      (every cell of a width at once)
      dT_c[c, d] = < W[(ks, c)], Q_ks[c, :, d, :] >,  Q_ks = Z_ks^T . dY                  (per parameter set and convolution)."""
    B, N, C = dims
    Kc = Tc.shape[0]
    # every product of the pass leaves its float64 partials in ONE (chunks, total) buffer, side by side, and one sum adds them all: per
    # product that was an allocation, a reduction and -- for dT_c -- a stack, three weight copies and an einsum of its own (~100 launches of a
    # few microseconds per step at the SF shape)
    graph_jobs = []                                               # (A, B, cells): dGs^T pieces
    if need_val:
        for w in (0, 1):
            if counts[w]:
                graph_jobs += [(dslab[w][1], slabs[w][1], counts[w]), (dslab[w][0], slabs[w][2], counts[w])]
    classes = {}                                                  # (width group, convolution) -> [(slab 0, slab 1, W, dY, first cell, step, cells)]
    dT_direct = []                                                # dT_c pieces formed directly on the matrix cores (stc_mix_dt_f32)
    if need_Tc:
        for s_id, (Wg, bg, Wc, bc) in enumerate(stacks):
            cells = [j for j, sc in enumerate(schedule) if sc[0] == s_id]
            if not cells:
                continue
            w, first = pos[cells[0]]
            where = [pos[j][1] for j in cells]
            step = where[1] - where[0] if len(where) > 1 else 1
            operands = ((slabs[w][1], slabs[w][0], Wg, dyg[w]), (slabs[w][2], slabs[w][3], Wc, dyc[w]))
            if step < 1 or any(b_ - a_ != step for a_, b_ in zip(where, where[1:])):
                # (a schedule STCGNN never builds: the set's cells are not evenly spaced inside their width group -- gather them)
                pick = lambda t: torch.stack([t[i] for i in where])
                operands = tuple((pick(s0), pick(s1), W, pick(dY)) for s0, s1, W, dY in operands)
                first, step = 0, 1
            for conv, (s0, s1, W, dY) in enumerate(operands):
                LP, Ho, cw = s0.shape[-1], W.shape[1], cin[cells[0]]
                if Ks == 2 and step == 1 and hasattr(k, 'mix_dT') and k.mix_dT_supported(Ks, Kc, C, LP, Ho):
                    # the set's cells are consecutive planes of their slabs: dT_c = sum over their rows of U_c . dY^T in ONE launch on tiles of
                    # floor(16 / C) nodes (U_c = [Z_0 | Z_1] . W_c re-formed inside) -- instead of Ks products Q = Z^T . dY with float64
                    # partials, their sum and a contraction with W per class (0.44 + ~0.2 ms of the 4.65 ms learned-graph SF step)
                    n = len(cells)
                    Wv = W.view(Ks, Kc, cw + H16, Ho)
                    Wp = W.new_zeros(Ks, Kc, LP, Ho)                 # W's rows in the slabs' column order [H (16) | X (cin) | 0]
                    Wp[:, :, :H16] = Wv[:, :, cw:]
                    Wp[:, :, H16:H16 + cw] = Wv[:, :, :cw]
                    rows = n * B * N
                    piece = Tc.new_empty(Kc, C, C)
                    k.mix_dT([s0[first:first + n].view(rows, C, LP), s1[first:first + n].view(rows, C, LP)], Wp.view(Ks * Kc * LP, Ho),
                             dY[first:first + n].view(rows, C, Ho), piece)
                    dT_direct.append(piece)
                    continue
                classes.setdefault((w, cin[cells[0]], conv), []).append((s0, s1, W, dY, first, step, len(cells)))
    block = lambda s0, dY: C * s0.shape[-1] * C * dY.shape[-1]
    total = sum(Ks * block(e[0], e[3]) for es in classes.values() for e in es)
    dS = None
    if need_val:                                                  # (N x N blocks are small: more, shorter workgroups -- a buffer of their own)
        dS = torch.zeros(N, N, dtype=torch.float64, device=Tc.device)
        if graph_jobs:
            gpart = k.grad_partials(Tc, len(graph_jobs) * N * N, chunks=max(1, min(256, min(n for _, _, n in graph_jobs) * B)))
            for i, (A, Bm, n_sel) in enumerate(graph_jobs):
                k.graph_grad(A, Bm, 0, 1, n_sel, N, into=(gpart, i * N * N))
            dS = gpart.view(-1, len(graph_jobs), N, N).sum((0, 1))
    sums = None
    if total:
        part, off = k.grad_partials(Tc, total), 0
        for es in classes.values():
            for s0, s1, W, dY, first, step, n_sel in es:
                for slab in (s0, s1)[:Ks]:
                    k.mix_grad(slab, dY, first, step, n_sel, N, into=(part, off))
                    off += block(s0, dY)
        sums = part.sum(0)
    dT = torch.zeros(Tc.shape, dtype=torch.float64, device=Tc.device) if need_Tc else None
    if dT_direct:
        dT += torch.stack(dT_direct).sum(0)
    off = 0
    for (w, cw, conv), es in classes.items():
        LP, Ho, L = es[0][0].shape[-1], es[0][2].shape[1], cw + H16
        size = len(es) * Ks * block(es[0][0], es[0][3])
        Q = sums[off:off + size].view(len(es), Ks, C, LP, C, Ho)
        off += size
        Wv = torch.stack([e[2] for e in es]).view(len(es), Ks, Kc, L, Ho)
        Wp = Wv.new_zeros(len(es), Ks, Kc, LP, Ho)              # W's rows in the slabs' column order [H (16) | X (cin) | 0]
        Wp[..., :H16, :] = Wv[..., cw:, :]
        Wp[..., H16:H16 + cw, :] = Wv[..., :cw, :]
        # sum_{p,s,l,o} Q[p,s,c,l,d,o] W[p,s,k,l,o] as a product and a sum (as an einsum: a float64 GEMM with a 50-element result, 190 us)
        dT += (Q[:, :, None] * Wp.double()[:, :, :, None, :, None, :]).sum((0, 1, 4, 6))
    return (None if dT is None else dT.to(Tc.dtype)), (None if dS is None else dS.to(Tc.dtype).reshape(-1))


def stc_small_graph(k, op: SpatialOperand, Tc, Ks: int, schedule, outputs, ext, stacks):
    flat = [p for st in stacks for p in st]
    return _StcSmallGraph.apply(k, op, Ks, list(schedule), list(outputs), len(ext), Tc, op.fwd_val, *ext, *flat)
