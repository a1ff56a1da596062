"""Two-ring patch aggregation (csrc/stc_spmm_ring2.hip, ``stc_ring2_sum_f32``): the state-gradient sum with its blend backward and the transpose
aggregation of the candidate's gradient in one launch, without the dY plane -- against the CPU twin (the two aggregations it replaces, composed)
and against the two HIP launches it replaces; host plan properties on CPU."""
import numpy as np
import pytest
import torch

from oracle.kernel_emul import EmulatedKernels
from stc_hip import CsrGraph
from stc_hip.graph import RING2_FIRST, RING2_INTERIOR, RING2_SECOND, csr_operand
from tests.conftest import rel_err

EM = EmulatedKernels()


def _dense(graph):
    return graph.to_dense().double().numpy()


def _renumbered(H, W, seed=1234):
    """The grid under a seeded random node order, renumbered by the host (reverse Cuthill-McKee): not a lattice in its numbering any more, so
    its two-ring patches are clusters grown under the ring bounds (graph.py _ring2_groups)."""
    graph, order = CsrGraph.queen_grid(H, W, normalize=True, permute_seed=seed).with_locality(collective=False)
    assert order is not None                                          # (the random order costs enough row fetches for the host to renumber)
    return graph


@pytest.mark.parametrize('H,W,renumbered', [(12, 20, False), (9, 33, False), (31, 8, False), (24, 30, True), (40, 17, True)])
def test_two_ring_plan_reproduces_the_matrix(H, W, renumbered):
    """The plan's tables ARE the matrix: level 2 rebuilds every row of S over the first ring's slots, level 1 every first-ring row over the
    staged rows; every row of S is some patch's own row exactly once; rings within their sizes.  Lattices (4 x 8 tiles) and a grid whose
    numbering is not a lattice (ring-bounded clusters)."""
    graph = _renumbered(H, W) if renumbered else CsrGraph.queen_grid(H, W, normalize=True)
    h = graph._host
    assert 'bwd_r2_l2' in h
    S = _dense(graph)                                                 # Gs itself: the backward operand
    l2, l1, own, t1, t2 = (h[f'bwd_r2_{k}'] for k in ('l2', 'l1', 'own', 't1', 't2'))
    n = H * W
    seen = np.zeros(n, dtype=int)
    for p in range(l2.shape[0]):
        rows = own[p][own[p] >= 0]
        seen[rows] += 1
        first = l1[p]
        assert (first[:len(rows)] & 0x3FFFFFFF == rows).all() and ((first[:len(rows)] >> 30) & 1).all()      # own rows lead, flagged
        live = first >= 0
        assert not ((first[live][len(rows):] >> 30) & 1).any()
        for r, u in enumerate(rows):                                  # level 2
            got = np.zeros(n)
            for off, bits in t2[p, r]:
                got[first[off // 512] & 0x3FFFFFFF] += np.int32(bits).view(np.float32)
            assert np.allclose(got, S[u], atol=1e-7)
        for slot in np.nonzero(live)[0]:                              # level 1
            u = first[slot] & 0x3FFFFFFF
            got = np.zeros(n)
            for off, bits in t1[p, slot]:
                got[l2[p, off // 512]] += np.int32(bits).view(np.float32)
            assert np.allclose(got, S[u], atol=1e-7)
    assert (seen == 1).all()
    assert l1.shape[1] == RING2_FIRST and l2.shape[1] == RING2_SECOND and own.shape[1] == RING2_INTERIOR
    if renumbered:
        assert 'bwd_pt_src' in h and n / l2.shape[0] >= 16          # the patch form's clusters did not fit the rings; these do, at a useful size


def test_graphs_whose_rings_do_not_fit_get_no_two_ring_plan():
    g = torch.Generator().manual_seed(0)
    dense = (torch.rand(200, 200, generator=g) < 0.08).float()        # ~16 entries per row: wider than the tables
    assert 'bwd_r2_l2' not in CsrGraph.from_dense(dense)._host
    assert csr_operand(CsrGraph.from_dense(dense), torch.device('cpu')).bwd_ring2 is None


@pytest.mark.gpu
@pytest.mark.parametrize('H,W,B,n_add,dual', [(12, 20, 2, 1, False), (40, 56, 2, 3, True), (9, 33, 1, 0, False), (31, 8, 3, 5, True), (224, 224, 1, 2, False)])
def test_two_ring_sum_on_the_gpu(H, W, B, n_add, dual):
    from stc_hip._lib import HipKernels
    hip = HipKernels()
    C, h = 32, 16
    graph = CsrGraph.queen_grid(H, W, normalize=True)
    n = H * W
    op_c, op_g = csr_operand(graph, torch.device('cpu')), csr_operand(graph, torch.device('cuda'))
    g = torch.Generator().manual_seed(H * W + n_add)
    rnd = lambda: torch.randn(B, n, C, h, generator=g)
    X, X2 = rnd(), (rnd() if dual else None)
    adds = [rnd() for _ in range(n_add)]
    U, Cand = torch.sigmoid(rnd()), torch.tanh(rnd())
    Y_w, Z_w = torch.empty(B, n, C, h, dtype=torch.float64), torch.empty(B, n, C, h, dtype=torch.float64)
    d = lambda t: None if t is None else t.double()
    EM.ring2_sum(op_c.bwd_rowptr, op_c.bwd_colidx, op_c.bwd_val.double(), None, d(X), d(X2), [d(t) for t in adds], d(U), d(Cand), Y_w, Z_w)
    cu = lambda t: None if t is None else t.cuda()
    Y, Z = torch.full((B, n, C, h), float('nan')).cuda(), torch.full((B, n, C, h), float('nan')).cuda()
    hip.ring2_sum(op_g.bwd_rowptr, op_g.bwd_colidx, op_g.bwd_val, op_g.bwd_ring2, cu(X), cu(X2), [cu(t) for t in adds], cu(U), cu(Cand), Y, Z)
    assert rel_err(Y, Y_w) < 1e-6 and rel_err(Z, Z_w) < 1e-6
    # the two launches it replaces, on the same operands
    Y2, dY2, Z2 = (torch.empty(B, n, C, h).cuda() for _ in range(3))
    hip.spmm_sum(op_g.bwd_rowptr, op_g.bwd_colidx, op_g.bwd_val, op_g.bwd_plan, cu(X), cu(X2), [(cu(t), 0) for t in adds], Y2, blend=(cu(U), cu(Cand), dY2))
    hip.csr_spmm(op_g.bwd_rowptr, op_g.bwd_colidx, op_g.bwd_val, n, n, dY2.view(B, n, C * h), None, Z2.view(B, n, C * h), 1.0, 0.0, plan=op_g.bwd_plan)
    assert rel_err(Y, Y2) < 1e-6 and rel_err(Z, Z2) < 1e-6
    # reproducible
    Y3, Z3 = torch.empty_like(Y), torch.empty_like(Z)
    hip.ring2_sum(op_g.bwd_rowptr, op_g.bwd_colidx, op_g.bwd_val, op_g.bwd_ring2, cu(X), cu(X2), [cu(t) for t in adds], cu(U), cu(Cand), Y3, Z3)
    assert torch.equal(Y3, Y) and torch.equal(Z3, Z)


@pytest.mark.gpu
@pytest.mark.parametrize('H,W,B', [(12, 20, 2), (40, 56, 2), (9, 33, 1), (224, 224, 1)])
def test_two_ring_blend_on_the_gpu(H, W, B):
    """stc_ring2_blend_f32 against the CPU twin (float64) and against the two HIP launches it replaces (blend aggregation, then S.Hnew)."""
    from stc_hip._lib import HipKernels
    hip = HipKernels()
    C, h = 32, 16
    graph = CsrGraph.queen_grid(H, W, normalize=True)
    n = H * W
    op_c, op_g = csr_operand(graph, torch.device('cpu')), csr_operand(graph, torch.device('cuda'))
    assert op_g.fwd_ring2 is not None
    g = torch.Generator().manual_seed(H * W)
    rnd = lambda: torch.randn(B, n, C, h, generator=g)
    Bm, A, U, Hp = rnd(), rnd(), torch.sigmoid(rnd()), torch.tanh(rnd())
    d = lambda t: t.double()
    want = [torch.empty(B, n, C, h, dtype=torch.float64) for _ in range(3)]
    EM.ring2_blend(op_c.fwd_rowptr, op_c.fwd_colidx, op_c.fwd_val.double(), None, d(Bm), d(A), d(U), d(Hp), *want)
    cu = lambda t: t.cuda()
    got = [torch.full((B, n, C, h), float('nan')).cuda() for _ in range(3)]
    hip.ring2_blend(op_g.fwd_rowptr, op_g.fwd_colidx, op_g.fwd_val, op_g.fwd_ring2, cu(Bm), cu(A), cu(U), cu(Hp), *got)
    for a, w in zip(got, want):
        assert rel_err(a, w) < 2e-6
    Cand2, Hn2, SH2 = (torch.empty(B, n, C, h).cuda() for _ in range(3))
    hip.spmm_blend_fwd(op_g.fwd_rowptr, op_g.fwd_colidx, op_g.fwd_val, op_g.fwd_plan, cu(Bm), cu(A), cu(U), cu(Hp), Cand2, Hn2)
    hip.csr_spmm(op_g.fwd_rowptr, op_g.fwd_colidx, op_g.fwd_val, n, n, Hn2.view(B, n, C * h), None, SH2.view(B, n, C * h), 1.0, 0.0, plan=op_g.fwd_plan)
    for a, w in zip(got, (Cand2, Hn2, SH2)):
        assert rel_err(a, w) < 2e-6


def _chain_cases():
    # (H, W, B, dual, n_add1, n_add0, store V, alpha1, alpha2): the forward recurrence, the transposes the order-3 backward asks for, the limits
    return [(12, 20, 2, False, 0, 1, True, 1.0, 2.0), (40, 56, 2, True, 2, 4, False, 2.0, 1.0), (9, 33, 1, False, 1, 2, False, 2.0, 1.0),
            (31, 8, 3, True, 2, 5, True, -0.5, 3.0), (100, 100, 4, True, 2, 5, False, 2.0, 1.0), (224, 224, 1, False, 0, 1, True, 1.0, 2.0)]


def test_chain_twin_is_the_order3_recurrence_and_its_transpose():
    """CPU: ``ring2_chain`` with the forward's arguments is [S.X, 2 S.(S.X) - X] (reference STC_GNN.py:24-29 on the feature side), and with the
    backward's arguments the transpose of  (d0, d1, d2) -> sum_n T_n(S)^T d_n  -- checked against dense float64 matrices."""
    graph = CsrGraph.queen_grid(7, 9, normalize=True)
    op = csr_operand(graph, torch.device('cpu'))
    B, n, C, h = 2, 63, 4, 16
    g = torch.Generator().manual_seed(5)
    rnd = lambda: torch.randn(B, n, C, h, generator=g, dtype=torch.float64)
    S = torch.zeros(n, n, dtype=torch.float64)                       # the forward operand as a dense matrix, from its CSR arrays
    rp, ci, vv = op.fwd_rowptr.long(), op.fwd_colidx.long(), op.fwd_val.double()
    for r in range(n):
        S[r, ci[rp[r]:rp[r + 1]]] = vv[rp[r]:rp[r + 1]]
    mul = lambda M, t: torch.einsum('uv,bvcf->bucf', M, t)
    X = rnd()
    T1, T2 = torch.empty_like(X), torch.empty_like(X)
    EM.ring2_chain(op.fwd_rowptr, op.fwd_colidx, op.fwd_val.double(), None, X, None, 1.0, [], T1, 2.0, [(X, -1.0)], T2)
    assert torch.allclose(T1, mul(S, X), atol=1e-12) and torch.allclose(T2, 2 * mul(S, mul(S, X)) - X, atol=1e-12)
    d0, d1, d2 = rnd(), rnd(), rnd()
    out = torch.empty_like(X)
    EM.ring2_chain(op.bwd_rowptr, op.bwd_colidx, op.bwd_val.double(), None, d2, None, 2.0, [d1], None, 1.0, [(d0, 1.0), (d2, -1.0)], out)
    St = S.T
    want = d0 + mul(St, d1) + (2 * mul(St, mul(St, d2)) - d2)          # T_0^T d0 + T_1^T d1 + T_2^T d2
    assert torch.allclose(out, want, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('H,W,B,dual,n1,n0,keep,alpha1,alpha2', _chain_cases())
def test_two_ring_chain_on_the_gpu(H, W, B, dual, n1, n0, keep, alpha1, alpha2):
    """stc_ring2_chain_f32 against the CPU twin (float64) and against the two HIP launches it replaces (stc_spmm_sum_f32 twice)."""
    from stc_hip._lib import HipKernels
    hip = HipKernels()
    C, h = 32, 16
    graph = CsrGraph.queen_grid(H, W, normalize=True)
    n = H * W
    op_c, op_g = csr_operand(graph, torch.device('cpu')), csr_operand(graph, torch.device('cuda'))
    g = torch.Generator().manual_seed(H * W + n1 + n0)
    rnd = lambda: torch.randn(B, n, C, h, generator=g)
    X, X2 = rnd(), (rnd() if dual else None)
    add1 = [rnd() for _ in range(n1)]
    add0 = [(rnd(), s) for s in ([1.0, -1.0, 1.0, -2.5, 0.5][:n0])]
    d = lambda t: None if t is None else t.double()
    V_w = torch.empty(B, n, C, h, dtype=torch.float64)
    Z_w = torch.empty(B, n, C, h, dtype=torch.float64)
    EM.ring2_chain(op_c.bwd_rowptr, op_c.bwd_colidx, op_c.bwd_val.double(), None, d(X), d(X2), alpha1, [d(t) for t in add1], V_w, alpha2,
                   [(d(t), s) for t, s in add0], Z_w)
    cu = lambda t: None if t is None else t.cuda()
    V = torch.full((B, n, C, h), float('nan')).cuda() if keep else None
    Z = torch.full((B, n, C, h), float('nan')).cuda()
    hip.ring2_chain(op_g.bwd_rowptr, op_g.bwd_colidx, op_g.bwd_val, op_g.bwd_ring2, cu(X), cu(X2), alpha1, [cu(t) for t in add1], V, alpha2,
                    [(cu(t), s) for t, s in add0], Z)
    assert rel_err(Z, Z_w) < 2e-6 and (V is None or rel_err(V, V_w) < 2e-6)
    V2, Z2 = torch.empty(B, n, C, h).cuda(), torch.empty(B, n, C, h).cuda()
    hip.spmm_sum(op_g.bwd_rowptr, op_g.bwd_colidx, op_g.bwd_val, op_g.bwd_plan, cu(X), cu(X2), [(cu(t), 0) for t in add1], V2, alpha=alpha1)
    hip.spmm_sum(op_g.bwd_rowptr, op_g.bwd_colidx, op_g.bwd_val, op_g.bwd_plan, V2, None, [(cu(t), 0, s) for t, s in add0], Z2, alpha=alpha2)
    assert rel_err(Z, Z2) < 2e-6 and (V is None or rel_err(V, V2) < 2e-6)
    Z3 = torch.empty_like(Z)
    hip.ring2_chain(op_g.bwd_rowptr, op_g.bwd_colidx, op_g.bwd_val, op_g.bwd_ring2, cu(X), cu(X2), alpha1, [cu(t) for t in add1], None, alpha2,
                    [(cu(t), s) for t, s in add0], Z3)
    assert torch.equal(Z3, Z)                                          # reproducible, with or without the stored V


@pytest.mark.gpu
def test_two_ring_launches_on_a_renumbered_grid_on_the_gpu():
    """The three entry points on ring-bounded CLUSTERS (a grid under a random node order, renumbered by the host: patches of 24 - 32 rows, ragged
    rings) against the launches they replace."""
    from stc_hip._lib import HipKernels
    hip = HipKernels()
    H, W, B, C, h = 48, 56, 2, 32, 16
    graph = _renumbered(H, W)
    n = H * W
    op = csr_operand(graph, torch.device('cuda'))
    assert op.bwd_ring2 is not None and op.fwd_ring2 is not None
    g = torch.Generator().manual_seed(11)
    rnd = lambda: torch.randn(B, n, C, h, generator=g).cuda()
    bwd, fwd = (op.bwd_rowptr, op.bwd_colidx, op.bwd_val), (op.fwd_rowptr, op.fwd_colidx, op.fwd_val)
    v3 = lambda t: t.view(B, n, C * h)
    # sum
    X, add, U, Cand = rnd(), rnd(), torch.sigmoid(rnd()), torch.tanh(rnd())
    Y, Z, Y2, dY2, Z2 = (torch.empty(B, n, C, h).cuda() for _ in range(5))
    hip.ring2_sum(*bwd, op.bwd_ring2, X, None, [add], U, Cand, Y, Z)
    hip.spmm_sum(*bwd, op.bwd_plan, X, None, [(add, 0)], Y2, blend=(U, Cand, dY2))
    hip.csr_spmm(*bwd, n, n, v3(dY2), None, v3(Z2), 1.0, 0.0, plan=op.bwd_plan)
    assert rel_err(Y, Y2) < 1e-6 and rel_err(Z, Z2) < 1e-6
    # blend
    Bm, A, Hp = rnd(), rnd(), torch.tanh(rnd())
    got = [torch.empty(B, n, C, h).cuda() for _ in range(3)]
    want = [torch.empty(B, n, C, h).cuda() for _ in range(3)]
    hip.ring2_blend(*fwd, op.fwd_ring2, Bm, A, U, Hp, *got)
    hip.spmm_blend_fwd(*fwd, op.fwd_plan, Bm, A, U, Hp, want[0], want[1])
    hip.csr_spmm(*fwd, n, n, v3(want[1]), None, v3(want[2]), 1.0, 0.0, plan=op.fwd_plan)
    for a, w in zip(got, want):
        assert rel_err(a, w) < 2e-6
    # chain: the forward recurrence
    T1, T2, S1, S2 = (torch.empty(B, n, C, h).cuda() for _ in range(4))
    hip.ring2_chain(*fwd, op.fwd_ring2, X, None, 1.0, [], T1, 2.0, [(X, -1.0)], T2)
    hip.csr_spmm(*fwd, n, n, v3(X), None, v3(S1), 1.0, 0.0, plan=op.fwd_plan)
    hip.csr_spmm(*fwd, n, n, v3(S1), v3(X), v3(S2), 2.0, -1.0, plan=op.fwd_plan)
    assert rel_err(T1, S1) < 2e-6 and rel_err(T2, S2) < 2e-6


def test_dispatch_limit_of_the_32_bit_piece_offsets():
    """The two-ring kernels address a plane by 32-bit piece offsets (< 2^28 pieces, checked at the entry points): the executor asks first and
    keeps the two launches beyond (twin and binding answer alike)."""
    from stc_hip._lib import HipKernels
    for k in (EM, HipKernels):
        assert k.ring2_fits(5, 50176, 32, 16) and k.ring2_fits(4, 10000, 32, 16)
        assert not k.ring2_fits(60, 50176, 32, 16)                   # 3.85e8 pieces
        assert not k.ring2_fits(4, 10000, 4, 16) and not k.ring2_fits(4, 10000, 32, 8)      # rows that are not whole 512-byte chunks / another hidden size


@pytest.mark.gpu
@pytest.mark.parametrize('K,renumbered', [(2, False), (2, True), (3, False), (3, True)])
def test_cell_graph_with_and_without_the_two_ring_launches(K, renumbered, monkeypatch):
    """Three chained planar cells through ``ops.stc_cell_graph`` (states consumed as X and as H, so every two-ring form of the order is on the
    path: sum + blend at order 2, chain forward and transposed at order 3) with the two-ring launches on and off: same states, same gradients."""
    from stc_hip import ops
    H, W, B, C, h = 40, 48, 2, 32, 16
    graph = _renumbered(H, W) if renumbered else CsrGraph.queen_grid(H, W, normalize=True)
    n = H * W
    op = csr_operand(graph, torch.device('cuda'))
    assert op.bwd_ring2 is not None and op.fwd_ring2 is not None and op.ring2_clusters == renumbered
    g = torch.Generator().manual_seed(K)
    Gc = torch.rand(C, C, generator=g)
    Tc = ops.cheby_dense((Gc / Gc.sum(1, keepdim=True)).cuda(), K)
    rows = K * K * 2 * h
    base = [torch.randn(B, n, C, h, generator=g), torch.tanh(torch.randn(B, n, C, h, generator=g)),
            torch.randn(rows, 2 * h, generator=g) * 0.1, torch.zeros(2 * h), torch.randn(rows, h, generator=g) * 0.1, torch.zeros(h),
            torch.randn(B, n, C, h, generator=g)]
    schedule = [(0, ('ext', 0), ('ext', 1)), (0, ('cell', 0), ('cell', 0)), (0, ('cell', 1), ('cell', 0))]
    assert ops.cell_graph_supported(op, Tc, K, C, h, [h])

    def run(on):
        monkeypatch.setattr(ops, '_RING2', on)
        monkeypatch.setattr(ops, '_RING2_FWD', on)
        Xt, Ht, Wg, bg, Wc, bc, R = (t.clone().cuda() for t in base)
        leaves = [t.requires_grad_() for t in (Xt, Ht, Wg, bg, Wc, bc)]
        out = ops.stc_cell_graph(op, Tc, K, schedule, [2], [Xt, Ht], [(Wg, bg, Wc, bc)])[0]
        (out * R).sum().backward()
        return [out.detach()] + [t.grad for t in leaves]

    seen = []
    real = type(ops.kernels())
    for name in ('ring2_sum', 'ring2_blend', 'ring2_chain'):
        fn = getattr(real, name)
        monkeypatch.setattr(real, name, (lambda f, nm: lambda self, *a, **kw: (seen.append(nm), f(self, *a, **kw))[1])(fn, name))
    with_ring2 = run(True)
    used = set(seen)
    seen.clear()
    without = run(False)
    assert not seen
    assert used == ({'ring2_chain'} if K == 3 else ({'ring2_sum'} if renumbered else {'ring2_sum', 'ring2_blend'})), used
    for a, b in zip(with_ring2, without):
        assert (a is None) == (b is None)                              # (external inputs get no gradient from the cell graph)
        if a is not None:
            assert rel_err(a, b) < 3e-6
