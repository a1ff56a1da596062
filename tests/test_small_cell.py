"""The small-graph cell kernels (stc_cell_small_fwd/bwd_f32: one STC_Cell step per launch, reference STC_GNN.py:65-79).

CPU: the kernels' CPU twin (oracle/kernel_emul.py) against the oracle's STC_Cell and ITS autograd, in float64 -- the twin restates the
launch's contract (saved aggregates in [H | X | 0] order, per-sample parameter-gradient partials that are added to, accumulate flags).
GPU (-m gpu): the HIP kernels against the twin on the same inputs, through the C ABI; bound 1e-5 (max-norm relative, as everywhere).
"""
import pytest
import torch

from oracle import stc_oracle as oracle
from oracle.kernel_emul import EmulatedKernels
from stc_hip import CsrGraph
from stc_hip.graph import csr_operand, dense_operand
from tests.conftest import rel_err

EM = EmulatedKernels()
TOL = 1e-5


def _graph(N, seed, dense=False):
    g = torch.Generator().manual_seed(seed)
    if dense:
        G = torch.softmax(torch.randn(N, N, generator=g), -1)
    else:
        G = (torch.rand(N, N, generator=g) < min(1.0, 6.0 / N)).float() * torch.rand(N, N, generator=g)
        G[0] = 0.0                                                  # an empty row, an empty column
        G[:, N - 1] = 0.0
        G = G + 0.5 * torch.eye(N)
        G[0, 0] = 0.0
    return CsrGraph.from_dense(G)


def _inputs(B, N, C, cin, seed, dtype=torch.float32, bias=True, K=2):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dtype)
    L = cin + 16
    Gc = torch.softmax(torch.randn(C, C, generator=g), -1).to(dtype)
    Tc = torch.stack([torch.eye(C, dtype=dtype), Gc] + ([2 * Gc @ Gc - torch.eye(C, dtype=dtype)] if K == 3 else []))     # cheby_poly, STC_GNN.py:24-29
    ws = 0.3 if K == 2 else 0.15                 # (order 3: 9 blocks instead of 4 and |T_2| up to ~2 -- the same pre-activation spread as at order 2)
    return dict(X=r(B, N, C, cin), H=0.5 * r(B, N, C, 16), Gc=Gc, Tc=Tc, Wg=ws * r(K * K * L, 32), bg=0.1 * r(32) if bias else None,
                Wc=ws * r(K * K * L, 16), bc=0.1 * r(16) if bias else None, dHnew=r(B, N, C, 16))


def _buffers(B, N, C, cin, dtype, k, K=2):
    new = lambda *s: torch.full(s, float('nan'), dtype=dtype)
    P = k.cell_small_params(K, K, cin) + 5                          # a leading dimension larger than needed
    third = dict(Zg2=new(B, N * C, k.cell_small_zg_width(cin)), Zc2=new(B, N * C, 16)) if K == 3 else {}
    return dict(U=new(B, N, C, 16), R=new(B, N, C, 16), Cand=new(B, N, C, 16), Hnew=new(B, N, C, 16), RH=new(B, N, C, 16),
                Zg=new(B, N * C, k.cell_small_zg_width(cin)), Zc=new(B, N * C, 16), **third), P


def _run(k, op, t, buf, P, to, acc_x=False, acc_h=False, want_x=True, want_h=True, want_dumps=False, splits=1):
    """Forward + backward through kernel set ``k`` on device / dtype converter ``to``; returns plain CPU tensors."""
    d = {n: (None if v is None else to(v)) for n, v in t.items()}
    b = {n: to(v) for n, v in buf.items()}
    dumps = {n: to(torch.full_like(buf['Zg'], float('nan'))) for n in ('Z0', 'dZ1c', 'dZ1g')} if want_dumps else {}
    third_f = third_b = {}
    if t['Tc'].shape[0] == 3:                    # order 3: T_2(S) = 2 S^2 - I as the second graph, in each launch's orientation
        g2 = op.source.second_order(b['Zg'].device)
        third_f = dict(graph2=tuple(g2[f'fwd2_{n}'] for n in ('rowptr', 'colidx', 'val')), Zg2=b['Zg2'], Zc2=b['Zc2'])
        third_b = dict(third_f, graph2=tuple(g2[f'bwd2_{n}'] for n in ('rowptr', 'colidx', 'val')))
        if t['H'].dtype == torch.float64:        # (the twin in float64: the second graph's values formed in float64 as well)
            dense2 = lambda G: 2 * G @ G - torch.eye(G.shape[0], dtype=torch.float64)
            G = op.source.to_dense().double()
            vals = lambda M, rp, ci: M[torch.repeat_interleave(torch.arange(M.shape[0]), (rp[1:] - rp[:-1]).long()), ci.long()]
            third_f['graph2'] = third_f['graph2'][:2] + (vals(dense2(G).t().contiguous(), *third_f['graph2'][:2]),)
            third_b['graph2'] = third_b['graph2'][:2] + (vals(dense2(G), *third_b['graph2'][:2]),)
    k.cell_small_fwd(op.fwd_rowptr, op.fwd_colidx, to(op.fwd_val), d['X'], d['H'], d['Tc'], d['Wg'], d['bg'], d['Wc'], d['bc'],
                     b['U'], b['R'], b['Cand'], b['Hnew'], b['RH'], b['Zg'], b['Zc'], Z0=dumps.get('Z0'), splits=splits, **third_f)
    B = d['H'].shape[0]
    dX = to(torch.full(t['X'].shape, 0.25, dtype=t['X'].dtype)) if want_x else None
    dH = to(torch.full(t['H'].shape, -0.5, dtype=t['H'].dtype)) if want_h else None
    dP = to(torch.full((B * splits * k.cell_small_param_rows, P), 0.125, dtype=t['H'].dtype))
    k.cell_small_bwd(op.bwd_rowptr, op.bwd_colidx, to(op.bwd_val), d['X'], d['H'], d['Tc'], d['Wg'], d['Wc'], b['U'], b['R'], b['Cand'], b['RH'],
                     b['Zg'], b['Zc'], d['dHnew'], dX, acc_x, dH, acc_h, dP, t['bg'] is not None, t['bc'] is not None,
                     dZ1c=dumps.get('dZ1c'), dZ1g=dumps.get('dZ1g'), dYg=dumps.setdefault('dYg', to(torch.full((B, buf['Zg'].shape[1], 32), float('nan'),
                                                                                                    dtype=t['H'].dtype))) if want_dumps else None,
                     splits=splits, **third_b)
    out = dict(b, dX=dX, dH=dH, dP=dP, **dumps)
    return {n: (None if v is None else v.detach().cpu()) for n, v in out.items()}


def _split_params(dP, cin, K=2):
    L = cin + 16
    nW = K * K * L
    return (dP[:, :nW * 32].sum(0).view(nW, 32), dP[:, nW * 32:nW * 32 + 32].sum(0), dP[:, nW * 32 + 32:nW * 48 + 32].sum(0).view(nW, 16),
            dP[:, nW * 48 + 32:nW * 48 + 48].sum(0), dP[:, nW * 48 + 48:])


@pytest.mark.parametrize('B,N,C,cin,bias', [(3, 12, 5, 1, True), (2, 9, 4, 16, True), (2, 7, 3, 3, False)])
def test_small_cell_twin_is_the_reference_cell(B, N, C, cin, bias):
    """oracle.stc_cell (dense Gs, the reference's op order) and its autograd, float64, against the twin's forward / backward."""
    dt = torch.float64
    graph = _graph(N, seed=N + cin)
    op = csr_operand(graph, torch.device('cpu'))
    t = _inputs(B, N, C, cin, seed=7 * N + C, dtype=dt, bias=bias)
    buf, P = _buffers(B, N, C, cin, dt, EM)
    got = _run(EM, op, t, buf, P, lambda v: v.clone().to(dt) if v.is_floating_point() else v)
    leaves = {n: t[n].clone().requires_grad_(True) for n in ('X', 'H', 'Wg', 'Wc') + (('bg', 'bc') if bias else ())}
    Hnew = oracle.stc_cell(graph.to_dense().to(dt), t['Gc'], leaves['X'], leaves['H'], leaves['Wg'], leaves.get('bg'), leaves['Wc'], leaves.get('bc'), 2, 2)
    Hnew.backward(t['dHnew'])
    assert rel_err(got['Hnew'], Hnew.detach()) < 1e-12
    dWg, dbg, dWc, dbc, rest = _split_params(got['dP'] - 0.125, cin)
    assert float(rest.abs().max()) == 0.0                           # columns beyond the parameters: untouched
    assert rel_err(got['dX'], leaves['X'].grad) < 1e-12 and rel_err(got['dH'], leaves['H'].grad) < 1e-12
    assert rel_err(dWg, leaves['Wg'].grad) < 1e-12 and rel_err(dWc, leaves['Wc'].grad) < 1e-12
    if bias:
        assert rel_err(dbg, leaves['bg'].grad) < 1e-12 and rel_err(dbc, leaves['bc'].grad) < 1e-12
    else:
        assert float(dbg.abs().max()) == 0.0 and float(dbc.abs().max()) == 0.0


@pytest.mark.parametrize('B,N,C,cin,bias', [(3, 12, 5, 1, True), (2, 9, 4, 16, True), (2, 7, 3, 3, False)])
def test_small_cell_twin_at_order_3_is_the_reference_cell(B, N, C, cin, bias):
    """Chebyshev order 3 (Main.py:24 -cheby_order; cheby_poly STC_GNN.py:24-29): the twin takes T_2(S) = 2 S^2 - I as a SECOND graph, formed on
    the matrix side as the reference forms it -- against oracle.stc_cell at Ks = Kc = 3 and its autograd, float64."""
    dt = torch.float64
    graph = _graph(N, seed=N + cin)
    op = csr_operand(graph, torch.device('cpu'))
    t = _inputs(B, N, C, cin, seed=7 * N + C, dtype=dt, bias=bias, K=3)
    buf, P = _buffers(B, N, C, cin, dt, EM, K=3)
    got = _run(EM, op, t, buf, P, lambda v: v.clone().to(dt) if v.is_floating_point() else v)
    leaves = {n: t[n].clone().requires_grad_(True) for n in ('X', 'H', 'Wg', 'Wc') + (('bg', 'bc') if bias else ())}
    Hnew = oracle.stc_cell(graph.to_dense().to(dt), t['Gc'], leaves['X'], leaves['H'], leaves['Wg'], leaves.get('bg'), leaves['Wc'], leaves.get('bc'), 3, 3)
    Hnew.backward(t['dHnew'])
    assert rel_err(got['Hnew'], Hnew.detach()) < 1e-12
    dWg, dbg, dWc, dbc, rest = _split_params(got['dP'] - 0.125, cin, K=3)
    assert float(rest.abs().max()) == 0.0
    assert rel_err(got['dX'], leaves['X'].grad) < 1e-12 and rel_err(got['dH'], leaves['H'].grad) < 1e-12
    assert rel_err(dWg, leaves['Wg'].grad) < 1e-12 and rel_err(dWc, leaves['Wc'].grad) < 1e-12
    if bias:
        assert rel_err(dbg, leaves['bg'].grad) < 1e-12 and rel_err(dbc, leaves['bc'].grad) < 1e-12


def test_second_chebyshev_matrix_of_a_graph():
    """CsrGraph.second_order: 2 S^2 - I in both orientations against the dense product, on a ragged graph with an empty row and an empty column."""
    graph = _graph(30, seed=4)
    G = graph.to_dense().double()
    T2 = 2 * G @ G - torch.eye(30, dtype=torch.float64)
    d = graph.second_order('cpu')
    for side, want in (('bwd2', T2), ('fwd2', T2.t())):
        rp, ci, v = d[f'{side}_rowptr'], d[f'{side}_colidx'], d[f'{side}_val']
        M = torch.zeros(30, 30, dtype=torch.float64)
        M[torch.repeat_interleave(torch.arange(30), (rp[1:] - rp[:-1]).long()), ci.long()] = v.double()
        assert rp.dtype == torch.int32 and ci.dtype == torch.int32 and v.dtype == torch.float32 and float((M - want).abs().max()) < 1e-6
        assert all(bool((ci[rp[i]:rp[i + 1]][1:] > ci[rp[i]:rp[i + 1]][:-1]).all()) for i in range(30))      # columns ascending within a row
    assert graph.second_order('cpu') is d                           # cached per device


@pytest.mark.gpu
@pytest.mark.parametrize('B,N,C,cin,splits', [(32, 100, 5, 1, 1), (32, 100, 5, 16, 1), (32, 100, 5, 16, 8), (3, 100, 5, 1, 4), (2, 37, 8, 3, 2), (1, 10, 16, 4, 1),
                                              (2, 7, 1, 2, 3), (2, 33, 7, 16, 1), (2, 200, 8, 16, 8), (2, 5000, 5, 1, 8), (1, 4500, 7, 3, 1), (1, 13000, 5, 16, 16)])
@pytest.mark.parametrize('bias,acc', [(True, False), (False, True)])
def test_small_cell_kernels_at_order_3(B, N, C, cin, splits, bias, acc):
    """stc_cell_small_fwd/bwd_f32 at Ks = Kc = 3 (ABI v32) against the twin: one launch per cell step and the split forms, narrow and wide inputs,
    ragged tiles, one category, the top of the row range; accumulate flags and absent biases."""
    from stc_hip._lib import HipKernels
    hip = HipKernels()
    assert hip.cell_small_supported(3, 3, C, cin, 16, N)
    graph = _graph(N, seed=N + cin)
    t = _inputs(B, N, C, cin, seed=3 * N + C + cin, bias=bias, K=3)
    buf, P = _buffers(B, N, C, cin, torch.float32, hip, K=3)
    want = _run(EM, csr_operand(graph, torch.device('cpu')), t, buf, P, lambda v: v.clone(), acc_x=acc, acc_h=acc)
    got = _run(hip, csr_operand(graph, torch.device('cuda')), t, buf, P, lambda v: v.cuda(), acc_x=acc, acc_h=acc, splits=splits)
    for name in ('U', 'R', 'RH', 'Zg', 'Zc', 'Zg2', 'Zc2', 'Cand', 'Hnew', 'dX', 'dH'):
        assert rel_err(got[name], want[name]) < TOL, name
    extra = 0.125 * (got['dP'].shape[0] - want['dP'].shape[0])      # (both sides start from 0.125 in every row: the split form has `splits` times the rows)
    for a, b, name in zip(_split_params(got['dP'], cin, K=3), _split_params(want['dP'], cin, K=3), ('dWg', 'dbg', 'dWc', 'dbc')):
        assert rel_err(a - extra, b) < TOL, name


@pytest.mark.gpu
@pytest.mark.parametrize('B,N,C,cin,dense', [
    (32, 100, 5, 1, False),      # the SF shape, layer 0
    (32, 100, 5, 16, False),     # the SF shape, layers above
    (3, 100, 5, 16, True),       # a dense graph as a CSR the kernels know nothing about: row gathers
    (32, 100, 5, 16, 'operand'), # the reference's learned dense Gs (graph.dense_operand): aggregation as matrix products on the staged planes
    (4, 100, 5, 1, 'operand'),
    (2, 37, 8, 3, 'operand'),    # N not a multiple of 16 nor of 4
    (2, 37, 8, 16, False),       # two whole nodes per row tile, ragged last tile
    (2, 200, 8, 3, False),       # BASELINE configuration 3's nominal size (N ~ 200, C ~ 8)
    (1, 10, 16, 4, False),       # one node per tile
    (2, 7, 1, 2, False),         # one category: no mix partner rows
    (2, 33, 7, 16, False),       # 2 nodes of 7 rows per tile (14 of 16 rows used)
    (2, 5000, 5, 1, False),      # rows beyond 4096 C: the row -> (node, category) division must be exact up to 65535 rows (round 3's wrapped)
    (1, 5000, 5, 16, False),
    (1, 13000, 5, 16, False),    # 65 000 rows: the top of the accepted range
    (1, 4500, 7, 3, False),
])
@pytest.mark.parametrize('bias,acc', [(True, False), (False, True)])
def test_small_cell_kernels(B, N, C, cin, dense, bias, acc):
    from stc_hip._lib import HipKernels
    hip = HipKernels()
    assert hip.cell_small_supported(2, 2, C, cin, 16, N)
    graph = _graph(N, seed=N + cin, dense=bool(dense))
    operand = (lambda dev: dense_operand(graph.to_dense().to(dev))) if dense == 'operand' else (lambda dev: csr_operand(graph, torch.device(dev)))
    t = _inputs(B, N, C, cin, seed=3 * N + C + cin, bias=bias)
    buf, P = _buffers(B, N, C, cin, torch.float32, hip)
    want = _run(EM, operand('cpu'), t, buf, P, lambda v: v.clone(), acc_x=acc, acc_h=acc, want_dumps=acc)
    got = _run(hip, operand('cuda'), t, buf, P, lambda v: v.cuda(), acc_x=acc, acc_h=acc, want_dumps=acc)
    for name in ('U', 'R', 'RH', 'Zg', 'Zc', 'Cand', 'Hnew') + (('Z0', 'dZ1c', 'dZ1g', 'dYg') if acc else ()):   # (dumps: what learned graphs keep)
        assert rel_err(got[name], want[name]) < TOL, name
    for name in ('dX', 'dH'):
        assert rel_err(got[name], want[name]) < TOL, name
    for a, b, name in zip(_split_params(got['dP'], cin), _split_params(want['dP'], cin), ('dWg', 'dbg', 'dWc', 'dbc', 'rest')):
        assert rel_err(a, b) < TOL, name
    # inputs that need no gradient: NULL outputs
    got = _run(hip, operand('cuda'), t, buf, P, lambda v: v.cuda(), want_x=False, want_h=False)
    assert got['dX'] is None and got['dH'] is None
    for a, b, name in zip(_split_params(got['dP'], cin), _split_params(want['dP'], cin), ('dWg', 'dbg', 'dWc', 'dbc', 'rest')):
        assert rel_err(a, b) < TOL, name


@pytest.mark.gpu
@pytest.mark.parametrize('B,N,C,cin,splits,dense', [(2, 100, 5, 16, 8, False), (3, 100, 5, 1, 4, False), (1, 37, 8, 16, 2, False), (2, 200, 8, 3, 8, False),
                                                     (1, 7, 1, 2, 3, False), (2, 100, 5, 16, 8, True), (2, 37, 8, 3, 4, True),
                                                     (2, 130, 3, 1, 8, True), (1, 16, 16, 16, 2, True), (2, 200, 8, 16, 8, True), (1, 65, 5, 4, 16, True),
                                                     (2, 5000, 5, 16, 8, False), (1, 4500, 7, 3, 4, False), (1, 13000, 5, 1, 16, False)])
def test_small_cell_kernels_split_over_workgroups(B, N, C, cin, splits, dense):
    """The same cell step as four launches per direction (one per phase) over ``splits`` workgroups per sample -- what the executor uses when
    the batch is too small to fill the chip with one workgroup per sample: same buffers, same results (parameter-gradient partials in
    splits x as many rows), dumps included."""
    from stc_hip._lib import HipKernels
    hip = HipKernels()
    graph = _graph(N, seed=N + cin, dense=dense)
    # (dense: the learned graph's operand -- the split launches then aggregate as matrix products over the node tiles that cover a workgroup's own
    # rows: shapes where those tiles straddle workgroups, N below / at / not a multiple of the 16-node tile, narrow and wide inputs)
    operand = (lambda dev: dense_operand(graph.to_dense().to(dev))) if dense else (lambda dev: csr_operand(graph, torch.device(dev)))
    t = _inputs(B, N, C, cin, seed=5 * N + C + cin)
    buf, P = _buffers(B, N, C, cin, torch.float32, hip)
    want = _run(EM, operand('cpu'), t, buf, P, lambda v: v.clone(), acc_x=True, acc_h=True, want_dumps=True)
    got = _run(hip, operand('cuda'), t, buf, P, lambda v: v.cuda(), acc_x=True, acc_h=True, want_dumps=True, splits=splits)
    for name in ('U', 'R', 'RH', 'Zg', 'Zc', 'Cand', 'Hnew', 'Z0', 'dZ1c', 'dZ1g', 'dYg', 'dX', 'dH'):
        assert rel_err(got[name], want[name]) < TOL, name
    for a, b, name in zip(_split_params(got['dP'], cin), _split_params(want['dP'], cin), ('dWg', 'dbg', 'dWc', 'dbc')):
        # (both sides start from 0.125 in every row: the split form has `splits` times the rows)
        extra = 0.125 * (got['dP'].shape[0] - want['dP'].shape[0])
        assert rel_err(a - extra, b) < TOL, name
    assert hip.cell_small_splits(32, 500) == 8 and hip.cell_small_splits(256, 500) == 1 and hip.cell_small_splits(4, 100) == 2


@pytest.mark.gpu
def test_small_cell_refuses_other_shapes():
    from stc_hip._lib import HipKernels, StcError
    hip = HipKernels()
    assert not hip.cell_small_supported(3, 2, 5, 16, 16, 100) and not hip.cell_small_supported(2, 2, 17, 16, 16, 100)
    assert not hip.cell_small_supported(2, 2, 5, 8, 16, 100) and not hip.cell_small_supported(2, 2, 32, 16, 16, 50176)
    graph = _graph(12, seed=1)
    t = _inputs(2, 12, 5, 8, seed=2)
    buf, P = _buffers(2, 12, 5, 8, torch.float32, hip)
    with pytest.raises(StcError, match='outside the small-graph cell kernels'):
        _run(hip, csr_operand(graph, torch.device('cuda')), t, buf, P, lambda v: v.cuda())


@pytest.mark.gpu
@pytest.mark.parametrize('cells,B,N,C,wa,wb,sel', [(9, 32, 100, 5, 32, 32, (0, 1, 9)), (6, 3, 100, 5, 32, 16, (1, 2, 3)), (4, 2, 37, 8, 20, 32, (0, 3, 2)),
                                                  (3, 2, 12, 3, 20, 16, (2, 1, 1)), (2, 1, 130, 1, 32, 32, (0, 1, 2))])
def test_graph_gradient_products(cells, B, N, C, wa, wb, sel):
    """stc_graph_grad_f32 / stc_mix_grad_f32 (the dGs / dGc products of learned graphs: sums over selected cells and samples, float64
    accumulation) against float64 einsums; strided cell selections, N and widths off the tile boundaries."""
    from stc_hip._lib import HipKernels
    hip = HipKernels()
    g = torch.Generator().manual_seed(cells * N + wa)
    A, Bm, Bn = (torch.randn(cells, B, N * C, w_, generator=g) for w_ in (wa, wa, wb))
    want = EM.graph_grad(A, Bm, *sel, N)
    got = hip.graph_grad(A.cuda(), Bm.cuda(), *sel, N).cpu()
    assert got.dtype == torch.float64 and rel_err(got, want) < 1e-6
    want = EM.mix_grad(A, Bn, *sel, N)
    got = hip.mix_grad(A.cuda(), Bn.cuda(), *sel, N).cpu()
    assert got.shape == (C * wa, C * wb) and rel_err(got, want) < 1e-6
