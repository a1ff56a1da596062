"""Checks at BASELINE.json's full size (N = 50 176 = 224 x 224 queen grid) on the MI355X.

* one ``STC_Cell`` (reference ``STC_GNN.py:65-79``) at C = 32 and C = 64, hidden 16, against the CPU oracle's sparse
  restatement (``oracle.stc_cell(conv=bdg_dif_sparse)``, itself pinned to the dense reference at N = 1 024 / 10 000 by
  g7 / g8 / g8b) on 256 sampled rows: new state, dXt, dHt, and the full parameter gradients;
* the whole encoder-decoder at a batch whose stacked decoder states exceed 2^31 bytes, through size-independent
  properties: every sample of a batch equals the same sample run alone (samples are independent on this path, SURVEY
  8(e1)), and the batch's parameter gradients are the sum of the single-sample gradients (linearity of the backward).
  Together with the per-cell oracle check this guards the 32-bit index arithmetic at the sizes the bench runs.
* BASELINE configuration 5 at its own size: two chained planar cells at N = 50 176, C = 64 with bf16 state storage against the same
  cells on the fp32 HIP path (the parity-checked one), under the bf16 contract of DESIGN.md (the reference has no bf16 behaviour).
"""
import pytest
import torch

import STC_GNN as M
from oracle import stc_oracle as O
from stc_hip import CsrGraph, ops
from tests.conftest import rel_err
from tests.golden.make_golden import sample_rows

GRID = 224


def _sparse_T(graph):
    h = graph._host
    return torch.sparse_csr_tensor(torch.from_numpy(h['fwd_rowptr']).long(), torch.from_numpy(h['fwd_colidx']).long(),
                                   torch.from_numpy(h['fwd_val']), size=(graph.n, graph.n))


@pytest.mark.gpu
@pytest.mark.parametrize('C,cin', [(32, 16), (32, 1), (64, 16)])
def test_full_size_cell_against_oracle_rows(monkeypatch, C, cin):
    monkeypatch.setattr(ops, '_kernels', None)
    dev = torch.device('cuda')
    N, h, K = GRID * GRID, 16, 2
    graph = CsrGraph.queen_grid(GRID, GRID, normalize=True)
    assert graph.nnz == 398724
    g = torch.Generator().manual_seed(50 + C + cin)
    Gc = torch.softmax(torch.randn(C, C, generator=g), -1)
    Xt = (torch.rand(1, N, C, cin, generator=g) < 0.1635).float() if cin == 1 else torch.tanh(torch.randn(1, N, C, cin, generator=g))
    Ht = torch.tanh(torch.randn(1, N, C, h, generator=g))
    R = torch.randn(1, N, C, h, generator=g)
    torch.manual_seed(7)
    cell = M.STC_Cell(N, C, K, K, cin, h)
    with torch.no_grad():
        cell.gates.b.normal_(0, 0.3)
        cell.candi.b.normal_(0, 0.3)
    names = ('gates.W', 'gates.b', 'candi.W', 'candi.b')
    sd = cell.state_dict()

    torch.set_num_threads(min(32, torch.get_num_threads() or 32))
    p = [sd[n].clone().requires_grad_() for n in names]
    Xo, Ho = Xt.clone().requires_grad_(), Ht.clone().requires_grad_()
    want = O.stc_cell(_sparse_T(graph), Gc, Xo, Ho, *p, K, K, conv=O.bdg_dif_sparse)
    (want * R).sum().backward()

    cell = cell.to(dev)
    Xd, Hd = Xt.to(dev).requires_grad_(), Ht.to(dev).requires_grad_()
    got = cell(graph, Gc.to(dev), Xd, Hd)
    (got * R.to(dev)).sum().backward()
    torch.cuda.synchronize()
    rows = sample_rows(N, 256, seed=5)
    assert rel_err(got[:, rows], want[:, rows]) < 1e-5
    assert rel_err(Hd.grad[:, rows], Ho.grad[:, rows]) < 1e-5
    assert rel_err(Xd.grad[:, rows], Xo.grad[:, rows]) < 1e-5
    params = dict(cell.named_parameters())
    for n, t in zip(names, p):
        assert rel_err(params[n].grad, t.grad) < 2e-5, n             # sums over 1.6 M rows: the long-reduction bound


@pytest.mark.gpu
def test_full_size_batch_equals_single_samples_beyond_2gb(monkeypatch):
    monkeypatch.setattr(ops, '_kernels', None)
    dev = torch.device('cuda')
    N, C, h, K, B, T, horizon = GRID * GRID, 32, 16, 2, 4, 1, 6
    assert horizon * B * N * C * h * 4 > 2 ** 31                      # the stacked decoder states of this batch: 2.47 GB
    graph = CsrGraph.queen_grid(GRID, GRID, normalize=True, device=dev)
    torch.manual_seed(42)
    model = M.STCGNN(N, C, K, K, 1, h, 2, horizon, graph_mode='csr-fixed').to(dev)
    g = torch.Generator().manual_seed(3)
    Gc = torch.softmax(torch.randn(C, C, generator=g), -1).to(dev)
    X = (torch.rand(B, T, N, C, generator=g) < 0.1635).float().to(dev)
    R = torch.randn(B, horizon, N, C, generator=g).to(dev)
    calls = []
    real = ops.stc_cell_graph
    monkeypatch.setattr(ops, 'stc_cell_graph', lambda *a, **k: (calls.append(1), real(*a, **k))[1])

    def run(xs, rs):
        model.zero_grad(set_to_none=True)
        y = model(X_seq=xs, As=graph, Ac=Gc)
        (y * rs).sum().backward()
        return y.detach(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}

    y_all, g_all = run(X, R)
    assert calls, 'the cell-graph path (the one the bench runs) was not taken'
    assert torch.isfinite(y_all).all()
    g_sum = None
    for b in range(B):
        y_b, g_b = run(X[b:b + 1], R[b:b + 1])
        assert rel_err(y_all[b:b + 1], y_b) < 1e-6, b                  # same kernels on the same rows: equal up to nothing
        g_sum = g_b if g_sum is None else {n: g_sum[n] + g_b[n] for n in g_b}
    for n in g_all:
        assert rel_err(g_all[n], g_sum[n]) < 1e-5, n


@pytest.mark.gpu
def test_full_size_model_against_oracle_end_to_end(monkeypatch):
    """The whole path at N = 50 176 -- encoder (1 observed step) + decoder (1 predicted step), 2 layers = 4 cells, head, ComboLoss,
    backward -- against the CPU oracle's sparse restatement: every prediction and every parameter gradient."""
    monkeypatch.setattr(ops, '_kernels', None)
    dev = torch.device('cuda')
    N, C, h, K, B = GRID * GRID, 32, 16, 2, 1
    graph = CsrGraph.queen_grid(GRID, GRID, normalize=True)
    torch.manual_seed(42)
    model = M.STCGNN(N, C, K, K, 1, h, 2, 1, graph_mode='csr-fixed')
    with torch.no_grad():
        for cell in list(model.encoder.cell_list) + list(model.decoder.cell_list):
            cell.gates.b.normal_(0, 0.3)
            cell.candi.b.normal_(0, 0.3)
    g = torch.Generator().manual_seed(8)
    Gc = torch.softmax(torch.randn(C, C, generator=g), -1)
    X = (torch.rand(B, 1, N, C, generator=g) < 0.1635).float()
    Y = (torch.rand(B, 1, N, C, generator=g) < 0.1635).float()
    torch.set_num_threads(min(32, torch.get_num_threads() or 32))
    # The oracle runs in float64 here (it is dtype-generic): at 1.6 M rows the fp32 CPU sums of the head's weight gradients are
    # themselves off by 1e-5 .. 1e-3 (tools/probes/head_grad_precision.py: torch CPU fp32 1.7e-5 vs this build's head kernel 4.6e-8
    # against float64), so a float32 oracle would be the less exact side of the comparison.
    sd = {k: v.double().clone().requires_grad_() for k, v in model.state_dict().items()}
    want = O.encdec_forward(X.double(), _sparse_T(graph).double(), Gc.double(), sd, K, K, h, 2, 1, conv=O.bdg_dif_sparse)
    loss_w = O.combo_loss(want, Y.double())
    loss_w.backward()
    model = model.to(dev)
    calls = []
    real = ops.stc_cell_graph
    monkeypatch.setattr(ops, 'stc_cell_graph', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    got = model(X_seq=X.to(dev), As=graph, Ac=Gc.to(dev))
    assert calls, 'the cell-graph path (the one the bench runs) was not taken'
    loss = O.combo_loss(got, Y.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    assert rel_err(got, want) < 1e-5 and abs(float(loss.detach()) - float(loss_w.detach())) < 1e-5
    for name, p in model.named_parameters():
        assert rel_err(p.grad, sd[name].grad) < 2e-5, name


@pytest.mark.gpu
def test_full_size_bf16_storage_cells_track_the_fp32_path(monkeypatch):
    """BASELINE configuration 5 (N = 50 176, C = 64, bf16 state storage): cell 0 reads external planes, cell 1 reads cell 0's state as
    X and as H (so the state's gradient is assembled from its consumers' pieces) -- the bf16-storage kernels against the fp32 HIP path
    on the same bf16-exact inputs and parameters: new states on 256 sampled rows within 2e-2 (states are in (-1, 1)), every parameter
    gradient with cosine > 0.995 and norm within 5 % (the contract of tests/test_bf16_kernels.py, at the size the configuration names)."""
    monkeypatch.setattr(ops, '_kernels', None)
    from stc_hip.graph import csr_operand
    dev = torch.device('cuda')
    N, C, h, K = GRID * GRID, 64, 16, 2
    graph = CsrGraph.queen_grid(GRID, GRID, normalize=True, device=dev)
    op = csr_operand(graph, dev)
    g = torch.Generator().manual_seed(21)
    Gc = torch.softmax(torch.randn(C, C, generator=g), -1).to(dev)
    Tc = ops.cheby_dense(Gc, K)
    bfx = lambda t: t.to(torch.bfloat16).float()                     # bf16-exact values in both runs
    X = bfx(torch.rand(1, N, C, h, generator=g) - 0.5).to(dev)
    H = bfx(torch.tanh(torch.randn(1, N, C, h, generator=g))).to(dev)
    R = torch.randn(1, N, C, h, generator=g).to(dev)
    L = 2 * h
    P = [torch.randn(K * K * L, 2 * h, generator=g) * (2.0 / (K * K * L + 2 * h)) ** 0.5, torch.randn(2 * h, generator=g) * 0.1,
         torch.randn(K * K * L, h, generator=g) * (2.0 / (K * K * L + h)) ** 0.5, torch.randn(h, generator=g) * 0.1]
    sched = [(0, ('ext', 0), ('ext', 1)), (0, ('cell', 0), ('cell', 0))]
    rows = sample_rows(N, 256, seed=7).to(dev)
    got = {}
    for name, dt in (('f32', torch.float32), ('bf16', torch.bfloat16)):
        p = [t.clone().to(dev).requires_grad_() for t in P]
        assert ops.cell_graph_supported(op, Tc, K, C, h, [h], dtype=dt)
        out = ops.stc_cell_graph(op, Tc, K, sched, [0, 1], [X.to(dt), H.to(dt)], [tuple(p)])
        assert out.dtype == dt
        (out[1].float() * R).sum().backward()
        torch.cuda.synchronize()
        got[name] = (out.detach().float()[:, :, rows], [t.grad.detach().clone() for t in p])
        del out
    (s32, g32), (s16, g16) = got['f32'], got['bf16']
    assert torch.isfinite(s16).all() and float((s16 - s32).abs().max()) < 2e-2
    for a_, b_, name in zip(g16, g32, ('gates.W', 'gates.b', 'candi.W', 'candi.b')):
        a, b = a_.flatten().double(), b_.flatten().double()
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
        assert cos > 0.995, f'd{name}: cosine {cos}'
        assert abs(float(a.norm() / (b.norm() + 1e-300)) - 1) < 5e-2, f'd{name}: norm ratio {float(a.norm() / b.norm())}'
