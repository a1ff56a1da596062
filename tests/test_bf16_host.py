"""CPU (-m "not gpu"): the bf16-storage path's host logic and CPU twins.

* the bf16 node-kernel twins (same rounding points as the hardware path) against the fp32 twins;
* ``STCGNN(storage_dtype=bfloat16)`` through the cell-graph executor on the emulated kernel set -- the schedule, the planar
  piece bookkeeping and the dtype plumbing are the same code the GPU runs -- against the fp32 model.
"""
import pytest
import torch

import STC_GNN as M
from oracle.kernel_emul import EmulatedKernels
from stc_hip import CsrGraph, ops
from tests.conftest import rel_err

EM = EmulatedKernels()


@pytest.mark.parametrize('K,C,L,Lw,Ho', [(2, 32, 32, 32, 32), (3, 32, 16, 9, 16), (1, 64, 32, 17, 32)])
def test_bf16_node_twins_track_the_fp32_twins(K, C, L, Lw, Ho):
    g = torch.Generator().manual_seed(K + C + L)
    n = 11
    Zs = [torch.randn(n, C, L, generator=g).bfloat16() for _ in range(K)]
    Tc = torch.softmax(torch.randn(K, C, C, generator=g), -1)
    Tc[0] = torch.eye(C)
    W, b = torch.randn(K * K * Lw, Ho, generator=g) * 0.2, torch.randn(Ho, generator=g)
    assert EM.node_bf16_supported(K, K, C, L, Ho)
    Y = torch.empty(n, C, Ho, dtype=torch.bfloat16)
    EM.bdg_node_fwd_bf16(Zs, Tc, W, b, Y)
    E = torch.empty(n, C, Ho)
    EM.bdg_node_fwd([z.float() for z in Zs], Tc, W, b, E)
    assert rel_err(Y.float(), E) < 2e-2
    dY = torch.randn(n, C, Ho, generator=g).bfloat16()
    dZ = [torch.empty(n, C, L, dtype=torch.bfloat16) for _ in range(K)]
    dW, db = torch.empty_like(W), torch.empty(Ho)
    EM.bdg_node_bwd_bf16(Zs, Tc, W, dY, dZ, dW, db)
    eZ = [torch.empty(n, C, L) for _ in range(K)]
    eW, eb = torch.empty_like(W), torch.empty(Ho)
    EM.bdg_node_bwd([z.float() for z in Zs], Tc, W, dY.float(), eZ, eW, eb, None)
    for a, e in zip(dZ, eZ):
        assert rel_err(a.float(), e) < 2e-2
        assert float(a[..., Lw:].float().abs().max() if Lw < L else 0.0) == 0.0
    assert rel_err(dW, eW) < 2e-2 and rel_err(db, eb) < 1e-5


def test_spmm_bf16_twin_rounds_once():
    graph = CsrGraph.queen_grid(4, 5, normalize=True)
    h = graph._host
    csr = tuple(torch.from_numpy(h[k]) for k in ('fwd_rowptr', 'fwd_colidx', 'fwd_val'))
    X = torch.randn(2, 20, 64).bfloat16()
    Y, E = torch.empty(2, 20, 64, dtype=torch.bfloat16), torch.empty(2, 20, 64)
    EM.csr_spmm_bf16(*csr, 20, 20, X, None, Y, 1.0, 0.0)
    EM.csr_spmm(*csr, 20, 20, X.float(), None, E, 1.0, 0.0)
    assert torch.equal(Y, E.bfloat16())


@pytest.mark.parametrize('layers,T,horizon,cin', [(2, 3, 2, 1), (1, 2, 1, 4), (3, 2, 2, 1)])
def test_bf16_storage_model_on_the_emulated_kernels(monkeypatch, layers, T, horizon, cin):
    monkeypatch.setattr(ops, '_kernels', EM)
    C, Hh, Ww, h, K, B = 5, 4, 5, 16, 2, 2
    torch.manual_seed(layers + T)
    graph = CsrGraph.queen_grid(Hh, Ww, normalize=True)
    N = Hh * Ww
    kw = dict(num_nodes=N, num_categories=C, Ks=K, Kc=K, input_dim=cin, hidden_dim=h, num_layers=layers, out_horizon=horizon, graph_mode='csr-fixed')
    m32, m16 = M.STCGNN(**kw), M.STCGNN(**kw, storage_dtype=torch.bfloat16)
    m16.load_state_dict(m32.state_dict())
    Gc = torch.softmax(torch.randn(C, C), -1)
    X = (torch.rand(B, T, N, C, cin) < 0.3).float()
    Rw = torch.randn(B, horizon, N, C)
    # the emulated set has no shape limits, the real bf16 kernels do (C in {32, 64}): ask the emulated front
    monkeypatch.setattr(ops, 'cell_graph_supported', lambda op, Tc, Ks, C_, h_, widths, dtype=torch.float32: True)
    out = {}
    for name, m in (('f32', m32), ('bf16', m16)):
        pair = M._graphs(graph, Gc, K, K)
        stacked = m._run_cell_graph(pair, X.to(m.storage_dtype))
        assert stacked is not None and stacked.dtype == m.storage_dtype
        y = torch.sigmoid(stacked.float().sum(-1)).transpose(0, 1)
        (y * Rw).sum().backward()
        out[name] = (y.detach(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    (y32, g32), (y16, g16) = out['f32'], out['bf16']
    assert float((y16 - y32).abs().max()) < 2e-2
    assert set(g16) == set(g32) and all(g.dtype == torch.float32 for g in g16.values())
    for n in g32:
        a, b = g16[n].flatten().double(), g32[n].flatten().double()
        assert float((a @ b) / (a.norm() * b.norm() + 1e-30)) > 0.995, n


def test_bf16_storage_constructor_checks():
    with pytest.raises(ValueError):
        M.STCGNN(30, 32, 2, 2, 1, 16, 2, 2, storage_dtype=torch.bfloat16)                       # learned graphs: fp32 only
    with pytest.raises(ValueError):
        M.STCGNN(30, 32, 2, 2, 1, 16, 2, 2, graph_mode='csr-fixed', storage_dtype=torch.float16)
    m = M.STCGNN(30, 32, 2, 2, 1, 16, 2, 2, graph_mode='csr-fixed', storage_dtype=torch.bfloat16)
    assert all(p.dtype == torch.float32 for p in m.parameters())                                # master weights stay fp32


@pytest.mark.parametrize('K', [1, 2, 3])
def test_bdg_dif_module_on_bf16_features_emulated(monkeypatch, K):
    """BDG_Dif fed bfloat16 features through the host path (SpMM hops, feature-side recurrence and its backward, node kernels) on
    the emulated kernel set, against the same module in fp32."""
    monkeypatch.setattr(ops, '_kernels', EM)
    torch.manual_seed(K)
    C, L, Ho, N = 32, 32, 16, 20
    graph = CsrGraph.queen_grid(4, 5, normalize=True)
    conv = M.BDG_Dif(K, K, L, Ho)
    Gc = torch.softmax(torch.randn(C, C), -1)
    X = torch.randn(2, N, C, L).bfloat16()
    Rw = torch.randn(2, N, C, Ho)
    out = {}
    for name, x in (('f32', X.float().requires_grad_()), ('bf16', X.clone().requires_grad_())):
        conv.zero_grad(set_to_none=True)
        y = conv(x, graph, Gc)
        assert y.dtype == x.dtype
        (y.float() * Rw).sum().backward()
        out[name] = (y.detach().float(), x.grad.float(), conv.W.grad.clone(), conv.b.grad.clone())
    for a, b in zip(out['bf16'], out['f32']):
        assert rel_err(a, b) < 3e-2
    with pytest.raises(ValueError):
        conv(torch.zeros(1, N, 16, L, dtype=torch.bfloat16), graph, torch.eye(16))          # C = 16 is off the bf16 kernels
