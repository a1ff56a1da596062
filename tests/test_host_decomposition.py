"""CPU check of the HOST LOGIC: the launch sequences in ``stc_hip.ops`` and the drop-in modules
of ``STC_GNN.py`` reproduce the reference's golden vectors when the kernels are stood in for by
``oracle/kernel_emul.py`` (test infrastructure, injected here; the product has no CPU path).

What this proves before any GPU time is spent: the decomposition the HIP path uses --
feature-side Chebyshev recurrence over CSR operands, project-then-mix node kernel, hand-derived
backward incl. dGs/dGc, fused gate math, module wiring and state_dict keys -- is the reference's
math.  The ``-m gpu`` tests then only have to show each HIP kernel equals its emulated twin.
"""
import pytest
import torch

import STC_GNN as M
from oracle import stc_oracle as O
from oracle.kernel_emul import EmulatedKernels
from stc_hip import CsrGraph, ops
from tests.conftest import load_golden, rel_err, sub_dict
from tests.golden.make_golden import synth_inputs

TOL = 2e-6


@pytest.fixture(autouse=True)
def emulated_kernels(monkeypatch):
    monkeypatch.setattr(ops, '_kernels', EmulatedKernels())
    yield


def _leaf(t):
    return t.clone().requires_grad_()


@pytest.mark.parametrize('K', [1, 2, 3])
def test_bdg_dif_module_matches_reference_golden(K):
    g = load_golden(f'g1_bdg_k{K}')
    B, N, C, L = g['X'].shape
    layer = M.BDG_Dif(K, K, L, g['W'].shape[1])
    layer.load_state_dict({'W': g['W'], 'b': g['b']})
    X, Gs, Gc = _leaf(g['X']), _leaf(g['Gs']), _leaf(g['Gc'])
    Y = layer(X, Gs, Gc)
    assert rel_err(Y, g['Y']) < TOL
    (Y * g['R']).sum().backward()
    assert rel_err(X.grad, g['dX']) < TOL
    assert rel_err(layer.W.grad, g['dW']) < TOL
    assert rel_err(layer.b.grad, g['db']) < TOL
    if K > 1:
        assert rel_err(Gs.grad, g['dGs']) < TOL
        assert rel_err(Gc.grad, g['dGc']) < TOL
    else:   # K=1: only T_0 = I is used; the drop-in reports zeros where the reference reports None
        assert Gs.grad is None or float(Gs.grad.abs().max()) == 0.0
        assert Gc.grad is None or float(Gc.grad.abs().max()) == 0.0


def test_bdg_dif_no_bias_and_activation():
    g = load_golden('g1_bdg_nobias')
    layer = M.BDG_Dif(2, 2, g['X'].shape[-1], g['W'].shape[1], use_bias=False)
    assert [k for k, _ in layer.state_dict().items()] == ['W']
    layer.load_state_dict({'W': g['W']})
    assert rel_err(layer(g['X'], g['Gs'], g['Gc']), g['Y']) < TOL
    act = M.BDG_Dif(2, 2, g['X'].shape[-1], g['W'].shape[1], use_bias=False, activation=torch.nn.ReLU)
    act.load_state_dict({'W': g['W']})
    assert rel_err(act(g['X'], g['Gs'], g['Gc']), torch.relu(g['Y'])) < TOL


@pytest.mark.parametrize('cin,K', [(1, 2), (1, 3), (4, 2), (4, 3)])
def test_stc_cell_module(cin, K):
    g = load_golden(f'g2_cell_in{cin}_k{K}')
    B, N, C, h = g['Ht'].shape
    cell = M.STC_Cell(N, C, K, K, cin, h)
    cell.load_state_dict(sub_dict(g, 'sd/'))
    Xt, Ht, Gs, Gc = (_leaf(g[k]) for k in ('Xt', 'Ht', 'Gs', 'Gc'))
    out = cell(Gs=Gs, Gc=Gc, Xt=Xt, Ht_1=Ht)
    assert rel_err(out, g['Hout']) < TOL
    (out * g['R']).sum().backward()
    for name, leaf in (('dXt', Xt), ('dHt', Ht), ('dGs', Gs), ('dGc', Gc)):
        assert rel_err(leaf.grad, g[name]) < 5e-6, name
    grads = dict(cell.named_parameters())
    for k, v in sub_dict(g, 'grad/').items():
        assert rel_err(grads[k].grad, v) < 5e-6, k


def test_encoder_decoder_modules():
    g = load_golden('g3_encdec')
    K, h, layers = int(g['K']), int(g['h']), int(g['layers'])
    B, T, N, C, _ = g['X_seq'].shape
    enc = M.STC_Encoder(N, C, K, K, 1, h, layers, return_all_layers=True)
    enc.load_state_dict(sub_dict(g, 'enc_sd/'))
    X_seq, Gs, Gc = _leaf(g['X_seq']), _leaf(g['Gs']), _leaf(g['Gc'])
    seqs, lasts = enc(Gs=Gs, Gc=Gc, X_seq=X_seq, H0_l=None)
    assert rel_err(seqs[0], g['seq0']) < TOL and rel_err(seqs[1], g['seq1']) < TOL
    assert rel_err(lasts[1], g['last1']) < TOL
    ((seqs[0] * g['R0']).sum() + (seqs[1] * g['R1']).sum() + (lasts[0] * g['RL']).sum()).backward()
    assert rel_err(X_seq.grad, g['dX_seq']) < 5e-6
    assert rel_err(Gs.grad, g['enc_dGs']) < 5e-6 and rel_err(Gc.grad, g['enc_dGc']) < 5e-6
    grads = dict(enc.named_parameters())
    for k, v in sub_dict(g, 'enc_grad/').items():
        assert rel_err(grads[k].grad, v) < 5e-6, k
    enc.return_all_layers = False
    s2, l2 = enc(g['Gs'], g['Gc'], g['X_seq'])
    assert len(s2) == 1 and len(l2) == 1 and rel_err(s2[0], g['seq_last_only']) < TOL

    dec = M.STC_Decoder(N, C, K, K, h, h, layers, out_horizon=2)
    dec.load_state_dict(sub_dict(g, 'dec_sd/'))
    Gs, Gc, Xd = _leaf(g['Gs']), _leaf(g['Gc']), _leaf(g['Xd'])
    H0 = [_leaf(g['H00']), _leaf(g['H01'])]
    top, states = dec(Gs=Gs, Gc=Gc, Xt=Xd, H0_l=H0)
    assert rel_err(top, g['dec_top']) < TOL and rel_err(states[0], g['dec_s0']) < TOL
    ((top * g['Rd']).sum() + (states[0] * g['Rs']).sum()).backward()
    assert rel_err(Xd.grad, g['dXd']) < 5e-6
    assert rel_err(H0[0].grad, g['dH00']) < 5e-6 and rel_err(H0[1].grad, g['dH01']) < 5e-6
    assert rel_err(Gs.grad, g['dec_dGs']) < 5e-6 and rel_err(Gc.grad, g['dec_dGc']) < 5e-6


def test_full_model_state_dict_keys_loss_and_grads():
    g = load_golden('g4_stcgnn_small')
    model = M.STCGNN(num_nodes=int(g['N']), num_categories=int(g['C']), Ks=int(g['K']), Kc=int(g['K']), input_dim=1,
                     hidden_dim=int(g['h']), num_layers=int(g['layers']), out_horizon=int(g['horizon']))
    sd = sub_dict(g, 'sd/')
    assert list(model.state_dict().keys()) == list(sd.keys())        # same keys, same order as the reference
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd)
    Gs, Gc = model.mix_graph_pair(g['X'], g['As'], g['Ac'])
    assert rel_err(Gs, g['Gs']) < TOL and rel_err(Gc, g['Gc']) < TOL
    yhat = model(X_seq=g['X'], As=g['As'], Ac=g['Ac'])
    assert yhat.shape == g['yhat'].shape and rel_err(yhat, g['yhat']) < TOL
    loss = O.combo_loss(yhat, g['Y'])
    assert abs(float(loss.detach()) - float(g['loss'])) < 1e-6
    loss.backward()
    grads = dict(model.named_parameters())
    for k, v in sub_dict(g, 'grad/').items():
        assert rel_err(grads[k].grad, v) < 1e-5, k


def test_adam_trajectory_through_drop_in():
    """Five steps of the reference's train step (Model_Trainer.py:71-87) on the drop-in module."""
    g = load_golden('g4_stcgnn_small')
    model = M.STCGNN(int(g['N']), int(g['C']), int(g['K']), int(g['K']), 1, int(g['h']), int(g['layers']), int(g['horizon']))
    model.load_state_dict(sub_dict(g, 'sd/'))
    opt = torch.optim.Adam(model.parameters(), lr=2e-3, weight_decay=1e-4)
    losses = []
    for _ in range(5):
        loss = O.combo_loss(model(X_seq=g['X'], As=g['As'], Ac=g['Ac']), g['Y'])
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert torch.allclose(torch.tensor(losses, dtype=torch.float64), g['adam_losses'], rtol=0, atol=2e-5)


def test_sf_shape_fixed_graphs_through_modules():
    g = load_golden('g5_sf_shape')
    model = M.STCGNN(int(g['N']), int(g['C']), int(g['K']), int(g['K']), 1, int(g['h']), int(g['layers']),
                     int(g['horizon']), graph_mode='csr-fixed')
    sd = sub_dict(g, 'sd/')
    assert sorted(model.state_dict().keys()) == sorted(sd.keys())      # no mix_graph_pair.* in csr-fixed mode
    model.load_state_dict(sd)
    Gs, Gc = _leaf(g['Gs']), _leaf(g['Gc'])
    yhat = model(X_seq=g['X'].float(), As=Gs, Ac=Gc)                  # dense Gs handed in directly stays differentiable
    assert rel_err(yhat, g['yhat']) < TOL
    O.combo_loss(yhat, g['Y'].float()).backward()
    grads = dict(model.named_parameters())
    for k, v in sub_dict(g, 'grad/').items():
        assert rel_err(grads[k].grad, v) < 1e-5, k
    assert rel_err(Gs.grad, g['dGs']) < 1e-5 and rel_err(Gc.grad, g['dGc']) < 1e-5


@pytest.mark.parametrize('tag,fname', [('g7', 'g7_csr_n1024'), ('g7p', 'g7_csr_n1024_perm')])
@pytest.mark.parametrize('form', ['CsrGraph', 'torch_sparse'])
def test_csr_fixed_graph_equals_dense_reference(tag, fname, form):
    """A sparse Gs through the CSR path equals the reference fed the same matrix densely (K=3)."""
    g = load_golden(fname)
    s = synth_inputs(tag)
    graph = CsrGraph.from_dense(s['Gs']) if form == 'CsrGraph' else s['Gs'].to_sparse_coo()
    cell = M.STC_Cell(s['N'], s['C'], s['K'], s['K'], s['cin'], s['h'])
    cell.load_state_dict({'gates.W': s['gates_W'], 'gates.b': s['gates_b'], 'candi.W': s['candi_W'], 'candi.b': s['candi_b']})
    Xt, Ht = _leaf(s['Xt']), _leaf(s['Ht'])
    out = cell(graph, s['Gc'], Xt, Ht)
    rows = g['rows']
    assert rel_err(out[:, rows], g['Hout']) < TOL
    (out * s['R']).sum().backward()
    assert rel_err(Xt.grad[:, rows], g['dXt']) < 5e-6
    assert rel_err(Ht.grad[:, rows], g['dHt']) < 5e-6
    assert rel_err(cell.gates.W.grad, g['d_gates_W']) < 2e-5
    assert rel_err(cell.candi.W.grad, g['d_candi_W']) < 2e-5
    assert rel_err(cell.gates.b.grad, g['d_gates_b']) < 2e-5


def test_shape_errors_are_python_exceptions():
    layer = M.BDG_Dif(2, 2, 5, 4)
    with pytest.raises(ValueError):
        layer(torch.randn(2, 12, 3), torch.randn(12, 12), torch.randn(3, 3))          # rank
    with pytest.raises(ValueError):
        layer(torch.randn(2, 12, 3, 5), torch.randn(11, 11), torch.randn(3, 3))       # node count
    with pytest.raises(ValueError):
        layer(torch.randn(2, 12, 3, 6), torch.randn(12, 12), torch.randn(3, 3))       # feature width vs W
    cell = M.STC_Cell(12, 3, 2, 2, 1, 4)
    with pytest.raises(AssertionError):
        cell(torch.randn(12, 12), torch.randn(3, 3), torch.randn(2, 12, 3), torch.randn(2, 12, 3, 4))
    with pytest.raises(AssertionError):
        M.STCGNN(12, 3, 2, 2, 1, 4, 2, 2)(torch.randn(2, 4, 12), torch.randn(12, 12), torch.randn(3, 3))
    with pytest.raises(ValueError):
        M.STCGNN(12, 3, 2, 2, 1, 4, 2, 2, graph_mode='nope')
