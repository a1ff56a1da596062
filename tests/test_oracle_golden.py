"""Pin the CPU oracle to the reference: golden vectors (always) + live reference (when present).

Tolerance 1e-6 relative: the oracle follows the reference op for op in fp32,
so differences are MKL blocking noise only (the reference's own fp32-vs-fp64
noise floor is 5e-8).
"""
import numpy as np
import pytest
import torch

from oracle import stc_oracle as O
from tests.conftest import load_golden, rel_err, sub_dict
from tests.golden.make_golden import SF_SHAPE, bench_path_inputs, grid_graph_dense, sample_rows, synth_inputs

TOL = 1e-6


def _leaf(t):
    return t.clone().requires_grad_()


@pytest.mark.parametrize('K', [1, 2, 3])
def test_g1_bdg_dif(K):
    g = load_golden(f'g1_bdg_k{K}')
    X, Gs, Gc, W, b = (_leaf(g[k]) for k in ('X', 'Gs', 'Gc', 'W', 'b'))
    Y = O.bdg_dif(X, Gs, Gc, W, b, K, K)
    assert rel_err(Y, g['Y']) < TOL
    (Y * g['R']).sum().backward()
    assert rel_err(X.grad, g['dX']) < TOL
    assert rel_err(W.grad, g['dW']) < TOL
    assert rel_err(b.grad, g['db']) < TOL
    if K > 1:          # K=1 uses the identity only: reference grads are exact zeros
        assert rel_err(Gs.grad, g['dGs']) < TOL
        assert rel_err(Gc.grad, g['dGc']) < TOL
    else:
        assert int(g['graph_grad_is_none']) == 1 and Gs.grad is None and Gc.grad is None


def test_g1_bdg_dif_no_bias():
    g = load_golden('g1_bdg_nobias')
    Y = O.bdg_dif(g['X'], g['Gs'], g['Gc'], g['W'], None, 2, 2)
    assert rel_err(Y, g['Y']) < TOL


@pytest.mark.parametrize('cin,K', [(1, 2), (1, 3), (4, 2), (4, 3)])
def test_g2_stc_cell(cin, K):
    g = load_golden(f'g2_cell_in{cin}_k{K}')
    sd = {k: _leaf(v) for k, v in sub_dict(g, 'sd/').items()}
    Xt, Ht, Gs, Gc = (_leaf(g[k]) for k in ('Xt', 'Ht', 'Gs', 'Gc'))
    out = O.stc_cell(Gs, Gc, Xt, Ht, sd['gates.W'], sd['gates.b'], sd['candi.W'], sd['candi.b'], K, K)
    assert rel_err(out, g['Hout']) < TOL
    (out * g['R']).sum().backward()
    for name, ref in (('dXt', Xt), ('dHt', Ht), ('dGs', Gs), ('dGc', Gc)):
        assert rel_err(ref.grad, g[name]) < TOL, name
    for k, v in sub_dict(g, 'grad/').items():
        assert rel_err(sd[k].grad, v) < TOL, k


def test_g3_encoder_decoder():
    g = load_golden('g3_encdec')
    K, h, layers = int(g['K']), int(g['h']), int(g['layers'])
    sd = {'encoder.' + k: _leaf(v) for k, v in sub_dict(g, 'enc_sd/').items()}
    X_seq, Gs, Gc = _leaf(g['X_seq']), _leaf(g['Gs']), _leaf(g['Gc'])
    seqs, lasts = O.stc_encoder(Gs, Gc, X_seq, sd, 'encoder', layers, K, K, h)
    assert rel_err(seqs[0], g['seq0']) < TOL and rel_err(seqs[1], g['seq1']) < TOL
    assert rel_err(lasts[0], g['last0']) < TOL and rel_err(lasts[1], g['last1']) < TOL
    ((seqs[0] * g['R0']).sum() + (seqs[1] * g['R1']).sum() + (lasts[0] * g['RL']).sum()).backward()
    assert rel_err(X_seq.grad, g['dX_seq']) < TOL
    assert rel_err(Gs.grad, g['enc_dGs']) < TOL and rel_err(Gc.grad, g['enc_dGc']) < TOL
    for k, v in sub_dict(g, 'enc_grad/').items():
        assert rel_err(sd['encoder.' + k].grad, v) < TOL, k
    s2, l2 = O.stc_encoder(Gs, Gc, X_seq, sd, 'encoder', layers, K, K, h, return_all_layers=False)
    assert len(s2) == int(g['n_last_only']) == 1 and len(l2) == 1
    assert rel_err(s2[0], g['seq_last_only']) < TOL

    sd = {'decoder.' + k: _leaf(v) for k, v in sub_dict(g, 'dec_sd/').items()}
    Gs, Gc, Xd = _leaf(g['Gs']), _leaf(g['Gc']), _leaf(g['Xd'])
    H0 = [_leaf(g['H00']), _leaf(g['H01'])]
    top, states = O.stc_decoder(Gs, Gc, Xd, H0, sd, 'decoder', layers, K, K)
    assert rel_err(top, g['dec_top']) < TOL
    assert rel_err(states[0], g['dec_s0']) < TOL and rel_err(states[1], g['dec_s1']) < TOL
    ((top * g['Rd']).sum() + (states[0] * g['Rs']).sum()).backward()
    assert rel_err(Xd.grad, g['dXd']) < TOL
    assert rel_err(H0[0].grad, g['dH00']) < TOL and rel_err(H0[1].grad, g['dH01']) < TOL
    assert rel_err(Gs.grad, g['dec_dGs']) < TOL and rel_err(Gc.grad, g['dec_dGc']) < TOL
    for k, v in sub_dict(g, 'dec_grad/').items():
        assert rel_err(sd['decoder.' + k].grad, v) < TOL, k


def _adam_step(params, grads, state, lr=2e-3, wd=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam with L2 weight decay (Model_Trainer.py:35), restated."""
    state['t'] += 1
    t = state['t']
    for k in params:
        g = grads[k] + wd * params[k]
        m = state['m'].setdefault(k, torch.zeros_like(g))
        v = state['v'].setdefault(k, torch.zeros_like(g))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / (1 - b2 ** t) ** 0.5).add_(eps)
        params[k] = params[k] - (lr / (1 - b1 ** t)) * m / denom


def test_g4_full_model_loss_grads_and_adam_trajectory():
    g = load_golden('g4_stcgnn_small')
    cfg = dict(Ks=int(g['K']), Kc=int(g['K']), hidden=int(g['h']), num_layers=int(g['layers']),
               out_horizon=int(g['horizon']))
    sd = {k: _leaf(v) for k, v in sub_dict(g, 'sd/').items()}
    Gs, Gc = O.mgp_gen(g['X'], g['As'], g['Ac'], sd)
    assert rel_err(Gs, g['Gs']) < TOL and rel_err(Gc, g['Gc']) < TOL
    yhat = O.stcgnn_forward(g['X'], g['As'], g['Ac'], sd, **cfg)
    assert yhat.shape == g['yhat'].shape
    assert rel_err(yhat, g['yhat']) < TOL
    loss = O.combo_loss(yhat, g['Y'])
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-6
    loss.backward()
    for k, v in sub_dict(g, 'grad/').items():
        assert rel_err(sd[k].grad, v) < 5e-6, k
    # five Adam steps from the stored initial parameters
    params = {k: v.detach().clone() for k, v in sd.items()}
    state = dict(t=0, m={}, v={})
    losses = []
    for _ in range(5):
        leaves = {k: _leaf(v) for k, v in params.items()}
        l = O.combo_loss(O.stcgnn_forward(g['X'], g['As'], g['Ac'], leaves, **cfg), g['Y'])
        l.backward()
        losses.append(float(l.detach()))
        _adam_step(params, {k: v.grad for k, v in leaves.items()}, state)
    assert np.allclose(losses, g['adam_losses'].numpy(), rtol=0, atol=2e-5), (losses, g['adam_losses'])
    with torch.no_grad():
        y5 = O.stcgnn_forward(g['X'], g['As'], g['Ac'], params, **cfg)
    assert rel_err(y5, g['yhat_after5']) < 1e-4


def test_g5_sf_shape_fixed_graphs():
    g = load_golden('g5_sf_shape')
    cfg = dict(Ks=int(g['K']), Kc=int(g['K']), hidden=int(g['h']), num_layers=int(g['layers']),
               out_horizon=int(g['horizon']))
    sd = {k: _leaf(v) for k, v in sub_dict(g, 'sd/').items()}
    assert sum(v.numel() for v in sd.values()) == 22033
    Gs, Gc = _leaf(g['Gs']), _leaf(g['Gc'])
    yhat = O.encdec_forward(g['X'].float(), Gs, Gc, sd, **cfg)
    assert yhat.shape == (32, 3, 100, 5)
    assert rel_err(yhat, g['yhat']) < TOL
    loss = O.combo_loss(yhat, g['Y'].float())
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-6
    loss.backward()
    for k, v in sub_dict(g, 'grad/').items():
        assert rel_err(sd[k].grad, v) < 5e-6, k
    assert rel_err(Gs.grad, g['dGs']) < 5e-6 and rel_err(Gc.grad, g['dGc']) < 5e-6


def test_g6_mgp_gen_and_mixed_fusion():
    g = load_golden('g6_mgp')
    sd = {'mix_graph_pair.' + k: v for k, v in sub_dict(g, 'small_sd/').items()}
    Gs, Gc = O.mgp_gen(g['small_X'], g['small_As'], g['small_Ac'], sd)
    assert rel_err(Gs, g['small_Gs']) < TOL and rel_err(Gc, g['small_Gc']) < TOL
    # SF size: regenerate the closed-form fusion weights of make_golden.py
    sd = {'mix_graph_pair.' + k: v for k, v in sub_dict(g, 'sf_sd/').items()}
    n = 100
    idx = torch.arange(n * n, dtype=torch.float32)
    for lin, (a, b) in (('lin_A', (0.37, 0.11)), ('lin_P', (0.23, 0.19))):
        sd[f'mix_graph_pair.aggreg_S.{lin}.weight'] = torch.sin(a * idx[:, None] + b * idx[None, :]) / (n * n)
        sd[f'mix_graph_pair.aggreg_S.{lin}.bias'] = torch.cos(0.05 * idx) * 0.1
    Gs, Gc = O.mgp_gen(g['sf_X'].float(), g['sf_As'], g['sf_Ac'], sd)
    assert rel_err(Gs, g['sf_Gs']) < TOL and rel_err(Gc, g['sf_Gc']) < TOL
    assert abs(float(Gs.sum(1).mean()) - float(g['sf_Gs'].sum(1).mean())) < 1e-5


def _dense_to_sparse_T(Gs):
    return Gs.t().contiguous().to_sparse_csr()


@pytest.mark.parametrize('tag,fname', [('g7', 'g7_csr_n1024'), ('g7p', 'g7_csr_n1024_perm')])
def test_g7_dense_reference_vs_sparse_restatement(tag, fname):
    """The sparse/feature-side form equals the reference fed the dense matrix (K=3)."""
    g = load_golden(fname)
    s = synth_inputs(tag)
    for chk, key in (('chk_Gs', 'Gs'), ('chk_Xt', 'Xt'), ('chk_Ht', 'Ht'), ('chk_W', 'gates_W')):
        assert abs(float(s[key].double().sum()) - float(g[chk])) < 1e-6 * max(1.0, abs(float(g[chk]))), chk
    rows = g['rows']
    GsT = _dense_to_sparse_T(s['Gs'])
    leaves = {k: _leaf(s[k]) for k in ('Xt', 'Ht', 'gates_W', 'gates_b', 'candi_W', 'candi_b')}
    out = O.stc_cell(GsT, s['Gc'], leaves['Xt'], leaves['Ht'], leaves['gates_W'], leaves['gates_b'],
                     leaves['candi_W'], leaves['candi_b'], s['K'], s['K'], conv=O.bdg_dif_sparse)
    assert rel_err(out[:, rows], g['Hout']) < 2e-6
    (out * s['R']).sum().backward()
    assert rel_err(leaves['Xt'].grad[:, rows], g['dXt']) < 5e-6
    assert rel_err(leaves['Ht'].grad[:, rows], g['dHt']) < 5e-6
    for k in ('gates_W', 'gates_b', 'candi_W', 'candi_b'):
        assert rel_err(leaves[k].grad, g['d_' + k]) < 2e-5, k
    # and the dense restatement itself
    with torch.no_grad():
        dense = O.stc_cell(s['Gs'], s['Gc'], s['Xt'], s['Ht'], s['gates_W'], s['gates_b'],
                           s['candi_W'], s['candi_b'], s['K'], s['K'])
    assert rel_err(dense[:, rows], g['Hout']) < TOL


def test_g8_large_n_sparse_restatement_vs_dense_reference():
    g = load_golden('g8_large_n10000')
    s = synth_inputs('g8')
    assert abs(float(s['Gs'].double().sum()) - float(g['chk_Gs'])) < 1e-3
    assert abs(float(s['Xt'].double().sum()) - float(g['chk_Xt'])) < 1e-3
    GsT = _dense_to_sparse_T(s['Gs'])
    del s['Gs']
    with torch.no_grad():
        out = O.stc_cell(GsT, s['Gc'], s['Xt'], s['Ht'], s['gates_W'], s['gates_b'],
                         s['candi_W'], s['candi_b'], s['K'], s['K'], conv=O.bdg_dif_sparse)
    assert rel_err(out[:, g['rows']], g['Hout']) < 2e-6


def test_queen_grid_matches_sf_formula():
    r, c = O.queen_grid_adjacency(10, 10)
    assert r.numel() == 684                                  # nnz of the SF s_adj
    for (H, W) in ((10, 20), (100, 100), (224, 224)):
        nnz = 8 * (H - 2) * (W - 2) + 5 * (2 * (H - 2) + 2 * (W - 2)) + 12
        assert O.queen_grid_adjacency(H, W)[0].numel() == nnz
    A = grid_graph_dense(3, 4)
    assert torch.allclose(A.sum(1), torch.ones(12))
    assert not torch.equal(A, A.t())                         # row-normalised grid is non-symmetric
    assert sample_rows(100, 10).numel() == 10


# ------------------------------------------------------------------ live reference (build container only)
@pytest.mark.parametrize('seed', range(6))
def test_live_reference_random_shapes(reference_module, seed):
    ref = reference_module
    g = torch.Generator().manual_seed(1000 + seed)
    B = int(torch.randint(1, 4, (1,), generator=g))
    N = int(torch.randint(2, 20, (1,), generator=g))
    C = int(torch.randint(1, 7, (1,), generator=g))
    cin = int(torch.randint(1, 5, (1,), generator=g))
    h = int(torch.randint(1, 9, (1,), generator=g))
    K = int(torch.randint(1, 5, (1,), generator=g))
    torch.manual_seed(seed)
    cell = ref.STC_Cell(N, C, K, K, cin, h)
    Xt, Ht = torch.randn(B, N, C, cin), torch.randn(B, N, C, h)
    Gs, Gc = torch.randn(N, N) * 0.2, torch.randn(C, C) * 0.3
    want = cell(Gs, Gc, Xt, Ht)
    got = O.stc_cell(Gs, Gc, Xt, Ht, cell.gates.W, cell.gates.b, cell.candi.W, cell.candi.b, K, K)
    assert rel_err(got, want) < TOL


def test_live_reference_sf_data_adjacency(reference_module):
    import os
    path = '/root/reference/data/SF-incidents-4h.npz'
    if not os.path.exists(path):
        pytest.skip('SF data not present')
    z = np.load(path)
    r, c = O.queen_grid_adjacency(10, 10)
    A = np.zeros((100, 100), dtype=z['s_adj'].dtype)
    A[r.numpy(), c.numpy()] = 1
    assert np.array_equal(A, z['s_adj'])


@pytest.mark.parametrize('name,C,K', [('g11_bench_c32', 32, 2), ('g12_bench_c64', 64, 2), ('g13_bench_c32_k3', 32, 3), ('g14_sf_shape', 5, 2), ('g15_sf_shape_k3', 5, 3)])
def test_g11_g13_bench_path_widths(name, C, K):
    """The widths the bench runs (C = 32 / 64, hidden 16, K = 2 / 3) through the REFERENCE's encoder-decoder-head: the oracle
    (dense form and sparse feature-side form) reproduces prediction, ComboLoss and every parameter gradient."""
    g = load_golden(name)
    s = bench_path_inputs(C, K, **(SF_SHAPE if name in ('g14_sf_shape', 'g15_sf_shape_k3') else {}))
    assert abs(float(s['Gs'].double().sum()) - float(g['chk_Gs'])) < 1e-9 and abs(float(s['X'].double().sum()) - float(g['chk_X'])) < 1e-9
    assert abs(float(s['Gc'].double().sum()) - float(g['chk_Gc'])) < 1e-6
    for conv, Gs in ((O.bdg_dif, s['Gs']), (O.bdg_dif_sparse, s['Gs'].t().contiguous().to_sparse_csr())):
        sd = {k: _leaf(v) for k, v in sub_dict(g, 'sd/').items()}
        yhat = O.encdec_forward(s['X'], Gs, s['Gc'], sd, K, K, int(g['h']), int(g['layers']), int(g['horizon']), conv=conv)
        assert rel_err(yhat, g['yhat']) < TOL
        loss = O.combo_loss(yhat, s['Y'])
        assert abs(float(loss.detach()) - float(g['loss'])) < 2e-6
        loss.backward()
        for k, v in sub_dict(g, 'grad/').items():
            assert rel_err(sd[k].grad, v) < (2e-6 if conv is O.bdg_dif else 5e-6), (k, conv.__name__)


def test_g8b_large_n_gradients_sparse_oracle():
    """N = 10 000 with backward: the sparse feature-side oracle (the CPU baseline's form) against sampled rows of the dense
    reference's input gradients and its full parameter gradients."""
    g = load_golden('g8b_large_n10000_grads')
    s = synth_inputs('g8')
    GsT = s['Gs'].t().contiguous().to_sparse_csr()
    Xt, Ht = _leaf(s['Xt']), _leaf(s['Ht'])
    p = [_leaf(s[k]) for k in ('gates_W', 'gates_b', 'candi_W', 'candi_b')]
    out = O.stc_cell(GsT, s['Gc'], Xt, Ht, *p, s['K'], s['K'], conv=O.bdg_dif_sparse)
    rows = g['rows']
    assert rel_err(out[:, rows], g['Hout']) < 2e-6
    (out * s['R']).sum().backward()
    assert rel_err(Xt.grad[:, rows], g['dXt']) < 5e-6 and rel_err(Ht.grad[:, rows], g['dHt']) < 5e-6
    for t, k in zip(p, ('d_gates_W', 'd_gates_b', 'd_candi_W', 'd_candi_b')):
        assert rel_err(t.grad, g[k]) < 2e-5, k


def test_g8c_large_n_order_3_sparse_oracle():
    """N = 10 000 at Chebyshev order 3 (BASELINE configuration 4) with backward: the sparse feature-side oracle against sampled rows of
    the dense reference (matrix-side cheby_poly, STC_GNN.py:24-29) and its full parameter gradients."""
    g = load_golden('g8c_large_n10000_k3')
    s = synth_inputs('g8c')
    assert s['K'] == 3
    GsT = s['Gs'].t().contiguous().to_sparse_csr()
    Xt, Ht = _leaf(s['Xt']), _leaf(s['Ht'])
    p = [_leaf(s[k]) for k in ('gates_W', 'gates_b', 'candi_W', 'candi_b')]
    out = O.stc_cell(GsT, s['Gc'], Xt, Ht, *p, 3, 3, conv=O.bdg_dif_sparse)
    rows = g['rows']
    assert rel_err(out[:, rows], g['Hout']) < 2e-6
    (out * s['R']).sum().backward()
    assert rel_err(Xt.grad[:, rows], g['dXt']) < 5e-6 and rel_err(Ht.grad[:, rows], g['dHt']) < 5e-6
    for t, k in zip(p, ('d_gates_W', 'd_gates_b', 'd_candi_W', 'd_candi_b')):
        assert rel_err(t.grad, g[k]) < 2e-5, k
