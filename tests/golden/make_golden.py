#!/usr/bin/env python3
"""Generate the golden vectors in this directory FROM THE REFERENCE.

Run in the build container only (the reference does not travel):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports ``/root/reference/framework/STC_GNN.py`` and
``Model_Trainer.ComboLoss`` unmodified, runs them on seeded inputs and stores
inputs, parameters (the reference's own ``state_dict`` keys), outputs and
gradients as small ``.npz`` files.  Nothing of the reference's source text is
stored, only numbers.  Sets follow SURVEY.md section 8(c4):

  g1_bdg_k{1,2,3}   BDG_Dif fwd + all grads, non-symmetric Gs/Gc
  g2_cell_*         STC_Cell fwd + all grads
  g3_encdec         STC_Encoder (both return_all_layers) and STC_Decoder
  g4_stcgnn_small   full STCGNN (with MGP_Gen) + ComboLoss + grads + 5 Adam steps
  g5_sf_shape       SF shape (B=32,T=9,N=100,C=5,h=16), Gs/Gc stored, enc/dec/head only
  g6_mgp            MGP_Gen / MixedFusion at N=12 and N=100 (closed-form weights)
  g7_csr_n1024      one STC_Cell on a 32x32 row-normalised queen grid (+permuted), dense reference
  g8_large_n10000   one STC_Cell at N=10 000, C=32, h=16 through the dense reference, sampled rows
  g9_pipeline       Data_Container windows/split/batches and a 2-epoch Model_Trainer run on a synthetic series
  g10_metrics       Metrics.mask_data and ModelEvaluator.one_step_eval_bi on synthetic predictions
  g11_bench_c32     the BENCH path's widths through the reference: encoder-decoder-head (graphs given, as g5) at C=32, h=16, K=2,
                    2 layers, 6x8 non-symmetric grid graph, B=2, T=3+2 -> yhat, ComboLoss, every gradient
  g12_bench_c64     the same at C=64;  g13_bench_c32_k3  the same at C=32, K=3 (configuration 4's order)
  g14_sf_shape      the same at the SF-incidents shape: N=100 (10 x 10), C=5, T=9+3, K=2, h=16 (the small-graph cell kernels' shape)
  g15_sf_shape_k3   the same at K=3 (the small-graph cell kernels' order-3 form: T_2(S) as a second graph)
  g8b_large_n10000_grads   g8 with backward: sampled rows of Ht, dXt, dHt + the full parameter gradients
  g8c_large_n10000_k3      the same cell at Chebyshev order K = 3 (BASELINE configuration 4) through the dense reference, with backward

Large inputs (g7, g8) are regenerated from seeds by ``synth_inputs`` below,
which the tests import too; a few checksums are stored to catch RNG drift.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from oracle.stc_oracle import queen_grid_adjacency  # noqa: E402  (graph generator only)


# ---------------------------------------------------------------- shared, importable by tests
def grid_graph_dense(H, W, permute_seed=None):
    """Row-stochastic queen-grid adjacency A/rowsum(A) as a dense (N,N) fp32 tensor."""
    r, c = queen_grid_adjacency(H, W)
    N = H * W
    A = torch.zeros(N, N)
    A[r, c] = 1.0
    A = A / A.sum(1, keepdim=True)
    if permute_seed is not None:
        g = torch.Generator().manual_seed(permute_seed)
        p = torch.randperm(N, generator=g)
        A = A[p][:, p]
    return A


def synth_inputs(name):
    """Seeded inputs for the large sets; identical in make_golden and in the tests."""
    if name in ('g7', 'g7p'):
        H = W = 32
        N, C, cin, h, K, B = H * W, 8, 2, 8, 3, 2
        Gs = grid_graph_dense(H, W, permute_seed=1234 if name == 'g7p' else None)
    elif name in ('g8', 'g8c'):
        H = W = 100
        N, C, cin, h, K, B = H * W, 32, 16, 16, (3 if name == 'g8c' else 2), 1
        Gs = grid_graph_dense(H, W)
    else:
        raise KeyError(name)
    g = torch.Generator().manual_seed(7)
    Gc = torch.softmax(torch.randn(C, C, generator=g), -1)
    g = torch.Generator().manual_seed(0)
    Xt = (torch.rand(B, N, C, cin, generator=g) < 0.1635).float()
    Ht = torch.tanh(torch.randn(B, N, C, h, generator=g))
    g = torch.Generator().manual_seed(42)
    L = cin + h
    gw = torch.randn(K * K * L, 2 * h, generator=g) * (2.0 / (K * K * L + 2 * h)) ** 0.5
    cw = torch.randn(K * K * L, h, generator=g) * (2.0 / (K * K * L + h)) ** 0.5
    gb = torch.randn(2 * h, generator=g) * 0.1
    cb = torch.randn(h, generator=g) * 0.1
    R = torch.randn(B, N, C, h, generator=g)
    return dict(N=N, C=C, cin=cin, h=h, K=K, B=B, Gs=Gs, Gc=Gc, Xt=Xt, Ht=Ht,
                gates_W=gw, gates_b=gb, candi_W=cw, candi_b=cb, R=R)


def bench_path_inputs(C, K, H=6, W=8, h=16, layers=2, horizon=2, B=2, T=3):
    """Inputs of g11-g13: the widths the bench runs (C in {32, 64}, hidden 16) on a small NON-symmetric graph -- the queen grid
    with every entry scaled by a random factor in [0.5, 1.5), then row-normalised -- and a non-symmetric category graph."""
    N = H * W
    g = torch.Generator().manual_seed(1100 + C + K)
    A = grid_graph_dense(H, W)
    A = A * (0.5 + torch.rand(N, N, generator=g)) * (A > 0)
    Gs = A / A.sum(1, keepdim=True)
    Gc = torch.softmax(torch.randn(C, C, generator=g), -1)
    X = (torch.rand(B, T, N, C, generator=g) < 0.3).float()
    Y = (torch.rand(B, horizon, N, C, generator=g) < 0.3).float()
    return dict(N=N, C=C, K=K, h=h, layers=layers, horizon=horizon, Gs=Gs, Gc=Gc, X=X, Y=Y)


def sample_rows(N, count, seed=99):
    g = torch.Generator().manual_seed(seed)
    return torch.sort(torch.randperm(N, generator=g)[:count]).values


# ---------------------------------------------------------------- generation
def _np(t):
    return t.detach().cpu().numpy()


def _save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: (_np(v) if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f'{name}.npz  {os.path.getsize(path)/1024:.1f} KB')


def _sd_arrays(module, prefix='sd/'):
    return {prefix + k: v.clone() for k, v in module.state_dict().items()}


def _grad_arrays(module, prefix='grad/'):
    return {prefix + k: p.grad.clone() for k, p in module.named_parameters()}


def main():
    sys.path.insert(0, '/root/reference/framework')
    import STC_GNN as ref                       # the reference, unmodified
    from Model_Trainer import ComboLoss          # reference loss

    torch.set_num_threads(8)

    # ---------------- G1: BDG_Dif
    B, N, C, L, Ho = 2, 12, 3, 5, 4
    for K in (1, 2, 3):
        torch.manual_seed(100 + K)
        layer = ref.BDG_Dif(K, K, L, Ho)
        with torch.no_grad():
            layer.b.normal_(0, 0.3)
        X = torch.randn(B, N, C, L, requires_grad=True)
        Gs = (torch.randn(N, N) * 0.3).requires_grad_()
        Gc = (torch.randn(C, C) * 0.5).requires_grad_()
        R = torch.randn(B, N, C, Ho)
        Y = layer(X, Gs, Gc)
        (Y * R).sum().backward()
        # K=1 touches only T_0 = I: the reference leaves Gs.grad/Gc.grad at None -> stored as zeros
        dGs = Gs.grad if Gs.grad is not None else torch.zeros_like(Gs)
        dGc = Gc.grad if Gc.grad is not None else torch.zeros_like(Gc)
        _save(f'g1_bdg_k{K}', K=K, X=X, Gs=Gs, Gc=Gc, R=R, W=layer.W, b=layer.b, Y=Y,
              dX=X.grad, dGs=dGs, dGc=dGc, dW=layer.W.grad, db=layer.b.grad,
              graph_grad_is_none=int(Gs.grad is None))
    # no-bias variant at K=2
    torch.manual_seed(111)
    layer = ref.BDG_Dif(2, 2, L, Ho, use_bias=False)
    X = torch.randn(B, N, C, L)
    Gs = torch.randn(N, N) * 0.3
    Gc = torch.randn(C, C) * 0.5
    _save('g1_bdg_nobias', K=2, X=X, Gs=Gs, Gc=Gc, W=layer.W, Y=layer(X, Gs, Gc))

    # ---------------- G2: STC_Cell
    h = 4
    for cin in (1, 4):
        for K in (2, 3):
            torch.manual_seed(200 + 10 * cin + K)
            cell = ref.STC_Cell(N, C, K, K, cin, h)
            with torch.no_grad():
                cell.gates.b.normal_(0, 0.3)
                cell.candi.b.normal_(0, 0.3)
            Xt = torch.randn(B, N, C, cin, requires_grad=True)
            Ht = torch.randn(B, N, C, h, requires_grad=True)
            Gs = (torch.randn(N, N) * 0.3).requires_grad_()
            Gc = (torch.randn(C, C) * 0.5).requires_grad_()
            R = torch.randn(B, N, C, h)
            out = cell(Gs, Gc, Xt, Ht)
            (out * R).sum().backward()
            _save(f'g2_cell_in{cin}_k{K}', K=K, cin=cin, h=h, Xt=Xt, Ht=Ht, Gs=Gs, Gc=Gc, R=R,
                  Hout=out, dXt=Xt.grad, dHt=Ht.grad, dGs=Gs.grad, dGc=Gc.grad,
                  **_sd_arrays(cell), **_grad_arrays(cell))

    # ---------------- G3: encoder / decoder
    torch.manual_seed(300)
    T, layers, K = 4, 2, 2
    enc = ref.STC_Encoder(N, C, K, K, 1, h, layers, return_all_layers=True)
    dec = ref.STC_Decoder(N, C, K, K, h, h, layers, out_horizon=2)
    for m in list(enc.cell_list) + list(dec.cell_list):
        with torch.no_grad():
            m.gates.b.normal_(0, 0.3)
            m.candi.b.normal_(0, 0.3)
    X_seq = torch.randn(B, T, N, C, 1, requires_grad=True)
    Gs = (torch.randn(N, N) * 0.3).requires_grad_()
    Gc = (torch.randn(C, C) * 0.5).requires_grad_()
    seqs, lasts = enc(Gs, Gc, X_seq)
    R0 = torch.randn_like(seqs[0])
    R1 = torch.randn_like(seqs[1])
    RL = torch.randn_like(lasts[0])
    ((seqs[0] * R0).sum() + (seqs[1] * R1).sum() + (lasts[0] * RL).sum()).backward()
    enc.return_all_layers = False
    seqs_last, lasts_last = enc(Gs, Gc, X_seq)
    enc_arrays = dict(X_seq=X_seq, Gs=Gs, Gc=Gc, R0=R0, R1=R1, RL=RL,
                      seq0=seqs[0], seq1=seqs[1], last0=lasts[0], last1=lasts[1],
                      n_last_only=len(seqs_last), seq_last_only=seqs_last[0],
                      dX_seq=X_seq.grad, enc_dGs=Gs.grad, enc_dGc=Gc.grad,
                      **_sd_arrays(enc, 'enc_sd/'), **_grad_arrays(enc, 'enc_grad/'))
    Gs2 = Gs.detach().clone().requires_grad_()
    Gc2 = Gc.detach().clone().requires_grad_()
    Xd = torch.randn(B, N, C, h, requires_grad=True)
    H0 = [torch.randn(B, N, C, h, requires_grad=True) for _ in range(layers)]
    top, states = dec(Gs2, Gc2, Xd, H0)
    Rd = torch.randn_like(top)
    Rs = torch.randn_like(states[0])
    ((top * Rd).sum() + (states[0] * Rs).sum()).backward()
    _save('g3_encdec', K=K, h=h, layers=layers, **enc_arrays,
          Xd=Xd, H00=H0[0], H01=H0[1], Rd=Rd, Rs=Rs, dec_top=top, dec_s0=states[0], dec_s1=states[1],
          dXd=Xd.grad, dH00=H0[0].grad, dH01=H0[1].grad, dec_dGs=Gs2.grad, dec_dGc=Gc2.grad,
          **_sd_arrays(dec, 'dec_sd/'), **_grad_arrays(dec, 'dec_grad/'))

    # ---------------- G4: full STCGNN small + ComboLoss + Adam trajectory
    torch.manual_seed(400)
    Hh, Ww = 3, 4
    N4, C4, h4, K4, layers4, hor4, B4, T4 = Hh * Ww, 3, 4, 2, 2, 2, 3, 4
    r, c = queen_grid_adjacency(Hh, Ww)
    As = torch.zeros(N4, N4)
    As[r, c] = 1.0
    Ac = torch.rand(C4, C4)
    model = ref.STCGNN(N4, C4, K4, K4, 1, h4, layers4, hor4)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    X4 = (torch.rand(B4, T4, N4, C4) < 0.3).float()
    Y4 = (torch.rand(B4, hor4, N4, C4) < 0.3).float()
    crit = ComboLoss()
    Gs4, Gc4 = model.mix_graph_pair(X4, As, Ac)
    yhat = model(X4, As, Ac)
    loss = crit(yhat, Y4)
    loss.backward()
    g4 = dict(N=N4, C=C4, h=h4, K=K4, layers=layers4, horizon=hor4, As=As, Ac=Ac, X=X4, Y=Y4,
              Gs=Gs4, Gc=Gc4, yhat=yhat, loss=loss)
    g4.update({'sd/' + k: v for k, v in sd0.items()})
    g4.update(_grad_arrays(model))
    opt = torch.optim.Adam(model.parameters(), lr=2e-3, weight_decay=1e-4)
    traj = []
    for _ in range(5):
        out = model(X_seq=X4, As=As, Ac=Ac)
        l = crit(out, Y4)
        opt.zero_grad()
        l.backward()
        opt.step()
        traj.append(float(l.detach()))
    g4['adam_losses'] = np.array(traj, dtype=np.float64)
    g4['yhat_after5'] = model(X4, As, Ac)
    _save('g4_stcgnn_small', **g4)

    # ---------------- G5: SF shape, graphs fixed (outputs of the reference's MGP_Gen)
    torch.manual_seed(500)
    N5, C5, h5, K5, layers5, hor5, B5, T5 = 100, 5, 16, 2, 2, 3, 32, 9
    r, c = queen_grid_adjacency(10, 10)
    As5 = torch.zeros(N5, N5)
    As5[r, c] = 1.0
    Ac5 = torch.rand(C5, C5)
    X5 = (torch.rand(B5, T5, N5, C5) < 0.1635).float()
    Y5 = (torch.rand(B5, hor5, N5, C5) < 0.1635).float()
    # closed-form MixedFusion weights keep the 800 MB out of the RNG stream; the
    # resulting Gs is still the reference's MGP_Gen output for these weights.
    full = ref.STCGNN(N5, C5, K5, K5, 1, h5, layers5, hor5)
    with torch.no_grad():
        for fus, n in ((full.mix_graph_pair.aggreg_S, N5), (full.mix_graph_pair.aggreg_C, C5)):
            idx = torch.arange(n * n, dtype=torch.float32)
            for lin, (a, bb) in ((fus.lin_A, (0.37, 0.11)), (fus.lin_P, (0.23, 0.19))):
                lin.weight.copy_(torch.sin(a * idx[:, None] + bb * idx[None, :]) / (n * n))
                lin.bias.copy_(torch.cos(0.05 * idx) * 0.1)
        Gs5, Gc5 = full.mix_graph_pair(X5, As5, Ac5)
    Gs5 = Gs5.detach().clone().requires_grad_()
    Gc5 = Gc5.detach().clone().requires_grad_()
    x5 = X5.unsqueeze(-1)
    _, states = full.encoder(Gs=Gs5, Gc=Gc5, X_seq=x5, H0_l=None)
    dec_in = states[-1]
    outs = []
    for _ in range(hor5):
        dec_in, states = full.decoder(Gs=Gs5, Gc=Gc5, Xt=dec_in, H0_l=states)
        outs.append(dec_in)
    yhat5 = torch.sigmoid(full.out_proj(torch.stack(outs, 1))).squeeze(-1)
    loss5 = ComboLoss()(yhat5, Y5)
    loss5.backward()
    g5 = dict(N=N5, C=C5, h=h5, K=K5, layers=layers5, horizon=hor5,
              X=X5.to(torch.uint8), Y=Y5.to(torch.uint8), Gs=Gs5, Gc=Gc5, yhat=yhat5, loss=loss5,
              dGs=Gs5.grad, dGc=Gc5.grad)
    for k, v in full.state_dict().items():
        if not k.startswith('mix_graph_pair'):
            g5['sd/' + k] = v.clone()
    for k, p in full.named_parameters():
        if not k.startswith('mix_graph_pair'):
            g5['grad/' + k] = p.grad.clone()
    _save('g5_sf_shape', **g5)

    # ---------------- G6: MGP_Gen / MixedFusion
    torch.manual_seed(600)
    gen = ref.MGP_Gen(N4, C4, h4)
    Xg = (torch.rand(B4, T4, N4, C4) < 0.3).float()
    Gs6, Gc6 = gen(Xg, As, Ac)
    g6 = dict(small_X=Xg, small_As=As, small_Ac=Ac, small_Gs=Gs6, small_Gc=Gc6)
    g6.update({'small_sd/' + k: v.clone() for k, v in gen.state_dict().items()})
    # N=100 with closed-form fusion weights (regenerated by the test), random Wu/Wv stored
    g6.update(dict(sf_X=X5.to(torch.uint8), sf_As=As5, sf_Ac=Ac5, sf_Gs=Gs5.detach(), sf_Gc=Gc5.detach()))
    for k, v in full.mix_graph_pair.state_dict().items():
        if 'params_' in k or 'aggreg_C' in k:
            g6['sf_sd/' + k] = v.clone()
    _save('g6_mgp', **g6)

    # ---------------- G7: 32x32 grid (+permuted), dense reference, one cell, sampled rows
    for tag in ('g7', 'g7p'):
        s = synth_inputs(tag)
        cell = ref.STC_Cell(s['N'], s['C'], s['K'], s['K'], s['cin'], s['h'])
        with torch.no_grad():
            cell.gates.W.copy_(s['gates_W']); cell.gates.b.copy_(s['gates_b'])
            cell.candi.W.copy_(s['candi_W']); cell.candi.b.copy_(s['candi_b'])
        Xt = s['Xt'].clone().requires_grad_()
        Ht = s['Ht'].clone().requires_grad_()
        out = cell(s['Gs'], s['Gc'], Xt, Ht)
        (out * s['R']).sum().backward()
        rows = sample_rows(s['N'], 128)
        _save('g7_csr_n1024' + ('_perm' if tag == 'g7p' else ''), rows=rows,
              Hout=out[:, rows], dXt=Xt.grad[:, rows], dHt=Ht.grad[:, rows],
              d_gates_W=cell.gates.W.grad, d_gates_b=cell.gates.b.grad,
              d_candi_W=cell.candi.W.grad, d_candi_b=cell.candi.b.grad,
              chk_Gs=s['Gs'].double().sum(), chk_Xt=s['Xt'].double().sum(),
              chk_Ht=s['Ht'].double().sum(), chk_W=s['gates_W'].double().sum())

    # ---------------- G8: N = 10 000 through the dense reference, forward only
    s = synth_inputs('g8')
    cell = ref.STC_Cell(s['N'], s['C'], s['K'], s['K'], s['cin'], s['h'])
    with torch.no_grad():
        cell.gates.W.copy_(s['gates_W']); cell.gates.b.copy_(s['gates_b'])
        cell.candi.W.copy_(s['candi_W']); cell.candi.b.copy_(s['candi_b'])
        out = cell(s['Gs'], s['Gc'], s['Xt'], s['Ht'])
    rows = sample_rows(s['N'], 128)
    _save('g8_large_n10000', rows=rows, Hout=out[:, rows],
          chk_Gs=s['Gs'].double().sum(), chk_Xt=s['Xt'].double().sum(),
          chk_Ht=s['Ht'].double().sum(), chk_W=s['gates_W'].double().sum())


SF_SHAPE = dict(H=10, W=10, horizon=3, B=3, T=9)          # g14: the SF-incidents shape (N = 100, C = 5, T = 9 + 3), graphs given
BENCH_PATH_GOLDENS = (('g11_bench_c32', 32, 2, {}), ('g12_bench_c64', 64, 2, {}), ('g13_bench_c32_k3', 32, 3, {}), ('g14_sf_shape', 5, 2, SF_SHAPE),
                      ('g15_sf_shape_k3', 5, 3, SF_SHAPE))      # g15: the SF shape at Chebyshev order 3 (Main.py:24 -cheby_order 3)


def bench_path_golden(ref_framework='/root/reference/framework', only=None):
    """g11 / g12 / g13 / g14 and g8b: the reference itself at the widths (and, g8b, a size) the bench runs; g14 at the SF shape."""
    sys.path.insert(0, ref_framework)
    import STC_GNN as ref
    from Model_Trainer import ComboLoss
    torch.set_num_threads(8)
    for name, C, K, kw in BENCH_PATH_GOLDENS:
        if only is not None and name != only:
            continue
        s = bench_path_inputs(C, K, **kw)
        torch.manual_seed(1100 + C + 10 * K)
        full = ref.STCGNN(s['N'], C, K, K, 1, s['h'], s['layers'], s['horizon'])
        with torch.no_grad():
            for m in list(full.encoder.cell_list) + list(full.decoder.cell_list):
                m.gates.b.normal_(0, 0.3)
                m.candi.b.normal_(0, 0.3)
        Gs = s['Gs'].clone().requires_grad_()
        Gc = s['Gc'].clone().requires_grad_()
        # STCGNN.forward after MGP_Gen (STC_GNN.py:189-207) on the reference's own modules, graphs handed in
        _, states = full.encoder(Gs=Gs, Gc=Gc, X_seq=s['X'].unsqueeze(-1), H0_l=None)
        dec_in, outs = states[-1], []
        for _ in range(s['horizon']):
            dec_in, states = full.decoder(Gs=Gs, Gc=Gc, Xt=dec_in, H0_l=states)
            outs.append(dec_in)
        yhat = torch.sigmoid(full.out_proj(torch.stack(outs, 1))).squeeze(-1)
        loss = ComboLoss()(yhat, s['Y'])
        loss.backward()
        out = dict(N=s['N'], C=C, K=K, h=s['h'], layers=s['layers'], horizon=s['horizon'], yhat=yhat, loss=loss,
                   chk_Gs=s['Gs'].double().sum(), chk_X=s['X'].double().sum(), chk_Gc=s['Gc'].double().sum())
        for k, v in full.state_dict().items():
            if not k.startswith('mix_graph_pair'):
                out['sd/' + k] = v.clone()
        for k, p in full.named_parameters():
            if not k.startswith('mix_graph_pair'):
                out['grad/' + k] = p.grad.clone()
        _save(name, **out)
    if only is not None:
        return
    # g8b: N = 10 000 through the dense reference WITH backward (sampled rows of the input gradients, full parameter gradients)
    s = synth_inputs('g8')
    cell = ref.STC_Cell(s['N'], s['C'], s['K'], s['K'], s['cin'], s['h'])
    with torch.no_grad():
        cell.gates.W.copy_(s['gates_W']); cell.gates.b.copy_(s['gates_b'])
        cell.candi.W.copy_(s['candi_W']); cell.candi.b.copy_(s['candi_b'])
    Xt = s['Xt'].clone().requires_grad_()
    Ht = s['Ht'].clone().requires_grad_()
    out = cell(s['Gs'], s['Gc'], Xt, Ht)
    (out * s['R']).sum().backward()
    rows = sample_rows(s['N'], 128)
    _save('g8b_large_n10000_grads', rows=rows, Hout=out[:, rows], dXt=Xt.grad[:, rows], dHt=Ht.grad[:, rows],
          d_gates_W=cell.gates.W.grad, d_gates_b=cell.gates.b.grad, d_candi_W=cell.candi.W.grad, d_candi_b=cell.candi.b.grad,
          chk_Gs=s['Gs'].double().sum(), chk_Xt=s['Xt'].double().sum())


def large_k3_golden(ref_framework='/root/reference/framework'):
    """g8c: one reference STC_Cell at N = 10 000, C = 32, h = 16, K = 3 (dense Gs, matrix-side cheby_poly with its two N^3 products per
    BDG_Dif call, STC_GNN.py:24-29) forward + backward: 256 sampled rows of Ht, dXt, dHt and the full parameter gradients."""
    sys.path.insert(0, ref_framework)
    import STC_GNN as ref
    torch.set_num_threads(8)
    s = synth_inputs('g8c')
    cell = ref.STC_Cell(s['N'], s['C'], s['K'], s['K'], s['cin'], s['h'])
    with torch.no_grad():
        cell.gates.W.copy_(s['gates_W']); cell.gates.b.copy_(s['gates_b'])
        cell.candi.W.copy_(s['candi_W']); cell.candi.b.copy_(s['candi_b'])
    Xt = s['Xt'].clone().requires_grad_()
    Ht = s['Ht'].clone().requires_grad_()
    out = cell(s['Gs'], s['Gc'], Xt, Ht)
    (out * s['R']).sum().backward()
    rows = sample_rows(s['N'], 256, seed=98)
    _save('g8c_large_n10000_k3', rows=rows, Hout=out[:, rows], dXt=Xt.grad[:, rows], dHt=Ht.grad[:, rows],
          d_gates_W=cell.gates.W.grad, d_gates_b=cell.gates.b.grad, d_candi_W=cell.candi.W.grad, d_candi_b=cell.candi.b.grad,
          chk_Gs=s['Gs'].double().sum(), chk_Xt=s['Xt'].double().sum())


def pipeline_inputs():
    """Synthetic incident series + trainer params shared by make_golden and the tests (g9)."""
    g = torch.Generator().manual_seed(9)
    T, H, W, C = 30, 2, 3, 2
    inc = (torch.rand(T, H, W, C, generator=g) < 0.3).to(torch.int32).numpy()
    r, c = queen_grid_adjacency(H, W)
    s_adj = np.zeros((H * W, H * W), dtype=np.float64)
    s_adj[r.numpy(), c.numpy()] = 1.0
    c_cor = torch.rand(C, C, generator=g).double().numpy()
    data = dict(inc=inc, mask=[(0, 0)], HA=inc.reshape(T, -1, C).mean((0, 1)), s_adj=s_adj, c_cor=c_cor)
    params = dict(device='cpu', H=H, W=W, C=C, batch_size=4, obs_len=3, pred_len=2, split_ratio=[6, 1, 1],
                  model='STC-GNN', cheby_order=2, hidden_dim=4, nn_layers=1, learn_rate=2e-3, decay_rate=1e-4,
                  num_epochs=2, time_slice=4, city='SYN')
    return data, params


def pipeline_golden(ref_framework='/root/reference/framework'):
    import contextlib
    import io
    import re
    import tempfile
    sys.path.insert(0, ref_framework)
    import Data_Container as DC
    import Model_Trainer as MT
    data, params = pipeline_inputs()
    gen = DC.DataGenerator(obs_len=params['obs_len'], pred_len=params['pred_len'], data_split_ratio=params['split_ratio'])
    loaders = gen.get_data_loader(params=params, data=data)
    out = {}
    for mode in ('train', 'validate', 'test'):
        batches = list(loaders[mode])
        out[f'{mode}_len'] = len(loaders[mode].dataset)
        out[f'{mode}_batches'] = len(batches)
        out[f'{mode}_x0'], out[f'{mode}_y0'] = batches[0]
        out[f'{mode}_xlast'], out[f'{mode}_ylast'] = batches[-1]
    # the reference's train loop (Model_Trainer.py:61-92) on the reference's classes, exact epoch losses
    torch.manual_seed(123)
    with tempfile.TemporaryDirectory() as tmp:
        params = dict(params, output_dir=tmp)
        trainer = MT.ModelTrainer(params=params, data=data)
        sd0 = {k: v.clone() for k, v in trainer.model.state_dict().items()}
        curves = {'train': [], 'validate': []}
        for _ in range(params['num_epochs']):
            for mode in ('train', 'validate'):
                trainer.model.train(mode == 'train')
                tot, seen = 0.0, 0
                for x, y in loaders[mode]:
                    with torch.set_grad_enabled(mode == 'train'):
                        loss = trainer.criterion(trainer.model(X_seq=x, As=trainer.prior_graph[0], Ac=trainer.prior_graph[1]), y)
                        if mode == 'train':
                            trainer.optimizer.zero_grad(); loss.backward(); trainer.optimizer.step()
                    tot += float(loss.detach()) * y.shape[0]; seen += y.shape[0]
                curves[mode].append(tot / seen)
        # cross-check against the reference's own ModelTrainer.train printout (4 significant digits)
        torch.manual_seed(123)
        t2 = MT.ModelTrainer(params=params, data=data)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            t2.train(data_loader=loaders, modes=['train', 'validate'])
        printed = [float(v) for v in re.findall(r'training loss: ([0-9.]+)', buf.getvalue())]
        assert len(printed) == params['num_epochs'] and all(abs(a - b) < 2e-3 * b for a, b in zip(printed, curves['train'])), (printed, curves)
        ck = torch.load(os.path.join(tmp, 'STC-GNN-4.pkl'))
        out['ckpt_keys'] = np.array(sorted(ck.keys()))
        out['ckpt_epoch'] = ck['epoch']
    out['train_curve'] = np.array(curves['train'])
    out['val_curve'] = np.array(curves['validate'])
    out.update({'sd0/' + k: v for k, v in sd0.items()})
    _save('g9_pipeline', **out)


def metrics_inputs():
    rng = np.random.default_rng(10)
    S, hor, H, W, C = 40, 2, 4, 5, 3
    true = (rng.random((S, hor, H * W, C)) < 0.25).astype(np.float32)
    prob = np.clip(0.6 * true + 0.5 * rng.random((S, hor, H * W, C)), 0.001, 0.999).astype(np.float32)
    prob = np.round(prob, 2)                                   # plenty of ties for the AUC code
    thr = true.reshape(-1, C).mean(0)
    mask = [(0, 0), (3, 4), (2, 2)]
    return prob, true, thr, mask, H, W


def metrics_golden(ref_framework='/root/reference/framework'):
    import contextlib
    import io
    sys.path.insert(0, ref_framework)
    import Metrics as RM
    prob, true, thr, mask, H, W = metrics_inputs()
    pm, tm = RM.mask_data(prob, H, W, mask), RM.mask_data(true, H, W, mask)
    ev = RM.ModelEvaluator(dict(C=prob.shape[-1], pred_len=prob.shape[1]))
    out = dict(prob_masked=pm, true_masked=tm, unmasked=RM.mask_data(prob, H, W, None))
    with contextlib.redirect_stdout(io.StringIO()):
        for step in range(prob.shape[1]):
            m = ev.one_step_eval_bi(pm[:, step], tm[:, step], list(thr))
            out[f'names{step}'] = np.array(list(m.keys()))
            out[f'values{step}'] = np.array([float(v) for v in m.values()])
    _save('g10_metrics', **out)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'metrics':
        metrics_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == 'pipeline':
        pipeline_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == 'bench_path':
        bench_path_golden(only=sys.argv[2] if len(sys.argv) > 2 else None)
    elif len(sys.argv) > 1 and sys.argv[1] == 'large_k3':
        large_k3_golden()
    else:
        main()
        pipeline_golden()
        metrics_golden()
        bench_path_golden()
        large_k3_golden()
