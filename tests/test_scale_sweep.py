"""Amplitude independence of the kernel families behind the drop-in model (reference STC_GNN.py:37-42, 72-78: torch.einsum / sigmoid / tanh are
scale-free -- a model whose inputs, parameters or graph are scaled by 10^+-k is as accurate as at amplitude 1).

The split-operand matrix-core kernels' default operand format (two fp16 pieces, csrc/stc_x3_frag.h) is not scale-free by itself: fp16 has an
absolute floor (2^-25 for a pair whose value is below 2^-3) and nothing above 65504.  Activations therefore carry powers of two as well -- one
per node in the forward, one per plane in the backward's dW products -- and the gate tanh switches to its Taylor polynomial below 1/4.  This
file sweeps what the goldens never did (they all have O(1) activations):

    X_seq x s, s in {1e-6, 1e-3, 1, 1e3, 1e5};  parameters x {1e-2, 1};  biases zero / 0.3 sigma;  Gs row sums x {1, 50} (and 4, 8, 16 at
    amplitude 1: the reference's raw 0/1 adjacency has row sums of 8)

through the four kernel families (C = 32 planar one-launch backward, C = 64 two-launch backward, C = 32 order 3, C = 5 few-category cells) and
the bf16 x 3 format, against the float64 oracle of the reference's encoder-decoder + ComboLoss:

    forward   max-norm relative error <= 1e-5 (north_star), or THREE times the reference's own fp32 noise on the prediction where that is larger
              (round 4 let 40x pass: 4.8e-5 with row sums of 50, 10x the noise).  Where the model amplifies rounding noise the max-norm error
              of ANY fp32-grade implementation is a draw of 1 - 2.2x the reference's noise: the few-category kernels (exact fp32 products)
              measure 1.9x at row sums of 16, 0.9x with libm gate functions and 2.1x with both gates refined (round 5 probes,
              profiles/r05/gate_function_probe.txt) -- a bound of 1x cannot be met by construction, 3x holds the line.  Graphs whose row
              sums exceed 24 leave the fp16 x 2 format (22-bit operands: 4 - 5x the noise) for bf16 x 3 by themselves (_lib.HEAVY_ROW_SUM)
    gradient  <= the bounds every other parity test uses (1e-5; 2e-5 for the long reductions)
    or (gradients), where that is larger, a multiple of the reference's OWN
    fp32 noise on that tensor -- the error of the oracle run in float32 (op for op
    the reference's arithmetic) against its float64 run.  That clause only matters where the model amplifies rounding noise: at X x 1e3 and
    above (pre-activations of 1e5 .. 5e6 carry an absolute fp32 rounding error of 1e-2 .. 0.3 in the reference itself, gates saturate and lose
    1 - U to cancellation; the reference's own gradients are off by up to 60 % against float64 at X x 1e5 with Gs x 50) and with graph row sums
    of 50 (the reference's noise: 5e-6 .. 1e-4).  The multiple: TEN, for both operand formats (round 6; rounds 4 - 5 gave fp16 x 2 forty).  bf16 x 3
    holds 24 significant bits, as fp32: one rounding pattern is one sample of a noise process.  fp16 x 2 holds 22 -- but since heavy graphs leave it
    by themselves (_lib.HEAVY_ROW_SUM, round 5) the regimes where its representation error showed (7 - 10x with products and sums exact, 10 - 25x
    measured in round 4) no longer run on it: what round 5's GPU log holds for the schedules that DO run on fp16 x 2 is at most 2.8x (C = 32), 1.7x
    (C = 64), 2.6x (few categories), 1.5x (order 3, row sums up to 16) -- profiles/r05/parity_errors.tsv.  One clause is wider, and it is not about a
    format: ORDER 3 on a graph with row sums of 50 (routed to bf16 x 3 like every heavy graph) gets 40x.  T_2(S) = 2 S^2 - I has norm ~5 000 there; the
    reference forms it on the MATRIX side and multiplies once, the kernels run the recurrence 2 S (S X) - X on the features and its transpose in
    Clenshaw form -- two roundings of a 5 000x larger intermediate against one, in a regime where the reference's own noise is 1e-4 .. 1e-2: measured
    16 - 18x on two bias / weight gradients at X x 1e-6 and 1e-3, <= 9.4x elsewhere.  Where that matters, STC_OPERAND_FORMAT=bf16x3 is the format to run.  A tensor on which the reference's own
    noise exceeds 10 % is not compared at all (logged as void): there is no parity to establish where float32 itself has no digits left
    (order 3 with Gs x 50 at X x 1e5: T_2(S) has norm ~5 000, pre-activations reach 5e8).

Both a max-norm and a relative-L2 error per tensor go to gpurun_out/parity_errors.txt.  CPU (always runs): the same sweep through the kernels'
CPU twin with the operand format EMULATED (oracle/kernel_emul.py, operand_format='f16x2'), plus the negative control: with the activation
scales switched off -- the round-3 kernels -- the sweep fails at amplitude 1e-3, i.e. these tests would have caught the floor.
"""
import os

import pytest
import torch

import STC_GNN as M
from oracle import stc_oracle as O
from oracle.kernel_emul import EmulatedKernels
from stc_hip import CsrGraph, ops
from tests.conftest import REPO, rel_err
from tests.golden.make_golden import SF_SHAPE, bench_path_inputs

FAMILIES = {'c32': (32, 2, {}), 'c64': (64, 2, {}), 'c32k3': (32, 3, {}), 'sf': (5, 2, SF_SHAPE)}
X_SCALES = [1e-6, 1e-3, 1.0, 1e3, 1e5]
# (parameter scale, bias sigma, Gs row-sum factor)
SETTINGS = {'plain': (1.0, 0.3, 1.0), 'small-params-zero-bias': (1e-2, 0.0, 1.0), 'heavy-graph-zero-bias': (1.0, 0.0, 50.0), 'small-params-heavy-graph': (1e-2, 0.3, 50.0),
            'graph-x4': (1.0, 0.0, 4.0), 'graph-x8': (1.0, 0.0, 8.0), 'graph-x16': (1.0, 0.0, 16.0)}
SWEPT = [(x, st) for st in ('plain', 'small-params-zero-bias', 'heavy-graph-zero-bias', 'small-params-heavy-graph') for x in X_SCALES] + \
        [(1.0, st) for st in ('graph-x4', 'graph-x8', 'graph-x16')]
FWD_BOUND, GRAD_BOUND = 1e-5, 2e-5
FWD_NOISE_FACTOR = 3.0            # the prediction: what an independent fp32 implementation draws (module docstring)


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    d = float(b.norm())
    return float((a - b).norm()) / (d if d > 0 else 1.0)


def _case(family, x_scale, setting, seed=0):
    C, K, shape = FAMILIES[family]
    p_scale, bias_sigma, gs_factor = SETTINGS[setting]
    s = bench_path_inputs(C, K, **shape)
    g = torch.Generator().manual_seed(4242 + seed)
    torch.manual_seed(977 + seed)                                     # (the module's own initialisers draw from the global generator)
    model = M.STCGNN(num_nodes=s['N'], num_categories=C, Ks=K, Kc=K, input_dim=1, hidden_dim=s['h'], num_layers=s['layers'], out_horizon=s['horizon'],
                     graph_mode='csr-fixed')
    sd = {}
    for k, v in model.state_dict().items():
        if k.endswith('.b') or k.endswith('.bias'):
            sd[k] = bias_sigma * torch.randn(v.shape, generator=g)
        else:
            sd[k] = p_scale * v.clone()
    model.load_state_dict(sd)
    return model, sd, s, s['X'] * x_scale, s['Gs'] * gs_factor, (C, K)


def _oracle(sd, s, X, Gs, K, dtype):
    leaves = {k: v.clone().to(dtype).requires_grad_() for k, v in sd.items()}
    yhat = O.encdec_forward(X.to(dtype), Gs.to(dtype), s['Gc'].to(dtype), leaves, K, K, s['h'], s['layers'], s['horizon'])
    O.combo_loss(yhat, s['Y'].to(dtype)).backward()
    return yhat.detach(), {k: v.grad for k, v in leaves.items()}


def _run(model, s, X, Gs, dev):
    model = model.to(dev)
    graph = CsrGraph.from_dense(Gs)
    yhat = model(X_seq=X.to(dev), As=graph, Ac=s['Gc'].to(dev))
    O.combo_loss(yhat, s['Y'].to(dev)).backward()
    return yhat.detach(), {k: p.grad for k, p in model.named_parameters()}


def _check(tag, got, want64, want32, dev, fwd_bound=FWD_BOUND, grad_bound=GRAD_BOUND, noise_factor=10.0):
    lines, worst = [], []
    (y, grads), (y64, g64), (y32, g32) = got, want64, want32
    tensors = [('yhat', y, y64, y32, fwd_bound, False)] + [('d' + k, grads[k], g64[k], g32[k], grad_bound, True) for k in g64]
    for name, a, b, b32, bound, is_grad in tensors:
        e_max, e_l2 = rel_err(a, b), rel_l2(a, b)
        noise = max(rel_err(b32, b), rel_l2(b32, b))                             # the reference's own fp32 noise on this tensor
        void = noise > 0.1
        lines.append(f'{tag}\t{name}\t{e_max:.3e}\tl2={e_l2:.3e}\tref_fp32_noise={noise:.1e}{" (void)" if void else ""}\n')
        allowed = max(bound, (noise_factor if is_grad else FWD_NOISE_FACTOR) * noise)
        if not torch.isfinite(a).all() or (not void and max(e_max, e_l2) >= allowed):
            worst.append((name, e_max, e_l2, noise))
    if dev == 'cuda':
        out = os.path.join(REPO, 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_errors.txt'), 'a') as f:
            f.writelines(lines)
    return worst


@pytest.mark.gpu
@pytest.mark.parametrize('x_scale,setting', SWEPT)
@pytest.mark.parametrize('family,fmt', [('c32', 'f16x2'), ('c32', 'bf16x3'), ('c64', 'f16x2'), ('c32k3', 'f16x2'), ('sf', 'f16x2')])
def test_scale_sweep_on_the_gpu(monkeypatch, family, fmt, x_scale, setting):
    from stc_hip import _lib
    monkeypatch.setattr(ops, '_kernels', None)
    k = ops.kernels()
    monkeypatch.setattr(k, 'operand_format', {'f16x2': _lib.FMT_F16X2, 'bf16x3': _lib.FMT_BF16X3}[fmt], raising=False)
    model, sd, s, X, Gs, (C, K) = _case(family, x_scale, setting)
    small_calls = []
    real_small = ops.stc_small_graph
    monkeypatch.setattr(ops, 'stc_small_graph', lambda *a, **kw: (small_calls.append(1), real_small(*a, **kw))[1])
    got = _run(model, s, X, Gs, 'cuda')
    assert bool(small_calls) == (family == 'sf')                      # the few-category kernels take the SF shape, the planar ones the rest
    bad = _check(f'scale_sweep[{family}-{fmt}-x{x_scale:g}-{setting}]', got, _oracle(sd, s, X, Gs, K, torch.float64), _oracle(sd, s, X, Gs, K, torch.float32), 'cuda',
                 # 10x the reference's own fp32 noise for BOTH formats; order 3 on a graph with row sums of 50 (which runs on bf16 x 3: heavy graphs
                 # leave fp16 x 2): 40x -- the feature-side recurrence against the reference's matrix-side T_2(S) of norm 5 000, module docstring
                 noise_factor=40.0 if (family == 'c32k3' and 'heavy' in setting) else 10.0)
    assert not bad, bad


@pytest.mark.parametrize('x_scale,setting', [(1e-6, 'small-params-zero-bias'), (1e-3, 'plain'), (1.0, 'plain'), (1e5, 'heavy-graph-zero-bias'),
                                             (1.0, 'heavy-graph-zero-bias'), (1.0, 'graph-x16')])
@pytest.mark.parametrize('family', ['c32', 'c32k3'])
def test_scale_sweep_through_the_emulated_format(monkeypatch, family, x_scale, setting):
    """CPU: host logic + the fp16 x 2 operand representation (scales as the kernels take them) against the float64 oracle."""
    monkeypatch.setattr(ops, '_kernels', EmulatedKernels(operand_format='f16x2'))
    model, sd, s, X, Gs, (C, K) = _case(family, x_scale, setting)
    bad = _check('emulated', _run(model, s, X, Gs, 'cpu'), _oracle(sd, s, X, Gs, K, torch.float64), _oracle(sd, s, X, Gs, K, torch.float32), 'cpu')
    assert not bad, bad


def test_the_sweep_catches_unscaled_activations(monkeypatch):
    """Negative control: activations fed to the fp16 x 2 format unscaled (round 3's kernels) break the bounds at amplitude 1e-3 -- the sweep
    sees the format's absolute floor (3.3e-5 on a 32-term dot product at that amplitude)."""
    monkeypatch.setattr(ops, '_kernels', EmulatedKernels(operand_format='f16x2', act_scales=False))
    model, sd, s, X, Gs, (C, K) = _case('c32', 1e-3, 'small-params-zero-bias')
    bad = _check('emulated-unscaled', _run(model, s, X, Gs, 'cpu'), _oracle(sd, s, X, Gs, K, torch.float64), _oracle(sd, s, X, Gs, K, torch.float32), 'cpu')
    assert bad, 'unscaled activations should have failed the sweep'


def test_heavy_graphs_leave_the_fp16x2_format(monkeypatch):
    """A graph whose row sums exceed HEAVY_ROW_SUM is run on the 24-bit operand format even when fp16 x 2 is the default: with row sums of 50 the
    fp16 x 2 format emulated on CPU puts the prediction at 7x the reference's noise (beyond 1e-5); routed, the schedule runs on the exact twin."""
    em = EmulatedKernels(operand_format='f16x2')
    monkeypatch.setattr(ops, '_kernels', em)
    model, sd, s, X, Gs, (C, K) = _case('c32', 1.0, 'heavy-graph-zero-bias')
    graph = CsrGraph.from_dense(Gs)
    assert graph.row_sum_bound > em.HEAVY_ROW_SUM and em.for_graph(graph.row_sum_bound) is not em
    assert CsrGraph.from_dense(_case('c32', 1.0, 'graph-x8')[4]).row_sum_bound <= em.HEAVY_ROW_SUM
    want64, want32 = _oracle(sd, s, X, Gs, K, torch.float64), _oracle(sd, s, X, Gs, K, torch.float32)
    bad = _check('emulated-routed', _run(model, s, X, Gs, 'cpu'), want64, want32, 'cpu')
    assert not bad, bad
    monkeypatch.setattr(em, 'HEAVY_ROW_SUM', 1e9, raising=False)      # negative control: the same graph kept on the emulated fp16 x 2 format
    model, sd, s, X, Gs, (C, K) = _case('c32', 1.0, 'heavy-graph-zero-bias')
    bad = _check('emulated-unrouted', _run(model, s, X, Gs, 'cpu'), want64, want32, 'cpu')
    assert any(name == 'yhat' for name, *_ in bad), bad
