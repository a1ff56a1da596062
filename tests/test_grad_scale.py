"""Gradient operands of the fp16 x 2 format where they vary by many binades from node to node INSIDE one wave's sums (csrc/stc_x3_frag.h: RunScale).

Every backward kernel that runs the format keeps its dW / db sums in registers over the nodes a wave owns, at the wave's reference scale; a node more
than 2^12 above the reference ends the wave's pass (sums to the partial row, the node becomes the next pass's reference).  Rounds 3-4 zeroed the sums
there instead: node maxima 1, 64, 4096, 8192 inside one wave lost the 4096 node -- half of what was kept -- and no test reached it (the sweep's graphs
gave a wave one node, the full-size tests O(1) gradients).  Here: >= 8 nodes per wave (9 000 nodes over the <= 1 024 waves of a launch), state
gradients scaled per node by

    staircase    2^(3 (i mod 12))                       jumps of 2^12 that JOIN and of 2^24 that end a pass, in every wave
    ramp         1, 64, 4096, 8192 by position in wave   the verdict's example: the fourth node ends the pass that holds the third
    first-tiny   2^-40 for a wave's first two nodes      the reference starts 40 binades too low
    gates-only   staircase on dHnew with H = 0 and a      the candidate phase's maximum stays put, the gates phase's jumps: the one-launch
                 dominant constant dBm                    backward stops in the MIDDLE of a node (candidate sums kept, gates phase redone)
    zeros        staircase with every third node zero    a node without gradient must not move the reference

against the kernels' CPU twin in float64 (reference STC_GNN.py:65-79 through autograd): dW, db <= 2e-5 of their maximum, and every node's gradient
planes <= 2e-5 of THAT NODE's maximum (a launch-wide norm would hide the small nodes).  CPU: the same patterns through the twin with the operand
format emulated per wave (oracle/kernel_emul.py: _grad_scaled, GRAD_WAVES lowered so that waves own several nodes).
"""
import pytest
import torch

from oracle.kernel_emul import EmulatedKernels
from tests.conftest import rel_err

EM = EmulatedKernels()
BOUND = 2e-5
PATTERNS = ['staircase', 'ramp', 'first-tiny', 'gates-only', 'zeros']


def node_factors(pattern, nodes, waves):
    i = torch.arange(nodes)
    if pattern in ('staircase', 'gates-only'):
        e = 3 * (i % 12)
    elif pattern == 'ramp':
        e = torch.tensor([0, 6, 12, 13])[(i // waves) % 4]
    elif pattern == 'first-tiny':
        e = torch.where(i < 2 * waves, torch.full_like(i, -40), torch.zeros_like(i))
    elif pattern == 'zeros':
        f = torch.pow(2.0, (3 * (i % 12)).double())
        return torch.where(i % 3 == 1, torch.zeros_like(f), f)
    return torch.pow(2.0, e.double())


def per_node_err(a, b):
    """max over nodes of max |a - b| / max |b| within the node (nodes whose reference is all zero must be zero)."""
    a, b = a.detach().double().cpu().flatten(1), b.detach().double().cpu().flatten(1)
    d, m = (a - b).abs().amax(1), b.abs().amax(1)
    assert bool((d[m == 0] == 0).all())
    return float((d[m > 0] / m[m > 0]).max()) if bool((m > 0).any()) else 0.0


def _cell_case(pattern, nodes, cin, waves, seed=0, C=32):
    h, K = 16, 2
    Lw = cin + h
    g = torch.Generator().manual_seed(seed + nodes + cin)
    rnd = lambda *s_: torch.randn(*s_, generator=g)
    X, SX, H, SH = rnd(nodes, C, cin), rnd(nodes, C, cin), torch.tanh(rnd(nodes, C, h)), rnd(nodes, C, h)
    Tc = rnd(K, C, C) / C ** 0.5
    Tc[0] = torch.eye(C)
    Wg, Wc = rnd(K * K * Lw, 2 * h) / (4 * Lw) ** 0.5, rnd(K * K * Lw, h) / (4 * Lw) ** 0.5
    U, R, Cand = torch.sigmoid(rnd(nodes, C, h)), torch.sigmoid(rnd(nodes, C, h)), torch.tanh(rnd(nodes, C, h))
    f = node_factors(pattern, nodes, waves).float().view(-1, 1, 1)
    dHn, dBm = rnd(nodes, C, h) * f, rnd(nodes, C, h) * f
    if pattern == 'gates-only':
        H = torch.zeros_like(H)
        dBm = rnd(nodes, C, h) * 2.0 ** 36
    return dict(X=X, H=H, SX=SX, SH=SH, Tc=Tc, Wg=Wg, Wc=Wc, U=U, R=R, Cand=Cand, dHn=dHn, dBm=dBm, cin=cin, h=h, C=C, nodes=nodes)


def _cell_bwd(k, c, to, acc=None):
    """One stc_cell_bwd_planar call on kernel set ``k``; ``to`` maps the case's fp32 tensors to the set's device / dtype."""
    nodes, C, h, wide = c['nodes'], c['C'], c['h'], c['cin'] == c['h']
    names = ('X', 'H', 'SX', 'SH', 'Tc', 'Wg', 'Wc', 'U', 'R', 'Cand', 'dHn', 'dBm')
    ops_ = [to(c[n]) for n in names]
    like = ops_[1]
    dZ = [(to(acc[i]).clone() if acc is not None else torch.full((nodes, C, h), float('nan'), dtype=like.dtype, device=like.device)) if (wide or i >= 2) else None
          for i in range(4)]
    dWg, dWc = torch.empty_like(ops_[5]), torch.empty_like(ops_[6])
    dbg, dbc = like.new_empty(2 * h), like.new_empty(h)
    kw = {} if acc is None else dict(accumulate_x=wide, accumulate_h=True)
    k.cell_bwd_planar(*ops_, dZ, dWg, dbg, dWc, dbc, **kw)
    return dZ, dict(dWg=dWg, dbg=dbg, dWc=dWc, dbc=dbc)


def _compare(got, want, tag):
    (dZ, par), (dZ_w, par_w) = got, want
    for name in par_w:
        e = rel_err(par[name], par_w[name])
        assert e < BOUND, (tag, name, e)
    for i, (a, w) in enumerate(zip(dZ, dZ_w)):
        assert (a is None) == (w is None)
        if a is not None:
            assert torch.isfinite(a).all(), (tag, 'plane', i)
            e = per_node_err(a, w)
            assert e < BOUND, (tag, 'plane', i, e)


# ------------------------------------------------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize('pattern', PATTERNS)
@pytest.mark.parametrize('cin', [16, 1])
def test_emulated_format_with_several_nodes_per_wave(monkeypatch, pattern, cin):
    """CPU twin with the operand format emulated per WAVE (four waves over 48 nodes: twelve nodes share a wave's sums) against its float64 run."""
    em = EmulatedKernels(operand_format='f16x2')
    monkeypatch.setattr(em, 'GRAD_WAVES', 4, raising=False)
    c = _cell_case(pattern, 48, cin, waves=4)
    _compare(_cell_bwd(em, c, lambda t: t), _cell_bwd(EM, c, lambda t: t.double()), (pattern, cin))


def test_emulated_reference_follows_the_wave_not_the_launch():
    """The emulation's scales: node maxima 1, 64, 4096, 8192 in one wave -- the fourth node ends the pass (reference moves to it), the third joined
    with its activation operand scaled up by 2^8 and the rest inside its own scale; another wave with small gradients keeps its own reference."""
    em = EmulatedKernels(operand_format='f16x2')
    em.GRAD_WAVES = 2
    m = torch.tensor([1.0, 2.0 ** -30, 64.0, 2.0 ** -30, 4096.0, 2.0 ** -31, 8192.0, 2.0 ** -29])      # wave 0: nodes 0, 2, 4, 6; wave 1: the others
    grads = [(m.view(-1, 1, 1) * torch.ones(8, 2, 2)).float()]
    plane = torch.full((8, 2, 2), 1.0 + 2.0 ** -20)                                                       # needs 21 bits: visible once scaled below 2^-3
    (gq,), (pq,) = em._grad_scaled(grads, [plane], [1.0])
    assert torch.equal(gq, grads[0])                                                                      # (powers of two survive any scale)
    exact = lambda i: float(pq[i, 0, 0]) == float(plane[i, 0, 0])
    assert exact(0) and exact(2) and exact(4) and exact(6)        # joins at 2^0, 2^6, 2^8 (+ 2^4 in a_n), then a new pass: nothing shrinks
    assert exact(1) and exact(3)                                  # wave 1: reference 2^-30, equal node
    assert exact(7)                                               # 2^-29 above 2^-30: joins upward
    # a node 2^20 BELOW its wave's reference joins with its activation operand shrunk by as much: it loses low bits in proportion
    m2 = torch.tensor([1.0, 1.0, 2.0 ** -20, 1.0])
    (_,), (pq2,) = em._grad_scaled([(m2.view(-1, 1, 1) * torch.ones(4, 2, 2)).float()], [plane[:4]], [1.0])
    assert float(pq2[2, 0, 0]) != float(plane[2, 0, 0]) and float(pq2[0, 0, 0]) == float(plane[0, 0, 0])


# ------------------------------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope='module')
def hip():
    from stc_hip._lib import HipKernels
    return HipKernels()


def _f16(hip, monkeypatch):
    from stc_hip import _lib
    monkeypatch.setattr(hip, 'operand_format', _lib.FMT_F16X2, raising=False)


@pytest.mark.gpu
@pytest.mark.parametrize('pattern', PATTERNS)
@pytest.mark.parametrize('cin', [16, 1])
def test_one_launch_cell_backward_with_gradient_jumps_inside_a_wave(hip, monkeypatch, pattern, cin):
    """stc_cell_bwd_planar_f32 (C = 32): 9 000 nodes = 8-9 per wave; plain and accumulate forms."""
    _f16(hip, monkeypatch)
    c = _cell_case(pattern, 9000, cin, waves=1024)
    want = _cell_bwd(EM, c, lambda t: t.double())
    _compare(_cell_bwd(hip, c, lambda t: t.cuda()), want, (pattern, cin))
    g = torch.Generator().manual_seed(5)
    base = [torch.randn(9000, 32, 16, generator=g) for _ in range(4)]
    wide = cin == 16
    want_acc = ([None if w is None else w + base[i].double() * (1.0 if (i >= 2 or wide) else 0.0) for i, w in enumerate(want[0])], want[1])
    # (the planes' own content joins the per-node norm: compare what the launch ADDED)
    got = _cell_bwd(hip, c, lambda t: t.cuda(), acc=base)
    added = [None if a is None else a.double().cpu() - base[i].double() for i, a in enumerate(got[0])]
    for i, (a, w) in enumerate(zip(added, want[0])):
        if a is not None:
            scale = w.abs().flatten(1).amax(1).clamp(min=1.0)                   # fp32 planes holding O(1) values: absolute floor 1e-7 per entry
            assert float(((a - w).abs().flatten(1).amax(1) / scale).max()) < BOUND, (pattern, cin, 'accumulate', i)
    for name in want[1]:
        assert rel_err(got[1][name], want[1][name]) < BOUND, (pattern, cin, 'accumulate', name)


@pytest.mark.gpu
@pytest.mark.parametrize('pattern', ['staircase', 'ramp', 'first-tiny', 'zeros'])
@pytest.mark.parametrize('C', [32, 64])
def test_two_launch_backward_with_gradient_jumps_inside_a_wave(hip, monkeypatch, pattern, C):
    """stc_cell_gates_bwd_planar_f32 and stc_bdg_node_post_bwd_f32 (the C = 64 cells' backward; C = 32 when the one-launch form is off)."""
    _f16(hip, monkeypatch)
    nodes, h, K = 9000, 16, 2
    g = torch.Generator().manual_seed(C)
    rnd = lambda *s_: torch.randn(*s_, generator=g)
    X, H, SX, SH = rnd(nodes, C, h), torch.tanh(rnd(nodes, C, h)), rnd(nodes, C, h), rnd(nodes, C, h)
    Tc = rnd(K, C, C) / C ** 0.5
    Tc[0] = torch.eye(C)
    Wg, Wc = rnd(K * K * 2 * h, 2 * h) / (8 * h) ** 0.5, rnd(K * K * 2 * h, h) / (8 * h) ** 0.5
    U, R, Cand = torch.sigmoid(rnd(nodes, C, h)), torch.sigmoid(rnd(nodes, C, h)), torch.tanh(rnd(nodes, C, h))
    f = node_factors(pattern, nodes, 1024).float().view(-1, 1, 1)
    dRH, dHn, dA, dB = rnd(nodes, C, h) * f, rnd(nodes, C, h) * f, rnd(nodes, C, h) * f, rnd(nodes, C, h) * f

    def gates(k, to):
        like = to(X)
        dZ = [torch.full((nodes, C, h), float('nan'), dtype=like.dtype, device=like.device) for _ in range(4)]
        dW, db = torch.empty_like(to(Wg)), like.new_empty(2 * h)
        k.cell_gates_bwd_planar(*[to(t) for t in (X, H, SX, SH, Tc, Wg, dRH, Cand, U, R, dHn)], dZ, dW, db, None)
        return dZ, dict(dW=dW, db=db)

    def post(k, to):
        like = to(X)
        dX, dX2 = (torch.full((nodes, C, h), float('nan'), dtype=like.dtype, device=like.device) for _ in range(2))
        dW, db = torch.empty_like(to(Wc)), like.new_empty(h)
        k.node_post_bwd(to(X), to(Tc), to(Wc), to(dA), to(dB), dX, dW, db, X2=to(R * H), dX2=dX2)
        return [dX, dX2], dict(dW=dW, db=db)

    for fn in (gates, post):
        _compare(fn(hip, lambda t: t.cuda()), fn(EM, lambda t: t.double()), (pattern, C, fn.__name__))


@pytest.mark.gpu
@pytest.mark.parametrize('pattern', ['staircase', 'ramp', 'zeros'])
@pytest.mark.parametrize('cin', [16, 1])
def test_order3_backward_with_gradient_jumps_inside_a_wave(hip, monkeypatch, pattern, cin):
    """stc_cell_cand_bwd_planar_k_f32 and stc_cell_gates_bwd_planar_k_f32 (order 3, C = 32)."""
    _f16(hip, monkeypatch)
    nodes, h, K, C = 9000, 16, 3, 32
    Lw = cin + h
    g = torch.Generator().manual_seed(cin)
    rnd = lambda *s_: torch.randn(*s_, generator=g)
    Zx, Zh, Zr = [rnd(nodes, C, cin) for _ in range(K)], [rnd(nodes, C, h) for _ in range(K)], [rnd(nodes, C, h) for _ in range(K)]
    Tc = rnd(K, C, C) / C ** 0.5
    Tc[0] = torch.eye(C)
    Wg, Wc = rnd(K * K * Lw, 2 * h) / (K * K * Lw) ** 0.5, rnd(K * K * Lw, h) / (K * K * Lw) ** 0.5
    U, R, Cand = torch.sigmoid(rnd(nodes, C, h)), torch.sigmoid(rnd(nodes, C, h)), torch.tanh(rnd(nodes, C, h))
    f = node_factors(pattern, nodes, 1024).float().view(-1, 1, 1)
    dHn, dRH = rnd(nodes, C, h) * f, rnd(nodes, C, h) * f
    wide = cin == h

    def run(k, to, which):
        like = to(U)
        new = lambda w: torch.full((nodes, C, w), float('nan'), dtype=like.dtype, device=like.device)
        ls = lambda ts: [to(t) for t in ts]
        dXs = [new(cin) for _ in range(K)] if wide else [None] * K
        dHs = [new(h) for _ in range(K)]
        if which == 'cand':
            dW, db = torch.empty_like(to(Wc)), like.new_empty(h)
            k.cell_cand_bwd_planar_k(ls(Zx), ls(Zr), to(Tc), to(Wc), to(dHn), to(U), to(Cand), dXs, dHs, dW, db)
        else:
            dW, db = torch.empty_like(to(Wg)), like.new_empty(2 * h)
            k.cell_gates_bwd_planar_k(ls(Zx), ls(Zh), to(Tc), to(Wg), to(dRH), to(Cand), to(U), to(R), to(dHn), dXs, dHs, dW, db, None)
        return dHs + (dXs if wide else []), dict(dW=dW, db=db)

    for which in ('cand', 'gates'):
        _compare(run(hip, lambda t: t.cuda(), which), run(EM, lambda t: t.double(), which), (pattern, cin, which))
