"""The front end of the learned graph generator (reference STC_GNN.py:229-232 spatial branch, :237-240 category branch) on the HIP kernels
``stc_mgp_uv_fwd/bwd_f32`` + ``stc_mgp_softmax_fwd/bwd_f32`` (csrc/stc_mgp_front.hip): forward and parameter gradients against the reference's
own sequence of torch operations in float64; the module's two paths (kernels / torch operations) against each other; the launch count."""
import pytest
import torch

import STC_GNN as M
from stc_hip import ops
from tests.conftest import rel_err


def _reference(X, Wu, Wv, alpha, transpose):
    """MGP_Gen.forward's branch, op for op (STC_GNN.py:229-232 / :236-240)."""
    if transpose:
        X = X.transpose(2, 3)
    U = torch.tanh(alpha * torch.matmul(X, Wu))
    V = torch.tanh(alpha * torch.matmul(X, Wv))
    P = torch.einsum('btnh,btmh->nm', U, V) - torch.einsum('btmh,btnh->mn', V, U)
    return torch.softmax(torch.relu(P), dim=-1)


@pytest.mark.gpu
@pytest.mark.parametrize('B,T,N,C,h', [(32, 9, 100, 5, 16), (2, 3, 7, 3, 4), (1, 1, 70, 13, 8), (3, 2, 1, 1, 16)])
@pytest.mark.parametrize('rows_axis', [2, 3])
def test_mgp_front_matches_the_reference_ops(B, T, N, C, h, rows_axis):
    g = torch.Generator().manual_seed(B * 1000 + N)
    X = (torch.rand(B, T, N, C, generator=g) < 0.3).float()
    F = C if rows_axis == 2 else N
    Wu, Wv = (torch.randn(F, h, generator=g) * (2.0 / (F + h)) ** 0.5 for _ in range(2))
    R = N if rows_axis == 2 else C
    W = torch.randn(R, R, generator=g)
    a64, b64 = Wu.double().requires_grad_(), Wv.double().requires_grad_()
    want = _reference(X.double(), a64, b64, 3, rows_axis == 3)
    (want * W.double()).sum().backward()
    # the reference's own float32 run: P sums B T h = 4 608 products at the SF shape and the softmax exponentiates DIFFERENCES of such sums, so the
    # float32 result itself sits ~1e-5 from float64 there; the bound is 1e-5 / 2e-5 or three times that noise, as in tests/test_scale_sweep.py
    a32, b32 = Wu.clone().requires_grad_(), Wv.clone().requires_grad_()
    want32 = _reference(X, a32, b32, 3, rows_axis == 3)
    (want32 * W).sum().backward()
    fwd_bound = max(1e-5, 3 * rel_err(want32, want))
    grad_bound = max(2e-5, 3 * max(rel_err(a32.grad, a64.grad), rel_err(b32.grad, b64.grad)))
    a, b = Wu.cuda().requires_grad_(), Wv.cuda().requires_grad_()
    assert ops.mgp_front_supported(X.cuda(), a, b)
    got = ops.mgp_front(X.cuda(), a, b, rows_axis, 3.0)
    (got * W.cuda()).sum().backward()
    assert got.shape == (R, R) and rel_err(got, want) < fwd_bound, (rel_err(got, want), fwd_bound)
    assert rel_err(a.grad, a64.grad) < grad_bound and rel_err(b.grad, b64.grad) < grad_bound, (rel_err(a.grad, a64.grad), rel_err(b.grad, b64.grad), grad_bound)


@pytest.mark.gpu
def test_generator_takes_the_kernels_and_equals_its_torch_path(monkeypatch):
    """MGP_Gen on the GPU: the fused front end is what runs (asserted), and Gs, Gc and every parameter gradient equal the torch-operation path."""
    torch.manual_seed(5)
    gen = M.MGP_Gen(num_nodes=12, num_categories=4, hidden_dim=8).cuda()
    g = torch.Generator().manual_seed(1)
    X = (torch.rand(3, 4, 12, 4, generator=g) < 0.3).float().cuda()
    As, Ac = torch.rand(12, 12, generator=g).cuda(), torch.rand(4, 4, generator=g).cuda()
    Ws, Wc = torch.randn(12, 12, generator=g).cuda(), torch.randn(4, 4, generator=g).cuda()
    calls = []
    real = ops.mgp_front
    monkeypatch.setattr(ops, 'mgp_front', lambda *a, **k: (calls.append(a[3]), real(*a, **k))[1])

    def run():
        gen.zero_grad(set_to_none=True)
        Gs, Gc = gen(X, As, Ac)
        ((Gs * Ws).sum() + (Gc * Wc).sum()).backward()
        return Gs.detach(), Gc.detach(), {k: p.grad.clone() for k, p in gen.named_parameters()}

    Gs, Gc, grads = run()
    assert calls == [2, 3]
    monkeypatch.setattr(ops, 'mgp_front_supported', lambda *a: False)
    Gs_t, Gc_t, grads_t = run()
    assert rel_err(Gs, Gs_t) < 1e-5 and rel_err(Gc, Gc_t) < 1e-5
    for k in grads:
        assert rel_err(grads[k], grads_t[k]) < 2e-5, k


def test_front_end_argument_validation_needs_no_gpu():
    from stc_hip import _lib
    lib = _lib.load_library()
    assert lib.stc_mgp_uv_fwd_f32(None, 0, 0, 0, None, None, 3.0, None, None, 0, 5, 3, 4, None) == 0          # no slices: no launch
    assert lib.stc_mgp_uv_fwd_f32(None, 0, 0, 0, None, None, 3.0, None, None, 2, 5, 3, 4, None) == -1 and b'null' in lib.stc_last_error()
    assert lib.stc_mgp_uv_fwd_f32(None, 0, 0, 0, None, None, 3.0, None, None, 2, 5, 0, 4, None) == -1
    assert lib.stc_mgp_softmax_fwd_f32(None, None, 0, None) == 0 and lib.stc_mgp_softmax_fwd_f32(None, None, 3, None) == -1
    assert lib.stc_mgp_softmax_bwd_f32(None, None, None, None, None, -1, None) == -1
