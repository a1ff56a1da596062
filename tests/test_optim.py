"""``stc_hip.optim.Adam``: torch.optim.Adam with the large parameters on ``stc_adam_f32`` (the optimizer of the reference's harness step,
Model_Trainer.py:71-87).  CPU: nothing is large enough to leave torch -- identical to torch.optim.Adam.  GPU: the kernel against torch's
own update over several steps, state dict interchange, a captured HIP graph."""
import copy

import pytest
import torch

from stc_hip.optim import Adam


def _params(device, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(device)) for s in ((257, 64), (33,), (1000, 1000))]


def _grads(params, step, seed=100):
    g = torch.Generator().manual_seed(seed + step)
    for p in params:
        p.grad = (torch.randn(p.shape, generator=g) * (10.0 ** (step - 2))).to(p.device)


def test_on_cpu_tensors_it_is_torch_adam():
    ours, ref = _params('cpu'), _params('cpu')
    a, b = Adam(ours, lr=2e-3, weight_decay=1e-4), torch.optim.Adam(ref, lr=2e-3, weight_decay=1e-4)
    for step in range(4):
        _grads(ours, step)
        _grads(ref, step)
        a.step()
        b.step()
    assert all(torch.equal(x, y) for x, y in zip(ours, ref))


def test_step_hooks_fire_once():
    """torch wraps ``step`` of every optimizer class in its hook wrapper; this class's step calls the parent's UNWRAPPED implementation, so a
    registered step hook runs once per step even after a plain torch.optim.Adam was constructed in the process."""
    torch.optim.Adam(_params('cpu'), lr=1e-3)
    ours = _params('cpu')
    opt = Adam(ours, lr=2e-3)
    calls = {'pre': 0, 'post': 0}
    opt.register_step_pre_hook(lambda *a, **k: calls.__setitem__('pre', calls['pre'] + 1))
    opt.register_step_post_hook(lambda *a, **k: calls.__setitem__('post', calls['post'] + 1))
    for step in range(3):
        _grads(ours, step)
        opt.step()
    assert calls == {'pre': 3, 'post': 3}


@pytest.mark.gpu
@pytest.mark.parametrize('weight_decay', [0.0, 1e-4])
def test_large_parameters_follow_torch_adam_on_the_gpu(weight_decay, monkeypatch):
    monkeypatch.setattr(Adam, 'LARGE_BYTES', 1 << 20)                 # the (1000, 1000) matrix is "large"; the other two stay with torch
    ours, ref = _params('cuda'), _params('cuda')
    a = Adam(ours, lr=2e-3, weight_decay=weight_decay)
    b = torch.optim.Adam(ref, lr=2e-3, weight_decay=weight_decay)
    for step in range(6):                                             # gradient magnitudes 1e-2 .. 1e3
        _grads(ours, step)
        _grads(ref, step)
        a.step()
        b.step()
        for x, y in zip(ours, ref):
            assert float((x - y).abs().max()) <= 2e-6 * float(y.abs().max()), step
    st = a.state[ours[2]]
    assert st['step'].is_cuda and float(st['step']) == 6 and set(st) == {'step', 'exp_avg', 'exp_avg_sq'}
    for name in ('exp_avg', 'exp_avg_sq'):                            # (max-norm: an element that cancels to ~0 carries the rounding of its terms)
        want = b.state[ref[2]][name]
        assert float((st[name] - want).abs().max()) <= 2e-6 * float(want.abs().max()), name
    # a state dict written by torch's optimizer continues on ours
    c = Adam(_params('cuda'), lr=2e-3, weight_decay=weight_decay)
    c.load_state_dict(copy.deepcopy(b.state_dict()))                  # (load_state_dict keeps tensors it need not convert: without the copy c and b would share moments)
    for x, y in zip(c.param_groups[0]['params'], ref):
        x.data.copy_(y.data)
    _grads(c.param_groups[0]['params'], 6)
    _grads(ref, 6)
    c.step()
    b.step()
    for x, y in zip(c.param_groups[0]['params'], ref):
        assert float((x - y).abs().max()) <= 2e-6 * float(y.abs().max())


@pytest.mark.gpu
def test_the_update_replays_from_a_captured_graph(monkeypatch):
    """The step count of a large parameter is a device scalar: replays of one captured update advance the bias corrections."""
    monkeypatch.setattr(Adam, 'LARGE_BYTES', 1 << 20)
    ours, ref = _params('cuda'), _params('cuda')
    a = Adam(ours, lr=2e-3, weight_decay=1e-4, capturable=True)
    b = torch.optim.Adam(ref, lr=2e-3, weight_decay=1e-4)
    _grads(ours, 2)
    _grads(ref, 2)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        a.step()                                                      # (state created outside the capture)
    torch.cuda.current_stream().wait_stream(side)
    b.step()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        a.step()
    for _ in range(3):
        graph.replay()
        b.step()
    torch.cuda.synchronize()
    for x, y in zip(ours, ref):
        assert float((x - y).abs().max()) <= 4e-6 * float(y.abs().max())
