"""Probe: relative error of the output head's parameter gradients at N = 50 176, C = 32 (1.6 M rows) against float64 -- torch CPU fp32,
this build's fused head kernel, torch GPU fp32.  The fp32 CPU sums are the least exact: full-size parity tests use a float64 oracle."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'stc-gnn_amd'))
import torch
from stc_hip import ops
torch.manual_seed(0)
dev = torch.device('cuda')
R, h = 50176 * 32, 16
H = torch.tanh(torch.randn(1, R, h))
lin1, lin2 = torch.nn.Linear(h, h // 2), torch.nn.Linear(h // 2, 1)
Y = (torch.rand(1, R) < 0.1635).float()
def run(dtype, device, use_kernel):
    l1 = torch.nn.Linear(h, h // 2).to(dtype).to(device); l2 = torch.nn.Linear(h // 2, 1).to(dtype).to(device)
    l1.load_state_dict({k: v.to(dtype) for k, v in lin1.state_dict().items()}); l2.load_state_dict({k: v.to(dtype) for k, v in lin2.state_dict().items()})
    Hd = H.to(dtype).to(device)
    if use_kernel:
        w = (l2.weight @ l1.weight).reshape(h); b = l2.weight @ l1.bias + l2.bias
        y = ops.head(Hd, w, b)
    else:
        y = torch.sigmoid(l2(l1(Hd))).squeeze(-1)
    Yd = Y.to(dtype).to(device)
    loss = torch.nn.functional.binary_cross_entropy(y, Yd) + (1 - 2 * (y * Yd).sum() / (y + Yd).sum())
    loss.backward()
    return {f'{i}.{n}': p.grad.detach().double().cpu() for i, l in enumerate((l1, l2)) for n, p in l.named_parameters()}
ref = run(torch.float64, 'cpu', False)
cpu32 = run(torch.float32, 'cpu', False)
gpu32 = run(torch.float32, dev, True)
gput = run(torch.float32, dev, False)
for i, k in enumerate(['l1.w', 'l1.b', 'l2.w', 'l2.b']):
    r = list(ref.values())[i]
    f = lambda d: float((list(d.values())[i] - r).abs().max() / r.abs().max())
    print(k, 'cpu fp32 %.2e   hip head kernel %.2e   torch gpu fp32 %.2e' % (f(cpu32), f(gpu32), f(gput)))
