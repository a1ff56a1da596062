"""Non-finite states must come out as non-finite numbers, not as a crash: one small csr-fixed train step per kernel family with a NaN / an Inf / huge
pre-activations injected (python tools/probes/nan_robustness.py; prints one line per case)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'stc-gnn_amd')]
import STC_GNN as M          # noqa: E402
from oracle import stc_oracle as O          # noqa: E402
from stc_hip import CsrGraph          # noqa: E402

dev = torch.device('cuda', 0)
for C, K, tag in ((32, 2, 'c32'), (64, 2, 'c64'), (32, 3, 'c32k3'), (5, 2, 'small')):
    for what in ('nan', 'inf', 'huge'):
        torch.manual_seed(0)
        H, W, B, T, horizon = 12, 20, 2, 4, 2
        graph = CsrGraph.queen_grid(H, W, normalize=True)
        model = M.STCGNN(H * W, C, K, K, 1, 16, 2, horizon, graph_mode='csr-fixed').to(dev)
        X = (torch.rand(B, T, H * W, C) < 0.3).float()
        if what == 'nan':
            X[0, 1, 7, 3] = float('nan')
        elif what == 'inf':
            X[1, 0, 100, 0] = float('inf')
        else:
            X = X * 1e30
        Y = (torch.rand(B, horizon, H * W, C) < 0.3).float()
        out = model(X_seq=X.to(dev), As=graph, Ac=torch.softmax(torch.randn(C, C), -1).to(dev))
        loss = O.combo_loss(out.clamp(0, 1).nan_to_num(0.5), Y.to(dev))
        loss.backward()
        torch.cuda.synchronize()
        g = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
        print(f'{tag:6s} {what:5s} out finite {bool(torch.isfinite(out).all())}  grads finite {bool(torch.isfinite(g).all())}  ok', flush=True)
