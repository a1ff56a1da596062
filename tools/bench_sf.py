#!/usr/bin/env python3
"""SF-shape train step (config 3 of BASELINE.json: B=32, T=9+3, N=100, C=5, h=16, K=2, 2 layers) on one MI355X.
--mode dense-learned: the reference's full model incl. MGP_Gen/MixedFusion (2e8 parameters, Adam over 800 MB);
--mode csr-fixed: encoder/decoder/head only on the fixed 10x10 queen grid (22 033 parameters).
Prints ms/step and samples/s (reference on 8 CPU cores: 67 samples/s fwd+bwd, ~30 with Adam)."""
import argparse
import os
import sys
import time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'stc-gnn_amd')):
    sys.path.insert(0, p)
import torch
import STC_GNN as M
from stc_hip import CsrGraph
from stc_hip.loss import ComboLoss

ap = argparse.ArgumentParser()
ap.add_argument('--mode', default='csr-fixed')
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--graph', action='store_true', help='capture the whole train step in a HIP graph and replay it')
ap.add_argument('--fused-adam', action='store_true', help='torch.optim.Adam(fused=True): one multi-tensor kernel per step instead of ~10 passes')
ap.add_argument('--profile', action='store_true', help='print the kernels of 3 steps by GPU time (torch.profiler)')
a = ap.parse_args()
dev = torch.device('cuda')
torch.manual_seed(0)
B, T, N, C, h, K, layers, hor = 32, 9, 100, 5, 16, 2, 2, 3
model = M.STCGNN(N, C, K, K, 1, h, layers, hor, graph_mode=a.mode).to(dev)
X = (torch.rand(B, T, N, C, device=dev) < 0.1635).float()
Y = (torch.rand(B, hor, N, C, device=dev) < 0.1635).float()
if a.mode == 'csr-fixed':
    As = CsrGraph.queen_grid(10, 10, normalize=True, device=dev)
else:
    As = CsrGraph.queen_grid(10, 10, normalize=False).to_dense().to(dev)
Ac = torch.rand(C, C, device=dev)
crit = ComboLoss()
opt = torch.optim.Adam(model.parameters(), lr=2e-3, weight_decay=1e-4, capturable=a.graph, **({'fused': True} if a.fused_adam else {}))


def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(model(X_seq=X, As=As, Ac=Ac), Y)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
if a.graph:
    # static inputs X, Y; grads live in the graph's private pool (set_to_none=True recreates them at each replay)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = step()
    eager_step, step = step, (lambda: (graph.replay(), static_loss)[1])
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f'SF shape {a.mode}{" hipGraph" if a.graph else ""}{" fused-adam" if a.fused_adam else ""}: {1e3 * dt:.2f} ms/step, {B / dt:.1f} samples/s, loss {float(loss.detach()):.4f}, '
      f'{sum(p.numel() for p in model.parameters())} parameters', flush=True)
if a.profile:
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=22, max_name_column_width=60), flush=True)
