#!/usr/bin/env python3
"""Probe: host time the Python side needs to ENQUEUE one train step of the metric shape (no synchronisation inside the step) against
the GPU time of the step -- how much headroom the launching thread has before the step would become launch-bound."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, 'stc-gnn_amd')):
    sys.path.insert(0, p)
import torch
import STC_GNN as M
from stc_hip import CsrGraph
from stc_hip import dist as sdist
from stc_hip.loss import ComboLoss
dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N, C = 224 * 224, 32
graph = CsrGraph.queen_grid(224, 224, device=dev)
torch.manual_seed(42)
model = M.STCGNN(N, C, 2, 2, 1, 16, 2, 6, graph_mode='csr-fixed').to(dev)
Gc = torch.softmax(torch.randn(C, C), -1).to(dev)
X = (torch.rand(B, 18, N, C) < 0.1635).float().to(dev)
Y = (torch.rand(B, 6, N, C) < 0.1635).float().to(dev)
crit = ComboLoss(); bucket = sdist.GradBucket(model.parameters()); opt = torch.optim.Adam(model.parameters(), lr=2e-3, weight_decay=1e-4)
def step():
    bucket.zero(); loss = crit(model(X_seq=X, As=graph, Ac=Gc), Y); loss.backward(); bucket.allreduce_mean(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
host, total = [], []
for _ in range(5):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append(1e3 * (t1 - t0)); total.append(1e3 * (t2 - t0))
print(f'batch {B}: host enqueue {sorted(host)[2]:.1f} ms per step, step {sorted(total)[2]:.1f} ms (median of 5)')
