#!/usr/bin/env python3
"""Launch the node kernels (and the SpMM) a few times at the headline shape: a target for rocprofv3 --pmc."""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'stc-gnn_amd')):
    sys.path.insert(0, p)
import torch
from stc_hip import CsrGraph
from stc_hip._lib import HipKernels

hip = HipKernels()
dev = torch.device('cuda')
N, C, L, K = 224 * 224, 32, 32, 2
iters = int(os.environ.get('ITERS', '5'))
Tc = torch.softmax(torch.randn(K, C, C, device=dev), -1)
Tc[0] = torch.eye(C, device=dev)
Zs = [torch.randn(N, C, L, device=dev) for _ in range(K)]
for Ho in (32, 16):
    W = torch.randn(K * K * L, Ho, device=dev) * 0.1
    b = torch.randn(Ho, device=dev)
    Y = torch.empty(N, C, Ho, device=dev)
    dZs = [torch.empty_like(z) for z in Zs]
    dW, db = torch.empty_like(W), torch.empty_like(b)
    for _ in range(iters):
        hip.bdg_node_fwd(Zs, Tc, W, b, Y)
    for _ in range(iters):
        hip.bdg_node_bwd(Zs, Tc, W, Y, dZs, dW, db, None)
g = CsrGraph.queen_grid(224, 224, device=dev).on(dev)
X = Zs[0].view(1, N, C * L)
Yo = torch.empty_like(X)
Xs = [torch.randn(1, N, C * L, device=dev) for _ in range(3)]          # rotate: nothing left in the Infinity Cache
plan = (g['fwd_blk_ptr'], g['fwd_blk_cols'], g['fwd_blk_vals'])
for i in range(iters):
    hip.csr_spmm(g['fwd_rowptr'], g['fwd_colidx'], g['fwd_val'], N, N, Xs[i % 3], None, Yo, 1.0, 0.0, plan=plan)
for i in range(iters):
    hip.csr_spmm(g['fwd_rowptr'], g['fwd_colidx'], g['fwd_val'], N, N, Xs[i % 3], None, Yo, 1.0, 0.0)
torch.cuda.synchronize()
print('done')
