#!/usr/bin/env python3
"""Probe: what the gather costs the plain row-blocked SpMM.  Same kernel, same output size, graphs with 1 (identity), 3 (i-1, i, i+1),
and the grid's 8 neighbours per row; bytes counted as graph + X once + Y once."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'stc-gnn_amd'))
import numpy as np, torch
from stc_hip import CsrGraph
from stc_hip._lib import HipKernels
hip = HipKernels(); dev = torch.device('cuda')
H = W = 224; N = H * W
i = np.arange(N)
def band(offs):
    r = np.concatenate([i[(i + o >= 0) & (i + o < N)] for o in offs]); c = np.concatenate([(i + o)[(i + o >= 0) & (i + o < N)] for o in offs])
    return CsrGraph(N, r, c, np.ones(r.size, np.float32))
graphs = {'identity (1/row)': band([0]), 'i-1,i,i+1 (3/row)': band([-1, 0, 1]), 'i-W,i,i+W (3/row, far)': band([-W, 0, W]), 'queen grid (8/row)': CsrGraph.queen_grid(H, W)}
for name, g in graphs.items():
    d = g.on(dev)
    plan = (d['fwd_blk_ptr'], d['fwd_blk_cols'], d['fwd_blk_vals'])
    for B, F in ((1, 1024), (5, 512)):
        Xs = [torch.randn(B, N, F, device=dev) for _ in range(6)]; Ys = [torch.empty(B, N, F, device=dev) for _ in range(6)]
        run = lambda k: hip.csr_spmm(d['fwd_rowptr'], d['fwd_colidx'], d['fwd_val'], N, N, Xs[k % 6], None, Ys[k % 6], 1.0, 0.0, plan=plan)
        for k in range(6): run(k)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for k in range(60): run(k)
        e.record(); torch.cuda.synchronize()
        us = 1e3 * s.elapsed_time(e) / 60
        nb = g.nnz * 8 + 4 * (N + 1) + 2 * B * N * F * 4
        print(f'{name:26s} entries/row (padded) {d["fwd_blk_cols"].numel() / N:5.2f}  B={B} F={F}: {us:7.1f} us  {nb / us / 1e3:7.1f} GB/s')
        del Xs, Ys
Xs = [torch.randn(5, N, 512, device=dev) for _ in range(6)]; Ys = [torch.empty_like(x) for x in Xs]
for k in range(6): Ys[k].copy_(Xs[k])
torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record()
for k in range(60): Ys[k % 6].copy_(Xs[k % 6])
e.record(); torch.cuda.synchronize(); us = 1e3 * s.elapsed_time(e) / 60
print(f'torch copy, same bytes (B=5 F=512): {us:7.1f} us  {2 * Xs[0].numel() * 4 / us / 1e3:7.1f} GB/s')
