#!/usr/bin/env python3
"""Probe: does the plain row-blocked SpMM get faster when 4 consecutive node ids form a 2 x 2 patch of the grid (3.96 distinct
neighbour rows fetched per output row instead of 4.46 for row-major ids)?  Same kernel, same bytes, different node numbering."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'stc-gnn_amd'))
import numpy as np, torch
from stc_hip import CsrGraph
from stc_hip._lib import HipKernels
hip = HipKernels(); dev = torch.device('cuda')
H = W = 224; N = H * W
base = CsrGraph.queen_grid(H, W)
idx = np.arange(N).reshape(H, W)
orders = {'row-major': None,
          '2x2 patches': idx.reshape(H // 2, 2, W).transpose(0, 2, 1).reshape(-1),
          '2x4 patches (8 ids)': idx.reshape(H // 2, 2, W // 4, 4).transpose(0, 2, 1, 3).reshape(-1),
          '4x4 patches (16 ids)': idx.reshape(H // 4, 4, W // 4, 4).transpose(0, 2, 1, 3).reshape(-1)}
for name, order in orders.items():
    g = base if order is None else base.permuted(order)
    d = g.on(dev)
    plan = (d['fwd_blk_ptr'], d['fwd_blk_cols'], d['fwd_blk_vals'])
    for B, F in ((1, 1024), (5, 512)):
        Xs = [torch.randn(B, N, F, device=dev) for _ in range(6)]; Ys = [torch.empty(B, N, F, device=dev) for _ in range(6)]
        run = lambda i: hip.csr_spmm(d['fwd_rowptr'], d['fwd_colidx'], d['fwd_val'], N, N, Xs[i % 6], None, Ys[i % 6], 1.0, 0.0, plan=plan)
        for i in range(6): run(i)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(60): run(i)
        e.record(); torch.cuda.synchronize()
        us = 1e3 * s.elapsed_time(e) / 60
        nb = g.nnz * 8 + 4 * (N + 1) + 2 * B * N * F * 4
        print(f'{name:22s} fetches/row {g.fetches_per_row[0]:.2f}  entries/row (padded) {d["fwd_blk_cols"].numel() / N:.2f}  B={B} F={F}: {us:7.1f} us  {nb / us / 1e3:7.1f} GB/s')
        del Xs, Ys
