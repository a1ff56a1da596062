"""Configuration 4's aggregation launches (100 x 100 grid, 4 samples, rows of 512 floats: 28 - 42 us each) are said to be tail-bound: would two
independent ones (the X-side and the H-side recurrence of an order-3 cell) overlap?  Pairs of launches sequentially on one stream against on two
streams with event waits.  Measured: 58.6 against 69.8 us per pair (plain), 84.3 against 95.8 (Y0 form) -- no overlap to be had, the fork / join
costs more than the tails.  python tools/probes/spmm_two_streams.py"""
import os, sys, torch
sys.path.insert(0, '/root/repo/stc-gnn_amd')
from stc_hip import CsrGraph
from stc_hip._lib import HipKernels
hip = HipKernels()
g = CsrGraph.queen_grid(100, 100); d = g.on(torch.device('cuda')); n = g.n; B, F = 4, 512
plan = (d['fwd_blk_ptr'], d['fwd_blk_cols'], d['fwd_blk_vals'], tuple(d[f'fwd_pt_{k}'] for k in ('src', 'rows', 'cnt', 'idx', 'val')))
Xs = [torch.randn(B, n, F, device='cuda') for _ in range(4)]; Ys = [torch.empty(B, n, F, device='cuda') for _ in range(4)]
s2 = torch.cuda.Stream()
def one(i, y0):
    hip.csr_spmm(d['fwd_rowptr'], d['fwd_colidx'], d['fwd_val'], n, n, Xs[i], Ys[i] if y0 else None, Ys[i], 2.0 if y0 else 1.0, -1.0 if y0 else 0.0, plan=plan)
for y0 in (False, True):
    for mode in ('sequential', 'two streams'):
        def pair():
            if mode == 'sequential':
                one(0, y0); one(1, y0)
            else:
                s2.wait_stream(torch.cuda.current_stream())
                one(0, y0)
                with torch.cuda.stream(s2):
                    one(1, y0)
                torch.cuda.current_stream().wait_stream(s2)
        for _ in range(5): pair()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): pair()
        e1.record(); torch.cuda.synchronize()
        print(f'{"Y0 form" if y0 else "plain  "} {mode:12s} {1e3 * e0.elapsed_time(e1) / 50:7.1f} us per pair')
