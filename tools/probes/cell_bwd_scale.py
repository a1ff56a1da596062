"""Probe (MI355X): stc_cell_bwd_planar_f32 in the fp16 x 2 operand format against the CPU twin with gradient operands of different magnitudes."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'stc-gnn_amd'))
import torch
from stc_hip._lib import HipKernels
from oracle.kernel_emul import EmulatedKernels
hip, EM = HipKernels(), EmulatedKernels()
cu = lambda t: t.cuda().contiguous()
rel = lambda a, b: float((a.cpu().double() - b.double()).abs().max() / b.double().abs().max())
C, h, K = 32, 16, 2
for nodes, cin in ((50, 16), (50, 1), (4500, 16)):
    for gscale in (1.0, 1e-4, 1e-8, 1e4):
        Lw = cin + h
        g = torch.Generator().manual_seed(nodes + cin)
        rnd = lambda *s_: torch.randn(*s_, generator=g)
        X, SX, H, SH = rnd(nodes, C, cin), rnd(nodes, C, cin), torch.tanh(rnd(nodes, C, h)), rnd(nodes, C, h)
        Tc = rnd(K, C, C) / C ** 0.5
        Tc[0] = torch.eye(C)
        Wg, Wc = rnd(K * K * Lw, 2 * h) / (4 * Lw) ** 0.5, rnd(K * K * Lw, h) / (4 * Lw) ** 0.5
        U, R, Cand = torch.sigmoid(rnd(nodes, C, h)), torch.sigmoid(rnd(nodes, C, h)), torch.tanh(rnd(nodes, C, h))
        dHn, dBm = rnd(nodes, C, h) * gscale, rnd(nodes, C, h) * gscale
        wide = cin == h
        new = lambda dev: [torch.full((nodes, C, h), float('nan'), device=dev) if (wide or i >= 2) else None for i in range(4)]
        dZ_w, dWg_w, dWc_w, dbg_w, dbc_w = new('cpu'), torch.empty_like(Wg), torch.empty_like(Wc), torch.empty(2 * h), torch.empty(h)
        EM.cell_bwd_planar(X, H, SX, SH, Tc, Wg, Wc, U, R, Cand, dHn, dBm, dZ_w, dWg_w, dbg_w, dWc_w, dbc_w)
        nan = lambda *s_: torch.full(s_, float('nan')).cuda()
        dZ, dWg, dWc, dbg, dbc = new('cuda'), nan(*Wg.shape), nan(*Wc.shape), nan(2 * h), nan(h)
        hip.cell_bwd_planar(*[cu(t) for t in (X, H, SX, SH, Tc, Wg, Wc, U, R, Cand, dHn, dBm)], dZ, dWg, dbg, dWc, dbc)
        torch.cuda.synchronize()
        errs = ['%.1e' % rel(a_, w_) for a_, w_ in zip(dZ, dZ_w) if a_ is not None]
        print(f'nodes {nodes} cin {cin} gradient scale {gscale:g}: dZ {errs} dWg {rel(dWg, dWg_w):.1e} dWc {rel(dWc, dWc_w):.1e} dbg {rel(dbg, dbg_w):.1e} dbc {rel(dbc, dbc_w):.1e}')
