"""Probe (MI355X): one planar cell step's backward as ONE launch (stc_cell_bwd_planar_f32) against the two launches it replaces
(stc_bdg_node_post_bwd_f32 + stc_cell_gates_bwd_planar_f32), at the bench's size: 5 samples x 50 176 nodes, C = 32, hidden 16.
Operands are rotated over several sets so that nothing is served from the 256 MiB Infinity Cache."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'stc-gnn_amd'))
from stc_hip import _lib  # noqa: E402


def main():
    k = _lib.HipKernels()
    dev = torch.device('cuda')
    nodes, C, h, K = 5 * 50176, 32, 16, 2
    for cin in (16, 1):
        Lw = cin + h
        sets = []
        for s in range(3):
            g = torch.Generator(device='cuda').manual_seed(s)
            rnd = lambda *sh: torch.randn(*sh, generator=g, device=dev)
            X, SX, H, SH = rnd(nodes, C, cin), rnd(nodes, C, cin), torch.tanh(rnd(nodes, C, h)), rnd(nodes, C, h)
            U, R, Cand = torch.sigmoid(rnd(nodes, C, h)), torch.sigmoid(rnd(nodes, C, h)), torch.tanh(rnd(nodes, C, h))
            dHn, dBm = rnd(nodes, C, h), rnd(nodes, C, h)
            dY, RH = dHn * U * (1 - Cand * Cand), R * H
            outs = [torch.empty(nodes, C, h, device=dev) for _ in range(6)]
            sets.append((X, SX, H, SH, U, R, Cand, dHn, dBm, dY, RH, outs))
        Tc = torch.randn(K, C, C, device=dev) / C ** 0.5
        Wg, Wc = torch.randn(4 * Lw, 2 * h, device=dev) / 8, torch.randn(4 * Lw, h, device=dev) / 8
        dWg, dWc, dbg, dbc = torch.empty_like(Wg), torch.empty_like(Wc), torch.empty(2 * h, device=dev), torch.empty(h, device=dev)
        wide = cin == h

        def fused(st):
            X, SX, H, SH, U, R, Cand, dHn, dBm, dY, RH, o = st
            dZ = [o[0], o[1], o[2], o[3]] if wide else [None, None, o[2], o[3]]
            k.cell_bwd_planar(X, H, SX, SH, Tc, Wg, Wc, U, R, Cand, dHn, dBm, dZ, dWg, dbg, dWc, dbc)

        def separate(st):
            X, SX, H, SH, U, R, Cand, dHn, dBm, dY, RH, o = st
            if wide:
                k.node_post_bwd(X, Tc, Wc, dY, dBm, o[4], dWc, dbc, X2=RH, dX2=o[5])
            else:
                k.node_post_bwd(RH, Tc, Wc, dY, dBm, o[5], dWc, dbc, X2=X)
            dZ = [o[0], o[1], o[2], o[3]] if wide else [None, None, o[2], o[3]]
            k.cell_gates_bwd_planar(X, H, SX, SH, Tc, Wg, o[5], Cand, U, R, dHn, dZ, dWg, dbg, None)

        label = 'one launch (' + ('one wave per SIMD' if os.environ.get('STC_CELL_BWD_PAIR') == '0' else 'C/G wave pairs') + ')'
        for name, fn in ((label, fused), ('two launches', separate)):
            for st in sets:
                fn(st)
            torch.cuda.synchronize()
            reps = 12
            t0 = time.perf_counter()
            for i in range(reps):
                fn(sets[i % len(sets)])
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / reps * 1e6
            planes = (13 if wide else 11) if name != 'two launches' else (19 if wide else 15)
            print(f'cin={cin:2d} {name:32s}: {us:8.1f} us per cell step   ({planes} planes of {nodes * C * h * 4 / 1e6:.0f} MB -> {planes * nodes * C * h * 4 / us / 1e6:.2f} TB/s)', flush=True)


if __name__ == '__main__':
    main()
