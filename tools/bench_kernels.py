#!/usr/bin/env python3
"""Per-kernel timings at the headline shape (N=50176, C=32, h=16, K=2) through the C ABI.
Prints one line per kernel: mean us over --iters launches (HIP events on the launch stream),
algorithmic bytes and GB/s.  Used for A/B work between builds; not the contract benchmark."""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'stc-gnn_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from stc_hip import CsrGraph  # noqa: E402
from stc_hip._lib import HipKernels  # noqa: E402


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--grid', type=int, default=224)
    ap.add_argument('--C', type=int, default=32)
    ap.add_argument('--h', type=int, default=16)
    ap.add_argument('--K', type=int, default=2)
    ap.add_argument('--B', type=int, default=1)
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--permute', action='store_true')
    ap.add_argument('--only', default='', help="'spmm' = just the SpMM lines")
    a = ap.parse_args()
    hip = HipKernels()
    dev = torch.device('cuda')
    N, C, h, K, B = a.grid ** 2, a.C, a.h, a.K, a.B
    g = CsrGraph.queen_grid(a.grid, a.grid, permute_seed=1234 if a.permute else None, device=dev).on(dev)
    nnz = g['fwd_colidx'].numel()
    print(f'# N={N} nnz={nnz} C={C} h={h} K={K} B={B} permute={a.permute} MFMA={"off" if os.environ.get("STC_DISABLE_MFMA") == "1" else "on"} '
          f'SPMM_VARIANT={os.environ.get("STC_SPMM_VARIANT", "default")}')

    def report(name, us, nbytes):
        print(f'{name:34s} {us:9.1f} us   {nbytes/1e6:9.1f} MB   {nbytes/us/1e3:8.1f} GB/s', flush=True)

    for L in (32, 20):
        F = C * L
        X = torch.randn(B, N, F, device=dev)
        Y = torch.empty_like(X)
        us = timeit(lambda: hip.csr_spmm(g['fwd_rowptr'], g['fwd_colidx'], g['fwd_val'], N, N, X, None, Y, 1.0, 0.0), a.iters)
        report(f'spmm fwd F={F}', us, nnz * 8 + 4 * (N + 1) + 2 * B * N * F * 4)
        us = timeit(lambda: hip.csr_spmm(g['bwd_rowptr'], g['bwd_colidx'], g['bwd_val'], N, N, X, Y, Y, 1.0, 1.0), a.iters)
        report(f'spmm bwd F={F} (+=, in place)', us, nnz * 8 + 4 * (N + 1) + 3 * B * N * F * 4)
        if a.only == 'spmm':
            continue
        Tc = torch.softmax(torch.randn(K, C, C, device=dev), -1)
        Tc[0] = torch.eye(C, device=dev)
        for Ho in (2 * h, h):
            Zs = [torch.randn(B * N, C, L, device=dev) for _ in range(K)]
            W = torch.randn(K * K * L, Ho, device=dev) * 0.1
            b = torch.randn(Ho, device=dev)
            Yn = torch.empty(B * N, C, Ho, device=dev)
            us = timeit(lambda: hip.bdg_node_fwd(Zs, Tc, W, b, Yn), a.iters)
            report(f'node fwd L={L} Ho={Ho}', us, (K * L + Ho) * B * N * C * 4)
            dY = torch.randn_like(Yn)
            dZs = [torch.empty_like(z) for z in Zs]
            dW, db = torch.empty_like(W), torch.empty_like(b)
            us = timeit(lambda: hip.bdg_node_bwd(Zs, Tc, W, dY, dZs, dW, db, None), a.iters)
            report(f'node bwd L={L} Ho={Ho}', us, (2 * K * L + Ho) * B * N * C * 4)
    if a.only == 'spmm':
        return
    rows = (B, N, C)
    G = torch.randn(*rows, 2 * h, device=dev)
    Xt = torch.randn(*rows, h, device=dev)
    H = torch.randn(*rows, h, device=dev)
    U, R, Ci = torch.empty_like(H), torch.empty_like(H), torch.empty(*rows, 2 * h, device=dev)
    n = B * N * C
    us = timeit(lambda: hip.gru_gates_fwd(G, Xt, H, U, R, Ci), a.iters)
    report('gru_gates_fwd cin=16', us, n * 4 * (2 * h + h + h + h + h + 2 * h))
    dG, dX, dH = torch.empty_like(G), torch.empty_like(Xt), torch.empty_like(H)
    us = timeit(lambda: hip.gru_gates_bwd(Ci, U, H, U, R, dG, dX, dH), a.iters)
    report('gru_gates_bwd cin=16', us, n * 4 * (2 * h + h + h + h + h + 2 * h + h + h))
    us = timeit(lambda: hip.gru_blend_fwd(H, U, H, R, dH), a.iters)
    report('gru_blend_fwd', us, n * 4 * 5 * h)
    us = timeit(lambda: hip.gru_blend_bwd(H, U, H, R, dH, dX, Xt), a.iters)
    report('gru_blend_bwd', us, n * 4 * 7 * h)
    us = timeit(lambda: hip.concat2(Xt, H, Ci), a.iters)
    report('concat2 16+16', us, n * 4 * 4 * h)
    Xc = torch.empty_like(X)
    us = timeit(lambda: Xc.copy_(X), a.iters)
    report('torch copy (HBM reference)', us, 2 * X.numel() * 4)


if __name__ == '__main__':
    main()
