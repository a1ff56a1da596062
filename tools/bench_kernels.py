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
    """fn(i): the caller rotates over several buffer sets by i so that no launch finds its operands in the
    256 MiB Infinity Cache left there by the previous one (a re-used 205 MB tensor would read far too fast)."""
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fn(i)
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
    ap.add_argument('--no-tile', action='store_true', help='CSR SpMM kernel instead of the row-blocked one')
    a = ap.parse_args()
    hip = HipKernels()
    dev = torch.device('cuda')
    N, C, h, K, B = a.grid ** 2, a.C, a.h, a.K, a.B
    g = CsrGraph.queen_grid(a.grid, a.grid, permute_seed=1234 if a.permute else None, device=dev).on(dev)
    nnz = g['fwd_colidx'].numel()
    plan_f = (g['fwd_blk_ptr'], g['fwd_blk_cols'], g['fwd_blk_vals']) if not a.no_tile else None
    plan_b = (g['bwd_blk_ptr'], g['bwd_blk_cols'], g['bwd_blk_vals']) if not a.no_tile else None
    print(f'# N={N} nnz={nnz} C={C} h={h} K={K} B={B} permute={a.permute} '
          f"row_blocked={plan_f is not None}")

    def report(name, us, nbytes):
        print(f'{name:44s} {us:9.1f} us   {nbytes/1e6:9.1f} MB   {nbytes/us/1e3:8.1f} GB/s', flush=True)

    R = 4                                  # buffer sets rotated through (4 x >= 200 MB >> 256 MiB of Infinity Cache)
    if a.only == 'spmm-bf16':              # bf16 storage (configuration 5 runs this with --C 64): rows of C*L bf16
        for L in (32, 16):
            F = C * L
            Xs = [torch.randn(B, N, F, device=dev).bfloat16() for _ in range(R)]
            Ys = [torch.empty(B, N, F, device=dev, dtype=torch.bfloat16) for _ in range(R)]
            us = timeit(lambda i: hip.csr_spmm_bf16(g['fwd_rowptr'], g['fwd_colidx'], g['fwd_val'], N, N, Xs[i % R], None, Ys[i % R], 1.0, 0.0, plan=plan_f), a.iters)
            report(f'spmm bf16 fwd F={F}', us, nnz * 8 + 4 * (N + 1) + 2 * B * N * F * 2)
            us = timeit(lambda i: hip.csr_spmm_bf16(g['bwd_rowptr'], g['bwd_colidx'], g['bwd_val'], N, N, Xs[i % R], Ys[i % R], Ys[i % R], 1.0, 1.0, plan=plan_b), a.iters)
            report(f'spmm bf16 bwd F={F} (+=, in place)', us, nnz * 8 + 4 * (N + 1) + 3 * B * N * F * 2)
            del Xs, Ys
        bf = torch.bfloat16
        Tc = torch.softmax(torch.randn(K, C, C, device=dev), -1)
        Tc[0] = torch.eye(C, device=dev)
        L = 32
        for Ho in (2 * h, h):
            Zs = [[torch.randn(B * N, C, L, device=dev).to(bf) for _ in range(K)] for _ in range(R)]
            W = torch.randn(K * K * L, Ho, device=dev) * 0.1
            b = torch.randn(Ho, device=dev)
            Yn = [torch.empty(B * N, C, Ho, device=dev, dtype=bf) for _ in range(R)]
            us = timeit(lambda i: hip.bdg_node_fwd_bf16(Zs[i % R], Tc, W, b, Yn[i % R]), a.iters)
            flops = 2 * B * N * C * Ho * K * (K * L + C)            # projection + category mix (T_0 = I counted: it runs)
            report(f'node bf16 fwd L={L} Ho={Ho} ({flops / us / 1e6:.1f} TFLOP/s)', us, (K * L + Ho) * B * N * C * 2)
            dY = [torch.randn(B * N, C, Ho, device=dev).to(bf) for _ in range(R)]
            dZs = [[torch.empty_like(z) for z in zs] for zs in Zs]
            dW, db = torch.empty_like(W), torch.empty_like(b)
            us = timeit(lambda i: hip.bdg_node_bwd_bf16(Zs[i % R], Tc, W, dY[i % R], dZs[i % R], dW, db), a.iters)
            report(f'node bf16 bwd L={L} Ho={Ho}', us, (2 * K * L + Ho) * B * N * C * 2)
            del Zs, Yn, dY, dZs
        Xc = [torch.randn(B, N, C * 16, device=dev) for _ in range(R)]
        Yc = [torch.empty_like(x) for x in Xc]
        us = timeit(lambda i: Yc[i % R].copy_(Xc[i % R]), a.iters)
        report('torch copy (HBM reference)', us, 2 * Xc[0].numel() * 4)
        return
    for L in ((32, 20, 16) if a.only == 'spmm' else (32, 20)):
        F = C * L
        Xs = [torch.randn(B, N, F, device=dev) for _ in range(R)]
        Ys = [torch.empty(B, N, F, device=dev) for _ in range(R)]
        us = timeit(lambda i: hip.csr_spmm(g['fwd_rowptr'], g['fwd_colidx'], g['fwd_val'], N, N, Xs[i % R], None, Ys[i % R], 1.0, 0.0, plan=plan_f), a.iters)
        report(f'spmm fwd F={F}', us, nnz * 8 + 4 * (N + 1) + 2 * B * N * F * 4)
        us = timeit(lambda i: hip.csr_spmm(g['bwd_rowptr'], g['bwd_colidx'], g['bwd_val'], N, N, Xs[i % R], Ys[i % R], Ys[i % R], 1.0, 1.0, plan=plan_b), a.iters)
        report(f'spmm bwd F={F} (+=, in place)', us, nnz * 8 + 4 * (N + 1) + 3 * B * N * F * 4)
        if a.only == 'spmm':
            continue
        Tc = torch.softmax(torch.randn(K, C, C, device=dev), -1)
        Tc[0] = torch.eye(C, device=dev)
        for Ho in (2 * h, h):
            Zs = [[torch.randn(B * N, C, L, device=dev) for _ in range(K)] for _ in range(R)]
            W = torch.randn(K * K * L, Ho, device=dev) * 0.1
            b = torch.randn(Ho, device=dev)
            Yn = [torch.empty(B * N, C, Ho, device=dev) for _ in range(R)]
            us = timeit(lambda i: hip.bdg_node_fwd(Zs[i % R], Tc, W, b, Yn[i % R]), a.iters)
            report(f'node fwd L={L} Ho={Ho}', us, (K * L + Ho) * B * N * C * 4)
            dY = [torch.randn_like(y) for y in Yn]
            dZs = [[torch.empty_like(z) for z in zs] for zs in Zs]
            dW, db = torch.empty_like(W), torch.empty_like(b)
            us = timeit(lambda i: hip.bdg_node_bwd(Zs[i % R], Tc, W, dY[i % R], dZs[i % R], dW, db, None), a.iters)
            report(f'node bwd L={L} Ho={Ho}', us, (2 * K * L + Ho) * B * N * C * 4)
        del Xs, Ys, Zs, Yn, dY, dZs
    if a.only == 'spmm':
        return
    rows = (B, N, C)
    mk = lambda w: [torch.randn(*rows, w, device=dev) for _ in range(R)]
    G, Xt, H, U, Rg, Ci = mk(2 * h), mk(h), mk(h), mk(h), mk(h), mk(2 * h)
    n = B * N * C
    us = timeit(lambda i: hip.gru_gates_fwd(G[i % R], Xt[i % R], H[i % R], U[i % R], Rg[i % R], Ci[i % R]), a.iters)
    report('gru_gates_fwd cin=16', us, n * 4 * (2 * h + h + h + h + h + 2 * h))
    dG, dX, dH = mk(2 * h), mk(h), mk(h)
    us = timeit(lambda i: hip.gru_gates_bwd(Ci[i % R], U[i % R], H[i % R], U[i % R], Rg[i % R], dG[i % R], dX[i % R], dH[i % R]), a.iters)
    report('gru_gates_bwd cin=16', us, n * 4 * (2 * h + h + h + h + h + 2 * h + h + h))
    us = timeit(lambda i: hip.gru_blend_fwd(H[i % R], U[i % R], Xt[i % R], Rg[i % R], dH[i % R]), a.iters)
    report('gru_blend_fwd', us, n * 4 * 5 * h)
    us = timeit(lambda i: hip.gru_blend_bwd(H[i % R], U[i % R], Xt[i % R], Rg[i % R], dH[i % R], dX[i % R], G[i % R][..., :h].contiguous() if False else dG[i % R][..., :h].reshape(-1)[:n * h].view(*rows, h)), a.iters)
    report('gru_blend_bwd', us, n * 4 * 7 * h)
    us = timeit(lambda i: hip.concat2(Xt[i % R], H[i % R], Ci[i % R]), a.iters)
    report('concat2 16+16', us, n * 4 * 4 * h)
    us = timeit(lambda i: hip.split2(Ci[i % R], dX[i % R], dH[i % R], addA=dX[i % R], addB=dH[i % R]), a.iters)
    report('split2 16+16 (+=)', us, n * 4 * 6 * h)
    Hh = [torch.randn(B, 6, N, C, h, device=dev) for _ in range(2)]
    w, bb = torch.randn(h, device=dev), torch.randn(1, device=dev)
    yy = torch.empty(B, 6, N, C, device=dev)
    us = timeit(lambda i: hip.head_fwd(Hh[i % 2], w, bb, yy), a.iters)
    report('head fwd (6 steps)', us, 6 * n * 4 * (h + 1))
    dHh, dwb = torch.empty_like(Hh[0]), torch.empty(h + 1, device=dev)
    us = timeit(lambda i: hip.head_bwd(Hh[i % 2], w, yy, yy, dHh, dwb), a.iters)
    report('head bwd (6 steps)', us, 6 * n * 4 * (2 * h + 2))
    Xc = [torch.randn(B, N, C * 32, device=dev) for _ in range(R)]
    Yc = [torch.empty_like(x) for x in Xc]
    us = timeit(lambda i: Yc[i % R].copy_(Xc[i % R]), a.iters)
    report('torch copy (HBM reference)', us, 2 * Xc[0].numel() * 4)


if __name__ == '__main__':
    main()
