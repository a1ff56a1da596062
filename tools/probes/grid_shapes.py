"""The plain patch product and the two-ring sum on lattices of other shapes than the bench's 224 x 224 (the tile regions of graph._grid_tiles are chosen
"nearest 7 tiles wide", the width that suits that grid): python tools/probes/grid_shapes.py   -- per shape: planes of ~514 MB, us and TB/s."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'stc-gnn_amd')]
import torch          # noqa: E402
from stc_hip import CsrGraph          # noqa: E402
from stc_hip._lib import HipKernels          # noqa: E402
from stc_hip.graph import csr_operand          # noqa: E402

dev = torch.device('cuda')
hip = HipKernels()
C, h = 32, 16


def timed(fn, reps=12):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


for H, W in ((224, 224), (160, 320), (320, 160), (512, 96), (96, 512), (97, 113), (100, 100), (333, 151), (1000, 50)):
    n = H * W
    B = max(1, round(5 * 50176 / n))
    graph = CsrGraph.queen_grid(H, W, normalize=True)
    op = csr_operand(graph, dev)
    planes = [torch.randn(B, n, C, h, device=dev) for _ in range(6)]
    outs = [torch.empty(B, n, C, h, device=dev) for _ in range(4)]
    U, Cand = torch.sigmoid(planes[4]), torch.tanh(planes[5])
    fwd, bwd = (op.fwd_rowptr, op.fwd_colidx, op.fwd_val), (op.bwd_rowptr, op.bwd_colidx, op.bwd_val)
    mb = B * n * C * h * 4 / 1e6
    t_plain = timed(lambda i: hip.csr_spmm(*fwd, n, n, planes[i % 4].view(B, n, C * h), None, outs[i % 2].view(B, n, C * h), 1.0, 0.0, plan=op.fwd_plan))
    t_sum = timed(lambda i: hip.ring2_sum(*bwd, op.bwd_ring2, planes[i % 3], None, [planes[3]], U, Cand, outs[i % 2], outs[2 + i % 2])) if op.bwd_ring2 is not None else float('nan')
    print(f'{H:5d} x {W:4d}  B {B:3d}  plane {mb:6.1f} MB   plain {t_plain:7.1f} us {2 * mb / t_plain:5.2f} TB/s   two-ring sum {t_sum:7.1f} us {6 * mb / t_sum:5.2f} TB/s'
          f'   patches {op.fwd_plan[3][0].shape[0] if len(op.fwd_plan) > 3 else 0}')
    del planes, outs, U, Cand, op, graph
    torch.cuda.empty_cache()
