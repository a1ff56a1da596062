"""The two-ring launches alone at the metric shape (5 samples x 50 176 nodes, rows of 512 floats), operands rotated so that nothing is cache-resident:
python tools/probes/ring2_unit.py [reps]   -- microseconds per launch and the rate on the algorithmic bytes."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'stc-gnn_amd')]
import torch          # noqa: E402
from stc_hip import CsrGraph          # noqa: E402
from stc_hip._lib import HipKernels          # noqa: E402
from stc_hip.graph import csr_operand          # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda')
hip = HipKernels()
B, H, W, C, h = 5, 224, 224, 32, 16
n = H * W
op = csr_operand(CsrGraph.queen_grid(H, W, normalize=True), dev)
planes = [torch.randn(B, n, C, h, device=dev) for _ in range(12)]
outs = [torch.empty(B, n, C, h, device=dev) for _ in range(6)]
U, Cand = torch.sigmoid(planes[10]), torch.tanh(planes[11])
bwd, fwd = (op.bwd_rowptr, op.bwd_colidx, op.bwd_val), (op.fwd_rowptr, op.fwd_colidx, op.fwd_val)


def timed(fn):
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


plane_mb = B * n * C * h * 4 / 1e6
X, Y = planes[0].view(B, n, C * h), outs[0].view(B, n, C * h)
for name, n_planes, fn in (
        ('plain patch product', 2, lambda i: hip.csr_spmm(*fwd, n, n, planes[i % 4].view(B, n, C * h), None, outs[i % 3].view(B, n, C * h), 1.0, 0.0, plan=op.fwd_plan)),
        ('ring2_sum, 1 addend', 6, lambda i: hip.ring2_sum(*bwd, op.bwd_ring2, planes[i % 4], None, [planes[4 + i % 3]], U, Cand, outs[i % 3], outs[3 + i % 3])),
        ('ring2_sum, 2 addends', 7, lambda i: hip.ring2_sum(*bwd, op.bwd_ring2, planes[i % 4], None, [planes[4 + i % 3], planes[7 + i % 3]], U, Cand, outs[i % 3], outs[3 + i % 3])),
        ('ring2_blend', 7, lambda i: hip.ring2_blend(*fwd, op.fwd_ring2, planes[i % 4], planes[4 + i % 3], U, planes[7 + i % 3], outs[i % 2], outs[2 + i % 2], outs[4 + i % 2])),
        ('ring2_chain fwd (store V)', 4, lambda i: hip.ring2_chain(*fwd, op.fwd_ring2, planes[i % 4], None, 1.0, [], outs[i % 3], 2.0, [(planes[i % 4], -1.0)], outs[3 + i % 3]))):
    us = timed(fn)
    print(f'{name:28s} {us:7.1f} us   {n_planes * plane_mb / us:5.2f} TB/s on {n_planes} planes')
