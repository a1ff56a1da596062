"""Where the time of a small-graph cell launch goes: builds csrc/stc_cell_small.hip with -DSTC_PROBE -DSC_STOP_AFTER=n (the launch returns after
phase n) into scratch libraries, times each truncated launch with HIP events at the SF shape, and prints the phase durations as
differences.  Run on the GPU box:  python tools/probes/small_cell_phases.py [cin] [N] [C] [B]"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'stc-gnn_amd'))
from stc_hip import CsrGraph                                     # noqa: E402
from stc_hip.graph import csr_operand                            # noqa: E402

cin = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
Cc = int(sys.argv[3]) if len(sys.argv) > 3 else 5
B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
side = int(round(N ** 0.5))
assert side * side == N
csrc = os.path.join(REPO, 'stc-gnn_amd', 'csrc')
tmp = tempfile.mkdtemp()
dev = torch.device('cuda')
op = csr_operand(CsrGraph.queen_grid(side, side), dev)
L, LP = cin + 16, (32 if cin == 16 else 20)
r = lambda *s: torch.randn(*s, device=dev)
X, H, dHn = r(B, N, Cc, cin), r(B, N, Cc, 16), r(B, N, Cc, 16)
Tc = torch.stack([torch.eye(Cc, device=dev), torch.softmax(r(Cc, Cc), -1)])
Wg, bg, Wc, bc = 0.3 * r(4 * L, 32), 0.1 * r(32), 0.3 * r(4 * L, 16), 0.1 * r(16)
U, R, Cand, Hnew, RH, Zc = (torch.zeros(B, N, Cc, 16, device=dev) for _ in range(6))
Zg = torch.zeros(B, N * Cc, LP, device=dev)
dX, dH = torch.zeros_like(X), torch.zeros_like(H)
P = 4 * L * 48 + 48
dP = torch.zeros(B * 4, P, device=dev)
ws = torch.zeros(B * N * Cc * (2 * LP + 32), device=dev)
p = lambda t: C.c_void_p(t.data_ptr())


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


rows = []
extra = [a for a in sys.argv[5:] if a.startswith('-D')]          # e.g. -DSC_SKIP_ROLE=1
for stop in (1, 2, 3, 99):
    so = os.path.join(tmp, f'small_{stop}.so')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-DSTC_PROBE', f'-DSC_STOP_AFTER={stop}', *extra,
                           '-I' + os.path.join(REPO, 'include'), os.path.join(csrc, 'stc_cell_small.hip'), os.path.join(csrc, 'stc_gates.hip'), '-o', so])      # (stc_gates.hip: stc_last_error's buffer)
    lib = C.CDLL(so)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fwd = lambda: lib.stc_cell_small_fwd_f32(p(op.fwd_rowptr), p(op.fwd_colidx), p(op.fwd_val), N, op.fwd_val.numel(), 0, p(X), cin, p(H), p(Tc), 2, p(Wg), p(bg), p(Wc), p(bc),
                                             p(U), p(R), p(Cand), p(Hnew), p(RH), p(Zg), p(Zc), None, 0, 1, B, Cc, stream)
    bwd = lambda: lib.stc_cell_small_bwd_f32(p(op.bwd_rowptr), p(op.bwd_colidx), p(op.bwd_val), N, op.bwd_val.numel(), 0, p(X), cin, p(H), p(Tc), 2, p(Wg), p(Wc), p(U), p(R),
                                             p(Cand), p(RH), p(Zg), p(Zc), p(dHn), p(dX), 0, p(dH), 0, p(dP), C.c_int64(P), 1, 1, None, None, None, p(ws),
                                             C.c_size_t(ws.numel() * 4), 0, 1, B, Cc, stream)
    assert fwd() == 0 and bwd() == 0
    rows.append((stop, timed(fwd), timed(bwd)))
print(f'small cell launch, N={N} C={Cc} cin={cin} B={B} {" ".join(extra)}: cumulative us after each phase (forward | backward)')
prev = (0.0, 0.0)
for stop, f, b_ in rows:
    print(f'  phase {stop if stop < 99 else 4}: fwd {f:7.1f} (+{f - prev[0]:6.1f})   bwd {b_:7.1f} (+{b_ - prev[1]:6.1f})')
    prev = (f, b_)
