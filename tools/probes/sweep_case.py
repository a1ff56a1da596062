"""One case of tests/test_scale_sweep.py on the GPU, every tensor's error printed (max-norm, relative L2, the reference's own fp32 noise).

    python tools/probes/sweep_case.py c64 1e-3 small-params-zero-bias [f16x2|bf16x3]
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'stc-gnn_amd')]
import torch  # noqa: E402

from stc_hip import _lib, ops  # noqa: E402
from tests.conftest import rel_err  # noqa: E402
from tests.test_scale_sweep import _case, _oracle, _run, rel_l2  # noqa: E402


def main():
    family, x_scale, setting = sys.argv[1], float(sys.argv[2]), sys.argv[3]
    fmt = sys.argv[4] if len(sys.argv) > 4 else 'f16x2'
    k = ops.kernels()
    k.operand_format = {'f16x2': _lib.FMT_F16X2, 'bf16x3': _lib.FMT_BF16X3}[fmt]
    model, sd, s, X, Gs, (C, K) = _case(family, x_scale, setting)
    y, g = _run(model, s, X, Gs, 'cuda')
    y64, g64 = _oracle(sd, s, X, Gs, K, torch.float64)
    y32, g32 = _oracle(sd, s, X, Gs, K, torch.float32)
    print(f'{family} x{x_scale:g} {setting} {fmt}')
    print(f'  {"yhat":40s} max {rel_err(y, y64):.2e}  l2 {rel_l2(y, y64):.2e}  ref noise {rel_err(y32, y64):.1e}')
    for name in g64:
        print(f'  {name:40s} max {rel_err(g[name], g64[name]):.2e}  l2 {rel_l2(g[name], g64[name]):.2e}  ref noise {rel_err(g32[name], g64[name]):.1e}   |g| {float(g64[name].abs().max()):.2e}')


if __name__ == '__main__':
    main()
