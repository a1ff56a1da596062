"""Probe: the few-category cell kernels on learned dense graphs BEYOND the staged size (``HipKernels.SMALL_STAGED_ROWS``): their split form
aggregates from global memory (MODE 3 of csrc/stc_cell_small.hip) and needs no staging.
    python tools/probes/small_dense_limit.py <rows limit> --graph-mode dense-learned --grid 14 --categories 8 --obs 9 --pred 3 --batch-per-gpu 32 ..."""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, 'stc-gnn_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
from stc_hip import _lib                                             # noqa: E402

_lib.HipKernels.SMALL_STAGED_ROWS = int(sys.argv[1])
sys.argv = [os.path.join(REPO, 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
