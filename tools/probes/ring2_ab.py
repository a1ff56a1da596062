"""Same-process A/B of the two-ring launches inside the metric step: python tools/probes/ring2_ab.py <bwd 0|1> <fwd 0|1> [bench.py arguments]."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'stc-gnn_amd')]
from stc_hip import ops          # noqa: E402

ops._RING2, ops._RING2_FWD = sys.argv[1] == '1', sys.argv[2] == '1'
sys.argv = ['bench.py'] + sys.argv[3:]
import bench          # noqa: E402

bench.main()
