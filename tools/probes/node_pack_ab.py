"""Same-process-tree A/B of ``ops._NODE_PACK`` on a learned-graph step with few categories that the few-category cell kernels do not take
(dense learned Gs beyond the staged size: the general per-cell path), e.g. the reference's NYC shape (Main.py: 20 x 15 cells, C = 8):
    python tools/probes/node_pack_ab.py 0|1 --graph-mode dense-learned --grid 14 --categories 8 --obs 9 --pred 3 --batch-per-gpu 32 --steps 5 ...
runs bench.py with the switch off / on."""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, 'stc-gnn_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
from stc_hip import ops                                              # noqa: E402

ops._NODE_PACK = bool(int(sys.argv[1]))
sys.argv = [os.path.join(REPO, 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
