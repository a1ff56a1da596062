import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), 'stc-gnn_amd')); sys.path.insert(0, os.getcwd())
from stc_hip import ops
ops._SMALL = False
import bench
sys.argv = ['bench.py'] + sys.argv[1:]
bench.main()
