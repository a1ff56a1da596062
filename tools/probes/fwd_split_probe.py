import os, sys, json, io, contextlib
REPO = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
for p in (REPO, os.path.join(REPO, 'stc-gnn_amd')):
    sys.path.insert(0, p)
import stc_hip._lib as L
L.HipKernels.SMALL_STAGED_ROWS = int(sys.argv[1])
sys.argv = ['bench.py', '--preset', 'sf', '--steps', '20', '--warmup', '3', '--no-cpu-baseline']
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().split('\n')[-1])
print(L.HipKernels.SMALL_STAGED_ROWS, round(d['ms_per_step'], 3), {k: (v['launches'], round(v['ms_per_step'], 3)) for k, v in d['kernels'].items() if 'small' in k})
