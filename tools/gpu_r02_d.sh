#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out && export TMPDIR=/tmp
echo "== order-3 kernel + module tests"
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_module_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "order3 or scales or cell_graph or bench_path or pieces" > gpurun_out/d_pytest_k3.log 2>&1; echo "pytest exit $?"; tail -12 gpurun_out/d_pytest_k3.log
for pk in 0 1; do
  STC_PLANAR_K3=$pk timeout 600 python3 bench.py --grid 100 --order 3 --batch-per-gpu 4 --steps 5 --warmup 2 --no-cpu-baseline --no-unit-d3 > gpurun_out/d_cfg4_k3_planar$pk.json 2>gpurun_out/d_cfg4_err$pk.log || tail -5 gpurun_out/d_cfg4_err$pk.log
  python3 - <<PY
import json
d=json.load(open('gpurun_out/d_cfg4_k3_planar$pk.json'))
print('cfg4 K=3 PLANAR_K3=$pk', round(d['value'],2), 'samples/s', round(d['ms_per_step'],2), 'ms, mem', round(d['hbm_peak_allocated_gb'],1), 'loss', d['loss'])
for k,v in d['kernels'].items(): print('   ', k, v['launches']//5, round(v['ms_per_step'],2), round(v.get('GBps',0)))
PY
done
