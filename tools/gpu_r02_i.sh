#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_module_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "order3 or cell_graph or bench_path" > gpurun_out/i_pytest.log 2>&1; echo "pytest exit $?"; tail -8 gpurun_out/i_pytest.log
for pk in 0 1; do
  STC_POST_K3=$pk timeout 600 python3 bench.py --grid 100 --order 3 --batch-per-gpu 4 --steps 5 --warmup 2 --no-cpu-baseline --no-unit-d3 > gpurun_out/i_cfg4_post$pk.json 2>gpurun_out/i_err$pk.log || tail -5 gpurun_out/i_err$pk.log
  python3 - <<PY
import json
d=json.load(open('gpurun_out/i_cfg4_post$pk.json'))
print('cfg4 K=3 POST_K3=$pk', round(d['value'],2), 'samples/s', round(d['ms_per_step'],2), 'ms, mem', round(d['hbm_peak_allocated_gb'],1), 'loss', d['loss'])
for k,v in d['kernels'].items(): print('   ', k, v['launches']//5, round(v['ms_per_step'],2), round(v.get('GBps',0)))
PY
done
STC_POST_K3=1 timeout 600 python3 bench.py --order 3 --batch-per-gpu 2 --steps 3 --warmup 1 --no-cpu-baseline --no-unit-d3 > gpurun_out/i_n50k_k3.json 2>/dev/null; python3 -c "import json;d=json.load(open('gpurun_out/i_n50k_k3.json'));print('N=50176 K=3', d['value'], d['ms_per_step'], d['hbm_peak_allocated_gb'])"
