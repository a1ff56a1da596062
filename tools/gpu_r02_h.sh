#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_module_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "planar or cell_graph or bench_path" > gpurun_out/h_pytest.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/h_pytest.log
for fold in 0 1; do
STC_FOLD_DH=$fold timeout 600 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-unit-d3 > gpurun_out/h_bench_fold$fold.json 2>/dev/null
python3 - <<PY
import json
d=json.load(open('gpurun_out/h_bench_fold$fold.json'))
print('FOLD=$fold', round(d['value'],3), 'samples/s', round(d['ms_per_step'],2), 'ms; mem', round(d['hbm_peak_allocated_gb'],1), 'loss', d['loss'])
for k,v in d['kernels'].items(): print('   ', k, v['launches']//8, round(v['ms_per_step'],2), round(v.get('GBps',0)))
PY
done
STC_FOLD_DH=1 timeout 600 python3 bench.py --grid 100 --order 3 --batch-per-gpu 4 --steps 5 --warmup 2 --no-cpu-baseline --no-unit-d3 > gpurun_out/h_cfg4.json 2>/dev/null; python3 -c "import json;d=json.load(open('gpurun_out/h_cfg4.json'));print('cfg4 K=3', d['value'], d['ms_per_step'])"
