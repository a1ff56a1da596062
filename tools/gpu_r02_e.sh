#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out && export TMPDIR=/tmp
rm -f gpurun_out/parity_errors.txt
echo "== full gpu suite"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/e_pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -6 gpurun_out/e_pytest_gpu.log
cp gpurun_out/parity_errors.txt gpurun_out/e_parity_errors.tsv 2>/dev/null
echo "== default bench"
timeout 900 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/e_bench.json 2> gpurun_out/e_bench.err; echo "bench exit $?"
python3 - <<PY
import json
d=json.load(open('gpurun_out/e_bench.json'))
print(round(d['value'],3), 'samples/s', round(d['ms_per_step'],2), 'ms; plain', round(d['roofline']['achieved']), 'GB/s; agg', round(d['roofline']['aggregate']['achieved']), 'd3', round(d['roofline']['unit_d3']['achieved']), 'loss', d['loss'])
for k,v in d['kernels'].items(): print('   ', k, v['launches']//10, round(v['ms_per_step'],2), round(v.get('GBps',0)))
PY
