#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "bcsr_spmm or csr_spmm" > gpurun_out/g_pytest.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/g_pytest.log
timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-unit-d3 > gpurun_out/g_bench.json 2>/dev/null
python3 - <<PY
import json
d=json.load(open('gpurun_out/g_bench.json'))
print(round(d['value'],3), 'samples/s', round(d['ms_per_step'],2), 'ms; plain', round(d['roofline']['achieved']), 'GB/s; loss', d['loss'])
for k,v in d['kernels'].items(): print('   ', k, v['launches']//6, round(v['ms_per_step'],2), round(v.get('GBps',0)))
PY
