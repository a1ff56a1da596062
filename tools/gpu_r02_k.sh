#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bf16_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/k_pytest_bf16.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/k_pytest_bf16.log
for c in 64 32; do
  b=5; [ $c = 32 ] && b=10
  timeout 600 python3 bench.py --storage bf16 --categories $c --batch-per-gpu $b --steps 4 --warmup 2 --no-cpu-baseline --no-unit-d3 > gpurun_out/k_bench_bf16_c$c.json 2>/dev/null
  python3 - <<PY
import json
d=json.load(open('gpurun_out/k_bench_bf16_c$c.json'))
print('bf16 C=$c', round(d['value'],2), 'samples/s', round(d['ms_per_step'],1), 'ms loss', d['loss'])
for k,v in list(d['kernels'].items())[:6]: print('   ', k, v['launches']//4, round(v['ms_per_step'],2))
PY
done
